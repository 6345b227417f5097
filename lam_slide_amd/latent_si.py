"""``LatentSIV3``: drop-in for ``src.models.components.latent.latent_si_v31.LatentSIV3`` whose forward runs
the hand-written gfx950 kernels of liblamslide_hip.so.

Same constructor keywords (``configs/model/*/second-stage.yaml`` ``backbone:`` blocks), same parameter names
and shapes (so ``load_state_dict`` of a reference checkpoint / EMA state works, lightning_base.py:63-70),
same ``forward(x, t, x_cond, x_cond_mask, y=None)`` (latent_si_v31.py:168-170).  Inference only
(the sampling path runs under ``torch.no_grad``, lightning_base.py:217); there is no CPU implementation:
calling forward without the HIP library or with CPU tensors raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib
from .packing import PackedWeights, make_dims


class _Holder(nn.Module):
    """Parameter container: gives state_dict the reference's dotted names without any compute."""


def _linear_params(out_f: int, in_f: int) -> _Holder:
    m = _Holder()
    m.weight = nn.Parameter(torch.empty(out_f, in_f))
    m.bias = nn.Parameter(torch.empty(out_f))
    bound = 1.0 / math.sqrt(in_f)  # nn.Linear default: kaiming_uniform(a=sqrt 5) == U(-1/sqrt(in), 1/sqrt(in))
    nn.init.uniform_(m.weight, -bound, bound)
    nn.init.uniform_(m.bias, -bound, bound)
    return m


def _embedder(in_dim: int, hidden: int) -> _Holder:
    m = _Holder()
    m.in_layer = _linear_params(hidden, in_dim)
    m.out_layer = _linear_params(hidden, hidden)
    return m


def _scale_param(dim: int) -> _Holder:
    m = _Holder()
    m.scale = nn.Parameter(torch.ones(dim))
    return m


def _attn_mlp_params(hidden: int, heads: int, mlp_dim: int) -> _Holder:
    m = _Holder()
    m.linear1 = _linear_params(3 * hidden + mlp_dim, hidden)
    m.linear2 = _linear_params(hidden, hidden + mlp_dim)
    m.norm = _Holder()
    m.norm.query_norm = _scale_param(hidden // heads)
    m.norm.key_norm = _scale_param(hidden // heads)
    return m


def _layer_params(hidden: int, heads: int, mlp_dim: int) -> _Holder:
    m = _Holder()
    m.modulation = _Holder()
    m.modulation.lin = _linear_params(6 * hidden, hidden)
    m.spatial_block = _attn_mlp_params(hidden, heads, mlp_dim)
    m.temporal_block = _attn_mlp_params(hidden, heads, mlp_dim)
    return m


class LatentSIV3(nn.Module):
    def __init__(
        self,
        depth: int,
        in_dim: int,
        hidden_size: int,
        num_heads: int,
        vec_in_dim: Optional[int] = None,
        mlp_ratio: int = 2,
        n_timesteps: int = 10,
        theta: int = 10_000,
        checkpointing: bool = False,
        normalize: bool = False,
        attention_mode: str = "scaled_dot_product",
        share_weights: bool = False,
        reset_parameters: bool = True,
    ):
        super().__init__()
        from . import dropin
        dropin.install()  # no-op unless a reference checkout's sampler modules are already imported (dropin.py)
        self.in_dim = in_dim
        self.out_dim = in_dim
        self.n_timesteps = n_timesteps      # accepted and ignored, as in the reference
        self.checkpointing = checkpointing  # idem (inference path)
        self.normalize = normalize
        # mmdit.py:50-53: "scaled_dot_product" is softmax attention, ANY other string selects attention_linear (mmdit.py:58-72); every
        # shipped config uses the former (configs/model/*/second-stage.yaml)
        self.attention_mode = attention_mode
        self.dims = make_dims(depth, in_dim, hidden_size, num_heads, mlp_ratio, vec_in_dim, normalize, theta)
        self.depth, self.share_weights = depth, share_weights
        mlp_dim = self.dims.mlp_dim

        self.x_in = _linear_params(hidden_size, in_dim)
        self.cond_to_emb = _linear_params(hidden_size, in_dim)
        self.mask_to_emb = _Holder()
        self.mask_to_emb.weight = nn.Parameter(torch.randn(2, hidden_size))
        self.time_in = _embedder(256, hidden_size)
        if vec_in_dim is not None:
            self.vec_in = _embedder(vec_in_dim, hidden_size)
        self.blocks = nn.ModuleList()
        if share_weights:
            layer = _layer_params(hidden_size, num_heads, mlp_dim)
            for _ in range(depth):
                self.blocks.append(layer)
        else:
            for _ in range(depth):
                self.blocks.append(_layer_params(hidden_size, num_heads, mlp_dim))
        self.adaLN_modulation = nn.ModuleList([_Holder(), _linear_params(2 * hidden_size, hidden_size)])  # keys "adaLN_modulation.1.*"
        self.linear = _linear_params(in_dim, hidden_size)
        if reset_parameters:
            self.reset_parameters()

        self._packed: Optional[PackedWeights] = None
        self._packed_key = None
        self._handle = C.c_void_p()
        self._workspaces: "OrderedDict[tuple, Tensor]" = OrderedDict()
        self._pinned: "OrderedDict[tuple, Tensor]" = OrderedDict()
        self._chunk = 0
        self._tail = None  # None: the library's default (LSL_TAIL); True / False: set_tail
        self._ln_fuse = None  # likewise (LSL_LN_FUSE / set_ln_fuse)
        self.last_path = None  # "hip" after a forward, for tests that must prove the native path ran

    # ---- initialisation recipe of the reference (latent_si_v31.py:123-156) ----------------------------
    def reset_parameters(self):
        def xavier(lin, gain):
            nn.init.xavier_uniform_(lin.weight, gain=gain)
            nn.init.constant_(lin.bias, 0)

        g = 1.0 / math.sqrt(2)
        lins = [self.x_in, self.cond_to_emb, self.time_in.in_layer, self.time_in.out_layer, self.adaLN_modulation[1], self.linear]
        if hasattr(self, "vec_in"):
            lins += [self.vec_in.in_layer, self.vec_in.out_layer]
        for blk in self.blocks:
            lins += [blk.modulation.lin, blk.spatial_block.linear1, blk.spatial_block.linear2, blk.temporal_block.linear1,
                     blk.temporal_block.linear2]
        for lin in lins:
            xavier(lin, g)
        for emb in [self.time_in] + ([self.vec_in] if hasattr(self, "vec_in") else []):
            nn.init.normal_(emb.in_layer.weight, std=0.02)
            nn.init.normal_(emb.out_layer.weight, std=0.02)
        for blk in self.blocks:
            nn.init.constant_(blk.modulation.lin.weight, 0.0)
            nn.init.constant_(blk.modulation.lin.bias, 0.0)
        nn.init.constant_(self.linear.weight, 0.0)
        nn.init.constant_(self.linear.bias, 0.0)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Also accepts the keys of a backbone that was saved while wrapped by ``torch.compile`` (``_orig_mod.`` prefix; the
        reference compiles its backbone when ``compile: True``, second_stage/md17.py:53-55)."""
        from .packing import strip_prefixes
        return super().load_state_dict(strip_prefixes(state_dict), strict=strict, assign=assign)

    # ---- native handle ---------------------------------------------------------------------------------
    def _weights_key(self, device):
        return (str(device), self.attention_mode) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def ensure_packed(self, device) -> PackedWeights:
        key = self._weights_key(device)
        if self._packed is None or key != self._packed_key:
            lib = _lib.load()
            self._packed = PackedWeights(self.state_dict(), self.dims, device)
            self._packed_key = key
            if self._handle:
                lib.lsl_model_destroy(self._handle)
                self._handle = C.c_void_p()
            desc = self._packed.desc()
            _lib.check(lib.lsl_model_create(C.byref(desc), C.byref(self._handle)))
            _lib.check(lib.lsl_model_set_weights(self._handle, C.byref(self._packed.c_weights)))
            if self._chunk:
                lib.lsl_model_set_chunk(self._handle, self._chunk)
            if self._tail is not None:
                _lib.check(lib.lsl_model_set_tail(self._handle, int(self._tail)))
            if self._ln_fuse is not None:
                _lib.check(lib.lsl_model_set_ln_fuse(self._handle, int(self._ln_fuse)))
            if self.attention_mode != "scaled_dot_product":
                _lib.check(lib.lsl_model_set_attention_mode(self._handle, 1))
        return self._packed

    def set_tail(self, on: bool):
        """Decomposition of every sub-block behind the attention (include/lsl_api.h, ``lsl_model_set_tail``): ``True`` = linear1 computes
        q | k | v only and one row-owning kernel runs mlp up-projection -> GELU -> linear2 -> gated residual -> next LayerNorm (fewer HBM
        bytes; faster from about 10^5 tokens per pass, slower below).  A property of the model object, never of the batch: a trajectory's
        bits are the same in any batch.  Raises ValueError if the model has no instance (hidden 256, heads * head_dim_pad = 256)."""
        self._tail = bool(on)
        if self._handle:
            _lib.check(_lib.load().lsl_model_set_tail(self._handle, int(self._tail)))

    def set_ln_fuse(self, on: bool):
        """LayerNorm + modulate inside linear1's activation load (include/lsl_api.h, ``lsl_model_set_ln_fuse``): no LayerNorm launch, linear1
        reads the fp32 residual stream.  Faster from about 10^5 tokens per pass, slower at small launches; one more bf16 rounding than the
        standalone kernel.  A property of the model object, never of the batch."""
        self._ln_fuse = bool(on)
        if self._handle:
            _lib.check(_lib.load().lsl_model_set_ln_fuse(self._handle, int(self._ln_fuse)))

    @property
    def ln_fuse(self) -> bool:
        """Whether the native handle fuses the LayerNorm into linear1 (False before the weights are packed)."""
        return bool(self._handle) and bool(_lib.load().lsl_model_ln_fuse(self._handle))

    @property
    def tail(self) -> bool:
        """Whether the native handle runs the tail form (False before the weights are packed)."""
        return bool(self._handle) and bool(_lib.load().lsl_model_tail(self._handle))

    def set_chunk(self, trajectories_per_pass: int):
        """Cache-residency knob: trajectories processed per pass through the layers (0 = library default)."""
        self._chunk = int(trajectories_per_pass)
        if self._handle:
            _lib.load().lsl_model_set_chunk(self._handle, self._chunk)

    def workspace(self, B: int, T: int, L: int, device) -> Tensor:
        """Scratch of one call.  One buffer per (device, stream): two sampling calls of one model in flight on different streams
        never share scratch (calls on ONE stream are ordered by the stream).  At most 4 buffers are kept."""
        device = torch.device(device)
        need = _lib.load().lsl_workspace_bytes(self._handle, B, T, L)
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        ws = self._workspaces.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._workspaces[key] = ws
        self._workspaces.move_to_end(key)
        while len(self._workspaces) > 4:
            self._workspaces.popitem(last=False)
        return ws

    @staticmethod
    def graph_replay_enabled(tokens: Optional[int] = None) -> bool:
        """Whether the library may replay a captured hipGraph for a call of ``tokens`` (= B T L) tokens: LSL_GRAPH=1 (the default) for
        launch-bound calls (at most 64 Ki tokens, the library's own rule in lsl_sample_ex), 2 for every call, 0 never."""
        mode = os.environ.get("LSL_GRAPH", "1")
        if mode in ("", "0"):
            return False
        return mode != "1" or tokens is None or tokens <= 65536

    def staged(self, tag: str, src: Tensor, dtype: torch.dtype, device, fresh: bool = False, persistent: bool = False) -> Tensor:
        """``src`` as a contiguous ``dtype`` tensor on ``device`` for the library.  Without ``persistent`` nothing persists: ``fresh`` (the
        state, which the sampler updates in place) gets a private copy, everything else is passed through when it already has the right
        layout.  ``persistent`` (calls the library may replay as a hipGraph: ``graph_replay_enabled``): the copy goes into a persistent
        buffer per (tag, shape, device), so that repeated calls hand the library the SAME pointers and ``lsl_sample`` can replay its
        captured graph; that cache is keyed per stream and LRU-bounded (16 buffers): one sampling call at a time per model object and stream."""
        device = torch.device(device)
        if not persistent:
            out = src.detach().to(device=device, dtype=dtype).contiguous()
            if fresh and out.data_ptr() == src.data_ptr():
                out = out.clone()
            return out
        key = (tag, tuple(src.shape), device, torch.cuda.current_stream(device).cuda_stream)  # (per stream, like the workspace)
        buf = self._pinned.get(key)
        if buf is None:
            buf = torch.empty(src.shape, dtype=dtype, device=device)
            self._pinned[key] = buf
        self._pinned.move_to_end(key)
        while len(self._pinned) > 16:
            self._pinned.popitem(last=False)
        buf.copy_(src)
        return buf

    def __del__(self):
        try:
            if self._handle:
                _lib.load().lsl_model_destroy(self._handle)
        except Exception:
            pass

    @staticmethod
    def _require_gpu(x: Tensor):
        if not x.is_cuda:
            raise RuntimeError("lam_slide_amd.LatentSIV3 runs only on an AMD GPU (HIP kernels); got a CPU tensor and there "
                               "is no CPU fallback")

    def make_io(self, x: Tensor, x_cond: Tensor, x_cond_mask: Tensor, y: Optional[Tensor], t: Optional[Tensor] = None,
                out: Optional[Tensor] = None):
        """Argument block of one library call.  Everything handed over as a pointer is checked here: a tensor on another device (or
        on the host) would otherwise reach the kernels as a foreign address and end in a GPU memory fault instead of the reference's
        device-mismatch error; ``t`` must cover the batch (the kernels read t[0..B))."""
        if x.dim() != 4:
            raise ValueError(f"x must be [B, T, L, C], got {tuple(x.shape)}")
        B, T, L, Cc = x.shape
        if Cc != self.in_dim or x_cond.shape != x.shape or tuple(x_cond_mask.shape) != (B, T, L):
            raise ValueError(f"shape mismatch: x {tuple(x.shape)}, x_cond {tuple(x_cond.shape)}, mask {tuple(x_cond_mask.shape)}")
        if y is not None and (not hasattr(self, "vec_in") or tuple(y.shape) != (B, self.dims.vec_in_dim)):
            raise ValueError("y given but the model has no vec_in, or y has the wrong shape")
        for name, ten in (("x_cond", x_cond), ("x_cond_mask", x_cond_mask), ("y", y), ("t", t), ("out", out)):
            if ten is not None and ten.device != x.device:
                raise RuntimeError(f"Expected all tensors to be on the same device, but {name} is on {ten.device} and x is on {x.device}")
        if not x.is_contiguous() or x.dtype != torch.float32:
            raise ValueError("x must be a contiguous float32 tensor")
        if t is not None and (tuple(t.shape) != (B,) or t.dtype != torch.float32 or not t.is_contiguous()):
            raise ValueError(f"t must be a contiguous float32 tensor of shape ({B},), got {tuple(t.shape)} {t.dtype}")
        if out is not None and (out.shape != x.shape or out.dtype != torch.float32 or not out.is_contiguous()):
            raise ValueError("out must be a contiguous float32 tensor shaped like x")
        keep = [x_cond.float().contiguous(), x_cond_mask.to(torch.int64).contiguous()]
        if y is not None:
            keep.append(y.float().contiguous())
        io = _lib.IO(x.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr() if y is not None else None,
                     t.data_ptr() if t is not None else None, out.data_ptr() if out is not None else None, B, T, L)
        return io, keep

    @torch.no_grad()
    def forward(self, x: Tensor, t: Tensor, x_cond: Tensor, x_cond_mask: Tensor, y: Tensor = None) -> Tensor:
        self._require_gpu(x)
        lib = _lib.load()
        if x.dim() != 4:
            raise ValueError(f"x must be [B, T, L, C], got {tuple(x.shape)}")
        if t.device != x.device:
            raise RuntimeError(f"Expected all tensors to be on the same device, but t is on {t.device} and x is on {x.device}")
        B = x.shape[0]
        tt = t.detach().float()
        if tt.dim() == 0 or tuple(tt.shape) == (1,):  # the reference broadcasts a scalar time over the batch (mmdit.py:107)
            tt = tt.reshape(1).expand(B)
        if tuple(tt.shape) != (B,):
            raise ValueError(f"t must have shape ({B},) (or be a scalar), got {tuple(t.shape)}")
        tt = tt.contiguous()
        with torch.cuda.device(x.device):
            self.ensure_packed(x.device)
            xin = x.detach().float().contiguous()
            out = torch.empty_like(xin)
            io, keep = self.make_io(xin, x_cond, x_cond_mask, y, tt, out)
            ws = self.workspace(io.B, io.T, io.L, x.device)
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _lib.check(lib.lsl_forward(self._handle, C.byref(io), ws.data_ptr(), ws.numel(), stream))
        self.last_path = "hip"
        del keep
        return out.to(x.dtype)
