"""Frozen stage-1 encode on the MI355X: ``quant(Encoder(x, entities, mask))`` of the reference
(models/composites/lightning_base.py:37-40, models/components/encoder.py:34-41,96-103), the step before
``setup_conditioning``.  ``x`` is the output of the dataset-specific ``prepare_inputs``
(first_stage/md17.py:52-58 and siblings), which stays the caller's.  Inference only, fp32, through ``lsl_encode``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch
from torch import Tensor

from . import _lib
from .decoder import _ACT, renorm_table


class Stage1Encoder:
    def __init__(self, state_dict: Dict[str, Tensor], *, num_head_cross: int, dim_head_cross: int, num_head_latent: int,
                 dim_head_latent: int, act: str = "gelu_erf", max_norm: Optional[float] = 1.0, device: Optional[torch.device] = None):
        if act not in _ACT:
            raise ValueError(f"unknown activation {act!r} (gelu_erf | gelu_tanh)")
        sd = {k.replace("_orig_mod.", ""): v for k, v in state_dict.items()}
        for k in ("quant.0.weight", "quant.0.bias", "encoder.latents", "encoder.mlp.0.weight", "encoder.mlp.2.weight",
                  "encoder.entity_embedding.embedding.weight"):
            if k not in sd:
                raise KeyError(k)
        self.num_block_cross = len({k.split(".")[2] for k in sd if k.startswith("encoder.cross_attn_blocks.")})
        self.num_block_attn = len({k.split(".")[2] for k in sd if k.startswith("encoder.blocks_attn.")})
        self.qk_norm = any(k.endswith("attn.fn.norm.query_norm.scale") for k in sd if k.startswith("encoder."))
        self.num_latents, self.dim_latent = sd["encoder.latents"].shape
        self.n_entities, self.dim_emb = sd["encoder.entity_embedding.embedding.weight"].shape
        self.dim_input = sd["encoder.mlp.0.weight"].shape[1] - self.dim_emb
        self.heads_cross, self.dim_head_cross = int(num_head_cross), int(dim_head_cross)
        self.heads_latent, self.dim_head_latent = int(num_head_latent), int(dim_head_latent)
        if self.num_block_cross and sd["encoder.cross_attn_blocks.0.attn.fn.to_q.weight"].shape[0] != self.heads_cross * self.dim_head_cross:
            raise ValueError("num_head_cross * dim_head_cross does not match encoder.cross_attn_blocks.0.attn.fn.to_q.weight")
        if self.num_block_attn and sd["encoder.blocks_attn.0.attn.fn.to_qkv.weight"].shape[0] != 3 * self.heads_latent * self.dim_head_latent:
            raise ValueError("num_head_latent * dim_head_latent does not match encoder.blocks_attn.0.attn.fn.to_qkv.weight")
        self.act = act
        self._sd = {k: v.detach().to(torch.float32) for k, v in sd.items() if k.startswith(("quant.", "encoder."))}
        self._sd["encoder.entity_embedding.embedding.weight"] = renorm_table(self._sd["encoder.entity_embedding.embedding.weight"], max_norm)
        self._handle = C.c_void_p()
        self._dev_tensors: List[Tensor] = []
        self._ws: Optional[Tensor] = None
        if device is not None:
            self.to(device)

    def _p(self, key: str, dev) -> Optional[int]:
        if key not in self._sd:
            return None
        t = self._sd[key].to(dev).contiguous()
        self._dev_tensors.append(t)
        return t.data_ptr()

    def _block(self, prefix: str, dev, cross: bool) -> "_lib.DecBlock":
        g = lambda name: self._p(f"{prefix}.{name}", dev)  # noqa: E731
        return _lib.DecBlock(
            ln_w=g("attn.norm.weight"), ln_b=g("attn.norm.bias"),
            lnc_w=g("attn.norm_context.weight") if cross else None, lnc_b=g("attn.norm_context.bias") if cross else None,
            w_q=g("attn.fn.to_q.weight") if cross else g("attn.fn.to_qkv.weight"), w_kv=g("attn.fn.to_kv.weight") if cross else None,
            w_out=g("attn.fn.to_out.weight"), b_out=g("attn.fn.to_out.bias"),
            q_scale=g("attn.fn.norm.query_norm.scale") if self.qk_norm else None, k_scale=g("attn.fn.norm.key_norm.scale") if self.qk_norm else None,
            ff_ln_w=g("ff.norm.weight"), ff_ln_b=g("ff.norm.bias"), ff_w1=g("ff.fn.net.0.0.weight"), ff_b1=g("ff.fn.net.0.0.bias"),
            ff_w2=g("ff.fn.net.1.weight"), ff_b2=g("ff.fn.net.1.bias"))

    def to(self, device) -> "Stage1Encoder":
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("Stage1Encoder runs on the MI355X only (no CPU path)")
        lib = _lib.load()
        if self._handle:
            lib.lsl_encoder_destroy(self._handle)
            self._handle = C.c_void_p()
        self._dev_tensors = []
        cross = (_lib.DecBlock * max(self.num_block_cross, 1))(*[self._block(f"encoder.cross_attn_blocks.{i}", dev, True) for i in range(self.num_block_cross)])
        selfs = (_lib.DecBlock * max(self.num_block_attn, 1))(*[self._block(f"encoder.blocks_attn.{i}", dev, False) for i in range(self.num_block_attn)])
        w = _lib.EncoderWeights(
            table=self._p("encoder.entity_embedding.embedding.weight", dev),
            mlp_w1=self._p("encoder.mlp.0.weight", dev), mlp_b1=self._p("encoder.mlp.0.bias", dev),
            mlp_w2=self._p("encoder.mlp.2.weight", dev), mlp_b2=self._p("encoder.mlp.2.bias", dev),
            latents=self._p("encoder.latents", dev), cross_blocks=cross, self_blocks=selfs,
            quant_w=self._p("quant.0.weight", dev), quant_b=self._p("quant.0.bias", dev))
        desc = _lib.EncoderDesc(self.dim_input, self.dim_emb, self.n_entities, self.dim_latent, self.num_latents, self.heads_cross,
                                self.dim_head_cross, self.heads_latent, self.dim_head_latent, self.num_block_cross, self.num_block_attn, _ACT[self.act])
        _lib.check(lib.lsl_encoder_create(C.byref(desc), C.byref(w), C.byref(self._handle)))
        self.device = dev
        return self

    def __del__(self):
        try:
            if self._handle:
                _lib.load().lsl_encoder_destroy(self._handle)
        except Exception:
            pass

    @torch.no_grad()
    def encode(self, x: Tensor, entities: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        """x [F, A, dim_input] fp32, entities [F, A] integer, mask [F, A] bool (True = real entity) -> latents [F, num_latents, dim_latent]."""
        if not x.is_cuda:
            raise RuntimeError("Stage1Encoder.encode needs CUDA/HIP tensors (no CPU path)")
        if not self._handle or self.device != x.device:
            self.to(x.device)
        if x.dim() != 3 or x.shape[-1] != self.dim_input or tuple(entities.shape) != tuple(x.shape[:2]):
            raise ValueError("expected x [F, A, dim_input] and entities [F, A]")
        if mask is not None and tuple(mask.shape) != tuple(x.shape[:2]):
            raise ValueError("expected mask [F, A]")
        lib = _lib.load()
        F_, A, _ = x.shape
        xx = x.contiguous().float()
        ent = entities.contiguous().to(torch.int64)
        mk = mask.contiguous().to(torch.uint8) if mask is not None else None
        need = lib.lsl_encode_workspace_bytes(self._handle, F_, A)
        if self._ws is None or self._ws.numel() < need or self._ws.device != xx.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=xx.device)
        out = torch.empty(F_, self.num_latents, self.dim_latent, dtype=torch.float32, device=xx.device)
        _lib.check(lib.lsl_encode(self._handle, xx.data_ptr(), ent.data_ptr(), mk.data_ptr() if mk is not None else None, F_, A, out.data_ptr(),
                                  self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream(xx.device).cuda_stream))
        return out

    __call__ = encode
