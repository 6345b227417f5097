"""Frozen stage-1 decode on the MI355X: ``first_stage.decode(latents, entities)`` of the reference
(models/composites/lightning_base.py:42-44 = ``Decoder(post_quant(latents), entities)``, models/components/decoder.py:12-102),
the step right after the sampler.  Inference only, fp32, through ``lsl_decode`` of liblamslide_hip.so.  The peptide
variant ``DecoderQuerySplitter`` (decoder.py:313-411) is recognised by its ``decoder.extender.1.*`` parameters.

The weights are taken from the first-stage state dict under the reference's own names (``post_quant.1.*``, ``decoder.*``); the
constructor arguments that cannot be read off the weight shapes carry the reference's names (decoder.py:14-29).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch
from torch import Tensor

from . import _lib

_ACT = {"gelu_erf": 1, "gelu": 1, "gelu_tanh": 2}


def renorm_table(table: Tensor, max_norm: Optional[float]) -> Tensor:
    """nn.Embedding(max_norm=...) rescales every looked-up row whose L2 norm exceeds max_norm, in place, at forward time
    (entity_embeddings.py:25; torch embedding_renorm_: scale = max_norm / (norm + 1e-7)).  Applied to the whole table once."""
    if max_norm is None:
        return table.clone()
    norms = table.norm(dim=-1, keepdim=True)
    return torch.where(norms > max_norm, table * (max_norm / (norms + 1e-7)), table)


class Stage1Decoder:
    def __init__(self, state_dict: Dict[str, Tensor], *, num_head_latent: int, dim_head_latent: int, num_head_cross: int,
                 dim_head_cross: int, act: str = "gelu_erf", output: str = "pos", max_norm: Optional[float] = 1.0,
                 device: Optional[torch.device] = None):
        if act not in _ACT:
            raise ValueError(f"unknown activation {act!r} (gelu_erf | gelu_tanh)")
        sd = {k.replace("_orig_mod.", ""): v for k, v in state_dict.items()}
        need = ["post_quant.1.weight", "post_quant.1.bias", "decoder.entity_embedding.embedding.weight", "decoder.query_mlp.1.weight",
                f"decoder.output_layers.{output}.0.weight", f"decoder.output_layers.{output}.2.weight"]
        for k in need:
            if k not in sd:
                raise KeyError(k)
        self.num_block_attn = len({k.split(".")[2] for k in sd if k.startswith("decoder.self_attn_blocks.")})
        self.num_block_cross = len({k.split(".")[2] for k in sd if k.startswith("decoder.cross_attn_blocks.")})
        self.qk_norm = "decoder.output_block.attn.fn.norm.query_norm.scale" in sd
        self.dim_latent, self.in_dim = sd["post_quant.1.weight"].shape
        self.n_entities, self.dim_emb = sd["decoder.entity_embedding.embedding.weight"].shape
        self.dim_query = sd["decoder.query_mlp.1.weight"].shape[0]
        self.out_dim = sd[f"decoder.output_layers.{output}.2.weight"].shape[0]
        self.heads_latent, self.dim_head_latent = int(num_head_latent), int(dim_head_latent)
        self.heads_cross, self.dim_head_cross = int(num_head_cross), int(dim_head_cross)
        inner_c = self.heads_cross * self.dim_head_cross
        if tuple(sd["decoder.output_block.attn.fn.to_q.weight"].shape) != (inner_c, self.dim_query):
            raise ValueError("num_head_cross * dim_head_cross does not match decoder.output_block.attn.fn.to_q.weight")
        if self.num_block_attn and sd["decoder.self_attn_blocks.0.attn.fn.to_qkv.weight"].shape[0] != 3 * self.heads_latent * self.dim_head_latent:
            raise ValueError("num_head_latent * dim_head_latent does not match decoder.self_attn_blocks.0.attn.fn.to_qkv.weight")
        self.act, self.output, self.max_norm = act, output, max_norm
        self._sd = {k: v.detach().to(torch.float32) for k, v in sd.items() if k.startswith(("post_quant.", "decoder."))}
        # DecoderQuerySplitter (decoder.py:384-388): Conv1d(D, D*N, 1) then "B (D N) L -> B (L N) D"; as a Linear whose rows are
        # reordered from channel d*N + n to n*D + d, the output of one latent IS its N context tokens back to back
        self.num_split = 0
        if "decoder.extender.1.weight" in self._sd:
            w, b = self._sd.pop("decoder.extender.1.weight")[..., 0], self._sd.pop("decoder.extender.1.bias")
            n = w.shape[0] // self.dim_latent
            self.num_split = int(n)
            self._sd["decoder.extender.rows"] = w.reshape(self.dim_latent, n, self.dim_latent).permute(1, 0, 2).reshape(n * self.dim_latent, self.dim_latent).contiguous()
            self._sd["decoder.extender.rows_bias"] = b.reshape(self.dim_latent, n).t().reshape(-1).contiguous()
        self._sd["decoder.entity_embedding.embedding.weight"] = renorm_table(self._sd["decoder.entity_embedding.embedding.weight"], max_norm)
        self._handle = C.c_void_p()
        self._dev_tensors: List[Tensor] = []
        self._ws: Optional[Tensor] = None
        if device is not None:
            self.to(device)

    # -- packing ---------------------------------------------------------------------------------------------
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

    def to(self, device) -> "Stage1Decoder":
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("Stage1Decoder runs on the MI355X only (no CPU path)")
        lib = _lib.load()
        if self._handle:
            lib.lsl_decoder_destroy(self._handle)
            self._handle = C.c_void_p()
        self._dev_tensors = []
        selfs = (_lib.DecBlock * max(self.num_block_attn, 1))(*[self._block(f"decoder.self_attn_blocks.{i}", dev, False) for i in range(self.num_block_attn)])
        cross = (_lib.DecBlock * max(self.num_block_cross, 1))(*[self._block(f"decoder.cross_attn_blocks.{i}", dev, True) for i in range(self.num_block_cross)])
        w = _lib.DecoderWeights(
            pq_w=self._p("post_quant.1.weight", dev), pq_b=self._p("post_quant.1.bias", dev),
            table=self._p("decoder.entity_embedding.embedding.weight", dev),
            qm_w=self._p("decoder.query_mlp.1.weight", dev), qm_b=self._p("decoder.query_mlp.1.bias", dev),
            self_blocks=selfs, cross_blocks=cross, out_block=self._block("decoder.output_block", dev, True),
            ext_w=self._p("decoder.extender.rows", dev), ext_b=self._p("decoder.extender.rows_bias", dev),
            head_w1=self._p(f"decoder.output_layers.{self.output}.0.weight", dev), head_b1=self._p(f"decoder.output_layers.{self.output}.0.bias", dev),
            head_w2=self._p(f"decoder.output_layers.{self.output}.2.weight", dev), head_b2=self._p(f"decoder.output_layers.{self.output}.2.bias", dev))
        desc = _lib.DecoderDesc(self.in_dim, self.dim_latent, self.dim_query, self.dim_emb, self.n_entities, self.heads_latent, self.dim_head_latent,
                                self.heads_cross, self.dim_head_cross, self.num_block_attn, self.num_block_cross, _ACT[self.act], self.out_dim, self.num_split)
        _lib.check(lib.lsl_decoder_create(C.byref(desc), C.byref(w), C.byref(self._handle)))
        self.device = dev
        return self

    def __del__(self):
        try:
            if self._handle:
                _lib.load().lsl_decoder_destroy(self._handle)
        except Exception:
            pass

    # -- decode ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, latents: Tensor, entities: Tensor) -> Tensor:
        """latents [F, L, C] fp32, entities [F, A] integer -> [F, A, out_dim] (the reference returns ``{"pos": ...}``;
        this is the tensor of the one head selected by ``output``)."""
        if not latents.is_cuda:
            raise RuntimeError("Stage1Decoder.decode needs CUDA/HIP tensors (no CPU path)")
        if not self._handle or self.device != latents.device:
            self.to(latents.device)
        if latents.dim() != 3 or latents.shape[-1] != self.in_dim or entities.dim() != 2 or entities.shape[0] != latents.shape[0]:
            raise ValueError("expected latents [F, L, C] and entities [F, A]")
        lib = _lib.load()
        F_, L, _ = latents.shape
        A = entities.shape[1]
        z = latents.contiguous().float()
        ent = entities.contiguous().to(torch.int64)
        need = lib.lsl_decode_workspace_bytes(self._handle, F_, L, A)
        if self._ws is None or self._ws.numel() < need or self._ws.device != z.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=z.device)
        out = torch.empty(F_, A, self.out_dim, dtype=torch.float32, device=z.device)
        _lib.check(lib.lsl_decode(self._handle, z.data_ptr(), ent.data_ptr(), F_, L, A, out.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                  torch.cuda.current_stream(z.device).cuda_stream))
        return out

    __call__ = decode
