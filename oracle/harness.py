"""CPU restatement of the caller-side harness around the sampling loop, and of the frozen
stage-1 decode that follows it (the "decoded-coordinate" half of the parity metric).

Test infrastructure only (see ``oracle/__init__.py``).

Reference lines restated (under /root/reference/src/):
  models/composites/lightning_base.py:240-263   setup_conditioning (cannot be imported here: needs lightning)
  models/composites/lightning_base.py:217-238   sample(): noise -> sample_fn(...)[-1] -> decode
  models/composites/lightning_base.py:28-31,42-44  post_quant (LayerNorm without affine, then Linear) + decoder
  models/components/decoder.py:12-102           Decoder: queries -> self-attn blocks -> cross blocks -> output block -> heads
  modules/torch_modules.py:104-147              PreNorm / FeedForward
  modules/torch_modules.py:150-264              Attention / SelfAttention / blocks (SDPA, optional QK RMS norm)
  modules/entity_embeddings.py:7-33             entity embedding with max_norm (renormalised rows)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import latent_net, transport


def setup_conditioning(latents: Tensor, cond_idx: Tuple[int, int], mask_cond_mean: bool) -> Tuple[Tensor, Tensor]:
    B, T, L, _ = latents.shape
    mask = torch.zeros(B, T, L, dtype=torch.int64)
    mask[:, cond_idx[0]: cond_idx[1]] = 1
    keep = mask.unsqueeze(-1).bool()
    if mask_cond_mean:
        fill = latents[:, cond_idx[0]: cond_idx[1]].mean(dim=1).unsqueeze(1)
        x_cond = torch.where(keep, latents, fill)
    else:
        x_cond = torch.where(keep, latents, torch.zeros((), dtype=latents.dtype))
    return x_cond, mask


def sample_latents(params, shape: latent_net.NetShape, tr: transport.Transport, init: Tensor, x_cond: Tensor,
                   x_cond_mask: Tensor, y: Optional[Tensor] = None, sampling_method="ODE", sampling_kwargs=None,
                   **extra) -> Tensor:
    """Counterpart of the sample_fn call in lightning_base.py:230-234; returns the final latents."""
    fn = transport.get_sample_fn(tr, sampling_method, sampling_kwargs, **extra)

    def model(xt, t, **kw):
        return latent_net.forward(params, shape, xt, t.to(xt.dtype), **kw)

    kw = {"x_cond": x_cond, "x_cond_mask": x_cond_mask}
    if y is not None:
        kw["y"] = y
    with torch.no_grad():
        return fn(init, model, **kw)[-1]


# ----------------------------------------------------------------------------------------------------------
# frozen stage-1 decoder


@dataclass(frozen=True)
class DecoderShape:
    dim_latent: int = 32
    dim_query: int = 128
    dim_head_cross: int = 16
    dim_head_latent: int = 16
    num_head_cross: int = 8
    num_head_latent: int = 2
    num_block_cross: int = 0
    num_block_attn: int = 1
    qk_norm: bool = True
    n_entities: int = 32
    out_pos: int = 3
    act: str = "gelu_erf"  # md17 first-stage config uses src.modules.torch_modules.GELU (exact erf)


def _ln(x, w=None, b=None, eps=1e-5):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)


def _act(x, kind):
    if kind == "gelu_erf":
        return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    return torch.nn.functional.gelu(x, approximate="tanh")


def _rms(x, scale):
    xf = x.float()
    r = torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + 1e-6)
    return (xf * r).to(x.dtype) * scale


def _heads(t, h):
    b, n, hd = t.shape
    return t.reshape(b, n, h, hd // h).permute(0, 2, 1, 3)


def _mha(p, pre, xq, ctx, heads, dim_head, qk_norm, fused_qkv, mask=None):
    if fused_qkv:
        q, k, v = torch.nn.functional.linear(xq, p[pre + ".to_qkv.weight"]).chunk(3, dim=-1)
    else:
        q = torch.nn.functional.linear(xq, p[pre + ".to_q.weight"])
        k, v = torch.nn.functional.linear(ctx, p[pre + ".to_kv.weight"]).chunk(2, dim=-1)
    q, k, v = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    if qk_norm:
        q = _rms(q, p[pre + ".norm.query_norm.scale"]).to(v.dtype)
        k = _rms(k, p[pre + ".norm.key_norm.scale"]).to(v.dtype)
    s = torch.matmul(q, k.transpose(-1, -2)) * dim_head ** -0.5
    if mask is not None:  # bool [b, keys], True = attend (torch_modules.py:196-199: attn_mask of F.scaled_dot_product_attention)
        s = s.masked_fill(~mask[:, None, None, :].bool(), float("-inf"))
    o = torch.matmul(torch.softmax(s, dim=-1), v)
    o = o.permute(0, 2, 1, 3).reshape(xq.shape[0], xq.shape[1], heads * dim_head)
    return torch.nn.functional.linear(o, p[pre + ".to_out.weight"], p[pre + ".to_out.bias"])


def _ff(p, pre, x, act):
    x = _ln(x, p[pre + ".norm.weight"], p[pre + ".norm.bias"])
    x = _act(torch.nn.functional.linear(x, p[pre + ".fn.net.0.0.weight"], p[pre + ".fn.net.0.0.bias"]), act)
    return torch.nn.functional.linear(x, p[pre + ".fn.net.1.weight"], p[pre + ".fn.net.1.bias"])


def _self_block(p, pre, x, ds):
    xn = _ln(x, p[pre + ".attn.norm.weight"], p[pre + ".attn.norm.bias"])
    x = _mha(p, pre + ".attn.fn", xn, xn, ds.num_head_latent, ds.dim_head_latent, ds.qk_norm, True) + x
    return _ff(p, pre + ".ff", x, ds.act) + x


def _cross_block(p, pre, x, ctx, heads, dim_head, ds, mask=None):
    xn = _ln(x, p[pre + ".attn.norm.weight"], p[pre + ".attn.norm.bias"])
    cn = _ln(ctx, p[pre + ".attn.norm_context.weight"], p[pre + ".attn.norm_context.bias"])
    x = _mha(p, pre + ".attn.fn", xn, cn, heads, dim_head, ds.qk_norm, False, mask) + x
    return _ff(p, pre + ".ff", x, ds.act) + x


def decode(p: Dict[str, Tensor], ds: DecoderShape, z: Tensor, entities: Tensor, output: str = "pos") -> Tensor:
    """z: [F, L, dim_latent] final latents of F frames; entities: [F, A] int -> positions [F, A, out_pos] (`output`: the name of the decoder head,
    decoder.py:62-71 - "pos" for the trajectory models, "atom14_pos" (42 wide) for the peptide first stage).

    Parameter names: ``post_quant.1.*`` (lightning_base.py:28-31) and ``decoder.*`` (decoder.py:31-80).
    The entity table is used with rows clipped to unit norm, which is what nn.Embedding(max_norm=1)
    does to the looked-up rows at forward time (entity_embeddings.py:25)."""
    lat = torch.nn.functional.linear(_ln(z), p["post_quant.1.weight"], p["post_quant.1.bias"])
    table = p["decoder.entity_embedding.embedding.weight"]
    norms = table.norm(dim=-1, keepdim=True)
    table = torch.where(norms > 1.0, table / (norms + 1e-7), table)  # torch's embedding_renorm_ formula
    q = torch.nn.functional.linear(table[entities], p["decoder.query_mlp.1.weight"], p["decoder.query_mlp.1.bias"])
    for i in range(ds.num_block_attn):
        lat = _self_block(p, f"decoder.self_attn_blocks.{i}", lat, ds)
    for i in range(ds.num_block_cross):
        lat = _cross_block(p, f"decoder.cross_attn_blocks.{i}", lat, q, ds.num_head_cross, ds.dim_head_cross, ds)
    if "decoder.extender.1.weight" in p:  # DecoderQuerySplitter (decoder.py:384-388,407): 1x1 Conv1d D -> D*N, "B (D N) L -> B (L N) D"
        w, b = p["decoder.extender.1.weight"][..., 0], p["decoder.extender.1.bias"]
        n_split = w.shape[0] // lat.shape[-1]
        y = torch.nn.functional.linear(lat, w, b)  # [F, L, D*N], channel = d * N + n
        lat = y.reshape(y.shape[0], y.shape[1], lat.shape[-1], n_split).permute(0, 1, 3, 2).reshape(y.shape[0], y.shape[1] * n_split, lat.shape[-1])
    o = _cross_block(p, "decoder.output_block", q, lat, ds.num_head_cross, ds.dim_head_cross, ds)
    o = _act(torch.nn.functional.linear(o, p[f"decoder.output_layers.{output}.0.weight"], p[f"decoder.output_layers.{output}.0.bias"]), ds.act)
    return torch.nn.functional.linear(o, p[f"decoder.output_layers.{output}.2.weight"], p[f"decoder.output_layers.{output}.2.bias"])


def rel_l2(a: Tensor, b: Tensor) -> float:
    return float((a.double() - b.double()).norm() / b.double().norm())


def compute_errors(selected_traj: Tensor, all_target: Tensor) -> Tuple[Tensor, Tensor]:
    """second_stage/pedestrian.py:178-185 (same in nba.py): best-of-K ADE / FDE.  selected_traj [N,K,T,D], all_target [N,T,D]."""
    error = torch.norm(selected_traj - all_target[:, None], dim=-1)  # [N, K, T]
    error_ave = error.mean(dim=-1)
    error_final = error[..., -1]
    return error_ave.min(dim=1).values, error_final.min(dim=1).values


# ----------------------------------------------------------------------------------------------------------
# frozen stage-1 encoder (the step before the path)


@dataclass(frozen=True)
class EncoderShape:
    dim_input: int = 128
    dim_latent: int = 32
    num_latents: int = 192
    dim_head_cross: int = 16
    dim_head_latent: int = 16
    num_head_cross: int = 8
    num_head_latent: int = 2
    num_block_cross: int = 1
    num_block_attn: int = 1
    qk_norm: bool = True
    act: str = "gelu_erf"


def encode(p: Dict[str, Tensor], es: EncoderShape, x: Tensor, entities: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """x: [F, A, dim_input] per-entity inputs (the dataset-specific ``prepare_inputs`` output, first_stage/md17.py:52-58);
    entities [F, A] int; mask [F, A] bool (True = real entity) -> latents [F, num_latents, dim_latent].

    models/components/encoder.py:34-41,96-103 (context = mlp(cat(x, entity_embedding)); learned latents cross-attend to the
    context under the entity mask, then self-attend) followed by ``quant`` = Linear + LayerNorm without affine
    (lightning_base.py:22-25,37-40).  Parameter names: ``encoder.*`` and ``quant.0.*``."""
    table = p["encoder.entity_embedding.embedding.weight"]
    norms = table.norm(dim=-1, keepdim=True)
    table = torch.where(norms > 1.0, table / (norms + 1e-7), table)
    ctx = torch.cat([x, table[entities]], dim=-1)
    ctx = torch.nn.functional.linear(_act(torch.nn.functional.linear(ctx, p["encoder.mlp.0.weight"], p["encoder.mlp.0.bias"]), es.act),
                                     p["encoder.mlp.2.weight"], p["encoder.mlp.2.bias"])
    lat = p["encoder.latents"][None].expand(x.shape[0], -1, -1)
    for i in range(es.num_block_cross):
        lat = _cross_block(p, f"encoder.cross_attn_blocks.{i}", lat, ctx, es.num_head_cross, es.dim_head_cross, es, mask)
    for i in range(es.num_block_attn):
        lat = _self_block(p, f"encoder.blocks_attn.{i}", lat, es)
    return _ln(torch.nn.functional.linear(lat, p["quant.0.weight"], p["quant.0.bias"]))
