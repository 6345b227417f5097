"""CPU restatement of one evaluation of the latent SiT network (LatentSIV3).

Test infrastructure only (see ``oracle/__init__.py``).  Written as a pure
function over a ``state_dict`` (reference parameter names) so it
needs none of the reference's classes.  Works in fp32 or fp64.

Reference lines restated here (all under /root/reference/src/models/components/latent/):
  mmdit.py:11-18     exact-erf GELU
  mmdit.py:21-22     modulate
  mmdit.py:75-82     RoPE table (fp64 angles, cast to fp32)
  mmdit.py:85-90     adjacent-pair rotation, computed in fp32
  mmdit.py:93-115    sinusoidal timestep embedding (fp32 arguments)
  mmdit.py:118-126   two-layer SiLU embedder
  mmdit.py:129-148   per-head RMS normalisation of q and k (eps 1e-6)
  mmdit.py:184-197   modulation: Linear(SiLU(vec)) split in six
  mmdit.py:240-249   parallel attention + MLP block
  latent_si_v31.py:45-63    spatial then temporal sub-block of one layer
  latent_si_v31.py:168-188  whole network
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor


@dataclass(frozen=True)
class NetShape:
    """Shape-independent hyper-parameters of the network (reference ctor kwargs)."""

    depth: int
    in_dim: int
    hidden_size: int
    num_heads: int
    mlp_ratio: float = 2
    vec_in_dim: Optional[int] = None
    theta: int = 10_000
    normalize: bool = False
    share_weights: bool = False
    attention_mode: str = "scaled_dot_product"  # anything else selects attention_linear (mmdit.py:50-53)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def mlp_dim(self) -> int:
        return int(self.hidden_size * self.mlp_ratio)


def _lin(p: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    return torch.nn.functional.linear(x, p[name + ".weight"], p[name + ".bias"])


def gelu_erf(x: Tensor) -> Tensor:
    # mmdit.py:11-18
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def time_features(t: Tensor, dim: int = 256, max_period: float = 10000.0, factor: float = 1000.0) -> Tensor:
    # mmdit.py:93-115 -- arguments are formed in fp32 whatever dtype t has.
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = (factor * t)[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    return emb.to(t.dtype) if torch.is_floating_point(t) else emb


def rope_cos_sin(n_pos: int, head_dim: int, theta: float):
    """cos/sin of p * theta**(-2j/hd), p < n_pos, j < hd/2: fp64 then rounded to fp32 (mmdit.py:75-82)."""
    j = torch.arange(0, head_dim, 2, dtype=torch.float64) / head_dim
    omega = 1.0 / (theta ** j)
    ang = torch.arange(n_pos, dtype=torch.float64)[:, None] * omega[None]
    return torch.cos(ang).float(), torch.sin(ang).float()


def rotate_pairs(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """x: [..., S, hd]; cos/sin: [S, hd/2].  Rotation done in fp32, result cast back (mmdit.py:85-90)."""
    xf = x.float()
    x0, x1 = xf[..., 0::2], xf[..., 1::2]
    y0 = cos * x0 - sin * x1
    y1 = sin * x0 + cos * x1
    return torch.stack([y0, y1], dim=-1).reshape(x.shape).to(x.dtype)


def head_rms(x: Tensor, scale: Tensor) -> Tensor:
    # mmdit.py:129-137: statistics in fp32, cast back, then learned per-channel scale.
    xf = x.float()
    r = torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + 1e-6)
    return (xf * r).to(x.dtype) * scale


def layer_norm(x: Tensor, eps: float) -> Tensor:
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), eps=eps)


def attn_mlp_block(p: Dict[str, Tensor], pre: str, u: Tensor, cos: Tensor, sin: Tensor, sh: NetShape,
                   taps: Optional[dict] = None, tag: str = "") -> Tensor:
    """ParallelMLPAttentionV2 on u: [G, S, D] (G independent sequences of length S)."""
    G, S, D = u.shape
    H, hd = sh.num_heads, sh.head_dim
    z = _lin(p, pre + ".linear1", u)
    qkv, mlp = z[..., : 3 * D], z[..., 3 * D:]
    # channel layout of the first 3D outputs is (K, H, hd), hd fastest (mmdit.py:244)
    qkv = qkv.reshape(G, S, 3, H, hd).permute(2, 0, 3, 1, 4)  # K G H S hd
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = head_rms(q, p[pre + ".norm.query_norm.scale"]).to(v.dtype)
    k = head_rms(k, p[pre + ".norm.key_norm.scale"]).to(v.dtype)
    if taps is not None:
        taps[tag + "z"] = z
        taps[tag + "q_norm"], taps[tag + "k_norm"] = q, k
    q, k = rotate_pairs(q, cos, sin), rotate_pairs(k, cos, sin)
    if sh.attention_mode == "scaled_dot_product":
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
        a = torch.matmul(torch.softmax(s, dim=-1), v)  # G H S hd
    else:
        # attention_linear (mmdit.py:58-72): q softmax over the head channels, k softmax over the positions, q scaled by hd^-1/2,
        # context[d][e] = sum_n k[n][d] v[n][e], out[n][e] = sum_d q[n][d] context[d][e]
        context = torch.matmul(torch.softmax(k, dim=-2).transpose(-1, -2), v)  # G H hd hd
        a = torch.matmul(torch.softmax(q, dim=-1) * hd ** -0.5, context)     # G H S hd
    a = a.permute(0, 2, 1, 3).reshape(G, S, D)
    if taps is not None:
        taps[tag + "q_rope"], taps[tag + "k_rope"] = q, k
        taps[tag + "attn"] = a
    out = _lin(p, pre + ".linear2", torch.cat([a, gelu_erf(mlp)], dim=-1))
    if taps is not None:
        taps[tag + "out"] = out
    return out


def conditioning_vector(p: Dict[str, Tensor], t: Tensor, y: Optional[Tensor]) -> Tensor:
    # latent_si_v31.py:176-178 with mmdit.py:118-126
    e = time_features(t, 256)
    vec = _lin(p, "time_in.out_layer", torch.nn.functional.silu(_lin(p, "time_in.in_layer", e)))
    if y is not None:
        vec = vec + _lin(p, "vec_in.out_layer", torch.nn.functional.silu(_lin(p, "vec_in.in_layer", y)))
    return vec


def forward(p: Dict[str, Tensor], sh: NetShape, x: Tensor, t: Tensor, x_cond: Tensor, x_cond_mask: Tensor,
            y: Optional[Tensor] = None, taps: Optional[dict] = None) -> Tensor:
    """One network evaluation.  x, x_cond: [B,T,L,C]; t: [B]; mask: [B,T,L] int; y: [B,V] or None."""
    B, T, L, _ = x.shape
    D = sh.hidden_size
    h = _lin(p, "x_in", x) + _lin(p, "cond_to_emb", x_cond) + p["mask_to_emb.weight"][x_cond_mask.long()]
    if sh.normalize:
        h = layer_norm(h, 1e-5)  # F.layer_norm default eps (latent_si_v31.py:173-174)
    vec = conditioning_vector(p, t, y)
    sv = torch.nn.functional.silu(vec)
    cs_l, sn_l = rope_cos_sin(L, sh.head_dim, sh.theta)
    cs_t, sn_t = rope_cos_sin(T, sh.head_dim, sh.theta)
    if taps is not None:
        taps["h0"], taps["vec"] = h, vec
    for i in range(sh.depth):
        pre = f"blocks.{i}"
        m = _lin(p, pre + ".modulation.lin", sv)  # [B, 6D]
        sh1, sc1, g1, sh2, sc2, g2 = [c[:, None, None, :] for c in m.chunk(6, dim=-1)]
        u = layer_norm(h, 1e-6) * (1 + sc1) + sh1
        o = attn_mlp_block(p, pre + ".spatial_block", u.reshape(B * T, L, D), cs_l, sn_l, sh, taps, f"l{i}.sp.")
        h = h + g1 * o.reshape(B, T, L, D)
        u = layer_norm(h, 1e-6) * (1 + sc2) + sh2
        u = u.permute(0, 2, 1, 3).reshape(B * L, T, D)
        o = attn_mlp_block(p, pre + ".temporal_block", u, cs_t, sn_t, sh, taps, f"l{i}.tm.")
        h = h + g2 * o.reshape(B, L, T, D).permute(0, 2, 1, 3)
        if taps is not None:
            taps[f"l{i}.mod"] = m
            taps[f"l{i}.h"] = h
    m = _lin(p, "adaLN_modulation.1", sv)
    shf, scf = [c[:, None, None, :] for c in m.chunk(2, dim=-1)]
    out = _lin(p, "linear", layer_norm(h, 1e-6) * (1 + scf) + shf)
    if taps is not None:
        taps["final_mod"] = m
        taps["out"] = out
    return out


def cast_params(p: Dict[str, Tensor], dtype: torch.dtype) -> Dict[str, Tensor]:
    out = {}
    for k, v in p.items():
        k = k.replace("_orig_mod.", "")
        out[k] = v.to(dtype) if v.is_floating_point() else v
    return out


def random_params(sh: NetShape, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Seeded default-PyTorch-style init with the reference's parameter names and shapes
    (latent_si_v31.py:68-121 with reset_parameters=False).  Not bit-identical to constructing the
    reference module (different draw order); use the golden fixtures when reference weights matter."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, Tensor] = {}

    def lin(name, out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        p[name + ".weight"] = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound
        p[name + ".bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * bound

    D, C, M, hd = sh.hidden_size, sh.in_dim, sh.mlp_dim, sh.head_dim
    lin("x_in", D, C)
    lin("cond_to_emb", D, C)
    p["mask_to_emb.weight"] = torch.randn(2, D, generator=g)
    lin("time_in.in_layer", D, 256)
    lin("time_in.out_layer", D, D)
    if sh.vec_in_dim is not None:
        lin("vec_in.in_layer", D, sh.vec_in_dim)
        lin("vec_in.out_layer", D, D)
    n_unique = 1 if sh.share_weights else sh.depth
    for i in range(n_unique):
        lin(f"blocks.{i}.modulation.lin", 6 * D, D)
        for blk in ("spatial_block", "temporal_block"):
            lin(f"blocks.{i}.{blk}.linear1", 3 * D + M, D)
            lin(f"blocks.{i}.{blk}.linear2", D, D + M)
            p[f"blocks.{i}.{blk}.norm.query_norm.scale"] = 1.0 + 0.1 * torch.randn(hd, generator=g)
            p[f"blocks.{i}.{blk}.norm.key_norm.scale"] = 1.0 + 0.1 * torch.randn(hd, generator=g)
    if sh.share_weights:
        for i in range(1, sh.depth):
            for k in [k for k in p if k.startswith("blocks.0.")]:
                p[k.replace("blocks.0.", f"blocks.{i}.")] = p[k]
    lin("adaLN_modulation.1", 2 * D, D)
    lin("linear", C, D)
    return cast_params(p, dtype)
