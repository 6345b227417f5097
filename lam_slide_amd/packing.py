"""Weight packing: reference ``state_dict`` (SURVEY.md a2 names) -> device blobs in the layout the HIP
kernels read (include/lsl_api.h, ``lsl_weights`` / ``lsl_block_weights``).  Done once per weight version.

Layout decisions
  * every attention head is padded to ``head_dim_pad`` (16 or 32) rows so a 32x32 MFMA tile never straddles
    a head: linear1 rows are reordered to [q heads | k heads | v heads | mlp] with zero rows in the padding,
    linear2 columns to [attention heads (padded) | mlp] with zero columns in the padding;
  * linear1 / linear2 weights are bf16 (MFMA operands), everything else stays fp32;
  * all ``blocks.i.modulation.lin`` and ``adaLN_modulation.1`` are stacked into one [(6*depth+2)*D, D] matrix
    so one launch produces every modulation vector of an evaluation.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


@dataclass(frozen=True)
class PackedDims:
    in_dim: int
    hidden: int
    heads: int
    head_dim: int
    head_dim_pad: int
    mlp_dim: int        # true M
    mlp_dim_pad: int    # packed M
    depth: int
    vec_in_dim: int
    normalize: bool
    theta: float

    @property
    def hhd(self) -> int:
        return self.heads * self.head_dim_pad

    @property
    def f1(self) -> int:
        return 3 * self.hhd + self.mlp_dim_pad

    @property
    def k2(self) -> int:
        return self.hhd + self.mlp_dim_pad

    @property
    def modw(self) -> int:
        return (6 * self.depth + 2) * self.hidden


def make_dims(depth: int, in_dim: int, hidden_size: int, num_heads: int, mlp_ratio, vec_in_dim: Optional[int], normalize: bool,
              theta) -> PackedDims:
    if hidden_size % num_heads != 0:
        raise ValueError(f"Hidden size {hidden_size} must be divisible by num_heads {num_heads}")  # latent_si_v31.py:92-95
    hd = hidden_size // num_heads
    hdp = 16 if hd <= 16 else 32
    m = int(hidden_size * mlp_ratio)
    mp = _round_up(m, 32)
    if (num_heads * hdp + mp) % 64:
        mp += 32
    return PackedDims(in_dim, hidden_size, num_heads, hd, hdp, m, mp, depth, int(vec_in_dim or 0), bool(normalize), float(theta))


def strip_prefixes(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """torch.compile wraps the backbone in the reference launch scripts (second_stage/md17.py:53-55);
    its state_dict keys then carry ``_orig_mod.``."""
    return {k.replace("_orig_mod.", ""): v for k, v in sd.items()}


def pack_block(sd, pre: str, dm: PackedDims, device) -> Dict[str, torch.Tensor]:
    D, H, hd, hdp, M = dm.hidden, dm.heads, dm.head_dim, dm.head_dim_pad, dm.mlp_dim
    w1 = sd[pre + ".linear1.weight"].detach().float().cpu()
    b1 = sd[pre + ".linear1.bias"].detach().float().cpu()
    w2 = sd[pre + ".linear2.weight"].detach().float().cpu()
    b2 = sd[pre + ".linear2.bias"].detach().float().cpu()
    assert w1.shape == (3 * D + M, D) and w2.shape == (D, D + M), (w1.shape, w2.shape)
    # rows are padded with zeros to whole 256-row GEMM tiles: the LDS-DMA loader reads full tiles without clamping
    w1p = torch.zeros(_round_up(dm.f1, 256), D)
    b1p = torch.zeros(_round_up(dm.f1, 256))
    for sec in range(3):
        src_w = w1[sec * D:(sec + 1) * D].reshape(H, hd, D)
        src_b = b1[sec * D:(sec + 1) * D].reshape(H, hd)
        dst_w = w1p[sec * dm.hhd:(sec + 1) * dm.hhd].view(H, hdp, D)
        dst_b = b1p[sec * dm.hhd:(sec + 1) * dm.hhd].view(H, hdp)
        dst_w[:, :hd] = src_w
        dst_b[:, :hd] = src_b
    w1p[3 * dm.hhd:3 * dm.hhd + M] = w1[3 * D:]
    b1p[3 * dm.hhd:3 * dm.hhd + M] = b1[3 * D:]
    w2p = torch.zeros(_round_up(D, 256), dm.k2)
    w2p[:D, :dm.hhd].view(D, H, hdp)[:, :, :hd] = w2[:, :D].reshape(D, H, hd)
    w2p[:D, dm.hhd:dm.hhd + M] = w2[:, D:]
    qs = torch.zeros(hdp)
    ks = torch.zeros(hdp)
    qs[:hd] = sd[pre + ".norm.query_norm.scale"].detach().float().cpu()
    ks[:hd] = sd[pre + ".norm.key_norm.scale"].detach().float().cpu()
    return {
        "w1": w1p.to(torch.bfloat16).contiguous().to(device), "b1": b1p.to(device), "qs": qs.to(device), "ks": ks.to(device),
        "w2": w2p.to(torch.bfloat16).contiguous().to(device), "b2": b2.contiguous().to(device),
    }


class PackedWeights:
    """Owns the packed device tensors and the ctypes structs that point at them."""

    def __init__(self, sd: Dict[str, torch.Tensor], dm: PackedDims, device):
        sd = strip_prefixes(sd)
        f32 = lambda k: sd[k].detach().float().contiguous().to(device)  # noqa: E731
        self.dims = dm
        t: Dict[str, torch.Tensor] = {}
        t["x_in_w"], t["x_in_b"] = f32("x_in.weight"), f32("x_in.bias")
        t["cond_w"], t["cond_b"] = f32("cond_to_emb.weight"), f32("cond_to_emb.bias")
        t["mask_emb"] = f32("mask_to_emb.weight")
        # mmdit.py:103-105, built with the same fp32 torch ops so the table is bit-identical to the reference's
        t["time_freqs"] = torch.exp(-math.log(10000.0) * torch.arange(0, 128, dtype=torch.float32) / 128).to(device)
        t["time_w1"], t["time_b1"] = f32("time_in.in_layer.weight"), f32("time_in.in_layer.bias")
        t["time_w2"], t["time_b2"] = f32("time_in.out_layer.weight"), f32("time_in.out_layer.bias")
        if dm.vec_in_dim:
            t["vec_w1"], t["vec_b1"] = f32("vec_in.in_layer.weight"), f32("vec_in.in_layer.bias")
            t["vec_w2"], t["vec_b2"] = f32("vec_in.out_layer.weight"), f32("vec_in.out_layer.bias")
        mw = [sd[f"blocks.{i}.modulation.lin.weight"].detach().float() for i in range(dm.depth)] + [sd["adaLN_modulation.1.weight"].detach().float()]
        mb = [sd[f"blocks.{i}.modulation.lin.bias"].detach().float() for i in range(dm.depth)] + [sd["adaLN_modulation.1.bias"].detach().float()]
        t["mod_w"] = torch.cat([w.cpu() for w in mw]).contiguous().to(device)
        t["mod_b"] = torch.cat([b.cpu() for b in mb]).contiguous().to(device)
        t["out_w"], t["out_b"] = f32("linear.weight"), f32("linear.bias")
        self.tensors = t
        self.blocks = []
        for i in range(dm.depth):
            for name in ("spatial_block", "temporal_block"):
                self.blocks.append(pack_block(sd, f"blocks.{i}.{name}", dm, device))
        self.c_blocks = (_lib.BlockWeights * len(self.blocks))()
        for cb, b in zip(self.c_blocks, self.blocks):
            for k in ("w1", "b1", "qs", "ks", "w2", "b2"):
                setattr(cb, k, b[k].data_ptr())
        self.c_weights = _lib.Weights()
        for name, _ in _lib.Weights._fields_:
            if name == "blocks":
                continue
            setattr(self.c_weights, name, t[name].data_ptr() if name in t else None)
        self.c_weights.blocks = C.cast(self.c_blocks, C.POINTER(_lib.BlockWeights))

    def desc(self) -> "_lib.ModelDesc":
        dm = self.dims
        return _lib.ModelDesc(dm.in_dim, dm.hidden, dm.heads, dm.head_dim, dm.head_dim_pad, dm.mlp_dim_pad, dm.depth, dm.vec_in_dim,
                              int(dm.normalize), dm.theta)

    def nbytes(self) -> int:
        n = sum(v.numel() * v.element_size() for v in self.tensors.values())
        return n + sum(v.numel() * v.element_size() for b in self.blocks for v in b.values())
