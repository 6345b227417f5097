"""Seeded synthetic weights and inputs for benchmarks and smoke runs (no dataset or checkpoint exists offline).

``seeded_state_dict`` draws every parameter of a ``LatentSIV3`` with the default-PyTorch-style ranges
(U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for Linear, N(0,1) for the mask embedding, 1 + 0.1 N(0,1) for the QK-norm
scales) from one CPU generator, i.e. the ``reset_parameters=False`` regime of the shipped configs
(configs/model/md17/second-stage.yaml:22) where modulation and output layers are NOT zero-initialised."""
from __future__ import annotations

import math
from typing import Dict

import torch


def seeded_state_dict(module: torch.nn.Module, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    sd = module.state_dict()
    seen = {}
    for name, ref in sd.items():
        key = ref.data_ptr()
        if key in seen:  # share_weights: the same tensor appears under several block indices
            out[name] = out[seen[key]]
            continue
        seen[key] = name
        if name.endswith("_norm.scale"):
            v = 1.0 + 0.1 * torch.randn(ref.shape, generator=g)
        elif name == "mask_to_emb.weight":
            v = torch.randn(ref.shape, generator=g)
        else:
            fan_in = ref.shape[1] if ref.dim() == 2 else sd[name[: -len("bias")] + "weight"].shape[1]
            bound = 1.0 / math.sqrt(fan_in)
            v = (torch.rand(ref.shape, generator=g) * 2 - 1) * bound
        out[name] = v
    return out


def seeded_decoder_state_dict(in_dim: int = 32, dim_latent: int = 32, dim_query: int = 128, dim_emb: int = 128, n_entities: int = 32,
                              num_head_latent: int = 2, dim_head_latent: int = 16, num_head_cross: int = 8, dim_head_cross: int = 16,
                              num_block_attn: int = 1, num_block_cross: int = 0, out_dim: int = 3, seed: int = 0) -> Dict[str, torch.Tensor]:
    """A frozen stage-1 ``post_quant`` + ``Decoder`` state dict under the reference's parameter names (decoder.py:31-80,
    lightning_base.py:28-31) with PyTorch-default-style ranges; defaults = the MD17 first-stage shape (configs/model/md17/first-stage.yaml)."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}

    def lin(name, o, i, bias=True):
        b = 1.0 / math.sqrt(i)
        out[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * b
        if bias:
            out[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * b

    def ln(name, d):
        out[name + ".weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
        out[name + ".bias"] = 0.1 * torch.randn(d, generator=g)

    def block(pre, dim, ctx, heads, dh):
        inner = heads * dh
        ln(pre + ".attn.norm", dim)
        if ctx is None:
            lin(pre + ".attn.fn.to_qkv", 3 * inner, dim, bias=False)
        else:
            ln(pre + ".attn.norm_context", ctx)
            lin(pre + ".attn.fn.to_q", inner, dim, bias=False)
            lin(pre + ".attn.fn.to_kv", 2 * inner, ctx, bias=False)
        lin(pre + ".attn.fn.to_out", dim, inner)
        out[pre + ".attn.fn.norm.query_norm.scale"] = 1.0 + 0.1 * torch.randn(dh, generator=g)
        out[pre + ".attn.fn.norm.key_norm.scale"] = 1.0 + 0.1 * torch.randn(dh, generator=g)
        ln(pre + ".ff.norm", dim)
        lin(pre + ".ff.fn.net.0.0", dim, dim)
        lin(pre + ".ff.fn.net.1", dim, dim)

    lin("post_quant.1", dim_latent, in_dim)
    table = torch.randn(n_entities, dim_emb, generator=g) / math.sqrt(dim_emb)
    out["decoder.entity_embedding.embedding.weight"] = table * torch.linspace(0.6, 1.6, n_entities)[:, None]  # some rows above max_norm
    lin("decoder.query_mlp.1", dim_query, dim_emb)
    for i in range(num_block_attn):
        block(f"decoder.self_attn_blocks.{i}", dim_latent, None, num_head_latent, dim_head_latent)
    for i in range(num_block_cross):
        block(f"decoder.cross_attn_blocks.{i}", dim_latent, dim_query, num_head_cross, dim_head_cross)
    block("decoder.output_block", dim_query, dim_latent, num_head_cross, dim_head_cross)
    lin("decoder.output_layers.pos.0", dim_query, dim_query)
    lin("decoder.output_layers.pos.2", out_dim, dim_query)
    return out


def seeded_encoder_state_dict(dim_input: int = 128, dim_latent: int = 32, num_latents: int = 192, dim_emb: int = 128, n_entities: int = 32,
                              num_head_cross: int = 8, dim_head_cross: int = 16, num_head_latent: int = 2, dim_head_latent: int = 16,
                              num_block_cross: int = 1, num_block_attn: int = 1, seed: int = 0) -> Dict[str, torch.Tensor]:
    """A frozen stage-1 ``Encoder`` + ``quant`` state dict under the reference's parameter names (encoder.py:11-103,
    lightning_base.py:22-25); defaults = the MD17 first-stage shape (configs/model/md17/first-stage.yaml:52-67)."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    dim_ctx = dim_input + dim_emb

    def lin(name, o, i, bias=True):
        b = 1.0 / math.sqrt(i)
        out[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * b
        if bias:
            out[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * b

    def ln(name, d):
        out[name + ".weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
        out[name + ".bias"] = 0.1 * torch.randn(d, generator=g)

    def block(pre, dim, ctx, heads, dh):
        inner = heads * dh
        ln(pre + ".attn.norm", dim)
        if ctx is None:
            lin(pre + ".attn.fn.to_qkv", 3 * inner, dim, bias=False)
        else:
            ln(pre + ".attn.norm_context", ctx)
            lin(pre + ".attn.fn.to_q", inner, dim, bias=False)
            lin(pre + ".attn.fn.to_kv", 2 * inner, ctx, bias=False)
        lin(pre + ".attn.fn.to_out", dim, inner)
        out[pre + ".attn.fn.norm.query_norm.scale"] = 1.0 + 0.1 * torch.randn(dh, generator=g)
        out[pre + ".attn.fn.norm.key_norm.scale"] = 1.0 + 0.1 * torch.randn(dh, generator=g)
        ln(pre + ".ff.norm", dim)
        lin(pre + ".ff.fn.net.0.0", dim, dim)
        lin(pre + ".ff.fn.net.1", dim, dim)

    table = torch.randn(n_entities, dim_emb, generator=g) / math.sqrt(dim_emb)
    out["encoder.entity_embedding.embedding.weight"] = table * torch.linspace(0.6, 1.6, n_entities)[:, None]
    out["encoder.latents"] = torch.randn(num_latents, dim_latent, generator=g)
    lin("encoder.mlp.0", dim_latent, dim_ctx)
    lin("encoder.mlp.2", dim_ctx, dim_latent)
    for i in range(num_block_cross):
        block(f"encoder.cross_attn_blocks.{i}", dim_latent, dim_ctx, num_head_cross, dim_head_cross)
    for i in range(num_block_attn):
        block(f"encoder.blocks_attn.{i}", dim_latent, None, num_head_latent, dim_head_latent)
    lin("quant.0", dim_latent, dim_latent)
    return out
