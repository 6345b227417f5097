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
