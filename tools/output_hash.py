#!/usr/bin/env python3
"""SHA-256 of the fused sampler's output on a few seeded shapes: run it once per library build (tools/gpu.sh libab swaps the in-tree library) to
show that a kernel change is bit-preserving.  Used for the round-6 attention changes (profiles/r06_experiments.txt sections 9 and 13).
Usage (GPU box): python tools/output_hash.py [peptide|long]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lam_slide_amd import CreateTransport, LatentSIV3, SecondStageSampler  # noqa: E402
from lam_slide_amd.synthetic import seeded_state_dict  # noqa: E402

CASES = {
    # (hidden, heads, mlp_ratio, in_dim, B, T, L)
    "peptide": [(384, 16, 4, 96, 3, T, 2) for T in (1000, 700, 513, 300, 257)],                                  # chunked-key temporal attention, 24 -> 32 padded heads
    "long": [(512, 16, 2, 32, 2, 30, 256), (512, 16, 2, 32, 3, 5, 200), (256, 8, 2, 32, 2, 4, 251), (512, 16, 2, 32, 2, 3, 129), (128, 4, 2, 32, 2, 3, 160)],
}
dev = torch.device("cuda:0")
for (D, H, mr, C, B, T, L) in CASES[sys.argv[1] if len(sys.argv) > 1 else "long"]:
    net = LatentSIV3(reset_parameters=False, depth=2, in_dim=C, hidden_size=D, num_heads=H, mlp_ratio=mr)
    net.load_state_dict(seeded_state_dict(net, seed=0))
    net.to(dev)
    g = torch.Generator().manual_seed(1)
    lat, init = torch.randn(B, T, L, C, generator=g).to(dev), torch.randn(B, T, L, C, generator=g).to(dev)
    drv = SecondStageSampler(net, CreateTransport("GVP", "data")(), cond_idx=(0, 1), sampling_kwargs={"sampling_method": "euler", "num_steps": 5})
    out = drv.sample_latents(lat, init=init)
    torch.cuda.synchronize()
    print(f"D={D} H={H} T={T} L={L}: {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}  mean |x| {float(out.abs().mean()):.6f}")
