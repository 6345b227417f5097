#!/usr/bin/env python3
"""Wall time of small-batch sampling calls through the Python surface (no per-kernel profiling): what LSL_GRAPH changes.
Usage (GPU box): LSL_GRAPH=0|1 python tools/latency_small_batch.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lam_slide_amd import CreateTransport, LatentSIV3, SecondStageSampler  # noqa: E402
from lam_slide_amd.synthetic import seeded_state_dict  # noqa: E402

dev = torch.device("cuda:0")
CASES = {
    "pedestrian (D=128, depth 6, T=20, L=2, y), 10 updates": (dict(depth=6, in_dim=32, hidden_size=128, num_heads=4, mlp_ratio=2, vec_in_dim=256, normalize=True), 20, 2, 11),
    "md17 cfg-1 (D=256, depth 4, T=30, L=192), 10 updates": (dict(depth=4, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=2), 30, 192, 11),
}
for name, (kw, T, L, ns) in CASES.items():
    net = LatentSIV3(reset_parameters=False, **kw)
    net.load_state_dict(seeded_state_dict(net, seed=0))
    net.to(dev)
    for B in (1, 20, 160):
        if L == 192 and B > 20:
            continue
        g = torch.Generator().manual_seed(1)
        lat = torch.randn(B, T, L, 32, generator=g).to(dev)
        init = torch.randn(B, T, L, 32, generator=g).to(dev)
        y = torch.randn(B, kw["vec_in_dim"], generator=g).to(dev) if kw.get("vec_in_dim") else None
        drv = SecondStageSampler(net, CreateTransport("GVP", "data")(), cond_idx=(0, 8), sampling_kwargs={"sampling_method": "euler", "num_steps": ns})
        for _ in range(4):
            out = drv.sample_latents(lat, y=y, init=init)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            out = drv.sample_latents(lat, y=y, init=init)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"LSL_GRAPH={os.environ.get('LSL_GRAPH', '0')}  {name}  batch {B:4d}: {dt * 1e3:7.2f} ms per call  ({B / dt:9.1f} trajectories/s)  checksum {float(out.abs().mean()):.6f}")
