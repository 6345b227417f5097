#!/usr/bin/env python3
"""Does running two half-batches of the headline workload on two HIP streams beat one full batch?  (Kernels of different phases can
share the chip: e.g. the HBM-bound LayerNorm of one half under the MFMA-bound GEMM of the other.)  Usage (GPU box): python tools/two_streams.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from lam_slide_amd import CreateTransport, LatentSIV3, SecondStageSampler
from lam_slide_amd.synthetic import seeded_state_dict
dev = torch.device("cuda:0")
kw = dict(depth=4, in_dim=32, hidden_size=512, num_heads=16, mlp_ratio=2)
def make():
    net = LatentSIV3(reset_parameters=False, **kw); net.load_state_dict(seeded_state_dict(net, seed=0)); net.to(dev)
    return SecondStageSampler(net, CreateTransport("Linear", "velocity")(), cond_idx=(0, 5), sampling_kwargs={"sampling_method": "euler", "num_steps": 51})
T, L, C = 30, 256, 32
g = torch.Generator().manual_seed(1)
def data(B):
    return torch.randn(B, T, L, C, generator=g).to(dev), torch.randn(B, T, L, C, generator=g).to(dev)
one = make(); lat32, init32 = data(32)
for _ in range(2): one.sample_latents(lat32, init=init32)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): one.sample_latents(lat32, init=init32)
torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 3
print(f"one stream,  B=32: {t1*1e3:8.1f} ms per step  {32/t1:6.2f} traj/s")
for split in ((16, 16), (8, 8, 8, 8)):
    drv = [make() for _ in split]; dat = [data(b) for b in split]; streams = [torch.cuda.Stream() for _ in split]
    def step():
        for d, (la, ini), s in zip(drv, dat, streams):
            with torch.cuda.stream(s):
                d.sample_latents(la, init=ini)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 3
    print(f"{len(split)} streams, B={split}: {t2*1e3:8.1f} ms per step  {sum(split)/t2:6.2f} traj/s")
