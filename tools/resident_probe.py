#!/usr/bin/env python3
"""Phase decomposition of the trajectory-resident kernel with the -DLSL_EXPERIMENTS build (LSL_RES_SKIP bits: 1 attention, 2 linear1,
4 linear2, 8 LayerNorm; results WRONG on purpose).  Usage (GPU box): LSL_RES_SKIP=<mask> python tools/resident_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lam_slide_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "_exp", "liblamslide_hip_exp.so")
from lam_slide_amd import CreateTransport, LatentSIV3, SecondStageSampler  # noqa: E402
from lam_slide_amd.synthetic import seeded_state_dict  # noqa: E402

dev = torch.device("cuda:0")
kw = dict(depth=6, in_dim=32, hidden_size=128, num_heads=4, mlp_ratio=2, vec_in_dim=256, normalize=True)
net = LatentSIV3(reset_parameters=False, **kw)
net.load_state_dict(seeded_state_dict(net, seed=0))
net.to(dev)
for B, ns in ((20, 11), (20, 2), (1280, 11), (1280, 2)):
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(B, 20, 2, 32, generator=g).to(dev)
    init = torch.randn(B, 20, 2, 32, generator=g).to(dev)
    y = torch.randn(B, 256, generator=g).to(dev)
    drv = SecondStageSampler(net, CreateTransport("GVP", "data")(), cond_idx=(0, 8), sampling_kwargs={"sampling_method": "euler", "num_steps": ns})
    for _ in range(3):
        drv.sample_latents(lat, y=y, init=init)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        drv.sample_latents(lat, y=y, init=init)
    torch.cuda.synchronize()
    if "--stamps" in sys.argv:
        import ctypes as C
        lib = _lib.load()
        buf = (C.c_ulonglong * 8)()
        lib.lsl_debug_res_stamps(buf)  # clear
        drv.sample_latents(lat, y=y, init=init)
        lib.lsl_debug_res_stamps(buf)
        tot = sum(buf)
        names = ["LayerNorm", "linear1", "attention", "linear2", "embed", "head", "-", "-"]
        print("  phase cycles (workgroup 0, wave 0): " + "  ".join(f"{n} {v} ({100 * v / tot:.0f} %)" for n, v in zip(names, buf) if v))
    print(f"LSL_RES_SKIP={os.environ.get('LSL_RES_SKIP', '0'):>2}  B={B:5d}: {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms per {ns - 1}-update call")
