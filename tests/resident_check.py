"""Checker script (test infrastructure, not collected by pytest): the trajectory-resident kernel against the oracle at the pedestrian
shape, batch independence of a trajectory's bits, and wall time per 10-update call at B = 1 / 20 / 160 / 1280.
Usage (GPU box): python tests/resident_check.py   (LSL_LIB=<path> selects another build of the library)"""
import sys, time, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from lam_slide_amd import _lib
if os.environ.get('LSL_LIB'): _lib.LIB_PATH = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), os.environ['LSL_LIB'])
from lam_slide_amd import CreateTransport, LatentSIV3, SecondStageSampler, Sampler
from lam_slide_amd.synthetic import seeded_state_dict
from oracle import harness, latent_net, transport as otr
dev = torch.device("cuda:0")
kw = dict(depth=6, in_dim=32, hidden_size=128, num_heads=4, mlp_ratio=2, vec_in_dim=256, normalize=True)
for use_y in (True, False):
    k2 = dict(kw)
    if not use_y:
        k2.pop("vec_in_dim")
    sh = latent_net.NetShape(**k2)
    p = latent_net.random_params(sh, seed=11)
    net = LatentSIV3(reset_parameters=False, **k2); net.load_state_dict(p); net.to(dev)
    B, T, L = 5, 20, 2
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(B, T, L, 32, generator=g); init = torch.randn(B, T, L, 32, generator=g)
    y = torch.randn(B, 256, generator=g) if use_y else None
    xc, m = harness.setup_conditioning(lat, (0, 8), True)
    mk = {"x_cond": xc.to(dev), "x_cond_mask": m.to(dev)}
    if y is not None: mk["y"] = y.to(dev)
    skw = {"sampling_method": "euler", "num_steps": 11}
    s = Sampler(CreateTransport("GVP", "data")(), fused=True)
    got = s.get_sample_fn("ODE", skw)(init.to(dev), net.forward, **mk)[-1].cpu()
    want = harness.sample_latents(p, sh, otr.Transport("GVP", "data"), init, xc, m, y, "ODE", skw)
    print("y" if use_y else "no-y", "rel l2", harness.rel_l2(got, want), "finite", bool(torch.isfinite(got).all()))
    one = s.get_sample_fn("ODE", skw)(init[2:3].to(dev), net.forward, **{k: v[2:3] for k, v in mk.items()})[-1].cpu()
    print("  batch-independence bits:", torch.equal(one, got[2:3]))
    for Bt in (1, 20, 160, 1280):
        g = torch.Generator().manual_seed(1)
        lat = torch.randn(Bt, T, L, 32, generator=g).to(dev); init = torch.randn(Bt, T, L, 32, generator=g).to(dev)
        yy = torch.randn(Bt, 256, generator=g).to(dev) if use_y else None
        drv = SecondStageSampler(net, CreateTransport("GVP", "data")(), cond_idx=(0, 8), sampling_kwargs=skw)
        for _ in range(3): out = drv.sample_latents(lat, y=yy, init=init)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
        for _ in range(n): out = drv.sample_latents(lat, y=yy, init=init)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"  B={Bt:5d}: {dt*1e3:8.3f} ms per 10-update call  ({Bt/dt:10.1f} traj/s)")
