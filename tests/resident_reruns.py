"""Rerun determinism of the trajectory-resident kernel: the same sampling call repeated RERUNS times (default 40) per shape, count of
results that differ from the first one (must be 0).  CASES="[(T, L, C, normalize, depth), ...]" selects shapes, LSL_LIB=<path> another
build of the library.  Usage (GPU box): python tests/resident_reruns.py"""
import sys, os, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lam_slide_amd import _lib
if os.environ.get('LSL_LIB'): _lib.LIB_PATH = os.path.join(ROOT, os.environ['LSL_LIB'])
from lam_slide_amd import CreateTransport, LatentSIV3, Sampler
from oracle import harness, latent_net
dev = torch.device("cuda:0")
def case(T, L, C, norm, depth, B=5):
    kw = dict(depth=depth, in_dim=C, hidden_size=128, num_heads=4, mlp_ratio=2, normalize=norm)
    sh = latent_net.NetShape(**kw); p = latent_net.random_params(sh, seed=31)
    net = LatentSIV3(reset_parameters=False, **kw); net.load_state_dict(p); net.to(dev)
    g = torch.Generator().manual_seed(5)
    lat, init = torch.randn(B, T, L, C, generator=g), torch.randn(B, T, L, C, generator=g)
    xc, m = harness.setup_conditioning(lat, (0, max(1, min(3, T - 1))), True)
    mk = {"x_cond": xc.to(dev), "x_cond_mask": m.to(dev)}
    for ns in (9,):
        skw = {"sampling_method": "euler", "num_steps": ns}
        s = Sampler(CreateTransport("GVP", "data")(), fused=True)
        f = lambda lo, hi: s.get_sample_fn("ODE", skw)(init[lo:hi].to(dev), net.forward, **{k: v[lo:hi] for k, v in mk.items()})[-1]
        ref = f(0, B)
        bad = sum(0 if torch.equal(f(0, B), ref) else 1 for _ in range(int(os.environ.get('RERUNS', '40'))))
        print(f"T{T} L{L} C{C} norm{norm} d{depth} steps {ns-1}: {bad}/" + os.environ.get('RERUNS', '40') + " reruns differ")
import ast
for c in ast.literal_eval(os.environ.get("CASES", "[(5,6,32,False,2),(3,1,8,True,1),(20,2,32,True,6)]")):
    case(*c)
