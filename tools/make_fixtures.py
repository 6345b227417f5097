#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python modules (imported from
/root/reference, build container only) and check the oracle restatement against them.

The reference ships no tests or golden vectors for this path, so these fixtures are what pins the
oracle.  Only data is written (inputs, weights of tiny models, expected outputs); no reference source
text is stored.  Run:  python tools/make_fixtures.py   (needs /root/reference; never runs on the GPU box)

Fixture families:
  f1_block.npz      tiny model: state_dict, inputs, output and per-stage intermediates
  f2_shapes.npz     shape classes (hd 16/24/32, S 2/8/30/64, normalize, share_weights, y): inputs + output
  f3_transport.npz  interval / drift / score / diffusion scalars on a t grid for every path x prediction
  f4_sampler.npz    ODE-euler and SDE (EM, Heun) end-to-end with explicit noise, small model
  f4_cfg1.npz       true cfg-1 shape (B=1, 10 Euler updates), weights from oracle.random_params(seed)
  f5_cond.npz       setup_conditioning, executed from the reference source via ast extraction
  f6_decode.npz     frozen stage-1 decode of latents -> positions (MD17 decoder shape, seeded weights)
  f7_encode.npz     frozen stage-1 Encoder + quant with a ragged entity mask (MD17 encoder shape, 48 latents)
  f8_decode_split.npz  DecoderQuerySplitter (peptide decoder: 1x1-conv latent extender), one latent<-query cross block, tanh GELU
  f11_pedestrian_k.npz  the reference's REAL pedestrian CondWrapper: prepare_batch (class vector y) and the K = 20 test_step loop (ADE / FDE)
  f13_peptide.npz       the reference's REAL peptide Wrapper at T = 1000: encode -> sample -> decode to atom14 positions
  f12_nba_k.npz         the reference's REAL NBA CondWrapper at the NBA shape: prepare_batch and the K = 60 / num_runs = 20 test_step loop
  f9_sample.npz     the reference's REAL LightningModule (second_stage/md17.py Wrapper built by its own __init__ from the reference YAML,
                    lightning_base.py sample / prepare_batch / setup_conditioning unchanged; tools/ref_env.py supplies the Lightning / Hydra
                    stand-ins): stage-1 inputs -> encode -> conditioning -> 5 Euler updates -> decode, with the initial noise fixed
"""
import ast
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# torchdiffeq is not installed and not vendored by the reference; give the import a fixed-grid Euler
# with the library's published semantics (grid == t, states stacked).  Lives only in a temp dir.
_stub = tempfile.mkdtemp()
os.makedirs(os.path.join(_stub, "torchdiffeq"))
with open(os.path.join(_stub, "torchdiffeq", "__init__.py"), "w") as fh:
    fh.write(
        "import torch\n"
        "def odeint(f, y0, t, method=None, atol=None, rtol=None, **kw):\n"
        "    assert method == 'euler'\n"
        "    ys=[y0]; y=y0\n"
        "    for i in range(len(t)-1):\n"
        "        y = y + (t[i+1]-t[i]) * f(t[i], y)\n"
        "        ys.append(y)\n"
        "    return torch.stack(ys)\n"
    )
sys.path.insert(0, _stub)

from src.models.components.latent import mmdit as ref_mmdit  # noqa: E402
from src.models.components.latent.latent_si_v31 import LatentSIV3  # noqa: E402
from src.modules.transport import CreateTransport  # noqa: E402
from src.modules.transport.transport import Sampler  # noqa: E402
from src.models.components.decoder import Decoder  # noqa: E402
from src.modules.entity_embeddings import EntityEmbeddingOrthogonal  # noqa: E402
from src.modules.torch_modules import GELU as RefGELU  # noqa: E402

from oracle import harness, latent_net, transport as otr  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)


def npz(name, **arrays):
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = vv.detach().cpu().numpy() if torch.is_tensor(vv) else np.asarray(vv)
        else:
            flat[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **flat)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.0f} KiB, {len(flat)} arrays")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def make_ref(sh: latent_net.NetShape, seed=0):
    torch.manual_seed(seed)
    m = LatentSIV3(depth=sh.depth, in_dim=sh.in_dim, hidden_size=sh.hidden_size, num_heads=sh.num_heads,
                   vec_in_dim=sh.vec_in_dim, mlp_ratio=sh.mlp_ratio, theta=sh.theta, normalize=sh.normalize,
                   share_weights=sh.share_weights, reset_parameters=False).eval()
    # make the QK-norm scales non-trivial so their channel mapping is pinned too
    g = torch.Generator().manual_seed(seed + 100)
    for n, p_ in m.named_parameters():
        if n.endswith("_norm.scale"):
            p_.copy_(1.0 + 0.2 * torch.randn(p_.shape, generator=g))
    return m


def make_inputs(sh, B, T, L, seed, cond_frames=2):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, L, sh.in_dim, generator=g)
    lat = torch.randn(B, T, L, sh.in_dim, generator=g)
    t = torch.rand(B, generator=g) * 0.98 + 0.01
    x_cond, mask = harness.setup_conditioning(lat, (0, cond_frames), True)
    y = torch.randn(B, sh.vec_in_dim, generator=g) if sh.vec_in_dim else None
    return x, t, x_cond, mask, y


def shape_dict(sh):
    return dict(depth=sh.depth, in_dim=sh.in_dim, hidden_size=sh.hidden_size, num_heads=sh.num_heads,
                mlp_ratio=sh.mlp_ratio, vec_in_dim=-1 if sh.vec_in_dim is None else sh.vec_in_dim,
                theta=sh.theta, normalize=int(sh.normalize), share_weights=int(sh.share_weights))


# ------------------------------------------------------------------------------------------- F1
def f1():
    sh = latent_net.NetShape(depth=2, in_dim=8, hidden_size=64, num_heads=4, mlp_ratio=2, vec_in_dim=16)
    m = make_ref(sh, 0)
    x, t, xc, mask, y = make_inputs(sh, 2, 5, 6, 1)
    taps = {}
    calls = {"rope": 0, "attn": 0}
    names = [f"l{i}.{b}." for i in range(sh.depth) for b in ("sp", "tm")]
    orig_rope, orig_attn = ref_mmdit.apply_rope, ref_mmdit.attention

    def rope_spy(q, k, pe):
        tag = names[calls["rope"]]
        calls["rope"] += 1
        taps[tag + "q_norm"], taps[tag + "k_norm"] = q.clone(), k.clone()
        qo, ko = orig_rope(q, k, pe)
        taps[tag + "q_rope"], taps[tag + "k_rope"] = qo.clone(), ko.clone()
        return qo, ko

    def attn_spy(q, k, v, pe=None, mode="scaled_dot_product"):
        tag = names[calls["attn"]]
        calls["attn"] += 1
        o = orig_attn(q, k, v, pe=pe, mode=mode)
        taps[tag + "attn"] = o.clone()
        return o

    ref_mmdit.apply_rope, ref_mmdit.attention = rope_spy, attn_spy
    hooks = []
    for i, blk in enumerate(m.blocks):
        hooks.append(blk.modulation.lin.register_forward_hook(lambda _m, _i, o, i=i: taps.__setitem__(f"l{i}.mod", o.clone())))
        hooks.append(blk.register_forward_hook(lambda _m, _i, o, i=i: taps.__setitem__(f"l{i}.h", o.clone())))
        for b, sub in (("sp", blk.spatial_block), ("tm", blk.temporal_block)):
            hooks.append(sub.linear1.register_forward_hook(lambda _m, _i, o, i=i, b=b: taps.__setitem__(f"l{i}.{b}.z", o.clone())))
            hooks.append(sub.register_forward_hook(lambda _m, _i, o, i=i, b=b: taps.__setitem__(f"l{i}.{b}.out", o.clone())))
    hooks.append(m.adaLN_modulation.register_forward_hook(lambda _m, _i, o: taps.__setitem__("final_mod", o.clone())))
    out = m(x, t, xc, mask, y)
    for h in hooks:
        h.remove()
    ref_mmdit.apply_rope, ref_mmdit.attention = orig_rope, orig_attn
    taps["vec"] = m.time_in(ref_mmdit.timestep_embedding(t, 256)) + m.vec_in(y)
    taps["out"] = out

    sd = {k: v for k, v in m.state_dict().items()}
    o_taps = {}
    o_out = latent_net.forward(sd, sh, x, t, xc, mask, y, taps=o_taps)
    worst = rel(o_out, out)
    for k, v in taps.items():
        e = rel(o_taps[k].reshape(v.shape), v)
        worst = max(worst, e)
        assert e < 2e-6, (k, e)
    # fp64 agreement of the restatement with the reference run in fp64
    m64 = make_ref(sh, 0).double()
    out64 = m64(x.double(), t.double(), xc.double(), mask, y.double())
    e64 = rel(latent_net.forward(latent_net.cast_params(sd, torch.float64), sh, x.double(), t.double(), xc.double(), mask, y.double()), out64)
    print(f"F1 oracle-vs-reference worst rel (fp32) {worst:.2e}; fp64 {e64:.2e}")
    assert e64 < 1e-6  # RoPE table and time features stay fp32 in both
    npz("f1_block.npz", shape=shape_dict(sh), p=sd, x=x, t=t, x_cond=xc, mask=mask, y=y, taps=taps, out64=out64)


# ------------------------------------------------------------------------------------------- F2
F2_CASES = {
    # name: (NetShape kwargs, B, T, L, weight seed).  Weights are oracle.random_params(shape, seed): not stored.
    "hd16_s8x30": (dict(depth=2, in_dim=8, hidden_size=64, num_heads=4, mlp_ratio=2), 2, 30, 8, 21),
    "hd24_s2x64": (dict(depth=1, in_dim=12, hidden_size=192, num_heads=8, mlp_ratio=1), 1, 64, 2, 22),
    "hd32_s64x5_norm_y": (dict(depth=1, in_dim=16, hidden_size=128, num_heads=4, mlp_ratio=2, vec_in_dim=32, normalize=True), 2, 5, 64, 23),
    "hd16_share": (dict(depth=3, in_dim=8, hidden_size=64, num_heads=4, mlp_ratio=2, share_weights=True), 1, 6, 10, 24),
    "hd32_s40x33": (dict(depth=1, in_dim=32, hidden_size=128, num_heads=4, mlp_ratio=2), 1, 33, 40, 25),
}


def load_ref(sh, sd):
    m = LatentSIV3(depth=sh.depth, in_dim=sh.in_dim, hidden_size=sh.hidden_size, num_heads=sh.num_heads,
                   vec_in_dim=sh.vec_in_dim, mlp_ratio=sh.mlp_ratio, theta=sh.theta, normalize=sh.normalize,
                   share_weights=sh.share_weights, reset_parameters=False).eval()
    m.load_state_dict(sd)
    return m


def f2():
    arrays = {}
    for name, (kw, B, T, L, wseed) in F2_CASES.items():
        sh = latent_net.NetShape(**kw)
        sd = latent_net.random_params(sh, seed=wseed)
        m = load_ref(sh, sd)
        x, t, xc, mask, y = make_inputs(sh, B, T, L, 11)
        out = m(x, t, xc, mask, y)
        e = rel(latent_net.forward(sd, sh, x, t, xc, mask, y), out)
        print(f"F2 {name}: oracle rel {e:.2e}")
        assert e < 2e-6
        arrays[name] = dict(**{"shape." + k: v for k, v in shape_dict(sh).items()}, weight_seed=wseed,
                            x=x, t=t, x_cond=xc, mask=mask, out=out, **({"y": y} if y is not None else {}))
    npz("f2_shapes.npz", **arrays)


# ------------------------------------------------------------------------------------------- F3
def f3():
    arrays = {}
    tg = torch.linspace(0.02, 0.98, 13)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(13, 3, 4, generator=g)
    mo = torch.randn(13, 3, 4, generator=g)
    worst = 0.0
    for path in otr.PATHS:
        for pred in otr.PREDICTIONS:
            rt = CreateTransport(path_type=path, prediction=pred)()
            ot = otr.Transport(path, pred)
            key = f"{path}.{pred}"
            iv = []
            for sde in (False, True):
                for form in ("SBDM", "linear"):
                    for ls in (0.0, 0.04):
                        a = rt.check_interval(rt.train_eps, rt.sample_eps, diffusion_form=form, sde=sde, eval=True, reverse=False, last_step_size=ls)
                        b = ot.interval(diffusion_form=form, sde=sde, last_step_size=ls)
                        assert tuple(map(float, a)) == tuple(map(float, b)), (key, sde, form, ls, a, b)
                        iv.append([float(a[0]), float(a[1])])
            smp = Sampler(rt)
            model = lambda xx, tt, **kw: mo  # noqa: E731
            v = smp.drift(x, tg, model)
            s = smp.score(x, tg, model)
            ov, os_ = ot.velocity(x, tg, model), ot.score(x, tg, model)
            worst = max(worst, rel(ov, v), rel(os_, s))
            arrays[key] = dict(intervals=np.array(iv), velocity=v, score=s)
            for form in ("constant", "SBDM", "sigma", "linear", "decreasing", "inccreasing-decreasing"):
                d = rt.path_sampler.compute_diffusion(x, tg, form=form, norm=0.7)
                od = ot.plan.diffusion(x, tg, form, 0.7)
                d = torch.as_tensor(d) * torch.ones(13, 1, 1)
                od = torch.as_tensor(od) * torch.ones(13, 1, 1)
                worst = max(worst, rel(od, d))
                arrays[key]["diff." + form] = d
    assert worst < 1e-6, worst
    print(f"F3 transport oracle-vs-reference worst rel {worst:.2e}")
    npz("f3_transport.npz", t=tg, x=x, model_out=mo, **arrays)


# ------------------------------------------------------------------------------------------- F4
def ref_sample(m, rt, init, xc, mask, y, method, kw, noise=None):
    fn = Sampler(rt).get_sample_fn(method, kw)
    model = lambda xt, t, **mk: m(x=xt, t=t, **mk)  # noqa: E731  (lightning_base.py:173-174)
    mk = dict(x_cond=xc, x_cond_mask=mask)
    if y is not None:
        mk["y"] = y
    if noise is not None:
        it = iter(noise)
        orig = torch.randn
        import src.modules.transport.integrators as integ
        integ.th.randn = lambda *a, **k: next(it).clone()
        try:
            out = fn(init, model, **mk)
        finally:
            integ.th.randn = orig
        return out
    return fn(init, model, **mk)


def f4():
    sh = latent_net.NetShape(depth=2, in_dim=8, hidden_size=64, num_heads=4, mlp_ratio=2)
    m = make_ref(sh, 3)
    sd = dict(m.state_dict())
    B, T, L = 2, 6, 16
    g = torch.Generator().manual_seed(2)
    lat = torch.randn(B, T, L, sh.in_dim, generator=g)
    xc, mask = harness.setup_conditioning(lat, (0, 2), True)
    init = torch.randn(B, T, L, sh.in_dim, generator=g)
    arrays = dict(shape=shape_dict(sh), p=sd, init=init, x_cond=xc, mask=mask)
    rt = CreateTransport(path_type="GVP", prediction="data")()
    ot = otr.Transport("GVP", "data")
    for n in (2, 11, 51):
        ref = ref_sample(m, rt, init, xc, mask, None, "ODE", {"sampling_method": "euler", "num_steps": n})
        assert ref.shape[0] == n
        mine = harness.sample_latents(sd, sh, ot, init, xc, mask, None, "ODE", {"sampling_method": "euler", "num_steps": n})
        e = rel(mine, ref[-1])
        print(f"F4 ode n={n}: oracle rel {e:.2e}")
        assert e < 5e-5, e  # late steps amplify fp32 rounding (gain pi/2/cos)
        arrays[f"ode{n}"] = ref[-1]
    # other path / prediction combinations through the same network (treated as the stated prediction)
    for path, pred in (("Linear", "velocity"), ("Linear", "data"), ("VP", "noise"), ("GVP", "score")):
        rt2 = CreateTransport(path_type=path, prediction=pred)()
        ref = ref_sample(m, rt2, init, xc, mask, None, "ODE", {"sampling_method": "euler", "num_steps": 6})
        mine = harness.sample_latents(sd, sh, otr.Transport(path, pred), init, xc, mask, None, "ODE", {"sampling_method": "euler", "num_steps": 6})
        e = rel(mine, ref[-1])
        print(f"F4 ode {path}/{pred}: oracle rel {e:.2e}")
        assert e < 5e-5, e
        arrays[f"ode6.{path}.{pred}"] = ref[-1]
    for n, form, last, meth in ((3, "linear", "Mean", "Euler"), (10, "linear", "Mean", "Euler"), (10, "SBDM", None, "Euler"),
                                (6, "sigma", "Euler", "Euler"), (5, "linear", "Mean", "Heun"), (6, "decreasing", "Tweedie", "Euler")):
        gn = torch.Generator().manual_seed(3 + n)
        noise = [torch.randn(B, T, L, sh.in_dim, generator=gn) for _ in range(n - 1)]
        kw = {"sampling_method": meth, "diffusion_form": form, "last_step": last, "num_steps": n}
        ref = ref_sample(m, rt, init, xc, mask, None, "SDE", kw, noise=noise)
        assert len(ref) == n
        fn = otr.get_sample_fn(ot, "SDE", kw, noise=noise)
        model = lambda xt, t, **mk: latent_net.forward(sd, sh, xt, t, **mk)  # noqa: E731
        mine = fn(init, model, x_cond=xc, x_cond_mask=mask)
        e = max(rel(a, b) for a, b in zip(mine, ref))
        fn1 = otr.get_sample_fn(ot, "SDE", kw, noise=noise, single_eval=True)
        e1 = rel(fn1(init, model, x_cond=xc, x_cond_mask=mask)[-1], ref[-1])
        print(f"F4 sde n={n} {form} {last} {meth}: oracle rel {e:.2e} (single-eval {e1:.2e})")
        assert e < 5e-5 and e1 < 5e-5
        tag = f"sde{n}.{form}.{last}.{meth}"
        arrays[tag + ".noise"] = torch.stack(noise)
        arrays[tag + ".final"] = ref[-1]
        arrays[tag + ".penultimate"] = ref[-2]
    npz("f4_sampler.npz", **arrays)

    # true cfg-1 shape, B=1; weights are NOT stored: they are oracle.random_params(shape, seed=0)
    sh1 = latent_net.NetShape(depth=4, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=2)
    sd1 = latent_net.random_params(sh1, seed=0)
    m1 = load_ref(sh1, sd1)
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(1, 30, 192, 32, generator=g)
    xc, mask = harness.setup_conditioning(lat, (0, 10), True)
    init = torch.randn(1, 30, 192, 32, generator=torch.Generator().manual_seed(2))
    ref = ref_sample(m1, rt, init, xc, mask, None, "ODE", {"sampling_method": "euler", "num_steps": 11})[-1]
    mine = harness.sample_latents(sd1, sh1, ot, init, xc, mask, None, "ODE", {"sampling_method": "euler", "num_steps": 11})
    e = rel(mine, ref)
    print(f"F4 cfg1 B=1: oracle rel {e:.2e}")
    assert e < 5e-5
    npz("f4_cfg1.npz", shape=shape_dict(sh1), weight_seed=0, latent_seed=1, init_seed=2, cond_idx=np.array([0, 10]),
        num_steps=11, final=ref)


# ------------------------------------------------------------------------------------------- F5
def f5():
    """Run the reference's own setup_conditioning source (lightning_base.py:240-263) without importing
    lightning: extract the function with ast and bind it to a stand-in ``self``."""
    src = open(os.path.join(REF, "src/models/composites/lightning_base.py")).read()
    tree = ast.parse(src)
    fn_node = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "setup_conditioning":
            fn_node = node
    fn_node.decorator_list = []
    mod = ast.Module(body=[fn_node], type_ignores=[])
    ns = {"torch": torch, "Tensor": torch.Tensor, "Tuple": tuple}
    exec(compile(mod, "<ref:setup_conditioning>", "exec"), ns)
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(3, 7, 4, 5, generator=g)
    arrays = dict(latents=lat)
    for mean in (True, False):
        for ci in ((0, 3), (0, 1), (2, 5)):
            self_ = types.SimpleNamespace(device="cpu", hparams=types.SimpleNamespace(cond_idx=list(ci), mask_cond_mean=mean))
            xc, mask = ns["setup_conditioning"](self_, lat)
            oxc, omask = harness.setup_conditioning(lat, ci, mean)
            assert torch.equal(xc, oxc) and torch.equal(mask, omask)
            arrays[f"m{int(mean)}.{ci[0]}_{ci[1]}.x_cond"] = xc
            arrays[f"m{int(mean)}.{ci[0]}_{ci[1]}.mask"] = mask
    print("F5 conditioning: oracle bit-equal to reference source")
    npz("f5_cond.npz", **arrays)


# ------------------------------------------------------------------------------------------- F6
def f6():
    from functools import partial
    ds = harness.DecoderShape()
    torch.manual_seed(5)
    emb = EntityEmbeddingOrthogonal(n_entiy_embeddings=ds.n_entities, embedding_dim=128, max_norm=1)
    dec = Decoder(outputs={"pos": 3, "atom": 10}, dim_query=128, dim_latent=32, entity_embedding=emb, dim_head_cross=16,
                  dim_head_latent=16, num_head_cross=8, num_head_latent=2, num_block_cross=0, num_block_attn=1,
                  dropout_query=0.1, qk_norm=True, act=partial(RefGELU)).eval()
    post_quant = torch.nn.Sequential(torch.nn.LayerNorm(32, elementwise_affine=False), torch.nn.Linear(32, 32)).eval()
    # give the entity table some rows above unit norm so the max_norm path is exercised
    emb.embedding.weight.mul_(torch.linspace(0.5, 1.8, ds.n_entities)[:, None])
    p = {"post_quant.1.weight": post_quant[1].weight.clone(), "post_quant.1.bias": post_quant[1].bias.clone()}
    p.update({"decoder." + k: v.clone() for k, v in dec.state_dict().items() if "output_layers.atom" not in k})
    g = torch.Generator().manual_seed(6)
    z = torch.randn(4, 192, 32, generator=g)
    ent = torch.stack([torch.randperm(ds.n_entities, generator=g)[:21] for _ in range(4)])
    pos = dec(post_quant(z), ent)["pos"]
    mine = harness.decode(p, ds, z, ent)
    e = rel(mine, pos)
    print(f"F6 decode: oracle rel {e:.2e}; params {sum(v.numel() for v in p.values())}")
    assert e < 2e-6
    npz("f6_decode.npz", p=p, z=z, entities=ent, pos=pos)


# ------------------------------------------------------------------------------------------- F7
def f7():
    from functools import partial
    from src.models.components.encoder import Encoder
    es = harness.EncoderShape(num_latents=48)
    torch.manual_seed(9)
    emb = EntityEmbeddingOrthogonal(n_entiy_embeddings=32, embedding_dim=128, max_norm=1)
    enc = Encoder(dim_input=es.dim_input, dim_latent=es.dim_latent, dim_head_cross=16, dim_head_latent=16, num_latents=es.num_latents,
                  num_head_cross=8, num_head_latent=2, num_block_cross=1, num_block_attn=1, qk_norm=True, entity_embedding=emb,
                  act=partial(RefGELU)).eval()
    quant = torch.nn.Sequential(torch.nn.Linear(32, 32), torch.nn.LayerNorm(32, elementwise_affine=False)).eval()
    emb.embedding.weight.mul_(torch.linspace(0.5, 1.8, 32)[:, None])
    p = {"quant.0.weight": quant[0].weight.clone(), "quant.0.bias": quant[0].bias.clone()}
    p.update({"encoder." + k: v.clone() for k, v in enc.state_dict().items()})
    g = torch.Generator().manual_seed(10)
    x = torch.randn(5, 21, es.dim_input, generator=g)
    ent = torch.stack([torch.randperm(32, generator=g)[:21] for _ in range(5)])
    mask = torch.ones(5, 21, dtype=torch.bool)
    mask[1, 15:] = False   # ragged systems: padded entities are masked out of the cross-attention
    mask[3, 9:] = False
    with torch.no_grad():
        z = quant(enc(x=x, entities=ent, mask=mask))
    mine = harness.encode(p, es, x, ent, mask)
    e = rel(mine, z)
    print(f"F7 encode: oracle rel {e:.2e}; params {sum(v.numel() for v in p.values())}")
    assert e < 2e-6
    npz("f7_encode.npz", p=p, x=x, entities=ent, mask=mask, z=z)


# ------------------------------------------------------------------------------------------- F8
def f8():
    from functools import partial
    from src.models.components.decoder import DecoderQuerySplitter
    ds = harness.DecoderShape(num_block_cross=1)
    torch.manual_seed(11)
    emb = EntityEmbeddingOrthogonal(n_entiy_embeddings=ds.n_entities, embedding_dim=128, max_norm=1)
    dec = DecoderQuerySplitter(outputs={"pos": 3}, dim_query=128, dim_latent=32, entity_embedding=emb, dim_head_cross=16, dim_head_latent=16,
                               num_head_cross=8, num_head_latent=2, num_block_cross=1, num_block_attn=1, qk_norm=True,
                               act=partial(torch.nn.GELU, approximate="tanh"), num_split=4).eval()
    post_quant = torch.nn.Sequential(torch.nn.LayerNorm(32, elementwise_affine=False), torch.nn.Linear(32, 32)).eval()
    emb.embedding.weight.mul_(torch.linspace(0.5, 1.8, ds.n_entities)[:, None])
    p = {"post_quant.1.weight": post_quant[1].weight.clone(), "post_quant.1.bias": post_quant[1].bias.clone()}
    p.update({"decoder." + k: v.clone() for k, v in dec.state_dict().items()})
    g = torch.Generator().manual_seed(12)
    z = torch.randn(3, 24, 32, generator=g)
    ent = torch.stack([torch.randperm(ds.n_entities, generator=g)[:14] for _ in range(3)])
    with torch.no_grad():
        pos = dec(post_quant(z), ent)["pos"]
    mine = harness.decode(p, harness.DecoderShape(num_block_cross=1, act="gelu_tanh"), z, ent)
    e = rel(mine, pos)
    print(f"F8 decode (query splitter, cross block, tanh GELU): oracle rel {e:.2e}")
    assert e < 2e-6
    npz("f8_decode_split.npz", p=p, z=z, entities=ent, pos=pos)


# ------------------------------------------------------------------------------------------- F9
def f9():
    """SecondStageCondLightningBase.sample (lightning_base.py:217-238) executed UNCHANGED on the reference's real second-stage Wrapper.
    Run twice: with the reference's own Sampler, and after lam_slide_amd.install() (the module-level Sampler rebound: this package's
    Sampler then steps the reference backbone through its generic loop) - the two must agree; the first is the fixture."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_env
    import lam_slide_amd
    from lam_slide_amd import dropin
    dropin.uninstall()
    ns = ref_env.setup()
    F = ref_env.F9
    first, first_cls = ref_env.build_first_stage(ns)
    w = ref_env.build_wrapper(ns, "src.models.components.latent.latent_si_v31.LatentSIV3", "src.modules.transport.CreateTransport", first, first_cls)
    w.eval()
    assert ns.lightning_base.Sampler is ns.transport.Sampler and ns.lightning_base.Sampler is not lam_slide_amd.Sampler
    batch = ref_env.f9_batch()
    noise = torch.randn(F["B"], F["T"], F["L"], 32, generator=torch.Generator().manual_seed(24))
    seen = {}
    real_decode = w.decode

    def tap(latents, entities):
        seen["final"] = latents.clone()
        return real_decode(latents, entities)

    w.decode = tap
    with ref_env.fixed_randn_like(noise):
        ref_pos = w.sample(dict(batch))["pos"]
    final_ref = seen["final"]
    enc_lat = w.encode(dict(batch))
    xc, mask = w.setup_conditioning(enc_lat)
    # the same call with this package's Sampler behind the reference's module-level name
    done = dropin.install(force=True)
    assert "src.models.composites.lightning_base.Sampler" in done and ns.lightning_base.Sampler is lam_slide_amd.Sampler
    with ref_env.fixed_randn_like(noise):
        mine_pos = w.sample(dict(batch))["pos"]
    dropin.uninstall()
    e_swap = rel(mine_pos, ref_pos)
    # oracle chain on the same inputs
    sd = {k: v.clone() for k, v in w.backbone.state_dict().items()}
    s1 = {k: v.clone() for k, v in first.backbone.state_dict().items()}
    flat = lambda t: t.reshape(F["B"] * F["T"], *t.shape[2:])  # noqa: E731
    es = harness.EncoderShape(num_latents=F["L"])
    o_lat = harness.encode(s1, es, flat(batch["pos"]), flat(batch["entities"]), flat(batch["attention_mask"])).reshape(F["B"], F["T"], F["L"], 32)
    oxc, omask = harness.setup_conditioning(o_lat, tuple(F["cond_idx"]), True)
    sh = latent_net.NetShape(**F["backbone"])
    o_final = harness.sample_latents(sd, sh, otr.Transport("GVP", "data"), noise, oxc, omask, None, "ODE", {"sampling_method": "euler", "num_steps": F["num_steps"]})
    o_pos = harness.decode(s1, harness.DecoderShape(), flat(o_final), flat(batch["entities"])).reshape(ref_pos.shape)
    print(f"F9 real LightningModule: install() swap rel {e_swap:.2e}; oracle encode {rel(o_lat, enc_lat):.2e} latents {rel(o_final, final_ref.reshape(o_final.shape)):.2e} "
          f"positions {rel(o_pos, ref_pos):.2e}")
    assert e_swap < 2e-6 and rel(o_pos, ref_pos) < 1e-5 and torch.equal(omask, mask)
    npz("f9_sample.npz", backbone=sd, stage1=s1, x=batch["pos"], entities=batch["entities"], attention_mask=batch["attention_mask"], noise=noise,
        latents=enc_lat, x_cond=xc, mask=mask, final=final_ref.reshape(o_final.shape), pos=ref_pos,
        meta=np.array([F["B"], F["T"], F["A"], F["L"], F["cond_idx"][0], F["cond_idx"][1], F["num_steps"]]))


# ------------------------------------------------------------------------------------------- F10
F10_CASES = {
    # attention_mode="linear" (mmdit.py:50-53, 58-72; no shipped config selects it).  name: (NetShape kwargs, B, T, L, weight seed)
    "lin_hd16_s8x70": (dict(depth=2, in_dim=8, hidden_size=64, num_heads=4, mlp_ratio=2), 2, 70, 8, 31),
    "lin_hd24_s2x33": (dict(depth=1, in_dim=12, hidden_size=192, num_heads=8, mlp_ratio=1), 1, 33, 2, 32),
    "lin_hd32_s300x3_norm_y": (dict(depth=1, in_dim=16, hidden_size=128, num_heads=4, mlp_ratio=2, vec_in_dim=32, normalize=True), 2, 3, 300, 33),
}


def f10():
    arrays = {}
    for name, (kw, B, T, L, wseed) in F10_CASES.items():
        sh = latent_net.NetShape(**kw, attention_mode="linear")
        sd = latent_net.random_params(sh, seed=wseed)
        m = LatentSIV3(depth=sh.depth, in_dim=sh.in_dim, hidden_size=sh.hidden_size, num_heads=sh.num_heads, vec_in_dim=sh.vec_in_dim,
                       mlp_ratio=sh.mlp_ratio, theta=sh.theta, normalize=sh.normalize, attention_mode="linear", reset_parameters=False).eval()
        m.load_state_dict(sd)
        x, t, xc, mask, y = make_inputs(sh, B, T, L, 12)
        with torch.no_grad():
            out = m(x, t, xc, mask, y)
        e = rel(latent_net.forward(sd, sh, x, t, xc, mask, y), out)
        sdpa = rel(latent_net.forward(sd, latent_net.NetShape(**kw), x, t, xc, mask, y), out)
        print(f"F10 {name}: oracle rel {e:.2e} (the softmax-attention oracle on the same weights: {sdpa:.2e})")
        assert e < 2e-6 and sdpa > 1e-3
        arrays[name] = dict(**{"shape." + k: v for k, v in shape_dict(sh).items()}, weight_seed=wseed,
                            x=x, t=t, x_cond=xc, mask=mask, out=out, **({"y": y} if y is not None else {}))
    npz("f10_linear_attention.npz", **arrays)


# ------------------------------------------------------------------------------------------- F11
def f11():
    """The conditioned caller, REAL files: second_stage/pedestrian.py CondWrapper built by its own __init__ from the reference YAMLs;
    `CondWrapper.prepare_batch` (:242-251: y = Embedding(cond_scene)) and the K = 20 `test_step` loop (:186-212: K sequential sample()
    calls, future frames, real agents only, best-of-K ADE / FDE) executed UNCHANGED, each sample() drawing its initial state from a stored
    noise.  Backbone weights = oracle.random_params(seed) loaded into the reference module (only the seed is stored)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_env
    from lam_slide_amd import dropin
    dropin.uninstall()
    ns = ref_env.setup()
    F = ref_env.F11
    B, T, A, L, K = F["B"], F["T"], F["A"], F["L"], F["K"]
    lift = torch.randn(3, 128, generator=torch.Generator().manual_seed(30)) * 0.5
    first, first_cls = ref_env.build_first_stage(ns, seed=31, num_latents=L, lift=lift)
    w = ref_env.build_pedestrian_wrapper(ns, first, first_cls)
    w.eval()
    sh = latent_net.NetShape(**F["backbone"])
    wseed = 34
    w.backbone.load_state_dict(latent_net.random_params(sh, seed=wseed))
    batch = ref_env.f11_batch()
    g = torch.Generator().manual_seed(35)
    batch["pos"] = torch.randn(B, T, A, 3, generator=g)  # positions; the first stage lifts them (ref_env.build_first_stage)
    true_future = batch["pos"][:, F["cond_idx"][1]:].clone()
    noises = torch.randn(K, B, T, L, 32, generator=g)
    # prepare_batch on its own (the conditioning the sampler receives)
    pb = w.prepare_batch({k: v.clone() for k, v in batch.items()})
    y, xc, mask, lat = pb["model_kwargs"]["y"], pb["model_kwargs"]["x_cond"], pb["model_kwargs"]["x_cond_mask"], pb["x1"]
    assert torch.equal(y, w.vec_in_embedding.weight[batch["cond_scene"]])
    # the K-sample evaluation loop
    finals, poss = [], []
    real_decode = w.decode

    def tap(latents, entities):
        # As committed, the reference's test_step cannot run on its own decode(): decode() returns positions as [B, T, A, D]
        # (pedestrian.py:140-143, nba.py:145-148) and test_step rearranges "(B T) L D -> B T L D" again (pedestrian.py:194, nba.py:207),
        # which einops refuses for a 4-D tensor.  The loop is executed unchanged around a decode() that hands it the layout it asks for.
        out = real_decode(latents, entities)
        finals.append(latents.clone().reshape(B, T, L, 32))
        poss.append(out["pos"].clone())
        return {"pos": out["pos"].reshape(B * T, A, -1)}

    w.decode = tap
    w.on_test_epoch_start()
    tb = {k: v.clone() for k, v in batch.items()}
    with ref_env.randn_like_sequence(noises) as seq:
        w.test_step(tb, 0)
    assert seq.i == K and len(finals) == K
    ades, fdes = torch.cat(w.test_step_outputs["eth"]["ades"]), torch.cat(w.test_step_outputs["eth"]["fdes"])
    finals, poss = torch.stack(finals), torch.stack(poss)  # [K,B,T,L,32], [K,B,T,A,3]
    # the oracle chain on the same inputs
    s1 = {k: v.clone() for k, v in first.backbone.state_dict().items()}
    sd = latent_net.random_params(sh, seed=wseed)
    flat = lambda t: t.reshape(B * T, *t.shape[2:])  # noqa: E731
    zeroed = batch["pos"].clone()
    zeroed[:, F["cond_idx"][1]:] = 0  # (test_step zeroes the future frames before sampling: pedestrian.py:175-176)
    o_lat = harness.encode(s1, harness.EncoderShape(num_latents=L), flat(zeroed @ lift), flat(batch["entities"]), flat(batch["attention_mask"])).reshape(B, T, L, 32)
    oxc, omask = harness.setup_conditioning(o_lat, tuple(F["cond_idx"]), True)
    o_final = torch.stack([harness.sample_latents(sd, sh, otr.Transport("GVP", "data"), noises[k], oxc, omask, y, "ODE",
                                                  {"sampling_method": "euler", "num_steps": F["num_steps"]}) for k in range(K)])
    o_pos = harness.decode(s1, harness.DecoderShape(), o_final.reshape(K * B * T, L, 32), flat(batch["entities"]).repeat(K, 1)).reshape(poss.shape)
    print(f"F11 real pedestrian CondWrapper: conditioning {rel(oxc[:, :F['cond_idx'][1]], xc[:, :F['cond_idx'][1]]):.2e} finals {rel(o_final, finals):.2e} positions {rel(o_pos, poss):.2e}; "
          f"ADE {ades.mean():.4f} FDE {fdes.mean():.4f} over {ades.numel()} agents")
    assert rel(o_final, finals) < 1e-5 and rel(o_pos, poss) < 1e-5 and torch.equal(omask, mask)
    npz("f11_pedestrian_k.npz", stage1=s1, lift=lift, weight_seed=np.array(wseed), embedding=w.vec_in_embedding.weight.detach().clone(),
        pos=batch["pos"], entities=batch["entities"], attention_mask=batch["attention_mask"], cond_scene=batch["cond_scene"], noises=noises,
        y=y, x_cond=xc, mask=mask, finals=finals, positions=poss, ades=ades, fdes=fdes, true_future=true_future,
        meta=np.array([B, T, A, L, K, F["cond_idx"][0], F["cond_idx"][1], F["num_steps"]]), shape=shape_dict(sh))


def f12():
    """The NBA conditioned caller, REAL files: second_stage/nba.py CondWrapper built by its own __init__ from the reference YAMLs (class
    defaults K = 60, num_runs = 20); `CondWrapper.prepare_batch` (:254-263) and the `test_step` loop (:205-225: K sequential sample() calls,
    best-of-the-first-num_runs ADE / FDE over the real agents) executed UNCHANGED at the NBA shape (T = 20, L = 8, hidden 256, 16 heads,
    mlp 4, class vector; depth 2).  The K initial noises are torch.randn of a stored seed (not stored themselves); of the K final states
    and decoded positions the fixture keeps samples 0, num_runs - 1 and K - 1, of the errors everything."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_env
    from lam_slide_amd import dropin
    dropin.uninstall()
    ns = ref_env.setup()
    F = ref_env.F12
    B, T, A, L, K, R = F["B"], F["T"], F["A"], F["L"], F["K"], F["num_runs"]
    lift = torch.randn(3, 128, generator=torch.Generator().manual_seed(40)) * 0.5
    first, first_cls = ref_env.build_first_stage(ns, seed=41, num_latents=L, lift=lift)
    w = ref_env.build_nba_wrapper(ns, first, first_cls)
    w.eval()
    sh = latent_net.NetShape(**F["backbone"])
    wseed, nseed = 44, 45
    w.backbone.load_state_dict(latent_net.random_params(sh, seed=wseed))
    batch = ref_env.f12_batch()
    true_future = batch["pos"][:, F["cond_idx"][1]:].clone()
    noises = torch.randn(K, B, T, L, 32, generator=torch.Generator().manual_seed(nseed))
    pb = w.prepare_batch({k: v.clone() for k, v in batch.items()})
    y, xc, mask = pb["model_kwargs"]["y"], pb["model_kwargs"]["x_cond"], pb["model_kwargs"]["x_cond_mask"]
    assert torch.equal(y, w.vec_in_embedding.weight[batch["cond_scene"]])
    finals, poss = [], []
    real_decode = w.decode

    def tap(latents, entities):  # (the same layout hand-over as F11: decode() returns [B, T, A, D], test_step rearranges "(B T) L D": nba.py:145-148, :207)
        out = real_decode(latents, entities)
        finals.append(latents.clone().reshape(B, T, L, 32))
        poss.append(out["pos"].clone())
        return {"pos": out["pos"].reshape(B * T, A, -1)}

    w.decode = tap
    w.on_test_epoch_start()
    with ref_env.randn_like_sequence(noises) as seq:
        w.test_step({k: v.clone() for k, v in batch.items()}, 0)
    assert seq.i == K and len(finals) == K
    ades, fdes = torch.cat(w.test_step_outputs["score"]["ades"]), torch.cat(w.test_step_outputs["score"]["fdes"])
    finals, poss = torch.stack(finals), torch.stack(poss)
    s1 = {k: v.clone() for k, v in first.backbone.state_dict().items()}
    sd = latent_net.random_params(sh, seed=wseed)
    flat = lambda t: t.reshape(B * T, *t.shape[2:])  # noqa: E731
    zeroed = batch["pos"].clone()
    zeroed[:, F["cond_idx"][1]:] = 0  # (test_step zeroes the future frames before sampling: nba.py:188-189)
    o_lat = harness.encode(s1, harness.EncoderShape(num_latents=L), flat(zeroed @ lift), flat(batch["entities"]), flat(batch["attention_mask"])).reshape(B, T, L, 32)
    oxc, omask = harness.setup_conditioning(o_lat, tuple(F["cond_idx"]), True)
    keep_k = [0, R - 1, K - 1]
    o_final = torch.stack([harness.sample_latents(sd, sh, otr.Transport("GVP", "data"), noises[k], oxc, omask, y, "ODE",
                                                  {"sampling_method": "euler", "num_steps": F["num_steps"]}) for k in keep_k])
    o_pos = harness.decode(s1, harness.DecoderShape(), o_final.reshape(len(keep_k) * B * T, L, 32), flat(batch["entities"]).repeat(len(keep_k), 1)).reshape(len(keep_k), B, T, A, 3)
    print(f"F12 real NBA CondWrapper: conditioning {rel(oxc[:, :F['cond_idx'][1]], xc[:, :F['cond_idx'][1]]):.2e} finals {rel(o_final, finals[keep_k]):.2e} "
          f"positions {rel(o_pos, poss[keep_k]):.2e}; ADE {ades.mean():.4f} FDE {fdes.mean():.4f} over {ades.numel()} agents, K = {K}, first {R} count")
    assert rel(o_final, finals[keep_k]) < 1e-5 and rel(o_pos, poss[keep_k]) < 1e-5 and torch.equal(omask, mask)
    npz("f12_nba_k.npz", stage1=s1, lift=lift, weight_seed=np.array(wseed), noise_seed=np.array(nseed), embedding=w.vec_in_embedding.weight.detach().clone(),
        pos=batch["pos"], entities=batch["entities"], attention_mask=batch["attention_mask"], cond_scene=batch["cond_scene"],
        y=y, x_cond=xc, mask=mask, kept=np.array(keep_k), finals=finals[keep_k], positions=poss[keep_k], ades=ades, fdes=fdes, true_future=true_future,
        meta=np.array([B, T, A, L, K, F["cond_idx"][0], F["cond_idx"][1], F["num_steps"], R]), shape=shape_dict(sh))


def f13():
    """The peptide caller, REAL files: second_stage/peptide.py Wrapper built by its own __init__ from the reference YAML at the true
    T = 1000 (L = 2, C = 96, hidden 384, 16 heads of 24, mlp 4; depth 2), its `encode` (:85-95: keys atom14_pos / aatype / attention_mask /
    entities, frames flattened), the base class's `sample` (lightning_base.py:217-238) and its `decode` (:97-102: "(B T) L (A D) -> B T L A D")
    executed UNCHANGED over a first stage of the peptide sizes (2 latents of 96, DecoderQuerySplitter with 8 splits, atom14 head of 42).
    The initial state is torch.randn of a stored seed; of the 1000 frames the fixture keeps every 8th (final latents and atom14 positions)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_env
    from lam_slide_amd import dropin
    dropin.uninstall()
    ns = ref_env.setup()
    F = ref_env.F13
    B, T, R, L = F["B"], F["T"], F["R"], F["L"]
    lift = torch.randn(42, F["dim_input"], generator=torch.Generator().manual_seed(50)) * 0.3
    first, first_cls = ref_env.build_peptide_first_stage(ns, lift)
    w = ref_env.build_peptide_wrapper(ns, first, first_cls)
    w.eval()
    sh = latent_net.NetShape(**F["backbone"])
    wseed, nseed = 54, 55
    w.backbone.load_state_dict(latent_net.random_params(sh, seed=wseed))
    bseed = 53
    batch = ref_env.f13_batch(bseed)
    noise = torch.randn(B, T, L, 96, generator=torch.Generator().manual_seed(nseed))
    seen = {}
    real_decode = w.decode

    def tap(latents, entities):
        seen["final"] = latents.clone().reshape(B, T, L, 96)
        return real_decode(latents, entities)

    w.decode = tap
    with ref_env.fixed_randn_like(noise):
        out = w.sample({k: v.clone() for k, v in batch.items()})["atom14_pos"]
    assert out.shape == (B, T, R, 14, 3)
    pb = w.prepare_batch({k: v.clone() for k, v in batch.items()})
    xc, mask = pb["model_kwargs"]["x_cond"], pb["model_kwargs"]["x_cond_mask"]
    # the oracle chain on the same inputs
    s1 = {k: v.clone() for k, v in first.backbone.state_dict().items()}
    sd = latent_net.random_params(sh, seed=wseed)
    flat = lambda t: t.reshape(B * T, *t.shape[2:])  # noqa: E731
    es = harness.EncoderShape(dim_input=F["dim_input"], dim_latent=96, num_latents=L, num_head_cross=2, num_head_latent=2)
    ds = harness.DecoderShape(dim_latent=96, num_head_cross=2, num_head_latent=2)
    o_lat = harness.encode(s1, es, flat(batch["atom14_pos"].flatten(-2) @ lift), flat(batch["entities"]), None).reshape(B, T, L, 96)
    oxc, omask = harness.setup_conditioning(o_lat, tuple(F["cond_idx"]), True)
    o_final = harness.sample_latents(sd, sh, otr.Transport("GVP", "data"), noise, oxc, omask, None, "ODE", {"sampling_method": "euler", "num_steps": F["num_steps"]})
    o_pos = harness.decode(s1, ds, o_final.reshape(B * T, L, 96), flat(batch["entities"]), output="atom14_pos").reshape(B, T, R, 14, 3)
    print(f"F13 real peptide Wrapper (T = {T}): conditioning {rel(oxc, xc):.2e} finals {rel(o_final, seen['final']):.2e} atom14 positions {rel(o_pos, out):.2e}")
    assert rel(o_final, seen["final"]) < 1e-5 and rel(o_pos, out) < 1e-5 and torch.equal(omask, mask)
    keep = {k: v for k, v in s1.items() if "output_layers.aatype" not in k}  # (the residue-type head is not on this path)
    npz("f13_peptide.npz", stage1=keep, lift=lift, weight_seed=np.array(wseed), noise_seed=np.array(nseed), batch_seed=np.array(bseed),
        atom14_frame0=batch["atom14_pos"][:, :1].clone(), frame_stride=np.array(8), cond_latents=o_lat[:, :1].clone(), finals=seen["final"][:, ::8].clone(), positions=out[:, ::8].clone(),
        x_cond_frame0=xc[:, :1].clone(), meta=np.array([B, T, R, L, F["cond_idx"][0], F["cond_idx"][1], F["num_steps"]]), shape=shape_dict(sh))


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2", "f3", "f4", "f5", "f6", "f7", "f8", "f9", "f10", "f11", "f12", "f13"]
    for w in which:
        globals()[w]()
