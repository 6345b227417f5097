"""The drop-in boundary: the shipped Hydra overrides, the zero-source-edit rebinding of the reference's module-level
``Sampler``, duck typing of a reference ``Transport``, ``torch.compile`` wrappers, noise-stream semantics, and the evaluation helpers
next to the path.  CPU tests exercise the host logic; the ``gpu`` ones run a fused sample through each override."""
import glob
import importlib
import os
import sys
import types

import pytest
import torch
import yaml

from conftest import ROOT, rel_l2

CONFIGS = sorted(glob.glob(os.path.join(ROOT, "configs", "model", "*", "second-stage*_mi355x.yaml")))


def _instantiate(block):
    """What hydra.utils.instantiate does for these flat blocks: import ``_target_`` and call it with the other keys."""
    kw = dict(block)
    mod, name = kw.pop("_target_").rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)(**kw)


def test_override_files_cover_the_four_experiments():
    names = {os.path.relpath(c, os.path.join(ROOT, "configs", "model")) for c in CONFIGS}
    assert names == {"md17/second-stage_mi355x.yaml", "md17/second-stage_cond_mi355x.yaml", "pedestrian/second-stage_mi355x.yaml",
                     "pedestrian/second-stage_cond_mi355x.yaml", "nba/second-stage_mi355x.yaml", "nba/second-stage_cond_mi355x.yaml",
                     "peptide/second-stage_mi355x.yaml"}


@pytest.mark.parametrize("path", CONFIGS, ids=lambda p: "/".join(p.split(os.sep)[-2:]))
def test_override_yaml_instantiates(path):
    cfg = yaml.safe_load(open(path))
    base = "second-stage_cond" if "_cond_" in os.path.basename(path) else "second-stage"
    assert cfg["defaults"] == [base, "_self_"]                      # everything else is inherited from the reference's own file
    assert set(cfg) <= {"defaults", "backbone", "transport", "vec_in_dim"}
    assert cfg["backbone"]["_target_"] == "lam_slide_amd.LatentSIV3" and cfg["transport"]["_target_"] == "lam_slide_amd.CreateTransport"
    import lam_slide_amd
    net = _instantiate(cfg["backbone"])
    assert isinstance(net, lam_slide_amd.LatentSIV3)
    assert net.dims.hidden == cfg["backbone"]["hidden_size"] and net.depth == cfg["backbone"]["depth"]
    assert hasattr(net, "vec_in") == ("vec_in_dim" in cfg["backbone"])
    tr = _instantiate(cfg["transport"])()                          # the LightningModule calls instantiate(transport)() (second_stage/md17.py:43)
    assert tr.path_type is lam_slide_amd.PathType.GVP and tr.model_type is lam_slide_amd.ModelType.DATA
    assert (tr.train_eps, tr.sample_eps) == (1e-3, 1e-3)
    ref = f"/root/reference/configs/model/{path.split(os.sep)[-2]}/{base}.yaml"
    if os.path.exists(ref):  # build container only: the restated hyper-parameters equal the reference's block
        rb = yaml.safe_load(open(ref))
        if base == "second-stage_cond":
            parent = yaml.safe_load(open(ref.replace("_cond", "")))
            rb = {**parent, **rb, "backbone": {**parent["backbone"], **rb.get("backbone", {})}}
        for k, v in cfg["backbone"].items():
            if k in ("_target_",):
                continue
            rv = rb["backbone"].get(k)
            if isinstance(rv, str) and rv.startswith("${"):
                rv = rb.get("vec_in_dim") if "vec_in_dim" in rv else None
            assert rv == v, (k, rv, v)
        assert {k: v for k, v in rb["transport"].items() if k != "_target_"} == {k: v for k, v in cfg["transport"].items() if k != "_target_"}


def test_install_rebinds_the_module_level_sampler_without_source_edits():
    import lam_slide_amd
    from lam_slide_amd import dropin

    class RefSampler:  # stands for src.modules.transport.transport.Sampler
        def __init__(self, transport):
            self.transport = transport

        def sample_ode(self, *, sampling_method, num_steps, atol, rtol, reverse):
            return ("reference " + sampling_method, self.transport, num_steps)

    class RefTransport:  # the attributes as_transport() reads from a reference Transport (transport.py:36-60) + what marks it as one
        def __init__(self):
            import enum
            self.model_type = enum.Enum("ModelType", "NOISE SCORE VELOCITY DATA").DATA
            self.path_type = enum.Enum("PathType", "LINEAR GVP VP").GVP
            self.loss_type = enum.Enum("WeightType", "NONE VELOCITY LIKELIHOOD").NONE
            self.train_eps = self.sample_eps = 1e-3

        def get_drift(self):
            return None

    fakes = {}
    for name in ("src", "src.modules", "src.modules.transport", "src.modules.transport.transport", "src.models", "src.models.composites",
                 "src.models.composites.lightning_base"):
        fakes[name] = types.ModuleType(name)
    for name in ("src.modules.transport", "src.modules.transport.transport", "src.models.composites.lightning_base"):
        fakes[name].Sampler = RefSampler
    saved = {k: sys.modules.get(k) for k in fakes}
    sys.modules.update(fakes)
    try:
        # the trigger a config override provides: constructing the overridden backbone / transport factory
        lam_slide_amd.LatentSIV3(depth=1, in_dim=8, hidden_size=64, num_heads=4)
        lb = sys.modules["src.models.composites.lightning_base"]
        assert lb.Sampler is lam_slide_amd.Sampler
        assert sys.modules["src.modules.transport.transport"].Sampler is lam_slide_amd.Sampler
        assert sorted(dropin.installed()) == sorted(["src.models.composites.lightning_base.Sampler", "src.modules.transport.Sampler",
                                                     "src.modules.transport.transport.Sampler"])
        assert dropin.install() == []                              # idempotent
        # lightning_base.sample() does `Sampler(self.si).get_sample_fn(method, kwargs)` with the module-level name
        fn = lb.Sampler(lam_slide_amd.CreateTransport("GVP", "data")()).get_sample_fn("ODE", {"sampling_method": "euler", "num_steps": 3})
        out = fn(torch.zeros(1, 2, 2, 2), lambda x, t, **kw: x)
        assert out.shape == (3, 1, 2, 2, 2)
        # what this package does not implement (torchdiffeq solvers other than euler / midpoint / heun3 / rk4 / dopri5) goes back to the class
        # that was replaced when the call carries the reference's own Transport object; with this package's Transport there is nobody to hand it to
        ref_tr = RefTransport()
        got = lb.Sampler(ref_tr).get_sample_fn("ODE", {"sampling_method": "dopri8", "num_steps": 7})
        assert got == ("reference dopri8", ref_tr, 7)
        with pytest.raises(NotImplementedError):
            lb.Sampler(lam_slide_amd.CreateTransport("GVP", "data")()).get_sample_fn("ODE", {"sampling_method": "dopri8"})
        # torchdiffeq's other fixed-grid methods are native too
        assert callable(lb.Sampler(ref_tr).get_sample_fn("ODE", {"sampling_method": "rk4", "num_steps": 7}))
        # the reference's ODE default (dopri5) is native: a callable sample function, also for the reference's own Transport
        assert callable(lb.Sampler(ref_tr).get_sample_fn("ODE", {"num_steps": 7}))
        dropin.uninstall()
        assert lb.Sampler is RefSampler and dropin.installed() == [] and dropin.original_sampler() is None
    finally:
        dropin.uninstall()
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_reference_transport_object_is_accepted_by_duck_typing():
    import enum

    import lam_slide_amd as la

    class RefModelType(enum.Enum):
        NOISE = enum.auto(); SCORE = enum.auto(); VELOCITY = enum.auto(); DATA = enum.auto()  # noqa: E702

    class RefWeight(enum.Enum):
        NONE = enum.auto(); VELOCITY = enum.auto(); LIKELIHOOD = enum.auto()  # noqa: E702

    class ICPlan: pass  # noqa: E701
    class GVPCPlan: pass  # noqa: E701
    class VPCPlan: pass  # noqa: E701

    class RefTransport:  # attribute set of src/modules/transport/transport.py:40-58
        def __init__(self, mt, plan, eps):
            self.model_type, self.loss_type, self.path_sampler, self.train_eps, self.sample_eps = mt, RefWeight.NONE, plan(), eps, eps

    for plan, pt in ((ICPlan, la.PathType.LINEAR), (GVPCPlan, la.PathType.GVP), (VPCPlan, la.PathType.VP)):
        tr = la.as_transport(RefTransport(RefModelType.DATA, plan, 1e-3))
        assert isinstance(tr, la.Transport) and tr.path_type is pt and tr.model_type is la.ModelType.DATA and tr.sample_eps == 1e-3
    mine = la.CreateTransport("GVP", "data")()
    assert la.as_transport(mine) is mine
    s = la.Sampler(RefTransport(RefModelType.DATA, GVPCPlan, 1e-3))
    want = la.Sampler(mine).ode_steps(5)[0]
    assert s.ode_steps(5)[0] == want
    with pytest.raises(TypeError):
        la.as_transport(object())


def test_fused_path_is_found_behind_torch_compile_wrappers():
    import lam_slide_amd as la
    from lam_slide_amd.transport import resolve_backbone
    net = la.LatentSIV3(depth=1, in_dim=8, hidden_size=64, num_heads=4)
    compiled = torch.compile(net)                                   # OptimizedModule; nothing is traced until it is called
    assert type(compiled).__name__ == "OptimizedModule" and resolve_backbone(compiled) is net

    class Lit:  # the two lines of the LightningModule that matter (second_stage/md17.py:53-55, lightning_base.py:173-174)
        def __init__(self, backbone):
            self.backbone = backbone

        def forward(self, xt, t, **kw):
            return self.backbone(x=xt, t=t, **kw)

    assert resolve_backbone(Lit(net).forward) is net
    assert resolve_backbone(Lit(compiled).forward) is net
    assert resolve_backbone(lambda x, t: x) is None


def test_noise_streams_advance_per_call():
    import lam_slide_amd as la
    assert la.mix_seed(0, 0) != la.mix_seed(0, 1) != la.mix_seed(1, 0) and 0 <= la.mix_seed(7, 3) < 1 << 64
    tr = la.CreateTransport("GVP", "data")()
    s = la.Sampler(tr, seed=5)
    a, b = s.next_call_seed(), s.next_call_seed()
    assert a != b and (a, b) == (la.mix_seed(5, 0), la.mix_seed(5, 1))
    assert la.Sampler(tr, seed=5).next_call_seed() == a             # rebuilt with the same seed: reproducible
    torch.manual_seed(11)
    u = la.Sampler(tr)                                              # seed=None: torch's global generator, fresh per call
    c, d = u.next_call_seed(), u.next_call_seed()
    torch.manual_seed(11)
    assert c != d and la.Sampler(tr).next_call_seed() == c
    drv = la.SecondStageSampler(la.LatentSIV3(depth=1, in_dim=8, hidden_size=64, num_heads=4), tr, seed=3)
    x, y = drv._next_call_seed(), drv._next_call_seed()
    drv.reseed()
    assert x != y and drv._next_call_seed() == x
    with pytest.raises(RuntimeError):
        la.device_randn((2, 2), "cpu", 0)


def test_make_io_validates_what_it_hands_to_the_library():
    import lam_slide_amd as la
    net = la.LatentSIV3(depth=1, in_dim=8, hidden_size=64, num_heads=4, vec_in_dim=16)
    x, xc, m = torch.zeros(2, 3, 4, 8), torch.zeros(2, 3, 4, 8), torch.zeros(2, 3, 4, dtype=torch.long)
    net.make_io(x, xc, m, torch.zeros(2, 16), torch.zeros(2), torch.zeros_like(x))
    with pytest.raises(ValueError):
        net.make_io(x, xc, m, None, torch.zeros(1))                 # t must cover the batch: the kernels read t[0..B)
    with pytest.raises(ValueError):
        net.make_io(x, xc, m, None, torch.zeros(()))
    with pytest.raises(ValueError):
        net.make_io(x, xc[:1], m, None)
    with pytest.raises(ValueError):
        net.make_io(x, xc, m, torch.zeros(2, 15))
    with pytest.raises(ValueError):
        net.make_io(x[..., :4], xc[..., :4], m, None)
    with pytest.raises(ValueError):
        net.make_io(x.double(), xc, m, None)


def test_rollout_sampler_builds_the_batch_and_chains():
    """modules/sampling.py:23-63 with a toy model: every frame of the batch repeats the (masked) conditioning frame, rollout r starts
    from the last frame of rollout r-1, frame 0 of the result is the conditioning frame, shift / scale removed and restored."""
    from lam_slide_amd import RolloutSampler

    class Toy:
        n_timesteps, shift, scale = 4, 1.0, 2.0

        def __init__(self):
            self.batches = []

        def sample(self, batch):
            self.batches.append(batch)
            pos = batch["atom14_pos"]
            return {"atom14_pos": pos + torch.arange(1, 5, dtype=pos.dtype).view(1, 4, 1, 1, 1)}

    toy = Toy()
    cond = torch.arange(2 * 14 * 3, dtype=torch.float32).reshape(2, 14, 3)
    res, res_mask = torch.tensor([3, 7]), torch.ones(2, 14, dtype=torch.bool)
    res_mask[1, 5:] = False
    out = RolloutSampler(toy).sample_rollout(cond, res, res_mask, num_rollouts=2)
    b0 = toy.batches[0]
    assert b0["atom14_pos"].shape == (1, 4, 2, 14, 3) and b0["aatype"].shape == (1, 4, 2) and b0["entities"].shape == (1, 4, 2)
    assert bool(b0["attention_mask"].all()) and torch.equal(b0["entities"][0, 2], torch.arange(2))
    norm = (cond - 1.0) / 2.0
    assert torch.equal(b0["atom14_pos"][0, 3], norm * res_mask[..., None])
    assert torch.equal(toy.batches[1]["atom14_pos"][0, 0], (norm * res_mask[..., None] + 4) * res_mask[..., None])
    assert out.shape == (8, 2, 14, 3) and torch.equal(out[0], cond)


def test_best_of_k_errors_layout_against_the_oracle_formula():
    """The K-loop of second_stage/pedestrian.py:186-212 folded into one call: agent rows ordered (B L), masked agents dropped,
    future frames only; the reductions equal the reference formula (oracle.compute_errors, itself checked against the source)."""
    from lam_slide_amd import best_of_k_errors
    from oracle import harness
    B, T, L, C, A, K, c1 = 3, 6, 2, 4, 5, 4, 2
    g = torch.Generator().manual_seed(0)
    samples = torch.randn(K, B, T, L, C, generator=g)

    class Stub:
        cond_idx = (0, c1)

        def sample_latents_k(self, latents, K, y=None, inits=None):
            return samples

    proj = torch.randn(L * C, A * 2, generator=g)

    def decode(z):  # [K*B, T, L, C] -> [K*B, T, A, 2]
        return (z.reshape(z.shape[0], T, L * C) @ proj).reshape(z.shape[0], T, A, 2)

    target = torch.randn(B, T - c1, A, 2, generator=g)
    mask = torch.rand(B, A, generator=g) > 0.3
    ade, fde = best_of_k_errors(Stub(), torch.zeros(B, T, L, C), target, K, decode, agent_mask=mask)
    # the reference's loop: K separate samples, "B T L D -> (B L) T D", [mask], stack over K
    per_k = [decode(samples[k])[:, c1:].permute(0, 2, 1, 3).reshape(B * A, T - c1, 2)[mask.reshape(-1)] for k in range(K)]
    want = harness.compute_errors(torch.stack(per_k, dim=1), target.permute(0, 2, 1, 3).reshape(B * A, T - c1, 2)[mask.reshape(-1)])
    assert torch.equal(ade, want[0]) and torch.equal(fde, want[1]) and ade.shape == (int(mask.sum()),)


# ---- GPU: one fused sample through every shipped override ---------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("path", CONFIGS, ids=lambda p: "/".join(p.split(os.sep)[-2:]))
def test_override_yaml_runs_the_fused_path_on_gpu(path):
    """`_target_`s from the YAML, wired the way the reference LightningModule wires them (backbone optionally behind torch.compile,
    `Sampler(self.si).get_sample_fn(...)(noise, self.forward, **model_kwargs)[-1]`), must take the fused HIP loop and agree with the
    oracle on the same weights."""
    import lam_slide_amd as la
    from oracle import harness, latent_net, transport as otr
    dev = torch.device("cuda:0")
    cfg = yaml.safe_load(open(path))
    net = _instantiate({**cfg["backbone"], "reset_parameters": False})
    kw = {k: v for k, v in cfg["backbone"].items() if k in ("depth", "in_dim", "hidden_size", "num_heads", "mlp_ratio", "vec_in_dim", "normalize")}
    sh = latent_net.NetShape(**kw)
    p = latent_net.random_params(sh, seed=3)
    net.load_state_dict(p)
    net.to(dev)

    class Lit:
        def __init__(self, backbone, si):
            self.backbone, self.si = backbone, si

        def forward(self, xt, t, **kw):
            return self.backbone(x=xt, t=t, **kw)

    lit = Lit(torch.compile(net) if "md17" in path else net, _instantiate(cfg["transport"])())
    T, L = (8, 24) if sh.hidden_size == 256 and not sh.normalize else (6, 2) if sh.in_dim == 96 else (20, 2 if sh.hidden_size == 128 else 8)
    B = 3
    g = torch.Generator().manual_seed(2)
    lat, init = torch.randn(B, T, L, sh.in_dim, generator=g), torch.randn(B, T, L, sh.in_dim, generator=g)
    y = torch.randn(B, sh.vec_in_dim, generator=g) if sh.vec_in_dim else None
    xc, m = harness.setup_conditioning(lat, (0, 2), True)
    mk = {"x_cond": xc.to(dev), "x_cond_mask": m.to(dev)}
    if y is not None:
        mk["y"] = y.to(dev)
    skw = {"sampling_method": "euler", "num_steps": 6}
    sampler = la.Sampler(lit.si)
    got = sampler.get_sample_fn("ODE", skw)(init.to(dev), lit.forward, **mk)[-1]
    assert sampler.last_path == "fused" and net.last_path == "hip"
    want = harness.sample_latents(p, sh, otr.Transport("GVP", "data"), init, xc, m, y, "ODE", skw)
    err = rel_l2(got.cpu(), want)
    from conftest import parity
    parity(f"dropin.{'/'.join(path.split(os.sep)[-2:])}", err, 1e-3)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_install_on_the_real_reference_modules_and_real_lightning_sample(golden):
    """The drop-in against the reference's REAL modules (nothing faked but Lightning / Hydra themselves, tools/ref_env.py): the real
    `src.modules.transport` package and `models/composites/lightning_base.py` are imported, `install()` rebinds `Sampler` on those module
    objects, a real `Transport` passes `as_transport`, and `SecondStageCondLightningBase.sample` (lightning_base.py:217-238) - unchanged,
    on the real second_stage/md17.py Wrapper built by its own __init__ from the reference YAML - gives the F9 fixture with this package's
    Sampler behind the module-level name (generic per-step loop around the reference backbone on the CPU)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_env

    import lam_slide_amd
    from lam_slide_amd import dropin
    from lam_slide_amd.transport import as_transport
    dropin.uninstall()
    ns = ref_env.setup()
    ref_sampler = ns.transport.Sampler
    assert ns.lightning_base.Sampler is ref_sampler and ref_sampler is not lam_slide_amd.Sampler
    assert not hasattr(ns.transport_pkg, "Sampler")  # (the package __init__ re-exports the Transport names only)
    try:
        # the trigger a config override provides: the overridden transport factory is constructed (LSL_NO_INSTALL unset)
        lam_slide_amd.CreateTransport("GVP", "data")
        assert ns.lightning_base.Sampler is lam_slide_amd.Sampler and ns.transport.Sampler is lam_slide_amd.Sampler
        assert dropin.original_sampler() is ref_sampler
        # a REAL reference Transport object (what `instantiate(transport)()` of an un-overridden config returns)
        real_tr = ns.transport_pkg.CreateTransport("GVP", "data")()
        assert type(real_tr).__module__ == "src.modules.transport.transport"
        mine = as_transport(real_tr)
        assert mine.path_type is lam_slide_amd.PathType.GVP and mine.model_type is lam_slide_amd.ModelType.DATA
        assert (mine.train_eps, mine.sample_eps) == (real_tr.train_eps, real_tr.sample_eps) == (1e-3, 1e-3)
        assert mine.check_interval(mine.train_eps, mine.sample_eps, sde=False, eval=True) == real_tr.check_interval(
            real_tr.train_eps, real_tr.sample_eps, sde=False, eval=True)
        # the real LightningModule, reference backbone + reference transport factory, Sampler = this package's
        f = golden("f9_sample.npz")
        first, first_cls = ref_env.build_first_stage(ns)
        w = ref_env.build_wrapper(ns, "src.models.components.latent.latent_si_v31.LatentSIV3", "src.modules.transport.CreateTransport", first, first_cls)
        w.eval()
        assert type(w.si).__module__ == "src.modules.transport.transport"
        w.backbone.load_state_dict(f.group("backbone"))
        first.backbone.load_state_dict(f.group("stage1"))
        batch = {"pos": f["x"], "entities": f["entities"], "attention_mask": f["attention_mask"]}
        with ref_env.fixed_randn_like(f["noise"]):
            pos = w.sample(dict(batch))["pos"]
        assert rel_l2(pos, f["pos"]) < 2e-6
        # ... and with the override `_target_`s: the transport factory of this package inside the real Wrapper (CPU: generic loop again)
        w2 = ref_env.build_wrapper(ns, "src.models.components.latent.latent_si_v31.LatentSIV3", "lam_slide_amd.CreateTransport", first, first_cls)
        w2.eval()
        assert isinstance(w2.si, lam_slide_amd.Transport)
        w2.backbone.load_state_dict(f.group("backbone"))
        with ref_env.fixed_randn_like(f["noise"]):
            pos2 = w2.sample(dict(batch))["pos"]
        assert rel_l2(pos2, f["pos"]) < 2e-6
        # the override backbone constructs inside the real Wrapper and takes the reference state dict (it runs on the GPU only)
        w3 = ref_env.build_wrapper(ns, "lam_slide_amd.LatentSIV3", "lam_slide_amd.CreateTransport", first, first_cls)
        assert isinstance(w3.backbone, lam_slide_amd.LatentSIV3)
        w3.backbone.load_state_dict(f.group("backbone"))
        with pytest.raises(RuntimeError):
            with ref_env.fixed_randn_like(f["noise"]):
                w3.sample(dict(batch))   # CPU tensors: the product path refuses, it does not fall back
    finally:
        dropin.uninstall()
    assert ns.lightning_base.Sampler is ref_sampler
