"""GPU parity tests of the fused LayerNorm (``LatentSIV3.set_ln_fuse`` / ``lsl_model_set_ln_fuse``): sub-blocks behind a weight-stationary linear2
run no LayerNorm kernel - linear2 leaves per-wave row statistics beside the residual stream (k_lin2.hip.h, LNS), k_ln_finalize combines them,
and linear1 reads the fp32 rows and applies LayerNorm + modulate while it turns them into MFMA fragments (k_lin1.hip.h, LNF;
latent_si_v31.py:50-51,57-58).

The same arithmetic per element as the standalone kernel with the statistics summed in another order: not bit-identical to the default
path (an ulp of the bf16 operand here and there).  Every test compares with the CPU oracle at the bars of tests/test_hip_parity.py and with
the default path at the size of that rounding; what must stay bit-exact does: a trajectory's result in any batch, pass or repeated call.
"""
import pytest
import torch

from conftest import parity, rel_l2

pytestmark = pytest.mark.gpu

LNF_MODELS = {
    # NetShape kwargs, B, T, L
    "md17_bench_like": (dict(depth=2, in_dim=32, hidden_size=512, num_heads=16, mlp_ratio=2), 2, 30, 256),   # shared modulation row, K = 512
    "md17_ref_like": (dict(depth=2, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=2), 3, 30, 192),
    "nba_like_y": (dict(depth=2, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=4, vec_in_dim=24, normalize=True), 37, 20, 8),  # 160 tokens per trajectory: 3 table slots
    "d128_ragged": (dict(depth=3, in_dim=16, hidden_size=128, num_heads=4, mlp_ratio=2), 3, 7, 251),           # ragged last tile
    "d512_y_tpt256": (dict(depth=2, in_dim=32, hidden_size=512, num_heads=16, mlp_ratio=2, vec_in_dim=8), 5, 2, 130),  # per-trajectory rows at K = 512 (260 tokens each)
}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def build(kw, dev, fuse, seed=23):
    from lam_slide_amd import LatentSIV3
    from oracle import latent_net
    sh = latent_net.NetShape(**kw)
    p = latent_net.random_params(sh, seed=seed)
    net = LatentSIV3(depth=sh.depth, in_dim=sh.in_dim, hidden_size=sh.hidden_size, num_heads=sh.num_heads, vec_in_dim=sh.vec_in_dim,
                     mlp_ratio=sh.mlp_ratio, theta=sh.theta, normalize=sh.normalize, reset_parameters=False)
    net.load_state_dict(p)
    net = net.to(dev)
    net.set_ln_fuse(fuse)
    net.ensure_packed(dev)
    assert net.ln_fuse == fuse
    return sh, p, net


def inputs(sh, B, T, L, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, L, sh.in_dim, generator=g)
    xc = torch.randn(B, T, L, sh.in_dim, generator=g)
    mask = (torch.rand(B, T, L, generator=g) < 0.3).long()
    t = torch.rand(B, generator=g)
    y = torch.randn(B, sh.vec_in_dim, generator=g) if sh.vec_in_dim else None
    return x, t, xc, mask, y


@pytest.mark.parametrize("name", sorted(LNF_MODELS))
def test_forward_with_fused_layernorm_vs_oracle_and_default(name, dev):
    """One network evaluation: against the oracle at the default path's bar, and against the default path at the size of one bf16 ulp of
    the operand; the profile names the fused linear1 instance (the LayerNorm class is launched once per evaluation only)."""
    from oracle import latent_net
    kw, B, T, L = LNF_MODELS[name]
    sh, p, net = build(kw, dev, True)
    _, _, ref = build(kw, dev, False)
    x, t, xc, mask, y = inputs(sh, B, T, L)
    to = lambda v: v.to(dev) if v is not None else None  # noqa: E731
    got = net(to(x), to(t), to(xc), to(mask), to(y)).cpu()
    base = ref(to(x), to(t), to(xc), to(mask), to(y)).cpu()
    nb = min(B, 3)
    want = latent_net.forward(p, sh, x[:nb], t[:nb], xc[:nb], mask[:nb], y[:nb] if y is not None else None)
    assert torch.isfinite(got).all()
    parity(f"lnf.{name}.forward", rel_l2(got[:nb], want), 6e-4)
    parity(f"lnf.{name}.forward.default_form", rel_l2(base[:nb], want), 6e-4)
    parity(f"lnf.{name}.forward.vs_default", rel_l2(got, base), 3e-5)
    assert not torch.equal(got, base), "the fused form was expected to run (its operand differs from the default path's by an ulp somewhere)"


def test_fused_layernorm_sampler_batch_independence_and_passes(dev):
    """The fused sampling call on an ln_fuse handle: a trajectory's bits are the same alone, in the batch and in passes of 2 (the form is
    chosen by the handle and the shape, never by the launch), repeated calls reproduce the bits, and the result agrees with the oracle."""
    from lam_slide_amd import CreateTransport, SecondStageSampler
    from oracle import harness, transport as otr
    kw, _, T, L = LNF_MODELS["md17_ref_like"]
    sh, p, net = build(kw, dev, True)
    B = 5
    g = torch.Generator().manual_seed(9)
    lat, init = torch.randn(B, T, L, 32, generator=g), torch.randn(B, T, L, 32, generator=g)
    drv = SecondStageSampler(net, CreateTransport("GVP", "data")(), cond_idx=(0, 3), mask_cond_mean=True, sampling_kwargs={"sampling_method": "euler", "num_steps": 5})
    full = drv.sample_latents(lat.to(dev), init=init.to(dev))
    assert drv.last_sampler.last_path == "fused"
    assert torch.equal(full, drv.sample_latents(lat.to(dev), init=init.to(dev))), "repeated call"
    for k in (0, 4):
        assert torch.equal(full[k:k + 1], drv.sample_latents(lat[k:k + 1].to(dev), init=init[k:k + 1].to(dev))), f"trajectory {k}: alone vs in the batch"
    net.set_chunk(2)
    assert torch.equal(full, drv.sample_latents(lat.to(dev), init=init.to(dev))), "passes of 2 trajectories"
    net.set_chunk(0)
    xc, mask = harness.setup_conditioning(lat[:2], (0, 3), True)
    want = harness.sample_latents(p, sh, otr.Transport("GVP", "data"), init[:2], xc, mask, None, "ODE", {"sampling_method": "euler", "num_steps": 5})
    parity("lnf.sampler.md17_ref_like", rel_l2(full[:2].cpu(), want), 1e-3)


def test_fused_layernorm_where_it_does_not_apply_is_the_default_path(dev):
    """384-wide models (linear2 K = 2 048: no weight-stationary instance, hence no row statistics) and handles in the tail form keep the
    standalone LayerNorm: the switch is accepted and changes nothing - bit for bit."""
    kw = dict(depth=2, in_dim=96, hidden_size=384, num_heads=16, mlp_ratio=4)
    sh, _, a = build(kw, dev, True)
    _, _, b = build(kw, dev, False)
    x, t, xc, mask, y = inputs(sh, 2, 300, 2)
    to = lambda v: v.to(dev) if v is not None else None  # noqa: E731
    assert torch.equal(a(to(x), to(t), to(xc), to(mask), None), b(to(x), to(t), to(xc), to(mask), None))
    kw = dict(depth=2, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=2)
    sh, _, a = build(kw, dev, True)
    a.set_tail(True)
    _, _, b = build(kw, dev, False)
    b.set_tail(True)
    x, t, xc, mask, y = inputs(sh, 2, 30, 64)
    assert a.tail and a.ln_fuse
    assert torch.equal(a(to(x), to(t), to(xc), to(mask), None), b(to(x), to(t), to(xc), to(mask), None)), "tail handles run k_tail's own LayerNorm"
