"""GPU parity tests of the tail decomposition (lam_slide_amd/csrc/k_tail.hip.h, ``LatentSIV3.set_tail`` / ``lsl_model_set_tail``): linear1
computes q | k | v only and ONE row-owning kernel runs the mlp up-projection, GELU, linear2 over [attention | gelu(mlp)], the gated residual
update and the next sub-block's LayerNorm + modulate (mmdit.py:240-249, latent_si_v31.py:45-63).

The tail form is not bit-identical to the default decomposition (other summation order in linear2 and in the row statistics): every test
compares it with the CPU oracle at the bars of tests/test_hip_parity.py (one block update 1e-2 of the update, one evaluation 6e-4, sampler
finals 1e-3), and with the default form at the size of their common rounding error.  What must stay bit-exact does: a trajectory's result in
any batch, repeated calls, passes of any size.
"""
import pytest
import torch

from conftest import parity, rel_l2

pytestmark = pytest.mark.gpu

TAIL_MODELS = {
    # NetShape kwargs, B, T, L  (B*T*L is not a multiple of 32 in the first two: ragged last wave tile, rows beyond N never written)
    "nba_like_y": (dict(depth=2, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=4, vec_in_dim=24, normalize=True), 7, 5, 8),
    "heads32_shared": (dict(depth=2, in_dim=16, hidden_size=256, num_heads=8, mlp_ratio=2), 3, 7, 13),
    "md17_ref_like": (dict(depth=2, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=2), 2, 30, 192),
}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def build(kw, dev, tail, seed=21):
    from lam_slide_amd import LatentSIV3
    from oracle import latent_net
    sh = latent_net.NetShape(**kw)
    p = latent_net.random_params(sh, seed=seed)
    net = LatentSIV3(depth=sh.depth, in_dim=sh.in_dim, hidden_size=sh.hidden_size, num_heads=sh.num_heads, vec_in_dim=sh.vec_in_dim,
                     mlp_ratio=sh.mlp_ratio, theta=sh.theta, normalize=sh.normalize, reset_parameters=False)
    net.load_state_dict(p)
    net = net.to(dev)
    net.set_tail(tail)
    net.ensure_packed(dev)
    assert net.tail == tail
    return sh, p, net


def inputs(sh, B, T, L, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, L, sh.in_dim, generator=g)
    xc = torch.randn(B, T, L, sh.in_dim, generator=g)
    mask = (torch.rand(B, T, L, generator=g) < 0.3).long()
    t = torch.rand(B, generator=g)
    y = torch.randn(B, sh.vec_in_dim, generator=g) if sh.vec_in_dim else None
    return x, t, xc, mask, y


def test_tail_is_refused_where_no_instance_exists(dev):
    """hidden 512 / 128 / 384 have no tail kernel: the setter fails loudly (nothing silently keeps the other form)."""
    for kw in (dict(depth=1, in_dim=8, hidden_size=512, num_heads=16), dict(depth=1, in_dim=8, hidden_size=128, num_heads=4),
               dict(depth=1, in_dim=8, hidden_size=384, num_heads=16)):
        from lam_slide_amd import LatentSIV3
        net = LatentSIV3(reset_parameters=False, **kw).to(dev)
        net.ensure_packed(dev)
        with pytest.raises(ValueError, match="no tail kernel"):
            net.set_tail(True)
        assert not net.tail


@pytest.mark.parametrize("name", sorted(TAIL_MODELS))
def test_tail_block_updates_against_oracle_and_default_form(name, dev):
    """Every sub-block fed the ORACLE's input state: the update the tail form produces against the oracle's (bar of the default form: 1e-2 of
    the update) and against the default form's (their difference is rounding of the same class)."""
    from lam_slide_amd import _lib
    from oracle import latent_net
    kw, B, T, L = TAIL_MODELS[name]
    sh, p, net_t = build(kw, dev, True)
    _, _, net_d = build(kw, dev, False)
    x, t, xc, mask, y = inputs(sh, B, T, L)
    taps = {}
    latent_net.forward(p, sh, x, t, xc, mask, y, taps=taps)
    D = sh.hidden_size
    mods = torch.cat([taps[f"l{i}.mod"].reshape(B, 6 * D) for i in range(sh.depth)] + [taps["final_mod"].reshape(B, 2 * D)], dim=1).to(dev).contiguous()
    lib = _lib.load()
    ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    h_prev = taps["h0"]
    for i in range(sh.depth):
        g1 = taps[f"l{i}.mod"].reshape(B, 6 * D)[:, 2 * D:3 * D][:, None, None, :]
        h_mid = h_prev + g1 * taps[f"l{i}.sp.out"].reshape(B, T, L, D)
        h_end = taps[f"l{i}.h"]
        for bi, (hin, hout) in ((2 * i, (h_prev, h_mid)), (2 * i + 1, (h_mid, h_end))):
            a = hin.to(dev).contiguous()
            outs = []
            for net in (net_t, net_d):
                o = torch.empty_like(a)
                _lib.check(lib.lsl_debug_block(net._handle, bi, a.data_ptr(), o.data_ptr(), mods.data_ptr(), B, T, L, ws.data_ptr(), ws.numel(),
                                               torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                outs.append(o.cpu())
            want = hout - hin
            parity(f"tail.{name}.block{bi}.update", rel_l2(outs[0] - hin, want), 1e-2)
            parity(f"tail.{name}.block{bi}.vs_default", rel_l2(outs[0] - hin, outs[1] - hin), 2e-5)
        h_prev = h_end


@pytest.mark.parametrize("name", sorted(TAIL_MODELS))
def test_tail_forward_against_oracle(name, dev):
    """One evaluation (the LayerNorm + modulate the tail writes for the NEXT sub-block feeds every later linear1): against the oracle at the
    bar of the default form, and no farther from the default form than both are from the oracle."""
    from oracle import latent_net
    kw, B, T, L = TAIL_MODELS[name]
    sh, p, net_t = build(kw, dev, True)
    _, _, net_d = build(kw, dev, False)
    x, t, xc, mask, y = inputs(sh, B, T, L)
    want = latent_net.forward(p, sh, x, t, xc, mask, y)
    args = [v.to(dev) for v in (x, t, xc, mask)] + ([y.to(dev)] if y is not None else [])
    got_t, got_d = net_t(*args).cpu(), net_d(*args).cpu()
    assert net_t.last_path == "hip"
    parity(f"tail.{name}.forward", rel_l2(got_t, want), 6e-4)
    parity(f"tail.{name}.forward.default_form", rel_l2(got_d, want), 6e-4)
    parity(f"tail.{name}.forward.vs_default", rel_l2(got_t, got_d), 3e-5)


def test_tail_sampler_against_oracle_and_batch_independence(dev):
    """NBA family through the fused sampler with the tail form: final latents against the oracle (bar of the default form), a trajectory
    sampled alone, inside a batch of 6 and in passes of 2 gives the SAME BITS, and a repeated call repeats them."""
    from lam_slide_amd import CreateTransport, Sampler
    from oracle import harness, latent_net, transport as otr
    kw = dict(depth=3, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=4, vec_in_dim=256, normalize=True)
    sh, p, net = build(kw, dev, True, seed=11)
    B, T, L = 6, 20, 8
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(B, T, L, sh.in_dim, generator=g)
    init = torch.randn(B, T, L, sh.in_dim, generator=g)
    y = torch.randn(B, sh.vec_in_dim, generator=g)
    xc, mask = harness.setup_conditioning(lat, (0, 5), True)
    skw = {"sampling_method": "euler", "num_steps": 11}
    s = Sampler(CreateTransport("GVP", "data")(), fused=True)

    def run(sl):
        out = s.get_sample_fn("ODE", skw)(init[sl].to(dev), net.forward, x_cond=xc[sl].to(dev), x_cond_mask=mask[sl].to(dev), y=y[sl].to(dev))[-1]
        assert s.last_path == "fused"
        return out.cpu()

    full = run(slice(0, B))
    want = harness.sample_latents(p, sh, otr.Transport("GVP", "data"), init, xc, mask, y, "ODE", skw)
    parity("tail.nba.sampler.latents", rel_l2(full, want), 9e-4)
    assert torch.equal(full, run(slice(0, B))), "a repeated call must repeat the bits"
    for k in (0, 3, 5):
        assert torch.equal(full[k:k + 1], run(slice(k, k + 1))), f"trajectory {k}: alone vs in the batch"
    net.set_chunk(2)
    assert torch.equal(full, run(slice(0, B))), "passes of 2 trajectories"
    net.set_chunk(0)


def test_tail_at_a_large_pass_matches_default_form_and_is_labelled(dev):
    """163 840 tokens (1024 NBA trajectories: several rounds per workgroup and a partial last one): the tail form against the default form
    over one evaluation; the library names the kernel its profile class launched."""
    import ctypes as C
    from lam_slide_amd import _lib
    kw = dict(depth=1, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=4, vec_in_dim=32, normalize=True)
    sh, p, net_t = build(kw, dev, True)
    _, _, net_d = build(kw, dev, False)
    B, T, L = 1024, 20, 8
    x, t, xc, mask, y = inputs(sh, B, T, L, seed=9)
    args = [v.to(dev) for v in (x, t, xc, mask, y)]
    lib = _lib.load()
    names = []
    outs = []
    for net in (net_t, net_d):
        _lib.check(lib.lsl_profile_enable(net._handle, 1, 8))
        outs.append(net(*args))
        tm, ln = C.c_double(), C.c_int32()
        _lib.check(lib.lsl_profile_read(net._handle, C.byref(tm), C.byref(ln)))
        names.append(lib.lsl_profile_kernel_name(net._handle).decode())
        lib.lsl_profile_enable(net._handle, -1, 0)
        assert ln.value == 2 and tm.value > 0
    assert names[0].startswith("k_tail<256, 256>") and names[1].startswith("k_linear2_ws<1280>"), names
    parity("tail.large_pass.vs_default", rel_l2(outs[0].cpu(), outs[1].cpu()), 2e-5)
