"""CPU: the oracle restatement against the golden vectors produced by the reference's own modules
(tools/make_fixtures.py).  Tolerances: 2e-6 relative L2 for one fp32 network evaluation, 5e-5 for
multi-step samplers (late steps amplify fp32 rounding by (pi/2)/cos(pi t/2)), exact for integers."""
import torch

from conftest import rel_l2, shape_from
from oracle import harness, latent_net, transport as otr


def test_f1_block_intermediates(golden):
    f = golden("f1_block.npz")
    sh = shape_from(f.group("shape"))
    p = f.group("p")
    taps = {}
    out = latent_net.forward(p, sh, f["x"], f["t"], f["x_cond"], f["mask"], f["y"], taps=taps)
    ref = f.group("taps")
    assert len(ref) == 35
    for k, v in ref.items():
        assert rel_l2(taps[k].reshape(v.shape), v) < 2e-6, k
    assert rel_l2(out, ref["out"]) < 2e-6
    p64 = latent_net.cast_params(p, torch.float64)
    out64 = latent_net.forward(p64, sh, f["x"].double(), f["t"].double(), f["x_cond"].double(), f["mask"], f["y"].double())
    assert rel_l2(out64, f["out64"]) < 1e-12


def test_f2_shape_classes(golden):
    f = golden("f2_shapes.npz")
    names = sorted({k.split("/")[0] for k in f.raw.files})
    assert len(names) == 5
    for n in names:
        g = f.group(n)
        sh = shape_from({k[6:]: v for k, v in g.items() if k.startswith("shape.")})
        p = latent_net.random_params(sh, seed=int(g["weight_seed"]))
        out = latent_net.forward(p, sh, g["x"], g["t"], g["x_cond"], g["mask"], g.get("y"))
        assert rel_l2(out, g["out"]) < 2e-6, n


def test_f10_linear_attention_mode(golden):
    """attention_mode="linear" (mmdit.py:58-72): outputs of the reference module built with that mode; the softmax form must NOT match them."""
    import dataclasses
    f = golden("f10_linear_attention.npz")
    names = sorted({k.split("/")[0] for k in f.raw.files})
    assert len(names) == 3
    for n in names:
        g = f.group(n)
        sdpa = shape_from({k[6:]: v for k, v in g.items() if k.startswith("shape.")})
        sh = dataclasses.replace(sdpa, attention_mode="linear")
        p = latent_net.random_params(sh, seed=int(g["weight_seed"]))
        args = (g["x"], g["t"], g["x_cond"], g["mask"], g.get("y"))
        assert rel_l2(latent_net.forward(p, sh, *args), g["out"]) < 2e-6, n
        assert rel_l2(latent_net.forward(p, sdpa, *args), g["out"]) > 5e-3, n


def test_f3_transport_scalars(golden):
    f = golden("f3_transport.npz")
    t, x, mo = f["t"], f["x"], f["model_out"]
    model = lambda xx, tt, **kw: mo  # noqa: E731
    for path in otr.PATHS:
        for pred in otr.PREDICTIONS:
            g = f.group(f"{path}.{pred}")
            tr = otr.Transport(path, pred)
            iv = []
            for sde in (False, True):
                for form in ("SBDM", "linear"):
                    for ls in (0.0, 0.04):
                        iv.append(list(map(float, tr.interval(diffusion_form=form, sde=sde, last_step_size=ls))))
            assert torch.allclose(torch.tensor(iv, dtype=torch.float64), g["intervals"], atol=0, rtol=0)
            assert rel_l2(tr.velocity(x, t, model), g["velocity"]) < 1e-6
            assert rel_l2(tr.score(x, t, model), g["score"]) < 1e-6
            for form in ("constant", "SBDM", "sigma", "linear", "decreasing", "inccreasing-decreasing"):
                d = torch.as_tensor(tr.plan.diffusion(x, t, form, 0.7)) * torch.ones(13, 1, 1)
                assert rel_l2(d, g["diff." + form]) < 1e-6


def test_f3_error_behaviour():
    import pytest
    with pytest.raises(KeyError):
        otr.Transport("nope", "data")
    tr = otr.Transport("GVP", "data")
    with pytest.raises(NotImplementedError):
        tr.plan.diffusion(torch.zeros(1, 1), torch.ones(1) * 0.5, "bogus", 1.0)
    assert tr.interval(sde=False) == (1e-3, 1 - 1e-3)
    assert tr.interval(sde=True, diffusion_form="linear", last_step_size=0.04) == (1e-3, 1 - 0.04)
    assert otr.Transport("Linear", "velocity").interval(sde=False) == (0, 1)


def test_f4_samplers(golden):
    f = golden("f4_sampler.npz")
    sh = shape_from(f.group("shape"))
    p = f.group("p")
    init, xc, mask = f["init"], f["x_cond"], f["mask"]
    tr = otr.Transport("GVP", "data")
    for n in (2, 11, 51):
        got = harness.sample_latents(p, sh, tr, init, xc, mask, None, "ODE", {"sampling_method": "euler", "num_steps": n})
        assert rel_l2(got, f[f"ode{n}"]) < 5e-5, n
    for path, pred in (("Linear", "velocity"), ("Linear", "data"), ("VP", "noise"), ("GVP", "score")):
        got = harness.sample_latents(p, sh, otr.Transport(path, pred), init, xc, mask, None, "ODE",
                                     {"sampling_method": "euler", "num_steps": 6})
        assert rel_l2(got, f[f"ode6.{path}.{pred}"]) < 5e-5
    model = lambda xt, t, **mk: latent_net.forward(p, sh, xt, t, **mk)  # noqa: E731
    for n, form, last, meth in ((3, "linear", "Mean", "Euler"), (10, "linear", "Mean", "Euler"), (10, "SBDM", None, "Euler"),
                                (6, "sigma", "Euler", "Euler"), (5, "linear", "Mean", "Heun"), (6, "decreasing", "Tweedie", "Euler")):
        tag = f"sde{n}.{form}.{last}.{meth}"
        noise = list(f[tag + ".noise"])
        kw = {"sampling_method": meth, "diffusion_form": form, "last_step": last, "num_steps": n}
        for single in (False, True):
            xs = otr.get_sample_fn(tr, "SDE", kw, noise=noise, single_eval=single)(init, model, x_cond=xc, x_cond_mask=mask)
            assert len(xs) == n
            assert rel_l2(xs[-1], f[tag + ".final"]) < 5e-5, tag
            assert rel_l2(xs[-2], f[tag + ".penultimate"]) < 5e-5, tag


def test_f4_cfg1_full_shape(golden):
    """True cfg-1 shape (T=30, L=192, C=32, D=256, H=16, depth 4), B=1, 10 Euler updates."""
    f = golden("f4_cfg1.npz")
    sh = shape_from(f.group("shape"))
    p = latent_net.random_params(sh, seed=int(f["weight_seed"]))
    g = torch.Generator().manual_seed(int(f["latent_seed"]))
    lat = torch.randn(1, 30, 192, 32, generator=g)
    xc, mask = harness.setup_conditioning(lat, tuple(f["cond_idx"].tolist()), True)
    init = torch.randn(1, 30, 192, 32, generator=torch.Generator().manual_seed(int(f["init_seed"])))
    got = harness.sample_latents(p, sh, otr.Transport("GVP", "data"), init, xc, mask, None, "ODE",
                                 {"sampling_method": "euler", "num_steps": int(f["num_steps"])})
    assert rel_l2(got, f["final"]) < 5e-5


def test_f5_conditioning_bit_exact(golden):
    f = golden("f5_cond.npz")
    lat = f["latents"]
    for mean in (True, False):
        for ci in ((0, 3), (0, 1), (2, 5)):
            xc, mask = harness.setup_conditioning(lat, ci, mean)
            assert torch.equal(xc, f[f"m{int(mean)}.{ci[0]}_{ci[1]}.x_cond"])
            assert torch.equal(mask, f[f"m{int(mean)}.{ci[0]}_{ci[1]}.mask"])


def test_f6_decode(golden):
    f = golden("f6_decode.npz")
    pos = harness.decode(f.group("p"), harness.DecoderShape(), f["z"], f["entities"])
    assert rel_l2(pos, f["pos"]) < 2e-6


def test_f7_encoder_restatement_matches_reference_module(golden):
    """oracle.harness.encode against latents produced by the reference's Encoder + quant (ragged entity mask)."""
    from oracle import harness
    d = golden("f7_encode.npz")
    z = harness.encode(d.group("p"), harness.EncoderShape(num_latents=48), d["x"], d["entities"], d["mask"])
    assert rel_l2(z, d["z"]) < 2e-6


def test_f8_decoder_query_splitter_restatement(golden):
    """oracle.harness.decode (extender + cross block + tanh GELU) against the reference's DecoderQuerySplitter output."""
    from oracle import harness
    d = golden("f8_decode_split.npz")
    pos = harness.decode(d.group("p"), harness.DecoderShape(num_block_cross=1, act="gelu_tanh"), d["z"], d["entities"])
    assert rel_l2(pos, d["pos"]) < 2e-6


def test_f9_real_lightning_module_sample_chain(golden):
    """F9 = the reference's real LightningModule (second_stage/md17.py Wrapper, lightning_base.py sample / prepare_batch /
    setup_conditioning, executed unchanged in the build container by tools/make_fixtures.py f9): stage-1 inputs -> encode -> conditioning
    -> 5 Euler updates from a fixed initial noise -> decode.  The oracle chain reproduces every stage."""
    f = golden("f9_sample.npz")
    B, T, A, L, c0, c1, n = (int(v) for v in f["meta"])
    s1, sd = f.group("stage1"), f.group("backbone")
    flat = lambda t: t.reshape(B * T, *t.shape[2:])  # noqa: E731
    lat = harness.encode(s1, harness.EncoderShape(num_latents=L), flat(f["x"]), flat(f["entities"]), flat(f["attention_mask"])).reshape(B, T, L, 32)
    assert rel_l2(lat, f["latents"]) < 2e-6
    xc, mask = harness.setup_conditioning(lat, (c0, c1), True)
    assert torch.equal(mask, f["mask"]) and rel_l2(xc, f["x_cond"]) < 2e-6
    sh = latent_net.NetShape(depth=2, in_dim=32, hidden_size=64, mlp_ratio=2, num_heads=4)
    final = harness.sample_latents(sd, sh, otr.Transport("GVP", "data"), f["noise"], xc, mask, None, "ODE", {"sampling_method": "euler", "num_steps": n})
    assert rel_l2(final, f["final"]) < 5e-6
    pos = harness.decode(s1, harness.DecoderShape(), flat(final), flat(f["entities"])).reshape(f["pos"].shape)
    assert rel_l2(pos, f["pos"]) < 5e-6


def test_f11_real_pedestrian_cond_wrapper_k_loop(golden):
    """F11 = the reference's real pedestrian CondWrapper (second_stage/pedestrian.py, built by its own __init__ from the reference YAMLs
    in the build container): `prepare_batch` with the class vector y = Embedding(cond_scene) (:242-251) and the K = 20 `test_step` loop
    (:186-212) with one stored initial noise per sample() call.  The oracle chain reproduces the conditioning, every sample's final latents
    and decoded positions, and the best-of-K ADE / FDE of the real (unpadded) agents."""
    f = golden("f11_pedestrian_k.npz")
    B, T, A, L, K, c0, c1, n = (int(v) for v in f["meta"])
    sh = shape_from(f.group("shape"))
    sd = latent_net.random_params(sh, seed=int(f["weight_seed"]))
    s1 = f.group("stage1")
    flat = lambda t: t.reshape(B * T, *t.shape[2:])  # noqa: E731
    y = f["embedding"][f["cond_scene"].long()]
    assert torch.equal(y, f["y"])  # CondWrapper.prepare_batch: an embedding lookup
    pos = f["pos"].clone()
    pos[:, c1:] = 0  # test_step hides the future frames from the encoder (:175-176)
    lat = harness.encode(s1, harness.EncoderShape(num_latents=L), flat(pos @ f["lift"]), flat(f["entities"]), flat(f["attention_mask"])).reshape(B, T, L, 32)
    xc, mask = harness.setup_conditioning(lat, (c0, c1), True)
    assert torch.equal(mask, f["mask"]) and rel_l2(xc[:, :c1], f["x_cond"][:, :c1]) < 2e-6
    finals = torch.stack([harness.sample_latents(sd, sh, otr.Transport("GVP", "data"), f["noises"][k], xc, mask, y, "ODE",
                                                 {"sampling_method": "euler", "num_steps": n}) for k in range(K)])
    assert rel_l2(finals, f["finals"]) < 5e-6
    positions = harness.decode(s1, harness.DecoderShape(), finals.reshape(K * B * T, L, 32), flat(f["entities"]).repeat(K, 1)).reshape(K, B, T, A, 3)
    assert rel_l2(positions, f["positions"]) < 5e-6
    keep = f["attention_mask"][:, -1].reshape(-1).bool()
    traj = positions[:, :, c1:].permute(1, 3, 0, 2, 4).reshape(B * A, K, T - c1, 3)[keep]
    tgt = f["true_future"].permute(0, 2, 1, 3).reshape(B * A, T - c1, 3)[keep]
    ades, fdes = harness.compute_errors(traj, tgt)
    assert ades.shape == f["ades"].shape and rel_l2(ades, f["ades"]) < 5e-6 and rel_l2(fdes, f["fdes"]) < 5e-6


def test_f12_real_nba_cond_wrapper_k_loop(golden):
    """F12 = the reference's real NBA CondWrapper (second_stage/nba.py, built by its own __init__ from the reference YAMLs in the build
    container; class defaults K = 60, num_runs = 20) at the NBA shape (T = 20, L = 8, hidden 256, 16 heads, mlp 4, class vector):
    `prepare_batch` (:254-263) and the `test_step` loop (:205-225).  The oracle chain reproduces the conditioning, the kept samples' final
    latents and decoded positions, and the ADE / FDE over the FIRST num_runs samples of the real agents (what `selected_traj` is, :229)."""
    f = golden("f12_nba_k.npz")
    B, T, A, L, K, c0, c1, n, R = (int(v) for v in f["meta"])
    sh = shape_from(f.group("shape"))
    sd = latent_net.random_params(sh, seed=int(f["weight_seed"]))
    s1 = f.group("stage1")
    flat = lambda t: t.reshape(B * T, *t.shape[2:])  # noqa: E731
    y = f["embedding"][f["cond_scene"].long()]
    assert torch.equal(y, f["y"])
    noises = torch.randn(K, B, T, L, 32, generator=torch.Generator().manual_seed(int(f["noise_seed"])))
    pos = f["pos"].clone()
    pos[:, c1:] = 0  # test_step hides the future frames from the encoder (:188-189)
    lat = harness.encode(s1, harness.EncoderShape(num_latents=L), flat(pos @ f["lift"]), flat(f["entities"]), flat(f["attention_mask"])).reshape(B, T, L, 32)
    xc, mask = harness.setup_conditioning(lat, (c0, c1), True)
    assert torch.equal(mask, f["mask"]) and rel_l2(xc[:, :c1], f["x_cond"][:, :c1]) < 2e-6
    kept = [int(k) for k in f["kept"]]
    need = sorted(set(range(R)) | set(kept))
    fin = {k: harness.sample_latents(sd, sh, otr.Transport("GVP", "data"), noises[k], xc, mask, y, "ODE", {"sampling_method": "euler", "num_steps": n})
           for k in need}
    assert rel_l2(torch.stack([fin[k] for k in kept]), f["finals"]) < 5e-6
    dec = lambda z: harness.decode(s1, harness.DecoderShape(), z.reshape(-1, L, 32), flat(f["entities"]).repeat(z.shape[0], 1)).reshape(z.shape[0], B, T, A, 3)  # noqa: E731
    assert rel_l2(dec(torch.stack([fin[k] for k in kept])), f["positions"]) < 5e-6
    positions = dec(torch.stack([fin[k] for k in range(R)]))
    keep = f["attention_mask"][:, -1].reshape(-1).bool()
    traj = positions[:, :, c1:].permute(1, 3, 0, 2, 4).reshape(B * A, R, T - c1, 3)[keep]
    tgt = f["true_future"].permute(0, 2, 1, 3).reshape(B * A, T - c1, 3)[keep]
    ades, fdes = harness.compute_errors(traj, tgt)
    assert ades.shape == f["ades"].shape and rel_l2(ades, f["ades"]) < 5e-6 and rel_l2(fdes, f["fdes"]) < 5e-6


def test_f13_real_peptide_wrapper_sample(golden):
    """F13 = the reference's real peptide second-stage Wrapper (second_stage/peptide.py, built by its own __init__ from the reference YAML in
    the build container) at T = 1000: `encode` (:85-95), the base class's `sample` (lightning_base.py:217-238), `decode` to atom14 positions
    (:97-102), over a first stage of the peptide sizes (2 latents of 96, DecoderQuerySplitter, atom14 head of 42).  The oracle chain
    reproduces the conditioning frame, the final latents and the decoded positions on the kept frames (every 8th).  The full restatement at
    T = 1000 takes ~ 20 s on the CPU."""
    from golden_inputs import peptide_frames
    f = golden("f13_peptide.npz")
    B, T, R, L, c0, c1, n = (int(v) for v in f["meta"])
    st = int(f["frame_stride"])
    sh = shape_from(f.group("shape"))
    sd = latent_net.random_params(sh, seed=int(f["weight_seed"]))
    s1 = f.group("stage1")
    batch = peptide_frames(int(f["batch_seed"]), B, T, R)
    assert torch.equal(batch["atom14_pos"][:, :1], f["atom14_frame0"])
    noise = torch.randn(B, T, L, 96, generator=torch.Generator().manual_seed(int(f["noise_seed"])))
    flat = lambda t: t.reshape(B * T, *t.shape[2:])  # noqa: E731
    es = harness.EncoderShape(dim_input=f["lift"].shape[1], dim_latent=96, num_latents=L, num_head_cross=2, num_head_latent=2)
    ds = harness.DecoderShape(dim_latent=96, num_head_cross=2, num_head_latent=2)
    lat = harness.encode(s1, es, flat(batch["atom14_pos"].flatten(-2) @ f["lift"]), flat(batch["entities"]), None).reshape(B, T, L, 96)
    assert rel_l2(lat[:, :1], f["cond_latents"]) < 2e-6
    xc, mask = harness.setup_conditioning(lat, (c0, c1), True)
    assert rel_l2(xc[:, :1], f["x_cond_frame0"]) < 2e-6 and int(mask.sum()) == B * (c1 - c0) * L
    final = harness.sample_latents(sd, sh, otr.Transport("GVP", "data"), noise, xc, mask, None, "ODE", {"sampling_method": "euler", "num_steps": n})
    assert rel_l2(final[:, ::st], f["finals"]) < 5e-6
    pos = harness.decode(s1, ds, final.reshape(B * T, L, 96), flat(batch["entities"]), output="atom14_pos").reshape(B, T, R, 14, 3)
    assert rel_l2(pos[:, ::st], f["positions"]) < 5e-6
