#!/usr/bin/env python3
"""Throughput bench of the hot path: sampled trajectories / second of the second-stage latent SiT sampler.

One "step" = one complete sampling call (noise -> final latents) of a batch of independent trajectories on
each GPU.  Default workload = BASELINE.json configs[1]: MD17 aspirin benchmark shape, T=30 frames x L=256 latent
tokens, C=32, D=512, H=16, depth 4, mlp_ratio 2, GVP path / data prediction, ODE Euler with 50 state updates
(reference ``num_steps=51``), bf16 MFMA operands with fp32 accumulate/state.  Inputs are synthetic (seeded
random weights of that architecture, random conditioning latents) and resident in HBM before the timed
region.  Multi-GPU: one process per GPU, the batch is sharded (weak scaling: per-GPU batch fixed), no data-path
collective, one all_gather (RCCL) of the final latents per step inside the timed region.

Extra legs on rank 0 at N=1: ``roofline`` (HIP-event timing of the dominant kernel, the linear1 MFMA GEMM,
inside the timed region) and ``cpu_baseline`` (the CPU oracle restatement timed on the host cores on a
bounded sample of the same workload).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (net kwargs, T, L, cond_idx, sampler, sampler kwargs, default per-GPU batch)
    "md17_bench": (dict(depth=4, in_dim=32, hidden_size=512, num_heads=16, mlp_ratio=2), 30, 256, (0, 10), "ODE",
                   {"sampling_method": "euler", "num_steps": 51}, 32),
    "md17_ref": (dict(depth=4, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=2), 30, 192, (0, 10), "ODE",
                 {"sampling_method": "euler", "num_steps": 11}, 4),
    "pedestrian": (dict(depth=6, in_dim=32, hidden_size=128, num_heads=4, mlp_ratio=2, vec_in_dim=256, normalize=True), 20, 2, (0, 8), "ODE",
                   {"sampling_method": "euler", "num_steps": 11}, 1280),
    "nba": (dict(depth=6, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=4, vec_in_dim=256, normalize=True), 20, 8, (0, 5), "ODE",
            {"sampling_method": "euler", "num_steps": 51}, 1024),
    "peptide": (dict(depth=7, in_dim=96, hidden_size=384, num_heads=16, mlp_ratio=4), 1000, 2, (0, 1), "SDE",
                {"num_steps": 1000}, 8),
}
PEAK_BF16_DENSE_TFLOPS = 2500.0  # MI355X_MICROARCH.md, chip-level parameters (dense, no sparsity)


def flops_per_eval_per_traj(kw, T, L):
    """F_fwd = 6 N C D + depth * 4 N D (4D + 2M + L + T), N = T*L tokens of one trajectory."""
    D, Cc, M, n = kw["hidden_size"], kw["in_dim"], int(kw["hidden_size"] * kw["mlp_ratio"]), T * L
    return 6 * n * Cc * D + kw["depth"] * 4 * n * D * (4 * D + 2 * M + L + T)


def done_updates(method, n_sample):
    return n_sample if method == "ODE" else n_sample - 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="md17_bench", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU per step (0 = workload default)")
    ap.add_argument("--chunk", type=int, default=0, help="trajectories per pass inside the library (0 = default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--profile-kernel", type=int, default=0, help="kernel class timed with HIP events (lsl_api.h)")
    ap.add_argument("--breakdown", action="store_true", help="extra untimed passes: per-kernel-class time shares")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lam_slide_amd import CreateTransport, LatentSIV3, Sampler, _lib, setup_conditioning

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback of the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kw, T, L, cond_idx, method, skw, default_b = WORKLOADS[args.workload]
    B = args.batch or default_b
    from lam_slide_amd.synthetic import seeded_state_dict
    net = LatentSIV3(reset_parameters=False, **kw)
    params = seeded_state_dict(net, seed=0)
    net.load_state_dict(params)
    net.to(dev)
    if args.chunk:
        net.set_chunk(args.chunk)
    tr = CreateTransport("GVP", "data")()

    g = torch.Generator().manual_seed(1 + rank)
    lat = torch.randn(B, T, L, kw["in_dim"], generator=g).to(dev)
    init = torch.randn(B, T, L, kw["in_dim"], generator=g).to(dev)
    y = torch.randn(B, kw["vec_in_dim"], generator=g).to(dev) if kw.get("vec_in_dim") else None
    x_cond, mask = setup_conditioning(lat, cond_idx, True)
    mk = {"x_cond": x_cond, "x_cond_mask": mask}
    if y is not None:
        mk["y"] = y
    sampler = Sampler(tr, fused=True, seed=1234)
    sampler.elem_offset = rank * B * T * L * kw["in_dim"]
    fn = sampler.get_sample_fn(method, skw)
    gather = [torch.empty_like(init) for _ in range(world)] if world > 1 else None

    def one_step():
        final = fn(init, net.forward, **mk)[-1]
        if world > 1:
            dist.all_gather(gather, final.contiguous())
        return final

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    net.ensure_packed(dev)
    for _ in range(args.warmup):
        one_step()
    lib = _lib.load()
    n_evals = (skw["num_steps"] - 1) if method == "ODE" else skw["num_steps"]
    if rank == 0:
        _lib.check(lib.lsl_profile_enable(net._handle, args.profile_kernel, 4096))
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        final = one_step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    assert os.environ.get("LSL_PROBE") or torch.isfinite(final).all()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_ms, launches = C.c_double(), C.c_int32()
    _lib.check(lib.lsl_profile_read(net._handle, C.byref(total_ms), C.byref(launches)))
    lib.lsl_profile_enable(net._handle, -1, 0)
    traj = B * world * args.steps
    value = traj / dt
    f_eval = flops_per_eval_per_traj(kw, T, L)
    D, M = kw["hidden_size"], int(kw["hidden_size"] * kw["mlp_ratio"])
    ws_bytes = lib.lsl_workspace_bytes(net._handle, B, T, L)
    pass_size = lib.lsl_pass_size(net._handle, B, T, L)
    passes = -(-B // pass_size)
    block_evals = 2 * kw["depth"] * n_evals * args.steps          # launches of each block kernel per pass
    launches_total = passes * block_evals
    tok_total = B * T * L
    kname, kflops_total = {
        0: ("k_gemm_glds<EpiLinear1> (linear1 + QK-norm/RoPE/GELU epilogue)", 2.0 * tok_total * D * (3 * D + M) * block_evals),
        1: ("k_gemm_glds<EpiLinear2> (linear2 + gate/residual epilogue)", 2.0 * tok_total * (D + M) * D * block_evals),
        2: ("k_attention", 4.0 * tok_total * D * (L + T) / 2 * block_evals),
    }.get(args.profile_kernel, (f"kernel class {args.profile_kernel}", 0.0))
    avg_ms = total_ms.value / max(1, launches.value)
    flops_per_launch = kflops_total / launches_total            # algorithmic FLOPs of one launch
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 and kflops_total else None
    # HBM-side bytes per launch from the committed rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, see profiles/): measured
    # on a 61 440-token launch of this workload; both operand and output bytes scale with the tokens of a launch (weights
    # are < 1 % of them), so the per-token figure is scaled to this run's tokens per launch.
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        key = {0: "linear1", 1: "linear2"}.get(args.profile_kernel)
        if key and tj["workload"] == args.workload:
            traffic = int(tj[key]["bytes"] / tj["tokens_per_launch"] * min(pass_size, B) * T * L)
    except (OSError, KeyError, ValueError):
        pass
    roofline = {
        "bound": "mfma", "kernel": kname, "achieved": achieved, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
        "frac": (achieved / PEAK_BF16_DENSE_TFLOPS) if achieved else None, "traffic": traffic,
        "launches_timed": launches.value, "launches_total": launches_total, "avg_launch_ms": avg_ms,
        "flops_per_launch": flops_per_launch, "trajectories_per_pass": pass_size,
        "kernel_time_share": avg_ms * launches_total * 1e-3 / dt,
        "whole_path_tflops": value * f_eval * n_evals / 1e12,
        "whole_path_frac": value * f_eval * n_evals / 1e12 / PEAK_BF16_DENSE_TFLOPS,
    }

    breakdown = None
    if args.breakdown:
        breakdown = {}
        names = ["linear1", "linear2", "attention", "ln_modulate", "head_step", "embed", "modulation"]
        for kid, nm in enumerate(names):
            _lib.check(lib.lsl_profile_enable(net._handle, kid, 8192))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            one_step()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t1
            _lib.check(lib.lsl_profile_read(net._handle, C.byref(total_ms), C.byref(launches)))
            breakdown[nm] = {"ms": round(total_ms.value, 3), "launches": launches.value, "share_of_step": round(total_ms.value * 1e-3 / wall, 4)}
        lib.lsl_profile_enable(net._handle, -1, 0)

    cpu = None
    if not args.no_cpu and world == 1:
        from oracle import harness, latent_net, transport as otr  # the checker, timed as the CPU baseline only
        import torch as th
        sh = latent_net.NetShape(**kw)
        host_cores = os.cpu_count() or 1
        xc_c, m_c = x_cond[:1].cpu(), mask[:1].cpu()
        y_c = y[:1].cpu() if y is not None else None
        x_c = init[:1].cpu()
        n_sample = max(2, min(n_evals, 5))
        otr_t = otr.Transport("GVP", "data")
        if method == "ODE":
            run = lambda n: harness.sample_latents(params, sh, otr_t, x_c, xc_c, m_c, y_c, "ODE", {"sampling_method": "euler", "num_steps": n + 1})  # noqa: E731
        else:  # the reference evaluates the network twice per SDE step (drift and score)
            run = lambda n: harness.sample_latents(params, sh, otr_t, x_c, xc_c, m_c, y_c, "SDE", {"num_steps": n, "last_step": None})  # noqa: E731
        # Thread count: all cores is far from best for these small fp32 ops (256 threads measured 57 s per update on the
        # GPU box); calibrate on one update per candidate, then time the sample with the fastest and report it as `cores`.
        one = 1 if method == "ODE" else 2
        best, cal = None, {}
        for c in [c for c in (8, 16, 32, 64, 128) if c <= host_cores] or [host_cores]:
            th.set_num_threads(c)
            if best is None:
                run(one)  # warm-up (allocator, oneDNN primitives)
            tc = time.perf_counter()
            run(one)
            cal[c] = time.perf_counter() - tc
            if best is None or cal[c] < cal[best]:
                best = c
            if cal[c] > 20.0:
                break
        th.set_num_threads(best)
        n_sample = max(2, min(n_evals, int(round(15.0 * one / cal[best]))))  # about 15 s of CPU work
        tc = time.perf_counter()
        cpu_final = run(n_sample)
        el = time.perf_counter() - tc
        # parity of THIS workload in the same run: the same n_sample-update solve of trajectory 0 on the HIP path, both decoded to
        # coordinates (HIP: lsl_decode; CPU: the oracle's decoder restatement) with a seeded frozen decoder of the MD17 shape
        parity = None
        if kw["in_dim"] == 32 and method == "ODE":  # (SDE runs draw their noise on different generators on the two sides)
            from lam_slide_amd import Stage1Decoder
            from lam_slide_amd.synthetic import seeded_decoder_state_dict
            skw_n = dict(skw, num_steps=n_sample + 1)
            mk1 = {k: v[:1] for k, v in mk.items()}
            hip_final = Sampler(tr, fused=True, seed=1234).get_sample_fn(method, skw_n)(init[:1], net.forward, **mk1)[-1]
            dsd = seeded_decoder_state_dict(seed=7)
            dec = Stage1Decoder(dsd, num_head_latent=2, dim_head_latent=16, num_head_cross=8, dim_head_cross=16)
            ent = th.arange(21)[None].expand(T, 21)
            pos_hip = dec.decode(hip_final[0], ent.to(dev)).cpu()
            pos_cpu = harness.decode(dsd, harness.DecoderShape(), cpu_final[0], ent)
            parity = {"latents_rel_l2": harness.rel_l2(hip_final.cpu(), cpu_final), "decoded_coord_rel_l2": harness.rel_l2(pos_hip, pos_cpu),
                      "what": f"trajectory 0, {done_updates(method, n_sample)} state updates, HIP sampler + HIP decode vs CPU oracle sampler + oracle decode"}
        done = n_sample if method == "ODE" else n_sample - 1
        per_update = el / done
        cpu = {"value": 1.0 / (per_update * n_evals), "unit": "trajectories/s", "cores": best, "kind": "port",
               "host_cores": host_cores, "calibration_s_per_update": {str(k): round(v, 2) for k, v in cal.items()},
               "sample": f"oracle restatement (pure PyTorch fp32, reference op structure, {'1' if method == 'ODE' else '2'} network "
                         f"evaluation(s) per update), B=1, {done} of {n_evals} state updates timed ({el:.1f} s) with {best} threads, "
                         f"extrapolated linearly to {n_evals}", "parity": parity}

    out = {
        "metric": "sampled trajectories/sec (50-step ODE) + decoded-coord L2 vs ref, MD17" if args.workload == "md17_bench"
        else f"sampled trajectories/sec ({args.workload})",
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "dtype_detail": "bf16 MFMA operands (linear1, linear2, attention), fp32 accumulators / residual state / small GEMMs", "data": "synthetic (seeded random weights and latents)",
        "config": {"workload": args.workload, "T": T, "L": L, "C": kw["in_dim"], "D": D, "H": kw["num_heads"], "depth": kw["depth"],
                   "mlp_ratio": kw["mlp_ratio"], "sampler": method, "state_updates": n_evals, "batch_per_gpu": B,
                   "global_batch": B * world, "trajectories_per_pass": pass_size,
                   "workspace_mib": round(ws_bytes / 2 ** 20, 1), "parallelism": f"batch-shard x{world}, 1 all_gather/step"},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    if cpu:
        out["speedup_vs_cpu_baseline"] = value / cpu["value"]
    if breakdown:
        out["breakdown"] = breakdown
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
