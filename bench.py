#!/usr/bin/env python3
"""Throughput bench of the hot path: sampled trajectories / second of the second-stage latent SiT sampler.

One "step" = one complete sampling call (noise -> final latents) of a batch of independent trajectories on
each GPU.  Default workload = BASELINE.json configs[1]: MD17 aspirin benchmark shape, T=30 frames x L=256 latent
tokens, C=32, D=512, H=16, depth 4, mlp_ratio 2, GVP path / data prediction, ODE Euler with 50 state updates
(reference ``num_steps=51``), bf16 MFMA operands with fp32 accumulate/state.  Inputs are synthetic (seeded
random weights of that architecture, random conditioning latents) and resident in HBM before the timed
region.

Multi-GPU (``--gpus N``): one process per GPU, the batch is sharded (weak scaling: per-GPU batch fixed), no
data-path collective, one gather onto rank 0 (RCCL over xGMI, backend "nccl") of the final latents per step inside the
timed region.  Launched by ``torch.distributed.run`` the ranks are taken from the environment; invoked plainly
(``python bench.py --gpus 8``) the script starts its N ranks itself as child processes BEFORE any GPU call and
relays rank 0's single JSON line.  Documented weak-scaling lines beside the headline:
    python bench.py --gpus N --workload nba --batch 1024        (BASELINE configs[4]: 1024 trajectories per GPU, 8192 at N=8)
    python bench.py --gpus N --workload peptide --batch 8       (BASELINE configs[3]: 1000-step SDE, 8 trajectories per GPU)

Extra legs on rank 0 at N=1: ``roofline`` / ``roofline2`` (HIP-event timing of every launch of the dominant kernel, the linear1
MFMA GEMM, and of linear2, in a separate un-overlapped pass after the timed region), ``roofline_step`` (the whole step's HBM-side bytes, from
the committed PMC profile, over this run's step time: the step as it is cut into kernels is HBM-bound), ``gpu_small_batch`` (the
same call at B = 1, 8 and 64), ``stage1`` (device encode / decode
of the batch, the steps either side of the loop) and ``cpu_baseline`` (the CPU oracle restatement timed on the
host cores: the full solve of one trajectory, which is also this run's parity check, a 1-thread figure and an
all-core figure with several trajectories in flight).
"""
import argparse
import ctypes as C
import glob
import json
import os
import socket
import subprocess
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
    "pedestrian_scene": (dict(depth=6, in_dim=32, hidden_size=128, num_heads=4, mlp_ratio=2, vec_in_dim=256, normalize=True), 20, 2, (0, 8), "ODE",
                         {"sampling_method": "euler", "num_steps": 11}, 20),   # BASELINE configs[2] literally: 1 scene x 20 samples
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="md17_bench", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU per step (0 = workload default)")
    ap.add_argument("--chunk", type=int, default=0, help="trajectories per pass inside the library (0 = default)")
    ap.add_argument("--updates", type=int, default=0, help="profiling only: state updates per sampling call instead of the workload's (the line is marked "
                                                           "config.reduced_updates; never a headline number).  PMC passes of the 1000-step peptide call need it")
    ap.add_argument("--tail", default="auto", choices=("auto", "on", "off"),
                    help="sub-block decomposition of the model handle (lam_slide_amd.LatentSIV3.set_tail): auto = the tail form for hidden-256 "
                         "models (NBA, md17_ref) at >= 131 072 tokens per GPU (where it is faster), the default form elsewhere")
    ap.add_argument("--ln-fuse", default="auto", choices=("auto", "on", "off"),
                    help="LayerNorm + modulate inside linear1's activation load (lam_slide_amd.LatentSIV3.set_ln_fuse): auto = on at >= 131 072 tokens "
                         "per GPU (where it is faster) for handles that do not run the tail form, off elsewhere")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the small-batch and stage-1 legs")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel HIP-event passes (rocprofv3 runs: every launch of the trace then belongs to a warm-up or timed sampling call)")
    ap.add_argument("--profile-kernel", type=int, default=0, help="kernel class timed with HIP events (lsl_api.h)")
    ap.add_argument("--breakdown", action="store_true", help="extra untimed passes: per-kernel-class time shares")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"), help="gloo only with --stub-compute (launcher test on CPU)")
    ap.add_argument("--stub-compute", action="store_true",
                    help="launcher / collective plumbing test without a GPU: the sampling call is replaced by a trivial CPU op and the "
                         "line is marked data=stub (never a measurement)")
    ap.add_argument("--launch-timeout", type=float, default=3600.0, help="self-launched ranks (--gpus N without a launcher): stop all of them after this many seconds")
    ap.add_argument("--rendezvous-timeout", type=float, default=180.0, help="seconds init_process_group (and every collective) may wait for the other ranks")
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)  # launcher test: this rank exits with code 7 before the rendezvous
    ap.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)  # internal: one worker of the all-core CPU leg
    ap.add_argument("--cpu-threads", type=int, default=1, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher starts the N ranks itself (no GPU call happens in this parent)


def launch_ranks(args) -> int:
    """Start one fresh child process per rank (this parent never touches a GPU), watch ALL of them, and fail fast: as soon as one rank
    exits non-zero the others are stopped and the exit code is that rank's - a rank that dies before the rendezvous must not leave the
    rest waiting for the collective library's own timeout.  Rank 0's stdout goes to a temporary file (nothing can block on a pipe)."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.monotonic() + args.launch_timeout
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            failed = f"no result after --launch-timeout {args.launch_timeout:.0f} s (ranks still running: {[r for r, rc in enumerate(rcs) if rc is None]})"
            break
        time.sleep(0.05)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    out0.seek(0)
    text = out0.read()
    rcs = [p.returncode for p in procs]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if failed or not lines:
        sys.stderr.write(f"bench.py: {failed or 'rank 0 printed no result line'}; rank exit codes {rcs} (negative = stopped by the launcher)\n{text}\n")
        return next((rc for rc in rcs if rc and rc > 0), 1)
    print(lines[-1])
    return 0


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU baseline helpers (the oracle is the checker; here it is timed, never shipped)


def cpu_oracle_runner(workload, inputs=None):
    """n -> final latents after n state updates of ONE trajectory on the CPU oracle.  inputs = (init, x_cond, mask, y) of that trajectory
    (the parity leg hands over the tensors of the GPU run); None: seeded stand-ins of the same shape (timing-only workers)."""
    import torch as th
    from lam_slide_amd import LatentSIV3, setup_conditioning
    from lam_slide_amd.synthetic import seeded_state_dict
    from oracle import harness, latent_net, transport as otr
    kw, T, L, cond_idx, method, skw, _ = WORKLOADS[workload]
    params = seeded_state_dict(LatentSIV3(reset_parameters=False, **kw), seed=0)
    sh = latent_net.NetShape(**kw)
    if inputs is None:
        g = th.Generator().manual_seed(1)
        lat = th.randn(1, T, L, kw["in_dim"], generator=g)
        init = th.randn(1, T, L, kw["in_dim"], generator=g)
        y = th.randn(1, kw["vec_in_dim"], generator=g) if kw.get("vec_in_dim") else None
        xc, m = setup_conditioning(lat, cond_idx, True)
    else:
        init, xc, m, y = inputs
    tr = otr.Transport("GVP", "data")
    if method == "ODE":
        return lambda n: harness.sample_latents(params, sh, tr, init, xc, m, y, "ODE", {"sampling_method": "euler", "num_steps": n + 1})
    return lambda n: harness.sample_latents(params, sh, tr, init, xc, m, y, "SDE", {"num_steps": n, "last_step": None})


def usable_cores() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_worker(args) -> int:
    """One worker of the all-core leg: `--cpu-worker n` runs n state updates of one trajectory with --cpu-threads threads."""
    import torch as th
    th.set_num_threads(args.cpu_threads)
    run = cpu_oracle_runner(args.workload)
    one = 1 if WORKLOADS[args.workload][4] == "ODE" else 2
    run(one)
    t0 = time.perf_counter()
    run(args.cpu_worker)
    print(json.dumps({"seconds": time.perf_counter() - t0, "updates": args.cpu_worker if one == 1 else args.cpu_worker - 1}))
    return 0


# ---------------------------------------------------------------------------------------------------------------------------------


def run_rank(args) -> int:
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    stub = args.stub_compute
    if args.backend == "gloo" and not stub:
        raise SystemExit("--backend gloo is only for --stub-compute (the product path has no CPU implementation)")
    if stub and rank == args.stub_fail_rank:
        sys.stderr.write(f"bench.py: rank {rank} fails on purpose (--stub-fail-rank)\n")
        return 7
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tmo = datetime.timedelta(seconds=args.rendezvous_timeout)  # a rank that never arrives is an error after this long, not a hang
        if stub:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=tmo)

    kw, T, L, cond_idx, method, skw, default_b = WORKLOADS[args.workload]
    B = args.batch or default_b
    if args.updates > 0:
        skw = dict(skw, num_steps=args.updates + 1 if method == "ODE" else args.updates)
    n_evals = (skw["num_steps"] - 1) if method == "ODE" else skw["num_steps"]
    g = torch.Generator().manual_seed(1 + rank)

    from lam_slide_amd import sample_sharded  # (pure torch.distributed logic: importable without the HIP library)
    if stub:
        dev = torch.device("cpu")
        init = torch.randn(B, 4, 4, kw["in_dim"], generator=g)

        def sample_call(first_traj=0):
            return init * 0.5

        def sync():
            pass
    else:
        from lam_slide_amd import CreateTransport, LatentSIV3, Sampler, _lib, setup_conditioning
        from lam_slide_amd.synthetic import seeded_state_dict
        assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback of the product path)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        net = LatentSIV3(reset_parameters=False, **kw)
        params = seeded_state_dict(net, seed=0)
        net.load_state_dict(params)
        net.to(dev)
        if args.chunk:
            net.set_chunk(args.chunk)
        # the caller's choice of decomposition, made per MODEL OBJECT (never per call): large NBA batches take the tail form
        tail_on = args.tail == "on" or (args.tail == "auto" and kw["hidden_size"] == 256 and B * T * L >= 131072)
        if tail_on or args.tail == "off":
            net.set_tail(tail_on)
        lnf_on = args.ln_fuse == "on" or (args.ln_fuse == "auto" and not tail_on and B * T * L >= 131072)
        if lnf_on or args.ln_fuse == "off":
            net.set_ln_fuse(lnf_on)
        tr = CreateTransport("GVP", "data")()
        lat = torch.randn(B, T, L, kw["in_dim"], generator=g).to(dev)
        init = torch.randn(B, T, L, kw["in_dim"], generator=g).to(dev)
        y = torch.randn(B, kw["vec_in_dim"], generator=g).to(dev) if kw.get("vec_in_dim") else None
        x_cond, mask = setup_conditioning(lat, cond_idx, True)
        mk = {"x_cond": x_cond, "x_cond_mask": mask}
        if y is not None:
            mk["y"] = y
        sampler = Sampler(tr, fused=True, seed=1234)
        fn = sampler.get_sample_fn(method, skw)
        net.ensure_packed(dev)
        lib = _lib.load()

        def sample_call(first_traj=rank * B):
            # (device noise, if the sampler draws any: the slice of the unsharded stream that belongs to this rank's trajectories)
            sampler.elem_offset = first_traj * T * L * kw["in_dim"]
            return fn(init, net.forward, **mk)[-1]

        def sync():
            torch.cuda.synchronize()

    # Sharding and the one collective of the path go through the package's documented entry point (INTEGRATION.md section 3,
    # lam_slide_amd.sample_sharded: contiguous batch split, no data-path collective, ONE gather of the final latents onto rank 0 - RCCL
    # over xGMI with backend "nccl"): the function the 2-rank gloo test covers is the one a scaling run times.  Weak scaling: the global
    # batch is world x B trajectories and rank r owns [r B, (r + 1) B); its inputs were generated on its own device above, so the
    # "global" tensor handed to sample_sharded only carries the batch shape (an expanded view, no memory).
    global_shape = torch.empty(1, device=dev).expand(B * world, *init.shape[1:])

    def one_step():
        box = {}

        def shard_fn(local, lo):
            assert local.shape[0] == B and lo == rank * B, (local.shape, lo, rank, B)
            box["final"] = sample_call(lo)
            return box["final"]

        sample_sharded(shard_fn, global_shape, dst=0)  # rank 0 gets the [world * B, T, L, C] result, the others None
        return box["final"]

    def gather_final(final):  # the collective alone (timed outside the timed region): the same call with the sampling replaced by its result
        sample_sharded(lambda local, lo: final, global_shape, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        sync()

    for _ in range(args.warmup):
        one_step()
    # small-trajectory models run the trajectory-resident kernel (one launch per group of state updates, no per-kernel classes to
    # bracket).  The timed region runs the product path as it ships, without the profiler; the per-kernel HIP-event timings of the
    # roofline come from a separate pass afterwards
    resident = (not stub) and lib.lsl_sampler_path(net._handle, T, L) == 1
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        final = one_step()
    fence()
    dt = time.perf_counter() - t0
    gather_ms = None
    per_rank = [{"rank": 0, "device": local_rank, "ms_per_step": dt / args.steps * 1e3}]
    if world > 1:
        # every rank's own time: the step is the MAX over ranks, and the per-rank list shows a straggler GPU instead of hiding it in that max
        mine = torch.tensor([dt, float(local_rank)], device=dev, dtype=torch.float64)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": r, "device": int(e[1].item()), "ms_per_step": float(e[0].item()) / args.steps * 1e3} for r, e in enumerate(every)]
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
        # the one collective of the path, timed on its own (outside the timed region above): RCCL gather of the final latents
        fence()
        tg = time.perf_counter()
        for _ in range(5):
            gather_final(final)
        fence()
        gather_ms = (time.perf_counter() - tg) / 5 * 1e3
    assert torch.isfinite(final).all()
    rccl_ranks = 1
    if world > 1:  # counted by the collective itself: a rank that silently fell out of the group cannot be reported
        ones = torch.ones(1, device=dev, dtype=torch.int32)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == world, (rccl_ranks, dist.get_world_size(), world)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return 0

    traj = B * world * args.steps
    value = traj / dt
    D, M = kw["hidden_size"], int(kw["hidden_size"] * kw["mlp_ratio"])
    out = {
        "metric": "sampled trajectories/sec (50-step ODE) + decoded-coord L2 vs ref, MD17" if args.workload == "md17_bench"
        else f"sampled trajectories/sec ({args.workload})",
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "build": None if stub else lib.lsl_build_info().decode(), "dtype": "bf16", "dtype_detail": "bf16 MFMA operands (linear1, linear2, attention), fp32 accumulators / residual state / small GEMMs",
        "data": "stub (launcher test, not a measurement)" if stub else "synthetic (seeded random weights and latents)",
        "config": {"workload": args.workload, "tail": bool(not stub and net.tail), "ln_fuse": bool(not stub and net.ln_fuse), "T": T, "L": L, "C": kw["in_dim"], "D": D, "H": kw["num_heads"], "depth": kw["depth"],
                   "mlp_ratio": kw["mlp_ratio"], "sampler": method, "state_updates": n_evals, "batch_per_gpu": B,
                   "global_batch": B * world, "parallelism": f"batch-shard x{world}, 1 gather to rank 0 per step",
                   # rank r draws the device noise of global elements [r * stride, (r + 1) * stride): the slice of the unsharded stream
                   "noise_elem_stride_per_rank": B * T * L * kw["in_dim"],
                   **({"reduced_updates": True} if args.updates > 0 else {})},
        "rccl_ranks": rccl_ranks, "collective_backend": args.backend if world > 1 else None, "gather_ms": gather_ms,
        "per_rank": per_rank,
    }
    if stub:
        out.update({"roofline": None, "cpu_baseline": None})
        print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return 0

    total_ms, launches = C.c_double(), C.c_int32()

    def profiled_pass(kid):
        """One more sampling call with every launch of kernel class `kid` bracketed by HIP events on its launch stream (the library
        then keeps all passes on one stream: un-overlapped durations).  -> (summed ms, launches)"""
        _lib.check(lib.lsl_profile_enable(net._handle, kid, 8192))
        torch.cuda.synchronize()
        sample_call()
        torch.cuda.synchronize()
        tm, ln = C.c_double(), C.c_int32()
        _lib.check(lib.lsl_profile_read(net._handle, C.byref(tm), C.byref(ln)))
        profiled_pass.name = lib.lsl_profile_kernel_name(net._handle).decode()
        lib.lsl_profile_enable(net._handle, -1, 0)
        return tm.value, ln.value

    profiled_pass.name = ""

    if not resident and not args.no_roofline:
        total_ms.value, launches.value = profiled_pass(args.profile_kernel)
    out["config"]["kernels"] = "trajectory-resident (k_resident)" if resident else "general"
    f_eval = flops_per_eval_per_traj(kw, T, L)
    ws_bytes = lib.lsl_workspace_bytes(net._handle, B, T, L)
    pass_size = lib.lsl_pass_size(net._handle, B, T, L)
    out["config"].update({"trajectories_per_pass": pass_size, "workspace_mib": round(ws_bytes / 2 ** 20, 1)})
    passes = -(-B // pass_size)
    block_evals = 2 * kw["depth"] * n_evals                       # launches of each block kernel per pass and sampling call
    launches_total = passes * block_evals
    tok_total = B * T * L
    if resident:  # the whole call is (groups of) one kernel: its rate is the whole-path rate
        launches_total = 1
        args.profile_kernel = -2
    # Which kernel a profile class launched is reported by the LIBRARY (lsl_profile_kernel_name: its own dispatch, recorded while the
    # class was bracketed), not re-derived here; the algorithmic FLOPs of a class follow from that name's decomposition.
    def class_info(kid, name):
        tailed = name.startswith("k_tail") or "(q | k | v)" in name
        flops = {0: 2.0 * tok_total * D * (3 * D + (0 if tailed else M)),
                 1: 2.0 * tok_total * ((2 * M + D) * D if tailed else (D + M) * D),
                 2: 4.0 * tok_total * D * (L + T) / 2}.get(kid, 0.0) * block_evals
        what = {0: ("LayerNorm + modulate on load + " if "LayerNorm" in name else "") + "linear1 + bias / QK-norm / RoPE" + ("" if tailed else " / GELU") + " epilogue",
                1: "mlp up-projection + GELU + linear2 + gated residual + next LayerNorm" if tailed else "linear2 + gate / residual epilogue",
                2: "attention"}.get(kid, f"kernel class {kid}")
        return f"{name} ({what})" if name else what, flops

    prof_name = profiled_pass.name if not resident else ""
    kinfo = {args.profile_kernel: class_info(args.profile_kernel, prof_name)}
    kname, kflops_total = kinfo.get(args.profile_kernel, (f"kernel class {args.profile_kernel}", 0.0))
    if resident:
        kname, kflops_total = "k_resident (all state updates of a trajectory in one workgroup)", float(f_eval) * n_evals * B
        total_ms.value, launches.value = dt / args.steps * 1e3, 1
    avg_ms = total_ms.value / max(1, launches.value)
    flops_per_launch = kflops_total / max(1, launches_total)      # algorithmic FLOPs of one launch
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 and kflops_total else None
    # HBM-side bytes per launch from the committed rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, see profiles/): both operand and
    # output bytes scale with the tokens of a launch (weights are < 1 % of them), so the per-token figure is scaled to this run's
    # tokens per launch.  The newest profiles/r*_traffic.json of this workload is used.
    def committed_traffic(key, name):
        """HBM-side bytes per launch of class `key` from the newest committed PMC profile of this workload in which that class ran the
        kernel this run timed (`name` as the library reports it), scaled to this run's tokens per launch; not measured in this run (PMC
        counters need rocprofv3): the source file is named beside the number.  Also the committed kernel trace's average duration of that
        kernel, for `profile_agrees`.  -> (bytes, source, trace_avg_us, trace_tokens_per_launch)"""
        fam = name.split("<")[0] if name else ""
        for tf, tj in traffic_files():
            if key and key in tj and fam and tj[key].get("kernel", "").startswith(fam) and (("(q | k | v)" in name) == bool(tj.get("tail"))) and (("LayerNorm" in name) == bool(tj.get("ln_fuse"))):
                scale = min(pass_size, B) * T * L / tj["tokens_per_launch"]
                return int(tj[key]["bytes"] * scale), os.path.relpath(tf, ROOT), tj[key].get("avg_us"), tj["tokens_per_launch"]
        return None, None, None, None

    def agrees(avg_ms_, trace_us, trace_tok):
        """The HIP-event average of this run against the committed rocprofv3 trace's average for the same kernel (same tokens per launch
        only): within 3 %?  None when no committed trace of this kernel at this launch size exists."""
        if not trace_us or trace_tok != min(pass_size, B) * T * L or avg_ms_ <= 0:
            return None
        return bool(abs(avg_ms_ * 1e3 / trace_us - 1.0) <= 0.03)

    def traffic_files():
        """(path, content) of the committed PMC-derived traffic files of THIS workload - profiles/rNN_traffic.json (headline) and
        profiles/rNN_traffic_<workload>[_bNN].json -, newest round first, within a round the file profiled at this run's batch first."""
        found = []
        for tf in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")):
            try:
                tj = json.load(open(tf))
            except (OSError, ValueError):
                continue
            if tj.get("workload") == args.workload and "tokens_per_launch" in tj:
                found.append((os.path.basename(tf)[:3], tj.get("batch") == B, tf, tj))
        found.sort(key=lambda e: (e[0], e[1]), reverse=True)
        return [(tf, tj) for _, _, tf, tj in found]

    how = "separate sampling call after the timed region, per-launch HIP events on the launch stream (one stream, nothing co-running)"
    traffic, traffic_src, trace_us, trace_tok = committed_traffic({0: "linear1", 1: "linear2"}.get(args.profile_kernel), prof_name)
    step_ms = dt / args.steps * 1e3
    out["roofline"] = {
        "bound": "mfma", "kernel": kname, "achieved": achieved, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
        "frac": (achieved / PEAK_BF16_DENSE_TFLOPS) if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
        "measured": how if not resident else "whole call (one kernel)",
        "launches_timed": launches.value, "launches_per_call": launches_total, "avg_launch_ms": avg_ms,
        "profile_avg_launch_ms": trace_us * 1e-3 if trace_us else None, "profile_agrees": agrees(avg_ms, trace_us, trace_tok),
        "flops_per_launch": flops_per_launch, "trajectories_per_pass": pass_size,
        "kernel_time_share": avg_ms * launches_total / step_ms,
        "whole_path_tflops": value * f_eval * n_evals / 1e12,
        "whole_path_frac": value * f_eval * n_evals / 1e12 / PEAK_BF16_DENSE_TFLOPS,
    }
    if args.no_roofline:
        out["roofline"] = None
    # The whole step against the HBM roof.  Bytes = the HBM-side traffic of every kernel of one step (2 * FETCH_SIZE + WRITE_SIZE from the
    # committed rocprofv3 PMC passes, derived by tools/traffic_from_pmc.py), which scales with the tokens of a step (weights are < 1 %), so
    # the per-token figure is scaled to this run's batch; time = this run's measured step.  As the step is cut into kernels, its HBM floor
    # is ABOVE its MFMA floor: the decomposition, not any single kernel, is HBM-bound.
    for tf, tj in traffic_files():
        try:
            if "step" not in tj:
                continue
            # bytes per token and state update (the profile may have been taken with fewer updates per call: --updates), scaled to this run
            per_tok_upd = tj["step"]["bytes_per_token"] / tj.get("state_updates", n_evals)
            step_bytes = per_tok_upd * n_evals * tok_total
            step_flops = float(f_eval) * n_evals * B
            out["roofline_step"] = {
                "bound": "hbm", "achieved": step_bytes / (step_ms * 1e-3) / 1e12, "peak": 8.0, "unit": "TB/s",
                "frac": step_bytes / (step_ms * 1e-3) / 8e12, "frac_of_copy_rate": step_bytes / (step_ms * 1e-3) / 6.29e12,
                "bytes_per_step": step_bytes, "bytes_per_token_per_update": per_tok_upd,
                "hbm_floor_ms": step_bytes / 8e12 * 1e3, "hbm_floor_ms_at_copy_rate": step_bytes / 6.29e12 * 1e3,
                "mfma_floor_ms": step_flops / (PEAK_BF16_DENSE_TFLOPS * 1e12) * 1e3, "ms_per_step": step_ms,
                "bytes_are": "profile-derived (PMC counters of the committed rocprofv3 profile named in traffic_source, taken at profile_commit), "
                             "scaled to this run's tokens and updates; only the TIME is this run's",
                "traffic_source": os.path.relpath(tf, ROOT), "profile_commit": tj.get("commit"), "profile_batch": tj.get("batch"),
                "derived_by": "tools/traffic_from_pmc.py"}
            break
        except (OSError, KeyError, ValueError, TypeError):
            continue
    if not resident and args.profile_kernel == 0 and not args.no_roofline:  # the second GEMM of the block beside it
        ms2, ln2 = profiled_pass(1)
        avg2 = ms2 / max(1, ln2)
        name2, flops2 = class_info(1, profiled_pass.name)
        fl2 = flops2 / max(1, launches_total)
        tr2, tr2_src, tr2_us, tr2_tok = committed_traffic("linear2", profiled_pass.name)
        out["roofline2"] = {"bound": "mfma / hbm", "kernel": name2, "profile_avg_launch_ms": tr2_us * 1e-3 if tr2_us else None,
                            "profile_agrees": agrees(avg2, tr2_us, tr2_tok), "achieved": fl2 / (avg2 * 1e-3) / 1e12 if avg2 > 0 else None,
                            "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": fl2 / (avg2 * 1e-3) / 1e12 / PEAK_BF16_DENSE_TFLOPS if avg2 > 0 else None,
                            "traffic": tr2, "traffic_source": tr2_src, "measured": how, "launches_timed": ln2, "avg_launch_ms": avg2,
                            "flops_per_launch": fl2, "kernel_time_share": avg2 * launches_total / step_ms}

    def timed_call(fn_, reps=1):
        fn_()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            fn_()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / reps

    if args.breakdown and not resident:
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
        out["breakdown"] = breakdown

    if world == 1 and not args.no_extras:
        # the same call at small batches (SURVEY 8d asks B = 1 and 8 beside the throughput batch)
        small = {}
        for b in (1, 8, 64):  # (B = 1, 8 and 64 beside the throughput batch)
            if b == B:
                continue
            if b < B:
                initb, mkb = init[:b], {k: v[:b] for k, v in mk.items()}
            else:  # a larger batch than the headline's: the same trajectories repeated (throughput does not depend on the values)
                rep = -(-b // B)
                initb = init.repeat(rep, 1, 1, 1)[:b]
                mkb = {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:b] for k, v in mk.items()}
            sec = timed_call(lambda: fn(initb, net.forward, **mkb)[-1], reps=2)
            small[f"B{b}"] = {"value": b / sec, "unit": "trajectories/s", "ms_per_call": sec * 1e3}
        out["gpu_small_batch"] = small
        # the steps either side of the loop, on the device, for the batch of one step (frozen stage-1 models of the MD17 shape)
        if kw["in_dim"] == 32:
            from lam_slide_amd import Stage1Decoder, Stage1Encoder
            from lam_slide_amd.synthetic import seeded_decoder_state_dict, seeded_encoder_state_dict
            A = 21
            enc = Stage1Encoder(seeded_encoder_state_dict(num_latents=L, seed=6), num_head_cross=8, dim_head_cross=16, num_head_latent=2, dim_head_latent=16)
            dec = Stage1Decoder(seeded_decoder_state_dict(seed=7), num_head_latent=2, dim_head_latent=16, num_head_cross=8, dim_head_cross=16)
            xin = torch.randn(B * T, A, 128, generator=g).to(dev)
            ent = torch.arange(A, device=dev)[None].expand(B * T, A).contiguous()
            em = torch.ones(B * T, A, dtype=torch.bool, device=dev)
            flat = final.reshape(B * T, L, kw["in_dim"])
            out["stage1"] = {"frames": B * T, "entities": A,
                             "encode_ms": timed_call(lambda: enc.encode(xin, ent, em), reps=3) * 1e3,
                             "decode_ms": timed_call(lambda: dec.decode(flat, ent), reps=3) * 1e3,
                             "what": "device encode / decode of one step's batch (seeded frozen stage-1 weights), outside the timed region"}

    if not args.no_cpu and world == 1:
        out["cpu_baseline"] = cpu_leg(args, kw, T, L, method, skw, n_evals, init, mk, net, tr, dev)
        if out["cpu_baseline"]:
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return 0


def cpu_leg(args, kw, T, L, method, skw, n_evals, init, mk, net, tr, dev):
    """The CPU oracle on the host cores.  `value`: one trajectory (the rank's trajectory 0) solved with the best intra-op thread count -
    for the headline workload the FULL solve (all 50 updates, no extrapolation), which is also this run's parity check of latents and
    decoded coordinates; `one_thread`: the reference launch scripts pin OMP_NUM_THREADS=1; `all_cores`: several trajectories in flight,
    one worker process per group of threads."""
    import torch as th
    from oracle import harness
    host_cores = os.cpu_count() or 1
    one = 1 if method == "ODE" else 2  # the reference evaluates the network twice per SDE step (drift and score)
    # trajectory 0 of the GPU run: same seeded weights, the very tensors the HIP path sampled from
    run = cpu_oracle_runner(args.workload, (init[:1].cpu(), mk["x_cond"][:1].cpu(), mk["x_cond_mask"][:1].cpu(),
                                            mk["y"][:1].cpu() if "y" in mk else None))
    best, cal = None, {}
    usable = usable_cores()  # what this process may run on (affinity mask and cgroup quota): less than os.cpu_count() on a shared box
    for c in [c for c in (4, 8, 16, 32, 64, 128) if c <= usable] or [usable]:
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
    full = cal[best] / one * n_evals <= 60.0
    n_sample = n_evals if full else max(2, min(n_evals, int(round(15.0 * one / cal[best]))))
    tc = time.perf_counter()
    cpu_final = run(n_sample)
    el = time.perf_counter() - tc
    done = n_sample if method == "ODE" else n_sample - 1
    per_update = el / done
    parity = None
    if kw["in_dim"] == 32 and method == "ODE":  # (SDE runs draw their noise on different generators on the two sides)
        from lam_slide_amd import Sampler, Stage1Decoder
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
                  "state_updates": done, "bar": 1e-3,
                  "what": f"trajectory 0, {done} of {n_evals} state updates, HIP sampler + HIP decode vs CPU oracle sampler + oracle decode"}
    single = {"value": 1.0 / (per_update * n_evals), "unit": "trajectories/s", "cores": best,
              "sample": f"B=1, {done} of {n_evals} state updates timed ({el:.1f} s) with {best} threads (the best intra-op thread count)"
                        + ("" if full else f", extrapolated linearly to {n_evals}")}
    cpu = {"value": single["value"], "unit": "trajectories/s", "cores": best, "kind": "port", "host_cores": host_cores, "usable_cores": usable,
           "calibration_s_per_update": {str(k): round(v / one, 2) for k, v in cal.items()},
           "sample": "oracle restatement (pure PyTorch fp32, reference op structure, "
                     f"{one} network evaluation(s) per update): " + single["sample"],
           "single_trajectory": single, "parity": parity}
    # 1 thread (scripts/md17/second-stage.sh pins OMP_NUM_THREADS=1): one update, extrapolated (per-update cost is constant)
    th.set_num_threads(1)
    tc = time.perf_counter()
    run(one)
    t1 = (time.perf_counter() - tc) / one
    cpu["one_thread"] = {"value": 1.0 / (t1 * n_evals), "unit": "trajectories/s", "cores": 1,
                         "sample": f"1 state update timed ({t1:.1f} s), extrapolated linearly to {n_evals}"}
    th.set_num_threads(best)
    # all usable cores: several trajectories in flight, one worker process each with a few threads (the calibration above shows the
    # intra-op scaling of these small fp32 ops saturates early).  "Usable" = what this process may run on (affinity mask and cgroup
    # quota), which on a shared box is less than os.cpu_count(); wall time is bounded (workers that overrun are stopped and excluded).
    tpw = min(8, best)
    workers = max(1, min(usable // tpw, 32))
    n_w = one + 3
    env = dict(os.environ, OMP_NUM_THREADS=str(tpw), MKL_NUM_THREADS=str(tpw))
    tc = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--cpu-worker", str(n_w), "--cpu-threads", str(tpw)],
                              env=env, stdout=subprocess.PIPE, text=True) for _ in range(workers)]
    res, deadline = [], time.perf_counter() + 75.0
    for p in procs:
        try:
            o, _ = p.communicate(timeout=max(1.0, deadline - time.perf_counter()))
            if p.returncode == 0:
                res.append(json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1]))
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
    if res:
        agg = sum(r["updates"] / r["seconds"] for r in res) / n_evals
        cpu["all_cores"] = {"value": agg, "unit": "trajectories/s", "cores": len(res) * tpw, "usable_cores": usable, "workers": len(res),
                            "workers_started": workers, "threads_per_worker": tpw,
                            "sample": f"{len(res)} trajectories in flight (one worker process each), {res[0]['updates']} state updates per worker, "
                                      f"{time.perf_counter() - tc:.1f} s wall incl. start-up, extrapolated linearly to {n_evals}"}
        # the like-for-like figure against a GPU that is kept full: every usable core busy -> that is `value`
        cpu.update({"value": agg, "cores": len(res) * tpw,
                    "sample": "oracle restatement (pure PyTorch fp32, reference op structure, "
                              f"{one} network evaluation(s) per update): " + cpu["all_cores"]["sample"]
                              + f"; {usable} usable cores (of {host_cores} on the host)"})
    return cpu


def main():
    args = parse_args()
    if args.cpu_worker:
        return cpu_worker(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)  # nothing above this line touches a GPU
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
