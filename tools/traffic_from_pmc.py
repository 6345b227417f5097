#!/usr/bin/env python3
"""HBM-side traffic of one sampling step from a rocprofv3 profile of bench.py (tools/gpu.sh prof | profile -> gpurun_out/prof_<tag>.summary.txt).

    python tools/traffic_from_pmc.py gpurun_out/prof_r04.summary.txt --workload md17_bench --batch 32 --calls 13 -o profiles/r04_traffic.json

Per kernel: launches per sampling call (= per bench step), average duration from the kernel trace, FETCH_SIZE / WRITE_SIZE per dispatch
from the two PMC passes (each its own run), HBM-side bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE in KiB -> bytes (on gfx950 FETCH_SIZE
reports half of the bytes of a wide streaming read: MI355X_MICROARCH.md, HBM section; WRITE_SIZE is exact for 16-byte streaming stores).
Totals: bytes per step, and the two floors of this kernel decomposition (HBM at 8 TB/s spec and at the 6.29 TB/s a copy reaches; bf16 MFMA
at 2.5 PFLOP/s) that bench.py's `roofline_step` scales to its own run.  `--calls` = sampling calls in the kernel-trace run (bench.py
--steps K --warmup W --no-roofline: K + W); the PMC runs are --steps 1 --warmup 0 (one call: `dispatches` = launches per call)."""
import argparse
import json
import os
import re
import sys

CLASSES = [  # (class, regex on the kernel name)
    ("linear1", r"k_linear1_ts|k_gemm_glds<.*EpiLinear1"),
    ("linear2", r"k_linear2_ws|k_gemm_glds<.*EpiLinear2|k_tail<"),  # (the back half of a sub-block: profile class 1 of lsl_api.h)
    ("attention", r"k_attention"),
    ("ln_modulate", r"k_ln_modulate"),
    ("head", r"k_head_step"),
    ("embed", r"k_embed"),
]


def parse(path):
    stats, counters = {}, {}
    cur = None
    for line in open(path):
        m = re.match(r"\s*([\d.]+)%\s+calls\s+(\d+)\s+avg\s+([\d.]+) us\s+min\s+[\d.]+\s+max\s+[\d.]+\s+(.*)$", line)
        if m:
            stats[m.group(4).strip()] = {"calls": int(m.group(2)), "avg_us": float(m.group(3))}
            continue
        m = re.match(r"# counters \(mean per dispatch\): (.*)$", line)
        if m:
            cur = m.group(1).strip()
            counters.setdefault(cur, {})
            continue
        m = re.match(r"\s+([A-Z_a-z0-9]+)\s+([\d.]+)\s+\(dispatches (\d+)\)", line)
        if m and cur:
            counters[cur][m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return stats, counters


def bench_line(path):
    """The bench.py JSON line that tools/gpu.sh writes into the head of a summary ("# bench line of the traced run:")."""
    for line in open(path):
        if line.startswith("{") and '"metric"' in line:
            try:
                return json.loads(line)
            except ValueError:
                return None
    return None


def flops_per_step(cfg):
    """F_fwd = 6 N C D + depth * 4 N D (4D + 2M + L + T) per trajectory and evaluation (bench.py: flops_per_eval_per_traj)."""
    D, C, T, L = cfg["D"], cfg["C"], cfg["T"], cfg["L"]
    M, n = int(D * cfg["mlp_ratio"]), T * L
    return (6 * n * C * D + cfg["depth"] * 4 * n * D * (4 * D + 2 * M + L + T)) * cfg["state_updates"] * cfg["batch_per_gpu"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("--workload", default=None, help="(default: from the bench line in the summary)")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--tokens-per-traj", type=int, default=0)
    ap.add_argument("--calls", type=int, default=0, help="sampling calls in the kernel-trace run (default: steps + warmup of the bench line)")
    ap.add_argument("--updates", type=int, default=0, help="state updates per sampling call of the profiled command (bench.py --updates; 0 = not recorded)")
    ap.add_argument("--commit", default="", help="git commit of the library the profile was taken with (recorded; bench.py names it beside the derived figures)")
    ap.add_argument("--flops-per-step", type=float, default=0.0, help="algorithmic FLOPs of one step (default: SURVEY 8(d)'s formula on the bench line's config)")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    bl = bench_line(a.summary)
    if bl:
        cfg = bl["config"]
        a.workload = a.workload or cfg["workload"]
        a.batch = a.batch or cfg["batch_per_gpu"]
        a.tokens_per_traj = a.tokens_per_traj or cfg["T"] * cfg["L"]
        a.calls = a.calls or bl["steps"] + bl["warmup"]
        a.updates = a.updates or cfg["state_updates"]
        a.flops_per_step = a.flops_per_step or float(flops_per_step(cfg))
    if not (a.workload and a.batch and a.tokens_per_traj and a.calls):
        ap.error("no bench line in the summary: give --workload --batch --tokens-per-traj --calls")
    stats, counters = parse(a.summary)
    # tools/gpu.sh records the exit status of every rocprofv3 pass in the summary's head ("# pass <name> rc <code>"): a crashed FETCH_SIZE /
    # WRITE_SIZE pass leaves an under-counted (or empty) counter table, which must not become a traffic file
    failed = [ln.strip() for ln in open(a.summary) if re.match(r"# pass \S+ rc (?!0\b)", ln)]
    if failed:
        sys.exit("refusing " + a.summary + ": " + "; ".join(failed))
    kernels = {}
    for name, st in stats.items():
        c = counters.get(name, {})
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        fetch, n_f = c["FETCH_SIZE"]
        write, _ = c["WRITE_SIZE"]
        per_launch = int(2 * fetch * 1024 + write * 1024)
        kernels[name] = {"launches_per_step": n_f, "trace_calls": st["calls"], "avg_us": st["avg_us"], "fetch_kib": fetch, "write_kib": write,
                         "bytes": per_launch, "bytes_per_step": per_launch * n_f, "ms_per_step": st["avg_us"] * n_f * 1e-3}
        if st["calls"] != n_f * a.calls:
            kernels[name]["note"] = f"trace calls {st['calls']} != {n_f} launches x {a.calls} sampling calls"
    out = {"note": __doc__.split("\n\n")[2].replace("\n", " "), "source": os.path.relpath(a.summary), "workload": a.workload, "batch": a.batch,
           "tokens_per_step": a.batch * a.tokens_per_traj, "kernels": kernels}
    if bl and bl["config"].get("ln_fuse"):
        out["ln_fuse"] = True  # the handle fused the LayerNorm into linear1 (bench.py config.ln_fuse): class "linear1" reads the fp32 residual stream
    if bl and bl["config"].get("tail"):
        out["tail"] = True  # the handle ran the tail decomposition (bench.py config.tail): class "linear2" is k_tail, "linear1" computes q | k | v only
    if a.updates:
        out["state_updates"] = a.updates
    if a.commit:
        out["commit"] = a.commit
    # per class (what bench.py's roofline / roofline2 look up): the kernel of the class with the most bytes per step
    for cls, rx in CLASSES:
        ks = [(v["bytes_per_step"], k) for k, v in kernels.items() if re.search(rx, k)]
        if not ks:
            continue
        tot = sum(b for b, _ in ks)
        _, big = max(ks)
        out[cls] = {"kernel": big, "bytes": kernels[big]["bytes"], "fetch_kib": kernels[big]["fetch_kib"], "write_kib": kernels[big]["write_kib"],
                    "avg_us": kernels[big]["avg_us"], "bytes_per_step": tot, "ms_per_step": sum(kernels[k]["ms_per_step"] for _, k in ks)}
    out["tokens_per_launch"] = a.batch * a.tokens_per_traj  # (one pass per step at this batch: every block kernel sees all tokens)
    total = sum(v["bytes_per_step"] for v in kernels.values())
    kernel_ms = sum(v["ms_per_step"] for v in kernels.values())
    out["step"] = {"bytes": total, "kernel_ms": kernel_ms, "tb_per_s_over_kernel_time": total / (kernel_ms * 1e-3) / 1e12 if kernel_ms else None,
                   "hbm_floor_ms_at_8_tb_s": total / 8e12 * 1e3, "hbm_floor_ms_at_6_29_tb_s": total / 6.29e12 * 1e3,
                   "mfma_floor_ms_at_2_5_pflop_s": a.flops_per_step / 2.5e15 * 1e3 if a.flops_per_step else None,
                   "bytes_per_token": total / (a.batch * a.tokens_per_traj)}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out["step"], indent=1))
    for cls, _ in CLASSES:
        if cls in out:
            print(f"  {cls:12s} {out[cls]['bytes_per_step'] / 1e9:8.1f} GB/step  {out[cls]['ms_per_step']:7.1f} ms/step  "
                  f"{out[cls]['bytes_per_step'] / (out[cls]['ms_per_step'] * 1e-3) / 1e12:5.2f} TB/s  {out[cls]['kernel'][:60]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
