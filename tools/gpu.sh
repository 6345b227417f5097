#!/bin/bash
# Everything this repo runs on the GPU box besides pytest / bench.py, as ONE parameterised script (run it through gpurun from the repo root):
#
#   tools/gpu.sh trace   <tag> [bench args]            rocprofv3 kernel trace of `bench.py --steps 3 --warmup 1 <args>` -> gpurun_out/trace_<tag>.txt
#   tools/gpu.sh prof    <tag> [bench args]            kernel trace + SQ / TCC / FETCH_SIZE / WRITE_SIZE passes (each a run of its own, one step)
#                                                      -> gpurun_out/prof_<tag>.summary.txt, prof_<tag>.bench.json  (copy to profiles/rNN_<tag>.txt)
#   tools/gpu.sh profile <tag> [bench args]            the same for the driver's default command (trace over the default steps) + the wait / LDS counters
#   tools/gpu.sh pmcbin  <tag> <binary> [args]         the same passes on a stand-alone binary (tools/_exp/lin1_harness ...)
#   tools/gpu.sh workloads                             one throughput line per BASELINE.json configuration family (profiles/rNN_workloads.txt)
#   tools/gpu.sh ab      "<bench args>" "VAR=v .." ..  whole-bench A/B of environment arms (each arm in its own process and environment)
#   tools/gpu.sh libab   "<bench args>" lib.so ...     A/B of library builds (`product` = the in-tree one); the in-tree library is restored on exit
#   tools/gpu.sh harness <binary> "<shape>" ...        tools/_exp/<binary> on each shape: bit comparison + timing lines
#
# rocprofv3 always gets `python3 <script>` (or the binary) directly behind `--`: no env / bash -c hop (the profiler's preloaded library has
# already initialised the GPU, and a re-exec from such a process takes the box down).
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cmd=${1:?subcommand}; shift
mkdir -p "$root/gpurun_out"

SQ_WAIT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
SQ_INST="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
TCC="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"

pmc_pass() {  # <out dir> <name> "<counters>" <program...>   (the exit status of every pass goes into $out/passes.txt -> the summary's head)
  local out=$1 name=$2 ctr=$3; shift 3
  rocprofv3 --pmc $ctr --output-format csv -d "$out/$name" -- "$@" > "$out/$name.log" 2>&1
  echo "# pass $name rc $?" >> "$out/passes.txt"
}
bench_line() { grep "^{" "$1" | tail -n 1; }
short_line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d.get('breakdown') or {}
print('$1: %.2f traj/s  %.3f ms/step' % (d['value'], d['ms_per_step']) + (' | ms: ' + ' '.join('%s %.1f' % (k, v['ms']) for k, v in b.items()) if b else ''))"; }

case $cmd in
trace)
  tag=$1; shift
  out=$root/gpurun_out/trace_$tag; rm -rf "$out"; mkdir -p "$out"
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/bench.py" --steps ${STEPS:-3} --warmup ${WARM:-1} --no-cpu --no-extras --no-roofline "$@" > "$out/log.txt" 2>&1
  cd "$root"
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps ${STEPS:-3} --warmup ${WARM:-1} --no-cpu --no-extras --no-roofline $*"
    bench_line "$out/log.txt" | short_line "$tag"
    python3 tools/pmc_summary.py "$out" | head -${TOP:-24}; } | tee "gpurun_out/trace_$tag.txt"
  find "$out" -name "*.csv" -size +1M -delete
  ;;
prof|profile)
  tag=$1; shift
  out=$root/gpurun_out/prof_$tag; rm -rf "$out"; mkdir -p "$out"
  cd /tmp && export TMPDIR=/tmp
  one=(python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu --no-extras --no-roofline "$@")
  if [ "$cmd" = profile ]; then  # the driver's default step count in the trace, so its averages compare with bench.py's HIP-event figure
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --no-cpu --no-extras --no-roofline "$@" > "$out/trace.log" 2>&1
  else
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "${one[@]}" > "$out/trace.log" 2>&1
  fi
  echo "# pass trace rc $?" >> "$out/passes.txt"
  pmc_pass "$out" pmc_sq "$SQ_WAIT" "${one[@]}"
  pmc_pass "$out" pmc_sq2 "$SQ_INST" "${one[@]}"
  pmc_pass "$out" pmc_tcc "$TCC" "${one[@]}"
  pmc_pass "$out" pmc_fetch FETCH_SIZE "${one[@]}"
  pmc_pass "$out" pmc_write WRITE_SIZE "${one[@]}"
  cd "$root"
  dirs=("$out/trace"); for d in pmc_sq pmc_sq2 pmc_tcc pmc_fetch pmc_write; do [ -d "$out/$d" ] && dirs+=("$out/$d"); done
  { echo "# tools/gpu.sh $cmd $tag $*   (kernel trace, then PMC passes in runs of their own; one step per PMC pass)"
    cat "$out/passes.txt"
    echo "# bench line of the traced run:"; bench_line "$out/trace.log"
    python3 tools/pmc_summary.py "${dirs[@]}"; } > "gpurun_out/prof_$tag.summary.txt" 2>&1
  bench_line "$out/trace.log" > "gpurun_out/prof_$tag.bench.json"
  tail -3 "$out"/*.log | grep -iE "error|refus|Traceback|Segmentation" | head
  head -30 "gpurun_out/prof_$tag.summary.txt"
  find "$out" -name "*.csv" -size +2M -delete
  ;;
pmcbin)
  tag=$1; bin=$root/$2; shift 2
  out=$root/gpurun_out/pmc_$tag; rm -rf "$out"; mkdir -p "$out"
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "$bin" "$@" > "$out/trace.log" 2>&1
  pmc_pass "$out" pmc_sq "$SQ_WAIT" "$bin" "$@"
  pmc_pass "$out" pmc_sq2 "$SQ_INST" "$bin" "$@"
  pmc_pass "$out" pmc_sq3 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_TRANS SQ_LDS_DATA_FIFO_FULL" "$bin" "$@"
  pmc_pass "$out" pmc_tcc "$TCC" "$bin" "$@"
  pmc_pass "$out" pmc_fetch FETCH_SIZE "$bin" "$@"
  pmc_pass "$out" pmc_write WRITE_SIZE "$bin" "$@"
  cd "$root"
  python3 tools/pmc_summary.py "$out/trace" "$out/pmc_sq" "$out/pmc_sq2" "$out/pmc_sq3" "$out/pmc_tcc" "$out/pmc_fetch" "$out/pmc_write" > "gpurun_out/pmc_$tag.summary.txt" 2>&1
  head -40 "gpurun_out/pmc_$tag.summary.txt"
  find "$out" -name "*.csv" -size +2M -delete
  ;;
workloads)
  cd "$root"
  run() {
    python3 bench.py --no-cpu --no-extras "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('%-18s B=%-5d updates=%-4d  %10.2f traj/s  %9.3f ms/call  whole-path %6.1f TFLOP/s (%.1f %% of bf16 peak)' % (c['workload'], c['batch_per_gpu'], c['state_updates'], d['value'], d['ms_per_step'], r['whole_path_tflops'], 100*r['whole_path_frac']))"
  }
  run --workload md17_bench --steps 3 --warmup 1
  run --workload md17_bench --batch 1 --steps 20 --warmup 5
  run --workload md17_bench --batch 8 --steps 3 --warmup 1
  run --workload md17_ref --steps 60 --warmup 20
  run --workload md17_ref --batch 64 --steps 5 --warmup 2
  run --workload pedestrian_scene --steps 50 --warmup 5
  run --workload pedestrian --batch 160 --steps 20 --warmup 5
  run --workload pedestrian --steps 10 --warmup 3
  run --workload nba --steps 3 --warmup 1
  run --workload nba --batch 64 --steps 5 --warmup 2
  run --workload peptide --steps 2 --warmup 1
  ;;
ab)
  cd "$root"
  args=$1; shift
  script=bench.py; [ -n "${EXP:-}" ] && script=tools/bench_exp.py   # EXP=1: the -DLSL_EXPERIMENTS build (tuning knobs, probes)
  for arm in "$@"; do  # `env` scopes every variable of the arm to its own process
    env $arm python3 $script $args --no-cpu --no-extras --breakdown 2>&1 | tail -1 | short_line "[$arm]"
  done
  ;;
libab)
  cd "$root"
  args=$1; shift
  keep=$(mktemp /tmp/_product.XXXXXX.so)  # (a path of this run's own: two concurrent runs must not share the backup)
  cp lam_slide_amd/liblamslide_hip.so "$keep"
  trap 'cp "$keep" "$root/lam_slide_amd/liblamslide_hip.so"; rm -f "$keep"' EXIT
  for round in 1 2; do
    for lib in "$@"; do
      if [ "$lib" = product ]; then cp "$keep" lam_slide_amd/liblamslide_hip.so; else cp "$lib" lam_slide_amd/liblamslide_hip.so; fi
      python3 bench.py $args --no-cpu --no-extras --breakdown 2>&1 | tail -1 | short_line "$lib"
    done
  done
  ;;
harness)
  cd "$root"
  bin=$1; shift
  for shape in "$@"; do
    echo "== $bin: $shape"
    timeout 180 tools/_exp/$bin $shape 2>&1 | grep -E "${GREP:-BITS|DIFF|round [12]|mismatch|rror|unsupported|grid|us }"
  done
  ;;
*)
  echo "unknown subcommand $cmd" >&2; exit 2 ;;
esac
