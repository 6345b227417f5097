#!/bin/bash
# Per-workload rocprofv3 profile: kernel-trace stats + the SQ / FETCH / WRITE PMC passes (each its own run) -> gpurun_out/prof_<tag>.summary.txt
#   tools/gpu_prof_wl.sh <tag> --workload <w> [--batch B] ...
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --no-roofline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $B > $out/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq2 -- python3 $B > $out/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $B > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $B > $out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $out/trace $out/pmc_sq2 $out/pmc_fetch $out/pmc_write > gpurun_out/prof_$tag.summary.txt 2>&1
grep "^{" $out/trace.log | tail -n 1 > gpurun_out/prof_$tag.bench.json
find $out -name "*.csv" -size +2M -delete
