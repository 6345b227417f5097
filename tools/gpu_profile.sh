#!/bin/bash
# rocprofv3 passes on the GPU box: kernel-trace stats, then PMC passes (each in its own run, no trace domains mixed in).
# Usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# kernel trace: the default bench command (what the driver runs) minus the CPU leg and the small-batch / stage-1 legs, so that every
# launch in the trace has the timed region's size and the averages compare with bench.py's HIP-event figure; PMC passes: one step
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --no-roofline $* > $out/trace.log 2>&1
B="$GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --no-roofline $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq -- python3 $B > $out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc_sq2 -- python3 $B > $out/pmc_sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_tcc -- python3 $B > $out/pmc_tcc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $B > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $B > $out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $out/trace $out/pmc_sq $out/pmc_sq2 $out/pmc_tcc $out/pmc_fetch $out/pmc_write > gpurun_out/prof_$tag.summary.txt 2>&1
tail -3 $out/*.log | grep -iE "error|refus|Traceback" | head
# keep only the summaries (raw CSVs are large)
find $out -name "*.csv" -size +2M -delete
