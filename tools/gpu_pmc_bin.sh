#!/bin/bash
# rocprofv3 PMC passes (each in its own run) on a stand-alone binary: tools/gpu_pmc_bin.sh <tag> <binary> [args...]
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
bin=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $bin $* > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq -- $bin $* > $out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc_sq2 -- $bin $* > $out/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_TRANS SQ_LDS_DATA_FIFO_FULL --output-format csv -d $out/pmc_sq3 -- $bin $* > $out/pmc_sq3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_tcc -- $bin $* > $out/pmc_tcc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $bin $* > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $bin $* > $out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $out/trace $out/pmc_sq $out/pmc_sq2 $out/pmc_sq3 $out/pmc_tcc $out/pmc_fetch $out/pmc_write > gpurun_out/pmc_$tag.summary.txt 2>&1
tail -3 $out/*.log | grep -iE "error|refus|Traceback" | head
find $out -name "*.csv" -size +2M -delete
