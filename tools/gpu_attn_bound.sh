#!/bin/bash
set -u
mkdir -p gpurun_out
{
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "attention_softmax_shift or stage_taps or edge_shapes or forward" 2>&1 | tail -15
python -m pytest tests/test_hip_cfg.py -m gpu -x -q 2>&1 | tail -5
bash tools/gpu_wl.sh peptide "LSL_ATTN_BOUND=0" "LSL_ATTN_BOUND=1"
bash tools/gpu_wl.sh md17_bench "LSL_ATTN_BOUND=0" "LSL_ATTN_BOUND=1"
bash tools/gpu_wl.sh nba "LSL_ATTN_BOUND=0" "LSL_ATTN_BOUND=1"
} > gpurun_out/attn_bound.log 2>&1
cat gpurun_out/attn_bound.log
