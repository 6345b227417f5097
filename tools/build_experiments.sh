#!/bin/bash
# Tools-only build of the library with -DLSL_EXPERIMENTS: timing probes (LSL_PROBE: results WRONG on purpose), the rejected GEMM
# structures (tools/experiments/k_gemm_pp, k_gemm_drain, variants 13 / 20-22 / 30), the scalar output head and every LSL_* tuning knob.
# The product library (lam_slide_amd/liblamslide_hip.so, built by __graft_entry__.build()) contains none of this.
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/tools/_exp"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -DLSL_EXPERIMENTS -I"$root/tools/experiments" -I"$root/lam_slide_amd/csrc" \
    -o "$root/tools/_exp/liblamslide_hip_exp.so" "$root/lam_slide_amd/csrc/lsl_api.hip"
echo "built $root/tools/_exp/liblamslide_hip_exp.so"
