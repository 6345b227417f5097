#!/bin/bash
# VGPR / scratch / occupancy of every kernel whose mangled name matches $1 (default k_gemm_glds)
cd /tmp && hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -o /tmp/_regs.so /root/repo/lam_slide_amd/csrc/lsl_api.hip ${REGS_FLAGS:-} -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy" | grep -A4 "${1:-k_gemm_glds}" | sed 's/.*Function Name: /K /; s/.*remark: *//; s/\[-Rpass.*//' | grep -v "^--" | paste - - - - - | sed 's/  */ /g; s/GemmArgsT6_//'
