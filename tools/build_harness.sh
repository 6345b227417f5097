#!/bin/bash
# Stand-alone kernel harnesses (tools/{lin1,lin2,tail}_harness.hip -> tools/_exp/<name>): bit comparison against the tile kernels + timing.
#   tools/build_harness.sh lin1|lin2|tail [name] [-DFLAGS ...]
# The product kernels carry no timing scaffolding.  The probe arms of linear1 / linear2 (skip the epilogue / the MFMAs / the stores / the
# requests, cycle stamps per step phase: results WRONG when set) live in tools/experiments/lin{1,2}_probes.patch; a build that passes
# -DLIN1_PROBE=.. / -DLIN2_PROBE=.. (or any other LIN1_* / LIN2_* arm of the patch) gets the patched copy of the kernel under tools/_exp/.
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
which=${1:?lin1|lin2|tail}; shift
name=${which}_harness
if [ $# -gt 0 ] && [ "${1#-}" = "$1" ]; then name=$1; shift; fi
mkdir -p "$root/tools/_exp"
flags=("$@")
case "$which" in
lin1|lin2)
  if printf '%s\n' "${flags[@]:-}" | grep -qiE "PROBE|PRIO|ALL_PLAIN|DRAIN_NT|HI_FIRST|DEPHASE"; then
    cp "$root/lam_slide_amd/csrc/k_${which}.hip.h" "$root/tools/_exp/k_${which}_probed.hip.h"
    patch -s "$root/tools/_exp/k_${which}_probed.hip.h" < "$root/tools/experiments/${which}_probes.patch"
    sed -i 's#"common.hip.h"#"../../lam_slide_amd/csrc/common.hip.h"#' "$root/tools/_exp/k_${which}_probed.hip.h"
    flags+=("-D$(echo $which | tr a-z A-Z)_PROBED")
  fi ;;
tail)
  if printf '%s\n' "${flags[@]:-}" | grep -q "TAIL_STAMP"; then  # cycle sums per phase (z rows / O phase / a rows / mlp / epilogue)
    cp "$root/lam_slide_amd/csrc/k_tail.hip.h" "$root/tools/_exp/k_tail_stamped.hip.h"
    patch -s "$root/tools/_exp/k_tail_stamped.hip.h" < "$root/tools/experiments/tail_stamps.patch"
    sed -i 's#"common.hip.h"#"../../lam_slide_amd/csrc/common.hip.h"#' "$root/tools/_exp/k_tail_stamped.hip.h"
    flags+=("-DTAIL_STAMPED")
  fi ;;
esac
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value -I"$root/tools" "${flags[@]:-}" "$root/tools/${which}_harness.hip" -o "$root/tools/_exp/$name"
echo "built tools/_exp/$name"
