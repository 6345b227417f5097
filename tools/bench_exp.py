#!/usr/bin/env python3
"""bench.py against the -DLSL_EXPERIMENTS build (tools/build_experiments.sh): the only way the LSL_GEMM / LSL_PROBE / LSL_NT / ...
knobs (`EXP=1 tools/gpu.sh ab ...`) take effect.  The library path is switched in this process only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lam_slide_amd import _lib  # noqa: E402

exp = os.path.join(ROOT, os.environ["LSL_LIB"]) if os.environ.get("LSL_LIB") else os.path.join(ROOT, "tools", "_exp", "liblamslide_hip_exp.so")
if not os.path.exists(exp):
    raise SystemExit("run tools/build_experiments.sh first")
_lib.LIB_PATH = exp
import bench  # noqa: E402

sys.exit(bench.main())
