#!/bin/bash
# phase clock of the resident kernel under each LSL_RES_SKIP mask given on the command line (tools build)
for m in "$@"; do LSL_RES_SKIP=$m python tools/resident_probe.py --stamps 2>&1 | grep -A1 "phase cycles" | head -2; done
