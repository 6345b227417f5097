#!/bin/bash
# ping-pong linear1 variants on the cfg-2 launch: tools/_exp/lin1_<variant> ...
set -u
for v in "$@"; do
  echo "== $v"; timeout 120 tools/_exp/lin1_$v 245760 512 16 2 20 256 0 2>&1 | grep -E "BITS|DIFF|round [12]|per block|MHz" | head -14
done
