#!/bin/bash
# Ablation ladder of headline_scan_kernel (needs tools/build_ablation.sh): 1 = staging + title bits + results only,
# 2 = + token pass (no verification), 3 = + token packing, 4 = no keyword bookkeeping, 5 = no company patterns, 0 = full.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for L in ${LEVELS:-1 2 3 4 5 0}; do
  echo -n "dbg=$L  "
  OI_LIB=ablation OI_HEADLINE_DBG=$L python3 $R/tools/headline_bench.py 10000000 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],4))"
done
OI_LIB=ablation OI_HEADLINE_TIMING=1 python3 $R/tools/headline_bench.py 10000000 3 2>&1 | grep -i "headline timing" | tail -2
