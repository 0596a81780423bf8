#!/bin/bash
# Workgroups-per-CU sweep of lexicon_scan_kernel's tile loop (needs tools/build_ablation.sh): kernel ms at 10M posts.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for G in ${GRIDS:-16 24 32 48 64 80 16}; do
  echo -n "wgs_per_cu=$G  "
  OI_LIB=ablation OI_LEX_GRID=$G python3 $R/tools/lexicon_bench.py 10000000 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['lexicon_kernel_ms'],4), round(d['fused_scan_summary']['kernel_ms'],4))"
done
