#!/bin/bash
# Where the per-ticker reduction's time goes (needs tools/build_ablation.sh): sizes, and ablations of the kernel's two halves.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
pick='import json,sys; d=json.loads(sys.stdin.read()); print(d["posts"], d["tickers"], round(d["segmented_summary_kernel_ms"],4), round(d["scan_kernel_ms"],4))'
for A in "10000000 100000" "1000000 10000" "3000000 30000" "10000000 1000000"; do
  echo -n "posts tickers = $A: "; OI_LIB=ablation python3 $R/tools/scan_bench.py $A 20 2>/dev/null | python3 -c "$pick"
done
for D in 1 2; do
  echo -n "dbg=$D (10M / 100K): "; OI_LIB=ablation OI_SEG_DBG=$D python3 $R/tools/scan_bench.py 10000000 100000 20 2>/dev/null | python3 -c "$pick"
done
