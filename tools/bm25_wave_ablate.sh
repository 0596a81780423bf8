#!/bin/bash
# Ablation ladder of bm25_wave_kernel (needs tools/build_ablation.sh): kernel time with the task cut short at
# 1 = bounds only, 2 = + front loads, 3 = + pass A and ranks, 4 = + pass B, 0 = full.  Results of 1..4 are wrong by construction.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-10000000}
for L in ${LEVELS:-1 2 6 5 3 4 0}; do
  echo -n "dbg=$L  "
  OI_LIB=ablation OI_BM25_MODE=wave OI_BM25_WAVE_DBG=$L python3 $R/tools/bm25_bench.py $N 10 64 wave-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['wave'])"
done
