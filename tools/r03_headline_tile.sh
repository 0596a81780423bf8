#!/bin/bash
# Tile-size sweep of headline_scan_kernel (needs tools/build_ablation.sh): kernel ms at 10M titles.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for T in ${TILES:-160 192 224 240 256}; do
  echo -n "tile=$T  "
  OI_LIB=ablation OI_HEADLINE_TILE=$T python3 $R/tools/headline_bench.py 10000000 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],4), d['bit_exact_vs_oracle_on_slice'])"
done
