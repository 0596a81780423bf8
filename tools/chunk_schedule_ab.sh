#!/bin/bash
# A/B of the cosine chunk schedule (first chunk rows = max(8192, 32*depth) * OI_FIRST_CHUNK_MULT, then x OI_CHUNK_GROWTH)
# at a shard size and at 10M rows; needs the ablation build (tools/build_ablation.sh).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for cfg in "1 8" "2 64" "4 64" "1 64" "2 16" "1 16"; do
  set -- $cfg
  echo -n "mult=$1 growth=$2  shard: "
  OI_LIB=ablation OI_FIRST_CHUNK_MULT=$1 OI_CHUNK_GROWTH=$2 python3 $R/tools/shard_step_bench.py 1250000 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lists %.3f ms  cosine %.3f select %.3f'%(d['lists_ms'], d['lists_kernels_ms']['cosine'], d['lists_kernels_ms']['select']))"
done
