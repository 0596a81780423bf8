#!/bin/bash
# A/B of cosine_bf16_quad variants in ONE session on one box: ms of the cosine leg per 256-query batch at 12.5M x 1024 bf16
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for V in ${VARIANTS:-agpr pipe}; do
  for D in ${DBGS:-0 1}; do
    echo -n "variant=$V dbg=$D  "
    OI_LIB=ablation_$V OI_QUAD_DBG=$D python3 $R/tools/cosine_bf16_bench.py 12500000 1024 256 6 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['cosine_ms_per_batch'],3), 'ms', round(d['hbm_GBs']/4), 'GB/s streamed per pass(2 passes)')"
  done
done
done
