#!/bin/bash
# A/B (ablation build): the stream kernel's two phases (OI_BM25_TWO_PHASE=1) vs the per-term impact floors (default) --
# BM25 alone, the full step at 10M rows, the 1.25M-row shard step; alternating on one box.
R=$GRAFT_REPO_ROOT
export OI_LIB=ablation
for rep in 1 2; do
for e in "OI_BM25_TWO_PHASE=1" "A=1"; do
  b=$(env $e python3 $R/tools/bm25_bench.py 10000000 10 64 stream-only 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['stream']['ms_per_batch'],4))")
  f=$(env $e python3 $R/tools/step_ab.py 10000000 40 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  s=$(env $e python3 $R/tools/shard_step_bench.py 1250000 50 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['lists_ms'],4), round(d['qps_8gpu_pipelined_if_exchange_hidden']))")
  echo "[$e] bm25 alone $b ms | full step $f ms | shard lists, pipelined QPS: $s"
done
done
