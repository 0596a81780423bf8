#!/bin/bash
# NOTE: the OI_* A/B and ablation switches only exist in an ablation build of the library:
#   OI_EXTRA_HIPCC_FLAGS=-DOI_ABLATION python -m openintel_amd.build --force
# (the product build ignores them; rebuild without the flag afterwards).
# A/B of the corpus-chunk schedule (OI_CHUNK_GROWTH, OI_FIRST_CHUNK_MULT): full step, shard step, batch-1 query.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/growth
mkdir -p $OUT
for cfg in "8 1" "16 1" "32 1" "64 1" "16 2" "8 1"; do
  set -- $cfg
  export OI_CHUNK_GROWTH=$1 OI_FIRST_CHUNK_MULT=$2
  timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --steps 30 > $OUT/full_$1_$2.json || exit 1
  timeout -k 10 100 python3 $R/tools/shard_step_bench.py 1250000 50 > $OUT/shard_$1_$2.json || exit 1
  timeout -k 10 100 python3 $R/bench.py --docs 1000000 --batch 1 --depth 100 --steps 300 --no-cpu-baseline > $OUT/b1_$1_$2.json || exit 1
  python3 - <<P
import json
f=json.load(open("$OUT/full_$1_$2.json")); s=json.load(open("$OUT/shard_$1_$2.json")); b=json.load(open("$OUT/b1_$1_$2.json"))
print("growth=$1 mult=$2 full %.3f ms (launches %.0f) | shard %.3f ms | b1 %.4f ms p50 %.4f (launches %.0f)" % (f["ms_per_step"], f["roofline"]["launches_per_step"], s["lists_ms"], b["ms_per_step"], b["p50_ms"], b["roofline"]["launches_per_step"]), flush=True)
P
done
