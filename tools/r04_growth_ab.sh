#!/bin/bash
# A/B of the corpus-chunk schedule on round 4's code (ablation build: bash tools/build_ablation.sh; OI_LIB=ablation):
# full step at 10M rows (tools/step_ab.py) and the 1.25M-row shard step, per (growth, first-chunk multiplier).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/growth4
mkdir -p $OUT
export OI_LIB=ablation
for cfg in "8 1" "64 1" "64 2" "32 1" "8 1" "64 1"; do
  set -- $cfg
  export OI_CHUNK_GROWTH=$1 OI_FIRST_CHUNK_MULT=$2
  timeout -k 10 200 python3 $R/tools/step_ab.py 10000000 40 > $OUT/full_$1_$2.json 2> $OUT/full_$1_$2.err || exit 1
  timeout -k 10 100 python3 $R/tools/shard_step_bench.py 1250000 50 > $OUT/shard_$1_$2.json 2> $OUT/shard_$1_$2.err || exit 1
  python3 - <<P
import json
f=json.load(open("$OUT/full_$1_$2.json")); s=json.load(open("$OUT/shard_$1_$2.json"))
print("growth=$1 mult=$2 full %s ms | shard lists %.3f ms | %s" % (f["ms_per_step"], s["lists_ms"], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in s.items() if not isinstance(v, (dict, list))}), flush=True)
P
done
