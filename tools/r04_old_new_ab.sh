#!/bin/bash
# Same-box A/B of two ablation builds (OI_LIB=ablation_old: a build of the previous commit kept beside the current one):
# full step at 10M rows and the 1.25M-row shard step, alternating.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/oldnew4
mkdir -p $OUT
i=0
for cfg in "OI_LIB=ablation_old" "OI_LIB=ablation" "OI_LIB=ablation_old" "OI_LIB=ablation"; do
  i=$((i+1))
  env $cfg timeout -k 10 200 python3 $R/tools/step_ab.py 10000000 40 > $OUT/full_$i.json 2> $OUT/full_$i.err || exit 1
  env $cfg timeout -k 10 100 python3 $R/tools/shard_step_bench.py 1250000 50 > $OUT/shard_$i.json 2> $OUT/shard_$i.err || exit 1
  python3 - <<P
import json
f=json.load(open("$OUT/full_$i.json")); s=json.load(open("$OUT/shard_$i.json"))
print("[$cfg] full %s ms (checksum %s) | shard lists %.3f ms | 8gpu pipelined-if-hidden QPS %.0f" % (f["ms_per_step"], f["docs_checksum"], s["lists_ms"], s["qps_8gpu_pipelined_if_exchange_hidden"]), flush=True)
P
done
