#!/bin/bash
# Round 5: one rank's shard of an 8-GPU job (1.25M x 768, 64 queries) through bench.py with a process group of ONE rank (real RCCL calls):
# does a third calibrated lane pay?   bash tools/r05_lanes_ab.sh TAG   ->  gpurun_out/TAG/lanes.txt
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$R/gpurun_out/${1:-lanesab}; mkdir -p $OUT; : > $OUT/lanes.txt
for rep in 1 2 3; do
  for cfg in "torch 2" "torch 3" "native 2" "native 3" "torch 4"; do
    set -- $cfg
    OI_BENCH_FORCE_DIST=1 python3 $R/bench.py --gpus 1 --docs 1250000 --exchange $1 --lanes $2 --steps 100 --warmup 10 --no-cpu-baseline --no-text-paths --latency-batches 1 --latency-warmup 0 > $OUT/run.json 2> $OUT/run.err || { echo "FAILED $cfg"; tail -n 5 $OUT/run.err; exit 1; }
    python3 - "$cfg" $OUT/run.json >> $OUT/lanes.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
c=d.get("config",{})
print(sys.argv[1], "ms_per_step %.4f"%d["ms_per_step"], "qps %.0f"%d["value"], "lanes_kept", c.get("lanes_kept", c.get("lanes")), "calib", json.dumps(c.get("calibration", c.get("pipeline", "")))[:200])
PY
  done
done
cat $OUT/lanes.txt
