#!/bin/bash
# What the card's clocks and power do while bench.py's timed loops run (an explanation for the 0.82-0.85 band of roofline.frac
# across boxes and runs): rocm-smi samples every 0.5 s beside a longer timed region.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/smi
mkdir -p $OUT
python3 $R/bench.py --steps 400 --warmup 20 --no-cpu-baseline --latency-batches 200 > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
for i in $(seq 1 60); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "--- t=$i" >> $OUT/smi.txt
  rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -v "^=\|^$" >> $OUT/smi.txt
  sleep 0.5
done
wait $BP
python3 - <<P
import json,re
d=json.load(open("$OUT/bench.json")); print("bench", d["value"], d["ms_per_step"], d["p50_ms"], d["roofline"]["frac"])
txt=open("$OUT/smi.txt").read()
for key in ("sclk", "mclk", "fclk", "socclk", "Average Graphics Package Power", "Current Socket Graphics Package Power", "Temperature (Sensor junction)", "Temperature (Sensor memory)", "GPU use"):
    vals=re.findall(re.escape(key)+r"[^:\n]*:\s*\(?([0-9.]+)", txt)
    if vals: print(key, "min", min(map(float,vals)), "max", max(map(float,vals)), "n", len(vals), "last", vals[-5:])
P
