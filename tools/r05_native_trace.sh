#!/bin/bash
# Round 5: kernel trace of bench.py --exchange native at world 1 (process group of one rank): is the GPU starved or do the lanes serialise?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
OI_BENCH_FORCE_DIST=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/bench.py --docs 1250000 --no-cpu-baseline --no-text-paths --no-stream-side --steps 60 --warmup 10 --latency-batches 2 --latency-warmup 1 --exchange ${2:-native} --lanes 2 > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 - "$OUT" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pf_stage_queries" in r["Kernel_Name"]]
# the timed region: 60 steps after 10 warm-ups (+ calibration batches for the torch path): take stage launches 30..33 from the END of the first 75
a, b = idx[40], idx[43]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    print("%9.1f +%8.1f q=%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                    r["Queue_Id"], r["Kernel_Name"].split("(")[0][-36:]))
P
