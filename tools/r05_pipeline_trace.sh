#!/bin/bash
# Round 5: kernel trace of the native pipeline (tools/r05_pipeline_probe.py) -- which hardware queue every lane's kernels ran
# on and whether two lanes' launches overlap in time.  Usage: tools/r05_pipeline_trace.sh <outdir> <n_docs> <steps>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/r05_pipeline_probe.py $2 $3 > $OUT/probe.json 2> $OUT/probe.err || exit 1
python3 - "$OUT" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last 400 launches: the 3-lane run's steady state; before it the 2-lane run ... print a window of the 2-lane run
names = [r["Kernel_Name"].split("(")[0][-34:] for r in rows]
qs = sorted(set(r["Queue_Id"] for r in rows))
print("queues seen:", qs)
idx = [i for i, r in enumerate(rows) if "rrf_kernel" in r["Kernel_Name"]]
# pick a window around 60 % of the rrf launches (inside the lanes=2 timed loop for the default step counts)
a = idx[int(len(idx) * 0.55)]
b = idx[int(len(idx) * 0.55) + 3]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    print("%9.1f +%8.1f q=%s %s grid=%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                             r["Queue_Id"], r["Kernel_Name"].split("(")[0][-36:], r["Grid_Size_X"]))
P
