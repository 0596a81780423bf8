#!/bin/bash
# Kernel trace of the BM25 kernels alone (tools/bm25_bench.py): per-launch durations, registers, scratch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-trace_bm25}
N=${2:-10000000}
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/bm25_bench.py $N 3 64 ${3:-all} > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/t/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "bm25" in r["Kernel_Name"] or "select" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for r in rows[-40:]:
    print("%8.1f us %-28s grid=%s wg=%s vgpr=%s sgpr=%s scratch=%s lds=%s"%((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"].split("(")[0][-28:],r["Grid_Size_X"],r["Workgroup_Size_X"],r.get("VGPR_Count"),r.get("SGPR_Count"),r.get("Scratch_Size"),r.get("LDS_Block_Size")))
PY
