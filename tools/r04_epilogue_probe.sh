#!/bin/bash
# Per-chunk launch durations of the screen and the whole step's span (rocprofv3 kernel trace of tools/step_ab.py) for a list of
# "ENV=... ENV=..." configurations, same box.  Default: the previous build (OI_LIB=ablation_old), the current one, the current
# one with thresholds nothing passes (OI_SCREEN_TAU_MAX: what the filter + append costs; results of that run are wrong on purpose).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/epi4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in ${CFGS:-"OI_LIB=ablation_old" "OI_LIB=ablation" "OI_LIB=ablation_old" "OI_LIB=ablation"}; do
  i=$((i+1))
  for kv in $cfg; do export $kv; done
  rocprofv3 --kernel-trace --output-format csv -d $OUT/run$i -- python3 $R/tools/step_ab.py 10000000 20 > $OUT/run$i.json 2> $OUT/run$i.err || exit 1
  unset OI_SCREEN_TAU_MAX OI_BM25_LATE
  python3 - <<P
import csv, glob, collections
f = glob.glob("$OUT/run$i/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "cosine_screen_filter" in r["Kernel_Name"]]
per = collections.defaultdict(list)
for j, x in enumerate(d):
    per[j % 4].append(x)
med = lambda v: sorted(v)[len(v) // 2]
st = [int(r["Start_Timestamp"]) for r in rows if "pf_stage_queries" in r["Kernel_Name"]]
spans = [(b - a) / 1e3 for a, b in zip(st, st[1:])]
other = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split()[-1][:28]
    if "cosine_screen" not in n:
        other[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("[$cfg] chunks", {k: round(med(v), 1) for k, v in per.items()}, "sum %.1f" % sum(med(v) for v in per.values()), "| step span median %.1f us" % med(spans),
      "|", {k: round(med(v), 1) for k, v in other.items() if k.startswith(("select", "pf_rescore", "rrf", "bm25_stream"))}, flush=True)
P
done
