#!/bin/bash
# Round 5: where the 128-query bf16 kernel's time goes -- matrix-pipe busy share, effective clock and wait share (PMC, two passes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
N=${2:-12500000}
mkdir -p $OUT
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  OI_LIB=${OI_LIB:-ablation} timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $R/tools/r05_sib_ab.py $N 3 2 > $OUT/p$i.json 2> $OUT/p$i.err || exit 1
done
python3 - "$OUT" <<'P'
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(float); ns = 0; seen = set()
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "cosine_bf16_q" not in r["Kernel_Name"]:
            continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        k = (f, r["Dispatch_Id"])
        if k not in seen and "GRBM" in r["Counter_Name"]:
            seen.add(k); ns += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print({k: round(v) for k, v in agg.items()})
if ns:
    print("effective clock GHz", agg["GRBM_GUI_ACTIVE"] / 8 / ns, "(GRBM_GUI_ACTIVE / 8 XCDs / ns)")
print("MFMA busy / (GUI_ACTIVE/8 * 1024 SIMDs):", agg["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, agg["GRBM_GUI_ACTIVE"] / 8 * 1024))
print("wait share SQ_WAIT_ANY / SQ_WAVE_CYCLES:", agg["SQ_WAIT_ANY"] / max(1.0, agg["SQ_WAVE_CYCLES"]) if agg["SQ_WAVE_CYCLES"] else None)
print("LDS bank conflict share:", agg["SQ_LDS_BANK_CONFLICT"] / max(1.0, agg["SQ_LDS_IDX_ACTIVE"]))
P
