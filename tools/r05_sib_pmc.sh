#!/bin/bash
# Round 5: HBM bytes per launch of the quad kernel with and without sibling workgroups (FETCH_SIZE, its own pass; the guide's
# gfx950 correction x2 on wide coalesced streams).  Usage: tools/r05_sib_pmc.sh <outdir> [n_docs]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
N=${2:-12500000}
mkdir -p $OUT
for sib in 0 2; do
  OI_LIB=${OI_LIB:-ablation} timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/sib$sib -- python3 $R/tools/r05_sib_ab.py $N 3 $sib > $OUT/sib$sib.json 2> $OUT/sib$sib.err || exit 1
  python3 - "$OUT/sib$sib" $sib $N <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
tot, n, ns = 0.0, 0, 0
seen = set()
for r in csv.DictReader(open(f)):
    if "cosine_bf16_quad" not in r["Kernel_Name"] or r["Counter_Name"] != "FETCH_SIZE":
        continue
    tot += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n += 1; ns += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = int(sys.argv[3])
alg = rows * 1024 * 2
launches_per_batch = 12 if sys.argv[2] == "0" else 6
batches = n / launches_per_batch
print("sib=%s: %d quad launches (%.1f batches), HBM read %.2f GB per batch (x2-corrected FETCH_SIZE) vs %.2f GB of rows; %.3f ms per batch under the counter pass"
      % (sys.argv[2], n, batches, tot * 1024 * 2 / batches / 1e9, alg / 1e9, ns / 1e6 / batches))
P
done
