#!/bin/bash
# Counters of the headline gate's scan and of the batch callers' per-ticker reduction (GPU box): HBM bytes, VALU issue,
# waits, LDS.  Separate --pmc passes with --kernel-trace only; FETCH_SIZE and WRITE_SIZE each in a pass of its own (the guide's
# rule: together they do not fit and the run aborts).  Every pass is bounded by `timeout` and reports as it ends.  Writes gpurun_out/<tag>/text_pmc.json.
cd "$(dirname "$0")/.."
R=$(pwd); T=${1:-text_pmc}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace -d $R/$O/h$i -o pmc --output-format csv -- python3 $R/tools/headline_bench.py 10000000 3 > $R/$O/h$i.log 2>&1 && echo "headline pmc set $i ok: $set" || echo "headline pmc set $i failed: $set"
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace -d $R/$O/s$i -o pmc --output-format csv -- python3 $R/tools/scan_bench.py 10000000 100000 3 > $R/$O/s$i.log 2>&1 && echo "scan pmc set $i ok: $set" || echo "scan pmc set $i failed: $set"
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for tag, names in (("h", ("headline_scan_kernel",)), ("s", ("social_summary_segmented_kernel", "lexicon_scan_kernel"))):
    for f in sorted(glob.glob('$R/$O/%s*/pmc_counter_collection.csv' % tag)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            for nm in names:
                if nm in r['Kernel_Name']:
                    acc[nm][r['Counter_Name']].append(float(r['Counter_Value']))
        for nm, cs in acc.items():
            out.setdefault(nm, {}).update({k: sum(v) / len(v) for k, v in cs.items()})
json.dump(out, open('$R/$O/text_pmc.json', 'w'), indent=1)
print(json.dumps(out))
PY
rm -rf $R/$O/h[0-9] $R/$O/s[0-9]
