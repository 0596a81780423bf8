#!/bin/bash
# Where the lexicon scan's cycles go (GPU box): LDS activity / conflicts, wait cycles, wave cycles.  gpurun_out/lex_pmc/.
set -e
cd "$(dirname "$0")/.."
R=$(pwd); O=gpurun_out/lex_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/$O/p$i -o pmc --output-format csv -- python3 $R/tools/lexicon_ladder.py 10000000 3 > $R/$O/p$i.log 2>&1 || echo "pmc set $i failed: $set"
done
python3 - <<PY
import csv, glob, collections, json
for f in sorted(glob.glob('$R/$O/p*/pmc_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'lexicon' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(json.dumps({k: sum(v) / len(v) for k, v in acc.items()}))
PY
