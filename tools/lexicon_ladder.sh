#!/bin/bash
# Ablation ladder of the lexicon scan, times and VALU/SALU/LDS instruction counts per level (GPU box).
#   bash tools/lexicon_ladder.sh [levels]   (OI_LEXICON_V2=1 for the second-generation kernel: levels 0..5)
# Results under gpurun_out/lex_ladder/.
set -e
cd "$(dirname "$0")/.."
R=$(pwd)
O=gpurun_out/lex_ladder; mkdir -p $O
export OI_LIB=ablation
LV=${1:-"0 1 2 3"}
for d in $LV; do OI_LEX_DBG=$d python tools/lexicon_ladder.py 10000000 8 2>/dev/null | tail -1 | tee -a $O/ladder.jsonl; done
cd /tmp && export TMPDIR=/tmp
for d in $LV; do
  export OI_LEX_DBG=$d
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --kernel-trace -d $R/$O/pmc_l$d -o pmc --output-format csv -- python3 $R/tools/lexicon_ladder.py 10000000 3 > $R/$O/pmc_l$d.log 2>&1 || echo "pmc level $d failed"
done
python3 - <<PY
import csv, glob, collections, json
for f in sorted(glob.glob('$R/$O/pmc_l*/pmc_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'lexicon' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-2], json.dumps({k: sum(v) / len(v) for k, v in acc.items()}))
PY
