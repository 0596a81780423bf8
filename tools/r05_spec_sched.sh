#!/bin/bash
# Round 5: chunk schedules of the copy screen WITH speculative thresholds (ablation build: OI_FIRST_CHUNK_MULT x OI_CHUNK_GROWTH; OI_NO_SPEC=1 = without),
# 10M x 768, 64 queries and the 1.25M-row shard.   bash tools/r05_spec_sched.sh TAG
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$R/gpurun_out/${1:-specsched}; mkdir -p $OUT; : > $OUT/sched.txt
export OI_LIB=ablation
for docs in 10000000 1250000; do
  for combo in "1 8" "1 8 nospec" "1 16" "1 32" "1 128" "2 16" "2 64" "4 16" "4 64"; do
    set -- $combo
    if [ "$3" = "nospec" ]; then export OI_NO_SPEC=1; else unset OI_NO_SPEC; fi
    line=$(OI_FIRST_CHUNK_MULT=$1 OI_CHUNK_GROWTH=$2 python3 $R/tools/step_ab.py $docs 40 64 1000 2>/dev/null | tail -n 1)
    echo "docs $docs first_mult/growth $combo : $line" | tee -a $OUT/sched.txt
  done
done
