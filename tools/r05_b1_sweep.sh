#!/bin/bash
# Round 5: configs[1] (1M x 768, ONE query) through the copy screen -- chunk schedule A/B (ablation build: OI_FIRST_CHUNK_MULT x OI_CHUNK_GROWTH).
# bash tools/r05_b1_sweep.sh TAG  ->  gpurun_out/TAG/b1_sweep.txt
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$R/gpurun_out/${1:-b1sweep}; mkdir -p $OUT
export OI_LIB=ablation
: > $OUT/b1_sweep.txt
for depth in 100 1000; do
  for combo in "1 16" "1 128" "2 64" "4 64" "8 64" "16 64" "4 8" "1 8" "gemv"; do
    set -- $combo
    if [ "$1" = "gemv" ]; then line=$(OI_SMALL_BATCH_GEMV=1 python3 $R/tools/step_ab.py 1000000 300 1 $depth 2>/dev/null | tail -n 1)
    else line=$(OI_FIRST_CHUNK_MULT=$1 OI_CHUNK_GROWTH=$2 python3 $R/tools/step_ab.py 1000000 300 1 $depth 2>/dev/null | tail -n 1); fi
    echo "depth $depth first_mult/growth $combo : $line" | tee -a $OUT/b1_sweep.txt
  done
done
