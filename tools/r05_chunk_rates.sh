#!/bin/bash
# Round 5: per-chunk stream rate of the copy screen at 10M x 768, 64 queries, under different chunk schedules (ablation build):
# does the weak-threshold chunk stream slower because of its survivors?   bash tools/r05_chunk_rates.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-chunkrates}; mkdir -p $OUT
export OI_LIB=ablation
for combo in "1 8" "4 8" "1 4" "2 16"; do
  set -- $combo
  d=$OUT/m$1_g$2
  OI_FIRST_CHUNK_MULT=$1 OI_CHUNK_GROWTH=$2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/tools/step_ab.py 10000000 12 > $d.json 2> $d.err || exit 1
  echo "== first_mult $1 growth $2: $(tail -n 1 $d.json)" >> $OUT/rates.txt
  python3 $R/tools/trace_one_step.py $d | grep "cosine_copy_screen\|select_flat\|pf_rescore" >> $OUT/rates.txt
done
cat $OUT/rates.txt
