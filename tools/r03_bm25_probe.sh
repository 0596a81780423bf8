#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03bm
mkdir -p $OUT
bash $R/tools/build_ablation.sh > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for N in 10000000 1250000; do
  echo "== N=$N ladder" 
  bash $R/tools/bm25_wave_ablate.sh $N
  echo "== N=$N timing"
  OI_LIB=ablation OI_BM25_MODE=wave OI_BM25_WAVE_TIMING=1 python3 $R/tools/bm25_bench.py $N 2 64 wave-only 2>&1 | grep "bm25 wave timing" | tail -2
  for W in 1 2 3; do
    echo -n "wgs/CU=$W "
    OI_LIB=ablation OI_BM25_MODE=wave OI_BM25_WAVE_WGS=$W python3 $R/tools/bm25_bench.py $N 10 64 wave-only 2>/dev/null | tail -1
  done
done
