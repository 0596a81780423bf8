#!/bin/bash
# PMC passes over bench.py (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not
# fit one pass; counters are collected WITHOUT any trace domain besides --kernel-trace).
# Usage (on the GPU box): tools/pmc_profile.sh <outdir-under-gpurun_out> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-text-paths --no-other-configs --no-pipelined-side --latency-batches 1 --latency-warmup 0 "$@" > $OUT/pass$i.log 2>&1
  echo "pass $i ($C): exit $?"
done
python3 $R/tools/pmc_summary.py $OUT
