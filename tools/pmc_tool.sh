#!/bin/bash
# PMC passes over one of the tools/*.py benches (counters without any trace domain besides --kernel-trace).
# Usage (on the GPU box): tools/pmc_tool.sh <outdir-under-gpurun_out> <tool.py> [tool args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
TOOL=$1; shift
mkdir -p $OUT
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/tools/$TOOL "$@" > $OUT/pass$i.log 2>&1
  echo "pass $i ($C): exit $?"
done
python3 $R/tools/pmc_summary.py $OUT
