#!/bin/bash
# Round 5: kernel trace of the native pipeline probe; prints a steady-state window of the 2-lane (default) run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/r05_pipeline_probe.py ${2:-1250000} ${3:-40} > $OUT/probe.json 2> $OUT/probe.err || exit 1
python3 $R/tools/r05_lane_trace.py $OUT/t ${4:-1}
