#!/bin/bash
# Round 5: kernel trace of the hybrid step at the bench shape (10M x 768, 64 queries; default scorer = screening copy):
# (bash tools/r05_trace_step.sh TAG [docs] [batch] [depth])
# the per-launch timeline of one step (tools/trace_one_step.py) and the per-kernel totals.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r05_trace}
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/step -- python3 $R/tools/step_ab.py ${2:-10000000} 12 ${3:-64} ${4:-1000} > $OUT/step.json 2> $OUT/step.err || exit 1
python3 $R/tools/trace_one_step.py $OUT/step > $OUT/timeline.txt
python3 $R/tools/trace_summary.py $OUT/step > $OUT/summary.txt
cat $OUT/timeline.txt
