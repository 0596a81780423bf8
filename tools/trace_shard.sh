#!/bin/bash
# Kernel trace of one rank's share of an 8-GPU step (1.25M rows) and of the full 10M step: per-launch timeline.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-trace_shard}
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/shard -- python3 $R/tools/shard_step_bench.py 1250000 10 > $OUT/shard.json 2> $OUT/shard.err && \
python3 $R/tools/trace_summary.py $OUT/shard 60 > $OUT/shard_summary.txt && \
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/full -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/full.json 2> $OUT/full.err && \
python3 $R/tools/trace_summary.py $OUT/full 40 > $OUT/full_summary.txt
