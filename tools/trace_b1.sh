#!/bin/bash
# Kernel trace of BASELINE configs[1] (1M rows, batch 1): the per-launch timeline of one query.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-trace_b1}
mkdir -p $OUT
timeout -k 10 300 python3 $R/bench.py --docs 1000000 --batch 1 --depth 100 --steps 200 --no-cpu-baseline > $OUT/b1_plain.json 2> $OUT/b1_plain.err && \
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/b1 -- python3 $R/bench.py --docs 1000000 --batch 1 --depth 100 --steps 20 --warmup 2 --no-cpu-baseline > $OUT/b1.json 2> $OUT/b1.err
