#!/bin/bash
# round 3, first probe: timeline of the two-lane shard pipeline, the launcher rehearsals at shard size, baselines of the kernels to be worked on
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03p1
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/shard -- python3 $R/tools/shard_step_bench.py 1250000 12 > $OUT/shard_traced.json 2> $OUT/shard.err && \
python3 $R/tools/trace_timeline.py $OUT/shard 4 260 > $OUT/shard_timeline.txt && echo "trace ok"
python3 $R/tools/shard_step_bench.py 1250000 50 > $OUT/shard_step.json 2> $OUT/shard_step.err && echo "shard ok"
OI_BENCH_FORCE_DIST=1 python3 $R/bench.py --gpus 1 --docs 1250000 --steps 200 --warmup 20 --no-cpu-baseline --no-screen-copy > $OUT/rccl_world1_shard.json 2> $OUT/rccl_world1.err && echo "rccl world1 ok"
OI_BENCH_BACKEND=gloo OI_BENCH_SINGLE_DEVICE=1 python3 $R/bench.py --gpus 2 --docs 2500000 --steps 100 --warmup 10 --no-cpu-baseline --no-screen-copy > $OUT/launcher_gloo2.json 2> $OUT/launcher_gloo2.err && echo "gloo2 ok"
python3 $R/bench.py --corpus bf16 --dim 1024 --batch 256 --docs 12500000 --steps 10 --warmup 2 --no-cpu-baseline --latency-batches 20 --latency-warmup 3 > $OUT/bench_config4_shard.json 2> $OUT/bench_config4_shard.err && echo "config4 shard ok"
python3 $R/tools/headline_bench.py 10000000 10 > $OUT/headline_bench.json 2> $OUT/headline_bench.err && echo "headline ok"
python3 $R/tools/bm25_bench.py 10000000 10 > $OUT/bm25_bench.json 2> $OUT/bm25_bench.err && echo "bm25 ok"
python3 $R/tools/bm25_bench.py 10000000 5 256 > $OUT/bm25_bench_b256.json 2> $OUT/bm25_bench_b256.err && echo "bm25 b256 ok"
