#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03p3
mkdir -p $OUT
python3 -m pytest $R/tests/test_gpu_graph.py $R/tests/test_gpu_parity.py::test_index_view_and_pipeline_lanes $R/tests/test_gpu_parity.py::test_sharded_pipeline_overlaps_fusion_and_returns_the_same_results $R/tests/test_gpu_two_ranks.py $R/tests/test_gpu_rccl.py -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -15 $OUT/tests.log
python3 $R/tools/shard_step_bench.py 1250000 50 > $OUT/shard_step.json 2> $OUT/shard_step.err && echo "shard ok"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/shard -- python3 $R/tools/shard_step_bench.py 1250000 12 > $OUT/shard_traced.json 2> $OUT/shard.err && echo "trace ok"
OI_BENCH_FORCE_DIST=1 python3 $R/bench.py --gpus 1 --docs 1250000 --steps 200 --warmup 20 --no-cpu-baseline --no-screen-copy > $OUT/rccl_world1_shard.json 2> $OUT/rccl_world1.err && echo "rccl world1 ok"
OI_BENCH_BACKEND=gloo OI_BENCH_SINGLE_DEVICE=1 python3 $R/bench.py --gpus 2 --docs 2500000 --steps 100 --warmup 10 --no-cpu-baseline --no-screen-copy > $OUT/launcher_gloo2.json 2> $OUT/launcher_gloo2.err && echo "gloo2 ok"
