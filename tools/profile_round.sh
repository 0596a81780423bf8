#!/bin/bash
# Round-end profile collection on the GPU box (one gpurun call):
#   bench.py (default + configs[1] + configs[4] shard + exact scorer), rocprofv3 --kernel-trace --stats over bench.py and the
#   side benches, the shard step, then the PMC passes over bench.py (tools/pmc_profile.sh).
# Usage: tools/profile_round.sh <tag>      (writes under gpurun_out/<tag>/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-prof}
OUT=$R/gpurun_out/$T
mkdir -p $OUT
# SKIP_PMC=1: everything but the counter passes; PMC_ONLY=1: only those (two gpurun calls fit the per-call limit)
if [ -z "$PMC_ONLY" ]; then
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err && echo "bench ok"
OI_BENCH_FORCE_DIST=1 python3 $R/bench.py --gpus 1 --docs 1250000 --steps 200 --warmup 20 --no-cpu-baseline --no-text-paths > $OUT/rccl_world1_shard.json 2> $OUT/rccl_world1.err && echo "rccl world1 ok"
OI_BENCH_BACKEND=gloo OI_BENCH_SINGLE_DEVICE=1 python3 $R/bench.py --gpus 2 --docs 2500000 --steps 100 --warmup 10 --no-cpu-baseline --no-text-paths > $OUT/launcher_gloo2.json 2> $OUT/launcher_gloo2.err && echo "gloo2 ok"
# (four ranks sharing the card: the box admits six GPU processes and counted five ranks + their launcher as seven, round 4)
OI_BENCH_BACKEND=gloo OI_BENCH_SINGLE_DEVICE=1 python3 $R/bench.py --gpus 4 --docs 1600000 --steps 50 --warmup 10 --no-cpu-baseline --no-text-paths > $OUT/launcher_gloo4.json 2> $OUT/launcher_gloo4.err && echo "gloo4 ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-text-paths --no-other-configs --no-pipelined-side --latency-batches 1 --latency-warmup 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats_bench.err && echo "stats bench ok"
python3 $R/bench.py --cosine exact --steps 10 --no-cpu-baseline --no-text-paths > $OUT/bench_exact.json 2> $OUT/bench_exact.err && echo "exact ok"
python3 $R/bench.py --docs 1000000 --batch 1 --depth 100 --steps 300 --no-cpu-baseline --no-text-paths > $OUT/bench_b1.json 2> $OUT/bench_b1.err && echo "b1 ok"
python3 $R/bench.py --corpus bf16 --dim 1024 --batch 256 --docs 12500000 --steps 10 --warmup 2 --no-cpu-baseline --no-text-paths > $OUT/bench_config4_shard.json 2> $OUT/bench_config4_shard.err && echo "config4 shard ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_config4 -- python3 $R/bench.py --corpus bf16 --dim 1024 --batch 256 --docs 12500000 --steps 3 --warmup 1 --no-cpu-baseline --no-text-paths > /dev/null 2> $OUT/stats_config4.err && echo "stats config4 ok"
python3 $R/tools/headline_bench.py 10000000 10 > $OUT/headline_bench.json 2> $OUT/headline_bench.err && echo "headline ok"
python3 $R/tools/lexicon_bench.py 10000000 10 > $OUT/lexicon_bench.json 2> $OUT/lexicon_bench.err && echo "lexicon ok"
python3 $R/tools/bm25_bench.py 10000000 10 > $OUT/bm25_bench.json 2> $OUT/bm25_bench.err && echo "bm25 ok"
python3 $R/tools/bm25_bench.py 10000000 5 256 > $OUT/bm25_bench_b256.json 2> $OUT/bm25_bench_b256.err && echo "bm25 b256 ok"
python3 $R/tools/bm25_bench.py 1250000 20 64 > $OUT/bm25_bench_shard.json 2> $OUT/bm25_bench_shard.err && echo "bm25 shard ok"
bash $R/tools/pmc_bm25_stream.sh $T/pmc_bm25_stream 10000000 > $OUT/pmc_bm25_stream.txt 2>&1 && echo "pmc bm25 stream ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -- python3 $R/tools/headline_bench.py 10000000 5 > /dev/null 2> $OUT/stats_headline.err && echo "stats headline ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bm25 -- python3 $R/tools/bm25_bench.py 10000000 3 > /dev/null 2> $OUT/stats_bm25.err && echo "stats bm25 ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lexicon -- python3 $R/tools/lexicon_bench.py 10000000 5 > /dev/null 2> $OUT/stats_lexicon.err && echo "stats lexicon ok"
python3 $R/tools/shard_step_bench.py 1250000 50 > $OUT/shard_step.json 2> $OUT/shard_step.err && echo "shard ok"
python3 $R/tools/r05_pipeline_probe.py 1250000 100 2> $OUT/pipeline_probe_shard.err | tail -1 > $OUT/pipeline_probe_shard.json && echo "pipeline probe shard ok"
python3 $R/tools/r05_pipeline_probe.py 10000000 40 2> $OUT/pipeline_probe_10M.err | tail -1 > $OUT/pipeline_probe_10M.json && echo "pipeline probe 10M ok"
OI_BENCH_FORCE_DIST=1 python3 $R/bench.py --gpus 1 --docs 1250000 --steps 200 --warmup 20 --no-cpu-baseline --no-text-paths --exchange native > $OUT/native_world1_shard.json 2> $OUT/native_world1.err && echo "native world1 ok"
fi
if [ -z "$SKIP_PMC" ]; then
bash $R/tools/pmc_profile.sh $T/pmc > $OUT/pmc.log 2>&1 && echo "pmc ok"
bash $R/tools/pmc_profile.sh $T/pmc_exact --cosine exact > $OUT/pmc_exact.log 2>&1 && echo "pmc exact ok"
bash $R/tools/pmc_profile.sh $T/pmc_stream --cosine screen-stream > $OUT/pmc_stream.log 2>&1 && echo "pmc stream ok"
fi
