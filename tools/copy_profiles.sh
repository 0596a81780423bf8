#!/bin/bash
# gpurun_out/<tag>/ (tools/profile_round.sh) -> profiles/<name>_*: the summaries that are committed.
#   bash tools/copy_profiles.sh r04z r04z
T=$1; N=${2:-$1}
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/$T; D=$R/profiles
cp $S/bench.json $D/${N}_bench.json
cp $S/bench_under_rocprof.json $D/${N}_bench_under_rocprof.json
cp $S/bench_exact.json $D/${N}_bench_exact.json
cp $S/bench_b1.json $D/${N}_bench_b1_1M.json
cp $S/bench_config4_shard.json $D/${N}_bench_config4_shard_12_5M_1024_bf16_b256.json
cp $S/rccl_world1_shard.json $D/${N}_rccl_world1_shard_1_25M.json
cp $S/launcher_gloo2.json $D/${N}_launcher_gloo2_ranks_2_5M.json
cp $S/launcher_gloo4.json $D/${N}_launcher_gloo4_ranks_1_6M.json
cp $S/shard_step.json $D/${N}_shard_step.json
cp $S/bm25_bench.json $D/${N}_bm25_bench.json
cp $S/bm25_bench_b256.json $D/${N}_bm25_bench_b256.json
cp $S/bm25_bench_shard.json $D/${N}_bm25_bench_shard_1_25M.json
cp $S/headline_bench.json $D/${N}_headline_bench.json
cp $S/lexicon_bench.json $D/${N}_lexicon_bench.json
cp $S/pmc_bm25_stream.txt $D/${N}_bm25_stream_pmc.txt
for k in bench config4 bm25 headline lexicon; do
  f=$(ls $S/stats_$k/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $D/${N}_${k}_rocprofv3_kernel_stats.csv
done
cp $S/pmc/pmc_summary.json $D/${N}_pmc_summary.json
cp $S/pmc_exact/pmc_summary.json $D/${N}_exact_pmc_summary.json
cp $S/pmc_stream/pmc_summary.json $D/${N}_stream_pmc_summary.json
cp $S/pipeline_probe_shard.json $D/${N}_pipeline_probe_shard_1_25M.json
cp $S/pipeline_probe_10M.json $D/${N}_pipeline_probe_10M.json
cp $S/native_world1_shard.json $D/${N}_native_world1_shard_1_25M.json
python3 $R/tools/make_pmc_traffic.py $D/${N}_pmc_summary.json $D/${N}_exact_pmc_summary.json $N $D/${N}_stream_pmc_summary.json > /dev/null
ls $D | grep "^${N}_" | wc -l
