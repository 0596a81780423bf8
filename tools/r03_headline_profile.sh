#!/bin/bash
# Text-scan artefacts of a round (one gpurun call): headline gate (bench line, rocprofv3 kernel stats, ablation ladder +
# stamps, tile sweep) and the batch callers' pooled scan + per-ticker reduction (bench line, kernel stats, ablation probe).
# Usage: tools/r03_headline_profile.sh <tag>   (writes under gpurun_out/<tag>/; the ladder needs tools/build_ablation.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-hl}
OUT=$R/gpurun_out/$T
mkdir -p $OUT
python3 $R/tools/headline_bench.py 10000000 10 > $OUT/headline_bench.json 2> $OUT/headline_bench.err && echo "headline ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -- python3 $R/tools/headline_bench.py 10000000 5 > /dev/null 2> $OUT/stats_headline.err && echo "stats headline ok"
cp $(find $OUT/stats_headline -name "*kernel_stats.csv" | head -1) $OUT/headline_rocprofv3_kernel_stats.csv && rm -rf $OUT/stats_headline
LEVELS="1 2 3 4 5 0" bash $R/tools/headline_ladder.sh > $OUT/headline_ladder.txt 2>&1 && echo "ladder ok"
TILES="160 192 224 256" bash $R/tools/r03_headline_tile.sh > $OUT/headline_tile_sweep.txt 2>&1 && echo "tile sweep ok"
# the batch callers' device path: bench line + rocprofv3 kernel stats (lexicon_scan_kernel, social_summary_segmented_kernel)
python3 $R/tools/scan_bench.py 10000000 100000 10 > $OUT/scan_bench.json 2> $OUT/scan_bench.err && echo "scan ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_scan -- python3 $R/tools/scan_bench.py 10000000 100000 5 > /dev/null 2> $OUT/stats_scan.err && echo "stats scan ok"
cp $(find $OUT/stats_scan -name "*kernel_stats.csv" | head -1) $OUT/scan_rocprofv3_kernel_stats.csv && rm -rf $OUT/stats_scan
bash $R/tools/r03_seg_probe.sh > $OUT/seg_probe.txt 2>&1 && echo "seg probe ok"
