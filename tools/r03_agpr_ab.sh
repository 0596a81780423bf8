#!/bin/bash
# one-session A/B: MFMA operands read from AGPRs directly (variant agprs) vs the builtin (variant base), screen kernel (10M x 768 f32
# hybrid step) and the solo bf16 kernel (10M x 768 bf16 cosine leg); libraries from tools/build_variant.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 -m pytest $R/tests/test_gpu_prefilter.py $R/tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for V in base agprs; do
  echo -n "$V step:  "; OI_LIB=ablation_$V python3 $R/tools/step_ab.py 10000000 40 2>/dev/null | tail -1
done
done
for V in base agprs base agprs; do
  echo -n "$V bf16 768 B=64:  "; OI_LIB=ablation_$V python3 $R/tools/cosine_bf16_bench.py 10000000 768 64 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['cosine_ms_per_batch'],4), 'ms', round(d['hbm_GBs']), 'GB/s')"
done
