#!/bin/bash
# Chunk schedule of the batch-1 path (BASELINE configs[1]: 1M posts, batch 1, top-100): first-chunk size x growth
# (needs tools/build_ablation.sh; tools/step_ab.py-style A/B through the ablation library).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for M in ${MULTS:-1 2 4 8}; do
for G in ${GROWTHS:-16 32 128}; do
  echo -n "first x$M growth $G: "
  OI_LIB=ablation OI_FIRST_CHUNK_MULT=$M OI_CHUNK_GROWTH=$G python3 - <<PY 2>/dev/null
import os, sys, time
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import torch
import openintel_amd as oi
import _ablation
from openintel_amd import synth
n, B, DIM, DEPTH, K = 1_000_000, 1, 768, 100, 100
dev = torch.device("cuda:0")
ctx = oi.HipContext(0); ctx.use_torch_current_stream()
rows = synth.embeddings_torch(n, DIM, dev)
terms, offs = synth.forward_index_torch(n, dev)
idx = oi.HybridIndex(ctx, n, DIM, synth.VOCAB)
idx.set_embeddings(rows, normalize=False); idx.set_forward(terms, offs); idx.set_max_query_terms(4); idx.finalize()
qs = [synth.query_batch_torch(B, DIM, dev, seed=synth.SEED_QUERY + 7919 * i) for i in range(4)]
for i in range(20): r = idx.search(*qs[i % 4], k=K, depth=DEPTH)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 300
for i in range(N): r = idx.search(*qs[i % 4], k=K, depth=DEPTH)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / N * 1e3
print("%.4f ms/query  %.0f QPS  checksum %d" % (ms, 1e3 / ms, int(r.docs.sum().item())))
PY
done
done
