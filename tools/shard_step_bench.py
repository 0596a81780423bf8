#!/usr/bin/env python3
"""One rank's share of an 8-GPU step, on one GPU: the shard's lists (oi_search_lists_packed over N/8 rows) and
the merge + fusion of eight shards' packed lists (oi_fuse_packed), timed separately with torch events.  The
all-gather itself (8 x 1 MB over xGMI) cannot be measured on a one-GPU box.
    python tools/shard_step_bench.py [n_docs_per_shard] [reps] [n_shards]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import openintel_amd as oi
from openintel_amd import retriever, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
S = int(sys.argv[3]) if len(sys.argv) > 3 else 8
B, DIM, DEPTH, K = 64, 768, 1000, 100
dev = torch.device("cuda:0")
ctx = oi.HipContext(0)
ctx.use_torch_current_stream()
rows = synth.embeddings_torch(n, DIM, dev)
terms, offs = synth.forward_index_torch(n, dev)
idx = oi.HybridIndex(ctx, n, DIM, synth.VOCAB)
idx.set_embeddings(rows, normalize=False)
idx.set_forward(terms, offs)
idx.set_max_query_terms(4)
idx.finalize()
qv, qt, qo = synth.query_batch_torch(B, DIM, dev)
words = retriever.packed_words(B, DEPTH)
packed = torch.zeros(words, dtype=torch.int32, device=dev)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t_lists = timed(lambda: idx.search_lists_packed(qv, qt, qo, depth=DEPTH, out=packed))
ctx.profile_reset(True)
for _ in range(reps):
    idx.search_lists_packed(qv, qt, qo, depth=DEPTH, out=packed)
torch.cuda.synchronize()
prof = {t: ctx.profile_read(t)[0] / reps for t in ("cosine", "bm25", "select", "rrf")}
ctx.profile_reset(False)
# eight shards with disjoint doc ids: the same lists, doc ids shifted by the shard's base
L = B * DEPTH
allp = packed.repeat(S).view(S, words).clone()
for s_ in range(S):
    allp[s_, 2 * L:4 * L] += s_ * n
flat = allp.reshape(-1).contiguous()
out = oi.SearchResult(torch.zeros((B, K), dtype=torch.float32, device=dev), torch.zeros((B, K), dtype=torch.int32, device=dev),
                      torch.zeros((B,), dtype=torch.int32, device=dev))
t_fuse = timed(lambda: retriever.fuse_packed(ctx, flat, S, B, DEPTH, K, out=out))
ctx.profile_reset(True)
for _ in range(reps):
    retriever.fuse_packed(ctx, flat, S, B, DEPTH, K, out=out)
torch.cuda.synchronize()
prof_f = {t: ctx.profile_read(t)[0] / reps for t in ("select", "rrf", "merge")}
ctx.profile_reset(False)
print(json.dumps({"docs_per_shard": n, "shards": S, "batch": B, "lists_ms": t_lists, "lists_kernels_ms": prof,
                  "fuse_ms": t_fuse, "fuse_kernels_ms": prof_f, "step_without_exchange_ms": t_lists + t_fuse,
                  "qps_8gpu_if_exchange_free": B / ((t_lists + t_fuse) / 1e3)}))
