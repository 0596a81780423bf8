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
import _ablation  # noqa: F401  (OI_LIB=ablation: tools only)
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

# ---- the same work as bench.py runs it for N > 1: batches in flight (sharded.ShardedPipeline) -- lists of batch i+1 beside
# the merge + fusion of batch i (S shards' lists, as above), and with two lanes the lists of two batches at once through a
# view of the shard.  Period per batch over `reps` batches; the all-gather is absent (one GPU).
import time

from openintel_amd import sharded


class _OneRankOfS:
    """A ShardedRetriever stand-in whose 'exchange' is the S-shard buffer built above (world 1 has nothing to gather)."""

    def __init__(self):
        self.local, self.device, self.world, self.group, self.dist, self.exchange = idx, dev, 1, None, None, False

    def check(self):
        ctx.synchronize()


class _Pipe(sharded.ShardedPipeline):
    """ShardedPipeline whose exchange returns the S-shard buffer built above (the timing of fuse_packed over 8 shards' lists)."""

    def _exchange(self, slot):
        return flat


batches = [synth.query_batch_torch(B, DIM, dev, seed=synth.SEED_QUERY + 7919 * i) for i in range(4)]
fctx = oi.HipContext(0)
pipe = _Pipe(_OneRankOfS(), fctx, B, DEPTH, K)
pipe.n_shards = S
cal = pipe.calibrate(batches, lambda: oi.HipContext(0), reps=reps, placements=4)   # one lane, then four placements of a second
print(json.dumps({"docs_per_shard": n, "shards": S, "batch": B, "lists_ms": t_lists, "lists_kernels_ms": prof,
                  "fuse_ms": t_fuse, "fuse_kernels_ms": prof_f, "step_without_exchange_ms": t_lists + t_fuse,
                  "qps_8gpu_if_exchange_free": B / ((t_lists + t_fuse) / 1e3),
                  "pipelined": dict(cal, note="batches in flight as bench.py --gpus N runs them (ShardedPipeline: fusion of batch i "
                                              "beside the lists of batch i+1; a second lane = two batches' lists at once through a view "
                                              "of the shard, kept if its stream placement pays -- ShardedPipeline.calibrate); four "
                                              "rotating query batches; no all-gather on one GPU"),
                  "qps_8gpu_pipelined_if_exchange_hidden": B / (cal["period_ms"] / 1e3)}))
