#!/usr/bin/env python3
"""Round 5: configs[4]'s shard -- 12.5M x 1024 bf16 rows, 256 queries -- with the two 128-query
passes of the quad kernel as two launches (OI_BF16_SIB=0, rounds 2-4) or as ONE launch of sibling workgroups that share every
tile through cache (1: neighbouring XCDs / Infinity Cache, 2: same XCD / L2).  Same box, same index, one process:
    OI_LIB=ablation python tools/r05_sib_ab.py [n_docs] [reps]
Per setting: cosine kernel ms per batch (HIP events inside the library), launches per batch, the fraction of the
once-per-batch HBM roof (n x d x 2 B / 8 TB/s), whole search_lists ms, and a checksum of the cosine lists (must not move)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import openintel_amd as oi
import _ablation  # noqa: F401  (OI_LIB=ablation: tools only)
from openintel_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dim, B, depth = 1024, 256, 1000
dev = torch.device("cuda:0")
ctx = oi.HipContext(0)
ctx.use_torch_current_stream()
rows = torch.empty((n, dim), dtype=torch.bfloat16, device=dev)
step = 2_500_000
for r in range(0, n, step):
    e = min(n, r + step)
    rows[r:e] = synth.embeddings_torch(e - r, dim, dev, seed=synth.SEED_EMB + r).to(torch.bfloat16)
terms, offs = synth.forward_index_torch(n, dev, vocab=4096)
idx = oi.HybridIndex(ctx, n, dim, 4096)
idx.set_embeddings_bf16(rows)
idx.set_forward(terms, offs)
idx.set_max_query_terms(4)
idx.finalize()
del terms, offs
batches = [synth.query_batch_torch(B, dim, dev, vocab=4096, seed=synth.SEED_QUERY + 7919 * i) for i in range(2)]


def run(tag):
    for i in range(2):
        L = idx.search_lists(*batches[i], depth=depth)
    torch.cuda.synchronize()
    ctx.profile_reset(2)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        L = idx.search_lists(*batches[i % 2], depth=depth)
    b.record()
    torch.cuda.synchronize()
    ms, launches = ctx.profile_read("cosine")
    ctx.profile_reset(False)
    ctx.synchronize()
    L = idx.search_lists(*batches[0], depth=depth)
    torch.cuda.synchronize()
    print(json.dumps({"tag": tag, "cosine_ms_per_batch": round(ms / reps, 4), "launches_per_batch": launches / reps,
                      "frac_of_once_per_batch_roof": round(2.0 * n * dim / (ms / reps / 1e3) / 8e12, 4),
                      "lists_ms_per_batch": round(a.elapsed_time(b) / reps, 4),
                      "docs_checksum": int(L.cos_docs.to(torch.int64).sum().item()),
                      "score_checksum": float(L.cos_scores.double().sum().item())}), flush=True)


modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ("0", "1", "2")
for rnd in range(2 if len(modes) > 1 else 1):
    for sib in modes:
        os.environ["OI_BF16_SIB"] = sib
        run("OI_BF16_SIB=%s" % sib)
