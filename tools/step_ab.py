#!/usr/bin/env python3
"""ms per hybrid step (oi_search, device buffers, 4 rotating batches) at the bench shape, for A/B runs of ablation switches:
    OI_LIB=ablation [OI_NO_LATE_FORK=1 ...] python tools/step_ab.py [n_docs] [steps] [batch] [depth]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import openintel_amd as oi
import _ablation  # noqa: F401  (OI_LIB=ablation: tools only)
from openintel_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
DEPTH = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
DIM, K = 768, min(100, DEPTH)
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
del terms, offs
batches = [synth.query_batch_torch(B, DIM, dev, seed=synth.SEED_QUERY + 7919 * i) for i in range(4)]
out = oi.SearchResult(torch.zeros((B, K), dtype=torch.float32, device=dev), torch.zeros((B, K), dtype=torch.int32, device=dev),
                      torch.zeros((B,), dtype=torch.int32, device=dev))
for i in range(4):
    idx.search(*batches[i], k=K, depth=DEPTH, out=out)
torch.cuda.synchronize()
res = []
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(steps):
        idx.search(*batches[i % 4], k=K, depth=DEPTH, out=out)
    b.record()
    torch.cuda.synchronize()
    res.append(a.elapsed_time(b) / steps)
ctx.synchronize()
print(json.dumps({"docs": n, "batch": B, "depth": DEPTH, "ms_per_step": [round(x, 4) for x in res], "docs_checksum": int(out.docs.sum().item())}))
