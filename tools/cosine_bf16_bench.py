#!/usr/bin/env python3
"""The cosine leg over a bf16 corpus (cosine_bf16_filter): N x dim bf16 rows in HBM, B queries, depth 1000.
Prints one JSON line with the kernel time per batch (HIP events inside the library) and the HBM rate.
    python tools/cosine_bf16_bench.py [n_docs] [dim] [batch] [reps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import openintel_amd as oi
import _ablation  # noqa: F401  (OI_LIB=ablation: tools only)
from openintel_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
ctx = oi.HipContext(0)
ctx.use_torch_current_stream()
rows = torch.empty((n, dim), dtype=torch.bfloat16, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(7)
for s in range(0, n, 1 << 20):
    e = min(n, s + (1 << 20))
    x = torch.randn((e - s, dim), generator=g, dtype=torch.float32, device=dev)
    rows[s:e] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
terms, offs = synth.forward_index_torch(min(n, 100_000), dev)   # a token BM25 side: the index needs one
idx = oi.HybridIndex(ctx, n, dim, synth.VOCAB)
idx.set_embeddings_bf16(rows)
pad_t = torch.zeros(1, dtype=torch.int32, device=dev)
full_offs = torch.cat([offs, offs[-1:].expand(n - (offs.numel() - 1))]) if n > offs.numel() - 1 else offs
idx.set_forward(terms, full_offs)
idx.set_max_query_terms(4)
idx.finalize()
qv, qt, qo = synth.query_batch_torch(B, dim, dev)
for _ in range(2):
    L = idx.search_lists(qv, qt, qo, depth=1000)
torch.cuda.synchronize()
ctx.profile_reset(True)
for _ in range(reps):
    L = idx.search_lists(qv, qt, qo, depth=1000)
torch.cuda.synchronize()
ms, launches = ctx.profile_read("cosine")
ctx.profile_reset(False)
passes = (B + (31 if dim == 1024 else 63)) // (32 if dim == 1024 else 64)
bytes_batch = 2.0 * n * dim * passes
# spot check against torch on a slice
qs = qv[:1].to(torch.bfloat16).float()
ref = (rows[:200_000].float() @ qs.T).squeeze(1)
top = torch.topk(ref, 5)
print(json.dumps({"docs": n, "dim": dim, "batch": B, "cosine_ms_per_batch": ms / reps, "launches_per_batch": launches / reps,
                  "corpus_passes_per_batch": passes, "hbm_GBs": bytes_batch / (ms / reps / 1e3) / 1e9,
                  "frac_of_8TBs": bytes_batch / (ms / reps / 1e3) / 8e12,
                  "tflops": 2.0 * n * dim * B / (ms / reps / 1e3) / 1e12,
                  "top1_score": float(L.cos_scores[0][0]), "slice_top1_ref": float(top.values[0])}))
