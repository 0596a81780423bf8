#!/usr/bin/env python3
"""Experiment: two batches in flight (two contexts, two streams, two full indexes over the SAME rows tensor) vs one.
    python tools/dual_stream_probe.py [n_docs] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import openintel_amd as oi
import _ablation  # noqa: F401  (OI_LIB=ablation: tools only)
from openintel_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B, DIM, DEPTH, K = 64, 768, 1000, 100
dev = torch.device("cuda:0")
rows = synth.embeddings_torch(n, DIM, dev)
terms, offs = synth.forward_index_torch(n, dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
ctxs, idxs, outs = [], [], []
for s in streams:
    c = oi.HipContext(0)
    c.set_stream(s)
    ix = oi.HybridIndex(c, n, DIM, synth.VOCAB)
    ix.set_embeddings(rows, normalize=False)
    ix.set_forward(terms, offs)
    ix.set_max_query_terms(4)
    ix.finalize()
    ctxs.append(c); idxs.append(ix)
    outs.append(oi.SearchResult(torch.zeros((B, K), dtype=torch.float32, device=dev), torch.zeros((B, K), dtype=torch.int32, device=dev),
                                torch.zeros((B,), dtype=torch.int32, device=dev)))
batches = [synth.query_batch_torch(B, DIM, dev, seed=synth.SEED_QUERY + 7919 * i) for i in range(4)]
torch.cuda.synchronize()


def run(n_streams):
    for i in range(6):
        j = i % n_streams
        qv, qt, qo = batches[i % 4]
        with torch.cuda.stream(streams[j]):
            idxs[j].search(qv, qt, qo, k=K, depth=DEPTH, out=outs[j])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        j = i % n_streams
        qv, qt, qo = batches[i % 4]
        with torch.cuda.stream(streams[j]):
            idxs[j].search(qv, qt, qo, k=K, depth=DEPTH, out=outs[j])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


r1 = run(1); r2 = run(2); r1b = run(1); r2b = run(2)
for c in ctxs:
    c.synchronize()
print(json.dumps({"docs": n, "one_stream_ms": [r1, r1b], "two_streams_ms": [r2, r2b]}))
