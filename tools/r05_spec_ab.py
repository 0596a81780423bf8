#!/usr/bin/env python3
"""Round 5: speculative thresholds on / off in ONE process (same index, same box, alternating): serial oi_search and the library's
two- and three-lane pipeline at a shard or the full corpus.    python tools/r05_spec_ab.py [n_docs] [reps] [rounds]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import openintel_amd as oi
import _ablation  # noqa: F401
from openintel_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
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
del terms, offs
batches = [synth.query_batch_torch(B, DIM, dev, seed=synth.SEED_QUERY + 7919 * i) for i in range(4)]
out = oi.SearchResult(torch.zeros((B, K), dtype=torch.float32, device=dev), torch.zeros((B, K), dtype=torch.int32, device=dev),
                      torch.zeros((B,), dtype=torch.int32, device=dev))


def serial():
    for i in range(8):
        idx.search(*batches[i % 4], k=K, depth=DEPTH, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        idx.search(*batches[i % 4], k=K, depth=DEPTH, out=out)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def piped(lanes):
    import time
    pipe = oi.NativePipeline(idx, lanes=lanes, max_queries=B, max_query_terms=4, depth=DEPTH, k=K)
    outs = [oi.SearchResult(torch.zeros((B, K), dtype=torch.float32, device=dev), torch.zeros((B, K), dtype=torch.int32, device=dev),
                            torch.zeros((B,), dtype=torch.int32, device=dev)) for _ in range(8)]
    for i in range(16):
        pipe.submit(*batches[i % 4], out=outs[i % 8])
    pipe.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        pipe.submit(*batches[i % 4], out=outs[i % 8])
    pipe.drain()
    ms = (time.perf_counter() - t0) / reps * 1e3
    pipe.close()
    return ms


res = {"docs": n, "serial": {"spec": [], "proven": []}, "lanes2": {"spec": [], "proven": []}, "lanes3": {"spec": [], "proven": []}}
for rnd in range(rounds):
    for mode in ("spec", "proven"):
        ctx.set_screen_speculation(mode == "spec")
        res["serial"][mode].append(round(serial(), 4))
        try:
            res["lanes2"][mode].append(round(piped(2), 4))
            res["lanes3"][mode].append(round(piped(3), 4))
        except TypeError:
            pass
ctx.set_screen_speculation(True)
res["speculation_state"] = ctx.speculation_state()
print(json.dumps(res))
