#!/usr/bin/env python3
"""Host time of one batch's enqueue at a 1.25M-row shard, piece by piece (GPU idle before each piece: pure enqueue cost)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import openintel_amd as oi
import _ablation  # noqa: F401  (OI_LIB=ablation: tools only)
from openintel_amd import retriever, sharded, synth

dev = torch.device("cuda:0")
n, B, DIM, DEPTH, K = 1_250_000, 64, 768, 1000, 100
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
packed = torch.zeros(retriever.packed_words(B, DEPTH), dtype=torch.int32, device=dev)
out = oi.SearchResult(torch.zeros((B, K), dtype=torch.float32, device=dev), torch.zeros((B, K), dtype=torch.int32, device=dev),
                      torch.zeros((B,), dtype=torch.int32, device=dev))
st2 = torch.cuda.Stream(device=dev)
ev = torch.cuda.Event()


def host_us(fn, reps=30):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    ts.sort()
    return ts[len(ts) // 2]


def torch_ops():
    main = torch.cuda.current_stream(dev)
    st2.wait_stream(main)
    with torch.cuda.stream(st2):
        ev.record(st2)
    with torch.cuda.stream(st2):
        st2.wait_event(ev)
        ev.record(st2)


res = {
    "search_lists_packed_call_us": host_us(lambda: idx.search_lists_packed(qv, qt, qo, depth=DEPTH, out=packed)),
    "fuse_packed_call_us": host_us(lambda: retriever.fuse_packed(ctx, packed, 1, B, DEPTH, K, out=out)),
    "torch_stream_event_ops_us": host_us(torch_ops),
    "one_empty_ctypes_call_us": host_us(lambda: ctx.lib.oi_abi_version()),
}
ctx.set_overlap(False)
res["search_lists_packed_call_us_no_overlap"] = host_us(lambda: idx.search_lists_packed(qv, qt, qo, depth=DEPTH, out=packed))
ctx.set_overlap(True)
sr = sharded.make_hip_sharded(ctx, idx, dev)
batches = [synth.query_batch_torch(B, DIM, dev, seed=synth.SEED_QUERY + 7919 * i) for i in range(4)]
for graphs in (False,):
    pipe = sharded.ShardedPipeline(sr, oi.HipContext(0), B, DEPTH, K, graphs=graphs)
    for i in range(8):
        pipe.submit(*batches[i % 4])
    pipe.drain()
    res["submit_call_us_graphs_%s" % graphs] = host_us(lambda: pipe.submit(*batches[0]))
    pipe.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        pipe.submit(*batches[i % 4])
    pipe.drain()
    torch.cuda.synchronize()
    res["period_ms_one_lane_graphs_%s" % graphs] = (time.perf_counter() - t0) / 200 * 1e3
    pipe.close()
print(json.dumps(res))
