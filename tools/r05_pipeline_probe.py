#!/usr/bin/env python3
"""Round 5: the native pipeline (oi_pipeline_*) against the serial oi_search, same index, same box:
    python tools/r05_pipeline_probe.py [n_docs] [steps]
ms per batch of 64 queries for: oi_search one batch at a time; NativePipeline with 1, 2, 3 lanes (device buffers, four
rotating batches, a ring of output slots); the fused doc ids of every mode must equal the serial ones."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import openintel_amd as oi
from openintel_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
flags = sys.argv[3:]
with_comm = "comm" in flags     # an RCCL communicator of one rank: the real ncclAllGather per batch
with_dist = "dist" in flags     # a torch.distributed process group of one rank (nccl) alive in the process, as under bench.py
with_prof = "prof" in flags     # HIP events around the lanes' cosine launches, as in bench.py's timed region
B, DIM, DEPTH, K = 64, 768, 1000, 100
dev = torch.device("cuda:0")
if with_dist:
    import socket
    import torch.distributed as dist
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ctx = oi.HipContext(0)
ctx.use_torch_current_stream()
rows = synth.embeddings_torch(n, DIM, dev)
terms, offs = synth.forward_index_torch(n, dev)
idx = oi.HybridIndex(ctx, n, DIM, synth.VOCAB)
idx.set_embeddings(rows, normalize=False)
idx.set_forward(terms, offs)
idx.set_max_query_terms(4)
comm = None
if with_comm:
    comm = oi.NativeComm(ctx, oi.NativeComm.unique_id(), 0, 1)
    idx.finalize_sharded(comm)
else:
    idx.finalize()
del terms, offs
NB = 4
batches = [synth.query_batch_torch(B, DIM, dev, seed=synth.SEED_QUERY + 7919 * i) for i in range(NB)]
mk = lambda: oi.SearchResult(torch.zeros((B, K), dtype=torch.float32, device=dev), torch.zeros((B, K), dtype=torch.int32, device=dev),
                             torch.zeros((B,), dtype=torch.int32, device=dev))
ref = [mk() for _ in range(NB)]
for i in range(NB):
    idx.search(*batches[i], k=K, depth=DEPTH, out=ref[i])
torch.cuda.synchronize()
res = {"docs": n}
t0 = time.perf_counter()
o = mk()
for i in range(steps):
    idx.search(*batches[i % NB], k=K, depth=DEPTH, out=o)
torch.cuda.synchronize()
res["serial_oi_search_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
for lanes in (1, 2, 3):
    pipe = oi.NativePipeline(idx, lanes=lanes, max_queries=B, max_query_terms=4, depth=DEPTH, k=K, comm=comm)
    outs = [mk() for _ in range(8)]
    for i in range(8):
        pipe.submit(*batches[i % NB], out=outs[i])
    pipe.drain()
    ok = all(torch.equal(outs[i].docs, ref[i % NB].docs) and torch.equal(outs[i].scores, ref[i % NB].scores) for i in range(8))
    best, host = None, None
    if with_prof:
        pipe.profile_reset(2)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(*batches[i % NB], out=outs[i % 8])
        th = (time.perf_counter() - t0) / steps * 1e3
        pipe.drain()
        ms = (time.perf_counter() - t0) / steps * 1e3
        if best is None or ms < best:
            best, host = ms, th
    ok = ok and all(torch.equal(outs[i].docs, ref[((steps - 8 + i) if False else i) % NB].docs) for i in range(0))
    res["pipeline_lanes_%d" % lanes] = {"ms_per_batch": round(best, 4), "host_ms_per_submit": round(host, 4), "bit_identical": bool(ok),
                                        "workspace_GB": round(pipe.workspace_bytes()[0] / 1e9, 3)}
    pipe.close()
print(json.dumps(res))
