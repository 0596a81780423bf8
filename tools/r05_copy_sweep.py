#!/usr/bin/env python3
"""Round 5: the screening-copy kernel's ring depth and grid, same box, same index, one process (ablation build):
    OI_LIB=ablation python tools/r05_copy_sweep.py [n_docs] [steps]
For every (OI_COPY_NBUF, OI_SCREEN_CUS) setting: ms per hybrid step, the screen launches' summed ms per step, the fraction of
the 8 TB/s HBM spec on the copy's bytes, and a checksum of the fused doc ids (must not move)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import openintel_amd as oi
import _ablation  # noqa: F401  (OI_LIB=ablation: tools only)
from openintel_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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
assert idx.index_bytes()[1] > 0
batches = [synth.query_batch_torch(B, DIM, dev, seed=synth.SEED_QUERY + 7919 * i) for i in range(4)]
out = oi.SearchResult(torch.zeros((B, K), dtype=torch.float32, device=dev), torch.zeros((B, K), dtype=torch.int32, device=dev),
                      torch.zeros((B,), dtype=torch.int32, device=dev))


def run(tag, mode=_lib.OI_COSINE_SCREEN, bytes_per=2):
    ctx.set_cosine_mode(mode)
    for i in range(4):
        idx.search(*batches[i], k=K, depth=DEPTH, out=out)
    torch.cuda.synchronize()
    best = None
    for rep in range(2):
        ctx.profile_reset(2)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(steps):
            idx.search(*batches[i % 4], k=K, depth=DEPTH, out=out)
        b.record()
        torch.cuda.synchronize()
        ms, launches = ctx.profile_read("cosine")
        ctx.profile_reset(False)
        r = {"tag": tag, "ms_per_step": round(a.elapsed_time(b) / steps, 4), "screen_ms_per_step": round(ms / steps, 4),
             "launches_per_step": launches / steps, "hbm_frac": round(bytes_per * n * DIM * steps / (ms / 1e3) / 8e12, 4),
             "checksum": int(out.docs.sum().item())}
        if best is None or r["ms_per_step"] < best["ms_per_step"]:
            best = r
    ctx.synchronize()
    print(json.dumps(best), flush=True)


for rnd in range(2):
    for nbuf in ("9", "8", "7", "6"):
        for cus in ("224", "240", "256"):
            os.environ["OI_COPY_NBUF"], os.environ["OI_SCREEN_CUS"] = nbuf, cus
            run("copy nbuf=%s cus=%s" % (nbuf, cus))
    os.environ.pop("OI_COPY_NBUF"); os.environ.pop("OI_SCREEN_CUS")
    run("f32 stream (default grid)", _lib.OI_COSINE_SCREEN_STREAM, 4)
