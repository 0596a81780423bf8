#!/usr/bin/env python3
"""BASELINE configs[0]'s shape on the GPU path: 1000 posts, 384-d, ONE hybrid query (BM25 + cosine + RRF, top-10) through
the OI_HOST entry point -- host microseconds per query and the kernels' share.  Prints one JSON line."""
import sys, time, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openintel_amd as oi
from openintel_amd import synth
n, dim, K, DEPTH = 1000, 384, 10, 100
ctx = oi.HipContext(0)
rows = synth.embeddings_np(n, dim)
terms, offs = synth.forward_index_np(n)
idx = oi.HybridIndex(ctx, n, dim, synth.VOCAB)
idx.set_embeddings(rows, normalize=False); idx.set_forward(terms, offs); idx.set_max_query_terms(4); idx.finalize()
rng = np.random.default_rng(1)
qv = rows[123:124].copy()
qt = terms[int(offs[123]):int(offs[123]) + 3].astype(np.uint32)
qo = np.array([0, qt.size], dtype=np.uint32)
for _ in range(20): r = idx.search(qv, qt, qo, k=K, depth=DEPTH)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); r = idx.search(qv, qt, qo, k=K, depth=DEPTH); ts.append((time.perf_counter() - t0) * 1e6)
ts.sort()
ctx.profile_reset(True)
for _ in range(20): r = idx.search(qv, qt, qo, k=K, depth=DEPTH)
prof = {t: ctx.profile_read(t) for t in ("cosine", "bm25", "select", "rrf")}
print(json.dumps({"config0_us_per_query_p50": ts[100], "p95": ts[190], "first_doc": int(r.docs[0][0]), "kernels_ms_over_20": prof}))
