#!/usr/bin/env python3
"""OI_LIB=ablation OI_SELECT_STAMPS=1 python tools/sel_debug.py : the small screened search of tests/test_gpu_prefilter.py through the
ablation build (select_flat's self-check prints a MISMATCH line when the super-bin search and the plain walk disagree)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import openintel_amd as oi
import _ablation  # noqa: F401
from openintel_amd import synth

B, dim, n = 9, 768, 5000
ctx = oi.HipContext(0)
rows = synth.embeddings_np(n, dim, seed=3 + B)
q = synth.embeddings_np(B, dim, seed=55 + B)
rng = np.random.default_rng(B)
lens = rng.integers(1, 8, size=n)
offs = np.zeros(n + 1, np.uint64)
offs[1:] = np.cumsum(lens)
terms = rng.integers(0, 50, size=int(offs[-1])).astype(np.uint32)
idx = oi.HybridIndex(ctx, n, dim, 50, doc_id_base=77)
idx.set_embeddings(rows, normalize=False)
idx.set_forward(terms, offs)
idx.finalize()
qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
for depth in (10, 1000):
    L = idx.search_lists(q, qt, qo, depth=depth)
    print(depth, L.cos_counts)
