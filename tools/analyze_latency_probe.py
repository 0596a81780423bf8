#!/usr/bin/env python3
"""Host-side cost of ONE PostAnalyzer::analyze call (oi_lexicon_analyze, host buffers) at the reference's batch sizes:
a ticker's posts (10 / 100) up to a pooled scan (10 K / 1 M).  Prints one JSON line: median microseconds per call."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import openintel_amd as oi
from openintel_amd import synth
from openintel_amd.analyzer import pack_posts

ctx = oi.HipContext(0)
an = oi.HipLexiconAnalyzer(ctx)
res = {}
for n in (10, 100, 1000, 10_000, 1_000_000):
    texts = synth.posts_np(n, seed=3) if hasattr(synth, "posts_np") else None
    if texts is None:
        raise SystemExit("synth.posts_np missing")
    blob, offs = pack_posts(texts)
    for _ in range(5):
        an.analyze_packed(blob, offs)
    ts = []
    for _ in range(50 if n <= 10_000 else 10):
        t0 = time.perf_counter()
        an.analyze_packed(blob, offs)
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    res["posts_%d" % n] = {"median_us": round(ts[len(ts) // 2], 1), "min_us": round(ts[0], 1), "bytes": int(blob.size)}
print(json.dumps(res))
