#!/usr/bin/env python3
"""Host-side cost of ONE PostAnalyzer::analyze call (oi_lexicon_analyze, host buffers) at the reference's batch sizes:
a ticker's posts (10 / 100) up to a pooled scan (10 K / 1 M); and of one headline-gate scan (oi_headline_scan) of a
ticker's titles.  Prints one JSON line: median microseconds per call."""
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
rng = np.random.default_rng(1)
for n, t in ((1000, 10), (10_000, 100)):  # a scan_watchlist call's reduction: t tickers' signals pooled
    pol = np.round(rng.uniform(-1, 1, n), 2)
    spec = (rng.random(n) < 0.3).astype(np.uint8)
    src = (rng.random(n) < 0.5).astype(np.uint8)
    seg = np.linspace(0, n, t + 1).astype(np.uint64)
    for _ in range(5):
        an.summary_segments(src, pol, spec, seg)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        an.summary_segments(src, pol, spec, seg)
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    res["segments_%d_signals_%d" % (t, n)] = {"median_us": round(ts[len(ts) // 2], 1), "min_us": round(ts[0], 1)}
from openintel_amd import dip
sc = oi.HeadlineScanner(ctx)
forms = dip.company_name_forms([synth.HEADLINE_COMPANY])
for n in (10, 100, 1000):  # the dip gate's call: a ticker's headlines (dip.rs:617-626)
    blob, offs = pack_posts(synth.headlines_np(n, seed=4))
    for _ in range(5):
        sc.scan_packed(blob, offs, synth.HEADLINE_TICKER, forms)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        sc.scan_packed(blob, offs, synth.HEADLINE_TICKER, forms)
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    res["titles_%d" % n] = {"median_us": round(ts[len(ts) // 2], 1), "min_us": round(ts[0], 1), "bytes": int(blob.size)}
rows = [(synth.headlines_np(8, seed=40 + r), synth.HEADLINE_TICKER, forms) for r in range(25)]  # a dip scan: 25 rows x 8 titles
for _ in range(5):
    sc.scan_rows(rows)
ts, ts1 = [], []
for _ in range(30):
    t0 = time.perf_counter()
    sc.scan_rows(rows)
    ts.append((time.perf_counter() - t0) * 1e6)
    t0 = time.perf_counter()
    for titles, tk, fm in rows:
        sc.scan(titles, tk, fm)
    ts1.append((time.perf_counter() - t0) * 1e6)
ts.sort(); ts1.sort()
res["dip_rows_25x8_one_call"] = {"median_us": round(ts[len(ts) // 2], 1), "min_us": round(ts[0], 1)}
res["dip_rows_25x8_row_by_row"] = {"median_us": round(ts1[len(ts1) // 2], 1), "min_us": round(ts1[0], 1)}
print(json.dumps(res))
