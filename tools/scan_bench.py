#!/usr/bin/env python3
"""The batch callers' device path (run_scan / run_compare, mcp/tools.rs:193-352): N synthetic posts pooled from T tickers,
resident in HBM -> one lexicon scan + one per-ticker social_summary reduction (oi_lexicon_scan_segments_device).  Prints
one JSON line: the scan's and the reduction's kernel time, tickers/s, the reduction's algorithmic GB/s, and the CPU
oracle's per-ticker loop timed on a slice.      python tools/scan_bench.py [n_posts] [n_tickers] [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import openintel_amd as oi
import _ablation  # noqa: F401  (OI_LIB=ablation: tools only)
from openintel_amd import synth
from openintel_amd.analyzer import COUNTERS_DTYPE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_tick = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
ctx = oi.HipContext(0)
ctx.use_torch_current_stream()
blob, offs = synth.posts_torch(n, dev)
rng = np.random.default_rng(17)
cuts = np.sort(rng.integers(0, n + 1, n_tick - 1))  # ragged tickers: 0 .. a few hundred posts, mean n / n_tick
seg = np.concatenate([[0], cuts, [n]]).astype(np.int64)
d_seg = torch.from_numpy(seg).to(dev)
src = (torch.arange(n, device=dev) % 3 == 0).to(torch.uint8)
d_out = torch.zeros(n_tick * 8, dtype=torch.int64, device=dev)
d_pol = torch.zeros(n, dtype=torch.float64, device=dev)
d_spec = torch.zeros(n, dtype=torch.uint8, device=dev)
an = oi.HipLexiconAnalyzer(ctx)
for _ in range(2):
    an.scan_segments_device(blob, offs, src, d_seg, d_out, 0.2, d_pol, d_spec)
torch.cuda.synchronize()
ctx.profile_reset(True)
t0 = time.perf_counter()
for _ in range(reps):
    an.scan_segments_device(blob, offs, src, d_seg, d_out, 0.2, d_pol, d_spec)
torch.cuda.synchronize()
t_call = (time.perf_counter() - t0) / reps
scan_ms, scan_n = ctx.profile_read("lexicon")
seg_ms, seg_n = ctx.profile_read("social_summary_segmented")
ctx.profile_reset(False)
got = d_out.cpu().numpy().view(COUNTERS_DTYPE)

from oracle import lib as O
ts = min(n_tick, 4000)  # the oracle's per-ticker loop on the first tickers
ns = int(seg[ts])
hb = blob[: int(offs[ns])].cpu().numpy()
ho = offs[: ns + 1].cpu().numpy().astype(np.uint64)
hs = src[:ns].cpu().numpy()
t0 = time.perf_counter()
pol, spec = O.lexicon_analyze(hb, ho)
ref = O.social_summary_segmented(hs, pol, spec, seg[: ts + 1].astype(np.uint64))
t_cpu = time.perf_counter() - t0
ok = all(int(got["total"][k]) == r.total_mentions and int(got["bullish"][k]) == r.bullish and
         int(got["bearish"][k]) == r.bearish and int(got["spec_count"][k]) == r.spec_count and
         np.float64(got["polarity_sum"][k]).tobytes() == np.float64(r.polarity_sum).tobytes() for k, r in enumerate(ref))
seg_s = seg_ms / seg_n / 1e3
seg_bytes = 10 * n + 8 * (n_tick + 1) + 64 * n_tick  # f64 + u8 + u8 per post in, offsets in, one record per ticker out
print(json.dumps({
    "path": "batch callers: pooled lexicon scan + per-ticker social_summary (run_scan / run_compare, reference-pinned)",
    "posts": n, "tickers": n_tick, "posts_per_ticker_max": int(np.diff(seg).max()),
    "scan_kernel_ms": scan_ms / scan_n, "segmented_summary_kernel_ms": seg_ms / seg_n,
    "call_ms": t_call * 1e3, "tickers_per_s": n_tick / t_call, "posts_per_s": n / t_call,
    "segmented_summary_algorithmic_GBs": seg_bytes / seg_s / 1e9, "segmented_summary_frac_of_8TBs": seg_bytes / seg_s / 8e12,
    "bit_exact_vs_oracle_on_slice": bool(ok),
    "cpu_oracle": {"tickers_per_s": ts / t_cpu, "posts_per_s": ns / t_cpu, "cores": 1, "sample_tickers": ts,
                   "sample_posts": ns, "seconds": t_cpu},
    "library": os.path.basename(oi._lib.LIB_PATH) if hasattr(oi._lib, "LIB_PATH") else None,
}))
