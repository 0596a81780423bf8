#!/usr/bin/env python3
"""Throughput of the reference-pinned path on the GPU: lexicon scan (oi_lexicon_analyze_device) +
social summary (oi_social_summary) over N synthetic posts resident in HBM, with the CPU oracle timed on
a slice.  Prints one JSON line.   python tools/lexicon_bench.py [n_posts] [reps]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
ctx = oi.HipContext(0)
ctx.use_torch_current_stream()
blob, offs = synth.posts_torch(n, dev)
pol = torch.zeros(n, dtype=torch.float64, device=dev)
spec = torch.zeros(n, dtype=torch.uint8, device=dev)
src = (torch.arange(n, device=dev) % 3 == 0).to(torch.uint8)
an = oi.HipLexiconAnalyzer(ctx)
cfg = oi.EngineConfig()
for _ in range(2):
    an.analyze_device(blob, offs, pol, spec)
torch.cuda.synchronize()
ctx.profile_reset(True)
t0 = time.perf_counter()
for _ in range(reps):
    an.analyze_device(blob, offs, pol, spec)
torch.cuda.synchronize()
t_lex = (time.perf_counter() - t0) / reps
k_ms, k_n = ctx.profile_read("lexicon")
ctx.profile_reset(True)
t0 = time.perf_counter()
for _ in range(reps):
    cnt = oi.SpeculationEngine.social_counters(ctx, src, pol, spec, cfg)
t_sum = (time.perf_counter() - t0) / reps
s_ms, s_n = ctx.profile_read("social_summary")
# the fused form: scan + summary in one pass, nothing per post written (oi_lexicon_summary_device)
for _ in range(2):
    fc = an.summary_device(blob, offs, src, tau=cfg.bull_bear_threshold)
ctx.profile_reset(True)
t0 = time.perf_counter()
for _ in range(reps):
    fc = an.summary_device(blob, offs, src, tau=cfg.bull_bear_threshold)
t_fused = (time.perf_counter() - t0) / reps
f_ms, f_n = ctx.profile_read("lexicon")
ctx.profile_reset(False)
text_bytes = blob.numel()
alg_bytes = text_bytes + 8 * (n + 1) + 9 * n      # text + offsets in, (f64 + u8) out
# CPU oracle on a slice
from oracle import lib as O
ns = min(n, 400_000)
hb = blob[: int(offs[ns])].cpu().numpy()
ho = offs[: ns + 1].cpu().numpy().astype(np.uint64)
t0 = time.perf_counter()
rpol, rspec = O.lexicon_analyze(hb, ho)
t_cpu = time.perf_counter() - t0
ok = bool(np.array_equal(pol[:ns].cpu().numpy().view(np.uint64), rpol.view(np.uint64)) and
          np.array_equal(spec[:ns].cpu().numpy(), rspec))
print(json.dumps({
    "path": "lexicon score + social summary (reference-pinned)", "posts": n, "text_bytes": text_bytes,
    "lexicon_kernel_ms": k_ms / k_n, "lexicon_posts_per_s": n / (k_ms / k_n / 1e3),
    "lexicon_algorithmic_GBs": alg_bytes / (k_ms / k_n / 1e3) / 1e9, "lexicon_frac_of_8TBs": alg_bytes / (k_ms / k_n / 1e3) / 8e12,
    "lexicon_call_ms": t_lex * 1e3,
    "summary_kernel_ms": s_ms / s_n, "summary_algorithmic_GBs": 10 * n / (s_ms / s_n / 1e3) / 1e9, "summary_call_ms": t_sum * 1e3,
    "fused_scan_summary": {"kernel_ms": f_ms / f_n, "call_ms": t_fused * 1e3,
                           "algorithmic_GBs": (text_bytes + 8 * (n + 1) + n) / (f_ms / f_n / 1e3) / 1e9,   # text + offsets + sources in, 64 B/WG out
                           "frac_of_8TBs": (text_bytes + 8 * (n + 1) + n) / (f_ms / f_n / 1e3) / 8e12,
                           "integers_equal_unfused": (fc.total, fc.bullish, fc.bearish, fc.neutral, fc.spec_count, fc.by_source[0], fc.by_source[1]) ==
                                                     (cnt.total, cnt.bullish, cnt.bearish, cnt.neutral, cnt.spec_count, cnt.by_source[0], cnt.by_source[1]),
                           "polarity_sum": fc.polarity_sum},
    "bit_exact_vs_oracle_on_slice": ok,
    "cpu_oracle": {"posts_per_s": ns / t_cpu, "cores": 1, "sample_posts": ns, "seconds": t_cpu},
    "counters": {"total": cnt.total, "bullish": cnt.bullish, "bearish": cnt.bearish, "neutral": cnt.neutral,
                 "spec": cnt.spec_count, "polarity_sum": cnt.polarity_sum},
}))
