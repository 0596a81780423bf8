#!/usr/bin/env python3
"""Throughput of the headline gate's title scan on the GPU (oi_headline_scan_device) over N synthetic
titles resident in HBM, with the CPU oracle timed on a slice.  Prints one JSON line.
    python tools/headline_bench.py [n_titles] [reps]"""
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
from openintel_amd import dip, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
ctx = oi.HipContext(0)
ctx.use_torch_current_stream()
blob, offs = synth.headlines_torch(n, dev)
mask = torch.zeros(n, dtype=torch.int16, device=dev)
order = torch.zeros(n, dtype=torch.int64, device=dev)
about = torch.zeros(n, dtype=torch.uint8, device=dev)
sc = oi.HeadlineScanner(ctx)
forms = dip.company_name_forms([synth.HEADLINE_COMPANY])
for _ in range(2):
    sc.scan_device(blob, offs, synth.HEADLINE_TICKER, forms, mask, order, about)
torch.cuda.synchronize()
ctx.profile_reset(True)
t0 = time.perf_counter()
for _ in range(reps):
    sc.scan_device(blob, offs, synth.HEADLINE_TICKER, forms, mask, order, about)
torch.cuda.synchronize()
t_call = (time.perf_counter() - t0) / reps
k_ms, k_n = ctx.profile_read("headline")
ctx.profile_reset(False)
text_bytes = blob.numel()
alg_bytes = text_bytes + 8 * (n + 1) + 11 * n  # titles + offsets in, (u16 + u64 + u8) out
from oracle import lib as O
ns = min(n, 400_000)
hb = blob[: int(offs[ns])].cpu().numpy()
ho = offs[: ns + 1].cpu().numpy().astype(np.uint64)
t0 = time.perf_counter()
rm, ro, ra = O.headline_scan(hb, ho, synth.HEADLINE_TICKER, forms)
t_cpu = time.perf_counter() - t0
ok = bool(np.array_equal(mask[:ns].cpu().numpy().view(np.uint16), rm) and
          np.array_equal(order[:ns].cpu().numpy().view(np.uint64), ro) and
          np.array_equal(about[:ns].cpu().numpy(), ra))
sec = k_ms / k_n / 1e3
print(json.dumps({
    "path": "headline gate title scan (catalyst_hits + headline_mentions_company, reference-pinned)",
    "titles": n, "text_bytes": text_bytes, "kernel_ms": k_ms / k_n, "titles_per_s": n / sec,
    "algorithmic_GBs": alg_bytes / sec / 1e9, "frac_of_8TBs": alg_bytes / sec / 8e12, "call_ms": t_call * 1e3,
    "titles_with_hits": int((mask != 0).sum().item()), "titles_about_company": int(about.sum().item()),
    "bit_exact_vs_oracle_on_slice": ok,
    "cpu_oracle": {"titles_per_s": ns / t_cpu, "cores": 1, "sample_titles": ns, "seconds": t_cpu},
}))
