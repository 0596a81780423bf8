#!/usr/bin/env python3
"""Kernel time of the lexicon scan for one setting of the ablation ladder (OI_LIB=ablation OI_LEX_DBG=k; see
lexicon.hip).  python tools/lexicon_ladder.py [n_posts] [reps]  -> one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
an = oi.HipLexiconAnalyzer(ctx)
for _ in range(2):
    an.analyze_device(blob, offs, pol, spec)
torch.cuda.synchronize()
ctx.profile_reset(True)
for _ in range(reps):
    an.analyze_device(blob, offs, pol, spec)
torch.cuda.synchronize()
k_ms, k_n = ctx.profile_read("lexicon")
print(json.dumps({"dbg": os.environ.get("OI_LEX_DBG", "0"), "variant": os.environ.get("OI_LEX_V", ""), "posts": n,
                  "text_bytes": blob.numel(), "kernel_ms": k_ms / k_n,
                  "text_GBs": blob.numel() / (k_ms / k_n / 1e3) / 1e9}))
