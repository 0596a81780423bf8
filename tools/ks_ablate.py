#!/usr/bin/env python3
"""Diagnostics: run bench.py with the K-split kernel's ablation switches (OI_KS_DEBUG bits:
1 = no DMA, 2 = no MFMA, 4 = no epilogue) and print the cosine kernel time of each variant.
Results of ablated variants are WRONG by construction; only the timings are meaningful.
Needs an ablation build: OI_EXTRA_HIPCC_FLAGS=-DOI_ABLATION python -m openintel_amd.build --force
(the product build compiles these variants out and ignores OI_KS_DEBUG)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
modes = [int(m) for m in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 4, 1, 5, 2, 6]
extra = sys.argv[2:]
for m in modes:
    env = dict(os.environ, OI_KS_DEBUG=str(m))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", *extra], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("dbg=%d FAILED\n%s" % (m, r.stderr[-2000:]))
        continue
    d = json.loads(line[-1])
    rf = d["roofline"]
    print("dbg=%d  cosine %.3f ms/step  %.1f TF  %.0f GB/s  | step %.2f ms  other %s" % (
        m, rf["kernel_ms_per_step"], rf["achieved"], rf["hbm_GBs_algorithmic"], d["ms_per_step"],
        {k: round(v, 3) for k, v in d["other_kernels_ms_per_step"].items() if k != "note"}), flush=True)
