#!/usr/bin/env python3
"""Summarise tools/pmc_profile.sh output: per kernel (bench kernels only) sum of each counter over the
run, launches, total time; writes <out>/pmc_summary.json and prints a table.
HBM bytes = FETCH_SIZE * 1024 * 2 (gfx950 counts 64 B per 128-B request on wide coalesced streams,
MI355X_MICROARCH.md section HBM) + WRITE_SIZE * 1024."""
import csv
import glob
import json
import sys
from collections import defaultdict

out = sys.argv[1]
KEEP = ("cosine_copy_screen", "cosine_screen", "cosine_ksplit", "cosine_bf16_quad", "cosine", "bm25_stream", "bm25_plan", "bm25_wave", "bm25_block", "bm25_scan", "select_flat", "select_topk", "pf_rescore", "rrf_kernel",
        "lexicon_kernel", "lists_to_pool", "headline_scan", "social_summary")   # first match wins
agg = defaultdict(lambda: defaultdict(float))
for f in sorted(glob.glob(out + "/pass*/*/*counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = next((k for k in KEEP if k in name), None)
        if key is None:
            continue
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        did = (f, r["Dispatch_Id"])
        if did not in seen:
            seen.add(did)
            agg[key]["_launches:" + f] += 1
            agg[key]["_ns:" + f] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
res = {}
for k, c in agg.items():
    launches = max(v for n, v in c.items() if n.startswith("_launches:"))
    ns = max(v for n, v in c.items() if n.startswith("_ns:"))
    d = {n: v for n, v in c.items() if not n.startswith("_")}
    d["launches"] = launches
    d["total_ms"] = ns / 1e6
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes"] = d["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    if "GRBM_GUI_ACTIVE" in d:
        d["eff_clock_GHz"] = d["GRBM_GUI_ACTIVE"] / 8 / ns
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
        d["mfma_busy_over_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]
    res[k] = d
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
for k, d in res.items():
    print(k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in d.items()}))
