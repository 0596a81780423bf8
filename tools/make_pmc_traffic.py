#!/usr/bin/env python3
"""profiles/pmc_traffic.json (what bench.py's roofline.traffic reads) from three tools/pmc_profile.sh summaries:
    python tools/make_pmc_traffic.py <default-mode pmc_summary.json> <--cosine exact pmc_summary.json> <tag> [<--cosine screen-stream pmc_summary.json>]
The default-mode run supplies the screen kernel over the screening copy (round 5's default) and the BM25 kernel, the exact-mode
run the f32 MFMA kernel (in a default-mode run that kernel also has gated launches per batch that exit at once and would dilute
the averages), the screen-stream run the f32-stream screen (rounds 1-4's default)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1]))
x = json.load(open(sys.argv[2]))
tag = sys.argv[3]
st = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None
ALG = 10_000_000 * 768 * 4
ALG_COPY = 10_000_000 * 768 * 2
sc = d["cosine_copy_screen"]
# launches per step (batch): in the single-mode runs every step ends in ONE rrf_kernel launch; the default run mixes three scorers
# (headline + the f32-stream and exact side blocks), so the copy screen's count is taken from the stream run -- the same chunk
# schedule (round 5: 3 launches at 10M rows with speculative thresholds, 4 without)
lps_exact = round(x["cosine_ksplit"]["launches"] / x["rrf_kernel"]["launches"]) if "rrf_kernel" in x else 4
lps_stream = round(st["cosine_screen"]["launches"] / st["rrf_kernel"]["launches"]) if st and "cosine_screen" in st and "rrf_kernel" in st else 4
lps_copy = lps_stream
assert sc["launches"] % lps_copy == 0, (sc["launches"], lps_copy)
out = {
    "source": "profiles/%s_pmc_summary.json (tools/pmc_profile.sh; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes over "
              "bench.py --steps 3 --warmup 1 --latency-batches 1 --latency-warmup 0, default scorer; collected on the round's code: the tag says which)" % tag,
    "correction": "HBM read bytes = FETCH_SIZE*1024*2 (gfx950 tallies 128-B requests at 64 B on wide coalesced streams, "
                  "MI355X_MICROARCH.md section HBM); write bytes = WRITE_SIZE*1024",
    "kernel": "cosine_copy_screen<768,2,8>",
    "cosine_launches": sc["launches"],
    "cosine_hbm_bytes_per_launch": (sc["hbm_read_bytes"] + sc.get("hbm_write_bytes", 0.0)) / sc["launches"],
    "cosine_launches_per_step": lps_copy,
    "cosine_hbm_read_bytes_per_step": sc["hbm_read_bytes"] / sc["launches"] * lps_copy,
    "cosine_algorithmic_bytes_per_step": ALG_COPY,
    "cosine_eff_clock_GHz": sc.get("eff_clock_GHz"),
    "cosine_lds_bank_conflict_cycles_frac": sc.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, sc.get("SQ_LDS_IDX_ACTIVE", 1.0)),
}
bw = d.get("bm25_wave")
if bw:
    out["bm25_wave_kernel"] = {
        "launches": bw["launches"], "hbm_read_bytes_per_launch": bw["hbm_read_bytes"] / bw["launches"],
        "hbm_write_bytes_per_launch": bw.get("hbm_write_bytes", 0.0) / bw["launches"],
        "lds_bank_conflict_cycles_frac": bw.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, bw.get("SQ_LDS_IDX_ACTIVE", 1.0)),
        "waves_waiting_frac": bw.get("SQ_WAIT_ANY", 0.0) / max(1.0, bw.get("SQ_WAVE_CYCLES", 1.0)),
        "note": "narrow (8-B per lane) loads: the FETCH_SIZE x2 correction is calibrated for 16-B-per-lane streams only; read as an upper bound"}
for name, key in (("bm25_stream_kernel", "bm25_stream"), ("select_flat_kernel", "select_flat"), ("pf_rescore_kernel", "pf_rescore")):
    k = d.get(key)
    if k:
        out[name] = {
            "launches": k["launches"], "hbm_read_bytes_per_launch": k["hbm_read_bytes"] / k["launches"],
            "hbm_write_bytes_per_launch": k.get("hbm_write_bytes", 0.0) / k["launches"],
            "lds_bank_conflict_cycles_frac": k.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, k.get("SQ_LDS_IDX_ACTIVE", 1.0)),
            "waves_waiting_frac": k.get("SQ_WAIT_ANY", 0.0) / max(1.0, k.get("SQ_WAVE_CYCLES", 1.0))}
if "bm25_stream_kernel" in out:
    out["bm25_stream_kernel"]["note"] = ("one launch per batch since the per-term impact floors (two before); the postings are read twice, the second "
                                         "time from L2 (the reads above are HBM: the algorithmic bytes once); the writes are the emitted candidate keys")
ex = x["cosine_ksplit"]
out["exact_kernel"] = {
    "kernel": "cosine_ksplit16_filter<768,2>",
    "cosine_launches": ex["launches"],
    "cosine_hbm_bytes_per_launch": (ex["hbm_read_bytes"] + ex.get("hbm_write_bytes", 0.0)) / ex["launches"],
    "cosine_launches_per_step": lps_exact,
    "cosine_hbm_read_bytes_per_step": ex["hbm_read_bytes"] / ex["launches"] * lps_exact,
    "cosine_algorithmic_bytes_per_step": ALG,
    "cosine_mfma_busy_frac_at_delivered_clock": ex.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, ex.get("GRBM_GUI_ACTIVE", 1.0) / 8 * 1024),
    "cosine_mfma_busy_over_sq_busy": ex.get("mfma_busy_over_sq_busy"),
    "cosine_eff_clock_GHz": ex.get("eff_clock_GHz"),
    "cosine_lds_bank_conflict_cycles_frac": ex.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, ex.get("SQ_LDS_IDX_ACTIVE", 1.0)),
}
if st and "cosine_screen" in st:
    fs = st["cosine_screen"]
    out["f32_stream_kernel"] = {
        "kernel": "cosine_screen_filter<768,2>",
        "cosine_launches": fs["launches"],
        "cosine_hbm_bytes_per_launch": (fs["hbm_read_bytes"] + fs.get("hbm_write_bytes", 0.0)) / fs["launches"],
        "cosine_launches_per_step": lps_stream,
        "cosine_hbm_read_bytes_per_step": fs["hbm_read_bytes"] / fs["launches"] * lps_stream,
        "cosine_algorithmic_bytes_per_step": ALG,
        "cosine_eff_clock_GHz": fs.get("eff_clock_GHz"),
        "cosine_lds_bank_conflict_cycles_frac": fs.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, fs.get("SQ_LDS_IDX_ACTIVE", 1.0)),
    }
    out["f32_stream_kernel_source"] = "profiles/%s_stream_pmc_summary.json (the same passes over bench.py --cosine screen-stream)" % tag
out["exact_kernel_source"] = ("profiles/%s_exact_pmc_summary.json (the same passes over bench.py --cosine exact --steps 3 --warmup 1; "
                              "the kernel with non-temporal loads)" % tag)
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
