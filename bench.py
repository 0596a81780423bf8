#!/usr/bin/env python3
"""bench.py -- hybrid BM25 + cosine + RRF top-100 queries/sec on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2]): 10M posts x 768-d f32 synthetic corpus,
batches of 64 queries (embedding + 4 BM25 terms), per-list depth k'=1000, RRF top-100.
A step = one batch through the whole hot path with corpus, index and queries resident in HBM.
With N > 1 a 10M-row corpus of the same distribution is row-sharded over the N ranks (strong scaling: the TOTAL
row count is fixed; every rank generates its own rows from seed + rank, so the rows differ from the N=1 corpus but
their number, shape and statistics do not) and the per-shard lists are exchanged with one RCCL all-gather per batch.

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N ... bench.py ...`) BEFORE anything touches the GPU, relays
rank 0's JSON line and exits with the child's code; under torch.distributed.run it is a rank.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share one run one after the
# other.  Two batches in flight (N > 1, and the pipelined side measurement at N = 1) have seven streams in play: 8 queues keep
# them apart (csrc/api.hip has the measurement).  Read by the runtime when it initialises: set before torch touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak


def cpu_baseline(n_total, dim, vocab, depth, k, sample_docs, sample_queries):
    """The CPU oracle (kind "port": the build's own scalar C restatement -- the reference has no retrieval code and
    its Rust cannot be built here, so this is NOT "the reference Rust CPU path") timed on a bounded slice of the
    same workload, single-thread AND on all host cores (OpenMP over the queries of the batch; every query is the
    same scalar pipeline, tests/test_oracle_retrieval.py).  Brute force is linear in the corpus size (dot products
    N*d, the BM25 scan of the forward index, the O(N) selection), so the full-corpus rate is the slice rate scaled
    by sample_docs / n_total; the all-cores leg is also run at half the slice to show that linearity in the line."""
    import numpy as np
    from openintel_amd import synth
    from oracle import lib as O
    rows = synth.embeddings_np(sample_docs, dim)
    nproc = os.cpu_count() or 1
    try:
        nproc = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # the all-cores leg is parallel over queries: give it as many queries as there are threads (up to 4 batches)
    n_qn = max(sample_queries, min(4 * sample_queries, nproc))
    threads = max(1, min(nproc, O.max_threads(), n_qn))
    q = synth.embeddings_np(n_qn, dim, seed=synth.SEED_QUERY)
    terms, offs = synth.forward_index_np(sample_docs, vocab)
    qt, qo = synth.query_terms_np(n_qn, vocab)
    df, _ = O.bm25_df(terms, offs, vocab)

    def run(n_docs, n_q, n_threads, blocked=False):
        o = offs[:n_docs + 1]
        t0 = time.perf_counter()
        _, _, _, used = O.hybrid_search_batch(rows[:n_docs], terms[:int(o[-1])], o, vocab, q[:n_q], qt[:int(qo[n_q])],
                                              qo[:n_q + 1], k, depth, n_threads=n_threads, df=df, blocked=blocked)
        return time.perf_counter() - t0, used

    n_q1 = max(1, min(sample_queries, 16))            # the single-thread leg: a quarter of the batch is ~2.5 s
    dt1, _ = run(sample_docs, n_q1, 1)
    dtn, used = run(sample_docs, n_qn, threads)        # round-2 driver: parallel over the queries, each streams the corpus
    # round 3: the same scalar arithmetic with the loops blocked for a CPU -- rows / docs outermost and
    # split over ALL host threads, every row block scored against all the batch's queries while it is cached.  Same results
    # (tests/test_oracle_retrieval.py).  This is the stated baseline (`value`); the other two legs stay in the line.
    all_threads = max(1, min(nproc, O.max_threads()))
    dtb, used_b = run(sample_docs, sample_queries, all_threads, blocked=True)
    dth, _ = run(sample_docs // 2, sample_queries, all_threads, blocked=True)
    scale = sample_docs / n_total
    single = n_q1 / dt1 * scale
    multi = n_qn / dtn * scale
    blocked = sample_queries / dtb * scale
    return {
        "value": blocked, "unit": "queries/s", "cores": used_b, "kind": "port",
        "sample": "one %d-query batch against a %d-doc slice on %d threads, loops blocked for the CPU: rows / docs outermost, "
                  "every 64-row block scored against all queries while cached (%.1f s); rate scaled by %d/%d (brute force is "
                  "linear in corpus size: the same batch on half the slice took %.2fx the time)" % (
                      sample_queries, sample_docs, used_b, dtb, sample_docs, n_total, dth / dtb),
        "per_query_parallel": {"value": multi, "unit": "queries/s", "cores": used,
                               "sample": "%d queries against the same slice, one query per thread, each streaming the corpus "
                                         "(round 2's all-cores leg; %.1f s)" % (n_qn, dtn)},
        "single_thread": {"value": single, "unit": "queries/s", "cores": 1,
                          "sample": "%d queries against the same %d-doc slice, one thread (%.1f s)" % (n_q1, sample_docs, dt1)},
        "nproc": nproc, "parallel_speedup": blocked / single,
        "compiler": O.CFLAGS,
        "label": "build's CPU restatement (scalar C oracle), not the reference's Rust; EXTRAPOLATED x%g: measured on a %d-doc slice, "
                 "scaled to %d docs (linear in corpus size)" % (n_total / sample_docs, sample_docs, n_total),
    }


def text_paths(oi, ctx, dev, n_items, reps, cpu_sample):
    """The reference-pinned paths in the driver-run line: the lexicon scan + social summary over
    `n_items` synthetic posts and the headline gate's title scan over as many synthetic titles, resident in HBM (SURVEY 8d
    seeds); kernel time from HIP events inside the library, algorithmic bytes per SURVEY 8d, the host-inclusive latency of ONE
    call at the reference's real size (100 posts: src/adapters/sources/reddit/mod.rs:93; a ticker's headlines), and the CPU
    oracle -- for THESE paths a line-by-line restatement of the reference (lexicon.rs:53-87, polarity.rs:8-14,
    speculation_engine.rs:70-125, dip.rs:247-272), pinned by the reference's own vectors -- single-thread (how the reference
    runs it) and on all host cores (the same scalar function per post, OpenMP static chunks).  Outside the timed region."""
    import numpy as np
    import torch
    from openintel_amd import dip, synth
    from oracle import lib as O
    nproc = os.cpu_count() or 1
    try:
        nproc = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = max(1, min(nproc, O.max_threads()))
    out = {}

    def pct(v, q):
        v = sorted(v)
        return v[min(len(v) - 1, max(0, int(round(q * (len(v) - 1)))))]

    # ------------------------------------------------------------ lexicon scan + social summary (A1-A4)
    an = oi.HipLexiconAnalyzer(ctx)
    cfg = oi.EngineConfig()
    blob, offs = synth.posts_torch(n_items, dev)
    n = n_items
    pol = torch.zeros(n, dtype=torch.float64, device=dev)
    spec = torch.zeros(n, dtype=torch.uint8, device=dev)
    src = (torch.arange(n, device=dev) % 3 == 0).to(torch.uint8)
    text_bytes = int(blob.numel())
    for _ in range(2):
        an.analyze_device(blob, offs, pol, spec)
    torch.cuda.synchronize()
    ctx.profile_reset(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        an.analyze_device(blob, offs, pol, spec)
    torch.cuda.synchronize()
    call_ms = (time.perf_counter() - t0) / reps * 1e3
    k_ms, k_n = ctx.profile_read("lexicon")
    ctx.profile_reset(False)
    for _ in range(2):
        fc = an.summary_device(blob, offs, src, tau=cfg.bull_bear_threshold)
    ctx.profile_reset(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fc = an.summary_device(blob, offs, src, tau=cfg.bull_bear_threshold)
    f_call_ms = (time.perf_counter() - t0) / reps * 1e3
    f_ms, f_n = ctx.profile_read("lexicon")
    ctx.profile_reset(False)
    alg = text_bytes + 8 * (n + 1) + 9 * n          # text + offsets in, (f64 + u8) per post out
    alg_f = text_bytes + 8 * (n + 1) + n            # fused with the A4 reduction: + sources in, 64 B of counters out
    # one call at the reference's size, host buffers in and out (OI_HOST: H2D, scan, D2H, synchronous)
    texts100 = synth.posts_np(100)
    b100, o100 = oi.pack_posts(texts100)
    for _ in range(20):
        an.analyze_packed(b100, o100)
    lat = []
    for _ in range(200):
        t0 = time.perf_counter()
        an.analyze_packed(b100, o100)
        lat.append((time.perf_counter() - t0) * 1e6)
    # CPU: the reference-faithful oracle on a slice of the same posts
    ns = min(n, cpu_sample)
    hb = blob[: int(offs[ns])].cpu().numpy()
    ho = offs[: ns + 1].cpu().numpy().astype(np.uint64)
    hs = src[:ns].cpu().numpy()
    t0 = time.perf_counter()
    rpol, rspec = O.lexicon_analyze(hb, ho)
    t_cpu1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.social_summary(hs, rpol, rspec)
    t_sum = time.perf_counter() - t0
    O.lexicon_analyze(hb[: int(ho[min(ns, 50_000)])], ho[: min(ns, 50_000) + 1], n_threads=threads)   # (spin the thread pool up)
    t0 = time.perf_counter()
    mpol, mspec = O.lexicon_analyze(hb, ho, n_threads=threads)
    t_cpun = time.perf_counter() - t0
    ok = bool(np.array_equal(pol[:ns].cpu().numpy().view(np.uint64), rpol.view(np.uint64)) and np.array_equal(spec[:ns].cpu().numpy(), rspec)
              and np.array_equal(mpol.view(np.uint64), rpol.view(np.uint64)) and np.array_equal(mspec, rspec))
    ksec = k_ms / max(1, k_n) / 1e3
    fsec = f_ms / max(1, f_n) / 1e3
    out["lexicon_path"] = {
        "rows": "SURVEY 8 A1-A4: LexiconAnalyzer::score / analyze (lexicon.rs:53-87), Polarity::new, social_summary (speculation_engine.rs:70-125)",
        "posts": n, "text_bytes": text_bytes,
        "scan": {"kernel_ms": k_ms / max(1, k_n), "call_ms": call_ms, "posts_per_s": n / ksec, "algorithmic_bytes": alg,
                 "algorithmic_GBs": alg / ksec / 1e9, "frac_of_hbm_peak": alg / ksec / 1e9 / PEAK_HBM_GBS,
                 "note": "oi_lexicon_analyze_device: per-post (f64 polarity, u8 speculative) written; bytes = text + 8(n+1) offsets in + 9n out"},
        "fused_scan_and_summary": {"kernel_ms": f_ms / max(1, f_n), "call_ms": f_call_ms, "algorithmic_bytes": alg_f,
                                   "algorithmic_GBs": alg_f / fsec / 1e9, "frac_of_hbm_peak": alg_f / fsec / 1e9 / PEAK_HBM_GBS,
                                   "note": "oi_lexicon_summary_device: the scan with the A4 reduction fused in, nothing written per post",
                                   "counters": {"total": int(fc.total), "bullish": int(fc.bullish), "bearish": int(fc.bearish),
                                                "neutral": int(fc.neutral), "speculative": int(fc.spec_count)}},
        "one_call_of_100_posts_host_inclusive_us": {"p50": pct(lat, 0.5), "p95": pct(lat, 0.95), "calls": len(lat),
                                                    "note": "oi_lexicon_analyze (OI_HOST): pack-free host arrays in, H2D, scan, D2H, synchronous; "
                                                            "the reference fetches at most 100 posts per source and call (reddit/mod.rs:93)"},
        "bit_exact_vs_oracle_on_slice": ok,
        "cpu_baseline": {"kind": "port", "reference_faithful": True, "sample": "%d of the same posts" % ns,
                         "single_thread": {"posts_per_s": ns / t_cpu1, "cores": 1, "seconds": t_cpu1,
                                           "social_summary_seconds": t_sum,
                                           "note": "how the reference runs it: one thread maps `score` over the posts (lexicon.rs:82-87)"},
                         "all_cores": {"posts_per_s": ns / t_cpun, "cores": threads, "seconds": t_cpun,
                                       "note": "the same scalar function per post, OpenMP static chunks (oio_lexicon_analyze_mt)"},
                         "nproc": nproc, "compiler": O.CFLAGS,
                         "label": "build's CPU restatement of the reference's Rust, pinned by the reference's own vectors (tests/test_oracle_golden.py); "
                                  "the reference itself cannot be built here (no cargo/rustc)"},
    }
    del blob, offs, pol, spec, src
    # ------------------------------------------------------------ headline gate's title scan (f-3)
    sc = oi.HeadlineScanner(ctx)
    forms = dip.company_name_forms([synth.HEADLINE_COMPANY])
    blob, offs = synth.headlines_torch(n_items, dev)
    mask = torch.zeros(n, dtype=torch.int16, device=dev)
    order = torch.zeros(n, dtype=torch.int64, device=dev)
    about = torch.zeros(n, dtype=torch.uint8, device=dev)
    text_bytes = int(blob.numel())
    for _ in range(2):
        sc.scan_device(blob, offs, synth.HEADLINE_TICKER, forms, mask, order, about)
    torch.cuda.synchronize()
    ctx.profile_reset(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        sc.scan_device(blob, offs, synth.HEADLINE_TICKER, forms, mask, order, about)
    torch.cuda.synchronize()
    call_ms = (time.perf_counter() - t0) / reps * 1e3
    k_ms, k_n = ctx.profile_read("headline")
    ctx.profile_reset(False)
    alg = text_bytes + 8 * (n + 1) + 11 * n         # titles + offsets in, (u16 mask + u64 order + u8 about) per title out
    t100 = synth.headlines_np(100)
    tb, to = oi.pack_posts(t100)
    for _ in range(20):
        sc.scan_packed(tb, to, synth.HEADLINE_TICKER, forms)
    lat = []
    for _ in range(200):
        t0 = time.perf_counter()
        sc.scan_packed(tb, to, synth.HEADLINE_TICKER, forms)
        lat.append((time.perf_counter() - t0) * 1e6)
    ns = min(n, cpu_sample)
    hb = blob[: int(offs[ns])].cpu().numpy()
    ho = offs[: ns + 1].cpu().numpy().astype(np.uint64)
    t0 = time.perf_counter()
    rm, ro, ra = O.headline_scan(hb, ho, synth.HEADLINE_TICKER, forms)
    t_cpu1 = time.perf_counter() - t0
    O.headline_scan(hb[: int(ho[min(ns, 50_000)])], ho[: min(ns, 50_000) + 1], synth.HEADLINE_TICKER, forms, n_threads=threads)
    t0 = time.perf_counter()
    mm, mo, ma = O.headline_scan(hb, ho, synth.HEADLINE_TICKER, forms, n_threads=threads)
    t_cpun = time.perf_counter() - t0
    ok = bool(np.array_equal(mask[:ns].cpu().numpy().view(np.uint16), rm) and np.array_equal(order[:ns].cpu().numpy().view(np.uint64), ro)
              and np.array_equal(about[:ns].cpu().numpy(), ra) and np.array_equal(mm, rm) and np.array_equal(mo, ro) and np.array_equal(ma, ra))
    ksec = k_ms / max(1, k_n) / 1e3
    out["headline_path"] = {
        "rows": "SURVEY 8 f-3: catalyst_hits / headline_mentions_company (dip.rs:247-272)",
        "titles": n, "text_bytes": text_bytes,
        "scan": {"kernel_ms": k_ms / max(1, k_n), "call_ms": call_ms, "titles_per_s": n / ksec, "algorithmic_bytes": alg,
                 "algorithmic_GBs": alg / ksec / 1e9, "frac_of_hbm_peak": alg / ksec / 1e9 / PEAK_HBM_GBS,
                 "note": "oi_headline_scan_device; bytes = titles + 8(n+1) offsets in + 11n out (u16 mask, u64 first-hit order, u8 about)"},
        "one_call_of_100_titles_host_inclusive_us": {"p50": pct(lat, 0.5), "p95": pct(lat, 0.95), "calls": len(lat),
                                                     "note": "oi_headline_scan (OI_HOST): H2D, scan, D2H, synchronous"},
        "titles_with_hits": int((mask != 0).sum().item()), "titles_about_company": int(about.sum().item()),
        "bit_exact_vs_oracle_on_slice": ok,
        "cpu_baseline": {"kind": "port", "reference_faithful": True, "sample": "%d of the same titles" % ns,
                         "single_thread": {"titles_per_s": ns / t_cpu1, "cores": 1, "seconds": t_cpu1},
                         "all_cores": {"titles_per_s": ns / t_cpun, "cores": threads, "seconds": t_cpun,
                                       "note": "the same scalar functions per title, OpenMP static chunks (oio_headline_scan_mt)"},
                         "nproc": nproc, "compiler": O.CFLAGS,
                         "label": "build's CPU restatement of the reference's Rust, pinned by the reference's own vectors (tests/test_dip.py)"},
    }
    return out


def launcher_command(argv, n_gpus, port, python=None):
    """argv/env of the child that runs the N ranks (tests/test_bench_launcher.py).  `argv` = this script's own
    arguments, passed through unchanged so that every rank parses what the user typed."""
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
           "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.join(ROOT, "bench.py")] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL / device-tensor sharing across processes
    env.setdefault("OMP_NUM_THREADS", "1")                # torch.distributed.run would set (and warn about) it otherwise
    env["MASTER_ADDR"] = "127.0.0.1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)                                  # the child launcher sets its ranks' own
    return cmd, env


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(argv, n_gpus):
    """Parent of an N-rank run: never imports torch, never touches the GPU (a process that has initialised the GPU must
    not exec or be replaced); starts the ranks as a CHILD, relays their output line by line and returns their exit code."""
    import subprocess
    cmd, env = launcher_command(argv, n_gpus, free_port())
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in p.stdout:               # rank 0 prints the ONE JSON line; anything else the ranks print is passed on too
        sys.stdout.write(line)
        sys.stdout.flush()
    return p.wait()


def other_configs():
    """BASELINE.json's other single-GPU configurations in the driver-run line (N = 1, default workload only): each is THIS script run
    as a child process on its own workload after the parent's timed region (the same code path as a stand-alone run: own index, own
    warm-up, own timed region with HIP events); the parent keeps the child's headline fields.  Never fatal: a child that fails or
    times out leaves {"error": ...}."""
    import subprocess
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH")) or \
            any(k.startswith("ROCPROF") for k in os.environ):
        return {"skipped": "running under rocprofv3: the children would be profiled into this run's statistics"}
    here = os.path.abspath(__file__)
    common = ["--no-cpu-baseline", "--no-text-paths", "--no-other-configs", "--no-pipelined-side"]
    runs = {
        "config1_single_query": (["--docs", "1000000", "--batch", "1", "--depth", "100", "--steps", "300", "--warmup", "20"],
                                 "BASELINE configs[1]: 1M posts x 768-d f32, ONE query, hybrid top-100 (per-list depth 100)"),
        "config4_one_shard": (["--corpus", "bf16", "--dim", "1024", "--batch", "256", "--docs", "12500000", "--steps", "10", "--warmup", "2"],
                              "one rank's shard of BASELINE configs[4] (100M x 1024-d bf16 over 8 GPUs): 12.5M rows, 256 queries, on one GPU"),
    }
    out = {}
    for name, (argv, what) in runs.items():
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, here] + argv + common, capture_output=True, text=True, timeout=400)
            last = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not last:
                out[name] = {"workload": what, "error": "rc %d: %s" % (r.returncode, r.stderr.strip()[-300:])}
                continue
            d = json.loads(last[-1])
            roof = d.get("roofline", {})
            blk = {"workload": what, "argv": " ".join(argv), "queries_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                   "p50_ms": d.get("p50_ms"), "p95_ms": d.get("p95_ms"), "steps": d["steps"], "dtype": d.get("dtype"),
                   "roofline": {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "corpus_passes_per_batch")},
                   "wall_s": round(time.perf_counter() - t0, 1)}
            if "f32_stream_scorer" in d:   # configs[1]: rounds 1-4's path beside it (the f32 GEMV: no copy to stream)
                blk["without_screening_copy"] = {k: d["f32_stream_scorer"].get(k) for k in ("ms_per_step", "queries_per_s", "hbm_frac_on_its_bytes")}
            if "exact_scorer" in d:
                blk["exact_scorer_ms_per_step"] = d["exact_scorer"].get("ms_per_step")
            out[name] = blk
        except Exception as e:   # (timeout, a JSON line that does not parse)
            out[name] = {"workload": what, "error": repr(e)[:300]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--depth", type=int, default=1000)
    ap.add_argument("--vocab", type=int, default=131072)
    ap.add_argument("--bm25", choices=["stream", "wave", "taat", "scan"], default="stream",
                    help="BM25 kernel (default stream: term-at-a-time through a per-wave LDS ring; wave: one wave per task; "
                         "taat: the first-generation workgroup kernel)")
    ap.add_argument("--corpus", choices=["f32", "bf16"], default="f32",
                    help="embedding storage (default f32 = BASELINE configs[1]/[2]; bf16 = the configs[4] regime, HBM-bound)")
    ap.add_argument("--cosine", choices=["screen", "exact", "split", "screen-copy", "screen-stream"], default="screen",
                    help="f32 corpus scorer: a bf16 screen with a proven error bound + exact f32 rescoring of the survivors "
                         "(default: the exact scorer's lists, HBM-bound; the screen streams the index's bf16 screening copy, "
                         "made at finalize), screen-stream (the same screen converting the f32 rows on the fly: rounds 1-4's "
                         "default), exact f32 MFMA for every row, or split-precision products (six bf16 MFMAs)")
    ap.add_argument("--no-screen-copy-index", action="store_true",
                    help="oi_index_set_screen_copy(NEVER): the index holds no bf16 screening copy (the default scorer then streams the f32 rows)")
    ap.add_argument("--query-batches", type=int, default=4, help="distinct query batches rotated through the steps")
    ap.add_argument("--no-pipeline", action="store_true", help="N > 1: do not overlap exchange + fusion with the next batch's lists")
    ap.add_argument("--lanes", type=int, default=None,
                    help="N > 1, pipelined: batches whose lists are scored at once, each through its own view of the shard "
                         "(own stream and workspaces); 1 = one at a time.  Default: 2 for the torch exchange (calibrated: the second "
                         "lane is kept only where it pays), 3 for the native one (measured over three fresh processes each at a "
                         "1.25M-row shard: 0.470-0.490 ms per step against 0.499-0.513 with 2; tools/r05_lanes_ab.sh)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, default workload: skip the side blocks that run BASELINE.json's other single-GPU configurations "
                         "(configs[1]: 1M x 768, one query; one 12.5M-row shard of configs[4]: 1024-d bf16, 256 queries) as child "
                         "processes of this script after the timed region")
    ap.add_argument("--exchange", choices=["torch", "native"], default="torch",
                    help="N > 1: who runs the pipeline and the all-gather -- torch (sharded.ShardedPipeline over torch.distributed, "
                         "the default) or native (oi_pipeline_* + oi_comm_*: lanes, streams and RCCL inside the library, what a host on "
                         "the C ABI gets; torch.distributed then only ships the communicator's 128-byte id)")
    ap.add_argument("--no-pipelined-side", action="store_true",
                    help="N = 1: skip the side measurement of the native two-lane pipeline (pipelined_native)")
    ap.add_argument("--lane-bm25", choices=["side", "inline"], default="inline",
                    help="N > 1, lanes > 1 (torch exchange): a lane's BM25 leg on a side stream of its own (as in oi_search) or inside the "
                         "lane's stream (fewer streams contending for the hardware queues; the other lane's corpus stream is what it overlaps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-text-paths", action="store_true",
                    help="N = 1: skip the reference-pinned paths' blocks (lexicon_path, headline_path: SURVEY 8 rows A1-A4, f-3)")
    ap.add_argument("--text-items", type=int, default=10_000_000, help="synthetic posts / titles of the text-path blocks (SURVEY 8d: 10M)")
    ap.add_argument("--text-cpu-sample", type=int, default=2_000_000, help="posts / titles the CPU oracle is timed on")
    ap.add_argument("--settle-steps", type=int, default=0,
                    help="untimed steps (x the number of ranks) before the warm-up (an experiment: no effect measured, round 4); 0 = none")
    ap.add_argument("--placements", type=int, default=4,
                    help="N > 1, torch exchange: fresh streams tried for each further lane (HIP maps streams to hardware queues "
                         "round-robin; 8 were measured no better than 4; every rank tries the same number)")
    ap.add_argument("--no-speculation", action="store_true",
                    help="screen with proven thresholds only (oi_set_screen_speculation(ctx, 0): rounds 2-4's behaviour; A/B runs)")
    ap.add_argument("--no-stream-side", action="store_true", help="skip the f32-stream screen's side measurement (f32_stream_scorer)")
    ap.add_argument("--cpu-sample-docs", type=int, default=400_000)
    ap.add_argument("--cpu-sample-queries", type=int, default=64)
    ap.add_argument("--latency-batches", type=int, default=200, help="timed batches of the latency loops (SURVEY 8d: >= 200)")
    ap.add_argument("--latency-warmup", type=int, default=20, help="untimed batches before each latency loop (SURVEY 8d: 20)")
    args = ap.parse_args()
    if args.lanes is None:
        args.lanes = 3 if args.exchange == "native" else 2

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    # stdout carries ONE JSON line and nothing else: RCCL ("RCCL version : ...") and gloo ("[Gloo] Rank 0 is connected ...")
    # print banners to fd 1 from native code, so fd 1 is pointed at stderr for the life of the rank and the line is
    # written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # RCCL / device-tensor sharing across processes needs dmabuf IPC on this driver (the image exports it; keep it if a
    # launcher dropped the environment)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world   # under a launcher the launcher's world size is the truth
    # Rehearsal switches for a 1-GPU box (not for reported numbers): OI_BENCH_BACKEND=gloo lets two
    # ranks share cuda:0 (RCCL refuses duplicate devices) so the sharded path runs end to end.
    backend = os.environ.get("OI_BENCH_BACKEND", "nccl")
    if os.environ.get("OI_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # OI_BENCH_FORCE_DIST=1 (rehearsal on a one-GPU box): a process group of ONE rank, so that the real RCCL calls -- init
    # with device_id, the df all-reduce, all_gather_into_tensor on the side stream, calibrate, drain -- run at --gpus 1
    force_dist = world == 1 and bool(os.environ.get("OI_BENCH_FORCE_DIST"))
    if force_dist:
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import openintel_amd as oi
    from openintel_amd import sharded, synth

    ctx = oi.HipContext(local_rank)
    ctx.use_torch_current_stream()
    from openintel_amd import _lib as _oil
    MODES = {"screen": _oil.OI_COSINE_SCREEN, "exact": _oil.OI_COSINE_EXACT, "split": _oil.OI_COSINE_SPLIT,
             "screen-copy": _oil.OI_COSINE_SCREEN_COPY, "screen-stream": _oil.OI_COSINE_SCREEN_STREAM}
    ctx.set_cosine_mode(MODES[args.cosine])
    if args.no_speculation:
        ctx.set_screen_speculation(False)   # (lane contexts are made with oi_create_like: they inherit it)

    # ---------------------------------------------------------------- corpus shard in HBM
    lo, hi = sharded.shard_bounds(args.docs, world, rank)
    n_local = hi - lo
    t_build = time.perf_counter()
    idx = oi.HybridIndex(ctx, n_local, args.dim, args.vocab, doc_id_base=lo)
    sr = sharded.make_hip_sharded(ctx, idx, dev)
    if force_dist:
        sr.exchange = True
    native_comm = None
    if args.exchange == "native" and (world > 1 or force_dist):
        # RCCL inside the library (oi_comm_*): the 128-byte id goes from rank 0 to the others over the process group that
        # launched the ranks -- control plane only; statistics, all-gathers and lanes are the library's from here on
        if world > 1 and os.environ.get("OI_BENCH_SINGLE_DEVICE"):
            raise SystemExit("bench.py: --exchange native needs one GPU per rank (RCCL refuses two ranks on one device: "
                             "ncclInvalidUsage); the one-GPU rehearsals are OI_BENCH_FORCE_DIST=1 --gpus 1 --exchange native, "
                             "or the torch exchange over gloo")
        ids = [oi.NativeComm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        native_comm = oi.NativeComm(ctx, ids[0], rank, world)

    def set_text():
        terms, offs = synth.forward_index_torch(n_local, dev, vocab=args.vocab, seed=synth.SEED_TEXT + rank)
        idx.set_forward(terms, offs)
        idx.set_max_query_terms(4)                      # the synthetic queries have exactly 4 terms
        idx.set_bm25_mode({"scan": idx.BM25_SCAN, "taat": idx.BM25_TAAT, "wave": idx.BM25_WAVE, "stream": idx.BM25_STREAM}[args.bm25])
        nt = int(offs[-1].item())
        del terms, offs
        torch.cuda.empty_cache()
        return nt

    def finalize_index():
        if native_comm is not None:
            idx.finalize_sharded(native_comm)          # ncclAllReduce of df / N / tokens inside the library
        else:
            sr.finalize()                              # all-reduce of df / N / tokens when world > 1

    if args.corpus == "bf16":
        # configs[4]'s regime, sized for one GPU's 288 GB (100M x 1024 bf16 = 204.8 GB of rows): the text index is
        # built and finalized FIRST -- its build scratch (sort keys twice, unique keys, tf: ~36 B per token) is gone before the
        # rows exist -- and the rows are generated in slices straight into their bf16 matrix (a whole f32 copy would not fit)
        n_tokens_local = set_text()
        finalize_index()
        torch.cuda.empty_cache()
        rows = torch.empty((n_local, args.dim), dtype=torch.bfloat16, device=dev)
        slice_rows = 2_500_000
        for r in range(0, n_local, slice_rows):
            e = min(n_local, r + slice_rows)           # (unit norm up to the rounding; every slice its own seed)
            rows[r:e] = synth.embeddings_torch(e - r, args.dim, dev, seed=synth.SEED_EMB + rank + 1000003 * (r // slice_rows)).to(torch.bfloat16)
        torch.cuda.empty_cache()
        idx.set_embeddings_bf16(rows)
    else:
        rows = synth.embeddings_torch(n_local, args.dim, dev, seed=synth.SEED_EMB + rank)
        idx.set_embeddings(rows, normalize=False)      # rows are generated unit-norm
        if args.no_screen_copy_index:
            idx.set_screen_copy(idx.SCREEN_COPY_NEVER)
        n_tokens_local = set_text()
        finalize_index()                               # (and the bf16 screening copy, budget permitting)
    _, df_local = idx.local_stats()                    # for the BM25 leg's algorithmic bytes (rank 0 reports)
    rows_owned_b, screen_copy_b, bm25_index_b = idx.index_bytes()
    hbm_free_b, hbm_total_b = torch.cuda.mem_get_info(dev)     # after the index is resident, before the search workspaces exist
    # what the default scorer's screen streams on this rank: the index's bf16 screening copy if finalize made it
    # (round 5: batches of <= 8 queries too -- configs[1] -- half the bytes of the f32 GEMV they used to run)
    copy_streamed = (args.corpus == "f32" and args.cosine in ("screen", "screen-copy") and args.dim in (384, 768)
                     and (screen_copy_b > 0 or args.cosine == "screen-copy"))
    # Distinct query batches rotated through the steps (step i uses batch i mod NB): no step can profit from the
    # previous step's thresholds, pools or cache contents being those of the same queries.
    NB = max(1, args.query_batches)
    batches = [synth.query_batch_torch(args.batch, args.dim, dev, vocab=args.vocab, seed=synth.SEED_QUERY + 7919 * i)
               for i in range(NB)]
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    bm25_bytes = None
    if args.bm25 != "scan":   # mean over the rotated batches of 8 B x (postings of the batch's terms) + bounds x blocks x terms
        n_blocks = (n_local + 32767) // 32768   # (bounds: 8 B per (block, term); the stream kernel reads three cell words, 12 B)
        per = [8.0 * float(df_local[b[1].cpu().numpy()].astype("int64").sum()) +
               (12.0 if args.bm25 == "stream" else 8.0) * n_blocks * int(b[1].numel()) for b in batches]
        bm25_bytes = sum(per) / len(per)
    else:
        bm25_bytes = 4.0 * n_tokens_local + 8.0 * (n_local + 1)

    out = oi.SearchResult(torch.zeros((args.batch, args.k), dtype=torch.float32, device=dev),
                          torch.zeros((args.batch, args.k), dtype=torch.int32, device=dev),
                          torch.zeros((args.batch,), dtype=torch.int32, device=dev))

    step_no = [0]
    # N > 1: throughput mode -- the exchange + fusion of batch i overlap the lists of batch i + 1 (sharded.ShardedPipeline;
    # independent batches, one all-gather each, same results); the latency loop below runs the batches one at a time.
    pipe = None
    lane_ctxs = []
    fuse_ctx = None

    class NativePipe:
        """oi_pipeline_* behind the handful of names the loops below use of sharded.ShardedPipeline (submit -> slot, results,
        drain, close, lanes, calibration) and of a HipContext (profile_*, synchronize, workspace_bytes)."""

        def __init__(self, lanes):
            self.p = oi.NativePipeline(idx, lanes=lanes, max_queries=args.batch, max_query_terms=4, depth=args.depth, k=args.k,
                                       comm=native_comm)
            self.n_slots, self.n = 8, 0
            cs = self.p.concurrent_streams()
            self.calibration = {"chosen_lanes": lanes, "streams_measured_concurrent": cs[0], "streams": cs[1],
                                "note": "oi_pipeline_create draws candidate streams until lanes + 1 run at the same time (measured with a spin kernel)"}
            self.lanes = [(None, None)] * lanes
            self.results = [oi.SearchResult(torch.zeros((args.batch, args.k), dtype=torch.float32, device=dev),
                                            torch.zeros((args.batch, args.k), dtype=torch.int32, device=dev),
                                            torch.zeros((args.batch,), dtype=torch.int32, device=dev)) for _ in range(self.n_slots)]

        def submit(self, qv, qt, qo):
            slot = self.n % self.n_slots
            self.n += 1
            self.p.submit(qv, qt, qo, out=self.results[slot])
            return slot

        def drain(self):
            self.p.drain()

        synchronize = drain

        def profile_reset(self, enable):
            self.p.profile_reset(int(enable))

        def profile_read(self, tag):
            return self.p.profile_read(tag)

        def workspace_bytes(self):
            return self.p.workspace_bytes()

        def close(self):
            self.p.close()

    native_pipe = args.exchange == "native" and (world > 1 or force_dist) and not args.no_pipeline
    if native_pipe:
        pipe = NativePipe(max(1, args.lanes))
        lane_ctxs = [pipe]                                   # (its lanes' contexts are the library's: profiled through the pipeline)
    elif (world > 1 or force_dist or os.environ.get("OI_BENCH_PIPELINE_N1")) and not args.no_pipeline:   # (the env switch: an experiment)
        fuse_ctx = oi.HipContext(local_rank)
        pipe = sharded.ShardedPipeline(sr, fuse_ctx, args.batch, args.depth, args.k)
        if args.lanes > 1:
            # a second lane is kept only if it measurably pays on THIS rank's stream-to-queue placement (calibrate's docstring)
            def make_lane_ctx():
                c = oi.HipContext(local_rank)
                c.set_cosine_mode(MODES[args.cosine])
                if args.no_speculation:
                    c.set_screen_speculation(False)
                if args.lane_bm25 == "inline":
                    c.set_overlap(False)
                return c
            if args.lane_bm25 == "inline":
                pipe.lane0_ctx.set_overlap(False)
            pipe.calibrate(batches, make_lane_ctx, reps=16, placements=args.placements, max_lanes=args.lanes)
        lane_ctxs = [index.ctx for index, _ in pipe.lanes]   # every lane scores on a context of its own (lane 0 too)

    def step():
        qv, qt, qo = batches[step_no[0] % NB]
        step_no[0] += 1
        if world == 1 and not force_dist and not (pipe is not None and pipelined[0]):
            idx.search(qv, qt, qo, k=args.k, depth=args.depth, out=out)   # one C-ABI call: the whole query
            return out.docs
        if pipe is not None and pipelined[0]:
            return pipe.results[pipe.submit(qv, qt, qo)].docs
        if native_comm is not None:
            return idx.search_sharded(native_comm, qv, qt, qo, k=args.k, depth=args.depth, out=out).docs   # one C-ABI call, RCCL inside
        return sr.search(qv, qt, qo, args.k, args.depth, check=False)[1]  # overflow flag checked after the loops

    pipelined = [False]

    def fence():
        if pipe is not None:
            pipe.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pipelined[0] = True
    # --settle-steps (default 0; disclosed as config.settle_steps): extra untimed steps before the warm-up.  Tried in round 4 against
    # the 1-2 % by which the timed back-to-back steps trail the latency loop's isolated batches: no effect (13.31 / 13.21 K QPS
    # without, 13.28 / 13.14 K with 60, alternating on one box), so the gap is not a warm-up effect.
    settle_steps = max(0, args.settle_steps) * world
    for _ in range(settle_steps):
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    ctx.synchronize()   # also surfaces a pool overflow as an error

    # ---------------------------------------------------------------- timed region: exactly K steps
    for c in [ctx] + lane_ctxs:
        c.profile_reset(0 if os.environ.get("OI_BENCH_NO_LIVE_EVENTS") else 2)   # HIP events around the dominant kernel's launches only (each pair costs stream time)
    # The interpreter's cyclic garbage collector stays out of the K steps: a generation-2 pass over this process's heap (torch,
    # numpy, the corpus generators' leftovers) is a 60 ms host stall, and one landed inside the 47 ms timed region of the native
    # exchange in every run (its K calls cost 0.1 ms each on the host) -- round 5's "native is slower under bench.py" was that.
    import gc
    gc.collect()
    if not os.environ.get("OI_BENCH_KEEP_GC"):
        gc.disable()
    fence()
    t0 = time.perf_counter()
    step_host, step_at = [], []
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_host.append(time.perf_counter() - ts)
        step_at.append(ts - t0)
    t_enqueued = time.perf_counter() - t0    # (host time of the K calls alone: shows a host-bound loop)
    gc.enable()
    if os.environ.get("OI_BENCH_STEP_TIMES"):
        sys.stderr.write("[bench] host us per step() call: " + " ".join("%d" % (x * 1e6) for x in step_host) + "\n")
        sys.stderr.write("[bench] step() entered at us: " + " ".join("%d" % (x * 1e6) for x in step_at) + "\n")
    fence()
    elapsed = time.perf_counter() - t0
    tm_own = elapsed
    tm = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    elapsed = float(tm.item())
    if os.environ.get("OI_BENCH_NO_LIVE_EVENTS"):   # experiment: what the kernel's event pairs cost the timed region (no roofline possible)
        if rank == 0:
            print(json.dumps({"value": args.batch * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3, "p50_ms": None,
                              "note": "OI_BENCH_NO_LIVE_EVENTS: timed region without the dominant kernel's HIP events"}))
        return
    # every rank's own set-up and clock (rank 0 prints them all: a scaling number can be tied to the lanes that produced it)
    mine = {"rank": rank, "docs_per_gpu": n_local, "doc_id_base": lo, "own_elapsed_ms_per_step": float(tm_own) / args.steps * 1e3,
            "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
            "lane_calibration": pipe.calibration if pipe is not None else None,
            "lanes": (len(pipe.lanes) if pipe is not None else 1), "exchange": args.exchange if (world > 1 or force_dist) else None,
            # HBM of the searching contexts' workspaces on this rank (the index itself not counted)
            "workspace_GB": round(sum(c.workspace_bytes()[0] for c in [ctx] + lane_ctxs) / 1e9, 3)}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    cos_ms, cos_launches = 0.0, 0
    for c in [ctx] + lane_ctxs:   # (with lanes the launches of two batches overlap: their durations are summed as measured)
        m, l = c.profile_read("cosine")
        cos_ms, cos_launches = cos_ms + m, cos_launches + l
        c.profile_reset(False)
        c.synchronize()   # outside the timed region: a pool overflow in any of the K steps is an error, not a number
    pipelined[0] = False   # everything below (isolated kernel times, exact scorer, latency) runs batch by batch
    # The BM25 leg runs beside the cosine leg on a side stream, so the live cosine duration above includes
    # the CUs it lends to BM25 workgroups.  A few untimed steps with the legs one after the other give the
    # kernel's own duration as well (reported next to the live figure, never instead of it).
    iso_steps = max(1, min(5, args.steps))
    ctx.set_overlap(False)
    step()
    fence()
    ctx.profile_reset(True)
    for _ in range(iso_steps):
        step()
    fence()
    iso_ms, iso_launches = ctx.profile_read("cosine")
    other = {t: ctx.profile_read(t) for t in ("bm25", "select", "rrf")}
    ctx.profile_reset(False)
    ctx.set_overlap(True)

    # The default scorer screens in bf16 and rescores exactly; the same batch through the exact f32 MFMA kernel for
    # every row is measured beside it (untimed steps), so that the line carries both scorers.
    exact_side = None
    stream_side = None
    if args.cosine == "screen" and args.corpus == "f32" and args.batch > 8:
        gate_opened = ctx.profile_read("screen_gate")[0]
        ctx.set_cosine_mode(MODES["exact"])
        step()
        fence()
        ex_steps = max(1, args.steps)   # the same K as the headline: a first-class number, not a spot check
        ctx.profile_reset(2)
        fence()
        t1 = time.perf_counter()
        for _ in range(ex_steps):
            step()
        fence()
        ex_elapsed = time.perf_counter() - t1
        ex_ms, ex_launches = ctx.profile_read("cosine")
        ctx.profile_reset(False)
        ctx.set_cosine_mode(MODES["screen"])
        exact_side = {"ms_per_step": ex_elapsed / ex_steps * 1e3, "queries_per_s": args.batch * ex_steps / ex_elapsed,
                      "cosine_avg_launch_ms": ex_ms / max(1, ex_launches), "steps": ex_steps,
                      "mfma_frac": 2.0 * n_local * args.dim * args.batch * ex_steps / (ex_ms / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                      "note": "same batch, same K steps, oi_set_cosine_mode(OI_COSINE_EXACT): f32 MFMA for every row (rank 0's clock, outside the headline's timed region)"}
        screen_fallback = gate_opened != 0.0
    # rounds 1-4's default beside it: the same screen converting the f32 rows on the fly (4 d bytes per row instead of 2 d)
    # (batches of <= 8 queries too: there the f32 path is the exact GEMV)
    ex_steps = max(1, args.steps)
    if copy_streamed and not args.no_stream_side and args.cosine == "screen" and args.corpus == "f32":
        ctx.set_cosine_mode(MODES["screen-stream"])
        step(); step()
        fence()
        ctx.profile_reset(2)
        fence()
        t1 = time.perf_counter()
        for _ in range(ex_steps):
            step()
        fence()
        st_elapsed = time.perf_counter() - t1
        st_ms, st_launches = ctx.profile_read("cosine")
        ctx.profile_reset(False)
        ctx.set_cosine_mode(MODES["screen"])
        stream_side = {"ms_per_step": st_elapsed / ex_steps * 1e3, "queries_per_s": args.batch * ex_steps / st_elapsed, "steps": ex_steps,
                       "screen_ms_per_step": st_ms / ex_steps,
                       "hbm_frac_on_its_bytes": 4.0 * n_local * args.dim * ((args.batch + 63) // 64) * ex_steps / (st_ms / 1e3) / 1e9 / PEAK_HBM_GBS,
                       "note": "oi_set_cosine_mode(OI_COSINE_SCREEN_STREAM): rounds 1-4's headline -- the same bound, the same survivors, the same "
                               "exact f32 rescoring, identical lists (tests/test_gpu_prefilter.py); the screen converts the f32 rows on the fly "
                               "(4 d bytes per row and batch)" if args.batch > 8 else
                               "oi_set_cosine_mode(OI_COSINE_SCREEN_STREAM) with <= 8 queries: rounds 1-4's exact f32 GEMV over the f32 rows "
                               "(cosine_gemv_filter, 4 d bytes per row and batch) -- without a copy to stream a screen would read the same bytes"}

    # N = 1: the same K steps through the library's own two-lane pipeline (oi_pipeline_*: what a serving host on the C ABI would
    # run for throughput) -- batch i+1's corpus stream beside the selects / rescoring / fusion of batch i.  A side number: the
    # headline stays one oi_search call per batch, one batch in flight, as in every earlier round.
    pipelined_side = None
    if world == 1 and not force_dist and not args.no_pipelined_side and args.batch > 8:
        npipe = oi.NativePipeline(idx, lanes=2, max_queries=args.batch, max_query_terms=4, depth=args.depth, k=args.k)
        outs = [oi.SearchResult(torch.zeros((args.batch, args.k), dtype=torch.float32, device=dev),
                                torch.zeros((args.batch, args.k), dtype=torch.int32, device=dev),
                                torch.zeros((args.batch,), dtype=torch.int32, device=dev)) for _ in range(8)]
        for i in range(8):
            npipe.submit(*batches[i % NB], out=outs[i])
        npipe.drain()
        same = True
        for i in range(min(NB, 8)):
            idx.search(*batches[i], k=args.k, depth=args.depth, out=out)
            ctx.synchronize()
            same = same and bool(torch.equal(out.docs, outs[i].docs)) and bool(torch.equal(out.scores, outs[i].scores))
        fence()
        t1 = time.perf_counter()
        for i in range(args.steps):
            npipe.submit(*batches[i % NB], out=outs[i % 8])
        npipe.drain()
        torch.cuda.synchronize()
        pp_elapsed = time.perf_counter() - t1
        pipelined_side = {"ms_per_step": pp_elapsed / args.steps * 1e3, "queries_per_s": args.batch * args.steps / pp_elapsed,
                          "steps": args.steps, "lanes": 2, "bit_identical_to_oi_search": same,
                          "workspace_GB": round(npipe.workspace_bytes()[0] / 1e9, 3),
                          "streams_measured_concurrent": "%d of %d" % npipe.concurrent_streams(),
                          "note": "oi_pipeline_create(idx, NULL, 2 lanes) / oi_pipeline_submit x K / oi_pipeline_drain: two batches in flight "
                                  "through lanes the library owns; NOT the headline (value = one oi_search call per batch, one batch in flight)"}
        npipe.close()

    # ---------------------------------------------------------------- latency: >= 200 timed batches after 20 warm-ups,
    # whatever --steps says.  (a) device-resident: HIP events on the launch stream around ONE batch at a time (queries and
    # results in HBM), plus the host's wall clock around the same batch (call + sync); (b) host-inclusive: the OI_HOST entry
    # point -- H2D of the queries (196 KB of vectors + terms at B=64), the whole query, D2H of the top-k -- wall clock.
    NL, NW = max(1, args.latency_batches), max(0, args.latency_warmup)

    def pct(v, p):
        v = sorted(v)
        return v[min(len(v) - 1, max(0, int(round(p * (len(v) - 1)))))]

    for _ in range(NW):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(NL)]
    lat_wall = []
    for a, b in ev:
        if world > 1:
            dist.barrier()          # ranks enter every batch together (a collective inside would wait for the slowest anyway)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        a.record()
        step()
        b.record()
        torch.cuda.synchronize()
        lat_wall.append((time.perf_counter() - t1) * 1e3)
    lat = [a.elapsed_time(b) for a, b in ev]

    host_batches = [tuple(x.cpu().numpy() for x in b) for b in batches]
    host_batches = [(qv, qt.view(np.uint32) if qt.dtype != np.uint32 else qt, qo.view(np.uint32) if qo.dtype != np.uint32 else qo)
                    for qv, qt, qo in host_batches]

    def host_step(i):
        qv, qt, qo = host_batches[i % NB]
        if world == 1 and not force_dist:
            return idx.search(qv, qt, qo, k=args.k, depth=args.depth).docs     # OI_HOST: numpy in, numpy out, synchronous
        if native_comm is not None:
            return idx.search_sharded(native_comm, qv, qt, qo, k=args.k, depth=args.depth).docs       # OI_HOST through the library
        d = [torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) for x in (qv, qt, qo)]
        return sr.search(d[0], d[1], d[2], args.k, args.depth, check=False)[1].cpu().numpy()

    for i in range(NW):
        host_step(i)
    fence()
    lat_host = []
    for i in range(NL):
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        host_step(i)
        lat_host.append((time.perf_counter() - t1) * 1e3)
    fence()
    ctx.synchronize()

    # a result sanity check outside the timed region: every row full, ids in range
    docs = step().cpu().numpy()
    assert docs.shape == (args.batch, args.k) and docs.min() >= 0 and docs.max() < args.docs

    if rank == 0:
        qps = args.batch * args.steps / elapsed
        # dominant kernel: the cosine scorer.  Algorithmic work per step on this rank
        # (SURVEY.md 8d): flops = 2 * n_local * d * B, bytes = n_local * d * 4 (corpus read once per batch)
        flops_step = 2.0 * n_local * args.dim * args.batch
        # ALGORITHMIC bytes: the corpus is read once per batch, whatever the kernel does.  A scorer that
        # keeps the queries in registers takes one corpus pass per 64 queries (bf16 corpus, d = 1024: 64 per pass of
        # the pair kernel): its extra passes are re-reads and count AGAINST it here (frac falls), never for it.
        bytes_step = (2.0 if args.corpus == "bf16" else 4.0) * n_local * args.dim
        if copy_streamed:
            bytes_step = 2.0 * n_local * args.dim   # the screen streams the index's bf16 screening copy: 2 d bytes per row, once per batch
        passes = (args.batch + 63) // 64 if args.batch > 8 else 1
        if args.corpus == "bf16":   # the bf16-corpus scorer's plan (cosine_bf16.hip, cb_group)
            left, passes, solo = args.batch, 0, (32 if args.dim == 1024 else 64)
            while left > 0:
                g = solo if left <= solo else (128 if left > 96 else 96) if args.dim == 1024 else (96 if (left + 95) // 96 < (left + 63) // 64 else 64)
                left -= min(g, left)
                passes += 1
        cos_s = cos_ms / 1e3 if cos_ms > 0 else float('nan')   # (nan: OI_BENCH_NO_LIVE_EVENTS, an experiment without the kernel's events)
        if args.batch > 8 and args.corpus == "f32" and args.cosine == "exact" or (args.cosine in ("screen", "screen-copy", "screen-stream") and args.dim not in (384, 768) and args.batch > 8 and args.corpus == "f32"):
            roof = {"bound": "mfma", "achieved": flops_step * args.steps / cos_s / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": bytes_step * args.steps / cos_s / 1e9, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["traffic"] = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        # the committed PMC pass was taken on the default workload at N=1 only
        if os.path.exists(pmc) and (args.docs, args.dim, args.batch, world, args.corpus) == (10_000_000, 768, 64, 1, "f32"):
            try:
                pm = json.load(open(pmc))   # the screen kernel's counters; the exact kernel's under "exact_kernel"
                roof["traffic"] = (pm if (args.cosine in ("screen", "screen-copy") and copy_streamed and "cosine_copy_screen" in pm.get("kernel", ""))
                                   else pm.get("f32_stream_kernel", {}) if args.cosine in ("screen", "screen-stream") and not copy_streamed
                                   else pm.get("exact_kernel", {}) if args.cosine == "exact" else {}).get("cosine_hbm_bytes_per_launch")
                roof["traffic_source"] = ("NOT measured in this run: read from the committed profiles/pmc_traffic.json (%s), a separate "
                                          "rocprofv3 --pmc pass of this workload (tools/pmc_profile.sh; FETCH_SIZE/WRITE_SIZE with the "
                                          "guide's gfx950 corrections), per launch" % pm.get("source", "see file"))
            except Exception:
                pass
        roof["kernel"] = "cosine scorer (all corpus-chunk launches of a batch)"
        if args.corpus == "bf16":
            roof["kernel"] = "cosine over the bf16 corpus (bf16 MFMA, all corpus-chunk launches of a batch)"
        roof["corpus_passes_per_batch"] = passes
        if args.corpus == "bf16" and args.dim == 1024 and args.batch >= 256 and args.batch % 256 == 0:
            roof["sibling_workgroups"] = ("every 256 queries are ONE launch of sibling workgroups (cosine_bf16_quad<.., SIB=2>): two workgroups of an XCD walk "
                                          "the same tile sequence with 128 queries each and share every tile through that XCD's L2 -- the CUs stream the "
                                          "corpus twice (corpus_passes_per_batch), HBM sees it once (FETCH_SIZE: 25.63 GB per batch at 12.5M x 1024, "
                                          "profiles/r05_config4_sibling_pmc.txt)")
        roof["hbm_GBs_streamed"] = bytes_step * passes * args.steps / (cos_ms / 1e3) / 1e9
        if args.cosine in ("screen", "screen-copy", "screen-stream") and roof["bound"] == "hbm" and (args.batch > 8 or copy_streamed) and args.corpus == "f32":
            roof["kernel"] = ("cosine_copy_screen (bf16 MFMA over the index's bf16 screening copy, all corpus-chunk launches of a batch)" if copy_streamed
                              else "cosine_screen_filter (bf16 MFMA over the f32 corpus converted on the fly, all corpus-chunk launches of a batch)")
            roof["algorithmic_bytes_per_launch"] = bytes_step / max(1.0, cos_launches / max(1, args.steps))
            roof["algorithmic_bytes_note"] = ("%d rows x %d dims x %d B, read once per 64-query batch by the screen; the exact f32 rescoring of the "
                                              "~2.45 k' survivors per query (~0.48 GB of scattered 3-KB f32 rows per batch at 10M x 768) is a separate "
                                              "kernel (other_kernels_ms_per_step.rescore)" % (n_local, args.dim, 2 if copy_streamed else 4))
        roof["launches_per_step"] = cos_launches / max(1, args.steps)
        roof["avg_launch_ms"] = cos_ms / max(1, cos_launches)
        roof["kernel_ms_per_step"] = cos_ms / max(1, args.steps)
        roof["hbm_GBs_algorithmic"] = bytes_step * args.steps / cos_s / 1e9
        roof["legs"] = "BM25 leg overlapped on a side stream during the timed region (oi_set_overlap)"
        roof["step_level"] = {"achieved": bytes_step * passes * args.steps / elapsed / 1e9, "unit": "GB/s",
                              "frac": bytes_step * passes * args.steps / elapsed / 1e9 / PEAK_HBM_GBS,
                              "note": "this rank's corpus bytes streamed / the whole timed region (everything else included)"}
        if pipe is not None and len(pipe.lanes) > 1:
            # two batches' screens share the chip: each launch's duration includes the time it ran beside the other lane's,
            # so the per-launch figure above understates the rate the corpus is streamed at; the step-level figure does not
            roof["overlapped_launches"] = ("%d batches in flight per rank: launch durations overlap and are summed as measured "
                                           "(frac understates the kernel); see step_level and isolated" % len(pipe.lanes))
        roof["isolated"] = {"avg_launch_ms": iso_ms / max(1, iso_launches),
                            "frac": (flops_step if roof["bound"] == "mfma" else bytes_step) * iso_steps / (iso_ms / 1e3)
                            / (1e12 if roof["bound"] == "mfma" else 1e9) / roof["peak"],
                            "note": "%d extra untimed steps with the legs serialised" % iso_steps}
        line = {
            "metric": "queries/sec + p50 latency, 10M-post/768-d hybrid BM25+cosine+RRF top-100",
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 (bf16x3 split products, f32 accumulate)" if args.cosine == "split" else args.corpus,
            "data": "synthetic",
            "config": {"workload": "%s: %d posts x %d-d %s, batch %d queries x 4 BM25 terms, "
                                   "per-list depth %d, RRF top-%d; corpus row-sharded over %d GPU(s)" % (
                                       "custom (bf16 corpus, the configs[4] regime)" if args.corpus == "bf16" else
                                       "BASELINE configs[2]" if (args.docs, args.batch) == (10_000_000, 64) else
                                       "BASELINE configs[1]" if (args.docs, args.batch) == (1_000_000, 1) else "custom",
                                       args.docs, args.dim, args.corpus, args.batch, args.depth, args.k, world),
                       "docs": args.docs, "dim": args.dim, "batch": args.batch, "k": args.k, "depth": args.depth,
                       "vocab": args.vocab, "docs_per_gpu": n_local, "tokens_rank0": n_tokens_local,
                       "query_batches_rotated": NB, "settle_steps": settle_steps,
                       "cosine_scorer": {"screen": "bf16 screen with a proven error bound + exact f32 rescoring of the survivors from the f32 rows "
                                                   "(the exact scorer's lists; gated exact fallback); the screen streams " +
                                                   ("the index's bf16 screening copy (made at finalize, +%.2f GB of HBM)" % (screen_copy_b / 1e9) if copy_streamed
                                                    else "the f32 rows, converted on the fly (the index holds no screening copy)"),
                                         "exact": "f32 MFMA for every row", "split": "bf16x3 split products",
                                         "screen-copy": "bf16 screen over the bf16 screening copy (made on first use if missing) + exact f32 rescoring",
                                         "screen-stream": "bf16 screen over the f32 rows converted on the fly + exact f32 rescoring (rounds 1-4's default)"}[args.cosine]
                                        if args.corpus == "f32" and (args.batch > 8 or copy_streamed) else "exact",
                       "screen_thresholds": ("speculative between the corpus chunks (the r-th best screen score so far, r = 3 k' m / n + 12, minus the margin; "
                                             "checked on the device against the proven final threshold: a failed check opens the exact pipeline and the "
                                             "ctx backs off) -- oi_set_screen_speculation, default; --no-speculation: proven only"
                                             if (args.corpus == "f32" and args.batch > 8 and args.cosine in ("screen", "screen-copy", "screen-stream")
                                                 and args.dim in (384, 768) and not args.no_speculation) else "proven only"),
                       "index_GB": {"rows_owned_by_library": rows_owned_b / 1e9, "rows_borrowed_from_caller": (0.0 if rows_owned_b else
                                    (2.0 if args.corpus == "bf16" else 4.0) * n_local * args.dim / 1e9),
                                    "screening_copy": screen_copy_b / 1e9, "bm25_structures": bm25_index_b / 1e9,
                                    "hbm_free_after_build": hbm_free_b / 1e9, "hbm_total": hbm_total_b / 1e9},
                       "parallelism": "row-shard x%d + all-gather of per-shard lists" % world +
                                      ("; exchange + fusion of batch i overlap the lists of batch i+1 (two streams)" if pipe is not None else "") +
                                      ("; %d batches' lists in flight per rank, each through its own view of the shard" % len(pipe.lanes)
                                       if pipe is not None and len(pipe.lanes) > 1 else "") +
                                      ("; lanes, streams and the RCCL all-gather inside the library (oi_pipeline_*, oi_comm_*)" if native_pipe else ""),
                       "backend": (backend if (world > 1 or force_dist) else None),
                       "forced_process_group_of_one": force_dist,
                       "per_rank": per_rank},
            "p50_ms": pct(lat, 0.5), "p95_ms": pct(lat, 0.95),
            "p50_host_ms": pct(lat_host, 0.5), "p95_host_ms": pct(lat_host, 0.95),
            "latency": {"batches": NL, "warmup": NW,
                        "device": {"p50_ms": pct(lat, 0.5), "p95_ms": pct(lat, 0.95), "min_ms": min(lat), "max_ms": max(lat),
                                   "clock": "HIP events on the launch stream around one batch; queries and results resident in HBM"},
                        "device_wall": {"p50_ms": pct(lat_wall, 0.5), "p95_ms": pct(lat_wall, 0.95),
                                        "clock": "host perf_counter around the same batch: call + stream sync"},
                        "host_inclusive": {"p50_ms": pct(lat_host, 0.5), "p95_ms": pct(lat_host, 0.95), "min_ms": min(lat_host), "max_ms": max(lat_host),
                                           "clock": "host perf_counter around the OI_HOST entry point: H2D of %d query bytes, the query, D2H of %d result bytes"
                                                    % (sum(int(x.nbytes) for x in host_batches[0]), args.batch * args.k * 8 + args.batch * 4)},
                        "note": "rank 0's clocks; one batch in flight at a time (the timed throughput region above pipelines batches when N > 1)"},
            "roofline": roof,
            "other_kernels_ms_per_step": dict({t: v[0] / iso_steps for t, v in other.items()},
                                              note="from the %d serialised steps, not the timed region" % iso_steps),
            "build_s": t_build,
        }
        # BM25 leg: term-at-a-time reads 8 B per posting of the batch's terms + 8 B per (doc block, term) bounds
        # lookup; its kernels' time comes from the serialised steps above (the leg runs beside the cosine leg otherwise).
        if bm25_bytes is not None and other["bm25"][0] > 0:
            bm_ms = other["bm25"][0] / iso_steps
            line["bm25_roofline"] = {"bound": "hbm", "achieved": bm25_bytes / (bm_ms / 1e3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                     "frac": bm25_bytes / (bm_ms / 1e3) / 1e9 / PEAK_HBM_GBS, "kernel_ms_per_step": bm_ms,
                                     "launches_per_step": other["bm25"][1] / iso_steps, "algorithmic_bytes_per_step": bm25_bytes,
                                     "kernel": {"stream": "bm25_stream_kernel (term-at-a-time; every wave streams a weight-balanced range of "
                                                          "(query, block) tasks through its LDS ring) + bm25_plan_kernel",
                                                "wave": "bm25_wave_kernel (term-at-a-time, one wave per (doc block, query) task)"}.get(
                                                    args.bm25, "bm25 (%s)" % args.bm25),
                                     "note": "algorithmic bytes = 8 B x postings of the batch's terms + 12 B (stream; 8 B wave) x (blocks x "
                                             "terms); at this batch size the kernel is latency/issue-bound, not byte-bound (DESIGN 4.3)"}
        if exact_side is not None:
            line["headline_note"] = ("value/roofline are the default SCREENED scorer (bf16 screen + exact f32 rescoring from the f32 rows, bound hbm; "
                                     "error bound built from measured rounding errors and adversarially tested, DESIGN 4.1). Round 5: the screen "
                                     "streams the index's bf16 screening copy (a derived structure made at finalize, like the BM25 impact postings): "
                                     "roofline is computed on THAT kernel's bytes (2 d per row). f32_stream_scorer = rounds 1-4's headline (the same "
                                     "screen over the f32 rows, identical lists); exact_scorer = the north star's f32-MFMA GEMM over every row; "
                                     "same K steps each; --cosine screen-stream / exact make them the headline.")
            line["exact_scorer"] = exact_side
            line["screen_fell_back_to_exact"] = screen_fallback
            try:   # (rank 0's ctx: checks of speculative thresholds that failed -- each such batch was rescored by the exact pipeline)
                sf, sn = ctx.speculation_state()
                line["screen_speculation"] = {"failed_checks": sf, "searches_that_speculated": sn}
            except Exception:
                pass
        if stream_side is not None:
            line["f32_stream_scorer"] = stream_side
        if pipelined_side is not None:
            line["pipelined_native"] = pipelined_side
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed on rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(args.docs, args.dim, args.vocab, args.depth, args.k,
                                                args.cpu_sample_docs, args.cpu_sample_queries)
        if not args.no_text_paths and world == 1:    # the reference-pinned paths, on rank 0 at N=1 (their own inputs; outside the timed region)
            line.update(text_paths(oi, ctx, dev, args.text_items, 10, args.text_cpu_sample))
        if (not args.no_other_configs and world == 1 and not force_dist and
                (args.docs, args.dim, args.batch, args.corpus, args.cosine) == (10_000_000, 768, 64, "f32", "screen")):
            line["other_configs"] = other_configs()
        line["library"] = os.path.relpath(_oil.LIB_PATH, ROOT)   # what was measured (the package loader has no override)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if pipe is not None:
        pipe.close()
        if fuse_ctx is not None:
            fuse_ctx.close()
    if native_comm is not None:
        native_comm.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
