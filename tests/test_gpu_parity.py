"""GPU parity tests: the HIP path (through the C ABI of libopenintel_hip.so) vs the CPU oracle on
the same seeded inputs.  Run on a real MI355X:  python -m pytest tests -m gpu

Bars (written where they are applied):
  * lexicon path (reference-pinned): per-post polarity f64 BIT-EXACT, flags exact, integer
    summary counters exact, polarity_sum within n * 2^-52 * max|partial sum| of the input-order sum;
  * BM25 / RRF / merge: scores and doc-id order BIT-EXACT vs the (unpinned) oracle;
  * cosine: scores within 1e-5 absolute (f32 MFMA/FMA order differs from the f64 oracle).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COS_TOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    import openintel_amd as oi
    from openintel_amd import _lib
    c = oi.HipContext(0)
    c.set_cosine_mode(_lib.OI_COSINE_EXACT)   # this module pins the exact kernels; the default (screen + exact
    yield c                                   # rescoring) has the same bars in tests/test_gpu_prefilter.py
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import lib
    return lib


# ----------------------------------------------------------------------------- lexicon path
def _analyze(ctx, texts):
    import openintel_amd as oi
    blob, offs = oi.pack_posts(texts)
    return oi.HipLexiconAnalyzer(ctx).analyze_packed(blob, offs), (blob, offs)


def _check_lexicon(ctx, O, texts):
    (pol, spec), (blob, offs) = _analyze(ctx, texts)
    rpol, rspec = O.lexicon_analyze(blob if blob.size else np.zeros(1, np.uint8), offs)
    bad = np.nonzero((pol.view(np.uint64) != rpol.view(np.uint64)) | (spec != rspec))[0]
    assert bad.size == 0, "first mismatch at post %d: %r gpu=(%r,%r) ref=(%r,%r)" % (
        bad[0], texts[bad[0]][:120], pol[bad[0]], spec[bad[0]], rpol[bad[0]], rspec[bad[0]])
    return pol, spec


def test_lexicon_reference_fixture(ctx, O, golden):
    texts = [p["text"] for p in golden["fixture_posts"]]
    pol, spec = _check_lexicon(ctx, O, texts)
    assert pol.tolist() == [s["polarity"] for s in golden["derived"]["signals"]]
    assert spec.astype(bool).tolist() == [s["speculative"] for s in golden["derived"]["signals"]]
    # lexicon.rs:106-120
    pol, spec = _check_lexicon(ctx, O, [c["text"] for c in golden["lexicon_test"]])
    assert [int(np.sign(p)) for p in pol] == [c["polarity_sign"] for c in golden["lexicon_test"]]


def test_lexicon_port_object_and_engine_end_to_end(ctx, golden):
    import openintel_amd as oi
    posts = [oi.SocialPost(p["id"], oi.SourceKind.REDDIT if p["source"] == "reddit" else oi.SourceKind.BLUESKY,
                           p["author"], oi.PostText.parse(p["text"]), None, p["engagement"])
             for p in golden["fixture_posts"]]
    analyzer: oi.PostAnalyzer = oi.HipLexiconAnalyzer(ctx)
    signals = analyzer.analyze(posts)
    assert len(signals) == len(posts)                                  # post_analyzer.rs:9
    m = golden["mock_market"]
    tk = oi.Ticker.parse("AAPL")
    snap = oi.MarketSnapshot(tk, m["last_price"], m["previous_close"], m["volume"], m["avg_volume"],
                             m["realized_vol"], m["put_call_ratio"], m["iv_rank"])
    rep = oi.SpeculationEngine.aggregate(tk, posts, signals, snap, None, oi.EngineConfig())
    d = golden["derived"]["summary"]
    assert rep.social.total_mentions == 10 and rep.fusion.alignment.value == "confirming_bullish"  # analyze_flow.rs:128-129
    assert rep.social.net_sentiment == d["net_sentiment"] and rep.social.speculation_index == d["speculation_index"]
    assert (rep.social.bullish, rep.social.bearish, rep.social.neutral) == (7, 2, 1)
    assert rep.social.bull_bear_ratio == d["bull_bear_ratio"] and rep.fusion.crowding == d["crowding"]
    assert rep.market.pct_change == d["pct_change"] and rep.market.rvol == d["rvol"]
    assert rep.social_confidence.value == "medium"
    # same report from the GPU reduction
    src = np.array([int(p.source) for p in posts], np.uint8)
    pol = np.array([s.polarity for s in signals]); spec = np.array([s.speculative for s in signals], np.uint8)
    cnt = oi.SpeculationEngine.social_counters(ctx, src, pol, spec, oi.EngineConfig())
    rep2 = oi.SpeculationEngine.aggregate_counters(tk, cnt, snap, None, oi.EngineConfig())
    assert rep2.social == rep.social and rep2.fusion.crowding == rep.fusion.crowding
    assert rep2.fusion.alignment == rep.fusion.alignment
    # speculation_engine.rs:335-355 length mismatch
    with pytest.raises(oi.AnalyzerMismatch):
        oi.SpeculationEngine.social_counters(ctx, src, pol[:5], spec[:5], oi.EngineConfig(), n_posts=10)
    with pytest.raises(oi.AnalyzerMismatch):
        oi.SpeculationEngine.aggregate(tk, posts, signals[:3], None, None, oi.EngineConfig())


def test_lexicon_unicode_and_edge_cases(ctx, O):
    from tests.test_oracle_golden import UNICODE_CASES
    texts = list(UNICODE_CASES)
    texts += ["", "a", "up", "", "", "iv", "x" * 15 + " moon", "x" * 16 + "moon", "moon" * 4, "y" * 4095 + " up",
              "K" * 9, "pumK", "Kpump", "buK buy", "İv iv", "moonİ", "caKlls"]
    # tokens that straddle 16-byte lanes and 4 KiB sub-tiles at every phase
    for pad in range(0, 40):
        texts.append("z" * pad + " squeeze " + "q" * (4096 - pad) + " bagholder rocket")
    # a post far longer than one sub-tile, lexicon words everywhere
    texts.append(" ".join(["moon", "dump", "filler", "0dte", "UP", "Down."] * 3000))
    # posts that END exactly where a token would continue in the next post
    texts += ["to the mo", "on calls", "pu", "ts", "b", "uy", "sell"]
    _check_lexicon(ctx, O, texts)


def test_lexicon_chunk_half_and_subtile_phases(ctx, O):
    """The third-generation scan's own seams: 32-byte halves, 64-byte lane chunks, 16-byte units, 16 KiB sub-tiles,
    the per-lane candidate columns (more than four candidates in one chunk) and the exact-path window around
    U+212A / U+0130 -- every phase of each, plus posts that begin or end exactly on them."""
    texts = []
    for pad in range(0, 72):      # lexicon words sliding over a half / chunk seam, preceded and followed by separators
        texts.append("#" * pad + "squeeze bagholder up iv 0dte contracts" + "#" * 7)
        texts.append("x" * pad + " up")                              # a 2-char word whose second char opens the next chunk
        texts.append("." * pad + "pumK rocKet tanK caKlls İv moonİ up İ striKe")   # exact-path bytes at every unit / chunk offset
    for pad in range(0, 40):      # ... and over a 16 KiB sub-tile seam
        texts.append("q" * (16384 - 20 + pad - 1) + " rocket drilling theta")
        texts.append("e" * (16384 + pad - 3) + " " + "K" + "pump pumK")
    texts.append("up " * 3000)                                       # 21 candidates per chunk: the columns overflow
    texts.append("iv.up,0dte;itm otm" * 1500)
    texts.append(" ".join(["bull", "bears", "bear", "bulls", "red", "reds", "tank", "tanks"] * 1200))  # Bloom false positives
    # posts that start / end exactly on a chunk, a half, a sub-tile (the packed blob has no separators between posts)
    texts += ["m" * 64, "oon moon", "c" * 31 + " ", "up", "", "", "s" * 16383, "ell sell", "", "y" * 16384, "olo yolo"]
    texts += ["gamma"] * 700 + [""] * 300 + ["delta vega"] * 300       # many posts per chunk, runs of empty posts
    _check_lexicon(ctx, O, texts)


def test_lexicon_fuzz_against_the_oracle(ctx, O):
    """Random posts over an alphabet that is hostile to the scan: lexicon words and their near misses, every kind of
    separator, upper case, digits, U+212A / U+0130 and other multi-byte chars, empty posts, posts of 1 byte up to
    several sub-tiles -- 12 seeds, each one packed blob (no separators between posts), bit for bit against the oracle."""
    from openintel_amd import synth
    lex = list(synth.LEXICON_WORDS)
    near = [w[:-1] for w in lex if len(w) > 2] + [w + "s" for w in lex] + [w.upper() for w in lex] + [w.capitalize() for w in lex]
    pieces = lex * 3 + near + ["\u212a", "\u0130", "\u00e9", "\U0001F680", "\u0430", "\u00aa", "\u00b0", "\u4e2a",
                               "0", "7", "42", "a", "I", "x" * 9, "y" * 10, "z" * 17]
    seps = [" ", " ", " ", "", "", ".", ",", "\n", "\t", "-", "'", "  ", "$", "\u2014", "/", "_", "\x00", "\x11"]
    for seed in range(12):
        rng = np.random.default_rng(1000 + seed)
        texts = []
        for _ in range(int(rng.integers(200, 1500))):
            kind = rng.random()
            n_tok = 0 if kind < 0.1 else int(rng.integers(1, 4)) if kind < 0.4 else int(rng.integers(4, 60)) if kind < 0.97 \
                else int(rng.integers(2000, 9000))
            ids = rng.integers(0, len(pieces), size=n_tok)
            sp = rng.integers(0, len(seps), size=n_tok)
            texts.append("".join(pieces[i] + seps[j] for i, j in zip(ids, sp)))
        _check_lexicon(ctx, O, texts)


def test_lexicon_many_tiny_posts_and_random_corpus(ctx, O):
    from openintel_amd import synth
    rng = np.random.default_rng(5)
    words = synth.LEXICON_WORDS + ["the", "a", "zz", "é", "K", "UP", "Moon", "\U0001F680"]
    tiny = [str(rng.choice(words)) if rng.random() < 0.8 else "" for _ in range(5000)]
    _check_lexicon(ctx, O, tiny)
    texts = synth.posts_np(20000)
    for i in range(0, len(texts), 7):  # sprinkle upper case / punctuation / non-ASCII
        texts[i] = texts[i].upper().replace(" ", ", ", 3) + " — \U0001F680"
    pol, spec = _check_lexicon(ctx, O, texts)
    assert 0 < (pol != 0).mean() < 1 and 0 < spec.mean() < 1


def test_lexicon_device_buffers_and_summary(ctx, O):
    import torch
    import openintel_amd as oi
    from openintel_amd import synth
    dev = torch.device("cuda:0")
    blob, offs = synth.posts_torch(300_000, dev)
    n = offs.numel() - 1
    pol = torch.zeros(n, dtype=torch.float64, device=dev)
    spec = torch.zeros(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    oi.HipLexiconAnalyzer(ctx).analyze_device(blob, offs, pol, spec)
    ctx.synchronize()
    rpol, rspec = O.lexicon_analyze(blob.cpu().numpy(), offs.cpu().numpy().astype(np.uint64))
    assert np.array_equal(pol.cpu().numpy().view(np.uint64), rpol.view(np.uint64))
    assert np.array_equal(spec.cpu().numpy(), rspec)
    src = (torch.arange(n, device=dev) % 3 == 0).to(torch.uint8)
    cfg = oi.EngineConfig()
    cnt = oi.SpeculationEngine.social_counters(ctx, src, pol, spec, cfg)
    ref = O.social_summary(src.cpu().numpy(), rpol, rspec)
    assert (cnt.total, cnt.bullish, cnt.bearish, cnt.neutral, cnt.spec_count) == (
        n, ref.bullish, ref.bearish, ref.neutral, ref.spec_count)                      # exact
    assert list(cnt.by_source) == list(ref.mentions_by_source)                          # exact
    bound = n * 2.0 ** -52 * max(1.0, float(np.abs(np.cumsum(rpol)).max()))
    assert abs(cnt.polarity_sum - ref.polarity_sum) <= bound
    # and a second run gives the identical bits (fixed-shape tree)
    cnt2 = oi.SpeculationEngine.social_counters(ctx, src, pol, spec, cfg)
    assert cnt2.polarity_sum == cnt.polarity_sum


def test_lexicon_scan_with_fused_summary(ctx, O):
    """oi_lexicon_summary_device: A1-A4 in one pass ("0 out if fused with the A4 reduction").  With no per-post
    outputs the sums must be the oracle's (integers exact, polarity_sum inside the reassociation bound, identical run to
    run); with outputs given they are oi_lexicon_analyze_device's bit for bit; ragged sizes, one post, sources absent."""
    import torch
    import openintel_amd as oi
    from openintel_amd import synth
    dev = torch.device("cuda:0")
    an = oi.HipLexiconAnalyzer(ctx)
    blob_all, offs_all = synth.posts_torch(260_001, dev)
    for n in (260_001, 512, 513, 1):
        offs = offs_all[:n + 1].contiguous()
        blob = blob_all[:int(offs[-1])].contiguous()
        src = (torch.arange(n, device=dev) % 7 < 3).to(torch.uint8)
        c0 = an.summary_device(blob, offs, src)                                   # nothing per post written
        pol = torch.full((n,), 7.0, dtype=torch.float64, device=dev)
        spec = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        c1 = an.summary_device(blob, offs, src, d_polarity=pol, d_speculative=spec)
        c2 = an.summary_device(blob, offs, None)
        rpol, rspec = O.lexicon_analyze(blob.cpu().numpy(), offs.cpu().numpy().astype(np.uint64))
        ref = O.social_summary(src.cpu().numpy(), rpol, rspec)
        ints = lambda c: (c.total, c.by_source[0], c.by_source[1], c.bullish, c.bearish, c.neutral, c.spec_count)
        want = (n, ref.mentions_by_source[0], ref.mentions_by_source[1], ref.bullish, ref.bearish, ref.neutral, ref.spec_count)
        assert ints(c0) == want and ints(c1) == want
        assert ints(c2) == (n, 0, 0, ref.bullish, ref.bearish, ref.neutral, ref.spec_count)   # no sources: no histogram
        bound = n * 2.0 ** -52 * max(1.0, float(np.abs(np.cumsum(rpol)).max()))
        assert abs(c0.polarity_sum - ref.polarity_sum) <= bound
        assert c0.polarity_sum == c1.polarity_sum == c2.polarity_sum                       # fixed-shape tree: same bits
        assert np.array_equal(pol.cpu().numpy().view(np.uint64), rpol.view(np.uint64))
        assert np.array_equal(spec.cpu().numpy(), rspec)
    # the reference's fixture through the fused path: the pinned summary
    texts = ["AAPL to the moon", "buy AAPL calls", "AAPL puts printing, crash incoming", "nothing to see"]
    b, o = oi.pack_posts(texts)
    tb = torch.from_numpy(np.concatenate([b, np.zeros(64, np.uint8)])).to(dev)
    c = an.summary_device(tb[:b.size], torch.from_numpy(o.astype(np.int64)).to(dev), None)
    assert (c.total, c.bullish, c.bearish, c.neutral, c.spec_count) == (4, 2, 1, 1, 2) and c.polarity_sum == 1.0


def test_sharded_analyzer_hip_shards_equal_the_oracle(ctx, O):
    """SURVEY 8(e) row 2 on the GPU: the posts cut into 3 shards, each through the HIP scan + summary reduction, the
    per-shard counters combined exactly as ShardedAnalyzer.summary combines the all-gathered words (the gloo test
    covers the collective itself).  Integer fields exact; polarity_sum = the rank-order sum of the shard partials."""
    import torch
    import openintel_amd as oi
    from openintel_amd import sharded, synth
    dev = torch.device("cuda:0")
    blob, offs = synth.posts_torch(200_000, dev)
    n = offs.numel() - 1
    src = (torch.arange(n, device=dev) % 5 < 2).to(torch.uint8)
    sa = sharded.make_hip_sharded_analyzer(ctx, dev)
    whole = sa.summary(blob, offs, src)                                   # world 1
    rpol, rspec = O.lexicon_analyze(blob.cpu().numpy(), offs.cpu().numpy().astype(np.uint64))
    ref = O.social_summary(src.cpu().numpy(), rpol, rspec)
    ints = lambda c: (c.total, c.by_source[0], c.by_source[1], c.bullish, c.bearish, c.neutral, c.spec_count)
    assert ints(whole) == (n, ref.mentions_by_source[0], ref.mentions_by_source[1], ref.bullish, ref.bearish,
                           ref.neutral, ref.spec_count)
    bound = n * 2.0 ** -52 * max(1.0, float(np.abs(np.cumsum(rpol)).max()))
    assert abs(whole.polarity_sum - ref.polarity_sum) <= bound
    cuts = [0, 70_001, 70_001 + 50_000, n]                                # ragged shards, unaligned byte starts
    parts = []
    for lo, hi in zip(cuts, cuts[1:]):
        b0, b1 = int(offs[lo]), int(offs[hi])
        sb = torch.zeros(b1 - b0 + 64, dtype=torch.uint8, device=dev)     # the scan wants a 16-byte aligned blob:
        assert sb.data_ptr() % 16 == 0                                     # a shard's text is its own allocation
        sb[:b1 - b0] = blob[b0:b1]
        parts.append(sa.analyze_shard(sb[:b1 - b0], (offs[lo:hi + 1] - offs[lo]).contiguous(), src[lo:hi].contiguous()))
    tot = [sum(ints(p)[i] for p in parts) for i in range(7)]
    assert tuple(tot) == ints(whole)
    psum = 0.0
    for p in parts:
        psum += p.polarity_sum
    assert abs(psum - ref.polarity_sum) <= bound
    rep = oi.SpeculationEngine.aggregate_counters(oi.Ticker.parse("AAPL"), whole, None, None, oi.EngineConfig())
    assert rep.social.total_mentions == n and rep.social.bullish == ref.bullish


def test_full_size_lexicon_10M_posts(ctx, O):
    """10M synthetic posts in HBM (the size the lexicon figures are quoted on), through size-independent
    properties: the oracle on a slice, re-based sub-ranges (other tile boundaries, other 16-byte phase)
    reproducing the full scan bit for bit, and the summary counters against a device-side recount."""
    import torch
    import openintel_amd as oi
    from openintel_amd import synth
    dev = torch.device("cuda:0")
    n = 10_000_000
    blob, offs = synth.posts_torch(n, dev, seed=77)
    pol = torch.zeros(n, dtype=torch.float64, device=dev)
    spec = torch.zeros(n, dtype=torch.uint8, device=dev)
    an = oi.HipLexiconAnalyzer(ctx)
    an.analyze_device(blob, offs, pol, spec)
    ctx.synchronize()
    ns = 300_000
    rpol, rspec = O.lexicon_analyze(blob[: int(offs[ns])].cpu().numpy(), offs[: ns + 1].cpu().numpy().astype(np.uint64))
    assert np.array_equal(pol[:ns].cpu().numpy().view(np.uint64), rpol.view(np.uint64))
    assert np.array_equal(spec[:ns].cpu().numpy(), rspec)
    for start, count in ((2_345_678, 400_001), (9_600_001, 399_999), (3, 50_000)):
        b0, b1 = int(offs[start]), int(offs[start + count])
        pad = (16 - b0 % 16) % 16 + 32
        sub = torch.full((pad + (b1 - b0),), 32, dtype=torch.uint8, device=dev)   # spaces: no token
        sub[pad:] = blob[b0:b1]
        so = (offs[start:start + count + 1] - b0 + pad).contiguous()
        so[0] = 0
        sp = torch.zeros(count, dtype=torch.float64, device=dev)
        ss = torch.zeros(count, dtype=torch.uint8, device=dev)
        an.analyze_device(sub, so, sp, ss)
        ctx.synchronize()
        assert torch.equal(sp.view(torch.int64), pol[start:start + count].view(torch.int64))
        assert torch.equal(ss, spec[start:start + count])
    src = (torch.arange(n, device=dev) % 3 == 0).to(torch.uint8)
    cfg = oi.EngineConfig()
    cnt = oi.SpeculationEngine.social_counters(ctx, src, pol, spec, cfg)
    tau = cfg.bull_bear_threshold
    assert cnt.total == n and cnt.bullish == int((pol > tau).sum()) and cnt.bearish == int((pol < -tau).sum())
    assert cnt.neutral == n - cnt.bullish - cnt.bearish and cnt.spec_count == int((spec != 0).sum())
    assert list(cnt.by_source) == [n - int(src.sum()), int(src.sum())]
    assert abs(cnt.polarity_sum - float(pol.sum())) <= n * 2.0 ** -52 * n   # two different summation trees


# ----------------------------------------------------------------------------- retrieval helpers
def _check_cos_list(scores, docs, count, ref_dense, depth, doc_base=0):
    n = ref_dense.size
    assert count == min(depth, n)
    s, d = scores[:count], docs[:count].astype(np.int64) - doc_base
    assert np.all(d >= 0) and np.all(d < n) and np.unique(d).size == count
    order_ok = (s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (d[:-1] < d[1:]))
    assert order_ok.all(), "list not sorted by (score desc, doc asc)"
    assert np.abs(s.astype(np.float64) - ref_dense[d].astype(np.float64)).max() <= COS_TOL   # the 1e-5 bar
    kth = np.sort(ref_dense)[::-1][count - 1]
    must = np.nonzero(ref_dense > kth + 2 * COS_TOL)[0]
    assert np.isin(must, d).all(), "a clearly better doc is missing"
    assert (ref_dense[d] >= kth - 2 * COS_TOL).all(), "a clearly worse doc is present"


def _build(ctx, rows, terms, offs, vocab, normalize=False, doc_base=0):
    import openintel_amd as oi
    idx = oi.HybridIndex(ctx, rows.shape[0], rows.shape[1], vocab, doc_base)
    idx.set_embeddings(rows, normalize=normalize)
    idx.set_forward(terms, offs)
    idx.finalize()
    return idx


def _small_forward(rng, n, vocab, max_len=12, zipf=True):
    lens = rng.integers(1, max_len + 1, size=n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    if zipf:
        p = 1.0 / np.arange(1, vocab + 1) ** 1.07
        terms = rng.choice(vocab, size=int(offs[-1]), p=p / p.sum()).astype(np.uint32)
    else:
        terms = rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32)
    return terms, offs


@pytest.mark.parametrize("B,dim,n", [(1, 768, 5000), (3, 384, 3000), (8, 128, 9000), (20, 128, 9000),
                                     (64, 768, 4000), (70, 64, 6000), (1, 1024, 2000), (33, 100, 1000)])
def test_cosine_lists_within_tolerance(ctx, O, B, dim, n):
    from openintel_amd import synth
    rows = synth.embeddings_np(n, dim, seed=1 + B)
    q = synth.embeddings_np(B, dim, seed=99 + B)
    rng = np.random.default_rng(B)
    terms, offs = _small_forward(rng, n, 50)
    idx = _build(ctx, rows, terms, offs, 50)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    for depth in (10, 1000):
        L = idx.search_lists(q, qt, qo, depth=depth)
        for b in range(B):
            _check_cos_list(L.cos_scores[b], L.cos_docs[b], int(L.cos_counts[b]), O.dot_scores(rows, q[b]), depth)
    idx.close()


def test_cosine_normalize_on_device_matches_oracle(ctx, O):
    rng = np.random.default_rng(3)
    raw = (rng.standard_normal((3000, 96)) * rng.uniform(0.1, 30, size=(3000, 1))).astype(np.float32)
    raw[7] = 0.0  # zero row stays zero
    terms, offs = _small_forward(rng, 3000, 20)
    idx = _build(ctx, raw, terms, offs, 20, normalize=True)
    ref_rows = O.l2_normalize_rows(raw)
    q = O.l2_normalize_rows(rng.standard_normal((2, 96)).astype(np.float32))
    L = idx.search_lists(q, np.zeros(2, np.uint32), np.arange(3, dtype=np.uint32), depth=50)
    for b in range(2):
        _check_cos_list(L.cos_scores[b], L.cos_docs[b], int(L.cos_counts[b]), O.dot_scores(ref_rows, q[b]), 50)
    idx.close()


def _exact_case(rng, n, dim, vocab, B, n_q_terms=4, max_len=12):
    # small-integer embeddings: every dot product is an exact integer in f32 whatever the
    # summation order, so even the cosine list (with its many ties) must match bit for bit
    rows = rng.integers(-3, 4, size=(n, dim)).astype(np.float32)
    q = rng.integers(-3, 4, size=(B, dim)).astype(np.float32)
    terms, offs = _small_forward(rng, n, vocab, max_len)
    qt = rng.integers(0, min(vocab, 12), size=B * n_q_terms).astype(np.uint32)
    qo = (np.arange(B + 1) * n_q_terms).astype(np.uint32)
    return rows, q, terms, offs, qt, qo


@pytest.mark.parametrize("n,dim,vocab,B,depth,k", [
    (70_000, 32, 40, 4, 100, 100),      # 3 doc blocks, dense BM25 path (frequent terms), many ties
    (70_000, 32, 5000, 9, 1000, 100),   # sparse BM25 path, MFMA cosine kernel
    (40_000, 64, 300, 64, 10, 10),      # batch 64
    (1_000, 384, 64, 1, 10, 10),        # BASELINE configs[0] shape
    (33_000, 16, 8, 2, 1024, 1024),     # maximum depth / k
    (600_000, 8, 30, 3, 100, 50),       # 19 doc blocks: two-phase BM25 with a threshold, heavy terms
    (600_000, 8, 3000, 3, 1000, 100),   # same, sparse terms; several cosine chunks
])
def test_hybrid_pipeline_bit_exact(ctx, O, n, dim, vocab, B, depth, k):
    rng = np.random.default_rng(n + B)
    rows, q, terms, offs, qt, qo = _exact_case(rng, n, dim, vocab, B)
    idx = _build(ctx, rows, terms, offs, vocab, normalize=False, doc_base=1000)
    L = idx.search_lists(q, qt, qo, depth=depth)
    R = idx.search(q, qt, qo, k=k, depth=depth)
    for b in range(B):
        cs, cd = O.topk(O.dot_scores(rows, q[b]), depth, False, 1000)
        bs, bd = O.topk(O.bm25_scores(terms, offs, vocab, qt[qo[b]:qo[b + 1]]), depth, True, 1000)
        fs, fd = O.rrf_fuse(cd, bd, k)
        assert int(L.cos_counts[b]) == cd.size and int(L.bm25_counts[b]) == bd.size
        assert np.array_equal(L.cos_docs[b][:cd.size], cd) and np.array_equal(L.cos_scores[b][:cd.size], cs)
        assert np.array_equal(L.bm25_docs[b][:bd.size], bd), "BM25 doc order differs (query %d)" % b
        assert np.array_equal(L.bm25_scores[b][:bd.size].view(np.uint32), bs.view(np.uint32)), "BM25 score bits"
        assert int(R.counts[b]) == fd.size
        assert np.array_equal(R.docs[b][:fd.size], fd) and np.array_equal(
            R.scores[b][:fd.size].view(np.uint32), fs.view(np.uint32))
    idx.close()


def test_bm25_edge_cases(ctx, O):
    rng = np.random.default_rng(11)
    n, vocab = 5000, 30
    rows = rng.integers(-2, 3, size=(n, 8)).astype(np.float32)
    terms, offs = _small_forward(rng, n, vocab - 2)   # terms 28, 29 never occur
    idx = _build(ctx, rows, terms, offs, vocab)
    queries = [[28], [29, 28], [], [3, 3, 3], [0], [5, 28, 7], [999999]]   # absent / empty / repeated / out-of-vocab
    from openintel_amd import pack_query_terms
    qt, qo = pack_query_terms(queries)
    q = rng.integers(-2, 3, size=(len(queries), 8)).astype(np.float32)
    L = idx.search_lists(q, qt, qo, depth=20)
    for b, terms_b in enumerate(queries):
        tb = [t for t in terms_b if t < vocab]
        bs, bd = O.topk(O.bm25_scores(terms, offs, vocab, np.array(tb, np.uint32)), 20, True)
        assert int(L.bm25_counts[b]) == bd.size
        assert np.array_equal(L.bm25_docs[b][:bd.size], bd)
        assert np.array_equal(L.bm25_scores[b][:bd.size].view(np.uint32), bs.view(np.uint32))
    idx.close()


def test_bm25_stream_starts_from_the_terms_impact_floors(ctx, O):
    """Round 4: the stream kernel has no threshold-less phase -- a query starts at max over its terms of idf * (a lower bound of
    the term's r-th largest impact), r = 16 / 64 / 256 / 1024 >= depth (bm25.hip: bm_term_floor_kernel).  A bound that is too high
    drops docs of the list: depths on both sides of every rank, terms whose df sits on both sides of every rank (a Zipf
    vocabulary over several doc blocks), rare-only queries (no bound at all), repeated terms, a nonzero doc base -- every list
    bit-identical to the oracle's and to the wave kernel's."""
    from openintel_amd import pack_query_terms
    rng = np.random.default_rng(404)
    n, vocab = 70_000, 3000
    rows = rng.integers(-2, 3, size=(n, 8)).astype(np.float32)
    terms, offs = _small_forward(rng, n, vocab, max_len=10, zipf=True)
    df = np.bincount(terms, minlength=vocab)
    order = np.argsort(-df)
    pick = lambda lo, hi: [int(t) for t in order if lo <= df[t] < hi][:3]
    frequent, mid, low, rare = pick(2000, 10**9), pick(300, 1000), pick(70, 250), pick(1, 15)
    assert frequent and mid and low and rare, (df.max(), df.min())
    queries = [frequent[:1], frequent[:2] + mid[:1], mid[:2], low[:2], rare[:2], rare[:1] + frequent[:1], [frequent[0]] * 3,
               low[:1] + rare[:1], mid + low + rare + frequent, [], [int(order[-1])]]
    qt, qo = pack_query_terms(queries)
    q = rng.integers(-2, 3, size=(len(queries), 8)).astype(np.float32)
    idx = _build(ctx, rows, terms, offs, vocab, doc_base=1000)
    idx.set_max_query_terms(16)
    full = [O.bm25_scores(terms, offs, vocab, np.array(tq, np.uint32)) for tq in queries]
    for depth in (1, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 1024):
        lists = {}
        for mode in (idx.BM25_STREAM, idx.BM25_WAVE):
            idx.set_bm25_mode(mode)
            lists[mode] = idx.search_lists(q, qt, qo, depth=depth)
        L, W = lists[idx.BM25_STREAM], lists[idx.BM25_WAVE]
        for b in range(len(queries)):
            bs, bd = O.topk(full[b], depth, True, 1000)
            assert int(L.bm25_counts[b]) == bd.size == int(W.bm25_counts[b]), (depth, b)
            assert np.array_equal(L.bm25_docs[b][:bd.size], bd), (depth, b)
            assert np.array_equal(L.bm25_scores[b][:bd.size].view(np.uint32), bs.view(np.uint32)), (depth, b)
            assert np.array_equal(W.bm25_docs[b][:bd.size], bd), (depth, b)
    idx.close()


@pytest.mark.parametrize("n,vocab,B,max_terms,depth", [
    (70_000, 40, 33, 8, 100),     # heavy terms: most docs match several batch terms (> 6 distinct -> exact slow path)
    (70_000, 5000, 70, 8, 1000),  # sparse terms, two passes of the batch (1024/8 = 128 ... B=70 fits one; see next)
    (20_000, 300, 40, 64, 50),    # max_terms 64 -> 16 queries per pass -> three passes
    (600_000, 2000, 24, 6, 100),  # several doc chunks with thresholds
])
def test_bm25_batch_scan_bit_exact(ctx, O, n, vocab, B, max_terms, depth):
    """The forward-index scan (bm25_scan.hip, selected per index); same bits as the oracle and the TAAT kernel.
    Queries have 0..max_terms terms, with repeats and out-of-vocabulary ids."""
    import openintel_amd as oi
    rng = np.random.default_rng(n + B)
    rows = rng.integers(-2, 3, size=(n, 8)).astype(np.float32)
    terms, offs = _small_forward(rng, n, vocab, max_len=14)
    idx = oi.HybridIndex(ctx, n, 8, vocab, doc_id_base=77)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    idx.finalize()
    idx.set_max_query_terms(max_terms)
    idx.set_bm25_mode(idx.BM25_SCAN)
    queries = []
    for b in range(B):
        k = int(rng.integers(0, min(max_terms, 8) + 1))
        t = rng.integers(0, min(vocab, 40), size=k).tolist()
        if k >= 2 and b % 3 == 0:
            t[1] = t[0]                    # repeated term counts twice
        if k >= 1 and b % 5 == 0:
            t[-1] = vocab + 3              # outside the vocabulary: contributes nothing
        queries.append(t)
    qt, qo = oi.pack_query_terms(queries)
    q = rng.integers(-2, 3, size=(B, 8)).astype(np.float32)
    L = idx.search_lists(q, qt, qo, depth=depth)
    for other in (idx.BM25_TAAT, idx.BM25_WAVE, idx.BM25_STREAM):     # the four kernels agree bit for bit
        idx.set_bm25_mode(other)
        L2 = idx.search_lists(q, qt, qo, depth=depth)
        assert np.array_equal(L.bm25_docs, L2.bm25_docs) and np.array_equal(L.bm25_scores, L2.bm25_scores)
        assert np.array_equal(L.bm25_counts, L2.bm25_counts)
    for b, tb in enumerate(queries):
        tv = np.array([t for t in tb if t < vocab], np.uint32)
        bs, bd = O.topk(O.bm25_scores(terms, offs, vocab, tv), depth, True, 77)
        assert int(L.bm25_counts[b]) == bd.size, (b, tb)
        assert np.array_equal(L.bm25_docs[b][:bd.size], bd), (b, tb)
        assert np.array_equal(L.bm25_scores[b][:bd.size].view(np.uint32), bs.view(np.uint32)), (b, tb)
    idx.close()


@pytest.mark.parametrize("case", ["every-doc-term", "long-queries", "clustered"])
def test_bm25_wave_kernel_windows_and_long_queries(ctx, O, case):
    """bm25_wave.hip and bm25_stream.hip off their happy paths: a term in EVERY doc (32768 postings per block: the wave
    kernel's table is cut into doc-id windows; the stream kernel's 4096-key segment is pruned in place again and again, and
    a window holds far more than 512 multi docs: its extra rank rounds), queries of 70..200 terms (more than the 64 runs a
    wave's lanes describe: pages; more than 2048 staged terms per pass), and docs clustered at the start of a block.  Same
    bits as the oracle and as the workgroup-per-block kernel."""
    import openintel_amd as oi
    rng = np.random.default_rng(len(case))
    n, vocab, depth = 100_000, 400, 300
    lens = rng.integers(1, 12, size=n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    terms = rng.integers(2, vocab, size=int(offs[-1])).astype(np.uint32)
    if case == "every-doc-term":
        terms[offs[:-1].astype(np.int64)] = 0                       # term 0 in every doc
        terms[offs[:-1].astype(np.int64)[lens > 1] + 1] = 1          # term 1 in most docs
        queries = [[0], [0, 1], [1, 0, 5], [0, 0], [7, 0, 9, 1]] + [rng.integers(0, 40, size=4).tolist() for _ in range(11)]
    elif case == "long-queries":
        queries = [rng.integers(0, vocab + 5, size=int(k)).tolist() for k in (70, 128, 200, 65, 64, 1, 0, 150)] + \
                  [rng.integers(0, vocab, size=150).tolist() for _ in range(12)]   # 20 queries, > 2048 terms in the pass
    else:
        first = offs[:-1].astype(np.int64)
        sel = first[(np.arange(n) % 32768) < 3000]                   # term 3 only in the first 3000 docs of each block
        terms[terms == 3] = 4
        terms[sel] = 3
        queries = [[3], [3, 4], [4, 3, 3], [3, 10, 11, 12]] + [rng.integers(2, 30, size=5).tolist() for _ in range(12)]
    rows = rng.integers(-2, 3, size=(n, 8)).astype(np.float32)
    idx = oi.HybridIndex(ctx, n, 8, vocab, doc_id_base=5)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    idx.finalize()
    qt, qo = oi.pack_query_terms(queries)
    q = rng.integers(-2, 3, size=(len(queries), 8)).astype(np.float32)
    idx.set_bm25_mode(idx.BM25_WAVE)
    L = idx.search_lists(q, qt, qo, depth=depth)
    for other in (idx.BM25_TAAT, idx.BM25_STREAM):
        idx.set_bm25_mode(other)
        L2 = idx.search_lists(q, qt, qo, depth=depth)
        assert np.array_equal(L.bm25_counts, L2.bm25_counts), other
        assert np.array_equal(L.bm25_docs, L2.bm25_docs) and np.array_equal(L.bm25_scores, L2.bm25_scores), other
    for b, tb in enumerate(queries):
        tv = np.array([t for t in tb if t < vocab], np.uint32)
        bs, bd = O.topk(O.bm25_scores(terms, offs, vocab, tv), depth, True, 5)
        assert int(L.bm25_counts[b]) == bd.size, (b, len(tb))
        assert np.array_equal(L.bm25_docs[b][:bd.size], bd), (b, len(tb))
        assert np.array_equal(L.bm25_scores[b][:bd.size].view(np.uint32), bs.view(np.uint32)), (b, len(tb))
    idx.close()


def test_merge_and_rrf_kernels_bit_exact(ctx, O):
    from openintel_amd import merge_lists, rrf_fuse
    rng = np.random.default_rng(2)
    S, B, depth = 8, 5, 100
    scores = np.zeros((S, B, depth), np.float32); docs = np.zeros((S, B, depth), np.uint32)
    counts = rng.integers(0, depth + 1, size=(S, B)).astype(np.uint32)
    counts[0, 0] = depth; counts[1, 1] = 0
    for s in range(S):
        for b in range(B):
            c = counts[s, b]
            sc = np.round(rng.standard_normal(c), 1).astype(np.float32)       # lots of ties
            dd = rng.choice(100000, size=c, replace=False).astype(np.uint32) * S + s   # disjoint across shards
            o = np.lexsort((dd, -sc.astype(np.float64)))
            scores[s, b, :c], docs[s, b, :c] = sc[o], dd[o]
    so, do, co = merge_lists(ctx, scores, docs, counts)
    for b in range(B):
        ms, md = O.merge_ranked([scores[s, b, :counts[s, b]] for s in range(S)],
                                [docs[s, b, :counts[s, b]] for s in range(S)], depth)
        assert co[b] == md.size and np.array_equal(do[b][:md.size], md) and np.array_equal(so[b][:md.size], ms)
    for depth, k in ((100, 100), (1000, 100), (1024, 1024), (7, 3)):
        da = np.stack([rng.choice(5000, size=depth, replace=False) for _ in range(B)]).astype(np.uint32)
        db = np.stack([rng.choice(5000, size=depth, replace=False) for _ in range(B)]).astype(np.uint32)
        ca = rng.integers(0, depth + 1, size=B).astype(np.uint32); cb = rng.integers(0, depth + 1, size=B).astype(np.uint32)
        ca[0], cb[0] = depth, depth
        if B > 1:
            ca[1], cb[1] = 0, 0
        R = rrf_fuse(ctx, da, ca, db, cb, k)
        for b in range(B):
            fs, fd = O.rrf_fuse(da[b][:ca[b]], db[b][:cb[b]], k)
            assert R.counts[b] == fd.size and np.array_equal(R.docs[b][:fd.size], fd)
            assert np.array_equal(R.scores[b][:fd.size].view(np.uint32), fs.view(np.uint32))


@pytest.mark.parametrize("S,depth,mode", [
    (4, 1000, "normal"), (8, 1000, "normal"), (8, 1000, "ties"), (16, 1000, "normal"), (32, 1000, "normal"),
    (32, 1024, "flat"), (40, 1000, "normal"), (40, 1000, "ties"), (1024, 64, "normal"), (1024, 64, "sparse"),
    (300, 10, "sparse"), (64, 512, "skewed"), (9, 1, "normal")])
def test_select_pool_sizes_bit_exact(ctx, O, S, depth, mode):
    """The pool selection behind every ranked list (select.hip), driven through oi_merge_lists so that the pool
    is S segments of up to `depth` keys: every size class of the flat kernel (registers 4/8/16/32 keys per
    thread, the re-loading path above 32K keys), its sampled cut and that cut's fallbacks -- score ties that
    reach into the doc-id bits, one constant score, lists from shards of very different quality, empty and
    ragged segments."""
    from openintel_amd import merge_lists
    rng = np.random.default_rng(S * 1000 + depth)
    B = 3
    scores = np.zeros((S, B, depth), np.float32); docs = np.zeros((S, B, depth), np.uint32)
    if mode == "sparse":
        counts = (rng.integers(0, depth + 1, size=(S, B)) * (rng.random((S, B)) < 0.1)).astype(np.uint32)
    else:
        counts = rng.integers(depth // 2, depth + 1, size=(S, B)).astype(np.uint32)
        counts[0, 0] = depth
        if S > 1:
            counts[1, 1] = 0
    for s in range(S):
        for b in range(B):
            c = int(counts[s, b])
            if mode == "ties":
                sc = np.round(rng.standard_normal(c), 1).astype(np.float32)
            elif mode == "flat":
                sc = np.full(c, 0.25, np.float32)
            elif mode == "skewed":   # shard s's scores sit around s: the top of the pool comes from a few segments
                sc = (rng.standard_normal(c) * 0.01 + s).astype(np.float32)
            else:
                sc = (rng.standard_normal(c) * 0.05).astype(np.float32)
            dd = rng.choice(200000, size=c, replace=False).astype(np.uint32) * S + s   # disjoint across shards
            o = np.lexsort((dd, -sc.astype(np.float64)))
            scores[s, b, :c], docs[s, b, :c] = sc[o], dd[o]
    so, do, co = merge_lists(ctx, scores, docs, counts)
    for b in range(B):
        ms, md = O.merge_ranked([scores[s, b, :counts[s, b]] for s in range(S)],
                                [docs[s, b, :counts[s, b]] for s in range(S)], depth)
        assert co[b] == md.size
        assert np.array_equal(do[b][:md.size], md)
        assert np.array_equal(so[b][:md.size].view(np.uint32), ms.view(np.uint32))


def test_errors_are_loud(ctx):
    import openintel_amd as oi
    from openintel_amd._lib import OiError
    with pytest.raises(OiError):
        oi.HybridIndex(ctx, 10, 7, 10)            # dim not a multiple of 4
    idx = oi.HybridIndex(ctx, 10, 8, 10)
    idx.set_embeddings(np.zeros((10, 8), np.float32), normalize=False)
    with pytest.raises(OiError):                   # BM25 index never built
        idx.search_lists(np.zeros((1, 8), np.float32), np.zeros(1, np.uint32), np.array([0, 1], np.uint32), depth=5)
    with pytest.raises(OiError):
        idx.set_forward(np.array([11], np.uint32), np.array([0] + [1] * 10, np.uint64))   # term >= vocab
    idx.close()


# ----------------------------------------------------------------------------- full BASELINE sizes
def _planted_check(ctx, n, dim, B, depth, k):
    """Size-independent properties at full size: planted near-duplicates of the queries must come
    back first with score ~1; lists are sorted; results are reproducible; sharding the same corpus
    in two and merging gives the same answer as one shard."""
    import torch
    import openintel_amd as oi
    from openintel_amd import synth
    dev = torch.device("cuda:0")
    rows = synth.embeddings_torch(n, dim, dev)
    qv, qt, qo = synth.query_batch_torch(B, dim, dev, vocab=4096)
    plant = torch.arange(B, device=dev) * (n // B) + 17
    rows[plant] = qv
    terms, offs = synth.forward_index_torch(n, dev, vocab=4096)
    idx = oi.HybridIndex(ctx, n, dim, 4096)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    idx.finalize()
    L = idx.search_lists(qv, qt, qo, depth=depth)
    ctx.synchronize()
    cs, cd, cc = L.cos_scores.cpu().numpy(), L.cos_docs.cpu().numpy(), L.cos_counts.cpu().numpy()
    assert (cc == depth).all()
    assert np.array_equal(cd[:, 0], plant.cpu().numpy()) and np.abs(cs[:, 0] - 1.0).max() < 1e-5
    assert (np.diff(cs, axis=1) <= 0).all()
    bs, bc = L.bm25_scores.cpu().numpy(), L.bm25_counts.cpu().numpy()
    for b in range(B):
        assert (np.diff(bs[b, :bc[b]]) <= 0).all() and (bs[b, :bc[b]] > 0).all()
    R1 = idx.search(qv, qt, qo, k=k, depth=depth)
    R2 = idx.search(qv, qt, qo, k=k, depth=depth)
    ctx.synchronize()
    assert torch.equal(R1.docs, R2.docs) and torch.equal(R1.scores, R2.scores)     # idempotent
    # spot-check a few queries' cosine lists against exact torch scores of the candidates
    for b in range(min(B, 3)):
        full = (rows @ qv[b]).cpu().numpy()
        _check_cos_list(cs[b], cd[b], int(cc[b]), full, depth)
    idx.close()
    return rows, terms, offs, qv, qt, qo, (cs, cd, cc, bs, L.bm25_docs.cpu().numpy(), bc)


def test_full_size_config1_1M_768_batch1(ctx):
    _planted_check(ctx, 1_000_000, 768, 1, 100, 100)     # BASELINE.json configs[1]


def test_full_size_two_shards_equal_one(ctx, O):
    import torch
    import openintel_amd as oi
    n, dim, B, depth, k = 400_000, 128, 16, 100, 100
    rows, terms, offs, qv, qt, qo, one = _planted_check(ctx, n, dim, B, depth, k)
    cs, cd, cc, bs, bd, bc = one
    half = n // 2
    lists = []
    tot, dfs = 0, []
    shards = []
    for s, (lo, hi) in enumerate(((0, half), (half, n))):
        t_lo, t_hi = int(offs[lo]), int(offs[hi])
        ix = oi.HybridIndex(ctx, hi - lo, dim, 4096, doc_id_base=lo)
        ix.set_embeddings(rows[lo:hi], normalize=False)
        ix.set_forward(terms[t_lo:t_hi].contiguous(), (offs[lo:hi + 1] - offs[lo]).contiguous())
        t, df = ix.local_stats()
        tot += t; dfs.append(df); shards.append(ix)
    gdf = (dfs[0].astype(np.uint64) + dfs[1]).astype(np.uint32)
    for ix in shards:
        ix.finalize(n, tot, gdf)          # global N / tokens / df: the all-reduce a multi-GPU caller does
        lists.append(ix.search_lists(qv, qt, qo, depth=depth))
    ctx.synchronize()
    st = lambda f: torch.stack([getattr(l, f) for l in lists])
    ms, md, mc = oi.merge_lists(ctx, st("cos_scores"), st("cos_docs"), st("cos_counts"))
    bs2, bd2, bc2 = oi.merge_lists(ctx, st("bm25_scores"), st("bm25_docs"), st("bm25_counts"))
    ctx.synchronize()
    assert np.array_equal(md.cpu().numpy(), cd) and np.array_equal(ms.cpu().numpy(), cs)
    # the packed exchange path (what the sharded retriever all-gathers) gives the same fused answer
    packed = torch.cat([ix.search_lists_packed(qv, qt, qo, depth=depth) for ix in shards])
    Rp = oi.fuse_packed(ctx, packed, 2, B, depth, k)
    Rl = oi.rrf_fuse(ctx, md, mc, bd2, bc2, k)
    ctx.synchronize()
    assert torch.equal(Rp.docs, Rl.docs) and torch.equal(Rp.scores, Rl.scores) and torch.equal(Rp.counts, Rl.counts)
    assert np.array_equal(bc2.cpu().numpy(), bc)
    for b in range(B):   # BM25 with global statistics is bit-identical to the unsharded index
        assert np.array_equal(bd2.cpu().numpy()[b, :bc[b]], bd[b, :bc[b]])
        assert np.array_equal(bs2.cpu().numpy()[b, :bc[b]], bs[b, :bc[b]])
    for ix in shards:
        ix.close()


def test_full_size_config2_10M_768_batch64(ctx):
    _planted_check(ctx, 10_000_000, 768, 64, 100, 100)   # BASELINE.json configs[2]


def test_full_size_config2_composition_against_the_oracle(ctx, O):
    """VERDICT r03 missing #3, r04 next #5: the bench's own workload end to end -- 10M x 768-d, 64 queries x 4 terms, vocab
    131072, depth 1000, k 100 -- in EVERY cosine mode the bench line reports: the default (bf16 screen over the index's
    screening copy + exact f32 rescoring: the headline's scorer), the same screen over the f32 rows (rounds 1-4's headline) and
    the exact f32-MFMA kernel.  Per mode: (a) oi_search's fused output equals O.rrf_fuse of the GPU's two lists for ALL 64
    queries bit for bit (the lists themselves are held against the oracle piece by piece: BM25 at this shape in the next test,
    cosine below); (b) the 10M-row cosine lists of two queries against O.dot_scores -- the oracle's f64 dot products, streamed
    over the corpus in row chunks -- at the 1e-5 bar, not against torch's matmul.  And the two screen modes return the same
    lists bit for bit."""
    import torch
    import openintel_amd as oi
    from openintel_amd import _lib, synth
    dev = torch.device("cuda:0")
    n, dim, vocab, B, depth, k = 10_000_000, 768, 131072, 64, 1000, 100
    rows = synth.embeddings_torch(n, dim, dev)
    qv, qt, qo = synth.query_batch_torch(B, dim, dev, vocab=vocab)
    terms, offs = synth.forward_index_torch(n, dev, vocab=vocab)
    idx = oi.HybridIndex(ctx, n, dim, vocab)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    idx.set_max_query_terms(4)
    idx.finalize()
    del terms, offs
    assert idx.index_bytes()[1] >= 2 * n * dim          # the screening copy exists: the default mode streams it
    h_q = qv.cpu().numpy()
    full = {}
    for b in (0, B - 1):                   # 10M f64 dot products per query on the host, 500K rows at a time (once, for all modes)
        f = np.empty(n, dtype=np.float32)
        step = 500_000
        for r0 in range(0, n, step):
            f[r0:r0 + step] = O.dot_scores(rows[r0:r0 + step].cpu().numpy(), h_q[b])
        full[b] = f
    lists = {}
    try:
        for mode_name, mode in (("screen (screening copy: the default)", _lib.OI_COSINE_SCREEN), ("screen-stream", _lib.OI_COSINE_SCREEN_STREAM),
                                ("exact", _lib.OI_COSINE_EXACT)):
            ctx.set_cosine_mode(mode)
            L = idx.search_lists(qv, qt, qo, depth=depth)
            R = idx.search(qv, qt, qo, k=k, depth=depth)
            ctx.synchronize()
            cs, cd, cc = L.cos_scores.cpu().numpy(), L.cos_docs.cpu().numpy(), L.cos_counts.cpu().numpy()
            bd, bc = L.bm25_docs.cpu().numpy(), L.bm25_counts.cpu().numpy()
            rs, rd, rc = R.scores.cpu().numpy(), R.docs.cpu().numpy(), R.counts.cpu().numpy()
            assert (cc == depth).all() and (bc == depth).all(), mode_name
            for b in range(B):             # (a) fusion of the two lists: integer ranks, f32 reciprocal sums, ties by doc id
                fs, fd = O.rrf_fuse(cd[b, :cc[b]], bd[b, :bc[b]], k)
                assert int(rc[b]) == fd.size == k, (mode_name, b, int(rc[b]), fd.size)
                assert np.array_equal(rd[b, :k], fd), "%s: fused doc order differs from the oracle's fusion of the same lists (query %d)" % (mode_name, b)
                assert np.array_equal(rs[b, :k].view(np.uint32), fs.view(np.uint32)), "%s: fused score bits differ (query %d)" % (mode_name, b)
            for b in (0, B - 1):           # (b) the cosine list against the oracle's f64 dot products
                _check_cos_list(cs[b], cd[b], int(cc[b]), full[b], depth)
            lists[mode] = (cs, cd, cc)
    finally:
        ctx.set_cosine_mode(_lib.OI_COSINE_EXACT)   # (the module's mode)
    a, b2 = lists[_lib.OI_COSINE_SCREEN], lists[_lib.OI_COSINE_SCREEN_STREAM]
    assert np.array_equal(a[2], b2[2]) and np.array_equal(a[1], b2[1]) and np.array_equal(a[0].view(np.uint32), b2[0].view(np.uint32))
    idx.close()


def test_finalize_refuses_a_df_above_the_collection_size(ctx):
    """the stream kernel's first threshold (per-term impact floors) is a valid lower bound only while every idf is
    >= 0, i.e. df_t <= N.  A caller's global df vector that says otherwise is refused loudly."""
    import openintel_amd as oi
    from openintel_amd._lib import OiError
    rng = np.random.default_rng(3)
    n, vocab = 500, 20
    lens = rng.integers(1, 6, size=n)
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    terms = rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32)
    idx = oi.HybridIndex(ctx, n, 8, vocab)
    idx.set_embeddings(np.zeros((n, 8), np.float32), normalize=False)
    idx.set_forward(terms, offs)
    tot, df = idx.local_stats()
    bad = df.copy()
    bad[3] = n + 1
    with pytest.raises(OiError):
        idx.finalize(n, tot, bad)
    idx.finalize(n, tot, df)               # the index is still usable: nothing was finalized by the refused call
    idx.close()


def test_full_size_bm25_bench_shape_against_the_oracle(ctx, O):
    """VERDICT r02 missing #4: the bench's BM25 workload -- 10M docs, vocab 131072, Zipf postings, 4-term queries, depth 1000 --
    checked against the ORACLE (not kernel vs kernel): three queries' lists bit for bit (docs, order, score bits), all four
    BM25 kernels equal on the whole batch.  Oracle: one scalar pass over the 288M-token forward index per query."""
    import torch
    import openintel_amd as oi
    from openintel_amd import synth
    dev = torch.device("cuda:0")
    n, vocab, B, depth = 10_000_000, 131072, 64, 1000
    terms, offs = synth.forward_index_torch(n, dev, vocab=vocab)
    _, qt, qo = synth.query_batch_torch(B, 8, dev, vocab=vocab)
    idx = oi.HybridIndex(ctx, n, 8, vocab)
    idx.set_embeddings(torch.zeros((n, 8), dtype=torch.float32, device=dev), normalize=False)   # the cosine leg is not under test
    idx.set_forward(terms, offs)
    idx.set_max_query_terms(4)
    idx.finalize()
    qv = torch.zeros((B, 8), dtype=torch.float32, device=dev)
    lists = {}
    for name, mode in (("stream", idx.BM25_STREAM), ("wave", idx.BM25_WAVE), ("taat", idx.BM25_TAAT), ("scan", idx.BM25_SCAN)):
        idx.set_bm25_mode(mode)
        L = idx.search_lists(qv, qt, qo, depth=depth)
        ctx.synchronize()
        lists[name] = (L.bm25_scores.cpu().numpy(), L.bm25_docs.cpu().numpy(), L.bm25_counts.cpu().numpy())
    ws, wd, wc = lists["stream"]   # the default kernel's lists are the ones held against the oracle
    for other in ("wave", "taat", "scan"):
        s2, d2, c2 = lists[other]
        assert np.array_equal(wc, c2)
        for b in range(B):
            assert np.array_equal(wd[b, :wc[b]], d2[b, :wc[b]]) and np.array_equal(ws[b, :wc[b]].view(np.uint32), s2[b, :wc[b]].view(np.uint32)), (other, b)
    h_terms, h_offs = terms.cpu().numpy().view(np.uint32), offs.cpu().numpy().view(np.uint64)
    h_qt, h_qo = qt.cpu().numpy().view(np.uint32), qo.cpu().numpy().view(np.uint32)
    df, _ = O.bm25_df(h_terms, h_offs, vocab)
    heavy = int(np.argmax([int(df[h_qt[h_qo[b]:h_qo[b + 1]]].sum()) for b in range(B)]))   # the query with the most postings
    for b in sorted({0, B - 1, heavy}):
        bs, bd = O.topk(O.bm25_scores(h_terms, h_offs, vocab, h_qt[h_qo[b]:h_qo[b + 1]], df=df), depth, True)
        assert int(wc[b]) == bd.size == depth, (b, int(wc[b]), bd.size)
        assert np.array_equal(wd[b, :bd.size], bd), "BM25 doc order differs from the oracle (query %d)" % b
        assert np.array_equal(ws[b, :bd.size].view(np.uint32), bs.view(np.uint32)), "BM25 score bits differ (query %d)" % b
    idx.close()


def test_full_size_config3_eight_shards_compose_to_the_single_index(ctx):
    """BASELINE configs[3] composed at FULL size on one GPU: the 10M x 768 corpus as EIGHT row shards of 1.25M (doc_id_base =
    first row, global N / token count / df handed to every finalize -- what the all-reduce delivers), every shard's packed
    lists concatenated in rank order (what the all-gather delivers) and fused with oi_fuse_packed -- against oi_search on ONE
    index over the same 10M rows.  Small-integer embeddings: every dot product is exact in any summation order, so the cosine
    lists, the BM25 lists (global statistics: the same f32 operations) and the fused top-100 must be equal bit for bit."""
    import torch
    import openintel_amd as oi
    from openintel_amd import synth
    dev = torch.device("cuda:0")
    n, dim, B, depth, k, S, vocab = 10_000_000, 768, 64, 1000, 100, 8, 131072
    g = torch.Generator(device=dev)
    g.manual_seed(83)
    rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
    for s0 in range(0, n, 1 << 20):      # small integers, chunk by chunk (randint's int64 scratch stays small)
        e0 = min(n, s0 + (1 << 20))
        rows[s0:e0] = torch.randint(-3, 4, (e0 - s0, dim), generator=g, device=dev).to(torch.float32)
    qv = torch.randint(-3, 4, (B, dim), generator=g, device=dev).to(torch.float32)
    _, qt, qo = synth.query_batch_torch(B, dim, dev, vocab=vocab)
    terms, offs = synth.forward_index_torch(n, dev, vocab=vocab)
    one = oi.HybridIndex(ctx, n, dim, vocab)
    one.set_embeddings(rows, normalize=False)
    one.set_forward(terms, offs)
    one.set_max_query_terms(4)
    one.finalize()
    R = one.search(qv, qt, qo, k=k, depth=depth)
    L = one.search_lists(qv, qt, qo, depth=depth)
    ctx.synchronize()
    want = (R.docs.clone(), R.scores.clone(), R.counts.clone())
    want_l = [t.clone() for t in (L.cos_docs, L.cos_scores, L.bm25_docs, L.bm25_scores, L.bm25_counts)]
    one.close()
    del one, R, L
    bounds = [(r * n // S, (r + 1) * n // S) for r in range(S)]
    shards, tot, gdf = [], 0, None
    for lo, hi in bounds:
        t_lo, t_hi = int(offs[lo]), int(offs[hi])
        ix = oi.HybridIndex(ctx, hi - lo, dim, vocab, doc_id_base=lo)
        ix.set_embeddings(rows[lo:hi], normalize=False)
        ix.set_forward(terms[t_lo:t_hi].contiguous(), (offs[lo:hi + 1] - offs[lo]).contiguous())
        ix.set_max_query_terms(4)
        t, df = ix.local_stats()
        tot += t
        gdf = df.astype(np.uint64) if gdf is None else gdf + df
        shards.append(ix)
    del rows
    for ix in shards:
        ix.finalize(n, tot, gdf.astype(np.uint32))
    packed = torch.cat([ix.search_lists_packed(qv, qt, qo, depth=depth) for ix in shards])
    F = oi.fuse_packed(ctx, packed, S, B, depth, k)
    lists = [ix.search_lists(qv, qt, qo, depth=depth) for ix in shards]
    st = lambda f: torch.stack([getattr(l, f) for l in lists])
    ms, md, mc = oi.merge_lists(ctx, st("cos_scores"), st("cos_docs"), st("cos_counts"))
    bs, bd, bc = oi.merge_lists(ctx, st("bm25_scores"), st("bm25_docs"), st("bm25_counts"))
    ctx.synchronize()
    assert torch.equal(F.docs, want[0]) and torch.equal(F.scores, want[1]) and torch.equal(F.counts, want[2])
    assert torch.equal(md, want_l[0]) and torch.equal(ms, want_l[1])
    assert torch.equal(bc, want_l[4])
    for b in range(B):
        c = int(bc[b])
        assert torch.equal(bd[b, :c], want_l[2][b, :c]) and torch.equal(bs[b, :c], want_l[3][b, :c])
    for ix in shards:
        ix.close()


def test_full_size_config3_shard_through_sharded_retriever(ctx, O):
    """BASELINE configs[3] as ONE GPU of the 8 sees it: a 1.25M-row x 768-d f32 shard with a nonzero doc_id_base,
    driven through ShardedRetriever (world 1: finalize's statistics exchange and the packed list format are the
    real code path, the all-gather degenerates to the local buffer).  The fused result must equal oi_search on the
    same index bit for bit; planted copies of the queries come back first; every id is a global one."""
    import torch
    import openintel_amd as oi
    from openintel_amd import sharded, synth
    dev = torch.device("cuda:0")
    n, dim, B, depth, k, base = 1_250_000, 768, 64, 1000, 100, 3_750_000
    rows = synth.embeddings_torch(n, dim, dev)
    qv, qt, qo = synth.query_batch_torch(B, dim, dev, vocab=131072)
    plant = torch.arange(B, device=dev) * (n // B) + 5
    rows[plant] = qv
    terms, offs = synth.forward_index_torch(n, dev, vocab=131072)
    idx = oi.HybridIndex(ctx, n, dim, 131072, doc_id_base=base)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    idx.set_max_query_terms(4)
    sr = sharded.make_hip_sharded(ctx, idx, dev)
    sr.finalize()
    s1, d1, c1 = sr.search(qv, qt, qo, k, depth)
    R = idx.search(qv, qt, qo, k=k, depth=depth)
    ctx.synchronize()
    assert torch.equal(d1, R.docs) and torch.equal(s1, R.scores) and torch.equal(c1, R.counts)
    d = d1.cpu().numpy().astype(np.int64)
    assert (c1.cpu().numpy() == k).all() and d.min() >= base and d.max() < base + n
    L = idx.search_lists(qv, qt, qo, depth=depth)
    ctx.synchronize()
    cd, cs = L.cos_docs.cpu().numpy().astype(np.int64), L.cos_scores.cpu().numpy()
    assert np.array_equal(cd[:, 0], plant.cpu().numpy() + base) and np.abs(cs[:, 0] - 1.0).max() < 1e-5
    for b in range(3):
        full = (rows @ qv[b]).cpu().numpy()
        _check_cos_list(cs[b], cd[b], depth, full, depth, doc_base=base)
    idx.close()


def test_sharded_pipeline_overlaps_fusion_and_returns_the_same_results(ctx, O):
    """sharded.ShardedPipeline (bench.py's throughput mode for N > 1): the fusion of batch i runs on a second stream and
    a second ctx while batch i+1's lists are being scored; two slots.  Six different batches through the pipeline must
    come out exactly as through the plain, one-at-a-time ShardedRetriever.search (world 1: the exchange degenerates)."""
    import torch
    import openintel_amd as oi
    from openintel_amd import sharded, synth
    dev = torch.device("cuda:0")
    n, dim, B, depth, k = 200_000, 384, 32, 300, 50
    rows = synth.embeddings_torch(n, dim, dev)
    terms, offs = synth.forward_index_torch(n, dev, vocab=4096)
    idx = oi.HybridIndex(ctx, n, dim, 4096, doc_id_base=1000)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    ctx.use_torch_current_stream()
    sr = sharded.make_hip_sharded(ctx, idx, dev)
    sr.finalize()
    batches = [synth.query_batch_torch(B, dim, dev, vocab=4096, seed=500 + i) for i in range(6)]
    want = []
    for qv, qt, qo in batches:
        s, d, c = sr.search(qv, qt, qo, k, depth)
        want.append((s.clone(), d.clone(), c.clone()))
    fctx = oi.HipContext(0)
    pipe = sharded.ShardedPipeline(sr, fctx, B, depth, k)
    got = []
    for qv, qt, qo in batches:
        slot = pipe.submit(qv, qt, qo)
        # a slot's result is valid until the slot is reused two submits later: copy it out behind the fusion (side stream)
        with torch.cuda.stream(pipe.side):
            r = pipe.results[slot]
            got.append((r.scores.clone(), r.docs.clone(), r.counts.clone()))
    pipe.drain()
    torch.cuda.synchronize()
    for (ws, wd, wc), (gs, gd, gc) in zip(want, got):
        assert torch.equal(wd, gd) and torch.equal(ws, gs) and torch.equal(wc, gc)
    fctx.close()
    idx.close()


def test_index_view_and_pipeline_lanes(ctx, O):
    """oi_index_view: a second handle on a finalized shard, bound to another context -- searched alone, it returns the
    source's results bit for bit; build calls on it are refused; two handles driven from two host threads agree with
    the serial results; ShardedPipeline with a second lane (consecutive batches scored through the index and its view at
    once, exchange + fusion on a third stream) returns what the one-at-a-time retriever returns."""
    import threading
    import torch
    import openintel_amd as oi
    from openintel_amd import _lib, sharded, synth
    dev = torch.device("cuda:0")
    n, dim, B, depth, k = 150_000, 768, 64, 400, 60
    rows = synth.embeddings_torch(n, dim, dev)
    terms, offs = synth.forward_index_torch(n, dev, vocab=4096)
    c0 = oi.HipContext(0)   # default scorer (bf16 screen + rescoring): the mode the lanes run in bench.py
    idx = oi.HybridIndex(c0, n, dim, 4096, doc_id_base=77)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    c0.use_torch_current_stream()
    c1 = oi.HipContext(0)
    with pytest.raises(_lib.OiError):
        idx.view(c1)                      # not finalized yet
    sr = sharded.make_hip_sharded(c0, idx, dev)
    sr.finalize()
    with pytest.raises(_lib.OiError):
        idx.view(c0)                      # a view needs a context of its own
    view = idx.view(c1)
    with pytest.raises(_lib.OiError):
        view.set_embeddings(rows, normalize=False)
    with pytest.raises(_lib.OiError):
        view.finalize()
    batches = [synth.query_batch_torch(B, dim, dev, vocab=4096, seed=900 + i) for i in range(8)]
    want = []
    for qv, qt, qo in batches:
        r = idx.search(qv, qt, qo, k=k, depth=depth)
        want.append((r.scores.clone(), r.docs.clone(), r.counts.clone()))
    torch.cuda.synchronize()
    for (qv, qt, qo), (ws, wd, wc) in zip(batches, want):       # the view alone
        r = view.search(qv, qt, qo, k=k, depth=depth)
        c1.synchronize()
        assert torch.equal(r.docs, wd) and torch.equal(r.scores, ws) and torch.equal(r.counts, wc)
    # two host threads, one handle each, at the same time
    got = [[None] * len(batches) for _ in range(2)]

    def worker(which, handle, hctx):
        st = torch.cuda.Stream(device=dev)
        hctx.set_stream(st)
        with torch.cuda.stream(st):
            for rep in range(3):
                for i, (qv, qt, qo) in enumerate(batches):
                    r = handle.search(qv, qt, qo, k=k, depth=depth)
                    got[which][i] = (r.scores.clone(), r.docs.clone(), r.counts.clone())
        hctx.synchronize()

    ts = [threading.Thread(target=worker, args=(0, idx, c0)), threading.Thread(target=worker, args=(1, view, c1))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    for which in range(2):
        for (ws, wd, wc), (gs, gd, gc) in zip(want, got[which]):
            assert torch.equal(wd, gd) and torch.equal(ws, gs) and torch.equal(wc, gc)
    view.close()
    c0.use_torch_current_stream()
    # the pipeline with a second lane
    fctx, lctx = oi.HipContext(0), oi.HipContext(0)
    pipe = sharded.ShardedPipeline(sr, fctx, B, depth, k, lane_ctxs=[lctx])
    assert pipe.n_slots == 6 and len(pipe.lanes) == 2
    outs = []
    for rep in range(2):
        for qv, qt, qo in batches:
            slot = pipe.submit(qv, qt, qo)
            with torch.cuda.stream(pipe.side):   # a slot is reused n_slots submits later: copy out behind the fusion
                r = pipe.results[slot]
                outs.append((r.scores.clone(), r.docs.clone(), r.counts.clone()))
    pipe.drain()
    torch.cuda.synchronize()
    for i, (gs, gd, gc) in enumerate(outs):
        ws, wd, wc = want[i % len(batches)]
        assert torch.equal(wd, gd) and torch.equal(ws, gs) and torch.equal(wc, gc), "batch %d" % i
    pipe.close()
    # the empirical choice between one lane and two (bench.py, N > 1): whatever it keeps, the results do not change
    pipe2 = sharded.ShardedPipeline(sr, fctx, B, depth, k)
    cal = pipe2.calibrate(batches, lambda: oi.HipContext(0), reps=6, placements=2)
    assert cal["chosen_lanes"] == len(pipe2.lanes) and cal["chosen_lanes"] in (1, 2) and len(cal["tried"]) == 3
    outs = []
    for qv, qt, qo in batches:
        slot = pipe2.submit(qv, qt, qo)
        with torch.cuda.stream(pipe2.side):
            r = pipe2.results[slot]
            outs.append((r.scores.clone(), r.docs.clone(), r.counts.clone()))
    pipe2.drain()
    torch.cuda.synchronize()
    for (gs, gd, gc), (ws, wd, wc) in zip(outs, want):
        assert torch.equal(wd, gd) and torch.equal(ws, gs) and torch.equal(wc, gc)
    pipe2.close()
    for c in (fctx, lctx, c1):
        c.close()
    idx.close()
    c0.close()


def test_fuzz_small_shapes_bit_exact(ctx, O):
    """Thirty random small configurations through the whole C-ABI path (oi_search_lists + oi_search), every BM25 kernel
    on each: odd corpus sizes around the 32768-doc block boundary, 1..70 queries, 0..70 terms per query with repeats and
    out-of-vocabulary ids (vocab 1 or 2: every doc holds the term and the query repeats it -- the wave kernel's dense
    windows), depth and k from 1 to 1024, the batch scorers' dims plus odd ones, a nonzero doc_id_base.  Small-integer
    embeddings: every dot product is exact in any order, so cosine, BM25 and fused lists must equal the oracle's bit
    for bit.  (Progress goes to gpurun_out/fuzz_progress.log when that directory exists.)"""
    import os
    import time
    import openintel_amd as oi
    rng = np.random.default_rng(20261004)
    log = None
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        log = open(os.path.join(out_dir, "fuzz_progress.log"), "w")
    for case in range(30):
        n = int(rng.choice([1, 2, 63, 64, 65, 1000, 32767, 32768, 32769, 40_000]))
        dim = int(rng.choice([4, 8, 100, 384, 768, 1024]))
        vocab = int(rng.choice([1, 2, 7, 50, 3000]))
        B = int(rng.choice([1, 2, 8, 9, 33, 64, 70]))
        depth = int(rng.choice([1, 2, 10, 100, 1000, 1024]))
        k = int(rng.choice([1, 3, 10, 100, 1024]))
        base = int(rng.choice([0, 5, 4_000_000_000 - 80_000]))
        if n * dim * B > 150_000_000:      # keep the oracle's share of the test in seconds
            dim = 8
        t0 = time.perf_counter()
        rows = rng.integers(-3, 4, size=(n, dim)).astype(np.float32)
        lens = rng.integers(0, 9, size=n)                          # empty docs allowed
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        terms = rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32)
        queries = []
        for b in range(B):
            nt = int(rng.choice([0, 1, 2, 4, 4, 4, 9, 70]))
            t = rng.integers(0, vocab + (2 if b % 4 == 0 else 0), size=nt).tolist()   # some ids >= vocab
            queries.append(t)
        qt, qo = oi.pack_query_terms(queries)
        q = rng.integers(-3, 4, size=(B, dim)).astype(np.float32)
        idx = oi.HybridIndex(ctx, n, dim, vocab, doc_id_base=base)
        idx.set_embeddings(rows, normalize=False)
        idx.set_forward(terms, offs)
        idx.finalize()
        idx.set_max_query_terms(128)
        ref = []
        for b in range(B):
            cs, cd = O.topk(O.dot_scores(rows, q[b]), depth, False, base)
            tv = np.array([t for t in queries[b] if t < vocab], np.uint32)
            bs, bd = O.topk(O.bm25_scores(terms, offs, vocab, tv), depth, True, base)
            fs, fd = O.rrf_fuse(cd, bd, k)
            ref.append((cs, cd, bs, bd, fs, fd))
        t1 = time.perf_counter()
        for mode in (idx.BM25_STREAM, idx.BM25_WAVE, idx.BM25_TAAT, idx.BM25_SCAN):
            idx.set_bm25_mode(mode)
            L = idx.search_lists(q, qt, qo, depth=depth)
            R = idx.search(q, qt, qo, k=k, depth=depth)
            for b in range(B):
                cs, cd, bs, bd, fs, fd = ref[b]
                tag = (case, n, dim, vocab, B, depth, k, base, mode, b)
                assert int(L.cos_counts[b]) == cd.size and np.array_equal(L.cos_docs[b][:cd.size], cd), tag
                assert np.array_equal(L.cos_scores[b][:cd.size], cs), tag
                assert int(L.bm25_counts[b]) == bd.size and np.array_equal(L.bm25_docs[b][:bd.size], bd), tag
                assert np.array_equal(L.bm25_scores[b][:bd.size].view(np.uint32), bs.view(np.uint32)), tag
                assert int(R.counts[b]) == fd.size and np.array_equal(R.docs[b][:fd.size], fd), tag
                assert np.array_equal(R.scores[b][:fd.size].view(np.uint32), fs.view(np.uint32)), tag
        idx.close()
        if log:
            log.write("case %d n=%d dim=%d vocab=%d B=%d depth=%d k=%d: oracle %.1fs gpu %.1fs\n" % (
                case, n, dim, vocab, B, depth, k, t1 - t0, time.perf_counter() - t1))
            log.flush()
    if log:
        log.close()
