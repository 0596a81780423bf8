"""Self-checks of the PARITY-UNPINNED retrieval oracle (BM25 / cosine / top-k / RRF).
The reference has no such code; these tests only make sure the C
restatement of the textbook definitions agrees with an independent numpy statement."""
import numpy as np

from oracle import lib as O


def _forward(rng, n_docs, vocab, max_len=12):
    lens = rng.integers(1, max_len + 1, size=n_docs)
    offs = np.zeros(n_docs + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    terms = rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32)
    return terms, offs


def test_bm25_matches_numpy_f32_formula():
    rng = np.random.default_rng(0)
    n, vocab = 300, 40
    terms, offs = _forward(rng, n, vocab)
    df, tot = O.bm25_df(terms, offs, vocab)
    assert tot == offs[-1]
    q = np.array([3, 7, 7, 11], dtype=np.uint32)
    got = O.bm25_scores(terms, offs, vocab, q)
    f = np.float32
    avgdl = f(np.float64(tot) / np.float64(n))
    want = np.zeros(n, dtype=np.float32)
    for d in range(n):
        doc = terms[int(offs[d]):int(offs[d + 1])]
        kd = f(1.2) * (f(f(1.0) - f(0.75)) + f(0.75) * (f(doc.size) / avgdl))
        s = f(0.0)
        for t in q:
            tf = int((doc == t).sum())
            if tf == 0:
                continue
            idf = f(np.log(1.0 + (n - float(df[t]) + 0.5) / (float(df[t]) + 0.5)))
            w = (f(tf) * f(f(1.2) + f(1.0))) / (f(tf) + kd)
            s = f(s + f(idf * w))
        want[d] = s
        assert int((np.unique(doc) == np.unique(doc)).sum()) >= 1
    np.testing.assert_array_equal(got, want)
    assert df[7] == sum(1 for d in range(n) if 7 in terms[int(offs[d]):int(offs[d + 1])])


def test_topk_order_and_ties():
    s = np.array([0.5, 0.9, 0.5, -0.0, 0.0, 0.9, np.nan, 0.1], dtype=np.float32)
    sc, dc = O.topk(s, 5)
    assert dc.tolist() == [1, 5, 0, 2, 7]
    sc, dc = O.topk(s, 100)
    assert dc.tolist() == [1, 5, 0, 2, 7, 3, 4]           # -0.0 == +0.0 -> doc id asc; NaN dropped
    sc, dc = O.topk(s, 100, positive_only=True, doc_base=10)
    assert dc.tolist() == [11, 15, 10, 12, 17]
    rng = np.random.default_rng(1)
    x = rng.standard_normal(5000).astype(np.float32)
    x[rng.integers(0, 5000, 500)] = 0.25                  # many exact ties
    sc, dc = O.topk(x, 700)
    order = np.lexsort((np.arange(5000), -x.astype(np.float64)))[:700]
    assert dc.tolist() == order.tolist()
    np.testing.assert_array_equal(sc, x[order])


def test_merge_and_rrf():
    sa, da = np.array([0.9, 0.5, 0.1], np.float32), np.array([4, 2, 9], np.uint32)
    sb, db = np.array([0.9, 0.5], np.float32), np.array([1, 7], np.uint32)
    ms, md = O.merge_ranked([sa, sb], [da, db], 4)
    assert md.tolist() == [1, 4, 2, 7] and ms.tolist() == [np.float32(0.9)] * 2 + [np.float32(0.5)] * 2

    a = np.array([10, 20, 30, 40], np.uint32)
    b = np.array([30, 50, 10], np.uint32)
    fs, fd = O.rrf_fuse(a, b, 10)
    f = np.float32
    want = {}
    for r, d in enumerate(a, 1):
        want[int(d)] = f(1.0) / (f(60.0) + f(r))
    for r, d in enumerate(b, 1):
        c = f(1.0) / (f(60.0) + f(r))
        want[int(d)] = f(want[int(d)] + c) if int(d) in want else c
    order = sorted(want, key=lambda d: (-float(want[d]), d))
    assert fd.tolist() == order
    assert fs.tolist() == [want[d] for d in order]
    fs2, fd2 = O.rrf_fuse(a, b, 2)
    assert fd2.tolist() == order[:2]
    # doc only in A at rank r ties with doc only in B at rank r -> lower doc id first
    fs3, fd3 = O.rrf_fuse(np.array([9, 5], np.uint32), np.array([3, 8], np.uint32), 4)
    assert fd3.tolist() == [3, 9, 5, 8]


def test_cosine_and_hybrid_smoke_config0():
    # BASELINE.json configs[0]: 1k posts, 384-d, single query, top-10, CPU only
    rng = np.random.default_rng(0xA11CE)
    rows = O.l2_normalize_rows(rng.standard_normal((1000, 384)).astype(np.float32))
    np.testing.assert_allclose(np.linalg.norm(rows.astype(np.float64), axis=1), 1.0, atol=1e-6)
    q = O.l2_normalize_rows(rng.standard_normal((1, 384)).astype(np.float32))[0]
    cs = O.dot_scores(rows, q)
    np.testing.assert_allclose(cs, rows.astype(np.float64) @ q.astype(np.float64), atol=1e-7)
    terms, offs = _forward(rng, 1000, 64, max_len=20)
    res = O.hybrid_search(rows, terms, offs, 64, q, np.array([1, 2, 3, 4], np.uint32), k=10, depth=100)
    assert len(res["fused"][1]) == 10 and len(set(res["fused"][1].tolist())) == 10
    assert set(res["fused"][1].tolist()) <= set(res["cos"][1].tolist()) | set(res["bm25"][1].tolist())


def test_batch_driver_equals_per_query_calls_for_any_thread_count():
    """bench.py's cpu_baseline times oio_hybrid_search_batch on 1 and on all host threads: it must be the same scalar
    pipeline as the per-query calls, whatever the thread count."""
    from oracle import lib as O
    rng = np.random.default_rng(7)
    n, dim, vocab, B, depth, k = 1500, 24, 40, 9, 60, 15
    rows = O.l2_normalize_rows(rng.standard_normal((n, dim)).astype(np.float32))
    lens = rng.integers(1, 9, size=n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    terms = rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32)
    q = O.l2_normalize_rows(rng.standard_normal((B, dim)).astype(np.float32))
    qt = rng.integers(0, vocab, size=B * 3).astype(np.uint32)
    qo = (np.arange(B + 1) * 3).astype(np.uint32)
    outs = [O.hybrid_search_batch(rows, terms, offs, vocab, q, qt, qo, k, depth, n_threads=t) for t in (1, 2, 5)]
    assert outs[0][3] == 1 and outs[1][3] in (1, 2)
    # the cache-blocked driver (rows / docs outermost): the same results, with and without a df vector, any thread count
    df, _ = O.bm25_df(terms, offs, vocab)
    outs += [O.hybrid_search_batch(rows, terms, offs, vocab, q, qt, qo, k, depth, n_threads=t, blocked=True, df=d)
             for t, d in ((1, None), (3, df), (8, df))]
    for s, d, c, _ in outs:
        for b in range(B):
            fs, fd = O.hybrid_search(rows, terms, offs, vocab, q[b], qt[qo[b]:qo[b + 1]], k, depth)["fused"]
            assert c[b] == fd.size and np.array_equal(d[b, :fd.size], fd)
            assert np.array_equal(s[b, :fs.size].view(np.uint32), fs.view(np.uint32))
    assert O.max_threads() >= 1
