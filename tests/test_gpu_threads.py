"""The boundary under host threads.  The reference's port is `Send + Sync` (src/domain/ports/post_analyzer.rs:7) and
`run_scan` issues many concurrent `analyze` calls on shared adapters (src/mcp/tools.rs:205-220): one oi_ctx must
take calls from several host threads at once (it serialises them internally), and separate contexts must be usable
side by side (the per-kernel LDS attribute one-shots and the lexicon table are process-wide: oi_dyn_lds,
std::call_once)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _posts(rng, n):
    words = ["moon", "buy", "calls", "puts", "crash", "dump", "squeeze", "yolo", "hold", "the", "a", "stock", "AAPL",
             "bullish", "bearish", "rocket", "short", "long", "sell", "tendies"]
    return [" ".join(rng.choice(words, size=rng.integers(3, 30))) for _ in range(n)]


def _forward(rng, n, vocab=64):
    lens = rng.integers(1, 9, size=n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    return rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32), offs


def _run_threads(fns):
    errs = []

    def wrap(f):
        def g():
            try:
                f()
            except BaseException as e:  # noqa: BLE001 - reported below
                errs.append(e)
        return g
    ts = [threading.Thread(target=wrap(f)) for f in fns]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a thread hung"
    if errs:
        raise errs[0]


def test_eight_threads_share_one_ctx():
    import openintel_amd as oi
    from openintel_amd import synth
    rng = np.random.default_rng(0)
    ctx = oi.HipContext(0)
    an = oi.HipLexiconAnalyzer(ctx)
    n, dim, B = 30_000, 384, 16
    rows = synth.embeddings_np(n, dim, seed=1)
    terms, offs = _forward(rng, n)
    idx = oi.HybridIndex(ctx, n, dim, 64)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    idx.finalize()
    jobs = []
    for t in range(8):
        texts = _posts(np.random.default_rng(100 + t), 400 + 50 * t)
        q = synth.embeddings_np(B, dim, seed=200 + t)
        qt = np.random.default_rng(300 + t).integers(0, 64, size=B * 3).astype(np.uint32)
        qo = (np.arange(B + 1) * 3).astype(np.uint32)
        jobs.append((texts, q, qt, qo))
    # serial answers first
    serial = []
    for texts, q, qt, qo in jobs:
        pol, spec = an.score_texts(texts)
        R = idx.search(q, qt, qo, k=20, depth=100)
        serial.append((pol.copy(), spec.copy(), R.docs.copy(), R.scores.copy(), R.counts.copy()))
    got = [None] * 8

    def worker(t):
        def f():
            texts, q, qt, qo = jobs[t]
            out = None
            for _ in range(6):                       # interleave lexicon and search calls on the shared ctx
                pol, spec = an.score_texts(texts)
                R = idx.search(q, qt, qo, k=20, depth=100)
                out = (pol, spec, R.docs, R.scores, R.counts)
                for a, b in zip(out, serial[t]):
                    assert np.array_equal(a, b)
            got[t] = out
        return f
    _run_threads([worker(t) for t in range(8)])
    assert all(g is not None for g in got)
    idx.close()
    ctx.close()


def test_two_contexts_on_two_threads():
    import openintel_amd as oi
    from openintel_amd import synth
    n, dim, B = 20_000, 768, 40

    results = {}

    def worker(name, seed):
        def f():
            rng = np.random.default_rng(seed)
            ctx = oi.HipContext(0)                  # its own ctx, stream and workspaces
            rows = synth.embeddings_np(n, dim, seed=seed)
            terms, offs = _forward(rng, n)
            idx = oi.HybridIndex(ctx, n, dim, 64)
            idx.set_embeddings(rows, normalize=False)
            idx.set_forward(terms, offs)
            idx.finalize()
            q = synth.embeddings_np(B, dim, seed=seed + 1)
            qt = rng.integers(0, 64, size=B * 2).astype(np.uint32)
            qo = (np.arange(B + 1) * 2).astype(np.uint32)
            outs = []
            for _ in range(4):
                L = idx.search_lists(q, qt, qo, depth=200)
                outs.append((L.cos_docs.copy(), L.cos_scores.copy(), L.bm25_docs.copy(), L.bm25_scores.copy()))
            for o in outs[1:]:
                assert all(np.array_equal(a, b) for a, b in zip(o, outs[0]))
            an = oi.HipLexiconAnalyzer(ctx)
            pol, spec = an.score_texts(_posts(rng, 500))
            results[name] = (outs[0], pol, spec, rows, q)
            idx.close()
            ctx.close()
        return f
    _run_threads([worker("a", 11), worker("b", 22)])
    from oracle import lib as O
    for name in ("a", "b"):
        (cd, cs, _, _), _, _, rows, q = results[name]
        ref = O.dot_scores(rows, q[0])
        top = np.argsort(-ref, kind="stable")[:5]
        assert np.array_equal(cd[0][:5].astype(np.int64), top)
        assert np.abs(cs[0][:5] - ref[top]).max() <= 1e-5
