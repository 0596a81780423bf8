"""Split-precision cosine mode (oi_set_cosine_mode(OI_COSINE_SPLIT)): every f32 operand split exactly into three
bf16 values, six bf16 MFMAs per product, f32 accumulation -- over the same f32 corpus.  Same bar as the exact
kernel (1e-5 absolute vs the f64 oracle, written in COS_TOL), plus: the observed error must stay f32-grade
(<= 5e-7 on unit vectors), the small-integer pipeline must be bit-exact, and both modes must agree on the
lists wherever scores are not within rounding of each other."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
COS_TOL = 1e-5
F32_GRADE = 5e-7


@pytest.fixture(scope="module")
def ctx():
    import openintel_amd as oi
    from openintel_amd import _lib
    c = oi.HipContext(0)
    c.set_cosine_mode(_lib.OI_COSINE_SPLIT)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import lib
    return lib


def _forward(rng, n, vocab=50):
    lens = rng.integers(1, 9, size=n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    return rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32), offs


def _index(ctx, rows, terms, offs, vocab, base=0):
    import openintel_amd as oi
    idx = oi.HybridIndex(ctx, rows.shape[0], rows.shape[1], vocab, base)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    idx.finalize()
    return idx


@pytest.mark.parametrize("B,dim,n", [(9, 768, 5000), (40, 768, 9000), (64, 768, 60_000), (70, 384, 6000),
                                     (33, 384, 40_000), (130, 768, 3000)])
def test_split_cosine_is_f32_grade(ctx, O, B, dim, n):
    from openintel_amd import synth
    rows = synth.embeddings_np(n, dim, seed=3 + B)
    q = synth.embeddings_np(B, dim, seed=55 + B)
    rng = np.random.default_rng(B)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    worst = 0.0
    for depth in (10, 1000):
        L = idx.search_lists(q, qt, qo, depth=depth)
        for b in range(B):
            ref = O.dot_scores(rows, q[b])
            c = int(L.cos_counts[b])
            assert c == min(depth, n)
            d, s = L.cos_docs[b][:c], L.cos_scores[b][:c]
            err = np.abs(s.astype(np.float64) - ref[d]).max()
            worst = max(worst, err)
            assert np.unique(d).size == c and err <= COS_TOL and (np.diff(s) <= 0).all()
            kth = np.sort(ref)[::-1][c - 1]
            assert np.isin(np.nonzero(ref > kth + 2 * COS_TOL)[0], d).all() and (ref[d] >= kth - 2 * COS_TOL).all()
    assert worst <= F32_GRADE, "split products should be as accurate as f32 accumulation (got %.3g)" % worst
    idx.close()


def test_split_extreme_values_and_scales(ctx, O):
    # values whose three bf16 parts are all non-trivial, mixed magnitudes, denormal-range remainders
    rng = np.random.default_rng(11)
    n, dim, B = 4000, 768, 64
    rows = (rng.standard_normal((n, dim)) * np.exp(rng.uniform(-12, 3, size=(n, 1)))).astype(np.float32)
    rows[5] = 0.0
    rows[6] = np.float32(1.0) + np.float32(2.0 ** -23)            # needs all 24 bits
    rows[7, ::2] = np.float32(3.0e-39)                              # denormals
    q = (rng.standard_normal((B, dim)) * np.exp(rng.uniform(-6, 2, size=(B, 1)))).astype(np.float32)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    L = idx.search_lists(q, np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32), depth=200)
    for b in range(B):
        ref = O.dot_scores(rows, q[b])
        d, s = L.cos_docs[b][:200], L.cos_scores[b][:200]
        scale = np.abs(rows[d].astype(np.float64)) @ np.abs(q[b].astype(np.float64))   # sum |x_k q_k|
        assert (np.abs(s - ref[d]) <= 4e-7 * scale + 1e-30).all()   # relative to the products' magnitude
    idx.close()


@pytest.mark.parametrize("n,dim,B,depth,k", [(70_000, 384, 9, 1000, 100), (40_000, 768, 64, 10, 10),
                                             (300_000, 384, 33, 100, 50)])
def test_split_hybrid_pipeline_bit_exact(ctx, O, n, dim, B, depth, k):
    rng = np.random.default_rng(n + B)
    rows = rng.integers(-3, 4, size=(n, dim)).astype(np.float32)
    q = rng.integers(-3, 4, size=(B, dim)).astype(np.float32)
    vocab = 300
    terms, offs = _forward(rng, n, vocab)
    qt = rng.integers(0, 12, size=B * 4).astype(np.uint32)
    qo = (np.arange(B + 1) * 4).astype(np.uint32)
    idx = _index(ctx, rows, terms, offs, vocab, base=1000)
    L = idx.search_lists(q, qt, qo, depth=depth)
    R = idx.search(q, qt, qo, k=k, depth=depth)
    for b in range(B):
        cs, cd = O.topk(O.dot_scores(rows, q[b]), depth, False, 1000)
        bs, bd = O.topk(O.bm25_scores(terms, offs, vocab, qt[qo[b]:qo[b + 1]]), depth, True, 1000)
        fs, fd = O.rrf_fuse(cd, bd, k)
        assert int(L.cos_counts[b]) == cd.size
        assert np.array_equal(L.cos_docs[b][:cd.size], cd) and np.array_equal(L.cos_scores[b][:cd.size], cs)
        assert int(R.counts[b]) == fd.size and np.array_equal(R.docs[b][:fd.size], fd)
        assert np.array_equal(R.scores[b][:fd.size].view(np.uint32), fs.view(np.uint32))
    idx.close()


def test_mode_switch_and_unsupported_shapes(ctx, O):
    """Shapes without a split build (dim 1024, B <= 8) silently use the exact kernels; the mode can be switched
    per call sequence and both modes rank the same docs apart from near-ties."""
    from openintel_amd import _lib, synth
    n, dim, B = 30_000, 768, 64
    rows = synth.embeddings_np(n, dim, seed=21)
    q = synth.embeddings_np(B, dim, seed=22)
    rng = np.random.default_rng(0)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    Ls = idx.search_lists(q, qt, qo, depth=100)
    ctx.set_cosine_mode(_lib.OI_COSINE_EXACT)
    Le = idx.search_lists(q, qt, qo, depth=100)
    ctx.set_cosine_mode(_lib.OI_COSINE_SPLIT)
    assert np.abs(Ls.cos_scores - Le.cos_scores).max() <= 3e-7
    same = (Ls.cos_docs == Le.cos_docs).mean()
    assert same > 0.999          # order can only differ between scores closer than the two roundings
    L1 = idx.search_lists(q[:3], qt[:3], qo[:4], depth=50)   # B <= 8: the GEMV kernel, exact in any mode
    for b in range(3):
        ref = O.dot_scores(rows, q[b])
        assert np.abs(L1.cos_scores[b][:50] - ref[L1.cos_docs[b][:50]]).max() <= COS_TOL
    idx.close()
    with pytest.raises(_lib.OiError):
        ctx.set_cosine_mode(7)
