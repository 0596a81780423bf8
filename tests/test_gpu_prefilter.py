"""Screened cosine mode (oi_set_cosine_mode(OI_COSINE_SCREEN), csrc/cosine_prefilter.hip): a bf16 screen with a
proven error bound (round 2: built from the MEASURED rounding errors of corpus and query; the worst-case tests
are test_bound_holds_when_every_coordinate_is_a_bf16_tie and test_tie_rounding_adversary_*) chooses the rows that can reach the list, exact f32 scores are computed for those rows only,
and a query whose survivors do not fit falls back -- inside the same call -- to the exact kernel.  What comes out
must be the exact scorer's lists: same bar as the exact kernel (1e-5 absolute vs the f64 oracle, COS_TOL), the
small-integer pipeline bit for bit (that corpus cannot be screened: the fallback is what is tested there), and
agreement with the exact mode apart from near-ties.  `screen_gate` tells which regime a search ran in."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
COS_TOL = 1e-5


# Round 5: the default mode streams the index's bf16 SCREENING COPY (made at finalize); OI_COSINE_SCREEN_STREAM is rounds 1-4's
# default, the same screen converting the f32 rows on the fly.  Every test of this module runs in both: the adversarial cases
# are the new default's tests too.
@pytest.fixture(scope="module", params=["copy", "stream"])
def ctx(request):
    import openintel_amd as oi
    from openintel_amd import _lib
    c = oi.HipContext(0)
    c.screen_mode = _lib.OI_COSINE_SCREEN if request.param == "copy" else _lib.OI_COSINE_SCREEN_STREAM
    c.set_cosine_mode(c.screen_mode)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import lib
    return lib


def _gate(ctx):
    return ctx.profile_read("screen_gate")[0]


def _forward(rng, n, vocab=50):
    lens = rng.integers(1, 9, size=n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    return rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32), offs


def _index(ctx, rows, terms, offs, vocab, base=0, normalize=False):
    import openintel_amd as oi
    idx = oi.HybridIndex(ctx, rows.shape[0], rows.shape[1], vocab, base)
    idx.set_embeddings(rows, normalize=normalize)
    idx.set_forward(terms, offs)
    idx.finalize()
    return idx


def _check(L, b, ref, depth, n, base=0):
    c = int(L.cos_counts[b])
    assert c == min(depth, n)
    d, s = L.cos_docs[b][:c].astype(np.int64) - base, L.cos_scores[b][:c]
    assert np.unique(d).size == c and d.min() >= 0 and d.max() < n
    assert ((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (d[:-1] < d[1:]))).all(), "not sorted by (score desc, doc asc)"
    assert np.abs(s.astype(np.float64) - ref[d]).max() <= COS_TOL
    kth = np.sort(ref)[::-1][c - 1]
    assert np.isin(np.nonzero(ref > kth + 2 * COS_TOL)[0], d).all(), "a clearly better doc is missing"
    assert (ref[d] >= kth - 2 * COS_TOL).all(), "a clearly worse doc is present"


@pytest.mark.parametrize("shift", [-0.5, 0.0, 0.25])
def test_margin_selects_on_negative_zero_and_clustered_scores(ctx, O, shift):
    """Round 4: a margin select takes the lower edge of the k'-th key's 22-bit bin as its threshold, and the screen compares floats
    where it compared keys.  Scores of one sign bit or the other, a k'-th score at exactly 0.0 (half the rows orthogonal to every
    query), and thousands of scores inside one leading-digit bin: the lists must stay the exact scorer's (same bars as everywhere)."""
    from openintel_amd import synth
    n, dim, B = 70_000, 768, 40
    rng = np.random.default_rng(int(100 + 10 * shift))
    rows = synth.embeddings_np(n, dim, seed=901)
    q = synth.embeddings_np(B, dim, seed=902)
    axis = np.zeros(dim, np.float32)
    axis[0] = 1.0
    if shift == 0.0:
        rows[::2, :] = 0.0           # every second row scores exactly 0 against every query ...
        rows[::2, 1] = 1.0
        q[:, 1] = 0.0                # ... because the queries have nothing in that coordinate
    else:
        q = (q + shift * 8.0 * axis).astype(np.float32)        # all scores move by shift * 8 * x_0: one sign for most rows
        rows[:, 0] = np.abs(rows[:, 0])                         # (shift < 0: negative scores, shift > 0: positive)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50, base=5)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    for depth in (100, 1000):
        L = idx.search_lists(q, qt, qo, depth=depth)
        for b in range(0, B, 5):
            _check(L, b, O.dot_scores(rows, q[b]), depth, n, base=5)
    idx.close()


def _screened(ctx, B):
    """Is a batch of B queries screened in this module's mode?  B > 8 always; B <= 8 (round 5) only when there is a screening
    copy to stream -- the f32-stream screen would read the bytes the GEMV reads.  `screen_gate` is -1 for an unscreened search."""
    from openintel_amd import _lib
    return B > 8 or ctx.screen_mode == _lib.OI_COSINE_SCREEN


@pytest.mark.parametrize("B,dim,n", [(9, 768, 5000), (40, 768, 9000), (64, 768, 60_000), (70, 384, 6000),
                                     (33, 384, 40_000), (130, 768, 3000), (64, 768, 300_000),
                                     (1, 768, 40_000), (3, 384, 20_000), (8, 768, 60_000), (1, 384, 300_000)])
def test_screened_lists_meet_the_exact_bar(ctx, O, B, dim, n):
    from openintel_amd import synth
    rows = synth.embeddings_np(n, dim, seed=3 + B)
    q = synth.embeddings_np(B, dim, seed=55 + B)
    rng = np.random.default_rng(B)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50, base=77)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    for depth in (10, 1000):
        L = idx.search_lists(q, qt, qo, depth=depth)
        assert _gate(ctx) == (0.0 if _screened(ctx, B) else -1.0), "unit vectors: the screen must hold (no fallback)"
        for b in range(B if n <= 60_000 else min(B, 8)):
            _check(L, b, O.dot_scores(rows, q[b]), depth, n, base=77)
    idx.close()


def test_screen_and_exact_modes_agree(ctx, O):
    from openintel_amd import _lib, synth
    n, dim, B = 100_000, 768, 64
    rows = synth.embeddings_np(n, dim, seed=21)
    q = synth.embeddings_np(B, dim, seed=22)
    rng = np.random.default_rng(0)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    Ls = idx.search_lists(q, qt, qo, depth=1000)
    assert _gate(ctx) == 0.0
    ctx.set_cosine_mode(_lib.OI_COSINE_EXACT)
    Le = idx.search_lists(q, qt, qo, depth=1000)
    ctx.set_cosine_mode(ctx.screen_mode)
    assert np.array_equal(Ls.cos_counts, Le.cos_counts)
    assert np.abs(Ls.cos_scores - Le.cos_scores).max() <= 5e-7    # two f32 summation orders
    assert (Ls.cos_docs == Le.cos_docs).mean() > 0.999            # order can only differ between near-ties
    for b in range(B):                                            # and membership only at the k-th boundary
        odd = np.setxor1d(Ls.cos_docs[b], Le.cos_docs[b])
        if odd.size:
            ref = O.dot_scores(rows, q[b])
            assert np.abs(ref[odd] - np.sort(ref)[::-1][999]).max() <= 1e-6
    idx.close()


@pytest.mark.parametrize("n,dim,B,depth,k", [(70_000, 384, 9, 1000, 100), (40_000, 768, 64, 10, 10),
                                             (300_000, 384, 33, 100, 50), (50_000, 768, 1, 100, 10), (60_000, 384, 5, 1000, 100)])
def test_integer_pipeline_bit_exact(ctx, O, n, dim, B, depth, k):
    """Small-integer embeddings (exact dot products in any order, thousands of ties, norms far from 1): whichever
    regime a query lands in, the lists are the oracle's bit for bit, hybrid fusion included."""
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


def test_survivor_overflow_opens_the_exact_pipeline(ctx, O):
    """6000 exact copies of one row: every copy scores the same, so for a query near that row more keys sit inside
    the margin than the screen keeps (4096).  The gate must open and the exact kernel's lists come out: the copies
    in doc-id order first, bit for bit the oracle's ranking."""
    from openintel_amd import synth
    rng = np.random.default_rng(3)
    n, dim, B = 60_000, 768, 16
    rows = synth.embeddings_np(n, dim, seed=30)
    copies = rng.choice(n, size=6000, replace=False)
    rows[copies] = rows[copies[0]]
    q = synth.embeddings_np(B, dim, seed=31)
    q[0] = rows[copies[0]]
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    L = idx.search_lists(q, qt, qo, depth=1000)
    assert _gate(ctx) != 0.0, "6000 tied rows cannot fit the screen's 4096 survivors"
    ref = O.dot_scores(rows, q[0])
    d = L.cos_docs[0][:1000].astype(np.int64)
    assert np.array_equal(d, np.sort(copies)[:1000])                 # ties rank by doc id
    assert np.abs(L.cos_scores[0][:1000] - ref[d]).max() <= COS_TOL
    for b in range(1, B):
        _check(L, b, O.dot_scores(rows, q[b]), 1000, n)
    L = idx.search_lists(q[1:], qt[1:], qo[:-1], depth=1000)          # without that query the screen holds again
    assert _gate(ctx) == 0.0
    for b in range(B - 1):
        _check(L, b, O.dot_scores(rows, q[b + 1]), 1000, n)
    idx.close()


def test_hard_cases_for_the_bound(ctx, O):
    """Rows of very different norms (the bound uses the LARGEST), near-duplicates of the best rows (many keys inside
    the margin), a zero row, unnormalised queries, and a NaN in the corpus (no bound: the gate must open)."""
    from openintel_amd import synth
    rng = np.random.default_rng(5)
    n, dim, B = 50_000, 768, 40
    rows = synth.embeddings_np(n, dim, seed=8)
    rows[100:200] *= 3.0                                   # a hundred long rows: they own the top of every list
    rows[300] = 0.0
    q = synth.embeddings_np(B, dim, seed=9) * rng.uniform(0.1, 20.0, size=(B, 1)).astype(np.float32)
    best = int(np.argmax(rows @ q[0]))
    rows[1000:1400] = rows[best] * (1.0 + 1e-4 * rng.standard_normal((400, 1))).astype(np.float32)   # near-duplicates
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    for depth in (10, 500):
        L = idx.search_lists(q, qt, qo, depth=depth)
        for b in range(B):
            ref = O.dot_scores(rows, q[b])
            c = int(L.cos_counts[b])
            d, s = L.cos_docs[b][:c].astype(np.int64), L.cos_scores[b][:c]
            tol = COS_TOL * max(1.0, float(np.abs(ref).max()))   # scores here reach ~60: the bar scales with them
            assert c == depth and np.unique(d).size == c and (np.diff(s) <= 0).all()
            assert np.abs(s.astype(np.float64) - ref[d]).max() <= tol
            kth = np.sort(ref)[::-1][c - 1]
            assert np.isin(np.nonzero(ref > kth + 2 * tol)[0], d).all() and (ref[d] >= kth - 2 * tol).all()
    # a query without a bound (NaN / infinite / huge norm): that query is scored by the exact kernel against every
    # row (the others keep their screened thresholds), the gate says so, and the lists are the exact mode's
    from openintel_amd import _lib
    q2 = q.copy()
    q2[3, 5] = np.nan
    q2[4] *= np.float32(1e20)
    L = idx.search_lists(q2, qt, qo, depth=100)
    assert _gate(ctx) != 0.0
    ctx.set_cosine_mode(_lib.OI_COSINE_EXACT)
    Le = idx.search_lists(q2, qt, qo, depth=100)
    ctx.set_cosine_mode(ctx.screen_mode)
    assert np.array_equal(L.cos_counts, Le.cos_counts) and int(L.cos_counts[3]) == 0 and int(L.cos_counts[4]) == 100
    for b in range(B):       # (entries past a list's count are not defined: the screened pass may have left its own there)
        c = int(L.cos_counts[b])
        assert np.array_equal(L.cos_docs[b][:c], Le.cos_docs[b][:c]) and np.array_equal(L.cos_scores[b][:c], Le.cos_scores[b][:c]), b
    idx.close()
    # a NaN in the corpus: no bound for any query -- such an index is never screened (decided when the rows are set)
    rows[7, 3] = np.nan
    idx = _index(ctx, rows, terms, offs, 50)
    L = idx.search_lists(q, qt, qo, depth=10)
    assert _gate(ctx) == -1.0, "the screen must not have run"
    ref = O.dot_scores(np.delete(rows, 7, axis=0), q[0])
    assert int(L.cos_counts[0]) == 10 and 7 not in L.cos_docs[0][:10]   # the exact scorer drops NaN scores
    assert abs(float(L.cos_scores[0][0]) - float(ref.max())) <= COS_TOL * max(1.0, float(np.abs(ref).max()))
    idx.close()


@pytest.mark.parametrize("n_long", [100, 3000])
def test_a_few_long_rows_do_not_open_the_gate(ctx, O, n_long):
    """the margin used the LARGEST row norm of the corpus, so a hundred rows 3x longer than the
    rest sent every query to the exact fallback.  With the two-class margin (rows whose norms stand out -- at most 1024 -- are
    left out of the screen's thresholds and always rescored) the gate stays SHUT, the long rows own the top of every list,
    and the lists pass the exact kernel's bars against the oracle.  3000 long rows are too many to set aside: one class as
    before (the gate may open) -- and the lists are still right."""
    from openintel_amd import _lib, synth
    rng = np.random.default_rng(n_long)
    n, dim, B, depth = 120_000, 768, 24, 1000
    rows = synth.embeddings_np(n, dim, seed=21)
    long_ids = rng.choice(n, size=n_long, replace=False)
    rows[long_ids] *= 3.0
    q = synth.embeddings_np(B, dim, seed=22)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50, base=11)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    assert idx.long_rows() == (n_long if n_long <= 1024 else 0)
    g0 = _gate(ctx)
    L = idx.search_lists(q, qt, qo, depth=depth)
    if n_long <= 1024:
        assert _gate(ctx) == 0.0 or _gate(ctx) == g0 == 0.0, "the screen gave up although the long rows were set aside"
        assert _gate(ctx) == 0.0
    for b in range(B):
        ref = O.dot_scores(rows, q[b]).astype(np.float64)
        c = int(L.cos_counts[b])
        d, s = L.cos_docs[b][:c].astype(np.int64) - 11, L.cos_scores[b][:c]
        tol = COS_TOL * 3.0                                     # scores of the long rows reach 3x the unit rows'
        assert c == depth and np.unique(d).size == c
        assert ((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (d[:-1] < d[1:]))).all()
        assert np.abs(s.astype(np.float64) - ref[d]).max() <= tol
        kth = np.sort(ref)[::-1][c - 1]
        assert np.isin(np.nonzero(ref > kth + 2 * tol)[0], d).all(), "a clearly better doc is missing"
        assert (ref[d] >= kth - 2 * tol).all(), "a clearly worse doc is present"
    # the same lists from the exact mode: same docs at (nearly) every rank, same scores to the bar
    ctx.set_cosine_mode(_lib.OI_COSINE_EXACT)
    Le = idx.search_lists(q, qt, qo, depth=depth)
    ctx.set_cosine_mode(ctx.screen_mode)
    same = np.mean(L.cos_docs[:, :depth] == Le.cos_docs[:, :depth])
    assert same > 0.98, same
    assert np.abs(L.cos_scores[:, :depth] - Le.cos_scores[:, :depth]).max() <= COS_TOL * 3.0
    idx.close()


# ---------------------------------------------------------------------------------------------------------------
# The bound itself, on its worst case.  Round 1 assumed a bf16 unit roundoff of 2^-9; it is 2^-8, a product of
# two rounded values is off by up to 2^-7, and with EVERY coordinate at a bf16 tie the errors do not cancel.
# Random data never shows this (errors cancel like 1/sqrt(d)); these tests are deterministic worst cases.
def _tie(m, e=-5):
    """(1 + (m + 1/2)/128) * 2^e: exactly halfway between the bf16 values with mantissa m and m + 1 (RNE: to the even one)."""
    return np.float32((1.0 + (m + 0.5) / 128.0) * 2.0 ** e)


def _old_eps(rows, q):
    """Round 1's (unsound) bound: (2^-8 + 2^-12) * 1.001 * max|x| * |q|."""
    X = np.sqrt((rows.astype(np.float64) ** 2).sum(1)).max()
    return (2.0 ** -8 + 2.0 ** -12) * 1.001 * X * np.sqrt((q.astype(np.float64) ** 2).sum())


@pytest.mark.parametrize("dim", [768, 384])
def test_bound_holds_when_every_coordinate_is_a_bf16_tie(ctx, dim):
    rng = np.random.default_rng(dim)
    n, B = 4096, 16
    # every value a tie, a random mantissa and binade per coordinate; within a row all mantissas have one parity, so
    # all its coordinates round the same way (even m: down, odd m: up) and the errors add up instead of cancelling
    m = 2 * rng.integers(0, 32, size=(n, dim)) + (np.arange(n) & 1)[:, None]
    e = rng.integers(-7, -3, size=(n, dim))
    rows = ((1.0 + (m + 0.5) / 128.0) * 2.0 ** e).astype(np.float32)
    rows[::7] *= -1.0
    q = rows[:B].copy()                                   # x == q on the diagonal: errors of both operands line up
    q[B // 2:] = rows[n - B // 2:] * np.float32(4.0)      # another binade: ties stay ties
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    st, eps = idx.screen_probe(q, 0, n)
    s = q.astype(np.float64) @ rows.astype(np.float64).T
    err = np.abs(st.astype(np.float64) - s)
    assert np.isfinite(eps).all()
    assert (err <= eps[:, None]).all(), "the proven bound is violated: max err/eps = %g" % (err / eps[:, None]).max()
    # the test has teeth: on this data round 1's constant IS violated (by the diagonal pairs)
    old = np.array([_old_eps(rows, q[b]) for b in range(B)])
    assert (err > old[:, None]).any(), "this construction should break the round-1 bound"
    # and the bound is not vacuous: within 2.2x of the worst error seen
    assert (eps / err.max(axis=1)).min() < 2.2
    idx.close()


@pytest.mark.parametrize("n_comp,depth", [(150, 100), (600, 500), (5000, 1000)])
def test_tie_rounding_adversary_keeps_the_true_top_row(ctx, O, n_comp, depth):
    """VERDICT r01 / What's weak #1, spelled out: the planted row lives where q rounds DOWN (and rounds down itself),
    its >= k' competitors where q rounds UP (and round up themselves).  True scores: planted 0.389791 > competitor
    0.389769; screened: planted 0.386810 < competitor 0.392761.  Round 1's margin dropped the planted row.  The
    lists must be the exact mode's bit for bit, planted row first."""
    from openintel_amd import _lib, synth
    rng = np.random.default_rng(n_comp)
    dim, B, n = 768, 16, 20_000
    rows = (synth.embeddings_np(n, dim, seed=40) * np.float32(0.3)).astype(np.float32)   # filler, far below
    q = synth.embeddings_np(B, dim, seed=41)
    q[0, :384], q[0, 384:] = _tie(2), _tie(1)               # m=2 ties round down (to even), m=1 ties round up
    planted = 12_345
    comp = np.sort(rng.choice(np.setdiff1d(np.arange(n), [planted]), size=n_comp, replace=False))
    rows[planted, :384], rows[planted, 384:] = _tie(2), 0.0
    rows[comp, :384], rows[comp, 384:] = 0.0, _tie(3)       # m=3 ties round up (to even m=4)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    ref = O.dot_scores(rows, q[0])
    assert ref[planted] > ref[comp[0]] > np.delete(ref, np.append(comp, planted)).max()
    # what the screen sees, and what round 1 would have done with it
    st, eps = idx.screen_probe(q[:1], 0, n)
    assert st[0, planted] < st[0, comp[0]], "the screen must invert the pair for this test to mean anything"
    # Round 4: the planted row and its competitors are far longer than the filler -- up to 1024 such rows are SET ASIDE (left
    # out of the thresholds, rescored for every query), and the margin is the bound of the other rows.  5001 are too many:
    # one class, the corpus maxima, as in rounds 2-3.
    aside = idx.long_rows()
    assert aside == (n_comp + 1 if n_comp + 1 <= 1024 else 0)
    bound_rows = np.setdiff1d(np.arange(n), np.append(comp, planted)) if aside else np.arange(n)
    assert np.abs(st[0].astype(np.float64) - ref)[bound_rows].max() <= eps[0]
    if n_comp >= depth:
        tau = np.sort(st[0])[::-1][depth - 1]
        assert st[0, planted] < tau - 2 * _old_eps(rows, q[0]), "round 1's threshold would have dropped the planted row"
        if not aside:
            assert st[0, planted] >= tau - 2 * eps[0]
    L = idx.search_lists(q, qt, qo, depth=depth)
    gate = _gate(ctx)
    assert (gate != 0.0) == (n_comp > 4096), "only the 5000-competitor case overflows the survivors"
    ctx.set_cosine_mode(_lib.OI_COSINE_EXACT)
    Le = idx.search_lists(q, qt, qo, depth=depth)
    ctx.set_cosine_mode(ctx.screen_mode)
    c = int(L.cos_counts[0])
    assert c == depth and int(L.cos_docs[0][0]) == planted
    head = min(depth, 1 + n_comp)                          # planted + competitors in doc-id order, exact sums
    assert np.array_equal(L.cos_docs[0][:head], np.append([planted], comp)[:head].astype(np.uint32))
    assert np.array_equal(L.cos_docs[0][:head], Le.cos_docs[0][:head])
    assert np.array_equal(L.cos_scores[0][:head].view(np.uint32), Le.cos_scores[0][:head].view(np.uint32))
    assert np.array_equal(L.cos_scores[0][:head].astype(np.float64), ref[L.cos_docs[0][:head].astype(np.int64)])
    for b in range(B):
        _check(L, b, O.dot_scores(rows, q[b]) if b else ref, depth, n)
    idx.close()


def test_margin_is_data_dependent_and_tight_on_unit_vectors(ctx):
    """Unit Gaussian vectors at d = 768: the measured bound is about half the worst case 2^-7 (so configs[2] keeps
    ~2.4 k' survivors per query, inside the 4096-key carry), and it holds on a sample of rows."""
    from openintel_amd import synth
    rng = np.random.default_rng(9)
    n, dim, B = 50_000, 768, 32
    rows, q = synth.embeddings_np(n, dim, seed=1), synth.embeddings_np(B, dim, seed=2)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    st, eps = idx.screen_probe(q, 1000, 8192)
    s = q.astype(np.float64) @ rows[1000:1000 + 8192].astype(np.float64).T
    assert (np.abs(st - s) <= eps[:, None]).all()
    assert 0.0030 < eps.min() and eps.max() < 0.0048, eps      # worst case would be (2^-7 + ...) = 0.0080
    idx.close()


def test_screen_is_the_default_mode(O):
    import openintel_amd as oi
    from openintel_amd import synth
    c = oi.HipContext(0)                       # no oi_set_cosine_mode
    assert c.profile_read("screen_gate")[0] == -1.0
    rng = np.random.default_rng(2)
    n, dim, B = 20_000, 384, 12
    rows, q = synth.embeddings_np(n, dim, seed=5), synth.embeddings_np(B, dim, seed=6)
    terms, offs = _forward(rng, n)
    idx = _index(c, rows, terms, offs, 50)
    L = idx.search_lists(q, np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32), depth=100)
    assert c.profile_read("screen_gate")[0] == 0.0
    for b in range(B):
        _check(L, b, O.dot_scores(rows, q[b]), 100, n)
    idx.close()
    c.close()


def test_two_screened_shards_equal_one(ctx, O):
    """Row-sharding under the screen: each shard bounds with ITS largest norm and returns exact scores of its own
    rows; merged, the two shards' lists are the unsharded index's bit for bit (same rescoring arithmetic per row),
    and the packed exchange path fuses to the same answer."""
    import torch
    import openintel_amd as oi
    from openintel_amd import synth
    dev = torch.device("cuda:0")
    n, dim, B, depth, k = 300_000, 384, 24, 200, 50
    rows = synth.embeddings_torch(n, dim, dev)
    qv, qt, qo = synth.query_batch_torch(B, dim, dev, vocab=4096)
    terms, offs = synth.forward_index_torch(n, dev, vocab=4096)
    one = oi.HybridIndex(ctx, n, dim, 4096)
    one.set_embeddings(rows, normalize=False)
    one.set_forward(terms, offs)
    one.finalize()
    L1 = one.search_lists(qv, qt, qo, depth=depth)
    R1 = one.search(qv, qt, qo, k=k, depth=depth)
    ctx.synchronize()
    assert _gate(ctx) == 0.0
    half = 140_000
    shards, lists, tot, dfs = [], [], 0, []
    for lo, hi in ((0, half), (half, n)):
        t_lo, t_hi = int(offs[lo]), int(offs[hi])
        ix = oi.HybridIndex(ctx, hi - lo, dim, 4096, doc_id_base=lo)
        ix.set_embeddings(rows[lo:hi], normalize=False)
        ix.set_forward(terms[t_lo:t_hi].contiguous(), (offs[lo:hi + 1] - offs[lo]).contiguous())
        t, df = ix.local_stats()
        tot += t; dfs.append(df); shards.append(ix)
    gdf = (dfs[0].astype(np.uint64) + dfs[1]).astype(np.uint32)
    for ix in shards:
        ix.finalize(n, tot, gdf)
        lists.append(ix.search_lists(qv, qt, qo, depth=depth))
        assert _gate(ctx) == 0.0
    ctx.synchronize()
    st = lambda f: torch.stack([getattr(l, f) for l in lists])
    ms, md, mc = oi.merge_lists(ctx, st("cos_scores"), st("cos_docs"), st("cos_counts"))
    ctx.synchronize()
    assert torch.equal(md, L1.cos_docs) and torch.equal(ms, L1.cos_scores) and torch.equal(mc, L1.cos_counts)
    packed = torch.cat([ix.search_lists_packed(qv, qt, qo, depth=depth) for ix in shards])
    Rp = oi.fuse_packed(ctx, packed, 2, B, depth, k)
    ctx.synchronize()
    assert torch.equal(Rp.docs, R1.docs) and torch.equal(Rp.scores, R1.scores) and torch.equal(Rp.counts, R1.counts)
    for ix in shards + [one]:
        ix.close()


def test_unsupported_shapes_use_the_exact_kernels(ctx, O):
    from openintel_amd import synth
    rng = np.random.default_rng(1)
    for n, dim, B in ((3000, 1024, 12), (5000, 128, 20), (4000, 768, 3)):
        rows = synth.embeddings_np(n, dim, seed=2)
        q = synth.embeddings_np(B, dim, seed=4)
        terms, offs = _forward(rng, n)
        idx = _index(ctx, rows, terms, offs, 50)
        L = idx.search_lists(q, np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32), depth=50)
        for b in range(B):
            _check(L, b, O.dot_scores(rows, q[b]), 50, n)
        idx.close()


def test_full_size_screened_10M_768_batch64(ctx):
    """BASELINE configs[2] in screen mode: size-independent properties at full size -- planted copies of the
    queries come back first with score ~1, full sorted lists, the exact mode's lists on the same index (apart
    from near-ties), idempotence, and three queries checked against dense torch scores."""
    import torch
    import openintel_amd as oi
    from openintel_amd import _lib, synth
    dev = torch.device("cuda:0")
    n, dim, B, depth, k = 10_000_000, 768, 64, 1000, 100
    rows = synth.embeddings_torch(n, dim, dev)
    qv, qt, qo = synth.query_batch_torch(B, dim, dev, vocab=4096)
    plant = torch.arange(B, device=dev) * (n // B) + 17
    rows[plant] = qv
    terms, offs = synth.forward_index_torch(n, dev, vocab=4096)
    idx = oi.HybridIndex(ctx, n, dim, 4096)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    idx.finalize()
    del terms, offs
    L = idx.search_lists(qv, qt, qo, depth=depth)
    ctx.synchronize()
    assert _gate(ctx) == 0.0
    cs, cd, cc = L.cos_scores.cpu().numpy(), L.cos_docs.cpu().numpy(), L.cos_counts.cpu().numpy()
    assert (cc == depth).all() and cd.min() >= 0 and cd.max() < n
    assert np.array_equal(cd[:, 0], plant.cpu().numpy()) and np.abs(cs[:, 0] - 1.0).max() < 1e-5
    assert (np.diff(cs, axis=1) <= 0).all()
    R1 = idx.search(qv, qt, qo, k=k, depth=depth)
    R2 = idx.search(qv, qt, qo, k=k, depth=depth)
    ctx.synchronize()
    assert torch.equal(R1.docs, R2.docs) and torch.equal(R1.scores, R2.scores)
    ctx.set_cosine_mode(_lib.OI_COSINE_EXACT)
    Le = idx.search_lists(qv, qt, qo, depth=depth)
    Re = idx.search(qv, qt, qo, k=k, depth=depth)
    ctx.synchronize()
    ctx.set_cosine_mode(ctx.screen_mode)
    es, ed = Le.cos_scores.cpu().numpy(), Le.cos_docs.cpu().numpy()
    # same score at every rank (two f32 summation orders), same doc except where neighbours are closer than that
    assert np.abs(cs - es).max() <= 5e-7 and (cd == ed).mean() > 0.995
    assert all(np.setxor1d(cd[b], ed[b]).size <= 4 for b in range(B))
    assert (R1.docs == Re.docs).float().mean().item() > 0.98
    for b in range(3):
        full = (rows @ qv[b]).cpu().numpy().astype(np.float64)
        _check(L_np(cs, cd, cc), b, full, depth, n)
    idx.close()


class L_np:
    def __init__(self, s, d, c):
        self.cos_scores, self.cos_docs, self.cos_counts = s, d, c


# ---------------------------------------------------------------------------------------------------------------
# The screening copy (round 5: the default whenever the index holds one): the screen reads bf16(rows) made once.  Same products,
# same bound, same exact rescoring from the f32 rows: the lists must be the f32-stream screen's BIT FOR BIT, the adversary included.
@pytest.mark.parametrize("B,dim,n,depth", [(64, 768, 120_000, 1000), (40, 384, 50_000, 100), (130, 768, 30_000, 500)])
def test_screen_copy_mode_returns_the_same_lists(ctx, O, B, dim, n, depth):
    from openintel_amd import _lib, synth
    rows = synth.embeddings_np(n, dim, seed=90 + B)
    q = synth.embeddings_np(B, dim, seed=91 + B)
    rng = np.random.default_rng(B)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50, base=11)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    assert idx.index_bytes()[1] >= 2 * n * dim                  # finalize made the copy (AUTO: it fits the budget)
    ctx.set_cosine_mode(_lib.OI_COSINE_SCREEN_STREAM)            # the f32 rows converted on the fly: rounds 1-4's default
    L0 = idx.search_lists(q, qt, qo, depth=depth)
    assert _gate(ctx) == 0.0
    ctx.set_cosine_mode(_lib.OI_COSINE_SCREEN)                   # the default: streams the copy
    try:
        L1 = idx.search_lists(q, qt, qo, depth=depth)
        assert _gate(ctx) == 0.0
        idx.set_screen_copy(idx.SCREEN_COPY_NEVER)               # the copy is freed: the default mode streams the f32 rows
        assert idx.index_bytes()[1] == 0
        L3 = idx.search_lists(q, qt, qo, depth=depth)
        ctx.set_cosine_mode(_lib.OI_COSINE_SCREEN_COPY)          # rounds 2-4's opt-in mode: makes a missing copy on first use
        L2 = idx.search_lists(q, qt, qo, depth=depth)
        assert idx.index_bytes()[1] >= 2 * n * dim
        idx.set_screen_copy(idx.SCREEN_COPY_ALWAYS)
        L4 = idx.search_lists(q, qt, qo, depth=depth)
    finally:
        ctx.set_cosine_mode(ctx.screen_mode)
    for L in (L1, L2, L3, L4):
        assert np.array_equal(L.cos_counts, L0.cos_counts) and np.array_equal(L.cos_docs, L0.cos_docs)
        assert np.array_equal(L.cos_scores.view(np.uint32), L0.cos_scores.view(np.uint32))
    for b in range(0, B, 7):
        _check(L1, b, O.dot_scores(rows, q[b]), depth, n, base=11)
    idx.close()


def test_screen_copy_mode_on_the_tie_rounding_adversary(ctx, O):
    from openintel_amd import _lib, synth
    rng = np.random.default_rng(4)
    dim, B, n, depth, n_comp = 768, 16, 20_000, 100, 150
    rows = (synth.embeddings_np(n, dim, seed=40) * np.float32(0.3)).astype(np.float32)
    q = synth.embeddings_np(B, dim, seed=41)
    q[0, :384], q[0, 384:] = _tie(2), _tie(1)
    planted = 777
    comp = np.sort(rng.choice(np.setdiff1d(np.arange(n), [planted]), size=n_comp, replace=False))
    rows[planted, :384], rows[planted, 384:] = _tie(2), 0.0
    rows[comp, :384], rows[comp, 384:] = 0.0, _tie(3)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    ctx.set_cosine_mode(_lib.OI_COSINE_EXACT)
    Le = idx.search_lists(q, qt, qo, depth=depth)
    ctx.set_cosine_mode(_lib.OI_COSINE_SCREEN_COPY)
    try:
        L = idx.search_lists(q, qt, qo, depth=depth)
    finally:
        ctx.set_cosine_mode(ctx.screen_mode)
    assert int(L.cos_docs[0][0]) == planted
    assert np.array_equal(L.cos_docs[0], Le.cos_docs[0])
    assert np.array_equal(L.cos_scores[0].view(np.uint32), Le.cos_scores[0].view(np.uint32))
    for b in range(B):
        _check(L, b, O.dot_scores(rows, q[b]), depth, n)
    idx.close()


def test_fuzz_small_batches_through_the_screen(ctx, O):
    """Round 5: batches of 1..8 queries take the screen when the index holds a copy (module param "copy"; the f32 GEMV in "stream").
    Twenty-four random configurations -- corpus sizes around the screen's tile (32 rows), its first chunk (28 672 rows at depth
    <= 896) and odd ones, both screened dims, depths 1..1000, a zero query, duplicated rows (exact ties at the list's end), a nonzero
    doc-id base -- against the f64 oracle at the exact kernel's bar, and in "copy" mode the lists of the exact mode on the same index."""
    from openintel_amd import _lib, synth
    rng = np.random.default_rng(505)
    for case in range(24):
        n = int(rng.choice([1, 31, 33, 1000, 28_671, 28_672, 28_673, 40_000, 120_001]))
        dim = int(rng.choice([384, 768]))
        B = int(rng.integers(1, 9))
        depth = int(rng.choice([1, 10, 100, 1000]))
        base = int(rng.choice([0, 77, 4_000_000_000]))
        rows = synth.embeddings_np(n, dim, seed=1000 + case)
        if n > 64 and case % 3 == 0:
            rows[n // 2:n // 2 + 20] = rows[5]              # 21 identical rows: ties wherever row 5 ranks
        q = synth.embeddings_np(B, dim, seed=2000 + case)
        if case % 4 == 1:
            q[0] = 0.0                                      # a zero query: every score 0, doc-id order
        if n > 64:
            q[B - 1] = rows[5]                              # the tied rows at the top of this query's list
        terms, offs = _forward(rng, n)
        idx = _index(ctx, rows, terms, offs, 50, base=base)
        qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
        L = idx.search_lists(q, qt, qo, depth=depth)
        g = _gate(ctx)
        if not _screened(ctx, B):
            assert g == -1.0, (case, n, dim, B, depth, g)
        elif case % 4 == 1:
            assert g in (0.0, 1.0)   # (a zero query ties with every row: whether its survivors fit decides; the lists are checked either way)
        else:
            assert g == 0.0, (case, n, dim, B, depth, g)
        for b in range(B):
            _check(L, b, O.dot_scores(rows, q[b]), depth, n, base=base)
        if _screened(ctx, B):
            ctx.set_cosine_mode(_lib.OI_COSINE_EXACT)
            Le = idx.search_lists(q, qt, qo, depth=depth)
            ctx.set_cosine_mode(ctx.screen_mode)
            assert np.array_equal(L.cos_counts, Le.cos_counts), (case, n, dim, B, depth)
            for b in range(B):
                c = int(L.cos_counts[b])
                assert np.abs(L.cos_scores[b][:c] - Le.cos_scores[b][:c]).max(initial=0.0) <= 5e-7, (case, b)
        idx.close()


@pytest.mark.parametrize("n,dim,B,depth", [(300_000, 768, 64, 1000), (300_000, 768, 64, 100), (120_001, 384, 40, 10), (700_000, 768, 3, 100)])
def test_speculative_thresholds_leave_the_lists_alone(ctx, O, n, dim, B, depth):
    """Round 5: between chunks the screen may use a PREDICTED threshold (cosine_prefilter.hip, pf_spec_kernel), checked at the end
    against the proven one.  On a corpus in random order the check holds (no fallback) and the lists are those of the
    proven-threshold screen bit for bit -- both survivor sets hold the exact list, the scores are the same exact rescoring."""
    from openintel_amd import synth
    rows = synth.embeddings_np(n, dim, seed=91)
    q = synth.embeddings_np(B, dim, seed=92)
    rng = np.random.default_rng(9)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50, base=5)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    f0, s0 = ctx.speculation_state()
    L1 = idx.search_lists(q, qt, qo, depth=depth)
    g1 = _gate(ctx)
    f1, s1 = ctx.speculation_state()
    ctx.set_screen_speculation(False)
    L0 = idx.search_lists(q, qt, qo, depth=depth)
    g0 = _gate(ctx)
    ctx.set_screen_speculation(True)
    if _screened(ctx, B):
        assert g1 == 0.0 and g0 == 0.0
        # (depth 10: the rank 3 k' m / n + 12 is never within k' / 2 -- nothing to gain, no speculation)
        # (batches of <= 8 queries never speculate: nothing to gain there)
        assert s1 == s0 + (1 if depth >= 100 and B > 8 else 0) and f1 == f0, "the search must have speculated, and its check must have held"
    assert np.array_equal(L1.cos_counts, L0.cos_counts)
    assert np.array_equal(L1.cos_docs, L0.cos_docs) and np.array_equal(L1.cos_scores.view(np.uint32), L0.cos_scores.view(np.uint32))
    for b in range(min(B, 4)):
        _check(L1, b, O.dot_scores(rows, q[b]), depth, n, base=5)
    idx.close()


def test_a_failed_speculation_opens_the_gate_and_backs_off(ctx, O):
    """A corpus whose FIRST rows are not a fair sample: 60 near-copies of every query sit in the first chunk, none after it.  The
    predicted threshold (the 3 k' m / n + 12 = 15th best of the first 8 192 rows at k' = 100, n = 1M) then sits near 1 while only 60
    rows of the million reach it: the check at the end fails, the gate opens, the exact pipeline delivers the lists -- still the
    exact scorer's -- and the ctx stops speculating for the next 16 searches (which are screened with proven thresholds, no gate)."""
    from openintel_amd import synth
    n, dim, B, depth = 1_000_000, 768, 16, 100
    if not _screened(ctx, B):
        pytest.skip("not reached")
    rows = synth.embeddings_np(n, dim, seed=93)
    q = synth.embeddings_np(B, dim, seed=94)
    rng = np.random.default_rng(10)
    for b in range(B):                                  # rows 100 + 60 b .. 100 + 60 b + 59: query b plus a little noise, renormalised
        blk = q[b][None, :] + 0.02 * rng.standard_normal((60, dim)).astype(np.float32) / np.sqrt(dim)
        rows[100 + 60 * b:160 + 60 * b] = blk / np.linalg.norm(blk, axis=1, keepdims=True)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, rows, terms, offs, 50)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    ctx.set_screen_speculation(True)                    # (also clears any back-off left by an earlier test)
    f0, s0 = ctx.speculation_state()
    L = idx.search_lists(q, qt, qo, depth=depth)
    assert _gate(ctx) != 0.0, "the speculation must have failed its check"
    refs = [O.dot_scores(rows, q[b]) for b in range(4)]
    for b in range(4):
        _check(L, b, refs[b], depth, n)
        assert set(range(100 + 60 * b, 160 + 60 * b)) <= set(L.cos_docs[b][:depth].tolist())
    f1, s1 = ctx.speculation_state()
    assert f1 == f0 + 1 and s1 == s0 + 1
    for i in range(3):                                  # backed off: proven thresholds, no fallback, the same lists
        L2 = idx.search_lists(q, qt, qo, depth=depth)
        assert _gate(ctx) == 0.0
        assert np.array_equal(L2.cos_docs, L.cos_docs) or all(
            np.abs(L2.cos_scores[b] - L.cos_scores[b]).max() <= 5e-7 for b in range(B))
    f2, s2 = ctx.speculation_state()
    assert f2 == f1 and s2 == s1, "no speculation while backed off"
    ctx.set_screen_speculation(True)
    idx.close()
