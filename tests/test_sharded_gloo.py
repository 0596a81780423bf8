"""Multi-rank path on CPU: world_size 2 over gloo.  The collective choreography of
openintel_amd.sharded.ShardedRetriever (global df/N/token all-reduce at build, ONE packed all-gather
of the per-shard lists per batch, merge to global ranks, THEN fuse) is exercised with the CPU oracle
standing in for the per-GPU engine and for the merge / RRF kernels; the result of every rank must be
bit-identical to the unsharded oracle answer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lib as O
from openintel_amd.sharded import ShardedRetriever, shard_bounds
from openintel_amd.retriever import RankedLists

N, DIM, VOCAB, B, DEPTH, K = 3000, 16, 50, 5, 40, 10


def _corpus():
    rng = np.random.default_rng(123)
    rows = rng.integers(-3, 4, size=(N, DIM)).astype(np.float32)   # exact dot products
    lens = rng.integers(1, 10, size=N)
    offs = np.zeros(N + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    terms = rng.integers(0, VOCAB, size=int(offs[-1])).astype(np.uint32)
    q = rng.integers(-3, 4, size=(B, DIM)).astype(np.float32)
    qt = rng.integers(0, 12, size=B * 3).astype(np.uint32)
    qo = (np.arange(B + 1) * 3).astype(np.uint32)
    return rows, terms, offs, q, qt, qo


class OracleShard:
    """Stands in for HybridIndex on one rank: same interface, CPU oracle inside."""

    def __init__(self, rows, terms, offs, base):
        self.rows, self.terms, self.offs, self.base = rows, terms, offs, base
        self.n_docs, self.vocab = rows.shape[0], VOCAB

    def local_stats(self):
        df, tot = O.bm25_df(self.terms, self.offs, VOCAB)
        return tot, df

    def finalize(self, n_global, tok_global, df_global):
        self.n_global, self.tok_global, self.df_global = n_global, tok_global, df_global

    def search_lists(self, qv, qt, qo, depth):
        qv, qt, qo = qv.numpy(), qt.numpy().astype(np.uint32), qo.numpy().astype(np.uint32)
        nb = qv.shape[0]
        out = [np.zeros((nb, depth), np.float32), np.zeros((nb, depth), np.int32), np.zeros(nb, np.int32),
               np.zeros((nb, depth), np.float32), np.zeros((nb, depth), np.int32), np.zeros(nb, np.int32)]
        for b in range(nb):
            cs, cd = O.topk(O.dot_scores(self.rows, qv[b]), depth, False, self.base)
            bs, bd = O.topk(O.bm25_scores(self.terms, self.offs, VOCAB, qt[qo[b]:qo[b + 1]], df=self.df_global,
                                          n_docs_global=self.n_global, total_tokens_global=self.tok_global),
                            depth, True, self.base)
            out[0][b, :cs.size], out[1][b, :cd.size], out[2][b] = cs, cd, cs.size
            out[3][b, :bs.size], out[4][b, :bd.size], out[5][b] = bs, bd, bs.size
        return RankedLists(*[torch.from_numpy(x) for x in out])


def _search_lists_packed(self, qv, qt, qo, depth):
    L = self.search_lists(qv, qt, qo, depth)
    sc = torch.stack([L.cos_scores, L.bm25_scores]).contiguous().view(torch.int32).reshape(-1)
    dc = torch.stack([L.cos_docs, L.bm25_docs]).reshape(-1)
    cn = torch.stack([L.cos_counts, L.bm25_counts]).reshape(-1)
    return torch.cat([sc, dc, cn])           # the OI_PACKED_WORDS layout


class OraclePackedShard(OracleShard):
    search_lists_packed = _search_lists_packed


def _fuse_packed(flat, n_shards, nb, depth, k):
    from openintel_amd.retriever import packed_words, unpack_lists
    W = packed_words(nb, depth)
    shards = [unpack_lists(flat[s * W:(s + 1) * W], nb, depth) for s in range(n_shards)]
    st = lambda f: torch.stack([getattr(x, f) for x in shards])
    _, cd, cc = _merge(st("cos_scores"), st("cos_docs"), st("cos_counts"))
    _, bd, bc = _merge(st("bm25_scores"), st("bm25_docs"), st("bm25_counts"))
    return _fuse(cd, cc, bd, bc, k)


def _merge(scores, docs, counts):
    S, nb, depth = scores.shape
    so, do, co = np.zeros((nb, depth), np.float32), np.zeros((nb, depth), np.int32), np.zeros(nb, np.int32)
    for b in range(nb):
        ms, md = O.merge_ranked([scores[s, b, :counts[s, b]].numpy() for s in range(S)],
                                [docs[s, b, :counts[s, b]].numpy().astype(np.uint32) for s in range(S)], depth)
        so[b, :ms.size], do[b, :md.size], co[b] = ms, md, md.size
    return torch.from_numpy(so), torch.from_numpy(do), torch.from_numpy(co)


def _fuse(cd, cc, bd, bc, k):
    nb = cd.shape[0]
    fs, fd, fc = np.zeros((nb, k), np.float32), np.zeros((nb, k), np.int32), np.zeros(nb, np.int32)
    for b in range(nb):
        s, d = O.rrf_fuse(cd[b, :cc[b]].numpy().astype(np.uint32), bd[b, :bc[b]].numpy().astype(np.uint32), k)
        fs[b, :s.size], fd[b, :d.size], fc[b] = s, d, d.size
    return torch.from_numpy(fs), torch.from_numpy(fd), torch.from_numpy(fc)


def _worker(rank, world, port, ret, packed=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows, terms, offs, q, qt, qo = _corpus()
        lo, hi = shard_bounds(N, world, rank)
        t_lo, t_hi = int(offs[lo]), int(offs[hi])
        cls = OraclePackedShard if packed else OracleShard
        shard = cls(rows[lo:hi], terms[t_lo:t_hi], (offs[lo:hi + 1] - offs[lo]).astype(np.uint64), lo)
        sr = ShardedRetriever(shard, torch.device("cpu"), _merge, _fuse, fuse_packed=_fuse_packed if packed else None)
        assert (sr.fuse_packed is not None) == packed
        sr.finalize()
        assert shard.n_global == N and shard.tok_global == int(offs[-1])
        s, d, c = sr.search(torch.from_numpy(q), torch.from_numpy(qt.astype(np.int32)),
                            torch.from_numpy(qo.astype(np.int32)), K, DEPTH)
        ret[rank] = (s.numpy().copy(), d.numpy().copy(), c.numpy().copy(), shard.df_global.copy())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_cover_and_align():
    for n, w in ((10_000_000, 8), (1001, 3), (29, 8), (12_500_000, 8), (1, 1)):
        edges = [shard_bounds(n, w, r) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        assert all(lo % 4 == 0 and hi > lo for lo, hi in edges)


def test_shard_bounds_refuse_empty_shards_on_every_rank():
    """10 rows over 8 ranks in 4-row blocks would leave ranks 3..7 empty (an index shard needs a row): every rank
    must fail the same way BEFORE any collective, or the others hang in all_reduce."""
    for n, w in ((10, 8), (7, 8), (0, 2), (4, 2)):
        for r in range(w):
            with pytest.raises(ValueError):
                shard_bounds(n, w, r)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,packed", [(2, False), (2, True), (8, True)], ids=["2-lists", "2-packed-exchange", "8-packed-exchange"])
def test_two_ranks_match_unsharded_oracle(world, packed):
    """world 8: the shape the driver's N = 8 run has -- eight row shards from shard_bounds, the df / N / token
    all-reduce over eight ranks, ONE all-gather of eight packed lists per batch, merge THEN fuse -- with the oracle standing in
    for the engines (the GPU box admits six processes on its card: eight ranks can only meet on the CPU)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, packed), nprocs=world, join=True)
    rows, terms, offs, q, qt, qo = _corpus()
    df, _ = O.bm25_df(terms, offs, VOCAB)
    for rank in range(world):
        s, d, c, gdf = ret[rank]
        assert np.array_equal(gdf, df)                       # the df all-reduce
        for b in range(B):
            ref = O.hybrid_search(rows, terms, offs, VOCAB, q[b], qt[qo[b]:qo[b + 1]], K, DEPTH)
            fs, fd = ref["fused"]
            assert c[b] == fd.size
            assert np.array_equal(d[b, :fd.size].astype(np.uint32), fd)
            assert np.array_equal(s[b, :fs.size].view(np.uint32), fs.view(np.uint32))
    for rank in range(1, world):
        assert np.array_equal(ret[0][1], ret[rank][1])       # every rank holds the same answer


# ------------------------------------------------------------------ the sharded lexicon path
def _posts(n, seed):
    rng = np.random.default_rng(seed)
    words = ["moon", "buy", "calls", "puts", "crash", "dump", "squeeze", "yolo", "hold", "the", "a", "stock", "AAPL",
             "bullish", "bearish", "rocket", "short", "long", "sell", "tendies", "bull", "bear", "up", "down"]
    texts = [" ".join(rng.choice(words, size=rng.integers(1, 25))) for _ in range(n)]
    src = rng.integers(0, 2, size=n).astype(np.uint8)
    return texts, src


def _oracle_counters(texts, src):
    blob, offs = O.pack_texts(texts)
    pol, spec = O.lexicon_analyze(blob, offs)
    return O.social_summary(src, pol, spec), pol, spec


class _Counters:
    def __init__(self, s):
        self.total, self.bullish, self.bearish, self.neutral = s.total_mentions, s.bullish, s.bearish, s.neutral
        self.spec_count, self.polarity_sum = s.spec_count, s.polarity_sum
        self.by_source = list(s.mentions_by_source)


def _analyzer_worker(rank, world, port, ret, n):
    from openintel_amd.sharded import ShardedAnalyzer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        texts, src = _posts(n, 77)
        lo, hi = rank * n // world, (rank + 1) * n // world      # posts shard anywhere (no alignment rule)
        sa = ShardedAnalyzer(lambda t, s: _Counters(_oracle_counters(t, s)[0]), torch.device("cpu"))
        g = sa.summary(texts[lo:hi], src[lo:hi])
        ret[rank] = (g.total, g.by_source[0], g.by_source[1], g.bullish, g.bearish, g.neutral, g.spec_count,
                     g.polarity_sum)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("n", [10, 4001])
def test_two_ranks_lexicon_summary_matches_unsharded_oracle(n, golden):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_analyzer_worker, args=(world, _free_port(), ret, n), nprocs=world, join=True)
    texts, src = _posts(n, 77)
    ref, pol, _ = _oracle_counters(texts, src)
    assert ret[0] == ret[1], "every rank must hold the same global counters, bit for bit"
    g = ret[0]
    assert g[:7] == (ref.total_mentions, ref.mentions_by_source[0], ref.mentions_by_source[1], ref.bullish, ref.bearish,
                     ref.neutral, ref.spec_count)                                   # integer fields: exact
    half = _oracle_counters(texts[:n // 2], src[:n // 2])[0].polarity_sum
    rest = _oracle_counters(texts[n // 2:], src[n // 2:])[0].polarity_sum
    assert g[7] == half + rest                                                      # rank-order sum of the partials
    assert abs(g[7] - ref.polarity_sum) <= n * 2.0 ** -52 * max(1.0, float(np.abs(np.cumsum(pol)).max()))


# ------------------------------------------------------------------ the batch callers over ranks: whole tickers per rank
def _ticker_cuts(n, n_tickers, seed):
    rng = np.random.default_rng(seed)
    cuts = np.sort(rng.integers(0, n + 1, n_tickers - 1))
    return np.concatenate([[0], cuts, [n]]).astype(np.uint64)   # ragged, some tickers empty


def _oracle_records(texts, src, seg):
    """8 int64 words per ticker, laid out like oi_social_counters (the f64's bits in the last word)."""
    _, pol, spec = _oracle_counters(texts, src)
    out = np.zeros((seg.size - 1, 8), dtype=np.int64)
    for k, o in enumerate(O.social_summary_segmented(src, pol, spec, seg)):
        out[k, :7] = (o.total_mentions, o.mentions_by_source[0], o.mentions_by_source[1], o.bullish, o.bearish, o.neutral,
                      o.spec_count)
        out[k, 7] = np.array([o.polarity_sum], dtype=np.float64).view(np.int64)[0]
    return out


def _segments_worker(rank, world, port, ret, n, n_tickers):
    from openintel_amd.sharded import ShardedAnalyzer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        texts, src = _posts(n, 78)
        seg = _ticker_cuts(n, n_tickers, 5)
        lo, hi = ShardedAnalyzer.ticker_bounds(n_tickers, world, rank)
        p0, p1 = int(seg[lo]), int(seg[hi])                    # the rank's tickers' posts, pooled

        def scan(t, s_, sg):
            return torch.from_numpy(_oracle_records(t, s_, sg).reshape(-1))
        sa = ShardedAnalyzer(None, torch.device("cpu"), scan_segments_shard=scan)
        g = sa.segment_summaries(n_tickers, texts[p0:p1], src[p0:p1], seg[lo:hi + 1] - seg[lo])
        ret[rank] = g.tobytes()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("n,n_tickers", [(600, 7), (5000, 64), (3000, 101)])
def test_two_ranks_ticker_shards_equal_the_unsharded_per_ticker_sums(n, n_tickers):
    """Whole tickers per rank: every record, polarity_sum's bits included, equals the unsharded oracle's -- there is no
    reassociation to bound (unlike the one-report form above)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_segments_worker, args=(world, _free_port(), ret, n, n_tickers), nprocs=world, join=True)
    texts, src = _posts(n, 78)
    ref = _oracle_records(texts, src, _ticker_cuts(n, n_tickers, 5))
    assert ret[0] == ret[1] == ref.tobytes()



# ---------------------------------------------------------------------------------------------------------------------
# ShardedPipeline.calibrate(max_lanes >= 3) let ranks run different numbers of batches -- a rank whose
# second lane had not earned its 3 % ran no trials at the third level, one that kept it ran `placements` more -- and every
# batch is one all-gather: RCCL hangs, gloo mispairs.  The decision loop (sharded.calibrate_lanes) now times every level on
# every rank and discards what a stopped rank measures.  Here two gloo ranks DISAGREE (rank 0's second lane pays, rank 1's
# does not) with a period() that is a real collective: the counts must match, nobody may hang, and each keeps its own choice.
def _calibrate_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openintel_amd.sharded import calibrate_lanes
        state = {"lanes": None, "periods": 0, "made": 0, "dropped": 0}
        # scripted ms per (number of lanes): rank 0 gains 20 % from a second lane and nothing from a third; rank 1 gains nothing
        script = {0: {1: 1.00, 2: 0.80, 3: 0.79}, 1: {1: 1.00, 2: 0.99, 3: 0.60}}[rank]

        def period():
            state["periods"] += 1
            mine = torch.tensor([rank * 1000 + state["periods"]], dtype=torch.int64)
            allv = torch.zeros(world, dtype=torch.int64)
            dist.all_gather_into_tensor(allv, mine)                       # the collective every batch of the real period() runs
            assert [int(x) % 1000 for x in allv] == [state["periods"]] * world, "ranks are at different trials"
            return script[len(state["lanes"])]

        def make_lane():
            state["made"] += 1
            return ("lane", state["made"])

        def drop_lane(lane):
            state["dropped"] += 1

        def use(lanes):
            state["lanes"] = list(lanes)

        kept, best, tried = calibrate_lanes(("lane", 0), period, make_lane, drop_lane, use, placements=3, max_lanes=3)
        ret[rank] = {"kept": len(kept), "best": best, "periods": state["periods"], "made": state["made"], "dropped": state["dropped"],
                     "considered": [t.get("considered", True) for t in tried]}
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_calibrate_runs_the_same_number_of_collectives_on_ranks_that_disagree():
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_calibrate_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    assert a["periods"] == b["periods"] == 1 + 2 * 3                  # one lane, then 3 placements at each of two levels: on BOTH ranks
    assert a["kept"] == 2 and abs(a["best"] - 0.80) < 1e-12           # rank 0: the second lane pays, the third (0.79 vs 0.80: < 3 %) does not
    assert b["kept"] == 1 and abs(b["best"] - 1.00) < 1e-12           # rank 1: stopped at one lane -- its 0.60 at three lanes was timed and DISCARDED
    assert b["considered"] == [True, True, True, True, False, False, False]
    assert a["made"] == b["made"] == 6 and a["dropped"] == 5 and b["dropped"] == 6   # every rejected lane is closed
