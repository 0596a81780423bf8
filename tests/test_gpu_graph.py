"""oi_set_graph_replay: repeated device-buffer query calls are captured into hipGraphs and replayed with one launch.  A replay
must do exactly what the eager call does: the same lists bit for bit while the CONTENTS of the (same) buffers change from
call to call, across a workspace reallocation in between, for the screened and the exact scorer, with and without the
overlapped BM25 leg, and for oi_fuse_packed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(n=120_000, dim=384, vocab=500, seed=3):
    rng = np.random.default_rng(seed)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    lens = rng.integers(1, 14, size=n)
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    terms = rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32)
    return rows, terms, offs


def _batch(rng, B, dim, vocab):
    q = rng.standard_normal((B, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    nt = rng.integers(1, 6, size=B)
    qo = np.zeros(B + 1, np.int32)
    qo[1:] = np.cumsum(nt)
    qt = rng.integers(0, vocab, size=int(qo[-1])).astype(np.int32)
    return q, qt, qo


@pytest.mark.parametrize("mode", ["screen", "exact"])
def test_replayed_calls_equal_eager_calls_bit_for_bit(mode):
    import torch
    import openintel_amd as oi
    from openintel_amd import _lib
    dev = torch.device("cuda:0")
    rows, terms, offs = _data()
    n, dim = rows.shape
    B, DEPTH, K = 64, 300, 40
    d_rows = torch.from_numpy(rows).to(dev)

    def make(graphs):
        c = oi.HipContext(0)
        c.set_stream(torch.cuda.Stream(device=dev))          # replay needs a real stream (not the default one)
        c.set_cosine_mode(_lib.OI_COSINE_SCREEN if mode == "screen" else _lib.OI_COSINE_EXACT)
        c.set_graph_replay(graphs)
        ix = oi.HybridIndex(c, n, dim, 500, doc_id_base=7)
        ix.set_embeddings(d_rows, normalize=False)
        ix.set_forward(terms, offs)
        ix.finalize()
        return c, ix

    cg, ig = make(True)
    ce, ie = make(False)
    rng = np.random.default_rng(1)
    # the SAME device buffers every call; their contents change
    qv = torch.zeros((B, dim), dtype=torch.float32, device=dev)
    qt = torch.zeros(B * 6, dtype=torch.int32, device=dev)
    qo = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    out_g = oi.SearchResult(torch.zeros((B, K), dtype=torch.float32, device=dev), torch.zeros((B, K), dtype=torch.int32, device=dev),
                            torch.zeros(B, dtype=torch.int32, device=dev))
    packed_g = torch.zeros(oi.packed_words(B, DEPTH), dtype=torch.int32, device=dev)
    for it in range(9):
        q, t, o = _batch(rng, B, dim, 500)
        qv.copy_(torch.from_numpy(q)); qt[:t.size].copy_(torch.from_numpy(t)); qo.copy_(torch.from_numpy(o))
        torch.cuda.synchronize()
        if it == 3:            # another shape in between moves workspaces (bigger batch): the captured calls must notice
            big = ig.search(torch.cat([qv, qv, qv]), torch.cat([qt, qt, qt]), torch.cat([qo, qo[1:] + qo[-1], qo[1:] + 2 * qo[-1]]),
                            k=K, depth=DEPTH)
            cg.synchronize()
            assert int(big.counts.min()) > 0
        if it == 5:
            cg.set_overlap(False); ce.set_overlap(False)      # a mode change is part of the key
        ig.search(qv, qt, qo, k=K, depth=DEPTH, out=out_g)
        ig.search_lists_packed(qv, qt, qo, depth=DEPTH, out=packed_g)
        cg.synchronize()
        want = ie.search(qv, qt, qo, k=K, depth=DEPTH)
        want_p = ie.search_lists_packed(qv, qt, qo, depth=DEPTH)
        ce.synchronize()
        assert torch.equal(out_g.docs, want.docs) and torch.equal(out_g.scores, want.scores) and torch.equal(out_g.counts, want.counts), it
        cnt = packed_g[4 * B * DEPTH:].view(2, B)
        assert torch.equal(cnt, want_p[4 * B * DEPTH:].view(2, B))
        a, b = oi.unpack_lists(packed_g, B, DEPTH), oi.unpack_lists(want_p, B, DEPTH)
        for qq in range(B):            # entries past a list's count are not defined: compare the valid prefix
            nc, nb = int(a.cos_counts[qq]), int(a.bm25_counts[qq])
            assert torch.equal(a.cos_docs[qq, :nc], b.cos_docs[qq, :nc]) and torch.equal(a.cos_scores[qq, :nc], b.cos_scores[qq, :nc])
            assert torch.equal(a.bm25_docs[qq, :nb], b.bm25_docs[qq, :nb]) and torch.equal(a.bm25_scores[qq, :nb], b.bm25_scores[qq, :nb])
        # the fusion call too: same buffers, new contents
        fused = oi.fuse_packed(cg, packed_g, 1, B, DEPTH, K, out=out_g)
        cg.synchronize()
        assert torch.equal(fused.docs, want.docs) and torch.equal(fused.scores, want.scores)
    replays, captures = cg.graph_stats()
    assert captures >= 3 and replays >= 3, (replays, captures)
    assert ce.graph_stats() == (0, 0)
    ig.close(); ie.close(); cg.close(); ce.close()


def test_pipeline_with_staged_batches_and_graph_replay_returns_the_same_results():
    """ShardedPipeline(graphs=True): every batch is copied into its slot's staging buffers and the slot's two C calls are
    replayed as captured graphs.  Different query tensors every submit (fresh allocations, varying term counts): the
    results must equal the plain search of the same batch, and replays must actually have happened."""
    import torch
    import openintel_amd as oi
    from openintel_amd import sharded
    dev = torch.device("cuda:0")
    rows, terms, offs = _data(n=80_000)
    n, dim = rows.shape
    B, DEPTH, K = 64, 200, 30
    ctx = oi.HipContext(0)
    ctx.use_torch_current_stream()
    idx = oi.HybridIndex(ctx, n, dim, 500)
    idx.set_embeddings(torch.from_numpy(rows).to(dev), normalize=False)
    idx.set_forward(terms, offs)
    sr = sharded.make_hip_sharded(ctx, idx, dev)
    sr.finalize()
    ref_ctx = oi.HipContext(0)
    ref = idx.view(ref_ctx)                      # eager reference through a view (its own workspaces)
    fctx = oi.HipContext(0)
    lane = oi.HipContext(0)
    pipe = sharded.ShardedPipeline(sr, fctx, B, DEPTH, K, lane_ctxs=[lane], graphs=True)
    rng = np.random.default_rng(9)
    outs, wants = [], []
    for it in range(22):                             # (six slots: a slot's third use is its first replay)
        q, t, o = _batch(rng, B, dim, 500)
        dq = (torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(o).to(dev))
        slot = pipe.submit(*dq)
        del dq                                   # the inputs may go right after submit(): staged + record_stream
        pipe.wait(slot)
        r = pipe.results[slot]
        outs.append((r.scores.clone(), r.docs.clone(), r.counts.clone()))
        w = ref.search(torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(o).to(dev), k=K, depth=DEPTH)
        ref_ctx.synchronize()
        wants.append((w.scores.clone(), w.docs.clone(), w.counts.clone()))
    pipe.drain()
    torch.cuda.synchronize()
    for i, ((s, d, c), (ws, wd, wc)) in enumerate(zip(outs, wants)):
        assert torch.equal(c, wc) and torch.equal(d, wd) and torch.equal(s, ws), i
    # (lane 0 scores on a context the pipeline made for it: the retriever's own `ctx` replays nothing)
    replays = pipe.lane0_ctx.graph_stats()[0] + lane.graph_stats()[0] + fctx.graph_stats()[0]
    assert ctx.graph_stats()[0] == 0
    assert replays >= 6, replays
    pipe.close()
    ref.close(); ref_ctx.close(); fctx.close(); idx.close(); ctx.close()


def test_new_rows_between_replays_are_seen_by_the_replayed_call():
    """a captured call bakes in the index's rows pointer.  oi_index_set_embeddings with ANOTHER buffer (a caller's
    pointer: no workspace moves) must invalidate the capture -- the next call with the same arguments scores the new rows."""
    import torch
    import openintel_amd as oi
    from openintel_amd import _lib
    dev = torch.device("cuda:0")
    rows, terms, offs = _data(n=60_000)
    n, dim = rows.shape
    B, DEPTH, K = 64, 200, 30
    rows_a = torch.from_numpy(rows).to(dev)
    rows_b = torch.from_numpy(np.ascontiguousarray(rows[::-1])).to(dev)    # the same rows in reverse order: other docs win
    c = oi.HipContext(0)
    c.set_stream(torch.cuda.Stream(device=dev))
    c.set_cosine_mode(_lib.OI_COSINE_EXACT)
    c.set_graph_replay(True)
    ix = oi.HybridIndex(c, n, dim, 500)
    ix.set_embeddings(rows_a, normalize=False)
    ix.set_forward(terms, offs)
    ix.finalize()
    rng = np.random.default_rng(5)
    q, t, o = _batch(rng, B, dim, 500)
    qv, qt, qo = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(o).to(dev)
    out = oi.SearchResult(torch.zeros((B, K), dtype=torch.float32, device=dev), torch.zeros((B, K), dtype=torch.int32, device=dev),
                          torch.zeros(B, dtype=torch.int32, device=dev))
    lists_a = None
    for _ in range(3):                                     # eager, capture, replay
        lists_a = ix.search_lists(qv, qt, qo, depth=DEPTH)
        ix.search(qv, qt, qo, k=K, depth=DEPTH, out=out)
    c.synchronize()
    cos_a = lists_a.cos_docs.clone()
    ix.set_embeddings(rows_b, normalize=False)            # a caller's pointer again: nothing of the library's moves
    lists_b = ix.search_lists(qv, qt, qo, depth=DEPTH)
    c.synchronize()
    # row r of the new matrix is row n-1-r of the old one: the cosine list is the old one mirrored
    assert torch.equal(lists_b.cos_docs[:, 0], (n - 1) - cos_a[:, 0])
    assert torch.equal(lists_b.cos_scores[:, 0], lists_a.cos_scores[:, 0])
    ix.close(); c.close()
