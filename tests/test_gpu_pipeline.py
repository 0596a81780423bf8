"""The pipelined query behind the C ABI (oi_pipeline_*, csrc/pipeline.hip): several batches in flight
through lanes the LIBRARY owns (contexts, streams, views of the index, a fusing stream, event-ordered slots) -- no torch
streams, no torch.distributed.  Per batch it runs the same kernels on the same data as oi_search / oi_search_sharded, so every
result must equal the plain call's bit for bit: host and device buffers, 1-3 lanes, ragged batch sizes, more batches than
slots, with and without an RCCL communicator (of one rank: RCCL refuses two ranks on one device)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(n=60_001, dim=384, vocab=300, B=64, seed=21, n_batches=9):
    rng = np.random.default_rng(seed)
    from openintel_amd import synth
    rows = synth.embeddings_np(n, dim, seed=seed)
    lens = rng.integers(1, 12, size=n)
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    terms = rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32)
    batches = []
    for i in range(n_batches):
        b = B if i % 3 else max(9, B - 7 * i)          # ragged: some batches are smaller than max_queries
        q = synth.embeddings_np(b, dim, seed=seed + 100 + i)
        qo = (np.arange(b + 1) * 4).astype(np.uint32)
        qt = rng.integers(0, 60, size=4 * b).astype(np.uint32)
        batches.append((q, qt, qo))
    return rows, terms, offs, batches


def _index(ctx, rows, terms, offs, vocab=300, base=700):
    import openintel_amd as oi
    idx = oi.HybridIndex(ctx, rows.shape[0], rows.shape[1], vocab, doc_id_base=base)
    idx.set_embeddings(rows.copy(), normalize=False)
    idx.set_forward(terms, offs)
    idx.finalize()
    return idx


def _same(got, want):
    g = [np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x) for x in (got.scores, got.docs, got.counts)]
    w = [np.asarray(x.cpu().numpy() if hasattr(x, "cpu") else x) for x in (want.scores, want.docs, want.counts)]
    return (np.array_equal(g[2].view(np.uint32), w[2].view(np.uint32)) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32))
            and np.array_equal(g[0].view(np.uint32), w[0].view(np.uint32)))


@pytest.mark.parametrize("lanes", [1, 2, 3])
def test_pipeline_equals_oi_search_bit_for_bit_host_and_device(lanes):
    import torch
    import openintel_amd as oi
    K, DEPTH = 50, 200
    rows, terms, offs, batches = _case()
    ctx = oi.HipContext(0)
    idx = _index(ctx, rows, terms, offs)
    want = [idx.search(q, qt, qo, k=K, depth=DEPTH) for q, qt, qo in batches]
    pipe = oi.NativePipeline(idx, lanes=lanes, max_queries=64, max_query_terms=4, depth=DEPTH, k=K)
    assert pipe.workspace_bytes()[0] > 0
    conc, total = pipe.concurrent_streams()
    assert total == lanes + 1 and 1 <= conc <= total
    # (a) host buffers: submit everything (more batches than slots: early ones are delivered when their slot is reused), then wait
    sub = [pipe.submit(q, qt, qo) for q, qt, qo in batches]
    for t, _ in reversed(sub):
        pipe.wait(t)
    for (t, out), w in zip(sub, want):
        assert _same(out, w), "host batch %d" % t
    # (b) device buffers, distinct outputs per batch in flight; stream-ordered wait (no host sync) then one synchronise
    dev = torch.device("cuda:0")
    dbs = [[torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) for x in b] for b in batches]
    torch.cuda.synchronize()
    sub = [pipe.submit(*b) for b in dbs]
    for t, _ in sub:
        pipe.wait(t, host_sync=False)
    ctx.synchronize()
    for (t, out), w in zip(sub, want):
        assert _same(out, w), "device batch %d" % t
    # (c) interleaved submit / wait, twice round the slots
    for rnd in range(2):
        for i, b in enumerate(dbs):
            t, out = pipe.submit(*b)
            if i % 2:
                pipe.wait(t)
                assert _same(out, want[i])
    pipe.drain()
    # errors are loud
    from openintel_amd._lib import OiError
    with pytest.raises(OiError):
        pipe.submit(np.zeros((65, rows.shape[1]), np.float32), np.zeros(260, np.uint32), (np.arange(66) * 4).astype(np.uint32))
    with pytest.raises(OiError):
        pipe.wait(10_000)
    pipe.close()
    idx.close(); ctx.close()


def test_pipeline_with_a_native_communicator_equals_oi_search_sharded():
    import torch
    import openintel_amd as oi
    K, DEPTH = 40, 150
    rows, terms, offs, batches = _case(n=30_011, seed=5, n_batches=6)
    ctx = oi.HipContext(0)
    comm = oi.NativeComm(ctx, oi.NativeComm.unique_id(), 0, 1)
    shard = oi.HybridIndex(ctx, rows.shape[0], rows.shape[1], 300, doc_id_base=123)
    shard.set_embeddings(rows.copy(), normalize=False)
    shard.set_forward(terms, offs)
    shard.finalize_sharded(comm)
    want = [shard.search_sharded(comm, q, qt, qo, k=K, depth=DEPTH) for q, qt, qo in batches]
    pipe = oi.NativePipeline(shard, lanes=2, max_queries=64, max_query_terms=4, depth=DEPTH, k=K, comm=comm)
    sub = [pipe.submit(q, qt, qo) for q, qt, qo in batches]
    pipe.drain()
    for (t, out), w in zip(sub, want):
        assert _same(out, w)
    dev = torch.device("cuda:0")
    dbs = [[torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) for x in b] for b in batches]
    torch.cuda.synchronize()
    sub = [pipe.submit(*b) for b in dbs]
    pipe.drain()
    for (t, out), w in zip(sub, want):
        assert _same(out, w)
    pipe.close()
    comm.close(); shard.close(); ctx.close()


def test_pipeline_survives_any_destruction_order():
    import openintel_amd as oi
    rows, terms, offs, batches = _case(n=20_000, n_batches=3)
    ctx = oi.HipContext(0)
    idx = _index(ctx, rows, terms, offs)
    pipe = oi.NativePipeline(idx, lanes=2, max_queries=64, max_query_terms=4, depth=100, k=10)
    t, out = pipe.submit(*batches[1])
    ctx.close()                      # the caller's ctx handle first: the index and the lanes keep what they need alive
    pipe.wait(t)
    want_docs = out.docs.copy()
    t2, out2 = pipe.submit(*batches[1])
    pipe.close()                     # drains
    assert np.array_equal(out2.docs, want_docs)
    idx.close()


@pytest.mark.parametrize("mode", ["copy", "stream"])
def test_concurrent_lanes_return_the_serial_lists_bit_for_bit(mode):
    """Round 5's co-residency finding (csrc/oi_device.h; profiles/r05_coresidency_probe.txt): with lanes that REALLY run at the same
    time (eight hardware queues), a pf_rescore_kernel wave scheduled on a CU beside a d = 384 screen workgroup of another lane lost
    one product of its packed fma chain -- one wrong exact cosine score in 2-11 % of the batches.  The MFMA kernels now take their
    CUs whole and the rescoring chain is single v_fma_f32.  Three lanes, their own streams, views of one index, rotating ragged
    batches, 40 rounds (360 batches; the unfixed library failed 10-40 of them): every packed pair of lists must be the serial
    call's, word for word."""
    import torch
    import openintel_amd as oi
    from openintel_amd import _lib
    DEPTH, LANES, ROUNDS = 200, 3, 40
    rows, terms, offs, batches = _case()                       # d = 384: the screen wave leaves room on its SIMD unless it claims it
    ctx = oi.HipContext(0)
    ctx.set_cosine_mode(_lib.OI_COSINE_SCREEN if mode == "copy" else _lib.OI_COSINE_SCREEN_STREAM)
    idx = _index(ctx, rows, terms, offs)
    if mode == "copy":
        assert idx.index_bytes()[1] > 0                        # (made by default at this size)
    dev = torch.device("cuda:0")
    dbs = [[torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) for x in b] for b in batches]
    torch.cuda.synchronize()
    want = [idx.search_lists_packed(*b, depth=DEPTH).clone() for b in dbs]
    torch.cuda.synchronize()
    lanes = []
    for _ in range(LANES):
        c = oi.HipContext.like(ctx)
        st = torch.cuda.Stream(device=dev)
        c.set_stream(st)
        c.set_overlap(False)
        lanes.append((idx.view(c), st, c))
    bad = []
    for rnd in range(ROUNDS):
        outs = []
        for i, b in enumerate(dbs):
            v, st, _ = lanes[(i + rnd) % LANES]
            with torch.cuda.stream(st):
                outs.append(v.search_lists_packed(*b, depth=DEPTH))
        torch.cuda.synchronize()
        bad += [(rnd, i) for i, o in enumerate(outs) if not torch.equal(o, want[i])]
    assert not bad, "batches whose lists differ from the serial call's: %s" % bad[:10]
    for v, _, c in lanes:
        v.close(); c.close()
    idx.close(); ctx.close()
