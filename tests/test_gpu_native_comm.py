"""The sharded query with RCCL INSIDE the library (oi_comm_*, oi_index_finalize_sharded, oi_search_sharded; VERDICT r02
missing #3 / next #7): no torch.distributed anywhere.  A communicator of ONE rank (RCCL refuses two ranks on one device)
runs the real calls -- ncclGetUniqueId, ncclCommInitRank, the statistics all-reduce, ncclAllGather of the packed lists --
and the result must equal the plain oi_search bit for bit, host and device buffers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(n=50_001, dim=384, vocab=300, B=64, seed=11):
    rng = np.random.default_rng(seed)
    rows = rng.integers(-3, 4, size=(n, dim)).astype(np.float32)
    q = rng.integers(-3, 4, size=(B, dim)).astype(np.float32)
    lens = rng.integers(1, 12, size=n)
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    terms = rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32)
    qo = (np.arange(B + 1) * 4).astype(np.uint32)
    qt = rng.integers(0, 40, size=4 * B).astype(np.uint32)
    return rows, q, terms, offs, qt, qo


def test_native_comm_of_one_rank_equals_oi_search_bit_for_bit():
    import torch
    import openintel_amd as oi
    rows, q, terms, offs, qt, qo = _case()
    n, dim = rows.shape
    K, DEPTH, BASE = 50, 200, 4000
    ctx = oi.HipContext(0)
    plain = oi.HybridIndex(ctx, n, dim, 300, doc_id_base=BASE)
    plain.set_embeddings(rows.copy(), normalize=False)
    plain.set_forward(terms, offs)
    plain.finalize()
    want = plain.search(q, qt, qo, k=K, depth=DEPTH)

    uid = oi.NativeComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = oi.NativeComm(ctx, uid, 0, 1)
    shard = oi.HybridIndex(ctx, n, dim, 300, doc_id_base=BASE)
    shard.set_embeddings(rows.copy(), normalize=False)
    shard.set_forward(terms, offs)
    shard.finalize_sharded(comm)                      # ncclAllReduce of (N, tokens) and df inside the library
    got = shard.search_sharded(comm, q, qt, qo, k=K, depth=DEPTH)      # OI_HOST: synchronous
    assert np.array_equal(got.counts, want.counts)
    assert np.array_equal(got.docs, want.docs) and np.array_equal(got.scores.view(np.uint32), want.scores.view(np.uint32))
    assert int(got.docs.min()) >= BASE
    # device buffers, asynchronous on the ctx stream; twice (the workspaces are reused)
    dev = torch.device("cuda:0")
    dq = [torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) for x in (q, qt, qo)]
    for _ in range(2):
        r = shard.search_sharded(comm, dq[0], dq[1], dq[2], k=K, depth=DEPTH)
        ctx.synchronize()
        assert np.array_equal(r.docs.cpu().numpy().view(np.uint32), want.docs)
        assert np.array_equal(r.scores.cpu().numpy().view(np.uint32), want.scores.view(np.uint32))
    # errors are loud: a communicator of another context is refused, a second finalize too
    from openintel_amd._lib import OiError
    ctx2 = oi.HipContext(0)
    comm2 = oi.NativeComm(ctx2, oi.NativeComm.unique_id(), 0, 1)
    with pytest.raises(OiError):
        shard.search_sharded(comm2, q, qt, qo, k=K, depth=DEPTH)
    with pytest.raises(OiError):
        shard.finalize_sharded(comm)
    comm2.close(); ctx2.close()
    # any destruction order (reference counts): ctx handle first, then the index, the communicator last
    ctx.close()
    shard.close(); plain.close()
    comm.close()
