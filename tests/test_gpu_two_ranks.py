"""The N > 1 path with the REAL engine, on one GPU: two processes (gloo; RCCL refuses two ranks on one device) each hold a
row shard of the same corpus in HBM and run what `bench.py --gpus N` runs -- ShardedRetriever.finalize (df / N / token
all-reduce), then sharded.ShardedPipeline with its empirical lane calibration: lists of a batch through the shard or a
read-only view of it, ONE all-gather of the packed lists per batch on the exchange stream, merge + RRF on the fusion
context.  Small-integer embeddings: every dot product is exact in any order, so every rank's fused result must equal the
single-index oi_search over the whole corpus bit for bit."""
import socket

import numpy as np
import pytest

N, DIM, VOCAB, B, DEPTH, K, NB = 90_001, 384, 300, 64, 200, 50, 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _corpus():
    rng = np.random.default_rng(2024)
    rows = rng.integers(-3, 4, size=(N, DIM)).astype(np.float32)
    lens = rng.integers(1, 12, size=N)
    offs = np.zeros(N + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    terms = rng.integers(0, VOCAB, size=int(offs[-1])).astype(np.uint32)
    batches = []
    for i in range(NB):
        q = rng.integers(-3, 4, size=(B, DIM)).astype(np.float32)
        qt = rng.integers(0, 40, size=B * 4).astype(np.uint32)
        qo = (np.arange(B + 1) * 4).astype(np.uint32)
        batches.append((q, qt, qo))
    return rows, terms, offs, batches


def _worker(rank, world, port, ret):
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import openintel_amd as oi
    from openintel_amd import sharded
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rows, terms, offs, batches = _corpus()
    lo, hi = sharded.shard_bounds(N, world, rank)
    ctx = oi.HipContext(0)
    ctx.use_torch_current_stream()
    t_lo, t_hi = int(offs[lo]), int(offs[hi])
    idx = oi.HybridIndex(ctx, hi - lo, DIM, VOCAB, doc_id_base=lo)
    d_rows = torch.from_numpy(rows[lo:hi].copy()).to(dev)
    idx.set_embeddings(d_rows, normalize=False)
    idx.set_forward(terms[t_lo:t_hi].copy(), (offs[lo:hi + 1] - offs[lo]).astype(np.uint64))
    sr = sharded.make_hip_sharded(ctx, idx, dev)
    sr.finalize()
    dbatches = [tuple(torch.from_numpy(x).to(dev) for x in b) for b in batches]
    fctx = oi.HipContext(0)
    pipe = sharded.ShardedPipeline(sr, fctx, B, DEPTH, K)
    cal = pipe.calibrate(dbatches, lambda: oi.HipContext(0), reps=4, placements=2)
    if len(pipe.lanes) == 1:      # whatever the timing chose, the two-lane path is what this test is about
        c = oi.HipContext(0)
        st = torch.cuda.Stream(device=dev)
        c.set_stream(st)
        pipe.lanes.append((idx.view(c), st))
    outs = []
    for rep in range(2):
        for b in dbatches:
            slot = pipe.submit(*b)
            with torch.cuda.stream(pipe.side):
                r = pipe.results[slot]
                outs.append((r.scores.clone(), r.docs.clone(), r.counts.clone()))
    pipe.drain()
    torch.cuda.synchronize()
    ret[rank] = ([(s.cpu().numpy(), d.cpu().numpy(), c.cpu().numpy()) for s, d, c in outs], cal["chosen_lanes"])
    pipe.close()
    idx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_pipeline_with_lanes_equals_the_unsharded_search():
    import torch
    import torch.multiprocessing as mp
    import openintel_amd as oi
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    rows, terms, offs, batches = _corpus()
    ctx = oi.HipContext(0)
    idx = oi.HybridIndex(ctx, N, DIM, VOCAB)
    idx.set_embeddings(rows.copy(), normalize=False)
    idx.set_forward(terms, offs)
    idx.finalize()
    want = []
    for q, qt, qo in batches:
        r = idx.search(q, qt, qo, k=K, depth=DEPTH)
        want.append((np.asarray(r.scores).copy(), np.asarray(r.docs).copy(), np.asarray(r.counts).copy()))
    for rank in range(world):
        outs, lanes = ret[rank]
        assert lanes in (1, 2) and len(outs) == 2 * NB
        for i, (s, d, c) in enumerate(outs):
            ws, wd, wc = want[i % NB]
            assert np.array_equal(c, wc), "rank %d batch %d counts" % (rank, i)
            for b in range(B):
                n = int(wc[b])
                assert np.array_equal(d[b, :n], wd[b, :n].astype(d.dtype)), "rank %d batch %d query %d docs" % (rank, i, b)
                assert np.array_equal(s[b, :n].view(np.uint32), ws[b, :n].view(np.uint32)), "rank %d batch %d query %d scores" % (rank, i, b)
    idx.close()
    ctx.close()
    assert torch.cuda.is_available()
