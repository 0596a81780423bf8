"""bf16 corpus (BASELINE configs[4]: bf16 embeddings): the cosine leg through cosine_bf16_filter.
Builder-defined like the rest of the retrieval path (parity unpinned): the oracle is
the f64 dot of the rows AS STORED with the queries rounded to bf16 (what the kernel is specified to
compute, f32-accumulated); tolerance 1e-5 absolute, and bit-exact with small-integer embeddings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
COS_TOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    import openintel_amd as oi
    c = oi.HipContext(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import lib
    return lib


def to_bf16_bits(x):
    """f32 -> bfloat16 bit patterns, round to nearest even."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def from_bf16_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def _forward(rng, n, vocab=50):
    lens = rng.integers(1, 9, size=n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    return rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32), offs


def _index(ctx, bits, terms, offs, vocab, base=0):
    import openintel_amd as oi
    idx = oi.HybridIndex(ctx, bits.shape[0], bits.shape[1], vocab, base)
    idx.set_embeddings_bf16(bits)
    idx.set_forward(terms, offs)
    idx.finalize()
    return idx


@pytest.mark.parametrize("B,dim,n", [(1, 768, 5000), (9, 384, 3000), (40, 768, 9000), (64, 768, 40_000),
                                     (70, 384, 6000), (33, 1024, 5000), (64, 1024, 20_000), (3, 1024, 33),
                                     (256, 1024, 30_000),   # the batch and width of BASELINE configs[4]
                                     (128, 768, 20_000), (100, 384, 5000), (200, 768, 3000),   # pair kernel shapes
                                     (97, 1024, 2011), (128, 1024, 9001), (225, 1024, 4000)])   # quad kernel (a quarter of K per wave)
def test_bf16_cosine_within_tolerance(ctx, O, B, dim, n):
    from openintel_amd import synth
    bits = to_bf16_bits(synth.embeddings_np(n, dim, seed=5 + B))
    rows = from_bf16_bits(bits)                       # the corpus as stored
    q = synth.embeddings_np(B, dim, seed=77 + B)
    qr = from_bf16_bits(to_bf16_bits(q))              # what the kernel multiplies with
    rng = np.random.default_rng(B)
    terms, offs = _forward(rng, n)
    idx = _index(ctx, bits, terms, offs, 50)
    qt, qo = np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32)
    for depth in (10, 1000):
        L = idx.search_lists(q, qt, qo, depth=depth)
        for b in range(B):
            ref = O.dot_scores(rows, qr[b])
            c = int(L.cos_counts[b])
            assert c == min(depth, n)
            d, s = L.cos_docs[b][:c], L.cos_scores[b][:c]
            assert np.unique(d).size == c and np.abs(s - ref[d]).max() <= COS_TOL
            assert (np.diff(s) <= 0).all()
            kth = np.sort(ref)[::-1][c - 1]
            assert np.isin(np.nonzero(ref > kth + 2 * COS_TOL)[0], d).all() and (ref[d] >= kth - 2 * COS_TOL).all()
    idx.close()


@pytest.mark.parametrize("n,dim,B,depth,k", [(70_000, 384, 9, 1000, 100), (40_000, 768, 64, 10, 10),
                                             (300_000, 384, 3, 100, 50), (5_000, 1024, 40, 1024, 1024),
                                             (9_001, 1024, 128, 500, 100), (40_003, 1024, 250, 64, 32)])   # quad kernel
def test_bf16_hybrid_pipeline_bit_exact(ctx, O, n, dim, B, depth, k):
    # small integers are exact in bf16 and their dot products exact in f32 in any order: the whole
    # pipeline (several cosine chunks, BM25, RRF) must match the oracle bit for bit
    rng = np.random.default_rng(n + B)
    rows = rng.integers(-3, 4, size=(n, dim)).astype(np.float32)
    q = rng.integers(-3, 4, size=(B, dim)).astype(np.float32)
    vocab = 300
    terms, offs = _forward(rng, n, vocab)
    qt = rng.integers(0, 12, size=B * 4).astype(np.uint32)
    qo = (np.arange(B + 1) * 4).astype(np.uint32)
    idx = _index(ctx, to_bf16_bits(rows), terms, offs, vocab, base=1000)
    L = idx.search_lists(q, qt, qo, depth=depth)
    R = idx.search(q, qt, qo, k=k, depth=depth)
    for b in range(B):
        cs, cd = O.topk(O.dot_scores(rows, q[b]), depth, False, 1000)
        bs, bd = O.topk(O.bm25_scores(terms, offs, vocab, qt[qo[b]:qo[b + 1]]), depth, True, 1000)
        fs, fd = O.rrf_fuse(cd, bd, k)
        assert int(L.cos_counts[b]) == cd.size
        assert np.array_equal(L.cos_docs[b][:cd.size], cd) and np.array_equal(L.cos_scores[b][:cd.size], cs)
        assert np.array_equal(L.bm25_docs[b][:bd.size], bd)
        assert int(R.counts[b]) == fd.size and np.array_equal(R.docs[b][:fd.size], fd)
        assert np.array_equal(R.scores[b][:fd.size].view(np.uint32), fs.view(np.uint32))
    idx.close()


def test_bf16_device_tensor_and_errors(ctx, O):
    import torch
    import openintel_amd as oi
    from openintel_amd import _lib, synth
    n, dim, B = 10_000, 768, 5
    x = torch.from_numpy(synth.embeddings_np(n, dim, seed=9)).cuda().to(torch.bfloat16).contiguous()
    rows = x.float().cpu().numpy()
    rng = np.random.default_rng(0)
    terms, offs = _forward(rng, n)
    idx = oi.HybridIndex(ctx, n, dim, 50)
    idx.set_embeddings_bf16(x)                        # borrowed torch.bfloat16 tensor in HBM
    idx.set_forward(terms, offs)
    idx.finalize()
    q = synth.embeddings_np(B, dim, seed=10)
    qr = from_bf16_bits(to_bf16_bits(q))
    L = idx.search_lists(q, np.zeros(B, np.uint32), np.arange(B + 1, dtype=np.uint32), depth=100)
    for b in range(B):
        ref = O.dot_scores(rows, qr[b])
        d = L.cos_docs[b][:100]
        assert np.abs(L.cos_scores[b][:100] - ref[d]).max() <= COS_TOL
    idx.close()
    bad = oi.HybridIndex(ctx, 100, 96, 10)            # dim without a bf16 kernel: loud, not a fallback
    with pytest.raises(_lib.OiError):
        bad.set_embeddings_bf16(np.zeros((100, 96), np.uint16))
    bad.close()


def test_full_size_config4_shard_12_5M_1024_bf16_batch256(ctx):
    """BASELINE configs[4] as ONE GPU of the 8 sees it: a 12.5M-row x 1024-d bf16 shard (25.6 GB), batch 256, depth
    1000 -> RRF top-100, with a nonzero doc_id_base.  Size-independent properties: planted copies of the (bf16-
    rounded) queries come back first with the planted score, full sorted lists of in-range unique ids, idempotence,
    and three queries spot-checked against dense torch scores of the rows as stored."""
    import torch
    import openintel_amd as oi
    from openintel_amd import synth
    dev = torch.device("cuda:0")
    n, dim, B, depth, k, base = 12_500_000, 1024, 256, 1000, 100, 37_500_000   # shard 3 of 8 of the 100M corpus
    rows = torch.empty((n, dim), dtype=torch.bfloat16, device=dev)
    step = 2_500_000
    for r in range(0, n, step):                         # generated in slices: the f32 staging stays at 10 GB
        rows[r:r + step] = synth.embeddings_torch(step, dim, dev, seed=synth.SEED_EMB + r).to(torch.bfloat16)
    qv, qt, qo = synth.query_batch_torch(B, dim, dev, vocab=4096)
    qb = qv.to(torch.bfloat16)                          # what the kernel multiplies with
    plant = torch.arange(B, device=dev) * (n // B) + 29
    rows[plant] = qb
    terms, offs = synth.forward_index_torch(n, dev, vocab=4096)
    idx = oi.HybridIndex(ctx, n, dim, 4096, doc_id_base=base)
    idx.set_embeddings_bf16(rows)
    idx.set_forward(terms, offs)
    idx.set_max_query_terms(4)
    idx.finalize()
    del terms, offs
    L = idx.search_lists(qv, qt, qo, depth=depth)
    ctx.synchronize()
    cs, cd, cc = L.cos_scores.cpu().numpy(), L.cos_docs.cpu().numpy().astype(np.int64), L.cos_counts.cpu().numpy()
    assert (cc == depth).all() and cd.min() >= base and cd.max() < base + n
    self_score = (qb.float() * qb.float()).sum(1).cpu().numpy()
    assert np.array_equal(cd[:, 0], plant.cpu().numpy() + base)
    assert np.abs(cs[:, 0] - self_score).max() < 1e-5
    assert (np.diff(cs, axis=1) <= 0).all()
    assert all(np.unique(cd[b]).size == depth for b in range(0, B, 17))
    bs, bc = L.bm25_scores.cpu().numpy(), L.bm25_counts.cpu().numpy()
    assert (bc == depth).all() and (np.diff(bs, axis=1) <= 0).all() and (bs > 0).all()
    R1 = idx.search(qv, qt, qo, k=k, depth=depth)
    R2 = idx.search(qv, qt, qo, k=k, depth=depth)
    ctx.synchronize()
    assert torch.equal(R1.docs, R2.docs) and torch.equal(R1.scores, R2.scores) and bool((R1.counts == k).all())
    for b in (0, 100, 255):
        # (the whole matrix in f32 would be 51 GB: scored in slices)
        ref = torch.cat([(rows[r:r + step].float() @ qb[b].float()) for r in range(0, n, step)]).cpu().numpy().astype(np.float64)
        d = cd[b] - base
        assert np.abs(cs[b].astype(np.float64) - ref[d]).max() <= COS_TOL
        kth = np.sort(ref)[::-1][depth - 1]
        assert np.isin(np.nonzero(ref > kth + 2 * COS_TOL)[0], d).all() and (ref[d] >= kth - 2 * COS_TOL).all()
    idx.close()
