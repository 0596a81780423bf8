"""Handle lifetime at the C ABI (VERDICT r02 missing #5 / weak #8): ctx, index and view may be destroyed in ANY order.

Round 1's `gpurun_out/pf1.log` ended in `terminate called after throwing 'std::bad_variant_access'` at interpreter exit:
a failed test kept a HybridIndex alive past its ctx fixture, and oi_index_destroy then locked the mutex of a freed oi_ctx.
The library now counts references (include/openintel_hip.h, oi_destroy): these tests drop the ctx FIRST."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(n=20000, dim=64, vocab=300, B=12, seed=5):
    rng = np.random.default_rng(seed)
    rows = rng.integers(-3, 4, size=(n, dim)).astype(np.float32)      # small integers: every dot product exact in any order
    q = rng.integers(-3, 4, size=(B, dim)).astype(np.float32)
    lens = rng.integers(1, 12, size=n)
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    terms = rng.integers(0, vocab, size=int(offs[-1])).astype(np.uint32)
    qo = (np.arange(B + 1) * 3).astype(np.uint32)
    qt = rng.integers(0, vocab, size=3 * B).astype(np.uint32)
    return rows, q, terms, offs, qt, qo


def _index(ctx, rows, terms, offs, vocab=300):
    import openintel_amd as oi
    idx = oi.HybridIndex(ctx, rows.shape[0], rows.shape[1], vocab)
    idx.set_embeddings(rows, normalize=False)
    idx.set_forward(terms, offs)
    idx.finalize()
    return idx


def test_ctx_destroyed_before_its_index_and_view_before_or_after_source():
    import openintel_amd as oi
    rows, q, terms, offs, qt, qo = _case()
    ctx = oi.HipContext(0)
    idx = _index(ctx, rows, terms, offs)
    want = idx.search(q, qt, qo, k=10, depth=100)
    ctx2 = oi.HipContext(0)
    view = idx.view(ctx2)
    ctx.close()                         # oi_destroy with a live index AND a live view of it: deferred inside the library
    got = idx.search(q, qt, qo, k=10, depth=100)          # the index stays usable (default stream from here on)
    assert np.array_equal(got.docs, want.docs) and np.array_equal(got.scores, want.scores)
    idx.close()                         # source destroyed before its view: the buffers live on
    ctx2.close()                        # ... and the view's ctx handle too
    got = view.search(q, qt, qo, k=10, depth=100)
    assert np.array_equal(got.docs, want.docs) and np.array_equal(got.scores, want.scores)
    view.close()                        # last handle: everything is freed here
    # a fresh ctx works afterwards (nothing was left locked or half torn down)
    ctx3 = oi.HipContext(0)
    idx3 = _index(ctx3, rows, terms, offs)
    got = idx3.search(q, qt, qo, k=10, depth=100)
    assert np.array_equal(got.docs, want.docs)
    idx3.close()
    ctx3.close()


SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import openintel_amd as oi
sys.path.insert(0, %r)
from test_gpu_lifetime import _case, _index
rows, q, terms, offs, qt, qo = _case()
ctx = oi.HipContext(0)
LEAKED = _index(ctx, rows, terms, offs)          # module global: alive until interpreter teardown
LEAKED_VIEW_CTX = oi.HipContext(0)
LEAKED_VIEW = LEAKED.view(LEAKED_VIEW_CTX)
r = LEAKED.search(q, qt, qo, k=10, depth=100)
ctx.close()                                        # the pf1.log order: ctx gone, index destroyed later by __del__
del ctx
print("ok", int(r.counts.sum()))
"""


def test_interpreter_exit_with_a_leaked_index_whose_ctx_was_closed_first():
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "ok" in r.stdout and "terminate called" not in r.stderr and "core dumped" not in r.stderr


@pytest.mark.gpu
def test_workspace_bytes_reports_what_a_searching_context_holds():
    """oi_workspace_bytes: nothing before the first call, the query workspaces after a search, page-locked staging after an
    OI_HOST call; a view's context brings its own workspaces, not the index.  Round 4's bound: the BM25 pool is a 4096-key
    segment per (query, block) of the first phase -- not a 32768-key one for every block -- and the cosine scorers share
    ONE pool of at most (rows + slack) keys per query."""
    import numpy as np
    import openintel_amd as oi
    from openintel_amd import synth
    ctx = oi.HipContext(0)
    assert ctx.workspace_bytes() == (0, 0)
    n, dim, B = 200_000, 64, 8
    idx = oi.HybridIndex(ctx, n, dim, synth.VOCAB)
    idx.set_embeddings(synth.embeddings_np(n, dim), normalize=False)
    terms, offs = synth.forward_index_np(n)
    idx.set_forward(terms, offs)
    idx.set_max_query_terms(4)
    idx.finalize()
    built, _ = ctx.workspace_bytes()
    qv = synth.embeddings_np(B, dim, seed=5)
    qt, qo = synth.query_terms_np(B)
    idx.search(qv, qt, qo, k=10, depth=100)          # host arrays: the OI_HOST entry point
    dev, pinned = ctx.workspace_bytes()
    n_blocks = (n + 32767) // 32768
    bm25_pool = B * (1024 + n_blocks * 4096) * 8      # the stream kernel's pool (7 blocks: one phase)
    cos_pool = B * (n + 4096 + 128 * 257) * 8         # one pool for the cosine scorers, worst case one key per row
    assert bm25_pool <= dev - built <= bm25_pool + cos_pool + (8 << 20), (dev - built, bm25_pool, cos_pool)
    assert 4 * bm25_pool < B * n_blocks * 32768 * 8    # (round 3 held a 32768-key segment per (query, block) for BM25 alone)
    assert 0 < pinned <= 2 << 20
    ctx2 = oi.HipContext(0)
    view = idx.view(ctx2)
    assert ctx2.workspace_bytes()[0] == 0            # the view borrows the index buffers
    view.search(qv, qt, qo, k=10, depth=100)
    assert bm25_pool <= ctx2.workspace_bytes()[0] <= bm25_pool + cos_pool + (8 << 20)
    view.close(); ctx2.close(); idx.close(); ctx.close()

