"""Hybrid retrieval (cosine + BM25 + reciprocal-rank fusion) on one MI355X.

The reference has no retrieval port, so `PostRetriever` is a NEW port,
styled after the reference's existing ones (borrowed inputs, owned outputs, DomainError-style
failures -- compare src/domain/ports/post_analyzer.rs:7-11).  `HybridIndex` is its
libopenintel_hip.so implementation; `openintel_amd.sharded` scales it over RCCL.
"""
from __future__ import annotations

import abc
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .context import HipContext

DEFAULT_DEPTH = 1000  # per-list depth k' fed to RRF
DEFAULT_K = 100


@dataclass
class RankedLists:
    """Per-query ranked lists of one shard: rows sorted by (score desc, doc id asc)."""
    cos_scores: object
    cos_docs: object
    cos_counts: object
    bm25_scores: object
    bm25_docs: object
    bm25_counts: object


@dataclass
class SearchResult:
    scores: object   # [B, k] f32 RRF scores
    docs: object     # [B, k] u32 global doc ids
    counts: object   # [B] valid entries per row


class PostRetriever(abc.ABC):
    """New port: rank stored posts for a batch of (embedding, term-id) queries."""

    @abc.abstractmethod
    def search(self, query_vecs, query_terms, q_term_offsets, k: int = DEFAULT_K,
               depth: int = DEFAULT_DEPTH) -> SearchResult:
        ...


def packed_words(n_queries: int, depth: int) -> int:
    """OI_PACKED_WORDS of include/openintel_hip.h."""
    return 4 * n_queries * depth + 2 * n_queries


def unpack_lists(packed, n_queries: int, depth: int) -> "RankedLists":
    """Views into one shard's packed buffer (numpy array or torch tensor of 32-bit words)."""
    L = n_queries * depth
    if hasattr(packed, "data_ptr"):
        import torch
        sc = packed[:2 * L].view(torch.float32).reshape(2, n_queries, depth)
    else:
        sc = packed[:2 * L].view(np.float32).reshape(2, n_queries, depth)
    dc = packed[2 * L:4 * L].reshape(2, n_queries, depth)
    cn = packed[4 * L:].reshape(2, n_queries)
    return RankedLists(sc[0], dc[0], cn[0], sc[1], dc[1], cn[1])


def _is_dev(x) -> bool:
    return hasattr(x, "data_ptr")


def _np(x, dtype):
    return np.ascontiguousarray(x, dtype=dtype)


def pack_query_terms(term_lists: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
    offs = np.zeros(len(term_lists) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(t) for t in term_lists])
    flat = np.fromiter((t for ts in term_lists for t in ts), dtype=np.uint32, count=int(offs[-1]))
    return flat, offs


class HybridIndex(PostRetriever):
    """One corpus shard resident in HBM: n_docs x dim f32 rows + a blocked BM25 inverted index."""

    def __init__(self, ctx: HipContext, n_docs: int, dim: int, vocab: int, doc_id_base: int = 0):
        self.ctx = ctx
        self.lib = ctx.lib
        self.n_docs, self.dim, self.vocab, self.doc_id_base = int(n_docs), int(dim), int(vocab), int(doc_id_base)
        h = C.c_void_p()
        _lib.check(self.lib.oi_index_create(ctx.handle, self.n_docs, self.dim, self.vocab, self.doc_id_base,
                                            C.byref(h)))
        self.handle = h
        self._keep = []  # device tensors the library borrows

    # ---------------------------------------------------------------- build
    def set_embeddings(self, rows, normalize: bool = True) -> None:
        """rows: [n_docs, dim] f32 numpy array (copied to HBM) or torch CUDA tensor (borrowed;
        normalised in place when normalize=True)."""
        if _is_dev(rows):
            assert tuple(rows.shape) == (self.n_docs, self.dim) and rows.is_contiguous()
            self._keep.append(rows)
            loc = _lib.OI_DEVICE
        else:
            rows = _np(rows, np.float32)
            assert rows.shape == (self.n_docs, self.dim)
            loc = _lib.OI_HOST
        _lib.check(self.lib.oi_index_set_embeddings(self.handle, _lib.ptr(rows), loc, 1 if normalize else 0))

    def set_embeddings_bf16(self, rows) -> None:
        """A bf16 corpus: a torch.bfloat16 CUDA tensor [n_docs, dim] (borrowed, never copied) or a numpy
        uint16 array of bfloat16 bit patterns (copied to HBM).  Rows must be unit-norm as stored."""
        if _is_dev(rows):
            assert rows.is_contiguous() and tuple(rows.shape) == (self.n_docs, self.dim) and rows.element_size() == 2
            self._keep.append(rows)  # keep the borrowed matrix alive
            loc = _lib.OI_DEVICE
        else:
            rows = np.ascontiguousarray(rows, dtype=np.uint16)
            assert rows.shape == (self.n_docs, self.dim)
            loc = _lib.OI_HOST
        _lib.check(self.lib.oi_index_set_embeddings_bf16(self.handle, _lib.ptr(rows), loc))

    def set_forward(self, term_ids, doc_offsets) -> None:
        """Forward index: doc d owns term_ids[doc_offsets[d]:doc_offsets[d+1]] (u32 ids, u64 offsets)."""
        if _is_dev(term_ids):
            loc = _lib.OI_DEVICE
        else:
            term_ids, doc_offsets = _np(term_ids, np.uint32), _np(doc_offsets, np.uint64)
            assert doc_offsets.size == self.n_docs + 1
            loc = _lib.OI_HOST
        _lib.check(self.lib.oi_index_set_forward(self.handle, _lib.ptr(term_ids), _lib.ptr(doc_offsets), loc))

    def long_rows(self) -> int:
        """Rows the screened cosine scorer sets aside (always rescored, never part of its thresholds): oi_index_long_rows."""
        n = C.c_uint32()
        _lib.check(self.lib.oi_index_long_rows(self.handle, C.byref(n)))
        return int(n.value)

    SCREEN_COPY_AUTO, SCREEN_COPY_NEVER, SCREEN_COPY_ALWAYS = _lib.OI_SCREEN_COPY_AUTO, _lib.OI_SCREEN_COPY_NEVER, _lib.OI_SCREEN_COPY_ALWAYS

    def set_screen_copy(self, policy: int) -> None:
        """The bf16 screening copy of an f32 corpus (oi_index_set_screen_copy): SCREEN_COPY_AUTO (default: made at finalize
        when n_docs x dim x 2 B is at most a quarter of the free HBM), _NEVER, _ALWAYS.  The default scorer's screen streams
        it instead of the f32 rows -- half the bytes, the same lists."""
        _lib.check(self.lib.oi_index_set_screen_copy(self.handle, int(policy)))

    def index_bytes(self):
        """(rows owned by the library, screening copy, BM25 structures) in bytes of HBM: oi_index_bytes."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.oi_index_bytes(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    BM25_DEFAULT, BM25_TAAT, BM25_SCAN, BM25_WAVE, BM25_STREAM = 0, 1, 2, 3, 4

    def set_bm25_mode(self, mode: int) -> None:
        """BM25_STREAM (term-at-a-time as a stream through a per-wave LDS ring: the default), BM25_WAVE (one wave per
        (block, query) task), BM25_TAAT (the first-generation workgroup-per-block kernel) or BM25_SCAN (batch scan of
        the forward index).  Bit-identical lists."""
        _lib.check(self.lib.oi_index_set_bm25_mode(self.handle, int(mode)))

    def set_max_query_terms(self, max_terms: int) -> None:
        """Contract for the batch BM25 scan: no query has more terms than this (default 16)."""
        _lib.check(self.lib.oi_index_set_max_query_terms(self.handle, int(max_terms)))

    def local_stats(self) -> Tuple[int, np.ndarray]:
        tot = C.c_uint64()
        df = np.zeros(self.vocab, dtype=np.uint32)
        _lib.check(self.lib.oi_index_local_stats(self.handle, C.byref(tot), _lib.ptr(df)))
        return tot.value, df

    def finalize(self, global_n_docs: Optional[int] = None, global_total_tokens: Optional[int] = None,
                 global_df: Optional[np.ndarray] = None) -> None:
        if global_n_docs is None:
            tot, _ = self.local_stats()
            global_n_docs, global_total_tokens, global_df = self.n_docs, tot, None
        df = None if global_df is None else _np(global_df, np.uint32)
        _lib.check(self.lib.oi_index_finalize(self.handle, int(global_n_docs), int(global_total_tokens),
                                              _lib.ptr(df)))

    # ---------------------------------------------------------------- query
    def _alloc(self, like_device: bool, shape, dtype):
        if like_device:
            import torch
            tdt = {np.float32: torch.float32, np.uint32: torch.int32}[dtype]
            return torch.zeros(shape, dtype=tdt, device="cuda:%d" % self.ctx.device)
        return np.zeros(shape, dtype=dtype)

    def _queries(self, query_vecs, query_terms, q_term_offsets):
        dev = _is_dev(query_vecs)
        if not dev:
            query_vecs = _np(query_vecs, np.float32)
            query_terms = _np(query_terms, np.uint32)
            q_term_offsets = _np(q_term_offsets, np.uint32)
            if query_terms.size == 0:
                query_terms = np.zeros(1, dtype=np.uint32)
        B = int(query_vecs.shape[0])
        assert int(query_vecs.shape[1]) == self.dim
        return dev, B, query_vecs, query_terms, q_term_offsets

    def search_lists(self, query_vecs, query_terms, q_term_offsets, depth: int = DEFAULT_DEPTH) -> RankedLists:
        dev, B, qv, qt, qo = self._queries(query_vecs, query_terms, q_term_offsets)
        out = RankedLists(*(self._alloc(dev, s, d) for s, d in (
            ((B, depth), np.float32), ((B, depth), np.uint32), ((B,), np.uint32),
            ((B, depth), np.float32), ((B, depth), np.uint32), ((B,), np.uint32))))
        _lib.check(self.lib.oi_search_lists(
            self.handle, _lib.ptr(qv), _lib.ptr(qt), _lib.ptr(qo), B, int(depth),
            _lib.OI_DEVICE if dev else _lib.OI_HOST, _lib.ptr(out.cos_scores), _lib.ptr(out.cos_docs),
            _lib.ptr(out.cos_counts), _lib.ptr(out.bm25_scores), _lib.ptr(out.bm25_docs),
            _lib.ptr(out.bm25_counts)))
        return out

    def search_lists_packed(self, query_vecs, query_terms, q_term_offsets, depth: int = DEFAULT_DEPTH, out=None):
        """The shard's two lists in the multi-GPU exchange format (include/openintel_hip.h,
        OI_PACKED_WORDS): one flat int32/uint32 buffer, ready for all_gather_into_tensor."""
        dev, B, qv, qt, qo = self._queries(query_vecs, query_terms, q_term_offsets)
        words = packed_words(B, depth)
        if out is None:
            out = self._alloc(dev, (words,), np.uint32)
        _lib.check(self.lib.oi_search_lists_packed(self.handle, _lib.ptr(qv), _lib.ptr(qt), _lib.ptr(qo), B,
                                                   int(depth), _lib.OI_DEVICE if dev else _lib.OI_HOST,
                                                   _lib.ptr(out)))
        return out

    def search(self, query_vecs, query_terms, q_term_offsets, k: int = DEFAULT_K,
               depth: int = DEFAULT_DEPTH, out: Optional[SearchResult] = None) -> SearchResult:
        dev, B, qv, qt, qo = self._queries(query_vecs, query_terms, q_term_offsets)
        if out is None:
            out = SearchResult(self._alloc(dev, (B, k), np.float32), self._alloc(dev, (B, k), np.uint32),
                               self._alloc(dev, (B,), np.uint32))
        _lib.check(self.lib.oi_search(self.handle, _lib.ptr(qv), _lib.ptr(qt), _lib.ptr(qo), B, int(depth), int(k),
                                      _lib.OI_DEVICE if dev else _lib.OI_HOST, _lib.ptr(out.scores),
                                      _lib.ptr(out.docs), _lib.ptr(out.counts)))
        return out

    # ---------------------------------------------------------------- the sharded query with RCCL inside the library
    def finalize_sharded(self, comm: "NativeComm") -> None:
        """Collective over `comm`: all-reduce of (n_docs, tokens, df) inside the library, then the impacts from the global
        statistics (oi_index_finalize_sharded) -- what ShardedRetriever.finalize does with torch.distributed."""
        _lib.check(self.lib.oi_index_finalize_sharded(self.handle, comm.handle))

    def search_sharded(self, comm: "NativeComm", query_vecs, query_terms, q_term_offsets, k: int = DEFAULT_K,
                       depth: int = DEFAULT_DEPTH, out: Optional[SearchResult] = None) -> SearchResult:
        """Collective over `comm`, same queries on every rank: this shard's lists -> ONE ncclAllGather -> global merge ->
        RRF, in one C call (oi_search_sharded).  Identical result on every rank."""
        dev, B, qv, qt, qo = self._queries(query_vecs, query_terms, q_term_offsets)
        if out is None:
            out = SearchResult(self._alloc(dev, (B, k), np.float32), self._alloc(dev, (B, k), np.uint32),
                               self._alloc(dev, (B,), np.uint32))
        _lib.check(self.lib.oi_search_sharded(self.handle, comm.handle, _lib.ptr(qv), _lib.ptr(qt), _lib.ptr(qo), B,
                                              int(depth), int(k), _lib.OI_DEVICE if dev else _lib.OI_HOST,
                                              _lib.ptr(out.scores), _lib.ptr(out.docs), _lib.ptr(out.counts)))
        return out

    def screen_probe(self, query_vecs, row_begin: int = 0, n_rows: int = 0):
        """Diagnostics of the bf16 screen (oi_screen_probe): (s~ [B, n_rows] or None, eps [B]) for host queries --
        the screen's raw scores of rows [row_begin, row_begin + n_rows) and each query's proven bound."""
        qv = _np(query_vecs, np.float32)
        B = int(qv.shape[0])
        assert int(qv.shape[1]) == self.dim
        st = np.zeros((B, n_rows), np.float32) if n_rows else None
        eps = np.zeros(B, np.float32)
        _lib.check(self.lib.oi_screen_probe(self.handle, _lib.ptr(qv), B, int(row_begin), int(n_rows), _lib.ptr(st),
                                            _lib.ptr(eps)))
        return st, eps

    def view(self, ctx: HipContext) -> "HybridIndex":
        """A second handle on this (finalized) shard, bound to `ctx` -- another HipContext of the same device, with its
        own stream and workspaces -- so that two searches can be in flight at once (oi_index_view).  Borrows every
        buffer: read-only, no HBM; close it before this index."""
        v = HybridIndex.__new__(HybridIndex)
        v.ctx, v.lib = ctx, ctx.lib
        v.n_docs, v.dim, v.vocab, v.doc_id_base = self.n_docs, self.dim, self.vocab, self.doc_id_base
        h = C.c_void_p()
        _lib.check(self.lib.oi_index_view(self.handle, ctx.handle, C.byref(h)))
        v.handle = h
        v._keep = [self]  # the source outlives the view
        return v

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.oi_index_destroy(self.handle)
            self.handle = None
            self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeComm:
    """An RCCL communicator owned by the library (oi_comm_*): the multi-GPU exchange without torch.distributed.

        id = NativeComm.unique_id()            # rank 0; ship the 128 bytes to the other ranks over any host channel
        comm = NativeComm(ctx, id, rank, world)  # collective
        idx.finalize_sharded(comm); idx.search_sharded(comm, qv, qt, qo, k, depth)

    Collectives run on the ctx stream in call order; every rank issues them in the same order."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * _lib.OI_COMM_ID_BYTES)()
        _lib.check(_lib.load().oi_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def __init__(self, ctx: HipContext, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == _lib.OI_COMM_ID_BYTES
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        buf = (C.c_uint8 * _lib.OI_COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _lib.check(ctx.lib.oi_comm_create(ctx.handle, C.cast(buf, C.c_void_p), self.rank, self.world, C.byref(h)))
        self.handle = h

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.ctx.lib.oi_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativePipeline:
    """Several batches in flight through the library's own lanes (oi_pipeline_*): the pipelined -- and, with a NativeComm,
    row-sharded -- query without torch streams or torch.distributed on the data path.

        pipe = NativePipeline(idx, lanes=2, max_queries=64, max_query_terms=4, depth=1000, k=100)   # comm=NativeComm(...) to shard
        t = pipe.submit(qv, qt, qo, out=result_slot)     # asynchronous (device tensors or numpy arrays)
        pipe.wait(t)                                     # that batch's outputs are complete
        pipe.drain(); pipe.close()

    Device batches: `out` (a SearchResult of device tensors, shape (n_queries, k)) must be a distinct buffer per batch in
    flight; inputs stay unmodified until wait().  Results are bit-identical to HybridIndex.search / search_sharded."""

    def __init__(self, index: "HybridIndex", lanes: int = 2, max_queries: int = 64, max_query_terms: int = 16,
                 depth: int = DEFAULT_DEPTH, k: int = DEFAULT_K, comm: Optional["NativeComm"] = None):
        self.index, self.lib, self.comm = index, index.lib, comm
        self.k, self.depth, self.max_queries = int(k), int(depth), int(max_queries)
        h = C.c_void_p()
        _lib.check(self.lib.oi_pipeline_create(index.handle, comm.handle if comm is not None else None, int(lanes),
                                               int(max_queries), int(max_query_terms), int(depth), int(k), C.byref(h)))
        self.handle = h
        self._keep = {}   # ticket -> (inputs, outputs): host arrays / tensors the library still reads or writes

    def submit(self, query_vecs, query_terms, q_term_offsets, out: Optional[SearchResult] = None):
        dev, B, qv, qt, qo = self.index._queries(query_vecs, query_terms, q_term_offsets)
        if out is None:
            out = SearchResult(self.index._alloc(dev, (B, self.k), np.float32), self.index._alloc(dev, (B, self.k), np.uint32),
                               self.index._alloc(dev, (B,), np.uint32))
        t = C.c_uint64()
        _lib.check(self.lib.oi_pipeline_submit(self.handle, _lib.ptr(qv), _lib.ptr(qt), _lib.ptr(qo), B,
                                               _lib.OI_DEVICE if dev else _lib.OI_HOST, _lib.ptr(out.scores), _lib.ptr(out.docs),
                                               _lib.ptr(out.counts), C.byref(t)))
        self._keep[int(t.value)] = ((qv, qt, qo), out)
        if len(self._keep) > 64:
            for old in sorted(self._keep)[:-32]:      # (slots that old were reused long ago: the library is done with them)
                del self._keep[old]
        return int(t.value), out

    def wait(self, ticket: int, host_sync: bool = True) -> None:
        _lib.check(self.lib.oi_pipeline_wait(self.handle, int(ticket), 1 if host_sync else 0))
        if host_sync:
            self._keep.pop(int(ticket), None)

    def drain(self) -> None:
        _lib.check(self.lib.oi_pipeline_drain(self.handle))
        self._keep.clear()

    def workspace_bytes(self):
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.oi_pipeline_workspace_bytes(self.handle, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def concurrent_streams(self):
        """(how many of the pipeline's streams were measured to run at the same time, how many it has): oi_pipeline_concurrent_streams."""
        a, b = C.c_uint32(), C.c_uint32()
        _lib.check(self.lib.oi_pipeline_concurrent_streams(self.handle, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def profile_reset(self, enable) -> None:
        _lib.check(self.lib.oi_pipeline_profile_reset(self.handle, int(enable)))

    def profile_read(self, tag: str):
        """(summed ms, launches) of the lanes' launches with that tag since the last reset (oi_pipeline_profile_read)."""
        ms, n = C.c_double(), C.c_uint64()
        _lib.check(self.lib.oi_pipeline_profile_read(self.handle, tag.encode(), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.oi_pipeline_destroy(self.handle)
            self.handle = None
            self._keep = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- list-level ops (any ctx)
def rrf_fuse(ctx: HipContext, docs_a, counts_a, docs_b, counts_b, k: int) -> SearchResult:
    dev = _is_dev(docs_a)
    if not dev:
        docs_a, docs_b = _np(docs_a, np.uint32), _np(docs_b, np.uint32)
        counts_a, counts_b = _np(counts_a, np.uint32), _np(counts_b, np.uint32)
    B, depth = int(docs_a.shape[0]), int(docs_a.shape[1])
    if dev:
        import torch
        mk = lambda shape, dt: torch.zeros(shape, dtype=dt, device=docs_a.device)
        out = SearchResult(mk((B, k), torch.float32), mk((B, k), torch.int32), mk((B,), torch.int32))
    else:
        out = SearchResult(np.zeros((B, k), np.float32), np.zeros((B, k), np.uint32), np.zeros(B, np.uint32))
    _lib.check(ctx.lib.oi_rrf_fuse(ctx.handle, _lib.ptr(docs_a), _lib.ptr(counts_a), _lib.ptr(docs_b),
                                   _lib.ptr(counts_b), B, depth, int(k),
                                   _lib.OI_DEVICE if dev else _lib.OI_HOST, _lib.ptr(out.scores),
                                   _lib.ptr(out.docs), _lib.ptr(out.counts)))
    return out


def fuse_packed(ctx: HipContext, packed_all, n_shards: int, n_queries: int, depth: int, k: int,
                out: Optional[SearchResult] = None) -> SearchResult:
    """All shards' packed lists ([n_shards * OI_PACKED_WORDS] words) -> global top-depth per list -> RRF top-k."""
    dev = _is_dev(packed_all)
    if not dev:
        packed_all = np.ascontiguousarray(packed_all).view(np.uint32)
    if out is None:
        if dev:
            import torch
            mk = lambda shape, dt: torch.zeros(shape, dtype=dt, device=packed_all.device)
            out = SearchResult(mk((n_queries, k), torch.float32), mk((n_queries, k), torch.int32),
                               mk((n_queries,), torch.int32))
        else:
            out = SearchResult(np.zeros((n_queries, k), np.float32), np.zeros((n_queries, k), np.uint32),
                               np.zeros(n_queries, np.uint32))
    _lib.check(ctx.lib.oi_fuse_packed(ctx.handle, _lib.ptr(packed_all), int(n_shards), int(n_queries), int(depth),
                                      int(k), _lib.OI_DEVICE if dev else _lib.OI_HOST, _lib.ptr(out.scores),
                                      _lib.ptr(out.docs), _lib.ptr(out.counts)))
    return out


def merge_lists(ctx: HipContext, scores, docs, counts):
    """[S, B, depth] per-shard lists (+ counts [S, B]) -> global top-depth per query."""
    dev = _is_dev(scores)
    if not dev:
        scores, docs, counts = _np(scores, np.float32), _np(docs, np.uint32), _np(counts, np.uint32)
    S, B, depth = (int(x) for x in scores.shape)
    if dev:
        import torch
        mk = lambda shape, dt: torch.zeros(shape, dtype=dt, device=scores.device)
        so, do, co = mk((B, depth), torch.float32), mk((B, depth), torch.int32), mk((B,), torch.int32)
    else:
        so, do, co = np.zeros((B, depth), np.float32), np.zeros((B, depth), np.uint32), np.zeros(B, np.uint32)
    _lib.check(ctx.lib.oi_merge_lists(ctx.handle, _lib.ptr(scores), _lib.ptr(docs), _lib.ptr(counts), S, B, depth,
                                      _lib.OI_DEVICE if dev else _lib.OI_HOST, _lib.ptr(so), _lib.ptr(do),
                                      _lib.ptr(co)))
    return so, do, co
