"""ctypes/numpy wrapper over liboi_oracle.so -- TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboi_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C oracle (gcc, seconds).  Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("oi_oracle.c", "oi_oracle.h")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src
    )
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


class EngineConfig(C.Structure):
    _fields_ = [
        ("bull_bear_threshold", C.c_double),
        ("net_sentiment_threshold", C.c_double),
        ("price_move_threshold", C.c_double),
        ("crowding_weight_spec", C.c_double),
        ("crowding_weight_rvol", C.c_double),
        ("crowding_weight_iv", C.c_double),
        ("rvol_cap", C.c_double),
        ("min_sample", C.c_uint64),
        ("confidence_low", C.c_uint64),
        ("confidence_high", C.c_uint64),
    ]


class SocialSummary(C.Structure):
    _fields_ = [
        ("total_mentions", C.c_uint64),
        ("mentions_by_source", C.c_uint64 * 2),
        ("net_sentiment", C.c_double),
        ("bullish", C.c_uint64),
        ("bearish", C.c_uint64),
        ("neutral", C.c_uint64),
        ("has_bull_bear_ratio", C.c_int),
        ("bull_bear_ratio", C.c_double),
        ("speculation_index", C.c_double),
        ("spec_count", C.c_uint64),
        ("polarity_sum", C.c_double),
    ]


class MarketSnapshot(C.Structure):
    _fields_ = [
        ("last_price", C.c_double),
        ("previous_close", C.c_double),
        ("volume", C.c_uint64),
        ("avg_volume", C.c_uint64),
        ("has_realized_vol", C.c_int),
        ("realized_vol", C.c_double),
        ("has_put_call_ratio", C.c_int),
        ("put_call_ratio", C.c_double),
        ("has_iv_rank", C.c_int),
        ("iv_rank", C.c_double),
    ]


class MarketSummary(C.Structure):
    _fields_ = [
        ("last_price", C.c_double),
        ("pct_change", C.c_double),
        ("has_rvol", C.c_int),
        ("rvol", C.c_double),
        ("has_realized_vol", C.c_int),
        ("realized_vol", C.c_double),
        ("has_put_call_ratio", C.c_int),
        ("put_call_ratio", C.c_double),
        ("has_iv_rank", C.c_int),
        ("iv_rank", C.c_double),
        ("note_previous_close_zero", C.c_int),
        ("note_avg_volume_zero", C.c_int),
    ]


class Report(C.Structure):
    _fields_ = [
        ("social", SocialSummary),
        ("has_market", C.c_int),
        ("market", MarketSummary),
        ("alignment", C.c_int),
        ("crowding", C.c_double),
        ("note_social_only", C.c_int),
        ("social_confidence", C.c_int),
    ]


ALIGNMENT_NAMES = ["confirming_bullish", "confirming_bearish", "diverging", "quiet"]
CONFIDENCE_NAMES = ["low", "medium", "high"]
ERR_ANALYZER_MISMATCH = -3
ERR_MARKET_TICKER_MISMATCH = -4

_lib: Optional[C.CDLL] = None


def _L() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(_SO)
        lib.oio_polarity_new.restype = C.c_double
        lib.oio_polarity_new.argtypes = [C.c_double]
        lib.oio_speculation_index_new.restype = C.c_double
        lib.oio_speculation_index_new.argtypes = [C.c_double]
        lib.oio_confidence_from_sample.restype = C.c_int
        lib.oio_confidence_from_sample.argtypes = [C.c_uint64] * 3
        lib.oio_crowding.restype = C.c_double
        lib.oio_bm25_idf.restype = C.c_float
        lib.oio_bm25_idf.argtypes = [C.c_uint64, C.c_uint64]
        lib.oio_bm25_avgdl.restype = C.c_float
        lib.oio_bm25_avgdl.argtypes = [C.c_uint64, C.c_uint64]
        lib.oio_bm25_doc_norm.restype = C.c_float
        lib.oio_bm25_doc_norm.argtypes = [C.c_uint32, C.c_float]
        lib.oio_bm25_impact.restype = C.c_float
        lib.oio_bm25_impact.argtypes = [C.c_uint32, C.c_float]
        lib.oio_topk.restype = C.c_uint32
        lib.oio_merge_ranked.restype = C.c_uint32
        lib.oio_rrf_fuse.restype = C.c_uint32
        lib.oio_hybrid_search_batch.restype = C.c_int
        lib.oio_hybrid_search_batch_blocked.restype = C.c_int
        lib.oio_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- (1) pinned
def default_config() -> EngineConfig:
    cfg = EngineConfig()
    _L().oio_engine_config_default(C.byref(cfg))
    return cfg


def polarity_new(v: float) -> float:
    return _L().oio_polarity_new(v)


def speculation_index_new(v: float) -> float:
    return _L().oio_speculation_index_new(v)


def confidence_from_sample(n: int, low: int, high: int) -> str:
    return CONFIDENCE_NAMES[_L().oio_confidence_from_sample(n, low, high)]


def pack_texts(texts: Sequence[bytes | str]):
    """Gather posts into one blob + (n+1) u64 offsets (the FFI layout)."""
    enc = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in texts]
    offsets = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offsets[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
    blob = np.frombuffer(b"".join(enc), dtype=np.uint8).copy()
    if blob.size == 0:
        blob = np.zeros(1, dtype=np.uint8)[:0]
    return blob, offsets


def lexicon_score(text: bytes | str):
    """(polarity, speculative, bull_hits, bear_hits) of one post."""
    b = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    buf = np.frombuffer(b, dtype=np.uint8) if b else np.zeros(0, np.uint8)
    pol, spec = C.c_double(), C.c_uint8()
    bull, bear = C.c_uint32(), C.c_uint32()
    _L().oio_lexicon_score(_p(buf), C.c_uint64(len(b)), C.byref(pol), C.byref(spec),
                           C.byref(bull), C.byref(bear))
    return pol.value, bool(spec.value), bull.value, bear.value


def lexicon_analyze(blob: np.ndarray, offsets: np.ndarray, n_threads: int = 1):
    """(polarity f64[n], speculative u8[n]).  n_threads > 1: the same scalar `score` per post on that many host threads
    (bench.py's all-cores baseline; the reference itself maps on one thread, lexicon.rs:82-87)."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    pol = np.empty(n, dtype=np.float64)
    spec = np.empty(n, dtype=np.uint8)
    if n_threads > 1:
        _L().oio_lexicon_analyze_mt(_p(blob), _p(offsets), C.c_uint64(n), _p(pol), _p(spec), C.c_int(n_threads))
        return pol, spec
    rc = _L().oio_lexicon_analyze(_p(blob), _p(offsets), C.c_uint64(n), _p(pol), _p(spec))
    assert rc == 0
    return pol, spec


# ---- headline gate (dip.rs:204-272) ----------------------------------------------
def catalyst_keywords() -> List[str]:
    arr = (C.c_char_p * 16).in_dll(_L(), "OIO_CATALYST_KEYWORDS")
    return [a.decode() for a in arr]


def pack_forms(forms: Sequence[bytes | str]):
    enc = [f.encode("utf-8") if isinstance(f, str) else bytes(f) for f in forms]
    offs = np.zeros(len(enc) + 1, dtype=np.uint32)
    if enc:
        offs[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint32)
    blob = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy()
    return blob, offs


def headline_scan(blob: np.ndarray, offsets: np.ndarray, ticker: bytes | str, forms: Sequence[bytes | str], n_threads: int = 1):
    """(mask u16[n], order u64[n], about u8[n]) over a batch of titles (n_threads > 1: the same scalar functions per title on
    that many host threads -- bench.py's all-cores baseline)."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    if blob.size == 0:
        blob = np.zeros(1, np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    tk = ticker.encode("utf-8") if isinstance(ticker, str) else bytes(ticker)
    tkb = np.frombuffer(tk + b"\0", dtype=np.uint8)
    fblob, foffs = pack_forms(forms)
    mask = np.zeros(n, np.uint16)
    order = np.zeros(n, np.uint64)
    about = np.zeros(n, np.uint8)
    if n_threads > 1:
        _L().oio_headline_scan_mt(_p(blob), _p(offsets), C.c_uint64(n), _p(tkb), C.c_uint64(len(tk)), _p(fblob),
                                  _p(foffs), C.c_uint32(len(forms)), _p(mask), _p(order), _p(about), C.c_int(n_threads))
        return mask, order, about
    _L().oio_headline_scan(_p(blob), _p(offsets), C.c_uint64(n), _p(tkb), C.c_uint64(len(tk)), _p(fblob),
                           _p(foffs), C.c_uint32(len(forms)), _p(mask), _p(order), _p(about))
    return mask, order, about


def hits_from_order(mask: int, order: int) -> List[str]:
    kw = catalyst_keywords()
    return [kw[(int(order) >> (4 * j)) & 15] for j in range(bin(int(mask)).count("1"))]


def catalyst_hits(texts: Sequence[bytes | str]) -> List[str]:
    """dip.rs:261-272: keyword hits across the texts, deduped, first-occurrence order."""
    blob, offs = pack_texts(texts)
    mask, order, _ = headline_scan(blob, offs, b"", [])
    hits: List[str] = []
    for m, o in zip(mask, order):
        for h in hits_from_order(m, o):
            if h not in hits:
                hits.append(h)
    return hits


def headline_mentions_company(title: bytes | str, ticker: bytes | str, forms: Sequence[bytes | str]) -> bool:
    blob, offs = pack_texts([title])
    return bool(headline_scan(blob, offs, ticker, forms)[2][0])


def social_summary(sources, polarity, speculative, cfg: Optional[EngineConfig] = None) -> SocialSummary:
    sources = np.ascontiguousarray(sources, dtype=np.uint8)
    polarity = np.ascontiguousarray(polarity, dtype=np.float64)
    speculative = np.ascontiguousarray(speculative, dtype=np.uint8)
    assert sources.size == polarity.size == speculative.size
    cfg = cfg or default_config()
    out = SocialSummary()
    _L().oio_social_summary_compute(_p(sources), _p(polarity), _p(speculative),
                                    C.c_uint64(polarity.size), C.byref(cfg), C.byref(out))
    return out


def social_summary_segmented(sources, polarity, speculative, seg_offsets, cfg: Optional[EngineConfig] = None):
    """social_summary of every segment [seg_offsets[s], seg_offsets[s+1]) of a pooled batch: the per-ticker loop of
    run_scan / run_compare (mcp/tools.rs:193-225, :303-352).  Returns a ctypes array of SocialSummary."""
    sources = np.ascontiguousarray(sources, dtype=np.uint8)
    polarity = np.ascontiguousarray(polarity, dtype=np.float64)
    speculative = np.ascontiguousarray(speculative, dtype=np.uint8)
    seg = np.ascontiguousarray(seg_offsets, dtype=np.uint64)
    assert sources.size == polarity.size == speculative.size
    n_seg = seg.size - 1
    assert n_seg >= 0 and (n_seg == 0 or (np.all(seg[1:] >= seg[:-1]) and int(seg[-1]) <= polarity.size))
    cfg = cfg or default_config()
    out = (SocialSummary * max(n_seg, 1))()
    if n_seg:
        _L().oio_social_summary_segmented(_p(sources), _p(polarity), _p(speculative), _p(seg), C.c_uint64(n_seg),
                                          C.byref(cfg), out)
    return out[:n_seg] if n_seg else []


def aggregate(ticker: str, sources, polarity, speculative, market: Optional[MarketSnapshot] = None,
              market_ticker: Optional[str] = None, cfg: Optional[EngineConfig] = None):
    """Returns (rc, Report).  len(sources) models posts.len(); len(polarity) signals.len()."""
    sources = np.ascontiguousarray(sources, dtype=np.uint8)
    polarity = np.ascontiguousarray(polarity, dtype=np.float64)
    speculative = np.ascontiguousarray(speculative, dtype=np.uint8)
    cfg = cfg or default_config()
    out = Report()
    rc = _L().oio_aggregate(
        ticker.encode(), _p(sources), C.c_uint64(sources.size), _p(polarity), _p(speculative),
        C.c_uint64(polarity.size), C.byref(market) if market is not None else None,
        (market_ticker if market_ticker is not None else ticker).encode() if market is not None else None,
        C.byref(cfg), C.byref(out))
    return rc, out


def make_snapshot(last, prev, volume, avg_volume, realized_vol=None, put_call_ratio=None,
                  iv_rank=None) -> MarketSnapshot:
    m = MarketSnapshot()
    m.last_price, m.previous_close, m.volume, m.avg_volume = last, prev, volume, avg_volume
    for name, v in (("realized_vol", realized_vol), ("put_call_ratio", put_call_ratio),
                    ("iv_rank", iv_rank)):
        setattr(m, "has_" + name, 0 if v is None else 1)
        setattr(m, name, 0.0 if v is None else v)
    return m


# ------------------------------------------------------- (2) parity unpinned
BM25_K1 = 1.2
BM25_B = 0.75
RRF_K = 60.0


def bm25_idf(n_docs: int, df: int) -> np.float32:
    return np.float32(_L().oio_bm25_idf(n_docs, df))


def bm25_df(term_ids, doc_offsets, vocab: int):
    term_ids = np.ascontiguousarray(term_ids, dtype=np.uint32)
    doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
    df = np.zeros(vocab, dtype=np.uint32)
    tot = C.c_uint64()
    _L().oio_bm25_df(_p(term_ids), _p(doc_offsets), C.c_uint64(doc_offsets.size - 1),
                     C.c_uint32(vocab), _p(df), C.byref(tot))
    return df, tot.value


def bm25_scores(term_ids, doc_offsets, vocab: int, query_terms, df=None, n_docs_global=None,
                total_tokens_global=None) -> np.ndarray:
    term_ids = np.ascontiguousarray(term_ids, dtype=np.uint32)
    doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
    q = np.ascontiguousarray(query_terms, dtype=np.uint32)
    n = doc_offsets.size - 1
    if df is not None:
        df = np.ascontiguousarray(df, dtype=np.uint32)
    out = np.empty(n, dtype=np.float32)
    _L().oio_bm25_scores(
        _p(term_ids), _p(doc_offsets), C.c_uint64(n), C.c_uint32(vocab),
        _p(df) if df is not None else None,
        C.c_uint64(n if n_docs_global is None else n_docs_global),
        C.c_uint64(int(doc_offsets[-1]) if total_tokens_global is None else total_tokens_global),
        _p(q), C.c_uint32(q.size), _p(out))
    return out


def l2_normalize_rows(rows: np.ndarray) -> np.ndarray:
    rows = np.array(rows, dtype=np.float32, order="C", copy=True)
    _L().oio_l2_normalize_rows(_p(rows), C.c_uint64(rows.shape[0]), C.c_uint32(rows.shape[1]))
    return rows


def dot_scores(rows: np.ndarray, q: np.ndarray) -> np.ndarray:
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    assert rows.shape[1] == q.size
    out = np.empty(rows.shape[0], dtype=np.float32)
    _L().oio_dot_scores(_p(rows), C.c_uint64(rows.shape[0]), C.c_uint32(rows.shape[1]), _p(q), _p(out))
    return out


def topk(scores: np.ndarray, k: int, positive_only: bool = False, doc_base: int = 0):
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    so = np.empty(k, dtype=np.float32)
    do = np.empty(k, dtype=np.uint32)
    n = _L().oio_topk(_p(scores), C.c_uint64(scores.size), C.c_uint32(k), C.c_int(int(positive_only)),
                      C.c_uint32(doc_base), _p(so), _p(do))
    return so[:n].copy(), do[:n].copy()


def merge_ranked(score_lists, doc_lists, depth: int):
    sl = [np.ascontiguousarray(s, dtype=np.float32) for s in score_lists]
    dl = [np.ascontiguousarray(d, dtype=np.uint32) for d in doc_lists]
    n = len(sl)
    sp = (C.c_void_p * n)(*[s.ctypes.data for s in sl])
    dp = (C.c_void_p * n)(*[d.ctypes.data for d in dl])
    counts = np.array([s.size for s in sl], dtype=np.uint32)
    so = np.empty(depth, dtype=np.float32)
    do = np.empty(depth, dtype=np.uint32)
    m = _L().oio_merge_ranked(sp, dp, _p(counts), C.c_uint32(n), C.c_uint32(depth), _p(so), _p(do))
    return so[:m].copy(), do[:m].copy()


def rrf_fuse(docs_a, docs_b, k: int):
    a = np.ascontiguousarray(docs_a, dtype=np.uint32)
    b = np.ascontiguousarray(docs_b, dtype=np.uint32)
    so = np.empty(max(k, 1), dtype=np.float32)
    do = np.empty(max(k, 1), dtype=np.uint32)
    m = _L().oio_rrf_fuse(_p(a), C.c_uint32(a.size), _p(b), C.c_uint32(b.size), C.c_uint32(k),
                          _p(so), _p(do))
    return so[:m].copy(), do[:m].copy()


def hybrid_search(rows_normalized: np.ndarray, term_ids, doc_offsets, vocab: int, query_vec,
                  query_terms, k: int, depth: int, doc_base: int = 0):
    """Single-query hybrid reference: cosine top-depth + BM25 top-depth -> RRF top-k."""
    cs = dot_scores(rows_normalized, query_vec)
    c_s, c_d = topk(cs, depth, False, doc_base)
    bs = bm25_scores(term_ids, doc_offsets, vocab, query_terms)
    b_s, b_d = topk(bs, depth, True, doc_base)
    f_s, f_d = rrf_fuse(c_d, b_d, k)
    return dict(cos=(c_s, c_d), bm25=(b_s, b_d), fused=(f_s, f_d))


CFLAGS = "gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp"   # oracle/Makefile


def max_threads() -> int:
    return int(_L().oio_max_threads())


def hybrid_search_batch(rows_normalized, term_ids, doc_offsets, vocab: int, query_vecs, query_terms, q_term_offsets,
                        k: int, depth: int, n_threads: int = 1, df=None, blocked: bool = False):
    """A whole batch through the scalar pipeline, on n_threads host threads (blocked=False: parallel over queries, every
    query streams the corpus; blocked=True: rows / docs outermost and split over the threads, every row block scored
    against all queries while cached -- identical results either way, for any thread count).
    Returns (scores [B,k], docs [B,k], counts [B], threads_used)."""
    rows = np.ascontiguousarray(rows_normalized, dtype=np.float32)
    t = np.ascontiguousarray(term_ids, dtype=np.uint32)
    o = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
    qv = np.ascontiguousarray(query_vecs, dtype=np.float32)
    qt = np.ascontiguousarray(query_terms, dtype=np.uint32)
    qo = np.ascontiguousarray(q_term_offsets, dtype=np.uint32)
    if qt.size == 0:
        qt = np.zeros(1, np.uint32)
    B = qv.shape[0]
    if df is not None:
        df = np.ascontiguousarray(df, dtype=np.uint32)
    so, do, co = np.zeros((B, k), np.float32), np.zeros((B, k), np.uint32), np.zeros(B, np.uint32)
    fn = _L().oio_hybrid_search_batch_blocked if blocked else _L().oio_hybrid_search_batch
    used = fn(_p(rows), C.c_uint64(rows.shape[0]), C.c_uint32(rows.shape[1]), _p(t), _p(o),
                                        C.c_uint32(vocab), _p(df) if df is not None else None, _p(qv), _p(qt), _p(qo),
                                        C.c_uint32(B), C.c_uint32(depth), C.c_uint32(k), C.c_int(n_threads), _p(so),
                                        _p(do), _p(co))
    return so, do, co, int(used)
