"""PostAnalyzer port + its MI355X implementation.

Mirrors the reference's port and adapter (paths relative to the openintel repo):

    trait PostAnalyzer { async fn analyze(&self, posts: &[SocialPost])
                         -> Result<Vec<PostSignal>, DomainError>; }
                                         src/domain/ports/post_analyzer.rs:7-11
    impl PostAnalyzer for LexiconAnalyzer    src/adapters/analyzer/lexicon.rs:82-87

Contract (post_analyzer.rs:9): one PostSignal per input post, aligned to input order.
"""
from __future__ import annotations

import abc
import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib
from .context import HipContext
from .domain import PostSignal, SocialPost, SourceFailure


# oi_social_counters as a numpy record (64 bytes): what the segmented summary writes, one per segment
COUNTERS_DTYPE = np.dtype([("total", "<u8"), ("by_source", "<u8", (2,)), ("bullish", "<u8"), ("bearish", "<u8"),
                           ("neutral", "<u8"), ("spec_count", "<u8"), ("polarity_sum", "<f8")])
assert COUNTERS_DTYPE.itemsize == C.sizeof(_lib.SocialCounters)


def counters_record(rec) -> "_lib.SocialCounters":
    """One record of a COUNTERS_DTYPE array as the ctypes struct SpeculationEngine.aggregate_counters takes."""
    c = _lib.SocialCounters()
    c.total, c.bullish, c.bearish = int(rec["total"]), int(rec["bullish"]), int(rec["bearish"])
    c.neutral, c.spec_count, c.polarity_sum = int(rec["neutral"]), int(rec["spec_count"]), float(rec["polarity_sum"])
    c.by_source[0], c.by_source[1] = int(rec["by_source"][0]), int(rec["by_source"][1])
    return c


class PostAnalyzer(abc.ABC):
    @abc.abstractmethod
    def analyze(self, posts: Sequence[SocialPost]) -> List[PostSignal]:
        ...


def pack_posts(texts: Sequence[str]):
    """Gather post texts into the FFI layout: one UTF-8 blob + (n+1) u64 offsets.
    (In the reference posts are separate heap strings, social_post.rs:25-27.)"""
    enc = [t.encode("utf-8") for t in texts]
    offsets = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offsets[1:] = np.cumsum(np.fromiter((len(e) for e in enc), dtype=np.uint64, count=len(enc)))
    blob = np.frombuffer(b"".join(enc), dtype=np.uint8)
    return blob, offsets


class HipLexiconAnalyzer(PostAnalyzer):
    """LexiconAnalyzer on the GPU (oi_lexicon_analyze).  No CPU fallback."""

    def __init__(self, ctx: HipContext):
        self.ctx = ctx

    def analyze_packed(self, blob: np.ndarray, offsets: np.ndarray):
        """Host buffers in, host arrays out: (polarity f64[n], speculative u8[n])."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        pol = np.zeros(n, dtype=np.float64)
        spec = np.zeros(n, dtype=np.uint8)
        if n == 0:
            return pol, spec
        rc = self.ctx.lib.oi_lexicon_analyze(self.ctx.handle, _lib.ptr(blob) if blob.size else None,
                                             _lib.ptr(offsets), n, _lib.ptr(pol), _lib.ptr(spec))
        if rc != 0:
            msg = self.ctx.lib.oi_last_error().decode("utf-8", "replace")
            raise SourceFailure("hip-analyzer", msg)  # the mapping INTEGRATION.md prescribes
        return pol, spec

    def score_texts(self, texts: Sequence[str]):
        """(polarity f64[n], speculative u8[n]) of plain strings: pack_posts + analyze_packed."""
        return self.analyze_packed(*pack_posts(texts))

    def analyze_device(self, d_blob, d_offsets, d_polarity, d_speculative) -> None:
        """torch CUDA tensors in HBM (uint8 blob, int64/uint64 offsets[n+1], float64[n], uint8[n]);
        asynchronous on the ctx stream."""
        n = d_offsets.numel() - 1
        _lib.check(self.ctx.lib.oi_lexicon_analyze_device(
            self.ctx.handle, _lib.ptr(d_blob), _lib.ptr(d_offsets), n, d_blob.numel(),
            _lib.ptr(d_polarity), _lib.ptr(d_speculative)))

    def summary_device(self, d_blob, d_offsets, d_sources=None, tau: float = 0.2, d_polarity=None, d_speculative=None):
        """The scan and the social_summary reduction in one pass (oi_lexicon_summary_device): returns the raw sums
        (_lib.SocialCounters); the per-post outputs are written only if their tensors are given."""
        n = d_offsets.numel() - 1
        out = _lib.SocialCounters()
        _lib.check(self.ctx.lib.oi_lexicon_summary_device(
            self.ctx.handle, _lib.ptr(d_blob), _lib.ptr(d_offsets), n, d_blob.numel(), _lib.ptr(d_sources), float(tau),
            _lib.ptr(d_polarity), _lib.ptr(d_speculative), C.byref(out)))
        return out

    def summary_segments(self, sources, polarity, speculative, seg_offsets, tau: float = 0.2) -> np.ndarray:
        """Per-segment social_summary sums of a pooled batch (oi_social_summary_segmented, host buffers): segment s =
        signals [seg_offsets[s], seg_offsets[s+1]).  Returns a COUNTERS_DTYPE array; polarity_sum is the reference's
        input-order sum, bit for bit."""
        polarity = np.ascontiguousarray(polarity, dtype=np.float64)
        speculative = np.ascontiguousarray(speculative, dtype=np.uint8)
        seg_offsets = np.ascontiguousarray(seg_offsets, dtype=np.uint64)
        src = None if sources is None else np.ascontiguousarray(sources, dtype=np.uint8)
        n_seg = max(0, seg_offsets.size - 1)
        out = np.zeros(n_seg, dtype=COUNTERS_DTYPE)
        if n_seg:
            _lib.check(self.ctx.lib.oi_social_summary_segmented(
                self.ctx.handle, _lib.ptr(src) if src is not None and src.size else None,
                _lib.ptr(polarity) if polarity.size else None, _lib.ptr(speculative) if speculative.size else None,
                polarity.size, _lib.ptr(seg_offsets), n_seg, float(tau), _lib.OI_HOST, _lib.ptr(out)))
        return out

    def scan_segments_device(self, d_blob, d_offsets, d_sources, d_seg_offsets, d_out, tau: float = 0.2,
                             d_polarity=None, d_speculative=None) -> None:
        """The pooled posts of many tickers in, one 64-byte counters record per ticker out (torch CUDA tensors;
        d_out: uint8[n_segments * 64] or int64[n_segments * 8]); oi_lexicon_scan_segments_device, asynchronous."""
        n = d_offsets.numel() - 1
        n_seg = d_seg_offsets.numel() - 1
        _lib.check(self.ctx.lib.oi_lexicon_scan_segments_device(
            self.ctx.handle, _lib.ptr(d_blob), _lib.ptr(d_offsets), n, d_blob.numel(), _lib.ptr(d_sources),
            _lib.ptr(d_seg_offsets), n_seg, float(tau), _lib.ptr(d_polarity), _lib.ptr(d_speculative), _lib.ptr(d_out)))

    def analyze_segments(self, segments: Sequence[Sequence[SocialPost]], tau: float = 0.2):
        """The batch callers' form (tools.rs:193-225): the posts of all tickers through ONE scan, then every ticker's
        sums from one reduction.  Returns (signals per segment, COUNTERS_DTYPE array)."""
        flat = [p for seg in segments for p in seg]
        seg_offsets = np.zeros(len(segments) + 1, dtype=np.uint64)
        if segments:
            seg_offsets[1:] = np.cumsum([len(seg) for seg in segments])
        pol, spec = self.analyze_packed(*pack_posts([p.text.as_str() for p in flat]))
        sources = np.fromiter((int(p.source) for p in flat), dtype=np.uint8, count=len(flat))
        try:
            counters = self.summary_segments(sources, pol, spec, seg_offsets, tau)
        except _lib.OiError as e:  # the same mapping as the scan's failures: DomainError::SourceFailure
            raise SourceFailure("hip-analyzer", e.message)
        signals = [[PostSignal(float(pol[i]), bool(spec[i])) for i in range(int(seg_offsets[k]), int(seg_offsets[k + 1]))]
                   for k in range(len(segments))]
        return signals, counters

    def analyze(self, posts: Sequence[SocialPost]) -> List[PostSignal]:
        blob, offsets = pack_posts([p.text.as_str() for p in posts])
        pol, spec = self.analyze_packed(blob, offsets)
        return [PostSignal(float(p), bool(s)) for p, s in zip(pol, spec)]
