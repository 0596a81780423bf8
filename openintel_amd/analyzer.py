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

    def analyze(self, posts: Sequence[SocialPost]) -> List[PostSignal]:
        blob, offsets = pack_posts([p.text.as_str() for p in posts])
        pol, spec = self.analyze_packed(blob, offsets)
        return [PostSignal(float(p), bool(s)) for p, s in zip(pol, spec)]
