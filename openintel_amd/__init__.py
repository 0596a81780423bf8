"""openintel_amd -- MI355X-native hot path for Kloudy-Sky/openintel's per-post analysis, plus the
hybrid retrieval path (cosine + BM25 + RRF) BASELINE.json names.

Everything that computes runs in libopenintel_hip.so (hand-written HIP for gfx950) behind the
C ABI of include/openintel_hip.h; this package is the host-side mirror of the reference's
port/adapter interface.  There is no CPU fallback: without the library or a gfx950 device the
calls raise.
"""
from . import _lib
from .analyzer import HipLexiconAnalyzer, PostAnalyzer, pack_posts
from .context import HipContext
from .dip import HeadlineScanner, company_name_forms
from .domain import (Alignment, AnalyzerMismatch, Confidence, DomainError, EngineConfig, MarketSnapshot,
                     PostSignal, PostText, SocialPost, SourceFailure, SourceKind, Ticker)
from .engine import SpeculationEngine
from .sharded import ShardedAnalyzer, ShardedPipeline, ShardedRetriever, make_hip_sharded, make_hip_sharded_analyzer, shard_bounds
from .retriever import (HybridIndex, NativeComm, NativePipeline, PostRetriever, SearchResult, fuse_packed, merge_lists, pack_query_terms,
                        packed_words, rrf_fuse, unpack_lists)

__all__ = [
    "HipContext", "HipLexiconAnalyzer", "PostAnalyzer", "pack_posts", "SpeculationEngine", "HybridIndex",
    "PostRetriever", "SearchResult", "merge_lists", "rrf_fuse", "pack_query_terms", "fuse_packed", "packed_words",
    "unpack_lists", "HeadlineScanner", "company_name_forms", "Alignment",
    "AnalyzerMismatch", "Confidence", "DomainError", "EngineConfig", "MarketSnapshot", "PostSignal", "PostText",
    "SocialPost", "SourceFailure", "SourceKind", "Ticker", "ShardedAnalyzer", "ShardedPipeline", "ShardedRetriever", "make_hip_sharded",
    "make_hip_sharded_analyzer", "shard_bounds", "NativeComm", "NativePipeline",
]
