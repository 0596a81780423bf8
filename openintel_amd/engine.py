"""SpeculationEngine -- host mirror of the reference's aggregate step, plus the GPU reduction.

    SpeculationEngine::aggregate       src/domain/engine/speculation_engine.rs:21-68
    social_summary                     :70-125
    market_summary                     :127-148
    crowding                           :151-176
    alignment                          :178-208

`aggregate(posts, signals, ...)` is the reference's function: signals on the host, the
polarity sum taken in INPUT order (np.add.accumulate is a sequential scan, so the result is
bit-identical to the Rust loop at :83-86).

`aggregate_counters(...)` finishes the same report from the raw sums that
oi_social_summary reduced on the GPU (integer fields exact; polarity_sum a fixed-shape tree
sum, see DESIGN.md for the bound).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .context import HipContext
from .domain import (Alignment, AnalyzerMismatch, Confidence, EngineConfig, FusionSignals, MarketSnapshot,
                     MarketSummary, MarketTickerMismatch, PostSignal, SocialPost, SocialSummary, SourceKind,
                     SpeculationReport, Ticker, _clamp, polarity_new, speculation_index_new)


class SpeculationEngine:
    # ------------------------------------------------------------------ reference API
    @staticmethod
    def aggregate(ticker: Ticker, posts: Sequence[SocialPost], signals: Sequence[PostSignal],
                  market: Optional[MarketSnapshot], now, cfg: EngineConfig) -> SpeculationReport:
        if len(signals) != len(posts):  # :29-34
            raise AnalyzerMismatch(expected=len(posts), got=len(signals))
        SpeculationEngine._check_market_ticker(ticker, market)  # :36-43
        sources = np.fromiter((int(p.source) for p in posts), dtype=np.uint8, count=len(posts))
        pol = np.fromiter((s.polarity for s in signals), dtype=np.float64, count=len(signals))
        spec = np.fromiter((s.speculative for s in signals), dtype=np.uint8, count=len(signals))
        social = SpeculationEngine.social_summary_arrays(sources, pol, spec, cfg)
        return SpeculationEngine._finish(ticker, social, market, now, cfg)

    @staticmethod
    def social_summary_arrays(sources: np.ndarray, polarity: np.ndarray, speculative: np.ndarray,
                              cfg: EngineConfig) -> SocialSummary:
        """speculation_engine.rs:70-125 on arrays (same arithmetic, input-order sum)."""
        total = int(polarity.size)
        by_source = {}
        for kind in SourceKind.ALL:  # BTreeMap: only present keys, in Ord order
            c = int((sources == int(kind)).sum())
            if c:
                by_source[kind] = c
        tau = cfg.bull_bear_threshold
        bullish = int((polarity > tau).sum())
        bearish = int((polarity < -tau).sum())
        neutral = total - bullish - bearish
        spec_count = int((speculative != 0).sum())
        polarity_sum = float(np.add.accumulate(polarity)[-1]) if total else 0.0
        return SpeculationEngine._social_from_sums(total, by_source, bullish, bearish, neutral, spec_count,
                                                   polarity_sum)

    # ------------------------------------------------------------------ GPU reduction
    @staticmethod
    def social_counters(ctx: HipContext, sources, polarity, speculative, cfg: EngineConfig,
                        n_posts: Optional[int] = None) -> _lib.SocialCounters:
        """oi_social_summary.  Arguments are numpy arrays (host) or torch CUDA tensors (HBM)."""
        on_device = hasattr(polarity, "data_ptr")
        n_sig = int(polarity.numel() if on_device else polarity.size)
        n_posts = n_sig if n_posts is None else int(n_posts)
        out = _lib.SocialCounters()
        rc = ctx.lib.oi_social_summary(ctx.handle, _lib.ptr(sources), n_posts, _lib.ptr(polarity),
                                       _lib.ptr(speculative), n_sig, float(cfg.bull_bear_threshold),
                                       _lib.OI_DEVICE if on_device else _lib.OI_HOST, C.byref(out))
        if rc == _lib.OI_ERR_ANALYZER_MISMATCH:
            raise AnalyzerMismatch(expected=n_posts, got=n_sig)
        _lib.check(rc)
        return out

    @staticmethod
    def aggregate_counters(ticker: Ticker, counters: _lib.SocialCounters, market: Optional[MarketSnapshot], now,
                           cfg: EngineConfig) -> SpeculationReport:
        SpeculationEngine._check_market_ticker(ticker, market)
        by_source = {k: int(counters.by_source[int(k)]) for k in SourceKind.ALL if counters.by_source[int(k)]}
        social = SpeculationEngine._social_from_sums(int(counters.total), by_source, int(counters.bullish),
                                                     int(counters.bearish), int(counters.neutral),
                                                     int(counters.spec_count), float(counters.polarity_sum))
        return SpeculationEngine._finish(ticker, social, market, now, cfg)

    # ------------------------------------------------------------------ shared pieces
    @staticmethod
    def _check_market_ticker(ticker: Ticker, market: Optional[MarketSnapshot]) -> None:
        if market is not None and market.ticker.as_str() != ticker.as_str():
            raise MarketTickerMismatch(expected=ticker.as_str(), got=market.ticker.as_str())

    @staticmethod
    def _social_from_sums(total, by_source, bullish, bearish, neutral, spec_count, polarity_sum) -> SocialSummary:
        net = 0.0 if total == 0 else polarity_sum / float(total)                 # :99-103
        spec_index = 0.0 if total == 0 else float(spec_count) / float(total)     # :104-108
        ratio = None if bearish == 0 else float(bullish) / float(bearish)        # :109-113
        return SocialSummary(total_mentions=total, mentions_by_source=by_source,
                             net_sentiment=polarity_new(net), bullish=bullish, bearish=bearish,
                             neutral=neutral, bull_bear_ratio=ratio,
                             speculation_index=speculation_index_new(spec_index))

    @staticmethod
    def market_summary(m: MarketSnapshot, notes: List[str]) -> MarketSummary:   # :127-148
        if m.previous_close == 0.0:
            notes.append("previous_close is 0; pct_change set to 0")
            pct = 0.0
        else:
            pct = (m.last_price - m.previous_close) / m.previous_close * 100.0
        if m.avg_volume == 0:
            notes.append("avg_volume is 0; rvol unavailable")
            rvol = None
        else:
            rvol = float(m.volume) / float(m.avg_volume)
        return MarketSummary(m.last_price, pct, rvol, m.realized_vol, m.put_call_ratio, m.iv_rank)

    @staticmethod
    def crowding(social: SocialSummary, market: Optional[MarketSummary], cfg: EngineConfig) -> float:  # :151-176
        weighted = 0.0
        weight_sum = 0.0
        if social.total_mentions > 0:
            weighted += cfg.crowding_weight_spec * social.speculation_index
            weight_sum += cfg.crowding_weight_spec
        if market is not None:
            if market.rvol is not None:
                weighted += cfg.crowding_weight_rvol * _clamp(market.rvol / cfg.rvol_cap, 0.0, 1.0)
                weight_sum += cfg.crowding_weight_rvol
            if market.iv_rank is not None:
                weighted += cfg.crowding_weight_iv * _clamp(market.iv_rank, 0.0, 1.0)
                weight_sum += cfg.crowding_weight_iv
        if weight_sum == 0.0:
            return 0.0
        return _clamp(weighted / weight_sum, 0.0, 1.0)

    @staticmethod
    def alignment(social: SocialSummary, market: Optional[MarketSummary], cfg: EngineConfig,
                  notes: List[str]) -> Alignment:  # :178-208
        if market is None:
            notes.append("social-only, no price reference")
            return Alignment.QUIET
        if social.total_mentions < cfg.min_sample:
            return Alignment.QUIET
        s, p = social.net_sentiment, market.pct_change
        if not (abs(s) >= cfg.net_sentiment_threshold) or not (abs(p) >= cfg.price_move_threshold):
            return Alignment.QUIET
        if s > 0.0 and p > 0.0:
            return Alignment.CONFIRMING_BULLISH
        if not (s > 0.0) and not (p > 0.0):
            return Alignment.CONFIRMING_BEARISH
        return Alignment.DIVERGING

    @staticmethod
    def _finish(ticker, social, market, now, cfg) -> SpeculationReport:  # :45-67
        notes: List[str] = []
        ms = SpeculationEngine.market_summary(market, notes) if market is not None else None
        crowding = SpeculationEngine.crowding(social, ms, cfg)
        align = SpeculationEngine.alignment(social, ms, cfg, notes)
        conf = Confidence.from_sample(social.total_mentions, cfg.confidence_low, cfg.confidence_high)
        return SpeculationReport(ticker=ticker, generated_at=now, social=social, market=ms,
                                 fusion=FusionSignals(alignment=align, crowding=crowding, notes=notes),
                                 social_confidence=conf)
