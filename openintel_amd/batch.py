"""The batch callers of the PostAnalyzer path ("the only batch entry").

Host mirror of the reference's batch tools (paths relative to the openintel repo):

    run_list_sources                               src/mcp/tools.rs:23-37
    request_from                                   src/mcp/tools.rs:71-96
    summarize                                      src/mcp/tools.rs:99-108
    AnalyzeOutput / run_analyze (no dip deps)      src/mcp/tools.rs:53-66, 110-162
    ScanArgs / ScanEntry / ScanOutput / run_scan   src/mcp/tools.rs:163-225
    RankBy / CompareArgs / RankedEntry / CompareError / CompareOutput
    rank_metric / sort_ranked / run_compare        src/mcp/tools.rs:227-352
    SentimentSummary / sentiment_for (per dip row) src/domain/dip.rs:428-431, src/application/dip.rs:175-194

The reference runs `application::analyze` once per ticker (`join_all`, tools.rs:206-220) and never pools posts of
different tickers: each ticker's few dozen posts go through their own `LexiconAnalyzer::analyze`.  On the GPU the batch
IS the unit of work, so here the posts of ALL tickers of a call go through ONE analyzer call (one scan launch), and --
when the analyzer offers it (`HipLexiconAnalyzer.analyze_segments`) -- every ticker's `social_summary` sums come from one
segmented reduction on the device (`oi_social_summary_segmented`: the reference's input-order f64 sum per ticker, bit for
bit).  The reports are the ones `analyze` would have produced ticker by ticker: same posts, same notes, same numbers
(tests/test_batch.py compares them byte for byte); entries keep the input order (join_all keeps positions).

One difference follows from pooling and is deliberate: an analyzer FAILURE (a device error) fails every ticker that had
posts in the pooled call, each with the same message, where the reference would fail them one by one.  The transport
around these functions (the MCP server, JSON-RPC, tool schemas: src/mcp/server.rs) is control plane and out of scope.
"""
from __future__ import annotations

import datetime as _dt
import enum
import functools
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

from .analyzer import PostAnalyzer, counters_record
from .application import (DISCLAIMER, AnalysisRequest, MarketDataSource, SocialDataSource, _debug_name, _json_f64,
                          _pretty, _Raw, gather, report_to_wire)
from .domain import Alignment, AnalyzerMismatch, DomainError, EngineConfig, SourceKind, SpeculationReport
from .engine import SpeculationEngine


# ----------------------------------------------------------------------------- shared by the tools
def request_from(ticker: str, enable_reddit: Optional[bool] = None, enable_bluesky: Optional[bool] = None,
                 no_market: Optional[bool] = None, limit: Optional[int] = None) -> AnalysisRequest:
    """tools.rs:71-96: no source flag set -> every source; limit 50; market on unless `no_market`."""
    enabled: List[SourceKind] = []
    if enable_reddit:
        enabled.append(SourceKind.REDDIT)
    if enable_bluesky:
        enabled.append(SourceKind.BLUESKY)
    if not enabled:
        enabled = list(SourceKind.ALL)
    return AnalysisRequest(ticker=ticker, enabled_sources=enabled, market_enabled=not bool(no_market),
                           limit=50 if limit is None else limit, engine=EngineConfig())


def summarize(report: SpeculationReport) -> str:
    """tools.rs:99-108: `{} — {:?} · crowding {:.0}% · {} mentions ({:?})`."""
    return "%s — %s · crowding %.0f%% · %d mentions (%s)" % (
        report.ticker.as_str(), _debug_name(report.fusion.alignment), report.fusion.crowding * 100.0,
        report.social.total_mentions, _debug_name(report.social_confidence))


# ----------------------------------------------------------------------------- list_sources / analyze_ticker
def run_list_sources(social_sources: Sequence[SocialDataSource], market_source: MarketDataSource) -> dict:
    """tools.rs:23-37: the data sources actually wired (`social` reflects the injected list, not SourceKind::ALL)."""
    return {"social": [s.kind().as_str() for s in social_sources], "market": [market_source.name()]}


@dataclass
class AnalyzeOutput:  # tools.rs:53-66; dip_signal / dip_note need the dip dependencies (bars, filings, news: out of scope)
    summary: str
    report: SpeculationReport
    dip_signal: Optional[object] = None
    dip_note: Optional[str] = None
    disclaimer: str = DISCLAIMER


def run_analyze(ticker: str, social_sources: Sequence[SocialDataSource], market_source: MarketDataSource,
                analyzer: PostAnalyzer, enable_reddit: Optional[bool] = None, enable_bluesky: Optional[bool] = None,
                no_market: Optional[bool] = None, limit: Optional[int] = None,
                now: Optional[_dt.datetime] = None) -> AnalyzeOutput:
    """tools.rs:110-162 without the dip attachment (`dip_deps = None`: the reference's own default in its tests, :676-696):
    one ticker, the report and its one-line gloss.  Raises the DomainError the reference returns as Err."""
    from .application import analyze
    req = request_from(ticker, enable_reddit, enable_bluesky, no_market, limit)
    report = analyze(req, social_sources, market_source, analyzer, now=now)
    return AnalyzeOutput(summary=summarize(report), report=report)


# ----------------------------------------------------------------------------- the pooled analysis
def analyze_many(requests: Sequence[AnalysisRequest], social_sources: Sequence[SocialDataSource],
                 market_source: Optional[MarketDataSource], analyzer: PostAnalyzer,
                 now: Optional[_dt.datetime] = None) -> List[object]:
    """`application::analyze` for every request, with ONE analyzer call for the posts of all of them.

    Returns, per request and in order, the SpeculationReport or the DomainError the reference's analyze would have
    returned for it."""
    if now is None:
        now = _dt.datetime.now(_dt.timezone.utc)
    results: List[object] = [None] * len(requests)
    pending = []  # (position, request, ticker, posts, market, notes)
    for i, req in enumerate(requests):
        try:
            ticker, posts, market, notes = gather(req, social_sources, market_source)  # analyze.rs:21-60
        except DomainError as e:
            results[i] = e
            continue
        pending.append((i, req, ticker, posts, market, notes))
    if not pending:
        return results

    segments = [p[3] for p in pending]
    counters = None
    try:
        if hasattr(analyzer, "analyze_segments"):
            # one scan + one reduction per ticker on the device; every request_from config has the same threshold,
            # a caller mixing thresholds gets the host sums below
            taus = {p[1].engine.bull_bear_threshold for p in pending}
            if len(taus) == 1:
                per_segment, counters = analyzer.analyze_segments(segments, tau=taus.pop())
        if counters is None:
            flat = [post for seg in segments for post in seg]
            signals = analyzer.analyze(flat)  # analyze.rs:61-62 -- the hot path, once for the whole batch
            if len(signals) != len(flat):  # speculation_engine.rs:29-34, for the pooled call
                raise AnalyzerMismatch(expected=len(flat), got=len(signals))
            per_segment, at = [], 0
            for seg in segments:
                per_segment.append(signals[at:at + len(seg)])
                at += len(seg)
    except DomainError as e:  # the pooled call failed: every ticker in it fails the same way
        for p in pending:
            results[p[0]] = e
        return results

    for k, (i, req, ticker, posts, market, notes) in enumerate(pending):
        try:
            if counters is not None:
                if int(counters[k]["total"]) != len(posts):
                    raise AnalyzerMismatch(expected=len(posts), got=int(counters[k]["total"]))
                report = SpeculationEngine.aggregate_counters(ticker, counters_record(counters[k]), market, now, req.engine)
            else:
                report = SpeculationEngine.aggregate(ticker, posts, per_segment[k], market, now, req.engine)
            report.fusion.notes = notes + report.fusion.notes  # analyze.rs:68-69 request notes first
            results[i] = report
        except DomainError as e:
            results[i] = e
    return results


# ----------------------------------------------------------------------------- the dip screen's sentiment step
SOCIAL_LIMIT = 50  # application/dip.rs:44


@dataclass
class SentimentSummary:  # domain/dip.rs:428-431
    net_sentiment: float
    mentions: int


def sentiments_for(tickers: Sequence[str], social_sources: Sequence[SocialDataSource], analyzer: PostAnalyzer,
                   now: Optional[_dt.datetime] = None) -> List[Optional[SentimentSummary]]:
    """`sentiment_for` (application/dip.rs:175-194) for every row of a dip scan at once: the social-only analysis the
    screen runs per loser (all sources, no market, limit 50), pooled like run_scan.  Any failure -- no sources, an invalid
    ticker, no posts -- degrades to None, as in the reference (the domain then scores divergence 0 with a note)."""
    if not social_sources:
        return [None] * len(tickers)
    reqs = [AnalysisRequest(ticker=t, enabled_sources=list(SourceKind.ALL), market_enabled=False, limit=SOCIAL_LIMIT,
                            engine=EngineConfig()) for t in tickers]
    out: List[Optional[SentimentSummary]] = []
    for res in analyze_many(reqs, social_sources, None, analyzer, now):
        out.append(None if isinstance(res, DomainError) else
                   SentimentSummary(net_sentiment=float(res.social.net_sentiment), mentions=int(res.social.total_mentions)))
    return out


# ----------------------------------------------------------------------------- scan_watchlist
@dataclass
class ScanArgs:  # tools.rs:163-175
    tickers: List[str]
    enable_reddit: Optional[bool] = None
    enable_bluesky: Optional[bool] = None
    no_market: Optional[bool] = None
    limit: Optional[int] = None


@dataclass
class ScanEntry:  # tools.rs:177-184
    ticker: str
    report: Optional[SpeculationReport] = None
    error: Optional[str] = None


@dataclass
class ScanOutput:  # tools.rs:186-190
    entries: List[ScanEntry]
    disclaimer: str = DISCLAIMER


def run_scan(args: ScanArgs, social_sources: Sequence[SocialDataSource], market_source: MarketDataSource,
             analyzer: PostAnalyzer, now: Optional[_dt.datetime] = None) -> ScanOutput:
    """tools.rs:193-225.  One entry per input ticker, in input order: the report, or the error's Display string."""
    reqs = [request_from(t, args.enable_reddit, args.enable_bluesky, args.no_market, args.limit) for t in args.tickers]
    entries = []
    for t, res in zip(args.tickers, analyze_many(reqs, social_sources, market_source, analyzer, now)):
        if isinstance(res, DomainError):
            entries.append(ScanEntry(ticker=t, error=str(res)))
        else:
            entries.append(ScanEntry(ticker=t, report=res))
    return ScanOutput(entries=entries)


# ----------------------------------------------------------------------------- compare_tickers
class RankBy(enum.Enum):  # tools.rs:227-238, serde snake_case
    CROWDING = "crowding"
    SPECULATION_INDEX = "speculation_index"
    NET_SENTIMENT = "net_sentiment"
    DIVERGENCE = "divergence"


@dataclass
class CompareArgs:  # tools.rs:240-251
    tickers: List[str]
    rank_by: RankBy = RankBy.CROWDING
    enable_reddit: Optional[bool] = None
    enable_bluesky: Optional[bool] = None
    no_market: Optional[bool] = None
    limit: Optional[int] = None


@dataclass
class RankedEntry:  # tools.rs:253-258
    ticker: str
    rank_metric: float
    report: SpeculationReport


@dataclass
class CompareError:  # tools.rs:260-264
    ticker: str
    error: str


@dataclass
class CompareOutput:  # tools.rs:266-272
    rank_by: RankBy
    ranked: List[RankedEntry]
    errors: List[CompareError]
    disclaimer: str = DISCLAIMER


def rank_metric(report: SpeculationReport, rank_by: RankBy) -> float:
    """tools.rs:274-283: `divergence` ranks categorically first, its numeric metric is crowding."""
    if rank_by in (RankBy.CROWDING, RankBy.DIVERGENCE):
        return float(report.fusion.crowding)
    if rank_by is RankBy.SPECULATION_INDEX:
        return float(report.social.speculation_index)
    return float(report.social.net_sentiment)


def _desc_partial(a: float, b: float) -> int:
    # b.partial_cmp(&a).unwrap_or(Equal): descending, an unordered pair (NaN) compares equal
    if math.isnan(a) or math.isnan(b):
        return 0
    return -1 if b < a else (1 if b > a else 0)


def sort_ranked(ranked: List[RankedEntry], rank_by: RankBy) -> None:
    """tools.rs:285-301, in place; `sort_by` is stable and so is this."""
    def cmp(a: RankedEntry, b: RankedEntry) -> int:
        if rank_by is RankBy.DIVERGENCE:
            a_div = a.report.fusion.alignment is Alignment.DIVERGING
            b_div = b.report.fusion.alignment is Alignment.DIVERGING
            if a_div != b_div:  # b_div.cmp(&a_div): diverging first
                return -1 if a_div else 1
        return _desc_partial(a.rank_metric, b.rank_metric)
    ranked.sort(key=functools.cmp_to_key(cmp))


def run_compare(args: CompareArgs, social_sources: Sequence[SocialDataSource], market_source: MarketDataSource,
                analyzer: PostAnalyzer, now: Optional[_dt.datetime] = None) -> CompareOutput:
    """tools.rs:303-352: valid tickers ranked, invalid ones listed, both in input order before the (stable) sort."""
    reqs = [request_from(t, args.enable_reddit, args.enable_bluesky, args.no_market, args.limit) for t in args.tickers]
    ranked: List[RankedEntry] = []
    errors: List[CompareError] = []
    for t, res in zip(args.tickers, analyze_many(reqs, social_sources, market_source, analyzer, now)):
        if isinstance(res, DomainError):
            errors.append(CompareError(ticker=t, error=str(res)))
        else:
            ranked.append(RankedEntry(ticker=t, rank_metric=rank_metric(res, args.rank_by), report=res))
    sort_ranked(ranked, args.rank_by)
    return CompareOutput(rank_by=args.rank_by, ranked=ranked, errors=errors)


# ----------------------------------------------------------------------------- wire format (#[derive(Serialize)])
def scan_output_to_json(out: ScanOutput) -> str:
    """serde_json::to_string_pretty(&ScanOutput): `report` / `error` are skipped when None (tools.rs:180-183)."""
    entries = []
    for e in out.entries:
        row = {"ticker": e.ticker}
        if e.report is not None:
            row["report"] = report_to_wire(e.report)
        if e.error is not None:
            row["error"] = e.error
        entries.append(row)
    return _pretty({"entries": entries, "disclaimer": out.disclaimer})


def compare_output_to_json(out: CompareOutput) -> str:
    """serde_json::to_string_pretty(&CompareOutput): field order of the structs, snake_case RankBy."""
    return _pretty({
        "rank_by": out.rank_by.value,
        "ranked": [{"ticker": r.ticker, "rank_metric": _Raw(_json_f64(r.rank_metric)), "report": report_to_wire(r.report)}
                   for r in out.ranked],
        "errors": [{"ticker": e.ticker, "error": e.error} for e in out.errors],
        "disclaimer": out.disclaimer,
    })
