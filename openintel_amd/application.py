"""The callers either side of the PostAnalyzer path (SURVEY.md section 8, row f).

Host mirror of the reference's use case and of the report wire format, so the GPU
analyzer can be exercised exactly as the reference exercises its own (paths relative to
the openintel repo):

    AnalysisRequest                      src/application/request.rs
    application::analyze                 src/application/analyze.rs:16-73
    DISCLAIMER                           src/application/mod.rs
    AppConfig / OutputFormat             src/config/settings.rs:4-52
    cli::run::analyze / render_json / render_table
                                         src/cli/run.rs:8-131
    SocialDataSource / MarketDataSource  src/domain/ports/{social,market}_data_source.rs
    serde names of the report            src/domain/entities/speculation_report.rs:11-48,
                                         values/{speculation,source_kind,polarity}.rs

The one deliberate difference: the reference constructs LexiconAnalyzer inside
`analyze` (analyze.rs:61); here the PostAnalyzer is a required argument, because the
implementation is a device object bound to a HIP context (INTEGRATION.md shows the same
seam on the Rust side).  There is no default and no CPU analyzer in this package.

Network adapters (Reddit/Bluesky/Yahoo clients) are out of scope; sources are ports.
"""
from __future__ import annotations

import abc
import datetime as _dt
import decimal
import enum
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

from .analyzer import PostAnalyzer
from .domain import (DomainError, EngineConfig, MarketSnapshot, NoData, SocialPost, SourceKind,
                     SpeculationReport, Ticker)
from .engine import SpeculationEngine

DISCLAIMER = ("Not financial advice. OpenIntel is a research/screening tool; markets are risky and "
              "social data is easily manipulated. Do your own diligence.")


# ----------------------------------------------------------------------------- ports
class SocialDataSource(abc.ABC):
    @abc.abstractmethod
    def kind(self) -> SourceKind:
        ...

    @abc.abstractmethod
    def fetch(self, ticker: Ticker, limit: int) -> List[SocialPost]:
        """Raises DomainError on failure (the reference's Err arm)."""


class MarketDataSource(abc.ABC):
    @abc.abstractmethod
    def name(self) -> str:
        ...

    @abc.abstractmethod
    def snapshot(self, ticker: Ticker) -> MarketSnapshot:
        """Raises DomainError on failure."""


# ----------------------------------------------------------------------------- request / config
@dataclass
class AnalysisRequest:  # request.rs
    ticker: str
    enabled_sources: List[SourceKind]
    market_enabled: bool
    limit: int
    engine: EngineConfig = field(default_factory=EngineConfig)


class OutputFormat(enum.Enum):  # settings.rs:4-8
    TABLE = "table"
    JSON = "json"


@dataclass
class AppConfig:  # settings.rs:10-52
    ticker: str
    enabled_sources: List[SourceKind]
    market_enabled: bool
    limit: int
    format: OutputFormat
    engine: EngineConfig = field(default_factory=EngineConfig)

    @staticmethod
    def new(ticker: str, reddit: bool, bluesky: bool, no_market: bool, limit: int,
            format: OutputFormat) -> "AppConfig":
        enabled: List[SourceKind] = []
        if reddit:
            enabled.append(SourceKind.REDDIT)
        if bluesky:
            enabled.append(SourceKind.BLUESKY)
        if not enabled:  # no flags -> every source
            enabled = list(SourceKind.ALL)
        return AppConfig(ticker=ticker, enabled_sources=enabled, market_enabled=not no_market, limit=limit,
                         format=format, engine=EngineConfig())


# ----------------------------------------------------------------------------- use case
def gather(req: AnalysisRequest, social_sources: Sequence[SocialDataSource],
           market_source: Optional[MarketDataSource]):
    """Everything of application::analyze before the analyzer runs (analyze.rs:21-60): the parsed ticker, the posts of
    the enabled sources in list order, the market snapshot, the request notes.  Raises what the reference returns as Err
    (InvalidTicker, NoData).  Shared by `analyze` and the batch callers (batch.py), which pool the posts of many tickers
    into one analyzer call."""
    ticker = Ticker.parse(req.ticker)  # :21

    notes: List[str] = []
    for kind in req.enabled_sources:  # :24-28
        if not any(s.kind() == kind for s in social_sources):
            notes.append("%s enabled but not configured" % kind.as_str())

    posts: List[SocialPost] = []
    for source in social_sources:  # :30-45
        kind = source.kind()
        if kind not in req.enabled_sources:
            continue
        try:
            posts.extend(source.fetch(ticker, req.limit))
        except DomainError as e:
            notes.append("source %s failed: %s" % (kind.as_str(), e))

    market: Optional[MarketSnapshot] = None  # :47-56
    if req.market_enabled and market_source is not None:
        try:
            market = market_source.snapshot(ticker)
        except DomainError as e:
            notes.append("market source failed: %s" % e)

    if not posts and market is None:  # :58-60
        raise NoData()
    return ticker, posts, market, notes


def analyze(req: AnalysisRequest, social_sources: Sequence[SocialDataSource],
            market_source: Optional[MarketDataSource], analyzer: PostAnalyzer,
            now: Optional[_dt.datetime] = None) -> SpeculationReport:
    """application::analyze (analyze.rs:16-73), same note order and error behaviour.

    Sources are polled in list order; the reference joins them concurrently but consumes
    the results in the same list order (join_all keeps positions), so posts and notes come
    out identically."""
    ticker, posts, market, notes = gather(req, social_sources, market_source)
    signals = analyzer.analyze(posts)  # :61-62 -- the hot path
    if now is None:
        now = _dt.datetime.now(_dt.timezone.utc)
    report = SpeculationEngine.aggregate(ticker, posts, signals, market, now, req.engine)
    report.fusion.notes = notes + report.fusion.notes  # :68-69 request notes first
    return report


def run_analyze(config: AppConfig, social_sources: Sequence[SocialDataSource],
                market_source: Optional[MarketDataSource], analyzer: PostAnalyzer,
                now: Optional[_dt.datetime] = None) -> Tuple[SpeculationReport, str]:
    """cli::run::analyze (run.rs:8-23): report + its rendering in the configured format."""
    req = AnalysisRequest(ticker=config.ticker, enabled_sources=list(config.enabled_sources),
                          market_enabled=config.market_enabled, limit=config.limit, engine=config.engine)
    report = analyze(req, social_sources, market_source, analyzer, now=now)
    return report, render(report, config.format)


def render(report: SpeculationReport, format: OutputFormat) -> str:
    return render_json(report) if format is OutputFormat.JSON else render_table(report)


# ----------------------------------------------------------------------------- wire format
def format_f64(v: float) -> str:
    """A finite f64 as serde_json writes it (ryu shortest round-trip digits; plain decimal
    while the decimal point lies within [-5, 16], exponent form `1.2e-7` otherwise)."""
    if v == 0.0:
        return "-0.0" if math.copysign(1.0, v) < 0 else "0.0"
    sign, digits, exp = decimal.Decimal(repr(float(v))).as_tuple()
    digits = list(digits)
    while len(digits) > 1 and digits[-1] == 0:
        digits.pop()
        exp += 1
    ds = "".join(map(str, digits))
    n = len(ds)
    kk = n + exp  # position of the decimal point relative to the first digit
    if 0 <= exp and kk <= 16:
        body = ds + "0" * exp + ".0"
    elif 0 < kk <= 16:
        body = ds[:kk] + "." + ds[kk:]
    elif -5 < kk <= 0:
        body = "0." + "0" * (-kk) + ds
    elif n == 1:
        body = "%se%d" % (ds, kk - 1)
    else:
        body = "%s.%se%d" % (ds[0], ds[1:], kk - 1)
    return ("-" if sign else "") + body


def _json_f64(v: Optional[float]) -> str:
    if v is None or math.isnan(v) or math.isinf(v):  # serde_json writes non-finite as null
        return "null"
    return format_f64(v)


_ESC = {'"': '\\"', "\\": "\\\\", "\b": "\\b", "\f": "\\f", "\n": "\\n", "\r": "\\r", "\t": "\\t"}


def _json_str(s: str) -> str:
    out = []
    for ch in s:
        if ch in _ESC:
            out.append(_ESC[ch])
        elif ord(ch) < 0x20:
            out.append("\\u%04x" % ord(ch))
        else:
            out.append(ch)  # serde_json leaves non-ASCII as UTF-8
    return '"' + "".join(out) + '"'


class _Raw(str):
    """An already-serialised JSON scalar."""


def _pretty(v, indent: int = 0) -> str:
    # serde_json::to_string_pretty layout: two-space indent, ": " separator, [] and {} when empty.
    pad = "  " * (indent + 1)
    end = "  " * indent
    if isinstance(v, _Raw):
        return str(v)
    if isinstance(v, str):
        return _json_str(v)
    if isinstance(v, dict):
        if not v:
            return "{}"
        rows = ["%s%s: %s" % (pad, _json_str(k), _pretty(x, indent + 1)) for k, x in v.items()]
        return "{\n" + ",\n".join(rows) + "\n" + end + "}"
    if isinstance(v, list):
        if not v:
            return "[]"
        rows = [pad + _pretty(x, indent + 1) for x in v]
        return "[\n" + ",\n".join(rows) + "\n" + end + "]"
    raise TypeError(type(v))


def _rfc3339(t, zulu: bool) -> str:
    """chrono DateTime<Utc>: serde writes `...Z`, to_rfc3339() writes `...+00:00`; both use
    the shortest of 0/3/6 fractional digits that holds the value."""
    if not isinstance(t, _dt.datetime):
        return str(t)
    if t.tzinfo is not None:
        t = t.astimezone(_dt.timezone.utc)
    s = t.strftime("%Y-%m-%dT%H:%M:%S")
    if t.microsecond:
        s += (".%03d" % (t.microsecond // 1000)) if t.microsecond % 1000 == 0 else (".%06d" % t.microsecond)
    return s + ("Z" if zulu else "+00:00")


def report_to_wire(report: SpeculationReport) -> dict:
    """Field names and order of #[derive(Serialize)] on SpeculationReport and its parts:
    lowercase source keys (BTreeMap order), transparent Polarity / SpeculationIndex / Ticker,
    snake_case Alignment, lowercase Confidence, Option -> null."""
    s = report.social
    social = {
        "total_mentions": _Raw(int(s.total_mentions)),
        "mentions_by_source": {k.as_str(): _Raw(int(s.mentions_by_source[k]))
                               for k in sorted(s.mentions_by_source)},
        "net_sentiment": _Raw(_json_f64(s.net_sentiment)),
        "bullish": _Raw(int(s.bullish)),
        "bearish": _Raw(int(s.bearish)),
        "neutral": _Raw(int(s.neutral)),
        "bull_bear_ratio": _Raw(_json_f64(s.bull_bear_ratio)),
        "speculation_index": _Raw(_json_f64(s.speculation_index)),
    }
    m = report.market
    market = _Raw("null") if m is None else {
        "last_price": _Raw(_json_f64(m.last_price)),
        "pct_change": _Raw(_json_f64(m.pct_change)),
        "rvol": _Raw(_json_f64(m.rvol)),
        "realized_vol": _Raw(_json_f64(m.realized_vol)),
        "put_call_ratio": _Raw(_json_f64(m.put_call_ratio)),
        "iv_rank": _Raw(_json_f64(m.iv_rank)),
    }
    fusion = {
        "alignment": report.fusion.alignment.value,
        "crowding": _Raw(_json_f64(report.fusion.crowding)),
        "notes": list(report.fusion.notes),
    }
    return {
        "ticker": report.ticker.as_str(),
        "generated_at": _rfc3339(report.generated_at, zulu=True),
        "social": social,
        "market": market,
        "fusion": fusion,
        "social_confidence": report.social_confidence.value,
    }


def report_to_json(report: SpeculationReport) -> str:
    """serde_json::to_string_pretty(&report) (speculation_report.rs:56-83 tests this form)."""
    return _pretty(report_to_wire(report))


def render_json(report: SpeculationReport) -> str:
    """run.rs:33-47: the report's fields flattened, then `disclaimer`."""
    wire = report_to_wire(report)
    wire["disclaimer"] = DISCLAIMER
    return _pretty(wire)


def _debug_name(e: enum.Enum) -> str:
    # Rust {:?} of a unit variant: CamelCase
    return "".join(p.capitalize() for p in e.name.split("_"))


def render_table(report: SpeculationReport) -> str:
    """run.rs:49-131, line for line the same text."""
    s = report.social
    out = []
    out.append("=== OpenIntel — %s ===" % report.ticker.as_str())
    out.append("generated: %s" % _rfc3339(report.generated_at, zulu=False))
    out.append("confidence (social sample): %s" % _debug_name(report.social_confidence))
    out.append("\nSOCIAL")
    out.append("  mentions: %d (bull %d / bear %d / neutral %d)" % (s.total_mentions, s.bullish, s.bearish, s.neutral))
    out.append("  net sentiment: %+.2f" % s.net_sentiment)
    out.append("  speculation index: %.0f%%" % (s.speculation_index * 100.0))
    if s.bull_bear_ratio is not None:
        out.append("  bull/bear ratio: %.2f" % s.bull_bear_ratio)
    else:
        out.append("  bull/bear ratio: n/a (no bearish posts)")
    m = report.market
    if m is not None:
        rvol = ("%.2fx" % m.rvol) if m.rvol is not None else "n/a"
        out.append("\nMARKET")
        out.append("  last: %.2f  change: %+.2f%%  rvol: %s" % (m.last_price, m.pct_change, rvol))
    else:
        failed = any("market source failed" in n for n in report.fusion.notes)
        label = "(unavailable — fetch failed; see notes)" if failed else "(disabled)"
        out.append("\nMARKET\n  %s" % label)
    out.append("\nFUSION")
    out.append("  alignment: %s" % _debug_name(report.fusion.alignment))
    out.append("  crowding: %.0f%%" % (report.fusion.crowding * 100.0))
    for note in report.fusion.notes:
        out.append("  note: %s" % note)
    out.append("\n%s" % DISCLAIMER)
    return "\n".join(out) + "\n"
