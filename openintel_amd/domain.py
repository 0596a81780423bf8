"""Host-side mirror of the reference's domain types on the PostAnalyzer path.

Same names, argument meaning and error behaviour as the reference (paths relative to
the openintel repo), so the parity tests read like the reference's own tests:

    SourceKind      src/domain/values/source_kind.rs:5-21
    Polarity        src/domain/values/polarity.rs:8-14
    PostSignal      src/domain/values/post_signal.rs:4-7
    SpeculationIndex / Confidence / Alignment   src/domain/values/speculation.rs:7-51
    PostText / SocialPost                       src/domain/entities/social_post.rs:7-38
    Ticker                                      src/domain/entities/ticker.rs:10-36
    MarketSnapshot                              src/domain/entities/market_snapshot.rs:7-17
    SocialSummary / MarketSummary / FusionSignals / SpeculationReport
                                                src/domain/entities/speculation_report.rs:11-48
    EngineConfig                                src/domain/engine/config.rs:2-33
    DomainError                                 src/domain/error.rs:4-22
"""
from __future__ import annotations

import enum
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

MAX_POST_LEN = 10_000  # social_post.rs:7


# ----------------------------------------------------------------- errors (error.rs:4-22)
class DomainError(Exception):
    pass


class InvalidTicker(DomainError):
    def __init__(self, raw: str):
        super().__init__("invalid ticker: %s" % raw)


class InvalidPostText(DomainError):
    def __init__(self, why: str):
        super().__init__("invalid post text: %s" % why)


class AnalyzerMismatch(DomainError):
    def __init__(self, expected: int, got: int):
        super().__init__("analyzer returned %d signals for %d posts" % (got, expected))
        self.expected, self.got = expected, got


class MarketTickerMismatch(DomainError):
    def __init__(self, expected: str, got: str):
        super().__init__("market snapshot ticker '%s' does not match requested '%s'" % (got, expected))
        self.expected, self.got = expected, got


class SourceFailure(DomainError):
    def __init__(self, name: str, message: str):
        super().__init__("data source '%s' failed: %s" % (name, message))
        self.name, self.message = name, message


class NoData(DomainError):
    def __init__(self):
        super().__init__("no data: no posts and no market snapshot available")


# ----------------------------------------------------------------- values
class SourceKind(enum.IntEnum):
    REDDIT = 0
    BLUESKY = 1

    def as_str(self) -> str:
        return "reddit" if self is SourceKind.REDDIT else "bluesky"


SourceKind.ALL = [SourceKind.REDDIT, SourceKind.BLUESKY]


def _clamp(v: float, lo: float, hi: float) -> float:
    # Rust f64::clamp
    if v < lo:
        return lo
    if v > hi:
        return hi
    return v


def polarity_new(v: float) -> float:
    """Polarity::new -- NaN -> 0, else clamp to [-1, 1] (polarity.rs:8-14)."""
    if math.isnan(v):
        return 0.0
    return _clamp(v, -1.0, 1.0)


def speculation_index_new(v: float) -> float:
    """SpeculationIndex::new (speculation.rs:8-14)."""
    if math.isnan(v):
        return 0.0
    return _clamp(v, 0.0, 1.0)


@dataclass(frozen=True)
class PostSignal:
    polarity: float
    speculative: bool

    def __post_init__(self):
        object.__setattr__(self, "polarity", polarity_new(self.polarity))


class Confidence(enum.Enum):
    LOW = "low"
    MEDIUM = "medium"
    HIGH = "high"

    @staticmethod
    def from_sample(n: int, low: int, high: int) -> "Confidence":
        low, high = min(low, high), max(low, high)  # speculation.rs:33
        if n < low:
            return Confidence.LOW
        if n < high:
            return Confidence.MEDIUM
        return Confidence.HIGH


class Alignment(enum.Enum):
    CONFIRMING_BULLISH = "confirming_bullish"
    CONFIRMING_BEARISH = "confirming_bearish"
    DIVERGING = "diverging"
    QUIET = "quiet"


# ----------------------------------------------------------------- entities
class PostText:
    __slots__ = ("_s",)

    def __init__(self, s: str):
        self._s = s

    # char::is_whitespace (Unicode White_Space), which is what str::trim strips -- not Python's
    # str.strip(), which also strips U+001C..U+001F
    _WS = frozenset("\t\n\x0b\x0c\r \x85\xa0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008"
                    "\u2009\u200a\u2028\u2029\u202f\u205f\u3000")

    @staticmethod
    def rust_trim(s: str) -> str:
        a, b = 0, len(s)
        while a < b and s[a] in PostText._WS:
            a += 1
        while b > a and s[b - 1] in PostText._WS:
            b -= 1
        return s[a:b]

    @staticmethod
    def parse(raw: str) -> "PostText":  # social_post.rs:14-23
        trimmed = PostText.rust_trim(raw)
        if not trimmed:
            raise InvalidPostText("empty")
        if len(trimmed) > MAX_POST_LEN:  # chars, not bytes
            raise InvalidPostText("exceeds max length")
        return PostText(trimmed)

    def as_str(self) -> str:
        return self._s


@dataclass
class SocialPost:
    id: str
    source: SourceKind
    author: str
    text: PostText
    created_at: object = None
    engagement: int = 0


class Ticker:
    __slots__ = ("_s",)

    def __init__(self, s: str):
        self._s = s

    @staticmethod
    def parse(raw: str) -> "Ticker":  # ticker.rs:10-36
        trimmed = raw.strip()
        if not trimmed:
            raise InvalidTicker("empty")
        if not trimmed.isascii():
            raise InvalidTicker(raw)
        symbol = trimmed.upper()
        base, _, cls = symbol.partition(".")
        has_class = "." in symbol
        base_ok = 1 <= len(base) <= 5 and all("A" <= c <= "Z" for c in base)
        class_ok = (not has_class) or (len(cls) == 1 and "A" <= cls <= "Z")
        if base_ok and class_ok:
            return Ticker(symbol)
        raise InvalidTicker(raw)

    def as_str(self) -> str:
        return self._s


@dataclass
class MarketSnapshot:
    ticker: Ticker
    last_price: float
    previous_close: float
    volume: int
    avg_volume: int
    realized_vol: Optional[float] = None
    put_call_ratio: Optional[float] = None
    iv_rank: Optional[float] = None
    as_of: object = None


@dataclass
class EngineConfig:  # config.rs:18-33
    bull_bear_threshold: float = 0.2
    net_sentiment_threshold: float = 0.05
    price_move_threshold: float = 1.0
    crowding_weight_spec: float = 0.5
    crowding_weight_rvol: float = 0.3
    crowding_weight_iv: float = 0.2
    rvol_cap: float = 3.0
    min_sample: int = 10
    confidence_low: int = 10
    confidence_high: int = 50


@dataclass
class SocialSummary:
    total_mentions: int
    mentions_by_source: Dict[SourceKind, int]
    net_sentiment: float
    bullish: int
    bearish: int
    neutral: int
    bull_bear_ratio: Optional[float]
    speculation_index: float


@dataclass
class MarketSummary:
    last_price: float
    pct_change: float
    rvol: Optional[float]
    realized_vol: Optional[float]
    put_call_ratio: Optional[float]
    iv_rank: Optional[float]


@dataclass
class FusionSignals:
    alignment: Alignment
    crowding: float
    notes: List[str] = field(default_factory=list)


@dataclass
class SpeculationReport:
    ticker: Ticker
    generated_at: object
    social: SocialSummary
    market: Optional[MarketSummary]
    fusion: FusionSignals
    social_confidence: Confidence
