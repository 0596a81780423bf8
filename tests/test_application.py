"""The callers either side of the PostAnalyzer path (SURVEY.md section 8 row f): the
`application::analyze` contract and the report wire format.

These read like the reference's own tests and cite them:
    src/application/analyze.rs:93-152      use-case tests
    src/cli/run.rs:133-190                 run/render tests
    tests/analyze_flow.rs:118-160          end-to-end flow
    src/config/settings.rs:56-76           AppConfig
    src/domain/entities/speculation_report.rs:56-83   serialisation

CPU tests drive the flow with the oracle as the PostAnalyzer (the oracle is the checker and
tests may use it); the `gpu` tests run the identical assertions through HipLexiconAnalyzer
and require byte-identical renderings.
"""
import datetime as dt

import numpy as np
import pytest

from openintel_amd import application as app
from openintel_amd.analyzer import PostAnalyzer, pack_posts
from openintel_amd.domain import (Alignment, Confidence, DomainError, FusionSignals, InvalidTicker, MarketSnapshot,
                                  NoData, PostSignal, PostText, SocialPost, SocialSummary, SourceFailure, SourceKind,
                                  SpeculationReport, Ticker)

NOW = dt.datetime(2026, 6, 24, 0, 0, 0, tzinfo=dt.timezone.utc)


# ----------------------------------------------------------------------------- test doubles
class FixtureSource(app.SocialDataSource):
    """src/adapters/sources/test_fixtures.rs:15-44 -- rows come from tests/golden."""

    def __init__(self, kind, rows):
        self._kind, self.rows = kind, rows

    def kind(self):
        return self._kind

    def fetch(self, ticker, limit):
        sym = ticker.as_str()
        return [SocialPost(id=r["id"], source=self._kind, author=r["author"],
                           text=PostText.parse(r["text"].replace("AAPL", sym)),
                           created_at=dt.datetime(2026, 6, 24, 15, 0, 0, tzinfo=dt.timezone.utc),
                           engagement=r["engagement"]) for r in self.rows[:limit]]


class FailingSource(app.SocialDataSource):
    def __init__(self, kind):
        self._kind = kind

    def kind(self):
        return self._kind

    def fetch(self, ticker, limit):
        raise SourceFailure(self._kind.as_str(), "HTTP 503")


class MockMarketSource(app.MarketDataSource):
    """src/adapters/market/mock_market.rs:11-30 -- numbers from tests/golden."""

    def __init__(self, numbers):
        self.numbers = numbers

    def name(self):
        return "mock-market"

    def snapshot(self, ticker):
        return MarketSnapshot(ticker=ticker, as_of=dt.datetime(2026, 6, 24, 20, 0, 0, tzinfo=dt.timezone.utc),
                              **self.numbers)


class FailingMarket(app.MarketDataSource):
    def name(self):
        return "down-market"

    def snapshot(self, ticker):
        raise SourceFailure("down-market", "timeout")


class OracleAnalyzer(PostAnalyzer):
    """The CPU oracle behind the PostAnalyzer port (test infrastructure only)."""

    def analyze(self, posts):
        from oracle import lib
        if not posts:
            return []
        blob, offs = pack_posts([p.text.as_str() for p in posts])
        pol, spec = lib.lexicon_analyze(blob, offs)
        return [PostSignal(float(p), bool(s)) for p, s in zip(pol, spec)]


class ShortAnalyzer(PostAnalyzer):
    """Breaks the one-signal-per-post contract (post_analyzer.rs:9)."""

    def analyze(self, posts):
        return [PostSignal(0.0, False)] * max(len(posts) - 1, 0)


def _rows(golden, source):
    return [p for p in golden["fixture_posts"] if p["source"] == source]


def reddit_fixture(golden):
    return FixtureSource(SourceKind.REDDIT, _rows(golden, "reddit"))


def bluesky_fixture(golden):
    return FixtureSource(SourceKind.BLUESKY, _rows(golden, "bluesky"))


def fixture_social(golden):
    return [reddit_fixture(golden), bluesky_fixture(golden)]


def req(ticker, market):
    return app.AnalysisRequest(ticker=ticker, enabled_sources=list(SourceKind.ALL), market_enabled=market, limit=50)


EXPECTED_JSON = """{
  "ticker": "AAPL",
  "generated_at": "2026-06-24T00:00:00Z",
  "social": {
    "total_mentions": 10,
    "mentions_by_source": {
      "reddit": 4,
      "bluesky": 6
    },
    "net_sentiment": 0.5,
    "bullish": 7,
    "bearish": 2,
    "neutral": 1,
    "bull_bear_ratio": 3.5,
    "speculation_index": 0.3
  },
  "market": {
    "last_price": 192.5,
    "pct_change": 4.054054054054054,
    "rvol": 1.8269230769230769,
    "realized_vol": 0.38,
    "put_call_ratio": 0.7,
    "iv_rank": 0.82
  },
  "fusion": {
    "alignment": "confirming_bullish",
    "crowding": 0.49669230769230766,
    "notes": []
  },
  "social_confidence": "medium",
  "disclaimer": "Not financial advice. OpenIntel is a research/screening tool; markets are risky and social data is easily manipulated. Do your own diligence."
}"""

EXPECTED_TABLE = """=== OpenIntel — AAPL ===
generated: 2026-06-24T00:00:00+00:00
confidence (social sample): Medium

SOCIAL
  mentions: 10 (bull 7 / bear 2 / neutral 1)
  net sentiment: +0.50
  speculation index: 30%
  bull/bear ratio: 3.50

MARKET
  last: 192.50  change: +4.05%  rvol: 1.83x

FUSION
  alignment: ConfirmingBullish
  crowding: 50%

Not financial advice. OpenIntel is a research/screening tool; markets are risky and social data is easily manipulated. Do your own diligence.
"""


# ----------------------------------------------------------------------------- the shared assertions
def check_use_case(golden, analyzer):
    market = MockMarketSource(golden["mock_market"])
    # analyze.rs:93-106 analyzes_default_request_confirming_bullish
    r = app.analyze(req("AAPL", True), fixture_social(golden), market, analyzer, now=NOW)
    assert r.social.total_mentions == 10
    assert r.fusion.alignment is Alignment.CONFIRMING_BULLISH
    assert r.market is not None
    # analyze.rs:108-117 invalid_ticker_errors
    with pytest.raises(DomainError):
        app.analyze(req("$$$", True), fixture_social(golden), market, analyzer)
    with pytest.raises(InvalidTicker):
        app.analyze(req("$$$", True), fixture_social(golden), market, analyzer)
    # analyze.rs:119-127 social_only_when_no_source_provided
    r = app.analyze(req("AAPL", False), fixture_social(golden), None, analyzer)
    assert r.market is None and r.fusion.alignment is Alignment.QUIET
    assert r.fusion.notes == ["social-only, no price reference"]
    # analyze.rs:129-142 enabled_source_absent_is_noted
    r = app.analyze(req("AAPL", False), [bluesky_fixture(golden)], None, analyzer)
    assert r.social.total_mentions == 6
    assert any("reddit enabled but not configured" in n for n in r.fusion.notes)
    assert r.fusion.notes[0] == "reddit enabled but not configured"  # request notes precede engine notes
    # analyze.rs:144-152 zero_sources_and_no_market_is_no_data
    with pytest.raises(NoData):
        app.analyze(req("AAPL", False), [], None, analyzer)


def check_run(golden, analyzer):
    market = MockMarketSource(golden["mock_market"])
    J, T = app.OutputFormat.JSON, app.OutputFormat.TABLE
    # run.rs:144-157 full_run_confirms_bullish_with_market
    report, rendered = app.run_analyze(app.AppConfig.new("AAPL", False, False, False, 50, J),
                                       fixture_social(golden), market, analyzer, now=NOW)
    assert report.market is not None and report.fusion.alignment is Alignment.CONFIRMING_BULLISH
    assert "Not financial advice" in rendered and "speculation_index" in rendered
    # tests/analyze_flow.rs:118-133 end_to_end_all_sources_with_market
    assert report.social.total_mentions == 10
    assert '"alignment": "confirming_bullish"' in rendered
    assert rendered == EXPECTED_JSON
    # run.rs:159-166 no_market_run_is_quiet / analyze_flow.rs:148-155
    report, _ = app.run_analyze(app.AppConfig.new("AAPL", False, False, True, 50, T), fixture_social(golden), None,
                                analyzer)
    assert report.market is None and report.fusion.alignment is Alignment.QUIET
    # run.rs:168-181 table_output_has_sections_and_disclaimer
    _, rendered = app.run_analyze(app.AppConfig.new("AAPL", False, False, False, 50, T), fixture_social(golden),
                                  market, analyzer, now=NOW)
    for needle in ("SOCIAL", "MARKET", "FUSION", "Not financial advice"):
        assert needle in rendered
    assert rendered == EXPECTED_TABLE
    # run.rs:183-189 invalid_ticker_errors
    with pytest.raises(DomainError):
        app.run_analyze(app.AppConfig.new("$$$", False, False, False, 50, T), fixture_social(golden), market, analyzer)
    # analyze_flow.rs:135-146 single_source_only
    report, _ = app.run_analyze(app.AppConfig.new("AAPL", True, False, False, 50, J), fixture_social(golden), market,
                                analyzer)
    assert report.social.total_mentions == 4
    # reference_assertions block of the golden file (bluesky only)
    report, _ = app.run_analyze(app.AppConfig.new("AAPL", False, True, False, 50, J), fixture_social(golden), market,
                                analyzer)
    assert report.social.total_mentions == 6
    assert list(report.social.mentions_by_source) == [SourceKind.BLUESKY]


def check_failure_notes(golden, analyzer):
    # analyze.rs:40-45, :47-56: failures become notes, in source order, before the engine's notes
    srcs = [FailingSource(SourceKind.REDDIT), bluesky_fixture(golden)]
    r = app.analyze(req("AAPL", True), srcs, FailingMarket(), analyzer, now=NOW)
    assert r.social.total_mentions == 6 and r.market is None
    assert r.fusion.notes == ["source reddit failed: data source 'reddit' failed: HTTP 503",
                              "market source failed: data source 'down-market' failed: timeout",
                              "social-only, no price reference"]
    assert "(unavailable — fetch failed; see notes)" in app.render_table(r)
    assert "  note: market source failed: data source 'down-market' failed: timeout\n" in app.render_table(r)
    # everything failed -> NoData (posts empty and no snapshot), not a report
    with pytest.raises(NoData):
        app.analyze(req("AAPL", True), [FailingSource(SourceKind.REDDIT)], FailingMarket(), analyzer)
    # market alone is enough data: empty social + a snapshot gives a Quiet report (below min_sample)
    r = app.analyze(req("AAPL", True), [], MockMarketSource(golden["mock_market"]), analyzer, now=NOW)
    assert r.social.total_mentions == 0 and r.market is not None and r.fusion.alignment is Alignment.QUIET
    assert r.social_confidence is Confidence.LOW
    assert r.fusion.notes == ["reddit enabled but not configured", "bluesky enabled but not configured"]
    assert "(disabled)" not in app.render_table(r)
    # limit is honoured per source (test_fixtures.rs:29 `.take(limit)`)
    rq = req("AAPL", False)
    rq.limit = 2
    r = app.analyze(rq, fixture_social(golden), None, analyzer)
    assert r.social.total_mentions == 4 and r.social.mentions_by_source == {SourceKind.REDDIT: 2,
                                                                          SourceKind.BLUESKY: 2}
    # a disabled source is neither fetched nor noted
    rq = app.AnalysisRequest("AAPL", [SourceKind.BLUESKY], False, 50)
    r = app.analyze(rq, [FailingSource(SourceKind.REDDIT), bluesky_fixture(golden)], None, analyzer)
    assert r.fusion.notes == ["social-only, no price reference"]


# ----------------------------------------------------------------------------- CPU: host logic + wire format
def test_use_case_contract_cpu(golden):
    check_use_case(golden, OracleAnalyzer())


def test_run_and_renderings_cpu(golden):
    check_run(golden, OracleAnalyzer())


def test_failure_notes_cpu(golden):
    check_failure_notes(golden, OracleAnalyzer())


def test_analyzer_mismatch_propagates(golden):
    from openintel_amd.domain import AnalyzerMismatch
    with pytest.raises(AnalyzerMismatch):
        app.analyze(req("AAPL", False), fixture_social(golden), None, ShortAnalyzer())


def test_app_config():
    # settings.rs:56-76
    c = app.AppConfig.new("AAPL", False, False, False, 50, app.OutputFormat.TABLE)
    assert c.enabled_sources == [SourceKind.REDDIT, SourceKind.BLUESKY] and c.market_enabled
    c = app.AppConfig.new("AAPL", True, False, True, 50, app.OutputFormat.JSON)
    assert c.enabled_sources == [SourceKind.REDDIT] and not c.market_enabled


def test_report_serialisation_names():
    # speculation_report.rs:56-83 serializes_with_snake_case_alignment_and_transparent_newtypes
    import json
    r = SpeculationReport(
        ticker=Ticker.parse("AAPL"), generated_at=NOW,
        social=SocialSummary(total_mentions=2, mentions_by_source={SourceKind.REDDIT: 2}, net_sentiment=0.5,
                             bullish=1, bearish=0, neutral=1, bull_bear_ratio=None, speculation_index=0.5),
        market=None, fusion=FusionSignals(alignment=Alignment.QUIET, crowding=0.1, notes=[]),
        social_confidence=Confidence.LOW)
    s = app.report_to_json(r)
    compact = json.dumps(json.loads(s), separators=(",", ":"))
    assert '"reddit":2' in compact
    assert '"speculation_index":0.5' in compact
    assert '"alignment":"quiet"' in compact
    assert '"ticker":"AAPL"' in compact and '"market":null' in compact and '"bull_bear_ratio":null' in compact
    assert '"social_confidence":"low"' in compact and '"net_sentiment":0.5' in compact
    assert '"notes": []' in s and '"generated_at": "2026-06-24T00:00:00Z"' in s
    assert "disclaimer" not in s and "disclaimer" in app.render_json(r)
    # no-bearish / disabled-market branches of the table (run.rs:79-81, :100-112)
    t = app.render_table(r)
    assert "  bull/bear ratio: n/a (no bearish posts)\n" in t and "\nMARKET\n  (disabled)\n" in t
    assert "confidence (social sample): Low\n" in t and "  alignment: Quiet\n" in t


def test_f64_formatting_like_serde_json():
    # ryu's shortest round-trip text, serde_json's layout rules
    cases = {0.0: "0.0", 1.0: "1.0", -1.0: "-1.0", 0.5: "0.5", 192.5: "192.5", 1e15: "1000000000000000.0",
             1e16: "1e16", 1.5e16: "1.5e16", 123456789012345680.0: "1.2345678901234568e17", 1e-5: "0.00001",
             1.5e-5: "0.000015", 1e-6: "1e-6", 1.25e-7: "1.25e-7", 0.1 + 0.2: "0.30000000000000004",
             4.054054054054054: "4.054054054054054", 5e-324: "5e-324", 1.7976931348623157e308: "1.7976931348623157e308",
             100.0: "100.0", 95_000_000 / 52_000_000: "1.8269230769230769", 1e21: "1e21", 12345.678: "12345.678"}
    for v, want in cases.items():
        assert app.format_f64(v) == want, (v, app.format_f64(v), want)
        assert float(app.format_f64(v)) == v
    assert app.format_f64(-0.0) == "-0.0"
    rng = np.random.default_rng(5)
    for v in np.concatenate([rng.standard_normal(200), 10.0 ** rng.uniform(-30, 30, 200)]):
        assert float(app.format_f64(float(v))) == float(v)
    assert app._json_f64(float("nan")) == "null" and app._json_f64(float("inf")) == "null"


def test_json_string_escapes_and_timestamps():
    assert app._json_str('a"b\\c\n\t\x01é🚀') == '"a\\"b\\\\c\\n\\t\\u0001é🚀"'
    t = dt.datetime(2026, 6, 24, 1, 2, 3, 250000, tzinfo=dt.timezone.utc)
    assert app._rfc3339(t, True) == "2026-06-24T01:02:03.250Z"
    assert app._rfc3339(t.replace(microsecond=250001), False) == "2026-06-24T01:02:03.250001+00:00"
    assert app._rfc3339(t.replace(microsecond=0), True) == "2026-06-24T01:02:03Z"


# ----------------------------------------------------------------------------- GPU: same flow through the HIP analyzer
@pytest.fixture(scope="module")
def hip_analyzer():
    import openintel_amd as oi
    c = oi.HipContext(0)
    yield oi.HipLexiconAnalyzer(c)
    c.close()


@pytest.mark.gpu
def test_use_case_contract_gpu(golden, hip_analyzer):
    check_use_case(golden, hip_analyzer)


@pytest.mark.gpu
def test_run_and_renderings_gpu(golden, hip_analyzer):
    check_run(golden, hip_analyzer)


@pytest.mark.gpu
def test_failure_notes_gpu(golden, hip_analyzer):
    check_failure_notes(golden, hip_analyzer)


@pytest.mark.gpu
def test_large_feed_renders_identically_gpu(hip_analyzer):
    """200K synthetic posts through both analyzers: the rendered reports must be the same bytes
    (per-post polarity is bit-exact, the engine sums in input order on the host)."""
    from openintel_amd import synth
    texts = synth.posts_np(200_000, seed=11)
    half = len(texts) // 2

    class Feed(app.SocialDataSource):
        def __init__(self, kind, rows):
            self._k, self.rows = kind, rows

        def kind(self):
            return self._k

        def fetch(self, ticker, limit):
            return [SocialPost(id=str(i), source=self._k, author="a", text=PostText(t))
                    for i, t in enumerate(self.rows[:limit])]

    srcs = [Feed(SourceKind.REDDIT, texts[:half]), Feed(SourceKind.BLUESKY, texts[half:])]
    rq = app.AnalysisRequest("GME", list(SourceKind.ALL), False, 10 ** 9)
    a = app.analyze(rq, srcs, None, hip_analyzer, now=NOW)
    b = app.analyze(rq, srcs, None, OracleAnalyzer(), now=NOW)
    assert a.social.total_mentions == len(texts)
    assert app.render_json(a) == app.render_json(b)
    assert app.render_table(a) == app.render_table(b)
