"""The batch callers of the PostAnalyzer path: `run_scan`, `run_compare` and the
per-ticker `social_summary` over a pooled batch.

These read like the reference's own tests and cite them:
    src/mcp/tools.rs:715-745     run_scan_handles_mixed_batch / run_scan_empty_list_is_empty
    src/mcp/tools.rs:747-812     sort_ranked_orders_by_crowding_desc
    src/mcp/tools.rs:814-832     run_compare_partitions_valid_and_invalid
    src/mcp/tools.rs:71-108      request_from / summarize (exercised by run_analyze_returns_confirming_bullish_report)

CPU tests drive the tools with the oracle behind the PostAnalyzer port and pin the pooled form against the
ticker-by-ticker one; the `gpu` tests run the same assertions through HipLexiconAnalyzer (one scan + one segmented
reduction on the device) and compare `oi_social_summary_segmented` with the oracle bit for bit.
"""
import math
import zlib

import numpy as np
import pytest

from openintel_amd import application as app
from openintel_amd import batch
from openintel_amd.analyzer import PostAnalyzer
from openintel_amd.domain import (Alignment, DomainError, EngineConfig, PostSignal, PostText, SocialPost, SourceFailure,
                                  SourceKind, Ticker)
from openintel_amd.engine import SpeculationEngine
from test_application import NOW, MockMarketSource, OracleAnalyzer, ShortAnalyzer, fixture_social


# ----------------------------------------------------------------------------- test doubles
class CountingAnalyzer(OracleAnalyzer):
    """Counts port calls: the batch tools must make ONE for the whole batch."""

    def __init__(self):
        self.calls = []

    def analyze(self, posts):
        self.calls.append(len(posts))
        return super().analyze(posts)


class OracleSegmentAnalyzer(OracleAnalyzer):
    """The oracle behind the optional `analyze_segments` form (signals + per-ticker sums), so the counters path of
    batch.analyze_many is exercised on the CPU too."""

    def analyze_segments(self, segments, tau=0.2):
        from oracle import lib
        from openintel_amd.analyzer import COUNTERS_DTYPE
        flat = [p for seg in segments for p in seg]
        sig = self.analyze(flat)
        seg_off = np.concatenate([[0], np.cumsum([len(s) for s in segments])]).astype(np.uint64)
        cfg = lib.default_config()
        cfg.bull_bear_threshold = tau
        sums = lib.social_summary_segmented(np.array([int(p.source) for p in flat], dtype=np.uint8),
                                            np.array([s.polarity for s in sig], dtype=np.float64),
                                            np.array([s.speculative for s in sig], dtype=np.uint8), seg_off, cfg)
        out = np.zeros(len(segments), dtype=COUNTERS_DTYPE)
        for k, o in enumerate(sums):
            out[k] = (o.total_mentions, (o.mentions_by_source[0], o.mentions_by_source[1]), o.bullish, o.bearish, o.neutral,
                      o.spec_count, o.polarity_sum)
        per = [sig[int(seg_off[k]):int(seg_off[k + 1])] for k in range(len(segments))]
        return per, out


class TickerSource(app.SocialDataSource):
    """Synthetic posts that differ per ticker (length, wording, count), for the pooled-vs-single comparison."""

    WORDS = ["moon", "calls", "puts", "crash", "buy", "sell", "squeeze", "yolo", "bearish", "bullish", "the", "and", "hold",
             "dump", "rocket", "short", "earnings", "today", "İstanbul", "Kelvin"]

    def __init__(self, kind, seed, fail_for=()):
        self._kind, self.seed, self.fail_for = kind, seed, set(fail_for)

    def kind(self):
        return self._kind

    def fetch(self, ticker, limit):
        sym = ticker.as_str()
        if sym in self.fail_for:
            raise SourceFailure(self._kind.as_str(), "HTTP 429")
        rng = np.random.default_rng(zlib.crc32(sym.encode()) + self.seed)
        n = int(rng.integers(0, 40))
        out = []
        for i in range(min(n, limit)):
            k = int(rng.integers(1, 30))
            text = " ".join(self.WORDS[int(j)] for j in rng.integers(0, len(self.WORDS), k)) + " $" + sym
            out.append(SocialPost(id="%s-%d" % (sym, i), source=self._kind, author="a%d" % i, text=PostText.parse(text),
                                  created_at=NOW, engagement=int(rng.integers(0, 100))))
        return out


TICKERS = ["AAPL", "TSLA", "$$$", "GME", "BRK.B", "", "NVDA", "AMC", "toolongticker", "F", "PLTR", "aapl"]


# ----------------------------------------------------------------------------- the shared assertions
def check_tools(golden, analyzer):
    market = MockMarketSource(golden["mock_market"])
    social = fixture_social(golden)
    # tools.rs:715-734 run_scan_handles_mixed_batch
    out = batch.run_scan(batch.ScanArgs(tickers=["AAPL", "$$$"]), social, market, analyzer, now=NOW)
    assert len(out.entries) == 2
    assert out.entries[0].report is not None and out.entries[0].error is None
    assert out.entries[1].report is None and out.entries[1].error is not None
    assert "Not financial advice" in out.disclaimer
    assert out.entries[1].error == "invalid ticker: $$$"  # DomainError's Display (error.rs)
    # tools.rs:676-696 run_analyze_returns_confirming_bullish_report: the summary line and the report
    rep = out.entries[0].report
    assert "ConfirmingBullish" in batch.summarize(rep) and rep.social.total_mentions == 10
    assert batch.summarize(rep) == "AAPL — ConfirmingBullish · crowding 50% · 10 mentions (Medium)"
    # the pooled report is the single-ticker one, byte for byte
    single = app.analyze(batch.request_from("AAPL"), social, market, analyzer, now=NOW)
    assert app.report_to_json(rep) == app.report_to_json(single)
    # tools.rs:736-745 run_scan_empty_list_is_empty
    assert batch.run_scan(batch.ScanArgs(tickers=[]), social, market, analyzer).entries == []
    # tools.rs:814-832 run_compare_partitions_valid_and_invalid
    cmp_out = batch.run_compare(batch.CompareArgs(tickers=["AAPL", "$$$"], rank_by=batch.RankBy.CROWDING), social, market,
                                analyzer, now=NOW)
    assert len(cmp_out.ranked) == 1 and len(cmp_out.errors) == 1
    assert cmp_out.errors[0].ticker == "$$$" and math.isfinite(cmp_out.ranked[0].rank_metric)
    assert cmp_out.ranked[0].rank_metric == rep.fusion.crowding
    # wire format: skipped Nones, struct field order, snake_case rank_by
    js = batch.scan_output_to_json(out)
    assert js.startswith('{\n  "entries": [\n    {\n      "ticker": "AAPL",\n      "report": {\n        "ticker": "AAPL",')
    assert '    {\n      "ticker": "$$$",\n      "error": "invalid ticker: $$$"\n    }\n  ],\n  "disclaimer": "Not fin' in js
    cj = batch.compare_output_to_json(cmp_out)
    assert cj.startswith('{\n  "rank_by": "crowding",\n  "ranked": [\n    {\n      "ticker": "AAPL",\n      "rank_metric": 0.4966')
    assert '"errors": [\n    {\n      "ticker": "$$$",\n      "error": "invalid ticker: $$$"\n    }\n  ],' in cj


def check_pooled_equals_single(golden, analyzer, single_analyzer):
    """Every entry of a 12-ticker scan -- valid, invalid, empty, failing sources, lower case -- equals what
    application::analyze gives for that ticker alone: report JSON byte for byte, error strings equal."""
    market = MockMarketSource(golden["mock_market"])
    social = [TickerSource(SourceKind.REDDIT, 1, fail_for=("GME",)), TickerSource(SourceKind.BLUESKY, 2, fail_for=("GME", "F"))]
    for kw in ({}, {"enable_reddit": True}, {"no_market": True, "limit": 7}):
        out = batch.run_scan(batch.ScanArgs(tickers=TICKERS, **kw), social, market, analyzer, now=NOW)
        assert [e.ticker for e in out.entries] == TICKERS
        n_ok = 0
        for e in out.entries:
            rq = batch.request_from(e.ticker, kw.get("enable_reddit"), None, kw.get("no_market"), kw.get("limit"))
            try:
                single = app.analyze(rq, social, market, single_analyzer, now=NOW)
            except DomainError as err:
                assert e.report is None and e.error == str(err), e.ticker
                continue
            assert e.error is None and app.report_to_json(e.report) == app.report_to_json(single), e.ticker
            n_ok += 1
        assert n_ok >= 7
    # no market and every source failing for GME -> NoData for that ticker only
    out = batch.run_scan(batch.ScanArgs(tickers=["GME", "AAPL"], no_market=True), social, market, analyzer, now=NOW)
    assert out.entries[0].error == "no data: no posts and no market snapshot available" and out.entries[1].report is not None


# ----------------------------------------------------------------------------- CPU
def test_request_from_defaults_and_flags():
    # tools.rs:71-96
    r = batch.request_from("AAPL")
    assert r.enabled_sources == list(SourceKind.ALL) and r.market_enabled and r.limit == 50
    assert r.engine == EngineConfig()
    r = batch.request_from("AAPL", enable_reddit=True, no_market=True, limit=5)
    assert r.enabled_sources == [SourceKind.REDDIT] and not r.market_enabled and r.limit == 5
    r = batch.request_from("AAPL", enable_reddit=False, enable_bluesky=False)  # Some(false) x2 -> all, like None
    assert r.enabled_sources == list(SourceKind.ALL)
    r = batch.request_from("AAPL", enable_reddit=True, enable_bluesky=True)
    assert r.enabled_sources == [SourceKind.REDDIT, SourceKind.BLUESKY]


def test_list_sources_and_single_ticker_tool(golden):
    from openintel_amd.domain import InvalidTicker
    market = MockMarketSource(golden["mock_market"])
    # tools.rs:668-673 list_sources_reports_all_adapters
    out = batch.run_list_sources(fixture_social(golden), market)
    assert out == {"social": ["reddit", "bluesky"], "market": ["mock-market"]}
    # tools.rs:676-696 run_analyze_returns_confirming_bullish_report
    res = batch.run_analyze("AAPL", fixture_social(golden), market, OracleAnalyzer(), now=NOW)
    assert "ConfirmingBullish" in res.summary and res.report.social.total_mentions == 10
    assert res.dip_signal is None and "Not financial advice" in res.disclaimer  # mock market is UP +4 %: no dip signal
    # tools.rs:698-713 run_analyze_rejects_bad_ticker
    with pytest.raises(InvalidTicker):
        batch.run_analyze("$$$", fixture_social(golden), market, OracleAnalyzer())


def test_dip_rows_sentiment_is_the_single_ticker_analysis(golden):
    """application/dip.rs:175-194: social-only, all sources, limit 50; any failure is None.  Pooled over the rows of a scan
    it equals the ticker-by-ticker analysis."""
    social = [TickerSource(SourceKind.REDDIT, 1, fail_for=("GME",)), TickerSource(SourceKind.BLUESKY, 2, fail_for=("GME",))]
    rows = ["AAPL", "GME", "$$$", "TSLA", "F"]
    for analyzer in (OracleAnalyzer(), OracleSegmentAnalyzer()):
        got = batch.sentiments_for(rows, social, analyzer, now=NOW)
        assert got[1] is None and got[2] is None  # every source failed -> NoData; invalid ticker
        for t, g in zip(rows, got):
            rq = app.AnalysisRequest(ticker=t, enabled_sources=list(SourceKind.ALL), market_enabled=False, limit=50)
            try:
                rep = app.analyze(rq, social, None, OracleAnalyzer(), now=NOW)
            except DomainError:
                assert g is None
                continue
            assert g == batch.SentimentSummary(float(rep.social.net_sentiment), rep.social.total_mentions)
    assert batch.sentiments_for(rows, [], OracleAnalyzer()) == [None] * 5  # dip.rs:179-181
    # the fixture: 10 posts, net 0.5 (tests/golden: the reference's own numbers)
    s = batch.sentiments_for(["AAPL"], fixture_social(golden), OracleAnalyzer())[0]
    assert s.mentions == 10 and s.net_sentiment == 0.5


def test_tools_reference_cases_with_the_oracle(golden):
    check_tools(golden, OracleAnalyzer())
    check_tools(golden, OracleSegmentAnalyzer())


def test_pooled_scan_equals_ticker_by_ticker(golden):
    check_pooled_equals_single(golden, OracleAnalyzer(), OracleAnalyzer())
    check_pooled_equals_single(golden, OracleSegmentAnalyzer(), OracleAnalyzer())


def test_one_analyzer_call_for_the_whole_batch(golden):
    a = CountingAnalyzer()
    market = MockMarketSource(golden["mock_market"])
    out = batch.run_scan(batch.ScanArgs(tickers=["AAPL", "TSLA", "$$$", "GME"]), fixture_social(golden), market, a, now=NOW)
    assert a.calls == [30]  # 3 valid tickers x 10 fixture posts, one call (the reference: three calls of 10)
    assert [e.report is not None for e in out.entries] == [True, True, False, True]
    # nothing valid: no call at all
    a = CountingAnalyzer()
    batch.run_scan(batch.ScanArgs(tickers=["$$$", ""]), fixture_social(golden), market, a)
    assert a.calls == []


def test_analyzer_contract_violation_and_failure_reach_every_pooled_ticker(golden):
    market = MockMarketSource(golden["mock_market"])
    out = batch.run_scan(batch.ScanArgs(tickers=["AAPL", "$$$", "TSLA"]), fixture_social(golden), market, ShortAnalyzer())
    assert out.entries[0].error == out.entries[2].error == "analyzer returned 19 signals for 20 posts"  # error.rs Display
    assert out.entries[1].error == "invalid ticker: $$$"

    class Broken(PostAnalyzer):
        def analyze(self, posts):
            raise SourceFailure("hip-analyzer", "device lost")
    out = batch.run_compare(batch.CompareArgs(tickers=["AAPL", "TSLA"]), fixture_social(golden), market, Broken())
    assert out.ranked == [] and [e.error for e in out.errors] == ["data source 'hip-analyzer' failed: device lost"] * 2


def _entry(ticker, crowding, alignment=Alignment.QUIET, spec=0.0, net=0.0):
    t = Ticker.parse("AAPL")
    post = SocialPost(id="1", source=SourceKind.REDDIT, author="a", text=PostText.parse("x"), created_at=NOW, engagement=0)
    rep = SpeculationEngine.aggregate(t, [post], [PostSignal(0.0, False)], None, NOW, EngineConfig())
    rep.fusion.crowding, rep.fusion.alignment = crowding, alignment
    rep.social.speculation_index, rep.social.net_sentiment = spec, net
    return batch.RankedEntry(ticker=ticker, rank_metric=0.0, report=rep)


def test_sort_ranked_orders():
    # tools.rs:747-812 sort_ranked_orders_by_crowding_desc, with the reference's own construction
    t = Ticker.parse("AAPL")
    post = SocialPost(id="1", source=SourceKind.REDDIT, author="a", text=PostText.parse("x"), created_at=NOW, engagement=0)
    hi = SpeculationEngine.aggregate(t, [post], [PostSignal(0.0, True)], None, NOW, EngineConfig())
    lo = SpeculationEngine.aggregate(t, [post], [PostSignal(0.0, False)], None, NOW, EngineConfig())
    assert hi.fusion.crowding > lo.fusion.crowding
    ranked = [batch.RankedEntry("LO", lo.fusion.crowding, lo), batch.RankedEntry("HI", hi.fusion.crowding, hi)]
    batch.sort_ranked(ranked, batch.RankBy.CROWDING)
    assert [r.ticker for r in ranked] == ["HI", "LO"]
    # rank_metric per key (tools.rs:274-283)
    e = _entry("X", 0.25, spec=0.5, net=-0.75)
    assert batch.rank_metric(e.report, batch.RankBy.CROWDING) == 0.25 == batch.rank_metric(e.report, batch.RankBy.DIVERGENCE)
    assert batch.rank_metric(e.report, batch.RankBy.SPECULATION_INDEX) == 0.5
    assert batch.rank_metric(e.report, batch.RankBy.NET_SENTIMENT) == -0.75
    # divergence: diverging first, then by the metric; stable among equals; an unordered pair (NaN) compares equal
    rows = [_entry("A", 0.9), _entry("B", 0.2, Alignment.DIVERGING), _entry("C", 0.5), _entry("D", 0.7, Alignment.DIVERGING),
            _entry("E", 0.5)]
    for r in rows:
        r.rank_metric = r.report.fusion.crowding
    batch.sort_ranked(rows, batch.RankBy.DIVERGENCE)
    assert [r.ticker for r in rows] == ["D", "B", "A", "C", "E"]
    batch.sort_ranked(rows, batch.RankBy.CROWDING)
    assert [r.ticker for r in rows] == ["A", "D", "C", "E", "B"]
    rows = [_entry("A", 0.1), _entry("N", float("nan")), _entry("B", 0.3)]
    for r in rows:
        r.rank_metric = r.report.fusion.crowding
    batch.sort_ranked(rows, batch.RankBy.CROWDING)  # NaN is Equal to both neighbours: a stable sort leaves a valid order
    assert sorted(r.ticker for r in rows) == ["A", "B", "N"]


def test_oracle_segmented_summary_is_the_per_ticker_loop():
    from oracle import lib
    rng = np.random.default_rng(5)
    n = 5000
    pol = np.round(rng.uniform(-1, 1, n), 3)
    pol[rng.random(n) < 0.2] = 0.0
    spec = (rng.random(n) < 0.3).astype(np.uint8)
    src = (rng.random(n) < 0.5).astype(np.uint8)
    cuts = np.unique(np.concatenate([[0, n, 1, 64, 65, 128, 129], rng.integers(0, n, 60)]))
    seg = np.concatenate([cuts[:5], cuts[4:5], cuts[5:]]).astype(np.uint64)  # one empty segment
    out = lib.social_summary_segmented(src, pol, spec, seg)
    assert len(out) == seg.size - 1
    for k, o in enumerate(out):
        b, e = int(seg[k]), int(seg[k + 1])
        acc = 0.0
        for v in pol[b:e]:
            acc += v  # speculation_engine.rs:82-86
        assert o.polarity_sum == acc and o.total_mentions == e - b
        assert o.bullish == int((pol[b:e] > 0.2).sum()) and o.bearish == int((pol[b:e] < -0.2).sum())
        assert o.neutral == (e - b) - o.bullish - o.bearish and o.spec_count == int(spec[b:e].sum())
        assert o.mentions_by_source[1] == int(src[b:e].sum())
        assert o.net_sentiment == (0.0 if e == b else lib.polarity_new(acc / (e - b)))


# ----------------------------------------------------------------------------- GPU
pytest_gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import openintel_amd as oi
    c = oi.HipContext(0)
    yield oi.HipLexiconAnalyzer(c)
    c.close()


def _check_segments(hip, src, pol, spec, seg, tau=0.2):
    from oracle import lib
    cfg = lib.default_config()
    cfg.bull_bear_threshold = tau
    got = hip.summary_segments(src, pol, spec, seg, tau)
    ref = lib.social_summary_segmented(src if src is not None else np.zeros(pol.size, np.uint8), pol, spec, seg, cfg)
    assert got.size == len(ref)
    for k, o in enumerate(ref):
        g = got[k]
        assert int(g["total"]) == o.total_mentions and int(g["bullish"]) == o.bullish and int(g["bearish"]) == o.bearish, k
        assert int(g["neutral"]) == o.neutral and int(g["spec_count"]) == o.spec_count, k
        if src is not None:
            assert [int(x) for x in g["by_source"]] == [o.mentions_by_source[0], o.mentions_by_source[1]], k
        else:
            assert [int(x) for x in g["by_source"]] == [0, 0]
        # the reference's input-order sum, bit for bit (not a tolerance)
        assert np.float64(g["polarity_sum"]).tobytes() == np.float64(o.polarity_sum).tobytes(), (k, g["polarity_sum"], o.polarity_sum)
    return got


@pytest_gpu
def test_segmented_summary_bit_exact_gpu(hip):
    rng = np.random.default_rng(11)
    n = 300_000
    # polarities as the scan produces them (ratios of small integers) plus values whose sum depends on the order
    num = rng.integers(-9, 10, n)
    den = rng.integers(1, 10, n)
    pol = np.clip(num / den, -1.0, 1.0)
    pol[rng.random(n) < 0.3] = 0.0
    pol[rng.random(n) < 0.01] = -0.0
    pol[::997] = 1e-17 * rng.integers(1, 9, pol[::997].size)
    spec = (rng.random(n) < 0.25).astype(np.uint8)
    src = (rng.random(n) < 0.6).astype(np.uint8)
    lens = np.concatenate([[0, 1, 63, 64, 65, 127, 128, 129, 0, 0, 1000, 4097], rng.integers(0, 120, 3000)])
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    assert int(seg[-1]) < n
    seg = np.concatenate([seg, [n]]).astype(np.uint64)  # and one long tail segment
    _check_segments(hip, src, pol, spec, seg)
    _check_segments(hip, None, pol, spec, seg, tau=0.0)
    _check_segments(hip, src, pol, spec, seg, tau=-0.1)  # a negative threshold: bullish wins (the else-if order)
    # segments need not start at 0 nor cover every signal
    _check_segments(hip, src, pol, spec, np.array([5, 9, 9, 200], dtype=np.uint64))
    # nothing at all, segments over nothing
    assert hip.summary_segments(None, np.zeros(0), np.zeros(0, np.uint8), np.zeros(1, np.uint64)).size == 0
    z = hip.summary_segments(None, np.zeros(0), np.zeros(0, np.uint8), np.zeros(4, np.uint64))
    assert z.size == 3 and int(z["total"].sum()) == 0 and not z["polarity_sum"].any()


@pytest_gpu
def test_segmented_summary_argument_errors_gpu(hip):
    from openintel_amd import _lib
    pol, spec = np.zeros(10), np.zeros(10, np.uint8)
    with pytest.raises(_lib.OiError) as e:  # segments cover more posts than signals given (speculation_engine.rs:29-34)
        hip.summary_segments(None, pol, spec, np.array([0, 11], dtype=np.uint64))
    assert e.value.code == _lib.OI_ERR_ANALYZER_MISMATCH
    with pytest.raises(_lib.OiError) as e:
        hip.summary_segments(None, pol, spec, np.array([0, 5, 3, 10], dtype=np.uint64))
    assert e.value.code == _lib.OI_ERR_INVALID_ARG


@pytest_gpu
def test_tools_reference_cases_gpu(golden, hip):
    check_tools(golden, hip)


@pytest_gpu
def test_pooled_scan_equals_ticker_by_ticker_gpu(golden, hip):
    check_pooled_equals_single(golden, hip, OracleAnalyzer())


@pytest_gpu
def test_scan_segments_device_one_call_gpu(hip):
    """oi_lexicon_scan_segments_device: pooled posts in HBM in, one record per ticker in HBM out -- equal to the oracle's
    scan followed by the oracle's per-ticker loop."""
    import torch
    from oracle import lib
    from openintel_amd import synth
    from openintel_amd.analyzer import COUNTERS_DTYPE
    n = 200_000
    dev = torch.device("cuda:0")
    blob, offs = synth.posts_torch(n, dev, seed=41)
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 200, 2200)
    seg = np.concatenate([[0], np.cumsum(lens)])
    seg = seg[seg <= n]
    seg = np.concatenate([seg, [n]]).astype(np.int64)
    src = (torch.arange(n, device=dev) % 3 == 0).to(torch.uint8)
    d_seg = torch.from_numpy(seg).to(dev)
    d_out = torch.zeros((seg.size - 1) * 8, dtype=torch.int64, device=dev)
    d_pol = torch.zeros(n, dtype=torch.float64, device=dev)
    d_spec = torch.zeros(n, dtype=torch.uint8, device=dev)
    hip.scan_segments_device(blob, offs, src, d_seg, d_out, 0.2, d_pol, d_spec)
    d_out2 = torch.zeros_like(d_out)
    hip.scan_segments_device(blob, offs, src, d_seg, d_out2, 0.2)  # signals in the ctx workspace only
    hip.ctx.synchronize()
    assert torch.equal(d_out, d_out2)
    got = d_out.cpu().numpy().view(COUNTERS_DTYPE)
    hb, ho = blob.cpu().numpy(), offs.cpu().numpy().astype(np.uint64)
    pol, spec = lib.lexicon_analyze(hb, ho)
    assert np.array_equal(d_pol.cpu().numpy().view(np.uint64), pol.view(np.uint64)) and np.array_equal(d_spec.cpu().numpy(), spec)
    ref = lib.social_summary_segmented(src.cpu().numpy(), pol, spec, seg.astype(np.uint64))
    for k, o in enumerate(ref):
        g = got[k]
        assert (int(g["total"]), int(g["bullish"]), int(g["bearish"]), int(g["neutral"]), int(g["spec_count"])) == (
            o.total_mentions, o.bullish, o.bearish, o.neutral, o.spec_count), k
        assert [int(x) for x in g["by_source"]] == [o.mentions_by_source[0], o.mentions_by_source[1]]
        assert np.float64(g["polarity_sum"]).tobytes() == np.float64(o.polarity_sum).tobytes(), k


@pytest_gpu
def test_full_size_10M_posts_in_100K_tickers_gpu(hip):
    """10M posts pooled from ~100K tickers (the size the lexicon bench is quoted on), checked through size-independent
    properties: the integer counters of all segments add up to the unsegmented summary's; every segment's polarity_sum
    equals the input-order sum of its slice of the scan's own output (numpy, sequential) on a sample of segments."""
    import torch
    from openintel_amd import synth
    from openintel_amd.analyzer import COUNTERS_DTYPE
    n = 10_000_000
    dev = torch.device("cuda:0")
    blob, offs = synth.posts_torch(n, dev, seed=43)
    rng = np.random.default_rng(9)
    lens = rng.integers(0, 201, 100_500)
    seg = np.concatenate([[0], np.cumsum(lens)])
    seg = np.concatenate([seg[seg < n], [n]]).astype(np.int64)
    src = (torch.arange(n, device=dev) % 3 == 0).to(torch.uint8)
    d_seg = torch.from_numpy(seg).to(dev)
    n_seg = seg.size - 1
    d_out = torch.zeros(n_seg * 8, dtype=torch.int64, device=dev)
    d_pol = torch.zeros(n, dtype=torch.float64, device=dev)
    d_spec = torch.zeros(n, dtype=torch.uint8, device=dev)
    hip.scan_segments_device(blob, offs, src, d_seg, d_out, 0.2, d_pol, d_spec)
    hip.ctx.synchronize()
    got = d_out.cpu().numpy().view(COUNTERS_DTYPE)
    whole = hip.summary_device(blob, offs, src, 0.2)
    assert int(got["total"].sum()) == n == whole.total
    for f in ("bullish", "bearish", "neutral", "spec_count"):
        assert int(got[f].sum()) == getattr(whole, f), f
    assert [int(got["by_source"][:, i].sum()) for i in (0, 1)] == [whole.by_source[0], whole.by_source[1]]
    assert np.array_equal(got["total"], np.diff(seg).astype(np.uint64))
    pol = d_pol.cpu().numpy()
    for k in np.concatenate([[0, n_seg - 1], rng.integers(0, n_seg, 400)]):
        b, e = int(seg[k]), int(seg[k + 1])
        acc = np.add.accumulate(np.concatenate([[0.0], pol[b:e]]))[-1]  # sequential, input order
        assert np.float64(got["polarity_sum"][k]).tobytes() == np.float64(acc).tobytes(), k


@pytest_gpu
def test_ticker_sharded_records_equal_unsharded_gpu(hip):
    """ShardedAnalyzer.segment_summaries through the HIP path (world 1: the rank owns every ticker) and as two
    half-shards scanned one after the other on the one GPU: the records are those of the single pooled call."""
    import torch
    from openintel_amd import synth
    from openintel_amd.sharded import ShardedAnalyzer, make_hip_sharded_analyzer
    n, n_tick = 300_000, 4001
    dev = torch.device("cuda:0")
    blob, offs = synth.posts_torch(n, dev, seed=45)
    rng = np.random.default_rng(4)
    seg = np.concatenate([[0], np.sort(rng.integers(0, n + 1, n_tick - 1)), [n]]).astype(np.int64)
    src = (torch.arange(n, device=dev) % 2 == 0).to(torch.uint8)
    sa = make_hip_sharded_analyzer(hip.ctx, dev)
    whole = sa.segment_summaries(n_tick, blob, offs, src, torch.from_numpy(seg).to(dev))
    assert whole.shape == (n_tick, 8) and int(whole[:, 0].sum()) == n
    parts = []
    for rank in (0, 1):
        lo, hi = ShardedAnalyzer.ticker_bounds(n_tick, 2, rank)
        p0, p1 = int(seg[lo]), int(seg[hi])
        b0, b1 = int(offs[p0]), int(offs[p1])
        sub_offs = (offs[p0:p1 + 1] - b0).contiguous()
        sub_seg = torch.from_numpy(seg[lo:hi + 1] - seg[lo]).to(dev)
        rec = sa.scan_segments_shard(blob[b0:b1].clone(), sub_offs, src[p0:p1].contiguous(), sub_seg)
        parts.append(rec.cpu().numpy().reshape(hi - lo, 8))
    assert np.array_equal(np.concatenate(parts), whole)

