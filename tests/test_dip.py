"""Headline gate (SURVEY.md section 8 row f, rank 3): catalyst_hits / headline_mentions_company /
company_name_forms / the no_catalyst_headline gate -- src/domain/dip.rs:38-55, :204-272, :612-659.

Golden vectors are the reference's own test data (dip.rs:850-855, :1002-1100), transcribed in
tests/golden/reference_fixture.json["dip"].  CPU tests pin the oracle (C restatement and the
pure-Python one) and the host logic; `gpu` tests run the HIP kernel through the C ABI against
the same vectors and against the oracle on seeded synthetic titles (bit-exact: integer/byte work).
"""
import numpy as np
import pytest

from openintel_amd import dip, synth
from openintel_amd.analyzer import pack_posts


class OracleScanner:
    """The CPU oracle behind HeadlineScanner's interface (test infrastructure only)."""

    def __init__(self):
        from oracle import lib
        self.lib = lib
        self.keywords = lib.catalyst_keywords()

    def scan(self, titles, ticker, name_forms):
        blob, offs = pack_posts(titles)
        return self.lib.headline_scan(blob, offs, ticker, name_forms)

    def scan_rows(self, rows):  # the reference's loop: one scan per row (application/dip.rs `check` per loser)
        return [self.scan(ts, ticker, forms) for ts, ticker, forms in rows]


def _gate_cases(golden, scanner):
    for c in golden["dip"]["gate"]:
        hl = [dip.Headline(title=t, publisher=p) for p, t in c["headlines"]]
        status, evidence = dip.no_catalyst_headline(scanner, c["ticker"], c["company_names"], hl)
        e = c["expect"]
        assert status.status == e["status"], c["name"]
        if "evidence0_contains" in e:
            assert e["evidence0_contains"] in evidence[0], c["name"]
        if "evidence_empty" in e:
            assert (len(evidence) == 0) == e["evidence_empty"], c["name"]
    # the exact strings of dip.rs:627-632, :647-656
    st, ev = dip.no_catalyst_headline(scanner, "VIK", ["Viking Holdings Ltd"],
                                      [dip.Headline("Viking cuts guidance after weak bookings, earnings miss", "Wire"),
                                       dip.Headline("Sector roundup: FDA halt fears", "IBD")])
    assert st == dip.GateStatus("fail", "catalyst term(s) in company headlines: guidance, earnings, miss")
    assert ev == ['headline [Wire]: "Viking cuts guidance after weak bookings, earnings miss" '
                  '(terms: guidance, earnings, miss)']
    st, ev = dip.no_catalyst_headline(scanner, "VIK", ["Viking Holdings Ltd"],
                                      [dip.Headline("Sector roundup: FDA halt fears", "IBD"),
                                       dip.Headline("Halt lifted; fda again", "X")])
    assert st == dip.GateStatus("unknown", "catalyst term(s) only in headlines not clearly about VIK: fda, halt")
    assert ev == []
    st, ev = dip.no_catalyst_headline(scanner, "VIK", [], None, unavailable_reason="news feed down")
    assert st == dip.GateStatus("unknown", "news feed down") and ev == []


def _gate_rows(golden, scanner):
    """The gate over the rows of a scan, one scan call for all of them, equals the gate row by row -- on the reference's
    own gate cases as the rows, plus an unavailable feed, a row without headlines and a row without a company name."""
    rows = [(c["ticker"], c["company_names"], [dip.Headline(title=t, publisher=p) for p, t in c["headlines"]], "")
            for c in golden["dip"]["gate"]]
    rows.insert(2, ("VIK", [], None, "news feed down"))
    rows.append(("ZZZ", ["Zed Corp"], [], ""))
    rows.append(("UCTT", [], [dip.Headline("UCTT guidance cut", "Wire"), dip.Headline("fraud probe at rival", "X")], ""))
    got = dip.no_catalyst_headline_rows(scanner, rows)
    assert len(got) == len(rows)
    for (ticker, names, hs, why), g in zip(rows, got):
        assert g == dip.no_catalyst_headline(scanner, ticker, names, hs, unavailable_reason=why), ticker
    assert got[2] == (dip.GateStatus("unknown", "news feed down"), [])
    assert dip.no_catalyst_headline_rows(scanner, []) == []


# ----------------------------------------------------------------------------- CPU
def test_gate_over_rows_with_oracle_scanner(golden):
    _gate_rows(golden, OracleScanner())



def test_oracle_matches_reference_vectors(golden):
    from oracle import lib, pyref
    g = golden["dip"]
    assert lib.catalyst_keywords() == g["catalyst_keywords"] == pyref.CATALYST_KEYWORDS
    for c in g["catalyst_hits"]:
        assert lib.catalyst_hits(c["texts"]) == c["expect"]
        assert pyref.catalyst_hits(c["texts"]) == c["expect"]
    for c in g["headline_mentions_company"]:
        assert lib.headline_mentions_company(c["title"], c["ticker"], c["forms"]) is c["expect"], c
        assert pyref.headline_mentions_company(c["title"], c["ticker"], c["forms"]) is c["expect"], c


def test_company_name_forms_reference_vectors(golden):
    for c in golden["dip"]["company_name_forms"]:
        assert dip.company_name_forms(c["names"]) == c["expect"], c
    # the rules of dip.rs:216-243 one by one
    assert dip.company_name_forms(["Apple Inc.", "APPLE INC", "Apple"]) == ["apple"]          # dedupe
    assert dip.company_name_forms(["Box Inc"]) == []                                          # 1 word < 4 chars
    assert dip.company_name_forms(["The Trade Desk, Inc."]) == ["trade desk"]                 # `the` dropped, 2 words
    assert dip.company_name_forms(["Berkshire Hathaway Energy Co"]) == ["berkshire hathaway"]  # first two words
    assert dip.company_name_forms(["Société Générale SA"]) == ["soci t"]                      # non-ASCII splits
    assert dip.company_name_forms(["Holdings Group Trust"]) == []                             # all suffixes
    assert dip.normalize_words("Ultra-Clean  Holdings, Inc.") == ["ultra", "clean", "holdings", "inc"]


def test_oracle_c_matches_python_on_synthetic_titles():
    from oracle import lib, pyref
    titles = synth.headlines_np(3000, seed=21)
    forms = ["ultra clean", "uct", "", "Ultra Clean", "clean ", "ultra  clean", "a a b"]
    for ticker in ("UCTT", "U", "BRK.B", "uctt", ""):
        blob, offs = pack_posts(titles)
        mask, order, about = lib.headline_scan(blob, offs, ticker, forms)
        kw = lib.catalyst_keywords()
        for i, t in enumerate(titles):
            assert lib.hits_from_order(mask[i], order[i]) == pyref.catalyst_hits([t]), (i, t)
            assert bool(about[i]) == pyref.headline_mentions_company(t, ticker, forms), (i, t, ticker)
        assert all(k in kw for k in pyref.catalyst_hits(titles))
    assert mask.any() and about.any() and not about.all()


def test_gate_logic_with_oracle_scanner(golden):
    _gate_cases(golden, OracleScanner())


# ----------------------------------------------------------------------------- GPU
pytest_gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def scanner():
    import openintel_amd as oi
    c = oi.HipContext(0)
    yield oi.HeadlineScanner(c)
    c.close()


def _check_scan(scanner, titles, ticker, forms):
    from oracle import lib
    blob, offs = pack_posts(titles)
    m, o, a = scanner.scan_packed(blob, offs, ticker, forms)
    rm, ro, ra = lib.headline_scan(blob, offs, ticker, forms)
    bad = np.nonzero((m != rm) | (o != ro) | (a != ra))[0]
    assert bad.size == 0, "first mismatch at title %d %r: gpu=(%x,%x,%d) ref=(%x,%x,%d)" % (
        bad[0], titles[bad[0]][:160], m[bad[0]], o[bad[0]], a[bad[0]], rm[bad[0]], ro[bad[0]], ra[bad[0]])
    return m, o, a


@pytest_gpu
def test_reference_vectors_gpu(golden, scanner):
    g = golden["dip"]
    assert scanner.keywords == g["catalyst_keywords"]
    for c in g["catalyst_hits"]:
        assert scanner.catalyst_hits(c["texts"]) == c["expect"]
    for c in g["headline_mentions_company"]:
        assert scanner.headline_mentions_company(c["title"], c["ticker"], c["forms"]) is c["expect"], c
    _gate_cases(golden, scanner)


@pytest_gpu
def test_gate_over_rows_gpu(golden, scanner):
    _gate_rows(golden, scanner)


@pytest_gpu
def test_scan_rows_equals_row_by_row_and_the_oracle_gpu(scanner):
    """oi_headline_scan_rows: 40 rows of 0..60 titles, each with its own ticker and name forms (some without forms, one
    with an empty ticker, long rows next to empty ones) -- per row identical to oi_headline_scan and to the oracle."""
    from oracle import lib
    rng = np.random.default_rng(7)
    base = synth.headlines_np(3000, seed=27)
    companies = [("UCTT", ["ultra clean"]), ("BRK.B", ["brk b", "holdings"]), ("U", []), ("", ["ultra clean holdings"]),
                 ("VIK", dip.company_name_forms(["Viking Holdings Ltd"])), ("earnings", ["miss guidance"])]
    rows, at = [], 0
    for r in range(40):
        k = 0 if r % 9 == 4 else int(rng.integers(1, 60)) if r != 17 else 700
        ticker, forms = companies[r % len(companies)]
        rows.append((base[at:at + k], ticker, forms))
        at += k
    got = scanner.scan_rows(rows)
    assert len(got) == len(rows)
    for (titles, ticker, forms), (m, o, a) in zip(rows, got):
        blob, offs = pack_posts(titles)
        rm, ro, ra = lib.headline_scan(blob, offs, ticker, forms)
        assert np.array_equal(m, rm) and np.array_equal(o, ro) and np.array_equal(a, ra), (ticker, len(titles))
        sm, so_, sa = scanner.scan(titles, ticker, forms)
        assert np.array_equal(m, sm) and np.array_equal(o, so_) and np.array_equal(a, sa)
    assert scanner.scan_rows([]) == []
    # a call too large for the page-locked staging (> 1 MB of titles): pageable path, same results, nothing stays pinned
    pinned_before = scanner.ctx.workspace_bytes()[1]
    big_titles = synth.headlines_np(30_000, seed=5)
    big = scanner.scan_rows([(big_titles[:20_000], "UCTT", ["ultra clean"]), (big_titles[20_000:], "VIK", ["viking"])])
    for (titles, ticker, forms), (m, o, a) in zip([(big_titles[:20_000], "UCTT", ["ultra clean"]), (big_titles[20_000:], "VIK", ["viking"])], big):
        blob, offs = pack_posts(titles)
        rm, ro, ra = lib.headline_scan(blob, offs, ticker, forms)
        assert np.array_equal(m, rm) and np.array_equal(o, ro) and np.array_equal(a, ra), ticker
    assert scanner.ctx.workspace_bytes()[1] == pinned_before
    one = scanner.scan_rows([([], "UCTT", ["ultra clean"])])
    assert len(one) == 1 and one[0][0].size == 0
    # argument errors come back as errors, not as truncation: rows that do not cover the titles, descending row offsets,
    # too many name forms for one row
    from openintel_amd import _lib
    blob, offs = pack_posts(["fda halt", "guidance cut"])
    tk = np.frombuffer(b"UCTT\0", dtype=np.uint8)
    m, o, a = np.zeros(2, np.uint16), np.zeros(2, np.uint64), np.zeros(2, np.uint8)

    def rows_call(row_off):
        return scanner.ctx.lib.oi_headline_scan_rows(
            scanner.ctx.handle, _lib.ptr(blob), _lib.ptr(offs), 2, _lib.ptr(np.array(row_off, dtype=np.uint64)), len(row_off) - 1,
            _lib.ptr(tk), _lib.ptr(np.array([0] + [4] * (len(row_off) - 1), dtype=np.uint32)), None, None,
            _lib.ptr(np.zeros(len(row_off), np.uint32)), _lib.ptr(m), _lib.ptr(o), _lib.ptr(a))
    assert rows_call([0, 2]) == 0 and int(m[0]) != 0 and int(m[1]) != 0
    assert rows_call([0, 1]) == _lib.OI_ERR_INVALID_ARG       # the rows stop short of the titles
    assert rows_call([0, 2, 1, 2]) == _lib.OI_ERR_INVALID_ARG  # descending
    with pytest.raises(_lib.OiError):
        scanner.scan_rows([(["a"], "UCTT", ["w%d" % i for i in range(40)])])


@pytest_gpu
def test_synthetic_titles_match_oracle_gpu(scanner):
    titles = synth.headlines_np(200_000, seed=22)
    forms = dip.company_name_forms([synth.HEADLINE_COMPANY])
    assert forms == ["ultra clean"]
    m, o, a = _check_scan(scanner, titles, synth.HEADLINE_TICKER, forms)
    assert 0.05 < (m != 0).mean() < 0.9 and 0.001 < a.mean() < 0.5
    # every keyword is exercised, and some title holds several in a non-trivial order
    assert np.bitwise_or.reduce(m) == 0xFFFF
    assert any(bin(int(x)).count("1") >= 3 for x in m)


@pytest_gpu
def test_pattern_edge_cases_gpu(scanner):
    titles = synth.headlines_np(20_000, seed=23) + [
        "", " ", "—", "a a a b", "a a b", "x a a b y", "aab", "a  a - b", "A.A.B", "uctt", "UCTT.", "xuctt", "uctt2",
        "ultra clean", "ultra", "clean ultra", "ULTRA\tCLEAN!", "ultra cleaner", "nultra clean", "ultra é clean",
        "brk b", "BRK.B up", "u", "U U", "é", "investigation", "investigations", "investigatio", "xinvestigation",
        "resign" * 3, "fda" + "x" * 40, "q" * 5000 + " earnings", "cut " * 2000]
    for ticker, forms in [("UCTT", ["ultra clean"]), ("U", ["a a b"]), ("BRK.B", ["brk b", "", "BRK B"]),
                          ("", []), ("brk", ["ultra  clean", " ultra", "clean ", "Ultra Clean", "ultra-clean"]),
                          ("Ab", ["a", "b", "investigation resign", "x" * 600]), ("é", ["é"]), ("uc tt", [" "])]:
        _check_scan(scanner, titles, ticker, forms)


@pytest_gpu
def test_touching_titles_and_large_tiles_gpu(scanner):
    # adjacent titles share no separator in the blob: a word must end at the title's end
    titles = ["earn", "ings", "mi", "ss", "ultra", " clean", "ultra ", "clean", "uc", "tt", "cut", "cut", "fda"] * 50
    m, o, a = _check_scan(scanner, titles, "UCTT", ["ultra clean"])
    assert [int(x) for x in m[:13]] == [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 8, 1 << 11]
    assert a.sum() == 0
    # a tile of 256 titles larger than the LDS window takes the direct-from-HBM path; mix both kinds
    long_titles = [(" ".join(synth.headlines_np(6, seed=100 + i, ragged=False))) for i in range(300)]
    assert sum(len(t) for t in long_titles[:256]) > 48 * 1024
    _check_scan(scanner, long_titles + synth.headlines_np(1000, seed=24) + long_titles[:10], "UCTT", ["ultra clean"])


@pytest_gpu
def test_heavy_tailed_lengths_at_the_window_limit_gpu(scanner):
    """The tile is chosen from the AVERAGE title length (256 titles while they fill <= 7/8 of the 24 KB window): with a
    heavy-tailed length distribution some tiles fit the window, some overflow it and are redone one lane per title.
    Both kinds, next to each other, must equal the oracle; so must an average just under and just over each tile step."""
    rng = np.random.default_rng(77)
    base = synth.headlines_np(60_000, seed=26)
    titles, i = [], 0
    while i < len(base) - 64:
        k = 1 if rng.random() < 0.99 else int(rng.integers(8, 40))   # 1 % of the titles are 8..40 headlines long
        titles.append(" ".join(base[i:i + k]))
        i += k
    lens = np.array([len(t.encode()) for t in titles])
    win = 24 * 1024
    avg = int(lens.sum()) // len(lens) + 1
    tile = next(t for t in (256, 224, 192, 160, 128, 96, 64, 48, 32, 16, 8) if avg * t <= win * 7 // 8)   # headline.hip
    assert tile >= 192
    sums = np.add.reduceat(lens, np.arange(0, len(lens), tile))
    assert (sums > win).any() and (sums < win).any(), "both the windowed and the overflow path must be exercised"
    _check_scan(scanner, titles, synth.HEADLINE_TICKER, ["ultra clean"])
    # averages either side of the 7/8 rule for tiles of 256 and 224 titles (84 and 96 bytes)
    for avg in (83, 85, 95, 97):
        fixed = [(t + " " + t + " " + t)[:avg].ljust(avg, ".") for t in base[:3000]]
        _check_scan(scanner, fixed, synth.HEADLINE_TICKER, ["ultra clean"])


@pytest_gpu
def test_headline_fuzz_against_the_oracle_gpu(scanner):
    """Random titles over an alphabet that is hostile to the scan -- every keyword, its near misses and other cases, the
    company's words in and out of order, separators of every kind (none at all included: titles touch in the blob),
    multi-byte chars, digits -- in three densities: ordinary, candidate-dense (a wave's keyword ring and pattern ring
    both fill and are verified in the loop) and hit-dense (more keyword hits than list nodes: the tile is redone one lane
    per title); several tickers / name forms per seed, bit for bit against the oracle."""
    kw = list(scanner.keywords)
    near = [w[:-1] for w in kw] + [w + "s" for w in kw] + [w.upper() for w in kw] + [w.capitalize() for w in kw] + \
           ["x" + w for w in kw]
    company = ["ultra", "clean", "ultra clean", "Ultra Clean", "ULTRA  CLEAN", "ultraclean", "clean ultra", "uctt", "UCTT", "uctt2",
               "brk", "b", "BRK.B", "holdings", "ultra clean holdings"]
    filler = ["the", "stock", "up", "q3", "2026", "a", "of", "é", "\U0001F680", "\u4e2a", "x" * 13, "y" * 14, "z" * 40, "to", "in"]
    seps = [" ", " ", " ", " ", "", ".", ", ", "\n", "\t", "-", "'", "  ", "$", "\u2014", "/", "_", "\x00", ": "]
    cases = [("UCTT", ["ultra clean"]), ("BRK.B", ["brk b", "", "holdings"]), ("U", ["ultra clean holdings", "clean"]),
             ("", []), ("uctt", ["x" * 13 + " the", "é", "a of"])]
    for seed in range(8):
        rng = np.random.default_rng(2000 + seed)
        density = ("ordinary", "candidates", "hits")[seed % 3]
        pieces = {"ordinary": kw + near + company + filler * 6,
                  "candidates": kw * 4 + near * 2 + company * 4 + filler,
                  "hits": kw * 12 + company * 2 + filler}[density]
        titles = []
        for _ in range(int(rng.integers(300, 2500))):
            kind = rng.random()
            n_tok = 0 if kind < 0.05 else int(rng.integers(1, 4)) if kind < 0.3 else int(rng.integers(4, 30)) if kind < 0.985 \
                else int(rng.integers(300, 1500))
            ids = rng.integers(0, len(pieces), size=n_tok)
            sp = rng.integers(0, len(seps), size=n_tok)
            titles.append("".join(pieces[i] + seps[j] for i, j in zip(ids, sp)))
        for ticker, forms in cases[seed % 2::2] if seed % 4 else cases:
            _check_scan(scanner, titles, ticker, forms)


@pytest_gpu
def test_device_buffers_and_limits_gpu(scanner):
    import torch
    from oracle import lib
    from openintel_amd import _lib
    titles = synth.headlines_np(50_000, seed=25)
    blob, offs = pack_posts(titles)
    d_blob = torch.from_numpy(blob.copy()).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    n = len(titles)
    d_m = torch.zeros(n, dtype=torch.int16, device="cuda")
    d_o = torch.zeros(n, dtype=torch.int64, device="cuda")
    d_a = torch.zeros(n, dtype=torch.uint8, device="cuda")
    scanner.scan_device(d_blob, d_offs, "UCTT", ["ultra clean"], d_m, d_o, d_a)
    scanner.ctx.synchronize()
    rm, ro, ra = lib.headline_scan(blob, offs, "UCTT", ["ultra clean"])
    assert np.array_equal(d_m.cpu().numpy().view(np.uint16), rm)
    assert np.array_equal(d_o.cpu().numpy().view(np.uint64), ro)
    assert np.array_equal(d_a.cpu().numpy(), ra)
    # empty batch is a no-op; too many pattern bytes is an argument error, not a truncation
    assert scanner.scan([], "UCTT", ["x"])[0].size == 0
    with pytest.raises(_lib.OiError):
        scanner.scan(["a"], "UCTT", ["x" * 600, "y" * 600])
    with pytest.raises(_lib.OiError):
        scanner.scan(["a"], "UCTT", ["w%d" % i for i in range(40)])


@pytest_gpu
def test_full_size_10M_titles_gpu(scanner):
    """10M synthetic titles resident in HBM (the size the headline bench is quoted on), checked through
    size-independent properties: the oracle on a slice, planted titles, and tiling independence -- a
    sub-range scanned on its own (different tile boundaries, different window alignment) must reproduce
    the corresponding slice of the full scan bit for bit."""
    import torch
    from oracle import lib
    n = 10_000_000
    dev = torch.device("cuda:0")
    blob, offs = synth.headlines_torch(n, dev, seed=31)
    forms = dip.company_name_forms([synth.HEADLINE_COMPANY])
    d_m = torch.zeros(n, dtype=torch.int16, device=dev)
    d_o = torch.zeros(n, dtype=torch.int64, device=dev)
    d_a = torch.zeros(n, dtype=torch.uint8, device=dev)
    scanner.scan_device(blob, offs, synth.HEADLINE_TICKER, forms, d_m, d_o, d_a)
    scanner.ctx.synchronize()
    # (1) the oracle on the first 300K titles
    ns = 300_000
    hb = blob[: int(offs[ns])].cpu().numpy()
    ho = offs[: ns + 1].cpu().numpy().astype(np.uint64)
    rm, ro, ra = lib.headline_scan(hb, ho, synth.HEADLINE_TICKER, forms)
    assert np.array_equal(d_m[:ns].cpu().numpy().view(np.uint16), rm)
    assert np.array_equal(d_o[:ns].cpu().numpy().view(np.uint64), ro)
    assert np.array_equal(d_a[:ns].cpu().numpy(), ra)
    # (2) sub-ranges scanned on their own: start mid-blob at odd title indices and odd byte alignments
    for start, count in ((1_234_567, 500_001), (9_500_003, 499_997), (7, 100_000)):
        b0 = int(offs[start])
        b1 = int(offs[start + count])
        pad = (16 - b0 % 16) % 16 + 16          # re-base the slice to a different 16-byte phase
        sub = torch.zeros(pad + (b1 - b0), dtype=torch.uint8, device=dev)
        sub[pad:] = blob[b0:b1]
        so = offs[start:start + count + 1] - b0 + pad
        so[0] = 0                                # the padding joins the first title of the slice ...
        sm = torch.zeros(count, dtype=torch.int16, device=dev)
        so_ = torch.zeros(count, dtype=torch.int64, device=dev)
        sa = torch.zeros(count, dtype=torch.uint8, device=dev)
        scanner.scan_device(sub, so.contiguous(), synth.HEADLINE_TICKER, forms, sm, so_, sa)
        scanner.ctx.synchronize()
        # ... (zeros are separators: they add no token), so every title but possibly none differs
        assert torch.equal(sm, d_m[start:start + count]) and torch.equal(so_, d_o[start:start + count])
        assert torch.equal(sa, d_a[start:start + count])
    # (3) global sanity: only the 16 keyword bits, order nibbles consistent with the mask
    m = d_m.cpu().numpy().view(np.uint16)
    o = d_o.cpu().numpy().view(np.uint64)
    sample = np.random.default_rng(1).integers(0, n, size=200_000)
    for i in sample[:20_000]:
        k = bin(int(m[i])).count("1")
        nib = [(int(o[i]) >> (4 * j)) & 15 for j in range(k)]
        assert len(set(nib)) == k and all((int(m[i]) >> b) & 1 for b in nib) and int(o[i]) >> (4 * k) == 0
