"""The dip screen's headline gate on the GPU (SURVEY.md section 8, row f rank 3).

Host mirror of the reference functions either side of the title scan (paths relative to the
openintel repo):

    CATALYST_KEYWORDS                src/domain/dip.rs:38-55
    NAME_SUFFIXES / normalize_words  src/domain/dip.rs:183-210
    company_name_forms               src/domain/dip.rs:216-243   (host: a handful of names)
    headline_mentions_company        src/domain/dip.rs:247-258   (GPU: oi_headline_scan)
    catalyst_hits                    src/domain/dip.rs:261-272   (GPU: oi_headline_scan)
    the no_catalyst_headline gate    src/domain/dip.rs:612-659   (host, from the GPU's per-title results)
    GateStatus                       src/domain/dip.rs:360-366

The per-title work -- tokenising, keyword lookup, company match -- runs in one kernel over all
titles (headline.hip); the gate below only folds its per-title masks.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .analyzer import pack_posts
from .context import HipContext

NAME_SUFFIXES = ["inc", "incorporated", "corp", "corporation", "ltd", "limited", "plc", "co", "company", "holdings",
                 "holding", "group", "trust", "sa", "nv", "ag"]


def catalyst_keywords() -> List[str]:
    """CATALYST_KEYWORDS as the library holds them (declaration order = bit index)."""
    lib = _lib.load()
    return [lib.oi_catalyst_keyword(i).decode("ascii") for i in range(_lib.OI_N_CATALYST_KEYWORDS)]


def _ascii_alnum(c: str) -> bool:
    return ("0" <= c <= "9") or ("a" <= c <= "z") or ("A" <= c <= "Z")


def normalize_words(text: str) -> List[str]:
    """dip.rs:204-210: ASCII-lowercase, split on every char that is not ASCII alphanumeric."""
    out, cur = [], []
    for c in text:
        if _ascii_alnum(c):
            cur.append(c.lower())  # ASCII letters only reach here
        elif cur:
            out.append("".join(cur))
            cur = []
    if cur:
        out.append("".join(cur))
    return out


def company_name_forms(company_names: Sequence[str]) -> List[str]:
    """dip.rs:216-243."""
    forms: List[str] = []
    for name in company_names:
        words = normalize_words(name)
        while words and words[-1] in NAME_SUFFIXES:
            words.pop()
        if words and words[0] == "the":
            words.pop(0)
        if len(words) == 0:
            continue
        if len(words) == 1:
            if len(words[0]) < 4:  # single-word names need some length
                continue
            form = words[0]
        else:
            form = " ".join(words[:2])
        if form not in forms:
            forms.append(form)
    return forms


def pack_forms(forms: Sequence[str]):
    enc = [f.encode("utf-8") for f in forms]
    offs = np.zeros(len(enc) + 1, dtype=np.uint32)
    if enc:
        offs[1:] = np.cumsum(np.fromiter((len(e) for e in enc), dtype=np.uint64, count=len(enc)))
    blob = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8)
    return blob, offs


def hits_from_order(mask: int, order: int, keywords: Sequence[str]) -> List[str]:
    """Per-title result words -> the Vec<String> catalyst_hits(&[title]) returns."""
    return [keywords[(int(order) >> (4 * j)) & 15] for j in range(bin(int(mask)).count("1"))]


class HeadlineScanner:
    """oi_headline_scan behind the reference's two function names."""

    def __init__(self, ctx: HipContext):
        self.ctx = ctx
        self.keywords = catalyst_keywords()

    def scan_packed(self, blob: np.ndarray, offsets: np.ndarray, ticker: str, name_forms: Sequence[str]):
        """Host buffers in, host arrays out: (mask u16[n], order u64[n], about u8[n])."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        mask = np.zeros(n, np.uint16)
        order = np.zeros(n, np.uint64)
        about = np.zeros(n, np.uint8)
        if n == 0:
            return mask, order, about
        tk = np.frombuffer(ticker.encode("utf-8") + b"\0", dtype=np.uint8)
        fblob, foffs = pack_forms(name_forms)
        _lib.check(self.ctx.lib.oi_headline_scan(
            self.ctx.handle, _lib.ptr(blob) if blob.size else None, _lib.ptr(offsets), n, _lib.ptr(tk), tk.size - 1,
            _lib.ptr(fblob), _lib.ptr(foffs), len(name_forms), _lib.ptr(mask), _lib.ptr(order), _lib.ptr(about)))
        return mask, order, about

    def scan_device(self, d_blob, d_offsets, ticker: str, name_forms: Sequence[str], d_mask, d_order, d_about) -> None:
        """torch CUDA tensors in HBM (uint8 blob, int64/uint64 offsets[n+1]; int16/uint16, int64/uint64, uint8
        outputs of n); asynchronous on the ctx stream."""
        n = d_offsets.numel() - 1
        tk = np.frombuffer(ticker.encode("utf-8") + b"\0", dtype=np.uint8)
        fblob, foffs = pack_forms(name_forms)
        _lib.check(self.ctx.lib.oi_headline_scan_device(
            self.ctx.handle, _lib.ptr(d_blob), _lib.ptr(d_offsets), n, d_blob.numel(), _lib.ptr(tk), tk.size - 1,
            _lib.ptr(fblob), _lib.ptr(foffs), len(name_forms), _lib.ptr(d_mask), _lib.ptr(d_order),
            _lib.ptr(d_about)))

    def scan(self, titles: Sequence[str], ticker: str, name_forms: Sequence[str]):
        blob, offs = pack_posts(titles)
        return self.scan_packed(blob, offs, ticker, name_forms)

    def scan_rows(self, rows: Sequence[Tuple[Sequence[str], str, Sequence[str]]]):
        """The rows of a dip scan in ONE call (oi_headline_scan_rows): rows[r] = (titles, ticker, name_forms).  Returns, per
        row, what `scan(titles, ticker, name_forms)` returns -- (mask u16[n_r], order u64[n_r], about u8[n_r])."""
        titles = [t for r in rows for t in r[0]]
        blob, offs = pack_posts(titles)
        row_off = np.zeros(len(rows) + 1, dtype=np.uint64)
        tick_off = np.zeros(len(rows) + 1, dtype=np.uint32)
        rform_off = np.zeros(len(rows) + 1, dtype=np.uint32)
        tickers, forms = [], []
        for i, (ts, ticker, name_forms) in enumerate(rows):
            tb = ticker.encode("utf-8")
            tickers.append(tb)
            forms.extend(name_forms)
            row_off[i + 1] = row_off[i] + len(ts)
            tick_off[i + 1] = tick_off[i] + len(tb)
            rform_off[i + 1] = rform_off[i] + len(name_forms)
        tblob = np.frombuffer(b"".join(tickers) + b"\0", dtype=np.uint8)
        fblob, foffs = pack_forms(forms)
        n = len(titles)
        mask, order, about = np.zeros(n, np.uint16), np.zeros(n, np.uint64), np.zeros(n, np.uint8)
        if rows:
            _lib.check(self.ctx.lib.oi_headline_scan_rows(
                self.ctx.handle, _lib.ptr(blob) if blob.size else None, _lib.ptr(offs), n, _lib.ptr(row_off), len(rows),
                _lib.ptr(tblob), _lib.ptr(tick_off), _lib.ptr(fblob), _lib.ptr(foffs), _lib.ptr(rform_off),
                _lib.ptr(mask), _lib.ptr(order), _lib.ptr(about)))
        out = []
        for i in range(len(rows)):
            a, b = int(row_off[i]), int(row_off[i + 1])
            out.append((mask[a:b], order[a:b], about[a:b]))
        return out

    # ------------------------------------------------------------------ reference API
    def catalyst_hits(self, texts: Sequence[str]) -> List[str]:
        """dip.rs:261-272: hits across the texts, deduped, first-occurrence order."""
        mask, order, _ = self.scan(texts, "", [])
        hits: List[str] = []
        for m, o in zip(mask, order):
            for h in hits_from_order(m, o, self.keywords):
                if h not in hits:
                    hits.append(h)
        return hits

    def headline_mentions_company(self, title: str, ticker: str, name_forms: Sequence[str]) -> bool:
        """dip.rs:247-258."""
        return bool(self.scan([title], ticker, name_forms)[2][0])


# ----------------------------------------------------------------------------- the gate
@dataclass(frozen=True)
class GateStatus:  # dip.rs:360-366, serde: {"status": "pass"|"fail"|"unknown", "reason": ...}
    status: str
    reason: Optional[str] = None

    PASS = None  # filled below


GateStatus.PASS = GateStatus("pass")


@dataclass
class Headline:  # src/domain/values/headline.rs
    title: str
    publisher: str
    published_at: object = None


def no_catalyst_headline(scanner: HeadlineScanner, ticker: str, company_names: Sequence[str],
                         headlines: Optional[Sequence[Headline]],
                         unavailable_reason: str = "") -> Tuple[GateStatus, List[str]]:
    """dip.rs:612-659: (gate status, catalyst_evidence lines this gate contributes).
    `headlines is None` is GateEvidence::Unavailable(unavailable_reason)."""
    if headlines is None:
        return GateStatus("unknown", unavailable_reason), []
    name_forms = company_name_forms(company_names)  # blank/junk names derive no forms -> strict path
    mask, order, about = scanner.scan([h.title for h in headlines], ticker, name_forms)
    return _gate_from_scan(scanner, ticker, name_forms, headlines, mask, order, about)


def no_catalyst_headline_rows(scanner: HeadlineScanner, rows) -> List[Tuple[GateStatus, List[str]]]:
    """The gate for every row of a dip scan (application/dip.rs: `check` per loser) with ONE scan call for all rows'
    headlines.  rows[r] = (ticker, company_names, headlines or None, unavailable_reason); the result per row is what
    `no_catalyst_headline` returns for it."""
    live = [(i, t, company_name_forms(names), hs) for i, (t, names, hs, _) in enumerate(rows) if hs is not None]
    scans = scanner.scan_rows([([h.title for h in hs], t, forms) for _, t, forms, hs in live]) if live else []
    out: List[Optional[Tuple[GateStatus, List[str]]]] = [None] * len(rows)
    for (i, t, forms, hs), (mask, order, about) in zip(live, scans):
        out[i] = _gate_from_scan(scanner, t, forms, hs, mask, order, about)
    for i, (_, _, hs, why) in enumerate(rows):
        if hs is None:
            out[i] = (GateStatus("unknown", why), [])
    return out


def _gate_from_scan(scanner: HeadlineScanner, ticker: str, name_forms: Sequence[str], headlines: Sequence[Headline],
                    mask, order, about) -> Tuple[GateStatus, List[str]]:
    evidence: List[str] = []
    matched: List[str] = []
    unmatched: List[str] = []
    for h, m, o, a in zip(headlines, mask, order, about):
        title_hits = hits_from_order(m, o, scanner.keywords)
        if not title_hits:
            continue
        about_company = (not name_forms) or bool(a)  # :624-625
        if about_company:
            evidence.append('headline [%s]: "%s" (terms: %s)' % (h.publisher, h.title, ", ".join(title_hits)))
        bucket = matched if about_company else unmatched
        for hit in title_hits:
            if hit not in bucket:
                bucket.append(hit)
    if matched:
        return GateStatus("fail", "catalyst term(s) in company headlines: %s" % ", ".join(matched)), evidence
    if unmatched:
        return GateStatus("unknown", "catalyst term(s) only in headlines not clearly about %s: %s"
                          % (ticker, ", ".join(unmatched))), evidence
    return GateStatus.PASS, evidence
