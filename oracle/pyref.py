"""Independent pure-Python restatement of the lexicon scorer.

TEST INFRASTRUCTURE ONLY.  Small cases only (pure-Python loops).

Unlike ``oi_oracle.c`` (which argues that only three lowercase mappings are
observable), this one applies the FULL Unicode ``str.lower()`` -- the same
algorithm as Rust's ``str::to_lowercase`` (default case mapping +
SpecialCasing's unconditional entries + the Final_Sigma rule) -- and then the
reference's split.  Agreement of the two on adversarial Unicode input is what
tests/test_oracle_golden.py checks.

Follows /root/reference/src/adapters/analyzer/lexicon.rs:9-73.
"""
from __future__ import annotations

# lexicon.rs:9-12
BULL = ["moon", "calls", "long", "buy", "bullish", "squeeze", "breakout", "rocket", "pump",
        "rip", "green", "up", "rally", "bull"]
# lexicon.rs:13-27
BEAR = ["puts", "short", "sell", "bearish", "dump", "crash", "drilling", "bagholder", "rug",
        "red", "down", "tank", "bear"]
# lexicon.rs:28-44
JARGON = ["calls", "puts", "0dte", "yolo", "leaps", "theta", "gamma", "squeeze", "otm", "itm",
          "strike", "iv", "delta", "vega", "contracts"]


def _is_ascii_alnum(c: str) -> bool:
    return ("0" <= c <= "9") or ("a" <= c <= "z") or ("A" <= c <= "Z")


def tokens(text: str):
    lower = text.lower()  # lexicon.rs:54
    out, cur = [], []
    for ch in lower:  # lexicon.rs:55-58
        if _is_ascii_alnum(ch):
            cur.append(ch)
        else:
            if cur:
                out.append("".join(cur))
            cur = []
    if cur:
        out.append("".join(cur))
    return out


def polarity_new(v: float) -> float:  # polarity.rs:8-14
    if v != v:
        return 0.0
    return min(max(v, -1.0), 1.0)


def score(text: str):
    """(polarity, speculative, bull_hits, bear_hits) -- lexicon.rs:53-73."""
    toks = tokens(text)
    bull = float(sum(1 for t in toks if t in BULL))
    bear = float(sum(1 for t in toks if t in BEAR))
    pol = 0.0 if bull + bear == 0.0 else (bull - bear) / (bull + bear)
    spec = any(t in JARGON for t in toks)
    return polarity_new(pol), spec, int(bull), int(bear)


# ---- headline gate (src/domain/dip.rs:38-55, :204-272) -- small-case cross-check of the C
CATALYST_KEYWORDS = ["earnings", "miss", "guidance", "cut", "offering", "dilution", "downgrade", "halt", "fraud",
                     "lawsuit", "recall", "fda", "bankruptcy", "delisting", "investigation", "resign"]


def _ascii_lower(text: str) -> str:  # str::to_ascii_lowercase
    return "".join(chr(ord(c) + 32) if "A" <= c <= "Z" else c for c in text)


def _split_non_alnum(text: str):
    out, cur = [], []
    for c in text:
        if _is_ascii_alnum(c):
            cur.append(c)
        else:
            out.append("".join(cur))
            cur = []
    out.append("".join(cur))
    return out


def normalize_words(text: str):  # dip.rs:204-210
    return [w for w in _split_non_alnum(_ascii_lower(text)) if w]


def catalyst_hits(texts):  # dip.rs:261-272
    hits = []
    for text in texts:
        for token in _split_non_alnum(_ascii_lower(text)):
            if token in CATALYST_KEYWORDS and token not in hits:
                hits.append(token)
    return hits


def headline_mentions_company(title: str, ticker: str, name_forms) -> bool:  # dip.rs:247-258
    words = normalize_words(title)
    if len(ticker.encode("utf-8")) >= 2 and _ascii_lower(ticker) in words:
        return True
    joined = " %s " % " ".join(words)
    return any((" %s " % form) in joined for form in name_forms)
