"""Synthetic workloads of SURVEY.md section 8(d) (all builder-defined: the reference has no corpus).

numpy generators (host, exact seeds) feed the tests; torch generators (device) build the
bench corpora directly in HBM.  Distributions:

  embeddings   N x d f32, i.i.d. N(0,1), L2-normalised rows            seed 0x0A11CE / queries 0xB0B
  postings     vocab 131072, Zipf(s=1.07) terms, doc length LogNormal(ln 24, 0.6) clipped
               to [1,256]; queries: 4 terms from the same Zipf minus the 64 top ranks  seed 0xC0FFEE
  post text    4096-word list containing the 42 lexicon words at 3 % total mass,
               8..40 tokens per post, single-space separated                           seed 0xD1CE
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

SEED_EMB, SEED_QUERY, SEED_TEXT, SEED_LEX = 0x0A11CE, 0xB0B, 0xC0FFEE, 0xD1CE
VOCAB = 131072
ZIPF_S = 1.07
STOP_RANKS = 64

# lexicon.rs:9-44 (the 39 distinct words)
LEXICON_WORDS = sorted(set(
    ["moon", "calls", "long", "buy", "bullish", "squeeze", "breakout", "rocket", "pump", "rip", "green", "up",
     "rally", "bull", "puts", "short", "sell", "bearish", "dump", "crash", "drilling", "bagholder", "rug", "red",
     "down", "tank", "bear", "0dte", "yolo", "leaps", "theta", "gamma", "otm", "itm", "strike", "iv", "delta",
     "vega", "contracts"]))


# ------------------------------------------------------------------ numpy (tests)
def embeddings_np(n: int, dim: int, seed: int = SEED_EMB) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((n, dim), dtype=np.float32)
    x /= np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def zipf_cdf(vocab: int = VOCAB, s: float = ZIPF_S, skip: int = 0) -> np.ndarray:
    w = 1.0 / np.arange(1, vocab + 1, dtype=np.float64) ** s
    w[:skip] = 0.0
    c = np.cumsum(w)
    return c / c[-1]


def forward_index_np(n_docs: int, vocab: int = VOCAB, seed: int = SEED_TEXT, mean_len: float = 24.0,
                     max_len: int = 256) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.clip(np.rint(rng.lognormal(np.log(mean_len), 0.6, size=n_docs)), 1, max_len).astype(np.int64)
    offs = np.zeros(n_docs + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    cdf = zipf_cdf(vocab)
    terms = np.searchsorted(cdf, rng.random(int(offs[-1])), side="left").astype(np.uint32)
    np.minimum(terms, vocab - 1, out=terms)
    return terms, offs


def query_terms_np(n_queries: int, vocab: int = VOCAB, seed: int = SEED_QUERY, terms_per_query: int = 4):
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5A5A))
    cdf = zipf_cdf(vocab, skip=min(STOP_RANKS, vocab // 2))
    t = np.searchsorted(cdf, rng.random(n_queries * terms_per_query), side="left").astype(np.uint32)
    np.minimum(t, vocab - 1, out=t)
    offs = (np.arange(n_queries + 1, dtype=np.uint32) * terms_per_query).astype(np.uint32)
    return t, offs


def word_list(n_words: int = 4096, seed: int = SEED_LEX) -> Tuple[List[str], np.ndarray]:
    """(words, probabilities): the lexicon words share 3 % of the mass, the rest is uniform."""
    rng = np.random.Generator(np.random.PCG64(seed))
    words = list(LEXICON_WORDS)
    seen = set(words)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    while len(words) < n_words:
        w = "".join(rng.choice(letters, size=int(rng.integers(2, 11))))
        if w not in seen:
            seen.add(w)
            words.append(w)
    p = np.full(n_words, 0.97 / (n_words - len(LEXICON_WORDS)))
    p[:len(LEXICON_WORDS)] = 0.03 / len(LEXICON_WORDS)
    return words, p


def posts_np(n_posts: int, seed: int = SEED_LEX) -> List[str]:
    words, p = word_list(seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    n_tok = rng.integers(8, 41, size=n_posts)
    ids = rng.choice(len(words), size=int(n_tok.sum()), p=p)
    out, pos = [], 0
    for k in n_tok:
        out.append(" ".join(words[i] for i in ids[pos:pos + k]))
        pos += k
    return out


# ------------------------------------------------------------------ headlines (dip gate)
HEADLINE_COMPANY = "Ultra Clean Holdings, Inc."
HEADLINE_TICKER = "UCTT"
_CATALYST = ["earnings", "miss", "guidance", "cut", "offering", "dilution", "downgrade", "halt", "fraud", "lawsuit",
             "recall", "fda", "bankruptcy", "delisting", "investigation", "resign"]
_NEAR = ["dismissal", "cuts", "missed", "halts", "recalls", "offerings", "earning", "resigned", "fdas", "cutter",
         "lawsuits", "guidances", "investigations", "ultra", "clean", "cleanse", "ultraclean", "uctt", "UCTT", "Ultra",
         "Clean", "uct", "ucttx", "Ultra Clean", "ULTRA CLEAN", "ultra-clean", "Ultra  Clean", "ultra cleanse",
         "ultra, clean"]


def headline_words(n_words: int = 2048, seed: int = SEED_LEX) -> Tuple[List[str], np.ndarray]:
    """Words of synthetic titles: catalyst keywords in three casings (4 % of the mass), near
    misses and the company's words (6 %), random filler for the rest."""
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    hot = [f(w) for w in _CATALYST for f in (str.lower, str.upper, str.capitalize)]
    words = hot + list(_NEAR)
    seen = set(words)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    while len(words) < n_words:
        w = "".join(rng.choice(letters, size=int(rng.integers(2, 11))))
        if rng.random() < 0.3:
            w = w.capitalize()
        if rng.random() < 0.05:
            w = str(int(rng.integers(1, 5000)))
        if w not in seen:
            seen.add(w)
            words.append(w)
    p = np.full(n_words, 0.90 / (n_words - len(hot) - len(_NEAR)))
    p[:len(hot)] = 0.04 / len(hot)
    p[len(hot):len(hot) + len(_NEAR)] = 0.06 / len(_NEAR)
    return words, p


_SEPARATORS = [" ", " ", " ", " ", ", ", ": ", " - ", "'s ", "  ", ".", " $", "%, ", " — ", "é", "/", "\t", " (", ") "]


def headlines_np(n_titles: int, seed: int = SEED_LEX, ragged: bool = True) -> List[str]:
    """Titles with mixed separators (punctuation, doubled spaces, non-ASCII); when `ragged`,
    also empty titles, separator-only titles and titles that begin/end mid-word so that
    adjacent titles in the packed blob touch without a separator."""
    words, p = headline_words(seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed + 8))
    n_tok = rng.integers(1, 17, size=n_titles)
    ids = rng.choice(len(words), size=int(n_tok.sum()), p=p)
    seps = rng.integers(0, len(_SEPARATORS), size=int(n_tok.sum()))
    kind = rng.random(n_titles)
    out, pos = [], 0
    for t in range(n_titles):
        k = int(n_tok[t])
        parts = []
        for j in range(k):
            parts.append(words[ids[pos + j]])
            if j + 1 < k:
                parts.append(_SEPARATORS[seps[pos + j]])
        pos += k
        title = "".join(parts)
        if ragged:
            if kind[t] < 0.01:
                title = ""
            elif kind[t] < 0.02:
                title = " —  !"
            elif kind[t] < 0.10:
                title = _SEPARATORS[seps[pos - 1]] + title + _SEPARATORS[seps[pos - k]]
        out.append(title)
    return out


# ------------------------------------------------------------------ torch (bench, in HBM)
def embeddings_torch(n: int, dim: int, device, seed: int = SEED_EMB, chunk: int = 1 << 20):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = torch.randn((e - s, dim), generator=g, dtype=torch.float32, device=device)
        x /= x.norm(dim=1, keepdim=True)
        out[s:e] = x
    return out


def forward_index_torch(n_docs: int, device, vocab: int = VOCAB, seed: int = SEED_TEXT, mean_len: float = 24.0,
                        max_len: int = 256, chunk: int = 1 << 26):
    """(term_ids int32 [T], doc_offsets int64 [n_docs+1]) on `device`."""
    import math
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    z = torch.randn(n_docs, generator=g, device=device, dtype=torch.float32)
    lens = torch.exp(z * 0.6 + math.log(mean_len)).round().clamp_(1, max_len).to(torch.int64)
    offs = torch.zeros(n_docs + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=offs[1:])
    total = int(offs[-1].item())
    cdf = torch.from_numpy(zipf_cdf(vocab)).to(device=device, dtype=torch.float32)
    terms = torch.empty(total, dtype=torch.int32, device=device)
    for s in range(0, total, chunk):
        e = min(total, s + chunk)
        u = torch.rand(e - s, generator=g, device=device, dtype=torch.float32)
        terms[s:e] = torch.searchsorted(cdf, u).clamp_(max=vocab - 1).to(torch.int32)
    return terms, offs


def query_batch_torch(n_queries: int, dim: int, device, vocab: int = VOCAB, seed: int = SEED_QUERY,
                      terms_per_query: int = 4):
    """(query_vecs f32 [B, dim] normalised, query_terms int32 [B*4], q_term_offsets int32 [B+1])."""
    import torch
    qv = embeddings_torch(n_queries, dim, device, seed=seed)
    t, o = query_terms_np(n_queries, vocab, seed, terms_per_query)
    return (qv, torch.from_numpy(t.astype(np.int32)).to(device), torch.from_numpy(o.astype(np.int32)).to(device))


def posts_torch(n_posts: int, device, seed: int = SEED_LEX, chunk_posts: int = 1 << 19):
    """(blob uint8 [bytes], offsets int64 [n_posts+1]) on `device`, same distribution as posts_np."""
    words, p = word_list(seed=seed)
    return texts_torch(words, p, n_posts, 8, 41, device, seed, chunk_posts)


def headlines_torch(n_titles: int, device, seed: int = SEED_LEX, chunk_titles: int = 1 << 19):
    """Synthetic headline titles in HBM: headline_words() joined by single spaces, 6..16 words."""
    words, p = headline_words(seed=seed)
    return texts_torch(words, p, n_titles, 6, 17, device, seed, chunk_titles)


def texts_torch(words, p, n_posts: int, tok_lo: int, tok_hi: int, device, seed: int, chunk_posts: int = 1 << 19):
    import torch
    maxlen = max(len(w) for w in words) + 1
    table = np.full((len(words), maxlen), ord(" "), dtype=np.uint8)
    wl = np.zeros(len(words), dtype=np.int64)
    for i, w in enumerate(words):
        table[i, :len(w)] = np.frombuffer(w.encode(), dtype=np.uint8)
        wl[i] = len(w) + 1  # trailing separator
    d_table = torch.from_numpy(table).to(device)
    d_wl = torch.from_numpy(wl).to(device)
    d_p = torch.from_numpy(p).to(device=device, dtype=torch.float32)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    blobs, lens_all = [], []
    for s in range(0, n_posts, chunk_posts):
        m = min(chunk_posts, n_posts - s)
        n_tok = torch.randint(tok_lo, tok_hi, (m,), generator=g, device=device)
        T = int(n_tok.sum().item())
        ids = torch.multinomial(d_p, T, replacement=True, generator=g)
        tl = d_wl[ids]                                   # bytes per token (word + separator)
        tok_end = torch.cumsum(tl, 0)
        post_last_tok = torch.cumsum(n_tok, 0) - 1
        post_bytes_end = tok_end[post_last_tok] - 1      # drop the post's trailing separator
        tok_of_byte = torch.repeat_interleave(torch.arange(T, device=device), tl)
        within = torch.arange(tok_of_byte.numel(), device=device) - (tok_end - tl)[tok_of_byte]
        bytes_ = d_table[ids[tok_of_byte], within]
        # remove each post's final separator byte
        keep = torch.ones(bytes_.numel(), dtype=torch.bool, device=device)
        keep[post_bytes_end] = False
        blobs.append(bytes_[keep])
        prev = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), post_bytes_end[:-1] + 1])
        lens_all.append(post_bytes_end - prev)
        del tok_of_byte, within, bytes_, keep
    blob = torch.cat(blobs)
    lens = torch.cat(lens_all)
    offs = torch.zeros(n_posts + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=offs[1:])
    return blob, offs
