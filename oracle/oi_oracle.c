/*
 * oi_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY (see oi_oracle.h).
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off; contraction is off so
 * every f32/f64 operation below rounds exactly once, as the comments say).
 *
 * Part (1) restates the reference's per-post path; every function cites the
 * reference file:line (relative to /root/reference) it follows.
 * Part (2) is PARITY UNPINNED: the reference has no retrieval code.
 */
#include "oi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ================================================================== */
/* (1) reference-pinned path                                           */
/* ================================================================== */

/* src/adapters/analyzer/lexicon.rs:9-12 */
static const char *const BULL[] = {"moon",  "calls",    "long",   "buy",  "bullish",
                                   "squeeze", "breakout", "rocket", "pump", "rip",
                                   "green", "up",       "rally",  "bull"};
/* src/adapters/analyzer/lexicon.rs:13-27 */
static const char *const BEAR[] = {"puts",     "short",     "sell", "bearish", "dump",
                                   "crash",    "drilling",  "bagholder", "rug", "red",
                                   "down",     "tank",      "bear"};
/* src/adapters/analyzer/lexicon.rs:28-44 */
static const char *const JARGON[] = {"calls", "puts",  "0dte",   "yolo", "leaps",
                                     "theta", "gamma", "squeeze", "otm", "itm",
                                     "strike", "iv",   "delta",  "vega", "contracts"};
#define N_BULL (sizeof(BULL) / sizeof(BULL[0]))
#define N_BEAR (sizeof(BEAR) / sizeof(BEAR[0]))
#define N_JARGON (sizeof(JARGON) / sizeof(JARGON[0]))

/* src/domain/engine/config.rs:18-33 */
void oio_engine_config_default(oio_engine_config *cfg) {
    cfg->bull_bear_threshold = 0.2;
    cfg->net_sentiment_threshold = 0.05;
    cfg->price_move_threshold = 1.0;
    cfg->crowding_weight_spec = 0.5;
    cfg->crowding_weight_rvol = 0.3;
    cfg->crowding_weight_iv = 0.2;
    cfg->rvol_cap = 3.0;
    cfg->min_sample = 10;
    cfg->confidence_low = 10;
    cfg->confidence_high = 50;
}

/* Rust f64::clamp: NaN stays NaN, otherwise max(min)/min(max) by comparison. */
static double clamp_f64(double v, double lo, double hi) {
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

/* src/domain/values/polarity.rs:8-14 */
double oio_polarity_new(double v) {
    if (isnan(v)) return 0.0;
    return clamp_f64(v, -1.0, 1.0);
}

/* src/domain/values/speculation.rs:8-14 */
double oio_speculation_index_new(double v) {
    if (isnan(v)) return 0.0;
    return clamp_f64(v, 0.0, 1.0);
}

/* src/domain/values/speculation.rs:32-41 */
int oio_confidence_from_sample(uint64_t n, uint64_t low, uint64_t high) {
    uint64_t lo = low < high ? low : high, hi = low < high ? high : low;
    if (n < lo) return OIO_CONF_LOW;
    if (n < hi) return OIO_CONF_MEDIUM;
    return OIO_CONF_HIGH;
}

/* Decode one UTF-8 scalar (input is a Rust &str, hence valid). */
static uint32_t utf8_next(const uint8_t *s, uint64_t len, uint64_t *pos) {
    uint8_t b0 = s[*pos];
    uint32_t cp;
    int extra;
    if (b0 < 0x80) { cp = b0; extra = 0; }
    else if ((b0 & 0xE0) == 0xC0) { cp = b0 & 0x1F; extra = 1; }
    else if ((b0 & 0xF0) == 0xE0) { cp = b0 & 0x0F; extra = 2; }
    else if ((b0 & 0xF8) == 0xF0) { cp = b0 & 0x07; extra = 3; }
    else { cp = 0xFFFD; extra = 0; } /* stray continuation byte: non-ASCII */
    (*pos)++;
    while (extra-- > 0 && *pos < len && (s[*pos] & 0xC0) == 0x80) {
        cp = (cp << 6) | (s[*pos] & 0x3F);
        (*pos)++;
    }
    return cp;
}

/*
 * str::to_lowercase (lexicon.rs:54) restricted to what the next step can
 * observe.  lexicon.rs:56 splits on every char that is NOT ASCII alphanumeric,
 * so a lowercased char matters only if it is ASCII.  Over all of Unicode the
 * code points whose full lowercase mapping contains an ASCII char are
 *   - 'A'..'Z'            -> 'a'..'z'
 *   - U+212A KELVIN SIGN  -> 'k'
 *   - U+0130 (I with dot) -> 'i' U+0307   (SpecialCasing, unconditional)
 * (tests/golden/gen_unicode_lower_ascii.py enumerates this from the Unicode
 * database).  Every other non-ASCII char lowercases to non-ASCII chars
 * (final-sigma included) and is a separator either way.  Emits 1 or 2 chars.
 */
static int lower_observable(uint32_t cp, uint32_t out[2]) {
    if (cp >= 'A' && cp <= 'Z') { out[0] = cp + 32; return 1; }
    if (cp == 0x212A) { out[0] = 'k'; return 1; }
    if (cp == 0x0130) { out[0] = 'i'; out[1] = 0x0307; return 2; }
    out[0] = cp;
    return 1;
}

static int is_ascii_alnum(uint32_t c) {
    return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
}

static int in_list(const char *tok, size_t len, const char *const *list, size_t n) {
    for (size_t i = 0; i < n; i++)
        if (strlen(list[i]) == len && memcmp(list[i], tok, len) == 0) return 1;
    return 0;
}

/* src/adapters/analyzer/lexicon.rs:53-73 */
void oio_lexicon_score(const uint8_t *text, uint64_t len, double *polarity,
                       uint8_t *speculative, uint32_t *bull_hits_out,
                       uint32_t *bear_hits_out) {
    /* longest lexicon word is 9 chars; a longer token can never match, so the
     * buffer only needs to tell "too long" apart (tok_len keeps counting). */
    char tok[16];
    size_t tok_len = 0;
    double bull_hits = 0.0, bear_hits = 0.0; /* `.count() as f64`, :60-61 */
    int spec = 0;
    uint64_t pos = 0;
    for (;;) {
        uint32_t chars[2];
        int nch;
        int at_end = pos >= len;
        if (at_end) { chars[0] = 0x20; nch = 1; } /* flush the last token */
        else { nch = lower_observable(utf8_next(text, len, &pos), chars); } /* :54 */
        for (int c = 0; c < nch; c++) {
            if (is_ascii_alnum(chars[c])) { /* :56 split predicate */
                if (tok_len < sizeof(tok)) tok[tok_len] = (char)chars[c];
                tok_len++;
            } else if (tok_len > 0) { /* :57 empties dropped */
                if (tok_len <= sizeof(tok)) {
                    if (in_list(tok, tok_len, BULL, N_BULL)) bull_hits += 1.0;   /* :60 */
                    if (in_list(tok, tok_len, BEAR, N_BEAR)) bear_hits += 1.0;   /* :61 */
                    if (in_list(tok, tok_len, JARGON, N_JARGON)) spec = 1;       /* :67 */
                }
                tok_len = 0;
            }
        }
        if (at_end) break;
    }
    double p;
    if (bull_hits + bear_hits == 0.0) p = 0.0;                     /* :62-63 */
    else p = (bull_hits - bear_hits) / (bull_hits + bear_hits);   /* :65 */
    *polarity = oio_polarity_new(p);                               /* :70 */
    *speculative = (uint8_t)spec;
    if (bull_hits_out) *bull_hits_out = (uint32_t)bull_hits;
    if (bear_hits_out) *bear_hits_out = (uint32_t)bear_hits;
}

/* src/adapters/analyzer/lexicon.rs:82-87 */
int oio_lexicon_analyze(const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                        double *polarity_out, uint8_t *speculative_out) {
    for (uint64_t i = 0; i < n; i++)
        oio_lexicon_score(blob + offsets[i], offsets[i + 1] - offsets[i],
                          &polarity_out[i], &speculative_out[i], NULL, NULL);
    return OIO_OK;
}

/* ---- headline gate ------------------------------------------------- */

/* src/domain/dip.rs:38-55 */
const char *const OIO_CATALYST_KEYWORDS[OIO_N_CATALYST] = {
    "earnings", "miss", "guidance", "cut", "offering", "dilution", "downgrade",
    "halt", "fraud", "lawsuit", "recall", "fda", "bankruptcy", "delisting",
    "investigation", "resign"};

static int oio_ascii_alnum(uint8_t c) { /* char::is_ascii_alphanumeric */
    return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
}
static uint8_t oio_ascii_lower(uint8_t c) { /* u8::to_ascii_lowercase */
    return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c;
}

/* src/domain/dip.rs:261-272.  Every byte of a non-ASCII char is >= 0x80, hence
 * not ASCII alphanumeric: splitting on bytes equals splitting on chars. */
uint32_t oio_catalyst_hits(const uint8_t *text, uint64_t len, uint16_t *mask,
                           uint64_t *order) {
    uint16_t m = 0;
    uint64_t ord = 0;
    uint32_t n = 0;
    uint64_t i = 0;
    while (i <= len) {
        uint64_t start = i;
        while (i < len && oio_ascii_alnum(text[i])) i++;
        uint64_t tl = i - start; /* token = text[start..i), possibly empty */
        for (int k = 0; k < OIO_N_CATALYST && tl; k++) {
            const char *kw = OIO_CATALYST_KEYWORDS[k];
            if (strlen(kw) != tl) continue;
            uint64_t j = 0;
            while (j < tl && oio_ascii_lower(text[start + j]) == (uint8_t)kw[j]) j++;
            if (j == tl && !(m & (1u << k))) { /* :266 contains && not yet in hits */
                m |= (uint16_t)(1u << k);
                ord |= (uint64_t)k << (4 * n);
                n++;
            }
        }
        i++; /* the separator */
    }
    *mask = m;
    *order = ord;
    return n;
}

/* normalize_words (:204-210) + join(" ") padded with one space each side (:254) */
static uint8_t *oio_title_joined(const uint8_t *title, uint64_t len, uint64_t *out_len) {
    uint8_t *buf = (uint8_t *)malloc(len + 3);
    uint64_t o = 0;
    int first = 1;
    buf[o++] = ' ';
    for (uint64_t i = 0; i < len;) {
        if (!oio_ascii_alnum(title[i])) { i++; continue; }
        if (!first) buf[o++] = ' ';
        first = 0;
        while (i < len && oio_ascii_alnum(title[i])) buf[o++] = oio_ascii_lower(title[i++]);
    }
    buf[o++] = ' ';
    *out_len = o;
    return buf;
}

static int oio_contains(const uint8_t *hay, uint64_t hl, const uint8_t *a, uint64_t al,
                        int pad) {
    /* does hay contain (pad ? " a " : a) */
    uint64_t nl = al + (pad ? 2 : 0);
    if (nl > hl) return 0;
    for (uint64_t s = 0; s + nl <= hl; s++) {
        uint64_t j = 0;
        if (pad) {
            if (hay[s] != ' ' || hay[s + nl - 1] != ' ') continue;
            while (j < al && hay[s + 1 + j] == a[j]) j++;
        } else {
            while (j < al && hay[s + j] == a[j]) j++;
        }
        if (j == al) return 1;
    }
    return 0;
}

/* src/domain/dip.rs:247-258 */
int oio_headline_mentions_company(const uint8_t *title, uint64_t len,
                                  const uint8_t *ticker, uint64_t ticker_len,
                                  const uint8_t *forms_blob,
                                  const uint32_t *form_offsets, uint32_t n_forms) {
    uint64_t jl;
    uint8_t *joined = oio_title_joined(title, len, &jl);
    int hit = 0;
    if (ticker_len >= 2) { /* :250 title_words.contains(&ticker_lower) */
        uint8_t *tl = (uint8_t *)malloc(ticker_len);
        int wordlike = 1; /* a word never holds a space; a ticker with one cannot equal a word */
        for (uint64_t i = 0; i < ticker_len; i++) {
            tl[i] = oio_ascii_lower(ticker[i]);
            if (tl[i] == ' ') wordlike = 0;
        }
        if (wordlike) hit = oio_contains(joined, jl, tl, ticker_len, 1);
        free(tl);
    }
    for (uint32_t f = 0; f < n_forms && !hit; f++) /* :255-257 */
        hit = oio_contains(joined, jl, forms_blob + form_offsets[f],
                           form_offsets[f + 1] - form_offsets[f], 1);
    free(joined);
    return hit;
}

void oio_headline_scan(const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                       const uint8_t *ticker, uint64_t ticker_len,
                       const uint8_t *forms_blob, const uint32_t *form_offsets,
                       uint32_t n_forms, uint16_t *mask_out, uint64_t *order_out,
                       uint8_t *about_out) {
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t *t = blob + offsets[i];
        uint64_t l = offsets[i + 1] - offsets[i];
        oio_catalyst_hits(t, l, &mask_out[i], &order_out[i]);
        about_out[i] = (uint8_t)oio_headline_mentions_company(t, l, ticker, ticker_len, forms_blob,
                                                              form_offsets, n_forms);
    }
}

/* src/domain/engine/speculation_engine.rs:70-125 */
void oio_social_summary_compute(const uint8_t *sources, const double *polarity,
                                const uint8_t *speculative, uint64_t n,
                                const oio_engine_config *cfg,
                                oio_social_summary *out) {
    memset(out, 0, sizeof(*out));
    uint64_t total = n; /* :75 */
    for (uint64_t i = 0; i < n; i++) /* :77-79 */
        out->mentions_by_source[sources[i] ? OIO_SOURCE_BLUESKY : OIO_SOURCE_REDDIT]++;
    uint64_t bullish = 0, bearish = 0, neutral = 0, spec_count = 0;
    double polarity_sum = 0.0;
    for (uint64_t i = 0; i < n; i++) { /* :83-97, input order */
        double v = polarity[i];
        polarity_sum += v;
        if (v > cfg->bull_bear_threshold) bullish++;
        else if (v < -cfg->bull_bear_threshold) bearish++;
        else neutral++;
        if (speculative[i]) spec_count++;
    }
    double net = total == 0 ? 0.0 : polarity_sum / (double)total;       /* :99-103 */
    double spec_index = total == 0 ? 0.0 : (double)spec_count / (double)total; /* :104-108 */
    out->has_bull_bear_ratio = bearish != 0;                            /* :109-113 */
    out->bull_bear_ratio = bearish == 0 ? 0.0 : (double)bullish / (double)bearish;
    out->total_mentions = total;
    out->net_sentiment = oio_polarity_new(net);                         /* :118 */
    out->bullish = bullish;
    out->bearish = bearish;
    out->neutral = neutral;
    out->speculation_index = oio_speculation_index_new(spec_index);     /* :123 */
    out->spec_count = spec_count;
    out->polarity_sum = polarity_sum;
}

/* The batch callers (src/mcp/tools.rs:193-225 run_scan, :303-352 run_compare) call application::analyze once per
 * ticker, i.e. social_summary (above) once per ticker's posts.  Segment s = signals [seg[s], seg[s+1]): the same
 * function on each slice, nothing else -- the per-ticker loop of the reference written over a pooled batch. */
void oio_social_summary_segmented(const uint8_t *sources, const double *polarity, const uint8_t *speculative,
                                  const uint64_t *seg, uint64_t n_segments, const oio_engine_config *cfg,
                                  oio_social_summary *out) {
    static const uint8_t none = 0;
    for (uint64_t s = 0; s < n_segments; s++) {
        uint64_t b = seg[s], n = seg[s + 1] - seg[s];
        oio_social_summary_compute(n ? sources + b : &none, polarity + b, speculative + b, n, cfg, &out[s]);
    }
}

/* src/domain/engine/speculation_engine.rs:127-148 */
void oio_market_summary_compute(const oio_market_snapshot *m, oio_market_summary *out) {
    memset(out, 0, sizeof(*out));
    if (m->previous_close == 0.0) { /* :128-130 */
        out->note_previous_close_zero = 1;
        out->pct_change = 0.0;
    } else {
        out->pct_change = (m->last_price - m->previous_close) / m->previous_close * 100.0; /* :132 */
    }
    if (m->avg_volume == 0) { /* :134-136 */
        out->note_avg_volume_zero = 1;
        out->has_rvol = 0;
    } else {
        out->has_rvol = 1;
        out->rvol = (double)m->volume / (double)m->avg_volume; /* :138 */
    }
    out->last_price = m->last_price;
    out->has_realized_vol = m->has_realized_vol;     out->realized_vol = m->realized_vol;
    out->has_put_call_ratio = m->has_put_call_ratio; out->put_call_ratio = m->put_call_ratio;
    out->has_iv_rank = m->has_iv_rank;               out->iv_rank = m->iv_rank;
}

/* src/domain/engine/speculation_engine.rs:151-176 */
double oio_crowding(const oio_social_summary *social, const oio_market_summary *market,
                    const oio_engine_config *cfg) {
    double weighted = 0.0, weight_sum = 0.0;
    if (social->total_mentions > 0) { /* :155-158 */
        weighted += cfg->crowding_weight_spec * social->speculation_index;
        weight_sum += cfg->crowding_weight_spec;
    }
    if (market) {
        if (market->has_rvol) { /* :160-164 */
            double rvol_norm = clamp_f64(market->rvol / cfg->rvol_cap, 0.0, 1.0);
            weighted += cfg->crowding_weight_rvol * rvol_norm;
            weight_sum += cfg->crowding_weight_rvol;
        }
        if (market->has_iv_rank) { /* :165-168 */
            weighted += cfg->crowding_weight_iv * clamp_f64(market->iv_rank, 0.0, 1.0);
            weight_sum += cfg->crowding_weight_iv;
        }
    }
    if (weight_sum == 0.0) return 0.0;                  /* :171-172 */
    return clamp_f64(weighted / weight_sum, 0.0, 1.0);  /* :174 */
}

/* src/domain/engine/speculation_engine.rs:178-208 */
int oio_alignment(const oio_social_summary *social, const oio_market_summary *market,
                  const oio_engine_config *cfg, int *note_social_only) {
    if (note_social_only) *note_social_only = 0;
    if (!market) { /* :184-189 */
        if (note_social_only) *note_social_only = 1;
        return OIO_ALIGN_QUIET;
    }
    if (social->total_mentions < cfg->min_sample) return OIO_ALIGN_QUIET; /* :191-193 */
    double s = social->net_sentiment, p = market->pct_change;
    int sentiment_meaningful = fabs(s) >= cfg->net_sentiment_threshold; /* :197 */
    int price_meaningful = fabs(p) >= cfg->price_move_threshold;        /* :198 */
    if (!sentiment_meaningful || !price_meaningful) return OIO_ALIGN_QUIET;
    if (s > 0.0 && p > 0.0) return OIO_ALIGN_CONFIRMING_BULLISH;       /* :203-207 */
    if (!(s > 0.0) && !(p > 0.0)) return OIO_ALIGN_CONFIRMING_BEARISH;
    return OIO_ALIGN_DIVERGING;
}

/* src/domain/engine/speculation_engine.rs:21-68 */
int oio_aggregate(const char *ticker, const uint8_t *sources, uint64_t n_posts,
                  const double *polarity, const uint8_t *speculative,
                  uint64_t n_signals, const oio_market_snapshot *market,
                  const char *market_ticker, const oio_engine_config *cfg,
                  oio_report *out) {
    memset(out, 0, sizeof(*out));
    if (n_signals != n_posts) return OIO_ERR_ANALYZER_MISMATCH; /* :29-34 */
    if (market && market_ticker && strcmp(market_ticker, ticker) != 0)
        return OIO_ERR_MARKET_TICKER_MISMATCH; /* :36-43 */
    oio_social_summary_compute(sources, polarity, speculative, n_posts, cfg, &out->social); /* :46 */
    out->has_market = market != NULL;
    if (market) oio_market_summary_compute(market, &out->market); /* :47 */
    const oio_market_summary *ms = market ? &out->market : NULL;
    out->crowding = oio_crowding(&out->social, ms, cfg);                          /* :48 */
    out->alignment = oio_alignment(&out->social, ms, cfg, &out->note_social_only); /* :49 */
    out->social_confidence = oio_confidence_from_sample(                          /* :50-54 */
        out->social.total_mentions, cfg->confidence_low, cfg->confidence_high);
    return OIO_OK;
}

/* ================================================================== */
/* (2) PARITY UNPINNED -- builder-chosen retrieval definitions          */
/* ================================================================== */

float oio_bm25_idf(uint64_t n_docs, uint64_t df) {
    double n = (double)n_docs, d = (double)df;
    return (float)log(1.0 + (n - d + 0.5) / (d + 0.5));
}

float oio_bm25_avgdl(uint64_t total_tokens, uint64_t n_docs) {
    return (float)((double)total_tokens / (double)n_docs);
}

float oio_bm25_doc_norm(uint32_t doc_len, float avgdl) {
    const float k1 = OIO_BM25_K1, b = OIO_BM25_B;
    float ratio = (float)doc_len / avgdl;
    float t = b * ratio;
    float u = (1.0f - b) + t;
    return k1 * u;
}

float oio_bm25_impact(uint32_t tf, float doc_norm) {
    const float k1p1 = OIO_BM25_K1 + 1.0f;
    float num = (float)tf * k1p1;
    float den = (float)tf + doc_norm;
    return num / den;
}

void oio_bm25_df(const uint32_t *term_ids, const uint64_t *doc_offsets, uint64_t n_docs,
                 uint32_t vocab, uint32_t *df_out, uint64_t *total_tokens_out) {
    memset(df_out, 0, (size_t)vocab * sizeof(uint32_t));
    uint64_t *last_doc = (uint64_t *)malloc((size_t)vocab * sizeof(uint64_t));
    for (uint32_t t = 0; t < vocab; t++) last_doc[t] = UINT64_MAX;
    for (uint64_t d = 0; d < n_docs; d++)
        for (uint64_t i = doc_offsets[d]; i < doc_offsets[d + 1]; i++) {
            uint32_t t = term_ids[i];
            if (last_doc[t] != d) { last_doc[t] = d; df_out[t]++; }
        }
    free(last_doc);
    if (total_tokens_out) *total_tokens_out = doc_offsets[n_docs];
}

void oio_bm25_scores(const uint32_t *term_ids, const uint64_t *doc_offsets, uint64_t n_docs,
                     uint32_t vocab, const uint32_t *df, uint64_t n_docs_global,
                     uint64_t total_tokens_global, const uint32_t *query_terms,
                     uint32_t n_query_terms, float *scores_out) {
    uint32_t *df_local = NULL;
    if (!df) {
        df_local = (uint32_t *)malloc((size_t)vocab * sizeof(uint32_t));
        oio_bm25_df(term_ids, doc_offsets, n_docs, vocab, df_local, NULL);
        df = df_local;
    }
    float avgdl = oio_bm25_avgdl(total_tokens_global, n_docs_global);
    for (uint64_t d = 0; d < n_docs; d++) {
        uint64_t lo = doc_offsets[d], hi = doc_offsets[d + 1];
        float kd = oio_bm25_doc_norm((uint32_t)(hi - lo), avgdl);
        float s = 0.0f;
        for (uint32_t qi = 0; qi < n_query_terms; qi++) {
            uint32_t t = query_terms[qi];
            uint32_t tf = 0;
            for (uint64_t i = lo; i < hi; i++) tf += term_ids[i] == t;
            if (tf == 0) continue;
            float c = oio_bm25_idf(n_docs_global, df[t]) * oio_bm25_impact(tf, kd);
            s = s + c;
        }
        scores_out[d] = s;
    }
    free(df_local);
}

void oio_l2_normalize_rows(float *rows, uint64_t n, uint32_t dim) {
    for (uint64_t r = 0; r < n; r++) {
        float *x = rows + r * dim;
        double ss = 0.0;
        for (uint32_t k = 0; k < dim; k++) ss += (double)x[k] * (double)x[k];
        if (ss == 0.0) continue;
        float inv = (float)(1.0 / sqrt(ss));
        for (uint32_t k = 0; k < dim; k++) x[k] = x[k] * inv;
    }
}

void oio_dot_scores(const float *rows, uint64_t n, uint32_t dim, const float *q,
                    float *scores_out) {
    for (uint64_t r = 0; r < n; r++) {
        const float *x = rows + r * dim;
        double s = 0.0;
        for (uint32_t k = 0; k < dim; k++) s += (double)q[k] * (double)x[k];
        scores_out[r] = (float)s;
    }
}

/* a ranks before b: higher score first, then lower doc id. */
static int ranks_before(float sa, uint32_t da, float sb, uint32_t db) {
    if (sa > sb) return 1;
    if (sa < sb) return 0;
    return da < db;
}

typedef struct { float s; uint32_t d; } oio_entry;

static int entry_cmp(const void *pa, const void *pb) {
    const oio_entry *a = (const oio_entry *)pa, *b = (const oio_entry *)pb;
    if (ranks_before(a->s, a->d, b->s, b->d)) return -1;
    if (ranks_before(b->s, b->d, a->s, a->d)) return 1;
    return 0;
}

uint32_t oio_topk(const float *scores, uint64_t n, uint32_t k, int positive_only,
                  uint32_t doc_base, float *scores_out, uint32_t *docs_out) {
    /* bounded insertion into a sorted buffer of size k: O(n*k) worst case but
     * O(n + k log n) in expectation on unsorted input; fine for an oracle. */
    if (k == 0) return 0;
    oio_entry *best = (oio_entry *)malloc((size_t)k * sizeof(oio_entry));
    uint32_t cnt = 0;
    for (uint64_t i = 0; i < n; i++) {
        float s = scores[i] + 0.0f; /* -0.0 -> +0.0 */
        if (isnan(s)) continue;
        if (positive_only && !(s > 0.0f)) continue;
        uint32_t d = doc_base + (uint32_t)i;
        if (cnt == k && !ranks_before(s, d, best[k - 1].s, best[k - 1].d)) continue;
        uint32_t j = cnt < k ? cnt : k - 1;
        while (j > 0 && ranks_before(s, d, best[j - 1].s, best[j - 1].d)) {
            best[j] = best[j - 1];
            j--;
        }
        best[j].s = s;
        best[j].d = d;
        if (cnt < k) cnt++;
    }
    for (uint32_t i = 0; i < cnt; i++) { scores_out[i] = best[i].s; docs_out[i] = best[i].d; }
    free(best);
    return cnt;
}

uint32_t oio_merge_ranked(const float *const *scores, const uint32_t *const *docs,
                          const uint32_t *counts, uint32_t n_lists, uint32_t depth,
                          float *scores_out, uint32_t *docs_out) {
    uint64_t total = 0;
    for (uint32_t l = 0; l < n_lists; l++) total += counts[l];
    oio_entry *all = (oio_entry *)malloc((size_t)(total ? total : 1) * sizeof(oio_entry));
    uint64_t m = 0;
    for (uint32_t l = 0; l < n_lists; l++)
        for (uint32_t i = 0; i < counts[l]; i++) {
            all[m].s = scores[l][i] + 0.0f;
            all[m].d = docs[l][i];
            m++;
        }
    qsort(all, (size_t)m, sizeof(oio_entry), entry_cmp);
    uint32_t out = (uint32_t)(m < depth ? m : depth);
    for (uint32_t i = 0; i < out; i++) { scores_out[i] = all[i].s; docs_out[i] = all[i].d; }
    free(all);
    return out;
}

uint32_t oio_rrf_fuse(const uint32_t *docs_a, uint32_t n_a, const uint32_t *docs_b,
                      uint32_t n_b, uint32_t k, float *scores_out, uint32_t *docs_out) {
    uint32_t cap = n_a + n_b;
    oio_entry *u = (oio_entry *)malloc((size_t)(cap ? cap : 1) * sizeof(oio_entry));
    uint32_t m = 0;
    for (uint32_t i = 0; i < n_a; i++) {
        u[m].d = docs_a[i];
        u[m].s = 1.0f / (OIO_RRF_K + (float)(i + 1));
        m++;
    }
    for (uint32_t j = 0; j < n_b; j++) {
        float c = 1.0f / (OIO_RRF_K + (float)(j + 1));
        uint32_t hit = UINT32_MAX;
        for (uint32_t i = 0; i < n_a; i++)
            if (docs_a[i] == docs_b[j]) { hit = i; break; }
        if (hit != UINT32_MAX) u[hit].s = u[hit].s + c; /* list A first, then B */
        else { u[m].d = docs_b[j]; u[m].s = c; m++; }
    }
    qsort(u, (size_t)m, sizeof(oio_entry), entry_cmp);
    uint32_t out = m < k ? m : k;
    for (uint32_t i = 0; i < out; i++) { scores_out[i] = u[i].s; docs_out[i] = u[i].d; }
    free(u);
    return out;
}

/* ---- batch driver for the CPU baseline (bench.py) ------------------------------------------ */
#ifdef _OPENMP
#include <omp.h>
#endif

int oio_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int oio_hybrid_search_batch(const float *rows, uint64_t n_docs, uint32_t dim, const uint32_t *term_ids,
                            const uint64_t *doc_offsets, uint32_t vocab, const uint32_t *df,
                            const float *query_vecs, const uint32_t *query_terms,
                            const uint32_t *q_term_offsets, uint32_t n_queries, uint32_t depth, uint32_t k,
                            int n_threads, float *scores_out, uint32_t *docs_out, uint32_t *counts_out) {
    int used = 1;
    if (n_threads < 1) n_threads = 1;
    const uint64_t total_tokens = doc_offsets[n_docs];
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
        /* per-thread scratch: one dense score array and the two ranked lists of a query */
        float *dense = (float *)malloc((size_t)(n_docs ? n_docs : 1) * sizeof(float));
        float *cs = (float *)malloc((size_t)depth * sizeof(float)), *bs = (float *)malloc((size_t)depth * sizeof(float));
        uint32_t *cd = (uint32_t *)malloc((size_t)depth * sizeof(uint32_t));
        uint32_t *bd = (uint32_t *)malloc((size_t)depth * sizeof(uint32_t));
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t q = 0; q < (int64_t)n_queries; q++) {
            oio_dot_scores(rows, n_docs, dim, query_vecs + (size_t)q * dim, dense);
            const uint32_t nc = oio_topk(dense, n_docs, depth, 0, 0, cs, cd);
            oio_bm25_scores(term_ids, doc_offsets, n_docs, vocab, df, n_docs, total_tokens,
                            query_terms + q_term_offsets[q], q_term_offsets[q + 1] - q_term_offsets[q], dense);
            const uint32_t nb = oio_topk(dense, n_docs, depth, 1, 0, bs, bd);
            counts_out[q] = oio_rrf_fuse(cd, nc, bd, nb, k, scores_out + (size_t)q * k, docs_out + (size_t)q * k);
        }
        free(dense); free(cs); free(bs); free(cd); free(bd);
    }
    return used;
}

/* The same batch, loops re-blocked for a CPU (round 3; VERDICT r02 weak #7: the per-query driver above re-streams the
 * corpus once per query and scales 10x on 128 threads).  Rows / docs are the OUTER loop, split over the threads; every
 * row block is scored against ALL queries while it sits in cache, every doc's tokens are counted for all queries' terms
 * while they sit in L1.  Per (query, row) and per (query, doc) the arithmetic is the statement-for-statement same as
 * oio_dot_scores / oio_bm25_scores (f64 dot accumulated over k in order; the f32 BM25 sum in query order), so the
 * results are identical to oio_hybrid_search_batch's for any thread count (tests/test_oracle_retrieval.py).  The
 * selection and fusion then run per query, queries split over the threads. */
int oio_hybrid_search_batch_blocked(const float *rows, uint64_t n_docs, uint32_t dim, const uint32_t *term_ids,
                                    const uint64_t *doc_offsets, uint32_t vocab, const uint32_t *df,
                                    const float *query_vecs, const uint32_t *query_terms,
                                    const uint32_t *q_term_offsets, uint32_t n_queries, uint32_t depth, uint32_t k,
                                    int n_threads, float *scores_out, uint32_t *docs_out, uint32_t *counts_out) {
    int used = 1;
    if (n_threads < 1) n_threads = 1;
    uint32_t *df_local = NULL;
    if (!df) {
        df_local = (uint32_t *)malloc((size_t)vocab * sizeof(uint32_t));
        oio_bm25_df(term_ids, doc_offsets, n_docs, vocab, df_local, NULL);
        df = df_local;
    }
    const uint64_t total_tokens = doc_offsets[n_docs];
    const size_t nd = (size_t)(n_docs ? n_docs : 1);
    float *cosd = (float *)malloc(nd * n_queries * sizeof(float)); /* [query][doc] */
    float *bmd = (float *)malloc(nd * n_queries * sizeof(float));
    if (!cosd || !bmd) { free(cosd); free(bmd); free(df_local); return -1; }
    const float avgdl = oio_bm25_avgdl(total_tokens, n_docs);
    const uint64_t RB = 64; /* rows per block: 64 x 768 x 4 B = 192 KB, L2-resident while the queries pass over it */
    const int64_t n_blocks = (int64_t)((n_docs + RB - 1) / RB);
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(static)
#endif
        for (int64_t b = 0; b < n_blocks; b++) {
            const uint64_t r0 = (uint64_t)b * RB, r1 = r0 + RB < n_docs ? r0 + RB : n_docs;
            for (uint32_t q = 0; q < n_queries; q++) {
                const float *qv = query_vecs + (size_t)q * dim;
                float *out = cosd + (size_t)q * nd;
                for (uint64_t r = r0; r < r1; r++) { /* oio_dot_scores, one row */
                    const float *x = rows + r * dim;
                    double s = 0.0;
                    for (uint32_t kk = 0; kk < dim; kk++) s += (double)qv[kk] * (double)x[kk];
                    out[r] = (float)s;
                }
            }
            for (uint64_t d = r0; d < r1; d++) { /* oio_bm25_scores, one doc, every query */
                const uint64_t lo = doc_offsets[d], hi = doc_offsets[d + 1];
                const float kd = oio_bm25_doc_norm((uint32_t)(hi - lo), avgdl);
                for (uint32_t q = 0; q < n_queries; q++) {
                    float s = 0.0f;
                    for (uint32_t qi = q_term_offsets[q]; qi < q_term_offsets[q + 1]; qi++) {
                        const uint32_t t = query_terms[qi];
                        uint32_t tf = 0;
                        for (uint64_t i = lo; i < hi; i++) tf += term_ids[i] == t;
                        if (tf == 0) continue;
                        const float c = oio_bm25_idf(n_docs, df[t]) * oio_bm25_impact(tf, kd);
                        s = s + c;
                    }
                    bmd[(size_t)q * nd + d] = s;
                }
            }
        }
        /* (implicit barrier) selection + fusion, one query per iteration */
        float *cs = (float *)malloc((size_t)depth * sizeof(float)), *bs = (float *)malloc((size_t)depth * sizeof(float));
        uint32_t *cd = (uint32_t *)malloc((size_t)depth * sizeof(uint32_t));
        uint32_t *bd = (uint32_t *)malloc((size_t)depth * sizeof(uint32_t));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t q = 0; q < (int64_t)n_queries; q++) {
            const uint32_t nc = oio_topk(cosd + (size_t)q * nd, n_docs, depth, 0, 0, cs, cd);
            const uint32_t nb = oio_topk(bmd + (size_t)q * nd, n_docs, depth, 1, 0, bs, bd);
            counts_out[q] = oio_rrf_fuse(cd, nc, bd, nb, k, scores_out + (size_t)q * k, docs_out + (size_t)q * k);
        }
        free(cs); free(bs); free(cd); free(bd);
    }
    free(cosd);
    free(bmd);
    free(df_local);
    return used;
}

/* ---- all-cores drivers of the two text paths for the CPU baseline (bench.py) ---------------------------------------
 * The reference maps `score` over the posts on ONE thread (lexicon.rs:82-87) and scans titles one by one (dip.rs:247-272);
 * posts and titles are independent, so the all-cores baseline runs the SAME scalar function per item on n_threads threads
 * (static chunks of posts, outputs index-aligned as the port requires -- post_analyzer.rs:9).  Returns the threads used. */
int oio_lexicon_analyze_mt(const uint8_t *blob, const uint64_t *offsets, uint64_t n, double *polarity_out,
                           uint8_t *speculative_out, int n_threads) {
    int used = 1;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(static)
#endif
        for (int64_t i = 0; i < (int64_t)n; i++)
            oio_lexicon_score(blob + offsets[i], offsets[i + 1] - offsets[i], &polarity_out[i], &speculative_out[i], NULL, NULL);
    }
    return used;
}

int oio_headline_scan_mt(const uint8_t *blob, const uint64_t *offsets, uint64_t n, const uint8_t *ticker, uint64_t ticker_len,
                         const uint8_t *forms_blob, const uint32_t *form_offsets, uint32_t n_forms, uint16_t *mask_out,
                         uint64_t *order_out, uint8_t *about_out, int n_threads) {
    int used = 1;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(static)
#endif
        for (int64_t i = 0; i < (int64_t)n; i++) {
            const uint8_t *t = blob + offsets[i];
            const uint64_t l = offsets[i + 1] - offsets[i];
            oio_catalyst_hits(t, l, &mask_out[i], &order_out[i]);
            about_out[i] = (uint8_t)oio_headline_mentions_company(t, l, ticker, ticker_len, forms_blob, form_offsets, n_forms);
        }
    }
    return used;
}

