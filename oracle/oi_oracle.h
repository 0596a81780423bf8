/*
 * oi_oracle.h -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * import, link, call or execute anything under oracle/.  The product path
 * (openintel_amd/ and libopenintel_hip.so) never touches this code; it fails
 * loudly when the HIP library is missing instead of falling back to it.
 *
 * Two families of functions live here:
 *
 *  (1) REFERENCE-PINNED: a plain-C restatement of the one per-post path the
 *      reference (Kloudy-Sky/openintel, Rust) really has -- keyword-lexicon
 *      scoring + social summary + fusion scalars.  Each function cites the
 *      reference file:line it follows (paths relative to /root/reference).
 *      Pinned by the reference's own fixtures and assertions, see
 *      tests/golden/ and tests/test_oracle_golden.py.
 *
 *  (2) PARITY UNPINNED: BM25, cosine, top-k, RRF.  The reference contains no
 *      retrieval code at all, so these restate the
 *      published textbook definitions with BUILDER-CHOSEN parameters, all of
 *      which are collected in the OIO_* constants below.  They are the checker
 *      for the HIP kernels; nothing here "matches the reference".
 */
#ifndef OI_ORACLE_H
#define OI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* (1) reference-pinned path                                           */
/* ------------------------------------------------------------------ */

/* src/domain/values/source_kind.rs:5-8 -- Ord derives Reddit < Bluesky. */
enum { OIO_SOURCE_REDDIT = 0, OIO_SOURCE_BLUESKY = 1, OIO_N_SOURCES = 2 };

/* src/domain/values/speculation.rs:45-51 */
enum {
    OIO_ALIGN_CONFIRMING_BULLISH = 0,
    OIO_ALIGN_CONFIRMING_BEARISH = 1,
    OIO_ALIGN_DIVERGING = 2,
    OIO_ALIGN_QUIET = 3
};
/* src/domain/values/speculation.rs:24-28 */
enum { OIO_CONF_LOW = 0, OIO_CONF_MEDIUM = 1, OIO_CONF_HIGH = 2 };

/* src/domain/error.rs:4-22 (only the variants reachable on this path) */
enum {
    OIO_OK = 0,
    OIO_ERR_ANALYZER_MISMATCH = -3,
    OIO_ERR_MARKET_TICKER_MISMATCH = -4
};

/* src/domain/engine/config.rs:2-33 */
typedef struct {
    double bull_bear_threshold;     /* tau   = 0.2  */
    double net_sentiment_threshold; /* sigma = 0.05 */
    double price_move_threshold;    /* delta = 1.0  */
    double crowding_weight_spec;    /* 0.5 */
    double crowding_weight_rvol;    /* 0.3 */
    double crowding_weight_iv;      /* 0.2 */
    double rvol_cap;                /* 3.0 */
    uint64_t min_sample;            /* 10 */
    uint64_t confidence_low;        /* 10 */
    uint64_t confidence_high;       /* 50 */
} oio_engine_config;

/* src/domain/entities/speculation_report.rs:12-21 */
typedef struct {
    uint64_t total_mentions;
    uint64_t mentions_by_source[OIO_N_SOURCES];
    double net_sentiment;
    uint64_t bullish, bearish, neutral;
    int has_bull_bear_ratio; /* Option<f64> */
    double bull_bear_ratio;
    double speculation_index;
    /* not in the reference struct; kept for the parity tests */
    uint64_t spec_count;
    double polarity_sum;
} oio_social_summary;

/* src/domain/entities/market_snapshot.rs:7-17 (Option<f64> as has_/value) */
typedef struct {
    double last_price, previous_close;
    uint64_t volume, avg_volume;
    int has_realized_vol;   double realized_vol;
    int has_put_call_ratio; double put_call_ratio;
    int has_iv_rank;        double iv_rank;
} oio_market_snapshot;

/* src/domain/entities/speculation_report.rs:24-31 */
typedef struct {
    double last_price, pct_change;
    int has_rvol; double rvol;
    int has_realized_vol;   double realized_vol;
    int has_put_call_ratio; double put_call_ratio;
    int has_iv_rank;        double iv_rank;
    /* notes pushed by market_summary (speculation_engine.rs:129,135) */
    int note_previous_close_zero, note_avg_volume_zero;
} oio_market_summary;

/* src/domain/entities/speculation_report.rs:34-48, flattened */
typedef struct {
    oio_social_summary social;
    int has_market;
    oio_market_summary market;
    int alignment;
    double crowding;
    int note_social_only; /* speculation_engine.rs:186 */
    int social_confidence;
} oio_report;

void oio_engine_config_default(oio_engine_config *cfg);

/* A3: src/domain/values/polarity.rs:8-14 */
double oio_polarity_new(double v);
/* src/domain/values/speculation.rs:8-14 */
double oio_speculation_index_new(double v);
/* src/domain/values/speculation.rs:32-41 */
int oio_confidence_from_sample(uint64_t n, uint64_t low, uint64_t high);

/* A1: src/adapters/analyzer/lexicon.rs:53-73.  `text` is valid UTF-8.
 * bull_hits/bear_hits may be NULL. */
void oio_lexicon_score(const uint8_t *text, uint64_t len, double *polarity,
                       uint8_t *speculative, uint32_t *bull_hits,
                       uint32_t *bear_hits);

/* A2: src/adapters/analyzer/lexicon.rs:82-87.  Post i is
 * blob[offsets[i] .. offsets[i+1]).  Always returns OIO_OK (the reference
 * impl is infallible). */
int oio_lexicon_analyze(const uint8_t *blob, const uint64_t *offsets,
                        uint64_t n, double *polarity_out,
                        uint8_t *speculative_out);

/* A4: src/domain/engine/speculation_engine.rs:70-125 */
void oio_social_summary_compute(const uint8_t *sources, const double *polarity,
                                const uint8_t *speculative, uint64_t n,
                                const oio_engine_config *cfg,
                                oio_social_summary *out);

/* A5: src/domain/engine/speculation_engine.rs:127-148 */
void oio_market_summary_compute(const oio_market_snapshot *m,
                                oio_market_summary *out);
/* A5: src/domain/engine/speculation_engine.rs:151-176 */
void oio_social_summary_segmented(const uint8_t *sources, const double *polarity, const uint8_t *speculative,
                                  const uint64_t *seg, uint64_t n_segments, const oio_engine_config *cfg,
                                  oio_social_summary *out);
double oio_crowding(const oio_social_summary *social,
                    const oio_market_summary *market /* NULL = None */,
                    const oio_engine_config *cfg);
/* A5: src/domain/engine/speculation_engine.rs:178-208 */
int oio_alignment(const oio_social_summary *social,
                  const oio_market_summary *market /* NULL = None */,
                  const oio_engine_config *cfg, int *note_social_only);

/* A6: src/domain/engine/speculation_engine.rs:21-68.
 * n_posts/n_signals model posts.len()/signals.len(); ticker strings model the
 * MarketTickerMismatch check at :36-43 (market_ticker NULL = no market). */
int oio_aggregate(const char *ticker, const uint8_t *sources, uint64_t n_posts,
                  const double *polarity, const uint8_t *speculative,
                  uint64_t n_signals, const oio_market_snapshot *market,
                  const char *market_ticker, const oio_engine_config *cfg,
                  oio_report *out);

/* ---- headline gate, reference-pinned ------- */

/* src/domain/dip.rs:38-55, in declaration order (index = bit / nibble value). */
#define OIO_N_CATALYST 16
extern const char *const OIO_CATALYST_KEYWORDS[OIO_N_CATALYST];

/* src/domain/dip.rs:261-272 on ONE text: ASCII-lowercase, split on every char
 * that is not ASCII alphanumeric, keep tokens that are catalyst keywords,
 * deduped in first-occurrence order.  *mask: bit i set iff keyword i hit.
 * *order: nibble j (bits 4j..4j+3) = keyword index of the j-th distinct hit.
 * Returns the number of distinct hits.  (The multi-text call of the reference
 * is the concatenation of per-text results with the same dedupe.) */
uint32_t oio_catalyst_hits(const uint8_t *text, uint64_t len, uint16_t *mask,
                           uint64_t *order);

/* src/domain/dip.rs:247-258 (with normalize_words, :204-210).  Form i is
 * forms_blob[form_offsets[i] .. form_offsets[i+1]).  Returns 0/1. */
int oio_headline_mentions_company(const uint8_t *title, uint64_t len,
                                  const uint8_t *ticker, uint64_t ticker_len,
                                  const uint8_t *forms_blob,
                                  const uint32_t *form_offsets,
                                  uint32_t n_forms);

/* Both of the above over a batch of titles (the dip gate's loop, dip.rs:619-626). */
void oio_headline_scan(const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                       const uint8_t *ticker, uint64_t ticker_len,
                       const uint8_t *forms_blob, const uint32_t *form_offsets,
                       uint32_t n_forms, uint16_t *mask_out, uint64_t *order_out,
                       uint8_t *about_out);

/* ------------------------------------------------------------------ */
/* (2) PARITY UNPINNED -- builder-chosen retrieval definitions          */
/* ------------------------------------------------------------------ */

#define OIO_BM25_K1 1.2f
#define OIO_BM25_B 0.75f
#define OIO_RRF_K 60.0f

/* idf_t = (float) ln(1 + (N - df + 0.5) / (df + 0.5)), evaluated in double. */
float oio_bm25_idf(uint64_t n_docs, uint64_t df);
/* (float) ((double) total_tokens / (double) n_docs) */
float oio_bm25_avgdl(uint64_t total_tokens, uint64_t n_docs);
/* Kd = k1 * ((1 - b) + b * (dl / avgdl)), every op rounded to f32. */
float oio_bm25_doc_norm(uint32_t doc_len, float avgdl);
/* w = (tf * (k1 + 1)) / (tf + Kd), every op rounded to f32. */
float oio_bm25_impact(uint32_t tf, float doc_norm);

/* Dense BM25 scores of one query against a forward index.
 * doc d owns tokens term_ids[doc_offsets[d] .. doc_offsets[d+1]).
 * score[d] = sum over the query's terms IN QUERY ORDER (a repeated term
 * counts each time) of idf_t * w(t,d), f32 adds starting from +0.0f; terms
 * absent from d contribute nothing.  df/total_tokens/n_docs_global describe
 * the GLOBAL collection (equal to the local one when unsharded); df may be
 * NULL to derive it from this forward index. */
void oio_bm25_scores(const uint32_t *term_ids, const uint64_t *doc_offsets,
                     uint64_t n_docs, uint32_t vocab, const uint32_t *df,
                     uint64_t n_docs_global, uint64_t total_tokens_global,
                     const uint32_t *query_terms, uint32_t n_query_terms,
                     float *scores_out);

/* Document frequencies + token count of a forward index. */
void oio_bm25_df(const uint32_t *term_ids, const uint64_t *doc_offsets,
                 uint64_t n_docs, uint32_t vocab, uint32_t *df_out,
                 uint64_t *total_tokens_out);

/* Row-wise L2 normalisation in f32 (sum of squares accumulated in double,
 * inv = 1/sqrt, then one f32 multiply per element).  Zero rows stay zero. */
void oio_l2_normalize_rows(float *rows, uint64_t n, uint32_t dim);

/* scores[d] = (float) sum_k (double) q[k] * (double) rows[d][k].  The HIP
 * path sums in f32 in a different order: parity bar 1e-5 absolute. */
void oio_dot_scores(const float *rows, uint64_t n, uint32_t dim,
                    const float *q, float *scores_out);

/* Top-k of a dense score array: score descending, doc id ascending on ties
 * (-0.0 == +0.0); NaN never selected.  If positive_only != 0 only scores > 0
 * are candidates (BM25: untouched docs are not results).  Returns the number
 * of results written (<= k).  doc ids written are doc_base + index. */
uint32_t oio_topk(const float *scores, uint64_t n, uint32_t k,
                  int positive_only, uint32_t doc_base, float *scores_out,
                  uint32_t *docs_out);

/* Merge `n_lists` ranked lists (each sorted as oio_topk sorts, disjoint doc
 * ids -- the per-shard lists of one query) into the global top-`depth`. */
uint32_t oio_merge_ranked(const float *const *scores,
                          const uint32_t *const *docs, const uint32_t *counts,
                          uint32_t n_lists, uint32_t depth, float *scores_out,
                          uint32_t *docs_out);

/* Reciprocal-rank fusion of two ranked lists of one query.
 * rrf(d) = sum over lists containing d of 1.0f / (60.0f + (float) rank),
 * rank starting at 1, list A added before list B, f32 ops.  Output: top-k by
 * (rrf descending, doc id ascending).  Returns the number written. */
uint32_t oio_rrf_fuse(const uint32_t *docs_a, uint32_t n_a,
                      const uint32_t *docs_b, uint32_t n_b, uint32_t k,
                      float *scores_out, uint32_t *docs_out);

/* A batch of hybrid queries, one after the other or on `n_threads` host threads (OpenMP over the QUERIES: every
 * query is still the scalar single-thread pipeline above -- oio_dot_scores, oio_topk, oio_bm25_scores, oio_topk,
 * oio_rrf_fuse -- so the results are identical for any thread count).  bench.py's cpu_baseline times this
 * (n_threads = 1 and n_threads = all cores); tests compare it with the per-query calls.  Query q owns
 * query_terms[q_term_offsets[q] .. q_term_offsets[q+1]).  Outputs row stride k; counts_out[q] valid entries.
 * Returns the number of threads actually used (1 when built without OpenMP). */
int oio_hybrid_search_batch(const float *rows, uint64_t n_docs, uint32_t dim, const uint32_t *term_ids,
                            const uint64_t *doc_offsets, uint32_t vocab, const uint32_t *df,
                            const float *query_vecs, const uint32_t *query_terms,
                            const uint32_t *q_term_offsets, uint32_t n_queries, uint32_t depth, uint32_t k,
                            int n_threads, float *scores_out, uint32_t *docs_out, uint32_t *counts_out);
/* The same batch and the same results, loops re-blocked for a CPU (rows / docs outermost, split over the threads; every
 * row block scored against all queries while it is in cache).  bench.py's all-cores baseline. */
int oio_hybrid_search_batch_blocked(const float *rows, uint64_t n_docs, uint32_t dim, const uint32_t *term_ids,
                            const uint64_t *doc_offsets, uint32_t vocab, const uint32_t *df,
                            const float *query_vecs, const uint32_t *query_terms,
                            const uint32_t *q_term_offsets, uint32_t n_queries, uint32_t depth, uint32_t k,
                            int n_threads, float *scores_out, uint32_t *docs_out, uint32_t *counts_out);
/* Host threads OpenMP would use by default (1 when built without it). */
int oio_max_threads(void);
/* The two text paths on n_threads host threads (bench.py's all-cores CPU baselines): the same scalar function per post /
 * title as oio_lexicon_analyze / oio_headline_scan, static chunks, index-aligned outputs.  Return the threads used. */
int oio_lexicon_analyze_mt(const uint8_t *blob, const uint64_t *offsets, uint64_t n, double *polarity_out,
                           uint8_t *speculative_out, int n_threads);
int oio_headline_scan_mt(const uint8_t *blob, const uint64_t *offsets, uint64_t n, const uint8_t *ticker, uint64_t ticker_len,
                         const uint8_t *forms_blob, const uint32_t *form_offsets, uint32_t n_forms, uint16_t *mask_out,
                         uint64_t *order_out, uint8_t *about_out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* OI_ORACLE_H */
