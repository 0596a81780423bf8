/*
 * openintel_hip.h -- C ABI of libopenintel_hip.so (MI355X / gfx950, hand-written HIP).
 *
 * This is the drop-in boundary: plain pointers and sizes, int status codes, no
 * exceptions across the ABI, no framework types.  A Rust `extern "C"` block (or
 * ctypes) binds these symbols directly; see INTEGRATION.md for the Rust shim that
 * implements the reference's port trait on top of them.
 *
 * Reference interfaces replaced (paths relative to the openintel repo):
 *
 *   oi_lexicon_analyze*      <- trait PostAnalyzer::analyze
 *                               src/domain/ports/post_analyzer.rs:7-11, implemented by
 *                               LexiconAnalyzer   src/adapters/analyzer/lexicon.rs:53-87
 *                               (called at src/application/analyze.rs:62-63)
 *   oi_social_summary*       <- SpeculationEngine::social_summary
 *                               src/domain/engine/speculation_engine.rs:70-125
 *   oi_social_summary_segmented / oi_lexicon_scan_segments_device
 *                            <- the per-ticker loop of the batch callers run_scan / run_compare
 *                               src/mcp/tools.rs:193-225, :303-352 (each iteration = analyze.rs:61-63 +
 *                               speculation_engine.rs:70-125), as ONE pooled scan + one reduction per ticker
 *   oi_headline_scan*        <- catalyst_hits + headline_mentions_company over the dip
 *                               gate's titles   src/domain/dip.rs:247-272 (loop at :617-626)
 *   oi_index_* / oi_search*  <- NO reference interface exists: the
 *   oi_rrf_fuse / oi_merge_lists  reference has no retrieval port.  New, builder-defined
 *                               API styled after the reference's ports (borrowed inputs,
 *                               caller-owned outputs, DomainError-style failures).
 *
 * Conventions
 *   - Every function returns OI_OK (0) or a negative oi_status.  oi_last_error()
 *     returns a thread-local message for the last failure on the calling thread.
 *     A Rust shim maps nonzero -> DomainError::SourceFailure{name:"hip-analyzer",..}
 *     (src/domain/error.rs:16-17).
 *   - `location` arguments say where the caller's buffers live: OI_HOST or OI_DEVICE
 *     (HBM of the ctx's device).  With OI_DEVICE the call is asynchronous on the ctx
 *     stream (oi_set_stream / oi_synchronize); with OI_HOST it returns after the
 *     results are in the caller's buffers.
 *   - The library never keeps a caller pointer past return, except
 *     oi_index_set_embeddings(OI_DEVICE), which borrows the corpus matrix (30 GB at
 *     10M x 768 is not copied) until oi_index_destroy.
 *   - An oi_ctx serialises calls internally (one mutex): safe to share between host
 *     threads, as the reference's `Send + Sync` port requires.
 */
#ifndef OPENINTEL_HIP_H
#define OPENINTEL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OI_ABI_VERSION 1

typedef enum {
    OI_OK = 0,
    OI_ERR_INVALID_ARG = -1,
    OI_ERR_HIP = -2,           /* a HIP runtime call failed; message has the hipError */
    OI_ERR_ANALYZER_MISMATCH = -3, /* mirrors DomainError::AnalyzerMismatch, error.rs:13-14 */
    OI_ERR_STATE = -5,         /* call order violated (e.g. search before finalize) */
    OI_ERR_NO_DEVICE = -6,     /* no gfx950 device / HIP runtime unusable */
    OI_ERR_UNSUPPORTED = -7,   /* shape outside what the kernels are built for */
    OI_ERR_OVERFLOW = -8,      /* an internal candidate pool overflowed (bug guard) */
    OI_ERR_COMM = -9           /* an RCCL call failed; message has the ncclResult string */
} oi_status;

enum { OI_HOST = 0, OI_DEVICE = 1 };

/* Limits of the kernels in this build. */
#define OI_MAX_DEPTH 1024u      /* per-list depth k' and final k */
#define OI_MAX_DIM 1024u        /* embedding dimension (multiple of 4) */
#define OI_BM25_BLOCK_DOCS 32768u /* docs per BM25 doc block (one pool segment, one seen/multi map) */
#define OI_BM25_FINE_DOCS 16384u  /* docs per (term, window) cell of the inverted index: two windows per block */

typedef struct oi_ctx oi_ctx;
typedef struct oi_index oi_index;
typedef struct oi_comm oi_comm;

int oi_abi_version(void);
const char *oi_last_error(void);

int oi_create(int device_ordinal, oi_ctx **out);
/* A NEW context on `like`'s device with `like`'s settings (cosine mode, overlap of the legs, graph replay) and nothing
 * else of it: its own default stream, its own (still empty) workspaces.  For a host that scores several batches at once
 * through views of one index (oi_index_view): every lane gets a context of its own and the caller's is never rebound. */
int oi_create_like(oi_ctx *like, oi_ctx **out);
/* Lifetime: handles are reference-counted inside the library, so destruction is safe in ANY order.  oi_destroy gives up
 * the caller's handle (which must not be used again); if indexes created on -- or viewed through -- the ctx are still
 * alive, its stream, workspaces and mutex stay until the last of them is destroyed (those indexes remain fully usable
 * and from then on run on the device's default stream: oi_destroy forgets the stream given to oi_set_stream, which may
 * not outlive the caller).  Likewise oi_index_destroy of an index that still has views keeps its buffers until the
 * last view is destroyed.  (The reference shares its long-lived adapters the same way -- `Arc<...>` fields of the MCP
 * server, src/mcp/server.rs:17-20: the last owner frees.) */
void oi_destroy(oi_ctx *ctx);
/* `hip_stream` is a hipStream_t (NULL = the default stream). */
int oi_set_stream(oi_ctx *ctx, void *hip_stream);
int oi_synchronize(oi_ctx *ctx);

/* How the batch cosine scorer works over an f32 corpus (dim 384 / 768; more than 8 queries, or -- since round 5 -- any
 * number of queries when the index holds a screening copy; other shapes always use the exact kernels).  The corpus handed to the library stays f32 in HBM in every mode, and every score the
 * library returns is computed in f32 from those f32 rows.
 *   OI_COSINE_SCREEN (default) a bf16 screen with a proven error bound picks the rows that can reach the list
 *                              (one bf16 MFMA per product: HBM-bound; the bound is built from the MEASURED rounding
 *                              errors of the corpus and of each query), exact f32 scores are then computed for those
 *                              rows only; a query whose survivors do not fit falls back, inside the same call, to the
 *                              exact kernel.  The screen reads the index's bf16 SCREENING COPY of the rows when the
 *                              index holds one (oi_index_set_screen_copy: made at finalize when it fits the budget --
 *                              2 d bytes per row and batch) and converts the f32 rows on the fly otherwise (4 d bytes).
 *                              Same products, same bound, same survivors, same rescoring either way.
 *                              What comes out: the SET of rows of the exact scorer's list up to ties -- two rows whose
 *                              exact f32 scores differ by less than the f32 rounding of two summation orders (measured
 *                              <= 5e-7 at d = 768) may swap neighbouring ranks or, at the list's last rank, places;
 *                              every score within the 1e-5 bar (csrc/cosine_prefilter.hip has the argument).  RRF
 *                              output is bit-exact GIVEN the two lists.
 *   OI_COSINE_EXACT            v_mfma_f32_* on the f32 values for every row: matrix-pipe-bound.
 *   OI_COSINE_SPLIT            every f32 operand is split exactly into three bf16 values and the dot product
 *                              taken as six bf16 MFMAs with f32 accumulation (the three smallest of the nine
 *                              cross terms, <= 2^-23 relative, are dropped): f32-grade scores (measured error
 *                              vs f64 ~4e-8 on unit vectors; the exact kernel's f32 accumulation ~1e-7).
 *   OI_COSINE_SCREEN_COPY      OI_COSINE_SCREEN, and an index without a screening copy gets one on the first search
 *                              whatever its policy (rounds 2-4's opt-in mode; kept for callers that used it).
 *   OI_COSINE_SCREEN_STREAM    OI_COSINE_SCREEN with the screen ALWAYS converting the f32 rows on the fly, copy or not
 *                              (rounds 1-4's default; A/B runs and bench.py's `f32_stream_scorer`).
 * Also selectable with OI_COSINE_MODE=screen|exact|split|screen-copy|screen-stream at oi_create. */
#define OI_COSINE_EXACT 0
#define OI_COSINE_SPLIT 1
#define OI_COSINE_SCREEN 2
#define OI_COSINE_SCREEN_COPY 3
#define OI_COSINE_SCREEN_STREAM 4
int oi_set_cosine_mode(oi_ctx *ctx, int mode);

/* A hybrid query has two independent legs until fusion.  By default the BM25 leg is issued on an
 * internal side stream (forked from and joined back into the ctx stream inside the call) so that it
 * runs beside the MFMA-bound cosine leg; enable = 0 runs them one after the other. */
int oi_set_overlap(oi_ctx *ctx, int enable);

/* Speculative thresholds of the screen (round 5; default on).  Between the corpus chunks of a screened search the threshold is the
 * proven one -- the k'-th best screen score of the rows seen so far, minus the margin -- which is weak while few rows have been seen.
 * With speculation the next chunk is screened against the larger of it and a PREDICTION of the final threshold (the r-th best score
 * so far, r = 3 k' m / n + 12 after m of n rows), and the prediction is checked at the end against the proven final threshold: if it
 * holds, the survivors are a superset of the proven screen's and the lists are the same; if not (a corpus whose first rows are not a
 * fair sample of it), the exact pipeline rescores the batch inside the same call -- the same lists, later -- and the ctx stops
 * speculating for 16, 32, ... 1024 searches.  enable = 0: proven thresholds only (rounds 2-4).  Never used with graph replay. */
int oi_set_screen_speculation(oi_ctx *ctx, int enable);

/* One device-buffer query call is ~30 kernel launches, memsets and event operations: ~0.3 ms of host time, which at a
 * 1.25M-row shard (one of 8 GPUs) is what limits the rate, not the GPU (0.7 ms of work that two batches in flight overlap).
 * enable != 0: oi_search_lists_packed / oi_fuse_packed / oi_search calls with OI_DEVICE buffers are CAPTURED into a
 * hipGraph the second time they are made with the same arguments (index, pointers, sizes, modes) on this ctx and
 * replayed with one launch call from then on.  The launch sequence does not depend on the data (chunk schedules and
 * pool capacities are functions of the sizes alone), so a replay does exactly what the call would have done: same
 * kernels, same buffers, same results.  The caller's part: keep using the SAME buffers (a serving loop copies each batch
 * into per-slot staging buffers -- openintel_amd/sharded.py does).  A call with new arguments runs eagerly once more;
 * profiling (oi_profile_reset) and the default stream disable replay; at most 32 captured calls per ctx are kept.
 * Default off.  (No reference counterpart: the reference makes no device calls, src/adapters/analyzer/lexicon.rs:82-87.) */
int oi_set_graph_replay(oi_ctx *ctx, int enable);

/* ------------------------------------------------------------------------- */
/* PostAnalyzer path (reference-pinned)                                        */
/* ------------------------------------------------------------------------- */

/*
 * One (polarity, speculative) per post, index-aligned with the input
 * (post_analyzer.rs:9).  Post i is text_blob[offsets[i] .. offsets[i+1]), valid
 * UTF-8 (the shim gathers SocialPost.text into one blob: posts are not contiguous
 * in the reference, social_post.rs:25-27).  Infallible in the reference; here it
 * fails only on device errors.
 */
int oi_lexicon_analyze(oi_ctx *ctx, const uint8_t *text_blob, const uint64_t *offsets,
                       uint64_t n_posts, double *polarity_out, uint8_t *speculative_out);

/* Same, buffers already in HBM; asynchronous on the ctx stream. */
int oi_lexicon_analyze_device(oi_ctx *ctx, const uint8_t *d_text_blob, const uint64_t *d_offsets,
                              uint64_t n_posts, uint64_t blob_bytes, double *d_polarity_out,
                              uint8_t *d_speculative_out);

/* Raw sums of SpeculationEngine::social_summary (speculation_engine.rs:76-97).
 * Integer fields are exact.  polarity_sum is a fixed-shape tree sum (bitwise
 * reproducible run to run) and differs from the reference's input-order sum by
 * at most n * 2^-52 * max|partial sum| (the bound tests/ assert). */
typedef struct {
    uint64_t total;
    uint64_t by_source[2]; /* [reddit, bluesky]  source_kind.rs:5-8 */
    uint64_t bullish, bearish, neutral, spec_count;
    double polarity_sum;
} oi_social_counters;

/* sources[i]: 0 = reddit, 1 = bluesky.  n_posts != n_signals ->
 * OI_ERR_ANALYZER_MISMATCH (speculation_engine.rs:29-34). */
int oi_social_summary(oi_ctx *ctx, const uint8_t *sources, uint64_t n_posts,
                      const double *polarity, const uint8_t *speculative, uint64_t n_signals,
                      double bull_bear_threshold, int location, oi_social_counters *out_host);

/* LexiconAnalyzer::analyze and the two loops of social_summary in ONE pass over the text (lexicon.rs:53-87 +
 * speculation_engine.rs:76-97): the per-post signals are reduced where they are computed, so with
 * d_polarity_out == d_speculative_out == NULL nothing per post is written at all ("0 out if fused
 * with the A4 reduction").  Buffers in HBM; d_sources (0 = reddit, 1 = bluesky) may be NULL; the outputs, when
 * given, are exactly oi_lexicon_analyze_device's.  Integer fields exact; polarity_sum a fixed-shape tree sum
 * (bitwise reproducible; same bound as oi_social_summary's).  Synchronises the ctx stream (the sums come back). */
int oi_lexicon_summary_device(oi_ctx *ctx, const uint8_t *d_text_blob, const uint64_t *d_offsets, uint64_t n_posts,
                              uint64_t blob_bytes, const uint8_t *d_sources, double bull_bear_threshold,
                              double *d_polarity_out, uint8_t *d_speculative_out, oi_social_counters *out_host);

/* The batch callers: run_scan (src/mcp/tools.rs:193-225) and run_compare (:303-352) run application::analyze once per
 * ticker.  Here the posts of ALL tickers are pooled into one batch -- one scan -- and every ticker gets its
 * social_summary sums from one reduction: segment s = posts [seg_offsets[s], seg_offsets[s+1]) (non-decreasing,
 * seg_offsets[n_segments] <= n_posts; an empty segment gives zeros), out[s] = that ticker's counters.
 * polarity_sum is the reference's own loop (speculation_engine.rs:82-86): added one signal at a time, in input order, from
 * +0.0 -- BIT-IDENTICAL to the reference's f64 (and so is net_sentiment = polarity_sum / total), unlike the tree of
 * oi_social_summary.  One wave per segment: meant for many segments of a ticker's size (50 posts per source,
 * tools.rs:101); a single segment of 10M posts takes ~50 ms.
 * location OI_HOST: every pointer is host memory, the call returns with `out` filled; seg_offsets[n_segments] > n_posts
 * -> OI_ERR_ANALYZER_MISMATCH (speculation_engine.rs:29-34), decreasing offsets -> OI_ERR_INVALID_ARG.
 * location OI_DEVICE: every pointer (out too: n_segments records) is HBM, asynchronous on the ctx stream; offsets past
 * n_posts are clamped by the kernel.  sources may be NULL (by_source stays 0). */
int oi_social_summary_segmented(oi_ctx *ctx, const uint8_t *sources, const double *polarity, const uint8_t *speculative,
                                uint64_t n_posts, const uint64_t *seg_offsets, uint64_t n_segments,
                                double bull_bear_threshold, int location, oi_social_counters *out);

/* oi_lexicon_analyze_device + oi_social_summary_segmented(OI_DEVICE) as one call on the ctx stream: the pooled posts of
 * n_segments tickers in, one record per ticker out (d_out, HBM).  d_polarity_out / d_speculative_out may be NULL (the
 * signals then live in the ctx's workspace only).  Asynchronous. */
int oi_lexicon_scan_segments_device(oi_ctx *ctx, const uint8_t *d_text_blob, const uint64_t *d_offsets, uint64_t n_posts,
                                    uint64_t blob_bytes, const uint8_t *d_sources, const uint64_t *d_seg_offsets,
                                    uint64_t n_segments, double bull_bear_threshold, double *d_polarity_out,
                                    uint8_t *d_speculative_out, oi_social_counters *d_out);

/* ------------------------------------------------------------------------- */
/* Headline gate (src/domain/dip.rs:204-272)                                   */
/* ------------------------------------------------------------------------- */

/* CATALYST_KEYWORDS (dip.rs:38-55) in declaration order; NULL past the end. */
#define OI_N_CATALYST_KEYWORDS 16
const char *oi_catalyst_keyword(uint32_t index);

/*
 * For each title i = blob[offsets[i] .. offsets[i+1]) (UTF-8, offsets[0] == 0):
 *   mask_out[i]   bit k set iff keyword k occurs as a whole word        catalyst_hits(&[title])
 *   order_out[i]  nibble j = keyword index of the j-th distinct hit,    (dip.rs:261-272; the
 *                 first-occurrence order of the reference's Vec          multi-text call is the
 *                                                                        deduped concatenation)
 *   about_out[i]  headline_mentions_company(title, ticker, name_forms)  dip.rs:247-258
 * ticker is the raw symbol (ticker_len bytes, compared ASCII-lowercased, ignored when
 * shorter than 2 bytes -- dip.rs:250); form i is forms_blob[form_offsets[i] ..
 * form_offsets[i+1]) exactly as company_name_forms returned it (n_forms may be 0; the
 * caller keeps the reference's `name_forms.is_empty() ||` short-circuit, dip.rs:624).
 * At most 32 usable patterns / 1024 pattern bytes per call -> OI_ERR_INVALID_ARG beyond.
 */
int oi_headline_scan(oi_ctx *ctx, const uint8_t *blob, const uint64_t *offsets, uint64_t n_titles,
                     const uint8_t *ticker, uint64_t ticker_len, const uint8_t *forms_blob,
                     const uint32_t *form_offsets, uint32_t n_forms, uint16_t *mask_out,
                     uint64_t *order_out, uint8_t *about_out);

/* Same, titles and outputs already in HBM (blob 16-byte aligned, each title < 4 GiB);
 * ticker/forms stay host pointers.  Asynchronous on the ctx stream. */
int oi_headline_scan_device(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets,
                            uint64_t n_titles, uint64_t blob_bytes, const uint8_t *ticker,
                            uint64_t ticker_len, const uint8_t *forms_blob,
                            const uint32_t *form_offsets, uint32_t n_forms, uint16_t *d_mask_out,
                            uint64_t *d_order_out, uint8_t *d_about_out);

/* The gate over the ROWS of a dip scan (src/application/dip.rs: one `check` per loser -> domain/dip.rs:612-659 on that
 * row's headlines, ticker and name forms).  Row r owns titles [row_offsets[r], row_offsets[r+1]) (row_offsets[0] = 0,
 * row_offsets[n_rows] = n_titles), ticker tickers_blob[ticker_offsets[r] .. ticker_offsets[r+1]) and name forms
 * row_form_offsets[r] .. row_form_offsets[r+1] (indices into form_offsets, which holds total_forms + 1 byte offsets into
 * forms_blob).  Outputs as oi_headline_scan, index-aligned with the titles.  Host buffers; one staging copy, one launch
 * per row back to back, one copy back -- a 25-row scan costs about what four single-row calls do.  At most 4096 rows. */
int oi_headline_scan_rows(oi_ctx *ctx, const uint8_t *blob, const uint64_t *offsets, uint64_t n_titles,
                          const uint64_t *row_offsets, uint32_t n_rows, const uint8_t *tickers_blob,
                          const uint32_t *ticker_offsets, const uint8_t *forms_blob, const uint32_t *form_offsets,
                          const uint32_t *row_form_offsets, uint16_t *mask_out, uint64_t *order_out, uint8_t *about_out);

/* ------------------------------------------------------------------------- */
/* Hybrid retrieval (builder-defined; parity unpinned vs the reference)        */
/* ------------------------------------------------------------------------- */

/* A shard of the corpus: rows [doc_id_base, doc_id_base + n_docs) of the global
 * collection.  Result doc ids are global (doc_id_base + local row). */
int oi_index_create(oi_ctx *ctx, uint64_t n_docs, uint32_t dim, uint32_t vocab,
                    uint32_t doc_id_base, oi_index **out);
void oi_index_destroy(oi_index *idx);

/* A second handle on a FINALIZED index, bound to another context of the same device (its own stream and workspaces), so
 * that two searches over the same shard can be in flight at once -- e.g. batch i+1 scored while the selects and the
 * rescoring of batch i drain.  The view borrows every buffer of `src`: it is read-only (the set_* /
 * finalize calls return OI_ERR_STATE), costs no HBM for the index (its ctx allocates its own search workspaces on
 * first use, see oi_search_lists), may be destroyed before or after `src` (the buffers live until the last handle on them
 * is gone), and is searched with the ordinary
 * oi_search* calls.  Thread-safety is per context as everywhere else: the two handles may be driven from two host threads.
 * (No reference counterpart: the reference's port is synchronous, src/domain/ports/post_analyzer.rs:7-11 is its model.) */
int oi_index_view(oi_index *src, oi_ctx *ctx, oi_index **out);

/* rows: n_docs x dim f32, row-major.  normalize != 0: L2-normalise each row (in
 * place when OI_DEVICE).  OI_DEVICE borrows the pointer; OI_HOST copies to HBM. */
int oi_index_set_embeddings(oi_index *idx, float *rows, int location, int normalize);

/* A bf16 corpus instead (BASELINE configs[4]): rows n_docs x dim of bfloat16 bit patterns, row-major,
 * unit-norm as stored (no normalisation is applied), dim in {384, 768, 1024}.  OI_DEVICE borrows the
 * pointer (16-byte aligned); OI_HOST copies to HBM.  Queries stay f32 at the API and are rounded to
 * bf16 (nearest even) inside: a score is sum_k bf16(q_k) * x_k accumulated in f32.  Replaces any f32
 * matrix set before, and vice versa. */
int oi_index_set_embeddings_bf16(oi_index *idx, const uint16_t *rows, int location);

/* Forward index: doc d owns term_ids[doc_offsets[d] .. doc_offsets[d+1]),
 * every id < vocab.  Stages the postings and computes local statistics. */
int oi_index_set_forward(oi_index *idx, const uint32_t *term_ids, const uint64_t *doc_offsets,
                         int location);
/* Local document frequencies (vocab entries, host) and token count: what a
 * multi-shard caller all-reduces before oi_index_finalize. */
int oi_index_local_stats(oi_index *idx, uint64_t *total_tokens_out, uint32_t *df_out_host);
/* Fix the collection statistics (global N, token count, df; df NULL = local) and
 * build the blocked inverted index with precomputed BM25 impacts. */
int oi_index_finalize(oi_index *idx, uint64_t global_n_docs, uint64_t global_total_tokens,
                      const uint32_t *global_df_host);

/* How many rows the screened cosine scorer sets aside (diagnostics, tests).  When the largest row norms of an f32 corpus
 * stand out from the rest (> 1.5 x the RMS norm, or the same for the norm of bf16(x) - x) and the rows responsible are few
 * (<= 1024), they are left out of the screen's thresholds and rescored exactly for every query; the screen's margin is then
 * built from the other rows' maxima.  0: one class (a normalised corpus, or too many such rows).  Builder-defined like the
 * whole retrieval path (the reference has no embeddings: SURVEY.md section 0). */
int oi_index_long_rows(oi_index *idx, uint32_t *n_out);

/* The bf16 SCREENING COPY of an f32 corpus: a derived index structure like the BM25 impact postings -- bf16(x) of every
 * row, made by the library with the screen's own conversion (so the measured error bound of OI_COSINE_SCREEN is the copy's),
 * n_docs x dim x 2 bytes of HBM on top of the f32 rows.  The screen streams it instead of the f32 rows (half the bytes of
 * an HBM-bound kernel); survivors are still rescored in f32 from the f32 rows, so the lists do not change.
 *   OI_SCREEN_COPY_AUTO (default)  made by oi_index_finalize / oi_index_finalize_sharded (and again by
 *                                  oi_index_set_embeddings on a finalized index) when the corpus can be screened (dim 384 /
 *                                  768, finite norms) and the copy is at most a quarter of the device memory that is FREE at
 *                                  that moment (OI_SCREEN_COPY_MAX_FRAC=0.25 overrides the fraction); otherwise no copy:
 *                                  the screen converts the f32 rows on the fly.
 *   OI_SCREEN_COPY_NEVER           no copy (an existing one is freed).
 *   OI_SCREEN_COPY_ALWAYS          made whatever the budget (OI_ERR_HIP if the allocation fails).
 * Call before finalize (or any time: a finalized index applies the policy at once).  OI_SCREEN_COPY=auto|never|always sets
 * the process-wide default.  Views (oi_index_view) borrow the source's copy.  No reference counterpart (the reference has no
 * embeddings: SURVEY.md section 0). */
#define OI_SCREEN_COPY_AUTO 0
#define OI_SCREEN_COPY_NEVER 1
#define OI_SCREEN_COPY_ALWAYS 2
int oi_index_set_screen_copy(oi_index *idx, int policy);
/* HBM the index itself holds right now (oi_workspace_bytes reports the searching contexts' workspaces): the embedding
 * rows as handed over (0 when the caller's device pointer is borrowed), the screening copy, and the BM25 structures
 * (postings, cell table, idf / floors, doc lengths, the kept forward index).  Any output may be NULL. */
int oi_index_bytes(oi_index *idx, uint64_t *rows_owned_bytes_out, uint64_t *screen_copy_bytes_out, uint64_t *bm25_bytes_out);

/* BM25 kernel choice: 0 = default (= 4), 1 = term-at-a-time with one workgroup per doc block (bm25.hip, the
 * first-generation kernel), 2 = batch scan of the forward index (bm25_scan.hip), 3 = term-at-a-time with one wave
 * per (doc block, query) task (bm25_wave.hip), 4 = term-at-a-time as a stream: every wave walks a weight-balanced
 * range of (query, block) tasks with the postings ahead of it in flight through an LDS ring (bm25_stream.hip).
 * All produce bit-identical lists.  OI_BM25_MODE=stream|wave|taat|scan selects the default process-wide. */
int oi_index_set_bm25_mode(oi_index *idx, int mode);

/* Contract for the batch BM25 scan (mode 2): no query of a batch has more
 * than `max_terms` terms (default 16).  The scan handles 1024 / max_terms queries per pass over the
 * forward index; a longer query sets the ctx's error flag (OI_ERR_OVERFLOW at the next host-visible
 * point).  The term-at-a-time path (smaller batches, or OI_BM25_MODE=taat) has no such limit. */
int oi_index_set_max_query_terms(oi_index *idx, uint32_t max_terms);

/*
 * Per-shard ranked lists for a batch of queries.
 *   query_vecs      n_queries x dim f32 (normalised by the caller if cosine is wanted)
 *   query_terms     concatenated term ids; query q owns [q_term_offsets[q], q_term_offsets[q+1])
 * Outputs, each n_queries x depth, row q sorted by (score desc, doc id asc),
 * first counts[q] entries valid:
 *   cosine list: dot(query, row);  BM25 list: only docs with score > 0.
 */
int oi_search_lists(oi_index *idx, const float *query_vecs, const uint32_t *query_terms,
                    const uint32_t *q_term_offsets, uint32_t n_queries, uint32_t depth,
                    int location, float *cos_scores, uint32_t *cos_docs, uint32_t *cos_counts,
                    float *bm25_scores, uint32_t *bm25_docs, uint32_t *bm25_counts);

/* Merge per-shard lists ([n_shards][n_queries][depth], counts [n_shards][n_queries])
 * into the global top-`depth` per query (the step after the RCCL all-gather). */
int oi_merge_lists(oi_ctx *ctx, const float *scores, const uint32_t *docs, const uint32_t *counts,
                   uint32_t n_shards, uint32_t n_queries, uint32_t depth, int location,
                   float *scores_out, uint32_t *docs_out, uint32_t *counts_out);

/*
 * The multi-GPU exchange format.  A shard's two lists for a batch, packed in ONE buffer of
 * OI_PACKED_WORDS(n_queries, depth) 32-bit words:
 *     float    scores[2][n_queries][depth]     list 0 = cosine, list 1 = BM25
 *     uint32_t docs  [2][n_queries][depth]
 *     uint32_t counts[2][n_queries]
 * oi_search_lists_packed fills it (same contents as oi_search_lists); every rank all-gathers the
 * buffers (RCCL) into [n_shards][OI_PACKED_WORDS] and calls oi_fuse_packed, which merges each list to
 * its global top-`depth` and THEN fuses (global ranks are needed: fusing per shard is not equivalent).
 */
#define OI_PACKED_WORDS(n_queries, depth) (4ull * (n_queries) * (depth) + 2ull * (n_queries))
int oi_search_lists_packed(oi_index *idx, const float *query_vecs, const uint32_t *query_terms,
                           const uint32_t *q_term_offsets, uint32_t n_queries, uint32_t depth,
                           int location, uint32_t *packed_out);
int oi_fuse_packed(oi_ctx *ctx, const uint32_t *packed_all, uint32_t n_shards, uint32_t n_queries,
                   uint32_t depth, uint32_t k, int location, float *scores_out, uint32_t *docs_out,
                   uint32_t *counts_out);

/* Reciprocal-rank fusion of two ranked lists per query (row stride `depth`):
 * rrf(d) = sum over lists containing d of 1/(60 + rank), rank from 1; output top-k
 * by (rrf desc, doc id asc), row stride k. */
int oi_rrf_fuse(oi_ctx *ctx, const uint32_t *docs_a, const uint32_t *counts_a,
                const uint32_t *docs_b, const uint32_t *counts_b, uint32_t n_queries,
                uint32_t depth, uint32_t k, int location, float *scores_out, uint32_t *docs_out,
                uint32_t *counts_out);

/* oi_search_lists + oi_rrf_fuse on one shard: the whole hybrid query. */
int oi_search(oi_index *idx, const float *query_vecs, const uint32_t *query_terms,
              const uint32_t *q_term_offsets, uint32_t n_queries, uint32_t depth, uint32_t k,
              int location, float *scores_out, uint32_t *docs_out, uint32_t *counts_out);

/* ------------------------------------------------------------------------- */
/* The row-sharded query behind the C ABI: one process per GPU, RCCL inside    */
/* ------------------------------------------------------------------------- */
/*
 * A host with no collective library of its own (the Rust composition root, src/main.rs:17-39, has none -- the reference
 * is single-process, src/domain/ports/mod.rs:1-8) gets the whole multi-GPU path from three calls.  RCCL is loaded on
 * first use (dlopen "librccl.so.1", or the library OI_RCCL_LIB names; OI_ERR_UNSUPPORTED with the loader's message if
 * it cannot be loaded -- tests/test_abi.py takes that path).  A Python host that already runs torch.distributed
 * can keep using oi_search_lists_packed + its own all-gather + oi_fuse_packed (openintel_amd/sharded.py): same results.
 *
 *   rank 0:  oi_comm_unique_id(id)  -> ship the OI_COMM_ID_BYTES to every rank over the host's own channel (file, socket, env)
 *   all:     oi_comm_create(ctx, id, rank, world, &comm)          collective: returns when every rank has called it
 *            ... oi_index_create / set_embeddings / set_forward on the rank's row shard (doc_id_base = first global row) ...
 *            oi_index_finalize_sharded(idx, comm)                  collective: all-reduce of (n_docs, tokens) and of the
 *                                                                  df vector, then the impacts from the GLOBAL statistics
 *            oi_search_sharded(idx, comm, queries..., out...)      collective, once per batch, same queries on every rank:
 *                                                                  the shard's two lists -> ONE ncclAllGather of
 *                                                                  OI_PACKED_WORDS words per rank -> merge to the global
 *                                                                  top-depth per list -> RRF.  Identical output on every rank.
 * Collectives run on the ctx stream in call order: every rank must issue them in the same order (one host thread per
 * comm, or external ordering).  With OI_DEVICE buffers oi_search_sharded is asynchronous like oi_search.
 * Executed so far with a communicator of ONE rank only (tests/test_gpu_native_comm.py): the build's GPU lease has one
 * device and RCCL refuses two ranks on one device.  The same choreography over torch.distributed (sharded.py) has run
 * with 2 and 8 gloo ranks; N > 1 through THESE entry points is unexecuted code.
 */
#define OI_COMM_ID_BYTES 128
int oi_comm_unique_id(uint8_t id_out[OI_COMM_ID_BYTES]);
int oi_comm_create(oi_ctx *ctx, const uint8_t id[OI_COMM_ID_BYTES], uint32_t rank, uint32_t world, oi_comm **out);
void oi_comm_destroy(oi_comm *comm);
int oi_index_finalize_sharded(oi_index *idx, oi_comm *comm);
int oi_search_sharded(oi_index *idx, oi_comm *comm, const float *query_vecs, const uint32_t *query_terms,
                      const uint32_t *q_term_offsets, uint32_t n_queries, uint32_t depth, uint32_t k, int location,
                      float *scores_out, uint32_t *docs_out, uint32_t *counts_out);

/* ------------------------------------------------------------------------- */
/* The pipelined query behind the C ABI: several batches in flight             */
/* ------------------------------------------------------------------------- */
/*
 * Throughput mode of oi_search / oi_search_sharded for a host without streams or a collective library of its own (the
 * reference's port is one synchronous call -- src/domain/ports/post_analyzer.rs:7-11, `Send + Sync`, borrowed in / owned
 * out; composition root src/main.rs:17-39 -- so this has no reference counterpart).  Batches are independent: batch n is
 * scored on lane n % lanes (every lane = a context, a stream and a view of the index inside the library), its exchange
 * (ONE RCCL all-gather when `comm` is given, none otherwise) and its fusion run on a further stream, and the next batch's
 * corpus stream overlaps the selects / rescoring / fusion of the previous one.  Per batch: the same kernels on the same
 * data as oi_search (comm == NULL) / oi_search_sharded -- bit-identical results (tests/test_gpu_pipeline.py).
 *
 *   oi_pipeline_create(idx, comm_or_NULL, lanes, max_queries, max_query_terms, depth, k, &p)
 *        idx finalized, not a view; lanes in [1,4] (2 is the measured optimum on one MI355X); every later batch has at most
 *        max_queries queries of at most max_query_terms terms each (sizes of the staging buffers).  HBM: one set of search
 *        workspaces per lane (oi_pipeline_workspace_bytes).
 *   oi_pipeline_submit(p, queries..., n_queries, location, scores_out, docs_out, counts_out, &ticket)
 *        asynchronous in BOTH locations; tickets count from 1.
 *        OI_DEVICE: inputs were produced on the index's context stream (oi_set_stream) and must stay unmodified until the
 *                   batch is waited for; outputs (n_queries x k, k as created) are written by the pipeline's streams.
 *        OI_HOST:   inputs are copied during the call (reusable at return); the host output arrays are filled by
 *                   oi_pipeline_wait(ticket) -- or, if nobody waits, when the slot is reused 2 x lanes (>= 4) submits later
 *                   or at drain -- and must stay valid until then.
 *   oi_pipeline_wait(p, ticket, host_sync)
 *        host_sync != 0: returns when that batch's outputs are complete (host outputs delivered).
 *        host_sync == 0 (OI_DEVICE batches): no host stall -- orders the index's context stream after the batch, so work
 *                   queued there afterwards may read the outputs.
 *   oi_pipeline_drain(p)      everything submitted is complete; a candidate-pool overflow in any lane is reported here.
 *   oi_pipeline_destroy(p)    drains, then frees the lanes (before the index and the communicator).
 * With a communicator every rank must submit the same batches in the same order from ONE host thread (the collectives
 * are issued in submission order on one stream), as with oi_search_sharded.  One submitting thread per pipeline; wait /
 * drain may be called from another.
 */
typedef struct oi_pipeline oi_pipeline;
int oi_pipeline_create(oi_index *idx, oi_comm *comm, uint32_t lanes, uint32_t max_queries, uint32_t max_query_terms,
                       uint32_t depth, uint32_t k, oi_pipeline **out);
void oi_pipeline_destroy(oi_pipeline *p);
int oi_pipeline_submit(oi_pipeline *p, const float *query_vecs, const uint32_t *query_terms, const uint32_t *q_term_offsets,
                       uint32_t n_queries, int location, float *scores_out, uint32_t *docs_out, uint32_t *counts_out,
                       uint64_t *ticket_out);
int oi_pipeline_wait(oi_pipeline *p, uint64_t ticket, int host_sync);
int oi_pipeline_drain(oi_pipeline *p);
int oi_pipeline_workspace_bytes(oi_pipeline *p, uint64_t *device_bytes_out, uint64_t *pinned_host_bytes_out);
/* oi_profile_reset / oi_profile_read (below) over the pipeline's lanes, summed (the lanes' contexts are the library's own).
 * Two lanes' launches overlap in time: the summed durations then exceed the wall time they covered. */
/* oi_pipeline_create MEASURES which of its streams run at the same time (HIP maps streams onto a few hardware queues, and
 * which queue a new stream gets depends on every stream the process already has: two lanes on one queue serialise) and
 * keeps lanes + 1 that do, drawing up to 12 candidates.  This reports how many of the lanes + 1 were pairwise concurrent. */
int oi_pipeline_concurrent_streams(oi_pipeline *p, uint32_t *concurrent_out, uint32_t *streams_out);
int oi_pipeline_profile_reset(oi_pipeline *p, int enable);
int oi_pipeline_profile_read(oi_pipeline *p, const char *kernel_tag, double *total_ms_out, uint64_t *launches_out);

/* Diagnostics of the OI_COSINE_SCREEN mode (tests/test_gpu_prefilter.py; not on the query path).  For host
 * queries [n_queries][dim] against rows [row_begin, row_begin + n_rows) of an f32 index:
 *   screen_scores_out[q * n_rows + r]  the screen's raw score s~ (bf16 operands, the screen's own conversion and
 *                                      matrix instruction);  NULL = skip
 *   eps_out[q]                         the proven bound of that query: |s~ - s| <= eps for EVERY row of the index
 *                                      (infinite when the query has no bound and takes the exact kernel)
 * Both outputs are host buffers; the call is synchronous. */
int oi_screen_probe(oi_index *idx, const float *query_vecs, uint32_t n_queries, uint64_t row_begin,
                    uint32_t n_rows, float *screen_scores_out, float *eps_out);

/* Timing hooks for bench.py: when enabled, HIP events are recorded on the ctx stream
 * around every kernel launch, grouped by tag ("cosine", "bm25", "select", "rrf",
 * "lexicon", "social_summary").  oi_profile_read returns the summed duration (ms) of
 * the launches with that tag and their count since the last reset.  enable: 0 off, 1 every
 * tagged launch, 2 only the "cosine" launches (two event packets per launch cost a few us of
 * stream time each: a timed region that only needs its dominant kernel asks for 2). */
int oi_profile_reset(oi_ctx *ctx, int enable);
int oi_profile_read(oi_ctx *ctx, const char *kernel_tag, double *total_ms_out, uint64_t *launches_out);

/* Memory this context holds right now: HBM of its workspaces (candidate pools, staging, lists -- INTEGRATION.md 5b: the
 * BM25 pool of the wave kernel alone reserves ~5 GB per searching context at 10M docs) and page-locked host memory of
 * its staging buffers.  Index buffers are not counted (oi_index owns them; a view borrows them).  Either output may be
 * NULL.  (No reference counterpart: the reference allocates nothing on a device.) */
int oi_workspace_bytes(oi_ctx *ctx, uint64_t *device_bytes_out, uint64_t *pinned_host_bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* OPENINTEL_HIP_H */
