// oi_internal.h -- shared host-side plumbing of libopenintel_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "openintel_hip.h"

// ---------------------------------------------------------------- errors
void oi_set_error(const char *fmt, ...);

#define OI_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t oi_e_ = (expr);                                                           \
        if (oi_e_ != hipSuccess) {                                                           \
            oi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(oi_e_), __FILE__, \
                         __LINE__);                                                          \
            return OI_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define OI_CHECK(expr)              \
    do {                            \
        int oi_rc_ = (expr);        \
        if (oi_rc_ != OI_OK) return oi_rc_; \
    } while (0)

#define OI_REQUIRE(cond, ...)         \
    do {                              \
        if (!(cond)) {                \
            oi_set_error(__VA_ARGS__); \
            return OI_ERR_INVALID_ARG; \
        }                             \
    } while (0)

// ---------------------------------------------------------------- process-wide one-shots and build switches
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: set once per (kernel, device),
// under a process-wide mutex -- contexts on different devices / host threads all pass through here (api.hip).
struct oi_ctx;
int oi_dyn_lds(oi_ctx *ctx, const void *kernel, size_t bytes);

// A/B switches, ablation builds ("timings only, results wrong by construction") and the first-generation kernels are
// reachable through environment variables ONLY in a -DOI_ABLATION build (tools/*.sh, tools/ks_ablate.py build one
// with OI_EXTRA_HIPCC_FLAGS=-DOI_ABLATION).  The product build ignores them: a stray variable cannot change what
// the C ABI returns.  (OI_COSINE_MODE / OI_BM25_MODE select between documented, tested, equivalent modes and stay.)
#ifdef OI_ABLATION
inline const char *oi_ablation_env(const char *name) { return getenv(name); }
#else
inline const char *oi_ablation_env(const char *) { return nullptr; }
#endif

// ---------------------------------------------------------------- buffers
// Grow-only device buffer: the hot path never calls hipMalloc once warmed up.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool borrowed = false; // a view's alias of another index's buffer (oi_index_view): never freed, never resized here
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int ensure(size_t bytes);
    void release();
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Page-locked host staging of a context (the OI_HOST entry points of the query path): the caller's pageable arrays are
// packed into it and moved with ONE DMA each way -- three pageable copies each way are staged by the runtime one after the
// other.  Every OI_HOST call ends with a stream synchronise inside the ctx mutex, so a buffer is free again at the next call.
#define OI_PINNED_STAGE_MAX ((size_t)1 << 20) // calls moving more than this use the caller's pageable arrays directly (measured: no gain above ~1 MB)
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool pinned = false; // false: hipHostMalloc refused, plain memory stands in (slower copies, same results)
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    int ensure(size_t bytes);
    void release();
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct ProfSpan {
    hipEvent_t a, b;
};

// Bumped whenever any DevBuf allocates, grows or is released: a captured launch sequence (below) holds raw workspace
// pointers and is only replayed while no workspace has moved since it was captured.
extern std::atomic<uint64_t> g_oi_ws_epoch;

// One device-buffer call of the query path (oi_search_lists_packed, oi_fuse_packed, oi_search: ~30 launches, memsets and
// event operations; 0.3 ms of host time at a 1.25M-row shard, where the GPU needs 0.7) captured into a hipGraph the second
// time it is made with the same arguments, and replayed with ONE launch call from then on (oi_set_graph_replay).
struct GraphEntry {
    uint64_t key[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t epoch = 0, last_use = 0;
    int state = 0; // 0 = seen once (ran eagerly: every workspace exists now), 1 = captured, 2 = not capturable (stays eager)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

struct oi_ctx {
    // Lifetime: the caller's handle holds one reference and every oi_index created on / viewed through the ctx one more;
    // oi_destroy drops the caller's, the teardown runs when the last one goes (api.hip: ctx_release).  An index can
    // therefore always be destroyed, whatever became of its ctx handle.
    std::atomic<int> refs{1};
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;     // the BM25 leg of a hybrid query runs here, beside the cosine leg (made by the first hybrid search)
    bool side_stream_failed = false;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap_legs = true;              // oi_set_overlap
    // speculative screen thresholds (cosine_prefilter.hip, pf_spec_kernel): on by default (oi_set_screen_speculation); a batch whose
    // speculation failed its check (the device writes *spec_fail_host) switches them off for spec_backoff searches, doubling up to 1024
    bool speculate = true;
    uint32_t spec_skip = 0, spec_backoff = 0;
    uint32_t *spec_fail_host = nullptr;    // 4 bytes of pinned host memory, written by pf_rescore_kernel when a check fails
    uint64_t spec_searches = 0, spec_failures = 0; // (diagnostics: profile "spec_state")
    int cosine_mode = 2;                   // oi_set_cosine_mode: 0 exact-f32 MFMA, 1 split-precision products, 2 screen + rescore (default),
                                           // 3 = 2 + make a missing screening copy on first use, 4 = 2 but never read the copy
    std::mutex mu;
    std::map<std::string, DevBuf> ws; // named workspaces
    int prof_enabled = 0; // 0 off, 1 every tagged launch, 2 the cosine scorer only
    // set by api.hip around the gated exact pipeline that follows a bf16 screen (cosine_prefilter.hip): the exact
    // cosine kernel and the selects launched meanwhile exit at once unless *run_gate is nonzero
    const uint32_t *run_gate = nullptr;
    const uint32_t *last_screen_gate = nullptr; // the gate word of the last screened search (diagnostics)
    std::map<std::string, std::vector<ProfSpan>> prof;
    std::vector<hipEvent_t> event_pool;
    bool use_graphs = false;          // oi_set_graph_replay
    std::vector<GraphEntry> graphs;   // <= OI_MAX_GRAPHS, least recently used evicted
    uint64_t graph_clock = 0, graph_replays = 0, graph_captures = 0;

    PinBuf pin_in, pin_out;

    DevBuf &buf(const char *name) { return ws[name]; }
    void prof_begin(const char *tag);
    void prof_end(const char *tag);
};

struct ProfScope {
    oi_ctx *c;
    const char *tag;
    bool on;
    ProfScope(oi_ctx *ctx, const char *t) : c(ctx), tag(t) {
        on = c->prof_enabled == 1 || (c->prof_enabled == 2 && strcmp(t, "cosine") == 0);
        if (on) c->prof_begin(tag);
    }
    ~ProfScope() {
        if (on) c->prof_end(tag);
    }
};

#define OI_BM25_FLOOR_RANKS 4
// ---------------------------------------------------------------- index
struct oi_index {
    uint64_t uid = 0;         // process-unique (graph cache keys: an address can be reused, a uid cannot)
    std::atomic<int> refs{1}; // the caller's handle + one per live view (a view borrows this index's buffers)
    oi_index *src = nullptr;  // a view: the index whose buffers it borrows (holds a reference on it)
    oi_ctx *ctx = nullptr;    // holds a reference
    uint64_t n_docs = 0;
    uint32_t dim = 0, vocab = 0, doc_id_base = 0;

    // embeddings
    float *rows = nullptr; // device
    bool rows_owned = false;
    uint16_t *rows_bf16 = nullptr; // device; set instead of `rows` for a bf16 corpus
    bool rows_bf16_owned = false;
    DevBuf max_row_norm; // u32[2]: bits of X = max_r |row r| and E = max_r |bf16(row r) - row r| (f32), taken when the f32 rows
                         // are set; NaN if any norm is
    bool screen_ok = false; // those maxima are finite and < 1e15: the bf16 screen's bound holds for this corpus
    // Two classes of rows (cosine_prefilter.hip): n_long > 0 = the maxima above are those of the rows that are NOT long; the
    // long ones are listed (local row numbers), marked in a bitmap, skipped by the margin selects and always rescored
    uint32_t n_long = 0;
    DevBuf long_list, long_bitmap;
    DevBuf screen_copy;     // the bf16 screening copy: bf16(rows), n_docs x dim x 2 B, made at finalize when the policy allows
    int screen_copy_policy = -1; // oi_index_set_screen_copy; -1 = the process default (OI_SCREEN_COPY, else AUTO)

    // staged forward index (between set_forward and finalize)
    bool forward_set = false, finalized = false;
    uint64_t total_tokens = 0;  // local
    uint64_t n_postings = 0;    // unique (doc, term) pairs
    uint32_t n_blocks = 0;      // ceil(n_docs / OI_BM25_BLOCK_DOCS)
    uint32_t n_win = 0;         // 2 * n_blocks: windows of OI_BM25_FINE_DOCS docs (the cells of a term's posting list)
    DevBuf uniq_keys;           // u64 (block | term | doc_in_block), sorted
    DevBuf tf;                  // u32 per unique key
    DevBuf doc_len;             // u32 per doc
    DevBuf df_local;            // u32 per term

    // finalized inverted index
    DevBuf postings;   // {u32 doc_in_block, f32 impact} per unique key, (term, doc) order: one contiguous list per term
    DevBuf cell_start; // u32 [vocab * n_win + 1]: start of the (term, window) run; the next entry is its end
    DevBuf idf;        // f32 per term
    // u32 [vocab][OI_BM25_FLOOR_RANKS] (round 4): bits of a lower bound of the term's r-th largest posting impact, r = 16, 64,
    // 256, 1024 (0: fewer than r postings).  A query's first BM25 threshold is read off this table (bm25_stream.hip: the plan).
    DevBuf impact_floor;
    // forward index kept for the batch scan (bm25_scan.hip)
    DevBuf fwd_terms;   // u32 per token
    DevBuf fwd_offsets; // u64 per doc + 1
    float avgdl = 0.f;  // global average doc length fixed at finalize
    uint32_t max_query_terms = 16; // contract for the batch-scan path (oi_index_set_max_query_terms)
    int bm25_mode = 0;             // 0 default (= 4 unless OI_BM25_MODE says otherwise), 1 term-at-a-time per workgroup (bm25.hip),
                                   // 2 scan of the forward index (bm25_scan.hip), 3 term-at-a-time per wave (bm25_wave.hip),
                                   // 4 the stream kernel (bm25_stream.hip)
    bool is_view = false;          // oi_index_view: the data belongs to another index; this handle only searches
};

// ---------------------------------------------------------------- kernels (host launchers)
// lexicon.hip
int oi_launch_lexicon(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n,
                      uint64_t blob_bytes, double *d_pol, uint8_t *d_spec);
int oi_launch_lexicon_fused(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n, uint64_t blob_bytes,
                            double *d_pol, uint8_t *d_spec, const uint8_t *d_sources, double tau, oi_social_counters *summary);
int oi_launch_social_summary(oi_ctx *ctx, const uint8_t *d_sources, const double *d_pol,
                             const uint8_t *d_spec, uint64_t n, double tau,
                             oi_social_counters *out_host);
int oi_launch_social_summary_segmented(oi_ctx *ctx, const uint8_t *d_sources, const double *d_pol, const uint8_t *d_spec,
                                       uint64_t n, const uint64_t *d_seg, uint64_t n_seg, double tau,
                                       oi_social_counters *d_out);
// headline.hip
int oi_launch_headline_scan(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n,
                            uint64_t blob_bytes, const uint8_t *ticker, uint64_t ticker_len,
                            const uint8_t *forms_blob, const uint32_t *form_offsets, uint32_t n_forms,
                            uint16_t *d_mask, uint64_t *d_order, uint8_t *d_about);
size_t oi_headline_params_bytes();
int oi_headline_build_params(void *dst, const uint8_t *ticker, uint64_t ticker_len, const uint8_t *forms_blob,
                             const uint32_t *form_offsets, uint32_t n_forms);
int oi_launch_headline_scan_params(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n,
                                   uint64_t blob_bytes, uint64_t text_bytes, const void *d_params, uint16_t *d_mask,
                                   uint64_t *d_order, uint8_t *d_about);
// select.hip
// Candidate pools, one per query, never touched by a global atomic in the batch kernels:
//   keys[q*stride + 0 .. carry_cap)                    the top-k carried over from earlier corpus chunks
//   keys[q*stride + carry_cap + s*seg_cap ..)          segment s: written by exactly one producer
//                                                      (cosine: workgroup s; BM25: doc block s)
//   carry_cnt[q], seg_cnt[q*seg_cnt_stride + s]        fill counts
struct PoolView {
    uint64_t *keys;
    uint32_t *carry_cnt;      // [n_queries]
    uint32_t *seg_cnt;        // [n_queries][seg_cnt_stride]
    uint32_t *tau_keys;       // [n_queries] orderable-u32 threshold (0 = accept all); may be null
    uint64_t stride;          // u64 entries between consecutive queries' pools
    uint32_t carry_cap;       // entries reserved for the carried top-k (>= k)
    uint32_t seg_cap;         // entries per segment
    uint32_t n_segs;          // segments in use
    uint32_t seg_cnt_stride;  // allocated segments per query
    uint32_t *overflow;       // single device word, set nonzero if a segment overflowed (bug guard)
};
// Optional behaviour of a select launch (cosine_prefilter.hip): eps2 != null = margin mode (keep every key within
// eps2[q] of the k-th score; needs compact, no sorted output, carry_cap >= 4096; an overflowing query sets
// *margin_gate); run_gate != null = the launch exits at once unless *run_gate is nonzero.
struct SelectExtra {
    const float *eps2 = nullptr;
    uint32_t *margin_gate = nullptr;
    const uint32_t *run_gate = nullptr;
    const uint32_t *skip_bitmap = nullptr; // margin mode: keys of docs whose bit (doc - skip_base) is set are left out of the selection
    uint32_t skip_base = 0;
};
int oi_launch_select(oi_ctx *ctx, const PoolView &pool, uint32_t n_queries, uint32_t k, bool compact,
                     float *out_scores, uint32_t *out_docs, uint32_t *out_counts, uint32_t out_stride,
                     const SelectExtra *extra = nullptr);
// shard s's lists start at scores + s*shard_stride (same for docs) and counts + s*count_stride
int oi_launch_lists_to_pool(oi_ctx *ctx, const float *scores, const uint32_t *docs,
                            const uint32_t *counts, uint64_t shard_stride, uint64_t count_stride,
                            uint32_t n_shards, uint32_t n_queries, uint32_t depth, const PoolView &pool);
int oi_launch_rrf(oi_ctx *ctx, const uint32_t *docs_a, const uint32_t *counts_a, const uint32_t *docs_b,
                  const uint32_t *counts_b, uint32_t n_queries, uint32_t depth, uint32_t k,
                  float *scores_out, uint32_t *docs_out, uint32_t *counts_out);
// cosine.hip
// Sets pool.n_segs / pool.seg_cap for this chunk (the following select must use the same view).
int oi_launch_cosine_chunk(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end,
                           uint32_t dim, const float *d_queries_padded, uint32_t n_queries,
                           uint32_t n_queries_padded, uint32_t doc_id_base, PoolView &pool);
// Largest chunk (rows) that can never overflow a pool of `stride` entries per query.
uint64_t oi_cosine_max_chunk_rows(const oi_ctx *ctx, uint32_t dim, uint32_t n_queries, uint64_t stride,
                                  uint32_t carry_cap);
int oi_launch_l2_normalize(oi_ctx *ctx, float *rows, uint64_t n, uint32_t dim);
// cosine_split.hip
bool oi_cosine_split_supported(uint32_t dim);
int oi_launch_cosine_split(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, uint32_t dim,
                           const float *d_queries, uint32_t nq, uint32_t doc_id_base, const PoolView &p);
// cosine_bf16.hip
bool oi_cosine_bf16_supported(uint32_t dim);
int oi_launch_cosine_bf16_chunk(oi_ctx *ctx, const uint16_t *rows, uint64_t row_begin, uint64_t row_end, uint32_t dim,
                                const float *d_queries, uint32_t n_queries, uint32_t doc_id_base, PoolView &pool);
uint32_t oi_cosine_query_padding(uint32_t n_queries);
// cosine_ksplit.hip
bool oi_cosine_ksplit_supported(uint32_t dim);
void oi_cosine_ksplit_geometry(const oi_ctx *ctx, uint64_t n_rows, uint32_t *n_segs, uint32_t *seg_cap);
// cosine_prefilter.hip: the bf16 screen + exact rescoring of an f32 corpus
bool oi_cosine_screen_supported(uint32_t dim);
void oi_cosine_screen_geometry(const oi_ctx *ctx, uint64_t n_rows, uint32_t *n_segs, uint32_t *seg_cap);
int oi_launch_row_norm_max(oi_ctx *ctx, const float *rows, uint64_t n, uint32_t dim, uint32_t *max_bits);
int oi_launch_screen_stage(oi_ctx *ctx, const float *d_queries, uint32_t n_queries, uint32_t dim,
                           const uint32_t *max_norm_bits, uint16_t *q_bf16, float *eps2, uint32_t *gate);
int oi_launch_cosine_screen_chunk(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, uint32_t dim,
                                  const uint16_t *q_bf16, uint32_t n_queries, uint32_t doc_id_base, PoolView &pool);
int oi_launch_make_screen_copy(oi_ctx *ctx, const float *rows, uint64_t n, uint32_t dim, uint16_t *out);
// cosine_screen_copy.hip
int oi_launch_cosine_screen_copy_chunk(oi_ctx *ctx, const uint16_t *copy_rows, uint64_t row_begin, uint64_t row_end, uint32_t dim,
                                       const uint16_t *q_bf16, uint32_t n_queries, uint32_t doc_id_base, PoolView &pool);
int oi_launch_screen_probe(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint32_t n_rows, uint32_t dim,
                           const uint16_t *q_bf16, uint32_t n_queries, float *d_out);
// speculative thresholds of the screen (cosine_prefilter.hip): the r-th best screen score so far - 2 eps, if above the proven
// threshold, for the NEXT chunk (spec_tau), the largest one used per query (spec_max); the check against the final one rides in
// oi_launch_rescore (spec_max, tau_final, gate, fail_host)
int oi_launch_spec_threshold(oi_ctx *ctx, const PoolView &pool, uint32_t n_queries, uint32_t r, const float *eps2, uint32_t *spec_tau,
                             uint32_t *spec_max);
int oi_launch_rescore(oi_ctx *ctx, const float *rows, uint64_t n_rows, uint32_t dim, uint32_t doc_id_base,
                      const float *d_queries, uint32_t n_queries, const PoolView &in, const PoolView &out,
                      const uint32_t *extra_docs = nullptr, uint32_t n_extra = 0, const uint32_t *spec_max = nullptr,
                      const uint32_t *tau_final = nullptr, uint32_t *gate = nullptr, uint32_t *fail_host = nullptr);
#define OI_LONG_ROWS_MAX 1024u // rows the two-class margin may set aside (more: one class, the corpus maxima, as before)
int oi_launch_row_norm_classes(oi_ctx *ctx, const float *rows, uint64_t n, uint32_t dim, float X0, float E0, uint32_t *cls,
                               uint32_t *bitmap, uint32_t *list, uint32_t cap);
int oi_launch_cosine_ksplit(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, uint32_t dim,
                            const float *q, uint32_t nq, bool two_tiles, uint32_t doc_id_base, const PoolView &p);
// bm25.hip
int oi_bm25_stage_forward(oi_index *idx, const uint32_t *d_terms, const uint64_t *d_offsets);
int oi_bm25_finalize(oi_index *idx, uint64_t global_n, uint64_t global_tokens, const uint32_t *global_df_host);
// bm25_scan.hip
void oi_bm25_scan_geometry(const oi_ctx *ctx, uint64_t n_docs, uint32_t *n_segs, uint32_t *seg_cap);
uint32_t oi_bm25_scan_pass_queries(uint32_t max_terms_per_query);
int oi_launch_bm25_scan(oi_index *idx, const uint32_t *d_q_terms, const uint32_t *d_q_offsets, uint32_t q_begin,
                        uint32_t nq, uint64_t doc_begin, uint64_t doc_end, float avgdl, bool run_setup,
                        const PoolView &pool);
// bm25_wave.hip: one wave per (block, query) task; `pool` = the view of queries [q_begin, q_begin + nq)
uint32_t oi_bm25_wave_pass_queries(void);
int oi_launch_bm25_wave(oi_index *idx, const uint32_t *d_q_terms, const uint32_t *d_q_offsets, uint32_t q_begin,
                        uint32_t nq, const PoolView &pool, uint32_t block_begin, uint32_t block_end);
// bm25_stream.hip: every wave streams a weight-balanced range of (query, block) tasks through an LDS ring; `pool` = the view
// of queries [q_begin, q_begin + nq) with segments of oi_bm25_stream_seg_cap() keys.  oi_launch_bm25_plan runs once per
// pass, before the first phase: it zeroes the pass's pool state (`state_words` words at `state`) and weighs the queries.
uint32_t oi_bm25_stream_pass_queries(void);
uint32_t oi_bm25_stream_seg_cap(uint32_t depth, bool first_phase);
int oi_launch_bm25_plan(oi_index *idx, const uint32_t *d_q_terms, const uint32_t *d_q_offsets, uint32_t q_begin, uint32_t nq,
                        uint32_t *state, uint64_t state_words, uint32_t depth, uint32_t tau_off, uint32_t tau_words, bool with_floors);
int oi_launch_bm25_stream(oi_index *idx, const uint32_t *d_q_terms, const uint32_t *d_q_offsets, uint32_t q_begin,
                          uint32_t nq, uint32_t depth, const PoolView &pool, uint32_t block_begin, uint32_t block_end);
// Doc blocks [block_begin, block_end); candidates below pool.tau_keys (if set) are dropped.
int oi_launch_bm25(oi_index *idx, const uint32_t *d_q_terms, const uint32_t *d_q_offsets,
                   uint32_t n_queries, uint32_t depth, const PoolView &pool, uint32_t block_begin,
                   uint32_t block_end);

// comm.hip: RCCL through dlopen (no link-time dependency)
struct oi_comm {
    oi_ctx *ctx = nullptr; // holds a reference
    void *nccl = nullptr;  // ncclComm_t
    uint32_t rank = 0, world = 1;
};
int oi_rccl_unique_id(uint8_t *id_out);
int oi_rccl_init(void **comm_out, uint32_t world, const uint8_t *id_bytes, uint32_t rank);
void oi_rccl_destroy(void *comm);
int oi_rccl_all_gather_u32(void *comm, const uint32_t *send, uint32_t *recv, size_t words, hipStream_t st);
int oi_rccl_all_reduce_sum(void *comm, void *buf, size_t count, bool u64, hipStream_t st);
