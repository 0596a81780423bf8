//! Raw bindings, one-to-one with include/openintel_hip.h (ABI version 1).
//! tests/test_rust_shim_sources.py keeps this list and the header in step.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct OiCtx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct OiIndex {
    _p: [u8; 0],
}
#[repr(C)]
pub struct OiPipeline {
    _p: [u8; 0],
}
#[repr(C)]
pub struct OiComm {
    _p: [u8; 0],
}
pub const OI_COMM_ID_BYTES: usize = 128;

/// `oi_social_counters`: the raw sums of SpeculationEngine::social_summary (speculation_engine.rs:76-97).
#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct OiSocialCounters {
    pub total: u64,
    pub by_source: [u64; 2], // [reddit, bluesky]  source_kind.rs:5-8
    pub bullish: u64,
    pub bearish: u64,
    pub neutral: u64,
    pub spec_count: u64,
    pub polarity_sum: f64,
}

pub const OI_OK: c_int = 0;
pub const OI_ERR_ANALYZER_MISMATCH: c_int = -3;
pub const OI_ERR_OVERFLOW: c_int = -8;
pub const OI_ERR_COMM: c_int = -9;
pub const OI_HOST: c_int = 0;
pub const OI_DEVICE: c_int = 1;
pub const OI_COSINE_EXACT: c_int = 0;
pub const OI_COSINE_SPLIT: c_int = 1;
pub const OI_COSINE_SCREEN: c_int = 2;
pub const OI_COSINE_SCREEN_COPY: c_int = 3;
pub const OI_COSINE_SCREEN_STREAM: c_int = 4;
pub const OI_SCREEN_COPY_AUTO: c_int = 0;
pub const OI_SCREEN_COPY_NEVER: c_int = 1;
pub const OI_SCREEN_COPY_ALWAYS: c_int = 2;

extern "C" {
    pub fn oi_abi_version() -> c_int;
    pub fn oi_last_error() -> *const c_char; // thread-local
    pub fn oi_create(device_ordinal: c_int, out: *mut *mut OiCtx) -> c_int;
    pub fn oi_create_like(like: *mut OiCtx, out: *mut *mut OiCtx) -> c_int;
    pub fn oi_destroy(ctx: *mut OiCtx);
    pub fn oi_workspace_bytes(ctx: *mut OiCtx, device_bytes_out: *mut u64, pinned_host_bytes_out: *mut u64) -> c_int;
    pub fn oi_set_stream(ctx: *mut OiCtx, hip_stream: *mut c_void) -> c_int;
    pub fn oi_synchronize(ctx: *mut OiCtx) -> c_int;
    pub fn oi_set_cosine_mode(ctx: *mut OiCtx, mode: c_int) -> c_int;
    pub fn oi_set_overlap(ctx: *mut OiCtx, enable: c_int) -> c_int;
    pub fn oi_set_screen_speculation(ctx: *mut OiCtx, enable: c_int) -> c_int;
    pub fn oi_set_graph_replay(ctx: *mut OiCtx, enable: c_int) -> c_int;

    pub fn oi_lexicon_analyze(ctx: *mut OiCtx, text_blob: *const u8, offsets: *const u64, n_posts: u64,
                              polarity_out: *mut f64, speculative_out: *mut u8) -> c_int;
    pub fn oi_lexicon_analyze_device(ctx: *mut OiCtx, d_text_blob: *const u8, d_offsets: *const u64, n_posts: u64,
                                     blob_bytes: u64, d_polarity_out: *mut f64, d_speculative_out: *mut u8) -> c_int;
    pub fn oi_social_summary(ctx: *mut OiCtx, sources: *const u8, n_posts: u64, polarity: *const f64,
                             speculative: *const u8, n_signals: u64, bull_bear_threshold: f64, location: c_int,
                             out_host: *mut OiSocialCounters) -> c_int;
    pub fn oi_lexicon_summary_device(ctx: *mut OiCtx, d_text_blob: *const u8, d_offsets: *const u64, n_posts: u64,
                                     blob_bytes: u64, d_sources: *const u8, bull_bear_threshold: f64,
                                     d_polarity_out: *mut f64, d_speculative_out: *mut u8,
                                     out_host: *mut OiSocialCounters) -> c_int;

    pub fn oi_social_summary_segmented(ctx: *mut OiCtx, sources: *const u8, polarity: *const f64, speculative: *const u8,
                                       n_posts: u64, seg_offsets: *const u64, n_segments: u64, bull_bear_threshold: f64,
                                       location: c_int, out: *mut OiSocialCounters) -> c_int;
    pub fn oi_lexicon_scan_segments_device(ctx: *mut OiCtx, d_text_blob: *const u8, d_offsets: *const u64, n_posts: u64,
                                           blob_bytes: u64, d_sources: *const u8, d_seg_offsets: *const u64,
                                           n_segments: u64, bull_bear_threshold: f64, d_polarity_out: *mut f64,
                                           d_speculative_out: *mut u8, d_out: *mut OiSocialCounters) -> c_int;

    pub fn oi_catalyst_keyword(index: u32) -> *const c_char;
    pub fn oi_headline_scan(ctx: *mut OiCtx, blob: *const u8, offsets: *const u64, n_titles: u64, ticker: *const u8,
                            ticker_len: u64, forms_blob: *const u8, form_offsets: *const u32, n_forms: u32,
                            mask_out: *mut u16, order_out: *mut u64, about_out: *mut u8) -> c_int;
    pub fn oi_headline_scan_rows(ctx: *mut OiCtx, blob: *const u8, offsets: *const u64, n_titles: u64,
                                 row_offsets: *const u64, n_rows: u32, tickers_blob: *const u8,
                                 ticker_offsets: *const u32, forms_blob: *const u8, form_offsets: *const u32,
                                 row_form_offsets: *const u32, mask_out: *mut u16, order_out: *mut u64,
                                 about_out: *mut u8) -> c_int;
    pub fn oi_headline_scan_device(ctx: *mut OiCtx, d_blob: *const u8, d_offsets: *const u64, n_titles: u64,
                                   blob_bytes: u64, ticker: *const u8, ticker_len: u64, forms_blob: *const u8,
                                   form_offsets: *const u32, n_forms: u32, d_mask_out: *mut u16,
                                   d_order_out: *mut u64, d_about_out: *mut u8) -> c_int;

    pub fn oi_index_create(ctx: *mut OiCtx, n_docs: u64, dim: u32, vocab: u32, doc_id_base: u32,
                           out: *mut *mut OiIndex) -> c_int;
    pub fn oi_index_destroy(idx: *mut OiIndex);
    pub fn oi_index_view(src: *mut OiIndex, ctx: *mut OiCtx, out: *mut *mut OiIndex) -> c_int;
    pub fn oi_index_set_embeddings(idx: *mut OiIndex, rows: *mut f32, location: c_int, normalize: c_int) -> c_int;
    pub fn oi_index_set_embeddings_bf16(idx: *mut OiIndex, rows: *const u16, location: c_int) -> c_int;
    pub fn oi_index_set_forward(idx: *mut OiIndex, term_ids: *const u32, doc_offsets: *const u64, location: c_int) -> c_int;
    pub fn oi_index_local_stats(idx: *mut OiIndex, total_tokens_out: *mut u64, df_out_host: *mut u32) -> c_int;
    pub fn oi_index_finalize(idx: *mut OiIndex, global_n_docs: u64, global_total_tokens: u64,
                             global_df_host: *const u32) -> c_int;
    pub fn oi_index_long_rows(idx: *mut OiIndex, n_out: *mut u32) -> c_int;
    pub fn oi_index_set_screen_copy(idx: *mut OiIndex, policy: c_int) -> c_int;
    pub fn oi_index_bytes(idx: *mut OiIndex, rows_owned_bytes_out: *mut u64, screen_copy_bytes_out: *mut u64,
                          bm25_bytes_out: *mut u64) -> c_int;
    pub fn oi_index_set_bm25_mode(idx: *mut OiIndex, mode: c_int) -> c_int;
    pub fn oi_index_set_max_query_terms(idx: *mut OiIndex, max_terms: u32) -> c_int;

    pub fn oi_search_lists(idx: *mut OiIndex, query_vecs: *const f32, query_terms: *const u32,
                           q_term_offsets: *const u32, n_queries: u32, depth: u32, location: c_int,
                           cos_scores: *mut f32, cos_docs: *mut u32, cos_counts: *mut u32, bm25_scores: *mut f32,
                           bm25_docs: *mut u32, bm25_counts: *mut u32) -> c_int;
    pub fn oi_merge_lists(ctx: *mut OiCtx, scores: *const f32, docs: *const u32, counts: *const u32, n_shards: u32,
                          n_queries: u32, depth: u32, location: c_int, scores_out: *mut f32, docs_out: *mut u32,
                          counts_out: *mut u32) -> c_int;
    pub fn oi_search_lists_packed(idx: *mut OiIndex, query_vecs: *const f32, query_terms: *const u32,
                                  q_term_offsets: *const u32, n_queries: u32, depth: u32, location: c_int,
                                  packed_out: *mut u32) -> c_int;
    pub fn oi_fuse_packed(ctx: *mut OiCtx, packed_all: *const u32, n_shards: u32, n_queries: u32, depth: u32, k: u32,
                          location: c_int, scores_out: *mut f32, docs_out: *mut u32, counts_out: *mut u32) -> c_int;
    pub fn oi_rrf_fuse(ctx: *mut OiCtx, docs_a: *const u32, counts_a: *const u32, docs_b: *const u32,
                       counts_b: *const u32, n_queries: u32, depth: u32, k: u32, location: c_int,
                       scores_out: *mut f32, docs_out: *mut u32, counts_out: *mut u32) -> c_int;
    pub fn oi_search(idx: *mut OiIndex, query_vecs: *const f32, query_terms: *const u32, q_term_offsets: *const u32,
                     n_queries: u32, depth: u32, k: u32, location: c_int, scores_out: *mut f32, docs_out: *mut u32,
                     counts_out: *mut u32) -> c_int;

    // the row-sharded query with RCCL inside the library (one process per GPU)
    pub fn oi_comm_unique_id(id_out: *mut u8) -> c_int;
    pub fn oi_comm_create(ctx: *mut OiCtx, id: *const u8, rank: u32, world: u32, out: *mut *mut OiComm) -> c_int;
    pub fn oi_comm_destroy(comm: *mut OiComm);
    pub fn oi_index_finalize_sharded(idx: *mut OiIndex, comm: *mut OiComm) -> c_int;
    pub fn oi_search_sharded(idx: *mut OiIndex, comm: *mut OiComm, query_vecs: *const f32, query_terms: *const u32,
                             q_term_offsets: *const u32, n_queries: u32, depth: u32, k: u32, location: c_int,
                             scores_out: *mut f32, docs_out: *mut u32, counts_out: *mut u32) -> c_int;

    pub fn oi_pipeline_create(idx: *mut OiIndex, comm: *mut OiComm, lanes: u32, max_queries: u32, max_query_terms: u32,
                              depth: u32, k: u32, out: *mut *mut OiPipeline) -> c_int;
    pub fn oi_pipeline_destroy(p: *mut OiPipeline);
    pub fn oi_pipeline_submit(p: *mut OiPipeline, query_vecs: *const f32, query_terms: *const u32,
                              q_term_offsets: *const u32, n_queries: u32, location: c_int, scores_out: *mut f32,
                              docs_out: *mut u32, counts_out: *mut u32, ticket_out: *mut u64) -> c_int;
    pub fn oi_pipeline_wait(p: *mut OiPipeline, ticket: u64, host_sync: c_int) -> c_int;
    pub fn oi_pipeline_drain(p: *mut OiPipeline) -> c_int;
    pub fn oi_pipeline_workspace_bytes(p: *mut OiPipeline, device_bytes_out: *mut u64,
                                       pinned_host_bytes_out: *mut u64) -> c_int;
    pub fn oi_pipeline_concurrent_streams(p: *mut OiPipeline, concurrent_out: *mut u32, streams_out: *mut u32) -> c_int;
    pub fn oi_pipeline_profile_reset(p: *mut OiPipeline, enable: c_int) -> c_int;
    pub fn oi_pipeline_profile_read(p: *mut OiPipeline, kernel_tag: *const c_char, total_ms_out: *mut f64,
                                    launches_out: *mut u64) -> c_int;

    pub fn oi_screen_probe(idx: *mut OiIndex, query_vecs: *const f32, n_queries: u32, row_begin: u64, n_rows: u32,
                           screen_scores_out: *mut f32, eps_out: *mut f32) -> c_int;
    pub fn oi_profile_reset(ctx: *mut OiCtx, enable: c_int) -> c_int;
    pub fn oi_profile_read(ctx: *mut OiCtx, kernel_tag: *const c_char, total_ms_out: *mut f64,
                           launches_out: *mut u64) -> c_int;
}
