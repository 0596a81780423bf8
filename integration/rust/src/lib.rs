//! Safe wrappers over libopenintel_hip.so and, with feature "reference", the adapter that implements the
//! reference's `PostAnalyzer` port (src/domain/ports/post_analyzer.rs:7-11) on top of them.
//!
//! NOT compile-checked in the build image (no cargo/rustc there); see Cargo.toml.
pub mod ffi;

use std::ffi::CStr;
use std::sync::Arc;

/// A failed library call: the status code and the thread-local message (`oi_last_error`).
#[derive(Debug, Clone)]
pub struct HipError {
    pub code: i32,
    pub message: String,
}
impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "libopenintel_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for HipError {}

fn check(rc: i32) -> Result<(), HipError> {
    if rc == 0 {
        return Ok(());
    }
    // never unwind across the ABI: the library reports, the shim converts
    let message = unsafe { CStr::from_ptr(ffi::oi_last_error()) }.to_string_lossy().into_owned();
    Err(HipError { code: rc, message })
}

/// Owns an `oi_ctx`.  The library serialises calls on a ctx with an internal mutex, so the handle may be shared
/// between threads: that is what makes the `Send + Sync` below sound (`PostAnalyzer: Send + Sync`,
/// post_analyzer.rs:7; tests/test_gpu_threads.py drives one ctx from eight host threads).
pub struct HipCtx(*mut ffi::OiCtx);
unsafe impl Send for HipCtx {}
unsafe impl Sync for HipCtx {}
impl Drop for HipCtx {
    fn drop(&mut self) {
        unsafe { ffi::oi_destroy(self.0) }
    }
}
impl HipCtx {
    pub fn new(device: i32) -> Result<Arc<Self>, HipError> {
        let mut p = std::ptr::null_mut();
        check(unsafe { ffi::oi_create(device, &mut p) })?;
        Ok(Arc::new(HipCtx(p)))
    }
    pub fn raw(&self) -> *mut ffi::OiCtx {
        self.0
    }
}

/// Gather strings into the FFI layout: one UTF-8 blob + (n + 1) u64 offsets.  In the reference every post owns
/// its text as a separate heap `String` (social_post.rs:25-27).
pub fn gather<'a, I: IntoIterator<Item = &'a str>>(texts: I) -> (Vec<u8>, Vec<u64>) {
    let mut blob = Vec::new();
    let mut offsets = vec![0u64];
    for t in texts {
        blob.extend_from_slice(t.as_bytes());
        offsets.push(blob.len() as u64);
    }
    (blob, offsets)
}

/// `LexiconAnalyzer::score` for every text (lexicon.rs:53-73): (polarity, speculative), index-aligned.
pub fn analyze_texts(ctx: &HipCtx, texts: &[&str]) -> Result<Vec<(f64, bool)>, HipError> {
    let (blob, offsets) = gather(texts.iter().copied());
    let n = texts.len();
    let (mut pol, mut spec) = (vec![0f64; n], vec![0u8; n]);
    check(unsafe {
        ffi::oi_lexicon_analyze(ctx.raw(), blob.as_ptr(), offsets.as_ptr(), n as u64, pol.as_mut_ptr(), spec.as_mut_ptr())
    })?;
    Ok(pol.into_iter().zip(spec).map(|(p, s)| (p, s != 0)).collect())
}

/// The batch tools' unit of work (mcp/tools.rs:193-225 `run_scan`, :303-352 `run_compare`): the texts of MANY tickers
/// through one scan, and every ticker's `social_summary` sums (speculation_engine.rs:76-97) from one reduction.
/// `segments[s]` = that ticker's (text, source) pairs in fetch order; returns the per-post signals, index-aligned with
/// the flattened input, and one counters record per ticker whose `polarity_sum` is the reference's input-order sum.
pub fn analyze_segments(ctx: &HipCtx, segments: &[Vec<(&str, u8)>], bull_bear_threshold: f64)
    -> Result<(Vec<(f64, bool)>, Vec<ffi::OiSocialCounters>), HipError> {
    let (blob, offsets) = gather(segments.iter().flatten().map(|(t, _)| *t));
    let sources: Vec<u8> = segments.iter().flatten().map(|(_, s)| *s).collect();
    let n = sources.len();
    let mut seg = vec![0u64];
    for s in segments {
        seg.push(seg.last().unwrap() + s.len() as u64);
    }
    let (mut pol, mut spec) = (vec![0f64; n], vec![0u8; n]);
    check(unsafe {
        ffi::oi_lexicon_analyze(ctx.raw(), blob.as_ptr(), offsets.as_ptr(), n as u64, pol.as_mut_ptr(), spec.as_mut_ptr())
    })?;
    let mut out: Vec<ffi::OiSocialCounters> = (0..segments.len()).map(|_| unsafe { std::mem::zeroed() }).collect();
    check(unsafe {
        ffi::oi_social_summary_segmented(ctx.raw(), sources.as_ptr(), pol.as_ptr(), spec.as_ptr(), n as u64, seg.as_ptr(),
                                         segments.len() as u64, bull_bear_threshold, ffi::OI_HOST, out.as_mut_ptr())
    })?;
    Ok((pol.into_iter().zip(spec).map(|(p, s)| (p, s != 0)).collect(), out))
}

/// Per-title result of the headline gate's scan (dip.rs:247-272).
pub struct TitleScan {
    /// indices into CATALYST_KEYWORDS (dip.rs:38-55) in first-occurrence order: `catalyst_hits(&[title])`
    pub hits: Vec<usize>,
    /// `headline_mentions_company(title, ticker, name_forms)`
    pub about_company: bool,
}

pub fn scan_titles(ctx: &HipCtx, titles: &[&str], ticker: &str, name_forms: &[String]) -> Result<Vec<TitleScan>, HipError> {
    let (blob, offsets) = gather(titles.iter().copied());
    let mut fblob = Vec::new();
    let mut foffs = vec![0u32];
    for f in name_forms {
        fblob.extend_from_slice(f.as_bytes());
        foffs.push(fblob.len() as u32);
    }
    let n = titles.len();
    let (mut mask, mut order, mut about) = (vec![0u16; n], vec![0u64; n], vec![0u8; n]);
    check(unsafe {
        ffi::oi_headline_scan(ctx.raw(), blob.as_ptr(), offsets.as_ptr(), n as u64, ticker.as_ptr(), ticker.len() as u64,
                              fblob.as_ptr(), foffs.as_ptr(), name_forms.len() as u32, mask.as_mut_ptr(),
                              order.as_mut_ptr(), about.as_mut_ptr())
    })?;
    Ok((0..n)
        .map(|i| TitleScan {
            hits: (0..mask[i].count_ones()).map(|j| ((order[i] >> (4 * j)) & 15) as usize).collect(),
            about_company: about[i] != 0,
        })
        .collect())
}

/// The gate's scan for every row of a dip scan in one call (application/dip.rs: `check` per loser; `oi_headline_scan_rows`):
/// `rows[r]` = (that loser's headlines, its ticker, its `company_name_forms`).  Per row what `scan_titles` returns.
pub fn scan_title_rows(ctx: &HipCtx, rows: &[(Vec<&str>, &str, Vec<String>)]) -> Result<Vec<Vec<TitleScan>>, HipError> {
    let (blob, offsets) = gather(rows.iter().flat_map(|r| r.0.iter().copied()));
    let (mut row_off, mut tick_off, mut rform_off) = (vec![0u64], vec![0u32], vec![0u32]);
    let (mut tblob, mut fblob, mut foffs) = (Vec::new(), Vec::new(), vec![0u32]);
    for (titles, ticker, forms) in rows {
        row_off.push(row_off.last().unwrap() + titles.len() as u64);
        tblob.extend_from_slice(ticker.as_bytes());
        tick_off.push(tblob.len() as u32);
        for f in forms {
            fblob.extend_from_slice(f.as_bytes());
            foffs.push(fblob.len() as u32);
        }
        rform_off.push(foffs.len() as u32 - 1);
    }
    let n = *row_off.last().unwrap() as usize;
    let (mut mask, mut order, mut about) = (vec![0u16; n], vec![0u64; n], vec![0u8; n]);
    check(unsafe {
        ffi::oi_headline_scan_rows(ctx.raw(), blob.as_ptr(), offsets.as_ptr(), n as u64, row_off.as_ptr(), rows.len() as u32,
                                   tblob.as_ptr(), tick_off.as_ptr(), fblob.as_ptr(), foffs.as_ptr(), rform_off.as_ptr(),
                                   mask.as_mut_ptr(), order.as_mut_ptr(), about.as_mut_ptr())
    })?;
    Ok(rows
        .iter()
        .enumerate()
        .map(|(r, _)| {
            (row_off[r] as usize..row_off[r + 1] as usize)
                .map(|i| TitleScan {
                    hits: (0..mask[i].count_ones()).map(|j| ((order[i] >> (4 * j)) & 15) as usize).collect(),
                    about_company: about[i] != 0,
                })
                .collect()
        })
        .collect())
}

/// One corpus shard in HBM (`oi_index`).  New API: the reference has no retrieval port.
pub struct HipIndex {
    ctx: Arc<HipCtx>,
    idx: *mut ffi::OiIndex,
    dim: usize,
    source: Option<Arc<HipIndex>>, // a view keeps the index it borrows from alive (oi_index_view)
}
unsafe impl Send for HipIndex {}
unsafe impl Sync for HipIndex {}
impl Drop for HipIndex {
    fn drop(&mut self) {
        unsafe { ffi::oi_index_destroy(self.idx) }
    }
}
pub struct RankedPost {
    pub doc_id: u32,
    pub score: f32,
}
impl HipIndex {
    /// rows: n_docs x dim f32 (copied to HBM, L2-normalised); forward index: doc d owns terms[offsets[d]..offsets[d+1]].
    pub fn build(ctx: Arc<HipCtx>, rows: &mut [f32], dim: usize, vocab: u32, terms: &[u32], offsets: &[u64]) -> Result<Self, HipError> {
        let n_docs = (rows.len() / dim) as u64;
        let mut idx = std::ptr::null_mut();
        check(unsafe { ffi::oi_index_create(ctx.raw(), n_docs, dim as u32, vocab, 0, &mut idx) })?;
        let me = HipIndex { ctx, idx, dim, source: None };
        check(unsafe { ffi::oi_index_set_embeddings(me.idx, rows.as_mut_ptr(), ffi::OI_HOST, 1) })?;
        check(unsafe { ffi::oi_index_set_forward(me.idx, terms.as_ptr(), offsets.as_ptr(), ffi::OI_HOST) })?;
        let mut tokens = 0u64;
        check(unsafe { ffi::oi_index_local_stats(me.idx, &mut tokens, std::ptr::null_mut()) })?;
        check(unsafe { ffi::oi_index_finalize(me.idx, n_docs, tokens, std::ptr::null()) })?;
        Ok(me)
    }
    /// A second, read-only handle on this shard bound to `ctx` (a context of its own: another stream, other workspaces),
    /// so that two searches can be in flight at once; costs no HBM.
    pub fn view(self: &Arc<Self>, ctx: Arc<HipCtx>) -> Result<HipIndex, HipError> {
        let mut idx = std::ptr::null_mut();
        check(unsafe { ffi::oi_index_view(self.idx, ctx.raw(), &mut idx) })?;
        Ok(HipIndex { ctx, idx, dim: self.dim, source: Some(Arc::clone(self)) })
    }
    /// Hybrid BM25 + cosine + RRF: one ranked list (<= k) per query, in query order.
    pub fn search(&self, query_vecs: &[f32], query_terms: &[Vec<u32>], k: usize, depth: usize) -> Result<Vec<Vec<RankedPost>>, HipError> {
        let b = query_terms.len();
        assert_eq!(query_vecs.len(), b * self.dim);
        let mut flat = Vec::new();
        let mut offs = vec![0u32];
        for t in query_terms {
            flat.extend_from_slice(t);
            offs.push(flat.len() as u32);
        }
        if flat.is_empty() {
            flat.push(0);
        }
        let (mut s, mut d, mut c) = (vec![0f32; b * k], vec![0u32; b * k], vec![0u32; b]);
        check(unsafe {
            ffi::oi_search(self.idx, query_vecs.as_ptr(), flat.as_ptr(), offs.as_ptr(), b as u32, depth as u32, k as u32,
                           ffi::OI_HOST, s.as_mut_ptr(), d.as_mut_ptr(), c.as_mut_ptr())
        })?;
        let _ = (&self.ctx, &self.source);
        Ok((0..b)
            .map(|q| (0..c[q] as usize).map(|i| RankedPost { doc_id: d[q * k + i], score: s[q * k + i] }).collect())
            .collect())
    }
}

/// An RCCL communicator owned by the library (`oi_comm`): the multi-GPU exchange for a host that brings no collective
/// library of its own -- the reference is single-process (src/domain/ports/mod.rs:1-8), its composition root
/// (src/main.rs:17-39) would start one process per GPU and ship the 128-byte id over any channel it likes.
pub struct HipComm {
    ctx: Arc<HipCtx>,
    comm: *mut ffi::OiComm,
    pub rank: u32,
    pub world: u32,
}
unsafe impl Send for HipComm {}
impl Drop for HipComm {
    fn drop(&mut self) {
        unsafe { ffi::oi_comm_destroy(self.comm) }
    }
}
impl HipComm {
    /// Rank 0 only; every rank passes the same bytes to `create`.
    pub fn unique_id() -> Result<[u8; ffi::OI_COMM_ID_BYTES], HipError> {
        let mut id = [0u8; ffi::OI_COMM_ID_BYTES];
        check(unsafe { ffi::oi_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }
    /// Collective: returns when every rank of `world` has called it.
    pub fn create(ctx: Arc<HipCtx>, id: &[u8; ffi::OI_COMM_ID_BYTES], rank: u32, world: u32) -> Result<Self, HipError> {
        let mut comm = std::ptr::null_mut();
        check(unsafe { ffi::oi_comm_create(ctx.raw(), id.as_ptr(), rank, world, &mut comm) })?;
        Ok(HipComm { ctx, comm, rank, world })
    }
}

impl HipIndex {
    /// One rank's row shard of a collection split over `comm.world` GPUs: rows [doc_id_base, doc_id_base + n) of the
    /// global collection.  Collective (the df / N / token all-reduce runs inside `oi_index_finalize_sharded`).
    pub fn build_shard(comm: &HipComm, doc_id_base: u32, rows: &mut [f32], dim: usize, vocab: u32, terms: &[u32],
                       offsets: &[u64]) -> Result<Self, HipError> {
        let n_docs = (rows.len() / dim) as u64;
        let mut idx = std::ptr::null_mut();
        check(unsafe { ffi::oi_index_create(comm.ctx.raw(), n_docs, dim as u32, vocab, doc_id_base, &mut idx) })?;
        let me = HipIndex { ctx: comm.ctx.clone(), idx, dim, source: None };
        check(unsafe { ffi::oi_index_set_embeddings(me.idx, rows.as_mut_ptr(), ffi::OI_HOST, 1) })?;
        check(unsafe { ffi::oi_index_set_forward(me.idx, terms.as_ptr(), offsets.as_ptr(), ffi::OI_HOST) })?;
        check(unsafe { ffi::oi_index_finalize_sharded(me.idx, comm.comm) })?;
        Ok(me)
    }
    /// The hybrid query over ALL shards (collective, same queries on every rank, one ncclAllGather per batch inside):
    /// identical lists on every rank, global doc ids.
    pub fn search_sharded(&self, comm: &HipComm, query_vecs: &[f32], query_terms: &[Vec<u32>], k: usize, depth: usize)
                          -> Result<Vec<Vec<RankedPost>>, HipError> {
        let b = query_terms.len();
        assert_eq!(query_vecs.len(), b * self.dim);
        let mut flat = Vec::new();
        let mut offs = vec![0u32];
        for t in query_terms {
            flat.extend_from_slice(t);
            offs.push(flat.len() as u32);
        }
        if flat.is_empty() {
            flat.push(0);
        }
        let (mut s, mut d, mut c) = (vec![0f32; b * k], vec![0u32; b * k], vec![0u32; b]);
        check(unsafe {
            ffi::oi_search_sharded(self.idx, comm.comm, query_vecs.as_ptr(), flat.as_ptr(), offs.as_ptr(), b as u32,
                                   depth as u32, k as u32, ffi::OI_HOST, s.as_mut_ptr(), d.as_mut_ptr(), c.as_mut_ptr())
        })?;
        Ok((0..b)
            .map(|q| (0..c[q] as usize).map(|i| RankedPost { doc_id: d[q * k + i], score: s[q * k + i] }).collect())
            .collect())
    }
}

/// Several batches in flight through lanes the LIBRARY owns (oi_pipeline_*): the throughput form of `search` /
/// `search_sharded` for a host without streams of its own.  Results are bit-identical to the one-call forms.
pub struct HipPipeline {
    p: *mut ffi::OiPipeline,
    dim: usize,
    k: usize,
    _idx: Arc<HipIndex>,
}
unsafe impl Send for HipPipeline {}
impl Drop for HipPipeline {
    fn drop(&mut self) {
        unsafe { ffi::oi_pipeline_destroy(self.p) } // drains first
    }
}
/// The owned output buffers of one submitted batch; filled when `HipPipeline::wait(ticket)` returns.
pub struct PendingBatch {
    pub ticket: u64,
    n: usize,
    scores: Vec<f32>,
    docs: Vec<u32>,
    counts: Vec<u32>,
}
impl HipPipeline {
    /// `comm`: None = one shard, no exchange; Some = row-sharded, ONE ncclAllGather per batch inside the library.
    pub fn create(idx: Arc<HipIndex>, comm: Option<&HipComm>, lanes: u32, max_queries: u32, max_query_terms: u32, depth: u32, k: u32)
                  -> Result<Self, HipError> {
        let mut p = std::ptr::null_mut();
        let c = comm.map(|c| c.comm).unwrap_or(std::ptr::null_mut());
        check(unsafe { ffi::oi_pipeline_create(idx.idx, c, lanes, max_queries, max_query_terms, depth, k, &mut p) })?;
        Ok(Self { p, dim: idx.dim, k: k as usize, _idx: idx })
    }
    /// Asynchronous: the inputs are copied during the call, the returned buffers are written by `wait`.
    pub fn submit(&self, query_vecs: &[f32], query_terms: &[Vec<u32>]) -> Result<PendingBatch, HipError> {
        let b = query_terms.len();
        assert_eq!(query_vecs.len(), b * self.dim);
        let mut flat = Vec::new();
        let mut offs = vec![0u32];
        for t in query_terms {
            flat.extend_from_slice(t);
            offs.push(flat.len() as u32);
        }
        if flat.is_empty() {
            flat.push(0);
        }
        let mut out = PendingBatch { ticket: 0, n: b, scores: vec![0f32; b * self.k], docs: vec![0u32; b * self.k], counts: vec![0u32; b] };
        check(unsafe {
            ffi::oi_pipeline_submit(self.p, query_vecs.as_ptr(), flat.as_ptr(), offs.as_ptr(), b as u32, ffi::OI_HOST,
                                    out.scores.as_mut_ptr(), out.docs.as_mut_ptr(), out.counts.as_mut_ptr(), &mut out.ticket)
        })?;
        Ok(out) // (the Vecs' heap buffers do not move with the struct: the library's pointers stay valid until wait)
    }
    pub fn wait(&self, batch: PendingBatch) -> Result<Vec<Vec<RankedPost>>, HipError> {
        check(unsafe { ffi::oi_pipeline_wait(self.p, batch.ticket, 1) })?;
        let k = self.k;
        Ok((0..batch.n)
            .map(|q| (0..batch.counts[q] as usize).map(|i| RankedPost { doc_id: batch.docs[q * k + i], score: batch.scores[q * k + i] }).collect())
            .collect())
    }
    pub fn drain(&self) -> Result<(), HipError> {
        check(unsafe { ffi::oi_pipeline_drain(self.p) })
    }
}

/// The adapter for the reference: `impl PostAnalyzer for HipLexiconAnalyzer`.
#[cfg(feature = "reference")]
pub mod adapter {
    use super::*;
    use async_trait::async_trait;
    use openintel::domain::entities::social_post::SocialPost;
    use openintel::domain::error::DomainError;
    use openintel::domain::ports::post_analyzer::PostAnalyzer;
    use openintel::domain::values::{polarity::Polarity, post_signal::PostSignal};

    fn fail(e: HipError) -> DomainError {
        // nonzero status -> DomainError::SourceFailure (src/domain/error.rs:16-17)
        DomainError::SourceFailure { name: "hip-analyzer".into(), message: e.to_string() }
    }

    pub struct HipLexiconAnalyzer {
        ctx: Arc<HipCtx>,
    }
    impl HipLexiconAnalyzer {
        pub fn new(device: i32) -> Result<Self, DomainError> {
            Ok(Self { ctx: HipCtx::new(device).map_err(fail)? })
        }
    }

    #[async_trait]
    impl PostAnalyzer for HipLexiconAnalyzer {
        /// One PostSignal per post, aligned to input order (post_analyzer.rs:9).
        async fn analyze(&self, posts: &[SocialPost]) -> Result<Vec<PostSignal>, DomainError> {
            let (blob, offsets) = gather(posts.iter().map(|p| p.text.as_str()));
            let n = posts.len();
            let ctx = self.ctx.clone();
            // The call blocks (H2D copy, kernel, D2H copy): keep it off the async workers.  The reference impl is an
            // async fn without an await point (lexicon.rs:84-86), so the semantics are unchanged.
            tokio::task::spawn_blocking(move || {
                let (mut pol, mut spec) = (vec![0f64; n], vec![0u8; n]);
                check(unsafe {
                    ffi::oi_lexicon_analyze(ctx.raw(), blob.as_ptr(), offsets.as_ptr(), n as u64, pol.as_mut_ptr(), spec.as_mut_ptr())
                })
                .map_err(fail)?;
                Ok(pol
                    .into_iter()
                    .zip(spec)
                    .map(|(p, s)| PostSignal { polarity: Polarity::new(p), speculative: s != 0 })
                    .collect())
            })
            .await
            .map_err(|e| DomainError::SourceFailure { name: "hip-analyzer".into(), message: e.to_string() })?
        }
    }
}
