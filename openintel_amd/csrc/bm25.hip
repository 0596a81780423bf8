// bm25.hip -- blocked inverted index with precomputed BM25 impacts + term-at-a-time query.
//
// Builder-defined (the reference has no BM25).  Parameters:
//   k1 = 1.2, b = 0.75, idf_t = (float) ln(1 + (N - df + 0.5)/(df + 0.5))   (double, then rounded)
//   Kd  = k1 * ((1 - b) + b * (dl / avgdl))          every op rounded to f32
//   w   = (tf * (k1 + 1)) / (tf + Kd)                every op rounded to f32  ("impact")
//   score(d) = sum over query terms, in query order, of idf_t * w(t, d)      f32 adds from +0
// Every f32 op is written with an explicit round-to-nearest intrinsic so no flag can
// fuse or reorder it: scores are bit-identical to the CPU oracle's.
//
// Layout (round 4: TERM-major).  Postings are sorted by (term, doc) -- a term's posting list is one contiguous
// array in doc order -- and stored as {doc_in_block u32, impact f32}, doc_in_block relative to the doc's block of
// R = 32768 consecutive ids.  Docs are also cut into WINDOWS of F = 16384 ids (two per block), and
// cell_start[term * n_windows + window] is where that (term, window) run begins; the next entry is where it
// ends, because consecutive windows of a term -- and then the next term -- follow each other in the array.  So
// the bounds of any run of consecutive windows or blocks of a term are ONE contiguous read, and a kernel that
// walks a query's blocks reads its terms' lists as contiguous streams (bm25_stream.hip).  The (block, term) run
// of the two older kernels is windows 2 blk and 2 blk + 1 of the term.
// This file's query kernel: a workgroup owns one doc block: its R f32
// accumulators live in LDS (128 KiB), each query term's run is streamed once with coalesced
// 8-byte loads, and a doc appears at most once per run, so accumulation needs no atomics and
// has a fixed order.  Candidates leave the block as 64-bit rank keys into the query's pool.
#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include "oi_device.h"
#include "oi_internal.h"

#define BM_R OI_BM25_BLOCK_DOCS
#define BM_R_LOG2 15
#define BM_THREADS 1024
#define BM_K1 1.2f
#define BM_B 0.75f

struct Posting {
    uint32_t dib;  // doc index within its block
    float impact;
};

#define BM_F_LOG2 14 // log2(OI_BM25_FINE_DOCS)
static_assert((1u << BM_F_LOG2) == OI_BM25_FINE_DOCS && OI_BM25_BLOCK_DOCS == 2 * OI_BM25_FINE_DOCS, "two windows per block");

// key = term (32 bits) | local doc (32 bits): sorted keys = every term's posting list in doc order
__device__ __forceinline__ uint64_t bm_key(uint64_t doc, uint32_t term) {
    return ((uint64_t)term << 32) | (doc & 0xFFFFFFFFull);
}

// ------------------------------------------------------------------ index build
__global__ void bm_make_keys_kernel(const uint32_t *terms, const uint64_t *offsets, uint64_t n_docs,
                                    uint32_t vocab, uint64_t *keys, uint32_t *doc_len, uint32_t *bad) {
    for (uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_docs;
         d += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lo = offsets[d], hi = offsets[d + 1];
        doc_len[d] = (uint32_t)(hi - lo);
        for (uint64_t i = lo; i < hi; ++i) {
            const uint32_t t = terms[i];
            if (t >= vocab) *bad = 1u;
            keys[i] = bm_key(d, t);
        }
    }
}

// (term, window) cell histogram of the sorted unique keys.  Entries of one cell are consecutive, so a
// wave first folds its lanes' runs (ballot of run heads) and issues ONE atomic per run and wave: the
// per-entry version serialised millions of adds on the cells of frequent terms (670 ms at 10M docs).
__global__ void bm_count_kernel(const uint64_t *uniq, uint64_t n, uint32_t n_win, uint32_t *cell_count) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(uint64_t)63; i0 < n;
         i0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = i0 + lane;
        const bool valid = i < n;
        const uint64_t cell_key = valid ? (uniq[i] >> BM_F_LOG2) : ~0ull; // term << 18 | window
        const uint64_t prev = __shfl_up(cell_key, 1, OI_WAVE);
        const bool head = valid && (lane == 0 || prev != cell_key);
        const unsigned long long heads = __ballot(head);
        const unsigned long long valids = __ballot(valid);
        if (head) {
            // run = lanes [lane, next head or first invalid lane)
            const unsigned long long above = lane == 63 ? 0ull : ((heads | ~valids) >> (lane + 1));
            const uint32_t run = above ? (uint32_t)__builtin_ctzll(above) + 1u : 64u - lane;
            const uint64_t term = cell_key >> (32 - BM_F_LOG2);
            const uint32_t win = (uint32_t)cell_key & ((1u << (32 - BM_F_LOG2)) - 1u);
            atomicAdd(&cell_count[term * n_win + win], run);
        }
    }
}

// df[t] = length of term t's posting list, read off the SCANNED cell array
__global__ void bm_df_kernel(const uint32_t *cell_start, uint32_t n_win, uint32_t vocab, uint32_t *df) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < vocab; t += gridDim.x * blockDim.x)
        df[t] = cell_start[(uint64_t)(t + 1) * n_win] - cell_start[(uint64_t)t * n_win];
}

__global__ void bm_impact_kernel(const uint64_t *uniq, const uint32_t *tf, uint64_t n,
                                 const uint32_t *doc_len, float avgdl, Posting *postings) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = uniq[i];
        const uint64_t doc = k & 0xFFFFFFFFull;
        const uint32_t dib = (uint32_t)(doc & (BM_R - 1));
        const float ratio = __fdiv_rn((float)doc_len[doc], avgdl);
        const float kd = __fmul_rn(BM_K1, __fadd_rn(1.0f - BM_B, __fmul_rn(BM_B, ratio)));
        const float ftf = (float)tf[i];
        Posting p;
        p.dib = dib;
        p.impact = __fdiv_rn(__fmul_rn(ftf, BM_K1 + 1.0f), __fadd_rn(ftf, kd));
        postings[i] = p;
    }
}

// Per-term impact floors (round 4).  floor[t][j] = the bits of a LOWER BOUND of term t's r_j-th largest posting impact,
// r = 16, 64, 256, 1024 (0.0f when the term has fewer postings): the lower edge of the 22-bit bin (two 11-bit histogram passes
// over the term's postings) that holds it.  What it is for: at least r_j docs hold term t with an impact >= floor, every one of
// them scores >= idf_t * floor for any query that contains t (the other terms add >= 0, f32 sums of non-negative values never
// fall below an addend), so max_t fl(idf_t * floor[t][j]) with r_j >= depth is a valid lower bound of the query's depth-th
// best BM25 score BEFORE a single posting is read -- the stream kernel starts with that threshold instead of a threshold-less
// first phase (bm25_stream.hip).  One 256-thread workgroup per term.
__device__ void bm_floor_find(const uint32_t *hist, uint32_t kk, uint32_t *bin_out, uint32_t *above_out) {
    // wave 0: the bin of hist[0..2048) holding the kk-th entry counted from the top; lane l owns bins 2047 - 32 l - (0..31)
    const uint32_t lane = threadIdx.x & 63;
    uint32_t mine = 0;
    for (uint32_t i = 0; i < 32; ++i) mine += hist[2047u - (lane * 32u + i)];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, OI_WAVE);
        if ((int)lane >= o) incl += v;
    }
    const unsigned long long ball = __ballot(incl >= kk);
    const uint32_t owner = ball ? (uint32_t)__builtin_ctzll(ball) : 63u;
    if (lane == owner) {
        uint32_t cum = incl - mine, b = 2047u - lane * 32u;
        for (uint32_t i = 0; i < 32; ++i, --b) {
            const uint32_t c = hist[b];
            if (cum + c >= kk || i == 31) break;
            cum += c;
        }
        *bin_out = b;
        *above_out = cum;
    }
}
__global__ __launch_bounds__(256) void bm_term_floor_kernel(const Posting *__restrict__ postings, const uint32_t *__restrict__ cell_start,
                                                            uint32_t n_win, uint32_t vocab, uint32_t *__restrict__ floors) {
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t s_bin[OI_BM25_FLOOR_RANKS], s_above[OI_BM25_FLOOR_RANKS], s_bin2, s_above2;
    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    if (t >= vocab) return;
    const uint32_t lo = cell_start[(uint64_t)t * n_win], hi = cell_start[(uint64_t)(t + 1) * n_win];
    const uint32_t df = hi - lo;
    const uint32_t ranks[OI_BM25_FLOOR_RANKS] = {16u, 64u, 256u, 1024u};
    if (df < ranks[0]) return; // (the table is zeroed before the launch)
    // ---- pass 1: the leading 11 bits (sign 0, exponent, two bits of mantissa) of every impact
    for (uint32_t i = tid; i < 2048; i += 256) hist[i] = 0;
    __syncthreads();
    for (uint32_t i = lo + tid; i < hi; i += 256) atomicAdd(&hist[__float_as_uint(postings[i].impact) >> 21], 1u);
    __syncthreads();
    for (int j = 0; j < OI_BM25_FLOOR_RANKS; ++j) { // (sequential: bm_floor_find is wave 0's, with its own results)
        if (tid < 64 && df >= ranks[j]) bm_floor_find(hist, ranks[j], &s_bin[j], &s_above[j]);
        __syncthreads();
    }
    // ---- pass 2 per rank: the next 11 bits among the impacts of its leading bin
    for (int j = 0; j < OI_BM25_FLOOR_RANKS; ++j) {
        if (df < ranks[j]) break; // uniform
        const uint32_t d1 = s_bin[j], kk = ranks[j] - s_above[j];
        for (uint32_t i = tid; i < 2048; i += 256) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = lo + tid; i < hi; i += 256) {
            const uint32_t b = __float_as_uint(postings[i].impact);
            if ((b >> 21) == d1) atomicAdd(&hist[(b >> 10) & 2047u], 1u);
        }
        __syncthreads();
        if (tid < 64) bm_floor_find(hist, kk, &s_bin2, &s_above2);
        __syncthreads();
        if (tid == 0) floors[(uint64_t)t * OI_BM25_FLOOR_RANKS + j] = (d1 << 21) | (s_bin2 << 10); // the bin's lower edge
        __syncthreads();
    }
}

int oi_bm25_stage_forward(oi_index *idx, const uint32_t *d_terms, const uint64_t *d_offsets) {
    oi_ctx *ctx = idx->ctx;
    hipStream_t st = ctx->stream;
    const uint64_t n = idx->n_docs;
    uint64_t total = 0;
    OI_HIP_CHECK(hipMemcpyAsync(&total, d_offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    OI_REQUIRE(total < 0xFFFFFFFFull, "bm25: %llu tokens in one shard (limit 2^32-1)", (unsigned long long)total);
    idx->total_tokens = total;
    // keep a copy of the forward index for the batch scan (4 B/token + 8 B/doc; +64 B so that the
    // scan's last aligned 16-byte group may read past the end)
    OI_CHECK(idx->fwd_terms.ensure(sizeof(uint32_t) * total + 64));
    OI_CHECK(idx->fwd_offsets.ensure(sizeof(uint64_t) * (n + 1)));
    if (total) OI_HIP_CHECK(hipMemcpyAsync(idx->fwd_terms.p, d_terms, sizeof(uint32_t) * total, hipMemcpyDeviceToDevice, st));
    OI_HIP_CHECK(hipMemcpyAsync(idx->fwd_offsets.p, d_offsets, sizeof(uint64_t) * (n + 1), hipMemcpyDeviceToDevice, st));
    idx->n_blocks = (uint32_t)((n + BM_R - 1) / BM_R);
    OI_REQUIRE(idx->n_blocks < (1u << 17), "bm25: too many doc blocks");
    idx->n_win = 2 * idx->n_blocks; // windows of OI_BM25_FINE_DOCS docs; a block is windows 2 blk, 2 blk + 1
    OI_REQUIRE((uint64_t)idx->n_win * idx->vocab < 0xFFFFFFFFull, "bm25: windows x vocab exceeds 2^32");

    OI_CHECK(idx->doc_len.ensure(sizeof(uint32_t) * (n ? n : 1)));
    OI_CHECK(idx->df_local.ensure(sizeof(uint32_t) * idx->vocab));
    OI_HIP_CHECK(hipMemsetAsync(idx->df_local.p, 0, sizeof(uint32_t) * idx->vocab, st));
    idx->n_postings = 0;
    if (total == 0) {
        if (n) OI_HIP_CHECK(hipMemsetAsync(idx->doc_len.p, 0, sizeof(uint32_t) * n, st));
        idx->forward_set = true;
        idx->finalized = false;
        return OI_OK;
    }
    DevBuf keys, keys_sorted, temp, flag, runs;
    OI_CHECK(keys.ensure(sizeof(uint64_t) * total));
    OI_CHECK(keys_sorted.ensure(sizeof(uint64_t) * total));
    OI_CHECK(flag.ensure(16));
    OI_CHECK(runs.ensure(16));
    OI_HIP_CHECK(hipMemsetAsync(flag.p, 0, 16, st));
    {
        uint64_t blocks = (n + 255) / 256;
        if (blocks > 65535) blocks = 65535;
        hipLaunchKernelGGL(bm_make_keys_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, d_terms, d_offsets, n,
                           idx->vocab, keys.as<uint64_t>(), idx->doc_len.as<uint32_t>(), flag.as<uint32_t>());
        OI_HIP_CHECK(hipGetLastError());
    }
    size_t temp_bytes = 0;
    OI_HIP_CHECK(rocprim::radix_sort_keys(nullptr, temp_bytes, keys.as<uint64_t>(), keys_sorted.as<uint64_t>(),
                                          (size_t)total, 0u, 64u, st));
    OI_CHECK(temp.ensure(temp_bytes ? temp_bytes : 16));
    OI_HIP_CHECK(rocprim::radix_sort_keys(temp.p, temp_bytes, keys.as<uint64_t>(), keys_sorted.as<uint64_t>(),
                                          (size_t)total, 0u, 64u, st));
    // run-length encode: unique (term, doc) keys, run length = term frequency
    OI_CHECK(idx->uniq_keys.ensure(sizeof(uint64_t) * total));
    OI_CHECK(idx->tf.ensure(sizeof(uint32_t) * total));
    size_t temp2 = 0;
    OI_HIP_CHECK(rocprim::run_length_encode(nullptr, temp2, keys_sorted.as<uint64_t>(), (unsigned int)total,
                                            idx->uniq_keys.as<uint64_t>(), idx->tf.as<uint32_t>(),
                                            runs.as<uint32_t>(), st));
    OI_CHECK(temp.ensure(temp2 ? temp2 : 16));
    OI_HIP_CHECK(rocprim::run_length_encode(temp.p, temp2, keys_sorted.as<uint64_t>(), (unsigned int)total,
                                            idx->uniq_keys.as<uint64_t>(), idx->tf.as<uint32_t>(),
                                            runs.as<uint32_t>(), st));
    uint32_t h_runs = 0, h_bad = 0;
    OI_HIP_CHECK(hipMemcpyAsync(&h_runs, runs.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(&h_bad, flag.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    keys.release(); keys_sorted.release(); temp.release(); flag.release(); runs.release();
    OI_REQUIRE(!h_bad, "bm25: a term id is >= vocab (%u)", idx->vocab);
    idx->n_postings = h_runs;

    // the (term, window) cell histogram, scanned into run starts; local df from it
    const uint64_t cells = (uint64_t)idx->n_win * idx->vocab;
    OI_CHECK(idx->cell_start.ensure(sizeof(uint32_t) * (cells + 1)));
    OI_HIP_CHECK(hipMemsetAsync(idx->cell_start.p, 0, sizeof(uint32_t) * (cells + 1), st));
    {
        uint64_t blocks = (idx->n_postings + 255) / 256;
        if (blocks > 65535) blocks = 65535;
        hipLaunchKernelGGL(bm_count_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, idx->uniq_keys.as<uint64_t>(),
                           idx->n_postings, idx->n_win, idx->cell_start.as<uint32_t>());
        OI_HIP_CHECK(hipGetLastError());
    }
    // exclusive scan in place -> cell_start[c] = first posting of cell c; last entry = n_postings
    size_t temp3 = 0;
    DevBuf t3;
    OI_HIP_CHECK(rocprim::exclusive_scan(nullptr, temp3, idx->cell_start.as<uint32_t>(),
                                         idx->cell_start.as<uint32_t>(), 0u, (size_t)(cells + 1),
                                         rocprim::plus<uint32_t>(), st));
    OI_CHECK(t3.ensure(temp3 ? temp3 : 16));
    OI_HIP_CHECK(rocprim::exclusive_scan(t3.p, temp3, idx->cell_start.as<uint32_t>(),
                                         idx->cell_start.as<uint32_t>(), 0u, (size_t)(cells + 1),
                                         rocprim::plus<uint32_t>(), st));
    hipLaunchKernelGGL(bm_df_kernel, dim3((idx->vocab + 255) / 256), dim3(256), 0, st, idx->cell_start.as<uint32_t>(),
                       idx->n_win, idx->vocab, idx->df_local.as<uint32_t>());
    OI_HIP_CHECK(hipGetLastError());
    OI_HIP_CHECK(hipStreamSynchronize(st));
    t3.release();
    idx->forward_set = true;
    idx->finalized = false;
    return OI_OK;
}

int oi_bm25_finalize(oi_index *idx, uint64_t global_n, uint64_t global_tokens, const uint32_t *global_df_host) {
    oi_ctx *ctx = idx->ctx;
    hipStream_t st = ctx->stream;
    OI_REQUIRE(global_n >= idx->n_docs && global_n > 0, "bm25: global_n_docs < local n_docs");
    std::vector<uint32_t> df(idx->vocab);
    if (global_df_host) memcpy(df.data(), global_df_host, sizeof(uint32_t) * idx->vocab);
    else {
        OI_HIP_CHECK(hipMemcpyAsync(df.data(), idx->df_local.p, sizeof(uint32_t) * idx->vocab,
                                    hipMemcpyDeviceToHost, st));
        OI_HIP_CHECK(hipStreamSynchronize(st));
    }
    std::vector<float> idf(idx->vocab);
    const double N = (double)global_n;
    // Every idf must be >= 0: the stream kernel's first threshold (max_t idf_t x floor_t, the impact floors below) is a valid
    // lower bound of the depth-th best score only then.  df_t <= N makes it so; a caller's global df vector that says otherwise
    // is refused here rather than silently dropping docs.
    for (uint32_t t = 0; t < idx->vocab; ++t)
        OI_REQUIRE((uint64_t)df[t] <= global_n, "bm25: df[%u] = %u exceeds global_n_docs = %llu", t, df[t], (unsigned long long)global_n);
    for (uint32_t t = 0; t < idx->vocab; ++t) {
        const double d = (double)df[t];
        idf[t] = (float)std::log(1.0 + (N - d + 0.5) / (d + 0.5));
    }
    OI_CHECK(idx->idf.ensure(sizeof(float) * idx->vocab));
    OI_HIP_CHECK(hipMemcpyAsync(idx->idf.p, idf.data(), sizeof(float) * idx->vocab, hipMemcpyHostToDevice, st));
    const float avgdl = (float)((double)global_tokens / (double)global_n);
    idx->avgdl = avgdl;
    if (idx->n_postings) {
        // (+1 KiB: the stream kernel moves postings in 1 KiB chunks, and a run's last chunk may extend past the array)
        OI_CHECK(idx->postings.ensure(sizeof(Posting) * idx->n_postings + 1024));
        uint64_t blocks = (idx->n_postings + 255) / 256;
        if (blocks > 65535) blocks = 65535;
        hipLaunchKernelGGL(bm_impact_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, idx->uniq_keys.as<uint64_t>(),
                           idx->tf.as<uint32_t>(), idx->n_postings, idx->doc_len.as<uint32_t>(), avgdl,
                           idx->postings.as<Posting>());
        OI_HIP_CHECK(hipGetLastError());
    }
    // the per-term impact floors the stream kernel's first threshold is read from
    OI_CHECK(idx->impact_floor.ensure(sizeof(uint32_t) * (size_t)idx->vocab * OI_BM25_FLOOR_RANKS));
    OI_HIP_CHECK(hipMemsetAsync(idx->impact_floor.p, 0, sizeof(uint32_t) * (size_t)idx->vocab * OI_BM25_FLOOR_RANKS, st));
    if (idx->n_postings) {
        hipLaunchKernelGGL(bm_term_floor_kernel, dim3(idx->vocab), dim3(256), 0, st, idx->postings.as<Posting>(),
                           idx->cell_start.as<uint32_t>(), idx->n_win, idx->vocab, idx->impact_floor.as<uint32_t>());
        OI_HIP_CHECK(hipGetLastError());
    }
    OI_HIP_CHECK(hipStreamSynchronize(st)); // idf vector goes out of scope
    idx->uniq_keys.release();
    idx->tf.release();
    idx->finalized = true;
    return OI_OK;
}

// ------------------------------------------------------------------ query
// Called by wave 0 (all 64 lanes).  Finds the bin holding the kk-th entry counted from the
// TOP bin down; *above = entries in higher bins.  nbins is a multiple of 64.
__device__ void bm_find_kth_from_top(const uint32_t *hist, uint32_t nbins, uint32_t kk, uint32_t *bin_out,
                                     uint32_t *above_out) {
    const uint32_t lane = threadIdx.x & 63, per = nbins >> 6;
    const uint32_t top = nbins - lane * per; // lane 0 owns the highest bins [top-per, top)
    uint32_t mine = 0;
    for (uint32_t i = 0; i < per; ++i) mine += hist[top - 1 - i];
    uint32_t incl = mine; // inclusive scan over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t v = __shfl_up(incl, o, OI_WAVE);
        if ((int)lane >= o) incl += v;
    }
    const unsigned long long ball = __ballot(incl >= kk);
    const uint32_t owner = ball ? (uint32_t)__builtin_ctzll(ball) : 63u;
    if (lane == owner) {
        uint32_t cum = incl - mine;
        uint32_t b = top - 1;
        for (uint32_t i = 0; i < per; ++i, --b) {
            const uint32_t c = hist[b];
            if (cum + c >= kk || i == per - 1) break;
            cum += c;
        }
        *bin_out = b;
        *above_out = cum;
    }
}

// Wave-aggregated append: one LDS atomic per wave instead of one per lane on the same word.
// Must be called by all active lanes of a converged region; returns the slot of lanes with pred.
__device__ __forceinline__ uint32_t bm_wave_slot(bool pred, uint32_t *counter) {
    const unsigned long long m = __ballot(pred);
    if (m == 0) return 0;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    const uint32_t leader = (uint32_t)__builtin_ctzll(m);
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
    base = __shfl(base, leader, OI_WAVE);
    return base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
}

#define BM_TB 4 // query terms handled per batch (their cell bounds and postings are fetched together)

__global__ __launch_bounds__(BM_THREADS) void bm25_block_kernel(
    const Posting *__restrict__ postings, const uint32_t *__restrict__ cell_start,
    const float *__restrict__ idf, uint32_t vocab, uint32_t n_win, uint32_t doc_id_base, uint32_t block0,
    const uint32_t *__restrict__ q_terms, const uint32_t *__restrict__ q_offsets, uint32_t n_queries,
    uint32_t depth, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride, const uint32_t *tau_keys,
    uint64_t pool_stride, uint32_t carry_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *acc = reinterpret_cast<float *>(smem);                       // BM_R
    uint32_t *hist = reinterpret_cast<uint32_t *>(acc + BM_R);          // 2048
    uint64_t *list = reinterpret_cast<uint64_t *>(hist + 2048);         // OI_MAX_DEPTH
    uint32_t *scan = reinterpret_cast<uint32_t *>(list + OI_MAX_DEPTH); // 16 wave totals + scratch
    uint32_t *sh = scan + 32;  // [0] touched [1] list_cnt [3] bin [4] above; [8..8+3*BM_TB) term info

    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t blk = block0 + blockIdx.x;
    // run of (term t, this block) = windows 2 blk, 2 blk + 1 of the term: [cell_start[c], cell_start[c + 2])
    auto cell_of = [&](uint32_t t) { return (uint64_t)t * n_win + 2u * blk; };
    const uint32_t doc0 = doc_id_base + blk * BM_R;
    // term info records: [0],[1] = the pipeline's double buffer (first batch of a query), [2] = scratch
    // for the later batches of queries with more than BM_TB terms.  Record = s[TB] e[TB] w[TB] tb te.
    constexpr uint32_t REC = 3 * BM_TB + 2;
    auto rec_s = [&](uint32_t b) { return sh + 8 + b * REC; };
    auto rec_e = [&](uint32_t b) { return sh + 8 + b * REC + BM_TB; };
    auto rec_w = [&](uint32_t b) { return reinterpret_cast<float *>(sh + 8 + b * REC + 2 * BM_TB); };
    auto rec_t = [&](uint32_t b) { return sh + 8 + b * REC + 3 * BM_TB; };

    for (uint32_t i = tid; i < BM_R / 4; i += BM_THREADS) reinterpret_cast<float4 *>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- software pipeline over this workgroup's queries q0, q0+G, q0+2G, ...:
    //   stage A (two queries ahead): the first batch's term ids -> (block, term) bounds and idf, into
    //            registers of threads 0..TB-1 (three dependent loads);
    //   stage B (one query ahead): the heads of those runs, one posting per thread and term;
    //   stage C: accumulate / emit the current query out of registers and LDS.
    // A (block, query) task is ~5 dependent memory steps; unpipelined they dominated the kernel.
    const uint32_t G = gridDim.y;
    uint32_t r_s = 0, r_e = 0, r_tb = 0, r_te = 0;
    float r_w = 0.f;
    auto stage_a = [&](uint32_t qa) { // threads 0..TB-1 (+ everyone gets tb/te of their own copy)
        r_s = r_e = r_tb = r_te = 0; r_w = 0.f;
        if (tid < BM_TB && qa < n_queries) {
            r_tb = q_offsets[qa]; r_te = q_offsets[qa + 1];
            if (r_tb + tid < r_te) {
                const uint32_t t = q_terms[r_tb + tid];
                if (t < vocab) { r_s = cell_start[cell_of(t)]; r_e = cell_start[cell_of(t) + 2]; r_w = idf[t]; }
            }
        }
    };
    auto store_a = [&](uint32_t b) {
        if (tid < BM_TB) {
            rec_s(b)[tid] = r_s; rec_e(b)[tid] = r_e; rec_w(b)[tid] = r_w;
            if (tid == 0) { rec_t(b)[0] = r_tb; rec_t(b)[1] = r_te; }
        }
    };
    auto stage_b = [&](uint32_t b, Posting (&p)[BM_TB]) {
#pragma unroll
        for (int j = 0; j < BM_TB; ++j) {
            p[j].dib = 0xFFFFFFFFu;
            p[j].impact = 0.f;
            if (rec_s(b)[j] + tid < rec_e(b)[j]) p[j] = postings[rec_s(b)[j] + tid];
        }
    };
    Posting pr[BM_TB], pr_nxt[BM_TB];
    {
        stage_a(blockIdx.y);
        __syncthreads();
        store_a(0);
        __syncthreads();
        stage_b(0, pr);
        stage_a(blockIdx.y + G);
    }

    uint32_t it = 0;
    for (uint32_t q = blockIdx.y; q < n_queries; q += G, ++it) {
        const uint32_t cb = it & 1u; // record holding THIS query's first batch
        __syncthreads();             // everyone is done with the previous query (record cb^1, sh, list)
        store_a(cb ^ 1u);            // first batch of query q+G
        if (tid < 8) sh[tid] = 0;
        __syncthreads();
        stage_b(cb ^ 1u, pr_nxt);    // in flight while this query is processed
        stage_a(q + 2 * G);          // dto.
        const uint32_t t_begin = rec_t(cb)[0], t_end = rec_t(cb)[1];
        const uint32_t tau = tau_keys ? tau_keys[q] : 0u;
        const bool single_batch = t_end - t_begin <= BM_TB; // then `pr` holds every run's head
        uint32_t *t_s = rec_s(cb), *t_e = rec_e(cb);
        float *t_w = rec_w(cb);
        // ---- accumulate: term by term (fixed order; a doc occurs at most once per run, so no two
        // lanes touch the same accumulator)
        uint32_t fresh = 0; // docs this thread touched first (summed per wave, then ONE LDS atomic per wave:
                            // a per-thread atomic on one LDS word serialised 4096 times per task)
        for (uint32_t tb = t_begin; tb < t_end; tb += BM_TB) {
            if (tb != t_begin) { // later batches of a long query: not pipelined
                t_s = rec_s(2); t_e = rec_e(2); t_w = rec_w(2);
                __syncthreads();
                if (tid < BM_TB) {
                    uint32_t s0 = 0, e0 = 0;
                    float w0 = 0.f;
                    if (tb + tid < t_end) {
                        const uint32_t t = q_terms[tb + tid];
                        if (t < vocab) { s0 = cell_start[cell_of(t)]; e0 = cell_start[cell_of(t) + 2]; w0 = idf[t]; }
                    }
                    t_s[tid] = s0; t_e[tid] = e0; t_w[tid] = w0;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < BM_TB; ++j) {
                    pr[j].dib = 0xFFFFFFFFu;
                    if (t_s[j] + tid < t_e[j]) pr[j] = postings[t_s[j] + tid];
                }
            }
#pragma unroll
            for (int j = 0; j < BM_TB; ++j) {
                const uint32_t s = t_s[j], e = t_e[j];
                if (s == e) continue; // uniform
                const float w = t_w[j];
                if (pr[j].dib != 0xFFFFFFFFu) {
                    const float old = acc[pr[j].dib];
                    acc[pr[j].dib] = __fadd_rn(old, __fmul_rn(w, pr[j].impact));
                    fresh += old == 0.0f;
                }
                for (uint32_t i = s + BM_THREADS + tid; i < e; i += BM_THREADS) {
                    const Posting p = postings[i];
                    const float old = acc[p.dib];
                    acc[p.dib] = __fadd_rn(old, __fmul_rn(w, p.impact));
                    fresh += old == 0.0f;
                }
                __syncthreads();
            }
        }
        fresh = oi_wave_sum(fresh);
        if (lane == 0 && fresh) atomicAdd(&sh[0], fresh);
        __syncthreads();
        const uint32_t touched = sh[0];
        if (touched == 0) {
#pragma unroll
            for (int j = 0; j < BM_TB; ++j) pr[j] = pr_nxt[j];
            continue;
        }

        // Walk every run of the query again, run by run (a doc occurs once per run, so within a run
        // no two lanes touch the same accumulator; the barrier orders the runs).
        auto walk = [&](auto &&visit) {
            for (uint32_t tb = t_begin; tb < t_end; tb += BM_TB) {
                if (!single_batch) {
                    t_s = rec_s(2); t_e = rec_e(2);
                    __syncthreads();
                    if (tid < BM_TB) {
                        uint32_t s0 = 0, e0 = 0;
                        if (tb + tid < t_end) {
                            const uint32_t t = q_terms[tb + tid];
                            if (t < vocab) { s0 = cell_start[cell_of(t)]; e0 = cell_start[cell_of(t) + 2]; }
                        }
                        t_s[tid] = s0; t_e[tid] = e0;
                    }
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < BM_TB; ++j) {
                        pr[j].dib = 0xFFFFFFFFu;
                        if (t_s[j] + tid < t_e[j]) pr[j] = postings[t_s[j] + tid];
                    }
                }
#pragma unroll
                for (int j = 0; j < BM_TB; ++j) {
                    const uint32_t s = t_s[j], e = t_e[j];
                    if (s == e) continue;
                    visit(pr[j].dib); // every lane calls (wave-aggregated appends inside); sentinel = no posting
                    for (uint32_t i0 = s + BM_THREADS; i0 < e; i0 += BM_THREADS) // uniform trip count
                        visit(i0 + tid < e ? postings[i0 + tid].dib : 0xFFFFFFFFu);
                    __syncthreads();
                }
            }
        };
        // emit + clear: the first run to reach a doc takes it (|acc|: a marking walk may have negated it)
        auto emit_visit = [&](uint32_t dib) { // dib == 0xFFFFFFFF: this lane has no posting in this step
            float v = 0.0f;
            if (dib != 0xFFFFFFFFu) {
                v = fabsf(acc[dib]);
                if (v != 0.0f) acc[dib] = 0.0f;
            }
            const bool keep = v != 0.0f && oi_f32_key(v) >= tau;
            const uint32_t slot = bm_wave_slot(keep, &sh[1]);
            if (keep) list[slot] = oi_rank_key(v, doc0 + dib);
        };
        bool dense = touched > depth;
        if (dense && tau != 0u) {
            // Many docs touched, but a threshold is known: count the docs at or above it (marking each
            // visited accumulator by its sign so that a doc is counted once); they almost always fit.
            walk([&](uint32_t dib) {
                bool hit = false;
                if (dib != 0xFFFFFFFFu) {
                    const float v = acc[dib];
                    if (v > 0.0f) {
                        acc[dib] = -v;
                        hit = oi_f32_key(v) >= tau;
                    }
                }
                (void)bm_wave_slot(hit, &sh[5]);
            });
            dense = sh[5] > depth;
        }
        if (!dense) {
            walk(emit_visit);
        } else {
            // ---- dense: exact local top-`depth` by radix select over the LDS accumulators.
            // Scores are > 0, so their bit patterns order like the floats.
            uint32_t kk = depth, prefix = 0;
            // pass A: bits 31..21, pass B: bits 20..10, pass C: bits 9..0
            for (int pass = 0; pass < 3; ++pass) {
                const uint32_t nb = pass == 2 ? 1024u : 2048u;
                for (uint32_t i = tid; i < 2048; i += BM_THREADS) hist[i] = 0;
                __syncthreads();
#pragma unroll 4
                for (uint32_t i = tid; i < BM_R; i += BM_THREADS) {
                    const uint32_t bits = __float_as_uint(acc[i]) & 0x7FFFFFFFu;
                    if (bits == 0) continue;
                    if (pass == 0) atomicAdd(&hist[bits >> 21], 1u);
                    else if (pass == 1) { if ((bits >> 21) == prefix) atomicAdd(&hist[(bits >> 10) & 2047u], 1u); }
                    else { if ((bits >> 10) == prefix) atomicAdd(&hist[bits & 1023u], 1u); }
                }
                __syncthreads();
                if (wv == 0) bm_find_kth_from_top(hist, nb, kk, &sh[3], &sh[4]);
                __syncthreads();
                prefix = pass == 2 ? ((prefix << 10) | sh[3]) : ((prefix << 11) | sh[3]);
                kk -= sh[4];
                __syncthreads();
            }
            const uint32_t T = prefix;   // bits of the depth-th largest score
            const uint32_t n_ties = kk;  // how many docs scoring exactly T to keep: lowest ids first
            // every thread owns 32 consecutive docs so that ties are taken in doc order
            const uint32_t base = tid * (BM_R / BM_THREADS);
            uint32_t eq = 0;
#pragma unroll 4
            for (uint32_t i = 0; i < BM_R / BM_THREADS; ++i) eq += (__float_as_uint(acc[base + i]) & 0x7FFFFFFFu) == T;
            uint32_t incl = eq;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t v = __shfl_up(incl, o, OI_WAVE);
                if ((int)lane >= o) incl += v;
            }
            if (lane == 63) scan[wv] = incl;
            __syncthreads();
            uint32_t before = incl - eq;
            for (uint32_t w2 = 0; w2 < wv; ++w2) before += scan[w2];
#pragma unroll 2
            for (uint32_t i = 0; i < BM_R / BM_THREADS; ++i) {
                const float v = fabsf(acc[base + i]);
                const uint32_t bits = __float_as_uint(v);
                bool take = bits > T;
                if (bits == T) { take = before < n_ties; ++before; }
                const bool keep = take && oi_f32_key(v) >= tau;
                const uint32_t slot = bm_wave_slot(keep, &sh[1]);
                if (keep) list[slot] = oi_rank_key(v, doc0 + base + i);
                acc[base + i] = 0.0f;
            }
            __syncthreads();
        }
        // ---- this block's candidates go to ITS segment of the query's pool (<= depth entries, no atomics)
        const uint32_t cnt = sh[1];
        uint64_t *seg = pools + (uint64_t)q * pool_stride + carry_cap + (uint64_t)blk * depth;
        for (uint32_t i = tid; i < cnt; i += BM_THREADS) seg[i] = list[i];
        if (tid == 0) seg_cnt[(uint64_t)q * seg_cnt_stride + blk] = cnt;
#pragma unroll
        for (int j = 0; j < BM_TB; ++j) pr[j] = pr_nxt[j];
    }
}

#define BM_SMEM (BM_R * 4 + 2048 * 4 + OI_MAX_DEPTH * 8 + 32 * 4 + (8 + 3 * (3 * BM_TB + 2)) * 4)

int oi_launch_bm25(oi_index *idx, const uint32_t *d_q_terms, const uint32_t *d_q_offsets,
                   uint32_t n_queries, uint32_t depth, const PoolView &pool, uint32_t block_begin,
                   uint32_t block_end) {
    oi_ctx *ctx = idx->ctx;
    if (n_queries == 0 || idx->n_postings == 0 || idx->n_blocks == 0 || block_end <= block_begin) return OI_OK;
    const uint32_t nb = block_end - block_begin;
    OI_REQUIRE(depth >= 1 && depth <= OI_MAX_DEPTH, "bm25: depth=%u outside [1,%u]", depth, OI_MAX_DEPTH);
    OI_REQUIRE(pool.seg_cap == depth && pool.n_segs == idx->n_blocks && pool.n_segs <= pool.seg_cnt_stride &&
                   pool.carry_cap + (uint64_t)pool.n_segs * depth <= pool.stride,
               "bm25: pool geometry mismatch");
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(bm25_block_kernel), (size_t)(BM_SMEM)));
    // one workgroup per CU is resident (128 KiB of LDS); split the batch so the grid has
    // about 4 workgroups per CU when the corpus has few blocks
    uint32_t ysplit = (uint32_t)((4ull * ctx->num_cus + nb - 1) / nb);
    if (ysplit < 1) ysplit = 1;
    if (ysplit > n_queries) ysplit = n_queries;
    ProfScope ps(ctx, "bm25");
    hipLaunchKernelGGL(bm25_block_kernel, dim3(nb, ysplit), dim3(BM_THREADS), BM_SMEM, ctx->stream,
                       idx->postings.as<Posting>(), idx->cell_start.as<uint32_t>(), idx->idf.as<float>(),
                       idx->vocab, idx->n_win, idx->doc_id_base, block_begin, d_q_terms, d_q_offsets, n_queries, depth,
                       pool.keys, pool.seg_cnt, pool.seg_cnt_stride, pool.tau_keys, pool.stride, pool.carry_cap);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}
