// bm25_stream.hip -- term-at-a-time BM25 as a STREAM: every wave walks a contiguous, weight-balanced range of
// (query, doc block) tasks and keeps the postings of the chunks ahead of it in flight through its own LDS ring.
//
// Builder-defined like the rest of the retrieval path (the reference has no BM25).  Same index
// and the same arithmetic as bm25.hip / bm25_wave.hip (score(d) = sum over the query's terms IN QUERY ORDER of
// idf_t * impact(t, d), every f32 op an explicit round-to-nearest intrinsic): the lists are bit-identical to the
// oracle's and to the other BM25 kernels' (tests/test_gpu_parity.py).
//
// Why.  bm25_wave_kernel gives a (block, query) task of ~730 postings to one wave and measured
// 0.137 ms per 64-query batch at 10M docs = 10 % of the HBM roof: ~4x the instructions the algorithm needs (group
// descriptors, 106 SGPRs, a compare-and-swap table for the docs that occur in several runs) and ~10-20 DEPENDENT
// memory round trips per 5.8 KB task (bounds -> postings -> passes), none of them overlapped with the next task.
// Two recodings of the same shape (bm25_lane, bm25_rep; round 3) came out slower.  This kernel changes the shape:
//
//   index     TERM-major (bm25.hip): a term's posting list is contiguous in doc order and cell_start[term][window]
//             bounds its 16384-doc windows, so a wave that walks consecutive blocks of one query reads each of the
//             query's lists as ONE contiguous stream, and the bounds of a block are three consecutive words;
//   tasks     the (query, block) tasks of a launch, query-major, are cut into G contiguous ranges of equal WEIGHT
//             (postings + a fixed cost per block; bm25_plan_kernel), one per wave: static, balanced, and a wave
//             stays inside one or two queries, so term ids / idf / list bases are loaded once per query;
//   producer  each wave runs a scalar state machine over its range -- task, window, pass, run, chunk -- that issues
//             one LDS-DMA (buffer_load_dwordx4 ... lds: 128 postings = 1 KiB, no VGPRs) per chunk into a ring of
//             BS_RING slots and pushes a descriptor into a FIFO held in the lanes of three VGPRs
//             (a lane select / v_readlane).  The producer runs BS_RING chunks ahead of the consumer ACROSS windows
//             and tasks, and the next task's bounds are themselves prefetched by LDS-DMA: in steady state
//             nothing waits for HBM.  Order is kept with counted s_waitcnt vmcnt(N) (a chunk's wait = the
//             number of DMAs issued after it; stores are left out of the count, which errs on the safe side);
//   consumer  pops descriptors in order.  Pass A of a window: every posting ORs its doc's bit into a wave-private
//             `seen` map (returning ds_or); a bit that was already set ORs the `multi` map.  Sweep: per-word ranks of
//             the multi map (one DPP scan), seen cleared.  Pass B (the same chunks streamed again: L2 hits): a
//             posting whose doc is not in multi is a finished score (+0) + idf * impact and goes to the threshold
//             test; a multi doc's postings are added in query order into acc[rank(doc)] -- a perfect hash from
//             the sweep, no table, no compare-and-swap (a run lists a doc once and a wave's LDS operations execute
//             in order: no two lanes ever update one accumulator).  Then the multi docs are emitted and the maps
//             cleared.  More than BS_CAP multi docs in a window: further rounds over rank ranges re-read the
//             window's runs from global memory (exact for any data; never taken on the bench's data);
//   emit      kept keys are staged in LDS and leave in 64-key stores into the task's pool segment.  Segments are
//             SMALL and fixed (phase 1: 4096 keys, phase 2: depth + 256), not one slot per doc of the block: a
//             segment that would overflow is PRUNED in place to its top `depth` keys by a wave-local radix select
//             (only a segment's top `depth` can reach the global top `depth`), which also raises the wave's local
//             threshold.  Exact for any data, and the BM25 pool is ~0.2 GB instead of 5.1 GB at 64 queries x 10M docs.
//
// HBM-bound by construction: 8 B per posting (read once from HBM, once more from L2) + 12 B per (block, term).
#include <algorithm>

#include "oi_device.h"
#include "oi_internal.h"

#define BS_BLOCK OI_BM25_BLOCK_DOCS
#define BS_FINE OI_BM25_FINE_DOCS
#define BS_WPB 2          // waves per workgroup (they never meet after the prologue)
#define BS_RING 8         // 1 KiB slots per wave
#define BS_CAP 512u       // multi-doc accumulators per window and round
#define BS_STAGE 192u     // staged keys (a chunk appends <= 128 to < 64 left over)
#define BS_MAX_Q 128u     // queries per pass
#define BS_FIX_COST 160u  // weight of a task beside its postings (windows, sweeps, run switches), in postings

typedef uint32_t bs_u32x4 __attribute__((ext_vector_type(4)));

struct BsArgs {
    const uint2 *postings;     // {doc_in_block, impact bits}, term-major
    uint64_t n_postings;
    const uint32_t *cells;     // [vocab * n_win + 1]
    const float *idf;
    const uint32_t *q_terms, *q_offsets;
    const uint32_t *unit;      // [nq] weight of one block of query r (plan)
    const uint64_t *cum;       // [nq + 1] exclusive prefix of unit
    uint64_t *pools;
    uint32_t *seg_cnt;
    const uint32_t *tau_keys;  // may be null
    uint32_t *overflow;
    uint64_t pool_stride;
    uint32_t n_win, vocab, doc_id_base, block0, nbh, q_begin, nq;
    uint32_t seg_cnt_stride, carry_cap, seg_cap, depth;
};

__device__ __forceinline__ uint32_t bs_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ uint32_t bs_rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t bs_readlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// One chunk: 64 lanes x 16 B from base + 16 * lane into the 1 KiB at lds_dst.  Bytes past `bytes_left` read as zero.
// hipcc does not see this load: it is ordered by bs_wait_vm().
__device__ __forceinline__ void bs_dma_chunk(const void *base, uint32_t bytes_left, uint32_t lds_dst, uint32_t lane16) {
    const uint64_t b = (uint64_t)base;
    bs_u32x4 srd;
    srd[0] = bs_rfl((uint32_t)b);
    srd[1] = bs_rfl((uint32_t)(b >> 32) & 0xFFFFu); // stride 0
    srd[2] = bs_rfl(bytes_left);
    srd[3] = 0x00020000u;
    uint32_t keep;
    const uint32_t d = bs_rfl(lds_dst);
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane16), "s"(srd), "s"(d)
        : "memory");
}
// One dword per lane from each lane's own address into lds_dst + 4 * lane (the bounds of the next task).
__device__ __forceinline__ void bs_dma_word(const uint32_t *gptr, uint32_t lds_dst) {
    uint32_t keep;
    const uint32_t d = bs_rfl(lds_dst);
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dword %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gptr), "s"(d)
        : "memory");
}
// Wait until at most y of this wave's vector-memory operations are outstanding (they complete in issue order, so
// "the y issued after the one I need" may stay in flight).  A smaller y than the truth only waits longer.
__device__ __forceinline__ void bs_wait_vm(uint32_t y) {
    switch (bs_rfl(y)) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}
// inclusive scan over the 64 lanes (the DPP sequence LLVM's atomic optimizer emits for gfx9: no LDS traffic)
__device__ __forceinline__ uint32_t bs_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return v;
}

enum { BS_CHUNK = 0, BS_END_A = 1, BS_END_B = 2, BS_START_TASK = 3, BS_END_TASK = 4 };

// ---------------------------------------------------------------------------------------------------------------
// Per-wave LDS (W docs per window, WORDS = W / 32):
//   seen u32[WORDS] | multi u32[WORDS] | pref u16[WORDS] | acc f32[BS_CAP] | accdoc u32[BS_CAP] | stage u64[BS_STAGE] |
//   desc u32[2][3][64] | (pad to 1 KiB) | ring [BS_RING][1 KiB]
template <int W>
struct BsLds {
    static constexpr uint32_t WORDS = W / 32;
    static constexpr uint32_t OFF_SEEN = 0, OFF_MULTI = WORDS * 4, OFF_PREF = WORDS * 8, OFF_ACC = WORDS * 10,
                              OFF_ACCDOC = OFF_ACC + BS_CAP * 4, OFF_STAGE = OFF_ACCDOC + BS_CAP * 4,
                              OFF_DESC = OFF_STAGE + BS_STAGE * 8, OFF_RING = (OFF_DESC + 2 * 3 * 256 + 1023) & ~1023u,
                              WAVE = OFF_RING + BS_RING * 1024;
    static constexpr uint32_t TOTAL = BS_WPB * WAVE + (BS_MAX_Q + 2) * 8;
};

template <int W>
__global__ __launch_bounds__(BS_WPB * 64) void bm25_stream_kernel(const BsArgs a) {
    using L = BsLds<W>;
    constexpr uint32_t WORDS = L::WORDS, WPL = WORDS / 64; // map words per lane in the sweep: 8 (W = 16384) or 16
    constexpr uint32_t NWB = BS_BLOCK / W;                  // windows per block: 2 or 1
    constexpr uint32_t WSTEP = W / BS_FINE;                 // cells per window: 1 or 2
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = bs_rfl(tid >> 6);
    unsigned char *wl = smem + wv * L::WAVE;
    uint32_t *seen = reinterpret_cast<uint32_t *>(wl + L::OFF_SEEN);
    uint32_t *multi = reinterpret_cast<uint32_t *>(wl + L::OFF_MULTI);
    uint16_t *pref = reinterpret_cast<uint16_t *>(wl + L::OFF_PREF);
    float *acc = reinterpret_cast<float *>(wl + L::OFF_ACC);
    uint32_t *accdoc = reinterpret_cast<uint32_t *>(wl + L::OFF_ACCDOC);
    uint64_t *stage = reinterpret_cast<uint64_t *>(wl + L::OFF_STAGE);
    uint32_t *desc = reinterpret_cast<uint32_t *>(wl + L::OFF_DESC);
    const uint4 *ring16 = reinterpret_cast<const uint4 *>(wl + L::OFF_RING);
    uint64_t *s_cum = reinterpret_cast<uint64_t *>(smem + BS_WPB * L::WAVE);
    const uint32_t ring_w = bs_lds_addr(wl + L::OFF_RING), desc_w = bs_lds_addr(desc);

    // ---- prologue: the plan's prefix in LDS, this wave's maps and accumulators zero
    for (uint32_t i = tid; i <= a.nq; i += BS_WPB * 64) s_cum[i] = a.cum[i];
    {
        uint4 *z = reinterpret_cast<uint4 *>(wl);
        for (uint32_t i = lane; i < (WORDS * 8) / 16; i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u); // seen | multi
        for (uint32_t i = lane; i < BS_CAP; i += 64) acc[i] = 0.0f;
    }
    __syncthreads(); // the only barrier

    // ---- this wave's range of the query-major task sequence: tasks whose START lies in [P0, P1) of the weight axis
    const uint32_t nq = a.nq, nbh = a.nbh;
    const uint32_t G = gridDim.x * BS_WPB, w = blockIdx.x * BS_WPB + wv;
    const uint64_t total = s_cum[nq] * nbh;
    auto locate = [&](uint64_t P, uint32_t &r_out, uint32_t &b_out) { // first task (r, b) with start >= P
        if (P >= total) { r_out = nq; b_out = 0; return; }
        uint32_t lo = 0, hi = nq; // s_cum[lo] * nbh <= P < s_cum[hi] * nbh
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_cum[mid] * nbh <= P) lo = mid; else hi = mid;
        }
        const uint64_t u = s_cum[lo + 1] - s_cum[lo], off = P - s_cum[lo] * nbh;
        uint32_t b = (uint32_t)((off + u - 1) / u);
        if (b >= nbh) { ++lo; b = 0; }
        r_out = lo; b_out = b;
    };
    uint32_t r0, b0, r1, b1;
    locate(total / G * w + total % G * w / G, r0, b0);
    if (w + 1 == G) { r1 = nq; b1 = 0; } else locate(total / G * (w + 1) + total % G * (w + 1) / G, r1, b1);
    r0 = bs_rfl(r0); b0 = bs_rfl(b0); r1 = bs_rfl(r1); b1 = bs_rfl(b1);
    if (r0 > r1 || (r0 == r1 && b0 >= b1)) return;

    const uint32_t lane16 = lane * 16;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // ================================================================ producer state (all wave-uniform)
    uint32_t p_r = r0, p_blk = b0, p_blk_hi = 0;      // current task; blocks [.., p_blk_hi) of query p_r are mine
    uint32_t p_T = 0, p_tb = 0, p_page = 0xFFFFFFFFu; // the query's terms; which 64 of its runs the lanes describe
    uint32_t d_cb = 0;                                // lane j: cell index of (term j of the page, window 0); 0 without a run
    bool d_ok = false;                                // lane j: the page has a term j and it is inside the vocabulary
    float d_idf = 0.f;                                // lane j: idf of that term
    uint32_t c0 = 0, c1 = 0, c2 = 0;                  // lane j: cell_start words of the current block (windows 0, 1, end)
    bool p_done = false, p_started = false, p_need_task = true, p_query_loaded = false;
    uint32_t p_win = 0, p_pass = 0, p_j = 0, p_pos = 0, p_end = 0, p_chunks = 0;
    float p_idf = 0.f;
    uint32_t p_pf_blk = 0xFFFFFFFFu, p_pf_buf = 0, p_pf_seq = 0; // prefetched bounds: of which block, where, issued when
    uint32_t vseq = 0;      // asm vector-memory operations issued so far (chunk and bounds DMAs)
    uint32_t n_issued = 0;  // chunk DMAs issued (ring slot = n_issued % BS_RING)
    uint32_t n_consumed = 0;
    uint32_t f0 = 0, f1 = 0, f2 = 0, f_head = 0, f_tail = 0; // FIFO of descriptors in the lanes of three VGPRs
    auto push = [&](uint32_t w0, uint32_t w1, uint32_t sq) {
        const bool here = lane == (f_tail & 63u); // (this clang has no writelane builtin: one compare, three selects)
        f0 = here ? w0 : f0;
        f1 = here ? w1 : f1;
        f2 = here ? sq : f2;
        ++f_tail;
    };
    // cell words of block `blk` for the lanes' runs, by LDS-DMA into bounds buffer `buf` (no VGPR destination)
    auto issue_bounds = [&](uint32_t blk, uint32_t buf) {
        const uint32_t *src = a.cells + d_cb + 2u * (a.block0 + blk); // (lanes without a run: d_cb = 0, any valid word)
        bs_dma_word(src, desc_w + (buf * 3u + 0u) * 256u);
        bs_dma_word(src + 1, desc_w + (buf * 3u + 1u) * 256u);
        bs_dma_word(src + 2, desc_w + (buf * 3u + 2u) * 256u);
        vseq += 3;
    };
    auto read_bounds = [&](uint32_t buf) {
        c0 = desc[(buf * 3u + 0u) * 64u + lane];
        c1 = desc[(buf * 3u + 1u) * 64u + lane];
        c2 = desc[(buf * 3u + 2u) * 64u + lane];
    };
    // the lanes' runs = terms [64 page, 64 page + 64) of query p_r (plain loads: once per query in the common case)
    auto load_page = [&](uint32_t page) {
        const uint32_t j = 64u * page + lane;
        uint32_t term = 0xFFFFFFFFu;
        if (j < p_T) term = a.q_terms[p_tb + j];
        d_ok = term < a.vocab;
        d_cb = d_ok ? term * a.n_win : 0u;
        d_idf = d_ok ? a.idf[term] : 0.f;
        p_page = page;
    };
    auto fix_bounds = [&]() { // lanes without a run: an empty one
        c0 = d_ok ? c0 : 0u; c1 = d_ok ? c1 : 0u; c2 = d_ok ? c2 : 0u;
    };
    auto bounds_now = [&](uint32_t blk) { // synchronously (query switch, page switch)
        issue_bounds(blk, 0);
        bs_wait_vm(0);
        read_bounds(0);
        fix_bounds();
        p_pf_blk = 0xFFFFFFFFu;
    };
    auto run_of = [&](uint32_t j, uint32_t win, uint32_t &s, uint32_t &e, float &wt) { // run j of the current page
        const uint32_t l = j & 63u;
        if (NWB == 1) { s = bs_readlane(c0, l); e = bs_readlane(c2, l); }
        else if (win == 0) { s = bs_readlane(c0, l); e = bs_readlane(c1, l); }
        else { s = bs_readlane(c1, l); e = bs_readlane(c2, l); }
        wt = __uint_as_float(bs_readlane(__float_as_uint(d_idf), l));
    };

    // One step of the sequence generator: pushes exactly ONE descriptor (or sets p_done).
    auto produce = [&]() {
        for (;;) {
            if (p_need_task) {
                if (p_r > r1 || (p_r == r1 && p_blk >= b1) || p_r >= nq) { p_done = true; return; }
                if (!p_query_loaded) {
                    const uint32_t q = a.q_begin + p_r;
                    p_tb = bs_rfl(a.q_offsets[q]);
                    p_T = bs_rfl(a.q_offsets[q + 1]) - p_tb;
                    p_blk_hi = p_r == r1 ? b1 : nbh;
                    load_page(0);
                    p_query_loaded = true;
                    p_pf_blk = 0xFFFFFFFFu;
                }
                if (p_page != 0u) { load_page(0); p_pf_blk = 0xFFFFFFFFu; }
                if (p_pf_blk == p_blk) { // the bounds were prefetched while the previous block was produced
                    bs_wait_vm(vseq - p_pf_seq - 3u);
                    read_bounds(p_pf_buf);
                    fix_bounds();
                } else bounds_now(p_blk);
                if (p_blk + 1 < p_blk_hi) { // the next block's bounds, in flight while this one is produced
                    p_pf_buf ^= 1u;
                    p_pf_seq = vseq;
                    issue_bounds(p_blk + 1, p_pf_buf);
                    p_pf_blk = p_blk + 1;
                } else p_pf_blk = 0xFFFFFFFFu;
                p_need_task = false;
                p_started = false;
                p_win = 0; p_pass = 0; p_j = 0xFFFFFFFFu; p_pos = p_end = 0; p_chunks = 0;
            }
            if (p_pos < p_end) { // ---- one chunk of the current run: 128 postings from an even index
                if (!p_started) { p_started = true; push(BS_START_TASK, p_r | (p_blk << 8), 0); return; }
                const uint32_t cs = p_pos & ~1u;
                const uint32_t lo = p_pos - cs, hi = p_end - cs < 128u ? p_end - cs : 128u;
                const uint64_t left = (a.n_postings - cs) * 8ull;
                const uint32_t slot = n_issued % BS_RING;
                bs_dma_chunk(a.postings + cs, left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left, ring_w + slot * 1024u, lane16);
                push(BS_CHUNK | (p_pass << 3) | (lo << 4) | (hi << 8) | (slot << 16), __float_as_uint(p_idf), vseq);
                ++vseq; ++n_issued; ++p_chunks;
                p_pos = cs + 128u;
                return;
            }
            // ---- next run of this (window, pass)
            ++p_j;
            if (p_j < p_T) {
                if ((p_j >> 6) != p_page) { load_page(p_j >> 6); bounds_now(p_blk); } // a query of more than 64 terms
                run_of(p_j, p_win, p_pos, p_end, p_idf);
                continue;
            }
            // ---- end of the pass
            if (p_pass == 0 && p_chunks) { // its postings again, as pass B
                push(BS_END_A, p_win, 0);
                p_pass = 1; p_j = 0xFFFFFFFFu; p_pos = p_end = 0;
                return;
            }
            const bool was_b = p_pass == 1;
            p_pass = 0; p_j = 0xFFFFFFFFu; p_pos = p_end = 0; p_chunks = 0;
            ++p_win;
            if (p_win == NWB) { // ---- end of the task
                p_need_task = true;
                ++p_blk;
                if (p_blk >= p_blk_hi) { ++p_r; p_blk = 0; p_query_loaded = false; }
                if (was_b) { push(BS_END_B | (1u << 3), 0, 0); return; } // (bit 3: the task ends with this window)
                if (p_started) { push(BS_END_TASK, 0, 0); return; }
                continue;
            }
            if (was_b) { push(BS_END_B, 0, 0); return; }
        }
    };

    // ================================================================ consumer state
    uint32_t t_q = 0, t_blk = 0, t_doc0 = 0, t_tau = 0, t_tau_q = 0, out_n = 0, st_n = 0, t_M = 0, t_win = 0;
    uint64_t *t_seg = a.pools;
    const uint32_t seg_cap = a.seg_cap;

    // The segment is full: keep its top `depth` keys (exact: only those can reach the global top `depth`).
    auto prune = [&]() {
        uint32_t *hist = seen; // zero and unused between a sweep and the next window's pass A
        bs_wait_vm(0);         // this wave's stores have landed
        const uint32_t n = out_n, kprime = a.depth;
        uint64_t prefix = 0;
        uint32_t kk = kprime;
        int shift = 56;
        for (;; shift -= 8) {
            for (uint32_t i = lane; i < n; i += 64) {
                const uint64_t key = __hip_atomic_load(&t_seg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (shift == 56 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
            }
            uint32_t mine = 0;
            for (int i = 0; i < 4; ++i) mine += hist[255u - (lane * 4u + i)];
            const uint32_t incl = bs_incl_scan(mine);
            const unsigned long long ball = __ballot(incl >= kk);
            const uint32_t owner = ball ? (uint32_t)__builtin_ctzll(ball) : 63u;
            uint32_t cum = incl - mine, d = 255u - lane * 4u;
            for (int i = 0; i < 3; ++i) {
                const uint32_t c = hist[d];
                if (cum + c >= kk) break;
                cum += c;
                --d;
            }
            const uint32_t bin_cnt = bs_readlane(hist[d], owner);
            d = bs_readlane(d, owner);
            cum = bs_readlane(cum, owner);
            prefix = (prefix << 8) | d;
            kk -= cum;
            for (int i = 0; i < 4; ++i) hist[lane * 4u + i] = 0u;
            if (bin_cnt == 1u || shift == 0) break;
        }
        uint32_t wr = 0; // compaction moves keys to the left of where they were read
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            uint64_t key = 0;
            if (i < n) key = __hip_atomic_load(&t_seg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool keepit = i < n && (key >> shift) >= prefix;
            const unsigned long long m = __ballot(keepit);
            if (keepit) t_seg[wr + (uint32_t)__popcll(m & lt_mask)] = key;
            wr += (uint32_t)__popcll(m);
        }
        out_n = wr; // == depth
        const uint32_t thr = (uint32_t)((prefix << shift) >> 32); // every kept key is >= prefix << shift
        t_tau = thr > t_tau ? thr : t_tau;
    };
    auto flush64 = [&]() { // 64 staged keys leave for the segment
        if (out_n + 64u > seg_cap) prune();
        t_seg[out_n + lane] = stage[lane];
        out_n += 64u;
        const uint32_t rem = st_n - 64u; // < 128
        uint64_t k0 = 0, k1 = 0;
        if (lane < rem) k0 = stage[64u + lane];
        if (64u + lane < rem) k1 = stage[128u + lane];
        if (lane < rem) stage[lane] = k0;
        if (64u + lane < rem) stage[64u + lane] = k1;
        st_n = rem;
    };
    auto emit2 = [&](bool k0, uint64_t key0, bool k1, uint64_t key1) { // all lanes call
        const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
        if (m0 | m1) {
            const uint32_t n0 = (uint32_t)__popcll(m0);
            if (k0) stage[st_n + (uint32_t)__popcll(m0 & lt_mask)] = key0;
            if (k1) stage[st_n + n0 + (uint32_t)__popcll(m1 & lt_mask)] = key1;
            st_n += n0 + (uint32_t)__popcll(m1);
            if (st_n >= 64u) flush64();
            if (st_n >= 64u) flush64();
        }
    };
    auto end_task = [&]() {
        if (st_n) {
            if (out_n + st_n > seg_cap) prune();
            if (lane < st_n) t_seg[out_n + lane] = stage[lane];
            out_n += st_n;
            st_n = 0;
        }
        if (out_n > seg_cap) { *a.overflow = 1u; out_n = seg_cap; } // bug guard
        if (lane == 0) a.seg_cnt[(uint64_t)t_q * a.seg_cnt_stride + a.block0 + t_blk] = out_n;
        out_n = 0;
    };
    // ranks [lo, lo + BS_CAP) of the window's multi docs: emit and clear their accumulators
    auto emit_multi = [&](uint32_t lo) {
        const uint32_t cnt = t_M - lo < BS_CAP ? t_M - lo : BS_CAP;
        for (uint32_t b = 0; b < cnt; b += 64) {
            const uint32_t r = b + lane;
            const bool ok = r < cnt;
            const float v = acc[ok ? r : 0u];
            const uint32_t d = accdoc[ok ? r : 0u];
            if (ok) acc[r] = 0.0f;
            emit2(ok && v > 0.0f && oi_f32_key(v) >= t_tau, oi_rank_key(v, t_doc0 + d), false, 0ull);
        }
    };
    // a multi doc's posting: acc[rank - lo] += x, in call order (ranks outside [lo, lo + BS_CAP) belong to another round)
    auto add_multi = [&](bool mul, uint32_t word, uint32_t mw, uint32_t bit, uint32_t dib, float x, uint32_t lo) {
        if (mul) {
            const uint32_t rr = (uint32_t)pref[word] + (uint32_t)__popc(mw & (bit - 1u)) - lo;
            if (rr < BS_CAP) {
                acc[rr] = __fadd_rn(acc[rr], x);
                accdoc[rr] = dib;
            }
        }
    };
    // Rounds beyond the first (more than BS_CAP multi docs in the window): the window's runs again, straight from
    // global memory, adding only the multi docs of ranks [lo, lo + BS_CAP).  Exact for any data, rare.
    auto slow_rounds = [&]() {
        const uint32_t q = a.q_begin + t_q;
        const uint32_t tb = bs_rfl(a.q_offsets[q]), te = bs_rfl(a.q_offsets[q + 1]);
        for (uint32_t lo = BS_CAP; lo < t_M; lo += BS_CAP) {
            for (uint32_t j = tb; j < te; ++j) {
                const uint32_t term = bs_rfl(a.q_terms[j]);
                if (term >= a.vocab) continue;
                const uint64_t cell = (uint64_t)term * a.n_win + 2u * (a.block0 + t_blk) + t_win * WSTEP;
                const uint32_t s = bs_rfl(a.cells[cell]), e = bs_rfl(a.cells[cell + WSTEP]);
                const float wt = a.idf[term];
                for (uint32_t i0 = s; i0 < e; i0 += 64) {
                    const bool ok = i0 + lane < e;
                    const uint2 p = a.postings[ok ? i0 + lane : s];
                    const uint32_t ix = p.x & (W - 1u), bit = 1u << (ix & 31u), mw = multi[ix >> 5];
                    add_multi(ok && (mw & bit), ix >> 5, mw, bit, p.x, __fmul_rn(wt, __uint_as_float(p.y)), lo);
                }
            }
            emit_multi(lo);
        }
    };

    // ================================================================ the stream
    for (;;) {
        while (!p_done && n_issued - n_consumed < BS_RING && f_tail - f_head < 56u) produce();
        if (f_head == f_tail) break; // (p_done, nothing queued)
        const uint32_t at = f_head & 63u;
        const uint32_t e0 = bs_readlane(f0, at), e1 = bs_readlane(f1, at), e2 = bs_readlane(f2, at);
        ++f_head;
        const uint32_t kind = e0 & 7u;
        if (kind == BS_CHUNK) {
            const uint32_t lo = (e0 >> 4) & 1u, hi = (e0 >> 8) & 255u, slot = (e0 >> 16) & 15u;
            bs_wait_vm(vseq - e2 - 1u);
            const uint4 v = ring16[slot * 64u + lane];
            const uint32_t i0 = v.x & (W - 1u), i1 = v.z & (W - 1u);
            const bool ok0 = 2u * lane >= lo && 2u * lane < hi, ok1 = 2u * lane + 1u < hi; // (2 lane + 1 >= lo always)
            const uint32_t bit0 = ok0 ? 1u << (i0 & 31u) : 0u, bit1 = ok1 ? 1u << (i1 & 31u) : 0u;
            if (!(e0 & 8u)) { // ---- pass A
                const uint32_t o0 = atomicOr(&seen[i0 >> 5], bit0);
                const uint32_t o1 = atomicOr(&seen[i1 >> 5], bit1);
                const uint32_t again0 = o0 & bit0, again1 = o1 & bit1; // the doc was in an earlier run
                if (again0) atomicOr(&multi[i0 >> 5], again0);
                if (again1) atomicOr(&multi[i1 >> 5], again1);
            } else { // ---- pass B
                const uint32_t mw0 = multi[i0 >> 5], mw1 = multi[i1 >> 5];
                const float wt = __uint_as_float(e1);
                const float x0 = __fmul_rn(wt, __uint_as_float(v.y)), x1 = __fmul_rn(wt, __uint_as_float(v.w));
                const bool mul0 = (mw0 & bit0) != 0u, mul1 = (mw1 & bit1) != 0u;
                const bool k0 = ok0 && !mul0 && x0 > 0.0f && oi_f32_key(x0) >= t_tau; // (BM25 lists hold scores > 0 only)
                const bool k1 = ok1 && !mul1 && x1 > 0.0f && oi_f32_key(x1) >= t_tau;
                emit2(k0, oi_rank_key(x0, t_doc0 + v.x), k1, oi_rank_key(x1, t_doc0 + v.z));
                if (__ballot(mul0 || mul1)) { // a lane's two postings are one run: distinct docs
                    add_multi(mul0, i0 >> 5, mw0, bit0, v.x, x0, 0u);
                    add_multi(mul1, i1 >> 5, mw1, bit1, v.z, x1, 0u);
                }
            }
            ++n_consumed;
        } else if (kind == BS_END_A) {
            // ---- sweep: ranks of the multi docs (exclusive popcount prefix per map word), seen cleared
            t_win = e1;
            const uint4 *m4 = reinterpret_cast<const uint4 *>(multi) + lane * (WPL / 4);
            uint4 *s4 = reinterpret_cast<uint4 *>(seen) + lane * (WPL / 4);
            uint32_t mwd[WPL], run = 0;
#pragma unroll
            for (uint32_t k = 0; k < WPL / 4; ++k) {
                const uint4 m = m4[k];
                mwd[4 * k] = m.x; mwd[4 * k + 1] = m.y; mwd[4 * k + 2] = m.z; mwd[4 * k + 3] = m.w;
            }
            uint32_t ex[WPL];
#pragma unroll
            for (uint32_t k = 0; k < WPL; ++k) { ex[k] = run; run += (uint32_t)__popc(mwd[k]); }
            const uint32_t incl = bs_incl_scan(run);
            const uint32_t base = incl - run;
            t_M = bs_readlane(incl, 63);
            if (t_M) {
                uint4 *p4 = reinterpret_cast<uint4 *>(pref) + lane * (WPL / 8);
#pragma unroll
                for (uint32_t k = 0; k < WPL / 8; ++k) {
                    uint4 o;
                    o.x = (base + ex[8 * k]) | ((base + ex[8 * k + 1]) << 16);
                    o.y = (base + ex[8 * k + 2]) | ((base + ex[8 * k + 3]) << 16);
                    o.z = (base + ex[8 * k + 4]) | ((base + ex[8 * k + 5]) << 16);
                    o.w = (base + ex[8 * k + 6]) | ((base + ex[8 * k + 7]) << 16);
                    p4[k] = o;
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < WPL / 4; ++k) s4[k] = make_uint4(0u, 0u, 0u, 0u);
        } else if (kind == BS_END_B) {
            if (t_M) {
                emit_multi(0u);
                if (t_M > BS_CAP) slow_rounds();
                uint4 *m4 = reinterpret_cast<uint4 *>(multi) + lane * (WPL / 4);
#pragma unroll
                for (uint32_t k = 0; k < WPL / 4; ++k) m4[k] = make_uint4(0u, 0u, 0u, 0u);
                t_M = 0;
            }
            if (e0 & 8u) end_task();
        } else if (kind == BS_START_TASK) {
            t_q = e1 & 255u;
            t_blk = e1 >> 8;
            t_doc0 = a.doc_id_base + (a.block0 + t_blk) * BS_BLOCK;
            t_tau_q = a.tau_keys ? bs_rfl(a.tau_keys[t_q]) : 0u;
            t_tau = t_tau_q;
            t_seg = a.pools + (uint64_t)t_q * a.pool_stride + a.carry_cap + (uint64_t)(a.block0 + t_blk) * seg_cap;
            out_n = 0;
        } else { // BS_END_TASK
            end_task();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Plan of a pass (one launch, before the first phase): zero the pass's pool state and weigh the queries.
//   unit[r] = postings of ONE block of query r on average (sum of its terms' local df / blocks) + a fixed cost;
//   cum = its exclusive prefix.  The stream kernel cuts cum[nq] * (blocks of the launch) into equal parts.
__global__ __launch_bounds__(256) void bm25_plan_kernel(const uint32_t *__restrict__ q_terms, const uint32_t *__restrict__ q_offsets,
                                                        const uint32_t *__restrict__ df, uint32_t vocab, uint32_t n_blocks,
                                                        uint32_t q_begin, uint32_t nq, uint32_t *unit, uint64_t *cum,
                                                        uint32_t *state, uint64_t state_words) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < state_words; i += (uint64_t)gridDim.x * 256) state[i] = 0u;
    if (blockIdx.x != 0) return;
    __shared__ uint32_t s_unit[BS_MAX_Q];
    for (uint32_t r = threadIdx.x; r < nq; r += 256) {
        uint64_t wsum = 0;
        for (uint32_t i = q_offsets[q_begin + r]; i < q_offsets[q_begin + r + 1]; ++i) {
            const uint32_t term = q_terms[i];
            wsum += term < vocab ? df[term] : 0u;
        }
        uint64_t u = wsum / (n_blocks ? n_blocks : 1u) + BS_FIX_COST;
        if (u > (1u << 24)) u = 1u << 24; // keeps cum[nq] * blocks * waves inside 64 bits
        s_unit[r] = (uint32_t)u;
        unit[r] = (uint32_t)u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t c = 0;
        for (uint32_t r = 0; r < nq; ++r) { cum[r] = c; c += s_unit[r]; }
        cum[nq] = c;
    }
}

uint32_t oi_bm25_stream_pass_queries(void) { return BS_MAX_Q; }
uint32_t oi_bm25_stream_seg_cap(uint32_t depth, bool first_phase) {
    const uint32_t c = depth + 256u;
    return first_phase ? std::max(c, 4096u) : c;
}

int oi_launch_bm25_plan(oi_index *idx, const uint32_t *d_q_terms, const uint32_t *d_q_offsets, uint32_t q_begin, uint32_t nq,
                        uint32_t *state, uint64_t state_words) {
    oi_ctx *ctx = idx->ctx;
    OI_REQUIRE(nq >= 1 && nq <= BS_MAX_Q, "bm25 (stream): %u queries in one pass (limit %u)", nq, BS_MAX_Q);
    DevBuf &pb = ctx->buf("bm25_stream_plan");
    OI_CHECK(pb.ensure(sizeof(uint64_t) * (BS_MAX_Q + 2) + sizeof(uint32_t) * BS_MAX_Q));
    uint64_t *cum = pb.as<uint64_t>();
    uint32_t *unit = reinterpret_cast<uint32_t *>(cum + BS_MAX_Q + 2);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, (state_words + 4095) / 4096));
    ProfScope ps(ctx, "bm25");
    hipLaunchKernelGGL(bm25_plan_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_q_terms, d_q_offsets,
                       idx->df_local.as<uint32_t>(), idx->vocab, idx->n_blocks, q_begin, nq, unit, cum, state, state_words);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

// Queries [q_begin, q_begin + nq) of the batch over doc blocks [block_begin, block_end); `pool` is the view of THESE nq
// queries (entry 0 = query q_begin); oi_launch_bm25_plan ran for the pass.  pool.seg_cap >= depth + 256.
int oi_launch_bm25_stream(oi_index *idx, const uint32_t *d_q_terms, const uint32_t *d_q_offsets, uint32_t q_begin,
                          uint32_t nq, uint32_t depth, const PoolView &pool, uint32_t block_begin, uint32_t block_end) {
    oi_ctx *ctx = idx->ctx;
    if (nq == 0 || idx->n_postings == 0 || idx->n_blocks == 0 || block_end <= block_begin) return OI_OK;
    OI_REQUIRE(nq <= BS_MAX_Q, "bm25 (stream): %u queries in one pass (limit %u)", nq, BS_MAX_Q);
    OI_REQUIRE(pool.seg_cap >= depth + 256u && pool.n_segs <= pool.seg_cnt_stride && block_end <= pool.seg_cnt_stride &&
                   pool.carry_cap + (uint64_t)block_end * pool.seg_cap <= pool.stride,
               "bm25 (stream): pool geometry mismatch");
    DevBuf &pb = ctx->buf("bm25_stream_plan");
    OI_REQUIRE(pb.p != nullptr, "bm25 (stream): no plan for this pass");
    BsArgs a;
    a.postings = reinterpret_cast<const uint2 *>(idx->postings.p);
    a.n_postings = idx->n_postings;
    a.cells = idx->cell_start.as<uint32_t>();
    a.idf = idx->idf.as<float>();
    a.q_terms = d_q_terms; a.q_offsets = d_q_offsets;
    a.cum = pb.as<uint64_t>();
    a.unit = reinterpret_cast<const uint32_t *>(a.cum + BS_MAX_Q + 2);
    a.pools = pool.keys; a.seg_cnt = pool.seg_cnt; a.tau_keys = pool.tau_keys; a.overflow = pool.overflow;
    a.pool_stride = pool.stride;
    a.n_win = idx->n_win; a.vocab = idx->vocab; a.doc_id_base = idx->doc_id_base;
    a.block0 = block_begin; a.nbh = block_end - block_begin; a.q_begin = q_begin; a.nq = nq;
    a.seg_cnt_stride = pool.seg_cnt_stride; a.carry_cap = pool.carry_cap; a.seg_cap = pool.seg_cap; a.depth = depth;
    using L = BsLds<BS_BLOCK>;
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(bm25_stream_kernel<BS_BLOCK>), (size_t)L::TOTAL));
    const uint64_t n_tasks = (uint64_t)a.nbh * nq;
    uint64_t per_cu = (160u * 1024u) / L::TOTAL; // resident workgroups per CU (LDS)
    if (const char *e = oi_ablation_env("OI_BM25_STREAM_WGS")) per_cu = std::max(1, atoi(e));
    uint64_t wgs = std::min<uint64_t>((n_tasks + BS_WPB - 1) / BS_WPB, per_cu * (uint64_t)ctx->num_cus);
    ProfScope ps(ctx, "bm25");
    hipLaunchKernelGGL(bm25_stream_kernel<BS_BLOCK>, dim3((uint32_t)wgs), dim3(BS_WPB * 64), L::TOTAL, ctx->stream, a);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}
