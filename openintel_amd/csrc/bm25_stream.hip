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
#ifndef BS_RING
#define BS_RING 4         // 1 KiB slots per wave (measured: 4 slots x 8 waves per CU beats 8 slots x 6 waves, 0.149 vs 0.176 ms)
#endif
#ifndef BS_CAP
#define BS_CAP 256u       // multi-doc accumulators per window and round
#endif
#define BS_STAGE 128u     // staged keys (a lane's two postings are appended one after the other: <= 64 to < 64 left over)
#define BS_MAX_Q 128u     // queries per pass
#define BS_RUN_COST 64u   // weight of a run beside its postings, in postings
#define BS_TASK_COST 384u // weight of a task beside its runs: table, sweep, multi docs, end of task
// (Round 4 measured what a wave's time is made of -- OI_BM25_STREAM_TIMING=light, per-wave records of the PRODUCT kernel
// regressed on what the waves did: lifetime = 0.383 us x chunk visits + 0.563 us x tasks + 11 us, residual 4.5 us; a run's
// chunks start at the run, so a run of X postings takes ceil(X / 128) chunks, each visited twice -- and rebuilt the weight as
// 766 ns x E[ceil(X_t / 128)] summed over the terms + 563 ns: modelled and measured visits per task then agree to 3 % per
// query, the launch's span went 102 -> 98 us, and the plan kernel's Poisson sums cost 5-8 us, more than they returned (0.111
// -> 0.116 ms at 10M docs, 0.046 -> 0.054 at a shard): not kept.  What is left of the spread -- mean 77 us, longest 98 -- is
// granularity: a range holds whole (query, block) tasks, 5 or 6 of a big query's 13-us tasks.)

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
    unsigned long long *timing; // TIMING instantiation only: per-section cycle sums over all waves
    const uint2 *qctx;          // [nq][64] per query and lane j: {cell index of (term j, window 0) or ~0 without a run, idf bits} (plan)
    uint32_t *queue;            // the launch's hand-out counter (zeroed by the plan kernel)
    uint32_t n_chunks;          // the weight axis is cut into this many ranges: wave w starts with range w, then draws
};

__device__ __forceinline__ uint32_t bs_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ uint32_t bs_rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t bs_readlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

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

// ---------------------------------------------------------------------------------------------------------------
// Per-wave LDS (W docs per window, WORDS = W / 32):
//   seen u32[WORDS] | multi u32[WORDS] | pref u16[WORDS] | acc f32[BS_CAP] | accdoc u32[BS_CAP] | stage u64[BS_STAGE] |
//   bounds u32[2][3][64] | (pad to 1 KiB) | ring [BS_RING][1 KiB];   head u32[64] lies over accdoc[0..64): a table is
//   built between two windows, when no accumulator is live
template <int W>
struct BsLds {
    static constexpr uint32_t WORDS = W / 32;
    static constexpr uint32_t OFF_SEEN = 0, OFF_MULTI = WORDS * 4, OFF_PREF = WORDS * 8, OFF_ACC = WORDS * 10,
                              OFF_ACCDOC = OFF_ACC + BS_CAP * 4, OFF_STAGE = OFF_ACCDOC + BS_CAP * 4,
                              OFF_DESC = OFF_STAGE + BS_STAGE * 8, OFF_HEAD = OFF_ACCDOC, // (head: see below)
                              OFF_RING = (OFF_DESC + 2 * 3 * 256 + 1023) & ~1023u, WAVE = OFF_RING + BS_RING * 1024;
    static constexpr uint32_t TOTAL = BS_WPB * WAVE + (BS_MAX_Q + 2) * 8 + BS_MAX_Q * 4; // + the plan's prefix and the thresholds
};

// The chunks of one window, one per lane (lane k = chunk k, in query order; a chunk never spans two runs): where its 128
// postings start (an even index: 16-byte aligned), which of them belong to the run, the run's idf.
struct BsTable {
    uint32_t pos;   // posting index of the chunk's first slot
    uint32_t lohi;  // first valid slot (0 or 1) | (one past the last valid slot, 1..128) << 8
    uint32_t idf;   // f32 bits
    uint32_t C;     // chunks of the window (uniform); 0 with slow = false: nothing to do
    bool slow;      // more than 64 chunks, or a query of more than 64 terms: the window goes through the direct passes
};

// TIMING (a diagnostic instantiation, never the product's launch): s_memtime stamps around the sections of a wave's life,
// summed over the waves into a.timing[0..9] = wait, pass A, sweep, pass B, table + bounds, finish + end of task, issue,
// query setup, whole wave, waves; [10] = the longest wave.
template <int W, bool TIMING = false>
__global__ __launch_bounds__(BS_WPB * 64) void bm25_stream_kernel(const BsArgs a) {
    using L = BsLds<W>;
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_big = 0, t_big_n = 0, t_first = 0, t_n_wait = 0, t_b_wait = 0; // long waits; a window's first wait; pass-B waits
    uint32_t t_stores = 0, t_tasks = 0; // (TIMING) 64-key stores and tasks so far
    auto stamp = [&]() __attribute__((always_inline)) -> unsigned long long { return TIMING ? __builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long t_wave0 = stamp();
#ifdef OI_ABLATION
    // OI_BM25_STREAM_TIMING=light: the PRODUCT instantiation with two constant-rate stamps per wave (start, end: s_memrealtime,
    // 100 MHz, the same clock on every XCD) -- the stamped instantiation's s_memtime reads drain the LDS queue at every
    // section and run 3.5x slower than this kernel, so its per-section shares are not this kernel's
    const unsigned long long t_light0 = (!TIMING && a.timing) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t t_first_q = 0xFFFFFFFFu; // the first query of the wave's range (light records)
#endif
    constexpr uint32_t WORDS = L::WORDS, WPL = WORDS / 64; // map words per lane in the sweep: 8 (W = 16384) or 16
    constexpr uint32_t NWB = BS_BLOCK / W;                  // windows per block: 2 or 1
    constexpr uint32_t WSTEP = W / BS_FINE;                 // cells per window: 1 or 2
    constexpr uint32_t R = BS_RING;
    static_assert((R & (R - 1)) == 0 && R >= 4, "ring slots: a power of two");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = bs_rfl(tid >> 6);
    unsigned char *wl = smem + wv * L::WAVE;
    uint32_t *seen = reinterpret_cast<uint32_t *>(wl + L::OFF_SEEN);
    uint32_t *multi = reinterpret_cast<uint32_t *>(wl + L::OFF_MULTI);
    uint32_t *pref = reinterpret_cast<uint32_t *>(wl + L::OFF_PREF); // two 16-bit ranks per word
    float *acc = reinterpret_cast<float *>(wl + L::OFF_ACC);
    uint32_t *accdoc = reinterpret_cast<uint32_t *>(wl + L::OFF_ACCDOC);
    uint64_t *stage = reinterpret_cast<uint64_t *>(wl + L::OFF_STAGE);
    uint32_t *desc = reinterpret_cast<uint32_t *>(wl + L::OFF_DESC);
    // (an explicit LDS pointer: a volatile access through a generic one becomes a FLAT instruction waited for with vmcnt(0))
    typedef volatile __attribute__((address_space(3))) uint32_t bs_lds_vu32;
    bs_lds_vu32 *head = (bs_lds_vu32 *)(__attribute__((address_space(3))) void *)(wl + L::OFF_HEAD);
    const uint4 *ring16 = reinterpret_cast<const uint4 *>(wl + L::OFF_RING);
    uint64_t *s_cum = reinterpret_cast<uint64_t *>(smem + BS_WPB * L::WAVE);
    uint32_t *s_tau = reinterpret_cast<uint32_t *>(s_cum + BS_MAX_Q + 2); // the queries' thresholds (fixed for the launch)
    const uint32_t ring_w = bs_lds_addr(wl + L::OFF_RING), desc_w = bs_lds_addr(desc);

    // ---- prologue: the plan's prefix and the thresholds in LDS, this wave's maps and accumulators zero
    for (uint32_t i = tid; i <= a.nq; i += BS_WPB * 64) s_cum[i] = a.cum[i];
    for (uint32_t i = tid; i < a.nq; i += BS_WPB * 64) s_tau[i] = a.tau_keys ? a.tau_keys[i] : 0u;
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
    auto locate = [&](uint64_t P, uint32_t &r_out, uint32_t &b_out) __attribute__((always_inline)) { // first task (r, b) with start >= P
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
    // Ranges are handed out DYNAMICALLY (round 4b): the weight axis is cut into n_chunks ranges (4 per wave), wave w takes
    // range w first and then draws the next free one from a.queue.  A static cut, one range per wave, ended with its
    // longest wave at 1.75x the average: waves with the same number of visits took 140K..805K cycles and XCDs 0, 1, 6, 7
    // ran ~40 % slower than 2..5 (per-wave records of the stamped build).
    const uint32_t NCH = a.n_chunks;
    auto cut = [&](uint32_t c) __attribute__((always_inline)) { return total / NCH * c + total % NCH * c / NCH; };
    const uint32_t lane16 = lane * 16;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    auto sample = [&](unsigned long long dur, uint32_t pass, uint32_t v, uint32_t C, uint32_t gcons, uint32_t ah) __attribute__((always_inline)) {
        if (TIMING && dur >= 2000 && a.timing && lane == 0) {
            const unsigned long long at = atomicAdd(&a.timing[16], 1ull);
            if (at < 2048) {
                unsigned long long *o = a.timing + 32 + at * 4;
                o[0] = dur; o[1] = ((unsigned long long)pass << 48) | ((unsigned long long)v << 32) | C;
                o[2] = ((unsigned long long)gcons << 32) | ((unsigned long long)t_stores << 16) | t_tasks;
                o[3] = ((unsigned long long)(stamp() - t_wave0) << 8) | ah;
            }
        }
    };

    // ================================================================ the task being scored
    uint32_t t_q = 0, t_blk = 0, t_doc0 = 0, t_tau = 0, t_tau_q = 0, out_n = 0, st_n = 0, t_M = 0;
    uint32_t q_tb = 0, q_T = 0; // the query's terms
    uint64_t *t_seg = a.pools;
    const uint32_t seg_cap = a.seg_cap;

    // The segment is full: keep its top `depth` keys (exact: only those can reach the global top `depth`).
    auto prune = [&]() __attribute__((always_inline)) {
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
    // stage[] (and accdoc[] / acc[] below) are exchanges between the lanes of one wave through LDS, like head[]: a lane reads
    // what OTHER lanes wrote.  The compiler barriers keep every such read behind the writes it depends on and in front of the
    // next round's writes (with plain accesses only, that rested on hipcc never forwarding a lane's own earlier store
    // to its load here -- the transformation that caused round 4's fault at head[]; cosine_prefilter.hip's PF_FLUSH does the same).
    auto flush64 = [&]() __attribute__((always_inline)) { // 64 staged keys leave for the segment
        if (out_n + 64u > seg_cap) prune();
        asm volatile("" ::: "memory");
        t_seg[out_n + lane] = stage[lane];
        out_n += 64u;
        if (TIMING) ++t_stores;
        const uint32_t rem = st_n - 64u; // < 64
        uint64_t k0 = 0;
        if (lane < rem) k0 = stage[64u + lane];
        asm volatile("" ::: "memory");
        if (lane < rem) stage[lane] = k0;
        asm volatile("" ::: "memory");
        st_n = rem;
    };
    auto emit1 = [&](bool k, uint32_t score_bits, uint32_t doc) __attribute__((always_inline)) { // all lanes call; score > 0: its key is bits | sign
        const unsigned long long m = __ballot(k);
        if (m) {
            if (k) stage[st_n + (uint32_t)__popcll(m & lt_mask)] = ((uint64_t)(score_bits | 0x80000000u) << 32) | (uint64_t)(~doc);
            st_n += (uint32_t)__popcll(m);
            if (st_n >= 64u) flush64();
        }
    };
    auto start_task = [&](uint32_t blk) __attribute__((always_inline)) {
        t_blk = blk;
        t_doc0 = a.doc_id_base + (a.block0 + blk) * BS_BLOCK;
        t_tau = t_tau_q;
        t_seg = a.pools + (uint64_t)t_q * a.pool_stride + a.carry_cap + (uint64_t)(a.block0 + blk) * seg_cap;
        out_n = 0;
    };
    auto end_task = [&]() __attribute__((always_inline)) {
        if (st_n) {
            if (out_n + st_n > seg_cap) prune();
            asm volatile("" ::: "memory");
            if (lane < st_n) t_seg[out_n + lane] = stage[lane];
            asm volatile("" ::: "memory");
            out_n += st_n;
            st_n = 0;
        }
        if (out_n > seg_cap) { *a.overflow = 1u; out_n = seg_cap; } // bug guard
        if (lane == 0) a.seg_cnt[(uint64_t)t_q * a.seg_cnt_stride + a.block0 + t_blk] = out_n;
        out_n = 0;
    };
    // ranks [lo, lo + BS_CAP) of the window's multi docs: emit and clear their accumulators
    auto emit_multi = [&](uint32_t lo) __attribute__((always_inline)) {
        const uint32_t cnt = t_M - lo < BS_CAP ? t_M - lo : BS_CAP;
        asm volatile("" ::: "memory"); // (acc[] / accdoc[] were written by whichever lane held the posting: add_multi)
        for (uint32_t b = 0; b < cnt; b += 64) {
            const uint32_t r = b + lane;
            const bool ok = r < cnt;
            const float v = acc[ok ? r : 0u];
            const uint32_t d = accdoc[ok ? r : 0u];
            if (ok) acc[r] = 0.0f;
            emit1(ok && v > 0.0f && (__float_as_uint(v) | 0x80000000u) >= t_tau, __float_as_uint(v), t_doc0 + d);
        }
    };
    // a multi doc's posting: acc[rank - lo] += x, in call order (ranks outside [lo, lo + BS_CAP) belong to another round)
    auto add_multi = [&](bool mul, uint32_t word, uint32_t mw, uint32_t bit, uint32_t dib, float x, uint32_t lo) __attribute__((always_inline)) {
        if (mul) {
            const uint32_t pw = pref[word >> 1];
            const uint32_t rr = ((word & 1u) ? pw >> 16 : pw & 0xFFFFu) + (uint32_t)__popc(mw & (bit - 1u)) - lo;
            if (rr < BS_CAP) {
                acc[rr] = __fadd_rn(acc[rr], x);
                accdoc[rr] = dib;
            }
        }
    };
    // pass A on two postings per lane (bit = 0: not a posting of the run): seen, then multi for the docs seen before
    auto pass_a = [&](uint32_t i0, uint32_t bit0, uint32_t i1, uint32_t bit1) __attribute__((always_inline)) {
        const uint32_t o0 = atomicOr(&seen[i0 >> 5], bit0);
        const uint32_t o1 = atomicOr(&seen[i1 >> 5], bit1);
        const uint32_t again0 = o0 & bit0, again1 = o1 & bit1; // the doc was in an earlier run
        if (again0) atomicOr(&multi[i0 >> 5], again0);
        if (again1) atomicOr(&multi[i1 >> 5], again1);
    };
    // pass B on two postings per lane, ranks [lo, lo + BS_CAP) of the multi docs; single-run docs are emitted iff `singles`
    auto pass_b = [&](uint32_t i0, uint32_t bit0, uint32_t d0, float x0, uint32_t i1, uint32_t bit1, uint32_t d1, float x1,
                      uint32_t lo, bool singles) __attribute__((always_inline)) {
        const uint32_t mw0 = multi[i0 >> 5], mw1 = multi[i1 >> 5];
        const bool mul0 = (mw0 & bit0) != 0u, mul1 = (mw1 & bit1) != 0u;
        if (singles) { // (BM25 lists hold scores > 0 only; the key of a score > 0 is its bits with the sign bit set)
            const bool k0 = bit0 && !mul0 && x0 > 0.0f && (__float_as_uint(x0) | 0x80000000u) >= t_tau;
            const bool k1 = bit1 && !mul1 && x1 > 0.0f && (__float_as_uint(x1) | 0x80000000u) >= t_tau;
            if (__ballot(k0 || k1)) {
                emit1(k0, __float_as_uint(x0), t_doc0 + d0);
                emit1(k1, __float_as_uint(x1), t_doc0 + d1);
            }
        }
        if (__ballot(mul0 || mul1)) { // a lane's two postings are one run: distinct docs
            add_multi(mul0, i0 >> 5, mw0, bit0, d0, x0, lo);
            add_multi(mul1, i1 >> 5, mw1, bit1, d1, x1, lo);
        }
    };
    // the sweep between the passes: ranks of the multi docs (exclusive popcount prefix per map word), seen cleared
    auto sweep = [&]() __attribute__((always_inline)) {
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
    };
    // Window `win` of the task straight from global memory, run by run in query order, 64 postings per step: the rounds
    // beyond the first of a window with more than BS_CAP multi docs, and whole windows the tables cannot describe.
    auto direct_runs = [&](uint32_t win, auto &&f) __attribute__((always_inline)) {
        for (uint32_t j = q_tb; j < q_tb + q_T; ++j) {
            const uint32_t term = bs_rfl(a.q_terms[j]);
            if (term >= a.vocab) continue;
            const uint64_t cell = (uint64_t)term * a.n_win + 2u * (a.block0 + t_blk) + win * WSTEP;
            const uint32_t s = bs_rfl(a.cells[cell]), e = bs_rfl(a.cells[cell + WSTEP]);
            const float wt = a.idf[term];
            for (uint32_t i0 = s; i0 < e; i0 += 64) {
                const bool ok = i0 + lane < e;
                const uint2 p = a.postings[ok ? i0 + lane : s];
                const uint32_t ix = p.x & (W - 1u);
                f(ix, ok ? 1u << (ix & 31u) : 0u, p.x, __fmul_rn(wt, __uint_as_float(p.y)));
            }
        }
    };
    auto direct_pass_b = [&](uint32_t win, uint32_t lo, bool singles) __attribute__((always_inline)) {
        direct_runs(win, [&](uint32_t ix, uint32_t bit, uint32_t dib, float x) __attribute__((always_inline)) { pass_b(ix, bit, dib, x, 0u, 0u, 0u, 0.f, lo, singles); });
    };
    auto finish_window = [&](uint32_t win) __attribute__((always_inline)) { // after pass B's first round: the multi docs out, further rounds, maps clean
        if (t_M) {
            emit_multi(0u);
            for (uint32_t lo = BS_CAP; lo < t_M; lo += BS_CAP) {
                direct_pass_b(win, lo, false);
                emit_multi(lo);
            }
            uint4 *m4 = reinterpret_cast<uint4 *>(multi) + lane * (WPL / 4);
#pragma unroll
            for (uint32_t k = 0; k < WPL / 4; ++k) m4[k] = make_uint4(0u, 0u, 0u, 0u);
            t_M = 0;
        }
    };

    // ================================================================ bounds and chunk tables
    uint32_t d_cb = 0;   // lane j: cell index of (term j of the query, window 0); 0 without a run
    bool d_ok = false;   // lane j: the query has a term j (< 64) and it is inside the vocabulary
    float d_idf = 0.f;   // lane j: idf of that term
    uint32_t c0 = 0, c1 = 0, c2 = 0; // lane j: cell_start words of a block (windows 0, 1, end)
    uint32_t vseq = 0;   // LDS-DMA operations issued so far (chunks and bounds)
    // cell words of block `blk` for the lanes' runs, by LDS-DMA into bounds buffer `buf` (no VGPR destination)
    auto issue_bounds = [&](uint32_t blk, uint32_t buf) __attribute__((always_inline)) {
        const uint32_t *src = a.cells + d_cb + 2u * (a.block0 + blk); // (lanes without a run: d_cb = 0, any valid word)
        bs_dma_word(src, desc_w + (buf * 3u + 0u) * 256u);
        bs_dma_word(src + 1, desc_w + (buf * 3u + 1u) * 256u);
        bs_dma_word(src + 2, desc_w + (buf * 3u + 2u) * 256u);
        vseq += 3;
    };
    auto read_bounds = [&](uint32_t buf) __attribute__((always_inline)) { // (lanes without a run: an empty one)
        c0 = desc[(buf * 3u + 0u) * 64u + lane];
        c1 = desc[(buf * 3u + 1u) * 64u + lane];
        c2 = desc[(buf * 3u + 2u) * 64u + lane];
        c0 = d_ok ? c0 : 0u; c1 = d_ok ? c1 : 0u; c2 = d_ok ? c2 : 0u;
    };
    auto build_table = [&](uint32_t win, bool long_query) __attribute__((always_inline)) {
        BsTable t;
        const uint32_t s = (NWB == 1 || win == 0) ? c0 : c1, e = NWB == 1 ? c2 : (win == 0 ? c1 : c2);
        const uint32_t cs = s & ~1u;
        const uint32_t n = e > s ? (e - cs + 127u) >> 7 : 0u; // chunks of run `lane`
        const uint32_t incl = bs_incl_scan(n), excl = incl - n;
        t.C = bs_readlane(incl, 63);
        t.slow = long_query || t.C > 64u;
        t.pos = 0; t.lohi = 0; t.idf = 0;
        if (t.slow) { t.C = 0; return t; }
        if (t.C == 0u) return t;
        // chunk k -> its run: every run with chunks marks the lane of its first chunk, a max-scan spreads the marks.
        // VOLATILE: lanes talk to each other through these words.  With plain accesses hipcc forwards a lane's own
        // "head[lane] = 0" to its read below (legal for unsynchronised threads) and the marks of the other lanes are lost:
        // chunks land in the wrong run, positions run past the end of a posting list -- a memory fault (round 4, found in the ISA).
        // Round 5: and a wave barrier on either side of the scatter, so that a later hipcc cannot move the accesses across it
        // either (the exchange is between lanes of ONE wave: LDS operations of a wave execute in program order).
        head[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (n) head[excl] = lane + 1u;
        __builtin_amdgcn_wave_barrier();
        uint32_t h = head[lane];
        h = max(h, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x111, 0xf, 0xf, false));
        h = max(h, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x112, 0xf, 0xf, false));
        h = max(h, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x114, 0xf, 0xf, false));
        h = max(h, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x118, 0xf, 0xf, false));
        h = max(h, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x142, 0xa, 0xf, false));
        h = max(h, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x143, 0xc, 0xf, false));
        const int j = lane < t.C ? (int)h - 1 : 0; // (lane 0 carries a mark whenever C > 0)
        const uint32_t cs_j = (uint32_t)__shfl((int)cs, j, OI_WAVE), s_j = (uint32_t)__shfl((int)s, j, OI_WAVE);
        const uint32_t e_j = (uint32_t)__shfl((int)e, j, OI_WAVE), ex_j = (uint32_t)__shfl((int)excl, j, OI_WAVE);
        t.idf = (uint32_t)__shfl((int)__float_as_uint(d_idf), j, OI_WAVE);
        t.pos = cs_j + 128u * (lane - ex_j);
        const uint32_t lo = lane == ex_j ? s_j - cs_j : 0u, left = e_j - t.pos;
        t.lohi = lo | ((left < 128u ? left : 128u) << 8);
        return t;
    };
    uint32_t g_issue = 0, g_consume = 0; // chunk DMAs issued / consumed so far: ring slot = counter % R
    uint32_t ahead = 0;                  // issued and not yet consumed (<= R)
    auto issue = [&](const BsTable &t, uint32_t k) __attribute__((always_inline)) {
        const uint32_t pos = bs_readlane(t.pos, k);
        const uint2 *src = a.postings + pos; // (the array is padded: a chunk may run 1 KiB past its run)
        const uint32_t dst = bs_rfl(ring_w + (g_issue & (R - 1u)) * 1024u);
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(lane16), "s"(src), "s"(dst)
            : "memory");
        ++g_issue; ++ahead; ++vseq;
    };
    // the oldest chunk in flight has landed (chunks complete in issue order; `ahead - 1` younger ones may stay in flight)
    auto wait_oldest = [&]() __attribute__((always_inline)) {
        if (ahead >= R) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory");
        else if (ahead >= R / 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R / 2 - 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto consume = [&](const BsTable &t, uint32_t k, bool pass_b_now) __attribute__((always_inline)) {
        const uint32_t lohi = bs_readlane(t.lohi, k);
        const uint32_t lo = lohi & 1u, hi = lohi >> 8;
        const uint4 v = ring16[(g_consume & (R - 1u)) * 64u + lane];
        const uint32_t i0 = v.x & (W - 1u), i1 = v.z & (W - 1u);
        const bool ok0 = 2u * lane >= lo && 2u * lane < hi, ok1 = 2u * lane + 1u < hi; // (2 lane + 1 >= lo always)
        const uint32_t bit0 = ok0 ? 1u << (i0 & 31u) : 0u, bit1 = ok1 ? 1u << (i1 & 31u) : 0u;
        if (!pass_b_now) pass_a(i0, bit0, i1, bit1);
        else {
            const float wt = __uint_as_float(bs_readlane(t.idf, k));
            pass_b(i0, bit0, v.x, __fmul_rn(wt, __uint_as_float(v.y)), i1, bit1, v.z, __fmul_rn(wt, __uint_as_float(v.w)), 0u, true);
        }
        ++g_consume; --ahead;
    };

    // ================================================================ the stream: range by range, query by query, block by block
    for (uint32_t ch = w; ch < NCH;) {
    uint32_t r0, b0, r1, b1;
    locate(cut(ch), r0, b0);
    if (ch + 1 == NCH) { r1 = nq; b1 = 0; } else locate(cut(ch + 1), r1, b1);
    r0 = bs_rfl(r0); b0 = bs_rfl(b0); r1 = bs_rfl(r1); b1 = bs_rfl(b1);
    for (uint32_t r = r0; r <= r1 && r < nq; ++r) {
        const uint32_t bA = r == r0 ? b0 : 0u, bB = r == r1 ? b1 : nbh;
        if (bA >= bB) continue;
        // ---- the query (plain loads: nothing is in flight here)
        const unsigned long long t_qs = stamp();
        t_q = r;
#ifdef OI_ABLATION
        if (t_first_q == 0xFFFFFFFFu) t_first_q = r;
#endif
        t_tau_q = bs_rfl(s_tau[r]);
        q_tb = bs_rfl(a.q_offsets[a.q_begin + r]);
        q_T = bs_rfl(a.q_offsets[a.q_begin + r + 1]) - q_tb;
        const bool long_query = q_T > 64u;
        {   // the lanes' runs: ONE coalesced load of what the plan kernel prepared (round 4b: term id -> idf / list base was a
            // chain of dependent loads at every range and query start, ~2.5 us each under load)
            const uint2 qc = a.qctx[(uint64_t)r * 64u + lane];
            d_ok = qc.x != 0xFFFFFFFFu;
            d_cb = d_ok ? qc.x : 0u;
            d_idf = d_ok ? __uint_as_float(qc.y) : 0.f;
            // Retire this load HERE: hipcc's s_waitcnt pass otherwise carries "d_idf may be in flight" around the whole
            // stream loop and drains the DMA ring with a vmcnt(0) wherever it is used.
            asm volatile("" : "+v"(d_idf), "+v"(d_cb));
        }
        uint32_t pf_buf = 0, pf_seq = 0;
        issue_bounds(bA, 0);
        bs_wait_vm(0);
        read_bounds(0);
        if (bA + 1 < bB) { pf_buf = 1; pf_seq = vseq; issue_bounds(bA + 1, 1); }
        BsTable cur = build_table(0, long_query);
        if (TIMING) t_acc[7] += stamp() - t_qs;
        uint32_t icur = 0; // visits of `cur` issued so far
        uint32_t blk = bA, win = 0;
        for (;;) {
            // ---- the next window and its table (its block's bounds were prefetched a block ago)
            uint32_t nblk = blk, nwin = win + 1;
            if (nwin == NWB) { nwin = 0; ++nblk; }
            const bool has_nxt = nblk < bB;
            BsTable nxt;
            nxt.pos = nxt.lohi = nxt.idf = nxt.C = 0; nxt.slow = false;
            const unsigned long long t_tb = stamp();
            if (has_nxt) {
                if (nwin == 0) {
                    bs_wait_vm(vseq - pf_seq - 3u);
                    read_bounds(pf_buf);
                    if (nblk + 1 < bB) { pf_buf ^= 1u; pf_seq = vseq; issue_bounds(nblk + 1, pf_buf); }
                }
                nxt = build_table(nwin, long_query);
            }
            if (TIMING) t_acc[4] += stamp() - t_tb;
            uint32_t inxt = 0;
            if (win == 0) {
                start_task(blk);
#ifdef OI_ABLATION
                ++t_tasks; // (the light records count tasks and visits too)
#else
                if (TIMING) ++t_tasks;
#endif
            }
            if (cur.slow) { // (nothing of this window is in the ring)
                direct_runs(win, [&](uint32_t ix, uint32_t bit, uint32_t, float) __attribute__((always_inline)) { pass_a(ix, bit, 0u, 0u); });
                sweep();
                direct_pass_b(win, 0u, true);
                finish_window(win);
            } else if (cur.C) {
                const uint32_t C = cur.C, V = 2u * C; // visits: the chunks as pass A, then again as pass B
                while (ahead < R && icur < V) { issue(cur, icur < C ? icur : icur - C); ++icur; }
                // (a macro, not a lambda: with icur / inxt captured by reference hipcc kept them in SCRATCH memory and waited for
                // every reload with vmcnt(0) -- the whole ring drained once per visit; found with in-kernel stamps, round 4)
#define BS_REFILL()                                                                           \
    do {                                                                                      \
        if (icur < V) { issue(cur, icur < C ? icur : icur - C); ++icur; }                     \
        else if (inxt < 2u * nxt.C) { issue(nxt, inxt < nxt.C ? inxt : inxt - nxt.C); ++inxt; } \
    } while (0)
                if (!TIMING) {
                    for (uint32_t v = 0; v < C; ++v) { wait_oldest(); consume(cur, v, false); BS_REFILL(); }
                    sweep();
                    for (uint32_t v = 0; v < C; ++v) { wait_oldest(); consume(cur, v, true); BS_REFILL(); }
#ifdef OI_ABLATION
                    t_n_wait += V;
#endif
                } else {
                    for (uint32_t v = 0; v < C; ++v) {
                        const unsigned long long t0 = stamp(); wait_oldest();
                        const unsigned long long t1 = stamp(); consume(cur, v, false);
                        const unsigned long long t2 = stamp(); BS_REFILL();
                        const unsigned long long t3 = stamp();
                        t_acc[0] += t1 - t0; t_acc[1] += t2 - t1; t_acc[6] += t3 - t2;
                        if (t1 - t0 >= 2000) { t_big += t1 - t0; ++t_big_n; }
                        sample(t1 - t0, 0, v, C, g_consume, ahead);
                        if (v == 0) t_first += t1 - t0;
                        ++t_n_wait;
                    }
                    const unsigned long long ts = stamp();
                    sweep();
                    t_acc[2] += stamp() - ts;
                    for (uint32_t v = 0; v < C; ++v) {
                        const unsigned long long t0 = stamp(); wait_oldest();
                        const unsigned long long t1 = stamp(); consume(cur, v, true);
                        const unsigned long long t2 = stamp(); BS_REFILL();
                        const unsigned long long t3 = stamp();
                        t_acc[0] += t1 - t0; t_acc[3] += t2 - t1; t_acc[6] += t3 - t2;
                        if (t1 - t0 >= 2000) { t_big += t1 - t0; ++t_big_n; }
                        sample(t1 - t0, 1, v, C, g_consume, ahead);
                        t_b_wait += t1 - t0;
                        ++t_n_wait;
                    }
                }
#undef BS_REFILL
                const unsigned long long tf = stamp();
                finish_window(win);
                if (TIMING) t_acc[5] += stamp() - tf;
            }
            const unsigned long long te = stamp();
            if (win == NWB - 1) end_task();
            if (TIMING) t_acc[5] += stamp() - te;
            if (!has_nxt) break;
            cur = nxt; icur = inxt; blk = nblk; win = nwin;
        }
    }
    {   // the next free range (nothing is in flight here; the result is used at once: no load left pending for hipcc's waits)
        uint32_t nx = 0;
        if (lane == 0) nx = atomicAdd(a.queue, 1u);
        ch = G + bs_rfl(nx);
    }
    }
#ifdef OI_ABLATION
    if (!TIMING && a.timing && lane == 0 && w < 4096) {
        unsigned long long *o = a.timing + 32 + 2048 * 4 + (unsigned long long)w * 4;
        uint32_t xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        o[0] = t_light0; o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = ((unsigned long long)t_n_wait << 32) | t_tasks;
        o[3] = ((unsigned long long)(xcc & 0xF) << 56) | ((unsigned long long)(hwid & 0xFFFFFF) << 32) | t_first_q;
    }
#endif
    if (TIMING && lane == 0 && a.timing) {
        const unsigned long long whole = stamp() - t_wave0;
        for (int i = 0; i < 8; ++i) atomicAdd(&a.timing[i], t_acc[i]);
        atomicAdd(&a.timing[8], whole);
        atomicAdd(&a.timing[9], 1ull);
        atomicMax(&a.timing[10], whole);
        atomicAdd(&a.timing[11], t_big); atomicAdd(&a.timing[12], t_big_n); atomicAdd(&a.timing[13], t_first);
        atomicAdd(&a.timing[14], t_n_wait); atomicAdd(&a.timing[15], t_b_wait);
        if (w < 4096) { // per-wave record: whole | wait | visits, tasks | hardware ids, start time
            unsigned long long *o = a.timing + 32 + 2048 * 4 + (unsigned long long)w * 4;
            uint32_t xcc = 0, hwid = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            o[0] = whole; o[1] = t_acc[0];
            o[2] = ((unsigned long long)t_n_wait << 32) | t_tasks;
            o[3] = ((unsigned long long)(xcc & 0xF) << 56) | ((unsigned long long)(hwid & 0xFFFFFF) << 32) | (uint32_t)(t_wave0 & 0xFFFFFFFFull);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Plan of a pass (one launch, before the first phase): zero the pass's pool state and weigh the queries.
//   unit[r] = postings of ONE block of query r on average (sum of its terms' local df / blocks) + a cost per run;
//   cum = its exclusive prefix.  The stream kernel cuts cum[nq] * (blocks of the launch) into equal parts.
__global__ __launch_bounds__(256) void bm25_plan_kernel(const uint32_t *__restrict__ q_terms, const uint32_t *__restrict__ q_offsets,
                                                        const uint32_t *__restrict__ df, uint32_t vocab, uint32_t n_blocks,
                                                        uint32_t q_begin, uint32_t nq, uint32_t *unit, uint64_t *cum,
                                                        uint32_t *state, uint64_t state_words, const float *__restrict__ idf,
                                                        uint32_t n_win, uint2 *qctx, const uint32_t *__restrict__ floors,
                                                        uint32_t floor_j, uint32_t tau_off, uint32_t tau_words) {
    // floors (round 4): the per-term impact floors of bm25.hip -- state[tau_off + r] (query r's threshold key) starts at
    // max over its terms of fl(idf * floor[term][floor_j]) instead of 0; those words are workgroup 0's, the rest is zeroed here
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < state_words; i += (uint64_t)gridDim.x * 256)
        if (!floors || i < tau_off || i >= (uint64_t)tau_off + tau_words) state[i] = 0u;
    // every query's first 64 runs as the stream kernel's lanes want them: {term * n_win, idf} or {~0, 0} (all workgroups)
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nq * 64u; i += gridDim.x * 256) {
        const uint32_t r = i >> 6, j = i & 63u;
        const uint32_t tb = q_offsets[q_begin + r], te = q_offsets[q_begin + r + 1];
        uint32_t term = 0xFFFFFFFFu;
        if (tb + j < te) term = q_terms[tb + j];
        const bool ok = term < vocab;
        qctx[i] = ok ? make_uint2(term * n_win, __float_as_uint(idf[term])) : make_uint2(0xFFFFFFFFu, 0u);
    }
    if (blockIdx.x != 0) return;
    __shared__ uint32_t s_unit[BS_MAX_Q];
    for (uint32_t r = threadIdx.x; r < nq; r += 256) {
        uint64_t wsum = 0, runs = 0;
        for (uint32_t i = q_offsets[q_begin + r]; i < q_offsets[q_begin + r + 1]; ++i) {
            const uint32_t term = q_terms[i];
            wsum += term < vocab ? df[term] : 0u;
            runs += term < vocab && df[term] ? 1u : 0u;
        }
        uint64_t u = wsum / (n_blocks ? n_blocks : 1u) + BS_RUN_COST * runs + BS_TASK_COST;
        if (u > (1u << 24)) u = 1u << 24; // keeps cum[nq] * blocks * waves inside 64 bits
        s_unit[r] = (uint32_t)u;
        unit[r] = (uint32_t)u;
    }
    if (floors) {
        for (uint32_t r = threadIdx.x; r < tau_words; r += 256) {
            float best = 0.f;
            if (r < nq)
                for (uint32_t i = q_offsets[q_begin + r]; i < q_offsets[q_begin + r + 1]; ++i) {
                    const uint32_t term = q_terms[i];
                    if (term < vocab) { // the product the kernel forms for a posting of this term with that impact (one rounding, monotone)
                        const float f = __fmul_rn(idf[term], __uint_as_float(floors[(uint64_t)term * OI_BM25_FLOOR_RANKS + floor_j]));
                        best = f > best ? f : best;
                    }
                }
            state[tau_off + r] = best > 0.f ? (__float_as_uint(best) | 0x80000000u) : 0u; // the key of a score > 0; 0 = no threshold
        }
    }
    if (threadIdx.x < 2) unit[BS_MAX_Q + threadIdx.x] = 0u; // the two phases' hand-out counters
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
                        uint32_t *state, uint64_t state_words, uint32_t depth, uint32_t tau_off, uint32_t tau_words, bool with_floors) {
    oi_ctx *ctx = idx->ctx;
    OI_REQUIRE(nq >= 1 && nq <= BS_MAX_Q, "bm25 (stream): %u queries in one pass (limit %u)", nq, BS_MAX_Q);
    DevBuf &pb = ctx->buf("bm25_stream_plan");
    OI_CHECK(pb.ensure(sizeof(uint64_t) * (BS_MAX_Q + 2) + sizeof(uint32_t) * (BS_MAX_Q + 2) + sizeof(uint2) * BS_MAX_Q * 64)); // cum | unit | two hand-out counters | qctx
    uint64_t *cum = pb.as<uint64_t>();
    uint32_t *unit = reinterpret_cast<uint32_t *>(cum + BS_MAX_Q + 2);
    uint2 *qctx = reinterpret_cast<uint2 *>(unit + BS_MAX_Q + 2);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>((nq * 64u + 255u) / 256u, (state_words + 4095) / 4096));
    ProfScope ps(ctx, "bm25");
    hipLaunchKernelGGL(bm25_plan_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_q_terms, d_q_offsets,
                       idx->df_local.as<uint32_t>(), idx->vocab, idx->n_blocks, q_begin, nq, unit, cum, state, state_words,
                       idx->idf.as<float>(), idx->n_win, qctx, with_floors && idx->impact_floor.p ? idx->impact_floor.as<uint32_t>() : nullptr,
                       depth <= 16 ? 0u : depth <= 64 ? 1u : depth <= 256 ? 2u : 3u, tau_off, tau_words);
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
    a.timing = nullptr;
    const uint64_t n_tasks = (uint64_t)a.nbh * nq;
    a.queue = reinterpret_cast<uint32_t *>(const_cast<uint32_t *>(a.unit) + BS_MAX_Q) + (block_begin ? 1 : 0); // one counter per phase
    a.qctx = reinterpret_cast<const uint2 *>(a.unit + BS_MAX_Q + 2);
    // A/B switches of the -DOI_ABLATION build (tools/): 16384-doc windows (10 waves per CU, a third more chunk visits: measured
    // slower, 0.157 vs 0.143 ms), resident workgroups per CU, the stamped instantiation
    static const bool half_windows = oi_ablation_env("OI_BM25_STREAM_W") && atoi(oi_ablation_env("OI_BM25_STREAM_W")) == 16384;
    ProfScope ps(ctx, "bm25");
    auto launch = [&](auto kernel, uint32_t lds_total) -> int {
        OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(kernel), (size_t)lds_total));
        uint64_t per_cu = (160u * 1024u) / lds_total; // resident workgroups per CU (LDS)
        if (const char *e = oi_ablation_env("OI_BM25_STREAM_WGS")) per_cu = std::max(1, atoi(e));
        const uint64_t wgs = std::min<uint64_t>((n_tasks + BS_WPB - 1) / BS_WPB, per_cu * (uint64_t)ctx->num_cus);
        uint64_t per_wave = 1; // ranges per wave (measured: 1 / 2 / 4 / 8 give 0.152 / 0.157 / 0.179 / 0.262 ms -- a range START costs more than the balance returns)
        if (const char *e = oi_ablation_env("OI_BM25_STREAM_CHUNKS")) per_wave = std::max(1, atoi(e));
        a.n_chunks = (uint32_t)std::max<uint64_t>(wgs * BS_WPB, std::min<uint64_t>(n_tasks, per_wave * wgs * BS_WPB));
        hipLaunchKernelGGL(kernel, dim3((uint32_t)wgs), dim3(BS_WPB * 64), lds_total, ctx->stream, a);
        return OI_OK;
    };
#ifdef OI_ABLATION
    if (oi_ablation_env("OI_BM25_STREAM_TIMING") && strcmp(oi_ablation_env("OI_BM25_STREAM_TIMING"), "light") == 0) {
        DevBuf &tb = ctx->buf("bm25_stream_timing");
        const size_t tbytes = (32 + 2048 * 4 + 4096 * 4) * sizeof(unsigned long long);
        OI_CHECK(tb.ensure(tbytes));
        OI_HIP_CHECK(hipMemsetAsync(tb.p, 0, tbytes, ctx->stream));
        a.timing = tb.as<unsigned long long>();
        OI_CHECK(launch(bm25_stream_kernel<BS_BLOCK>, BsLds<BS_BLOCK>::TOTAL));
        OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        std::vector<unsigned long long> smp(32 + 2048 * 4 + 4096 * 4);
        OI_HIP_CHECK(hipMemcpy(smp.data(), tb.p, smp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (unsigned i = 0; i < 4096; ++i) { const unsigned long long *o = &smp[32 + 2048 * 4 + (size_t)i * 4]; if (o[1]) { t0 = std::min(t0, o[0]); t1 = std::max(t1, o[1]); } }
        double xs[16] = {0}, xe[16] = {0}, xb[16] = {0}, xl[16] = {0}; unsigned xn[16] = {0};
        for (unsigned i = 0; i < 4096; ++i) {
            const unsigned long long *o = &smp[32 + 2048 * 4 + (size_t)i * 4];
            if (!o[1]) continue;
            const unsigned x = (unsigned)(o[3] >> 56) & 15u;
            xs[x] += (double)(o[1] - o[0]); xb[x] += (double)(o[0] - t0); xe[x] += (double)(o[1] - t0); xl[x] = std::max(xl[x], (double)(o[1] - t0)); ++xn[x];
        }
        fprintf(stderr, "[bm25 stream light] launch span %.1f us (100 MHz stamps)\n", (double)(t1 - t0) / 100.0);
        if (const char *path = oi_ablation_env("OI_BM25_STREAM_LIGHT_CSV")) { // raw per-wave rows for a regression
            if (FILE *f = fopen(path, "w")) {
                fprintf(f, "wave,xcc,start_us,life_us,tasks,visits,first_q\n");
                for (unsigned i = 0; i < 4096; ++i) {
                    const unsigned long long *o = &smp[32 + 2048 * 4 + (size_t)i * 4];
                    if (o[1]) fprintf(f, "%u,%u,%.2f,%.2f,%llu,%llu,%u\n", i, (unsigned)(o[3] >> 56) & 15u, (double)(o[0] - t0) / 100.0, (double)(o[1] - o[0]) / 100.0,
                                      o[2] & 0xFFFFFFFFull, o[2] >> 32, (unsigned)(o[3] & 0xFFFFFFFFull));
                }
                fclose(f);
            }
        }
        for (unsigned x = 0; x < 16; ++x)
            if (xn[x]) fprintf(stderr, "[bm25 stream light xcc %u] waves %u: start +%.1f us, lifetime %.1f us, end +%.1f us (last +%.1f)\n", x, xn[x], xb[x] / xn[x] / 100.0,
                               xs[x] / xn[x] / 100.0, xe[x] / xn[x] / 100.0, xl[x] / 100.0);
        return OI_OK;
    }
    if (oi_ablation_env("OI_BM25_STREAM_TIMING")) { // the stamped instantiation, sums and the first long waits printed per launch
        DevBuf &tb = ctx->buf("bm25_stream_timing");
        const size_t tbytes = (32 + 2048 * 4 + 4096 * 4) * sizeof(unsigned long long);
        OI_CHECK(tb.ensure(tbytes));
        OI_HIP_CHECK(hipMemsetAsync(tb.p, 0, tbytes, ctx->stream));
        a.timing = tb.as<unsigned long long>();
        OI_CHECK(launch(bm25_stream_kernel<BS_BLOCK, true>, BsLds<BS_BLOCK>::TOTAL));
        unsigned long long h[16];
        OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        OI_HIP_CHECK(hipMemcpy(h, tb.p, sizeof(h), hipMemcpyDeviceToHost));
        const double wv = h[9] ? (double)h[9] : 1.0;
        fprintf(stderr, "[bm25 stream timing] blocks %u tasks %llu waves %llu | cycles per wave: wait %.0f passA %.0f sweep %.0f "
                        "passB %.0f table %.0f finish %.0f issue %.0f query %.0f | whole %.0f longest %llu\n",
                a.nbh, (unsigned long long)n_tasks, h[9], h[0] / wv, h[1] / wv, h[2] / wv, h[3] / wv, h[4] / wv, h[5] / wv, h[6] / wv,
                h[7] / wv, h[8] / wv, h[10]);
        {
            std::vector<unsigned long long> smp(32 + 2048 * 4 + 4096 * 4);
            OI_HIP_CHECK(hipMemcpy(smp.data(), tb.p, smp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            const unsigned long long ns = std::min<unsigned long long>(smp[16], 2048);
            fprintf(stderr, "[bm25 stream sample] %llu samples\n", smp[16]);
            { // per-wave records: by XCC, and the ten longest waves
                const unsigned long long nw = std::min<unsigned long long>(h[9], 4096);
                double xs[16] = {0}, xw[16] = {0}, xv[16] = {0}; unsigned xn[16] = {0};
                std::vector<std::pair<unsigned long long, unsigned>> order;
                unsigned long long t0min = ~0ull;
                for (unsigned i = 0; i < nw; ++i) { const unsigned long long *o = &smp[32 + 2048 * 4 + (size_t)i * 4]; if (o[0]) t0min = std::min<unsigned long long>(t0min, o[3] & 0xFFFFFFFFull); }
                for (unsigned i = 0; i < nw; ++i) {
                    const unsigned long long *o = &smp[32 + 2048 * 4 + (size_t)i * 4];
                    if (!o[0]) continue;
                    const unsigned x = (unsigned)(o[3] >> 56) & 15u;
                    xs[x] += (double)o[0]; xw[x] += (double)o[1]; xv[x] += (double)(o[2] >> 32); ++xn[x];
                    order.push_back({o[0], i});
                }
                for (unsigned x = 0; x < 16; ++x) if (xn[x]) fprintf(stderr, "[bm25 stream xcc %u] waves %u whole %.0f wait %.0f visits %.1f\n", x, xn[x], xs[x] / xn[x], xw[x] / xn[x], xv[x] / xn[x]);
                std::sort(order.begin(), order.end());
                for (size_t k = 0; k < order.size(); k += std::max<size_t>(1, order.size() / 12)) {
                    const unsigned long long *o = &smp[32 + 2048 * 4 + (size_t)order[k].second * 4];
                    fprintf(stderr, "[bm25 stream wave pct %2.0f] wave %u whole %llu wait %llu visits %llu tasks %llu xcc %llu hwid %06llx start +%llu\n", 100.0 * k / order.size(),
                            order[k].second, o[0], o[1], o[2] >> 32, o[2] & 0xFFFFFFFF, o[3] >> 56, (o[3] >> 32) & 0xFFFFFF, (o[3] & 0xFFFFFFFFull) - t0min);
                }
                const unsigned long long *o = &smp[32 + 2048 * 4 + (size_t)order.back().second * 4];
                fprintf(stderr, "[bm25 stream wave longest] wave %u whole %llu wait %llu visits %llu tasks %llu xcc %llu start +%llu\n", order.back().second, o[0], o[1], o[2] >> 32,
                        o[2] & 0xFFFFFFFF, o[3] >> 56, (o[3] & 0xFFFFFFFFull) - t0min);
            }
            const unsigned long long n_print = std::max(6, atoi(oi_ablation_env("OI_BM25_STREAM_TIMING"))); // OI_BM25_STREAM_TIMING=N: N samples of long waits printed
            for (unsigned long long i = 0; i < ns && i < n_print; ++i) {
                const unsigned long long *o = &smp[32 + i * 4];
                fprintf(stderr, "[bm25 stream sample] dur %llu pass %llu v %llu C %llu | visit# %llu stores %llu task# %llu | at cycle %llu ahead %llu\n", o[0],
                        o[1] >> 48, (o[1] >> 32) & 0xFFFF, o[1] & 0xFFFFFFFF, o[2] >> 32, (o[2] >> 16) & 0xFFFF, o[2] & 0xFFFF, o[3] >> 8, o[3] & 255);
            }
        }
        fprintf(stderr, "[bm25 stream timing]   waits per wave %.0f, of them >= 2000 cycles: %.1f (sum %.0f); first wait of a window: sum %.0f; pass-B waits: sum %.0f\n",
                h[14] / wv, h[12] / wv, h[11] / wv, h[13] / wv, h[15] / wv);
        return OI_OK;
    }
    if (half_windows) { OI_CHECK(launch(bm25_stream_kernel<BS_FINE>, BsLds<BS_FINE>::TOTAL)); OI_HIP_CHECK(hipGetLastError()); return OI_OK; }
#endif
    (void)half_windows;
    OI_CHECK(launch(bm25_stream_kernel<BS_BLOCK>, BsLds<BS_BLOCK>::TOTAL));
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}
