// cosine_bf16.hip -- the cosine scorer over a bf16 corpus (BASELINE configs[4]: 100M x 1024-d bf16).
//
// Builder-defined like the rest of the retrieval path (the reference has none).  Scores are
// sum_k bf16(q_k) * x_k accumulated in f32 by v_mfma_f32_32x32x16_bf16 (products of two bf16 values
// are exact in f32), i.e. the cosine of the rows AS STORED with the queries rounded to bf16 -- the
// oracle for this kernel does the same rounding and sums in f64.
//
// Regime.  At 2 bytes per element the scorer needs 64 flop per corpus byte at B = 64 while the matrix
// pipes offer ~300: the kernel is HBM-bound (15.4 GB per pass at 10M x 768), so it is built to stream,
// not to keep the MFMA pipe full:
//   * one workgroup per CU, 4 waves, one per SIMD, each owning the full 512-register file.  There is
//     NO K-split here: a wave holds ALL 64 queries over the whole K in registers (64 x 768 bf16 =
//     96 KB = 384 VGPRs per lane, as B operands of the 32x32x16 MFMA) and owns whole 32-row tiles, so
//     waves never meet -- no cross-wave sum, no barrier after the prologue;
//   * each wave streams its tiles through its own LDS ring of 4 KiB slots (32 rows x 128 B = 64 bf16
//     of K) with buffer_load ... lds, P slots ahead, ordered by counted s_waitcnt vmcnt, the prefetch
//     running across tile boundaries (same DMA, descriptor and swizzle as cosine_ksplit.hip);
//   * a lane's A fragment (8 consecutive bf16 of its row) is one conflict-free ds_read_b128 per MFMA
//     group; the filter + pool append come straight out of the accumulators (the 32x32 D layout maps
//     a lane to one query column), into this workgroup's private pool segment.
// d = 1024 with 64 queries would need all 512 registers for the queries alone: it runs 32 queries per
// pass (NQT = 1); d = 384 and 768 run 64.
#include <cstdlib>
#include <type_traits>

#include "oi_device.h"
#include "oi_internal.h"

#ifndef OI_TILE_CONTIG
#define OI_TILE_CONTIG 0
#endif
#ifndef OI_BF16_SIB_DEFAULT
#define OI_BF16_SIB_DEFAULT 2 // (round 5: 256-query batches at d = 1024 take their two passes as sibling workgroups on one XCD)
#endif
#ifndef CB_QUAD_NBUF
#define CB_QUAD_NBUF 6 // ring slots per wave of the quad kernel (4 = one tile's worth, rounds 2-4; 6 is what the LDS holds)
#endif

typedef float cb_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 cb_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t cb_u32x4 __attribute__((ext_vector_type(4)));

#define CB_TILE_ROWS 32
#define CB_SLOT_K 64                 // bf16 of K per ring slot row (128 B)
#define CB_SLOT_BYTES (CB_TILE_ROWS * 128)

__device__ __forceinline__ uint32_t cb_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ cb_u32x4 cb_make_srd(const uint16_t *base, uint64_t bytes) {
    const uint64_t b = (uint64_t)base;
    cb_u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xFFFFu); // stride 0
    r[2] = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes));
    r[3] = 0x00020000u;
    return r;
}
// One 1-KiB LDS-DMA piece (8 rows x 128 B).  Lanes past the descriptor's end read as zero: the ragged
// last tile needs no clamping.  hipcc does not see these loads: they are ordered by cb_wait<N>().
// STREAM = true: the once-read policy of oi_device.h (non-temporal).  STREAM = false: the default cache policy -- the quad
// kernel's sibling workgroups (below) read every tile TWICE on one XCD and want the first read to stay in its L2.
template <bool STREAM = true>
__device__ __forceinline__ void cb_issue_piece(const cb_u32x4 &srd, uint32_t voff, uint32_t soff, uint32_t lds_dst,
                                               bool skip) {
    if (skip) return;
    uint32_t keep;
    const uint32_t d = __builtin_amdgcn_readfirstlane(lds_dst);
    const uint32_t so = __builtin_amdgcn_readfirstlane(soff);
    if constexpr (STREAM)
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %1, %2, %3 offen " OI_DMA_NT "lds\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(srd), "s"(so), "s"(d)
            : "memory");
    else
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(srd), "s"(so), "s"(d)
            : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void cb_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cb_static_for<I + 1, N>(f);
    }
}
// MFMA with the B operand named as AGPRs.  gfx950's matrix instructions take srcA / srcB from either half of the unified
// register file, but the builtin lets the compiler choose: with more than 256 registers of resident queries it parks the
// excess in AGPRs as SPILL slots and copies four registers back with v_accvgpr_read in front of every MFMA that needs them
// (quad kernel: 304 copies in the tile loop, 8 per group of four MFMAs -- as much vector issue time as the epilogue).  Here the
// whole query block lives in AGPRs and is read from there; the accumulators stay in VGPRs (the epilogue stores them to LDS).
// The compiler does not see the matrix pipe's latency through an asm: a reader of `acc` other than the next cb_mfma_agpr of
// the same accumulator must come after cb_mfma_drain().
__device__ __forceinline__ void cb_mfma_agpr(cb_f32x16 &acc, const cb_bf16x8 &a, const cb_bf16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
__device__ __forceinline__ void cb_mfma_agpr_first(cb_f32x16 &acc, const cb_bf16x8 &a, const cb_bf16x8 &b) { // acc = a x b
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
}
__device__ __forceinline__ void cb_mfma_drain() { // >= 19 wait states: the last MFMA's result is readable by any instruction
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
}

template <int N>
__device__ __forceinline__ void cb_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int D, int NQT>
__global__ __launch_bounds__(256, 1) void cosine_bf16_filter(
    const uint16_t *__restrict__ rows, uint64_t row_begin, uint64_t row_end,
    const uint16_t *__restrict__ queries, // bf16 [32*NQT][D], zero padded
    uint32_t n_queries, uint32_t doc_id_base, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride,
    const uint32_t *tau_keys, uint64_t pool_stride, uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow) {
    constexpr int NKC = D / CB_SLOT_K;                    // ring slots per tile
    constexpr int NBUF = NKC % 8 == 0 ? 8 : (NKC % 6 == 0 ? 6 : NKC);
    constexpr int P = NBUF - 1;                           // slots in flight ahead of the one being consumed
    constexpr int KSTEPS = D / 16;                        // MFMA groups per tile
    static_assert(D % CB_SLOT_K == 0 && NKC % NBUF == 0 && P >= 1 && P < NKC, "unsupported D");
    static_assert(NQT * KSTEPS * 4 <= 400, "the query block must fit the register file");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *ring = smem;                                                       // [4][NBUF][4 KiB]
    uint32_t *seg_fill = reinterpret_cast<uint32_t *>(smem + 4 * NBUF * CB_SLOT_BYTES); // [32*NQT]

    OI_CLAIM_WHOLE_SIMD(); // (MFMA kernel: nothing else may run on this CU -- oi_device.h)
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t li = lane & 31, lh = lane >> 5;

    // ---- every query over the whole K, in registers for the whole launch: B[k = 16 s + 8 lh + 0..7][n = li]
    cb_bf16x8 qreg[NQT][KSTEPS];
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            qreg[t][s] = *reinterpret_cast<const cb_bf16x8 *>(queries + (uint64_t)(32 * t + li) * D + 16 * s + 8 * lh);
    uint32_t tau[NQT]; // thresholds of the queries this lane filters
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const uint32_t q = 32u * t + li;
        tau[t] = q < n_queries ? tau_keys[q] : 0xFFFFFFFFu;
    }
    if (tid < 32 * NQT) seg_fill[tid] = 0;
    __syncthreads(); // the only barrier: seg_fill is zero before any wave appends

    // ---- tiles of this WAVE: (blockIdx.x * 4 + w), + 4 * gridDim.x, ...
    const uint64_t n_rows = row_end - row_begin;
    const uint64_t n_tiles = (n_rows + CB_TILE_ROWS - 1) / CB_TILE_ROWS;
    const uint64_t first = (uint64_t)blockIdx.x * 4 + w, stride = (uint64_t)gridDim.x * 4;
    const uint64_t my_nt = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    uint64_t *my_seg = pools + carry_cap + (uint64_t)blockIdx.x * seg_cap;

    if (my_nt) {
        // per-lane source of the 4 DMA pieces of a slot: piece m covers tile rows 8m..8m+7; lane l -> row
        // 8m + (l>>3), physical 16-B column l&7 holding LOGICAL column (l&7) ^ ((row>>1)&7)
        uint32_t voff[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint32_t prow = 8 * m + (lane >> 3);
            voff[m] = prow * (uint32_t)(D * 2) + (((lane & 7) ^ ((prow >> 1) & 7)) << 4);
        }
        const uint32_t ring_w = cb_lds_addr(ring) + w * (NBUF * CB_SLOT_BYTES);
        const unsigned char *ring_rd = ring + w * (NBUF * CB_SLOT_BYTES);
        // fragment read address inside a slot: row li, logical 16-B column (2g + lh)
        uint32_t frag_off[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) frag_off[g] = li * 128 + (((2 * g + lh) ^ ((li >> 1) & 7)) << 4);

        auto tile_row0 = [&](uint64_t ti) { return row_begin + (first + ti * stride) * (uint64_t)CB_TILE_ROWS; };
        auto tile_srd = [&](uint64_t ti) {
            const uint64_t r0 = tile_row0(ti);
            return cb_make_srd(rows + r0 * D, (row_end - r0) * (uint64_t)(D * 2));
        };
        cb_u32x4 cur = tile_srd(0), nxt = tile_srd(my_nt > 1 ? 1 : 0);
        // Every load hipcc knows about (queries, thresholds) is retired HERE, with a wait it models:
        // otherwise it re-waits for them inside the tile loop and drains the DMA ring.
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only
#pragma unroll
        for (int kc = 0; kc < P; ++kc) // prologue: slots 0..P-1 of the first tile
#pragma unroll
            for (int m = 0; m < 4; ++m)
                cb_issue_piece(cur, voff[m], kc * 128, ring_w + (kc % NBUF) * CB_SLOT_BYTES + m * 1024, false);

        for (uint64_t ti = 0; ti < my_nt; ++ti) {
            const bool has_next_tile = ti + 1 < my_nt;
            cb_f32x16 acc[NQT];
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

            // Slot s of this tile lives in ring buffer s % NBUF.  Per MFMA group (kc, g): read the next
            // fragment, NQT MFMAs on the current one, and DMA piece g of slot kc + P into the buffer slot
            // kc - 1 has vacated; at g == 3 the next fragment is (kc + 1, 0), behind the counted wait that
            // retires slot kc + 1 (P - 1 younger slots stay in flight).
            cb_wait<4 * (P - 1)>();
            cb_bf16x8 a_cur = *reinterpret_cast<const cb_bf16x8 *>(ring_rd + frag_off[0]);
            cb_static_for<0, NKC * 4>([&](auto gi_) {
                constexpr int gi = decltype(gi_)::value;
                constexpr int kc = gi / 4, g = gi % 4;
                constexpr int sn = kc + P; // slot refilled during this slot's groups
                cb_bf16x8 a_nxt = a_cur;
                if constexpr (g < 3)
                    a_nxt = *reinterpret_cast<const cb_bf16x8 *>(ring_rd + (kc % NBUF) * CB_SLOT_BYTES + frag_off[g + 1]);
#pragma unroll
                for (int t = 0; t < NQT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur, qreg[t][gi], acc[t], 0, 0, 0);
                if constexpr (sn < NKC)
                    cb_issue_piece(cur, voff[g], sn * 128, ring_w + (sn % NBUF) * CB_SLOT_BYTES + g * 1024, false);
                else
                    cb_issue_piece(nxt, voff[g], (sn - NKC) * 128, ring_w + (sn % NBUF) * CB_SLOT_BYTES + g * 1024,
                                   !has_next_tile);
                if constexpr (g == 3 && kc + 1 < NKC) {
                    if (kc + P < NKC || has_next_tile) cb_wait<4 * (P - 1)>();
                    else cb_wait<4 * (NKC - 2 - kc)>();
                    a_nxt = *reinterpret_cast<const cb_bf16x8 *>(ring_rd + ((kc + 1) % NBUF) * CB_SLOT_BYTES + frag_off[0]);
                }
                a_cur = a_nxt;
            });

            // ---- filter + append, straight out of the accumulators: register r of query tile t holds
            // D[row (r&3) + 8 (r>>2) + 4 lh][query 32 t + li]
            const uint64_t row0 = tile_row0(ti);
#pragma unroll
            for (int t = 0; t < NQT; ++t) {
                const uint32_t q = 32u * t + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float s = acc[t][r];
                    if (row < row_end && s == s && oi_f32_key(s) >= tau[t]) {
                        const uint32_t pos = atomicAdd(&seg_fill[q], 1u); // LDS
                        if (pos < seg_cap) my_seg[(uint64_t)q * pool_stride + pos] = oi_rank_key(s, doc_id_base + (uint32_t)row);
                        else *overflow = 1u;
                    }
                }
            }
            cur = nxt;
            if (ti + 2 < my_nt) nxt = tile_srd(ti + 2);
        }
    }
    __syncthreads(); // every wave's appends are counted
    if (tid < 32 * NQT && tid < n_queries) {
        const uint32_t c = seg_fill[tid];
        seg_cnt[(uint64_t)tid * seg_cnt_stride + blockIdx.x] = c < seg_cap ? c : seg_cap;
    }
}

// ------------------------------------------------------------------ more queries per corpus pass
// The kernel above reads the corpus once per 64 queries (32 at d = 1024): at B = 256 that is 4-8 passes
// of an HBM-bound scan.  Here the four waves work as TWO PAIRS; a pair owns whole 32-row tiles and each
// of its waves holds one HALF of K of up to 128 queries in registers (128 x 384 bf16 = 96 KB), streams
// its half of the rows through its own ring, and the two partial 32 x 128 tiles meet once per tile:
// one wave writes its accumulators to LDS, one workgroup barrier, the other adds them to its own and
// filters.  The roles alternate from tile to tile, which balances the epilogue work and makes the one
// barrier enough (a wave has finished reading the buffer before it becomes the writer).
template <int D, int NQT>
__global__ __launch_bounds__(256, 1) void cosine_bf16_pair(
    const uint16_t *__restrict__ rows, uint64_t row_begin, uint64_t row_end,
    const uint16_t *__restrict__ queries, // bf16 [32*NQT][D], zero padded
    uint32_t n_queries, uint32_t doc_id_base, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride,
    const uint32_t *tau_keys, uint64_t pool_stride, uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow) {
    constexpr int KH = D / 2;                 // K of one wave
    constexpr int NKC = KH / CB_SLOT_K;       // ring slots per tile and wave
    constexpr int NBUF = NKC;
    constexpr int P = NBUF - 1;
    constexpr int KSTEPS = KH / 16;
    constexpr int RED = NQT * 16 * 64;        // floats of one partial tile
    static_assert(KH % CB_SLOT_K == 0 && P >= 1, "unsupported D");
    static_assert(NQT * KSTEPS * 4 <= 400, "the query block must fit the register file");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *ring = smem;                                                      // [4][NBUF][4 KiB]
    float *red = reinterpret_cast<float *>(smem + 4 * NBUF * CB_SLOT_BYTES);         // [2 pairs][RED]
    uint32_t *seg_fill = reinterpret_cast<uint32_t *>(red + 2 * RED);                // [32*NQT]

    OI_CLAIM_WHOLE_SIMD(); // (MFMA kernel: nothing else may run on this CU -- oi_device.h)
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t pr = w >> 1, kh = w & 1;
    const uint32_t li = lane & 31, lh = lane >> 5;

    cb_bf16x8 qreg[NQT][KSTEPS]; // this wave's half of K of every query
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            qreg[t][s] = *reinterpret_cast<const cb_bf16x8 *>(queries + (uint64_t)(32 * t + li) * D + kh * KH + 16 * s + 8 * lh);
    uint32_t tau[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const uint32_t q = 32u * t + li;
        tau[t] = q < n_queries ? tau_keys[q] : 0xFFFFFFFFu;
    }
    if (tid < 32 * NQT) seg_fill[tid] = 0;
    __syncthreads();

    // ---- tiles of this PAIR: (blockIdx.x * 2 + pr), + 2 * gridDim.x, ...  Both pairs run the same number of
    // loop trips (the barrier is workgroup-wide); a pair past its last tile idles through the trip.
    const uint64_t n_rows = row_end - row_begin;
    const uint64_t n_tiles = (n_rows + CB_TILE_ROWS - 1) / CB_TILE_ROWS;
    const uint64_t first = (uint64_t)blockIdx.x * 2 + pr, stride = (uint64_t)gridDim.x * 2;
    const uint64_t my_nt = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    const uint64_t first0 = (uint64_t)blockIdx.x * 2;
    const uint64_t trips = first0 < n_tiles ? (n_tiles - first0 + stride - 1) / stride : 0; // pair 0 has the most
    uint64_t *my_seg = pools + carry_cap + (uint64_t)blockIdx.x * seg_cap;
    float *my_red = red + pr * RED;

    uint32_t voff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const uint32_t prow = 8 * m + (lane >> 3);
        voff[m] = prow * (uint32_t)(D * 2) + kh * (uint32_t)(KH * 2) + (((lane & 7) ^ ((prow >> 1) & 7)) << 4);
    }
    const uint32_t ring_w = cb_lds_addr(ring) + w * (NBUF * CB_SLOT_BYTES);
    const unsigned char *ring_rd = ring + w * (NBUF * CB_SLOT_BYTES);
    uint32_t frag_off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) frag_off[g] = li * 128 + (((2 * g + lh) ^ ((li >> 1) & 7)) << 4);

    auto tile_row0 = [&](uint64_t ti) { return row_begin + (first + ti * stride) * (uint64_t)CB_TILE_ROWS; };
    auto tile_srd = [&](uint64_t ti) {
        const uint64_t r0 = tile_row0(ti < my_nt ? ti : 0);
        return cb_make_srd(rows + r0 * D, (row_end - r0) * (uint64_t)(D * 2));
    };
    cb_u32x4 cur = tile_srd(0), nxt = tile_srd(1);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only: retire every load hipcc knows about, here
    if (my_nt) {
#pragma unroll
        for (int kc = 0; kc < P; ++kc)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                cb_issue_piece(cur, voff[m], kc * 128, ring_w + (kc % NBUF) * CB_SLOT_BYTES + m * 1024, false);
    }

    for (uint64_t ti = 0; ti < trips; ++ti) {
        const bool active = ti < my_nt;
        const bool has_next_tile = ti + 1 < my_nt;
        cb_f32x16 acc[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        if (active) {
            cb_wait<4 * (P - 1)>();
            cb_bf16x8 a_cur = *reinterpret_cast<const cb_bf16x8 *>(ring_rd + frag_off[0]);
            cb_static_for<0, NKC * 4>([&](auto gi_) {
                constexpr int gi = decltype(gi_)::value;
                constexpr int kc = gi / 4, g = gi % 4;
                constexpr int sn = kc + P;
                cb_bf16x8 a_nxt = a_cur;
                if constexpr (g < 3)
                    a_nxt = *reinterpret_cast<const cb_bf16x8 *>(ring_rd + (kc % NBUF) * CB_SLOT_BYTES + frag_off[g + 1]);
#pragma unroll
                for (int t = 0; t < NQT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur, qreg[t][gi], acc[t], 0, 0, 0);
                if constexpr (sn < NKC)
                    cb_issue_piece(cur, voff[g], sn * 128, ring_w + (sn % NBUF) * CB_SLOT_BYTES + g * 1024, false);
                else
                    cb_issue_piece(nxt, voff[g], (sn - NKC) * 128, ring_w + (sn % NBUF) * CB_SLOT_BYTES + g * 1024,
                                   !has_next_tile);
                if constexpr (g == 3 && kc + 1 < NKC) {
                    if (kc + P < NKC || has_next_tile) cb_wait<4 * (P - 1)>();
                    else cb_wait<4 * (NKC - 2 - kc)>();
                    a_nxt = *reinterpret_cast<const cb_bf16x8 *>(ring_rd + ((kc + 1) % NBUF) * CB_SLOT_BYTES + frag_off[0]);
                }
                a_cur = a_nxt;
            });
        }
        // ---- the pair's two halves meet: the writer of this trip parks its partial tile in LDS
        const bool writer = ((ti + kh) & 1) == 0;
        if (active && writer) {
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) my_red[(t * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (active && !writer) {
            const uint64_t row0 = tile_row0(ti);
#pragma unroll
            for (int t = 0; t < NQT; ++t) {
                const uint32_t q = 32u * t + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float other = my_red[(t * 16 + r) * 64 + lane];
                    const float s = kh == 0 ? acc[t][r] + other : other + acc[t][r]; // (K half 0) + (K half 1)
                    if (row < row_end && s == s && oi_f32_key(s) >= tau[t]) {
                        const uint32_t pos = atomicAdd(&seg_fill[q], 1u); // LDS
                        if (pos < seg_cap) my_seg[(uint64_t)q * pool_stride + pos] = oi_rank_key(s, doc_id_base + (uint32_t)row);
                        else *overflow = 1u;
                    }
                }
            }
        }
        cur = nxt;
        nxt = tile_srd(ti + 2);
    }
    __syncthreads();
    if (tid < 32 * NQT && tid < n_queries) {
        const uint32_t c = seg_fill[tid];
        seg_cnt[(uint64_t)tid * seg_cnt_stride + blockIdx.x] = c < seg_cap ? c : seg_cap;
    }
}

// ------------------------------------------------------------------ 128 queries per corpus pass
// configs[4] is 256 queries at d = 1024: 32 per pass with the solo kernel, 96 with the pair kernel (three passes).  Here
// the four waves of a workgroup share EVERY 32-row tile, each holding one QUARTER of K of 128 queries in registers
// (128 x 256 bf16 = 64 KB = 256 VGPRs) and streaming its quarter of the rows through its own ring: two passes.  The four
// partial 32 x 128 tiles meet once per tile, and the epilogue is split four ways too: wave w owns query tile w -- it parks
// the three partial tiles it does not own in LDS, and after one barrier adds the other waves' partials of ITS tile to its
// own (K quarters in order 0..3) and filters.  A second barrier lets the buffer be reused (the four epilogues are the
// same size, so nobody waits long at it; the DMA ring keeps the loads of the next tile in flight across both).
template <int D, int DBG = 0, int SIB = 0> // DBG (ablation builds): 1 = no reduction / epilogue / barriers (streaming + MFMA only), 2 = no epilogue work
__global__ __launch_bounds__(256, 1) void cosine_bf16_quad(
    const uint16_t *__restrict__ rows, uint64_t row_begin, uint64_t row_end,
    const uint16_t *__restrict__ queries, // bf16 [128 (x 2 with siblings)][D], zero padded
    uint32_t n_queries, uint32_t doc_id_base, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride,
    const uint32_t *tau_keys, uint64_t pool_stride, uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow) {
    constexpr int NQT = 4;
    constexpr uint32_t sib = SIB;
    constexpr bool STREAM = SIB == 0; // siblings: default cache policy (HBM then sees every tile ONCE: 25.63 GB per 256-query batch
                                      // at 12.5M x 1024 by the FETCH_SIZE counter, against 42 GB with non-temporal loads and 51.2 GB without siblings)
    // SIBLINGS (round 5).  256 queries at d = 1024 are two passes of this kernel over the corpus; one CU cannot hold more than
    // 128 queries (the 256 x 1024 bf16 block IS the register file of a CU).  With sib != 0 the two passes run as ONE launch:
    // the grid is cut into pairs of workgroups that walk the SAME tile sequence, one with queries 0..127 and one with 128..255
    // (own pool segments, own thresholds), so a tile fetched from HBM by whichever sibling gets there first is read by the other
    // out of cache -- HBM sees the corpus once per 256 queries instead of twice.  A sibling that hits cache runs faster until
    // it leads and misses: the pair stays together by itself, nothing synchronises them.
    //   sib == 1: siblings are blockIdx 2j, 2j + 1 (dispatched to neighbouring XCDs: they share the memory-side Infinity Cache)
    //   sib == 2: siblings are 16 a + x and 16 a + 8 + x (x = blockIdx % 8: the SAME XCD under round-robin dispatch -- they share its L2)
    uint32_t half = 0, wg = blockIdx.x, n_wg = gridDim.x;
    if (sib == 1) { half = blockIdx.x & 1u; wg = blockIdx.x >> 1; n_wg = gridDim.x >> 1; }
    else if (sib == 2) { half = (blockIdx.x >> 3) & 1u; wg = ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7u); n_wg = gridDim.x >> 1; }
    if (half) { // (uniform) the second 128 queries: their block of every per-query array
        queries += (uint64_t)128 * D;
        pools += (uint64_t)128 * pool_stride;
        seg_cnt += (uint64_t)128 * seg_cnt_stride;
        tau_keys += 128;
        n_queries = n_queries > 128u ? n_queries - 128u : 0u;
    } else if (sib && n_queries > 128u) n_queries = 128u;
    constexpr int KQ = D / 4;                 // K of one wave
    constexpr int NKC = KQ / CB_SLOT_K;       // ring slots per tile and wave
    // Round 5: the ring is SIX slots deep and indexed at run time (as cosine_screen_copy.hip's), where rounds 2-4 had the four
    // slots of one tile: 3 slots = 12 KB in flight per wave against 16 KB consumed per tile, and no refill is issued while a
    // wave parks / waits at the two barriers of a tile -- the kernel was bound by HBM LATENCY, not bandwidth: 12 KB x 1024 waves
    // per 2.2 us of loaded latency = the 5.6 TB/s it streamed at (tools/r05_sib_ab.py: 7080 cycles per tile, 2048 of them MFMA).
    // Five slots ahead (20 KB) is what the LDS has room for once the park buffer holds only the 12 partial tiles that are
    // actually parked (48 KB; it was laid out for 16).
    constexpr int NBUF = CB_QUAD_NBUF;
    constexpr int P = NBUF - 1;
    constexpr int KSTEPS = KQ / 16;
    constexpr int RED = 16 * 64;              // floats of one parked partial query tile (32 rows x 32 queries)
    constexpr uint32_t RING = NBUF * CB_SLOT_BYTES;
    static_assert(KQ % CB_SLOT_K == 0 && P >= NKC && P <= 2 * NKC, "unsupported ring depth");
    static_assert(NQT * KSTEPS * 4 <= 400, "the query block must fit the register file");
    static_assert(NKC * 4 == 16, "the epilogue is spread over 16 MFMA groups");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *ring = smem;                                                      // [4][NBUF][4 KiB]
    float *red = reinterpret_cast<float *>(smem + 4 * RING);                         // [4 writers][3 foreign query tiles][RED]
    uint32_t *seg_fill = reinterpret_cast<uint32_t *>(red + 12 * RED);               // [128]

    OI_CLAIM_WHOLE_SIMD(); // (MFMA kernel: nothing else may run on this CU -- oi_device.h)
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6); // K quarter, and the query tile this wave finishes
    const uint32_t li = lane & 31, lh = lane >> 5;

    cb_bf16x8 qreg[NQT][KSTEPS]; // this wave's quarter of K of every query
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            qreg[t][s] = *reinterpret_cast<const cb_bf16x8 *>(queries + (uint64_t)(32 * t + li) * D + w * KQ + 16 * s + 8 * lh);
    const uint32_t my_q = 32u * w + li;
    const uint32_t my_tau = my_q < n_queries ? tau_keys[my_q] : 0xFFFFFFFFu;
    // (a key of 0 = no threshold yet; -0.0 passes a +0.0 threshold here where the key order would stop it: a harmless extra
    // candidate, the selects work on keys)
    const float tau_f = my_tau == 0u ? -__builtin_inff() : oi_key_f32(my_tau);
    if (tid < 32 * NQT) seg_fill[tid] = 0;
    __syncthreads();

    // ---- tiles of this WORKGROUP (of this sibling pair): wg, + n_wg, ...
    const uint64_t n_rows = row_end - row_begin;
    const uint64_t n_tiles = (n_rows + CB_TILE_ROWS - 1) / CB_TILE_ROWS;
    const uint64_t first = wg, stride = n_wg;
    const uint64_t my_nt = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    uint64_t *my_seg = pools + carry_cap + (uint64_t)wg * seg_cap;

    uint32_t voff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const uint32_t prow = 8 * m + (lane >> 3);
        voff[m] = prow * (uint32_t)(D * 2) + w * (uint32_t)(KQ * 2) + (((lane & 7) ^ ((prow >> 1) & 7)) << 4);
    }
    const uint32_t ring_w = cb_lds_addr(ring) + w * RING;
    const unsigned char *ring_rd = ring + w * RING;
    uint32_t frag_off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) frag_off[g] = li * 128 + (((2 * g + lh) ^ ((li >> 1) & 7)) << 4);

    auto tile_row0 = [&](uint64_t ti) { return row_begin + (first + ti * stride) * (uint64_t)CB_TILE_ROWS; };
    auto tile_srd = [&](uint64_t ti) { // past this workgroup's last tile: an EMPTY descriptor (loads return zeros)
        const uint64_t r0 = tile_row0(ti < my_nt ? ti : 0);
        return cb_make_srd(rows + r0 * D, ti < my_nt ? (row_end - r0) * (uint64_t)(D * 2) : 0ull);
    };
    cb_u32x4 s0 = tile_srd(0), s1 = tile_srd(1), s2 = tile_srd(2);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only: retire every load hipcc knows about, here
    if (my_nt) {
        cb_static_for<0, P>([&](auto j_) { // prologue: logical slots 0..P-1 (tile j / NKC, slot j % NKC) into ring slots 0..P-1
            constexpr int j = decltype(j_)::value;
            constexpr int tj = j / NKC, kj = j % NKC;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                cb_issue_piece<STREAM>(tj == 0 ? s0 : s1, voff[m], kj * 128, ring_w + j * CB_SLOT_BYTES + m * 1024, false);
        });
    }
    // ring offsets (bytes, wave-uniform): the slot being consumed, and the one vacated before it = the refill target
    uint32_t rd_off = 0, wr_off = (NBUF - 1) * CB_SLOT_BYTES;

    // The epilogue of tile i-1 (sum of the four K quarters of this wave's query tile, filter, append) is spread over the
    // 16 MFMA groups of tile i: it runs in the shadow of the matrix pipe and of the DMA instead of holding both up
    // (stand-alone it cost 65 % on top of the streaming loop).  Round 3 cut its LDS and issue traffic (the loop alone
    // streams at 6.4-6.6 TB/s, the round-2 epilogue took it to 5.2):
    //   * a wave keeps the partial of the query tile it OWNS in registers (16 copies per tile) and parks only the three
    //     it does not own -- 48 KB per tile and workgroup instead of 64;
    //   * parked as [writer][query tile][register block rb][lane][4 registers]: 12 ds_write_b128 per wave instead of 64
    //     ds_write_b32, and the owner reads 12 ds_read_b128 (256 B/clk) instead of 64 ds_read_b32 (128 B/clk);
    //   * the three foreign partials of a register block are read in groups 4 rb, 4 rb + 1, 4 rb + 2 (before the MFMAs) and
    //     summed in group 4 rb + 3: score = ((own + p[w+1]) + p[w+2]) + p[w+3] (writers mod 4) -- a fixed order per query
    //     tile, so a score is the same bits run to run (the order depends on the query's slot: see finish_block); ONE branch per block tests max(four scores) >= tau, the per-register
    //     test and the ragged-tile cut are behind it (survivors are rare once a threshold exists).
    uint64_t row0_prev = 0;
    bool have_prev = false;
    uint32_t rows_prev = 0; // rows of the previous tile inside the chunk (32 but for the last)
    typedef float cb_f32x4 __attribute__((ext_vector_type(4)));
    cb_f32x4 *red4 = reinterpret_cast<cb_f32x4 *>(red); // [writer][query tile][rb][lane]
    // [writer x][the three query tiles x does not own, in tile order][rb][lane]: tile t sits at index t (t < x) or t - 1 (t > x)
    cb_f32x4 *park_base = red4 + (w * 12) * 64 + lane;          // + ((t < w ? t : t - 1) * 4 + rb) * 64
    const cb_f32x4 *rd_base[3];                                 // writer x = (w + 1 + j) % 4, query tile w: + rb * 64
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t x = (w + 1 + j) & 3u;
        rd_base[j] = red4 + ((x * 3 + (w < x ? w : w - 1u)) * 4) * 64 + lane;
    }
    cb_f32x16 own; // this wave's partial of its own query tile, previous tile
    auto finish_block = [&](auto rb_, const cb_f32x4 (&p)[3]) {
        constexpr int rb = decltype(rb_)::value;
        float sc[4];
        // own quarter first, then the others in writer order w+1, w+2, w+3 (mod 4): a fixed order per QUERY TILE, so a score is the
        // same bits run to run -- but the K-quarter order depends on the owning wave, i.e. on the query's slot in the batch: the
        // same (query, row) pair can differ in the last ulp between slots (inside the 1e-5 bar; tests/test_gpu_bf16.py compares
        // with the oracle at that bar, not bit for bit).  Summing in K order 0..3 whatever the owner was built in round 4 and
        // cost 26 % of the launch (0.772 -> 0.976 ms: the order is a run-time property of the wave, four variants per score): not kept.
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[i] = ((own[4 * rb + i] + p[0][i]) + p[1][i]) + p[2][i];
        if (DBG == 3) { if (sc[0] + sc[1] + sc[2] + sc[3] == 12345.678f) *overflow = 2u; return; } // sums only
        // tau_f is the threshold as a float (-inf while there is none, NaN for a padded query: every compare false); a NaN
        // score drops out of the max and fails its own compare
        const float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        if (mx >= tau_f) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t rr = (uint32_t)(i + 8 * rb) + 4u * lh; // row of register 4 rb + i inside the tile
                if (rr < rows_prev && sc[i] >= tau_f) {
                    const uint32_t pos = atomicAdd(&seg_fill[my_q], 1u); // LDS
                    if (pos < seg_cap) my_seg[(uint64_t)my_q * pool_stride + pos] = oi_rank_key(sc[i], doc_id_base + (uint32_t)(row0_prev + rr));
                    else *overflow = 1u;
                }
            }
        }
    };

    // The fragments of a WHOLE slot (4 x 16 B per lane) are read one slot ahead of the matrix pipe (round 5): a group is four
    // MFMAs = 128 cycles, and a fragment read issued one group ahead (rounds 2-4) came back later than that whenever the LDS
    // pipe was busy with the four waves' DMA writes and partial-tile traffic -- the wave then sat at lgkmcnt(0) in front of
    // every group (stream + MFMA alone: 4500 cycles per tile for 2048 of MFMA, with HBM no longer the limit once siblings
    // share the tiles).  The ring is continuous across tiles, so the next tile's first slot is read during this tile's last.
    cb_bf16x8 fr_cur[4], fr_nxt[4];
    if (my_nt) {
        cb_wait<4 * (P - 1)>();
#pragma unroll
        for (int g = 0; g < 4; ++g) fr_cur[g] = *reinterpret_cast<const cb_bf16x8 *>(ring_rd + rd_off + frag_off[g]);
    }
    for (uint64_t ti = 0; ti < my_nt; ++ti) {
        cb_f32x16 acc[NQT];
        {
            // Slot kc of this tile sits at rd_off, its fragments in fr_cur.  Per slot: the counted wait that retires slot
            // kc + 1 (P - 2 younger slots stay in flight), its four fragment reads, then per MFMA group (kc, g) four MFMAs and
            // DMA piece g of logical slot kc + P (one or two tiles ahead; an empty descriptor past the last tile: the refill
            // ALWAYS issues, so every counted wait is the same constant) into the slot vacated last (wr_off).
            cb_f32x4 pp[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            cb_static_for<0, NKC * 4>([&](auto gi_) {
                constexpr int gi = decltype(gi_)::value;
                constexpr int kc = gi / 4, g = gi % 4;
                constexpr int sn = kc + P;
                constexpr int tn = sn / NKC, kn = sn % NKC; // tn is 1 or 2 (P >= NKC)
                constexpr int rb = gi / 4, ej = gi % 4; // the previous tile's epilogue: register block rb, step ej
                if constexpr (g == 0) {
                    const uint32_t nx_off = rd_off + CB_SLOT_BYTES == RING ? 0u : rd_off + CB_SLOT_BYTES;
                    cb_wait<4 * (P - 2)>();
#pragma unroll
                    for (int h = 0; h < 4; ++h) fr_nxt[h] = *reinterpret_cast<const cb_bf16x8 *>(ring_rd + nx_off + frag_off[h]);
                }
                if constexpr ((DBG == 0 || DBG == 3) && ej < 3) { // a foreign partial of block rb: read before the MFMAs
                    if (have_prev) pp[ej] = rd_base[ej][rb * 64];
                }
#pragma unroll
                for (int t = 0; t < NQT; ++t) {
                    if constexpr (gi == 0) cb_mfma_agpr_first(acc[t], fr_cur[g], qreg[t][gi]);
                    else cb_mfma_agpr(acc[t], fr_cur[g], qreg[t][gi]);
                }
                cb_issue_piece<STREAM>(tn == 1 ? s1 : s2, voff[g], kn * 128, ring_w + wr_off + g * 1024, false);
                if constexpr ((DBG == 0 || DBG == 3) && ej == 3) { // ... block rb summed, filtered, appended
                    if (have_prev) finish_block(std::integral_constant<int, rb>{}, pp);
                }
                if constexpr (g == 3) {
                    wr_off = rd_off;
                    rd_off = rd_off + CB_SLOT_BYTES == RING ? 0u : rd_off + CB_SLOT_BYTES;
#pragma unroll
                    for (int h = 0; h < 4; ++h) fr_cur[h] = fr_nxt[h];
                }
            });
            cb_mfma_drain(); // the accumulators are read (parked / copied) next
        }
        if (DBG == 1) {
            float x = 0.f;
#pragma unroll
            for (int t = 0; t < NQT; ++t) x += acc[t][0] + acc[t][7] + acc[t][15];
            if (x == 12345.678f) *overflow = 2u;
            s0 = s1;
            s1 = s2;
            s2 = tile_srd(ti + 3);
            continue;
        }
        __syncthreads(); // everyone has consumed the parked partials of the previous tile
        // ---- the four quarters meet: a wave parks the three query tiles it does not own and keeps its own
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            if ((uint32_t)t != w) { // uniform
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    cb_f32x4 v = {acc[t][4 * rb], acc[t][4 * rb + 1], acc[t][4 * rb + 2], acc[t][4 * rb + 3]};
                    park_base[(((uint32_t)t < w ? t : t - 1) * 4 + rb) * 64] = v;
                }
            } else {
                own = acc[t];
            }
        }
        row0_prev = tile_row0(ti);
        rows_prev = row_end - row0_prev < 32 ? (uint32_t)(row_end - row0_prev) : 32u;
        have_prev = true;
        __syncthreads(); // parked: readable during the next tile's MFMA groups
        s0 = s1;
        s1 = s2;
        s2 = tile_srd(ti + 3);
    }
    cb_wait<0>(); // the zero-filling refills issued past the last tile have landed before the LDS goes back
    if ((DBG == 0 || DBG == 3) && have_prev) // the last tile's epilogue has no MFMA loop to hide in
        cb_static_for<0, 4>([&](auto rb_) {
            constexpr int rb = decltype(rb_)::value;
            cb_f32x4 p[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) p[j] = rd_base[j][rb * 64];
            finish_block(rb_, p);
        });
    __syncthreads();
    if (tid < 32 * NQT && tid < n_queries) {
        const uint32_t c = seg_fill[tid];
        seg_cnt[(uint64_t)tid * seg_cnt_stride + wg] = c < seg_cap ? c : seg_cap;
    }
}

// ------------------------------------------------------------------ 128 queries per pass, split by QUERY (round 5)
// The quad kernel above splits K over the four waves, so every 32 x 128 score tile exists as four partial tiles that meet
// through LDS: 48 KB parked and read back per tile, two barriers, three adds per score -- 2.2 of its 8.3 ms per 256-query
// batch at a 12.5M-row shard once sibling workgroups had taken HBM out of the way.  Here the four waves split
// the QUERIES: wave w holds queries 32 w .. 32 w + 31 over the WHOLE K (32 x 1024 bf16 = 64 KB = 256 AGPRs), every tile is
// DMA'd into LDS once (each wave loads a quarter of it) and read by all four waves.  A wave's accumulators then hold complete
// scores: no partial tiles, no reduction, and a score is one MFMA accumulation chain in K order -- the same bits whatever the
// query's slot in the batch.  The price is LDS read traffic: every fragment feeds ONE MFMA instead of four, 256 KB of
// ds_read_b128 per tile and CU = 2048 LDS cycles beside 2048 matrix-pipe cycles per wave; the kernel is built to keep both busy:
//   * the tile moves as two HALF tiles (512 k = 8 slots of 4 KiB, 32 KB) through a ring of CQ_NHB half-tile buffers; wave w
//     issues the DMA pieces of slots 2 w, 2 w + 1 of every half tile, three half tiles ahead;
//   * ONE barrier per half tile does both jobs: every wave has waited for its own pieces of half tile h (counted vmcnt) before
//     it, so after it all of h is in LDS; and every wave has finished reading h - 1, whose buffer is refilled right after;
//   * fragments are read a whole slot (4 x 16 B per lane) ahead of the matrix pipe.
// Sibling workgroups (SIB = 2: two workgroups of one XCD, 128 queries each, same tile sequence, default cache policy) as in
// the quad kernel: HBM sees the corpus once per 256 queries.
#define CQ_HT_SLOTS 8
#define CQ_HT_BYTES (CQ_HT_SLOTS * CB_SLOT_BYTES)
#ifndef CQ_NHB
#define CQ_NHB 4
#endif
#define CQ_STAGE 256
#define CQ_STAGE_FLUSH 64u
#define CQ_LDS (CQ_NHB * CQ_HT_BYTES + 128 * 4 + 4 * CQ_STAGE * 12)

__device__ __forceinline__ uint32_t cq_incl_scan(uint32_t v) { // wave-wide inclusive prefix sum (DPP, no LDS)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

template <int SIB>
__global__ __launch_bounds__(256, 1) void cosine_bf16_qsplit(
    const uint16_t *__restrict__ rows, uint64_t row_begin, uint64_t row_end,
    const uint16_t *__restrict__ queries, // bf16 [128 (x 2 with siblings)][1024], zero padded
    uint32_t n_queries, uint32_t doc_id_base, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride,
    const uint32_t *tau_keys, uint64_t pool_stride, uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow) {
    constexpr int D = 1024, NHB = CQ_NHB;
    constexpr bool STREAM = SIB == 0;
    constexpr uint32_t RING = NHB * CQ_HT_BYTES;
    static_assert(NHB == 4 && CQ_LDS <= 160 * 1024, "the refill of half tile h + NHB is addressed as tile ti + 2: NHB = 4");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *ring = smem;                                                  // [NHB][8 slots][4 KiB]: shared by the four waves
    uint32_t *seg_fill = reinterpret_cast<uint32_t *>(smem + RING);              // [128]
    OI_CLAIM_WHOLE_SIMD(); // (MFMA kernel: nothing else may run on this CU -- oi_device.h)
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);                 // the query tile of this wave
    const uint32_t li = lane & 31, lh = lane >> 5;
    uint64_t *stage_keys = reinterpret_cast<uint64_t *>(smem + RING + 512) + w * CQ_STAGE;
    uint32_t *stage_q = reinterpret_cast<uint32_t *>(smem + RING + 512 + 4 * CQ_STAGE * 8) + w * CQ_STAGE;
    uint32_t st_head = 0, st_n = 0;

    uint32_t half = 0, wg = blockIdx.x, n_wg = gridDim.x; // siblings: see cosine_bf16_quad
    if (SIB == 1) { half = blockIdx.x & 1u; wg = blockIdx.x >> 1; n_wg = gridDim.x >> 1; }
    else if (SIB == 2) { half = (blockIdx.x >> 3) & 1u; wg = ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7u); n_wg = gridDim.x >> 1; }
    if (half) {
        queries += (uint64_t)128 * D;
        pools += (uint64_t)128 * pool_stride;
        seg_cnt += (uint64_t)128 * seg_cnt_stride;
        tau_keys += 128;
        n_queries = n_queries > 128u ? n_queries - 128u : 0u;
    } else if (SIB && n_queries > 128u) n_queries = 128u;

    // ---- this wave's 32 queries over the whole K, in registers: B[k = 16 s + 8 lh + 0..7][n = li]
    cb_bf16x8 qreg[64];
#pragma unroll
    for (int s = 0; s < 64; ++s)
        qreg[s] = *reinterpret_cast<const cb_bf16x8 *>(queries + (uint64_t)(32 * w + li) * D + 16 * s + 8 * lh);
    const uint32_t my_q = 32u * w + li;
    const uint32_t my_tau = my_q < n_queries ? tau_keys[my_q] : 0xFFFFFFFFu;
    const float tau_f = my_tau <= 0x007FFFFFu ? -__builtin_inff() : oi_key_f32(my_tau); // (no query in the slot: NaN, nothing passes)
    if (tid < 128) seg_fill[tid] = 0;

    const uint64_t n_rows = row_end - row_begin;
    const uint64_t n_tiles = (n_rows + CB_TILE_ROWS - 1) / CB_TILE_ROWS;
#if OI_TILE_CONTIG // A/B: every workgroup (pair) owns a CONTIGUOUS run of tiles instead of every n_wg-th tile
    const uint64_t per_ = (n_tiles + n_wg - 1) / n_wg;
    const uint64_t first = (uint64_t)wg * per_, stride = 1;
    const uint64_t my_nt = first < n_tiles ? (n_tiles - first < per_ ? n_tiles - first : per_) : 0;
#else
    const uint64_t first = wg, stride = n_wg;
    const uint64_t my_nt = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
#endif
    uint64_t *my_seg = pools + carry_cap + (uint64_t)wg * seg_cap;

    // per-lane source of the 4 DMA pieces of a slot (piece m: tile rows 8m..8m+7; the swizzle of the other kernels)
    uint32_t voff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const uint32_t prow = 8 * m + (lane >> 3);
        voff[m] = prow * (uint32_t)(D * 2) + (((lane & 7) ^ ((prow >> 1) & 7)) << 4);
    }
    const uint32_t ring_w = cb_lds_addr(ring);
    uint32_t frag_off[4]; // fragment of MFMA group g of a slot: row li, logical 16-B column 2 g + lh
#pragma unroll
    for (int g = 0; g < 4; ++g) frag_off[g] = li * 128 + (((2 * g + lh) ^ ((li >> 1) & 7)) << 4);

    auto tile_row0 = [&](uint64_t ti) { return row_begin + (first + ti * stride) * (uint64_t)CB_TILE_ROWS; };
    auto tile_srd = [&](uint64_t ti) { // past this workgroup's last tile: an EMPTY descriptor (loads return zeros)
        const uint64_t r0 = tile_row0(ti < my_nt ? ti : 0);
        return cb_make_srd(rows + r0 * D, ti < my_nt ? (row_end - r0) * (uint64_t)(D * 2) : 0ull);
    };
    // this wave's pieces of one half tile (slots 2 w and 2 w + 1): K half `hh` of the tile described by `srd`
    auto issue_slot = [&](const cb_u32x4 &srd, uint32_t hh, uint32_t jj, uint32_t buf_off) __attribute__((always_inline)) {
        const uint32_t j = 2u * w + jj;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            cb_issue_piece<STREAM>(srd, voff[m], hh * 1024u + j * 128u, ring_w + buf_off + j * CB_SLOT_BYTES + m * 1024, false);
    };
    cb_u32x4 s0 = tile_srd(0), s1 = tile_srd(1), s2 = tile_srd(2);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only: retire every load hipcc knows about, here
    __syncthreads();                    // seg_fill is zero before any wave appends
    // The slot stream is continuous over half tiles and tiles.  Fragments are read THREE slots ahead of the matrix pipe (a slot
    // is four MFMAs = 128 cycles; one slot ahead, the LDS reads -- four waves' fragment traffic plus the DMA writes on one LDS
    // pipe -- came back after ~250 cycles and every slot waited for them: stream and MFMA time ADDED UP, 5.3 + 2.8 ms per batch
    // at a 12.5M-row shard with and without the matrix instructions).  So the reads of half tile h + 1 begin at slot 5 of half
    // tile h, and the ONE barrier per half tile sits there: before it a wave waits for its own DMA pieces of h + 1 (counted
    // vmcnt) -- after it all of h + 1 is in LDS; and every read of half tile h has been issued (the last ones at slot 4) and,
    // by the barrier's lgkmcnt(0), completed -- its buffer is refilled with half tile h + NHB right after the barrier.
    cb_bf16x8 fr[4][4]; // fragments of slots g, g + 1, g + 2 (and the one being read): slot g lives in fr[g % 4]
    if (my_nt) {
        cb_static_for<0, NHB>([&](auto h_) { // prologue: half tiles 0 .. NHB - 1 (tile h / 2, K half h % 2) fill the ring
            constexpr int h = decltype(h_)::value;
            issue_slot(h / 2 == 0 ? s0 : s1, h % 2, 0, h * CQ_HT_BYTES);
            issue_slot(h / 2 == 0 ? s0 : s1, h % 2, 1, h * CQ_HT_BYTES);
        });
        cb_wait<8 * (NHB - 1)>(); // this wave's pieces of half tile 0 ...
    }
    __syncthreads();              // ... and everyone's
    if (my_nt) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) fr[j][g] = *reinterpret_cast<const cb_bf16x8 *>(ring + j * CB_SLOT_BYTES + frag_off[g]);
    }
    uint32_t rd_off = 0; // the buffer of the half tile the matrix pipe is on

    for (uint64_t ti = 0; ti < my_nt; ++ti) {
        cb_f32x16 acc;
        cb_static_for<0, 2>([&](auto hh_) {
            constexpr int hh = decltype(hh_)::value;
            const uint32_t nx_off = rd_off + CQ_HT_BYTES == RING ? 0u : rd_off + CQ_HT_BYTES;
            cb_static_for<0, CQ_HT_SLOTS>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                if constexpr (j == 5) {
                    cb_wait<8 * (NHB - 2)>(); // this wave's pieces of the NEXT half tile are in LDS ...
                    __syncthreads();          // ... and everyone's; everyone has finished reading THIS half tile's buffer
                }
#if defined(CQ_NOREADS) // (variant builds, timings only: the DMA stream and the barriers alone -- no fragment reads, no matrix pipe)
                if constexpr (hh == 0 && j == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = -1.f - (float)tile_row0(ti) * 1e-6f; // (falls with the row: nothing passes once a threshold stands)
                }
#elif defined(CQ_NOMFMA) // (variant builds, timings only: the stream and the LDS traffic without the matrix pipe)
                if constexpr (hh == 0 && j == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] += (float)fr[j % 4][g][0];
#ifdef CQ_SLEEP // (with CQ_NOMFMA: the wave idles for the slot's 128 matrix cycles instead -- is it the pipe or the time that costs?)
                __builtin_amdgcn_s_sleep(2);
#endif
#else
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (hh == 0 && j == 0) { if (g == 0) cb_mfma_agpr_first(acc, fr[0][0], qreg[0]); else cb_mfma_agpr(acc, fr[0][g], qreg[g]); }
                    else cb_mfma_agpr(acc, fr[j % 4][g], qreg[hh * 32 + j * 4 + g]);
                }
#endif
                // fragments of the slot three ahead: this half tile's, or (from slot 5 on) the next one's.  AFTER this slot's MFMAs
                // in program order: the registers they go to fed the PREVIOUS slot's MFMAs, all issued a slot ago.
#ifndef CQ_NOREADS
                {
                    const unsigned char *src = j + 3 < CQ_HT_SLOTS ? ring + rd_off + (j + 3) * CB_SLOT_BYTES
                                                                   : ring + nx_off + (j + 3 - CQ_HT_SLOTS) * CB_SLOT_BYTES;
#pragma unroll
                    for (int g = 0; g < 4; ++g) fr[(j + 3) % 4][g] = *reinterpret_cast<const cb_bf16x8 *>(src + frag_off[g]);
                }
#endif
                // the refill of THIS half tile's buffer (free since the barrier): half tile h + NHB = K half hh of tile ti + NHB / 2
                if constexpr (j == 5) issue_slot(s2, hh, 0, rd_off);
                if constexpr (j == 6) issue_slot(s2, hh, 1, rd_off);
            });
            rd_off = nx_off;
        });
        cb_mfma_drain(); // the accumulators are read next

        // ---- filter + append (cosine_screen_filter's epilogue with one query tile): register r holds
        // D[row (r&3) + 8 (r>>2) + 4 lh][query 32 w + li]
        const uint64_t row0 = tile_row0(ti);
        uint32_t m = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) m |= acc[r] >= tau_f ? 1u << r : 0u;
        if (row_end - row0 < (uint64_t)CB_TILE_ROWS) { // the ragged last tile: rows past the end read as zeros
            const uint32_t left = (uint32_t)(row_end - row0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * lh >= left) m &= ~(1u << r);
        }
        if (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
            const uint32_t cnt = (uint32_t)__builtin_popcount(m);
            const uint32_t incl = cq_incl_scan(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total <= CQ_STAGE - CQ_STAGE_FLUSH) { // sparse tile: staged in LDS, 64 keys leave with one store instruction
                uint32_t idx = st_head + st_n + incl - cnt;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (m & (1u << r)) {
                        const uint32_t row = (uint32_t)row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        stage_keys[idx & (CQ_STAGE - 1)] = oi_rank_key(acc[r], doc_id_base + row);
                        stage_q[idx & (CQ_STAGE - 1)] = my_q;
                        ++idx;
                    }
                st_n += total;
                while (st_n >= CQ_STAGE_FLUSH) {
                    asm volatile("" ::: "memory");
                    {
                        const uint32_t i_ = (st_head + lane) & (CQ_STAGE - 1);
                        const uint64_t key_ = stage_keys[i_];
                        const uint32_t q_ = stage_q[i_];
                        const uint32_t pos_ = atomicAdd(&seg_fill[q_], 1u);
                        if (pos_ < seg_cap) my_seg[(uint64_t)q_ * pool_stride + pos_] = key_;
                        else *overflow = 1u;
                    }
                    asm volatile("" ::: "memory");
                    st_head = (st_head + CQ_STAGE_FLUSH) & (CQ_STAGE - 1);
                    st_n -= CQ_STAGE_FLUSH;
                }
            } else { // dense tile (no threshold yet): straight to the pool
                uint32_t pos = atomicAdd(&seg_fill[my_q], cnt);
                uint64_t *dst = my_seg + (uint64_t)my_q * pool_stride;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (m & (1u << r)) {
                        const uint32_t row = (uint32_t)row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (pos < seg_cap) dst[pos] = oi_rank_key(acc[r], doc_id_base + row);
                        else *overflow = 1u;
                        ++pos;
                    }
            }
        }
        s0 = s1;
        s1 = s2;
        s2 = tile_srd(ti + 3);
    }
    if (st_n) { // the staged rest
        asm volatile("" ::: "memory");
        if (lane < st_n) {
            const uint32_t i_ = (st_head + lane) & (CQ_STAGE - 1);
            const uint64_t key_ = stage_keys[i_];
            const uint32_t q_ = stage_q[i_];
            const uint32_t pos_ = atomicAdd(&seg_fill[q_], 1u);
            if (pos_ < seg_cap) my_seg[(uint64_t)q_ * pool_stride + pos_] = key_;
            else *overflow = 1u;
        }
        asm volatile("" ::: "memory");
    }
    cb_wait<0>(); // the zero-filling refills issued past the last tile have landed before the LDS goes back
    __syncthreads();
    if (tid < 128 && tid < n_queries) {
        const uint32_t c = seg_fill[tid];
        seg_cnt[(uint64_t)tid * seg_cnt_stride + wg] = c < seg_cap ? c : seg_cap;
    }
}

// ------------------------------------------------------------------ query staging: f32 -> bf16 (RNE), zero padded
__global__ __launch_bounds__(256) void cb_stage_queries(const float *__restrict__ q, uint32_t n_queries, uint32_t n_padded,
                                                        uint32_t dim, uint16_t *__restrict__ out) {
    const uint64_t total = (uint64_t)n_padded * dim;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / dim);
        uint16_t v = 0;
        if (row < n_queries) {
            const uint32_t u = __float_as_uint(q[i]);
            v = (u & 0x7F800000u) == 0x7F800000u ? (uint16_t)(u >> 16)                        // inf / NaN: truncate
                                                 : (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); // round to nearest even
        }
        out[i] = v;
    }
}

// ------------------------------------------------------------------ host
bool oi_cosine_bf16_supported(uint32_t dim) { return dim == 384 || dim == 768 || dim == 1024; }
// Queries of the next corpus pass.  One wave can hold 64 queries over the whole K (32 at d = 1024); a pair
// holds 96 (three query tiles: with four, the 768-d build spills and loses more than the saved pass).  The
// pair kernel is used when it shortens the plan: always at d = 1024, and at 384 / 768 when passes of 96
// cover what is left in fewer passes than passes of 64.
static uint32_t cb_group(uint32_t dim, uint32_t left) {
    const uint32_t solo = dim == 1024 ? 32u : 64u;
    if (left <= solo) return solo;
    if (dim == 1024) return left > 96u ? 128u : 96u; // 128: the quad kernel (a quarter of K per wave)
    return (left + 95u) / 96u < (left + 63u) / 64u ? 96u : 64u;
}

// Pool geometry of one chunk.  Segments are per workgroup; a workgroup's waves take 4 tiles per round
// (solo kernel) or 2 (pair kernel) -- the cap below covers both.
// How 256 queries at d = 1024 take their two passes of 128 (the quad kernel): 0 = two launches one after the other (rounds
// 2-4), 1 / 2 = ONE launch of sibling workgroups sharing every tile through the Infinity Cache / the XCD's L2 (see the kernel).
static uint32_t cb_sibling_mode() {
    const char *e = oi_ablation_env("OI_BF16_SIB");
    return e ? (uint32_t)atoi(e) : OI_BF16_SIB_DEFAULT;
}
void oi_cosine_bf16_geometry(const oi_ctx *ctx, uint64_t n_rows, uint32_t *n_segs, uint32_t *seg_cap, bool siblings) {
    const uint64_t n_tiles = (n_rows + CB_TILE_ROWS - 1) / CB_TILE_ROWS;
    const uint64_t quads = (n_tiles + 3) / 4;
    const uint64_t cus = siblings ? std::max<uint64_t>(8, ((uint64_t)ctx->num_cus / 16) * 8) : (uint64_t)ctx->num_cus; // pairs of workgroups: half the grid, whole XCD rounds
    uint64_t grid = quads < cus ? (quads ? quads : 1) : cus;
    if (siblings && grid >= 8) grid -= grid % 8; // (the same-XCD pairing addresses workgroups in groups of 16 = 8 pairs)
    *n_segs = (uint32_t)grid;
    *seg_cap = (uint32_t)((quads + grid - 1) / grid) * 4 * CB_TILE_ROWS;
}

template <int D, int NQT>
static int launch_bf16(oi_ctx *ctx, const uint16_t *rows, uint64_t row_begin, uint64_t row_end, const uint16_t *q,
                       uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
    constexpr int NKC = D / CB_SLOT_K, NBUF = NKC % 8 == 0 ? 8 : (NKC % 6 == 0 ? 6 : NKC);
    constexpr size_t smem = 4 * NBUF * CB_SLOT_BYTES + 64 * 4;
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(cosine_bf16_filter<D, NQT>), (size_t)(smem)));
    hipLaunchKernelGGL((cosine_bf16_filter<D, NQT>), dim3(p.n_segs), dim3(256), smem, ctx->stream, rows, row_begin,
                       row_end, q, nq, doc_id_base, p.keys, p.seg_cnt, p.seg_cnt_stride, p.tau_keys, p.stride,
                       p.carry_cap, p.seg_cap, p.overflow);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

template <int D, int NQT>
static int launch_bf16_pair(oi_ctx *ctx, const uint16_t *rows, uint64_t row_begin, uint64_t row_end, const uint16_t *q,
                            uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
    constexpr int NKC = D / 2 / CB_SLOT_K;
    constexpr size_t smem = 4 * NKC * CB_SLOT_BYTES + 2 * (NQT * 16 * 64) * 4 + 128 * 4;
    static_assert(smem <= 160 * 1024, "LDS");
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(cosine_bf16_pair<D, NQT>), (size_t)(smem)));
    hipLaunchKernelGGL((cosine_bf16_pair<D, NQT>), dim3(p.n_segs), dim3(256), smem, ctx->stream, rows, row_begin,
                       row_end, q, nq, doc_id_base, p.keys, p.seg_cnt, p.seg_cnt_stride, p.tau_keys, p.stride,
                       p.carry_cap, p.seg_cap, p.overflow);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

template <int SIB>
static int launch_bf16_qsplit(oi_ctx *ctx, const uint16_t *rows, uint64_t row_begin, uint64_t row_end, const uint16_t *q,
                              uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
    const uint32_t grid = SIB ? 2u * p.n_segs : p.n_segs;
    constexpr size_t smem = CQ_LDS;
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(cosine_bf16_qsplit<SIB>), (size_t)(smem)));
    hipLaunchKernelGGL((cosine_bf16_qsplit<SIB>), dim3(grid), dim3(256), smem, ctx->stream, rows, row_begin, row_end, q, nq,
                       doc_id_base, p.keys, p.seg_cnt, p.seg_cnt_stride, p.tau_keys, p.stride, p.carry_cap, p.seg_cap,
                       p.overflow);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

template <int D, int DBG, int SIB>
static int launch_bf16_quad_k(oi_ctx *ctx, const uint16_t *rows, uint64_t row_begin, uint64_t row_end, const uint16_t *q,
                              uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
    const uint32_t grid = SIB ? 2u * p.n_segs : p.n_segs; // (siblings: p.n_segs pairs, one pool segment per pair and query half)
    constexpr size_t smem = 4 * CB_QUAD_NBUF * CB_SLOT_BYTES + 12 * (16 * 64) * 4 + 128 * 4;
    static_assert(smem <= 160 * 1024, "LDS");
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(cosine_bf16_quad<D, DBG, SIB>), (size_t)(smem)));
    hipLaunchKernelGGL((cosine_bf16_quad<D, DBG, SIB>), dim3(grid), dim3(256), smem, ctx->stream, rows, row_begin, row_end, q, nq,
                       doc_id_base, p.keys, p.seg_cnt, p.seg_cnt_stride, p.tau_keys, p.stride, p.carry_cap, p.seg_cap,
                       p.overflow);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

template <int D>
static int launch_bf16_quad(oi_ctx *ctx, const uint16_t *rows, uint64_t row_begin, uint64_t row_end, const uint16_t *q,
                            uint32_t nq, uint32_t doc_id_base, const PoolView &p, uint32_t sib = 0) {
#ifdef OI_ABLATION
    static const char *dbg_s = oi_ablation_env("OI_QUAD_DBG");
    const int dbg = dbg_s ? atoi(dbg_s) : 0;
    if (dbg == 1) return sib ? launch_bf16_quad_k<D, 1, 2>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p)
                             : launch_bf16_quad_k<D, 1, 0>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
    if (dbg == 3) return sib ? launch_bf16_quad_k<D, 3, 2>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p)
                             : launch_bf16_quad_k<D, 3, 0>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
    if (sib == 1) return launch_bf16_quad_k<D, 0, 1>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
#endif
    // Round 5: 128 queries per pass split by QUERY over the waves (cosine_bf16_qsplit) instead of by K.  OI_BF16_QUAD=1 (A/B): the K split.
    static const bool k_split = oi_ablation_env("OI_BF16_QUAD") != nullptr;
    if (D == 1024 && !k_split) return sib ? launch_bf16_qsplit<2>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p)
                                          : launch_bf16_qsplit<0>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
    if (sib) return launch_bf16_quad_k<D, 0, 2>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
    return launch_bf16_quad_k<D, 0, 0>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
}

// All queries of a batch over rows [row_begin, row_end) of a bf16 corpus.  d_queries: f32 [n_queries][dim].
int oi_launch_cosine_bf16_chunk(oi_ctx *ctx, const uint16_t *rows, uint64_t row_begin, uint64_t row_end, uint32_t dim,
                                const float *d_queries, uint32_t n_queries, uint32_t doc_id_base, PoolView &pool) {
    OI_REQUIRE(oi_cosine_bf16_supported(dim), "cosine (bf16 corpus): dim %u not instantiated (384, 768, 1024)", dim);
    // siblings: every group of this batch is a full pair of 128-query passes (256, 512, ... queries at d = 1024) and the chunk
    // is long enough to give every pair of workgroups a tile
    const uint32_t sib_mode = cb_sibling_mode();
    const bool siblings = sib_mode != 0 && dim == 1024 && n_queries >= 256 && n_queries % 256 == 0 &&
                          row_end > row_begin && (row_end - row_begin) >= (uint64_t)CB_TILE_ROWS * ctx->num_cus;
    oi_cosine_bf16_geometry(ctx, row_end > row_begin ? row_end - row_begin : 0, &pool.n_segs, &pool.seg_cap, siblings);
    OI_REQUIRE(pool.n_segs <= pool.seg_cnt_stride && pool.carry_cap + (uint64_t)pool.n_segs * pool.seg_cap <= pool.stride,
               "cosine (bf16 corpus): chunk does not fit the candidate pool");
    if (row_end <= row_begin || n_queries == 0) return OI_OK;
    static const bool solo_only = oi_ablation_env("OI_BF16_SOLO") != nullptr; // A/B: never use the pair kernel
    static const bool no_quad = oi_ablation_env("OI_BF16_NO_QUAD") != nullptr;   // A/B: passes of at most 96 queries at d = 1024
    const uint32_t n_padded = (n_queries + 31u) & ~31u;
    DevBuf &qb = ctx->buf("q_bf16");
    OI_CHECK(qb.ensure(sizeof(uint16_t) * (size_t)(n_padded + 128) * dim));
    {
        const uint64_t total = (uint64_t)n_padded * dim;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 1024);
        hipLaunchKernelGGL(cb_stage_queries, dim3(blocks), dim3(256), 0, ctx->stream, d_queries, n_queries, n_padded, dim,
                           qb.as<uint16_t>());
        OI_HIP_CHECK(hipGetLastError());
    }
    ProfScope ps(ctx, "cosine");
    for (uint32_t q0 = 0; q0 < n_queries;) {
        const uint32_t left = n_queries - q0;
        uint32_t group = solo_only ? (dim == 1024 ? 32u : 64u) : cb_group(dim, left);
        if (no_quad && group > 96u) group = 96u;
        if (siblings) group = 256u;
        const uint32_t nq_here = std::min(group, left);
        const uint32_t nqt = (nq_here + 31u) / 32u; // query tiles of 32 in this launch
        PoolView p = pool;
        p.keys += (uint64_t)q0 * pool.stride;
        p.carry_cnt += q0;
        p.seg_cnt += (uint64_t)q0 * pool.seg_cnt_stride;
        p.tau_keys += q0;
        const uint16_t *qptr = qb.as<uint16_t>() + (uint64_t)q0 * dim;
#define CB_SOLO(DD, T) OI_CHECK((launch_bf16<DD, T>(ctx, rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)))
#define CB_PAIR(DD, T) OI_CHECK((launch_bf16_pair<DD, T>(ctx, rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)))
        const bool pair = nq_here > (dim == 1024 ? 32u : 64u);
        if (dim == 1024) {
            if (nq_here > 128u) OI_CHECK((launch_bf16_quad<1024>(ctx, rows, row_begin, row_end, qptr, nq_here, doc_id_base, p, sib_mode)));
            else if (nq_here > 96u) OI_CHECK((launch_bf16_quad<1024>(ctx, rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)));
            else if (!pair) CB_SOLO(1024, 1);
            else if (nqt == 2) CB_PAIR(1024, 2);
            else CB_PAIR(1024, 3);
        } else if (dim == 768) {
            if (pair) CB_PAIR(768, 3);
            else if (nqt == 1) CB_SOLO(768, 1);
            else CB_SOLO(768, 2);
        } else {
            if (pair) CB_PAIR(384, 3);
            else if (nqt == 1) CB_SOLO(384, 1);
            else CB_SOLO(384, 2);
        }
#undef CB_SOLO
#undef CB_PAIR
        q0 += nq_here;
    }
    return OI_OK;
}
