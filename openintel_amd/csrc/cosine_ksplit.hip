// cosine_ksplit.hip -- the batch (B > 8) cosine scorer for gfx950: exact-f32 MFMA with the
// query block RESIDENT IN REGISTERS and the corpus streamed once through per-wave LDS rings.
//
// Why this shape.  At B = 64, d = 768 the scorer needs 2*64 = 128 flop per corpus byte... per
// 4-byte element, i.e. 32 flop/B: the f32 MFMA peak (157 TF) and the HBM rate (~6.3 TB/s
// achievable) bind at almost the same time (6.3 ms vs 4.9 ms for 10M rows), so the kernel must
// keep the matrix pipes issuing back to back WHILE streaming ~5 TB/s.  A conventional LDS-tiled
// GEMM re-stages the 196 KB query block (it does not fit in 160 KB of LDS) for every row tile.
// Instead:
//   * one persistent workgroup per CU, 4 waves, one per SIMD, each owning the full 512-register
//     file: the K dimension is split over the 4 waves (wave w owns k in [w*D/4, (w+1)*D/4)), so a
//     wave's slice of all 64 queries is 2 * D/8 = 192 VGPRs (d = 768) -- loaded once per launch;
//   * each wave streams ITS K-slice of the rows into ITS OWN LDS ring (6 x 4 KiB) with
//     global_load_lds_dwordx4 (full 128-B lines, swizzled through the per-lane source address so
//     the ds_read_b128 fragment reads are conflict-free), 5 chunks ahead, ordered only by its own
//     counted s_waitcnt vmcnt -- no barrier in the main loop, and the prefetch runs across tile
//     boundaries, under the epilogue;
//   * per 32-row tile each wave issues D/4/2*2 = 192 v_mfma_f32_32x32x2_f32 (12288 cycles), then
//     the four partial 32x64 tiles are summed through LDS in a fixed order, compared with the
//     per-query threshold, and the survivors stored straight into THIS workgroup's private segment
//     of each query's candidate pool (fill counters in LDS): no global atomic, no returning
//     memory operation, hence nothing that would make a wave drain its DMA ring.
//
// Operand maps (cdna_hip_programming.md section 3): lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31] is accumulator register r.
#include <cstdlib>
#include <type_traits>

#include "oi_device.h"
#include "oi_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define KS_TILE_ROWS 32
#define KS_CHUNK_K 32                       // floats of K per ring slot row (128 B)
#define KS_SLOT_BYTES (KS_TILE_ROWS * KS_CHUNK_K * 4) // 4 KiB

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// Four 1-KiB LDS-DMA pieces = one ring slot.  hipcc does not see these loads: they are ordered
// by ks_wait<N>() below (cdna_hip_programming.md section 5.7).  M0 carries the LDS destination.
__device__ __forceinline__ void ks_issue_slot(const float *p0, const float *p1, const float *p2,
                                              const float *p3, uint32_t lds_dst, bool skip = false) {
    if (skip) return;
    uint32_t keep;
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(lds_dst);
    const uint32_t d1 = d0 + 1024, d2 = d0 + 2048, d3 = d0 + 3072;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, off\n\t"
        "s_mov_b32 m0, %8\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(d0), "s"(d1), "s"(d2), "s"(d3)
        : "memory");
}
// One 1-KiB LDS-DMA piece (8 rows x 128 B), issued one per MFMA group.  buffer_load ... lds with a
// wave-uniform descriptor and a 32-bit per-lane offset: its issue is short enough to hide under a
// 64-cycle matrix instruction (a global_load_lds with 64-bit per-lane addresses is not -- it cost
// ~80 exposed cycles per piece here).  Lanes past the descriptor's end read as zero: the ragged
// last tile needs no clamping.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ks_make_srd(const float *base, uint64_t bytes) {
    const uint64_t b = (uint64_t)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xFFFFu); // stride 0
    r[2] = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes));
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void ks_issue_piece(const u32x4 &srd, uint32_t voff, uint32_t soff, uint32_t lds_dst,
                                               bool skip) {
    if (skip) return;
    uint32_t keep;
    const uint32_t d = __builtin_amdgcn_readfirstlane(lds_dst);
    const uint32_t so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen " OI_DMA_NT "lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(srd), "s"(so), "s"(d)
        : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void ks_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ks_static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void ks_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// keep a value alive without cost (diagnostic builds).  Inline asm with AMDGPU constraints must
// live in __device__ functions: directly inside a __global__ body it silently drops the host stub.
__device__ __forceinline__ void ks_keep(const f32x4 &v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void ks_keep(const f32x16 &v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void ks_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void ks_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int D, int NQT, int DBG>
__global__ __launch_bounds__(256, 1) void cosine_ksplit_filter(
    const float *__restrict__ rows, uint64_t row_begin, uint64_t row_end,
    const float *__restrict__ queries, // [32*NQT][D], zero padded
    uint32_t n_queries, uint32_t doc_id_base, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride,
    const uint32_t *tau_keys, uint64_t pool_stride, uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow) {
    OI_CLAIM_WHOLE_SIMD(); // (MFMA kernel: nothing else may run on this CU -- oi_device.h)
    // DBG (diagnostic instantiations only, results are then wrong): 1 = no DMA, 2 = no MFMA, 4 = no epilogue,
    // 8 = every workgroup re-reads the same 64 tiles (DMA served from L2/MALL instead of HBM)
    constexpr int KS = D / 4;            // K-slice of one wave
    constexpr int NKC = KS / KS_CHUNK_K; // ring slots per tile and wave
    constexpr int NBUF = NKC <= 6 ? NKC : NKC / 2;
    constexpr int P = NBUF - 1;          // slots in flight ahead of the one being consumed
    constexpr int QR = KS / 2;           // query registers per 32-query tile
    static_assert(KS % KS_CHUNK_K == 0 && NKC % NBUF == 0 && P >= 1 && P < NKC, "unsupported D");
    constexpr int RED_FLOATS = NQT * 16 * 64; // one wave's partial tile

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *ring = smem;                                                 // [4][NBUF][4 KiB]
    float *red = reinterpret_cast<float *>(smem + 4 * NBUF * KS_SLOT_BYTES);    // [4][RED_FLOATS]
    uint32_t *seg_fill = reinterpret_cast<uint32_t *>(red + 4 * RED_FLOATS);    // [32*NQT] entries written so far

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t li = lane & 31, lh = lane >> 5;

    // ---- this wave's K-slice of every query, in registers for the whole launch
    float qreg[NQT][QR];
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
        for (int s = 0; s < KS / 8; ++s) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(queries + (uint64_t)(32 * t + li) * D + w * KS + 8 * s + 4 * lh);
            qreg[t][4 * s + 0] = v[0]; qreg[t][4 * s + 1] = v[1]; qreg[t][4 * s + 2] = v[2]; qreg[t][4 * s + 3] = v[3];
        }
    // thresholds of the two queries this thread filters in the epilogue
    uint32_t tau[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const uint32_t q = 32u * t + li;
        tau[t] = q < n_queries ? tau_keys[q] : 0xFFFFFFFFu;
    }
    if (tid < 64) seg_fill[tid] = 0;

    // ---- tiles of this workgroup: blockIdx.x, + gridDim.x, ...
    const uint64_t n_rows = row_end - row_begin;
    const uint64_t n_tiles = (n_rows + KS_TILE_ROWS - 1) / KS_TILE_ROWS;
    const uint64_t my_nt = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_nt == 0) return;
    // this workgroup's segment of query q: pools[q*stride + carry_cap + blockIdx.x*seg_cap ..)
    uint64_t *my_seg = pools + carry_cap + (uint64_t)blockIdx.x * seg_cap;

    // per-lane source of the 4 DMA pieces of a slot: piece m covers tile rows 8m..8m+7;
    // lane l -> row 8m + (l>>3), physical 16-B column l&7 holding LOGICAL column (l&7) ^ ((row>>1)&7)
    uint32_t piece_row[4], piece_col[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        piece_row[m] = 8 * m + (lane >> 3);
        piece_col[m] = ((lane & 7) ^ ((piece_row[m] >> 1) & 7)) * 4 + w * KS; // float offset in the row
    }
    const uint32_t ring_w = lds_addr(ring) + w * (NBUF * KS_SLOT_BYTES);
    const unsigned char *ring_rd = ring + w * (NBUF * KS_SLOT_BYTES);
    // fragment read address inside a slot: row li, logical 16-B column (2g + lh)
    uint32_t frag_off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) frag_off[g] = li * 128 + (((2 * g + lh) ^ ((li >> 1) & 7)) << 4);

    constexpr bool no_dma = DBG & 1, no_mfma = DBG & 2, no_epi = DBG & 4;
    // DMA addressing: a descriptor per tile (wave-uniform base = the tile's first row), one 32-bit
    // per-lane offset per piece (row * row bytes + swizzled column), the slot's K offset in soffset
    uint32_t voff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) voff[m] = piece_row[m] * (uint32_t)(D * 4) + piece_col[m] * 4u;
    auto tile_srd = [&](uint64_t ti) {
        uint64_t tile_id = blockIdx.x + ti * gridDim.x;
        if constexpr ((DBG & 8) != 0) tile_id &= 63;
        const uint64_t r0 = row_begin + tile_id * (uint64_t)KS_TILE_ROWS;
        return ks_make_srd(rows + r0 * D, (row_end - r0) * (uint64_t)(D * 4));
    };
    u32x4 cur = tile_srd(0), nxt = tile_srd(my_nt > 1 ? 1 : 0);
    // Every load hipcc knows about (queries, thresholds) is retired HERE, with a wait it models:
    // otherwise it re-waits for them at the top of the tile loop (vmcnt(1)) and drains the DMA ring.
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only
    // prologue: slots 0..P-1 of the first tile
#pragma unroll
    for (int kc = 0; kc < P; ++kc)
#pragma unroll
        for (int m = 0; m < 4; ++m)
            ks_issue_piece(cur, voff[m], kc * KS_CHUNK_K * 4, ring_w + (kc % NBUF) * KS_SLOT_BYTES + m * 1024, no_dma);

    // ---- deferred epilogue.  A tile's partial sums are written to LDS right after its last MFMA;
    // the cross-wave sum + filter of tile t then rides INSIDE tile t+1's MFMA stream (barrier A at
    // group 2, one output per group after it, barrier B behind the last one), so its LDS round
    // trips and VALU work issue in the shadow of the 64-cycle matrix instructions.
    float *my_red = red + w * RED_FLOATS;
    uint64_t prev_row0 = 0; // first corpus row of the tile whose partials sit in `red`
    auto epi_out = [&](int i) {
        const uint32_t e = tid + 256u * i; // (t = i>>2, r = (e>>6)&15, lane)
        const float s = (red[e] + red[RED_FLOATS + e]) + (red[2 * RED_FLOATS + e] + red[3 * RED_FLOATS + e]);
        const uint32_t t = i >> 2, r = (e >> 6) & 15u;
        const uint32_t q = 32u * t + li;
        const uint64_t row = prev_row0 + (r & 3u) + 8u * (r >> 2) + 4u * lh;
        if (row < row_end && s == s && oi_f32_key(s) >= tau[t]) {
            const uint32_t pos = atomicAdd(&seg_fill[q], 1u); // LDS
            if (pos < seg_cap) my_seg[(uint64_t)q * pool_stride + pos] = oi_rank_key(s, doc_id_base + (uint32_t)row);
            else *overflow = 1u;
        }
    };
    constexpr int EPI_G0 = 3, EPI_GB = EPI_G0 + NQT * 4; // groups of the outputs / of barrier B
    static_assert(EPI_GB <= NKC * 4 - 1, "tile too short to host the deferred epilogue");

    for (uint64_t ti = 0; ti < my_nt; ++ti) {
        const bool has_next_tile = ti + 1 < my_nt;
        const bool have_prev = ti > 0;
        f32x16 acc[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        // Slot s of this tile lives in ring buffer s % NBUF.  Schedule per tile:
        //   start       : wait -> slot 0 landed (slots 0..P-1 were issued during the previous tile),
        //                 read fragment (0,0)
        //   group (kc,g): read the NEXT fragment, 4*NQT MFMAs on the current one, and -- between
        //                 them -- DMA piece g of slot kc+P into the buffer slot kc-1 has vacated;
        //                 at g == 3 the next fragment is (kc+1,0), behind the counted wait that
        //                 retires slot kc+1 (P-1 younger slots stay in flight).
        ks_wait<4 * (P - 1)>();
        f32x4 a_cur = *reinterpret_cast<const f32x4 *>(ring_rd + frag_off[0]);
        ks_static_for<0, NKC * 4>([&](auto gi_) {
            constexpr int gi = decltype(gi_)::value;
            constexpr int kc = gi / 4, g = gi % 4;
            constexpr int sn = kc + P; // slot refilled during this slot's groups
            f32x4 a_nxt = a_cur;
            if constexpr (g < 3)
                a_nxt = *reinterpret_cast<const f32x4 *>(ring_rd + (kc % NBUF) * KS_SLOT_BYTES + frag_off[g + 1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (!no_mfma) {
#pragma unroll
                    for (int t = 0; t < NQT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[j], qreg[t][(kc * 4 + g) * 4 + j], acc[t], 0, 0, 0);
                } else {
                    ks_keep(a_cur);
                }
                if (j == 0) { // one DMA piece per group, right behind the group's first MFMAs
                    if constexpr (sn < NKC)
                        ks_issue_piece(cur, voff[g], sn * KS_CHUNK_K * 4, ring_w + (sn % NBUF) * KS_SLOT_BYTES + g * 1024,
                                       no_dma);
                    else
                        ks_issue_piece(nxt, voff[g], (sn - NKC) * KS_CHUNK_K * 4,
                                       ring_w + (sn % NBUF) * KS_SLOT_BYTES + g * 1024, no_dma || !has_next_tile);
                }
            }
            if constexpr (!no_epi) {
                if constexpr (gi == EPI_G0 - 1) { if (have_prev) ks_barrier(); }                 // (A) partials visible
                if constexpr (gi >= EPI_G0 && gi < EPI_GB) { if (have_prev) epi_out(gi - EPI_G0); }
                if constexpr (gi == EPI_GB) { if (have_prev) ks_barrier(); }                     // (B) `red` is free again
            }
            if constexpr (g == 3 && kc + 1 < NKC) {
                if (kc + P < NKC || has_next_tile) ks_wait<4 * (P - 1)>();
                else ks_wait<4 * (NKC - 2 - kc)>();
                a_nxt = *reinterpret_cast<const f32x4 *>(ring_rd + ((kc + 1) % NBUF) * KS_SLOT_BYTES + frag_off[0]);
            }
            a_cur = a_nxt;
        });

        if constexpr (no_epi) {
#pragma unroll
            for (int t = 0; t < NQT; ++t) ks_keep(acc[t]);
        } else {
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) my_red[(t * 16 + r) * 64 + lane] = acc[t][r];
            uint64_t tile_id = blockIdx.x + ti * gridDim.x;
            prev_row0 = row_begin + tile_id * (uint64_t)KS_TILE_ROWS;
        }
        cur = nxt;
        if (ti + 2 < my_nt) nxt = tile_srd(ti + 2);
    }
    if constexpr (!no_epi) { // the last tile's epilogue has no MFMA stream to hide in
        ks_barrier();
#pragma unroll
        for (int i = 0; i < NQT * 4; ++i) epi_out(i);
        ks_barrier();
    }
    // publish this segment's fill counts (LDS atomics of every wave are complete after barrier B)
    if (tid < 32 * NQT && tid < n_queries) {
        const uint32_t c = seg_fill[tid];
        seg_cnt[(uint64_t)tid * seg_cnt_stride + blockIdx.x] = c < seg_cap ? c : seg_cap;
    }
}

// =====================================================================================
// Same kernel on the 16x16x4 MFMA shape (v_mfma_f32_16x16x4_f32: 32-cycle issue, 4 accumulator
// registers, same FLOP/cycle).  Under the chip's power limit the two shapes can hold different
// clocks (MI355X_MICROARCH.md "DVFS give-back" item 7), so both are built and the faster one by wall
// time is used.  Operand map: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// D[(l>>4)*4 + r][l&15] is accumulator register r.  A wave's 32x64 tile = 2 row tiles x 4 query tiles.
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int D, int NQT>
__global__ __launch_bounds__(256, 1) void cosine_ksplit16_filter(
    const float *__restrict__ rows, uint64_t row_begin, uint64_t row_end,
    const float *__restrict__ queries, // [32*NQT][D], zero padded
    uint32_t n_queries, uint32_t doc_id_base, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride,
    const uint32_t *tau_keys, uint64_t pool_stride, uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow,
    const uint32_t *run_gate) { // run_gate != null: part of the gated exact pipeline (cosine_prefilter.hip)
    if (run_gate && *run_gate == 0u) return;
    constexpr int KS = D / 4, NKC = KS / KS_CHUNK_K, NBUF = NKC <= 6 ? NKC : NKC / 2, P = NBUF - 1;
    constexpr int NQ16 = 2 * NQT;        // query tiles of 16
    constexpr int QR = KS / 4;           // query registers per tile (one per k-step of 4)
    constexpr int NG = NKC * 2;          // MFMA groups (16 k each) per tile
    static_assert(KS % KS_CHUNK_K == 0 && NKC % NBUF == 0 && P >= 1 && P < NKC, "unsupported D");
    constexpr int RED_FLOATS = NQT * 16 * 64;

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *ring = smem;
    float *red = reinterpret_cast<float *>(smem + 4 * NBUF * KS_SLOT_BYTES);
    uint32_t *seg_fill = reinterpret_cast<uint32_t *>(red + 4 * RED_FLOATS);

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t li = lane & 15, kk = lane >> 4;

    // k-step (s, e) of this lane covers k = w*KS + 16 s + 4 kk + e  (any partition of k works as long
    // as A and B agree): both operands are then contiguous float4 per lane.
    float qreg[NQ16][QR];
#pragma unroll
    for (int t = 0; t < NQ16; ++t)
#pragma unroll
        for (int sg = 0; sg < KS / 16; ++sg) {
    OI_CLAIM_WHOLE_SIMD(); // (MFMA kernel: nothing else may run on this CU -- oi_device.h)
            const f32x4 v = *reinterpret_cast<const f32x4 *>(queries + (uint64_t)(16 * t + li) * D + w * KS + 16 * sg + 4 * kk);
            qreg[t][4 * sg + 0] = v[0]; qreg[t][4 * sg + 1] = v[1]; qreg[t][4 * sg + 2] = v[2]; qreg[t][4 * sg + 3] = v[3];
        }
    uint32_t tau[NQ16];
#pragma unroll
    for (int t = 0; t < NQ16; ++t) {
        const uint32_t q = 16u * t + li;
        tau[t] = q < n_queries ? tau_keys[q] : 0xFFFFFFFFu;
    }
    if (tid < 64) seg_fill[tid] = 0;

    const uint64_t n_rows = row_end - row_begin;
    const uint64_t n_tiles = (n_rows + KS_TILE_ROWS - 1) / KS_TILE_ROWS;
    const uint64_t my_nt = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_nt == 0) return;
    uint64_t *my_seg = pools + carry_cap + (uint64_t)blockIdx.x * seg_cap;

    uint32_t voff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const uint32_t prow = 8 * m + (lane >> 3);
        const uint32_t pcol = ((lane & 7) ^ ((prow >> 1) & 7)) * 4 + w * KS;
        voff[m] = prow * (uint32_t)(D * 4) + pcol * 4u;
    }
    const uint32_t ring_w = lds_addr(ring) + w * (NBUF * KS_SLOT_BYTES);
    const unsigned char *ring_rd = ring + w * (NBUF * KS_SLOT_BYTES);
    // fragment (row tile rt, group g in the slot): row 16 rt + li, logical 16-B column 4 g + kk
    uint32_t frag_off[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t row = 16 * rt + li;
            frag_off[rt][g] = row * 128 + (((4 * g + kk) ^ ((row >> 1) & 7)) << 4);
        }
    // Past the workgroup's last tile the descriptor is EMPTY (its loads return zeros): the tile loop then needs no
    // branch around the prefetch and a single set of wait counts -- a branch-free tile body.
    auto tile_srd = [&](uint64_t ti) {
        const uint64_t r0 = row_begin + (blockIdx.x + (ti < my_nt ? ti : 0) * gridDim.x) * (uint64_t)KS_TILE_ROWS;
        return ks_make_srd(rows + r0 * D, ti < my_nt ? (row_end - r0) * (uint64_t)(D * 4) : 0ull);
    };
    u32x4 cur = tile_srd(0), nxt = tile_srd(1);
    __builtin_amdgcn_s_waitcnt(0x0F70); // retire every load hipcc knows about before the DMA ring starts
#pragma unroll
    for (int kc = 0; kc < P; ++kc)
#pragma unroll
        for (int m = 0; m < 4; ++m)
            ks_issue_piece(cur, voff[m], kc * KS_CHUNK_K * 4, ring_w + (kc % NBUF) * KS_SLOT_BYTES + m * 1024, false);

    float *my_red = red + w * RED_FLOATS;
    uint64_t prev_row0 = 0;
    bool have_prev = false; // false during the first tile: `red` holds nothing yet
    // element e = tid + 256 i of a partial tile is ((rt*NQ16 + t)*4 + reg)*64 + lane with
    // reg = tid>>6, t = i % NQ16, rt = i / NQ16
    auto epi_out = [&](int i) {
        const uint32_t e = tid + 256u * i;
        const float s = (red[e] + red[RED_FLOATS + e]) + (red[2 * RED_FLOATS + e] + red[3 * RED_FLOATS + e]);
        const uint32_t t = (uint32_t)i % NQ16, rt = (uint32_t)i / NQ16;
        const uint32_t q = 16u * t + li;
        const uint64_t row = prev_row0 + 16u * rt + 4u * kk + w;
        if (have_prev && row < row_end && s == s && oi_f32_key(s) >= tau[t]) {
            const uint32_t pos = atomicAdd(&seg_fill[q], 1u); // LDS
            if (pos < seg_cap) my_seg[(uint64_t)q * pool_stride + pos] = oi_rank_key(s, doc_id_base + (uint32_t)row);
            else *overflow = 1u;
        }
    };
    constexpr int N_OUT = NQT * 4;                              // outputs per thread and tile
    constexpr int OPG = (N_OUT + (NG - 4) - 1) / (NG - 4);      // outputs hosted per group, groups 2..
    static_assert(NG >= 6, "tile too short to host the deferred epilogue");

    for (uint64_t ti = 0; ti < my_nt; ++ti) {
        have_prev = ti > 0;
        f32x4v acc[2][NQ16];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < NQ16; ++t) acc[rt][t] = f32x4v{0.f, 0.f, 0.f, 0.f};

        ks_wait<4 * (P - 1)>();
        f32x4 a0 = *reinterpret_cast<const f32x4 *>(ring_rd + frag_off[0][0]);
        f32x4 a1 = *reinterpret_cast<const f32x4 *>(ring_rd + frag_off[1][0]);
        ks_static_for<0, NG>([&](auto gi_) {
            constexpr int gi = decltype(gi_)::value;
            constexpr int kc = gi / 2, g = gi % 2;
            constexpr int sn = kc + P;
            f32x4 n0 = a0, n1 = a1;
            if constexpr (g == 0) {
                n0 = *reinterpret_cast<const f32x4 *>(ring_rd + (kc % NBUF) * KS_SLOT_BYTES + frag_off[0][1]);
                n1 = *reinterpret_cast<const f32x4 *>(ring_rd + (kc % NBUF) * KS_SLOT_BYTES + frag_off[1][1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int t = 0; t < NQ16; ++t) {
                    acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], qreg[t][gi * 4 + e], acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], qreg[t][gi * 4 + e], acc[1][t], 0, 0, 0);
                }
                if (e == 0 || e == 2) { // two DMA pieces per group
                    constexpr int dummy = 0; (void)dummy;
                    const int m = 2 * g + (e >> 1);
                    if constexpr (sn < NKC)
                        ks_issue_piece(cur, voff[m], sn * KS_CHUNK_K * 4, ring_w + (sn % NBUF) * KS_SLOT_BYTES + m * 1024, false);
                    else
                        ks_issue_piece(nxt, voff[m], (sn - NKC) * KS_CHUNK_K * 4,
                                       ring_w + (sn % NBUF) * KS_SLOT_BYTES + m * 1024, false);
                }
            }
            if constexpr (gi == 1) ks_barrier();                                          // (A) partials visible
            if constexpr (gi >= 2 && gi < NG - 1) {
#pragma unroll
                for (int o = 0; o < OPG; ++o)
                    if ((gi - 2) * OPG + o < N_OUT) epi_out((gi - 2) * OPG + o);
            }
            if constexpr (gi == NG - 1) ks_barrier();                                     // (B) `red` is free again
            if constexpr (g == 1 && kc + 1 < NKC) {
                ks_wait<4 * (P - 1)>();
                n0 = *reinterpret_cast<const f32x4 *>(ring_rd + ((kc + 1) % NBUF) * KS_SLOT_BYTES + frag_off[0][0]);
                n1 = *reinterpret_cast<const f32x4 *>(ring_rd + ((kc + 1) % NBUF) * KS_SLOT_BYTES + frag_off[1][0]);
            }
            a0 = n0; a1 = n1;
        });

#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < NQ16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) my_red[((rt * NQ16 + t) * 4 + r) * 64 + lane] = acc[rt][t][r];
        prev_row0 = row_begin + (blockIdx.x + ti * gridDim.x) * (uint64_t)KS_TILE_ROWS;
        cur = nxt;
        nxt = tile_srd(ti + 2);
    }
    have_prev = my_nt > 0;
    ks_barrier();
#pragma unroll
    for (int i = 0; i < N_OUT; ++i) epi_out(i);
    ks_barrier();
    if (tid < 32 * NQT && tid < n_queries) {
        const uint32_t c = seg_fill[tid];
        seg_cnt[(uint64_t)tid * seg_cnt_stride + blockIdx.x] = c < seg_cap ? c : seg_cap;
    }
}

template <int D, int NQT>
static int launch_ksplit16(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, const float *q,
                           uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
    constexpr int KS = D / 4, NKC = KS / KS_CHUNK_K, NBUF = NKC <= 6 ? NKC : NKC / 2;
    constexpr size_t smem = 4 * NBUF * KS_SLOT_BYTES + 4 * (NQT * 16 * 64) * 4 + 64 * 4;
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(cosine_ksplit16_filter<D, NQT>), (size_t)(smem)));
    hipLaunchKernelGGL((cosine_ksplit16_filter<D, NQT>), dim3(p.n_segs), dim3(256), smem, ctx->stream, rows,
                       row_begin, row_end, q, nq, doc_id_base, p.keys, p.seg_cnt, p.seg_cnt_stride, p.tau_keys,
                       p.stride, p.carry_cap, p.seg_cap, p.overflow, ctx->run_gate);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

template <int D, int NQT, int DBG>
static int launch_ksplit_dbg(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, const float *q,
                             uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
    constexpr int KS = D / 4, NKC = KS / KS_CHUNK_K, NBUF = NKC <= 6 ? NKC : NKC / 2;
    constexpr size_t smem = 4 * NBUF * KS_SLOT_BYTES + 4 * (NQT * 16 * 64) * 4 + 64 * 4;
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(cosine_ksplit_filter<D, NQT, DBG>), (size_t)(smem)));
    hipLaunchKernelGGL((cosine_ksplit_filter<D, NQT, DBG>), dim3(p.n_segs), dim3(256), smem, ctx->stream, rows,
                       row_begin, row_end, q, nq, doc_id_base, p.keys, p.seg_cnt, p.seg_cnt_stride, p.tau_keys,
                       p.stride, p.carry_cap, p.seg_cap, p.overflow);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

template <int D, int NQT>
static int launch_ksplit(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, const float *q,
                         uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
#ifdef OI_ABLATION
    if constexpr (D == 768 && NQT == 2) {
        // ablation builds for tools/ks_ablate.py (timings only; results are wrong by construction)
        static const int dbg = oi_ablation_env("OI_KS_DEBUG") ? atoi(oi_ablation_env("OI_KS_DEBUG")) : 0;
        switch (dbg) {
            case 1: return launch_ksplit_dbg<D, NQT, 1>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 2: return launch_ksplit_dbg<D, NQT, 2>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 4: return launch_ksplit_dbg<D, NQT, 4>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 5: return launch_ksplit_dbg<D, NQT, 5>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 6: return launch_ksplit_dbg<D, NQT, 6>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 8: return launch_ksplit_dbg<D, NQT, 8>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 12: return launch_ksplit_dbg<D, NQT, 12>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 14: return launch_ksplit_dbg<D, NQT, 14>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            default: break;
        }
    }
#endif
    return launch_ksplit_dbg<D, NQT, 0>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
}

bool oi_cosine_ksplit_supported(uint32_t dim) { return dim == 384 || dim == 768 || dim == 1024; }

// Pool geometry for one chunk: one segment per workgroup, sized for the worst case (every score of
// every tile that workgroup owns passes the filter).
void oi_cosine_ksplit_geometry(const oi_ctx *ctx, uint64_t n_rows, uint32_t *n_segs, uint32_t *seg_cap) {
    const uint64_t n_tiles = (n_rows + KS_TILE_ROWS - 1) / KS_TILE_ROWS;
    const uint64_t grid = n_tiles < (uint64_t)ctx->num_cus ? n_tiles : (uint64_t)ctx->num_cus;
    *n_segs = (uint32_t)grid;
    *seg_cap = (uint32_t)((n_tiles + grid - 1) / grid) * KS_TILE_ROWS;
}

// One group of <= 64 queries (zero padded to 32 or 64 rows at `q`).
int oi_launch_cosine_ksplit(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, uint32_t dim,
                            const float *q, uint32_t nq, bool two_tiles, uint32_t doc_id_base, const PoolView &p) {
    // 16x16x4 is the default: 3 % faster by wall time than 32x32x2 on the same tile (A/B in one
    // session, 10M x 768, B=64: 9.07 vs 9.35 ms); OI_KS_SHAPE=32 selects the other build.
    static const bool shape16 = !(oi_ablation_env("OI_KS_SHAPE") && atoi(oi_ablation_env("OI_KS_SHAPE")) == 32);
    OI_REQUIRE(shape16 || !ctx->run_gate, "cosine_ksplit: only the 16x16x4 build takes a run gate");
#define OI_KS(DD)                                                                                       \
    case DD:                                                                                            \
        if (shape16)                                                                                    \
            return two_tiles ? launch_ksplit16<DD, 2>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p) \
                             : launch_ksplit16<DD, 1>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p); \
        return two_tiles ? launch_ksplit<DD, 2>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p)   \
                         : launch_ksplit<DD, 1>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
    switch (dim) {
        OI_KS(384)
        OI_KS(768)
        OI_KS(1024)
        default:
            oi_set_error("cosine_ksplit: dim %u not instantiated", dim);
            return OI_ERR_UNSUPPORTED;
    }
#undef OI_KS
}
