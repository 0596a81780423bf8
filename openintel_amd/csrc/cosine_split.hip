// cosine_split.hip -- the batch cosine scorer over an f32 corpus with SPLIT-PRECISION products
// (opt-in: oi_set_cosine_mode(ctx, OI_COSINE_SPLIT); the default stays the exact-f32 MFMA kernel).
//
// Every f32 value is the exact sum of three bf16 values, x = h + m + l (8 significant bits each: h is
// x's upper half, m the upper half of the exact remainder x - h, l what is left).  A product of two
// bf16 values is exact in f32, so
//     x * q = (h + m + l)(H + M + L) = hH + (hM + mH) + (mM + hL + lH) + [mL + lM + lL]
// and dropping the bracket (<= 2^-23 relative, below f32's own rounding) gives an f32-grade dot product
// from SIX bf16 MFMAs.  On gfx950 v_mfma_f32_32x32x16_bf16 runs 16x the flop rate of the f32 MFMA, so
// the six cost 3/8 of the f32 instruction's cycles: the scorer stops being matrix-pipe-bound (6.25 ms
// of f32 MFMA per 10M x 768 x 64 batch) and becomes HBM-bound (30.7 GB, ~5.5 ms).  Measured error vs
// f64 on unit vectors: ~4e-8 (plain f32 accumulation: ~1e-7); the parity bar is 1e-5.
//
// Shape: one workgroup per CU, 4 waves splitting K (wave w owns k in [w D/4, (w+1) D/4)):
//   * the wave's K-slice of all 64 queries, already split into (H, M, L) bf16 planes by the staging
//     kernel, lives in registers as MFMA B operands (3 x 96 VGPRs at d = 768);
//   * the f32 rows stream through the wave's own LDS ring exactly as in cosine_ksplit.hip (4 KiB slots of
//     32 rows x 32 floats, buffer_load ... lds, counted vmcnt waits, prefetch across tiles);
//   * per 16 k: two ds_read_b128 (the lane's 8 consecutive floats), ~44 VALU to split and pack them into
//     the three A operands -- issued in the shadow of the 12 MFMAs (32 cycles each) they feed;
//   * the big term hH accumulates apart from the five small ones (added at the end), then the four waves'
//     partial tiles are summed through LDS in a fixed order, filtered against the per-query threshold and
//     appended to this workgroup's private pool segment.
#include <cstdlib>
#include <type_traits>

#include "oi_device.h"
#include "oi_internal.h"

typedef float cs_f32x16 __attribute__((ext_vector_type(16)));
typedef float cs_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cs_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t cs_u32x4 __attribute__((ext_vector_type(4)));

#define CS_TILE_ROWS 32
#define CS_SLOT_BYTES 4096 // 32 rows x 128 B (32 floats of K)

__device__ __forceinline__ uint32_t cs_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ cs_u32x4 cs_make_srd(const float *base, uint64_t bytes) {
    const uint64_t b = (uint64_t)base;
    cs_u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xFFFFu);
    r[2] = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes));
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void cs_issue_piece(const cs_u32x4 &srd, uint32_t voff, uint32_t soff, uint32_t lds_dst,
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
__device__ __forceinline__ void cs_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cs_static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void cs_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void cs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 8 consecutive floats of a row -> the three bf16x8 A operands (h, m, l), x = h + m + l exactly.
union CsPack {
    uint32_t u[4];
    cs_bf16x8 v;
};
__device__ __forceinline__ void cs_split8(const cs_f32x4 &a, const cs_f32x4 &b, cs_bf16x8 &h, cs_bf16x8 &m, cs_bf16x8 &l) {
    float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    float r1[8], r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r1[i] = x[i] - __uint_as_float(__float_as_uint(x[i]) & 0xFFFF0000u);   // exact
        r2[i] = r1[i] - __uint_as_float(__float_as_uint(r1[i]) & 0xFFFF0000u); // exact, <= 8 significant bits
    }
    CsPack ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) { // upper halves of two floats -> one dword: bytes [hi.3, hi.2, lo.3, lo.2]
        ph.u[j] = __builtin_amdgcn_perm(__float_as_uint(x[2 * j + 1]), __float_as_uint(x[2 * j]), 0x07060302u);
        pm.u[j] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * j + 1]), __float_as_uint(r1[2 * j]), 0x07060302u);
        pl.u[j] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * j + 1]), __float_as_uint(r2[2 * j]), 0x07060302u);
    }
    h = ph.v; m = pm.v; l = pl.v;
}

template <int D, int NQT, int DBG = 0> // DBG (diagnostic builds, wrong results): 1 no DMA, 2 no MFMA, 4 no split arithmetic
__global__ __launch_bounds__(256, 1) void cosine_split_filter(
    const float *__restrict__ rows, uint64_t row_begin, uint64_t row_end,
    const uint16_t *__restrict__ qsplit, // bf16 [3 planes: H, M, L][32*NQT][D], zero padded
    uint32_t n_queries, uint32_t doc_id_base, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride,
    const uint32_t *tau_keys, uint64_t pool_stride, uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow) {
    OI_CLAIM_WHOLE_SIMD(); // (MFMA kernel: nothing else may run on this CU -- oi_device.h)
    constexpr int KS = D / 4;            // K-slice of one wave (floats)
    constexpr int NKC = KS / 32;         // ring slots per tile and wave
    constexpr int NBUF = NKC <= 6 ? NKC : NKC / 2;
    constexpr int P = NBUF - 1;
    constexpr int KSTEPS = KS / 16;      // MFMA groups per tile and wave (two per slot)
    constexpr int RED = NQT * 16 * 64;   // floats of one wave's partial tile
    static_assert(KS % 32 == 0 && NKC % NBUF == 0 && P >= 1 && P < NKC, "unsupported D");
    static_assert(3 * NQT * KSTEPS * 4 <= 300, "the split query block must fit the register file");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *ring = smem;                                                   // [4][NBUF][4 KiB]
    float *red = reinterpret_cast<float *>(smem + 4 * NBUF * CS_SLOT_BYTES);      // [4][RED]
    uint32_t *seg_fill = reinterpret_cast<uint32_t *>(red + 4 * RED);             // [32*NQT]

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t li = lane & 31, lh = lane >> 5;

    // ---- this wave's K-slice of every query, three planes: B[k = 16 s + 8 lh + 0..7][n = li]
    cs_bf16x8 qh[NQT][KSTEPS], qm[NQT][KSTEPS], ql[NQT][KSTEPS];
    constexpr uint64_t PLANE = (uint64_t)32 * NQT * D;
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const uint64_t o = (uint64_t)(32 * t + li) * D + w * KS + 16 * s + 8 * lh;
            qh[t][s] = *reinterpret_cast<const cs_bf16x8 *>(qsplit + o);
            qm[t][s] = *reinterpret_cast<const cs_bf16x8 *>(qsplit + PLANE + o);
            ql[t][s] = *reinterpret_cast<const cs_bf16x8 *>(qsplit + 2 * PLANE + o);
        }
    uint32_t tau[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const uint32_t q = 32u * t + li;
        tau[t] = q < n_queries ? tau_keys[q] : 0xFFFFFFFFu;
    }
    if (tid < 32 * NQT) seg_fill[tid] = 0;

    // ---- tiles of this workgroup: blockIdx.x, + gridDim.x, ...
    const uint64_t n_rows = row_end - row_begin;
    const uint64_t n_tiles = (n_rows + CS_TILE_ROWS - 1) / CS_TILE_ROWS;
    const uint64_t my_nt = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_nt == 0) return;
    uint64_t *my_seg = pools + carry_cap + (uint64_t)blockIdx.x * seg_cap;

    uint32_t voff[4]; // per-lane source of the 4 DMA pieces of a slot (cosine_ksplit.hip: same swizzle)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const uint32_t prow = 8 * m + (lane >> 3);
        voff[m] = prow * (uint32_t)(D * 4) + w * (uint32_t)(KS * 4) + (((lane & 7) ^ ((prow >> 1) & 7)) << 4);
    }
    const uint32_t ring_w = cs_lds_addr(ring) + w * (NBUF * CS_SLOT_BYTES);
    const unsigned char *ring_rd = ring + w * (NBUF * CS_SLOT_BYTES);
    // fragment (slot half g in {0,1}): the lane's floats 16 g + 8 lh .. + 8 = logical 16-B columns 4g + 2lh, + 1
    uint32_t frag_off[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 2; ++c) frag_off[g][c] = li * 128 + (((4 * g + 2 * lh + c) ^ ((li >> 1) & 7)) << 4);

    auto tile_row0 = [&](uint64_t ti) { return row_begin + (blockIdx.x + ti * gridDim.x) * (uint64_t)CS_TILE_ROWS; };
    // Past the workgroup's last tile the descriptor is empty: its loads return zeros, so the tile loop needs no
    // branch around the prefetch (and no second set of wait counts) -- a branch-free body is also what lets the
    // scheduler interleave the split arithmetic with the MFMAs.
    auto tile_srd = [&](uint64_t ti) {
        const uint64_t r0 = tile_row0(ti < my_nt ? ti : 0);
        return cs_make_srd(rows + r0 * D, ti < my_nt ? (row_end - r0) * (uint64_t)(D * 4) : 0ull);
    };
    cs_u32x4 cur = tile_srd(0), nxt = tile_srd(1);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only: retire every load hipcc knows about, here
#pragma unroll
    for (int kc = 0; kc < P; ++kc)
#pragma unroll
        for (int m = 0; m < 4; ++m)
            cs_issue_piece(cur, voff[m], kc * 128, ring_w + (kc % NBUF) * CS_SLOT_BYTES + m * 1024, (DBG & 1) != 0);
    float *my_red = red + w * RED;

    // ---- deferred epilogue (as in cosine_ksplit.hip).  A tile's partial sums go to LDS right after its last
    // MFMA; the cross-wave sum + filter of tile t then rides INSIDE tile t+1's MFMA stream (barrier A before
    // k-step EPI_G0, one output per thread per k-step, barrier B behind the last one), so the DMA ring keeps
    // being refilled while the epilogue's LDS round trips and stores go out.
    uint64_t prev_row0 = 0; // first corpus row of the tile whose partials sit in `red`
    bool have_prev = false;  // false during the first tile: `red` holds nothing yet
    auto epi_out = [&](int i) {
        const uint32_t e = tid + 256u * i; // (t = i>>2, r = (e>>6)&15, lane)
        const float s = (red[e] + red[RED + e]) + (red[2 * RED + e] + red[3 * RED + e]);
        const uint32_t t = i >> 2, r = (e >> 6) & 15u;
        const uint32_t q = 32u * t + li;
        const uint64_t row = prev_row0 + (r & 3u) + 8u * (r >> 2) + 4u * lh;
        if (have_prev && row < row_end && s == s && oi_f32_key(s) >= tau[t]) {
            const uint32_t pos = atomicAdd(&seg_fill[q], 1u); // LDS
            if (pos < seg_cap) my_seg[(uint64_t)q * pool_stride + pos] = oi_rank_key(s, doc_id_base + (uint32_t)row);
            else *overflow = 1u;
        }
    };
    constexpr int NG = NKC * 2;                              // k-steps per tile
    constexpr int EPI_G0 = 1;                                // first k-step with outputs
    constexpr int EPI_PER = (NQT * 4 + (NG - 2) - 1) / (NG - 2); // outputs per thread and k-step (2 at d = 384)
    constexpr int EPI_STEPS = (NQT * 4 + EPI_PER - 1) / EPI_PER;
    constexpr int EPI_GB = EPI_G0 + EPI_STEPS;               // k-step of barrier B
    static_assert(EPI_GB <= NG - 1, "tile too short to host the deferred epilogue");

    for (uint64_t ti = 0; ti < my_nt; ++ti) {
        have_prev = ti > 0;
        // hH accumulates apart from the five small terms; with one query tile the small terms use two
        // accumulators so that no MFMA waits for the one issued right before it
        cs_f32x16 acc[NQT], cor[NQT], cor2[NQT == 1 ? 1 : 1];
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; cor[t][r] = 0.f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) cor2[0][r] = 0.f;

        cs_wait<4 * (P - 1)>();
        cs_bf16x8 ah, am, al; // A operands of the current k-step
        {
            const cs_f32x4 f0 = *reinterpret_cast<const cs_f32x4 *>(ring_rd + frag_off[0][0]);
            const cs_f32x4 f1 = *reinterpret_cast<const cs_f32x4 *>(ring_rd + frag_off[0][1]);
            if constexpr ((DBG & 4) != 0) { CsPack pk; pk.u[0] = __float_as_uint(f0[0]); pk.u[1] = __float_as_uint(f0[1]); pk.u[2] = __float_as_uint(f1[0]); pk.u[3] = __float_as_uint(f1[1]); ah = am = al = pk.v; }
            else cs_split8(f0, f1, ah, am, al);
        }
        cs_static_for<0, NG>([&](auto gi_) {
            constexpr int gi = decltype(gi_)::value; // k-step of the tile
            constexpr int kc = gi / 2, g = gi % 2;
            constexpr int sn = kc + P;               // slot refilled during this slot's k-steps
            // The NEXT k-step's floats are read now and split into its three operands WHILE this k-step's MFMAs
            // run: one wave per SIMD issues in order, so the split has to sit between the MFMAs in program order
            // (sched_group_barrier below) to execute in their shadow.
            cs_bf16x8 nh = ah, nm = am, nl = al;
            if constexpr (gi + 1 < NG) {
                constexpr int nslot = g == 0 ? kc : kc + 1, nhalf = g == 0 ? 1 : 0;
                const cs_f32x4 f0 = *reinterpret_cast<const cs_f32x4 *>(ring_rd + (nslot % NBUF) * CS_SLOT_BYTES + frag_off[nhalf][0]);
                const cs_f32x4 f1 = *reinterpret_cast<const cs_f32x4 *>(ring_rd + (nslot % NBUF) * CS_SLOT_BYTES + frag_off[nhalf][1]);
                if constexpr ((DBG & 4) != 0) { CsPack pk; pk.u[0] = __float_as_uint(f0[0]); pk.u[1] = __float_as_uint(f0[1]); pk.u[2] = __float_as_uint(f1[0]); pk.u[3] = __float_as_uint(f1[1]); nh = nm = nl = pk.v; }
                else cs_split8(f0, f1, nh, nm, nl);
            }
            // twelve (six) MFMAs, ordered so that consecutive ones never share an accumulator; the small terms
            // are added smallest first
            if constexpr ((DBG & 2) != 0) {
                asm volatile("" ::"v"(ah), "v"(am), "v"(al));
            } else if constexpr (NQT == 2) {
                cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[0][gi], cor[0], 0, 0, 0);
                cor[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[1][gi], cor[1], 0, 0, 0);
                cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[0][gi], cor[0], 0, 0, 0);
                cor[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[1][gi], cor[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[0][gi], acc[0], 0, 0, 0);
                cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qm[0][gi], cor[0], 0, 0, 0);
                cor[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qm[1][gi], cor[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[1][gi], acc[1], 0, 0, 0);
                cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qh[0][gi], cor[0], 0, 0, 0);
                cor[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qh[1][gi], cor[1], 0, 0, 0);
                cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qm[0][gi], cor[0], 0, 0, 0);
                cor[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qm[1][gi], cor[1], 0, 0, 0);
            } else {
                cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[0][gi], cor[0], 0, 0, 0);
                cor2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[0][gi], cor2[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[0][gi], acc[0], 0, 0, 0);
                cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qm[0][gi], cor[0], 0, 0, 0);
                cor2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qh[0][gi], cor2[0], 0, 0, 0);
                cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qm[0][gi], cor[0], 0, 0, 0);
            }
            if constexpr ((DBG & 6) == 0) {
                // issue order of this k-step: the two LDS reads of the next fragment first, two MFMAs to cover
                // their latency, then a handful of the split's VALU instructions after every further MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                for (int i = 2; i < NQT * 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x002, NQT == 2 ? 6 : 15, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
            // two DMA pieces per k-step: slot kc + P goes into the buffer slot kc - 1 has vacated
#pragma unroll
            for (int m = 2 * g; m < 2 * g + 2; ++m) {
                if constexpr (sn < NKC)
                    cs_issue_piece(cur, voff[m], sn * 128, ring_w + (sn % NBUF) * CS_SLOT_BYTES + m * 1024, (DBG & 1) != 0);
                else
                    cs_issue_piece(nxt, voff[m], (sn - NKC) * 128, ring_w + (sn % NBUF) * CS_SLOT_BYTES + m * 1024,
                                   (DBG & 1) != 0);
            }
            if constexpr (gi == EPI_G0 - 1 && !(DBG & 8)) cs_barrier();                        // (A) partials visible
            if constexpr (gi >= EPI_G0 && gi < EPI_GB && !(DBG & 8)) {
#pragma unroll
                for (int o = 0; o < EPI_PER; ++o)
                    if ((gi - EPI_G0) * EPI_PER + o < NQT * 4) epi_out((gi - EPI_G0) * EPI_PER + o);
            }
            if constexpr (gi == EPI_GB && !(DBG & 8)) cs_barrier();                            // (B) `red` is free again
            // the next k-step reads slot kc + 1 when this one is the slot's first half: it has to have landed
            // (the two pieces of slot kc + P issued just above may still be in flight with P - 2 younger slots)
            if constexpr (g == 0 && kc + 1 < NKC) cs_wait<4 * (P - 2) + 2>();
            ah = nh; am = nm; al = nl;
        });

        // this tile's partial sums -> LDS (summed across the waves during the next tile, or below)
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                my_red[(t * 16 + r) * 64 + lane] = acc[t][r] + (NQT == 1 ? cor[t][r] + cor2[0][r] : cor[t][r]);
        prev_row0 = tile_row0(ti);
        cur = nxt;
        nxt = tile_srd(ti + 2);
    }
    have_prev = true;
    cs_barrier(); // the last tile's epilogue has no MFMA stream to hide in
#pragma unroll
    for (int i = 0; i < NQT * 4; ++i) epi_out(i);
    cs_barrier();
    if (tid < 32 * NQT && tid < n_queries) {
        const uint32_t c = seg_fill[tid];
        seg_cnt[(uint64_t)tid * seg_cnt_stride + blockIdx.x] = c < seg_cap ? c : seg_cap;
    }
}

// ------------------------------------------------------------------ query staging: f32 -> (H, M, L) bf16 planes
__global__ __launch_bounds__(256) void cs_stage_queries(const float *__restrict__ q, uint32_t n_queries, uint32_t n_padded,
                                                        uint32_t dim, uint16_t *__restrict__ out) {
    const uint64_t plane = (uint64_t)n_padded * dim;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / dim);
        uint16_t h = 0, m = 0, l = 0;
        if (row < n_queries) {
            const float x = q[i];
            const uint32_t ux = __float_as_uint(x);
            const float r1 = x - __uint_as_float(ux & 0xFFFF0000u);
            const uint32_t u1 = __float_as_uint(r1);
            const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
            h = (uint16_t)(ux >> 16);
            m = (uint16_t)(u1 >> 16);
            l = (uint16_t)(__float_as_uint(r2) >> 16);
        }
        out[i] = h;
        out[plane + i] = m;
        out[2 * plane + i] = l;
    }
}

// ------------------------------------------------------------------ host
bool oi_cosine_split_supported(uint32_t dim) { return dim == 384 || dim == 768; }

template <int D, int NQT, int DBG = 0>
static int launch_split(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, const uint16_t *q,
                        uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
    constexpr int KS = D / 4, NKC = KS / 32, NBUF = NKC <= 6 ? NKC : NKC / 2;
    constexpr size_t smem = 4 * NBUF * CS_SLOT_BYTES + 4 * (NQT * 16 * 64) * 4 + 64 * 4;
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(cosine_split_filter<D, NQT, DBG>), (size_t)(smem)));
    hipLaunchKernelGGL((cosine_split_filter<D, NQT, DBG>), dim3(p.n_segs), dim3(256), smem, ctx->stream, rows, row_begin,
                       row_end, q, nq, doc_id_base, p.keys, p.seg_cnt, p.seg_cnt_stride, p.tau_keys, p.stride,
                       p.carry_cap, p.seg_cap, p.overflow);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

// One group of <= 64 queries (f32, device) over rows [row_begin, row_end); same pool geometry as the K-split
// kernel (one segment per workgroup, oi_cosine_ksplit_geometry).
int oi_launch_cosine_split(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, uint32_t dim,
                           const float *d_queries, uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
    OI_REQUIRE(oi_cosine_split_supported(dim), "cosine (split products): dim %u not instantiated (384, 768)", dim);
    const uint32_t n_padded = nq > 32 ? 64u : 32u;
    DevBuf &qb = ctx->buf("q_split");
    OI_CHECK(qb.ensure(sizeof(uint16_t) * 3ull * 64 * OI_MAX_DIM));
    {
        const uint64_t total = (uint64_t)n_padded * dim;
        hipLaunchKernelGGL(cs_stage_queries, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_queries, nq,
                           n_padded, dim, qb.as<uint16_t>());
        OI_HIP_CHECK(hipGetLastError());
    }
    const uint16_t *q = qb.as<uint16_t>();
#ifdef OI_ABLATION
    static const int dbg = oi_ablation_env("OI_CS_DEBUG") ? atoi(oi_ablation_env("OI_CS_DEBUG")) : 0; // ablation builds (timings only)
    if (dim == 768 && nq > 32 && dbg) {
        switch (dbg) {
            case 1: return launch_split<768, 2, 1>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 2: return launch_split<768, 2, 2>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 4: return launch_split<768, 2, 4>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 6: return launch_split<768, 2, 6>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 8: return launch_split<768, 2, 8>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 9: return launch_split<768, 2, 9>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            case 15: return launch_split<768, 2, 15>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
            default: break;
        }
    }
#endif
    if (dim == 768) return nq > 32 ? launch_split<768, 2>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p)
                                   : launch_split<768, 1>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
    return nq > 32 ? launch_split<384, 2>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p)
                   : launch_split<384, 1>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
}
