// cosine_prefilter.hip -- the f32 corpus scored at HBM speed: a bf16 screen with a PROVEN error bound, then
// exact f32 scores for the few rows that pass it.
//
// Builder-defined like the rest of the retrieval path (the reference has none).
//
// Why.  The exact f32 MFMA scorer (cosine_ksplit.hip) is bound by the f32 matrix pipes and, under them, by the
// chip's power limit (8.8 ms per 64-query batch at 10M x 768 = 70 % of the f32 MFMA peak; the same corpus
// streams from HBM in 5 ms).  A top-k' list does not need the exact score of every row, only of the rows that
// can reach the list.  So:
//   1. screen: s~ = sum_k bf16(x_k) * bf16(q_k), accumulated in f32 by v_mfma_f32_32x32x16_bf16 (1/16 of the
//      f32 MFMA cycles), the f32 rows converted on the fly between LDS and the matrix pipe (v_cvt_pk_bf16_f32,
//      round to nearest even).  The corpus stays f32 in HBM and is read once: the kernel is a stream.
//   2. bound (round 2: re-derived -- the round-1 constant assumed a unit roundoff of 2^-9 for bf16; it is 2^-8,
//      and a product of two rounded values can be off by 2^-7, so that bound was too small by 2x and a row of
//      the exact list could be screened out).  Nothing is assumed about the rounding now; it is MEASURED.  With
//      x~ = bf16(x), q~ = bf16(q), e_x = x~ - x, e_q = q~ - q (both exact in f32):
//          s~ - s  =  sum_k x~_k q~_k - x_k q_k  =  e_x . q~  +  x . e_q
//      so by Cauchy-Schwarz   |s~ - s| <= |e_x| |q~| + |x| |e_q|   for the exact sums, and the f32 accumulation
//      of the screen (products of bf16 values are exact in f32; <= d additions, each off by at most 2^-23 of
//      the running sum even if the matrix pipe truncates) and of the rescoring add <= d 2^-22 |x~| |q~|.  Hence
//          eps_q = E |q~| + X |e_q| + d 2^-22 (X + E) |q~|  (+ an absolute term for flushed denormals)
//      with X = max_r |x_r| and E = max_r |bf16(x_r) - x_r| taken over the corpus when the rows are set (one
//      pass, the same v_cvt_pk_bf16_f32 the screen uses) and |q~|, |e_q| per query at search time.  For
//      unit Gaussian rows at d = 768 that is E ~ 0.5 * 2^-8, |e_q| ~ 0.45 * 2^-8: eps ~ 0.0042, about half
//      of the worst case 2^-7 -- and when every coordinate sits at a bf16 tie (the adversarial case,
//      tests/test_gpu_prefilter.py) E and |e_q| grow to 2^-8 |x| and the margin grows with them: the
//      bound holds for ANY data, it is only tight for typical data.
//   3. keep: if tau~ is the k'-th largest s~ seen so far, at least k' rows have s >= tau~ - eps, so the final
//      k'-th exact score is >= tau~ - eps, and any row of the final list has s~ >= tau~ - 2 eps.  The select
//      after each corpus chunk therefore keeps EVERY key within 2 eps of the k'-th (select.hip, margin mode) and
//      the screen's threshold is tau~ - 2 eps: a superset of the exact list survives, about 2.4 k' keys for
//      unit vectors at d = 768.
//   4. rescore: exact f32 dot products of the survivors only (one wave per (query, row); ~150K rows per batch
//      instead of 640M), then the ordinary sorted top-k' selection over exact keys.
//   5. if a query's survivors do not fit (4096 keys: rows within 2 eps of the threshold -- duplicates, or norms
//      far above the typical row's) or its norm is not finite, a device flag opens the GATED exact launches that
//      follow in the same stream and overwrite the lists: the exact kernel over all rows with the screen's
//      thresholds (still valid lower bounds of the exact k'-th scores), then one selection; with the gate shut
//      both exit at once.  A corpus whose largest norm is not finite is never screened (decided on the host
//      when the rows are set).
// The lists that come out are those of the exact scorer (up to the f32 rounding of two different summation
// orders, which the parity tolerance already covers); tests/test_gpu_prefilter.py checks both regimes.
//
// Kernel shape: cosine_bf16.hip's solo kernel fed with f32 rows.  One workgroup per CU, 4 waves, no K-split:
// a wave holds all 64 queries over the whole K as bf16 B operands (384 VGPRs at d = 768), owns whole 32-row
// tiles and streams them through its own LDS ring of 4 KiB slots (32 rows x 32 floats) with
// buffer_load ... lds, 7 slots ahead, ordered by counted s_waitcnt vmcnt; waves never meet.  Per 16 k of a tile:
// two conflict-free ds_read_b128, four v_cvt_pk_bf16_f32, two MFMAs.  HBM-bound: 4 d bytes per row and batch.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "oi_device.h"
#include "oi_internal.h"

typedef float pf_f32x16 __attribute__((ext_vector_type(16)));
typedef float pf_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pf_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t pf_u32x4 __attribute__((ext_vector_type(4)));

#define PF_TILE_ROWS 32
#define PF_SLOT_K 32                 // floats of K per ring slot row (128 B)
#define PF_SLOT_BYTES (PF_TILE_ROWS * 128)

__device__ __forceinline__ uint32_t pf_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ pf_u32x4 pf_make_srd(const float *base, uint64_t bytes) {
    const uint64_t b = (uint64_t)base;
    pf_u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xFFFFu); // stride 0
    r[2] = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes));
    r[3] = 0x00020000u;
    return r;
}
// One 1-KiB LDS-DMA piece (8 rows x 128 B).  Lanes past the descriptor's end read as zero: the ragged last
// tile and the tile after the last one (empty descriptor) need no branch.  hipcc does not see these loads:
// they are ordered by pf_wait<N>().
__device__ __forceinline__ void pf_issue_piece(const pf_u32x4 &srd, uint32_t voff, uint32_t soff, uint32_t lds_dst) {
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
__device__ __forceinline__ void pf_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pf_static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void pf_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ pf_bf16x8 pf_pack(const pf_f32x4 &a, const pf_f32x4 &b) {
    pf_bf16x8 r;
    r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3]; // v_cvt_pk_bf16_f32 (RNE)
    r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
    return r;
}

// Survivor staging of the screen kernel: PF_STAGE entries per wave (a power of two), flushed PF_STAGE_FLUSH at a time (fewer than
// PF_STAGE_FLUSH stay between tiles, so a tile of up to PF_STAGE - PF_STAGE_FLUSH survivors is staged).
#define PF_STAGE 256
#define PF_STAGE_FLUSH 64u
#define PF_STAGE_LDS (4 * PF_STAGE * 12)
__device__ __forceinline__ uint32_t pf_incl_scan(uint32_t v) { // wave-wide inclusive prefix sum (DPP, no LDS)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return v;
}
// The first NF staged entries of the wave leave for the pool: lane l takes entry st_head + l, its position in its query's
// segment from ONE LDS atomic, one store instruction for all of them.  (A macro: a lambda would take st_head / st_n by
// reference and hipcc then keeps them in scratch.)  The compiler barriers keep the staging writes of other lanes in front of
// these reads, and these reads in front of the next tile's writes (LDS operations of a wave execute in order).
#define PF_FLUSH(NF)                                                                                                   \
    do {                                                                                                               \
        const uint32_t nf_ = (NF);                                                                                     \
        asm volatile("" ::: "memory");                                                                                 \
        if (lane < nf_) {                                                                                              \
            const uint32_t i_ = (st_head + lane) & (PF_STAGE - 1);                                                     \
            const uint64_t key_ = stage_keys[i_];                                                                      \
            const uint32_t q_ = stage_q[i_];                                                                           \
            const uint32_t pos_ = atomicAdd(&seg_fill[q_], 1u);                                                        \
            if (pos_ < seg_cap) my_seg[(uint64_t)q_ * pool_stride + pos_] = key_;                                      \
            else *overflow = 1u;                                                                                       \
        }                                                                                                              \
        asm volatile("" ::: "memory");                                                                                 \
        st_head = (st_head + nf_) & (PF_STAGE - 1);                                                                    \
        st_n -= nf_;                                                                                                   \
    } while (0)

template <int D, int NQT>
__global__ __launch_bounds__(256, 1) void cosine_screen_filter(
    const float *__restrict__ rows, uint64_t row_begin, uint64_t row_end,
    const uint16_t *__restrict__ queries, // bf16 [32*NQT][D], zero padded
    uint32_t n_queries, uint32_t doc_id_base, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride,
    const uint32_t *tau_keys, uint64_t pool_stride, uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow) {
    constexpr int NKC = D / PF_SLOT_K;                    // ring slots per tile
    constexpr int NBUF = NKC % 8 == 0 ? 8 : (NKC % 6 == 0 ? 6 : NKC);
    constexpr int P = NBUF - 1;                           // slots in flight ahead of the one being consumed
    constexpr int KSTEPS = D / 16;                        // MFMA groups per tile: two per slot
    static_assert(D % PF_SLOT_K == 0 && NKC % NBUF == 0 && P >= 1 && P < NKC, "unsupported D");
    static_assert(NQT * KSTEPS * 4 <= 400, "the query block must fit the register file");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *ring = smem;                                                         // [4][NBUF][4 KiB]
    uint32_t *seg_fill = reinterpret_cast<uint32_t *>(smem + 4 * NBUF * PF_SLOT_BYTES); // [64]

    OI_CLAIM_WHOLE_SIMD(); // (MFMA kernel: nothing else may run on this CU -- oi_device.h)
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t li = lane & 31, lh = lane >> 5;
    // the wave's staging ring of survivors (keys and their queries), behind seg_fill
    uint64_t *stage_keys = reinterpret_cast<uint64_t *>(smem + 4 * NBUF * PF_SLOT_BYTES + 256) + w * PF_STAGE;
    uint32_t *stage_q = reinterpret_cast<uint32_t *>(smem + 4 * NBUF * PF_SLOT_BYTES + 256 + 4 * PF_STAGE * 8) + w * PF_STAGE;
    uint32_t st_head = 0, st_n = 0; // wave-uniform: first staged entry (mod PF_STAGE), staged entries (< PF_STAGE_FLUSH between tiles)

    // ---- every query over the whole K, in registers for the whole launch: B[k = 16 s + 8 lh + 0..7][n = li]
    pf_bf16x8 qreg[NQT][KSTEPS];
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            qreg[t][s] = *reinterpret_cast<const pf_bf16x8 *>(queries + (uint64_t)(32 * t + li) * D + 16 * s + 8 * lh);
    // Screen thresholds (tau~ - 2 eps) of the queries this lane filters, as FLOATS: for a score s that is not a NaN,
    // oi_f32_key(s) >= key  <=>  s >= oi_key_f32(key) (the key is strictly monotone on floats after s + 0 has made -0 a +0, and
    // the comparison does not tell -0 from +0 either); keys at or below key(-inf) pass every such score (-inf), keys above
    // key(+inf) -- 0xFFFFFFFF: no query in this slot -- map to NaN bit patterns, which no score is >=.  A NaN score fails the
    // comparison by itself.  One v_cmp per score instead of the key's five instructions.
    float tauf[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const uint32_t q = 32u * t + li;
        const uint32_t k = q < n_queries ? tau_keys[q] : 0xFFFFFFFFu;
        tauf[t] = k <= 0x007FFFFFu ? -__builtin_inff() : oi_key_f32(k);
    }
    if (tid < 32 * NQT) seg_fill[tid] = 0;
    __syncthreads(); // the only barrier before the end: seg_fill is zero before any wave appends

    // ---- tiles of this WAVE: (blockIdx.x * 4 + w), + 4 * gridDim.x, ...
    const uint64_t n_rows = row_end - row_begin;
    const uint64_t n_tiles = (n_rows + PF_TILE_ROWS - 1) / PF_TILE_ROWS;
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
            voff[m] = prow * (uint32_t)(D * 4) + (((lane & 7) ^ ((prow >> 1) & 7)) << 4);
        }
        const uint32_t ring_w = pf_lds_addr(ring) + w * (NBUF * PF_SLOT_BYTES);
        const unsigned char *ring_rd = ring + w * (NBUF * PF_SLOT_BYTES);
        // fragment of k-step g of a slot: row li, floats 16 g + 8 lh + 0..7 = logical 16-B columns 4g + 2lh, + 1
        uint32_t frag_off[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) frag_off[g][h] = li * 128 + (((4 * g + 2 * lh + h) ^ ((li >> 1) & 7)) << 4);

        auto tile_row0 = [&](uint64_t ti) { return row_begin + (first + ti * stride) * (uint64_t)PF_TILE_ROWS; };
        auto tile_srd = [&](uint64_t ti) { // past this wave's last tile: an EMPTY descriptor (loads return zeros)
            const uint64_t r0 = tile_row0(ti < my_nt ? ti : 0);
            return pf_make_srd(rows + r0 * D, ti < my_nt ? (row_end - r0) * (uint64_t)(D * 4) : 0ull);
        };
        pf_u32x4 cur = tile_srd(0), nxt = tile_srd(1);
        // Every load hipcc knows about (queries, thresholds) is retired HERE, with a wait it models:
        // otherwise it re-waits for them inside the tile loop and drains the DMA ring.
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only
#pragma unroll
        for (int kc = 0; kc < P; ++kc) // prologue: slots 0..P-1 of the first tile
#pragma unroll
            for (int m = 0; m < 4; ++m)
                pf_issue_piece(cur, voff[m], kc * 128, ring_w + (kc % NBUF) * PF_SLOT_BYTES + m * 1024);

        for (uint64_t ti = 0; ti < my_nt; ++ti) {
            pf_f32x16 acc[NQT];
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

            // Slot s of this tile lives in ring buffer s % NBUF.  Per k-step (kc, g): read the next k-step's
            // floats, NQT MFMAs on the current operand, DMA pieces 2g, 2g+1 of slot kc + P into the buffer
            // slot kc - 1 has vacated, convert the floats read; at g == 1 the next k-step is (kc + 1, 0),
            // behind the counted wait that retires slot kc + 1 (P - 1 younger slots stay in flight).
            pf_wait<4 * (P - 1)>();
            pf_bf16x8 a_cur = pf_pack(*reinterpret_cast<const pf_f32x4 *>(ring_rd + frag_off[0][0]),
                                      *reinterpret_cast<const pf_f32x4 *>(ring_rd + frag_off[0][1]));
            pf_static_for<0, NKC * 2>([&](auto gi_) {
                constexpr int gi = decltype(gi_)::value;
                constexpr int kc = gi / 2, g = gi % 2;
                constexpr int sn = kc + P; // slot refilled during this slot's k-steps
                pf_f32x4 f0, f1;
                if constexpr (g == 0) {
                    f0 = *reinterpret_cast<const pf_f32x4 *>(ring_rd + (kc % NBUF) * PF_SLOT_BYTES + frag_off[1][0]);
                    f1 = *reinterpret_cast<const pf_f32x4 *>(ring_rd + (kc % NBUF) * PF_SLOT_BYTES + frag_off[1][1]);
                }
#pragma unroll
                for (int t = 0; t < NQT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur, qreg[t][gi], acc[t], 0, 0, 0);
#pragma unroll
                for (int m = 2 * g; m < 2 * g + 2; ++m) {
                    if constexpr (sn < NKC)
                        pf_issue_piece(cur, voff[m], sn * 128, ring_w + (sn % NBUF) * PF_SLOT_BYTES + m * 1024);
                    else
                        pf_issue_piece(nxt, voff[m], (sn - NKC) * 128, ring_w + (sn % NBUF) * PF_SLOT_BYTES + m * 1024);
                }
                if constexpr (g == 1 && kc + 1 < NKC) {
                    pf_wait<4 * (P - 1)>();
                    f0 = *reinterpret_cast<const pf_f32x4 *>(ring_rd + ((kc + 1) % NBUF) * PF_SLOT_BYTES + frag_off[0][0]);
                    f1 = *reinterpret_cast<const pf_f32x4 *>(ring_rd + ((kc + 1) % NBUF) * PF_SLOT_BYTES + frag_off[0][1]);
                }
                if constexpr (gi + 1 < NKC * 2) a_cur = pf_pack(f0, f1);
            });

            // ---- filter + append, straight out of the accumulators: register r of query tile t holds
            // D[row (r&3) + 8 (r>>2) + 4 lh][query 32 t + li]
            // Round 4: which of the lane's 16 NQT scores pass is collected in a mask first (one compare each); a tile without
            // a survivor -- most tiles of the large chunks -- leaves through one ballot, and a lane with survivors takes ONE
            // LDS atomic per query for all of them.  (Round 3 took an atomic and waited for it per score: 200-350 wave
            // cycles per survivor, 0.13 ms of the 4.9 ms step at 10M rows -- tools/r04_epilogue_probe.sh.)
            const uint64_t row0 = tile_row0(ti);
            uint32_t m = 0;
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) m |= acc[t][r] >= tauf[t] ? 1u << (16 * t + r) : 0u;
            if (row_end - row0 < (uint64_t)PF_TILE_ROWS) { // the ragged last tile: rows past the end read as zeros
                const uint32_t left = (uint32_t)(row_end - row0);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * lh >= left) m &= ~(0x00010001u << r);
            }
            if (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
                const uint32_t cnt = (uint32_t)__builtin_popcount(m);
                const uint32_t incl = pf_incl_scan(cnt);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (total <= PF_STAGE - PF_STAGE_FLUSH) {
                    // SPARSE tile (every tile once a threshold stands): the survivors go to the wave's LDS staging ring, and
                    // 64 of them leave with ONE store instruction.  A store per survivor sat in the same in-order vmcnt queue
                    // as the DMA pieces: every counted wait then also waited for slots it did not need yet (the stores behind
                    // them), 0.13 ms of the 4.9 ms step at 10M rows (tools/r04_epilogue_probe.sh).
                    uint32_t idx = st_head + st_n + incl - cnt;
#pragma unroll
                    for (int t = 0; t < NQT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (m & (1u << (16 * t + r))) {
                                const uint32_t row = (uint32_t)row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                stage_keys[idx & (PF_STAGE - 1)] = oi_rank_key(acc[t][r], doc_id_base + row);
                                stage_q[idx & (PF_STAGE - 1)] = 32u * t + li;
                                ++idx;
                            }
                    st_n += total;
                    while (st_n >= PF_STAGE_FLUSH) {
                        PF_FLUSH(PF_STAGE_FLUSH);
                    }
                } else {
                    // DENSE tile (the first chunk, scored without a threshold: every score passes): straight to the pool
                    uint32_t pos[NQT];
#pragma unroll
                    for (int t = 0; t < NQT; ++t) // (both atomics are in flight before the first is waited for; adding 0 is harmless)
                        pos[t] = atomicAdd(&seg_fill[32u * t + li], (uint32_t)__builtin_popcount((m >> (16 * t)) & 0xFFFFu));
#pragma unroll
                    for (int t = 0; t < NQT; ++t) {
                        uint64_t *dst = my_seg + (uint64_t)(32u * t + li) * pool_stride;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (m & (1u << (16 * t + r))) {
                                const uint32_t row = (uint32_t)row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                if (pos[t] < seg_cap) dst[pos[t]] = oi_rank_key(acc[t][r], doc_id_base + row);
                                else *overflow = 1u;
                                ++pos[t];
                            }
                        }
                    }
                }
            }
            cur = nxt;
            nxt = tile_srd(ti + 2);
        }
        if (st_n) {
            PF_FLUSH(st_n);
        }
    }
    __syncthreads(); // every wave's appends are counted
    if (tid < 32 * NQT && tid < n_queries) {
        const uint32_t c = seg_fill[tid];
        seg_cnt[(uint64_t)tid * seg_cnt_stride + blockIdx.x] = c < seg_cap ? c : seg_cap;
    }
}

// ------------------------------------------------------------------ norms, eps, query staging
// Over the rows: X = max_r |x_r| and E = max_r |bf16(x_r) - x_r| (f32, as the bits of non-negative floats:
// atomicMax on the bits orders them; a NaN has the largest bits and poisons the maximum on purpose -- the bound
// does not hold for such a corpus).  The conversion is the screen's own (pf_pack -> v_cvt_pk_bf16_f32), so
// whatever it does to a value (round to nearest even, a flushed denormal) is what E measures.
//   out[0] = bits(X), out[1] = bits(E); (double) out[2..3] = sum_r |x_r|^2, out[4..5] = sum_r |bf16(x_r) - x_r|^2 (the RMS
//   norms the two-class margin is cut at, oi_launch_row_norm_classes)
__global__ __launch_bounds__(256) void pf_row_norm_max_kernel(const float *__restrict__ rows, uint64_t n, uint32_t dim,
                                                              uint32_t *out) {
    double sum_x2 = 0.0, sum_e2 = 0.0;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t nvec = dim >> 2;
    float best = 0.f, best_e = 0.f;
    bool bad = false;
    for (uint64_t r = wave; r < n; r += n_waves) {
        const float4 *x = reinterpret_cast<const float4 *>(rows + r * dim);
        float ss = 0.f, se = 0.f;
        for (uint32_t v = lane; v < nvec; v += 64) {
            const float4 a = x[v];
            ss = fmaf(a.x, a.x, ss); ss = fmaf(a.y, a.y, ss); ss = fmaf(a.z, a.z, ss); ss = fmaf(a.w, a.w, ss);
            const pf_f32x4 f = {a.x, a.y, a.z, a.w};
            const pf_bf16x8 b = pf_pack(f, f);
            const float e0 = (float)b[0] - a.x, e1 = (float)b[1] - a.y, e2 = (float)b[2] - a.z, e3 = (float)b[3] - a.w;
            se = fmaf(e0, e0, se); se = fmaf(e1, e1, se); se = fmaf(e2, e2, se); se = fmaf(e3, e3, se);
        }
        ss = oi_wave_sum(ss);
        se = oi_wave_sum(se);
        const float nm = sqrtf(ss), ne = sqrtf(se);
        bad = bad || !(nm == nm) || !(ne == ne);
        best = nm > best ? nm : best;
        best_e = ne > best_e ? ne : best_e;
        sum_x2 += (double)ss;
        sum_e2 += (double)se;
    }
    if (lane == 0) {
        atomicMax(out, bad ? 0x7FC00000u : __float_as_uint(best));
        atomicMax(out + 1, bad ? 0x7FC00000u : __float_as_uint(best_e));
        atomicAdd(reinterpret_cast<double *>(out + 2), sum_x2);
        atomicAdd(reinterpret_cast<double *>(out + 4), sum_e2);
    }
}

// Two classes of rows (round 4).  The margin above uses the LARGEST |x_r| and |bf16(x_r) - x_r| of the
// corpus, so a handful of long rows widened every query's margin until the survivors no longer fitted and the exact kernel
// took over.  Rows with |x_r| > X0 or error norm > E0 (1.5 x the corpus RMS of each, api.hip) are LONG: listed (at most
// `cap`), marked in a bitmap, left out of the screen's threshold logic by the margin selects (select.hip: skip_bitmap) and
// rescored exactly for every query (pf_rescore_kernel: extra_docs).  The margin is then built from the maxima over the
// other rows, which this kernel takes in the same pass.
//   cls[0] = bits(max |x_r|) and cls[1] = bits(max error norm) over the rows that are NOT long, cls[2] = long rows found
__global__ __launch_bounds__(256) void pf_row_norm_class_kernel(const float *__restrict__ rows, uint64_t n, uint32_t dim, float X0,
                                                                float E0, uint32_t *cls, uint32_t *bitmap, uint32_t *list,
                                                                uint32_t cap) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t nvec = dim >> 2;
    float best = 0.f, best_e = 0.f;
    for (uint64_t r = wave; r < n; r += n_waves) {
        const float4 *x = reinterpret_cast<const float4 *>(rows + r * dim);
        float ss = 0.f, se = 0.f;
        for (uint32_t v = lane; v < nvec; v += 64) {
            const float4 a = x[v];
            ss = fmaf(a.x, a.x, ss); ss = fmaf(a.y, a.y, ss); ss = fmaf(a.z, a.z, ss); ss = fmaf(a.w, a.w, ss);
            const pf_f32x4 f = {a.x, a.y, a.z, a.w};
            const pf_bf16x8 b = pf_pack(f, f);
            const float e0 = (float)b[0] - a.x, e1 = (float)b[1] - a.y, e2 = (float)b[2] - a.z, e3 = (float)b[3] - a.w;
            se = fmaf(e0, e0, se); se = fmaf(e1, e1, se); se = fmaf(e2, e2, se); se = fmaf(e3, e3, se);
        }
        ss = oi_wave_sum(ss);
        se = oi_wave_sum(se);
        const float nm = sqrtf(ss), ne = sqrtf(se);
        if (nm > X0 || ne > E0) { // (the same sums in the same order as the pass that set X0 and E0: the same values)
            if (lane == 0) {
                const uint32_t pos = atomicAdd(cls + 2, 1u);
                if (pos < cap) list[pos] = (uint32_t)r;
                atomicOr(bitmap + (r >> 5), 1u << (r & 31));
            }
        } else {
            best = nm > best ? nm : best;
            best_e = ne > best_e ? ne : best_e;
        }
    }
    if (lane == 0) {
        atomicMax(cls, __float_as_uint(best));
        atomicMax(cls + 1, __float_as_uint(best_e));
    }
}

// The margin of one query (see the header): |s~ - s^| <= eps for every row of the corpus, s^ the rescoring
// kernel's f32 score.  X, E: the corpus maxima above; qn = |q|, qtn = |bf16(q)|, en = |bf16(q) - q|, all f32.
//   * 1.001 covers the f32 rounding of the five norms (sums of <= 1024 squares: 1e-4 at the very most);
//   * squares below 2^-126 may have been flushed out of a norm: each norm is short by at most
//     sqrt(d) 2^-63 < 4e-18, added back here;
//   * products / inputs below 2^-126 may be flushed by the conversions and the matrix pipe: d 2^-120 (X + |q|).
#define PF_NORM_LIMIT 1.0e15f
#define PF_QNORM_MIN 1.0e-12f
__device__ __forceinline__ float pf_eps(float X, float E, float qn, float qtn, float en, uint32_t dim) {
    const float tiny = 4.0e-18f, d = (float)dim;
    const float Xs = X + tiny, Es = E + tiny, qts = qtn + tiny, ens = en + tiny;
    const float acc = d * 2.384185791015625e-07f; // d * 2^-22
    return 1.001f * (Es * qts + Xs * ens + acc * (Xs + Es) * qts) + d * 7.5231638452626401e-37f * (Xs + qn) + 1.0e-30f;
}

// Per query: bf16 copy (RNE, zero padded to n_padded rows) and the screen's margin 2 eps; a norm that is not
// finite, too large for the bf16 products to stay finite, or too small for its rounding errors to be measured in
// f32 opens the exact pipeline instead (gate).
//   state words: eps2[q] (float) at state + q; *gate at gate.
__global__ __launch_bounds__(256) void pf_stage_queries_kernel(const float *__restrict__ q, uint32_t n_queries,
                                                               uint32_t n_padded, uint32_t dim,
                                                               const uint32_t *__restrict__ max_norm_bits,
                                                               uint16_t *__restrict__ out, float *__restrict__ eps2,
                                                               uint32_t *gate) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t row = wave; row < n_padded; row += n_waves) {
        float ss = 0.f, st = 0.f, se = 0.f;
        for (uint32_t k = lane; k < dim; k += 64) {
            uint16_t v = 0;
            if (row < n_queries) {
                const float f = q[(uint64_t)row * dim + k];
                const uint32_t u = __float_as_uint(f);
                v = (u & 0x7F800000u) == 0x7F800000u ? (uint16_t)(u >> 16)                        // inf / NaN: truncate
                                                     : (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); // RNE
                const float ft = __uint_as_float((uint32_t)v << 16), e = ft - f; // exact: f rounded to 8 of its 24 bits
                ss = fmaf(f, f, ss);
                st = fmaf(ft, ft, st);
                se = fmaf(e, e, se);
            }
            out[(uint64_t)row * dim + k] = v;
        }
        if (row < n_queries) {
            ss = oi_wave_sum(ss);
            st = oi_wave_sum(st);
            se = oi_wave_sum(se);
            const float qn = sqrtf(ss), qtn = sqrtf(st), en = sqrtf(se);
            const float X = __uint_as_float(max_norm_bits[0]), E = __uint_as_float(max_norm_bits[1]);
            const bool ok = qn < PF_NORM_LIMIT && qn >= PF_QNORM_MIN && qtn < PF_NORM_LIMIT && X < PF_NORM_LIMIT &&
                            E < PF_NORM_LIMIT && en == en; // false for NaN as well
            if (lane == 0) {
                // A query without a bound gets an infinite margin: its threshold never rises, every key stays
                // inside the margin, the survivors overflow and the exact kernel scores it against every row.
                eps2[row] = ok ? 2.0f * pf_eps(X, E, qn, qtn, en, dim) : __builtin_inff();
                if (!ok) *gate = 1u;
            }
        }
    }
}

// Diagnostics (oi_screen_probe): the screen's raw scores.  Same operands and the same instruction as the screen
// (bf16 rows by pf_pack, the staged bf16 queries, v_mfma_f32_32x32x16_bf16 accumulating 16 k at a time in K
// order), one wave per 32 rows x 32 queries; out[q * n_rows + r].
__global__ __launch_bounds__(64) void pf_probe_kernel(const float *__restrict__ rows, uint64_t row_begin, uint32_t n_rows,
                                                      uint32_t dim, const uint16_t *__restrict__ queries,
                                                      uint32_t n_queries, float *__restrict__ out) {
    const uint32_t lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    const uint32_t r0 = blockIdx.x * 32, q0 = blockIdx.y * 32;
    pf_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const uint32_t row = r0 + li < n_rows ? r0 + li : n_rows - 1; // (rows past the end are not written)
    const float *x = rows + (row_begin + row) * (uint64_t)dim;
    const uint16_t *qq = queries + (uint64_t)(q0 + li) * dim;     // the staged block is zero padded to 32 rows
    for (uint32_t k = 0; k < dim; k += 16) {
        const pf_bf16x8 a = pf_pack(*reinterpret_cast<const pf_f32x4 *>(x + k + 8 * lh),
                                    *reinterpret_cast<const pf_f32x4 *>(x + k + 8 * lh + 4));
        const pf_bf16x8 b = *reinterpret_cast<const pf_bf16x8 *>(qq + k + 8 * lh);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t rr = r0 + (r & 3) + 8 * (r >> 2) + 4 * lh, qx = q0 + li;
        if (rr < n_rows && qx < n_queries) out[(uint64_t)qx * n_rows + rr] = acc[r];
    }
}

// The bf16 screening copy (OI_COSINE_SCREEN_COPY, opt-in): every row converted ONCE, with the conversion the screen
// uses on the fly (pf_pack), so E -- and therefore the bound -- is the same; the screen then reads 2 d bytes per row.
__global__ __launch_bounds__(256) void pf_make_copy_kernel(const float *__restrict__ rows, uint64_t n_vec8, uint4 *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec8; i += (uint64_t)gridDim.x * blockDim.x) {
        const pf_f32x4 a = *reinterpret_cast<const pf_f32x4 *>(rows + i * 8), b = *reinterpret_cast<const pf_f32x4 *>(rows + i * 8 + 4);
        const pf_bf16x8 v = pf_pack(a, b);
        out[i] = *reinterpret_cast<const uint4 *>(&v);
    }
}
int oi_launch_make_screen_copy(oi_ctx *ctx, const float *rows, uint64_t n, uint32_t dim, uint16_t *out) {
    const uint64_t n_vec8 = n * dim / 8; // dim is a multiple of 16 on this path
    if (n_vec8 == 0) return OI_OK;
    const uint64_t blocks = std::min<uint64_t>((n_vec8 + 255) / 256, (uint64_t)ctx->num_cus * 16);
    hipLaunchKernelGGL(pf_make_copy_kernel, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, rows, n_vec8, reinterpret_cast<uint4 *>(out));
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

// ------------------------------------------------------------------ exact rescoring of the survivors
// One wave per (query, survivor): s = sum_k x_k q_k in f32 (fma chain per lane, butterfly sum), written as an
// exact rank key at the same slot of the output pool's carry region.
__global__ __launch_bounds__(256) void pf_rescore_kernel(const float *__restrict__ rows, uint32_t dim, uint32_t doc_id_base,
                                                         uint64_t n_rows, const float *__restrict__ queries,
                                                         const uint64_t *__restrict__ in_pools, const uint32_t *__restrict__ in_cnt,
                                                         uint64_t in_stride, uint32_t cap, uint64_t *out_pools,
                                                         uint32_t *out_cnt, uint64_t out_stride,
                                                         const uint32_t *__restrict__ extra_docs, uint32_t n_extra,
                                                         const uint32_t *__restrict__ spec_max, const uint32_t *__restrict__ tau_final,
                                                         uint32_t *gate, uint32_t *fail_host) {
    // spec_max != null: the check of the speculative thresholds rides along (spec_max[q] > tau_final[q] voids the batch's screen; no launch of its own): the
    // gated exact pipeline is enqueued after this kernel
    if (spec_max && blockIdx.x == 0 && threadIdx.x == 0 && spec_max[blockIdx.y] > tau_final[blockIdx.y]) {
        gate[0] = 1u;
        if (fail_host) *fail_host = 1u;
    }
    // extra_docs: the index's LONG rows (local row numbers, pf_row_norm_class_kernel) -- scored for every query after its
    // survivors, whatever the screen made of them
    const uint32_t q = blockIdx.y, lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    uint32_t c0 = in_cnt[q];
    c0 = c0 < cap ? c0 : cap;
    const uint32_t c = c0 + n_extra;
    const uint32_t nvec = dim >> 2;
    const float4 *qv = reinterpret_cast<const float4 *>(queries + (uint64_t)q * dim);
    // four survivors per wave and trip: their rows' loads are all in flight before the first reduction (one row at a
    // time this kernel was a chain of HBM round trips: 68 us for 2450 survivors x 64 queries at d = 768)
    for (uint32_t i0 = wave * 4u; i0 < c; i0 += n_waves * 4u) {
        uint32_t doc[4];
        const float4 *x[4];
        float a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u < c ? i0 + u : c - 1u; // (past the end: the last survivor again, not written)
            doc[u] = i < c0 ? oi_rank_key_doc(in_pools[(uint64_t)q * in_stride + i]) : doc_id_base + extra_docs[i - c0];
            const uint64_t r = (uint64_t)(doc[u] - doc_id_base);
            x[u] = reinterpret_cast<const float4 *>(rows + (r < n_rows ? r : 0) * dim);
            a[u] = 0.f;
        }
        for (uint32_t v = lane; v < nvec; v += 64) {
            const float4 yv = qv[v];
            float4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = oi_load_stream(x[u] + v); // (each survivor row is read once)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // (one v_fma_f32 each, NOT the packed form the vectoriser makes of two chains: oi_device.h, oi_fma_unpacked)
                a[u] = oi_fma_unpacked(xv[u].x, yv.x, a[u]); a[u] = oi_fma_unpacked(xv[u].y, yv.y, a[u]);
                a[u] = oi_fma_unpacked(xv[u].z, yv.z, a[u]); a[u] = oi_fma_unpacked(xv[u].w, yv.w, a[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t r = (uint64_t)(doc[u] - doc_id_base);
            float s = oi_wave_sum(a[u]);
            if (!(r < n_rows)) s = 0.f; // (a key outside the shard: cannot happen; scored 0 as before)
            if (lane == 0 && i0 + u < c) out_pools[(uint64_t)q * out_stride + i0 + u] = oi_rank_key(s, doc[u]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out_cnt[q] = c;
}

// ------------------------------------------------------------------ speculative thresholds (round 5)
// The screen's threshold after m of n rows is PROVEN: tau~_m - 2 eps, tau~_m the k'-th best screen score of those m rows.  It is
// weak while m << n (k' of 28 672 rows: 3.5 % of the next chunk pass; the final one passes 0.01 %), and every survivor of a weak
// threshold costs staging, pool traffic and select time.  What the final threshold will be can be PREDICTED from the same m rows:
// if they are a fair sample, the k'-th best of n sits near the (k' m / n)-th best of m.  pf_spec_kernel takes the r-th best screen
// score so far, r = 3 k' m / n + 12 (three times the expected rank plus twelve: for rows in any exchangeable order the chance
// that fewer than k' of the n rows reach it is < 1e-9 whatever m), lowers it by the same 2 eps, and hands the LARGER of it and the
// proven threshold to the next chunk.  A prediction is not a bound -- so it is CHECKED: pf_rescore_kernel (first thing) compares the largest
// speculative threshold used for a query with the proven one at the end, tau~_n - 2 eps.  T_spec <= tau~_n - 2 eps means every row
// the speculation dropped (s~ < T_spec) would have been dropped by the final proven threshold as well: the survivor set still
// holds every row of the exact list, the lists are the proven screen's.  Otherwise (a corpus whose first rows are not a fair
// sample: sorted by time or topic) the gate opens and the exact pipeline rescores the batch in the same call, and the host,
// told through a pinned flag, stops speculating for a while (api.hip: spec_backoff).
__global__ __launch_bounds__(256) void pf_spec_kernel(const uint64_t *__restrict__ pools, const uint32_t *__restrict__ carry_cnt,
                                                      uint64_t stride, uint32_t cap, uint32_t r, const float *__restrict__ eps2,
                                                      const uint32_t *__restrict__ tau_keys, uint32_t *spec_tau, uint32_t *spec_max) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_rank;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    uint32_t c = carry_cnt[q];
    if (c > cap) c = cap;
    const uint32_t tau = tau_keys[q];
    if (r == 0 || c < r) { // (uniform over the workgroup) too few keys kept: the proven threshold as it is
        if (tid == 0) spec_tau[q] = tau;
        return;
    }
    const uint64_t *k = pools + (uint64_t)q * stride; // the carry region: an unsorted superset of the k' best rank keys so far
    if (tid == 0) { s_prefix = 0; s_rank = r; }
    // the r-th LARGEST 32-bit score key: four passes over 8-bit digits, most significant first.  The digit that holds the rank is
    // found by wave 0 alone (lane l owns digits 4 l .. 4 l + 3; a shuffle suffix sum over the lanes): three barriers per pass.
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, rank = s_rank, himask = shift == 24 ? 0u : 0xFFFFFFFFu << (shift + 8);
        for (uint32_t i = tid; i < c; i += 256) {
            const uint32_t sk = (uint32_t)(k[i] >> 32);
            if ((sk & himask) == prefix) atomicAdd(&hist[(sk >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const uint32_t own = h0 + h1 + h2 + h3;
            uint32_t v = own; // -> keys whose digit lies in this lane's four or in a higher lane's
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_down(v, o, 64);
                if (tid + o < 64) v += t;
            }
            uint32_t above = v - own; // keys with a digit above this lane's
            const uint32_t hd[4] = {h3, h2, h1, h0};
#pragma unroll
            for (int d = 0; d < 4; ++d) { // this lane's digits from the top: exactly one digit of one lane holds the rank-th largest
                if (above < rank && rank <= above + hd[d]) {
                    s_prefix = prefix | ((uint32_t)(4 * tid + 3 - d) << shift);
                    s_rank = rank - above;
                }
                above += hd[d];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float s = oi_key_f32(s_prefix) - eps2[q]; // (a query without a bound has eps2 = +inf: -inf, no speculation)
        const uint32_t T = s == s ? oi_f32_key(s) : 0u;
        const uint32_t st = T > tau ? T : tau;
        spec_tau[q] = st;
        if (st > tau && st > spec_max[q]) spec_max[q] = st;
    }
}
int oi_launch_spec_threshold(oi_ctx *ctx, const PoolView &pool, uint32_t n_queries, uint32_t r, const float *eps2, uint32_t *spec_tau,
                             uint32_t *spec_max) {
    if (n_queries == 0) return OI_OK;
    ProfScope ps(ctx, "spec");
    hipLaunchKernelGGL(pf_spec_kernel, dim3(n_queries), dim3(256), 0, ctx->stream, pool.keys, pool.carry_cnt, pool.stride, pool.carry_cap, r,
                       eps2, pool.tau_keys, spec_tau, spec_max);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

// ------------------------------------------------------------------ host
bool oi_cosine_screen_supported(uint32_t dim) { return dim == 384 || dim == 768; }

// Pool geometry of one chunk: one segment per workgroup; its four waves take 4 tiles per round.
void oi_cosine_screen_geometry(const oi_ctx *ctx, uint64_t n_rows, uint32_t *n_segs, uint32_t *seg_cap) {
    const uint64_t n_tiles = (n_rows + PF_TILE_ROWS - 1) / PF_TILE_ROWS;
    const uint64_t quads = (n_tiles + 3) / 4;
    // The persistent screen takes 7/8 of the CUs, one workgroup each.  It is HBM-bound: 208..224 of 256 CUs stream the corpus as
    // fast as 256 (4.82-4.92 vs 4.97 ms per 10M-row batch), and the CUs left free run the small kernels of whatever else is in
    // flight -- the BM25 leg on the side stream, the selects and the rescoring of another batch scored through a view of the
    // same shard (oi_index_view): 0.817 -> 0.713 ms per batch at 1.25M rows with two batches in flight (tools/dual_stream_probe.py;
    // 192: 0.69-0.72, 160: 0.75, and from 176 down the single-stream time grows).  OI_SCREEN_CUS: A/B switch.
    const char *cus_s = oi_ablation_env("OI_SCREEN_CUS"); // (read per call: sweep tools change it between runs of one process)
    const uint64_t cus = cus_s ? (uint64_t)std::max(1, atoi(cus_s)) : std::max<uint64_t>(1, (uint64_t)ctx->num_cus * 7 / 8);
    static const bool small_full = oi_ablation_env("OI_SCREEN_SMALL_FULL") != nullptr; // A/B: a chunk of <= one quad per CU takes every CU
    const uint64_t grid = small_full && quads <= (uint64_t)ctx->num_cus ? (quads ? quads : 1) : (quads < cus ? (quads ? quads : 1) : cus);
    *n_segs = (uint32_t)grid;
    *seg_cap = (uint32_t)((quads + grid - 1) / grid) * 4 * PF_TILE_ROWS;
}

int oi_launch_row_norm_classes(oi_ctx *ctx, const float *rows, uint64_t n, uint32_t dim, float X0, float E0, uint32_t *cls,
                               uint32_t *bitmap, uint32_t *list, uint32_t cap) {
    OI_HIP_CHECK(hipMemsetAsync(cls, 0, 16, ctx->stream));
    OI_HIP_CHECK(hipMemsetAsync(bitmap, 0, ((n + 31) / 32) * 4, ctx->stream));
    uint64_t blocks = (n + 3) / 4;
    const uint64_t capb = (uint64_t)ctx->num_cus * 8;
    if (blocks > capb) blocks = capb;
    hipLaunchKernelGGL(pf_row_norm_class_kernel, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, rows, n, dim, X0, E0, cls, bitmap,
                       list, cap);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

int oi_launch_row_norm_max(oi_ctx *ctx, const float *rows, uint64_t n, uint32_t dim, uint32_t *max_bits) {
    OI_HIP_CHECK(hipMemsetAsync(max_bits, 0, 24, ctx->stream)); // [0] = X, [1] = E, [2..5] = the two sums of squares (double)
    if (n == 0) return OI_OK;
    uint64_t blocks = (n + 3) / 4;
    const uint64_t cap = (uint64_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(pf_row_norm_max_kernel, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, rows, n, dim, max_bits);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

// bf16 queries (padded to a multiple of 32 rows) + per-query margins + gate, once per search
int oi_launch_screen_stage(oi_ctx *ctx, const float *d_queries, uint32_t n_queries, uint32_t dim,
                           const uint32_t *max_norm_bits, uint16_t *q_bf16, float *eps2, uint32_t *gate) {
    const uint32_t n_padded = (n_queries + 31u) & ~31u;
    hipLaunchKernelGGL(pf_stage_queries_kernel, dim3((n_padded + 3) / 4), dim3(256), 0, ctx->stream, d_queries, n_queries,
                       n_padded, dim, max_norm_bits, q_bf16, eps2, gate);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

template <int D, int NQT>
static int launch_screen(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, const uint16_t *q,
                         uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
    constexpr int NKC = D / PF_SLOT_K, NBUF = NKC % 8 == 0 ? 8 : (NKC % 6 == 0 ? 6 : NKC);
    constexpr size_t smem = 4 * NBUF * PF_SLOT_BYTES + 64 * 4 + PF_STAGE_LDS;
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(cosine_screen_filter<D, NQT>), (size_t)(smem)));
    hipLaunchKernelGGL((cosine_screen_filter<D, NQT>), dim3(p.n_segs), dim3(256), smem, ctx->stream, rows, row_begin,
                       row_end, q, nq, doc_id_base, p.keys, p.seg_cnt, p.seg_cnt_stride, p.tau_keys, p.stride,
                       p.carry_cap, p.seg_cap, p.overflow);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

// All queries of a batch over rows [row_begin, row_end): the bf16 screen.  q_bf16: staged by
// oi_launch_screen_stage.  One corpus pass per 64 queries.
int oi_launch_cosine_screen_chunk(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end, uint32_t dim,
                                  const uint16_t *q_bf16, uint32_t n_queries, uint32_t doc_id_base, PoolView &pool) {
    OI_REQUIRE(oi_cosine_screen_supported(dim), "cosine screen: dim %u not instantiated (384, 768)", dim);
    oi_cosine_screen_geometry(ctx, row_end > row_begin ? row_end - row_begin : 0, &pool.n_segs, &pool.seg_cap);
    OI_REQUIRE(pool.n_segs <= pool.seg_cnt_stride && pool.carry_cap + (uint64_t)pool.n_segs * pool.seg_cap <= pool.stride,
               "cosine screen: chunk does not fit the candidate pool");
    if (row_end <= row_begin || n_queries == 0) return OI_OK;
    ProfScope ps(ctx, "cosine");
    for (uint32_t q0 = 0; q0 < n_queries; q0 += 64) {
        const uint32_t nq_here = std::min(64u, n_queries - q0);
        PoolView p = pool;
        p.keys += (uint64_t)q0 * pool.stride;
        p.carry_cnt += q0;
        p.seg_cnt += (uint64_t)q0 * pool.seg_cnt_stride;
        p.tau_keys += q0;
        static const bool tau_max = oi_ablation_env("OI_SCREEN_TAU_MAX") != nullptr; // A/B (WRONG results): chunks of >= 1M rows pass nothing
        if (tau_max && row_end - row_begin >= (1u << 20)) {
            DevBuf &tb = ctx->buf("abl_tau_max");
            if (!tb.p) { OI_CHECK(tb.ensure(4096 * 4)); OI_HIP_CHECK(hipMemsetAsync(tb.p, 0xFF, 4096 * 4, ctx->stream)); }
            p.tau_keys = tb.as<uint32_t>();
        }
        const uint16_t *qptr = q_bf16 + (uint64_t)q0 * dim;
        const bool two = nq_here > 32;
        if (dim == 768) {
            if (two) OI_CHECK((launch_screen<768, 2>(ctx, rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)));
            else OI_CHECK((launch_screen<768, 1>(ctx, rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)));
        } else {
            if (two) OI_CHECK((launch_screen<384, 2>(ctx, rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)));
            else OI_CHECK((launch_screen<384, 1>(ctx, rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)));
        }
    }
    return OI_OK;
}

// Exact scores of the screen's survivors: in.carry region (in.carry_cnt keys per query) -> out.carry region.
int oi_launch_rescore(oi_ctx *ctx, const float *rows, uint64_t n_rows, uint32_t dim, uint32_t doc_id_base,
                      const float *d_queries, uint32_t n_queries, const PoolView &in, const PoolView &out,
                      const uint32_t *extra_docs, uint32_t n_extra, const uint32_t *spec_max, const uint32_t *tau_final,
                      uint32_t *gate, uint32_t *fail_host) {
    if (n_queries == 0) return OI_OK;
    OI_REQUIRE(out.carry_cap >= in.carry_cap + n_extra, "rescore: output pool too small");
    ProfScope ps(ctx, "rescore");
    hipLaunchKernelGGL(pf_rescore_kernel, dim3(64, n_queries), dim3(256), 0, ctx->stream, rows, dim, doc_id_base, n_rows,
                       d_queries, in.keys, in.carry_cnt, in.stride, in.carry_cap, out.keys, out.carry_cnt, out.stride, extra_docs,
                       n_extra, spec_max, tau_final, gate, fail_host);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

// Diagnostics: raw screen scores s~ of rows [row_begin, row_begin + n_rows) for every query of the staged block.
int oi_launch_screen_probe(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint32_t n_rows, uint32_t dim,
                           const uint16_t *q_bf16, uint32_t n_queries, float *d_out) {
    if (n_rows == 0 || n_queries == 0) return OI_OK;
    OI_REQUIRE(dim % 16 == 0, "screen probe: dim %u is not a multiple of 16", dim);
    hipLaunchKernelGGL(pf_probe_kernel, dim3((n_rows + 31) / 32, (n_queries + 31) / 32), dim3(64), 0, ctx->stream, rows,
                       row_begin, n_rows, dim, q_bf16, n_queries, d_out);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}
