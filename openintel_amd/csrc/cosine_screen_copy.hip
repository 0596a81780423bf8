// cosine_screen_copy.hip -- the bf16 screen of cosine_prefilter.hip reading the index's bf16 SCREENING COPY of the rows
// (round 5: the default batch scorer of an f32 corpus whenever the index holds the copy -- oi_index_finalize makes it when it
// fits the stated HBM budget).
//
// Builder-defined like the rest of the retrieval path (the reference has none).
//
// What changes against cosine_screen_filter and what does not.  The copy is bf16(x) made ONCE with the screen's own
// conversion (pf_make_copy_kernel: v_cvt_pk_bf16_f32), so the screen's products, the measured E = max |bf16(x) - x|, the
// margin eps_q, the survivors and their exact f32 rescoring FROM THE f32 ROWS are the same: the lists are the same bit for
// bit (tests/test_gpu_prefilter.py).  Only the bytes differ: 2 d per row and batch instead of 4 d -- the kernel is an HBM
// stream, so that is the whole point.
//
// Kernel shape: cosine_screen_filter's with the conversion gone.  One persistent workgroup on 7/8 of the CUs, 4 waves, no
// K-split: a wave holds all 64 queries over the whole K as bf16 B operands, owns whole 32-row tiles and streams them through
// its own LDS ring of 4 KiB slots (32 rows x 64 bf16) with buffer_load ... lds, ordered by counted s_waitcnt vmcnt.  Per 16 k
// of a tile: one conflict-free ds_read_b128, NQT MFMAs.  Two things are new:
//   * the ring index is a RUN-TIME scalar (one s_add + s_cselect per slot, one v_add per fragment read), so the depth NBUF is
//     free of the tile's slot count (d = 768: 12 slots per tile; the compile-time ring of cosine_bf16.hip has to divide it:
//     6 slots = 20 KB in flight per wave, where the f32 screen keeps 28).  The kernel has 16 K cycles per tile and wave at
//     the HBM rate and needs ~3 K of matrix pipe: the extra scalar work is free, bytes in flight are what it is short of.
//   * the refill of slot s + P always issues (an empty descriptor past the wave's last tile returns zeros), so every
//     counted wait is the same constant and the tile loop has no tail cases.
// Survivors leave through the per-wave LDS staging ring of the f32 screen (64 keys per store instruction: a store per
// survivor sits in the same in-order vmcnt queue as the DMA pieces and makes every counted wait wait for more than it needs).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "oi_device.h"
#include "oi_internal.h"

typedef float sc_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 sc_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t sc_u32x4 __attribute__((ext_vector_type(4)));

#ifndef OI_TILE_CONTIG
#define OI_TILE_CONTIG 0
#endif
#define SC_TILE_ROWS 32
#define SC_SLOT_K 64                 // bf16 of K per ring slot row (128 B)
#define SC_SLOT_BYTES (SC_TILE_ROWS * 128)
#define SC_STAGE 256                 // staged survivors per wave (a power of two)
#define SC_STAGE_FLUSH 64u
#define SC_STAGE_LDS (4 * SC_STAGE * 12)

__device__ __forceinline__ uint32_t sc_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ sc_u32x4 sc_make_srd(const uint16_t *base, uint64_t bytes) {
    const uint64_t b = (uint64_t)base;
    sc_u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xFFFFu); // stride 0
    r[2] = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes));
    r[3] = 0x00020000u;
    return r;
}
// One 1-KiB LDS-DMA piece (8 rows x 128 B).  Lanes past the descriptor's end read as zero: the ragged last tile and the
// tiles after the last one (empty descriptor) need no branch.  hipcc does not see these loads: sc_wait<N>() orders them.
__device__ __forceinline__ void sc_issue_piece(const sc_u32x4 &srd, uint32_t voff, uint32_t soff, uint32_t lds_dst) {
#if defined(SC_AGG_NO_DMA) && SC_AGG_NO_DMA == 1
    // (variant builds, tools/r05_victim_probe.py: which trait of this kernel disturbs a neighbour wave?  The same bytes loaded
    // into registers nobody reads instead of into LDS; the ring is zeroed at the start, every score is 0: results WRONG.)
    const uint32_t so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("buffer_load_dwordx4 a[100:103], %0, %1, %2 offen " OI_DMA_NT : : "v"(voff), "s"(srd), "s"(so) : "memory", "a100", "a101", "a102", "a103");
#elif defined(SC_AGG_NO_DMA)
    (void)srd; (void)voff; (void)soff; (void)lds_dst; // (variant builds: no loads at all)
#else
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
#endif
}
template <int I, int N, class F>
__device__ __forceinline__ void sc_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sc_static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void sc_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ uint32_t sc_incl_scan(uint32_t v) { // wave-wide inclusive prefix sum (DPP, no LDS)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return v;
}
// The first NF staged entries of the wave leave for the pool (cosine_prefilter.hip: PF_FLUSH -- same contract; the compiler
// barriers keep other lanes' staging writes in front of these reads and these reads in front of the next tile's writes).
#define SC_FLUSH(NF)                                                                                                   \
    do {                                                                                                               \
        const uint32_t nf_ = (NF);                                                                                     \
        asm volatile("" ::: "memory");                                                                                 \
        if (lane < nf_) {                                                                                              \
            const uint32_t i_ = (st_head + lane) & (SC_STAGE - 1);                                                     \
            const uint64_t key_ = stage_keys[i_];                                                                      \
            const uint32_t q_ = stage_q[i_];                                                                           \
            const uint32_t pos_ = atomicAdd(&seg_fill[q_], 1u);                                                        \
            if (pos_ < seg_cap) my_seg[(uint64_t)q_ * pool_stride + pos_] = key_;                                      \
            else *overflow = 1u;                                                                                       \
        }                                                                                                              \
        asm volatile("" ::: "memory");                                                                                 \
        st_head = (st_head + nf_) & (SC_STAGE - 1);                                                                    \
        st_n -= nf_;                                                                                                   \
    } while (0)

template <int D, int NQT, int NBUF>
__global__ __launch_bounds__(256, 1) void cosine_copy_screen(
    const uint16_t *__restrict__ rows, uint64_t row_begin, uint64_t row_end,
    const uint16_t *__restrict__ queries, // bf16 [32*NQT][D], zero padded (pf_stage_queries_kernel)
    uint32_t n_queries, uint32_t doc_id_base, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride,
    const uint32_t *tau_keys, uint64_t pool_stride, uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow) {
    constexpr int NKC = D / SC_SLOT_K;    // ring slots per tile
    constexpr int P = NBUF - 1;           // slots in flight ahead of the one being consumed
    constexpr int KSTEPS = D / 16;        // MFMA groups per tile: four per slot
    constexpr uint32_t RING = NBUF * SC_SLOT_BYTES;
    static_assert(D % SC_SLOT_K == 0 && P >= 1 && P <= 2 * NKC, "unsupported ring depth for this D");
    static_assert(NQT * KSTEPS * 4 <= 400, "the query block must fit the register file");
    static_assert(4 * RING + 256 + SC_STAGE_LDS <= 160 * 1024, "LDS");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *ring = smem;                                          // [4][NBUF][4 KiB]
    uint32_t *seg_fill = reinterpret_cast<uint32_t *>(smem + 4 * RING);  // [64]

    OI_CLAIM_WHOLE_SIMD(); // (MFMA kernel: nothing else may run on this CU -- oi_device.h)
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t li = lane & 31, lh = lane >> 5;
    uint64_t *stage_keys = reinterpret_cast<uint64_t *>(smem + 4 * RING + 256) + w * SC_STAGE;
    uint32_t *stage_q = reinterpret_cast<uint32_t *>(smem + 4 * RING + 256 + 4 * SC_STAGE * 8) + w * SC_STAGE;
    uint32_t st_head = 0, st_n = 0; // wave-uniform: first staged entry (mod SC_STAGE), staged entries (< SC_STAGE_FLUSH between tiles)

    // ---- every query over the whole K, in registers for the whole launch: B[k = 16 s + 8 lh + 0..7][n = li]
    sc_bf16x8 qreg[NQT][KSTEPS];
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            qreg[t][s] = *reinterpret_cast<const sc_bf16x8 *>(queries + (uint64_t)(32 * t + li) * D + 16 * s + 8 * lh);
    // thresholds (tau~ - 2 eps) as floats: see cosine_screen_filter -- one v_cmp per score; no query in the slot = NaN
    float tauf[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const uint32_t q = 32u * t + li;
        const uint32_t k = q < n_queries ? tau_keys[q] : 0xFFFFFFFFu;
        tauf[t] = k <= 0x007FFFFFu ? -__builtin_inff() : oi_key_f32(k);
    }
    if (tid < 32 * NQT) seg_fill[tid] = 0;
#ifdef SC_AGG_NO_DMA
    for (uint32_t i = tid; i < 4 * RING / 4; i += 256) reinterpret_cast<uint32_t *>(ring)[i] = 0u;
#endif
    __syncthreads(); // the only barrier before the end: seg_fill is zero before any wave appends

    // ---- tiles of this WAVE: (blockIdx.x * 4 + w), + 4 * gridDim.x, ...
    const uint64_t n_rows = row_end - row_begin;
    const uint64_t n_tiles = (n_rows + SC_TILE_ROWS - 1) / SC_TILE_ROWS;
#if OI_TILE_CONTIG // A/B: every wave owns a CONTIGUOUS run of tiles (sequential pages) instead of every stride-th tile
    const uint64_t n_waves_ = (uint64_t)gridDim.x * 4, per_ = (n_tiles + n_waves_ - 1) / n_waves_;
    const uint64_t first = ((uint64_t)blockIdx.x * 4 + w) * per_, stride = 1;
    const uint64_t my_nt = first < n_tiles ? (n_tiles - first < per_ ? n_tiles - first : per_) : 0;
#else
    const uint64_t first = (uint64_t)blockIdx.x * 4 + w, stride = (uint64_t)gridDim.x * 4;
    const uint64_t my_nt = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
#endif
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
        const uint32_t ring_w = sc_lds_addr(ring) + w * RING;
        const unsigned char *ring_rd = ring + w * RING;
        // fragment of MFMA group g of a slot: row li, bf16 16 g + 8 lh + 0..7 = logical 16-B column 2g + lh
        uint32_t frag_off[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) frag_off[g] = li * 128 + (((2 * g + lh) ^ ((li >> 1) & 7)) << 4);

        auto tile_row0 = [&](uint64_t ti) { return row_begin + (first + ti * stride) * (uint64_t)SC_TILE_ROWS; };
        auto tile_srd = [&](uint64_t ti) { // past this wave's last tile: an EMPTY descriptor (loads return zeros)
            const uint64_t r0 = tile_row0(ti < my_nt ? ti : 0);
            return sc_make_srd(rows + r0 * D, ti < my_nt ? (row_end - r0) * (uint64_t)(D * 2) : 0ull);
        };
        sc_u32x4 s0 = tile_srd(0), s1 = tile_srd(1), s2 = tile_srd(2);
        // Every load hipcc knows about (queries, thresholds) is retired HERE, with a wait it models:
        // otherwise it re-waits for them inside the tile loop and drains the DMA ring.
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only
        sc_static_for<0, P>([&](auto j_) { // prologue: logical slots 0..P-1 (tile j / NKC, slot j % NKC) into ring slots 0..P-1
            constexpr int j = decltype(j_)::value;
            constexpr int tj = j / NKC, kj = j % NKC;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                sc_issue_piece(tj == 0 ? s0 : (tj == 1 ? s1 : s2), voff[m], kj * 128, ring_w + j * SC_SLOT_BYTES + m * 1024);
        });
        // ring offsets (bytes, wave-uniform): the slot being consumed, and the one vacated before it = the refill target
        uint32_t rd_off = 0, wr_off = (NBUF - 1) * SC_SLOT_BYTES;

        for (uint64_t ti = 0; ti < my_nt; ++ti) {
            sc_f32x16 acc[NQT];
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

            // Slot kc of this tile sits at rd_off.  Per MFMA group (kc, g): read the next fragment, NQT MFMAs on the
            // current one, DMA piece g of logical slot kc + P into the slot vacated last (wr_off); after g == 3 the
            // counted wait retires slot kc + 1 (P - 1 younger slots stay in flight) and the offsets move on.
            sc_wait<4 * (P - 1)>();
            sc_bf16x8 a_cur = *reinterpret_cast<const sc_bf16x8 *>(ring_rd + rd_off + frag_off[0]);
            sc_static_for<0, NKC * 4>([&](auto gi_) {
                constexpr int gi = decltype(gi_)::value;
                constexpr int kc = gi / 4, g = gi % 4;
                constexpr int sn = kc + P;           // logical slot (relative to this tile) refilled during this slot
                constexpr int tn = sn / NKC, kn = sn % NKC;
                sc_bf16x8 a_nxt = a_cur;
                if constexpr (g < 3) a_nxt = *reinterpret_cast<const sc_bf16x8 *>(ring_rd + rd_off + frag_off[g + 1]);
#ifndef SC_AGG_NO_MFMA // (variant builds: the stream without the matrix instructions; every score 0, results WRONG)
#pragma unroll
                for (int t = 0; t < NQT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur, qreg[t][gi], acc[t], 0, 0, 0);
#else
                asm volatile("" : : "v"(a_cur));
#endif
                sc_issue_piece(tn == 0 ? s0 : (tn == 1 ? s1 : s2), voff[g], kn * 128, ring_w + wr_off + g * 1024);
                if constexpr (g == 3) {
                    wr_off = rd_off;
                    rd_off = rd_off + SC_SLOT_BYTES == RING ? 0u : rd_off + SC_SLOT_BYTES;
                    if constexpr (kc + 1 < NKC) {
                        sc_wait<4 * (P - 1)>();
                        a_nxt = *reinterpret_cast<const sc_bf16x8 *>(ring_rd + rd_off + frag_off[0]);
                    }
                }
                a_cur = a_nxt;
            });

            // ---- filter + append, straight out of the accumulators (cosine_screen_filter's epilogue): register r of
            // query tile t holds D[row (r&3) + 8 (r>>2) + 4 lh][query 32 t + li]
            const uint64_t row0 = tile_row0(ti);
            uint32_t m = 0;
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) m |= acc[t][r] >= tauf[t] ? 1u << (16 * t + r) : 0u;
            if (row_end - row0 < (uint64_t)SC_TILE_ROWS) { // the ragged last tile: rows past the end read as zeros
                const uint32_t left = (uint32_t)(row_end - row0);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * lh >= left) m &= ~(0x00010001u << r);
            }
            if (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
                const uint32_t cnt = (uint32_t)__builtin_popcount(m);
                const uint32_t incl = sc_incl_scan(cnt);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (total <= SC_STAGE - SC_STAGE_FLUSH) {
                    // SPARSE tile (every tile once a threshold stands): staged, 64 leave with one store instruction
                    uint32_t idx = st_head + st_n + incl - cnt;
#pragma unroll
                    for (int t = 0; t < NQT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (m & (1u << (16 * t + r))) {
                                const uint32_t row = (uint32_t)row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                stage_keys[idx & (SC_STAGE - 1)] = oi_rank_key(acc[t][r], doc_id_base + row);
                                stage_q[idx & (SC_STAGE - 1)] = 32u * t + li;
                                ++idx;
                            }
                    st_n += total;
                    while (st_n >= SC_STAGE_FLUSH) {
                        SC_FLUSH(SC_STAGE_FLUSH);
                    }
                } else {
                    // DENSE tile (the first chunk, scored without a threshold: every score passes): straight to the pool
                    uint32_t pos[NQT];
#pragma unroll
                    for (int t = 0; t < NQT; ++t)
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
            s0 = s1;
            s1 = s2;
            s2 = tile_srd(ti + 3);
        }
        if (st_n) {
            SC_FLUSH(st_n);
        }
        sc_wait<0>(); // the zero-filling refills issued past the last tile have landed before the LDS goes back
    }
    __syncthreads(); // every wave's appends are counted
    if (tid < 32 * NQT && tid < n_queries) {
        const uint32_t c = seg_fill[tid];
        seg_cnt[(uint64_t)tid * seg_cnt_stride + blockIdx.x] = c < seg_cap ? c : seg_cap;
    }
}

// ------------------------------------------------------------------ host
// Ring depth.  Measured on one box at 10M x 768, 64 queries (tools/r05_copy_sweep.py, profiles/r05a_copy_sweep.jsonl): 6, 7, 8
// and 9 slots all stream at 0.788-0.790 of the HBM spec on 224 CUs and 0.807-0.811 on 240 -- 20 KB in flight per wave is
// already enough, depth is NOT what holds the kernel (the f32 screen's 0.83 on twice the bytes puts the stream itself at
// 7.0 TB/s and a fixed ~58 us per launch on top: ramp, query preload, tail).  8 it is.  OI_COPY_NBUF (ablation builds): 6 / 7 / 9.
#ifndef OI_COPY_NBUF_DEFAULT
#define OI_COPY_NBUF_DEFAULT 8
#endif

template <int D, int NQT, int NBUF>
static int launch_copy_screen(oi_ctx *ctx, const uint16_t *rows, uint64_t row_begin, uint64_t row_end, const uint16_t *q,
                              uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
    constexpr size_t smem = 4 * NBUF * SC_SLOT_BYTES + 64 * 4 + SC_STAGE_LDS;
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(cosine_copy_screen<D, NQT, NBUF>), (size_t)(smem)));
    hipLaunchKernelGGL((cosine_copy_screen<D, NQT, NBUF>), dim3(p.n_segs), dim3(256), smem, ctx->stream, rows, row_begin,
                       row_end, q, nq, doc_id_base, p.keys, p.seg_cnt, p.seg_cnt_stride, p.tau_keys, p.stride,
                       p.carry_cap, p.seg_cap, p.overflow);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

template <int D, int NQT>
static int launch_copy_screen_depth(oi_ctx *ctx, int nbuf, const uint16_t *rows, uint64_t row_begin, uint64_t row_end,
                                    const uint16_t *q, uint32_t nq, uint32_t doc_id_base, const PoolView &p) {
#ifdef OI_ABLATION
    if (nbuf == 6) return launch_copy_screen<D, NQT, 6>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
    if (nbuf == 9) return launch_copy_screen<D, NQT, 9>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
    if (nbuf == 7) return launch_copy_screen<D, NQT, 7>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
#endif
    (void)nbuf;
    return launch_copy_screen<D, NQT, OI_COPY_NBUF_DEFAULT>(ctx, rows, row_begin, row_end, q, nq, doc_id_base, p);
}

// All queries of a batch over rows [row_begin, row_end) of the index's bf16 screening copy.  q_bf16: staged by
// oi_launch_screen_stage (the same block the f32 screen reads).  One pass over the copy per 64 queries.  Pool geometry:
// the f32 screen's (oi_cosine_screen_geometry) -- the two kernels are interchangeable chunk by chunk.
int oi_launch_cosine_screen_copy_chunk(oi_ctx *ctx, const uint16_t *copy_rows, uint64_t row_begin, uint64_t row_end, uint32_t dim,
                                       const uint16_t *q_bf16, uint32_t n_queries, uint32_t doc_id_base, PoolView &pool) {
    OI_REQUIRE(oi_cosine_screen_supported(dim), "cosine screen (copy): dim %u not instantiated (384, 768)", dim);
    oi_cosine_screen_geometry(ctx, row_end > row_begin ? row_end - row_begin : 0, &pool.n_segs, &pool.seg_cap);
    OI_REQUIRE(pool.n_segs <= pool.seg_cnt_stride && pool.carry_cap + (uint64_t)pool.n_segs * pool.seg_cap <= pool.stride,
               "cosine screen (copy): chunk does not fit the candidate pool");
    if (row_end <= row_begin || n_queries == 0) return OI_OK;
    const char *nbuf_s = oi_ablation_env("OI_COPY_NBUF"); // (read per call: a sweep tool changes it between runs of one process)
    const int nbuf = nbuf_s ? atoi(nbuf_s) : OI_COPY_NBUF_DEFAULT;
    ProfScope ps(ctx, "cosine");
    for (uint32_t q0 = 0; q0 < n_queries; q0 += 64) {
        const uint32_t nq_here = std::min(64u, n_queries - q0);
        PoolView p = pool;
        p.keys += (uint64_t)q0 * pool.stride;
        p.carry_cnt += q0;
        p.seg_cnt += (uint64_t)q0 * pool.seg_cnt_stride;
        p.tau_keys += q0;
        const uint16_t *qptr = q_bf16 + (uint64_t)q0 * dim;
        const bool two = nq_here > 32;
        if (dim == 768) {
            if (two) OI_CHECK((launch_copy_screen_depth<768, 2>(ctx, nbuf, copy_rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)));
            else OI_CHECK((launch_copy_screen_depth<768, 1>(ctx, nbuf, copy_rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)));
        } else {
            if (two) OI_CHECK((launch_copy_screen_depth<384, 2>(ctx, nbuf, copy_rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)));
            else OI_CHECK((launch_copy_screen_depth<384, 1>(ctx, nbuf, copy_rows, row_begin, row_end, qptr, nq_here, doc_id_base, p)));
        }
    }
    return OI_OK;
}
