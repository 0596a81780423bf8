// cosine.hip -- brute-force dot-product scoring of a query batch against corpus rows,
// fused with a threshold filter: only scores >= the per-query running threshold are
// appended to that query's candidate pool, so the N x B score matrix never exists.
//
// Builder-defined (the reference has no embeddings).
//
//   cosine_gemv_filter<NQ>   B <= 8 : HBM-bound.  One wave per row, 16-byte loads,
//                            queries in registers, butterfly reduction.
//   cosine_mfma_filter<NQT>  B  > 8 : exact-f32 MFMA (v_mfma_f32_32x32x2_f32), LDS-tiled.
//                            (v1 tile kernel; the K-split register-resident kernel is in
//                            cosine_ksplit.hip)
#include <cstdlib>

#include "oi_device.h"
#include "oi_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ GEMV (B <= 8)
template <int NQ>
__global__ __launch_bounds__(256) void cosine_gemv_filter(const float *__restrict__ rows, uint64_t row_begin,
                                                           uint64_t row_end, uint32_t dim,
                                                           const float *__restrict__ queries,
                                                           uint32_t doc_id_base, uint64_t *pools,
                                                           uint32_t *seg_cnt, uint32_t cnt_stride,
                                                           const uint32_t *tau_keys, uint64_t pool_stride,
                                                           uint32_t seg_cap, uint32_t *overflow,
                                                           const uint32_t *run_gate) {
    // run_gate != null: a launch of the gated exact pipeline behind the screen (cosine_prefilter.hip) -- exits at once unless open
    if (run_gate && *run_gate == 0u) return;
    // Survivors go to THIS workgroup's segment of the query's pool (LDS fill counter, published once at the
    // end): no global atomic.  With one global counter per query the first chunk -- scored before any
    // threshold exists, every row a survivor -- spent 100 us on 8192 serialised returning atomics.
    __shared__ uint32_t fill[8];
    if (threadIdx.x < 8) fill[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t nvec = dim >> 2; // float4 per row
    // this lane's slice of every query, kept in registers (dim <= 1024 -> <= 4 float4 per query)
    float4 qv[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t v = lane + 64u * c;
            qv[q][c] = v < nvec ? reinterpret_cast<const float4 *>(queries + (uint64_t)q * dim)[v]
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    uint32_t tau = 0;
    if (lane < NQ) tau = tau_keys[lane];

    // Two rows per trip, every load unpredicated (a lane past the row's end re-reads the row's last float4
    // and multiplies it by the zeros its query registers hold there): all of a trip's loads are in flight
    // together instead of each waiting at the end of its own branch.
    uint32_t vidx[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t v = lane + 64u * c;
        vidx[c] = v < nvec ? v : nvec - 1u;
    }
    auto score_row = [&](const float4 (&x)[4], uint64_t r) {
        float mine = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a = fmaf(x[c].x, qv[q][c].x, a);
                a = fmaf(x[c].y, qv[q][c].y, a);
                a = fmaf(x[c].z, qv[q][c].z, a);
                a = fmaf(x[c].w, qv[q][c].w, a);
            }
            a = oi_wave_sum(a);
            if ((int)lane == q) mine = a;
        }
        if (lane < NQ && mine == mine && oi_f32_key(mine) >= tau) {
            const uint32_t pos = atomicAdd(&fill[lane], 1u); // LDS
            if (pos < seg_cap)
                pools[(uint64_t)lane * pool_stride + (uint64_t)blockIdx.x * seg_cap + pos] = oi_rank_key(mine, doc_id_base + (uint32_t)r);
            else *overflow = 1u;
        }
    };
    const uint32_t ncol = (nvec + 63u) >> 6; // float4 columns a lane really needs (3 at dim 768)
    uint64_t r = row_begin + wave;
    for (; r + n_waves < row_end; r += 2 * n_waves) {
        const float4 *row0 = reinterpret_cast<const float4 *>(rows + r * dim);
        const float4 *row1 = reinterpret_cast<const float4 *>(rows + (r + n_waves) * dim);
        float4 x0[4], x1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            x0[c] = x1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((uint32_t)c < ncol) { x0[c] = oi_load_stream(row0 + vidx[c]); x1[c] = oi_load_stream(row1 + vidx[c]); } // (uniform condition)
        }
        score_row(x0, r);
        score_row(x1, r + n_waves);
    }
    if (r < row_end) {
        const float4 *row0 = reinterpret_cast<const float4 *>(rows + r * dim);
        float4 x0[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            x0[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((uint32_t)c < ncol) x0[c] = oi_load_stream(row0 + vidx[c]);
        }
        score_row(x0, r);
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        const uint32_t c = fill[threadIdx.x];
        seg_cnt[(uint64_t)threadIdx.x * cnt_stride + blockIdx.x] = c < seg_cap ? c : seg_cap;
    }
}

// Grid and pool geometry of one GEMV chunk: one segment per workgroup; a workgroup's four waves take rows
// wave, wave + n_waves, ... so each scores at most ceil(n_rows / n_waves) of them.
static uint64_t gemv_blocks(const oi_ctx *ctx, uint64_t n_rows) {
    // ~8 waves per SIMD-quad worth of rows in flight; grid capped at 8 blocks per CU
    uint64_t blocks = (n_rows + 3) / 4;
    const uint64_t cap = (uint64_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    return blocks ? blocks : 1;
}
void oi_cosine_gemv_geometry(const oi_ctx *ctx, uint64_t n_rows, uint32_t *n_segs, uint32_t *seg_cap) {
    const uint64_t blocks = gemv_blocks(ctx, n_rows), n_waves = blocks * 4;
    *n_segs = (uint32_t)blocks;
    *seg_cap = (uint32_t)(4 * ((n_rows + n_waves - 1) / n_waves));
}

// ------------------------------------------------------------------ MFMA tile kernel (v1)
// Workgroup = 4 waves; tile = 128 rows x (32*NQT) queries; K stepped by 32 through LDS.
// Wave w owns rows [32w, 32w+32) of the tile and all query tiles.
// MFMA 32x32x2 operand map: lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31] in accumulator register r.
#define CM_ROWS 128
#define CM_BK 32
#define CM_LD (CM_BK + 1) // +1 float: conflict-free ds_read_b32 down a column

template <int NQT>
__global__ __launch_bounds__(256) void cosine_mfma_filter(const float *__restrict__ rows, uint64_t row_begin,
                                                           uint64_t row_end, uint32_t dim,
                                                           const float *__restrict__ queries, // [32*NQT][dim], zero padded
                                                           uint32_t n_queries, uint32_t doc_id_base,
                                                           uint64_t *pools, uint32_t *pool_counts,
                                                           uint32_t cnt_stride, const uint32_t *tau_keys,
                                                           uint64_t pool_stride, uint32_t pool_cap,
                                                           uint32_t *overflow) {
    __shared__ float sA[CM_ROWS * CM_LD];
    __shared__ float sQ[32 * NQT * CM_LD];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t li = lane & 31, lh = lane >> 5;
    const uint64_t n_rows = row_end - row_begin;
    const uint64_t n_tiles = (n_rows + CM_ROWS - 1) / CM_ROWS;

    uint32_t tau[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const uint32_t q = 32u * t + li;
        tau[t] = q < n_queries ? tau_keys[q] : 0xFFFFFFFFu;
    }

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t r0 = row_begin + tile * CM_ROWS;
        f32x16 acc[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        for (uint32_t k0 = 0; k0 < dim; k0 += CM_BK) {
            __syncthreads();
            // stage A: 128 rows x 32 floats = 1024 float4; thread -> 4 of them
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t f = tid + 256u * i; // float4 index
                const uint32_t rr = f >> 3, cc = (f & 7u) << 2;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r0 + rr < row_end && k0 + cc < dim)
                    v = *reinterpret_cast<const float4 *>(rows + (r0 + rr) * dim + k0 + cc);
                float *d = sA + rr * CM_LD + cc;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            // stage Q: 32*NQT rows x 32 floats
#pragma unroll
            for (int i = 0; i < NQT; ++i) {
                const uint32_t f = tid + 256u * i;
                const uint32_t rr = f >> 3, cc = (f & 7u) << 2;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + cc < dim) v = *reinterpret_cast<const float4 *>(queries + (uint64_t)rr * dim + k0 + cc);
                float *d = sQ + rr * CM_LD + cc;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < CM_BK; kk += 2) {
                const float a = sA[(32u * w + li) * CM_LD + kk + lh];
#pragma unroll
                for (int t = 0; t < NQT; ++t) {
                    const float b = sQ[(32u * t + li) * CM_LD + kk + lh];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
        // epilogue: this lane holds column (query) li of each query tile, 16 rows
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            const uint32_t q = 32u * t + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint64_t row = r0 + 32u * w + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float s = acc[t][r];
                if (row < row_end && s == s && oi_f32_key(s) >= tau[t])
                    oi_pool_append(pools + (uint64_t)q * pool_stride, pool_counts + (uint64_t)q * cnt_stride, pool_cap,
                                   overflow, oi_rank_key(s, doc_id_base + (uint32_t)row));
            }
        }
    }
}

uint64_t oi_cosine_max_chunk_rows(const oi_ctx *ctx, uint32_t dim, uint32_t n_queries, uint64_t stride,
                                  uint32_t carry_cap) {
    // K-split: segments are rounded up to whole tiles per workgroup -> up to 32 * (CUs + 1) slack
    const uint64_t slack = 32ull * ((uint64_t)ctx->num_cus + 1);
    const uint64_t room = stride - carry_cap;
    return room > slack ? room - slack : 0; // api.hip sizes the pool so that room >= min(n, 3*slack) + slack
}

uint32_t oi_cosine_query_padding(uint32_t n_queries) {
    if (n_queries <= 8) return n_queries;
    return (n_queries + 31u) & ~31u;
}

int oi_launch_cosine_chunk(oi_ctx *ctx, const float *rows, uint64_t row_begin, uint64_t row_end,
                           uint32_t dim, const float *d_queries, uint32_t n_queries,
                           uint32_t n_queries_padded, uint32_t doc_id_base, PoolView &pool) {
    static const bool force_v1 = oi_ablation_env("OI_COSINE_V1") != nullptr; // A/B switch for benchmarking
    const bool ksplit = n_queries > 8 && !force_v1 && oi_cosine_ksplit_supported(dim);
    if (ksplit) oi_cosine_ksplit_geometry(ctx, row_end > row_begin ? row_end - row_begin : 0, &pool.n_segs, &pool.seg_cap);
    else if (n_queries <= 8) oi_cosine_gemv_geometry(ctx, row_end > row_begin ? row_end - row_begin : 0, &pool.n_segs, &pool.seg_cap);
    else { // the v1 tile kernel appends with one global counter: one segment spanning the whole appended region
        pool.n_segs = 1;
        pool.seg_cap = (uint32_t)(pool.stride - pool.carry_cap);
    }
    OI_REQUIRE(pool.n_segs <= pool.seg_cnt_stride && pool.carry_cap + (uint64_t)pool.n_segs * pool.seg_cap <= pool.stride,
               "cosine: chunk does not fit the candidate pool");
    if (row_end <= row_begin || n_queries == 0) return OI_OK;
    OI_REQUIRE(dim % 4 == 0 && dim >= 4 && dim <= OI_MAX_DIM, "cosine: dim=%u must be a multiple of 4 in [4,%u]",
               dim, OI_MAX_DIM);
    const uint64_t n_rows = row_end - row_begin;
    ProfScope ps(ctx, ctx->run_gate ? "cosine_gated" : "cosine");
    if (n_queries <= 8) {
        dim3 g((uint32_t)gemv_blocks(ctx, n_rows)), b(256);
#define OI_GEMV(NQ)                                                                                     \
    hipLaunchKernelGGL(cosine_gemv_filter<NQ>, g, b, 0, ctx->stream, rows, row_begin, row_end, dim,     \
                       d_queries, doc_id_base, pool.keys + pool.carry_cap, pool.seg_cnt,                 \
                       pool.seg_cnt_stride, pool.tau_keys, pool.stride, pool.seg_cap, pool.overflow, ctx->run_gate)
        // queries beyond n_queries are not readable: dispatch on the exact count
        switch (n_queries) {
            case 1: OI_GEMV(1); break;
            case 2: OI_GEMV(2); break;
            case 3: OI_GEMV(3); break;
            case 4: OI_GEMV(4); break;
            case 5: OI_GEMV(5); break;
            case 6: OI_GEMV(6); break;
            case 7: OI_GEMV(7); break;
            default: OI_GEMV(8); break;
        }
#undef OI_GEMV
        OI_HIP_CHECK(hipGetLastError());
        return OI_OK;
    }
    OI_REQUIRE(n_queries_padded % 32 == 0 && n_queries_padded >= n_queries, "cosine: bad query padding");
    const uint64_t n_tiles = (n_rows + CM_ROWS - 1) / CM_ROWS;
    uint64_t blocks = n_tiles;
    const uint64_t cap = (uint64_t)ctx->num_cus * 4;
    if (blocks > cap) blocks = cap;
    for (uint32_t q0 = 0; q0 < n_queries_padded; q0 += 64) {
        const uint32_t left = n_queries_padded - q0;
        const uint32_t nq_here = (n_queries - q0) < 64 ? (n_queries - q0) : 64;
        PoolView p = pool;
        p.keys += (uint64_t)q0 * pool.stride;
        p.carry_cnt += q0;
        p.seg_cnt += (uint64_t)q0 * pool.seg_cnt_stride;
        p.tau_keys += q0;
        const float *qptr = d_queries + (uint64_t)q0 * dim;
        if (ksplit && ctx->cosine_mode == OI_COSINE_SPLIT && oi_cosine_split_supported(dim)) {
            OI_CHECK(oi_launch_cosine_split(ctx, rows, row_begin, row_end, dim, qptr, nq_here, doc_id_base, p));
        } else if (ksplit) {
            OI_CHECK(oi_launch_cosine_ksplit(ctx, rows, row_begin, row_end, dim, qptr, nq_here, left >= 64,
                                             doc_id_base, p));
        } else if (left >= 64)
            hipLaunchKernelGGL(cosine_mfma_filter<2>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, rows,
                               row_begin, row_end, dim, qptr, nq_here, doc_id_base, p.keys + p.carry_cap, p.seg_cnt,
                               p.seg_cnt_stride, p.tau_keys, p.stride, p.seg_cap, p.overflow);
        else
            hipLaunchKernelGGL(cosine_mfma_filter<1>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, rows,
                               row_begin, row_end, dim, qptr, nq_here, doc_id_base, p.keys + p.carry_cap, p.seg_cnt,
                               p.seg_cnt_stride, p.tau_keys, p.stride, p.seg_cap, p.overflow);
        OI_HIP_CHECK(hipGetLastError());
        if (q0 + 64 >= n_queries) break;
    }
    return OI_OK;
}

// ------------------------------------------------------------------ row normalisation
__global__ __launch_bounds__(256) void l2_normalize_kernel(float *rows, uint64_t n, uint32_t dim) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n; r += n_waves) {
        float *x = rows + r * dim;
        float ss = 0.f;
        for (uint32_t k = lane; k < dim; k += 64) ss = fmaf(x[k], x[k], ss);
        ss = oi_wave_sum(ss);
        if (ss == 0.f) continue;
        const float inv = 1.0f / sqrtf(ss);
        for (uint32_t k = lane; k < dim; k += 64) x[k] = x[k] * inv;
    }
}

int oi_launch_l2_normalize(oi_ctx *ctx, float *rows, uint64_t n, uint32_t dim) {
    if (n == 0) return OI_OK;
    uint64_t blocks = (n + 3) / 4;
    const uint64_t cap = (uint64_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, rows, n, dim);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}
