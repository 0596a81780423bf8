// tools/r05_victim.hip -- round 5, the co-residency finding (csrc/oi_device.h, OI_CLAIM_WHOLE_SIMD): self-checking VICTIMS.
// tools/r05_victim_probe.py runs them on their own stream beside searches of other streams: which of the library's kernels,
// resident on the same CU, disturbs a small kernel's wave -- and what exactly goes wrong in it.
//   victim_words: small waves (256 threads, few registers, no LDS) read 2-KiB rows of a buffer whose every word is a known function
//                 of its index with ordinary global loads, and count the words that come back wrong.
//   victim_dots:  pf_rescore_kernel's own loop (four rows per wave and trip, float4 loads strided by 64 lanes, an fma chain per
//                 lane, butterfly sum) over small-integer rows and queries, whose dot products are exact in f32 and known
//                 (expect[q][row], computed by the tool).  flavor bit 0: ordinary instead of non-temporal loads; bit 1: the loop
//                 without its ragged last trip (d = 384 is 96 float4: lanes 32..63 sit out the second trip -- every lane loads, the
//                 lanes past the row multiply by zero); bit 2 (value 4): every lane LOADS on every trip (clamped index, the loaded
//                 registers pinned before the branch) and only the fma chain is skipped past the row's end -- the hardened form of
//                 pf_rescore_kernel's loop; value 8: the ragged loop with every fma a single v_fma_f32 (inline asm: the form
//                 pf_rescore_kernel ships with).  Beside the dot product each lane sums its row elements and its query elements on their own:
//                 a wrong sum says whether the row's or the query's words came back wrong.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC -o tools/r05_victim.so tools/r05_victim.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__device__ __forceinline__ uint32_t word_of(uint64_t i) { return (uint32_t)i * 2654435761u + 12345u; }

// err[0] = wrong words, err[1] = log entries used, err[2..] = {index lo, got, want, hw_id} x 64
__global__ __launch_bounds__(256) void victim_words_kernel(const uint32_t *__restrict__ buf, uint64_t n_rows, uint32_t seed,
                                                           uint32_t rows_per_wave, uint32_t *err) {
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t h = seed * 0x9E3779B9u + wave * 0x85EBCA6Bu;
    for (uint32_t it = 0; it < rows_per_wave; ++it) {
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        const uint64_t row = (uint64_t)h % n_rows;
        const uint4 *p = reinterpret_cast<const uint4 *>(buf + row * 512);
        const uint4 a = p[lane], b = p[64 + lane];
        const uint64_t ia = row * 512 + lane * 4, ib = ia + 256;
        const uint32_t got[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t i = (k < 4 ? ia : ib) + (k & 3);
            if (got[k] != word_of(i)) {
                atomicAdd(&err[0], 1u);
                const uint32_t s = atomicAdd(&err[1], 1u);
                if (s < 64) {
                    uint32_t hw;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                    err[2 + 4 * s + 0] = (uint32_t)i; err[2 + 4 * s + 1] = got[k]; err[2 + 4 * s + 2] = word_of(i); err[2 + 4 * s + 3] = hw;
                }
            }
        }
    }
}

extern "C" int victim_launch(void *stream, const uint32_t *buf, uint64_t n_rows, uint32_t seed, uint32_t n_blocks,
                             uint32_t rows_per_wave, uint32_t *err) {
    hipLaunchKernelGGL(victim_words_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, buf, n_rows, seed, rows_per_wave, err);
    return (int)hipGetLastError();
}

__device__ __forceinline__ float4 ld_nt(const float4 *p) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// err[0] = wrong sums, err[1] = log entries used; log entry s (s < 8) at flog + s * 200: {q, row, got, want, hw_id, u, trip, 0,
//          partial dot[64], partial row-element sum[64], partial query-element sum[64]}
template <int FLAVOR>
__device__ __forceinline__ float vfma(float x, float y, float a) {
    if constexpr ((FLAVOR & 8) != 0) {
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
        return a;
    } else {
        return fmaf(x, y, a);
    }
}
template <int FLAVOR>
__global__ __launch_bounds__(256) void victim_dots_kernel(const float *__restrict__ rows, uint32_t dim, uint32_t n_rows,
                                                          const float *__restrict__ queries, const float *__restrict__ expect,
                                                          uint32_t seed, uint32_t trips, uint32_t *err, float *flog) {
    const uint32_t q = blockIdx.y, lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nvec = dim >> 2;
    const float4 *qv = reinterpret_cast<const float4 *>(queries + (uint64_t)q * dim);
    uint32_t h = seed * 0x9E3779B9u + wave * 0x85EBCA6Bu + q * 0xC2B2AE35u;
    for (uint32_t t = 0; t < trips; ++t) {
        uint32_t r[4];
        const float4 *x[4];
        float a[4], ax[4], ay = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ax[u] = 0.f;
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
            r[u] = __builtin_amdgcn_readfirstlane(h) % n_rows;
            x[u] = reinterpret_cast<const float4 *>(rows + (uint64_t)r[u] * dim);
            a[u] = 0.f;
        }
        if (FLAVOR & 2) {
            const uint32_t nv64 = (nvec + 63) & ~63u;
            for (uint32_t v = lane; v < nv64; v += 64) {
                const uint32_t vc = v < nvec ? v : nvec - 1;
                const float keep = v < nvec ? 1.f : 0.f;
                float4 yv = qv[vc];
                yv.x *= keep; yv.y *= keep; yv.z *= keep; yv.w *= keep;
                float4 xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = (FLAVOR & 1) ? x[u][vc] : ld_nt(x[u] + vc);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = fmaf(xv[u].x, yv.x, a[u]); a[u] = fmaf(xv[u].y, yv.y, a[u]);
                    a[u] = fmaf(xv[u].z, yv.z, a[u]); a[u] = fmaf(xv[u].w, yv.w, a[u]);
                }
            }
        } else if (FLAVOR & 4) {
            const uint32_t nv64 = (nvec + 63) & ~63u;
            for (uint32_t v = lane; v < nv64; v += 64) {
                const uint32_t vc = v < nvec ? v : nvec - 1;
                float4 yv = qv[vc];
                float4 xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = (FLAVOR & 1) ? x[u][vc] : ld_nt(x[u] + vc);
                asm volatile("" : "+v"(yv.x), "+v"(yv.y), "+v"(yv.z), "+v"(yv.w));
#pragma unroll
                for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(xv[u].x), "+v"(xv[u].y), "+v"(xv[u].z), "+v"(xv[u].w));
                if (v < nvec) {
                    ay += (yv.x + yv.y) + (yv.z + yv.w);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a[u] = fmaf(xv[u].x, yv.x, a[u]); a[u] = fmaf(xv[u].y, yv.y, a[u]);
                        a[u] = fmaf(xv[u].z, yv.z, a[u]); a[u] = fmaf(xv[u].w, yv.w, a[u]);
                        ax[u] += (xv[u].x + xv[u].y) + (xv[u].z + xv[u].w);
                    }
                }
            }
        } else {
            for (uint32_t v = lane; v < nvec; v += 64) {
                const float4 yv = qv[v];
                float4 xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = (FLAVOR & 1) ? x[u][v] : ld_nt(x[u] + v);
                ay += (yv.x + yv.y) + (yv.z + yv.w);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = vfma<FLAVOR>(xv[u].x, yv.x, a[u]); a[u] = vfma<FLAVOR>(xv[u].y, yv.y, a[u]);
                    a[u] = vfma<FLAVOR>(xv[u].z, yv.z, a[u]); a[u] = vfma<FLAVOR>(xv[u].w, yv.w, a[u]);
                    ax[u] += (xv[u].x + xv[u].y) + (xv[u].z + xv[u].w);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s = wave_sum(a[u]);
            const float want = expect[(uint64_t)q * n_rows + r[u]];
            if (s != want) {                                       // (uniform: every lane holds the same s)
                uint32_t slot = 0;
                if (lane == 0) {
                    atomicAdd(&err[0], 1u);
                    slot = atomicAdd(&err[1], 1u);
                }
                slot = __builtin_amdgcn_readfirstlane(slot);
                if (slot < 8) {
                    float *e = flog + slot * 200;
                    if (lane == 0) {
                        uint32_t hw;
                        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                        e[0] = (float)q; e[1] = (float)r[u]; e[2] = s; e[3] = want; e[4] = __uint_as_float(hw); e[5] = (float)u; e[6] = (float)t;
                    }
                    e[8 + lane] = a[u]; e[72 + lane] = ax[u]; e[136 + lane] = ay;
                }
            }
        }
    }
}

extern "C" int victim_dots_launch(void *stream, int flavor, const float *rows, uint32_t dim, uint32_t n_rows, const float *queries,
                                  uint32_t n_queries, const float *expect, uint32_t seed, uint32_t blocks_x, uint32_t trips,
                                  uint32_t *err, float *flog) {
    const dim3 g(blocks_x, n_queries), b(256);
    hipStream_t st = (hipStream_t)stream;
    if (flavor & 8) {
        hipLaunchKernelGGL(victim_dots_kernel<8>, g, b, 0, st, rows, dim, n_rows, queries, expect, seed, trips, err, flog);
        return (int)hipGetLastError();
    }
    if (flavor & 4) {
        if (flavor & 1) hipLaunchKernelGGL(victim_dots_kernel<5>, g, b, 0, st, rows, dim, n_rows, queries, expect, seed, trips, err, flog);
        else hipLaunchKernelGGL(victim_dots_kernel<4>, g, b, 0, st, rows, dim, n_rows, queries, expect, seed, trips, err, flog);
        return (int)hipGetLastError();
    }
    switch (flavor & 3) {
    case 0: hipLaunchKernelGGL(victim_dots_kernel<0>, g, b, 0, st, rows, dim, n_rows, queries, expect, seed, trips, err, flog); break;
    case 1: hipLaunchKernelGGL(victim_dots_kernel<1>, g, b, 0, st, rows, dim, n_rows, queries, expect, seed, trips, err, flog); break;
    case 2: hipLaunchKernelGGL(victim_dots_kernel<2>, g, b, 0, st, rows, dim, n_rows, queries, expect, seed, trips, err, flog); break;
    default: hipLaunchKernelGGL(victim_dots_kernel<3>, g, b, 0, st, rows, dim, n_rows, queries, expect, seed, trips, err, flog); break;
    }
    return (int)hipGetLastError();
}
