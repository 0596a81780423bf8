// oi_device.h -- device-side helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#define OI_WAVE 64

// Cache policy of the corpus stream (LDS-DMA `buffer_load ... lds` in the cosine kernels): the rows are read once per
// batch and are far larger than every cache, so they are loaded NON-TEMPORAL (MI355X_MICROARCH.md, row nt-weights:
// "set nt on streamed bytes that one CU reads once").  Measured on the screen kernel at 10M x 768: 5.2-5.4 -> 4.7 ms per
// batch, 72-74 % -> 82 % of the HBM spec.  -DOI_NO_NT (the ablation build) restores the default policy for A/B runs.
#if defined(OI_NO_NT)
#define OI_DMA_NT ""
#elif defined(OI_DMA_VARIANT) // A/B builds of the other cache-policy bit combinations (tools/build_ablation.sh, OI_ABL_EXTRA)
#if OI_DMA_VARIANT == 1
#define OI_DMA_NT "nt sc1 "
#elif OI_DMA_VARIANT == 2
#define OI_DMA_NT "sc0 nt sc1 "
#elif OI_DMA_VARIANT == 3
#define OI_DMA_NT "sc0 nt "
#elif OI_DMA_VARIANT == 4
#define OI_DMA_NT "sc1 "
#else
#define OI_DMA_NT "sc0 sc1 "
#endif
#else
#define OI_DMA_NT "nt "
#endif
__device__ __forceinline__ uint4 oi_load_stream(const uint4 *p) {
#ifdef OI_NO_NT
    return *p;
#else
    typedef uint32_t oi_u32x4_t __attribute__((ext_vector_type(4)));
    const oi_u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const oi_u32x4_t *>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
#endif
}
__device__ __forceinline__ uint2 oi_load_stream(const uint2 *p) {
#ifdef OI_NO_NT
    return *p;
#else
    typedef uint32_t oi_u32x2_t __attribute__((ext_vector_type(2)));
    const oi_u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const oi_u32x2_t *>(p));
    return make_uint2(v[0], v[1]);
#endif
}
// The same policy for a register load of a once-read stream (the batch-1 GEMV scorer's corpus rows).
__device__ __forceinline__ float4 oi_load_stream(const float4 *p) {
#ifdef OI_NO_NT
    return *p;
#else
    typedef float oi_f32x4_t __attribute__((ext_vector_type(4)));
    const oi_f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const oi_f32x4_t *>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
#endif
}

// A matrix-core kernel must have its CUs to itself.  Round 5 found it the hard way: searches through views of one index on streams
// that really ran at the same time (eight hardware queues) returned, in 2-7 % of the batches, a cosine list with ONE wrong exact
// score -- pf_rescore_kernel's dot product of a (query, row) pair off by 1e-3 .. 1e-2 -- and only when its waves had been scheduled
// on a CU beside a screen workgroup of ANOTHER stream (d = 384: the screen wave takes 304 of its SIMD's 512 registers, so a small
// kernel's wave fits beside it; never at d = 768, 494 registers).  tools/r05_victim_probe.py + tools/r05_victim.hip pin it down
// with a self-checking victim (pf_rescore_kernel's loop over small-integer rows, every sum exact and known) run beside searches:
//   * the victim's LOADS are right (row-element and query-element sums per lane are exact); what is lost is one PRODUCT: the
//     compiler fuses two of the four accumulator chains into v_pk_fma_f32, and the one written  v_pk_fma_f32 ... op_sel:[0,1,0]
//     (both halves take the HIGH half of src1: the .y term) drops its product in the LOW result of lanes 48..63 -- the last
//     quarter of the wave -- 180 to 5 300 wrong sums in 1800 launches; ordinary loads, op_sel_hi-only packed forms, single v_fma_f32: 0;
//   * the neighbour's trait that matters is the MFMA stream, not its LDS-DMA: the screen with its DMA replaced by loads into
//     unused registers still disturbs the victim (70 wrong sums against 204 in the same session), without its matrix instructions
//     it does not (0), and holding M0 longer after a DMA instruction changed nothing; non-temporal loads, the cross-lane sum and
//     the kernel descriptor's register counts were each ruled out on the way.
// A wave executing that packed form on a SIMD where another workgroup's wave issues MFMAs back to back can lose the product.
// Nothing in the ISA text we have allows for it, so it is handled from both sides:
//   (1) every MFMA kernel of this library runs one wave per SIMD and CLAIMS the SIMD's whole register file (v255 and a255 touched:
//       256 + 256 registers allocated), so no other wave -- ours, the runtime's, another library's -- is ever placed beside it
//       (free: these kernels were one wave per SIMD by design; tests/test_gpu_pipeline.py runs three concurrent lanes at d = 384);
//   (2) the exact-score chain of pf_rescore_kernel is written as single v_fma_f32 (oi_fma_unpacked: same fused rounding, bit-identical
//       scores; without the claim and with it alone: 0 mismatching batches of 2700), so it holds beside a FOREIGN matrix kernel too
//       (a host application's GEMM on another stream).  profiles/r05_coresidency_probe.txt lists every run.  It is the only
//       kernel without MFMA in this library that contained an op_sel packed-f32 instruction (ISA survey of every kernel, round 5).
#ifdef OI_NO_CLAIM // (variant builds: with -DOI_PACKED_RESCORE too, tools/r05_lane_race.py shows the corruption again; the victim probe needs this one only)
#define OI_CLAIM_WHOLE_SIMD() do { } while (0)
#else
#define OI_CLAIM_WHOLE_SIMD() asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255")
#endif
__device__ __forceinline__ float oi_fma_unpacked(float x, float y, float a) {
#ifdef OI_PACKED_RESCORE // (variant builds: the compiler's packed form back, for the tools above)
    return fmaf(x, y, a);
#else
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
    return a;
#endif
}

// Order-preserving map f32 -> u32 (ascending).  -0.0 is folded into +0.0 first so
// that equal scores compare equal; NaN must be rejected by the caller.
__device__ __forceinline__ uint32_t oi_f32_key(float s) {
    s += 0.0f;
    uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float oi_key_f32(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(b);
}
// 64-bit rank key: larger = ranks earlier (higher score, then LOWER doc id).
__device__ __forceinline__ uint64_t oi_rank_key(float s, uint32_t doc) {
    return ((uint64_t)oi_f32_key(s) << 32) | (uint64_t)(~doc);
}
__device__ __forceinline__ uint32_t oi_rank_key_doc(uint64_t k) { return ~(uint32_t)k; }
__device__ __forceinline__ float oi_rank_key_score(uint64_t k) { return oi_key_f32((uint32_t)(k >> 32)); }

__device__ __forceinline__ float oi_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, OI_WAVE);
    return v;
}
__device__ __forceinline__ double oi_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, OI_WAVE);
    return v;
}
__device__ __forceinline__ uint32_t oi_wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, OI_WAVE);
    return v;
}

// Append one entry to a candidate pool; entries past `cap` are dropped and flagged.
__device__ __forceinline__ void oi_pool_append(uint64_t *pool, uint32_t *count, uint32_t cap,
                                               uint32_t *overflow, uint64_t key) {
    uint32_t pos = atomicAdd(count, 1u);
    if (pos < cap) pool[pos] = key;
    else *overflow = 1u;
}

// ---- byte classification for the text scans (lexicon.hip, headline.hip) -------------------------------
// 0x80 in every byte of w that is [0-9A-Za-z].  Per range: bit 7 of (x + 0x80 - lo) is x >= lo and bit 7 of
// (x + 0x7F - hi) is x > hi (x < 0x80: no carries); "above hi" implies "at least lo", so their XOR is "in range",
// and the two ranges are disjoint, so the XOR of all four is "in either".  Letters are tested case-folded (|0x20 maps
// A-Z onto a-z and nothing else into that range), digits unfolded (0x10..0x19 fold onto the digits).
__device__ __forceinline__ uint32_t oi_alnum_flags(uint32_t w) {
    const uint32_t w7 = w & 0x7F7F7F7Fu, fold = w7 | 0x20202020u;
    const uint32_t r = (fold + 0x1F1F1F1Fu) ^ (fold + 0x05050505u) ^ (w7 + 0x50505050u) ^ (w7 + 0x46464646u);
    return r & ~w & 0x80808080u;
}
// 16 bytes -> 16 bits (bit i = byte i is alnum); each flagged byte is 128, the dot products weigh byte k by 2^k
__device__ __forceinline__ uint32_t oi_alnum16(uint4 x) {
    const uint32_t lo = __builtin_amdgcn_udot4(oi_alnum_flags(x.y), 0x80402010u,
                                               __builtin_amdgcn_udot4(oi_alnum_flags(x.x), 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(oi_alnum_flags(x.w), 0x80402010u,
                                               __builtin_amdgcn_udot4(oi_alnum_flags(x.z), 0x08040201u, 0u, false), false);
    return (lo >> 7) | (hi << 1);
}
