// select.hip -- candidate-pool top-k selection, list merge and reciprocal-rank fusion.
//
// Every ranked list in this library is a set of 64-bit rank keys
//   key = orderable(score) << 32 | ~doc_id        (oi_device.h)
// so "score descending, doc id ascending" is plain descending u64 order and keys of
// distinct docs are distinct.  One workgroup (1024 threads = 16 waves) owns one query.
#include "oi_device.h"
#include "oi_internal.h"

#define SEL_THREADS 1024
#define SEL_MAX 1024  // == OI_MAX_DEPTH

// In-LDS bitonic sort, descending, P a power of two <= 2048, all SEL_THREADS threads call.
__device__ void bitonic_sort_desc(uint64_t *a, uint32_t P) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < (P >> 1); t += SEL_THREADS) {
                uint32_t lo = 2 * t - (t & (stride - 1));
                uint32_t hi = lo + stride;
                bool desc = ((lo & size) == 0);
                uint64_t x = a[lo], y = a[hi];
                if ((x < y) == desc) { a[lo] = y; a[hi] = x; }
            }
            __syncthreads();
        }
    }
}

// Top-k of pool q (carry region + segments, see PoolView).  Small pools are gathered into LDS
// once; the radix select (8 passes of 8 bits from the top, stopping as soon as the k-th key is
// alone in its bin) then runs out of LDS.  Large pools are scanned in place, one wave per segment.
// compact: the pool is rewritten as carry = the selected keys, all segments empty, and the
// threshold raised to the k-th score, so the next corpus chunk appends after them.
#define SEL_LDS_KEYS 8192

struct SelSource {
    const uint64_t *pool;      // this query's pool
    const uint32_t *segc;      // this query's segment counts
    const uint64_t *lds_keys;  // non-null: everything already gathered here
    uint32_t n, c0, carry_cap, seg_cap, n_segs;
};

template <class F>
__device__ __forceinline__ void sel_for_each(const SelSource &S, F &&f) {
    const uint32_t tid = threadIdx.x;
    if (S.lds_keys) {
        for (uint32_t i = tid; i < S.n; i += SEL_THREADS) f(S.lds_keys[i]);
        return;
    }
    for (uint32_t i = tid; i < S.c0; i += SEL_THREADS) f(S.pool[i]);
    const uint32_t lane = tid & 63, wv = tid >> 6;
    for (uint32_t sg = wv; sg < S.n_segs; sg += SEL_THREADS / 64) {
        uint32_t c = S.segc[sg];
        if (c > S.seg_cap) c = S.seg_cap;
        const uint64_t *base = S.pool + S.carry_cap + (uint64_t)sg * S.seg_cap;
        for (uint32_t i = lane; i < c; i += 64) f(base[i]);
    }
}

__global__ __launch_bounds__(SEL_THREADS) void select_topk_kernel(
    uint64_t *pools, uint32_t *carry_cnt, uint32_t *seg_cnt, uint32_t *tau_keys, uint64_t pool_stride,
    uint32_t carry_cap, uint32_t seg_cap, uint32_t n_segs, uint32_t seg_cnt_stride, uint32_t *overflow,
    uint32_t k, int compact, float *out_scores, uint32_t *out_docs, uint32_t *out_counts, uint32_t out_stride) {
    __shared__ uint32_t hist[256];
    __shared__ uint64_t sel[2 * SEL_MAX];
    __shared__ uint64_t lds_keys[SEL_LDS_KEYS];
    __shared__ uint32_t s_cnt, s_kk, s_bin_cnt, s_total;
    __shared__ uint64_t s_prefix;

    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    uint64_t *pool = pools + (uint64_t)q * pool_stride;
    uint32_t *segc = seg_cnt + (uint64_t)q * seg_cnt_stride;
    SelSource S;
    S.pool = pool; S.segc = segc; S.lds_keys = nullptr;
    S.carry_cap = carry_cap; S.seg_cap = seg_cap; S.n_segs = n_segs;
    S.c0 = carry_cnt[q] < carry_cap ? carry_cnt[q] : carry_cap;
    if (tid == 0) { s_cnt = 0; s_total = 0; }
    __syncthreads();
    {
        uint32_t local = 0;
        for (uint32_t sg = tid; sg < n_segs; sg += SEL_THREADS) {
            uint32_t c = segc[sg];
            if (c > seg_cap) { *overflow = 1u; c = seg_cap; }
            local += c;
        }
        local = oi_wave_sum(local);
        if ((tid & 63) == 0 && local) atomicAdd(&s_total, local);
    }
    __syncthreads();
    const uint32_t n = S.c0 + s_total;
    S.n = n;
    if (n <= SEL_LDS_KEYS) {
        sel_for_each(S, [&](uint64_t key) { lds_keys[atomicAdd(&s_cnt, 1u)] = key; });
        __syncthreads();
        S.lds_keys = lds_keys;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
    }

    uint32_t m; // number selected
    if (n <= k) {
        sel_for_each(S, [&](uint64_t key) { sel[atomicAdd(&s_cnt, 1u)] = key; });
        m = n;
        __syncthreads();
    } else {
        if (tid == 0) { s_prefix = 0; s_kk = k; }
        int shift = 56;
        for (;; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const uint64_t prefix = s_prefix;
            if (shift == 56)
                sel_for_each(S, [&](uint64_t key) { atomicAdd(&hist[(uint32_t)(key >> 56)], 1u); });
            else
                sel_for_each(S, [&](uint64_t key) {
                    if ((key >> (shift + 8)) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
                });
            __syncthreads();
            if (tid < 64) { // wave 0: find the digit holding the kk-th key counted from the top
                const uint32_t kk = s_kk;
                uint32_t mine = 0;
                for (int i = 0; i < 4; ++i) mine += hist[255 - (tid * 4 + i)];
                uint32_t incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    uint32_t v = __shfl_up(incl, o, OI_WAVE);
                    if ((int)tid >= o) incl += v;
                }
                const unsigned long long ball = __ballot(incl >= kk);
                const uint32_t owner = ball ? (uint32_t)__builtin_ctzll(ball) : 63u;
                if (tid == owner) {
                    uint32_t cum = incl - mine;
                    int d = 255 - (int)(tid * 4);
                    for (int i = 0; i < 3; ++i, --d) {
                        const uint32_t c = hist[d];
                        if (cum + c >= kk) break;
                        cum += c;
                    }
                    s_prefix = (prefix << 8) | (uint64_t)d;
                    s_kk = kk - cum;
                    s_bin_cnt = hist[d];
                }
            }
            __syncthreads();
            if (s_bin_cnt == 1 || shift == 0) break;
        }
        // every key whose top bits are >= prefix is selected: exactly k of them (keys are distinct)
        const uint64_t prefix = s_prefix;
        sel_for_each(S, [&](uint64_t key) {
            if ((key >> shift) >= prefix) {
                const uint32_t pos = atomicAdd(&s_cnt, 1u);
                if (pos < SEL_MAX) sel[pos] = key;
            }
        });
        __syncthreads();
        m = s_cnt < k ? s_cnt : k;
    }
    // Sorted output is needed only for a final list; an intermediate compaction just needs the set
    // and its smallest key (the new threshold).
    __shared__ unsigned long long s_min;
    if (out_scores) {
        uint32_t P = 2;
        while (P < m) P <<= 1;
        for (uint32_t i = m + tid; i < P; i += SEL_THREADS) sel[i] = 0; // lowest possible key
        __syncthreads();
        bitonic_sort_desc(sel, P);
        for (uint32_t i = tid; i < m; i += SEL_THREADS) {
            const uint64_t key = sel[i];
            out_scores[(uint64_t)q * out_stride + i] = oi_rank_key_score(key);
            out_docs[(uint64_t)q * out_stride + i] = oi_rank_key_doc(key);
        }
        if (tid == 0) { out_counts[q] = m; s_min = m ? sel[m - 1] : 0; }
    } else {
        if (tid == 0) s_min = ~0ull;
        __syncthreads();
        unsigned long long lo = ~0ull;
        for (uint32_t i = tid; i < m; i += SEL_THREADS) lo = sel[i] < lo ? sel[i] : lo;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long v = __shfl_xor(lo, o, OI_WAVE);
            lo = v < lo ? v : lo;
        }
        if ((tid & 63) == 0 && lo != ~0ull) atomicMin(&s_min, lo);
    }
    __syncthreads();
    if (compact) {
        for (uint32_t i = tid; i < m; i += SEL_THREADS) pool[i] = sel[i];
        for (uint32_t sg = tid; sg < n_segs; sg += SEL_THREADS) segc[sg] = 0;
        if (tid == 0) {
            carry_cnt[q] = m;
            if (m == k && tau_keys) {
                // k docs at or above this score exist: a valid lower bound for the final
                // k-th score, so later chunks may drop anything strictly below it.
                const uint32_t t = (uint32_t)(s_min >> 32);
                if (t > tau_keys[q]) tau_keys[q] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// select_flat_kernel: the same selection with the pool seen as ONE flat array.
//
// The kernel above walks the pool a segment at a time (one wave per segment: a dependent count -> keys
// load chain per segment, repeated on every radix pass when the pool does not fit its LDS copy); with the
// 256 short segments a cosine chunk leaves behind, that chain IS the select (40-120 us per launch, on the
// critical path between two corpus chunks).  Here
//   * the segment counts are scanned once into LDS offsets; flat index -> (segment, slot) is one binary
//     search per lane and then a forward walk, because wave w owns the contiguous flat range
//     [w*kpt*64, (w+1)*kpt*64) and reads it 64 consecutive keys per load (coalesced, all loads in flight);
//   * the keys then live in REGISTERS (<= 32 per thread: pools up to 32K keys) for everything that follows;
//     larger pools re-load through the same flat view on every pass;
//   * pools above 4K keys are cut down first: the threshold that keeps ~2k keys is read off a 1024-key
//     sample (radix select over one key per thread), one filter pass moves the survivors to LDS, and the
//     exact selection runs over those.  A sample that keeps fewer than k or more than 4096 keys falls back
//     to the exact passes over the whole pool: the RESULT never depends on the sample;
//   * histogram and append atomics are aggregated per wave first (the leading digits of a pool of
//     near-threshold scores are nearly all equal, i.e. one LDS address).
#define SEL_MAX_SEGS 4096
#define SEL_KPT_MAX 32
#ifdef OI_ABLATION
// OI_SELECT_STAMPS=1 (ablation builds): query 0's thread 0 prints where a select_flat launch spends its cycles
__device__ int sel_dbg_on;
__device__ unsigned long long sel_dbg_t[16];
#define SEL_STAMP(i) do { if (sel_dbg_on && blockIdx.x == 0 && threadIdx.x == 0) sel_dbg_t[i] = __builtin_readcyclecounter(); } while (0)
#else
#define SEL_STAMP(i) do { } while (0)
#endif
#define SEL_CAND 4096

struct SelShared {
    uint32_t hist[256];
    uint32_t hist2k[2048]; // the margin selects' 11-bit digits (sel_flat_select, fast margin path)
    uint32_t seg_off[SEL_MAX_SEGS + 1];
    uint32_t wave_tot[SEL_THREADS / 64];
    uint32_t cnt, kk, bin_cnt, n_samples;
    uint64_t prefix;
    unsigned long long min_key;
};

struct SelFlat {
    const uint64_t *pool;
    const uint32_t *seg_off; // LDS, n_segs + 1 entries, seg_off[n_segs] = keys in segments
    uint32_t n, c0, carry_cap, seg_cap, n_segs, steps;
};
// A lane's position in the flat view: segment sg holds the segment-relative flat indices [lo, hi).
struct SelCursor {
    uint32_t sg, lo, hi;
};
// flat index i < n  ->  cursor at its segment (binary search; indices inside the carry region map to e = 0)
__device__ __forceinline__ SelCursor sel_seek(const SelFlat &K, uint32_t i) {
    const uint32_t e = i < K.c0 ? 0u : i - K.c0;
    uint32_t lo = 0, hi = K.n_segs; // seg_off[lo] <= e < seg_off[hi]
    for (uint32_t s = 0; s < K.steps; ++s) {
        const uint32_t mid = (lo + hi) >> 1;
        const bool up = K.seg_off[mid] <= e;
        lo = up ? mid : lo;
        hi = up ? hi : mid;
    }
    SelCursor c;
    c.sg = lo; c.lo = K.seg_off[lo]; c.hi = K.seg_off[lo + 1];
    return c;
}
// pool offset of flat index i < n, i >= every index this cursor was used for before (forward walk)
__device__ __forceinline__ uint64_t sel_at(const SelFlat &K, SelCursor &c, uint32_t i) {
    if (i < K.c0) return i;
    const uint32_t e = i - K.c0;
    while (e >= c.hi) { // e < seg_off[n_segs]: stops at the segment that holds e (skips empty ones)
        ++c.sg;
        c.lo = c.hi;
        c.hi = K.seg_off[c.sg + 1];
    }
    return (uint64_t)K.carry_cap + (uint64_t)c.sg * K.seg_cap + (e - c.lo);
}

// hist[digit] += 1 for the active lanes; the two most common digits of the wave go in as one atomic each
__device__ __forceinline__ void sel_hist_add(uint32_t *hist, bool active, uint32_t digit) {
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const unsigned long long m = __ballot(active);
        if (!m) return;
        const uint32_t leader = (uint32_t)__builtin_ctzll(m);
        const uint32_t d0 = (uint32_t)__shfl((int)digit, (int)leader, OI_WAVE);
        const unsigned long long same = __ballot(active && digit == d0);
        if (lane == leader) atomicAdd(&hist[d0], (uint32_t)__popcll(same));
        active = active && digit != d0;
    }
    if (active) atomicAdd(&hist[digit], 1u);
}

// wave-aggregated append of the lanes' keys to dst; *count counts every taken key, stored or not
__device__ __forceinline__ void sel_append(uint64_t *dst, uint32_t *count, uint32_t cap, bool take, uint64_t key) {
    const unsigned long long m = __ballot(take);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63, leader = (uint32_t)__builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)leader, OI_WAVE);
    const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (take && pos < cap) dst[pos] = key;
}

// wave-wide inclusive prefix sum in DPP steps (no LDS round trips, unlike __shfl_up)
__device__ __forceinline__ uint32_t sel_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return v;
}
// hist[digit] += weight for the active lanes; the two most common digits of the wave go in as one atomic each
__device__ __forceinline__ void sel_hist_add_w(uint32_t *hist, bool active, uint32_t digit, uint32_t weight) {
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const unsigned long long m = __ballot(active);
        if (!m) return;
        const uint32_t leader = (uint32_t)__builtin_ctzll(m);
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)digit, (int)leader);
        const bool same = active && digit == d0;
        const uint32_t sum = (uint32_t)__builtin_amdgcn_readlane((int)sel_incl_scan(same ? weight : 0u), 63);
        if (lane == leader) atomicAdd(&hist[d0], sum);
        active = active && !same;
    }
    if (active) atomicAdd(&hist[digit], weight);
}

// Radix select over the keys `for_each` enumerates (f(valid, key), same keys on every call): finds
// (shift, prefix) such that exactly kk of them have (key >> shift) >= prefix.  Needs kk <= their number,
// distinct keys; all threads call.
template <class FE>
__device__ __forceinline__ void sel_threshold(FE &&for_each, uint32_t kk_in, SelShared &sh, int &shift_out, uint64_t &prefix_out) {
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { sh.prefix = 0; sh.kk = kk_in; }
    int shift = 56;
    for (;; shift -= 8) {
        if (tid < 256) sh.hist[tid] = 0;
        __syncthreads();
        const uint64_t prefix = sh.prefix;
        const bool top = shift == 56;
        for_each([&](bool valid, uint64_t kv) {
            sel_hist_add(sh.hist, valid && (top || (kv >> ((shift + 8) & 63)) == prefix), (uint32_t)(kv >> shift) & 255u);
        });
        __syncthreads();
        if (tid < 64) { // wave 0: the digit holding the kk-th key counted from the top
            const uint32_t kk = sh.kk;
            uint32_t mine = 0;
            for (int i = 0; i < 4; ++i) mine += sh.hist[255 - (tid * 4 + i)];
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o, OI_WAVE);
                if ((int)tid >= o) incl += v;
            }
            const unsigned long long ball = __ballot(incl >= kk);
            const uint32_t owner = ball ? (uint32_t)__builtin_ctzll(ball) : 63u;
            if (tid == owner) {
                uint32_t cum = incl - mine;
                int d = 255 - (int)(tid * 4);
                for (int i = 0; i < 3; ++i, --d) {
                    const uint32_t c = sh.hist[d];
                    if (cum + c >= kk) break;
                    cum += c;
                }
                sh.prefix = (prefix << 8) | (uint64_t)d;
                sh.kk = kk - cum;
                sh.bin_cnt = sh.hist[d];
            }
        }
        __syncthreads();
        if (sh.bin_cnt == 1 || shift == 0) break;
    }
    shift_out = shift;
    prefix_out = sh.prefix;
    __syncthreads(); // sh.prefix / sh.kk may be rewritten by the next call
}

// Selects the top min(n, k) keys of K into sel[], returns their number.  KPT > 0: the keys live in registers
// (n <= KPT * SEL_THREADS); KPT == 0: every pass re-loads them through the flat view.  sh.cnt == 0 on entry.
//
// Margin mode (eps2 >= 0; the bf16 screen of cosine_prefilter.hip): the result is not the top k but EVERY key whose
// score is within eps2 of the k-th largest score -- the set that must survive for the exact top k to be inside
// it -- left in cand[] (*in_cand = true), at most SEL_CAND keys (more: *margin_overflow = true, the caller opens
// the exact pipeline), and *margin_tau = the orderable key of (k-th score - eps2), the next chunk's threshold.
template <int KPT>
__device__ __forceinline__ uint32_t sel_flat_select(const SelFlat &K, uint32_t k, SelShared &sh, uint64_t *sel,
                                                    uint64_t *cand, float eps2, bool *in_cand, uint32_t *margin_tau,
                                                    bool *margin_overflow, const uint32_t *skip, uint32_t skip_base) {
    // skip (the screen's two-class margin): keys of docs marked in this bitmap do not take part -- they are scored exactly
    // whatever happens here, and their screen scores must not move the threshold (cosine_prefilter.hip)
    auto skipped = [&](uint64_t kv) -> bool {
        const uint32_t d = oi_rank_key_doc(kv) - skip_base;
        return (skip[d >> 5] >> (d & 31u)) & 1u;
    };
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = K.n;
    *in_cand = false;
    *margin_overflow = false;
    if (n == 0) return 0;
    const uint32_t kpt = (n + SEL_THREADS - 1) / SEL_THREADS; // keys per thread; wave w owns [w*kpt*64, (w+1)*kpt*64)
    const uint32_t first = wv * kpt * 64 + lane, last = n - 1;
    constexpr int NR = KPT > 0 ? KPT : 1;
    uint64_t key[NR];
    uint32_t dead = 0; // bit j: key[j] is a skipped doc's
    if constexpr (KPT > 0) {
        SelCursor cur = sel_seek(K, first < n ? first : last);
        uint64_t at = sel_at(K, cur, first < n ? first : last);
#pragma unroll
        for (int j = 0; j < KPT; ++j) { // loads unpredicated (a slot past the end re-reads the previous key): all in flight
            const uint32_t i = first + (uint32_t)j * 64;
            if ((uint32_t)j < kpt && i < n) at = sel_at(K, cur, i); // the cursor only walks forward
            key[j] = K.pool[at];
        }
        if (skip) {
#pragma unroll
            for (int j = 0; j < KPT; ++j)
                if ((uint32_t)j < kpt && first + (uint32_t)j * 64 < n && skipped(key[j])) dead |= 1u << j;
        }
    }
    auto for_each = [&](auto &&f) {
        if constexpr (KPT > 0) {
#pragma unroll
            for (int j = 0; j < KPT; ++j)
                if ((uint32_t)j < kpt) f(first + (uint32_t)j * 64 < n && !((dead >> j) & 1u), key[j]); // uniform guard
        } else {
            SelCursor cur = sel_seek(K, first < n ? first : last);
            uint64_t at = sel_at(K, cur, first < n ? first : last);
            for (uint32_t j0 = 0; j0 < kpt; j0 += 4) { // four loads in flight per thread
                uint64_t kx[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t i = first + (j0 + u) * 64;
                    ok[u] = j0 + u < kpt && i < n;
                    if (ok[u]) at = sel_at(K, cur, i);
                    kx[u] = K.pool[at];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) f(ok[u] && !(skip && skipped(kx[u])), kx[u]);
            }
        }
    };
    if (n <= k) {
        for_each([&](bool valid, uint64_t kv) { sel_append(sel, &sh.cnt, 2 * SEL_MAX, valid, kv); });
        __syncthreads();
        return skip ? sh.cnt : n; // (skipped keys were not appended)
    }
    SEL_STAMP(2);
    if (eps2 >= 0.f) {
        // ---- margin mode, the fast path (round 4).  What the next chunk and the rescoring need is ANY lower bound tau' of the
        // k-th largest screen score and every key within eps2 of tau' -- not the k-th key itself.  Two passes of 11-bit digits
        // over the valid keys find the 22-bit bin (sign, exponent, 13 bits of mantissa) that holds the k-th largest; its LOWER
        // EDGE is tau': at least k keys are >= it, and it is below the k-th score by < 2^-13 of its size, ~1e-5 where the margin
        // is ~4e-3, so the survivor set grows by a fraction of a percent.  (The exact path below: a 1024-key sample, 4-7 passes of
        // 8-bit digits over the sample, a cut, 4-7 more over the cut, the k-th key, then the margin filter -- 25-30 us per launch
        // between two corpus chunks; this one: two histogram passes and the filter.)  Fewer than k valid keys: the exact path.
        uint32_t kk = k;
        uint32_t pfx = 0;
        bool enough = true;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int shift = 53 - 11 * pass;
            sh.hist2k[tid] = 0;
            sh.hist2k[tid + SEL_THREADS] = 0;
            __syncthreads();
            if (pass == 0) {
                // the leading digit (sign, exponent, two bits of mantissa) of a pool of near-threshold scores takes a handful of
                // values: a thread adds its keys up in runs first, and the wave its threads' last runs (one LDS address each)
                uint32_t run_d = 0, run_c = 0;
                for_each([&](bool valid, uint64_t kv) {
                    if (valid) {
                        const uint32_t d = (uint32_t)(kv >> 53);
                        if (run_c && d != run_d) { atomicAdd(&sh.hist2k[run_d], run_c); run_c = 0; }
                        run_d = d;
                        ++run_c;
                    }
                });
                sel_hist_add_w(sh.hist2k, run_c != 0u, run_d, run_c);
            } else { // 11 bits of mantissa inside one leading bin: spread over the 2048 addresses, plain atomics
                for_each([&](bool valid, uint64_t kv) {
                    if (valid && (uint32_t)(kv >> 53) == pfx) atomicAdd(&sh.hist2k[(uint32_t)(kv >> shift) & 2047u], 1u);
                });
            }
            __syncthreads();
            { // 64 super-bins of 32 bins: a thread adds two bins, a DPP row (16 lanes) the 32 of a super-bin -- conflict-free reads
                uint32_t v = sh.hist2k[2 * tid] + sh.hist2k[2 * tid + 1];
                v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
                v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
                v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
                v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
                if ((tid & 15u) == 15u) sh.hist[tid >> 4] = v; // super-bin s = bins [32 s, 32 s + 32)
            }
            __syncthreads();
            if (tid < 64) { // wave 0: the super-bin, then the bin, holding the kk-th key counted from the top -- two scans, no walk
                const uint32_t x = sh.hist[63u - tid];
                const uint32_t incl = sel_incl_scan(x);
                const unsigned long long ball = __ballot(incl >= kk);
                if (!ball) {
                    if (tid == 0) sh.bin_cnt = 0xFFFFFFFFu; // fewer than kk valid keys
                } else {
                    const uint32_t l1 = (uint32_t)__builtin_ctzll(ball), sb = 63u - l1;
                    const uint32_t kk2 = kk - (uint32_t)__builtin_amdgcn_readlane((int)(incl - x), (int)l1);
                    const uint32_t y = tid < 32u ? sh.hist2k[32u * sb + 31u - tid] : 0u;
                    const uint32_t incl2 = sel_incl_scan(y);
                    const unsigned long long ball2 = __ballot(incl2 >= kk2); // (non-empty: the super-bin holds >= kk2 keys)
                    const uint32_t l2 = ball2 ? (uint32_t)__builtin_ctzll(ball2) : 31u;
                    // (read with every lane active: inside the one-lane branch below hipcc computes incl2 - y for that lane only)
                    const uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)(incl2 - y), (int)l2);
                    if (tid == 0) {
                        sh.prefix = (uint64_t)(32u * sb + 31u - l2);
                        sh.kk = kk2 - above;
                        sh.bin_cnt = 1u;
#ifdef OI_ABLATION
                        if (sel_dbg_on > 1) { // OI_SELECT_STAMPS=2: self-check against the plain walk from the top
                            uint32_t cum = 0; int dd = 2047;
                            for (; dd >= 0; --dd) { if (cum + sh.hist2k[dd] >= kk) break; cum += sh.hist2k[dd]; }
                            if ((uint32_t)dd != (uint32_t)sh.prefix || kk - cum != sh.kk)
                                printf("MISMATCH q=%u pass=%d kk=%u: walk (%d, %u) super-bin (%u, %u) sb=%u l1=%u kk2=%u l2=%u\n", blockIdx.x, pass, kk, dd, kk - cum,
                                       (uint32_t)sh.prefix, sh.kk, sb, l1, kk2, l2);
                        }
#endif
                    }
                }
            }
            __syncthreads();
            enough = sh.bin_cnt != 0xFFFFFFFFu;
            const uint32_t d = (uint32_t)sh.prefix;
            kk = sh.kk;
            __syncthreads(); // (sh.prefix / sh.kk / sh.bin_cnt are rewritten by the next pass or the exact path)
            if (!enough) break;
            if (pass == 0) pfx = d;
            else pfx = (pfx << 11) | d;
            SEL_STAMP(3 + pass);
        }
        if (enough) {
            uint32_t tkey = pfx << 10; // the bin's lower edge as a 32-bit score key
            tkey = tkey < 0x007FFFFFu ? 0x007FFFFFu : tkey; // (below key(-inf) the bit pattern is a NaN's: -inf instead)
            const uint32_t t32 = eps2 < __builtin_inff() ? oi_f32_key(oi_key_f32(tkey) - eps2) : 0u;
            // the filter: a thread counts its survivors, the wave takes ONE slot range for all of them
            uint32_t mine = 0;
            for_each([&](bool valid, uint64_t kv) { mine += valid && (uint32_t)(kv >> 32) >= t32 ? 1u : 0u; });
            const uint32_t incl = sel_incl_scan(mine);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t base = 0;
            if (lane == 0 && tot) base = atomicAdd(&sh.cnt, tot);
            uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + incl - mine;
            for_each([&](bool valid, uint64_t kv) {
                if (valid && (uint32_t)(kv >> 32) >= t32) {
                    if (pos < SEL_CAND) cand[pos] = kv;
                    ++pos;
                }
            });
            __syncthreads();
            const uint32_t c = sh.cnt;
            *in_cand = true;
            *margin_tau = t32;
            *margin_overflow = c > SEL_CAND;
            SEL_STAMP(5);
            return c > SEL_CAND ? SEL_CAND : c;
        }
    }
    int shift;
    uint64_t prefix;
    bool selected = false; // sel[0..k) holds the top k
    if (KPT != 4 && n > SEL_CAND) {
        // ---- cut the pool down with a sampled threshold: one key per thread, spread over the wave's range
        const uint32_t js = lane % kpt, is = first + js * 64;
        bool sv = is < n;
        uint64_t sk;
        if constexpr (KPT > 0) {
            sk = key[0];
#pragma unroll
            for (int j = 1; j < KPT; ++j) sk = (uint32_t)j == js ? key[j] : sk;
            sv = sv && !((dead >> js) & 1u);
        } else {
            SelCursor cur = sel_seek(K, sv ? is : last);
            sk = K.pool[sel_at(K, cur, sv ? is : last)];
            sv = sv && !(skip && skipped(sk));
        }
        if (tid == 0) sh.n_samples = 0;
        __syncthreads();
        {
            const unsigned long long m = __ballot(sv);
            if (lane == 0 && m) atomicAdd(&sh.n_samples, (uint32_t)__popcll(m));
        }
        __syncthreads();
        const uint32_t ns = sh.n_samples;
        uint32_t r = (uint32_t)(((uint64_t)ns * (2 * k) + n - 1) / n) + 1; // keeps ~2k keys (+1 sample of margin)
        r = r > ns ? ns : r;
        sel_threshold([&](auto &&f) { f(sv, sk); }, r, sh, shift, prefix);
        for_each([&](bool valid, uint64_t kv) { sel_append(cand, &sh.cnt, SEL_CAND, valid && (kv >> shift) >= prefix, kv); });
        __syncthreads();
        const uint32_t c = sh.cnt;
        __syncthreads();
        if (tid == 0) sh.cnt = 0;
        __syncthreads();
        if (c >= k && c <= SEL_CAND) { // the top k of the pool are the top k of cand[0..c)
            uint64_t ck[SEL_CAND / SEL_THREADS];
#pragma unroll
            for (int j = 0; j < SEL_CAND / SEL_THREADS; ++j) {
                const uint32_t i = (uint32_t)j * SEL_THREADS + tid;
                ck[j] = cand[i < c ? i : c - 1];
            }
            auto cand_each = [&](auto &&f) {
#pragma unroll
                for (int j = 0; j < SEL_CAND / SEL_THREADS; ++j)
                    if ((uint32_t)j * SEL_THREADS < c) f((uint32_t)j * SEL_THREADS + tid < c, ck[j]);
            };
            if (c == k) { // nothing to select
                cand_each([&](bool valid, uint64_t kv) { sel_append(sel, &sh.cnt, SEL_MAX, valid, kv); });
            } else {
                sel_threshold(cand_each, k, sh, shift, prefix);
                cand_each([&](bool valid, uint64_t kv) { sel_append(sel, &sh.cnt, SEL_MAX, valid && (kv >> shift) >= prefix, kv); });
            }
            __syncthreads();
            selected = true;
        }
        // else: the sample misjudged the pool -- exact passes over all of it
    }
    if (!selected) {
        sel_threshold(for_each, k, sh, shift, prefix);
        // every key whose top bits are >= prefix is selected: exactly k of them (keys are distinct)
        for_each([&](bool valid, uint64_t kv) { sel_append(sel, &sh.cnt, SEL_MAX, valid && (kv >> shift) >= prefix, kv); });
        __syncthreads();
    }
    const uint32_t m = sh.cnt < k ? sh.cnt : k;
    if (!(eps2 >= 0.f) || m < k) return m;
    __syncthreads(); // everyone has read sh.cnt before it is reset below
    // ---- margin mode: the k-th key is the smallest selected one; keep everything within eps2 of its score
    unsigned long long lo = ~0ull;
    for (uint32_t i = tid; i < m; i += SEL_THREADS) lo = sel[i] < lo ? sel[i] : lo;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long v = __shfl_xor(lo, o, OI_WAVE);
        lo = v < lo ? v : lo;
    }
    if (lane == 0 && lo != ~0ull) atomicMin(&sh.min_key, lo);
    if (tid == 0) sh.cnt = 0;
    __syncthreads();
    // an infinite margin (a query without a bound, cosine_prefilter.hip) keeps everything and never raises the threshold
    const uint32_t t32 = eps2 < __builtin_inff() ? oi_f32_key(oi_key_f32((uint32_t)(sh.min_key >> 32)) - eps2) : 0u;
    for_each([&](bool valid, uint64_t kv) { sel_append(cand, &sh.cnt, SEL_CAND, valid && (uint32_t)(kv >> 32) >= t32, kv); });
    __syncthreads();
    const uint32_t c = sh.cnt;
    *in_cand = true;
    *margin_tau = t32;
    *margin_overflow = c > SEL_CAND;
    return c > SEL_CAND ? SEL_CAND : c;
}

__global__ __launch_bounds__(SEL_THREADS) void select_flat_kernel(
    uint64_t *pools, uint32_t *carry_cnt, uint32_t *seg_cnt, uint32_t *tau_keys, uint64_t pool_stride,
    uint32_t carry_cap, uint32_t seg_cap, uint32_t n_segs, uint32_t seg_cnt_stride, uint32_t *overflow,
    uint32_t k, int compact, float *out_scores, uint32_t *out_docs, uint32_t *out_counts, uint32_t out_stride,
    const float *eps2, uint32_t *margin_gate, const uint32_t *run_gate, const uint32_t *skip, uint32_t skip_base) {
    // run_gate: this launch belongs to the gated exact pipeline (cosine_prefilter.hip) and only runs when the
    // screen gave up.  eps2: margin mode (see sel_flat_select); its overflow opens that gate.
    if (run_gate && *run_gate == 0u) return;
    __shared__ SelShared sh;
    __shared__ uint64_t sel[2 * SEL_MAX];
    __shared__ uint64_t cand[SEL_CAND];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint64_t *pool = pools + (uint64_t)q * pool_stride;
    uint32_t *segc = seg_cnt + (uint64_t)q * seg_cnt_stride;
    SEL_STAMP(0);
    const uint32_t c0_raw = carry_cnt[q];    // (issued beside the segment counts: not a second round trip after the scan)
    const float e2 = eps2 ? eps2[q] : -1.f;

    // exclusive scan of the (clamped) segment counts -> sh.seg_off; SEL_MAX_SEGS / SEL_THREADS = 4 per thread
    constexpr int SPT = SEL_MAX_SEGS / SEL_THREADS;
    uint32_t c[SPT], local = 0;
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        const uint32_t sg = tid * SPT + u;
        uint32_t v = segc[sg < n_segs ? sg : 0];
        if (sg >= n_segs) v = 0;
        if (v > seg_cap) { *overflow = 1u; v = seg_cap; }
        c[u] = v;
        local += v;
    }
    const uint32_t incl = sel_incl_scan(local);
    if (lane == 63) sh.wave_tot[wv] = incl;
    if (tid == 0) { sh.cnt = 0; sh.min_key = ~0ull; }
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SEL_THREADS / 64; ++w) {
        const uint32_t t = sh.wave_tot[w];
        base += (uint32_t)w < wv ? t : 0u;
        total += t;
    }
    uint32_t run = base + incl - local;
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
        const uint32_t sg = tid * SPT + u;
        if (sg < n_segs) sh.seg_off[sg] = run;
        run += c[u];
    }
    if (tid == 0) sh.seg_off[n_segs] = total;
    __syncthreads();

    SelFlat K;
    K.pool = pool; K.seg_off = sh.seg_off; K.carry_cap = carry_cap; K.seg_cap = seg_cap; K.n_segs = n_segs;
    K.c0 = c0_raw < carry_cap ? c0_raw : carry_cap;
    K.n = K.c0 + total;
    K.steps = 0;
    while ((1u << K.steps) < n_segs) ++K.steps;
    SEL_STAMP(1);

    uint32_t m, m_tau = 0;
    bool in_cand = false, m_over = false;
    if (K.n <= 4 * SEL_THREADS) m = sel_flat_select<4>(K, k, sh, sel, cand, e2, &in_cand, &m_tau, &m_over, skip, skip_base);
    else if (K.n <= 8 * SEL_THREADS) m = sel_flat_select<8>(K, k, sh, sel, cand, e2, &in_cand, &m_tau, &m_over, skip, skip_base);
    else if (K.n <= 16 * SEL_THREADS) m = sel_flat_select<16>(K, k, sh, sel, cand, e2, &in_cand, &m_tau, &m_over, skip, skip_base);
    else if (K.n <= SEL_KPT_MAX * SEL_THREADS) m = sel_flat_select<SEL_KPT_MAX>(K, k, sh, sel, cand, e2, &in_cand, &m_tau, &m_over, skip, skip_base);
    else m = sel_flat_select<0>(K, k, sh, sel, cand, e2, &in_cand, &m_tau, &m_over, skip, skip_base);
    if (in_cand) { // margin mode: an unsorted superset of the top k in cand[]; only ever compacted
        if (m_over && tid == 0 && margin_gate) *margin_gate = 1u;
        for (uint32_t i = tid; i < m; i += SEL_THREADS) pool[i] = cand[i];
        for (uint32_t sg = tid; sg < n_segs; sg += SEL_THREADS) segc[sg] = 0;
        if (tid == 0) {
            carry_cnt[q] = m;
            if (tau_keys && m_tau > tau_keys[q]) tau_keys[q] = m_tau;
        }
#ifdef OI_ABLATION
        __syncthreads();
        if (sel_dbg_on && blockIdx.x == 0 && tid == 0) {
            const unsigned long long e = __builtin_readcyclecounter();
            printf("select margin n=%u m=%u segs=%u: scan %llu, loads+skip %llu, pass0 %llu, pass1 %llu, filter %llu, compact %llu (cycles)\n", K.n, m, n_segs,
                   sel_dbg_t[1] - sel_dbg_t[0], sel_dbg_t[2] - sel_dbg_t[1], sel_dbg_t[3] - sel_dbg_t[2], sel_dbg_t[4] - sel_dbg_t[3], sel_dbg_t[5] - sel_dbg_t[4], e - sel_dbg_t[5]);
        }
#endif
        return;
    }

    // Sorted output is needed only for a final list; an intermediate compaction just needs the set
    // and its smallest key (the new threshold).
    if (out_scores) {
        uint32_t P = 2;
        while (P < m) P <<= 1;
        for (uint32_t i = m + tid; i < P; i += SEL_THREADS) sel[i] = 0; // lowest possible key
        __syncthreads();
        bitonic_sort_desc(sel, P);
        for (uint32_t i = tid; i < m; i += SEL_THREADS) {
            const uint64_t key = sel[i];
            out_scores[(uint64_t)q * out_stride + i] = oi_rank_key_score(key);
            out_docs[(uint64_t)q * out_stride + i] = oi_rank_key_doc(key);
        }
        if (tid == 0) { out_counts[q] = m; sh.min_key = m ? sel[m - 1] : 0; }
    } else {
        unsigned long long lo = ~0ull;
        for (uint32_t i = tid; i < m; i += SEL_THREADS) lo = sel[i] < lo ? sel[i] : lo;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long v = __shfl_xor(lo, o, OI_WAVE);
            lo = v < lo ? v : lo;
        }
        if (lane == 0 && lo != ~0ull) atomicMin(&sh.min_key, lo);
    }
    __syncthreads();
    if (compact) {
        for (uint32_t i = tid; i < m; i += SEL_THREADS) pool[i] = sel[i];
        for (uint32_t sg = tid; sg < n_segs; sg += SEL_THREADS) segc[sg] = 0;
        if (tid == 0) {
            carry_cnt[q] = m;
            if (m == k && tau_keys && !eps2) { // k docs at or above this score exist: a valid lower bound of the final
                // k-th (a margin-mode pool of exactly k keys keeps its threshold: that bound needs the margin)
                const uint32_t t = (uint32_t)(sh.min_key >> 32);
                if (t > tau_keys[q]) tau_keys[q] = t;
            }
        }
    }
}

int oi_launch_select(oi_ctx *ctx, const PoolView &pool, uint32_t n_queries, uint32_t k, bool compact,
                     float *out_scores, uint32_t *out_docs, uint32_t *out_counts, uint32_t out_stride,
                     const SelectExtra *extra) {
    if (n_queries == 0) return OI_OK;
    OI_REQUIRE(k >= 1 && k <= OI_MAX_DEPTH && k <= pool.carry_cap, "select: k=%u outside [1,%u]", k, OI_MAX_DEPTH);
    ProfScope ps(ctx, "select");
    static const bool v1 = oi_ablation_env("OI_SELECT_V1") != nullptr; // A/B switch: the segment-walking kernel
#ifdef OI_ABLATION
    static const bool stamps = [] {
        const int on = oi_ablation_env("OI_SELECT_STAMPS") ? std::max(1, atoi(oi_ablation_env("OI_SELECT_STAMPS"))) : 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(sel_dbg_on), &on, sizeof(int));
        return on != 0;
    }();
    (void)stamps;
#endif
    const bool special = extra && (extra->eps2 || extra->run_gate);
    if (special) {
        OI_REQUIRE(pool.n_segs <= SEL_MAX_SEGS, "select: %u segments (margin / gated selects take <= %u)", pool.n_segs, SEL_MAX_SEGS);
        OI_REQUIRE(!extra->eps2 || (compact && !out_scores && pool.carry_cap >= SEL_CAND),
                   "select: margin mode compacts into a carry region of >= %u keys", SEL_CAND);
    }
    if ((!v1 || special) && pool.n_segs <= SEL_MAX_SEGS) {
        hipLaunchKernelGGL(select_flat_kernel, dim3(n_queries), dim3(SEL_THREADS), 0, ctx->stream, pool.keys,
                           pool.carry_cnt, pool.seg_cnt, pool.tau_keys, pool.stride, pool.carry_cap, pool.seg_cap,
                           pool.n_segs, pool.seg_cnt_stride, pool.overflow, k, compact ? 1 : 0, out_scores, out_docs,
                           out_counts, out_stride, extra ? extra->eps2 : nullptr, extra ? extra->margin_gate : nullptr,
                           extra ? extra->run_gate : nullptr, extra && extra->eps2 ? extra->skip_bitmap : nullptr,
                           extra ? extra->skip_base : 0u);
        OI_HIP_CHECK(hipGetLastError());
        return OI_OK;
    }
    hipLaunchKernelGGL(select_topk_kernel, dim3(n_queries), dim3(SEL_THREADS), 0, ctx->stream, pool.keys,
                       pool.carry_cnt, pool.seg_cnt, pool.tau_keys, pool.stride, pool.carry_cap, pool.seg_cap,
                       pool.n_segs, pool.seg_cnt_stride, pool.overflow, k, compact ? 1 : 0, out_scores, out_docs,
                       out_counts, out_stride);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

// [n_shards][n_queries][depth] lists -> pool segment s of query q = shard s's list (keys rebuilt).
__global__ void lists_to_pool_kernel(const float *scores, const uint32_t *docs, const uint32_t *counts,
                                     uint64_t shard_stride, uint64_t count_stride, uint32_t n_queries,
                                     uint32_t depth, uint64_t *pools, uint32_t *carry_cnt,
                                     uint32_t *seg_cnt, uint64_t pool_stride, uint32_t carry_cap,
                                     uint32_t seg_cnt_stride) {
    const uint32_t q = blockIdx.x, sh = blockIdx.y;
    uint32_t c = counts[(uint64_t)sh * count_stride + q];
    if (c > depth) c = depth;
    const uint64_t src = (uint64_t)sh * shard_stride + (uint64_t)q * depth;
    uint64_t *seg = pools + (uint64_t)q * pool_stride + carry_cap + (uint64_t)sh * depth;
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) seg[i] = oi_rank_key(scores[src + i], docs[src + i]);
    if (threadIdx.x == 0) {
        seg_cnt[(uint64_t)q * seg_cnt_stride + sh] = c;
        if (sh == 0) carry_cnt[q] = 0;
    }
}

int oi_launch_lists_to_pool(oi_ctx *ctx, const float *scores, const uint32_t *docs,
                            const uint32_t *counts, uint64_t shard_stride, uint64_t count_stride,
                            uint32_t n_shards, uint32_t n_queries, uint32_t depth, const PoolView &pool) {
    if (n_queries == 0) return OI_OK;
    OI_REQUIRE(pool.seg_cap == depth && pool.n_segs == n_shards, "merge: pool geometry mismatch");
    hipLaunchKernelGGL(lists_to_pool_kernel, dim3(n_queries, n_shards), dim3(256), 0, ctx->stream, scores, docs,
                       counts, shard_stride, count_stride, n_queries, depth, pool.keys, pool.carry_cnt, pool.seg_cnt, pool.stride,
                       pool.carry_cap, pool.seg_cnt_stride);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

// Reciprocal-rank fusion of two ranked lists per query.
//   rrf(d) = [d in A at rank i] 1/(60+i)  +  [d in B at rank j] 1/(60+j)     (f32, A before B)
// LDS hash table over list A's doc ids; union built in LDS; bitonic sort; top-k out.
#define RRF_HASH 4096
__global__ __launch_bounds__(SEL_THREADS) void rrf_kernel(const uint32_t *docs_a, const uint32_t *counts_a,
                                                          const uint32_t *docs_b, const uint32_t *counts_b,
                                                          uint32_t depth, uint32_t k, float *scores_out,
                                                          uint32_t *docs_out, uint32_t *counts_out) {
    __shared__ uint32_t h_doc[RRF_HASH];
    __shared__ uint32_t h_idx[RRF_HASH]; // rank index in A + 1, 0 = empty
    __shared__ uint32_t a_match[SEL_MAX]; // rank index in B + 1 of the same doc, 0 = none
    __shared__ uint64_t u[2 * SEL_MAX];
    __shared__ uint32_t s_extra;

    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    uint32_t na = counts_a[q], nb = counts_b[q];
    if (na > depth) na = depth;
    if (nb > depth) nb = depth;
    const uint32_t *da = docs_a + (uint64_t)q * depth, *db = docs_b + (uint64_t)q * depth;

    for (uint32_t i = tid; i < RRF_HASH; i += SEL_THREADS) h_idx[i] = 0;
    for (uint32_t i = tid; i < SEL_MAX; i += SEL_THREADS) a_match[i] = 0;
    if (tid == 0) s_extra = 0;
    __syncthreads();
    for (uint32_t i = tid; i < na; i += SEL_THREADS) {
        uint32_t d = da[i];
        uint32_t h = (d * 2654435761u) >> 20; // 12 bits
        for (;;) {
            uint32_t prev = atomicCAS(&h_idx[h], 0u, i + 1);
            if (prev == 0) { h_doc[h] = d; break; }
            h = (h + 1) & (RRF_HASH - 1);
        }
    }
    __syncthreads();
    // list B: look each doc up in A; unmatched docs become their own union entries
    for (uint32_t j = tid; j < nb; j += SEL_THREADS) {
        uint32_t d = db[j];
        uint32_t h = (d * 2654435761u) >> 20;
        uint32_t hit = 0;
        for (;;) {
            uint32_t ix = h_idx[h];
            if (ix == 0) break;
            if (h_doc[h] == d) { hit = ix; break; }
            h = (h + 1) & (RRF_HASH - 1);
        }
        if (hit) a_match[hit - 1] = j + 1;
        else {
            uint32_t pos = atomicAdd(&s_extra, 1u);
            float c = 1.0f / (60.0f + (float)(j + 1));
            u[na + pos] = oi_rank_key(c, d);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < na; i += SEL_THREADS) {
        float s = 1.0f / (60.0f + (float)(i + 1));
        uint32_t j1 = a_match[i];
        if (j1) s = s + 1.0f / (60.0f + (float)j1);
        u[i] = oi_rank_key(s, da[i]);
    }
    __syncthreads();
    const uint32_t m = na + s_extra;
    uint32_t P = 2;
    while (P < m) P <<= 1;
    for (uint32_t i = m + tid; i < P; i += SEL_THREADS) u[i] = 0;
    __syncthreads();
    bitonic_sort_desc(u, P);
    const uint32_t out = m < k ? m : k;
    for (uint32_t i = tid; i < out; i += SEL_THREADS) {
        scores_out[(uint64_t)q * k + i] = oi_rank_key_score(u[i]);
        docs_out[(uint64_t)q * k + i] = oi_rank_key_doc(u[i]);
    }
    if (tid == 0) counts_out[q] = out;
}

int oi_launch_rrf(oi_ctx *ctx, const uint32_t *docs_a, const uint32_t *counts_a, const uint32_t *docs_b,
                  const uint32_t *counts_b, uint32_t n_queries, uint32_t depth, uint32_t k,
                  float *scores_out, uint32_t *docs_out, uint32_t *counts_out) {
    if (n_queries == 0) return OI_OK;
    OI_REQUIRE(depth >= 1 && depth <= OI_MAX_DEPTH, "rrf: depth=%u outside [1,%u]", depth, OI_MAX_DEPTH);
    OI_REQUIRE(k >= 1 && k <= OI_MAX_DEPTH, "rrf: k=%u outside [1,%u]", k, OI_MAX_DEPTH);
    ProfScope ps(ctx, "rrf");
    hipLaunchKernelGGL(rrf_kernel, dim3(n_queries), dim3(SEL_THREADS), 0, ctx->stream, docs_a, counts_a,
                       docs_b, counts_b, depth, k, scores_out, docs_out, counts_out);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}
