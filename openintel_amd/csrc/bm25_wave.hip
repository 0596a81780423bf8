// bm25_wave.hip -- term-at-a-time BM25 over the blocked inverted index, one WAVE per (doc block, query) task.
//
// Builder-defined like the rest of the retrieval path (the reference has no BM25).  Same index
// and the same arithmetic as bm25.hip (score(d) = sum over the query's terms IN QUERY ORDER of idf_t * impact(t, d),
// every f32 op an explicit round-to-nearest intrinsic): the lists are bit-identical to the oracle's and to the other
// two BM25 kernels' (tests/test_gpu_parity.py).
//
// Why.  bm25_block_kernel gives a (block, query) task -- ~700 postings at 10M docs -- to a 1024-thread workgroup
// with 128 KiB of LDS accumulators: ~10 workgroup barriers and a handful of dependent memory steps per task, one
// workgroup per CU.  19.6K such tasks per 64-query batch took 0.6 ms for 115 MB of postings (2.4 % of the HBM roof;
// rocprofv3: 52 % LDS bank-conflict cycles, 132 B of scratch per lane, 129 MB of pool writes per launch).  Here a
// task belongs to ONE WAVE, 12 waves per CU walk their own task lists, and nothing in a task waits for another wave.
// Most docs of a task occur in ONE of the query's runs (~95 % for 4-term queries): their score is that one product,
// (+0) + idf * impact, and needs no accumulator at all.  So:
//   load     the task's postings are cut into 64-posting chunks in query order by a scalar cursor over the runs and
//            fetched a round of 8 chunks at a time, the next round in flight while this one is processed (the second
//            pass reads them again: L2 hits);
//   pass A   every posting ORs its doc's bit into a wave-private 32768-bit map (ds_or returning the old word); a bit
//            that was already set -- the doc was in an earlier run -- is ORed into a second map, "multi".  All the
//            chunks' atomics are issued before the first result is waited for (a wave's LDS operations execute in
//            order, so a later run sees the earlier runs' bits);
//   pass B   the chunks again, in QUERY ORDER: a posting whose doc is not in "multi" is a finished score and goes
//            straight from registers to the threshold test; the few multi docs are summed in query order in a small
//            wave-private hash table (one 64-bit compare-and-swap inserts {doc, first product} or returns the entry
//            to add to; a run lists a doc once, so no two lanes ever update one entry) and emitted at the end;
//   emit     straight into the task's OWN pool segment of capacity BM_R (a block cannot touch more docs than it
//            has): no overflow path, no in-kernel selection; with a threshold only scores >= tau are written.
// A task with more multi-doc postings than the table takes (384) is cut into doc-id windows (halved until they fit):
// exact for any data, one window in the common case.  (Two earlier forms, both measured: an open-addressing table for
// ALL docs -- a wave ran as slow as its longest probe chain, 26K cycles per task -- and a bitmap + rank perfect hash
// with a dense accumulator array -- 65K cycles per task, instruction-bound on 32 fixed register slots per lane, the
// rank sweep and a read-modify-write per posting.  DESIGN.md 4.3 has the ladders.)  HBM-bound by construction (8 B
// per posting + 8 B per (block, term) bounds lookup); at this batch size the floor is launch and latency, not bytes.
#include "oi_device.h"
#include "oi_internal.h"

#define BW_R OI_BM25_BLOCK_DOCS
#define BW_WORDS (BW_R / 32)      // 1024 words per bitmap
#define BW_HASH 512u              // entries {tag = doc-in-block + 1, f32 score bits} of the multi-doc table
#define BW_MULTI_CAP 384u         // multi-doc postings a window may hold (load factor <= 0.75)
#define BW_WAVES 4                // waves per workgroup
#define BW_STAGE_TERMS 256u       // query terms of a pass staged in LDS (more: read from global memory)
#define BW_MAX_Q 128u             // queries per pass (term offsets, weights and order staged in LDS)
#define BW_ROUND 4                 // 64-posting chunks fetched and processed together
#define BW_WAVE_LDS (BW_WORDS * 4 * 2 + BW_HASH * 8)   // bytes: seen map | multi map | multi table

struct BwPosting {
    uint32_t dib;
    float impact;
};

// block/query of task t.  Tasks are dealt query-major over the queries in DESCENDING weight (postings per block):
// wave w takes t = w, w + G, ... -- every wave starts in the heavy queries and ends in the light ones, so the waves
// finish together; s_order[rank] = query.
__device__ __forceinline__ void bw_task(uint32_t t, uint32_t nb, const uint32_t *s_order, uint32_t *blk, uint32_t *q) {
    const uint32_t r = t / nb;
    *blk = t - r * nb;
    *q = s_order[r];
}

// DBG != 0 (-DOI_ABLATION builds only; results WRONG by construction, timings only): 1 = bounds only, 2 = + front loads,
// 3 = + pass A, 4 = + pass B without the multi-doc table.
template <bool TIMING, int DBG = 0>
__global__ __launch_bounds__(BW_WAVES * 64, 3) void bm25_wave_kernel(
    const BwPosting *__restrict__ postings, const uint32_t *__restrict__ cell_start, const float *__restrict__ idf,
    const uint32_t *__restrict__ df, uint32_t vocab, uint32_t doc_id_base, uint32_t block0, uint32_t n_blocks_here,
    const uint32_t *__restrict__ q_terms, const uint32_t *__restrict__ q_offsets, uint32_t q_begin, uint32_t nq,
    uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride, const uint32_t *tau_keys, uint64_t pool_stride,
    uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow, unsigned long long *timing) {
    // TIMING (an -DOI_ABLATION diagnostic build only): per-section cycle sums of every wave into timing[0..7]
    unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0};
    auto stamp = [&]() -> unsigned long long { return TIMING ? __builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long t_kernel0 = stamp();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // [BW_WAVES][seen u32 x 1024 | multi u32 x 1024 | table u64 x 512] | q_off[MAX_Q + 1] | order[MAX_Q] | weight[MAX_Q] | terms[STAGE]
    uint32_t *s_qoff = reinterpret_cast<uint32_t *>(smem_raw + BW_WAVES * BW_WAVE_LDS);
    uint32_t *s_order = s_qoff + BW_MAX_Q + 1;
    uint32_t *s_weight = s_order + BW_MAX_Q;
    uint32_t *s_terms = s_weight + BW_MAX_Q;

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *seen = reinterpret_cast<uint32_t *>(smem_raw + wv * BW_WAVE_LDS);
    uint32_t *multi = seen + BW_WORDS;
    unsigned long long *tab = reinterpret_cast<unsigned long long *>(multi + BW_WORDS);
    uint4 *maps4 = reinterpret_cast<uint4 *>(seen); // both maps: 8 KiB = 512 x 16 B
    uint4 *tab4 = reinterpret_cast<uint4 *>(tab);   // 4 KiB = 256 x 16 B
    auto clear_maps = [&]() {
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) maps4[k * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
    };
    auto clear_table = [&]() {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) tab4[k * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
    };

    // ---- stage the pass's query terms (offsets relative to the pass's first term), weigh and rank the queries
    const uint32_t term0 = q_offsets[q_begin];
    for (uint32_t i = tid; i <= nq; i += BW_WAVES * 64) s_qoff[i] = q_offsets[q_begin + i] - term0;
    __syncthreads();
    const uint32_t n_terms_pass = s_qoff[nq];
    const bool staged = n_terms_pass <= BW_STAGE_TERMS;
    if (staged)
        for (uint32_t i = tid; i < n_terms_pass; i += BW_WAVES * 64) s_terms[i] = q_terms[term0 + i];
    for (uint32_t qq = tid; qq < nq; qq += BW_WAVES * 64) { // weight = sum of the terms' document frequencies (saturating)
        uint32_t wsum = 0;
        for (uint32_t i = s_qoff[qq]; i < s_qoff[qq + 1]; ++i) {
            const uint32_t term = q_terms[term0 + i];
            const uint32_t d = term < vocab ? df[term] : 0u;
            wsum = wsum + d < wsum ? 0xFFFFFFFFu : wsum + d;
        }
        s_weight[qq] = wsum;
    }
    clear_maps();
    clear_table();
    __syncthreads();
    for (uint32_t qq = tid; qq < nq; qq += BW_WAVES * 64) { // rank by (weight desc, query asc): nq <= 256 comparisons each
        const uint32_t wq = s_weight[qq];
        uint32_t rank = 0;
        for (uint32_t o = 0; o < nq; ++o) {
            const uint32_t wo = s_weight[o];
            rank += (wo > wq || (wo == wq && o < qq)) ? 1u : 0u;
        }
        s_order[rank] = qq;
    }
    __syncthreads(); // the only barriers of the kernel

    const uint32_t n_tasks = n_blocks_here * nq;
    const uint32_t G = gridDim.x * BW_WAVES, w = blockIdx.x * BW_WAVES + wv;
    if (TIMING) t_acc[0] += stamp() - t_kernel0; // staging

    // Bounds of terms [g0, g0 + 64) of a task, one term per lane: the run [s, e) in the postings array and idf.
    // RAW loads only (no arithmetic on the loaded values: the compiler waits for a load where its value is first
    // computed with, and this fetch is meant to stay in flight behind the previous task); the caller applies okm.
    auto fetch_bounds = [&](uint32_t t, uint32_t g0, uint32_t &s, uint32_t &e, float &wt, uint32_t &okm) {
        uint32_t blk, q;
        bw_task(t < n_tasks ? t : 0, n_blocks_here, s_order, &blk, &q);
        const uint32_t tb = s_qoff[q], te = s_qoff[q + 1];
        const bool have = t < n_tasks && tb + g0 + lane < te;
        const uint32_t ti = have ? tb + g0 + lane : 0u;
        uint32_t term = n_terms_pass == 0 ? 0u : (staged ? s_terms[ti] : q_terms[term0 + ti]);
        const bool ok = have && term < vocab;
        term = ok ? term : 0u;
        const uint64_t cell = (uint64_t)(block0 + blk) * vocab + term;
        s = cell_start[cell];
        e = cell_start[cell + 1];
        wt = idf[term];
        okm = ok ? 1u : 0u;
    };

    uint32_t nx_s, nx_e, nx_ok;
    float nx_w;
    fetch_bounds(w, 0, nx_s, nx_e, nx_w, nx_ok);

    for (uint32_t t = w; t < n_tasks; t += G) {
        const unsigned long long t_task0 = stamp();
        uint32_t blk, q;
        bw_task(t, n_blocks_here, s_order, &blk, &q);
        const uint32_t my_s = nx_ok ? nx_s : 0u, my_e = nx_ok ? nx_e : 0u;
        const float my_w = nx_ok ? nx_w : 0.f;
        const uint32_t tb = s_qoff[q], te = s_qoff[q + 1];
        const uint32_t nt = te - tb;
        const uint32_t tau = tau_keys ? tau_keys[q] : 0u;
        const uint32_t doc0 = doc_id_base + (block0 + blk) * BW_R;
        uint64_t *seg = pools + (uint64_t)q * pool_stride + carry_cap + (uint64_t)(block0 + blk) * seg_cap;

        // ---- the task's postings as 64-posting chunks IN QUERY ORDER (a cursor over the runs of the first 64 terms: a
        // run of n postings is ceil(n / 64) chunks, a chunk never spans two runs), fetched a ROUND of chunks at a
        // time, the next round in flight while this one is processed
        const uint32_t ntf = nt < 64u ? nt : 64u;
        struct Cursor { uint32_t j, i, e; };
        auto cursor_term = [&](Cursor &c, uint32_t j) {
            c.i = (uint32_t)__builtin_amdgcn_readlane((int)my_s, (int)j);
            c.e = (uint32_t)__builtin_amdgcn_readlane((int)my_e, (int)j);
        };
        auto cursor_skip_empty = [&](Cursor &c) { // uniform: stops at a run with postings left, or at j == ntf with i = e = 0
            while (c.j < ntf && c.i >= c.e) {
                ++c.j;
                if (c.j < ntf) cursor_term(c, c.j);
                else { c.e = 0; c.i = 0; }
            }
        };
        auto cursor_begin = [&]() {
            Cursor c{0u, 0u, 0u};
            if (ntf) cursor_term(c, 0);
            cursor_skip_empty(c);
            return c;
        };
        struct Round {
            BwPosting p[BW_ROUND];
            uint32_t meta[BW_ROUND / 2]; // uniform, two chunks per word: postings in the chunk (0: none) | term index << 7
        };
        auto load_round = [&](Cursor &c) { // unconditional loads (an exhausted cursor reads postings[0], n = 0)
            Round r;
#pragma unroll
            for (int k = 0; k < BW_ROUND / 2; ++k) r.meta[k] = 0u;
#pragma unroll
            for (int k = 0; k < BW_ROUND; ++k) {
                const bool live = c.i < c.e; // (an exhausted cursor stays at i = e = 0: it must not advance)
                const uint32_t left = live ? c.e - c.i : 0u;
                const uint32_t n = left < 64u ? left : 64u;
                r.meta[k >> 1] |= __builtin_amdgcn_readfirstlane(n | ((c.j & 63u) << 7)) << (16 * (k & 1));
                r.p[k] = postings[lane < n ? c.i + lane : (live ? c.e - 1u : 0u)];
                c.i = live ? c.i + 64u : c.i;
                cursor_skip_empty(c);
            }
            return r;
        };
        auto chunk_n = [&](const Round &r, int k) -> uint32_t { return (r.meta[k >> 1] >> (16 * (k & 1))) & 127u; };
        auto chunk_w = [&](const Round &r, int k) -> float { // the idf of the chunk's term: lane (term index) of my_w
            const uint32_t tj = __builtin_amdgcn_readfirstlane((r.meta[k >> 1] >> (16 * (k & 1) + 7)) & 63u);
            return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), (int)tj));
        };
        // One instantiation of f per pass (the code of a task is executed once per task: every kilobyte of it is a cold
        // instruction fetch for the wave's first task -- a 40 KB build of this kernel took 50 us for ONE task).  The next
        // round is requested before this one is processed; the copy at the end of the body is where it is waited for.
        auto for_each_round = [&](auto &&f) {
            Cursor c = cursor_begin();
            Round ra = load_round(c);
            while (chunk_n(ra, 0) != 0u) {
                const Round rb = load_round(c);
                f(ra);
                ra = rb;
            }
        };
        fetch_bounds(t + G, 0, nx_s, nx_e, nx_w, nx_ok); // the next task's bounds: in flight while this one is processed
        if (DBG == 1 || DBG == 2) {
            uint32_t x = my_s + my_e;
            if (DBG == 2)
                for_each_round([&](const Round &r) {
                    for (int k = 0; k < BW_ROUND; ++k) x += (r.p[k].dib ^ __float_as_uint(r.p[k].impact)) + chunk_n(r, k);
                });
            if (lane == 0 || x == 0xDEADBEEFu) seg_cnt[(uint64_t)q * seg_cnt_stride + block0 + blk] = x == 0xDEADBEEFu ? 1u : 0u;
            continue;
        }

        // Terms beyond the 64 a wave holds bounds for (long queries), in query order, one chunk at a time (not pipelined):
        // f(posting of this lane, lane holds one, idf weight)
        auto for_each_late_chunk = [&](auto &&f) {
            for (uint32_t g0 = 64; g0 < nt; g0 += 64) {
                uint32_t gs, ge, okm;
                float gw;
                fetch_bounds(t, g0, gs, ge, gw, okm);
                gs = okm ? gs : 0u; ge = okm ? ge : 0u; gw = okm ? gw : 0.f;
                const uint32_t gn = nt - g0 < 64u ? nt - g0 : 64u;
                for (uint32_t j = 0; j < gn; ++j) {
                    const uint32_t s = (uint32_t)__shfl((int)gs, (int)j, OI_WAVE), e = (uint32_t)__shfl((int)ge, (int)j, OI_WAVE);
                    const float wt = __shfl(gw, (int)j, OI_WAVE);
                    for (uint32_t i0 = s; i0 < e; i0 += 64u) f(postings[i0 + lane < e ? i0 + lane : e - 1u], i0 + lane < e, wt);
                }
            }
        };

        uint32_t out_cnt = 0;
        auto emit = [&](bool keep, float v, uint32_t dib) { // all lanes call; appends the kept scores to the segment
            const unsigned long long m = __ballot(keep);
            if (m) {
                if (keep) {
                    const uint32_t pos = out_cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (pos < seg_cap) seg[pos] = oi_rank_key(v, doc0 + dib);
                }
                out_cnt += (uint32_t)__popcll(m);
            }
        };

        uint32_t width = BW_R, lo = 0;
        while (lo < BW_R) {
            const uint32_t hi = lo + width;
            const unsigned long long t_a0 = stamp();
            // ---- pass A: seen / multi maps.  A round's eight atomics are issued, THEN their results are used.
            uint32_t n_multi = 0;
            auto mark = [&](bool inw, uint32_t d, uint32_t oldword) { // all lanes call
                const uint32_t bit = 1u << (d & 31u);
                const bool again = inw && (oldword & bit);
                const unsigned long long m = __ballot(again);
                if (m) { // uniform: rare
                    if (again) atomicOr(&multi[d >> 5], bit);
                    n_multi += (uint32_t)__popcll(m);
                }
            };
            for_each_round([&](const Round &r) {
                uint32_t oldw[BW_ROUND];
#pragma unroll
                for (int k = 0; k < BW_ROUND; ++k) {
                    const uint32_t d = r.p[k].dib;
                    const bool inw = lane < chunk_n(r, k) && d >= lo && d < hi;
                    oldw[k] = 0u;
                    if (inw) oldw[k] = atomicOr(&seen[d >> 5], 1u << (d & 31u));
                }
#pragma unroll
                for (int k = 0; k < BW_ROUND; ++k) mark(oldw[k] != 0u, r.p[k].dib, oldw[k]); // (outside window / chunk: oldw = 0)
            });
            for_each_late_chunk([&](const BwPosting &p, bool have, float) {
                const uint32_t d = p.dib;
                const bool inw = have && d >= lo && d < hi;
                uint32_t o = 0u;
                if (inw) o = atomicOr(&seen[d >> 5], 1u << (d & 31u));
                mark(inw, d, o);
            });
            if (TIMING) t_acc[3] += stamp() - t_a0; // pass A
            if (DBG == 3) { clear_maps(); lo = hi; continue; }
            if (n_multi > BW_MULTI_CAP) { // too many multi-doc postings for the table: clear, halve the window, again
                clear_maps();
                if (width > 1u) width >>= 1;
                else { *overflow = 1u; lo = hi; } // cannot happen (a doc has <= 1024 postings here); never loop forever
                continue;
            }
            const unsigned long long t_b0 = stamp();
            // ---- pass B: in query order.  Single-run docs are finished scores; multi docs go through the table.
            auto accumulate = [&](bool mul, uint32_t d, float x) { // lanes with mul: table[d] += x, in call order
                if (DBG == 4) return;
                bool pend = mul;
                uint32_t slot = (d * 0x9E3779B1u) >> 23; // 9 bits
                const uint32_t tag = d + 1u;
                while (__ballot(pend)) {
                    if (pend) {
                        const unsigned long long prev = atomicCAS(&tab[slot], 0ull, ((unsigned long long)__float_as_uint(x) << 32) | tag);
                        if (prev == 0ull) pend = false; // first run of this doc: (+0) + x
                        else if ((uint32_t)prev == tag) { // seen in an earlier run: add in query order, plain store
                            const float v = __fadd_rn(__uint_as_float((uint32_t)(prev >> 32)), x);
                            tab[slot] = ((unsigned long long)__float_as_uint(v) << 32) | tag;
                            pend = false;
                        } else slot = (slot + 1u) & (BW_HASH - 1u);
                    }
                }
            };
            auto score = [&](bool inw, uint32_t d, uint32_t mword, float x) { // all lanes call
                const bool mul = inw && ((mword >> (d & 31u)) & 1u);
                emit(inw && !mul && x > 0.0f && oi_f32_key(x) >= tau, x, d); // (BM25 lists hold scores > 0 only)
                if (__ballot(mul)) accumulate(mul, d, x);
            };
            for_each_round([&](const Round &r) {
                uint32_t mw[BW_ROUND];
#pragma unroll
                for (int k = 0; k < BW_ROUND; ++k) mw[k] = multi[lane < chunk_n(r, k) ? r.p[k].dib >> 5 : 0u]; // reads in flight together
#pragma unroll
                for (int k = 0; k < BW_ROUND; ++k) {
                    if (chunk_n(r, k)) { // uniform
                        const uint32_t d = r.p[k].dib;
                        score(lane < chunk_n(r, k) && d >= lo && d < hi, d, mw[k], __fmul_rn(chunk_w(r, k), r.p[k].impact));
                    }
                }
            });
            for_each_late_chunk([&](const BwPosting &p, bool have, float wt) {
                const uint32_t d = p.dib;
                const bool inw = have && d >= lo && d < hi;
                score(inw, d, multi[inw ? d >> 5 : 0u], __fmul_rn(wt, p.impact));
            });
            if (TIMING) t_acc[4] += stamp() - t_b0; // pass B
            const unsigned long long t_em0 = stamp();
            // ---- the multi docs' sums out of the table; clear it and the maps
            if (n_multi) { // uniform
                uint4 ent[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) ent[k] = tab4[k * 64 + lane];
                clear_table();
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t tg = h ? ent[k].z : ent[k].x;
                        const float v = __uint_as_float(h ? ent[k].w : ent[k].y);
                        emit(tg != 0u && v > 0.0f && oi_f32_key(v) >= tau, v, tg - 1u);
                    }
                }
            }
            clear_maps();
            if (TIMING) t_acc[5] += stamp() - t_em0; // table emission + clears
            lo = hi;
        }
        if (TIMING) t_acc[1] += stamp() - t_task0; // whole task
        if (out_cnt > seg_cap) { *overflow = 1u; out_cnt = seg_cap; } // bug guard: seg_cap >= docs per block
        if (lane == 0) seg_cnt[(uint64_t)q * seg_cnt_stride + block0 + blk] = out_cnt;
    }
    if (TIMING && lane == 0 && timing) {
        for (int i = 0; i < 6; ++i) atomicAdd(&timing[i], t_acc[i]);
        atomicAdd(&timing[6], stamp() - t_kernel0);
        atomicAdd(&timing[7], 1ull);
    }
}

#define BW_SMEM (BW_WAVES * BW_WAVE_LDS + (BW_MAX_Q + 1 + 2 * BW_MAX_Q + BW_STAGE_TERMS) * 4)

uint32_t oi_bm25_wave_pass_queries(void) { return BW_MAX_Q; }

// Queries [q_begin, q_begin + nq) of the batch over doc blocks [block_begin, block_end); `pool` is the view of THESE
// nq queries (entry 0 = query q_begin).  Every (block, query) task writes pool segment `block` of its query
// (pool.seg_cap >= OI_BM25_BLOCK_DOCS: it can never overflow).
int oi_launch_bm25_wave(oi_index *idx, const uint32_t *d_q_terms, const uint32_t *d_q_offsets, uint32_t q_begin,
                        uint32_t nq, const PoolView &pool, uint32_t block_begin, uint32_t block_end) {
    oi_ctx *ctx = idx->ctx;
    if (nq == 0 || idx->n_postings == 0 || idx->n_blocks == 0 || block_end <= block_begin) return OI_OK;
    OI_REQUIRE(nq <= BW_MAX_Q, "bm25 (wave): %u queries in one pass (limit %u)", nq, BW_MAX_Q);
    OI_REQUIRE(pool.seg_cap >= BW_R && pool.n_segs == idx->n_blocks && pool.n_segs <= pool.seg_cnt_stride &&
                   pool.carry_cap + (uint64_t)pool.n_segs * pool.seg_cap <= pool.stride,
               "bm25 (wave): pool geometry mismatch");
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(bm25_wave_kernel<false>), (size_t)BW_SMEM));
    const uint32_t nb = block_end - block_begin;
    const uint64_t n_tasks = (uint64_t)nb * nq;
    OI_REQUIRE(n_tasks < 0xFFFFFFFFull, "bm25 (wave): too many tasks");
    uint64_t wgs = (n_tasks + BW_WAVES - 1) / BW_WAVES;
    const uint64_t max_wgs = 3ull * (uint64_t)ctx->num_cus; // three resident workgroups per CU (LDS: 3 x 52 KiB)
    if (wgs > max_wgs) wgs = max_wgs;
#ifdef OI_ABLATION
    if (oi_ablation_env("OI_BM25_WAVE_TIMING")) { // diagnostic: per-section cycle sums (the stamps and forced waits cost time)
        OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(bm25_wave_kernel<true>), (size_t)BW_SMEM));
        DevBuf &tb = ctx->buf("bm25_wave_timing");
        OI_CHECK(tb.ensure(8 * sizeof(unsigned long long)));
        OI_HIP_CHECK(hipMemsetAsync(tb.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(bm25_wave_kernel<true>, dim3((uint32_t)wgs), dim3(BW_WAVES * 64), BW_SMEM, ctx->stream,
                           reinterpret_cast<const BwPosting *>(idx->postings.p), idx->cell_start.as<uint32_t>(),
                           idx->idf.as<float>(), idx->df_local.as<uint32_t>(), idx->vocab, idx->doc_id_base, block_begin, nb, d_q_terms, d_q_offsets,
                           q_begin, nq, pool.keys, pool.seg_cnt, pool.seg_cnt_stride, pool.tau_keys, pool.stride,
                           pool.carry_cap, pool.seg_cap, pool.overflow, tb.as<unsigned long long>());
        OI_HIP_CHECK(hipGetLastError());
        unsigned long long h[8];
        OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        OI_HIP_CHECK(hipMemcpy(h, tb.p, sizeof(h), hipMemcpyDeviceToHost));
        const double wv = h[7] ? (double)h[7] : 1.0;
        fprintf(stderr, "[bm25 wave timing] blocks %u tasks %llu waves %llu | cycles per wave: staging %.0f bounds-wait %.0f "
                        "(unused) %.0f passA %.0f passB %.0f table+clears %.0f kernel %.0f (bounds-wait column = whole tasks)\n",
                nb, (unsigned long long)n_tasks, h[7], h[0] / wv, h[1] / wv, h[2] / wv, h[3] / wv, h[4] / wv, h[5] / wv, h[6] / wv);
        return OI_OK;
    }
#endif
    ProfScope ps(ctx, "bm25");
#ifdef OI_ABLATION
    if (const char *d = oi_ablation_env("OI_BM25_WAVE_DBG")) {
        const int lvl = atoi(d);
#define OI_BW_DBG(L)                                                                                                     \
    if (lvl == L) {                                                                                                      \
        OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(bm25_wave_kernel<false, L>), (size_t)BW_SMEM));          \
        hipLaunchKernelGGL((bm25_wave_kernel<false, L>), dim3((uint32_t)wgs), dim3(BW_WAVES * 64), BW_SMEM, ctx->stream,  \
                           reinterpret_cast<const BwPosting *>(idx->postings.p), idx->cell_start.as<uint32_t>(),         \
                           idx->idf.as<float>(), idx->df_local.as<uint32_t>(), idx->vocab, idx->doc_id_base, block_begin, nb, d_q_terms, d_q_offsets,  \
                           q_begin, nq, pool.keys, pool.seg_cnt, pool.seg_cnt_stride, pool.tau_keys, pool.stride,        \
                           pool.carry_cap, pool.seg_cap, pool.overflow, (unsigned long long *)nullptr);                 \
        OI_HIP_CHECK(hipGetLastError());                                                                                 \
        return OI_OK;                                                                                                    \
    }
        OI_BW_DBG(1) OI_BW_DBG(2) OI_BW_DBG(3) OI_BW_DBG(4)
#undef OI_BW_DBG
    }
#endif
    hipLaunchKernelGGL(bm25_wave_kernel<false>, dim3((uint32_t)wgs), dim3(BW_WAVES * 64), BW_SMEM, ctx->stream,
                       reinterpret_cast<const BwPosting *>(idx->postings.p), idx->cell_start.as<uint32_t>(),
                       idx->idf.as<float>(), idx->df_local.as<uint32_t>(), idx->vocab, idx->doc_id_base, block_begin, nb, d_q_terms, d_q_offsets,
                       q_begin, nq, pool.keys, pool.seg_cnt, pool.seg_cnt_stride, pool.tau_keys, pool.stride,
                       pool.carry_cap, pool.seg_cap, pool.overflow, (unsigned long long *)nullptr);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}
