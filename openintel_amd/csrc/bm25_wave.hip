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
//   load     the task's postings are cut, run by run in query order, into groups of up to 256 (four per lane); lane g
//            holds group g's descriptor, built by one scalar loop over the TERMS; the first six groups (1536
//            postings: the whole task in the common case) are loaded together, up front, and stay in registers for
//            both passes; further groups stream one ahead;
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
// A task with more multi-doc postings than the table takes (BW_MULTI_CAP = 384) is cut into doc-id windows (halved until they fit, down
// to 2048 docs; below that the window's docs get dense accumulators instead -- at most 16 windows per task whatever the
// query repeats): exact for any data, one window in the common case.  (Two earlier forms, both measured: an open-addressing table for
// ALL docs -- a wave ran as slow as its longest probe chain, 26K cycles per task -- and a bitmap + rank perfect hash
// with a dense accumulator array -- 65K cycles per task, instruction-bound on 32 fixed register slots per lane, the
// rank sweep and a read-modify-write per posting.  DESIGN.md 4.3 has the ladders.)  HBM-bound by construction (8 B
// per posting + 8 B per (block, term) bounds lookup); at this batch size the floor is launch and latency, not bytes.
#include <algorithm>

#include "oi_device.h"
#include "oi_internal.h"

#define BW_R OI_BM25_BLOCK_DOCS
#define BW_WORDS (BW_R / 32)      // 1024 words per bitmap
#define BW_HASH_BITS 9
#define BW_HASH (1u << BW_HASH_BITS) // entries {tag = doc-in-block + 1, f32 score bits} of the multi-doc table (a multiple of 128)
#define BW_MULTI_CAP (BW_HASH * 3u / 4u) // multi-doc postings a window may hold (load factor <= 0.75)
#define BW_TAB_STEPS (BW_HASH / 128u)    // 16-byte pieces of the table per lane
#define BW_DENSE_W 2048u          // docs per window in dense mode (2048 f32 accumulators = the two maps' 8 KiB)
#define BW_WAVES 4                // waves per workgroup
#define BW_STAGE_TERMS 256u       // query terms of a pass staged in LDS (more: read from global memory)
#define BW_MAX_Q 128u             // queries per pass (term offsets, weights and order staged in LDS)
#define BW_KEEP 6                  // groups of 256 postings loaded up front and kept in registers for both passes
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
    const uint32_t *__restrict__ df, uint32_t vocab, uint32_t n_win, uint32_t doc_id_base, uint32_t block0, uint32_t n_blocks_here,
    const uint32_t *__restrict__ q_terms, const uint32_t *__restrict__ q_offsets, uint32_t q_begin, uint32_t nq,
    uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride, const uint32_t *tau_keys, uint64_t pool_stride,
    uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow, unsigned long long *timing) {
    // TIMING (an -DOI_ABLATION diagnostic build only): per-section cycle sums of every wave into timing[0..7]
    unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0};
    auto stamp = [&]() -> unsigned long long { return TIMING ? __builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long t_kernel0 = stamp();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // [BW_WAVES][seen u32 x 1024 | multi u32 x 1024 | table u64 x BW_HASH] | q_off[MAX_Q + 1] | order[MAX_Q] | weight[MAX_Q] | terms[STAGE]
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
    uint4 *tab4 = reinterpret_cast<uint4 *>(tab);   // BW_HASH x 8 B = BW_TAB_STEPS x 64 x 16 B
    auto clear_maps = [&]() {
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) maps4[k * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
    };
    auto clear_table = [&]() {
#pragma unroll
        for (uint32_t k = 0; k < BW_TAB_STEPS; ++k) tab4[k * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
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
        const uint64_t cell = (uint64_t)term * n_win + 2u * (block0 + blk); // windows 2 blk, 2 blk + 1 of the term's list
        s = cell_start[cell];
        e = cell_start[cell + 2];
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

        // ---- the task's postings as GROUPS of up to 256 (four per lane) IN QUERY ORDER; a group never spans two runs.
        // Lane g holds the descriptor of group g (run start of the group, run end, term): built by one uniform loop over the
        // terms -- scalar work per TERM, not per chunk (the first form walked a scalar cursor per 64 postings: ~1500 SALU
        // instructions per task, as many as the vector work; PMC in DESIGN.md 4.3).
        const uint32_t ntf = nt < 64u ? nt : 64u;
        uint32_t d_i0 = 0, d_e = 0, d_j = 0; // this lane's group of the current PAGE (page p: groups 64 p .. 64 p + 63)
        uint32_t ng = 0;                      // groups of the first 64 terms (uniform)
        const uint32_t n_mine = (my_e - my_s + 255u) >> 8; // groups of the term this lane holds the bounds of
        auto build_page = [&](uint32_t page) {
            d_i0 = 0; d_e = 0; d_j = 0;
            const uint32_t gl = 64u * page + lane; // the group this lane describes
            uint32_t acc = 0;
            for (uint32_t j = 0; j < ntf; ++j) {
                const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)n_mine, (int)j);
                if (nj == 0u) continue; // uniform
                const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)my_s, (int)j);
                const uint32_t ej = (uint32_t)__builtin_amdgcn_readlane((int)my_e, (int)j);
                const bool mine = gl >= acc && gl < acc + nj;
                d_i0 = mine ? sj + ((gl - acc) << 8) : d_i0;
                d_e = mine ? ej : d_e;
                d_j = mine ? j : d_j;
                acc += nj;
            }
            ng = acc;
        };
        build_page(0);
        struct Group {
            BwPosting p[4];
            uint32_t n; // postings in the group (uniform, 0..256)
            float w;    // idf of its term (uniform)
        };
        // group `l` of the current page (l uniform; a compile-time constant for the kept groups of page 0); past the end: n = 0
        auto load_group = [&](uint32_t page, uint32_t l) {
            Group G;
            const bool have = l < 64u && 64u * page + l < ng;
            const uint32_t gl = have ? l : 0u;
            const uint32_t i0 = (uint32_t)__builtin_amdgcn_readlane((int)d_i0, (int)gl);
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)d_e, (int)gl);
            const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)d_j, (int)gl);
            G.n = have ? (e - i0 < 256u ? e - i0 : 256u) : 0u;
            G.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), (int)tj));
            const uint32_t safe = have ? e - 1u : 0u; // unpredicated loads: a lane past the group re-reads its run's last posting
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + 64u * u + lane;
                G.p[u] = postings[have && i < e ? i : safe];
            }
            return G;
        };
        // the first BW_KEEP groups: all their loads in flight together, kept in registers for BOTH passes
        Group keep[BW_KEEP];
#pragma unroll
        for (int g = 0; g < BW_KEEP; ++g) keep[g] = load_group(0u, (uint32_t)g);
        fetch_bounds(t + G, 0, nx_s, nx_e, nx_w, nx_ok); // the next task's bounds: in flight while this one is processed
        if (DBG == 1 || DBG == 2) {
            uint32_t x = my_s + my_e;
            if (DBG == 2)
                for (int g = 0; g < BW_KEEP; ++g)
                    for (int u = 0; u < 4; ++u) x += (keep[g].p[u].dib ^ __float_as_uint(keep[g].p[u].impact)) + keep[g].n;
            if (lane == 0 || x == 0xDEADBEEFu) seg_cnt[(uint64_t)q * seg_cnt_stride + block0 + blk] = x == 0xDEADBEEFu ? 1u : 0u;
            continue;
        }
        // Every group of the first 64 terms in query order: f(group).  Kept groups come out of registers; the rest streams one
        // group ahead, page by page (a task of more than 64 groups -- 16K postings -- rebuilds the lanes' descriptors per
        // page).  `stop()`: uniform predicate checked between groups -- pass A gives up as soon as the window has failed.
        const uint32_t n_pages = (ng + 63u) >> 6;
        auto for_each_group = [&](auto &&f, auto &&stop) {
#pragma unroll
            for (int g = 0; g < BW_KEEP; ++g) {
                // The kept postings are made opaque before each use: otherwise everything pass A derives from them (word
                // address, bit, window test, product -- ~40 registers per group) is kept alive for pass B instead of being
                // recomputed (+45 VGPRs per kept group: spills at four groups).
#pragma unroll
                for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(keep[g].p[u].dib), "+v"(keep[g].p[u].impact));
                if ((uint32_t)g < ng) f(keep[g]); // uniform
            }
            if (ng <= BW_KEEP) return;
            for (uint32_t page = 0; page < n_pages; ++page) {
                if (n_pages > 1u) build_page(page);
                const uint32_t l0 = page == 0u ? BW_KEEP : 0u;
                const uint32_t l1 = ng - 64u * page < 64u ? ng - 64u * page : 64u;
                Group ga = load_group(page, l0);
                for (uint32_t l = l0; l < l1; ++l) {
                    if (stop()) { if (n_pages > 1u) build_page(0); return; }
                    const Group gb = load_group(page, l + 1u);
                    f(ga);
                    ga = gb;
                }
            }
            if (n_pages > 1u) build_page(0);
        };
        // slow form (terms beyond the 64th of a long query): f(posting of this lane, lane holds one, idf weight), one chunk at a time
        auto for_each_slow_chunk = [&](auto &&f, auto &&stop) {
            auto run = [&](uint32_t s, uint32_t e, float wt) {
                for (uint32_t i0 = s; i0 < e && !stop(); i0 += 64u) f(postings[i0 + lane < e ? i0 + lane : e - 1u], i0 + lane < e, wt);
            };
            for (uint32_t g0 = 64; g0 < nt; g0 += 64) { // terms beyond the 64 a wave holds bounds for
                uint32_t gs, ge, okm;
                float gw;
                fetch_bounds(t, g0, gs, ge, gw, okm);
                gs = okm ? gs : 0u; ge = okm ? ge : 0u; gw = okm ? gw : 0.f;
                const uint32_t gn = nt - g0 < 64u ? nt - g0 : 64u;
                for (uint32_t j = 0; j < gn; ++j)
                    run((uint32_t)__shfl((int)gs, (int)j, OI_WAVE), (uint32_t)__shfl((int)ge, (int)j, OI_WAVE), __shfl(gw, (int)j, OI_WAVE));
            }
        };
        uint32_t out_cnt = 0;
        auto emit = [&](bool keepit, float v, uint32_t dib) { // all lanes call; appends the kept scores to the segment
            const unsigned long long m = __ballot(keepit);
            if (m) {
                if (keepit) {
                    const uint32_t pos = out_cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (pos < seg_cap) seg[pos] = oi_rank_key(v, doc0 + dib);
                }
                out_cnt += (uint32_t)__popcll(m);
            }
        };

        auto never = []() { return false; };
        uint32_t width = BW_R, lo = 0;
        while (lo < BW_R) {
            const uint32_t hi = lo + width;
            const unsigned long long t_a0 = stamp();
            // ---- pass A: seen / multi maps.  A group's four atomics are issued, THEN their results are used; one ballot
            // per group decides whether anything of it was seen before (rare)
            uint32_t n_multi = 0;
            for_each_group([&](const Group &g) {
                uint32_t oldw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t d = g.p[u].dib;
                    const bool inw = 64u * u + lane < g.n && d >= lo && d < hi;
                    oldw[u] = 0u;
                    if (inw) oldw[u] = atomicOr(&seen[d >> 5], 1u << (d & 31u));
                }
                bool any = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) any = any || ((oldw[u] >> (g.p[u].dib & 31u)) & 1u); // (not in window: oldw = 0)
                if (__ballot(any)) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t d = g.p[u].dib;
                        const bool again = (oldw[u] >> (d & 31u)) & 1u;
                        if (again) atomicOr(&multi[d >> 5], 1u << (d & 31u));
                        n_multi += (uint32_t)__popcll(__ballot(again));
                    }
                }
            }, [&]() { return n_multi > BW_MULTI_CAP; });
            for_each_slow_chunk([&](const BwPosting &p, bool have, float) {
                const uint32_t d = p.dib, bit = 1u << (d & 31u);
                const bool inw = have && d >= lo && d < hi;
                uint32_t o = 0u;
                if (inw) o = atomicOr(&seen[d >> 5], bit);
                const bool again = (o & bit) != 0u;
                const unsigned long long m = __ballot(again);
                if (m) {
                    if (again) atomicOr(&multi[d >> 5], bit);
                    n_multi += (uint32_t)__popcll(m);
                }
            }, [&]() { return n_multi > BW_MULTI_CAP; });
            if (TIMING) t_acc[3] += stamp() - t_a0; // pass A
            if (DBG == 3) { clear_maps(); lo = hi; continue; }
            if (n_multi > BW_MULTI_CAP) { // too many multi-doc postings for the table: clear, halve the window, again
                clear_maps();
                if (width > BW_DENSE_W) { width >>= 1; continue; }
                // Still too many at 2048 docs per window (a query that repeats a frequent term many times makes every
                // doc of it "multi"): halving further would re-scan the task's runs once per handful of docs.  DENSE mode
                // instead, bounded at 16 windows per task: the map region becomes 2048 f32 accumulators indexed by
                // (doc - lo); every in-window posting is added in query order (a group is one run: distinct docs, so a
                // group's four read-modify-writes are independent; a wave's LDS operations execute in program order).
                float *dacc = reinterpret_cast<float *>(seen); // 8 KiB = both maps, all zero here (+0.0f)
                auto dense_add = [&](bool inw, uint32_t d, float x) {
                    const uint32_t a = inw ? d - lo : 0u;
                    const float v = dacc[a];
                    if (inw) dacc[a] = __fadd_rn(v, x);
                };
                for_each_group([&](const Group &g) {
                    float v[4];
                    bool inw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t d = g.p[u].dib;
                        inw[u] = 64u * u + lane < g.n && d >= lo && d < hi;
                        v[u] = dacc[inw[u] ? d - lo : 0u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (inw[u]) dacc[g.p[u].dib - lo] = __fadd_rn(v[u], __fmul_rn(g.w, g.p[u].impact));
                }, never);
                for_each_slow_chunk([&](const BwPosting &p, bool have, float wt) {
                    dense_add(have && p.dib >= lo && p.dib < hi, p.dib, __fmul_rn(wt, p.impact));
                }, never);
                for (uint32_t i = lane; i < BW_DENSE_W; i += 64) {
                    const float v = dacc[i];
                    emit(v > 0.0f && oi_f32_key(v) >= tau, v, lo + i);
                }
                clear_maps();
                lo = hi;
                continue;
            }
            const unsigned long long t_b0 = stamp();
            // ---- pass B: in query order.  Single-run docs are finished scores; multi docs go through the table.
            auto accumulate = [&](bool mul, uint32_t d, float x) { // lanes with mul: table[d] += x, in call order
                if (DBG == 4) return;
                bool pend = mul;
                uint32_t slot = (d * 0x9E3779B1u) >> (32 - BW_HASH_BITS);
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
            for_each_group([&](const Group &g) {
                uint32_t mw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) mw[u] = multi[64u * u + lane < g.n ? g.p[u].dib >> 5 : 0u]; // reads in flight together
                bool mul[4], kp[4], anym = false, anyk = false;
                float x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t d = g.p[u].dib;
                    const bool inw = 64u * u + lane < g.n && d >= lo && d < hi;
                    mul[u] = inw && ((mw[u] >> (d & 31u)) & 1u);
                    x[u] = __fmul_rn(g.w, g.p[u].impact);
                    kp[u] = inw && !mul[u] && x[u] > 0.0f && oi_f32_key(x[u]) >= tau; // (BM25 lists hold scores > 0 only)
                    anym = anym || mul[u];
                    anyk = anyk || kp[u];
                }
                if (__ballot(anyk)) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) emit(kp[u], x[u], g.p[u].dib);
                }
                if (DBG != 4 && __ballot(anym)) {
                    // the four chunks of a group are one run: their docs are distinct, so their table updates are independent --
                    // all four compare-and-swaps are in flight together, only stragglers (slot taken by another doc) loop
                    bool pend[4];
                    uint32_t slot[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { pend[u] = mul[u]; slot[u] = (g.p[u].dib * 0x9E3779B1u) >> (32 - BW_HASH_BITS); }
                    for (;;) {
                        unsigned long long prev[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            prev[u] = 0ull;
                            if (pend[u]) prev[u] = atomicCAS(&tab[slot[u]], 0ull, ((unsigned long long)__float_as_uint(x[u]) << 32) | (g.p[u].dib + 1u));
                        }
                        bool more = false;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (pend[u]) {
                                const uint32_t tag = g.p[u].dib + 1u;
                                if (prev[u] == 0ull) pend[u] = false; // first run of this doc: (+0) + x
                                else if ((uint32_t)prev[u] == tag) {   // seen in an earlier run: add in query order, plain store
                                    const float v = __fadd_rn(__uint_as_float((uint32_t)(prev[u] >> 32)), x[u]);
                                    tab[slot[u]] = ((unsigned long long)__float_as_uint(v) << 32) | tag;
                                    pend[u] = false;
                                } else { slot[u] = (slot[u] + 1u) & (BW_HASH - 1u); more = true; }
                            }
                        if (!__ballot(more)) break;
                    }
                }
            }, never);
            for_each_slow_chunk([&](const BwPosting &p, bool have, float wt) {
                const uint32_t d = p.dib;
                const bool inw = have && d >= lo && d < hi;
                const uint32_t mword = multi[inw ? d >> 5 : 0u];
                const bool mul = inw && ((mword >> (d & 31u)) & 1u);
                const float x = __fmul_rn(wt, p.impact);
                emit(inw && !mul && x > 0.0f && oi_f32_key(x) >= tau, x, d);
                if (__ballot(mul)) accumulate(mul, d, x);
            }, never);
            if (TIMING) t_acc[4] += stamp() - t_b0; // pass B
            const unsigned long long t_em0 = stamp();
            // ---- the multi docs' sums out of the table; clear it and the maps
            if (n_multi) { // uniform
                uint4 ent[BW_TAB_STEPS];
#pragma unroll
                for (uint32_t k = 0; k < BW_TAB_STEPS; ++k) ent[k] = tab4[k * 64 + lane];
                clear_table();
#pragma unroll
                for (uint32_t k = 0; k < BW_TAB_STEPS; ++k) {
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
    // three resident workgroups per CU (LDS: 3 x 52 KiB).  A 128-entry table (39 KiB: four per CU) was measured and is slower,
    // 0.209 vs 0.137 ms per batch: tasks with more than 96 multi-doc postings are common and each costs a rescan of a halved window
    uint64_t max_wgs = 3ull * (uint64_t)ctx->num_cus;
    if (const char *e = oi_ablation_env("OI_BM25_WAVE_WGS")) max_wgs = std::max(1, atoi(e)) * (uint64_t)ctx->num_cus; // A/B: resident workgroups per CU
    if (wgs > max_wgs) wgs = max_wgs;
#ifdef OI_ABLATION
    if (oi_ablation_env("OI_BM25_WAVE_TIMING")) { // diagnostic: per-section cycle sums (the stamps and forced waits cost time)
        OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(bm25_wave_kernel<true>), (size_t)BW_SMEM));
        DevBuf &tb = ctx->buf("bm25_wave_timing");
        OI_CHECK(tb.ensure(8 * sizeof(unsigned long long)));
        OI_HIP_CHECK(hipMemsetAsync(tb.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(bm25_wave_kernel<true>, dim3((uint32_t)wgs), dim3(BW_WAVES * 64), BW_SMEM, ctx->stream,
                           reinterpret_cast<const BwPosting *>(idx->postings.p), idx->cell_start.as<uint32_t>(),
                           idx->idf.as<float>(), idx->df_local.as<uint32_t>(), idx->vocab, idx->n_win, idx->doc_id_base, block_begin, nb, d_q_terms, d_q_offsets,
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
                           idx->idf.as<float>(), idx->df_local.as<uint32_t>(), idx->vocab, idx->n_win, idx->doc_id_base, block_begin, nb, d_q_terms, d_q_offsets,  \
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
                       idx->idf.as<float>(), idx->df_local.as<uint32_t>(), idx->vocab, idx->n_win, idx->doc_id_base, block_begin, nb, d_q_terms, d_q_offsets,
                       q_begin, nq, pool.keys, pool.seg_cnt, pool.seg_cnt_stride, pool.tau_keys, pool.stride,
                       pool.carry_cap, pool.seg_cap, pool.overflow, (unsigned long long *)nullptr);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}
