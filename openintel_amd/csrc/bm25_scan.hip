// bm25_scan.hip -- BM25 for a query BATCH as one coalesced HBM scan of the forward index.
//
// Builder-defined (the reference has no BM25); same score definition, bit for bit, as bm25.hip and
// the CPU oracle: score(q, d) = sum over q's terms in query order of idf_t * (tf*(k1+1))/(tf+Kd).
//
// Why a second BM25 kernel.  The term-at-a-time kernel (bm25.hip) touches only the postings of the
// query terms (115 MB for 64 queries at 10M docs) but is bound by its ~20 K (block, query) tasks of
// dependent steps, far from any roofline.  For a batch, the union of the query terms (256 of 131072
// here) makes a different trade attractive: stream the WHOLE forward index (4 B per token, 1.15 GB at
// 10M docs) once per batch at HBM speed and test every token against the batch's term set in LDS.
// Algorithmic bytes per batch: 4*T + 8*N (tokens + doc offsets); HBM-bound by construction.
//
//   phase 1  token-parallel: every lane loads 8 consecutive tokens (two 16-byte loads, fully
//            coalesced; the next step's loads are issued before this step is processed) and tests each
//            against a 4 KiB Bloom filter of the batch's terms in LDS (one LDS read per token).  The ~5 %
//            that pass are resolved to their term slot (LDS hash table; Bloom false positives die
//            here) and compacted per wave into the tile's hit array as (slot, token index).
//   link     one lane per hit: the doc by binary search in the tile's LDS-staged offsets, and a push
//            on that doc's LDS linked list (atomicExch on the list head).
//   phase 2  one lane per DOC: the doc's list is folded once into (slot, tf) pairs in registers; every
//            query using one of those terms is scored exactly once (f32 sum over its terms IN QUERY
//            ORDER, impact from tf and the doc length); survivors of the per-query threshold go to the
//            workgroup's private pool segment (LDS fill counters, no global atomics).
#include <cstdlib>

#include "oi_device.h"
#include "oi_internal.h"

#define BS_THREADS 512
#define BS_WAVES (BS_THREADS / 64)
#define BS_WD 64            // docs per wave tile, at most (the setup kernel picks the tile size per batch)
#define BS_TPT 4            // tokens per lane and step in phase 1: one 16-byte load
#define BS_STEP (64 * BS_TPT)
#define BS_PRE 6            // steps of a tile loaded before the first is scanned (a typical tile has 5)
#define BS_HASH 2048        // term hash table slots (power of two)
#define BS_MAX_TERMS 1024   // distinct batch terms (load factor <= 50 %)
#define BS_MAX_Q 256        // queries per pass
#define BS_MAX_QT 2048      // total (query, term) pairs per pass
#define BS_WAVE_HITS 192    // hit entries per wave tile (typ. ~55; ~200 at 256 queries); beyond: exact per-doc fallback
#define BS_BLOOM_WORDS 1024  // 4 KiB: <= 1024 keys in 32768 bits -> < 3 % false positives
#define BS_NIL 0xFFFFu
#define BS_LONGQ 0xFFFEu
#define BS_TF_SLOTS 512     // (doc, term) -> tf table of a wave tile (more slots than BS_WAVE_HITS)
#define BS_TF_EMPTY 0xFFFFFFFFu
#define BS_K1 1.2f
#define BS_B 0.75f

struct BsBatch { // built once per batch by bm25_scan_setup (global memory, ~40 KB)
    uint32_t key[BS_HASH];          // term id or 0xFFFFFFFF
    float idf[BS_HASH];             // idf of the term in that slot
    uint32_t users_off[BS_HASH + 1]; // CSR over slots: which (query, position) pairs use the term
    uint32_t users[BS_MAX_QT];      // query << 16 | position
    uint32_t q_off[BS_MAX_Q + 1];   // CSR over queries: their terms' slots in query order
    uint32_t q_slot[BS_MAX_QT];     // slot or 0xFFFF (term outside the vocabulary: contributes nothing)
    uint32_t bloom[BS_BLOOM_WORDS]; // 32768-bit filter over the batch's term ids (1 hash)
    // queries of at most 4 terms, ready for registers: their slots (0xFFFF: no term) and idfs in query
    // order; qp_slots[q][0] = BS_LONGQ marks a longer query (scored from q_slot / idf instead)
    uint16_t qp_slots[BS_MAX_Q][4];
    float qp_idf[BS_MAX_Q][4];
    uint32_t n_queries, n_pairs, error;
    uint32_t docs_per_tile; // chosen from the batch's expected hits per doc, see bm25_scan_setup
};
__device__ __forceinline__ uint32_t bs_bloom_bit(uint32_t t) { return (t * 0x85EBCA77u) >> 17; } // 15 bits

__device__ __forceinline__ uint32_t bs_hash(uint32_t t) { return (t * 0x9E3779B1u) >> 21; } // 11 bits

// ------------------------------------------------------------------ batch setup (one workgroup)
__global__ __launch_bounds__(1024) void bm25_scan_setup(const uint32_t *q_terms, const uint32_t *q_offsets,
                                                         uint32_t q_begin, uint32_t n_queries, uint32_t vocab,
                                                         uint32_t max_terms_per_query, const float *idf,
                                                         const uint32_t *df, uint64_t n_docs, float avgdl,
                                                         BsBatch *out) {
    __shared__ uint32_t cnt[BS_HASH];
    __shared__ uint32_t scan_tmp[1024];
    __shared__ uint32_t n_distinct;
    __shared__ unsigned long long sum_df; // docs holding each distinct batch term, summed: ~ hits of the batch
    const uint32_t tid = threadIdx.x;
    const uint32_t base = q_offsets[q_begin];
    const uint32_t n_pairs = q_offsets[q_begin + n_queries] - base;
    for (uint32_t i = tid; i < BS_HASH; i += 1024) { out->key[i] = 0xFFFFFFFFu; out->idf[i] = 0.f; cnt[i] = 0; }
    for (uint32_t i = tid; i < BS_BLOOM_WORDS; i += 1024) out->bloom[i] = 0;
    if (tid == 0) { n_distinct = 0; sum_df = 0; out->docs_per_tile = BS_WD; out->n_queries = n_queries; out->n_pairs = n_pairs; out->error = 0; }
    for (uint32_t q = tid; q <= n_queries; q += 1024) out->q_off[q] = q_offsets[q_begin + q] - base;
    __syncthreads();
    __shared__ uint32_t too_long;
    if (tid == 0) too_long = 0;
    __syncthreads();
    for (uint32_t q = tid; q < n_queries; q += 1024)
        if (q_offsets[q_begin + q + 1] - q_offsets[q_begin + q] > max_terms_per_query) too_long = 1;
    __syncthreads();
    // the pass size was chosen as 1024 / max_terms_per_query queries, so these hold unless a query
    // has more terms than the index was told to expect
    if (too_long || n_pairs > BS_MAX_TERMS || n_queries > BS_MAX_Q) { if (tid == 0) out->error = 1; return; }
    // insert every (query, position) pair's term; the hash POSITION is the term's slot id
    for (uint32_t p = tid; p < n_pairs; p += 1024) {
        const uint32_t t = q_terms[base + p];
        uint32_t slot = BS_NIL;
        if (t < vocab) {
            uint32_t h = bs_hash(t);
            for (;;) {
                const uint32_t prev = atomicCAS(&out->key[h], 0xFFFFFFFFu, t);
                if (prev == 0xFFFFFFFFu) {
                    atomicAdd(&n_distinct, 1u);
                    atomicAdd(&sum_df, (unsigned long long)df[t]);
                    out->idf[h] = idf[t];
                    const uint32_t bb = bs_bloom_bit(t);
                    atomicOr(&out->bloom[bb >> 5], 1u << (bb & 31u));
                    slot = h;
                    break;
                }
                if (prev == t) { slot = h; break; }
                h = (h + 1) & (BS_HASH - 1);
            }
            atomicAdd(&cnt[slot], 1u);
        }
        out->q_slot[p] = slot;
    }
    __syncthreads();
    if (n_distinct > BS_MAX_TERMS) { if (tid == 0) out->error = 2; return; }
    if (tid == 0) {
        // Tile size (docs per WAVE tile).  Phase 1 works in steps of BS_STEP tokens and the hit passes in
        // rounds of 64 hits, both rounded up per tile, so the tile that wastes least depends on the batch:
        // pick the docs per tile minimising (steps + 4 * hit rounds) / docs (roughly the cost ratio of a
        // hit round -- link + score -- to a token step).
        const float hpd = n_docs ? 1.05f * (float)sum_df / (float)n_docs : 0.f; // expected hits per doc
        float best = 3.4e38f;
        uint32_t best_d = 8;
        for (uint32_t d = 8; d <= BS_WD; d += 4) {
            const float steps = ceilf((float)d * avgdl / (float)BS_STEP);
            const float rounds = ceilf(fmaxf((float)d * hpd * 1.1f, 1.f) / 64.f);
            const float cost = (steps + 4.f * rounds) / (float)d;
            if (d > 8 && (float)d * hpd * 1.5f > (float)BS_WAVE_HITS) break; // the tile's hits must fit its hit array
            if (cost <= best) { best = cost; best_d = d; }
        }
        out->docs_per_tile = best_d;
    }
    // exclusive scan of cnt[2048] (two entries per thread)
    const uint32_t a = cnt[2 * tid], b = cnt[2 * tid + 1];
    scan_tmp[tid] = a + b;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t v = tid >= o ? scan_tmp[tid - o] : 0u;
        __syncthreads();
        scan_tmp[tid] += v;
        __syncthreads();
    }
    const uint32_t excl = scan_tmp[tid] - (a + b);
    out->users_off[2 * tid] = excl;
    out->users_off[2 * tid + 1] = excl + a;
    if (tid == 1023) out->users_off[BS_HASH] = scan_tmp[1023];
    cnt[2 * tid] = excl;       // reuse as fill cursors
    cnt[2 * tid + 1] = excl + a;
    __syncthreads();
    // users: filled in pair order per query so that a term repeated in one query keeps its positions
    // ascending (the order inside a slot's list is otherwise irrelevant)
    for (uint32_t q = tid; q < n_queries; q += 1024) {
        const uint32_t lo = out->q_off[q], hi = out->q_off[q + 1];
        for (uint32_t p = lo; p < hi; ++p) {
            const uint32_t slot = out->q_slot[p];
            if (slot != BS_NIL) out->users[atomicAdd(&cnt[slot], 1u)] = (q << 16) | (p - lo);
        }
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t slot = lo + j < hi ? out->q_slot[lo + j] : BS_NIL;
            out->qp_slots[q][j] = (uint16_t)slot;
            out->qp_idf[q][j] = slot != BS_NIL ? out->idf[slot] : 0.f;
        }
        if (hi - lo > 4) out->qp_slots[q][0] = (uint16_t)BS_LONGQ;
    }
}

// ------------------------------------------------------------------ the scan
// Every WAVE is autonomous: it owns tiles of up to 64 consecutive docs (one contiguous token range),
// scans, links and scores them out of its own slice of LDS, and never meets a workgroup barrier after
// the batch tables are staged.  A workgroup is only eight such waves sharing those read-only tables and
// one pool segment.
struct BsWave {
    uint32_t off[BS_WD + 1];       // token offsets of the tile's docs, relative to the tile's first token
    uint32_t tf[BS_TF_SLOTS];      // doc << 11 | slot in the low 20 bits, tf above; BS_TF_EMPTY when free
    uint32_t hit[BS_WAVE_HITS];    // slot | token index << 11 (phase 1), then slot | doc << 11 | representative << 20
    uint32_t overflow_tile, pad[2];
};
struct BsShared {
    uint32_t key[BS_HASH];
    uint32_t users[BS_MAX_QT];
    uint32_t q_off[BS_MAX_Q + 1];
    uint32_t tau[BS_MAX_Q];
    uint32_t seg_fill[BS_MAX_Q];
    uint32_t bloom[BS_BLOOM_WORDS];
    alignas(16) float qp_idf[BS_MAX_Q][4];
    alignas(8) uint16_t qp_slots[BS_MAX_Q][4];
    uint16_t users_off[BS_HASH + 2];
    uint16_t q_slot[BS_MAX_QT];
    BsWave wave[BS_WAVES];
};
static_assert(sizeof(BsShared) <= 80 * 1024, "two 512-thread workgroups per CU");

__device__ unsigned long long bs_timing[8]; // development aid (OI_BM25_SCAN_DBG=9): cycles per phase, summed over sampled waves

__device__ __forceinline__ uint32_t bs_tf_hash(uint32_t key20) { return (key20 * 0x9E3779B1u) >> 23; } // 9 bits

// tf of (doc, slot) in the wave tile's table, 0 if the doc does not hold the term
__device__ __forceinline__ uint32_t bs_tf_lookup(const BsWave &w, uint32_t key20) {
    for (uint32_t h = bs_tf_hash(key20);; h = (h + 1) & (BS_TF_SLOTS - 1)) {
        const uint32_t v = w.tf[h];
        if (v == BS_TF_EMPTY) return 0;
        if ((v & 0xFFFFFu) == key20) return v >> 20;
    }
}

// Between a wave's phases.  The hardware keeps one wave's LDS operations in order, so what the wave
// wrote is what it reads next; this only stops the compiler from moving or caching LDS accesses across
// the phase boundary.  (Deliberately not a fence: a fence would also wait for the global loads in
// flight -- the next tile's tokens -- and undo the software pipeline.)
__device__ __forceinline__ void bs_wave_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

template <bool TM> // TM: per-phase cycle counters (development aid, OI_BM25_SCAN_DBG=9)
__global__ __launch_bounds__(BS_THREADS) void bm25_scan_kernel(
    const uint32_t *__restrict__ terms, const uint64_t *__restrict__ doc_offsets, uint64_t doc_begin,
    uint64_t doc_end, float avgdl, const BsBatch *__restrict__ batch, const uint32_t *tau_keys, uint32_t q_begin,
    uint32_t doc_id_base, uint64_t *pools, uint32_t *seg_cnt, uint32_t seg_cnt_stride, uint64_t pool_stride,
    uint32_t carry_cap, uint32_t seg_cap, uint32_t *overflow, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    BsShared &s = *reinterpret_cast<BsShared *>(smem_raw);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (batch->error) { // a query broke the pass limits (see oi_index_set_max_query_terms): fail loudly
        if (tid == 0) *overflow = 1u;
        return;
    }
    const uint32_t nq = batch->n_queries;
    // ---- batch tables -> LDS (once per workgroup)
    for (uint32_t i = tid; i < BS_HASH; i += BS_THREADS) {
        s.key[i] = batch->key[i];
        s.users_off[i] = (uint16_t)batch->users_off[i];
    }
    if (tid == 0) s.users_off[BS_HASH] = (uint16_t)batch->users_off[BS_HASH];
    for (uint32_t i = tid; i < BS_BLOOM_WORDS; i += BS_THREADS) s.bloom[i] = batch->bloom[i];
    for (uint32_t i = tid; i < batch->n_pairs; i += BS_THREADS) { s.users[i] = batch->users[i]; s.q_slot[i] = (uint16_t)batch->q_slot[i]; }
    for (uint32_t i = tid; i <= nq; i += BS_THREADS) s.q_off[i] = batch->q_off[i];
    for (uint32_t i = tid; i < nq * 4; i += BS_THREADS) {
        s.qp_slots[i >> 2][i & 3] = batch->qp_slots[i >> 2][i & 3];
        s.qp_idf[i >> 2][i & 3] = batch->qp_idf[i >> 2][i & 3];
    }
    for (uint32_t i = tid; i < nq; i += BS_THREADS) { s.tau[i] = tau_keys ? tau_keys[q_begin + i] : 0u; s.seg_fill[i] = 0; }
    __syncthreads();

    BsWave &w = s.wave[wv];
    uint64_t *my_seg = pools + (uint64_t)q_begin * pool_stride + carry_cap + (uint64_t)blockIdx.x * seg_cap;
    const uint32_t D = batch->docs_per_tile; // docs per wave tile for this batch (<= BS_WD)
    const uint64_t n_tiles = (doc_end - doc_begin + D - 1) / D;
    const uint64_t tile_stride = (uint64_t)gridDim.x * BS_WAVES;

    // Software pipeline over the wave's tiles: while a tile is linked and scored out of LDS, the next
    // tile's tokens are already on their way into registers, and the offsets of the tile after that are
    // being fetched -- the dependent HBM round trips (offsets, then tokens) are off a tile's path.
    struct TileMeta {
        uint64_t t0, tend, mine; // first token of the tile, one past its last; this lane's doc offset (tend past the last doc)
        uint32_t nd;             // docs in the tile; 0 past the last tile
    };
    // No branch around the loads (past the last tile the last tile is re-read and nd = 0 marks it void),
    // and nothing is computed from the loaded values here: a wait would otherwise be placed at the end of
    // the branch, i.e. a whole HBM round trip inside the tile's setup.
    auto fetch_meta = [&](uint64_t tile) {
        TileMeta m;
        const uint64_t tl = tile < n_tiles ? tile : n_tiles - 1;
        const uint64_t d0 = doc_begin + tl * D;
        const uint32_t nd = (uint32_t)((doc_end - d0) < D ? (doc_end - d0) : D);
        // vector loads on purpose (a per-lane zero the compiler cannot see through): scalar loads would
        // share lgkmcnt with LDS and make the next LDS wait sit out an HBM round trip
        uint32_t z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        m.t0 = doc_offsets[d0 + z];
        m.tend = doc_offsets[d0 + nd + z];
        m.mine = doc_offsets[d0 + (lane < nd ? lane : nd)];
        m.nd = tile < n_tiles ? nd : 0u;
        return m;
    };
    auto meta_ntok = [](const TileMeta &m) { return m.nd ? (uint32_t)(m.tend - m.t0) : 0u; };
    // a tile's token window: loads are never predicated -- a group past the tile's end re-reads the
    // tile's last group and its tokens fail the range test in scan_step
    auto load_step = [&](const TileMeta &m, uint32_t sb, uint32_t (&tk)[BS_TPT]) {
        const uint32_t headpad = (uint32_t)(m.t0 & 3u);
        const uint32_t span = meta_ntok(m) + headpad;
        const uint32_t last_g = span ? ((span - 1u) & ~3u) : 0u;
        const uint32_t g = sb + lane * BS_TPT;
        const uint4 x = *reinterpret_cast<const uint4 *>(terms + (m.t0 - headpad) + (g < last_g ? g : last_g));
        tk[0] = x.x; tk[1] = x.y; tk[2] = x.z; tk[3] = x.w;
    };
    uint32_t tka[BS_PRE][BS_TPT]; // the first BS_PRE steps of a tile
    TileMeta cur = fetch_meta((uint64_t)blockIdx.x * BS_WAVES + wv);
    if (meta_ntok(cur)) {
#pragma unroll
        for (int st = 0; st < BS_PRE; ++st) load_step(cur, st * BS_STEP, tka[st]);
    }
    for (uint64_t tile = (uint64_t)blockIdx.x * BS_WAVES + wv; tile < n_tiles; tile += tile_stride) {
        const unsigned long long c0 = TM ? clock64() : 0ull;
        const uint64_t d0 = doc_begin + tile * D;
        const uint32_t nd = cur.nd;
        const uint64_t t0 = cur.t0;
        const uint32_t n_tok = meta_ntok(cur);
        w.off[lane] = (uint32_t)(cur.mine - t0); // (docs past nd hold the tile's end)
        for (uint32_t i = lane; i < BS_TF_SLOTS; i += 64) w.tf[i] = BS_TF_EMPTY;
        if (lane == 0) { // (nd may be 64: the end offset needs its own writer)
            w.off[nd] = n_tok;
            w.overflow_tile = n_tok >= (1u << 21); // a token index would not fit a hit entry
        }
        TileMeta nxt = fetch_meta(tile + tile_stride); // arrives while this tile is scanned
        bs_wave_sync();
        if (lane < nd && w.off[lane + 1] - w.off[lane] >= 4096u) w.overflow_tile = 1; // tf would not fit its 12 bits

        // ---- phase 1: token-parallel scan.  Steps start 16-byte aligned in the token array.
        const unsigned long long c1 = TM ? clock64() : 0ull;
        uint32_t wave_hits = 0; // wave-uniform: hits appended to the tile's hit array
        const uint32_t headpad = (uint32_t)(t0 & 3u); // tokens before t0 in the first aligned group
        const uint32_t span = n_tok + headpad;        // tokens from the first aligned group to the tile's end
        auto scan_step = [&](uint32_t sb, const uint32_t (&tk)[BS_TPT]) {
            // tile-relative index of my first token (32-bit: a tile has fewer than 2^21 tokens) and which of my
            // BS_TPT tokens lie inside the tile
            const int32_t rel0 = (int32_t)(sb + lane * BS_TPT) - (int32_t)headpad;
            const int32_t lo = rel0 < 0 ? -rel0 : 0, hi = (int32_t)n_tok - rel0;
            const uint32_t inside = hi <= lo ? 0u : (((hi >= BS_TPT ? 1u << BS_TPT : 1u << hi) - 1u) & ~((1u << lo) - 1u));
            // ONE LDS read per token (Bloom bit of its term id); ~4.5 % pass
            uint32_t hmask = 0;
#pragma unroll
            for (int i = 0; i < BS_TPT; ++i) {
                const uint32_t bb = bs_bloom_bit(tk[i]);
                hmask |= ((s.bloom[bb >> 5] >> (bb & 31u)) & 1u) << i;
            }
            hmask &= inside;
            // every trip, each lane with a hit left resolves one (term -> slot in the LDS hash table; Bloom
            // false positives die here) and the wave appends the survivors to the tile's hit array:
            // ballot + lane prefix, no atomics, no shuffles
            while (__ballot(hmask != 0)) {
                uint32_t ent = 0xFFFFFFFFu;
                if (hmask) {
                    const uint32_t i = __builtin_ctz(hmask);
                    hmask &= hmask - 1;
                    // the token's term: a select chain over the registers (i is not a constant)
                    uint32_t t = tk[0];
#pragma unroll
                    for (int j = 1; j < BS_TPT; ++j) t = (i == (uint32_t)j) ? tk[j] : t;
                    for (uint32_t h = bs_hash(t);; h = (h + 1) & (BS_HASH - 1)) {
                        const uint32_t k = s.key[h];
                        if (k == t) { ent = h | ((uint32_t)(rel0 + (int32_t)i) << 11); break; }
                        if (k == 0xFFFFFFFFu) break;
                    }
                }
                const uint64_t m = __ballot(ent != 0xFFFFFFFFu);
                if (ent != 0xFFFFFFFFu) {
                    const uint32_t at = wave_hits + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (at < BS_WAVE_HITS) w.hit[at] = ent;
                }
                wave_hits += (uint32_t)__popcll(m);
            }
        };
        if (span) {
#pragma unroll
            for (int st = 0; st < BS_PRE; ++st)
                if (st * BS_STEP < span) scan_step(st * BS_STEP, tka[st]);
            if (span > BS_PRE * BS_STEP) { // a tile of long docs: the rest one step ahead
                uint32_t tk[BS_TPT], tkn[BS_TPT];
                load_step(cur, BS_PRE * BS_STEP, tk);
                for (uint32_t sb = BS_PRE * BS_STEP; sb < span; sb += BS_STEP) {
                    const bool more = sb + BS_STEP < span;
                    if (more) load_step(cur, sb + BS_STEP, tkn);
                    scan_step(sb, tk);
                    if (more) {
#pragma unroll
                        for (int i = 0; i < BS_TPT; ++i) tk[i] = tkn[i];
                    }
                }
            }
        }
        const unsigned long long c2 = TM ? clock64() : 0ull;
        // (pins the first use of the prefetched offsets here: the compiler would otherwise compute with them
        // -- and wait for them -- right after issuing the loads)
        asm volatile("" : "+v"(nxt.t0), "+v"(nxt.tend), "+v"(nxt.mine));
        if (meta_ntok(nxt)) { // the next tile's tokens fly while this one is linked and scored
#pragma unroll
            for (int st = 0; st < BS_PRE; ++st) load_step(nxt, st * BS_STEP, tka[st]);
        }
        if (wave_hits > BS_WAVE_HITS) { // (uniform)
            if (lane == 0) w.overflow_tile = 1;
            wave_hits = BS_WAVE_HITS;
        }
        if (dbg == 1) { cur = nxt; continue; } // ablation: phase 1 only
        bs_wave_sync();
        // ---- link pass: every hit finds its doc (binary search in the tile's offsets) and counts itself in
        // the tile's (doc, term) -> tf table; the hit that creates the entry is the pair's representative
        for (uint32_t e = lane; e < wave_hits; e += 64) {
            const uint32_t x = w.hit[e];
            const uint32_t r = x >> 11;
            uint32_t lo = 0, hi = nd; // largest d with off[d] <= r
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (w.off[mid] <= r) lo = mid; else hi = mid;
            }
            const uint32_t key20 = (lo << 11) | (x & 0x7FFu);
            uint32_t rep = 0, probes = 0;
            for (uint32_t h = bs_tf_hash(key20);; h = (h + 1) & (BS_TF_SLOTS - 1)) {
                uint32_t v = w.tf[h];
                if (v == BS_TF_EMPTY) {
                    v = atomicCAS(&w.tf[h], BS_TF_EMPTY, key20 | (1u << 20));
                    if (v == BS_TF_EMPTY) { rep = 1; break; }
                }
                if ((v & 0xFFFFFu) == key20) { atomicAdd(&w.tf[h], 1u << 20); break; }
                if (++probes == BS_TF_SLOTS) { w.overflow_tile = 1; break; } // (cannot happen: more slots than hits)
            }
            w.hit[e] = key20 | (rep << 20);
        }
        bs_wave_sync();
        const unsigned long long c3 = TM ? clock64() : 0ull;
        // ---- phase 2: one lane per HIT (balanced: a long doc's hits spread over many lanes); only the
        // representative of a (doc, term) pair goes on.  For every query q using the term it looks up the
        // tf of q's terms in the tile's table, and scores the (doc, q) pair iff its term is the first of
        // q's terms (in query order) that the doc holds -- so each pair is scored exactly once, as the
        // f32 sum over q's terms IN QUERY ORDER.
        if (dbg == 2) { cur = nxt; continue; } // ablation: no scoring
        const bool tile_overflow = w.overflow_tile != 0; // (uniform)
        for (uint32_t he = lane; he < wave_hits && !tile_overflow && dbg != 4; he += 64) {
            const uint32_t x = w.hit[he];
            if (!((x >> 20) & 1u)) continue; // not the representative of its (doc, term) pair
            const uint32_t sl = x & 0x7FFu, d = (x >> 11) & 0x1FFu, dkey = x & 0xFF800u;
            const uint32_t dlen = w.off[d + 1] - w.off[d];
            const float ratio = __fdiv_rn((float)dlen, avgdl);
            const float kd = __fmul_rn(BS_K1, __fadd_rn(1.0f - BS_B, __fmul_rn(BS_B, ratio)));
            const uint32_t ub = s.users_off[sl], ue = s.users_off[sl + 1];
            for (uint32_t u = ub; u < ue; ++u) {
                const uint32_t q = s.users[u] >> 16, pos_in_q = s.users[u] & 0xFFFFu;
                const uint2 qs = *reinterpret_cast<const uint2 *>(&s.qp_slots[q][0]);
                float score = 0.0f;
                if ((qs.x & 0xFFFFu) != BS_LONGQ) {
                    const uint32_t t[4] = {qs.x & 0xFFFFu, qs.x >> 16, qs.y & 0xFFFFu, qs.y >> 16};
                    uint32_t tf[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) tf[p] = t[p] == BS_NIL ? 0u : bs_tf_lookup(w, dkey | t[p]);
                    bool mine = true; // no earlier term of q occurs in the doc
#pragma unroll
                    for (int p = 0; p < 3; ++p) mine = mine && !((uint32_t)p < pos_in_q && tf[p] != 0);
                    if (!mine) continue;
                    const float4 qi = *reinterpret_cast<const float4 *>(&s.qp_idf[q][0]);
                    const float idfs[4] = {qi.x, qi.y, qi.z, qi.w};
#pragma unroll
                    for (int p = 0; p < 4; ++p) { // q's terms in query order
                        if (tf[p] == 0) continue;
                        const float ftf = (float)tf[p];
                        const float im = __fdiv_rn(__fmul_rn(ftf, BS_K1 + 1.0f), __fadd_rn(ftf, kd));
                        score = __fadd_rn(score, __fmul_rn(idfs[p], im));
                    }
                } else { // a query of more than four terms: the same, out of the LDS tables
                    const uint32_t qb = s.q_off[q], qe = s.q_off[q + 1];
                    bool mine = true;
                    for (uint32_t p = qb; p < qb + pos_in_q; ++p) {
                        const uint32_t slp = s.q_slot[p];
                        mine = mine && (slp == BS_NIL || bs_tf_lookup(w, dkey | slp) == 0);
                    }
                    if (!mine) continue;
                    for (uint32_t p = qb; p < qe; ++p) {
                        const uint32_t slp = s.q_slot[p];
                        const uint32_t tf = slp == BS_NIL ? 0u : bs_tf_lookup(w, dkey | slp);
                        if (tf == 0) continue;
                        const float ftf = (float)tf;
                        const float im = __fdiv_rn(__fmul_rn(ftf, BS_K1 + 1.0f), __fadd_rn(ftf, kd));
                        score = __fadd_rn(score, __fmul_rn(batch->idf[slp], im));
                    }
                }
                if (score > 0.0f && oi_f32_key(score) >= s.tau[q]) {
                    const uint32_t pos = atomicAdd(&s.seg_fill[q], 1u);
                    if (pos < seg_cap) my_seg[(uint64_t)q * pool_stride + pos] = oi_rank_key(score, doc_id_base + (uint32_t)(d0 + d));
                    else *overflow = 1u;
                }
            }
        }
        const unsigned long long c4 = TM ? clock64() : 0ull;
        // ---- a tile whose hits overflowed the hit array (or with a 4096-token doc): the wave takes its
        // docs one at a time, ONE LANE PER QUERY, counting tf straight from the doc's tokens (every lane
        // reads the same token: a broadcast load that hits in cache, just streamed)
        for (uint32_t d = 0; d < nd && tile_overflow && dbg != 3; ++d) {
            const uint32_t dlen = w.off[d + 1] - w.off[d];
            const float ratio = __fdiv_rn((float)dlen, avgdl);
            const float kd = __fmul_rn(BS_K1, __fadd_rn(1.0f - BS_B, __fmul_rn(BS_B, ratio)));
            const uint32_t *dt = terms + t0 + w.off[d];
            for (uint32_t q = lane; q < nq; q += 64) {
                const uint2 qs = *reinterpret_cast<const uint2 *>(&s.qp_slots[q][0]);
                float score = 0.0f;
                if ((qs.x & 0xFFFFu) != BS_LONGQ) {
                    const uint32_t t[4] = {qs.x & 0xFFFFu, qs.x >> 16, qs.y & 0xFFFFu, qs.y >> 16};
                    uint32_t key[4], tf[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int p = 0; p < 4; ++p) key[p] = t[p] == BS_NIL ? 0xFFFFFFFFu : s.key[t[p]]; // no token has that id
                    for (uint32_t i = 0; i < dlen; ++i) {
                        const uint32_t tok = dt[i];
#pragma unroll
                        for (int p = 0; p < 4; ++p) tf[p] += tok == key[p];
                    }
                    const float4 qi = *reinterpret_cast<const float4 *>(&s.qp_idf[q][0]);
                    const float idfs[4] = {qi.x, qi.y, qi.z, qi.w};
#pragma unroll
                    for (int p = 0; p < 4; ++p) { // q's terms in query order
                        if (tf[p] == 0) continue;
                        const float ftf = (float)tf[p];
                        const float im = __fdiv_rn(__fmul_rn(ftf, BS_K1 + 1.0f), __fadd_rn(ftf, kd));
                        score = __fadd_rn(score, __fmul_rn(idfs[p], im));
                    }
                } else { // a query of more than four terms: term by term
                    for (uint32_t p = s.q_off[q]; p < s.q_off[q + 1]; ++p) {
                        const uint32_t sl = s.q_slot[p];
                        if (sl == BS_NIL) continue;
                        const uint32_t t = s.key[sl];
                        uint32_t tf = 0;
                        for (uint32_t i = 0; i < dlen; ++i) tf += dt[i] == t;
                        if (tf == 0) continue;
                        const float ftf = (float)tf;
                        const float im = __fdiv_rn(__fmul_rn(ftf, BS_K1 + 1.0f), __fadd_rn(ftf, kd));
                        score = __fadd_rn(score, __fmul_rn(batch->idf[sl], im));
                    }
                }
                if (score > 0.0f && oi_f32_key(score) >= s.tau[q]) {
                    const uint32_t pos = atomicAdd(&s.seg_fill[q], 1u);
                    if (pos < seg_cap) my_seg[(uint64_t)q * pool_stride + pos] = oi_rank_key(score, doc_id_base + (uint32_t)(d0 + d));
                    else *overflow = 1u;
                }
            }
        }
        bs_wave_sync(); // the next tile overwrites this wave's slice
        if (TM && lane == 0 && (blockIdx.x & 15u) == 0) {
            atomicAdd(&bs_timing[0], c1 - c0); // tile setup (offsets to LDS, tf table reset)
            atomicAdd(&bs_timing[1], c2 - c1); // phase 1
            atomicAdd(&bs_timing[2], c3 - c2); // next tile's loads issued, link
            atomicAdd(&bs_timing[3], c4 - c3); // phase 2
            atomicAdd(&bs_timing[4], clock64() - c4);
            atomicAdd(&bs_timing[5], 1ull);
            atomicAdd(&bs_timing[6], (unsigned long long)wave_hits);
        }
        cur = nxt;
    }
    __syncthreads();
    for (uint32_t q = tid; q < nq; q += BS_THREADS) {
        const uint32_t c = s.seg_fill[q];
        seg_cnt[(uint64_t)(q_begin + q) * seg_cnt_stride + blockIdx.x] = c < seg_cap ? c : seg_cap;
    }
}

// ------------------------------------------------------------------ host
void oi_bm25_scan_geometry(const oi_ctx *ctx, uint64_t n_docs, uint32_t *n_segs, uint32_t *seg_cap) {
    const uint64_t wg_docs = (uint64_t)BS_WAVES * BS_WD; // docs one workgroup takes per round, at most
    const uint64_t n_rounds = (n_docs + wg_docs - 1) / wg_docs;
    uint64_t grid = 2ull * (uint64_t)ctx->num_cus; // two 512-thread workgroups per CU
    if (grid > n_rounds) grid = n_rounds;
    if (grid == 0) grid = 1;
    *n_segs = (uint32_t)grid;
    // docs one workgroup can meet, whatever tile size (<= BS_WD per wave) the batch picks: n_docs / grid
    // rounded up to whole rounds plus one round
    *seg_cap = (uint32_t)(((n_rounds + grid - 1) / grid + 1) * wg_docs);
}

uint32_t oi_bm25_scan_max_queries() { return BS_MAX_Q; }

// One pass: queries [q_begin, q_begin + nq) of the batch over docs [doc_begin, doc_end).
// pool.n_segs / pool.seg_cap must come from oi_bm25_scan_geometry(doc_end - doc_begin).
uint32_t oi_bm25_scan_pass_queries(uint32_t max_terms_per_query) {
    uint32_t m = max_terms_per_query ? max_terms_per_query : 1;
    uint32_t p = BS_MAX_TERMS / m; // pairs per pass <= BS_MAX_TERMS, hence distinct terms too
    if (p > BS_MAX_Q) p = BS_MAX_Q;
    return p ? p : 1;
}

int oi_launch_bm25_scan(oi_index *idx, const uint32_t *d_q_terms, const uint32_t *d_q_offsets, uint32_t q_begin,
                        uint32_t nq, uint64_t doc_begin, uint64_t doc_end, float avgdl, bool run_setup,
                        const PoolView &pool) {
    oi_ctx *ctx = idx->ctx;
    if (nq == 0 || doc_end <= doc_begin) return OI_OK;
    OI_REQUIRE(nq <= BS_MAX_Q, "bm25 scan: %u queries in one pass (limit %u)", nq, BS_MAX_Q);
    DevBuf &bb = ctx->buf("bm25_scan_batch");
    OI_CHECK(bb.ensure(sizeof(BsBatch)));
    static const int dbg = oi_ablation_env("OI_BM25_SCAN_DBG") ? atoi(oi_ablation_env("OI_BM25_SCAN_DBG")) : 0; // ablations (wrong results)
    auto kernel = dbg == 9 ? bm25_scan_kernel<true> : bm25_scan_kernel<false>;
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(bm25_scan_kernel<true>), (size_t)(sizeof(BsShared))));
    OI_CHECK(oi_dyn_lds(ctx, reinterpret_cast<const void *>(bm25_scan_kernel<false>), (size_t)(sizeof(BsShared))));
    if (run_setup) {
        hipLaunchKernelGGL(bm25_scan_setup, dim3(1), dim3(1024), 0, ctx->stream, d_q_terms, d_q_offsets, q_begin, nq,
                           idx->vocab, idx->max_query_terms, idx->idf.as<float>(), idx->df_local.as<uint32_t>(),
                           (uint64_t)idx->n_docs, avgdl, bb.as<BsBatch>());
        OI_HIP_CHECK(hipGetLastError());
    }
    ProfScope ps(ctx, "bm25");
    hipLaunchKernelGGL(kernel, dim3(pool.n_segs), dim3(BS_THREADS), sizeof(BsShared), ctx->stream,
                       idx->fwd_terms.as<uint32_t>(), idx->fwd_offsets.as<uint64_t>(), doc_begin, doc_end, avgdl,
                       bb.as<BsBatch>(), pool.tau_keys, q_begin, idx->doc_id_base, pool.keys, pool.seg_cnt,
                       pool.seg_cnt_stride, pool.stride, pool.carry_cap, pool.seg_cap, pool.overflow, dbg);
    OI_HIP_CHECK(hipGetLastError());
    if (dbg == 9) {
        unsigned long long h[8] = {0};
        OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        OI_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(bs_timing), sizeof(h)));
        const double w = h[5] ? (double)h[5] : 1.0;
        fprintf(stderr, "[bm25 scan timing] cycles/wave/tile: setup %.0f phase1 %.0f link %.0f phase2 %.0f tail %.0f | samples %llu hits/wave %.1f\n",
                h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5], h[6] / w);
        unsigned long long z[8] = {0};
        OI_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(bs_timing), z, sizeof(z)));
    }
    return OI_OK;
}
