// headline.hip -- the dip gate's headline scan on gfx950: catalyst keyword hits and the
// "does this title name the company" test, for a batch of titles resident in HBM.
//
// Reference behaviour (openintel, paths relative to its repo):
//   CATALYST_KEYWORDS            src/domain/dip.rs:38-55
//   normalize_words              src/domain/dip.rs:204-210   ASCII lowercase, split on !is_ascii_alphanumeric
//   headline_mentions_company    src/domain/dip.rs:247-258
//   catalyst_hits                src/domain/dip.rs:261-272
//   the gate's loop over titles  src/domain/dip.rs:617-626
//
// Byte/integer work, HBM-bound by nature: each title byte is read from HBM once.  Non-ASCII chars
// are separators in the reference; all their UTF-8 bytes are >= 0x80, so splitting on bytes is the
// same split.
//
// Company match without building the joined string: `" form "` occurs in `" w0 w1 .. "` iff the
// form is itself single-space-joined words equal to consecutive title words -- so a form is
// matched word by word from a token start.  Forms that are not of that shape (upper case,
// punctuation, doubled or edge spaces) can never occur in the joined string and are dropped on
// the host; the empty form matches exactly the titles with no words (`"  "` contains `"  "`),
// carried as a flag.
//
// Kernel (headline_scan_kernel): a workgroup takes `tile` consecutive titles (chosen on the host
// from the average title length so that a tile's bytes fill at most 7/8 of the 24 KB LDS window),
// stages their bytes with coalesced 16-byte loads and turns every 16-byte chunk into 16 "alnum"
// bits on the way.  The token pass then works on bitmaps: token starts, lengths and the cut at title
// boundaries are a few bit operations per chunk; a lane walks its chunk's tokens on its own and
// marks a token as a candidate only if its (first char, length) can begin a keyword or a company
// pattern (one LDS table lookup).  The marked tokens are compacted with wave ballots into two
// per-wave rings, by kind, and verified a batch of ONE kind at a time (a mixed batch would run both
// paths in turn): packed 16-byte compares against a perfect hash (keywords) or the packed words of
// the patterns.  Hits are rare; each keyword hit becomes a node in a per-title list, folded at the
// end into the keyword mask and the first-occurrence order the reference's Vec has.  Tiles that do
// not fit the window, or that hold more hits than nodes, are done one lane per title
// (hl_scan_title), which is also kernel v1.
#include "oi_device.h"
#include "oi_internal.h"

#define HL_THREADS 256
#define HL_WAVES (HL_THREADS / 64)
#define HL_MAX_TILE 256            // titles per workgroup, at most
#define HL_WIN_BYTES (24 * 1024)   // LDS text window
#define HL_CHUNKS (HL_WIN_BYTES / 16)
#define HL_PER_LANE (HL_CHUNKS / HL_THREADS) // 16-byte chunks a lane stages and scans
#define HL_NODES 384u              // keyword hits per tile before the tile is redone lane-per-title
#define HL_RING 64u                // candidate ring per wave and kind (keyword / pattern): one batch of the wave
#define HL_MAX_PAT_BYTES 1024
#define HL_MAX_PATTERNS 32
#define HL_MAX_WORDS 64            // packed pattern words; patterns beyond take the byte walker
#define HL_KW_SLOTS 32
#define HL_N_KW 16
#define HL_LONG 14u                // token length class "14 or more": such tokens are never compared packed

// Everything the kernel needs per call.  The hot part is copied to LDS; the raw pattern bytes stay
// in HBM and are only read by the byte walker (patterns with a word of HL_LONG or more bytes).
struct __attribute__((aligned(16))) HlHot {
    uint32_t n_patterns;
    uint32_t empty_form;
    uint32_t kw_mult;
    uint32_t min_len; // shortest keyword / pattern first word: shorter tokens are dropped before the token loop
    uint64_t kw_lo[HL_KW_SLOTS]; // keyword bytes 0..7 packed little-endian, 0 = empty slot
    uint64_t kw_hi[HL_KW_SLOTS]; // bytes 8..15
    uint64_t wd_lo[HL_MAX_WORDS], wd_hi[HL_MAX_WORDS]; // pattern words, packed the same way
    // Token filter, indexed by first char (a-z -> 0..25, 0-9 -> 26..35): bits 0..15 = lengths L for
    // which a keyword starts with that char; bits 16..31 = length classes min(L, HL_LONG) of the
    // first words of the company patterns starting with it.
    uint32_t first_char[40];
    uint8_t kw_id[HL_KW_SLOTS];
    uint8_t wd_len[HL_MAX_WORDS];           // 1..13
    uint8_t pat_w0[HL_MAX_PATTERNS + 4];    // words of pattern p: wd[pat_w0[p] .. pat_w0[p + 1]) (33 used; dword-sized)
    uint8_t pat_bytewise[HL_MAX_PATTERNS];  // 1: has a long word (or the word table was full): byte walker
    uint8_t pat_first_len[HL_MAX_PATTERNS]; // length class of the first word
    uint8_t pad1[12];
};
static_assert(offsetof(HlHot, wd_len) % 4 == 0 && offsetof(HlHot, pat_w0) % 4 == 0 && offsetof(HlHot, pat_bytewise) % 4 == 0 &&
                  offsetof(HlHot, pat_first_len) % 4 == 0,
              "hl_u8 reads the byte tables as dwords");
struct HlParams {
    HlHot hot;
    uint16_t pat_off[HL_MAX_PATTERNS + 2];
    uint8_t pat[HL_MAX_PAT_BYTES];
};
static_assert(sizeof(HlHot) % 16 == 0, "HlHot is copied as uint4");

static const char *const kCatalyst[HL_N_KW] = { // dip.rs:38-55, declaration order
    "earnings", "miss",   "guidance", "cut", "offering",   "dilution",  "downgrade",     "halt",
    "fraud",    "lawsuit", "recall",  "fda", "bankruptcy", "delisting", "investigation", "resign"};

__host__ __device__ static inline uint32_t hl_kw_slot(uint64_t lo, uint32_t mult) {
    return (((uint32_t)lo ^ (uint32_t)(lo >> 32)) * mult) >> 27;
}
__host__ __device__ static inline uint32_t hl_char_index(uint32_t c) { // a-z (either case) -> 0..25, 0-9 -> 26..35
    return (c - '0' < 10u) ? 26u + (c - '0') : ((c | 0x20u) - 'a');
}
__device__ static inline bool hl_alnum(uint32_t c) {
    return (c - '0' < 10u) || ((c | 0x20u) - 'a' < 26u);
}
__device__ static inline uint32_t hl_lower(uint32_t c) { return (c - 'A' < 26u) ? c + 32u : c; }

// Byte readers: LDS window (position relative to the 16-byte aligned window base) or HBM.
struct HlLdsReader {
    const uint32_t *w;
    __device__ uint32_t operator()(uint32_t i) const { return (w[i >> 2] >> (8u * (i & 3u))) & 0xFFu; }
};
struct HlMemReader {
    const uint8_t *p;
    __device__ uint32_t operator()(uint32_t i) const { return p[i]; }
};

// ---------------------------------------------------------------- one lane per title
// Does pattern `pt[0..pl)` match the title words starting at the token that begins at `pos`?
template <class R>
__device__ static bool hl_match_at(const R &rd, uint32_t pos, uint32_t end, const uint8_t *pt, uint32_t pl) {
    uint32_t t = pos;
    for (uint32_t j = 0; j < pl; ++j) {
        const uint32_t f = pt[j];
        if (f == ' ') { // next word: the title must be at a separator run followed by a word
            if (t >= end || hl_alnum(rd(t))) return false;
            while (t < end && !hl_alnum(rd(t))) ++t;
            if (t >= end) return false;
        } else {
            if (t >= end || hl_lower(rd(t)) != f) return false;
            ++t;
        }
    }
    return t >= end || !hl_alnum(rd(t));
}

// The reference's loop, one byte at a time.  `hot` may live in LDS; `prm` is the HBM copy.
template <class R>
__device__ static void hl_scan_title(const R &rd, uint32_t beg, uint32_t end, const HlHot &hot, const HlParams *prm,
                                     uint32_t &mask_out, uint64_t &order_out, uint32_t &about_out) {
    uint32_t mask = 0, nh = 0, about = 0, words = 0;
    uint64_t order = 0;
    uint64_t lo = 0, hi = 0;
    uint32_t len = 0;
    const uint32_t np = hot.n_patterns;
    for (uint32_t i = beg; i <= end; ++i) {
        const uint32_t c = (i < end) ? rd(i) : 0u;
        if (hl_alnum(c)) {
            const uint32_t l = hl_lower(c);
            if (len == 0) {
                ++words;
                if (!about) { // company patterns are tried from token starts only
                    for (uint32_t p = 0; p < np; ++p) {
                        const uint32_t o = prm->pat_off[p];
                        if (prm->pat[o] == l && hl_match_at(rd, i, end, prm->pat + o, prm->pat_off[p + 1] - o)) {
                            about = 1;
                            break;
                        }
                    }
                }
            }
            if (len < 8) lo |= (uint64_t)l << (8u * len);
            else if (len < 16) hi |= (uint64_t)l << (8u * (len - 8u));
            ++len;
        } else {
            if (len >= 3 && len <= 13) { // keyword lengths
                const uint32_t s = hl_kw_slot(lo, hot.kw_mult);
                if (hot.kw_lo[s] == lo && hot.kw_hi[s] == hi) {
                    const uint32_t k = hot.kw_id[s];
                    if (!((mask >> k) & 1u)) { // dip.rs:266 first occurrence only
                        mask |= 1u << k;
                        order |= (uint64_t)k << (4u * nh);
                        ++nh;
                    }
                }
            }
            lo = hi = 0;
            len = 0;
        }
    }
    if (words == 0 && hot.empty_form) about = 1;
    mask_out = mask;
    order_out = order;
    about_out = about;
}

// Kernel v1 (OI_HEADLINE_V1): 256 titles per workgroup, one lane walks one title straight from HBM.
__global__ __launch_bounds__(HL_THREADS) void headline_scan_kernel_v1(const uint8_t *__restrict__ blob,
                                                                     const uint64_t *__restrict__ offsets, uint64_t n,
                                                                     const HlParams *__restrict__ params,
                                                                     uint16_t *__restrict__ mask_out,
                                                                     uint64_t *__restrict__ order_out,
                                                                     uint8_t *__restrict__ about_out) {
    __shared__ __attribute__((aligned(16))) HlHot hot;
    for (uint32_t i = threadIdx.x; i < sizeof(HlHot) / 16; i += HL_THREADS)
        reinterpret_cast<uint4 *>(&hot)[i] = reinterpret_cast<const uint4 *>(&params->hot)[i];
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * HL_THREADS + threadIdx.x;
    if (t >= n) return;
    const uint64_t tb = offsets[t], te = offsets[t + 1];
    HlMemReader rd{blob + tb};
    uint32_t mask, about;
    uint64_t order;
    hl_scan_title(rd, 0u, (uint32_t)(te - tb), hot, params, mask, order, about);
    mask_out[t] = (uint16_t)mask;
    order_out[t] = order;
    about_out[t] = (uint8_t)about;
}

// ---------------------------------------------------------------- byte-parallel kernel
struct HlShared {
    HlHot hot;
    uint32_t text[HL_WIN_BYTES / 4 + 8];   // slack: 20-byte token reads may run past the window
    uint16_t am[HL_CHUNKS + 8];            // am[1 + c] = alnum bits of chunk c; am[0] = 0
    uint16_t ts[HL_CHUNKS + 8];            // ts[1 + c] = title-start bits of chunk c
    uint32_t off[HL_MAX_TILE + 1];         // title offsets relative to the window
    uint32_t res[HL_MAX_TILE];             // bits 0..15 keyword mask, bit 16 about-company, bits 17.. head node + 1
    uint32_t node[HL_NODES];               // keyword (4 bits) | pos << 4 | (next node + 1) << 19
    uint32_t ctab[256];                    // by first byte of a token: hot.first_char of its class, 0 if not alnum
    uint32_t ring[HL_WAVES][2][HL_RING];   // candidates, [0] keyword [1] pattern: pos | length class << 16
    uint8_t tchunk[HL_CHUNKS];             // title owning the first byte of each chunk (0 before the first title)
    uint32_t n_nodes;
};
static_assert(sizeof(HlShared) <= 40 * 1024, "four workgroups per CU");

__device__ static inline uint32_t hl_title_of(const HlShared &s, uint32_t nt, uint32_t pos) {
    // largest j < nt with off[j] <= pos (empty titles share an offset; the last of them owns the byte):
    // start from the title owning the chunk's first byte, step over the titles that begin before pos
    uint32_t j = s.tchunk[pos >> 4];
    while (j + 1u < nt && s.off[j + 1u] <= pos) ++j;
    return j;
}

// The token at window position `pos` (`len` bytes, all ASCII alphanumeric, len < HL_LONG), lowercased
// and packed like the tables: |0x20 lowercases letters and keeps digits.
__device__ static inline void hl_pack_token(const HlShared &s, uint32_t pos, uint32_t len, uint64_t &lo, uint64_t &hi) {
    const uint32_t wi = pos >> 2, sh = pos & 3u;
    const uint32_t x0 = s.text[wi], x1 = s.text[wi + 1], x2 = s.text[wi + 2], x3 = s.text[wi + 3], x4 = s.text[wi + 4];
    const uint64_t t0 = __builtin_amdgcn_alignbyte(x1, x0, sh) | 0x20202020u;
    const uint64_t t1 = __builtin_amdgcn_alignbyte(x2, x1, sh) | 0x20202020u;
    const uint64_t t2 = __builtin_amdgcn_alignbyte(x3, x2, sh) | 0x20202020u;
    const uint64_t t3 = __builtin_amdgcn_alignbyte(x4, x3, sh) | 0x20202020u;
    lo = t0 | (t1 << 32);
    hi = t2 | (t3 << 32);
    if (len < 8) { lo &= (1ull << (8u * len)) - 1ull; hi = 0; }
    else if (len == 8) hi = 0;
    else if (len < 16) hi &= (1ull << (8u * (len - 8u))) - 1ull;
}

// Entry i of a dword-aligned byte table of the call's tables in HBM.  With a wave-uniform i this is a scalar load and
// two scalar operations: the pattern tables are walked by pattern and word number, the same for every lane, so reading
// them through the scalar cache instead of LDS takes them off the vector and LDS pipes.
__device__ __forceinline__ uint32_t hl_u8(const uint8_t *table, uint32_t i) {
    return (reinterpret_cast<const uint32_t *>(table)[i >> 2] >> (8u * (i & 3u))) & 0xFFu;
}

// 32 alnum bits starting at chunk c's first byte (bit i = byte 16c + i)
__device__ static inline uint32_t hl_am32(const HlShared &s, uint32_t c) {
    return (uint32_t)s.am[1 + c] | ((uint32_t)s.am[2 + c] << 16);
}

// Words 1.. of pattern p against the title's tokens after the one ending at `q`; `end` = title end.
__device__ static bool hl_match_rest(const HlShared &s, const HlHot &gh, uint32_t p, uint32_t q, uint32_t end) {
    const uint32_t w1 = hl_u8(gh.pat_w0, p + 1u);
    for (uint32_t w = hl_u8(gh.pat_w0, p) + 1u; w < w1; ++w) {
        // next token start: first alnum byte at or after q, before the title's end
        uint32_t st = q;
        for (;;) {
            if (st >= end) return false;
            const uint32_t x = hl_am32(s, st >> 4) >> (st & 15u); // at least 17 valid bits
            if (x) { st += __builtin_ctz(x); break; }
            st = (st & ~15u) + 32u; // separators up to the end of the next chunk
        }
        if (st >= end) return false;
        const uint32_t y = ~(hl_am32(s, st >> 4) >> (st & 15u)) | (1u << 16); // run length, capped at 16
        uint32_t len = __builtin_ctz(y);
        if (len > end - st) len = end - st;
        if (len != hl_u8(gh.wd_len, w)) return false; // wd_len < HL_LONG <= 16: the comparison is exact
        uint64_t lo, hi;
        hl_pack_token(s, st, len, lo, hi);
        if (lo != gh.wd_lo[w] || hi != gh.wd_hi[w]) return false;
        q = st + len;
    }
    return true;
}

// A candidate is verified by kind, a batch of one kind at a time: a wave runs ONE of the two paths per batch instead
// of both in turn for a mixed batch.  dbg (ablation builds): 3 = token packing only, 4 = no keyword bookkeeping
// (node lists), 5 = no company patterns.
//
// Keyword candidate: the token's (first char, exact length 3..13) begins a keyword.
__device__ static void hl_verify_keyword(HlShared &s, uint32_t nt, uint32_t e, int dbg = 0) {
    const uint32_t pos = e & 0xFFFFu, lc = e >> 16;
    uint64_t lo, hi;
    hl_pack_token(s, pos, lc, lo, hi);
    if (dbg == 3) { if ((lo ^ hi) == 0xDEADBEEFull) s.n_nodes = 0; return; }
    if (dbg == 4) return;
    const uint32_t sl = hl_kw_slot(lo, s.hot.kw_mult);
    if (s.hot.kw_lo[sl] == lo && s.hot.kw_hi[sl] == hi) {
        const uint32_t k = s.hot.kw_id[sl];
        const uint32_t j = hl_title_of(s, nt, pos);
        const uint32_t idx = atomicAdd(&s.n_nodes, 1u);
        if (idx < HL_NODES) { // push onto title j's list; past the cap the tile is redone (n_nodes tells)
            uint32_t old = s.res[j], assumed;
            do {
                assumed = old;
                s.node[idx] = k | (pos << 4) | ((assumed >> 17) << 19);
                old = atomicCAS(&s.res[j], assumed, (assumed & 0x1FFFFu) | (1u << k) | ((idx + 1u) << 17));
            } while (old != assumed);
        }
    }
}

// Pattern candidate: the token's (first char, length class) is that of a company pattern's first word.  Words by
// packed compare, long words by the byte walker.
__device__ static void hl_verify_pattern(HlShared &s, const HlParams *prm, uint32_t nt, uint32_t e, int dbg = 0) {
    const uint32_t pos = e & 0xFFFFu, lc = e >> 16;
    uint64_t lo, hi;
    hl_pack_token(s, pos, lc, lo, hi); // exact for lc < HL_LONG; longer tokens are never compared packed
    if (dbg == 3) { if ((lo ^ hi) == 0xDEADBEEFull) s.n_nodes = 0; return; }
    if (dbg == 5) return;
    uint32_t j = ~0u;
    const HlHot &gh = prm->hot; // the pattern tables by (uniform) pattern number: scalar loads, not LDS reads
    const uint32_t np = gh.n_patterns;
    for (uint32_t p = 0; p < np; ++p) {
        if (hl_u8(gh.pat_first_len, p) != lc) continue;
        const uint32_t w0 = hl_u8(gh.pat_w0, p);
        if (lc < HL_LONG && (gh.wd_lo[w0] != lo || gh.wd_hi[w0] != hi)) continue;
        if (j == ~0u) j = hl_title_of(s, nt, pos);
        if ((s.res[j] >> 16) & 1u) break;
        bool hit;
        if (hl_u8(gh.pat_bytewise, p)) {
            HlLdsReader rd{s.text};
            const uint32_t o = prm->pat_off[p];
            hit = hl_match_at(rd, pos, s.off[j + 1], prm->pat + o, prm->pat_off[p + 1] - o);
        } else {
            hit = hl_match_rest(s, gh, p, pos + lc, s.off[j + 1]);
        }
        if (hit) {
            atomicOr(&s.res[j], 1u << 16);
            break;
        }
    }
}

// Append this round's candidates of one kind (lanes with `has`) to the wave's ring of that kind; when they do not
// fit beside the ones waiting, the waiting ones (a nearly full batch) are verified first.  Wave-uniform cursors.
template <int KIND>
__device__ __forceinline__ void hl_ring_push(HlShared &s, const HlParams *prm, uint32_t nt, uint32_t wv, uint32_t lane,
                                             bool has, uint32_t e, uint32_t &head, uint32_t &tail, int dbg) {
    const uint64_t m = __ballot(has);
    const uint32_t cnt = (uint32_t)__popcll(m);
    if (tail - head + cnt > HL_RING) { // a wave's LDS operations complete in order: the reads precede the writes below
        if (lane < tail - head && dbg != 2) {
            const uint32_t w = s.ring[wv][KIND][(head + lane) & (HL_RING - 1u)];
            if (KIND == 0) hl_verify_keyword(s, nt, w, dbg);
            else hl_verify_pattern(s, prm, nt, w, dbg);
        }
        head = tail;
    }
    if (has) {
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        s.ring[wv][KIND][(tail + below) & (HL_RING - 1u)] = e;
    }
    tail += cnt;
}

// The piece the blob ends in, zero-filled past the end.
__device__ __noinline__ uint4 hl_load_tail(const uint8_t *blob, uint64_t src, uint64_t blob_bytes) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t q = 0; q < 16 && src + q < blob_bytes; ++q) w[q >> 2] |= (uint32_t)blob[src + q] << (8u * (q & 3u));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// A lane's share of a tile's window: 16-byte pieces tid, tid + 256, ... of [a0, a0 + 16 * n16).
__device__ __forceinline__ void hl_load_text(const uint8_t *__restrict__ blob, uint64_t blob_bytes, uint64_t a0,
                                             uint32_t n16, uint32_t tid, uint4 (&v)[HL_PER_LANE]) {
    if (n16 == 0) return;
    const uint64_t whole = (blob_bytes - a0) >> 4; // pieces wholly inside the blob (a0 < blob_bytes here)
    const uint32_t full = whole < n16 ? (uint32_t)whole : n16;
    if (full) {
        // No branch around the loads (a lane past the end re-reads the last piece and ignores it), so
        // all of them are in flight together instead of each waiting at its own join.
#pragma unroll
        for (uint32_t k = 0; k < HL_PER_LANE; ++k) {
            const uint32_t i = tid + k * HL_THREADS;
            v[k] = *reinterpret_cast<const uint4 *>(blob + a0 + 16ull * (i < full ? i : full - 1u));
        }
    }
    if (full < n16) { // the blob ends inside piece `full` (the last one): its owner loads it by bytes
#pragma unroll
        for (uint32_t k = 0; k < HL_PER_LANE; ++k) {
            if (tid + k * HL_THREADS == full) v[k] = hl_load_tail(blob, a0 + 16ull * full, blob_bytes);
        }
    }
}

template <bool TM> // TM: per-phase cycle counters (development aid)
// (4 waves per SIMD: the LDS budget allows four workgroups per CU, the registers must too.)
__global__ __launch_bounds__(HL_THREADS, 4) void headline_scan_kernel(const uint8_t *__restrict__ blob,
                                                                  const uint64_t *__restrict__ offsets, uint64_t n,
                                                                  uint64_t blob_bytes, uint32_t tile,
                                                                  const HlParams *__restrict__ params,
                                                                  uint16_t *__restrict__ mask_out,
                                                                  uint64_t *__restrict__ order_out,
                                                                  uint8_t *__restrict__ about_out, int dbg,
                                                                  unsigned long long *timing) {
    __shared__ __attribute__((aligned(16))) HlShared s;
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0};
    if (TM) tk[0] = clock64();
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint64_t t0 = (uint64_t)blockIdx.x * tile;
    const uint32_t nt = (uint32_t)((t0 + tile < n) ? tile : n - t0);
    const uint64_t b0 = offsets[t0], b1 = offsets[t0 + nt];
    const uint64_t my_off = offsets[t0 + (tid < nt ? tid : nt)];
    const uint64_t a0 = b0 & ~(uint64_t)15;

    for (uint32_t i = tid; i < sizeof(HlHot) / 16; i += HL_THREADS)
        reinterpret_cast<uint4 *>(&s.hot)[i] = reinterpret_cast<const uint4 *>(&params->hot)[i];

    if ((b1 - a0) > HL_WIN_BYTES) { // oversized tile: one lane per title, straight from HBM
        __syncthreads();
        if (tid < nt) {
            const uint64_t te = offsets[t0 + tid + 1];
            HlMemReader rd{blob + my_off};
            uint32_t mask, about;
            uint64_t order;
            hl_scan_title(rd, 0u, (uint32_t)(te - my_off), s.hot, params, mask, order, about);
            mask_out[t0 + tid] = (uint16_t)mask;
            order_out[t0 + tid] = order;
            about_out[t0 + tid] = (uint8_t)about;
        }
        return;
    }

    // ---- stage: all of the lane's loads in flight at once; text -> LDS, alnum bits on the way
    const uint32_t lo_rel = (uint32_t)(b0 - a0), hi_rel = (uint32_t)(b1 - a0);
    const uint32_t n16 = (hi_rel + 15u) >> 4;
    uint4 v[HL_PER_LANE];
    hl_load_text(blob, blob_bytes, a0, n16, tid, v);
    // per-title state while the loads fly
    for (uint32_t i = tid; i < (HL_CHUNKS + 8) / 2; i += HL_THREADS) reinterpret_cast<uint32_t *>(s.ts)[i] = 0;
    s.res[tid] = 0;
    s.off[tid] = (uint32_t)(my_off - a0);
    s.ctab[tid] = hl_alnum(tid) ? params->hot.first_char[hl_char_index(tid)] : 0u;
    if (tid < 8) {
        s.am[1 + n16 + tid] = 0; // lookahead of the last chunks
        if (tid == 0) { s.am[0] = 0; s.off[HL_MAX_TILE] = hi_rel; s.n_nodes = 0; s.tchunk[0] = 0; }
    }
#pragma unroll
    for (uint32_t k = 0; k < HL_PER_LANE; ++k) {
        const uint32_t i = tid + k * HL_THREADS;
        if (i < n16) {
            reinterpret_cast<uint4 *>(s.text)[i] = v[k];
            s.am[1 + i] = (uint16_t)oi_alnum16(v[k]);
        }
    }
    __syncthreads();
    if (TM) tk[1] = clock64();
    if (tid < nt) {
        const uint32_t tb = s.off[tid], te = s.off[tid + 1];
        // title starts: ts[1 + o/16] bit o%16, addressed as dwords for the atomic
        atomicOr(reinterpret_cast<uint32_t *>(s.ts) + ((16u + tb) >> 5), 1u << ((16u + tb) & 31u));
        if (te > tb) // chunks whose first byte lies in this title
            for (uint32_t c = (tb + 15u) >> 4; c <= ((te - 1u) >> 4); ++c) s.tchunk[c] = (uint8_t)tid;
    }
    if (tid == 0) // the end of the last title cuts tokens too: the next tile's bytes follow in the window
        atomicOr(reinterpret_cast<uint32_t *>(s.ts) + ((16u + hi_rel) >> 5), 1u << ((16u + hi_rel) & 31u));
    __syncthreads();
    if (TM) tk[2] = clock64();

    // ---- token pass over the lane's chunks.  Per chunk a lane first walks its tokens on its own (no ballots) and
    // marks the candidates in a bitmap -- bit b: the token starting at byte b may be a keyword, bit 16 + b: it may
    // begin a company pattern -- then the wave compacts the marked ones with ballots into its two rings, by kind.
    uint32_t kh = 0, kt = 0, ph = 0, pt = 0; // wave-uniform ring cursors: keyword ring, pattern ring
    const uint32_t min_len = s.hot.min_len;
    if (dbg != 1) {
        for (uint32_t k = 0; k < HL_PER_LANE; ++k) {
            if (wv * 64u + k * HL_THREADS >= n16) break; // uniform: the wave has no chunk left
            const uint32_t c = tid + k * HL_THREADS, c0 = c << 4;
            uint32_t starts = 0, Z = 0x80000000u; // bit 31 keeps every ctz below defined
            if (c < n16) {
                const uint32_t A = hl_am32(s, c);
                const uint32_t prev = (s.am[c] >> 15) & 1u;
                const uint32_t T = (uint32_t)s.ts[1 + c] | ((uint32_t)s.ts[2 + c] << 16);
                starts = A & (~((A << 1) | prev) | T) & 0xFFFFu;
                if (c0 < lo_rel) starts &= ~((1u << (lo_rel - c0)) - 1u); // bytes of the previous tile
                if (hi_rel - c0 < 16u) starts &= (1u << (hi_rel - c0)) - 1u;
                // a token cannot continue into a non-alnum byte or across a title start; bit 31 bounds the search
                Z = ~A | T | 0x80000000u;
                // tokens shorter than every keyword and every pattern's first word never enter the loop
                if (min_len >= 2u) starts &= ~(Z >> 1);
                if (min_len >= 3u) starts &= ~(Z >> 2);
            }
            const uint8_t *tbytes = reinterpret_cast<const uint8_t *>(s.text) + c0;
            uint32_t cm = 0;
            while (starts) {
                const uint32_t b = __builtin_ctz(starts);
                starts &= starts - 1;
                const uint32_t len = __builtin_ctz(Z >> (b + 1u)) + 1u; // exact below HL_LONG (b + 14 <= 31)
                const uint32_t lc = len < HL_LONG ? len : HL_LONG;
                const uint32_t fc = s.ctab[tbytes[b]];
                cm |= ((fc >> lc) & 0x10001u) << b; // (first char, length) begins a keyword / a pattern
            }
            uint32_t kb = cm & 0xFFFFu, pb = cm >> 16;
            while (__ballot(kb != 0)) {
                const bool has = kb != 0;
                const uint32_t b = has ? (uint32_t)__builtin_ctz(kb) : 0u;
                kb &= kb - 1u;
                const uint32_t len = __builtin_ctz(Z >> (b + 1u)) + 1u; // a keyword candidate's length is exact (3..13)
                hl_ring_push<0>(s, params, nt, wv, lane, has, (c0 + b) | (len << 16), kh, kt, dbg);
            }
            while (__ballot(pb != 0)) {
                const bool has = pb != 0;
                const uint32_t b = has ? (uint32_t)__builtin_ctz(pb) : 0u;
                pb &= pb - 1u;
                const uint32_t len = __builtin_ctz(Z >> (b + 1u)) + 1u;
                hl_ring_push<1>(s, params, nt, wv, lane, has, (c0 + b) | ((len < HL_LONG ? len : HL_LONG) << 16), ph, pt, dbg);
            }
        }
        if (TM) tk[3] = clock64();
        if (dbg != 2) { // what still waits: one batch per kind
            if (lane < kt - kh) hl_verify_keyword(s, nt, s.ring[wv][0][(kh + lane) & (HL_RING - 1u)], dbg);
            if (lane < pt - ph) hl_verify_pattern(s, params, nt, s.ring[wv][1][(ph + lane) & (HL_RING - 1u)], dbg);
        }
    } else {
        if (TM) tk[3] = clock64();
    }
    if (TM) tk[4] = clock64();
    __syncthreads();
    if (TM) tk[5] = clock64();

    // ---- one result per title
    if (tid < nt) {
        uint32_t mask, about;
        uint64_t order = 0;
        if (s.n_nodes > HL_NODES) { // more hits than nodes (a title repeating a keyword hundreds of times)
            HlLdsReader rd{s.text};
            hl_scan_title(rd, s.off[tid], s.off[tid + 1], s.hot, params, mask, order, about);
        } else {
            const uint32_t r = s.res[tid];
            mask = r & 0xFFFFu;
            about = (r >> 16) & 1u;
            uint32_t rem = mask, nh = 0;
            while (rem && nh < HL_N_KW) { // first-occurrence order (dip.rs:266): take the earliest remaining hit
                uint32_t best = ~0u;
                for (uint32_t nd = r >> 17, steps = 0; nd && steps < HL_NODES; ++steps) {
                    const uint32_t x = s.node[nd - 1u];
                    if (((rem >> (x & 15u)) & 1u) && (x & 0x7FFFFu) < best) best = x & 0x7FFFFu; // pos in the high bits decides
                    nd = x >> 19;
                }
                const uint32_t bk = best & 15u;
                order |= (uint64_t)bk << (4u * nh);
                ++nh;
                rem &= ~(1u << bk);
            }
            if (s.hot.empty_form && !about) { // the empty form matches exactly the titles without words
                const uint32_t tb = s.off[tid], te = s.off[tid + 1];
                uint32_t any = 0;
                for (uint32_t p = tb; p < te && !any; ++p) any = (s.am[1 + (p >> 4)] >> (p & 15u)) & 1u;
                if (!any) about = 1;
            }
        }
        mask_out[t0 + tid] = (uint16_t)mask;
        order_out[t0 + tid] = order;
        about_out[t0 + tid] = (uint8_t)about;
    }
    if (TM && timing && lane == 0 && (blockIdx.x & 63u) == 0) { // development aid, one workgroup in 64 (OI_HEADLINE_TIMING)
        const unsigned long long t6 = clock64();
        atomicAdd(&timing[0], tk[1] - tk[0]); // stage
        atomicAdd(&timing[1], tk[2] - tk[1]); // title-start bits, chunk owners
        atomicAdd(&timing[2], tk[3] - tk[2]); // token pass (with the full verify batches inside)
        atomicAdd(&timing[3], tk[4] - tk[3]); // last verify batch
        atomicAdd(&timing[4], tk[5] - tk[4]); // wait at the barrier
        atomicAdd(&timing[5], t6 - tk[5]);    // results
        atomicAdd(&timing[6], 1ull);
        atomicAdd(&timing[7], (unsigned long long)(kt + pt));
    }
}

// ---------------------------------------------------------------- host
static bool hl_word_char(uint8_t c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z'); }

// A pattern can occur in the joined title only if it is [a-z0-9]+ words joined by single spaces.
static bool hl_joinable(const uint8_t *p, uint32_t n) {
    if (n == 0 || !hl_word_char(p[0]) || !hl_word_char(p[n - 1])) return false;
    for (uint32_t i = 0; i < n; ++i) {
        if (p[i] == ' ') {
            if (p[i + 1] == ' ') return false;
        } else if (!hl_word_char(p[i])) return false;
    }
    return true;
}

static void hl_pack_word(const uint8_t *w, uint32_t n, uint64_t &lo, uint64_t &hi) {
    lo = hi = 0;
    for (uint32_t i = 0; i < n && i < 16; ++i) {
        if (i < 8) lo |= (uint64_t)w[i] << (8 * i);
        else hi |= (uint64_t)w[i] << (8 * (i - 8));
    }
}

static int hl_build_params(HlParams &prm, const uint8_t *ticker, uint64_t ticker_len, const uint8_t *forms_blob,
                           const uint32_t *form_offsets, uint32_t n_forms) {
    memset(&prm, 0, sizeof(prm));
    HlHot &hot = prm.hot;
    // keyword table: perfect hash over the packed first 8 bytes
    uint64_t lo[HL_N_KW], hi[HL_N_KW];
    for (int k = 0; k < HL_N_KW; ++k)
        hl_pack_word((const uint8_t *)kCatalyst[k], (uint32_t)strlen(kCatalyst[k]), lo[k], hi[k]);
    uint32_t mult = 0;
    for (uint32_t m = 0x9E3779B1u;; m += 2) {
        uint32_t used = 0;
        bool ok = true;
        for (int k = 0; k < HL_N_KW && ok; ++k) {
            const uint32_t sl = hl_kw_slot(lo[k], m);
            ok = !((used >> sl) & 1u);
            used |= 1u << sl;
        }
        if (ok) { mult = m; break; }
    }
    hot.kw_mult = mult;
    for (int k = 0; k < HL_N_KW; ++k) {
        const uint32_t sl = hl_kw_slot(lo[k], mult);
        hot.kw_lo[sl] = lo[k];
        hot.kw_hi[sl] = hi[k];
        hot.kw_id[sl] = (uint8_t)k;
    }
    // patterns: the ticker as a one-word form (dip.rs:249-252), then the usable name forms (:255-257)
    uint32_t used = 0, np = 0;
    auto add = [&](const uint8_t *p, uint32_t n) -> int {
        if (np >= HL_MAX_PATTERNS || used + n > HL_MAX_PAT_BYTES) {
            oi_set_error("headline scan: more than %d patterns or %d pattern bytes", HL_MAX_PATTERNS, HL_MAX_PAT_BYTES);
            return OI_ERR_INVALID_ARG;
        }
        memcpy(prm.pat + used, p, n);
        prm.pat_off[np++] = (uint16_t)used;
        used += n;
        prm.pat_off[np] = (uint16_t)used;
        return OI_OK;
    };
    if (ticker_len >= 2) {
        std::vector<uint8_t> tl(ticker, ticker + ticker_len);
        for (auto &c : tl)
            if (c >= 'A' && c <= 'Z') c = (uint8_t)(c + 32);
        bool word = true; // equal to a title word only if it is one word of [a-z0-9]
        for (auto c : tl) word = word && hl_word_char(c);
        if (word) OI_CHECK(add(tl.data(), (uint32_t)tl.size()));
    }
    for (uint32_t f = 0; f < n_forms; ++f) {
        const uint32_t o = form_offsets[f], l = form_offsets[f + 1] - o;
        if (l == 0) { hot.empty_form = 1; continue; }
        std::vector<uint8_t> tmp(forms_blob + o, forms_blob + o + l);
        tmp.push_back(0); // hl_joinable peeks one byte past a space
        if (!hl_joinable(tmp.data(), l)) continue;
        OI_CHECK(add(tmp.data(), l));
    }
    hot.n_patterns = np;
    // (first char, length) filter and the packed words
    for (int k = 0; k < HL_N_KW; ++k)
        hot.first_char[hl_char_index((uint8_t)kCatalyst[k][0])] |= 1u << strlen(kCatalyst[k]);
    uint32_t nw = 0;
    for (uint32_t p = 0; p < np; ++p) {
        const uint8_t *pt = prm.pat + prm.pat_off[p];
        const uint32_t pl = prm.pat_off[p + 1] - prm.pat_off[p];
        std::vector<std::pair<uint32_t, uint32_t>> words; // (offset, length)
        for (uint32_t i = 0; i < pl;) {
            uint32_t e = i;
            while (e < pl && pt[e] != ' ') ++e;
            words.push_back({i, e - i});
            i = e + 1;
        }
        bool packed = nw + words.size() <= HL_MAX_WORDS;
        for (auto &w : words) packed = packed && w.second < HL_LONG;
        const uint32_t fl = words[0].second < HL_LONG ? words[0].second : HL_LONG;
        hot.pat_first_len[p] = (uint8_t)fl;
        hot.first_char[hl_char_index(pt[0])] |= 1u << (16u + fl);
        hot.pat_w0[p] = (uint8_t)nw;
        hot.pat_bytewise[p] = (uint8_t)!packed;
        if (packed) {
            for (auto &w : words) {
                hl_pack_word(pt + w.first, w.second, hot.wd_lo[nw], hot.wd_hi[nw]);
                hot.wd_len[nw++] = (uint8_t)w.second;
            }
        } else if (fl < HL_LONG) { // the first word is still compared packed
            if (nw >= HL_MAX_WORDS) {
                oi_set_error("headline scan: more than %d pattern words", HL_MAX_WORDS);
                return OI_ERR_INVALID_ARG;
            }
            hl_pack_word(pt, fl, hot.wd_lo[nw], hot.wd_hi[nw]);
            hot.wd_len[nw++] = (uint8_t)fl;
        }
    }
    hot.pat_w0[np] = (uint8_t)nw;
    hot.min_len = 3; // the shortest keywords ("cut", "fda")
    for (uint32_t p = 0; p < np; ++p)
        if (hot.pat_first_len[p] < hot.min_len) hot.min_len = hot.pat_first_len[p];
    return OI_OK;
}

// The per-call tables are opaque outside this file: their size, and a builder into caller memory (oi_headline_scan_rows
// builds one per row of a dip scan into one staging buffer).
size_t oi_headline_params_bytes() { return sizeof(HlParams); }
int oi_headline_build_params(void *dst, const uint8_t *ticker, uint64_t ticker_len, const uint8_t *forms_blob,
                             const uint32_t *form_offsets, uint32_t n_forms) {
    return hl_build_params(*reinterpret_cast<HlParams *>(dst), ticker, ticker_len, forms_blob, form_offsets, n_forms);
}

int oi_launch_headline_scan(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n,
                            uint64_t blob_bytes, const uint8_t *ticker, uint64_t ticker_len,
                            const uint8_t *forms_blob, const uint32_t *form_offsets, uint32_t n_forms,
                            uint16_t *d_mask, uint64_t *d_order, uint8_t *d_about) {
    static thread_local HlParams prm; // staged synchronously by the pageable copy below
    OI_CHECK(hl_build_params(prm, ticker, ticker_len, forms_blob, form_offsets, n_forms));
    DevBuf &dp = ctx->buf("hl_params");
    OI_CHECK(dp.ensure(sizeof(HlParams)));
    OI_HIP_CHECK(hipMemcpyAsync(dp.p, &prm, sizeof(HlParams), hipMemcpyHostToDevice, ctx->stream));
    return oi_launch_headline_scan_params(ctx, d_blob, d_offsets, n, blob_bytes, blob_bytes, dp.p, d_mask, d_order, d_about);
}

// The scan of titles [0, n) of `d_offsets` (absolute byte offsets into d_blob: a sub-range of a larger batch works, its
// offsets pointer advanced) with tables already in HBM.  text_bytes: the bytes of these n titles, for the tile choice.
int oi_launch_headline_scan_params(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n,
                                   uint64_t blob_bytes, uint64_t text_bytes, const void *d_params, uint16_t *d_mask,
                                   uint64_t *d_order, uint8_t *d_about) {
    OI_REQUIRE(((uintptr_t)d_blob & 15u) == 0, "headline scan: title blob must be 16-byte aligned in HBM");
    const HlParams *dpp = reinterpret_cast<const HlParams *>(d_params);
    // titles per workgroup: the largest tile whose average bytes fill at most 7/8 of the LDS window.  The cost per
    // workgroup is largely fixed per wave (one partial verify batch, three barriers, two dependent HBM round trips),
    // so the time falls with the tile (10M titles of 76 bytes: 192 -> 0.664 ms, 224 -> 0.63, 256 -> 0.585); a tile
    // whose bytes overflow the window is redone one lane per title, ~10x slower, hence the 1/8 of slack: with 256
    // titles a tile is 14 % over its average only if the mean of 256 lengths is, > 4 sigma for sigma = mean / 2.
    static const uint32_t kTiles[] = {256, 224, 192, 160, 128, 96, 64, 48, 32, 16, 8};
    static const uint32_t forced = oi_ablation_env("OI_HEADLINE_TILE") ? (uint32_t)atoi(oi_ablation_env("OI_HEADLINE_TILE")) : 0u;
    const uint64_t avg = text_bytes / n + 1;
    uint32_t tile = 8;
    for (uint32_t t : kTiles)
        if (avg * t <= (HL_WIN_BYTES * 7) / 8) { tile = t; break; }
    // a small batch (the gate's own call is a ticker's handful of headlines) is spread over the chip instead: smaller tiles
    // until they number twice the CUs
    while (tile > 8u && (n + tile - 1) / tile < 2ull * (uint64_t)ctx->num_cus) tile >>= 1;
    if (forced >= 1 && forced <= HL_MAX_TILE) tile = forced;
    static const int dbg = oi_ablation_env("OI_HEADLINE_DBG") ? atoi(oi_ablation_env("OI_HEADLINE_DBG")) : 0; // ablations (wrong results)
    static const bool v1 = oi_ablation_env("OI_HEADLINE_V1") != nullptr; // one lane per title (kept for A/B runs)
    const uint64_t per_wg = v1 ? HL_THREADS : tile;
    OI_REQUIRE((n + per_wg - 1) / per_wg <= 0x7FFFFFFFull, "headline scan: too many titles for one launch");
    const uint32_t grid = (uint32_t)((n + per_wg - 1) / per_wg);
    unsigned long long *d_timing = nullptr;
    if (oi_ablation_env("OI_HEADLINE_TIMING")) {
        DevBuf &tb = ctx->buf("hl_timing");
        OI_CHECK(tb.ensure(8 * sizeof(unsigned long long)));
        OI_HIP_CHECK(hipMemsetAsync(tb.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
        d_timing = tb.as<unsigned long long>();
    }
    ctx->prof_begin("headline");
    if (v1)
        hipLaunchKernelGGL(headline_scan_kernel_v1, dim3(grid), dim3(HL_THREADS), 0, ctx->stream, d_blob, d_offsets, n,
                           dpp, d_mask, d_order, d_about);
    else if (d_timing)
        hipLaunchKernelGGL(headline_scan_kernel<true>, dim3(grid), dim3(HL_THREADS), 0, ctx->stream, d_blob, d_offsets,
                           n, blob_bytes, tile, dpp, d_mask, d_order, d_about, dbg, d_timing);
    else
        hipLaunchKernelGGL(headline_scan_kernel<false>, dim3(grid), dim3(HL_THREADS), 0, ctx->stream, d_blob, d_offsets,
                           n, blob_bytes, tile, dpp, d_mask, d_order, d_about, dbg, d_timing);
    ctx->prof_end("headline");
    OI_HIP_CHECK(hipGetLastError());
    if (d_timing) {
        unsigned long long h[8];
        OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        OI_HIP_CHECK(hipMemcpy(h, d_timing, sizeof(h), hipMemcpyDeviceToHost));
        const double w = h[6] ? (double)h[6] : 1.0;
        fprintf(stderr, "[headline timing] tile %u cycles/wave: stage %.0f tsbits %.0f token+verify %.0f last-verify %.0f "
                        "barrier %.0f results %.0f | waves %llu candidates/wave %.1f\n",
                tile, h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6], h[7] / w);
    }
    return OI_OK;
}

extern "C" const char *oi_catalyst_keyword(uint32_t index) { return index < HL_N_KW ? kCatalyst[index] : nullptr; }
