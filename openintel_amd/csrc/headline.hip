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
// Byte/integer work, HBM-bound: each title byte is read once.  A workgroup stages the bytes of
// HL_TILE titles into LDS with coalesced 16-byte loads, then one lane walks one title out of LDS.
// Non-ASCII chars are separators in the reference; all their UTF-8 bytes are >= 0x80, so the
// byte-wise split below is the same split.
//
// Company match without building the joined string: `" form "` occurs in `" w0 w1 .. "` iff the
// form is itself single-space-joined words equal to consecutive title words -- so a form is
// matched from a token start, its spaces standing for the title's separator runs.  Forms that
// are not of that shape (upper case, punctuation, doubled or edge spaces) can never occur in
// the joined string and are dropped on the host; the empty form matches exactly the titles
// with no words (`"  "` contains `"  "`), carried as a flag.
#include "oi_internal.h"

#define HL_THREADS 256
#define HL_TILE 256              // titles per workgroup
#define HL_TEXT_BYTES (48 * 1024) // LDS text window; larger tiles read HBM directly
#define HL_MAX_PAT_BYTES 1024
#define HL_MAX_PATTERNS 32
#define HL_KW_SLOTS 32
#define HL_N_KW 16
#define HL_LONG 14u // token length class "14 or more" of the byte-parallel kernel

struct __attribute__((aligned(16))) HlParams {
    uint32_t n_patterns;
    uint32_t empty_form;
    uint32_t kw_mult;
    uint32_t pad0;
    uint64_t kw_lo[HL_KW_SLOTS]; // keyword bytes 0..7 packed little-endian, 0 = empty slot
    uint64_t kw_hi[HL_KW_SLOTS]; // bytes 8..15
    uint8_t kw_id[HL_KW_SLOTS];
    uint16_t pat_off[HL_MAX_PATTERNS + 2];
    uint8_t pat[HL_MAX_PAT_BYTES];
    uint8_t pad1[12];
    // Token filter of the byte-parallel kernel, indexed by first char (a-z -> 0..25, 0-9 -> 26..35):
    // bits 0..15 = lengths L for which a keyword starts with that char; bits 16..31 = length classes
    // min(L, HL_LONG) of the first words of the company patterns starting with it.
    uint32_t first_char[40];
    // first word of each pattern, packed like a keyword (bytes 0..7 / 8..15); words of HL_LONG or more
    // bytes are not packed (pw_len = 0) and always go to the byte walker
    uint64_t pw_lo[HL_MAX_PATTERNS], pw_hi[HL_MAX_PATTERNS];
    uint8_t pw_len[HL_MAX_PATTERNS];  // length of the first word, 0 if >= HL_LONG
    uint8_t pw_only[HL_MAX_PATTERNS]; // 1: the pattern is that single word
};
static_assert(sizeof(HlParams) % 16 == 0, "HlParams is copied as uint4");

static const char *const kCatalyst[HL_N_KW] = { // dip.rs:38-55, declaration order
    "earnings", "miss",   "guidance", "cut", "offering",   "dilution",  "downgrade",     "halt",
    "fraud",    "lawsuit", "recall",  "fda", "bankruptcy", "delisting", "investigation", "resign"};

__host__ __device__ static inline uint32_t hl_kw_slot(uint64_t lo, uint32_t mult) {
    return (((uint32_t)lo ^ (uint32_t)(lo >> 32)) * mult) >> 27;
}
__device__ static inline bool hl_alnum(uint32_t c) {
    return (c - '0' < 10u) || ((c | 0x20u) - 'a' < 26u);
}
__device__ static inline uint32_t hl_lower(uint32_t c) { return (c - 'A' < 26u) ? c + 32u : c; }

struct HlShared {
    HlParams prm;
    uint32_t text[HL_TEXT_BYTES / 4];
};

// Byte readers: LDS window (position relative to the 16-byte aligned window base) or HBM.
struct HlLdsReader {
    const uint32_t *w;
    __device__ uint32_t operator()(uint32_t i) const { return (w[i >> 2] >> (8u * (i & 3u))) & 0xFFu; }
};
struct HlMemReader {
    const uint8_t *p;
    __device__ uint32_t operator()(uint32_t i) const { return p[i]; }
};

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

template <class R>
__device__ static void hl_scan_title(const R &rd, uint32_t beg, uint32_t end, const HlParams &prm, uint32_t &mask_out,
                                     uint64_t &order_out, uint32_t &about_out) {
    uint32_t mask = 0, nh = 0, about = 0, words = 0;
    uint64_t order = 0;
    uint64_t lo = 0, hi = 0;
    uint32_t len = 0;
    const uint32_t np = prm.n_patterns;
    for (uint32_t i = beg; i <= end; ++i) {
        const uint32_t c = (i < end) ? rd(i) : 0u;
        if (hl_alnum(c)) {
            const uint32_t l = hl_lower(c);
            if (len == 0) {
                ++words;
                if (!about) { // company patterns are tried from token starts only
                    for (uint32_t p = 0; p < np; ++p) {
                        const uint32_t o = prm.pat_off[p];
                        if (prm.pat[o] == l && hl_match_at(rd, i, end, prm.pat + o, prm.pat_off[p + 1] - o)) {
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
                const uint32_t s = hl_kw_slot(lo, prm.kw_mult);
                if (prm.kw_lo[s] == lo && prm.kw_hi[s] == hi) {
                    const uint32_t k = prm.kw_id[s];
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
    if (words == 0 && prm.empty_form) about = 1;
    mask_out = mask;
    order_out = order;
    about_out = about;
}

__global__ __launch_bounds__(HL_THREADS) void headline_scan_kernel(const uint8_t *__restrict__ blob,
                                                                  const uint64_t *__restrict__ offsets, uint64_t n,
                                                                  uint64_t blob_bytes,
                                                                  const HlParams *__restrict__ params,
                                                                  uint16_t *__restrict__ mask_out,
                                                                  uint64_t *__restrict__ order_out,
                                                                  uint8_t *__restrict__ about_out) {
    __shared__ __attribute__((aligned(16))) HlShared s;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < sizeof(HlParams) / 16; i += HL_THREADS)
        reinterpret_cast<uint4 *>(&s.prm)[i] = reinterpret_cast<const uint4 *>(params)[i];

    const uint64_t t0 = (uint64_t)blockIdx.x * HL_TILE;
    const uint64_t t1 = (t0 + HL_TILE < n) ? t0 + HL_TILE : n;
    const uint64_t b0 = offsets[t0], b1 = offsets[t1];
    const uint64_t a0 = b0 & ~(uint64_t)15;
    const bool fits = (b1 - a0) <= HL_TEXT_BYTES;
    if (fits) {
        const uint32_t n16 = (uint32_t)((b1 - a0 + 15) >> 4);
        for (uint32_t i = tid; i < n16; i += HL_THREADS) {
            const uint64_t src = a0 + 16ull * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (src + 16 <= blob_bytes) v = *reinterpret_cast<const uint4 *>(blob + src);
            else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; src + k < blob_bytes; ++k) w[k >> 2] |= (uint32_t)blob[src + k] << (8u * (k & 3u));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            reinterpret_cast<uint4 *>(s.text)[i] = v;
        }
    }
    __syncthreads();

    const uint64_t t = t0 + tid;
    if (t >= t1) return;
    const uint64_t tb = offsets[t], te = offsets[t + 1];
    uint32_t mask, about;
    uint64_t order;
    if (fits) {
        HlLdsReader rd{s.text};
        hl_scan_title(rd, (uint32_t)(tb - a0), (uint32_t)(te - a0), s.prm, mask, order, about);
    } else {
        // a single title of 4 GiB or more would not index with 32 bits; the launcher rejects it
        HlMemReader rd{blob + tb};
        hl_scan_title(rd, 0u, (uint32_t)(te - tb), s.prm, mask, order, about);
    }
    mask_out[t] = (uint16_t)mask;
    order_out[t] = order;
    about_out[t] = (uint8_t)about;
}

// ---------------------------------------------------------------- byte-parallel kernel
// The tile's text is staged once; while it passes through registers every lane turns its 16 bytes
// into 16 alnum bits, so the token pass works on bitmaps: token starts, token lengths and the cut
// at title boundaries are a handful of bit operations per 16-byte chunk.  A token becomes a
// candidate only if (first char, length) can begin a keyword or a company pattern -- a 256-bit
// Bloom filter held in registers, probed with the first char taken from the lane's own staged
// registers, so the streaming loop reads no LDS per token.  Candidates go to a per-wave LDS queue
// and are verified in a dense pass (packed compare against the perfect hash for keywords,
// hl_match_at for patterns).  Hits are rare: each becomes a node in a per-title list (one CAS on
// the title's result word), which the title's lane folds into mask and first-occurrence order.
#define HL_QCAP 192
#define HL_WAVES (HL_THREADS / 64)
#define HL2_TEXT_BYTES (32 * 1024)
#define HL2_CHUNKS (HL2_TEXT_BYTES / 16)
#define HL2_PER_LANE (HL2_CHUNKS / HL_THREADS) // 16-byte chunks a lane stages and scans
#define HL2_NODES 512u                         // keyword hits per tile before the tile is redone lane-per-title

struct Hl2Shared {
    HlParams prm;
    uint32_t text[HL2_TEXT_BYTES / 4 + 8]; // slack: 16-byte token reads may run past the window
    uint16_t am[HL2_CHUNKS + 8];           // am[1 + c] = alnum bits of chunk c; am[0] = 0
    uint16_t ts[HL2_CHUNKS + 8];           // ts[1 + c] = title-start bits of chunk c
    uint32_t off[HL_TILE + 1];             // title offsets relative to the window
    uint32_t res[HL_TILE];                 // bits 0..15 keyword mask, bit 16 about-company, bits 17.. head node + 1
    uint32_t node[HL2_NODES];              // keyword (4 bits) | pos << 4 | (next node + 1) << 19
    uint8_t tchunk[HL2_CHUNKS];            // title owning the first byte of each chunk (0 before the first title)
    uint32_t ctab[256];                    // by first byte of a token: prm.first_char of its class, 0 if not alnum
    uint32_t n_nodes;
    uint32_t q_cnt[HL_WAVES];
    uint32_t queue[HL_WAVES][HL_QCAP];     // pos | lenclass << 16
};
static_assert(sizeof(Hl2Shared) <= 53 * 1024, "three workgroups per CU");

__device__ static inline uint32_t hl_alnum4(uint32_t w) { // one bit per byte
    const uint32_t hi = w & 0x80808080u;
    const uint32_t w7 = w & 0x7F7F7F7Fu;
    const uint32_t ge_a = (w7 | 0x20202020u) + (0x80u - 'a') * 0x01010101u;
    const uint32_t gt_z = (w7 | 0x20202020u) + (0x7Fu - 'z') * 0x01010101u;
    const uint32_t ge_0 = w7 + (0x80u - '0') * 0x01010101u;
    const uint32_t gt_9 = w7 + (0x7Fu - '9') * 0x01010101u;
    const uint32_t f = ((ge_a & ~gt_z) | (ge_0 & ~gt_9)) & ~hi & 0x80808080u;
    return (((f >> 7) & 0x01010101u) * 0x00204081u >> 21) & 0xFu;
}

__host__ __device__ static inline uint32_t hl_char_index(uint32_t c) { // a-z (either case) -> 0..25, 0-9 -> 26..35
    return (c - '0' < 10u) ? 26u + (c - '0') : ((c | 0x20u) - 'a');
}

__device__ static inline uint32_t hl_title_of(const Hl2Shared &s, uint32_t nt, uint32_t pos) {
    // largest j < nt with off[j] <= pos (empty titles share an offset; the last of them owns the byte):
    // start from the title owning the chunk's first byte, step over the titles that begin before pos
    uint32_t j = s.tchunk[pos >> 4];
    while (j + 1u < nt && s.off[j + 1u] <= pos) ++j;
    return j;
}

__device__ static void hl2_verify(Hl2Shared &s, uint32_t nt, uint32_t e) {
    const uint32_t pos = e & 0xFFFFu, lc = e >> 16;
    const uint32_t wi = pos >> 2, sh = pos & 3u;
    const uint32_t x0 = s.text[wi], x1 = s.text[wi + 1], x2 = s.text[wi + 2], x3 = s.text[wi + 3], x4 = s.text[wi + 4];
    const uint32_t fc = s.ctab[__builtin_amdgcn_alignbyte(x1, x0, sh) & 0xFFu];
    // the token, lowercased and packed: every byte inside it is ASCII alphanumeric, so |0x20 lowercases
    // letters and keeps digits.  Exact for lc < HL_LONG; longer tokens are never compared packed.
    const uint64_t t0 = __builtin_amdgcn_alignbyte(x1, x0, sh) | 0x20202020u;
    const uint64_t t1 = __builtin_amdgcn_alignbyte(x2, x1, sh) | 0x20202020u;
    const uint64_t t2 = __builtin_amdgcn_alignbyte(x3, x2, sh) | 0x20202020u;
    const uint64_t t3 = __builtin_amdgcn_alignbyte(x4, x3, sh) | 0x20202020u;
    uint64_t lo = t0 | (t1 << 32), hi = t2 | (t3 << 32);
    if (lc < 8) { lo &= (1ull << (8u * lc)) - 1ull; hi = 0; }
    else if (lc == 8) hi = 0;
    else hi &= (1ull << (8u * (lc - 8u))) - 1ull;
    uint32_t j = ~0u;
    if ((fc >> lc) & 1u) { // keyword: lc is the exact length, 3..13
        const uint32_t sl = hl_kw_slot(lo, s.prm.kw_mult);
        if (s.prm.kw_lo[sl] == lo && s.prm.kw_hi[sl] == hi) {
            const uint32_t k = s.prm.kw_id[sl];
            j = hl_title_of(s, nt, pos);
            const uint32_t idx = atomicAdd(&s.n_nodes, 1u);
            if (idx < HL2_NODES) { // push onto title j's list; past the cap the tile is redone (n_nodes tells)
                uint32_t old = s.res[j], assumed;
                do {
                    assumed = old;
                    s.node[idx] = k | (pos << 4) | ((assumed >> 17) << 19);
                    old = atomicCAS(&s.res[j], assumed, (assumed & 0x1FFFFu) | (1u << k) | ((idx + 1u) << 17));
                } while (old != assumed);
            }
        }
    }
    if ((fc >> (16u + lc)) & 1u) { // company pattern: first word by packed compare, the rest by the byte walker
        const uint32_t np = s.prm.n_patterns;
        for (uint32_t p = 0; p < np; ++p) {
            const uint32_t wl = s.prm.pw_len[p];
            bool hit;
            if (wl) {
                if (wl != lc || s.prm.pw_lo[p] != lo || s.prm.pw_hi[p] != hi) continue;
                hit = s.prm.pw_only[p] != 0;
            } else {
                if (lc != HL_LONG) continue;
                hit = false;
            }
            if (j == ~0u) j = hl_title_of(s, nt, pos);
            if ((s.res[j] >> 16) & 1u) break;
            if (!hit) {
                HlLdsReader rd{s.text};
                const uint32_t o = s.prm.pat_off[p];
                hit = hl_match_at(rd, pos, s.off[j + 1], s.prm.pat + o, s.prm.pat_off[p + 1] - o);
            }
            if (hit) {
                atomicOr(&s.res[j], 1u << 16);
                break;
            }
        }
    }
}

__global__ __launch_bounds__(HL_THREADS) void headline_scan_kernel2(const uint8_t *__restrict__ blob,
                                                                   const uint64_t *__restrict__ offsets, uint64_t n,
                                                                   uint64_t blob_bytes,
                                                                   const HlParams *__restrict__ params,
                                                                   uint16_t *__restrict__ mask_out,
                                                                   uint64_t *__restrict__ order_out,
                                                                   uint8_t *__restrict__ about_out, int dbg,
                                                                   unsigned long long *timing) {
    __shared__ __attribute__((aligned(16))) Hl2Shared s;
    unsigned long long tk[6];
    tk[0] = clock64();
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint64_t t0 = (uint64_t)blockIdx.x * HL_TILE;
    const uint32_t nt = (uint32_t)((t0 + HL_TILE < n) ? HL_TILE : n - t0);
    const uint64_t b0 = offsets[t0], b1 = offsets[t0 + nt];
    const uint64_t my_off = offsets[t0 + (tid < nt ? tid : nt)];
    const uint64_t a0 = b0 & ~(uint64_t)15;

    for (uint32_t i = tid; i < sizeof(HlParams) / 16; i += HL_THREADS)
        reinterpret_cast<uint4 *>(&s.prm)[i] = reinterpret_cast<const uint4 *>(params)[i];

    if ((b1 - a0) > HL2_TEXT_BYTES) { // oversized tile: one lane per title, straight from HBM
        __syncthreads();
        if (tid < nt) {
            const uint64_t te = offsets[t0 + tid + 1];
            HlMemReader rd{blob + my_off};
            uint32_t mask, about;
            uint64_t order;
            hl_scan_title(rd, 0u, (uint32_t)(te - my_off), s.prm, mask, order, about);
            mask_out[t0 + tid] = (uint16_t)mask;
            order_out[t0 + tid] = order;
            about_out[t0 + tid] = (uint8_t)about;
        }
        return;
    }

    // ---- stage: all of the lane's loads in flight at once; text -> LDS, alnum bits on the way
    const uint32_t lo_rel = (uint32_t)(b0 - a0), hi_rel = (uint32_t)(b1 - a0);
    const uint32_t n16 = (hi_rel + 15u) >> 4;
    uint4 v[HL2_PER_LANE];
    if (a0 + 16ull * n16 <= blob_bytes) { // every 16-byte piece of the window lies inside the blob
#pragma unroll
        for (uint32_t k = 0; k < HL2_PER_LANE; ++k) {
            const uint32_t i = tid + k * HL_THREADS;
            v[k] = make_uint4(0, 0, 0, 0);
            if (i < n16) v[k] = *reinterpret_cast<const uint4 *>(blob + a0 + 16ull * i);
        }
    } else { // the blob ends inside the last piece: byte loads there
        for (uint32_t k = 0; k < HL2_PER_LANE; ++k) {
            const uint32_t i = tid + k * HL_THREADS;
            const uint64_t src = a0 + 16ull * i;
            uint32_t w[4] = {0, 0, 0, 0};
            if (i < n16)
                for (uint32_t q = 0; q < 16 && src + q < blob_bytes; ++q) w[q >> 2] |= (uint32_t)blob[src + q] << (8u * (q & 3u));
            v[k] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    // per-title state while the loads fly
    for (uint32_t i = tid; i < (HL2_CHUNKS + 8) / 2; i += HL_THREADS) reinterpret_cast<uint32_t *>(s.ts)[i] = 0;
    s.res[tid] = 0;
    s.off[tid] = (uint32_t)(my_off - a0);
    if (tid < 8) {
        s.am[1 + n16 + tid] = 0; // lookahead of the last chunks
        if (tid == 0) { s.am[0] = 0; s.off[HL_TILE] = hi_rel; s.n_nodes = 0; s.tchunk[0] = 0; }
        if (tid < HL_WAVES) s.q_cnt[tid] = 0;
    }
    s.ctab[tid] = hl_alnum(tid) ? params->first_char[hl_char_index(tid)] : 0u;
#pragma unroll
    for (uint32_t k = 0; k < HL2_PER_LANE; ++k) {
        const uint32_t i = tid + k * HL_THREADS;
        if (i < n16) {
            reinterpret_cast<uint4 *>(s.text)[i] = v[k];
            s.am[1 + i] = (uint16_t)(hl_alnum4(v[k].x) | (hl_alnum4(v[k].y) << 4) | (hl_alnum4(v[k].z) << 8) |
                                     (hl_alnum4(v[k].w) << 12));
        }
    }
    __syncthreads();
    tk[1] = clock64();
    if (tid < nt) { // title starts; ts[1 + o/16] bit o%16, addressed as dwords for the atomic
        const uint32_t o = (uint32_t)(my_off - a0);
        atomicOr(reinterpret_cast<uint32_t *>(s.ts) + ((16u + o) >> 5), 1u << ((16u + o) & 31u));
    }
    if (tid == 0) // the end of the last title cuts tokens too: the next tile's bytes follow in the window
        atomicOr(reinterpret_cast<uint32_t *>(s.ts) + ((16u + hi_rel) >> 5), 1u << ((16u + hi_rel) & 31u));
    if (tid < nt) { // chunks whose first byte lies in this title
        const uint32_t tb = s.off[tid], te = s.off[tid + 1];
        if (te > tb)
            for (uint32_t c = (tb + 15u) >> 4; c <= ((te - 1u) >> 4); ++c) s.tchunk[c] = (uint8_t)tid;
    }
    __syncthreads();

    // ---- token pass over the lane's own chunks
    tk[2] = clock64();
    if (dbg == 1) goto finalize;
#pragma unroll 1
    for (uint32_t k = 0; k < HL2_PER_LANE; ++k) {
        const uint32_t c = tid + k * HL_THREADS;
        if (c < n16) {
            const uint32_t c0 = c << 4;
            const uint32_t A = (uint32_t)s.am[1 + c] | ((uint32_t)s.am[2 + c] << 16);
            const uint32_t prev = (s.am[c] >> 15) & 1u;
            const uint32_t T = (uint32_t)s.ts[1 + c] | ((uint32_t)s.ts[2 + c] << 16);
            uint32_t starts = A & (~((A << 1) | prev) | T) & 0xFFFFu;
            if (c0 < lo_rel) starts &= ~((1u << (lo_rel - c0)) - 1u); // bytes of the previous tile
            if (hi_rel - c0 < 16u) starts &= (1u << (hi_rel - c0)) - 1u;
            // a token cannot continue into a non-alnum byte or across a title start; bit 31 bounds the search
            const uint32_t Z = ~A | T | 0x80000000u;
            const uint8_t *tbase = reinterpret_cast<const uint8_t *>(s.text) + c0;
            while (starts) {
                const uint32_t b = __builtin_ctz(starts);
                starts &= starts - 1;
                const uint32_t len = __builtin_ctz(Z >> (b + 1u)) + 1u; // exact below HL_LONG (b + 14 <= 31)
                const uint32_t lc = len < HL_LONG ? len : HL_LONG;
                const uint32_t fc = s.ctab[tbase[b]];
                if (!((fc >> lc) & 0x10001u)) continue; // (first char, length) begins no keyword and no pattern
                const uint32_t e = (c0 + b) | (lc << 16);
                const uint32_t qp = atomicAdd(&s.q_cnt[wv], 1u);
                if (qp < HL_QCAP) s.queue[wv][qp] = e;
                else hl2_verify(s, nt, e); // queue full: verify in place
            }
        }
    }
    // ---- dense pass over this wave's queue (a wave's LDS operations complete in order)
    tk[3] = clock64();
    if (dbg != 2) {
        uint32_t nq = s.q_cnt[wv];
        if (nq > HL_QCAP) nq = HL_QCAP;
        for (uint32_t q = lane; q < nq; q += 64) hl2_verify(s, nt, s.queue[wv][q]);
    }
    tk[4] = clock64();
finalize:
    __syncthreads();
    tk[5] = clock64();

    // ---- one result per title
    if (tid < nt) {
        uint32_t mask, about;
        uint64_t order = 0;
        if (s.n_nodes > HL2_NODES) { // more hits than nodes (a title repeating a keyword hundreds of times)
            HlLdsReader rd{s.text};
            hl_scan_title(rd, s.off[tid], s.off[tid + 1], s.prm, mask, order, about);
        } else {
            const uint32_t r = s.res[tid];
            mask = r & 0xFFFFu;
            about = (r >> 16) & 1u;
            uint32_t rem = mask, nh = 0;
            while (rem && nh < HL_N_KW) { // first-occurrence order (dip.rs:266): take the earliest remaining hit
                uint32_t best = ~0u;
                for (uint32_t nd = r >> 17, steps = 0; nd && steps < HL2_NODES; ++steps) {
                    const uint32_t x = s.node[nd - 1u];
                    if (((rem >> (x & 15u)) & 1u) && (x & 0x7FFFFu) < best) best = x & 0x7FFFFu; // pos in the high bits decides
                    nd = x >> 19;
                }
                const uint32_t bk = best & 15u;
                order |= (uint64_t)bk << (4u * nh);
                ++nh;
                rem &= ~(1u << bk);
            }
            if (s.prm.empty_form && !about) { // the empty form matches exactly the titles without words
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
    if (timing && lane == 0 && (blockIdx.x & 63u) == 0) { // development aid, one workgroup in 64: cycles per phase, summed over waves (OI_HEADLINE_TIMING)
        const unsigned long long t6 = clock64();
        atomicAdd(&timing[0], tk[1] - tk[0]); // stage
        atomicAdd(&timing[1], tk[2] - tk[1]); // title-start bits
        atomicAdd(&timing[2], tk[3] - tk[2]); // token pass
        atomicAdd(&timing[3], tk[4] - tk[3]); // verify pass
        atomicAdd(&timing[4], tk[5] - tk[4]); // wait at the barrier
        atomicAdd(&timing[5], t6 - tk[5]);    // results
        atomicAdd(&timing[6], 1ull);
        atomicAdd(&timing[7], (unsigned long long)s.q_cnt[wv]);
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

static int hl_build_params(HlParams &prm, const uint8_t *ticker, uint64_t ticker_len, const uint8_t *forms_blob,
                           const uint32_t *form_offsets, uint32_t n_forms) {
    memset(&prm, 0, sizeof(prm));
    // keyword table: perfect hash over the packed first 8 bytes
    uint64_t lo[HL_N_KW], hi[HL_N_KW];
    for (int k = 0; k < HL_N_KW; ++k) {
        lo[k] = hi[k] = 0;
        const size_t l = strlen(kCatalyst[k]);
        for (size_t i = 0; i < l; ++i) {
            if (i < 8) lo[k] |= (uint64_t)(uint8_t)kCatalyst[k][i] << (8 * i);
            else hi[k] |= (uint64_t)(uint8_t)kCatalyst[k][i] << (8 * (i - 8));
        }
    }
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
    prm.kw_mult = mult;
    for (int k = 0; k < HL_N_KW; ++k) {
        const uint32_t sl = hl_kw_slot(lo[k], mult);
        prm.kw_lo[sl] = lo[k];
        prm.kw_hi[sl] = hi[k];
        prm.kw_id[sl] = (uint8_t)k;
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
        if (l == 0) { prm.empty_form = 1; continue; }
        std::vector<uint8_t> tmp(forms_blob + o, forms_blob + o + l);
        tmp.push_back(0); // hl_joinable peeks one byte past a space
        if (!hl_joinable(tmp.data(), l)) continue;
        OI_CHECK(add(tmp.data(), l));
    }
    prm.n_patterns = np;
    // (first char, length) filter of the byte-parallel kernel
    auto mark = [&](uint8_t c, uint32_t lc, uint32_t shift) {
        const uint32_t ci = hl_char_index(c);
        prm.first_char[ci] |= 1u << (shift + lc);
    };
    for (int k = 0; k < HL_N_KW; ++k) mark((uint8_t)kCatalyst[k][0], (uint32_t)strlen(kCatalyst[k]), 0);
    for (uint32_t p = 0; p < np; ++p) {
        const uint8_t *pt = prm.pat + prm.pat_off[p];
        const uint32_t pl = prm.pat_off[p + 1] - prm.pat_off[p];
        uint32_t w = 0;
        while (w < pl && pt[w] != ' ') ++w;
        mark(pt[0], w < HL_LONG ? w : HL_LONG, 16);
        if (w < HL_LONG) {
            prm.pw_len[p] = (uint8_t)w;
            prm.pw_only[p] = (uint8_t)(w == pl);
            for (uint32_t i = 0; i < w; ++i) {
                if (i < 8) prm.pw_lo[p] |= (uint64_t)pt[i] << (8 * i);
                else prm.pw_hi[p] |= (uint64_t)pt[i] << (8 * (i - 8));
            }
        }
    }
    return OI_OK;
}

int oi_launch_headline_scan(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n,
                            uint64_t blob_bytes, const uint8_t *ticker, uint64_t ticker_len,
                            const uint8_t *forms_blob, const uint32_t *form_offsets, uint32_t n_forms,
                            uint16_t *d_mask, uint64_t *d_order, uint8_t *d_about) {
    OI_REQUIRE(((uintptr_t)d_blob & 15u) == 0, "headline scan: title blob must be 16-byte aligned in HBM");
    OI_REQUIRE((n + HL_TILE - 1) / HL_TILE <= 0x7FFFFFFFull, "headline scan: too many titles for one launch");
    static thread_local HlParams prm; // staged synchronously by the pageable copy below
    OI_CHECK(hl_build_params(prm, ticker, ticker_len, forms_blob, form_offsets, n_forms));
    DevBuf &dp = ctx->buf("hl_params");
    OI_CHECK(dp.ensure(sizeof(HlParams)));
    OI_HIP_CHECK(hipMemcpyAsync(dp.p, &prm, sizeof(HlParams), hipMemcpyHostToDevice, ctx->stream));
    const uint32_t grid = (uint32_t)((n + HL_TILE - 1) / HL_TILE);
    static const int dbg = getenv("OI_HEADLINE_DBG") ? atoi(getenv("OI_HEADLINE_DBG")) : 0; // ablations (wrong results)
    unsigned long long *d_timing = nullptr;
    if (getenv("OI_HEADLINE_TIMING")) {
        DevBuf &tb = ctx->buf("hl_timing");
        OI_CHECK(tb.ensure(8 * sizeof(unsigned long long)));
        OI_HIP_CHECK(hipMemsetAsync(tb.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
        d_timing = tb.as<unsigned long long>();
    }
    static const bool v1 = getenv("OI_HEADLINE_V1") != nullptr; // one lane per title (kept for A/B runs)
    ctx->prof_begin("headline");
    if (v1)
        hipLaunchKernelGGL(headline_scan_kernel, dim3(grid), dim3(HL_THREADS), 0, ctx->stream, d_blob, d_offsets, n,
                           blob_bytes, dp.as<HlParams>(), d_mask, d_order, d_about);
    else
        hipLaunchKernelGGL(headline_scan_kernel2, dim3(grid), dim3(HL_THREADS), 0, ctx->stream, d_blob, d_offsets, n,
                           blob_bytes, dp.as<HlParams>(), d_mask, d_order, d_about, dbg, d_timing);
    ctx->prof_end("headline");
    if (d_timing) {
        unsigned long long h[8];
        OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        OI_HIP_CHECK(hipMemcpy(h, d_timing, sizeof(h), hipMemcpyDeviceToHost));
        const double w = h[6] ? (double)h[6] : 1.0;
        fprintf(stderr, "[headline timing] cycles/wave: stage %.0f tsbits %.0f token %.0f verify %.0f barrier %.0f results %.0f | "
                        "waves %llu queue/wave %.1f\n", h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6], h[7] / w);
    }
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

extern "C" const char *oi_catalyst_keyword(uint32_t index) { return index < HL_N_KW ? kCatalyst[index] : nullptr; }
