// lexicon.hip -- the reference's per-post path on gfx950.
//
//   LexiconAnalyzer::score        openintel src/adapters/analyzer/lexicon.rs:53-73
//   LexiconAnalyzer::analyze      openintel src/adapters/analyzer/lexicon.rs:82-87
//   Polarity::new                 openintel src/domain/values/polarity.rs:8-14
//   SpeculationEngine::social_summary (the two reduction loops)
//                                 openintel src/domain/engine/speculation_engine.rs:76-97
//
// Byte-parallel HBM scan: a workgroup owns 256 consecutive posts; their bytes are
// one contiguous span of the blob, streamed through LDS in 4 KiB sub-tiles with
// 16-byte coalesced loads (lane = 16 bytes).  Each lane finds the tokens that START
// in its 16 bytes with SWAR byte tests, cuts them at post boundaries, and looks each
// one up in a 256-slot perfect-hash table of the 42 lexicon words held in LDS.  Hit
// counts accumulate per post in LDS (integer atomics: exact, order-free); the
// polarity division happens once per post, in f64, exactly as the reference does.
//
// Unicode: the reference lowercases with str::to_lowercase and then splits on every
// char that is not ASCII alphanumeric.  Only three mappings can put an ASCII char
// into the lowercased text: A-Z -> a-z, U+212A (E2 84 AA) -> 'k', U+0130 (C4 B0) ->
// 'i' + U+0307.  All other non-ASCII bytes separate tokens.  A lane whose window holds
// a 0xAA or 0xB0 byte takes a (rare) exact per-char path.
#include <cstdlib>
#include <mutex>
#include <vector>

#include "oi_device.h"
#include "oi_internal.h"

#define LEX_THREADS 256
#define LEX_PPT 256   // posts per workgroup tile
#define LEX_SUB 4096  // bytes per sub-tile = LEX_THREADS * 16
#define LEX_SLOTS 256

struct LexEntry {
    uint32_t k0, k1;   // chars 0-3, 4-7 (lowercase ASCII, zero padded)
    uint32_t c8_len;   // char 8 | len << 8
    uint32_t flags;    // 1 bull, 2 bear, 4 jargon; 0 = empty slot
};

__device__ __forceinline__ uint32_t lex_hash(uint32_t k0, uint32_t k1, uint32_t c8_len, uint32_t mult) {
    uint32_t x = k0 * 0x9E3779B1u ^ k1 * 0x85EBCA77u ^ c8_len * 0xC2B2AE3Du;
    x ^= x >> 15;
    return (x * mult) >> 24;
}

// ---- SWAR helpers on 4 packed bytes -------------------------------------------
// 0x80 in every byte whose value is in [lo, hi]; bytes >= 0x80 never match.
__device__ __forceinline__ uint32_t swar_range(uint32_t w7, uint32_t lo, uint32_t hi) {
    // w7 has every byte < 0x80, so the adds below never carry across bytes
    uint32_t ge = w7 + (0x80u - lo) * 0x01010101u;
    uint32_t gt = w7 + (0x7Fu - hi) * 0x01010101u;
    return ge & ~gt & 0x80808080u;
}
// ASCII-lowercase the bytes of w; *alnum gets 0x80 per byte that is [0-9a-z] afterwards.
__device__ __forceinline__ uint32_t swar_lower_alnum(uint32_t w, uint32_t *alnum) {
    const uint32_t hi = w & 0x80808080u;
    const uint32_t w7 = w & 0x7F7F7F7Fu;
    const uint32_t up = swar_range(w7, 'A', 'Z') & ~hi;
    const uint32_t lw = w | (up >> 2); // 0x80 >> 2 == 0x20
    const uint32_t l7 = lw & 0x7F7F7F7Fu;
    *alnum = (swar_range(l7, 'a', 'z') | swar_range(l7, '0', '9')) & ~hi;
    return lw;
}
// bit i = byte i's 0x80 flag
__device__ __forceinline__ uint32_t swar_movemask(uint32_t flags80) {
    return (((flags80 >> 7) & 0x01010101u) * 0x00204081u >> 21) & 0xFu;
}
__device__ __forceinline__ bool swar_has_byte(uint32_t w, uint32_t b) {
    uint32_t x = w ^ (b * 0x01010101u);
    return ((x - 0x01010101u) & ~x & 0x80808080u) != 0;
}
__device__ __forceinline__ uint32_t byte_mask(uint32_t nbytes) { // low nbytes bytes, 0..4
    return nbytes >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nbytes)) - 1u);
}

struct LexShared {
    LexEntry table[LEX_SLOTS];
    uint64_t off[LEX_PPT + 1];
    uint32_t bull[LEX_PPT], bear[LEX_PPT], spec[LEX_PPT];
    uint32_t text[(LEX_SUB + 32) / 4]; // [sb-16, sb+LEX_SUB+16)
};

__device__ __forceinline__ void lex_lookup(const LexShared &s, uint32_t mult, uint32_t k0, uint32_t k1,
                                           uint32_t c8, uint32_t len, uint32_t post, LexShared &sw) {
    const uint32_t c8_len = c8 | (len << 8);
    const LexEntry e = s.table[lex_hash(k0, k1, c8_len, mult)];
    if (e.flags != 0 && e.k0 == k0 && e.k1 == k1 && e.c8_len == c8_len) {
        if (e.flags & 1u) atomicAdd(&sw.bull[post], 1u);
        if (e.flags & 2u) atomicAdd(&sw.bear[post], 1u);
        if (e.flags & 4u) atomicOr(&sw.spec[post], 1u);
    }
}

// Exact per-char path for positions [lo, hi) of post range; reads the blob directly.
__device__ bool lex_is_alnum(uint32_t c) {
    return (c - 'a' < 26u) || (c - '0' < 10u);
}
__device__ void lex_slow_chunk(const uint8_t *blob, uint64_t lo, uint64_t hi, uint32_t j,
                               LexShared &s, uint32_t mult) {
    for (uint64_t pos = lo; pos < hi; ++pos) {
        while (pos >= s.off[j + 1]) ++j;
        const uint64_t pstart = s.off[j], pend = s.off[j + 1];
        uint32_t b = blob[pos];
        if ((b & 0xC0u) == 0x80u) continue; // continuation byte: not the start of a char
        // first lowercased char of the char at pos
        uint32_t first;
        if (b < 0x80u) first = (b - 'A' < 26u) ? b + 32u : b;
        else if (b == 0xE2u && pos + 2 < pend && blob[pos + 1] == 0x84u && blob[pos + 2] == 0xAAu) first = 'k';
        else if (b == 0xC4u && pos + 1 < pend && blob[pos + 1] == 0xB0u) first = 'i';
        else first = 0;
        if (!lex_is_alnum(first)) continue;
        if (pos > pstart) { // last lowercased char of the previous char
            uint64_t p = pos - 1;
            while (p > pstart && (blob[p] & 0xC0u) == 0x80u) --p;
            uint32_t pb = blob[p], last;
            if (pb < 0x80u) last = (pb - 'A' < 26u) ? pb + 32u : pb;
            else if (pb == 0xE2u && p + 3 == pos && blob[p + 1] == 0x84u && blob[p + 2] == 0xAAu) last = 'k';
            else last = 0; // includes U+0130 -> 'i' U+0307: the last char is the combining dot
            if (lex_is_alnum(last)) continue;
        }
        // token start: walk its chars
        uint32_t len = 0, k0 = 0, k1 = 0, c8 = 0;
        uint64_t p = pos;
        while (p < pend) {
            uint32_t c, adv;
            bool ends = false;
            uint32_t bb = blob[p];
            if (bb < 0x80u) { c = (bb - 'A' < 26u) ? bb + 32u : bb; adv = 1; }
            else if (bb == 0xE2u && p + 2 < pend && blob[p + 1] == 0x84u && blob[p + 2] == 0xAAu) { c = 'k'; adv = 3; }
            else if (bb == 0xC4u && p + 1 < pend && blob[p + 1] == 0xB0u) { c = 'i'; adv = 2; ends = true; }
            else break;
            if (!lex_is_alnum(c)) break;
            if (len < 4) k0 |= c << (8 * len);
            else if (len < 8) k1 |= c << (8 * (len - 4));
            else if (len == 8) c8 = c;
            ++len;
            p += adv;
            if (ends) break;
        }
        if (len <= 9) lex_lookup(s, mult, k0, k1, c8, len, j, s);
    }
}

__global__ __launch_bounds__(LEX_THREADS) void lexicon_kernel_v1(const uint8_t *blob, const uint64_t *offsets,
                                                              uint64_t n, uint64_t blob_bytes,
                                                              const LexEntry *table, uint32_t mult,
                                                              double *pol_out, uint8_t *spec_out) {
    __shared__ __attribute__((aligned(16))) LexShared s;
    const uint32_t tid = threadIdx.x;
    reinterpret_cast<uint4 *>(s.table)[tid] = reinterpret_cast<const uint4 *>(table)[tid];

    const uint64_t n_tiles = (n + LEX_PPT - 1) / LEX_PPT;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t p0 = tile * LEX_PPT;
        const uint32_t np = (uint32_t)((n - p0) < LEX_PPT ? (n - p0) : LEX_PPT);
        __syncthreads(); // previous tile fully written out
        if (tid <= np) s.off[tid] = offsets[p0 + tid];
        if (tid == 0 && np == LEX_PPT) s.off[LEX_PPT] = offsets[p0 + LEX_PPT];
        s.bull[tid] = 0; s.bear[tid] = 0; s.spec[tid] = 0;
        __syncthreads();
        const uint64_t byte_begin = s.off[0], byte_end = s.off[np];

        for (uint64_t sb = byte_begin & ~(uint64_t)15; sb < byte_end; sb += LEX_SUB) {
            // ---- stage [sb-16, sb+LEX_SUB+16) -> LDS, 16 B per lane, zeros outside the blob
            {
                const uint64_t a = sb + (uint64_t)tid * 16;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (a + 16 <= blob_bytes) v = *reinterpret_cast<const uint4 *>(blob + a);
                else if (a < blob_bytes) {
                    uint32_t w[4] = {0, 0, 0, 0};
                    for (uint32_t i = 0; a + i < blob_bytes; ++i) w[i >> 2] |= (uint32_t)blob[a + i] << (8 * (i & 3));
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
                reinterpret_cast<uint4 *>(s.text)[1 + tid] = v;
                if (tid < 2) {
                    // halo: tid 0 -> 16 bytes before sb, tid 1 -> 16 bytes after the sub-tile
                    const bool left = tid == 0;
                    uint4 h = make_uint4(0, 0, 0, 0);
                    if (left) {
                        if (sb >= 16) h = *reinterpret_cast<const uint4 *>(blob + sb - 16);
                    } else {
                        const uint64_t r = sb + LEX_SUB;
                        if (r + 16 <= blob_bytes) h = *reinterpret_cast<const uint4 *>(blob + r);
                        else if (r < blob_bytes) {
                            uint32_t w[4] = {0, 0, 0, 0};
                            for (uint32_t i = 0; r + i < blob_bytes; ++i) w[i >> 2] |= (uint32_t)blob[r + i] << (8 * (i & 3));
                            h = make_uint4(w[0], w[1], w[2], w[3]);
                        }
                    }
                    reinterpret_cast<uint4 *>(s.text)[left ? 0 : 1 + LEX_THREADS] = h;
                }
            }
            __syncthreads();

            const uint64_t c0 = sb + (uint64_t)tid * 16;
            const uint64_t lo = c0 > byte_begin ? c0 : byte_begin;
            const uint64_t hi = (c0 + 16) < byte_end ? (c0 + 16) : byte_end;
            if (lo < hi) {
                // post containing lo: largest j with off[j] <= lo
                uint32_t jl = 0, jr = np; // invariant off[jl] <= lo < off[jr]
                while (jr - jl > 1) {
                    uint32_t mid = (jl + jr) >> 1;
                    if (s.off[mid] <= lo) jl = mid; else jr = mid;
                }
                uint32_t j = jl;
                // window words: bytes [c0-4, c0+32)
                const uint32_t wbase = tid * 4 + 3;
                uint32_t W[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) W[i] = s.text[wbase + i];
                bool special = false;
#pragma unroll
                for (int i = 0; i < 9; ++i) special = special || swar_has_byte(W[i], 0xAAu) || swar_has_byte(W[i], 0xB0u);
                if (special) {
                    lex_slow_chunk(blob, lo, hi, j, s, mult);
                } else {
                    // alnum bit per byte of [c0-1, c0+16): bit 0 = byte c0-1
                    uint32_t am[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) { uint32_t a80; (void)swar_lower_alnum(W[i], &a80); am[i] = swar_movemask(a80); }
                    const uint32_t cand = am[1] | (am[2] << 4) | (am[3] << 8) | (am[4] << 12); // bytes c0..c0+15
                    const uint32_t prev = ((am[0] >> 3) & 1u) | (cand << 1);
                    uint32_t starts = cand & ~prev;
                    // a post's first byte starts a token whatever precedes it
                    {
                        uint32_t jj = j;
                        uint64_t e = s.off[jj + 1];
                        while (e < hi) {
                            if (e >= c0) starts |= cand & (1u << (uint32_t)(e - c0));
                            ++jj;
                            e = s.off[jj + 1];
                        }
                        if (s.off[j] >= c0 && s.off[j] < hi) starts |= cand & (1u << (uint32_t)(s.off[j] - c0));
                    }
                    // keep [lo, hi)
                    starts &= ~((1u << (uint32_t)(lo - c0)) - 1u);
                    if (hi - c0 < 16) starts &= (1u << (uint32_t)(hi - c0)) - 1u;
                    while (starts) {
                        const uint32_t b = __builtin_ctz(starts);
                        starts &= starts - 1;
                        const uint64_t pos = c0 + b;
                        while (pos >= s.off[j + 1]) ++j;
                        const uint64_t pend = s.off[j + 1];
                        // 12 bytes from pos, out of LDS (dynamic index)
                        const uint32_t bi = b + 4, wi = wbase + (bi >> 2), sh = bi & 3u;
                        const uint32_t x0 = s.text[wi], x1 = s.text[wi + 1], x2 = s.text[wi + 2], x3 = s.text[wi + 3];
                        uint32_t t0 = __builtin_amdgcn_alignbyte(x1, x0, sh);
                        uint32_t t1 = __builtin_amdgcn_alignbyte(x2, x1, sh);
                        uint32_t t2 = __builtin_amdgcn_alignbyte(x3, x2, sh);
                        uint32_t a0, a1, a2;
                        t0 = swar_lower_alnum(t0, &a0);
                        t1 = swar_lower_alnum(t1, &a1);
                        t2 = swar_lower_alnum(t2, &a2);
                        const uint32_t m12 = swar_movemask(a0) | (swar_movemask(a1) << 4) | (swar_movemask(a2) << 8);
                        uint32_t len = __builtin_ctz(~m12); // >= 1, <= 12
                        const uint64_t room = pend - pos;
                        if ((uint64_t)len > room) len = (uint32_t)room;
                        if (len <= 9) {
                            const uint32_t k0 = t0 & byte_mask(len);
                            const uint32_t k1 = len > 4 ? (t1 & byte_mask(len - 4)) : 0u;
                            const uint32_t c8 = len == 9 ? (t2 & 0xFFu) : 0u;
                            lex_lookup(s, mult, k0, k1, c8, len, j, s);
                        }
                    }
                }
            }
            __syncthreads(); // LDS text is restaged next iteration
        }
        // ---- one PostSignal per post (lexicon.rs:62-72; Polarity::new is the identity on [-1,1])
        if (tid < np) {
            const double bh = (double)s.bull[tid], rh = (double)s.bear[tid];
            const double p = (bh + rh == 0.0) ? 0.0 : (bh - rh) / (bh + rh);
            pol_out[p0 + tid] = p;
            spec_out[p0 + tid] = (uint8_t)(s.spec[tid] != 0);
        }
    }
}

// =====================================================================================
// v2 scan: 64 bytes per lane, cheap per-token filter, dense candidate pass.
//
// PMC on v1 (16 bytes per lane): 794 VALU wave-instructions per 1 KiB of text, VALU-bound at 10 % of
// the HBM roof.  Three things made it so: (1) per-lane fixed work (window load, SWAR masks, the
// post binary search) paid per 16 bytes; (2) the full token extraction (4 LDS reads + 3 SWAR
// lowercase/alnum passes) paid for EVERY token although 97 % are not lexicon words; (3) with 64
// lanes each holding a token, some lane almost always needs the expensive path, so a per-token
// early-out does not help a wave.  v2: (1) a lane owns 64 bytes; (2) tokens are screened by length
// (2..9, from the alnum bit mask alone) and by a 2048-bit Bloom filter on their first two
// case-folded chars (one ds_read2 + one ds_read); (3) survivors (~5 %) are queued per wave in LDS
// and looked up afterwards with all lanes busy.
#define LX_CH 64                      // bytes per lane
#define LX_SUB (LEX_THREADS * LX_CH)  // 16 KiB sub-tile
#define LX_PPT 512                    // posts per workgroup tile
#define LX_QCAP 160                   // candidate queue entries per wave and sub-tile

struct Lex2Shared {
    LexEntry table[LEX_SLOTS];
    uint32_t off[LX_PPT + 1];          // post offsets relative to the tile's first byte
    uint32_t bull[LX_PPT], bear[LX_PPT], spec[LX_PPT];
    // [g0-16, g0+LX_SUB+16) with ONE PAD DWORD after every 64-byte lane chunk: lane c's chunk starts at
    // dword 4 + 17c, so lanes reading the same offset of their chunks hit 32 different banks (an
    // unpadded 64-byte lane stride put them on 2 banks: 74 % of the LDS cycles were conflicts).
    uint32_t text[(LX_SUB + 32) / 4 + LEX_THREADS];
    uint32_t q_cnt[LEX_THREADS / 64];
    uint2 queue[LEX_THREADS / 64][LX_QCAP];
};

// 256-bit Bloom filter on a token's first two case-folded chars, held in 8 registers per lane
// (39 words -> <= 15 % false positives; the dense pass resolves them exactly).
// logical dword D of the staged text (D = 0..3 left halo, 4.. the sub-tile) -> physical LDS dword
__device__ __forceinline__ uint32_t lx_phys(uint32_t D) { return D + ((D - 4u) >> 4); } // D >= 4
__device__ __forceinline__ uint32_t lex_bloom_slot(uint32_t two_chars_folded) {
    return (two_chars_folded * 0x9E3779B1u) >> 24; // 8 bits
}

template <class Sh>
__device__ __forceinline__ void lex2_hit(Sh &s, uint32_t mult, uint32_t k0, uint32_t k1, uint32_t c8,
                                         uint32_t len, uint32_t post) {
    const uint32_t c8_len = c8 | (len << 8);
    const LexEntry e = s.table[lex_hash(k0, k1, c8_len, mult)];
    if (e.flags != 0 && e.k0 == k0 && e.k1 == k1 && e.c8_len == c8_len) {
        if (e.flags & 1u) atomicAdd(&s.bull[post], 1u);
        if (e.flags & 2u) atomicAdd(&s.bear[post], 1u);
        if (e.flags & 4u) atomicOr(&s.spec[post], 1u);
    }
}

// Exact per-char path (possible U+212A / U+0130 nearby); positions relative to the tile's first byte.
template <class Sh>
__device__ void lex2_slow_chunk(const uint8_t *tb, uint32_t lo, uint32_t hi, uint32_t j, Sh &s,
                                uint32_t mult) {
    for (uint32_t pos = lo; pos < hi; ++pos) {
        while (pos >= s.off[j + 1]) ++j;
        const uint32_t pstart = s.off[j], pend = s.off[j + 1];
        const uint32_t b = tb[pos];
        if ((b & 0xC0u) == 0x80u) continue;
        uint32_t first;
        if (b < 0x80u) first = (b - 'A' < 26u) ? b + 32u : b;
        else if (b == 0xE2u && pos + 2 < pend && tb[pos + 1] == 0x84u && tb[pos + 2] == 0xAAu) first = 'k';
        else if (b == 0xC4u && pos + 1 < pend && tb[pos + 1] == 0xB0u) first = 'i';
        else first = 0;
        if (!lex_is_alnum(first)) continue;
        if (pos > pstart) {
            uint32_t p = pos - 1;
            while (p > pstart && (tb[p] & 0xC0u) == 0x80u) --p;
            const uint32_t pb = tb[p];
            uint32_t last;
            if (pb < 0x80u) last = (pb - 'A' < 26u) ? pb + 32u : pb;
            else if (pb == 0xE2u && p + 3 == pos && tb[p + 1] == 0x84u && tb[p + 2] == 0xAAu) last = 'k';
            else last = 0;
            if (lex_is_alnum(last)) continue;
        }
        uint32_t len = 0, k0 = 0, k1 = 0, c8 = 0, p = pos;
        while (p < pend) {
            uint32_t c, adv;
            bool ends = false;
            const uint32_t bb = tb[p];
            if (bb < 0x80u) { c = (bb - 'A' < 26u) ? bb + 32u : bb; adv = 1; }
            else if (bb == 0xE2u && p + 2 < pend && tb[p + 1] == 0x84u && tb[p + 2] == 0xAAu) { c = 'k'; adv = 3; }
            else if (bb == 0xC4u && p + 1 < pend && tb[p + 1] == 0xB0u) { c = 'i'; adv = 2; ends = true; }
            else break;
            if (!lex_is_alnum(c)) break;
            if (len < 4) k0 |= c << (8 * len);
            else if (len < 8) k1 |= c << (8 * (len - 4));
            else if (len == 8) c8 = c;
            ++len;
            p += adv;
            if (ends) break;
        }
        if (len <= 9) lex2_hit(s, mult, k0, k1, c8, len, j);
    }
}

// alnum flags of 4 ASCII bytes, one bit per byte (letters by case folding: |0x20 maps A-Z onto a-z
// and nothing else into that range; bytes >= 0x80 never match)
__device__ __forceinline__ uint32_t lex2_alnum4(uint32_t w) {
    const uint32_t hi = w & 0x80808080u;
    const uint32_t w7 = w & 0x7F7F7F7Fu;
    const uint32_t f = (swar_range(w7 | 0x20202020u, 'a', 'z') | swar_range(w7, '0', '9')) & ~hi;
    return swar_movemask(f);
}

// Look one queued candidate up: 12 bytes at LDS byte index `ti`, `len` alnum chars, post `j`.
__device__ __forceinline__ void lex2_lookup(Lex2Shared &s, uint32_t mult, uint32_t ti, uint32_t len, uint32_t j) {
    const uint32_t wi = ti >> 2, sh = ti & 3u;
    const uint32_t x0 = s.text[lx_phys(wi)], x1 = s.text[lx_phys(wi + 1)], x2 = s.text[lx_phys(wi + 2)],
                   x3 = s.text[lx_phys(wi + 3)];
    // every char inside `len` is ASCII alphanumeric: |0x20 lowercases letters and leaves digits alone
    const uint32_t t0 = __builtin_amdgcn_alignbyte(x1, x0, sh) | 0x20202020u;
    const uint32_t t1 = __builtin_amdgcn_alignbyte(x2, x1, sh) | 0x20202020u;
    const uint32_t t2 = __builtin_amdgcn_alignbyte(x3, x2, sh) | 0x20202020u;
    const uint32_t k0 = t0 & byte_mask(len);
    const uint32_t k1 = len > 4 ? (t1 & byte_mask(len - 4)) : 0u;
    const uint32_t c8 = len == 9 ? (t2 & 0xFFu) : 0u;
    lex2_hit(s, mult, k0, k1, c8, len, j);
}

// The 16-byte piece the blob ends in, zero-filled past the end.
__device__ __noinline__ uint4 lex_load_tail(const uint8_t *blob, uint64_t src, uint64_t blob_bytes) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < 16 && src + i < blob_bytes; ++i) w[i >> 2] |= (uint32_t)blob[src + i] << (8 * (i & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Per-workgroup raw sums of SpeculationEngine::social_summary (speculation_engine.rs:76-97); the host adds the
// workgroups' partials in block order.
struct SumPartial {
    unsigned long long src0, src1, bull, bear, neu, spec;
    double psum;
    double pad;
};

// pol_out / spec_out may be null when `partials` is given (the A4 reduction fused into the scan: SURVEY 8d, "0 out if
// fused with the A4 reduction"): then nothing per post is written at all.  partials != null: every workgroup leaves
// its raw sums (sources[i] != 0 counts as source 1; tau = bull/bear threshold, config.rs:21).  The f64 polarity sum
// has a fixed shape -- per tile: thread t adds its posts (t, t + 256), then the wave tree; a wave adds its tiles in
// order; then the four waves in order -- so it is bitwise reproducible for a given grid.
// (launch bound 4 waves per SIMD = 128 VGPRs: the kernel sat at exactly 128 before the fused epilogue)
// DBG (ablation builds only): 1 staging only, 2 + windows and alnum masks, 3 + token starts / post boundaries,
// 4 + the token loop up to the length screen, 5 + Bloom screen and queue (no look-ups); 0 = the product.
template <int DBG>
__global__ __launch_bounds__(LEX_THREADS, 4) void lexicon_kernel(const uint8_t *blob, const uint64_t *offsets,
                                                              uint64_t n, uint64_t blob_bytes,
                                                              const LexEntry *table, const uint32_t *bloom,
                                                              uint32_t mult, double *pol_out, uint8_t *spec_out,
                                                              const uint8_t *sources, double tau, SumPartial *partials,
                                                              uint32_t ppt) {
    __shared__ __attribute__((aligned(16))) Lex2Shared s;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the summary's running sums live in LDS, one set per wave (loop-carried registers cost the scan a wave of occupancy:
    // 136 instead of 128 VGPRs, 1.29 -> 1.51 ms at 10M posts)
    __shared__ uint32_t r_u[5][LEX_THREADS / 64];
    __shared__ double r_d[LEX_THREADS / 64];
    if (tid < 5 * (LEX_THREADS / 64)) (&r_u[0][0])[tid] = 0u;
    if (tid < LEX_THREADS / 64) r_d[tid] = 0.0;
    reinterpret_cast<uint4 *>(s.table)[tid] = reinterpret_cast<const uint4 *>(table)[tid];
    uint32_t bl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bl[i] = bloom[i];

    const uint64_t n_tiles = (n + ppt - 1) / ppt; // ppt <= LX_PPT posts per tile: fewer for small batches (see the launcher)
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t p0 = tile * ppt;
        const uint32_t np = (uint32_t)((n - p0) < ppt ? (n - p0) : ppt);
        __syncthreads(); // previous tile fully written out
        const uint64_t byte_begin = offsets[p0];
        for (uint32_t i = tid; i <= np; i += LEX_THREADS) s.off[i] = (uint32_t)(offsets[p0 + i] - byte_begin);
        for (uint32_t i = tid; i < LX_PPT; i += LEX_THREADS) { s.bull[i] = 0; s.bear[i] = 0; s.spec[i] = 0; }
        __syncthreads();
        const uint32_t n_bytes = s.off[np];
        const uint8_t *tb = blob + byte_begin;
        // sub-tiles start 16-byte aligned in the blob: rel position of sub-tile start may be negative
        const uint32_t head = (uint32_t)(byte_begin & 15u); // bytes of the first sub-tile before byte_begin
        for (uint32_t sb = 0; sb < n_bytes + head; sb += LX_SUB) {
            // this sub-tile covers tile-relative positions [sb - head, sb - head + LX_SUB)
            const uint64_t g0 = byte_begin - head + sb; // absolute, 16-byte aligned
            // ---- stage [g0-16, g0+LX_SUB+16) -> LDS, 16 B per lane per step, zeros outside the blob.
            // The loads are not predicated (a unit outside the blob reads the blob's first 16 bytes and is
            // zeroed afterwards), so a lane's five are in flight together instead of each waiting at the
            // end of its own branch.
            constexpr uint32_t kUnits = LX_SUB / 16 + 2;
            constexpr uint32_t kSteps = (kUnits + LEX_THREADS - 1) / LEX_THREADS;
            uint4 xs[kSteps];
            if (blob_bytes >= 16) {
#pragma unroll
                for (uint32_t k = 0; k < kSteps; ++k) {
                    const uint64_t a = g0 + (uint64_t)(tid + k * LEX_THREADS) * 16; // absolute address + 16 (slot 0 = g0-16)
                    const bool whole = a >= 16 && a <= blob_bytes;                  // [a-16, a) inside the blob
                    xs[k] = *reinterpret_cast<const uint4 *>(blob + (whole ? a - 16 : 0));
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < kSteps; ++k) {
                const uint32_t v = tid + k * LEX_THREADS;
                if (v >= kUnits) continue;
                const uint64_t a = g0 + (uint64_t)v * 16;
                uint4 x = make_uint4(0, 0, 0, 0);
                if (a >= 16) {
                    const uint64_t src = a - 16;
                    if (src + 16 <= blob_bytes) x = xs[k];
                    else if (src < blob_bytes) x = lex_load_tail(blob, src, blob_bytes); // the piece the blob ends in
                }
                const uint32_t D = 4u * v; // logical dword of this 16-byte unit
                const uint32_t P = v == 0 ? 0u : lx_phys(D);
                s.text[P] = x.x; s.text[P + 1] = x.y; s.text[P + 2] = x.z; s.text[P + 3] = x.w;
            }
            if (lane == 0) s.q_cnt[wv] = 0;
            __syncthreads();
            if (DBG == 1) { if (s.text[tid] == 0xDEADBEEFu) s.bull[0] = 1; __syncthreads(); continue; }

            // lane chunk: tile-relative positions [c0, c0+64); may start before 0 in the first sub-tile
            const int64_t c0s = (int64_t)sb - head + (int64_t)tid * LX_CH;
            const int64_t lo_s = c0s > 0 ? c0s : 0;
            const int64_t hi_s = (c0s + LX_CH) < (int64_t)n_bytes ? (c0s + LX_CH) : (int64_t)n_bytes;
            if (lo_s < hi_s) {
                const uint32_t lo = (uint32_t)lo_s, hi = (uint32_t)hi_s;
                const uint32_t tbase = 16 + tid * LX_CH; // LDS byte index of the chunk's first byte
                // post containing lo: largest j with off[j] <= lo
                uint32_t jl = 0, jr = np;
                while (jr - jl > 1) {
                    const uint32_t mid = (jl + jr) >> 1;
                    if (s.off[mid] <= lo) jl = mid; else jr = mid;
                }
                uint32_t j = jl;
                // window: bytes [c0-16, c0+80) = 24 dwords (six aligned 16-byte LDS reads)
                uint32_t W[24];
                {
                    const uint32_t pc = 4u + 17u * tid; // physical dword of this lane's chunk
                    const uint32_t ph = tid == 0 ? 0u : pc - 5u; // the 4 dwords before it (skip the pad)
#pragma unroll
                    for (int i = 0; i < 4; ++i) W[i] = s.text[ph + i];
#pragma unroll
                    for (int i = 0; i < 16; ++i) W[4 + i] = s.text[pc + i];
#pragma unroll
                    for (int i = 0; i < 4; ++i) W[20 + i] = s.text[pc + 17 + i]; // next chunk / right halo
                }
                uint32_t any = 0;
#pragma unroll
                for (int i = 3; i < 23; ++i) any |= W[i];
                bool special = false;
                if (any & 0x80808080u) {
#pragma unroll
                    for (int i = 3; i < 23; ++i) special = special || swar_has_byte(W[i], 0xAAu) || swar_has_byte(W[i], 0xB0u);
                }
                if (special) {
                    lex2_slow_chunk(tb, lo, hi, j, s, mult);
                } else {
                    // alnum bit per byte: `cand` = my 64 bytes, `ext` = the 12 after, prev = the byte before
                    uint64_t cand = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) cand |= (uint64_t)lex2_alnum4(W[4 + i]) << (4 * i);
                    const uint64_t ext = (uint64_t)lex2_alnum4(W[20]) | ((uint64_t)lex2_alnum4(W[21]) << 4) |
                                         ((uint64_t)lex2_alnum4(W[22]) << 8);
                    const uint64_t prevbit = (lex2_alnum4(W[3]) >> 3) & 1u;
                    uint64_t starts = cand & ~((cand << 1) | prevbit);
                    if (DBG == 2) { if ((starts ^ ext) == 0xDEADBEEFull) s.bull[0] = 1; goto lane_done; }
                    // a post's first byte starts a token whatever precedes it
                    {
                        uint32_t jj = j;
                        uint32_t e = s.off[jj + 1];
                        while (e < hi) {
                            if ((int64_t)e >= c0s) starts |= cand & (1ull << (uint32_t)((int64_t)e - c0s));
                            ++jj;
                            e = s.off[jj + 1];
                        }
                        if ((int64_t)s.off[j] >= c0s && s.off[j] < hi) starts |= cand & (1ull << (uint32_t)((int64_t)s.off[j] - c0s));
                    }
                    // keep [lo, hi)
                    const uint32_t lb = (uint32_t)((int64_t)lo - c0s), hb = (uint32_t)((int64_t)hi - c0s);
                    if (lb) starts &= ~((1ull << lb) - 1ull);
                    if (hb < 64) starts &= (1ull << hb) - 1ull;
                    uint32_t pend = s.off[j + 1]; // end of the current post, kept in a register
                    if (DBG == 3) { if ((starts ^ pend) == 0xDEADBEEFull) s.bull[0] = 1; goto lane_done; }
                    while (starts) {
                        const uint32_t b = __builtin_ctzll(starts);
                        starts &= starts - 1;
                        // run of alnum bytes from b: bits of cand above b, then ext
                        uint64_t x = cand >> b;
                        if (b) x |= ext << (64 - b);
                        uint32_t len = (uint32_t)__builtin_ctzll(~x);
                        const uint32_t pos = (uint32_t)(c0s + b);
                        while (pos >= pend) { ++j; pend = s.off[j + 1]; }
                        const uint32_t room = pend - pos;
                        if (len > room) len = room;
                        if (len < 2 || len > 9) continue; // lexicon words are 2..9 chars
                        if (DBG == 4) { if ((len ^ pos) == 0xDEADBEEFu) s.bull[0] = 1; continue; }
                        // first two chars, case-folded, against the Bloom filter (the only LDS access here)
                        const uint32_t ti = tbase + b, wi = ti >> 2;
                        const uint32_t y0 = s.text[lx_phys(wi)], y1 = s.text[lx_phys(wi + 1)];
                        const uint32_t two = (__builtin_amdgcn_alignbyte(y1, y0, ti & 3u) & 0xFFFFu) | 0x2020u;
                        const uint32_t slot = lex_bloom_slot(two);
                        const uint32_t sel = slot >> 5;
                        const uint32_t lo4 = (sel & 1u) ? ((sel & 2u) ? bl[3] : bl[1]) : ((sel & 2u) ? bl[2] : bl[0]);
                        const uint32_t hi4 = (sel & 1u) ? ((sel & 2u) ? bl[7] : bl[5]) : ((sel & 2u) ? bl[6] : bl[4]);
                        const uint32_t bw = (sel & 4u) ? hi4 : lo4;
                        if (!((bw >> (slot & 31u)) & 1u)) continue;
                        const uint32_t qp = atomicAdd(&s.q_cnt[wv], 1u);
                        if (qp < LX_QCAP) s.queue[wv][qp] = make_uint2(ti | (len << 16), j);
                        else lex2_lookup(s, mult, ti, len, j); // queue full: look it up in place
                    }
                }
            lane_done:;
            }
            // ---- dense pass over this wave's queue (LDS ops of one wave complete in order)
            if (DBG != 5) {
                uint32_t nq = s.q_cnt[wv];
                if (nq > LX_QCAP) nq = LX_QCAP;
                for (uint32_t c = lane; c < nq; c += 64) {
                    const uint2 e = s.queue[wv][c];
                    lex2_lookup(s, mult, e.x & 0xFFFFu, e.x >> 16, e.y);
                }
            }
            __syncthreads(); // LDS text is restaged next iteration
        }
        // ---- one PostSignal per post (lexicon.rs:62-72; Polarity::new is the identity on [-1,1])
        uint32_t a_src1 = 0, a_bull = 0, a_bear = 0, a_neu = 0, a_spec = 0; // this thread's posts of THIS tile
        double a_psum = 0.0;
        for (uint32_t i = tid; i < np; i += LEX_THREADS) {
            const double bh = (double)s.bull[i], rh = (double)s.bear[i];
            const double p = (bh + rh == 0.0) ? 0.0 : (bh - rh) / (bh + rh);
            const bool sp = s.spec[i] != 0;
            if (pol_out) pol_out[p0 + i] = p;
            if (spec_out) spec_out[p0 + i] = (uint8_t)sp;
            if (partials) { // speculation_engine.rs:81-97, on the signal just computed
                a_psum += p;
                if (p > tau) ++a_bull; else if (p < -tau) ++a_bear; else ++a_neu;
                a_spec += sp ? 1u : 0u;
                if (sources) a_src1 += sources[p0 + i] != 0;
            }
        }
        if (partials) { // fold the tile into the wave's running sums (fixed order: tiles in sequence)
            uint32_t v5[5] = {a_src1, a_bull, a_bear, a_neu, a_spec};
#pragma unroll
            for (int k5 = 0; k5 < 5; ++k5) { const uint32_t r = oi_wave_sum(v5[k5]); if (lane == 0) r_u[k5][wv] += r; }
            const double d = oi_wave_sum(a_psum);
            if (lane == 0) r_d[wv] += d;
        }
    }
    if (partials) {
        __syncthreads();
        if (tid == 0) {
            unsigned long long t[5] = {0, 0, 0, 0, 0};
            double ds = 0.0;
            for (int ww = 0; ww < LEX_THREADS / 64; ++ww) {
                for (int k5 = 0; k5 < 5; ++k5) t[k5] += r_u[k5][ww];
                ds += r_d[ww];
            }
            SumPartial o;
            o.src1 = t[0]; o.bull = t[1]; o.bear = t[2]; o.neu = t[3]; o.spec = t[4];
            o.src0 = sources ? (t[1] + t[2] + t[3]) - t[0] : 0; // posts of this workgroup not from source 1
            o.psum = ds; o.pad = 0.0;
            partials[blockIdx.x] = o;
        }
    }
}

// ---- third generation ---------------------------------------------------------------------------------
// The v2 scan was VALU-issue-bound (PMC, 10M posts: 1574 VALU wave-instructions per 4 KiB wave chunk, 82 % of the issue
// slots; ladder: staging 0.34 ms, alnum windows +0.20, token loop +0.63, look-ups +0.12).  v3 removes instructions:
//  (1) every 16-byte unit is classified ONCE, by the lane that stages it (v2 re-classified a 20-byte halo per lane and
//      read a 24-dword window back from LDS): four range tests folded into one XOR chain, the per-byte flags gathered
//      with two v_dot4_u32_u8 per 8 bytes instead of a multiply-shift movemask per dword; the 16 alnum bits go to a
//      bitmap in LDS next to the (now unpadded, ds_write_b128) text;
//  (2) post starts are a second bitmap (one LDS atomic per post and sub-tile): "a post's first byte starts a token and
//      its end cuts one" become mask operations -- no binary search and no post walk in the token loop; the post of a
//      token is looked up only for the few that are lexicon words;
//  (3) the token loop works on 32-bit masks (ffbl, alignbit) instead of 64-bit ones, reads the two Bloom chars as
//      bytes, takes the Bloom word from LDS (v2: a 7-select chain over 8 registers), and compacts candidates with a
//      ballot (no LDS atomic with return).
#define L3_SUB 16384u
#define L3_UNITS (L3_SUB / 16u + 2u)       // halo units of 16 bytes: [g0-16, g0+L3_SUB+16)
#define L3_BITW 516u                       // u32 words of a bit-per-byte map: unit v is u16 number 3 + v
#define L3_BLOOM_MUL 0x9E3779u             // 24-bit: v_mul_u32_u24 is full rate
#define L3_SLOTS 4u                        // candidates a lane can park per sub-tile (more: looked up in place)

struct Lex3Shared {
    LexEntry table[LEX_SLOTS];
    uint32_t off[LX_PPT + 1];
    uint32_t bull[LX_PPT], bear[LX_PPT], spec[LX_PPT];
    uint4 text[L3_UNITS];                  // bytes [g0-16, g0+L3_SUB+16), unit v at text[v]
    uint32_t abits[L3_BITW];               // 1 = ASCII alphanumeric byte
    uint32_t tbits[2][L3_BITW];            // 1 = first byte of a post (or the tile's end); double buffered
    uint32_t spchunk[2][LEX_THREADS / 32]; // lane chunk needs the exact per-char path
    uint32_t bloom[8];
    uint32_t jchunk[LEX_THREADS];          // post holding the first byte of lane chunk c
    uint32_t cand[LEX_THREADS / 64][L3_SLOTS][64]; // per wave: lane l's k-th candidate, then the same region compacted
};

__device__ __forceinline__ uint32_t lex3_bloom_hash(uint32_t two_folded) { return __umul24(two_folded, L3_BLOOM_MUL); } // slot = bits 24..31

// Look one candidate up: `len` alnum chars at LDS text byte `ti`; pos = its tile-relative byte position.
__device__ __forceinline__ void lex3_lookup(Lex3Shared &s, uint32_t mult, uint32_t ti, uint32_t len, uint32_t pos) {
    const uint32_t *t32 = reinterpret_cast<const uint32_t *>(s.text);
    const uint32_t wi = ti >> 2, sh = ti & 3u;
    const uint32_t x0 = t32[wi], x1 = t32[wi + 1], x2 = t32[wi + 2], x3 = t32[wi + 3];
    const uint32_t t0 = __builtin_amdgcn_alignbyte(x1, x0, sh) | 0x20202020u;
    const uint32_t t1 = __builtin_amdgcn_alignbyte(x2, x1, sh) | 0x20202020u;
    const uint32_t t2 = __builtin_amdgcn_alignbyte(x3, x2, sh) | 0x20202020u;
    const uint32_t k0 = t0 & byte_mask(len);
    const uint32_t k1 = len > 4 ? (t1 & byte_mask(len - 4)) : 0u;
    const uint32_t c8_len = (len == 9 ? (t2 & 0xFFu) : 0u) | (len << 8);
    const LexEntry e = s.table[lex_hash(k0, k1, c8_len, mult)];
    if (e.flags != 0 && e.k0 == k0 && e.k1 == k1 && e.c8_len == c8_len) {
        // the post holding pos: largest j with off[j] <= pos (empty posts share an offset); from its chunk's first post
        uint32_t jl = s.jchunk[(ti - 16u) >> 6];
        while (s.off[jl + 1] <= pos) ++jl; // off[np] = the tile's end > pos
        if (e.flags & 1u) atomicAdd(&s.bull[jl], 1u);
        if (e.flags & 2u) atomicAdd(&s.bear[jl], 1u);
        if (e.flags & 4u) atomicOr(&s.spec[jl], 1u);
    }
}

// DBG (ablation builds only): 1 staging + classification only, 2 + masks and token starts, 3 + token loops (no look-ups).
template <int DBG>
__global__ __launch_bounds__(LEX_THREADS, 4) void lexicon_scan_kernel(const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                                                                   uint64_t blob_bytes, const LexEntry *table,
                                                                   const uint32_t *bloom, uint32_t mult, double *pol_out,
                                                                   uint8_t *spec_out, const uint8_t *sources, double tau,
                                                                   SumPartial *partials, uint32_t ppt) {
    __shared__ __attribute__((aligned(16))) Lex3Shared s;
    __shared__ uint32_t r_u[5][LEX_THREADS / 64]; // the summary's running sums, one set per wave (see lexicon_kernel)
    __shared__ double r_d[LEX_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 5 * (LEX_THREADS / 64)) (&r_u[0][0])[tid] = 0u;
    if (tid < LEX_THREADS / 64) r_d[tid] = 0.0;
    reinterpret_cast<uint4 *>(s.table)[tid] = reinterpret_cast<const uint4 *>(table)[tid];
    if (tid < 8) s.bloom[tid] = bloom[tid];
    uint16_t *ab16 = reinterpret_cast<uint16_t *>(s.abits);
    const uint8_t *text8 = reinterpret_cast<const uint8_t *>(s.text);

    const uint64_t n_tiles = (n + ppt - 1) / ppt; // ppt <= LX_PPT posts per tile: fewer for small batches (see the launcher)
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t p0 = tile * ppt;
        const uint32_t np = (uint32_t)((n - p0) < ppt ? (n - p0) : ppt);
        __syncthreads(); // previous tile fully written out
        const uint64_t byte_begin = offsets[p0];
        // this thread's post starts (tile-relative; the tile's end counts as one), kept for every sub-tile's bitmap
        uint32_t myoff[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t i = tid + (uint32_t)k * LEX_THREADS;
            myoff[k] = 0xFFFFFFFFu;
            if (i <= np) { myoff[k] = (uint32_t)(offsets[p0 + i] - byte_begin); s.off[i] = myoff[k]; }
        }
        for (uint32_t i = tid; i < LX_PPT; i += LEX_THREADS) { s.bull[i] = 0; s.bear[i] = 0; s.spec[i] = 0; }
        for (uint32_t i = tid; i < 2 * L3_BITW; i += LEX_THREADS) (&s.tbits[0][0])[i] = 0;
        if (tid < 2 * (LEX_THREADS / 32)) (&s.spchunk[0][0])[tid] = 0;
        __syncthreads();
        const uint32_t n_bytes = s.off[np];
        const uint8_t *tb = blob + byte_begin;
        const uint32_t head = (uint32_t)(byte_begin & 15u); // sub-tiles start 16-byte aligned in the blob
        // A sub-tile's loads are issued one sub-tile ahead and stay in flight across the token loops.  They are not
        // predicated (a unit outside the blob reads the blob's first 16 bytes and is zeroed by fixup): a lane's loads
        // go out together instead of each waiting at the end of its own branch.
        constexpr uint32_t kSteps = (L3_UNITS + LEX_THREADS - 1) / LEX_THREADS;
        uint4 xs[kSteps];
        auto issue = [&](const uint64_t g0_) {
            if (g0_ >= 16 && g0_ + L3_SUB + 16 <= blob_bytes) { // the whole window lies inside the blob
                const uint4 *src = reinterpret_cast<const uint4 *>(blob + (g0_ - 16));
#pragma unroll
                for (uint32_t k = 0; k < kSteps; ++k) {
                    const uint32_t v = tid + k * LEX_THREADS;
                    xs[k] = src[v < L3_UNITS ? v : 0u];
                }
            } else if (blob_bytes >= 16) {
#pragma unroll
                for (uint32_t k = 0; k < kSteps; ++k) {
                    const uint64_t a = g0_ + (uint64_t)(tid + k * LEX_THREADS) * 16; // unit v covers [a-16, a)
                    const bool whole = a >= 16 && a <= blob_bytes;
                    xs[k] = *reinterpret_cast<const uint4 *>(blob + (whole ? a - 16 : 0));
                }
            }
        };
        auto fixup = [&](const uint64_t g0_, uint4 (&x)[kSteps]) {
            if (g0_ >= 16 && g0_ + L3_SUB + 16 <= blob_bytes) return;
#pragma unroll
            for (uint32_t k = 0; k < kSteps; ++k) {
                const uint64_t a = g0_ + (uint64_t)(tid + k * LEX_THREADS) * 16;
                uint4 z = make_uint4(0, 0, 0, 0);
                if (a >= 16) {
                    const uint64_t src = a - 16;
                    if (src + 16 <= blob_bytes) z = x[k];
                    else if (src < blob_bytes) z = lex_load_tail(blob, src, blob_bytes); // the piece the blob ends in
                }
                x[k] = z;
            }
        };
        issue(byte_begin - head);
        uint32_t it = 0;
        for (uint32_t sb = 0; sb < n_bytes + head; sb += L3_SUB, ++it) {
            const uint32_t cur = it & 1u, nxt = cur ^ 1u;
            const uint64_t g0 = byte_begin - head + sb; // absolute, 16-byte aligned; tile-relative position sb - head
            // ---- stage [g0-16, g0+L3_SUB+16): the loads were issued a sub-tile ago (or before the loop)
            fixup(g0, xs);
            // post-start bits of this sub-tile (bit 48 + r for the byte r past g0-16); the other buffer is cleared for the next
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t r = myoff[k] + 16u + head - sb; // wraps far out of range for posts before the window
                if (myoff[k] != 0xFFFFFFFFu && r < L3_SUB + 32u) atomicOr(&s.tbits[cur][(48u + r) >> 5], 1u << ((48u + r) & 31u));
            }
            for (uint32_t i = tid; i < L3_BITW; i += LEX_THREADS) s.tbits[nxt][i] = 0;
            if (tid < LEX_THREADS / 32) s.spchunk[nxt][tid] = 0;
#pragma unroll
            for (uint32_t k = 0; k < kSteps; ++k) {
                const uint32_t v = tid + k * LEX_THREADS;
                if (v >= L3_UNITS) continue;
                const uint4 x = xs[k];
                s.text[v] = x;
                ab16[3u + v] = (uint16_t)oi_alnum16(x);
                if ((x.x | x.y | x.z | x.w) & 0x80808080u) { // a 0xAA / 0xB0 byte may be part of U+212A / U+0130
                    bool sp = false;
                    const uint32_t w4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) sp = sp || swar_has_byte(w4[i], 0xAAu) || swar_has_byte(w4[i], 0xB0u);
                    if (sp) { // lane chunk c looks at units 4c .. 4c+5 (16 bytes either side of its own four)
                        const uint32_t c_hi = v >> 2, c_lo = v >= 5 ? (v - 2) >> 2 : 0u;
                        for (uint32_t c = c_lo; c <= c_hi && c < LEX_THREADS; ++c) atomicOr(&s.spchunk[cur][c >> 5], 1u << (c & 31u));
                    }
                }
            }
            if (sb + L3_SUB < n_bytes + head) issue(g0 + L3_SUB); // the next sub-tile's window
            __syncthreads();
            if (DBG == 1) { if (s.abits[tid] == 0xDEADBEEFu) s.bull[0] = 1; __syncthreads(); continue; }

            // ---- lane chunk: tile-relative positions [c0s, c0s+64) (negative in front of the tile's first byte)
            const int64_t c0s = (int64_t)sb - head + (int64_t)tid * 64;
            const int64_t lo_s = c0s > 0 ? c0s : 0;
            const int64_t hi_s = (c0s + 64) < (int64_t)n_bytes ? (c0s + 64) : (int64_t)n_bytes;
            uint32_t S0 = 0, S1 = 0, E0 = 0, E1 = 0, En = 0; // token starts / token ends (non-alnum or a post's first byte)
            if (lo_s < hi_s) {
                const uint32_t lo = (uint32_t)lo_s, hi = (uint32_t)hi_s;
                uint32_t jl = 0, jr = np; // post containing lo: largest j with off[j] <= lo
                while (jr - jl > 1) {
                    const uint32_t mid = (jl + jr) >> 1;
                    if (s.off[mid] <= lo) jl = mid; else jr = mid;
                }
                s.jchunk[tid] = jl; // read by this wave's look-ups only
                if ((s.spchunk[cur][tid >> 5] >> (tid & 31u)) & 1u) {
                    lex2_slow_chunk(tb, lo, hi, jl, s, mult);
                } else {
                    const uint16_t *tb16 = reinterpret_cast<const uint16_t *>(s.tbits[cur]);
                    const uint2 a = *reinterpret_cast<const uint2 *>(ab16 + 4u * tid + 4u);
                    const uint2 t = *reinterpret_cast<const uint2 *>(tb16 + 4u * tid + 4u);
                    const uint32_t a_prev = ab16[4u * tid + 3u], a_next = ab16[4u * tid + 8u], t_next = tb16[4u * tid + 8u];
                    E0 = ~a.x | t.x;
                    E1 = ~a.y | t.y;
                    En = ~a_next | t_next | 0xFFFF0000u; // 16 bytes of look-ahead: enough to see that a token is longer than 9
                    S0 = a.x & (~((a.x << 1) | (a_prev >> 15)) | t.x);
                    S1 = a.y & (~((a.y << 1) | (a.x >> 31)) | t.y);
                    const uint32_t lb = (uint32_t)(lo_s - c0s), hb = (uint32_t)(hi_s - c0s); // keep [lo, hi)
                    const uint64_t keep = (hb < 64 ? (1ull << hb) - 1ull : ~0ull) & ~((1ull << lb) - 1ull);
                    S0 &= (uint32_t)keep;
                    S1 &= (uint32_t)(keep >> 32);
                }
            }
            if (DBG == 2) { if ((S0 ^ S1 ^ E0 ^ E1 ^ En) == 0xDEADBEEFu) s.bull[0] = 1; __syncthreads(); continue; }
            // ---- token loops, one per 32-byte half; every lane of the wave stays in them (qn is wave-uniform)
            const uint32_t pos_base = sb - head - 16u; // tile-relative position of text byte 0 (mod 2^32)
            // one token of a 32-byte half: its length from the end markers, the length screen, the Bloom screen
            auto token = [&](uint32_t &S, const uint32_t Ecur, const uint32_t Enext, const uint32_t tbase, bool &pass, uint32_t &ent) {
                pass = false;
                ent = 0;
                if (S) {
                    const uint32_t b = (uint32_t)__builtin_ctz(S);
                    S &= S - 1u;
                    const uint32_t y = __builtin_amdgcn_alignbit(Enext, Ecur, b) >> 1; // end markers after the first char
                    const uint32_t len = (uint32_t)__ffs((int)y);                      // chars up to the first end; 0: none in sight
                    if (len - 2u < 8u) { // lexicon words are 2..9 chars
                        const uint32_t ti = tbase + b;
                        const uint32_t two = ((uint32_t)text8[ti] | ((uint32_t)text8[ti + 1] << 8)) | 0x2020u;
                        const uint32_t h = lex3_bloom_hash(two);
                        pass = (s.bloom[h >> 29] >> ((h >> 24) & 31u)) & 1u;
                        ent = ti | (len << 16);
                    }
                }
            };
            // a candidate is parked in the lane's own column (no ballot, no atomic in the loop)
            uint32_t cnt = 0;
            auto push = [&](bool pass, uint32_t ent) {
                if (pass) {
                    if (cnt < L3_SLOTS) s.cand[wv][cnt][lane] = ent;
                    else if (DBG == 0) lex3_lookup(s, mult, ent & 0xFFFFu, ent >> 16, pos_base + (ent & 0xFFFFu)); // column full
                    ++cnt;
                }
            };
            // both halves in one loop: the trip count is the longer half's, and the two LDS chains overlap
            while (__any((S0 | S1) != 0u)) {
                bool pa, pb;
                uint32_t ea, eb;
                token(S0, E0, E1, 16u + tid * 64u, pa, ea);
                token(S1, E1, En, 48u + tid * 64u, pb, eb);
                push(pa, ea);
                push(pb, eb);
            }
            // ---- compact the columns (prefix sum of the 0..4 counts from three ballots), then look the candidates up with
            // all lanes busy (LDS ops of one wave complete in order; the entries pass through registers)
            if (DBG == 0) {
                cnt = cnt < L3_SLOTS ? cnt : L3_SLOTS;
                const unsigned long long below = (1ull << lane) - 1ull;
                const unsigned long long b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
                const uint32_t base = (uint32_t)__popcll(b0 & below) + 2u * (uint32_t)__popcll(b1 & below) + 4u * (uint32_t)__popcll(b2 & below);
                const uint32_t nq = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
                uint32_t mine[L3_SLOTS];
#pragma unroll
                for (uint32_t k = 0; k < L3_SLOTS; ++k) mine[k] = s.cand[wv][k][lane];
                uint32_t *flat = &s.cand[wv][0][0];
#pragma unroll
                for (uint32_t k = 0; k < L3_SLOTS; ++k)
                    if (k < cnt) flat[base + k] = mine[k];
                for (uint32_t c = lane; c < nq; c += 64) {
                    const uint32_t ent = flat[c];
                    lex3_lookup(s, mult, ent & 0xFFFFu, ent >> 16, pos_base + (ent & 0xFFFFu));
                }
            } else if (cnt == 0xDEADBEEFu) s.bull[0] = 1;
            __syncthreads(); // text and bitmaps are restaged next iteration
        }
        // ---- one PostSignal per post (lexicon.rs:62-72; Polarity::new is the identity on [-1,1])
        uint32_t a_src1 = 0, a_bull = 0, a_bear = 0, a_neu = 0, a_spec = 0; // this thread's posts of THIS tile
        double a_psum = 0.0;
        for (uint32_t i = tid; i < np; i += LEX_THREADS) {
            const double bh = (double)s.bull[i], rh = (double)s.bear[i];
            const double p = (bh + rh == 0.0) ? 0.0 : (bh - rh) / (bh + rh);
            const bool sp = s.spec[i] != 0;
            if (pol_out) pol_out[p0 + i] = p;
            if (spec_out) spec_out[p0 + i] = (uint8_t)sp;
            if (partials) { // speculation_engine.rs:81-97, on the signal just computed
                a_psum += p;
                if (p > tau) ++a_bull; else if (p < -tau) ++a_bear; else ++a_neu;
                a_spec += sp ? 1u : 0u;
                if (sources) a_src1 += sources[p0 + i] != 0;
            }
        }
        if (partials) { // fold the tile into the wave's running sums (fixed order: tiles in sequence)
            uint32_t v5[5] = {a_src1, a_bull, a_bear, a_neu, a_spec};
#pragma unroll
            for (int k5 = 0; k5 < 5; ++k5) { const uint32_t r = oi_wave_sum(v5[k5]); if (lane == 0) r_u[k5][wv] += r; }
            const double d = oi_wave_sum(a_psum);
            if (lane == 0) r_d[wv] += d;
        }
    }
    if (partials) {
        __syncthreads();
        if (tid == 0) {
            unsigned long long t[5] = {0, 0, 0, 0, 0};
            double ds = 0.0;
            for (int ww = 0; ww < LEX_THREADS / 64; ++ww) {
                for (int k5 = 0; k5 < 5; ++k5) t[k5] += r_u[k5][ww];
                ds += r_d[ww];
            }
            SumPartial o;
            o.src1 = t[0]; o.bull = t[1]; o.bear = t[2]; o.neu = t[3]; o.spec = t[4];
            o.src0 = sources ? (t[1] + t[2] + t[3]) - t[0] : 0; // posts of this workgroup not from source 1
            o.psum = ds; o.pad = 0.0;
            partials[blockIdx.x] = o;
        }
    }
}

// ---- host: word table ------------------------------------------------------------
// openintel src/adapters/analyzer/lexicon.rs:9-44 (`calls`/`squeeze` are bull+jargon, `puts` bear+jargon)
static const char *const kBull[] = {"moon", "calls", "long", "buy", "bullish", "squeeze", "breakout",
                                    "rocket", "pump", "rip", "green", "up", "rally", "bull"};
static const char *const kBear[] = {"puts", "short", "sell", "bearish", "dump", "crash", "drilling",
                                    "bagholder", "rug", "red", "down", "tank", "bear"};
static const char *const kJargon[] = {"calls", "puts", "0dte", "yolo", "leaps", "theta", "gamma", "squeeze",
                                      "otm", "itm", "strike", "iv", "delta", "vega", "contracts"};

static uint32_t host_lex_hash(uint32_t k0, uint32_t k1, uint32_t c8_len, uint32_t mult) {
    uint32_t x = k0 * 0x9E3779B1u ^ k1 * 0x85EBCA77u ^ c8_len * 0xC2B2AE3Du;
    x ^= x >> 15;
    return (x * mult) >> 24;
}

static bool build_lex_table(LexEntry *table, uint32_t *mult_out, uint32_t *bloom) {
    struct Word { uint32_t k0, k1, c8_len, flags; };
    std::vector<Word> words;
    auto add = [&](const char *w, uint32_t flag) {
        size_t len = strlen(w);
        Word x{0, 0, 0, flag};
        for (size_t i = 0; i < len; ++i) {
            uint32_t c = (uint8_t)w[i];
            if (i < 4) x.k0 |= c << (8 * i);
            else if (i < 8) x.k1 |= c << (8 * (i - 4));
            else x.c8_len |= c;
        }
        x.c8_len |= (uint32_t)len << 8;
        for (auto &o : words)
            if (o.k0 == x.k0 && o.k1 == x.k1 && o.c8_len == x.c8_len) { o.flags |= flag; return; }
        words.push_back(x);
    };
    for (auto w : kBull) add(w, 1);
    for (auto w : kBear) add(w, 2);
    for (auto w : kJargon) add(w, 4);
    memset(bloom, 0, 64 * sizeof(uint32_t));
    for (auto &w : words) { // first two chars (already lowercase; |0x20 is the kernel's case fold)
        const uint32_t slot = (((w.k0 & 0xFFFFu) | 0x2020u) * 0x9E3779B1u) >> 24;
        bloom[slot >> 5] |= 1u << (slot & 31u);
        const uint32_t h3 = (uint32_t)((uint64_t)((w.k0 & 0xFFFFu) | 0x2020u) * L3_BLOOM_MUL); // lex3_bloom_hash
        bloom[8 + (h3 >> 29)] |= 1u << ((h3 >> 24) & 31u);                                      // v3's filter: words 8..15
    }
    // smallest odd multiplier that makes the hash perfect over the 39 distinct words
    for (uint32_t mult = 1; mult < (1u << 24); mult += 2) {
        memset(table, 0, sizeof(LexEntry) * LEX_SLOTS);
        bool ok = true;
        for (auto &w : words) {
            LexEntry &e = table[host_lex_hash(w.k0, w.k1, w.c8_len, mult)];
            if (e.flags) { ok = false; break; }
            e.k0 = w.k0; e.k1 = w.k1; e.c8_len = w.c8_len; e.flags = w.flags;
        }
        if (ok) { *mult_out = mult; return true; }
    }
    return false;
}

// d_pol / d_spec may be null when `summary` is given (nothing per post is written then); summary != null: the fused A4
// reduction -- the call synchronises the stream and returns the raw sums.
int oi_launch_lexicon_fused(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n, uint64_t blob_bytes,
                            double *d_pol, uint8_t *d_spec, const uint8_t *d_sources, double tau, oi_social_counters *summary) {
    if (summary) { memset(summary, 0, sizeof(*summary)); summary->total = n; }
    if (n == 0) return OI_OK;
    OI_REQUIRE(((uintptr_t)d_blob & 15u) == 0, "lexicon: text blob must be 16-byte aligned in HBM");
    struct HostTables { LexEntry table[LEX_SLOTS]; uint32_t bloom[64]; };
    static HostTables h; // built once per process (std::call_once: contexts on several host threads share it)
    static uint32_t h_mult = 0;
    static bool built_ok = false;
    static std::once_flag once;
    std::call_once(once, [] { built_ok = build_lex_table(h.table, &h_mult, h.bloom); });
    if (!built_ok) { oi_set_error("lexicon: no perfect hash found"); return OI_ERR_STATE; }
    DevBuf &tb = ctx->buf("lex_table");
    if (!tb.p) {
        OI_CHECK(tb.ensure(sizeof(h)));
        OI_HIP_CHECK(hipMemcpyAsync(tb.p, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    }
    const LexEntry *d_table = tb.as<LexEntry>();
    const uint32_t *d_bloom = reinterpret_cast<const uint32_t *>(d_table + LEX_SLOTS);
    static const bool v1 = oi_ablation_env("OI_LEXICON_V1") != nullptr; // A/B switches: the first / second generation scans
    static const bool v2 = oi_ablation_env("OI_LEXICON_V2") != nullptr;
    (void)v2;
    // Posts per tile: LX_PPT for a batch that fills the chip; a small batch -- the reference's own call is one ticker's
    // <= 100 posts -- is cut into smaller tiles so that its sub-tiles run side by side on many CUs instead of one after the
    // other on one (1000 posts: 2 workgroups x 5 sub-tiles -> 63 workgroups x 1).  Per-post results do not depend on it.
    uint32_t ppt = LX_PPT;
    while (ppt > 16u && (n + ppt - 1) / ppt < 2ull * (uint64_t)ctx->num_cus) ppt >>= 1;
    const uint64_t n_tiles = (n + ppt - 1) / ppt;
    // (4 workgroups are resident per CU; 16 per CU over the tile loop balance the tail: 1.042 / 1.023 / 1.009 ms at 4 / 8 / 16)
    static const char *grid_s = oi_ablation_env("OI_LEX_GRID"); // A/B switch: workgroups per CU
    const uint32_t max_grid = (uint32_t)ctx->num_cus * (grid_s ? (uint32_t)atoi(grid_s) : 16u);
    const uint32_t grid = (uint32_t)(n_tiles < max_grid ? n_tiles : max_grid);
    SumPartial *d_partials = nullptr;
    if (summary) {
        DevBuf &pb = ctx->buf("lex_partials");
        OI_CHECK(pb.ensure(sizeof(SumPartial) * max_grid));
        d_partials = pb.as<SumPartial>();
    }
    {
        ProfScope ps(ctx, "lexicon");
        if (v1 && !summary) {
            const uint64_t n_tiles1 = (n + LEX_PPT - 1) / LEX_PPT;
            const uint32_t grid1 = (uint32_t)(n_tiles1 < max_grid ? n_tiles1 : max_grid);
            hipLaunchKernelGGL(lexicon_kernel_v1, dim3(grid1), dim3(LEX_THREADS), 0, ctx->stream, d_blob, d_offsets, n,
                               blob_bytes, d_table, h_mult, d_pol, d_spec);
        } else {
#ifdef OI_ABLATION
            static const char *dbg_s = oi_ablation_env("OI_LEX_DBG");
            const int dbg = dbg_s ? atoi(dbg_s) : 0;
#define LEX_GO(K, D, BL) hipLaunchKernelGGL(K<D>, dim3(grid), dim3(LEX_THREADS), 0, ctx->stream, d_blob, d_offsets, n, \
                                            blob_bytes, d_table, BL, h_mult, d_pol, d_spec, d_sources, tau, d_partials, ppt)
            if (v2) switch (dbg) { case 1: LEX_GO(lexicon_kernel, 1, d_bloom); break; case 2: LEX_GO(lexicon_kernel, 2, d_bloom); break;
                                   case 3: LEX_GO(lexicon_kernel, 3, d_bloom); break; case 4: LEX_GO(lexicon_kernel, 4, d_bloom); break;
                                   case 5: LEX_GO(lexicon_kernel, 5, d_bloom); break; default: LEX_GO(lexicon_kernel, 0, d_bloom); }
            else switch (dbg) { case 1: LEX_GO(lexicon_scan_kernel, 1, d_bloom + 8); break; case 2: LEX_GO(lexicon_scan_kernel, 2, d_bloom + 8); break;
                                case 3: LEX_GO(lexicon_scan_kernel, 3, d_bloom + 8); break; default: LEX_GO(lexicon_scan_kernel, 0, d_bloom + 8); }
#undef LEX_GO
#else
            hipLaunchKernelGGL(lexicon_scan_kernel<0>, dim3(grid), dim3(LEX_THREADS), 0, ctx->stream, d_blob, d_offsets, n,
                               blob_bytes, d_table, d_bloom + 8, h_mult, d_pol, d_spec, d_sources, tau, d_partials, ppt);
#endif
        }
        OI_HIP_CHECK(hipGetLastError());
    }
    if (!summary) return OI_OK;
    std::vector<SumPartial> hp(grid);
    OI_HIP_CHECK(hipMemcpyAsync(hp.data(), d_partials, sizeof(SumPartial) * grid, hipMemcpyDeviceToHost, ctx->stream));
    OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (uint32_t bb = 0; bb < grid; ++bb) { // block order: reproducible
        summary->by_source[0] += hp[bb].src0; summary->by_source[1] += hp[bb].src1;
        summary->bullish += hp[bb].bull; summary->bearish += hp[bb].bear; summary->neutral += hp[bb].neu;
        summary->spec_count += hp[bb].spec; summary->polarity_sum += hp[bb].psum;
    }
    return OI_OK;
}

int oi_launch_lexicon(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n,
                      uint64_t blob_bytes, double *d_pol, uint8_t *d_spec) {
    return oi_launch_lexicon_fused(ctx, d_blob, d_offsets, n, blob_bytes, d_pol, d_spec, nullptr, 0.0, nullptr);
}

// ---- social summary ---------------------------------------------------------------
// speculation_engine.rs:76-97: source histogram, polarity sum, bull/bear/neutral by
// threshold tau, speculative count.  Integer sums are exact; the f64 sum is a fixed
// tree: per-thread strided partial -> wave shuffle tree -> per-block partial, and the
// host adds the <= 1024 block partials in block order.
#define SUM_THREADS 256
#define SUM_MAX_BLOCKS 1024
__global__ __launch_bounds__(SUM_THREADS) void social_summary_kernel(const uint8_t *sources, const double *pol,
                                                                      const uint8_t *spec, uint64_t n, double tau,
                                                                      SumPartial *partials) {
    uint32_t src1 = 0, bull = 0, bear = 0, neu = 0, sp = 0, cnt = 0;
    double psum = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * SUM_THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * SUM_THREADS) {
        const double v = pol[i];
        psum += v;
        if (v > tau) ++bull; else if (v < -tau) ++bear; else ++neu;
        sp += spec[i] != 0;
        if (sources) src1 += sources[i] != 0;
        ++cnt;
    }
    __shared__ uint32_t s_u[6][SUM_THREADS / OI_WAVE];
    __shared__ double s_d[SUM_THREADS / OI_WAVE];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t v6[6] = {cnt, src1, bull, bear, neu, sp};
#pragma unroll
    for (int k = 0; k < 6; ++k) { uint32_t r = oi_wave_sum(v6[k]); if (lane == 0) s_u[k][w] = r; }
    double d = oi_wave_sum(psum);
    if (lane == 0) s_d[w] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t[6] = {0, 0, 0, 0, 0, 0};
        double ds = 0.0;
        for (int ww = 0; ww < SUM_THREADS / OI_WAVE; ++ww) {
            for (int k = 0; k < 6; ++k) t[k] += s_u[k][ww];
            ds += s_d[ww];
        }
        SumPartial p;
        p.src0 = sources ? t[0] - t[1] : 0; p.src1 = t[1]; p.bull = t[2]; p.bear = t[3]; p.neu = t[4]; p.spec = t[5];
        p.psum = ds; p.pad = 0.0;
        partials[blockIdx.x] = p;
    }
}

int oi_launch_social_summary(oi_ctx *ctx, const uint8_t *d_sources, const double *d_pol,
                             const uint8_t *d_spec, uint64_t n, double tau, oi_social_counters *out) {
    memset(out, 0, sizeof(*out));
    out->total = n;
    if (n == 0) return OI_OK;
    uint64_t blocks = (n + (uint64_t)SUM_THREADS * 8 - 1) / ((uint64_t)SUM_THREADS * 8);
    if (blocks > SUM_MAX_BLOCKS) blocks = SUM_MAX_BLOCKS;
    DevBuf &pb = ctx->buf("sum_partials");
    OI_CHECK(pb.ensure(sizeof(SumPartial) * SUM_MAX_BLOCKS));
    {
        ProfScope ps(ctx, "social_summary");
        hipLaunchKernelGGL(social_summary_kernel, dim3((uint32_t)blocks), dim3(SUM_THREADS), 0, ctx->stream,
                           d_sources, d_pol, d_spec, n, tau, pb.as<SumPartial>());
        OI_HIP_CHECK(hipGetLastError());
    }
    std::vector<SumPartial> h(blocks);
    OI_HIP_CHECK(hipMemcpyAsync(h.data(), pb.p, sizeof(SumPartial) * blocks, hipMemcpyDeviceToHost, ctx->stream));
    OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (uint64_t b = 0; b < blocks; ++b) {
        out->by_source[0] += h[b].src0; out->by_source[1] += h[b].src1;
        out->bullish += h[b].bull; out->bearish += h[b].bear; out->neutral += h[b].neu;
        out->spec_count += h[b].spec; out->polarity_sum += h[b].psum;
    }
    return OI_OK;
}

// ---- social summary per segment -----------------------------------------------------
// The reference's only batch entry (mcp/tools.rs:193-225 run_scan, :303-352 run_compare) runs application::analyze
// once per ticker; here the posts of ALL tickers go through one scan and this kernel gives every ticker (segment) its
// social_summary sums (speculation_engine.rs:76-97).  One wave per segment, segments strided over the grid.  The
// integer counters are lane-parallel (64 signals per step, ballots).  polarity_sum is added ONE SIGNAL AT A TIME in
// input order from +0.0 -- the reference's own `polarity_sum += v` loop (:83-86), so the sum and the net_sentiment made
// from it are bit-identical to the reference's, not a tree with a bound.  The chain costs ONE vector instruction per
// signal: the step's polarities, already in registers for the counters, are parked in the wave's own LDS slot and read
// back at wave-uniform addresses (a broadcast read, two signals per ds_read_b128), each added with one v_add_f64.
// (Measured at 10M posts in 100K tickers, tools/scan_bench.py: two v_readlane + add per signal 0.15-0.22 ms; the
// polarities through the scalar cache, s_load_dwordx16 + eight adds, 0.109 ms; this form 0.072 ms -- ordered sums 0.028,
// counters 0.011, loads and per-segment latency the rest, ~0.015 of it at any size (tools/r03_seg_probe.sh).  ONE segment
// of 10M posts is still one wave's 10M dependent adds, ~40 ms -- the unsegmented oi_social_summary with its tree is the
// call for that; 1M segments of 10 posts take 0.29 ms, a memory round trip per segment and wave.)
#define SEG_THREADS 256
#define SEG_ROUNDS 4 // 64-signal rounds of a segment whose loads are in flight together
__global__ __launch_bounds__(SEG_THREADS) void social_summary_segmented_kernel(const uint8_t *__restrict__ sources,
                                                                                const double *__restrict__ pol,
                                                                                const uint8_t *__restrict__ spec, uint64_t n,
                                                                                const uint64_t *__restrict__ seg, uint64_t n_seg,
                                                                                double tau, oi_social_counters *__restrict__ out, int dbg) {
    __shared__ __attribute__((aligned(16))) double seg_slot[SEG_THREADS / 64][64 * SEG_ROUNDS]; // a wave's step, in input order
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: the segment walk is scalar code
    const uint64_t n_waves = (uint64_t)gridDim.x * (SEG_THREADS / 64);
    uint64_t sg = (uint64_t)blockIdx.x * (SEG_THREADS / 64) + wv;
    if (sg >= n_seg) return;
    uint64_t nb = seg[sg], ne = seg[sg + 1];
    for (; sg < n_seg; sg += n_waves) {
        uint64_t b = nb, e = ne;
        // the NEXT segment's bounds now, used a segment later: bounds -> signals is two dependent round trips per segment
        // otherwise (the offsets of consecutive tickers share a line, but they go to different waves and CUs)
        const uint64_t nsg = sg + n_waves < n_seg ? sg + n_waves : sg;
        nb = seg[nsg];
        ne = seg[nsg + 1];
        if (e > n) e = n; // a malformed offsets array must not read past the signals (the host entry point rejects it)
        if (b > e) b = e;
        uint64_t src1 = 0, bull = 0, bear = 0, sp = 0;
        double sum = 0.0; // :82
        const uint8_t *srcp = sources ? sources : spec; // without sources the flags are read twice and the count dropped
        double *slot = seg_slot[wv];
        for (uint64_t base = b; base < e; base += 64 * SEG_ROUNDS) {
            // all of a step's loads first, never predicated (a lane past the end re-reads the segment's last signal and
            // ignores it): a ticker's ~100 posts are one step, one memory round trip
            double v[SEG_ROUNDS];
            uint8_t f[SEG_ROUNDS], s1[SEG_ROUNDS];
#pragma unroll
            for (int r = 0; r < SEG_ROUNDS; ++r) {
                const uint64_t i = base + 64u * r + lane, ic = i < e ? i : e - 1;
                v[r] = pol[ic];
                f[r] = spec[ic];
                s1[r] = srcp[ic]; // (no branch around a load: it would be waited for at the join)
            }
#pragma unroll
            for (int r = 0; r < SEG_ROUNDS; ++r) {
                const uint64_t r0 = base + 64u * r;
                if (r0 >= e || dbg == 2) break; // wave-uniform
                const bool ok = r0 + lane < e;
                const double x = ok ? v[r] : 0.0;
                const bool is_bull = ok && x > tau;               // :86-92: bullish, else bearish, else neutral
                const bool is_bear = ok && !is_bull && x < -tau;
                bull += (uint64_t)__popcll(__ballot(is_bull));
                bear += (uint64_t)__popcll(__ballot(is_bear));
                sp += (uint64_t)__popcll(__ballot(ok && f[r] != 0));
                src1 += (uint64_t)__popcll(__ballot(ok && s1[r] != 0));
                slot[64 * r + lane] = x;
            }
            // the ordered sum of the step's signals: a wave's LDS operations complete in order, so the reads below see
            // every lane's store; the addresses are wave-uniform
            __builtin_amdgcn_wave_barrier();
            const uint32_t m = e - base < 64u * SEG_ROUNDS ? (uint32_t)(e - base) : 64u * SEG_ROUNDS; // scalar
            if (lane == 0 && dbg != 1) { // ONE lane: a read broadcast to 64 lanes costs the LDS as many cycles as 64 different reads
                uint32_t k = 0;
                for (; k + 8 <= m; k += 8) {
                    double t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = slot[k + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum += t[j];
                }
                for (; k < m; ++k) sum += slot[k];
            }
            __builtin_amdgcn_wave_barrier(); // the next step's stores come after these reads
        }
        if (lane == 0) {
            oi_social_counters o;
            o.total = e - b;
            o.by_source[0] = sources ? (e - b) - src1 : 0; // as oi_social_summary: no sources, no histogram
            o.by_source[1] = sources ? src1 : 0;
            o.bullish = bull;
            o.bearish = bear;
            o.neutral = (e - b) - bull - bear;
            o.spec_count = sp;
            o.polarity_sum = sum;
            out[sg] = o;
        }
    }
}

int oi_launch_social_summary_segmented(oi_ctx *ctx, const uint8_t *d_sources, const double *d_pol, const uint8_t *d_spec,
                                       uint64_t n, const uint64_t *d_seg, uint64_t n_seg, double tau,
                                       oi_social_counters *d_out) {
    if (n_seg == 0) return OI_OK;
    const uint64_t per_wg = SEG_THREADS / 64;
    uint64_t blocks = (n_seg + per_wg - 1) / per_wg;
    const uint64_t cap = (uint64_t)ctx->num_cus * 16; // 64 waves per CU in the grid: the segments are strided over them
    if (blocks > cap) blocks = cap;
    static const int dbg = oi_ablation_env("OI_SEG_DBG") ? atoi(oi_ablation_env("OI_SEG_DBG")) : 0; // ablations (wrong results): 1 no ordered sum, 2 no counters / no stores to LDS
    ProfScope ps(ctx, "social_summary_segmented");
    hipLaunchKernelGGL(social_summary_segmented_kernel, dim3((uint32_t)blocks), dim3(SEG_THREADS), 0, ctx->stream, d_sources,
                       d_pol, d_spec, n, d_seg, n_seg, tau, d_out, dbg);
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}
