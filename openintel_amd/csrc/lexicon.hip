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

__device__ __forceinline__ void lex2_hit(Lex2Shared &s, uint32_t mult, uint32_t k0, uint32_t k1, uint32_t c8,
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
__device__ void lex2_slow_chunk(const uint8_t *tb, uint32_t lo, uint32_t hi, uint32_t j, Lex2Shared &s,
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
__global__ __launch_bounds__(LEX_THREADS, 4) void lexicon_kernel(const uint8_t *blob, const uint64_t *offsets,
                                                              uint64_t n, uint64_t blob_bytes,
                                                              const LexEntry *table, const uint32_t *bloom,
                                                              uint32_t mult, double *pol_out, uint8_t *spec_out,
                                                              const uint8_t *sources, double tau, SumPartial *partials) {
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

    const uint64_t n_tiles = (n + LX_PPT - 1) / LX_PPT;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t p0 = tile * LX_PPT;
        const uint32_t np = (uint32_t)((n - p0) < LX_PPT ? (n - p0) : LX_PPT);
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
            }
            // ---- dense pass over this wave's queue (LDS ops of one wave complete in order)
            {
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
    static const bool v1 = oi_ablation_env("OI_LEXICON_V1") != nullptr; // A/B switch: the first-generation scan
    const uint64_t n_tiles = (n + LX_PPT - 1) / LX_PPT;
    const uint32_t max_grid = (uint32_t)ctx->num_cus * 8u;
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
            hipLaunchKernelGGL(lexicon_kernel, dim3(grid), dim3(LEX_THREADS), 0, ctx->stream, d_blob, d_offsets, n,
                               blob_bytes, d_table, d_bloom, h_mult, d_pol, d_spec, d_sources, tau, d_partials);
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
