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
    ctx->prof_begin("headline");
    hipLaunchKernelGGL(headline_scan_kernel, dim3(grid), dim3(HL_THREADS), 0, ctx->stream, d_blob, d_offsets, n,
                       blob_bytes, dp.as<HlParams>(), d_mask, d_order, d_about);
    ctx->prof_end("headline");
    OI_HIP_CHECK(hipGetLastError());
    return OI_OK;
}

extern "C" const char *oi_catalyst_keyword(uint32_t index) { return index < HL_N_KW ? kCatalyst[index] : nullptr; }
