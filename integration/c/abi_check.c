/* abi_check.c -- a plain-C host of libopenintel_hip.so (C99, no C++ runtime of its own): what the reference's FFI sees.
 *
 *   gcc -std=c99 -Wall -Wextra -pedantic -I include integration/c/abi_check.c -L openintel_amd -lopenintel_hip \
 *       -Wl,-rpath,$PWD/openintel_amd -lm -o /tmp/abi_check && /tmp/abi_check
 *
 * Without a GPU it checks that the header is valid C, that the library links and that a missing device is a loud
 * OI_ERR_NO_DEVICE (there is no CPU path).  With a gfx950 device it scores three posts through the PostAnalyzer entry
 * point (post_analyzer.rs:7-11 / lexicon.rs:106-120's sentences), sums a pooled batch per ticker, and runs one hybrid query of a 1000-post index
 * (BASELINE configs[0]'s shape: 384-d, top-10). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "openintel_hip.h"

static int fail(const char *what, int rc) {
    fprintf(stderr, "abi_check: %s failed: %d (%s)\n", what, rc, oi_last_error());
    return 1;
}

int main(void) {
    oi_ctx *ctx = NULL;
    int rc;
    if (oi_abi_version() != OI_ABI_VERSION) return fail("oi_abi_version", oi_abi_version());
    rc = oi_create(0, &ctx);
    if (rc == OI_ERR_NO_DEVICE) {
        if (ctx != NULL || oi_last_error()[0] == '\0') return fail("no-device contract", rc);
        printf("abi_check: no gfx950 device (%s) -- link and header ok\n", oi_last_error());
        return 0;
    }
    if (rc != OI_OK) return fail("oi_create", rc);

    { /* PostAnalyzer::analyze */
        const char *posts[3] = {"AAPL to the moon, loading calls all day", "AAPL is going to dump, buying puts", "AAPL earnings are next week"};
        unsigned char blob[256];
        uint64_t offs[4] = {0, 0, 0, 0};
        double pol[3];
        uint8_t spec[3];
        int i;
        for (i = 0; i < 3; ++i) {
            memcpy(blob + offs[i], posts[i], strlen(posts[i]));
            offs[i + 1] = offs[i] + strlen(posts[i]);
        }
        rc = oi_lexicon_analyze(ctx, blob, offs, 3, pol, spec);
        if (rc != OI_OK) return fail("oi_lexicon_analyze", rc);
        if (!(pol[0] > 0.0 && spec[0] == 1 && pol[1] < 0.0 && spec[1] == 1 && pol[2] == 0.0 && spec[2] == 0)) {
            fprintf(stderr, "abi_check: lexicon signals differ from lexicon.rs:109-119: %g/%d %g/%d %g/%d\n", pol[0], spec[0], pol[1], spec[1], pol[2], spec[2]);
            return 1;
        }
    }
    { /* the batch tools' per-ticker sums (mcp/tools.rs:193-225): two tickers' signals pooled, the second one empty */
        const double pol[5] = {1.0, -1.0, 0.0, 0.5, 1e-17};
        const uint8_t spec[5] = {1, 0, 0, 1, 0}, src[5] = {0, 1, 1, 0, 0};
        const uint64_t seg[4] = {0, 5, 5, 5};
        oi_social_counters out[3];
        double want = 0.0;
        int i;
        for (i = 0; i < 5; ++i) want += pol[i]; /* speculation_engine.rs:82-86, input order */
        rc = oi_social_summary_segmented(ctx, src, pol, spec, 5, seg, 3, 0.2, OI_HOST, out);
        if (rc != OI_OK) return fail("oi_social_summary_segmented", rc);
        if (!(out[0].total == 5 && out[0].bullish == 2 && out[0].bearish == 1 && out[0].neutral == 2 && out[0].spec_count == 2 &&
              out[0].by_source[0] == 3 && out[0].by_source[1] == 2 && out[0].polarity_sum == want && out[1].total == 0 &&
              out[2].total == 0 && out[2].polarity_sum == 0.0)) {
            fprintf(stderr, "abi_check: per-ticker sums differ: total %llu bull %llu bear %llu sum %.17g (want %.17g)\n",
                    (unsigned long long)out[0].total, (unsigned long long)out[0].bullish, (unsigned long long)out[0].bearish,
                    out[0].polarity_sum, want);
            return 1;
        }
    }
    { /* the hybrid query: doc 123 holds the query's vector and its terms, so it must come first */
        enum { N = 1000, D = 384, V = 64, K = 10 };
        float *rows = (float *)calloc((size_t)N * D, sizeof(float)), *q = (float *)calloc(D, sizeof(float));
        uint32_t *terms = (uint32_t *)malloc(sizeof(uint32_t) * N * 3);
        uint64_t *offs = (uint64_t *)malloc(sizeof(uint64_t) * (N + 1));
        uint32_t qt[2] = {7, 9}, qo[2] = {0, 2}, docs[K], counts[1];
        float scores[K];
        oi_index *idx = NULL;
        uint64_t tokens = 0;
        uint32_t i, k, seed = 12345u;
        for (i = 0; i < N; ++i) {
            for (k = 0; k < D; ++k) { seed = seed * 1664525u + 1013904223u; rows[(size_t)i * D + k] = (float)((seed >> 9) & 0xFFFF) / 65536.0f - 0.5f; }
            offs[i] = 3ull * i;
            terms[3 * i] = i % V; terms[3 * i + 1] = (i * 7u) % V; terms[3 * i + 2] = (i * 13u) % V;
        }
        offs[N] = 3ull * N;
        terms[3 * 123] = 7; terms[3 * 123 + 1] = 9; terms[3 * 123 + 2] = 7;
        memcpy(q, rows + (size_t)123 * D, sizeof(float) * D);
        { double ss = 0.0; for (k = 0; k < D; ++k) ss += (double)q[k] * q[k]; for (k = 0; k < D; ++k) q[k] = (float)(q[k] / (ss > 0 ? sqrt(ss) : 1.0)); }
        if ((rc = oi_index_create(ctx, N, D, V, 0, &idx)) != OI_OK) return fail("oi_index_create", rc);
        if ((rc = oi_index_set_embeddings(idx, rows, OI_HOST, 1)) != OI_OK) return fail("oi_index_set_embeddings", rc);
        if ((rc = oi_index_set_forward(idx, terms, offs, OI_HOST)) != OI_OK) return fail("oi_index_set_forward", rc);
        if ((rc = oi_index_local_stats(idx, &tokens, NULL)) != OI_OK) return fail("oi_index_local_stats", rc);
        if ((rc = oi_index_finalize(idx, N, tokens, NULL)) != OI_OK) return fail("oi_index_finalize", rc);
        if ((rc = oi_search(idx, q, qt, qo, 1, 100, K, OI_HOST, scores, docs, counts)) != OI_OK) return fail("oi_search", rc);
        if (counts[0] != K || docs[0] != 123) {
            fprintf(stderr, "abi_check: hybrid query returned %u docs, first %u (expected 10, 123)\n", counts[0], docs[0]);
            return 1;
        }
        { /* the pipelined query (oi_pipeline_*): the same batch three times through two lanes the library owns; every
           * result must be oi_search's, bit for bit */
            oi_pipeline *pipe = NULL;
            float s2[3][10];
            uint32_t d2[3][10], c2[3];
            uint64_t ticket[3];
            int b;
            if ((rc = oi_pipeline_create(idx, NULL, 2, 1, 8, 100, K, &pipe)) != OI_OK) return fail("oi_pipeline_create", rc);
            for (b = 0; b < 3; ++b)
                if ((rc = oi_pipeline_submit(pipe, q, qt, qo, 1, OI_HOST, s2[b], d2[b], &c2[b], &ticket[b])) != OI_OK) return fail("oi_pipeline_submit", rc);
            if ((rc = oi_pipeline_wait(pipe, ticket[1], 1)) != OI_OK) return fail("oi_pipeline_wait", rc);
            if ((rc = oi_pipeline_drain(pipe)) != OI_OK) return fail("oi_pipeline_drain", rc);
            for (b = 0; b < 3; ++b)
                if (c2[b] != counts[0] || memcmp(d2[b], docs, sizeof(uint32_t) * K) != 0 || memcmp(s2[b], scores, sizeof(float) * K) != 0) {
                    fprintf(stderr, "abi_check: pipelined batch %d differs from oi_search\n", b);
                    return 1;
                }
            oi_pipeline_destroy(pipe);
        }
        oi_destroy(ctx); /* any destruction order is fine: the ctx first ... */
        ctx = NULL;
        oi_index_destroy(idx); /* ... the index after it */
        free(rows); free(q); free(terms); free(offs);
    }
    printf("abi_check: ok\n");
    return 0;
}
