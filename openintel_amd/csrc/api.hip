// api.hip -- the extern "C" surface of libopenintel_hip.so (see include/openintel_hip.h).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <set>

#include "oi_internal.h"

// ---------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

void oi_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Hardware queues.  HIP spreads the streams of a process over GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams
// that share a queue run one after the other whatever their events allow.  A two-lane pipeline (oi_pipeline_*, or a host that
// drives views of an index itself) has seven streams in play -- the caller's, two lanes, the fusing stream, a BM25 side stream
// each -- so with 4 queues some share, and the lanes' overlap is lost: 0.55 ms per batch at a 1.25M-row shard against 0.465
// with 8 queues, ten runs each, tools/r05_pipeline_probe.py.  The runtime reads the variable when it initialises, so it is set
// when this library is LOADED (never overriding the host's own choice); a host that initialises HIP before loading the library
// sets it itself (bench.py and tests/conftest.py do).  It changes scheduling only, never a result.
__attribute__((constructor)) static void oi_ask_for_more_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0); }

extern "C" const char *oi_last_error(void) { return g_err; }
extern "C" int oi_abi_version(void) { return OI_ABI_VERSION; }

// ---------------------------------------------------------------- per-(kernel, device) one-shots
int oi_dyn_lds(oi_ctx *ctx, const void *kernel, size_t bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_pair(kernel, ctx->device);
    if (done.count(key)) return OI_OK;
    OI_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.insert(key);
    return OI_OK;
}

// ---------------------------------------------------------------- buffers
std::atomic<uint64_t> g_oi_ws_epoch{1};

int DevBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return OI_OK;
    if (borrowed) { oi_set_error("internal: a borrowed buffer cannot grow"); return OI_ERR_STATE; }
    g_oi_ws_epoch.fetch_add(1); // captured launch sequences that hold the old pointer are stale from here on
    if (p) {
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = (bytes + 255) & ~(size_t)255;
    if (want == 0) want = 256;
    OI_HIP_CHECK(hipMalloc(&p, want));
    cap = want;
    return OI_OK;
}
void DevBuf::release() {
    if (p) g_oi_ws_epoch.fetch_add(1);
    if (p && !borrowed) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    borrowed = false;
}

int PinBuf::ensure(size_t bytes) {
    if (bytes <= cap) return OI_OK;
    release();
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) == hipSuccess) {
        pinned = true;
    } else { // no page-locked memory to be had (a locked-memory limit): pageable staging -- the same calls, staged by the runtime
        (void)hipGetLastError();
        p = aligned_alloc(4096, want);
        pinned = false;
        if (!p) { oi_set_error("host staging buffer: out of memory (%zu bytes)", want); return OI_ERR_HIP; }
    }
    cap = want;
    return OI_OK;
}
void PinBuf::release() {
    if (p) {
        if (pinned) (void)hipHostFree(p);
        else free(p);
    }
    p = nullptr;
    cap = 0;
}

// ---------------------------------------------------------------- profiling hooks
void oi_ctx::prof_begin(const char *tag) {
    ProfSpan s;
    // (timing only: without the system-scope fence a default event performs when it completes -- each pair sits between two
    // kernels of the timed step)
    if (hipEventCreateWithFlags(&s.a, hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&s.b, hipEventDisableSystemFence) != hipSuccess) return;
    (void)hipEventRecord(s.a, stream);
    prof[tag].push_back(s);
}
void oi_ctx::prof_end(const char *tag) {
    auto it = prof.find(tag);
    if (it == prof.end() || it->second.empty()) return;
    (void)hipEventRecord(it->second.back().b, stream);
}

extern "C" int oi_profile_reset(oi_ctx *ctx, int enable) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> g(ctx->mu);
    for (auto &kv : ctx->prof)
        for (auto &s : kv.second) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    ctx->prof.clear();
    ctx->prof_enabled = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return OI_OK;
}

extern "C" int oi_profile_read(oi_ctx *ctx, const char *kernel_tag, double *total_ms_out, uint64_t *launches_out) {
    if (!ctx || !kernel_tag) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double total = 0.0;
    uint64_t n = 0;
    if (strncmp(kernel_tag, "flagword", 8) == 0) { // diagnostics: raw words of the ctx state flag buffer
        uint32_t w[4] = {0, 0, 0, 0};
        DevBuf &st = ctx->buf("state_flag");
        if (st.p) OI_HIP_CHECK(hipMemcpy(w, st.p, 16, hipMemcpyDeviceToHost));
        if (total_ms_out) *total_ms_out = (double)w[(kernel_tag[8] - '0') & 3];
        if (launches_out) *launches_out = 0;
        return OI_OK;
    }
    if (strcmp(kernel_tag, "graph_replays") == 0 || strcmp(kernel_tag, "graph_captures") == 0) { // diagnostics of oi_set_graph_replay
        if (total_ms_out) *total_ms_out = 0.0;
        if (launches_out) *launches_out = kernel_tag[6] == 'r' ? ctx->graph_replays : ctx->graph_captures;
        return OI_OK;
    }
    if (strcmp(kernel_tag, "spec_state") == 0) { // diagnostics of the speculative screen thresholds: failed checks seen so far, searches that speculated
        if (ctx->spec_fail_host && *ctx->spec_fail_host) { *ctx->spec_fail_host = 0; ++ctx->spec_failures; ctx->spec_backoff = ctx->spec_backoff ? std::min(1024u, 2 * ctx->spec_backoff) : 16u; ctx->spec_skip = ctx->spec_backoff; }
        if (total_ms_out) *total_ms_out = (double)ctx->spec_failures;
        if (launches_out) *launches_out = ctx->spec_searches;
        return OI_OK;
    }
    if (strcmp(kernel_tag, "screen_gate") == 0) { // diagnostics: did the last screened search fall back to the exact kernel?
        uint32_t g = 0;                          // (0 no, nonzero yes; -1 when no search has used the screen)
        if (ctx->last_screen_gate) OI_HIP_CHECK(hipMemcpy(&g, ctx->last_screen_gate, 4, hipMemcpyDeviceToHost));
        if (total_ms_out) *total_ms_out = ctx->last_screen_gate ? (double)g : -1.0;
        if (launches_out) *launches_out = 0;
        return OI_OK;
    }
    auto it = ctx->prof.find(kernel_tag);
    if (it != ctx->prof.end())
        for (auto &s : it->second) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { total += ms; ++n; }
        }
    if (total_ms_out) *total_ms_out = total;
    if (launches_out) *launches_out = n;
    return OI_OK;
}

// ---------------------------------------------------------------- captured launch sequences (hipGraph replay)
#define OI_MAX_GRAPHS 32

// (callers make sure no launch of the graph is still in flight: every path that frees one synchronises the ctx stream first)
static void graph_free(GraphEntry &e) {
    if (e.exec) (void)hipGraphExecDestroy(e.exec);
    if (e.graph) (void)hipGraphDestroy(e.graph);
    e.exec = nullptr;
    e.graph = nullptr;
}

static void graphs_purge(oi_ctx *ctx, uint64_t index_uid) { // index_uid 0 = all
    for (size_t i = 0; i < ctx->graphs.size();)
        if (index_uid == 0 || ctx->graphs[i].key[0] == index_uid) {
            graph_free(ctx->graphs[i]);
            ctx->graphs.erase(ctx->graphs.begin() + (long)i);
        } else ++i;
}

// body(): the launches of one call, all on ctx->stream (and streams forked from and joined back into it), nothing that
// synchronises.  First call with a key: eager (it also allocates the workspaces and sets the kernels' attributes);
// second: captured, instantiated, launched; afterwards: one hipGraphLaunch.  Anything unusual -- profiling on, the default
// stream, a workspace that moved, a capture that fails -- takes the eager path, which is always correct.
template <class F> static int run_captured(oi_ctx *ctx, const uint64_t (&key)[10], F &&body) {
    if (!ctx->use_graphs || ctx->prof_enabled || !ctx->stream) return body();
    const uint64_t epoch = g_oi_ws_epoch.load();
    GraphEntry *e = nullptr;
    for (auto &g : ctx->graphs)
        if (memcmp(g.key, key, sizeof(key)) == 0) { e = &g; break; }
    if (!e) {
        if (ctx->graphs.size() >= OI_MAX_GRAPHS) {
            size_t lru = 0;
            for (size_t i = 1; i < ctx->graphs.size(); ++i)
                if (ctx->graphs[i].last_use < ctx->graphs[lru].last_use) lru = i;
            OI_HIP_CHECK(hipStreamSynchronize(ctx->stream)); // rare (more than OI_MAX_GRAPHS distinct calls): its last replay may still run
            graph_free(ctx->graphs[lru]);
            ctx->graphs.erase(ctx->graphs.begin() + (long)lru);
        }
        ctx->graphs.emplace_back();
        e = &ctx->graphs.back();
        memcpy(e->key, key, sizeof(key));
        e->last_use = ++ctx->graph_clock;
        const int rc = body();
        e->epoch = g_oi_ws_epoch.load(); // (after the call: whatever it allocated is in place now)
        return rc;
    }
    e->last_use = ++ctx->graph_clock;
    if (e->epoch != epoch) { // a workspace moved since: start over (eager now, capture next time)
        OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        graph_free(*e);
        e->state = 0;
        const int rc = body();
        e->epoch = g_oi_ws_epoch.load();
        return rc;
    }
    if (e->state == 2) return body();
    if (e->state == 1) {
        OI_HIP_CHECK(hipGraphLaunch(e->exec, ctx->stream));
        ++ctx->graph_replays;
        return OI_OK;
    }
    // state 0, same epoch: capture
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        e->state = 2;
        return body();
    }
    const int rc = body();
    hipGraph_t g = nullptr;
    const hipError_t end = hipStreamEndCapture(ctx->stream, &g);
    if (rc != OI_OK || end != hipSuccess || !g || g_oi_ws_epoch.load() != epoch) {
        (void)hipGetLastError();
        if (g) (void)hipGraphDestroy(g);
        e->state = 2;
        if (rc != OI_OK) return rc; // (nothing of the failed call was launched)
        return body();               // the captured launches never ran: run them
    }
    hipGraphExec_t x = nullptr;
    if (hipGraphInstantiate(&x, g, nullptr, nullptr, 0) != hipSuccess || !x) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(g);
        e->state = 2;
        return body();
    }
    e->graph = g;
    e->exec = x;
    e->state = 1;
    ++ctx->graph_captures;
    OI_HIP_CHECK(hipGraphLaunch(e->exec, ctx->stream));
    return OI_OK;
}

extern "C" int oi_set_graph_replay(oi_ctx *ctx, int enable) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->use_graphs = enable != 0;
    if (!ctx->use_graphs) {
        OI_HIP_CHECK(hipSetDevice(ctx->device));
        OI_HIP_CHECK(hipStreamSynchronize(ctx->stream)); // no replay in flight when its graph goes
        graphs_purge(ctx, 0);
    }
    return OI_OK;
}

// ---------------------------------------------------------------- ctx
extern "C" int oi_create(int device_ordinal, oi_ctx **out) {
    if (!out) { oi_set_error("oi_create: out is null"); return OI_ERR_INVALID_ARG; }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        oi_set_error("oi_create: no HIP device visible (this library has no CPU fallback)");
        return OI_ERR_NO_DEVICE;
    }
    OI_REQUIRE(device_ordinal >= 0 && device_ordinal < n, "oi_create: device %d of %d", device_ordinal, n);
    OI_HIP_CHECK(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    OI_HIP_CHECK(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        oi_set_error("oi_create: device %d is %s; this library is built for gfx950 only", device_ordinal,
                     prop.gcnArchName);
        return OI_ERR_NO_DEVICE;
    }
    oi_ctx *c = new oi_ctx();
    c->device = device_ordinal;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->stream = nullptr;
    if (const char *m = getenv("OI_COSINE_MODE"))
        c->cosine_mode = strcmp(m, "split") == 0 ? OI_COSINE_SPLIT : strcmp(m, "exact") == 0 ? OI_COSINE_EXACT
                         : strcmp(m, "screen-copy") == 0 ? OI_COSINE_SCREEN_COPY
                         : strcmp(m, "screen-stream") == 0 ? OI_COSINE_SCREEN_STREAM : OI_COSINE_SCREEN;
    // best effort: without these the two legs of a query simply run one after the other.  The side stream itself is made by
    // the first hybrid search of the context (ensure_side_stream): a context that never runs one -- the fusing context of a
    // pipeline, a lexicon-only host -- does not take one of the process's few hardware queues (HIP hands streams of one priority
    // out over GPU_MAX_HW_QUEUES = 4 of them; two streams on one queue run one after the other whatever the events say).
    // (fork / join ORDER the side stream's kernels against the main stream's through hipStreamWaitEvent: they keep the default
    // fence -- hipEventDisableSystemFence is documented for timing events only; the ~12 us it saved per step in round 4 rested
    // on the kernels' own release semantics, an implementation detail of the runtime)
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) c->ev_fork = nullptr;
    if (hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) c->ev_join = nullptr;
    *out = c;
    return OI_OK;
}

extern "C" int oi_create_like(oi_ctx *like, oi_ctx **out) {
    if (!out) { oi_set_error("oi_create_like: out is null"); return OI_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!like) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    OI_CHECK(oi_create(like->device, out));
    std::lock_guard<std::mutex> g(like->mu);
    (*out)->cosine_mode = like->cosine_mode;
    (*out)->overlap_legs = like->overlap_legs;
    (*out)->speculate = like->speculate;
    (*out)->use_graphs = like->use_graphs;
    return OI_OK;
}

// The teardown proper: runs when the last reference goes (the caller's handle, or the last index that outlived it).
static void ctx_release(oi_ctx *ctx) {
    if (ctx->refs.fetch_sub(1) != 1) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    oi_profile_reset(ctx, 0);
    graphs_purge(ctx, 0);
    for (auto &kv : ctx->ws) kv.second.release();
    ctx->pin_in.release();
    ctx->pin_out.release();
    if (ctx->spec_fail_host) (void)hipHostFree(ctx->spec_fail_host);
    delete ctx;
}

extern "C" void oi_destroy(oi_ctx *ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        // With indexes still alive the object stays (they search, and are destroyed, through it): from here on it runs
        // on the default stream -- the caller's stream may not outlive the caller's handle.
        if (ctx->refs.load() > 1) ctx->stream = nullptr;
    }
    ctx_release(ctx);
}

extern "C" int oi_workspace_bytes(oi_ctx *ctx, uint64_t *device_bytes, uint64_t *pinned_bytes) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> g(ctx->mu);
    uint64_t d = 0;
    for (auto &kv : ctx->ws)
        if (!kv.second.borrowed) d += kv.second.cap;
    if (device_bytes) *device_bytes = d;
    if (pinned_bytes) *pinned_bytes = (ctx->pin_in.pinned ? ctx->pin_in.cap : 0) + (ctx->pin_out.pinned ? ctx->pin_out.cap : 0);
    return OI_OK;
}

extern "C" int oi_set_stream(oi_ctx *ctx, void *hip_stream) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return OI_OK;
}

extern "C" int oi_set_cosine_mode(oi_ctx *ctx, int mode) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    OI_REQUIRE(mode == OI_COSINE_EXACT || mode == OI_COSINE_SPLIT || mode == OI_COSINE_SCREEN || mode == OI_COSINE_SCREEN_COPY ||
                   mode == OI_COSINE_SCREEN_STREAM,
               "oi_set_cosine_mode: unknown mode %d", mode);
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->cosine_mode = mode;
    return OI_OK;
}

extern "C" int oi_set_overlap(oi_ctx *ctx, int enable) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->overlap_legs = enable != 0;
    return OI_OK;
}

extern "C" int oi_set_screen_speculation(oi_ctx *ctx, int enable) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->speculate = enable != 0;
    ctx->spec_skip = ctx->spec_backoff = 0;
    return OI_OK;
}

static int check_overflow_locked(oi_ctx *ctx) {
    DevBuf &st = ctx->buf("state_flag");
    if (!st.p) return OI_OK;
    uint32_t f = 0;
    OI_HIP_CHECK(hipMemcpyAsync(&f, st.p, sizeof(f), hipMemcpyDeviceToHost, ctx->stream));
    OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (f) {
        OI_HIP_CHECK(hipMemsetAsync(st.p, 0, 16, ctx->stream));
        oi_set_error("internal candidate pool overflowed");
        return OI_ERR_OVERFLOW;
    }
    return OI_OK;
}

extern "C" int oi_synchronize(oi_ctx *ctx) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return check_overflow_locked(ctx);
}

// ---------------------------------------------------------------- PostAnalyzer path
extern "C" int oi_lexicon_analyze_device(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets,
                                         uint64_t n, uint64_t blob_bytes, double *d_pol, uint8_t *d_spec) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    if (n == 0) return OI_OK;
    OI_REQUIRE(d_offsets && d_pol && d_spec && (d_blob || blob_bytes == 0), "lexicon: null buffer");
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    return oi_launch_lexicon(ctx, d_blob, d_offsets, n, blob_bytes, d_pol, d_spec);
}

extern "C" int oi_lexicon_summary_device(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n,
                                         uint64_t blob_bytes, const uint8_t *d_sources, double tau, double *d_pol,
                                         uint8_t *d_spec, oi_social_counters *out) {
    if (!ctx || !out) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    if (n == 0) { memset(out, 0, sizeof(*out)); return OI_OK; }
    OI_REQUIRE(d_offsets && (d_blob || blob_bytes == 0), "lexicon summary: null buffer");
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    return oi_launch_lexicon_fused(ctx, d_blob, d_offsets, n, blob_bytes, d_pol, d_spec, d_sources, tau, out);
}

extern "C" int oi_lexicon_analyze(oi_ctx *ctx, const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                                  double *pol_out, uint8_t *spec_out) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    if (n == 0) return OI_OK;
    OI_REQUIRE(offsets && pol_out && spec_out, "lexicon: null buffer");
    OI_REQUIRE(offsets[0] == 0, "lexicon: offsets[0] must be 0");
    const uint64_t bytes = offsets[n];
    OI_REQUIRE(blob || bytes == 0, "lexicon: null text blob");
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // A ticker's posts (the reference's call: <= 100 posts, ~20 KB) are a launch-latency call: text and offsets go through
    // the context's page-locked buffer in ONE DMA, the signals come back in one (100 posts: 62 -> 41 us per call, 10 posts: 49 -> 29; tools/analyze_latency_probe.py).
    const size_t off_o = (bytes + 15) & ~(size_t)15, in_bytes = off_o + sizeof(uint64_t) * (n + 1);
    const size_t out_bytes = sizeof(double) * n + n;
    if (in_bytes <= OI_PINNED_STAGE_MAX) {
        DevBuf &si = ctx->buf("lex_stage_in"), &so = ctx->buf("lex_stage_out");
        OI_CHECK(si.ensure(in_bytes + 64));
        OI_CHECK(so.ensure(out_bytes));
        OI_CHECK(ctx->pin_in.ensure(in_bytes));
        OI_CHECK(ctx->pin_out.ensure(out_bytes));
        uint8_t *h = ctx->pin_in.as<uint8_t>();
        if (bytes) memcpy(h, blob, bytes);
        memcpy(h + off_o, offsets, sizeof(uint64_t) * (n + 1));
        OI_HIP_CHECK(hipMemcpyAsync(si.p, h, in_bytes, hipMemcpyHostToDevice, st));
        double *d_pol = so.as<double>();
        uint8_t *d_spec = so.as<uint8_t>() + sizeof(double) * n;
        OI_CHECK(oi_launch_lexicon(ctx, si.as<uint8_t>(), reinterpret_cast<const uint64_t *>(si.as<uint8_t>() + off_o), n, bytes,
                                   d_pol, d_spec));
        OI_HIP_CHECK(hipMemcpyAsync(ctx->pin_out.p, so.p, out_bytes, hipMemcpyDeviceToHost, st));
        OI_HIP_CHECK(hipStreamSynchronize(st));
        memcpy(pol_out, ctx->pin_out.p, sizeof(double) * n);
        memcpy(spec_out, ctx->pin_out.as<uint8_t>() + sizeof(double) * n, n);
        return OI_OK;
    }
    DevBuf &b = ctx->buf("lex_blob"), &o = ctx->buf("lex_off"), &p = ctx->buf("lex_pol"), &s = ctx->buf("lex_spec");
    OI_CHECK(b.ensure(bytes + 64));
    OI_CHECK(o.ensure(sizeof(uint64_t) * (n + 1)));
    OI_CHECK(p.ensure(sizeof(double) * n));
    OI_CHECK(s.ensure(n));
    if (bytes) OI_HIP_CHECK(hipMemcpyAsync(b.p, blob, bytes, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemcpyAsync(o.p, offsets, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, st));
    OI_CHECK(oi_launch_lexicon(ctx, b.as<uint8_t>(), o.as<uint64_t>(), n, bytes, p.as<double>(), s.as<uint8_t>()));
    OI_HIP_CHECK(hipMemcpyAsync(pol_out, p.p, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(spec_out, s.p, n, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    return OI_OK;
}

extern "C" int oi_headline_scan_device(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n,
                                       uint64_t blob_bytes, const uint8_t *ticker, uint64_t ticker_len,
                                       const uint8_t *forms_blob, const uint32_t *form_offsets, uint32_t n_forms,
                                       uint16_t *d_mask, uint64_t *d_order, uint8_t *d_about) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    if (n == 0) return OI_OK;
    OI_REQUIRE(d_offsets && d_mask && d_order && d_about && (d_blob || blob_bytes == 0), "headline scan: null buffer");
    OI_REQUIRE((ticker || ticker_len == 0) && (n_forms == 0 || (forms_blob && form_offsets)),
               "headline scan: null ticker/forms");
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    return oi_launch_headline_scan(ctx, d_blob, d_offsets, n, blob_bytes, ticker, ticker_len, forms_blob,
                                   form_offsets, n_forms, d_mask, d_order, d_about);
}

extern "C" int oi_headline_scan(oi_ctx *ctx, const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                                const uint8_t *ticker, uint64_t ticker_len, const uint8_t *forms_blob,
                                const uint32_t *form_offsets, uint32_t n_forms, uint16_t *mask_out,
                                uint64_t *order_out, uint8_t *about_out) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    if (n == 0) return OI_OK;
    OI_REQUIRE(offsets && mask_out && order_out && about_out, "headline scan: null buffer");
    OI_REQUIRE(offsets[0] == 0, "headline scan: offsets[0] must be 0");
    OI_REQUIRE((ticker || ticker_len == 0) && (n_forms == 0 || (forms_blob && form_offsets)),
               "headline scan: null ticker/forms");
    const uint64_t bytes = offsets[n];
    OI_REQUIRE(blob || bytes == 0, "headline scan: null title blob");
    for (uint64_t i = 0; i < n; ++i)
        OI_REQUIRE(offsets[i + 1] >= offsets[i] && offsets[i + 1] - offsets[i] < (1ull << 32),
                   "headline scan: offsets must ascend and a title must be shorter than 4 GiB");
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // the dip gate's own call is a handful of headlines per ticker (dip.rs:617-626): like oi_lexicon_analyze, a small call
    // goes through the page-locked buffers -- titles + offsets in one DMA, order | mask | about back in one
    const size_t off_o = (bytes + 15) & ~(size_t)15, in_bytes = off_o + sizeof(uint64_t) * (n + 1);
    const size_t out_m = sizeof(uint64_t) * n, out_a = out_m + sizeof(uint16_t) * n, out_bytes = out_a + n;
    if (in_bytes <= OI_PINNED_STAGE_MAX) {
        DevBuf &si = ctx->buf("hl_stage_in"), &so = ctx->buf("hl_stage_out");
        OI_CHECK(si.ensure(in_bytes + 64));
        OI_CHECK(so.ensure(out_bytes));
        OI_CHECK(ctx->pin_in.ensure(in_bytes));
        OI_CHECK(ctx->pin_out.ensure(out_bytes));
        uint8_t *h = ctx->pin_in.as<uint8_t>();
        if (bytes) memcpy(h, blob, bytes);
        memcpy(h + off_o, offsets, sizeof(uint64_t) * (n + 1));
        OI_HIP_CHECK(hipMemcpyAsync(si.p, h, in_bytes, hipMemcpyHostToDevice, st));
        uint8_t *d = so.as<uint8_t>();
        OI_CHECK(oi_launch_headline_scan(ctx, si.as<uint8_t>(), reinterpret_cast<const uint64_t *>(si.as<uint8_t>() + off_o), n,
                                         bytes, ticker, ticker_len, forms_blob, form_offsets, n_forms,
                                         reinterpret_cast<uint16_t *>(d + out_m), reinterpret_cast<uint64_t *>(d), d + out_a));
        OI_HIP_CHECK(hipMemcpyAsync(ctx->pin_out.p, so.p, out_bytes, hipMemcpyDeviceToHost, st));
        OI_HIP_CHECK(hipStreamSynchronize(st));
        const uint8_t *r = ctx->pin_out.as<uint8_t>();
        memcpy(order_out, r, out_m);
        memcpy(mask_out, r + out_m, sizeof(uint16_t) * n);
        memcpy(about_out, r + out_a, n);
        return OI_OK;
    }
    DevBuf &b = ctx->buf("hl_blob"), &o = ctx->buf("hl_off"), &m = ctx->buf("hl_mask"), &r = ctx->buf("hl_order"),
           &a = ctx->buf("hl_about");
    OI_CHECK(b.ensure(bytes + 64));
    OI_CHECK(o.ensure(sizeof(uint64_t) * (n + 1)));
    OI_CHECK(m.ensure(sizeof(uint16_t) * n));
    OI_CHECK(r.ensure(sizeof(uint64_t) * n));
    OI_CHECK(a.ensure(n));
    if (bytes) OI_HIP_CHECK(hipMemcpyAsync(b.p, blob, bytes, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemcpyAsync(o.p, offsets, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, st));
    OI_CHECK(oi_launch_headline_scan(ctx, b.as<uint8_t>(), o.as<uint64_t>(), n, bytes, ticker, ticker_len, forms_blob,
                                     form_offsets, n_forms, m.as<uint16_t>(), r.as<uint64_t>(), a.as<uint8_t>()));
    OI_HIP_CHECK(hipMemcpyAsync(mask_out, m.p, sizeof(uint16_t) * n, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(order_out, r.p, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(about_out, a.p, n, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    return OI_OK;
}

// The dip screen's headline gate over the ROWS of a scan (application/dip.rs: one `check` per loser, each calling the gate
// of domain/dip.rs:612-659 on its own headlines with its own ticker and name forms): one staging copy of all titles, one
// launch per row back to back on the stream, one copy back, one synchronise.
extern "C" int oi_headline_scan_rows(oi_ctx *ctx, const uint8_t *blob, const uint64_t *offsets, uint64_t n,
                                     const uint64_t *row_offsets, uint32_t n_rows, const uint8_t *tickers_blob,
                                     const uint32_t *ticker_offsets, const uint8_t *forms_blob, const uint32_t *form_offsets,
                                     const uint32_t *row_form_offsets, uint16_t *mask_out, uint64_t *order_out,
                                     uint8_t *about_out) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    if (n_rows == 0) return OI_OK;
    OI_REQUIRE(n_rows <= 4096, "headline rows: at most 4096 rows per call (%u given)", n_rows);
    OI_REQUIRE(row_offsets && ticker_offsets && row_form_offsets, "headline rows: null row table");
    OI_REQUIRE(row_offsets[0] == 0 && row_offsets[n_rows] == n, "headline rows: the rows must cover titles [0, %llu)",
               (unsigned long long)n);
    for (uint32_t r = 0; r < n_rows; ++r)
        OI_REQUIRE(row_offsets[r] <= row_offsets[r + 1] && ticker_offsets[r] <= ticker_offsets[r + 1] &&
                       row_form_offsets[r] <= row_form_offsets[r + 1],
                   "headline rows: row %u has descending offsets", r);
    OI_REQUIRE(tickers_blob || ticker_offsets[n_rows] == 0, "headline rows: null tickers");
    OI_REQUIRE(row_form_offsets[n_rows] == 0 || (forms_blob && form_offsets), "headline rows: null forms");
    if (n == 0) return OI_OK;
    OI_REQUIRE(offsets && mask_out && order_out && about_out, "headline rows: null buffer");
    OI_REQUIRE(offsets[0] == 0, "headline rows: offsets[0] must be 0");
    const uint64_t bytes = offsets[n];
    OI_REQUIRE(blob || bytes == 0, "headline rows: null title blob");
    for (uint64_t i = 0; i < n; ++i)
        OI_REQUIRE(offsets[i + 1] >= offsets[i] && offsets[i + 1] - offsets[i] < (1ull << 32),
                   "headline rows: offsets must ascend and a title must be shorter than 4 GiB");
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // staging: [titles | offsets | one table set per row] in, [order | mask | about] out
    const size_t pb = oi_headline_params_bytes();
    const size_t off_o = (bytes + 15) & ~(size_t)15, off_p = (off_o + sizeof(uint64_t) * (n + 1) + 15) & ~(size_t)15;
    const size_t in_bytes = off_p + pb * n_rows;
    const size_t out_m = sizeof(uint64_t) * n, out_a = out_m + sizeof(uint16_t) * n, out_bytes = out_a + n;
    DevBuf &si = ctx->buf("hl_rows_in"), &so = ctx->buf("hl_rows_out");
    OI_CHECK(si.ensure(in_bytes + 64));
    OI_CHECK(so.ensure(out_bytes));
    // page-locked staging for the calls it pays for (one ticker's or a dip scan's rows); a large scan packs into pageable
    // memory and copies its results straight into the caller's arrays -- PinBuf never shrinks, and one big call would leave
    // tens of MB page-locked for the life of the ctx (the other OI_HOST entry points follow the same rule)
    const bool pin = in_bytes <= OI_PINNED_STAGE_MAX && out_bytes <= OI_PINNED_STAGE_MAX;
    std::vector<uint8_t> pageable_in;
    uint8_t *h;
    if (pin) {
        OI_CHECK(ctx->pin_in.ensure(in_bytes));
        OI_CHECK(ctx->pin_out.ensure(out_bytes));
        h = ctx->pin_in.as<uint8_t>();
    } else {
        pageable_in.resize(in_bytes);
        h = pageable_in.data();
    }
    if (bytes) memcpy(h, blob, bytes);
    memcpy(h + off_o, offsets, sizeof(uint64_t) * (n + 1));
    for (uint32_t r = 0; r < n_rows; ++r)
        OI_CHECK(oi_headline_build_params(h + off_p + pb * r, tickers_blob ? tickers_blob + ticker_offsets[r] : nullptr,
                                          ticker_offsets[r + 1] - ticker_offsets[r], forms_blob,
                                          form_offsets ? form_offsets + row_form_offsets[r] : nullptr,
                                          row_form_offsets[r + 1] - row_form_offsets[r]));
    OI_HIP_CHECK(hipMemcpyAsync(si.p, h, in_bytes, hipMemcpyHostToDevice, st));
    uint8_t *d = si.as<uint8_t>(), *o = so.as<uint8_t>();
    const uint64_t *d_offs = reinterpret_cast<const uint64_t *>(d + off_o);
    for (uint32_t r = 0; r < n_rows; ++r) {
        const uint64_t t0 = row_offsets[r], t1 = row_offsets[r + 1];
        if (t1 == t0) continue;
        OI_CHECK(oi_launch_headline_scan_params(ctx, d, d_offs + t0, t1 - t0, bytes, offsets[t1] - offsets[t0], d + off_p + pb * r,
                                                reinterpret_cast<uint16_t *>(o + out_m) + t0,
                                                reinterpret_cast<uint64_t *>(o) + t0, o + out_a + t0));
    }
    if (!pin) {
        OI_HIP_CHECK(hipMemcpyAsync(order_out, o, out_m, hipMemcpyDeviceToHost, st));
        OI_HIP_CHECK(hipMemcpyAsync(mask_out, o + out_m, sizeof(uint16_t) * n, hipMemcpyDeviceToHost, st));
        OI_HIP_CHECK(hipMemcpyAsync(about_out, o + out_a, n, hipMemcpyDeviceToHost, st));
        OI_HIP_CHECK(hipStreamSynchronize(st)); // (before pageable_in goes out of scope)
        return OI_OK;
    }
    OI_HIP_CHECK(hipMemcpyAsync(ctx->pin_out.p, so.p, out_bytes, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    const uint8_t *rr = ctx->pin_out.as<uint8_t>();
    memcpy(order_out, rr, out_m);
    memcpy(mask_out, rr + out_m, sizeof(uint16_t) * n);
    memcpy(about_out, rr + out_a, n);
    return OI_OK;
}

extern "C" int oi_social_summary(oi_ctx *ctx, const uint8_t *sources, uint64_t n_posts, const double *polarity,
                                 const uint8_t *speculative, uint64_t n_signals, double tau, int location,
                                 oi_social_counters *out) {
    if (!ctx || !out) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    if (n_signals != n_posts) { // speculation_engine.rs:29-34
        oi_set_error("analyzer returned %llu signals for %llu posts", (unsigned long long)n_signals,
                     (unsigned long long)n_posts);
        return OI_ERR_ANALYZER_MISMATCH;
    }
    const uint64_t n = n_posts;
    if (n == 0) { memset(out, 0, sizeof(*out)); return OI_OK; }
    OI_REQUIRE(polarity && speculative, "social_summary: null buffer");
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    if (location == OI_DEVICE) return oi_launch_social_summary(ctx, sources, polarity, speculative, n, tau, out);
    hipStream_t st = ctx->stream;
    DevBuf &s = ctx->buf("sum_src"), &p = ctx->buf("sum_pol"), &f = ctx->buf("sum_spec");
    OI_CHECK(p.ensure(sizeof(double) * n));
    OI_CHECK(f.ensure(n));
    OI_HIP_CHECK(hipMemcpyAsync(p.p, polarity, sizeof(double) * n, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemcpyAsync(f.p, speculative, n, hipMemcpyHostToDevice, st));
    const uint8_t *dsrc = nullptr;
    if (sources) {
        OI_CHECK(s.ensure(n));
        OI_HIP_CHECK(hipMemcpyAsync(s.p, sources, n, hipMemcpyHostToDevice, st));
        dsrc = s.as<uint8_t>();
    }
    return oi_launch_social_summary(ctx, dsrc, p.as<double>(), f.as<uint8_t>(), n, tau, out);
}

// Per-segment sums of a pooled batch (the batch callers: mcp/tools.rs:193-225, :303-352).
extern "C" int oi_social_summary_segmented(oi_ctx *ctx, const uint8_t *sources, const double *polarity,
                                           const uint8_t *speculative, uint64_t n_posts, const uint64_t *seg_offsets,
                                           uint64_t n_segments, double tau, int location, oi_social_counters *out) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    if (n_segments == 0) return OI_OK;
    OI_REQUIRE(seg_offsets && out, "segmented summary: null buffer");
    OI_REQUIRE((polarity && speculative) || n_posts == 0, "segmented summary: null buffer");
    OI_REQUIRE(location == OI_HOST || location == OI_DEVICE, "segmented summary: bad location");
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    if (location == OI_DEVICE)
        return oi_launch_social_summary_segmented(ctx, sources, polarity, speculative, n_posts, seg_offsets, n_segments, tau, out);
    for (uint64_t s = 0; s < n_segments; ++s)
        OI_REQUIRE(seg_offsets[s] <= seg_offsets[s + 1], "segmented summary: segment %llu has offsets %llu > %llu",
                   (unsigned long long)s, (unsigned long long)seg_offsets[s], (unsigned long long)seg_offsets[s + 1]);
    if (seg_offsets[n_segments] > n_posts) { // the analyzer handed back fewer signals than the segments cover (speculation_engine.rs:29-34)
        oi_set_error("segments cover %llu posts, %llu signals given", (unsigned long long)seg_offsets[n_segments],
                     (unsigned long long)n_posts);
        return OI_ERR_ANALYZER_MISMATCH;
    }
    hipStream_t st = ctx->stream;
    // [polarity f64 | segment offsets u64 | speculative u8 | sources u8] in one buffer, one DMA for a call under 1 MB (a
    // scan_watchlist call: a few thousand signals); the records come back in one
    const size_t pb = sizeof(double) * n_posts, gb = sizeof(uint64_t) * (n_segments + 1);
    const size_t off_g = pb, off_f = off_g + gb, off_s = off_f + ((n_posts + 15) & ~(size_t)15), in_bytes = off_s + n_posts + 16;
    const size_t out_bytes = sizeof(oi_social_counters) * n_segments;
    const bool pinned = in_bytes <= OI_PINNED_STAGE_MAX && out_bytes <= OI_PINNED_STAGE_MAX;
    DevBuf &si = ctx->buf("sum_seg_in"), &ob = ctx->buf("sum_seg_out");
    OI_CHECK(si.ensure(in_bytes));
    OI_CHECK(ob.ensure(out_bytes));
    uint8_t *d = si.as<uint8_t>();
    if (pinned) {
        OI_CHECK(ctx->pin_in.ensure(in_bytes));
        OI_CHECK(ctx->pin_out.ensure(out_bytes));
        uint8_t *h = ctx->pin_in.as<uint8_t>();
        if (n_posts) {
            memcpy(h, polarity, pb);
            memcpy(h + off_f, speculative, n_posts);
            if (sources) memcpy(h + off_s, sources, n_posts);
        }
        memcpy(h + off_g, seg_offsets, gb);
        OI_HIP_CHECK(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st));
    } else {
        if (n_posts) {
            OI_HIP_CHECK(hipMemcpyAsync(d, polarity, pb, hipMemcpyHostToDevice, st));
            OI_HIP_CHECK(hipMemcpyAsync(d + off_f, speculative, n_posts, hipMemcpyHostToDevice, st));
            if (sources) OI_HIP_CHECK(hipMemcpyAsync(d + off_s, sources, n_posts, hipMemcpyHostToDevice, st));
        }
        OI_HIP_CHECK(hipMemcpyAsync(d + off_g, seg_offsets, gb, hipMemcpyHostToDevice, st));
    }
    OI_CHECK(oi_launch_social_summary_segmented(ctx, sources && n_posts ? d + off_s : nullptr, reinterpret_cast<const double *>(d),
                                                d + off_f, n_posts, reinterpret_cast<const uint64_t *>(d + off_g), n_segments, tau,
                                                ob.as<oi_social_counters>()));
    if (pinned) {
        OI_HIP_CHECK(hipMemcpyAsync(ctx->pin_out.p, ob.p, out_bytes, hipMemcpyDeviceToHost, st));
        OI_HIP_CHECK(hipStreamSynchronize(st));
        memcpy(out, ctx->pin_out.p, out_bytes);
    } else {
        OI_HIP_CHECK(hipMemcpyAsync(out, ob.p, out_bytes, hipMemcpyDeviceToHost, st));
        OI_HIP_CHECK(hipStreamSynchronize(st));
    }
    return OI_OK;
}

// LexiconAnalyzer::analyze over the pooled posts of many tickers + every ticker's social_summary sums: two launches on the
// ctx stream, nothing copied, nothing synchronised.
extern "C" int oi_lexicon_scan_segments_device(oi_ctx *ctx, const uint8_t *d_blob, const uint64_t *d_offsets, uint64_t n,
                                               uint64_t blob_bytes, const uint8_t *d_sources, const uint64_t *d_seg_offsets,
                                               uint64_t n_segments, double tau, double *d_pol, uint8_t *d_spec,
                                               oi_social_counters *d_out) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    if (n_segments == 0) return OI_OK;
    OI_REQUIRE(d_seg_offsets && d_out, "scan segments: null buffer");
    OI_REQUIRE(n == 0 || (d_offsets && (d_blob || blob_bytes == 0)), "scan segments: null buffer");
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    if (n) {
        if (!d_pol) { // the signals are wanted only as the reduction's input: the ctx's own workspace
            DevBuf &p = ctx->buf("lex_pol");
            OI_CHECK(p.ensure(sizeof(double) * n));
            d_pol = p.as<double>();
        }
        if (!d_spec) {
            DevBuf &f = ctx->buf("lex_spec");
            OI_CHECK(f.ensure(n));
            d_spec = f.as<uint8_t>();
        }
        OI_CHECK(oi_launch_lexicon(ctx, d_blob, d_offsets, n, blob_bytes, d_pol, d_spec));
    }
    return oi_launch_social_summary_segmented(ctx, d_sources, d_pol, d_spec, n, d_seg_offsets, n_segments, tau, d_out);
}

// ---------------------------------------------------------------- index
static std::atomic<uint64_t> g_index_uid{1};

extern "C" int oi_index_create(oi_ctx *ctx, uint64_t n_docs, uint32_t dim, uint32_t vocab, uint32_t doc_id_base,
                               oi_index **out) {
    if (!ctx || !out) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    *out = nullptr;
    OI_REQUIRE(n_docs > 0 && n_docs + doc_id_base <= 0xFFFFFFFFull, "index: doc ids must fit in 32 bits");
    OI_REQUIRE(dim % 4 == 0 && dim >= 4 && dim <= OI_MAX_DIM, "index: dim=%u must be a multiple of 4 in [4,%u]", dim,
               OI_MAX_DIM);
    OI_REQUIRE(vocab > 0, "index: vocab must be > 0");
    oi_index *idx = new oi_index();
    idx->uid = g_index_uid.fetch_add(1);
    idx->ctx = ctx;
    ctx->refs.fetch_add(1);
    idx->n_docs = n_docs;
    idx->dim = dim;
    idx->vocab = vocab;
    idx->doc_id_base = doc_id_base;
    *out = idx;
    return OI_OK;
}

// A second handle on a finalized index, bound to another context (its own stream and workspaces) of the same device:
// searches through the two handles can be in flight at the same time.  The view borrows every buffer.
extern "C" int oi_index_view(oi_index *src, oi_ctx *ctx, oi_index **out) {
    if (!src || !ctx || !out) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    *out = nullptr;
    OI_REQUIRE(ctx != src->ctx, "index view: give it a context of its own (the point is a second stream and second workspaces)");
    OI_REQUIRE(ctx->device == src->ctx->device, "index view: context on device %d, index on device %d", ctx->device, src->ctx->device);
    std::lock_guard<std::mutex> g(src->ctx->mu); // the source is not being built or searched while it is aliased
    if (!src->finalized || (!src->rows && !src->rows_bf16)) { oi_set_error("index view: the source must have rows and be finalized"); return OI_ERR_STATE; }
    oi_index *v = new oi_index();
    v->uid = g_index_uid.fetch_add(1);
    v->ctx = ctx;
    ctx->refs.fetch_add(1);
    v->src = src;
    src->refs.fetch_add(1); // the buffers it borrows live until the last handle on them is gone
    v->is_view = true;
    v->n_docs = src->n_docs; v->dim = src->dim; v->vocab = src->vocab; v->doc_id_base = src->doc_id_base;
    v->rows = src->rows; v->rows_bf16 = src->rows_bf16; // rows_owned / rows_bf16_owned stay false
    v->screen_ok = src->screen_ok; v->screen_copy_policy = src->screen_copy_policy;
    v->forward_set = src->forward_set; v->finalized = true;
    v->total_tokens = src->total_tokens; v->n_postings = src->n_postings; v->n_blocks = src->n_blocks; v->n_win = src->n_win;
    v->avgdl = src->avgdl; v->max_query_terms = src->max_query_terms; v->bm25_mode = src->bm25_mode;
    auto alias = [](DevBuf &dst, const DevBuf &from) { dst.p = from.p; dst.cap = from.cap; dst.borrowed = from.p != nullptr; };
    alias(v->max_row_norm, src->max_row_norm); alias(v->screen_copy, src->screen_copy);
    alias(v->uniq_keys, src->uniq_keys); alias(v->tf, src->tf); alias(v->doc_len, src->doc_len); alias(v->df_local, src->df_local);
    alias(v->postings, src->postings); alias(v->cell_start, src->cell_start); alias(v->idf, src->idf);
    alias(v->impact_floor, src->impact_floor);
    v->n_long = src->n_long; alias(v->long_list, src->long_list); alias(v->long_bitmap, src->long_bitmap);
    alias(v->fwd_terms, src->fwd_terms); alias(v->fwd_offsets, src->fwd_offsets);
    *out = v;
    return OI_OK;
}

// Drops one reference; the last one frees the buffers (a source index with live views stays until they are gone),
// then lets go of the source (a view) and of the ctx.
static void index_release(oi_index *idx) {
    if (idx->refs.fetch_sub(1) != 1) return;
    oi_ctx *ctx = idx->ctx;
    oi_index *src = idx->src;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        graphs_purge(ctx, idx->uid);
        if (idx->rows_owned && idx->rows) (void)hipFree(idx->rows);
        if (idx->rows_bf16_owned && idx->rows_bf16) (void)hipFree(idx->rows_bf16);
        idx->uniq_keys.release(); idx->tf.release(); idx->doc_len.release(); idx->df_local.release();
        idx->postings.release(); idx->cell_start.release(); idx->idf.release();
        idx->fwd_terms.release(); idx->fwd_offsets.release(); idx->max_row_norm.release(); idx->screen_copy.release();
    }
    delete idx;
    if (src) index_release(src);
    ctx_release(ctx);
}

extern "C" void oi_index_destroy(oi_index *idx) {
    if (!idx) return;
    {   // nothing of this handle's work stays in flight behind the call, whoever frees the buffers in the end
        std::lock_guard<std::mutex> g(idx->ctx->mu);
        (void)hipSetDevice(idx->ctx->device);
        (void)hipStreamSynchronize(idx->ctx->stream);
        if (idx->ctx->side_stream) (void)hipStreamSynchronize(idx->ctx->side_stream);
    }
    index_release(idx);
}

// The bf16 screening copy of an f32 corpus (include/openintel_hip.h: oi_index_set_screen_copy).  Called with the ctx mutex
// held, at finalize and whenever the rows or the policy of a finalized index change.  AUTO: only when the corpus can be
// screened at all and n x d x 2 bytes are at most a quarter (OI_SCREEN_COPY_MAX_FRAC) of the device memory free right now.
static int default_screen_copy_policy() {
    static const int p = [] {
        const char *e = getenv("OI_SCREEN_COPY");
        return !e ? OI_SCREEN_COPY_AUTO : strcmp(e, "never") == 0 ? OI_SCREEN_COPY_NEVER : strcmp(e, "always") == 0 ? OI_SCREEN_COPY_ALWAYS
                                                                                                                  : OI_SCREEN_COPY_AUTO;
    }();
    return p;
}
static int apply_screen_copy_policy(oi_index *idx) {
    oi_ctx *ctx = idx->ctx;
    if (idx->is_view) return OI_OK;
    const int policy = idx->screen_copy_policy < 0 ? default_screen_copy_policy() : idx->screen_copy_policy;
    const bool possible = idx->rows && !idx->rows_bf16 && idx->screen_ok && oi_cosine_screen_supported(idx->dim);
    if (policy == OI_SCREEN_COPY_NEVER || !possible) {
        if (idx->screen_copy.p) { OI_HIP_CHECK(hipStreamSynchronize(ctx->stream)); idx->screen_copy.release(); }
        return OI_OK;
    }
    if (idx->screen_copy.p) return OI_OK; // (set_embeddings releases a copy of the previous rows)
    const size_t bytes = sizeof(uint16_t) * (size_t)idx->n_docs * idx->dim + 64;
    if (policy == OI_SCREEN_COPY_AUTO) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return OI_OK; // no figure, no copy
        static const double frac = [] {
            const char *e = getenv("OI_SCREEN_COPY_MAX_FRAC");
            const double f = e ? atof(e) : 0.25;
            return f > 0.0 && f <= 1.0 ? f : 0.25;
        }();
        if ((double)bytes > frac * (double)free_b) return OI_OK;
    }
    OI_CHECK(idx->screen_copy.ensure(bytes));
    OI_CHECK(oi_launch_make_screen_copy(ctx, idx->rows, idx->n_docs, idx->dim, idx->screen_copy.as<uint16_t>()));
    OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OI_OK;
}

extern "C" int oi_index_set_screen_copy(oi_index *idx, int policy) {
    if (!idx) { oi_set_error("null index"); return OI_ERR_INVALID_ARG; }
    if (idx->is_view) { oi_set_error("index view: read-only (set the policy on the index it was taken from)"); return OI_ERR_STATE; }
    OI_REQUIRE(policy == OI_SCREEN_COPY_AUTO || policy == OI_SCREEN_COPY_NEVER || policy == OI_SCREEN_COPY_ALWAYS,
               "oi_index_set_screen_copy: unknown policy %d", policy);
    oi_ctx *ctx = idx->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    idx->screen_copy_policy = policy;
    if (!idx->finalized) return OI_OK; // applied by finalize
    g_oi_ws_epoch.fetch_add(1);        // captured query calls hold (or lack) the copy's pointer
    return apply_screen_copy_policy(idx);
}

extern "C" int oi_index_bytes(oi_index *idx, uint64_t *rows_owned, uint64_t *screen_copy, uint64_t *bm25) {
    if (!idx) { oi_set_error("null index"); return OI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> g(idx->ctx->mu);
    if (rows_owned)
        *rows_owned = (idx->rows_owned ? sizeof(float) * (uint64_t)idx->n_docs * idx->dim : 0) +
                      (idx->rows_bf16_owned ? sizeof(uint16_t) * (uint64_t)idx->n_docs * idx->dim : 0);
    if (screen_copy) *screen_copy = idx->screen_copy.p && !idx->screen_copy.borrowed ? (uint64_t)idx->screen_copy.cap : 0;
    if (bm25) {
        uint64_t b = 0;
        for (const DevBuf *d : {&idx->uniq_keys, &idx->tf, &idx->doc_len, &idx->df_local, &idx->postings, &idx->cell_start, &idx->idf,
                                &idx->impact_floor, &idx->fwd_terms, &idx->fwd_offsets})
            if (d->p && !d->borrowed) b += d->cap;
        *bm25 = b;
    }
    return OI_OK;
}

extern "C" int oi_index_set_embeddings(oi_index *idx, float *rows, int location, int normalize) {
    if (!idx || !rows) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    if (idx->is_view) { oi_set_error("index view: read-only (set the data on the index it was taken from)"); return OI_ERR_STATE; }
    oi_ctx *ctx = idx->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    // captured query calls (oi_set_graph_replay) hold the OLD rows pointer -- on this ctx and on every view's: stale from here
    // on (the epoch is process-wide; a raw hipMalloc / hipFree or a caller's pointer does not bump it by itself)
    g_oi_ws_epoch.fetch_add(1);
    if (idx->rows_owned && idx->rows) { (void)hipFree(idx->rows); idx->rows = nullptr; idx->rows_owned = false; }
    if (idx->rows_bf16_owned && idx->rows_bf16) (void)hipFree(idx->rows_bf16);
    idx->rows_bf16 = nullptr; idx->rows_bf16_owned = false;
    idx->screen_copy.release(); // a copy of the previous rows
    const size_t bytes = (size_t)idx->n_docs * idx->dim * sizeof(float);
    if (location == OI_DEVICE) {
        OI_REQUIRE(((uintptr_t)rows & 15u) == 0, "index: embedding matrix must be 16-byte aligned");
        idx->rows = rows;
    } else {
        void *p = nullptr;
        OI_HIP_CHECK(hipMalloc(&p, bytes));
        idx->rows = reinterpret_cast<float *>(p);
        idx->rows_owned = true;
        OI_HIP_CHECK(hipMemcpyAsync(idx->rows, rows, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    if (normalize) OI_CHECK(oi_launch_l2_normalize(ctx, idx->rows, idx->n_docs, idx->dim));
    // the bf16 screen's error bound needs max |row| and max |bf16(row) - row| (cosine_prefilter.hip): one more pass
    // over the rows, now
    OI_CHECK(idx->max_row_norm.ensure(64));
    OI_CHECK(oi_launch_row_norm_max(ctx, idx->rows, idx->n_docs, idx->dim, idx->max_row_norm.as<uint32_t>()));
    OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    idx->n_long = 0;
    idx->long_list.release(); idx->long_bitmap.release();
    {   // a corpus whose largest norm is not finite, or so large that bf16 products could overflow, is never screened
        struct { float X, E; double sx2, se2; } mx = {0.f, 0.f, 0.0, 0.0}; // X = max |row|, E = max |bf16(row) - row|, sums of squares
        OI_HIP_CHECK(hipMemcpy(&mx, idx->max_row_norm.p, 24, hipMemcpyDeviceToHost));
        idx->screen_ok = mx.X < 1.0e15f && mx.E < 1.0e15f; // false for NaN
        // Two classes (cosine_prefilter.hip): when the largest norms stand out from the corpus (> 1.5 x the RMS) and the rows
        // responsible are few, they are set aside -- always rescored, never part of the screen's thresholds -- and the margin
        // is built from the maxima over the REST.  A normalised corpus never gets here (no second pass over the rows).
        const uint64_t n = idx->n_docs;
        if (idx->screen_ok && n > 4 * OI_LONG_ROWS_MAX) {
            const float X0 = 1.5f * (float)std::sqrt(mx.sx2 / (double)n), E0 = 1.5f * (float)std::sqrt(mx.se2 / (double)n);
            if ((mx.X > X0 || mx.E > E0) && X0 > 0.f && E0 > 0.f) {
                DevBuf cls;
                OI_CHECK(cls.ensure(16));
                OI_CHECK(idx->long_list.ensure(sizeof(uint32_t) * OI_LONG_ROWS_MAX));
                OI_CHECK(idx->long_bitmap.ensure(((n + 31) / 32) * 4));
                OI_CHECK(oi_launch_row_norm_classes(ctx, idx->rows, n, idx->dim, X0, E0, cls.as<uint32_t>(), idx->long_bitmap.as<uint32_t>(),
                                                    idx->long_list.as<uint32_t>(), OI_LONG_ROWS_MAX));
                uint32_t h[4] = {0, 0, 0, 0};
                OI_HIP_CHECK(hipMemcpyAsync(h, cls.p, 16, hipMemcpyDeviceToHost, ctx->stream));
                OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                if (h[2] >= 1 && h[2] <= OI_LONG_ROWS_MAX) {
                    idx->n_long = h[2];
                    OI_HIP_CHECK(hipMemcpy(idx->max_row_norm.p, h, 8, hipMemcpyHostToDevice)); // the maxima over the other rows
                } else {
                    idx->long_list.release(); idx->long_bitmap.release(); // too many to set aside: one class, the corpus maxima
                }
            }
        }
    }
    if (idx->finalized) OI_CHECK(apply_screen_copy_policy(idx)); // new rows under a finalized index: a new copy, now
    return OI_OK;
}

extern "C" int oi_index_set_embeddings_bf16(oi_index *idx, const uint16_t *rows, int location) {
    if (!idx || !rows) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    if (idx->is_view) { oi_set_error("index view: read-only (set the data on the index it was taken from)"); return OI_ERR_STATE; }
    oi_ctx *ctx = idx->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    // captured query calls (oi_set_graph_replay) hold the OLD rows pointer -- on this ctx and on every view's: stale from here
    // on (the epoch is process-wide; a raw hipMalloc / hipFree or a caller's pointer does not bump it by itself)
    g_oi_ws_epoch.fetch_add(1);
    OI_REQUIRE(oi_cosine_bf16_supported(idx->dim), "index: a bf16 corpus needs dim 384, 768 or 1024 (got %u)", idx->dim);
    if (idx->rows_owned && idx->rows) (void)hipFree(idx->rows);
    idx->rows = nullptr; idx->rows_owned = false;
    idx->screen_copy.release(); // (a copy of f32 rows that are gone)
    if (idx->rows_bf16_owned && idx->rows_bf16) (void)hipFree(idx->rows_bf16);
    idx->rows_bf16 = nullptr; idx->rows_bf16_owned = false;
    const size_t bytes = (size_t)idx->n_docs * idx->dim * sizeof(uint16_t);
    if (location == OI_DEVICE) {
        OI_REQUIRE(((uintptr_t)rows & 15u) == 0, "index: embedding matrix must be 16-byte aligned");
        idx->rows_bf16 = const_cast<uint16_t *>(rows);
    } else {
        void *p = nullptr;
        OI_HIP_CHECK(hipMalloc(&p, bytes));
        idx->rows_bf16 = reinterpret_cast<uint16_t *>(p);
        idx->rows_bf16_owned = true;
        OI_HIP_CHECK(hipMemcpyAsync(idx->rows_bf16, rows, bytes, hipMemcpyHostToDevice, ctx->stream));
        OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return OI_OK;
}

extern "C" int oi_index_set_forward(oi_index *idx, const uint32_t *term_ids, const uint64_t *doc_offsets,
                                    int location) {
    if (!idx || !doc_offsets) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    if (idx->is_view) { oi_set_error("index view: read-only (set the data on the index it was taken from)"); return OI_ERR_STATE; }
    oi_ctx *ctx = idx->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    if (location == OI_DEVICE) return oi_bm25_stage_forward(idx, term_ids, doc_offsets);
    const uint64_t total = doc_offsets[idx->n_docs];
    DevBuf t, o;
    OI_CHECK(t.ensure(sizeof(uint32_t) * (total ? total : 1)));
    OI_CHECK(o.ensure(sizeof(uint64_t) * (idx->n_docs + 1)));
    if (total) OI_HIP_CHECK(hipMemcpyAsync(t.p, term_ids, sizeof(uint32_t) * total, hipMemcpyHostToDevice, ctx->stream));
    OI_HIP_CHECK(hipMemcpyAsync(o.p, doc_offsets, sizeof(uint64_t) * (idx->n_docs + 1), hipMemcpyHostToDevice,
                                ctx->stream));
    int rc = oi_bm25_stage_forward(idx, t.as<uint32_t>(), o.as<uint64_t>());
    (void)hipStreamSynchronize(ctx->stream);
    t.release();
    o.release();
    return rc;
}

extern "C" int oi_index_set_max_query_terms(oi_index *idx, uint32_t max_terms) {
    if (!idx) { oi_set_error("null index"); return OI_ERR_INVALID_ARG; }
    OI_REQUIRE(max_terms >= 1 && max_terms <= 1024, "max_query_terms=%u outside [1,1024]", max_terms);
    std::lock_guard<std::mutex> g(idx->ctx->mu);
    idx->max_query_terms = max_terms;
    return OI_OK;
}

extern "C" int oi_index_long_rows(oi_index *idx, uint32_t *n_out) {
    if (!idx || !n_out) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    *n_out = idx->n_long;
    return OI_OK;
}

extern "C" int oi_index_set_bm25_mode(oi_index *idx, int mode) {
    if (!idx) { oi_set_error("null index"); return OI_ERR_INVALID_ARG; }
    OI_REQUIRE(mode >= 0 && mode <= 4, "bm25 mode %d outside [0,4]", mode);
    std::lock_guard<std::mutex> g(idx->ctx->mu);
    idx->bm25_mode = mode;
    return OI_OK;
}

extern "C" int oi_index_local_stats(oi_index *idx, uint64_t *total_tokens_out, uint32_t *df_out_host) {
    if (!idx) { oi_set_error("null index"); return OI_ERR_INVALID_ARG; }
    if (idx->is_view) { oi_set_error("index view: read-only (set the data on the index it was taken from)"); return OI_ERR_STATE; }
    oi_ctx *ctx = idx->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!idx->forward_set) { oi_set_error("index: set_forward has not been called"); return OI_ERR_STATE; }
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    if (total_tokens_out) *total_tokens_out = idx->total_tokens;
    if (df_out_host) {
        OI_HIP_CHECK(hipMemcpyAsync(df_out_host, idx->df_local.p, sizeof(uint32_t) * idx->vocab,
                                    hipMemcpyDeviceToHost, ctx->stream));
        OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return OI_OK;
}

extern "C" int oi_index_finalize(oi_index *idx, uint64_t global_n_docs, uint64_t global_total_tokens,
                                 const uint32_t *global_df_host) {
    if (!idx) { oi_set_error("null index"); return OI_ERR_INVALID_ARG; }
    if (idx->is_view) { oi_set_error("index view: read-only (set the data on the index it was taken from)"); return OI_ERR_STATE; }
    oi_ctx *ctx = idx->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!idx->forward_set) { oi_set_error("index: set_forward has not been called"); return OI_ERR_STATE; }
    if (idx->finalized) { oi_set_error("index: already finalized"); return OI_ERR_STATE; }
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    OI_CHECK(oi_bm25_finalize(idx, global_n_docs, global_total_tokens, global_df_host));
    return apply_screen_copy_policy(idx); // the derived structure of the cosine leg, beside the BM25 leg's postings
}

// ---------------------------------------------------------------- search
namespace {

struct Pools {
    PoolView cos, bm;
};

// 1 term-at-a-time per workgroup (bm25.hip), 2 scan of the forward index, 3 one wave per task (bm25_wave.hip), 4 the stream
// kernel (bm25_stream.hip: the default).  The index's own setting wins over the process-wide OI_BM25_MODE.
int bm25_mode_of(const oi_index *idx) {
    static const char *mode_env = getenv("OI_BM25_MODE");
    int mode = idx->bm25_mode;
    if (mode == 0 && mode_env)
        mode = strcmp(mode_env, "scan") == 0 ? 2 : strcmp(mode_env, "taat") == 0 ? 1 : strcmp(mode_env, "wave") == 0 ? 3 : 4;
    return mode == 0 ? 4 : mode;
}

// State of both pools in one block, zeroed with ONE memset per search:
//   cosine: carry_cnt[B] tau[B] seg_cnt[B][CUs]      BM25: carry_cnt[B] seg_cnt[B][n_blocks]
// `extra_words` more zeroed words follow them (*extra): the bf16 screen's state, so that one memset kernel does both.
int prepare_pools(oi_ctx *ctx, uint32_t B, uint64_t cos_alloc_stride, uint64_t cos_stride, uint32_t carry_cap, uint32_t bm_blocks,
                  uint32_t depth, Pools *out, size_t extra_words = 0, uint32_t **extra = nullptr) {
    DevBuf &flag = ctx->buf("state_flag");
    if (!flag.p) {
        OI_CHECK(flag.ensure(16));
        OI_HIP_CHECK(hipMemsetAsync(flag.p, 0, 16, ctx->stream));
    }
    const uint32_t cos_segs = (uint32_t)ctx->num_cus * (B <= 8 ? 8u : 1u); // one per workgroup: GEMV grids are 8 per CU
    const uint32_t bm_segs = bm_blocks ? bm_blocks : 1;
    const size_t words = (size_t)B * (2 + cos_segs + 2 + bm_segs);
    DevBuf &st = ctx->buf("pool_state");
    OI_CHECK(st.ensure(sizeof(uint32_t) * (words + extra_words)));
    OI_HIP_CHECK(hipMemsetAsync(st.p, 0, sizeof(uint32_t) * (words + extra_words), ctx->stream));
    if (extra) *extra = st.as<uint32_t>() + words;
    const uint64_t bm_stride = (uint64_t)carry_cap + (uint64_t)bm_segs * depth;
    DevBuf &pc = ctx->buf("pool_cos"), &pb = ctx->buf("pool_bm");
    OI_CHECK(pc.ensure(sizeof(uint64_t) * (size_t)B * cos_alloc_stride)); // (room for the widest view of it: the screen's)
    OI_CHECK(pb.ensure(sizeof(uint64_t) * (size_t)B * bm_stride));
    uint32_t *s = st.as<uint32_t>();
    uint32_t *cos_carry = s, *cos_tau = s + B, *cos_seg = s + 2 * (size_t)B;
    uint32_t *bm_carry = cos_seg + (size_t)B * cos_segs, *bm_tau = bm_carry + B, *bm_seg = bm_tau + B;
    out->cos = PoolView{pc.as<uint64_t>(), cos_carry, cos_seg, cos_tau, cos_stride, carry_cap, 0, 0, cos_segs,
                        flag.as<uint32_t>()};
    out->bm = PoolView{pb.as<uint64_t>(), bm_carry, bm_seg, bm_tau, bm_stride, carry_cap, depth, bm_segs, bm_segs,
                       flag.as<uint32_t>()};
    return OI_OK;
}

// Rows of the first corpus chunk (scored with no threshold yet: every row lands in the pool, so it is kept
// small); each later chunk is 8x the one before.  OI_FIRST_CHUNK_MULT scales it (A/B runs).
static uint64_t oi_first_chunk_rows(uint32_t depth) {
    static const uint64_t mult = oi_ablation_env("OI_FIRST_CHUNK_MULT") ? std::max(1, atoi(oi_ablation_env("OI_FIRST_CHUNK_MULT"))) : 1;
    static const uint64_t div = oi_ablation_env("OI_FIRST_CHUNK_DIV") ? std::max(1, atoi(oi_ablation_env("OI_FIRST_CHUNK_DIV"))) : 1; // (A/B)
    return std::max<uint64_t>(std::max<uint64_t>(8192, 32ull * depth) * mult / div, 2ull * depth);
}
// The screen's first chunk (round 4): a whole number of ROUNDS of the persistent grid -- 7/8 of the CUs x 4 waves x 32-row tiles
// (cosine_prefilter.hip: oi_cosine_screen_geometry) -- so that no wave of the two short first launches runs one tile more than
// the others (32000 rows = 1000 tiles on 896 waves: 104 waves with two tiles; 256000 rows: 9.1 per wave, i.e. 10 rounds).
// Chunk k is 8^k times the first and keeps the property.  OI_SCREEN_NO_ROUND=1 (A/B): as before.
static uint64_t oi_screen_first_chunk_rows(const oi_ctx *ctx, uint32_t depth) {
    static const bool no_round = oi_ablation_env("OI_SCREEN_NO_ROUND") != nullptr;
    uint64_t rows = oi_first_chunk_rows(depth);
    const uint64_t round = 32ull * 4 * std::max<uint64_t>(1, (uint64_t)ctx->num_cus * 7 / 8);
    if (!no_round && rows >= round) rows -= rows % round;
    return rows;
}
// Measured (tools/growth_ab.sh): 8 is best for the MFMA batch path at 10M and 1.25M rows (more survivors per
// chunk cost more in the epilogue and the select than the launch they save); the GEMV path (B <= 8) gains
// 3 % from 16 (1M rows: 3 launches instead of 4).
static uint64_t oi_chunk_growth(uint32_t B) {
    static const uint64_t g = oi_ablation_env("OI_CHUNK_GROWTH") ? std::max(2, atoi(oi_ablation_env("OI_CHUNK_GROWTH"))) : 0;
    return g ? g : (B <= 8 ? 16 : 8);
}

// End of the corpus chunk that starts at row r: `chunk` rows, but a tail shorter than a quarter of the chunk is taken along
// (a 2.5M-row shard: 32K, 256K, 2.2M rows instead of 32K, 256K, 2M and a fourth launch + select for 0.2M).
// And when what is left after this chunk would not fit ONE more chunk but fits two, this chunk grows so that the last one is
// exactly the largest the pool takes (10M rows, 6.8M-row pool: 32K, 256K, 2.9M, 6.8M instead of 32K, 256K, 2M, 6.8M, 0.9M).
// (only where the chunk AFTER this one would be cut by the pool anyway -- `next_chunk`, its planned size, reaches max_chunk --
// never for the small first chunks, which run without a threshold).
static uint64_t oi_chunk_end(uint64_t r, uint64_t chunk, uint64_t n, uint64_t max_chunk, uint64_t next_chunk) {
    uint64_t e = std::min(n, r + chunk);
    if (e < n && (n - e) * 4 <= (e - r) && n - r <= max_chunk) e = n;
    if (e < n && next_chunk >= max_chunk && n - e > max_chunk && n - r <= 2 * max_chunk) e = n - max_chunk;
    return e;
}

// Device-side ranked lists for a batch; all pointers device.
int search_lists_device(oi_index *idx, const float *d_qv, const uint32_t *d_qt, const uint32_t *d_qo, uint32_t B,
                        uint32_t depth, float *cos_s, uint32_t *cos_d, uint32_t *cos_c, float *bm_s,
                        uint32_t *bm_d, uint32_t *bm_c) {
    oi_ctx *ctx = idx->ctx;
    hipStream_t st = ctx->stream;
    const uint64_t n = idx->n_docs;
    // ---- pool capacities
    // cosine: the corpus is scored in chunks; a chunk can append at most one entry per row and
    // query, so a chunk sized from the pool's free room can never overflow it (no overflow path
    // to handle, no data-dependent sizing).  BM25: every doc block contributes <= depth entries.
    const uint32_t carry_cap = OI_MAX_DEPTH;
    const uint64_t slack = (idx->rows_bf16 ? 128ull : 32ull) * ((uint64_t)ctx->num_cus + 1);
    // Large pools = few launches: at 10M rows the schedule is 32K, 256K, 3.2M, 6.5M rows (4 launches).  The room is worst
    // case (every row of a chunk passes the threshold), only entries that pass are written.  Round 4: an f32 corpus gets
    // 3.25 GiB of pool instead of 8 (the last two chunks are balanced so that the launch count stays), and the screen's pool
    // and the exact fallback's pool are ONE buffer (they are never live together: the gated exact pipeline starts after the
    // rescoring has consumed the screen's survivors).  A bf16 corpus (configs[4]: 256 queries) keeps 8 GiB.
    uint64_t cos_stride = 1ull << 24;
    const uint64_t budget = (idx->rows_bf16 ? (8ull << 30) : (13ull << 28)) / 8 / B;
    if (cos_stride > budget) cos_stride = budget;
    if (cos_stride < carry_cap + 4 * slack) cos_stride = carry_cap + 4 * slack;
    if (cos_stride > carry_cap + n + slack) cos_stride = carry_cap + n + slack;
    // the screen's view of the same buffer keeps up to 4096 keys per query between chunks and rounds its segments to 4 tiles
    const uint32_t pf_carry = 4096;
    const uint64_t pf_slack = 128ull * ((uint64_t)ctx->num_cus + 1);
    uint64_t pf_stride = 1ull << 24;
    if (pf_stride > budget) pf_stride = budget;
    if (pf_stride < pf_carry + 4 * pf_slack) pf_stride = pf_carry + 4 * pf_slack;
    if (pf_stride > pf_carry + n + pf_slack) pf_stride = pf_carry + n + pf_slack;
    Pools P;
    // the bf16 screen's state words (carry_cnt[B] tau[B] rs_cnt[B] eps2[B] gate[4] seg_cnt[B][CUs]) ride in the same memset
    const size_t screen_words = (size_t)B * (6 + (size_t)ctx->num_cus) + 4; // (+ spec_tau[B] spec_max[B]: speculative thresholds)
    uint32_t *screen_state = nullptr;
    // (the depth-sized segments of P.bm belong to the workgroup-per-block kernel: no room is set aside for them otherwise)
    OI_CHECK(prepare_pools(ctx, B, std::max<uint64_t>(cos_stride, idx->rows_bf16 ? 0ull : pf_stride), cos_stride, carry_cap,
                           bm25_mode_of(idx) == 1 ? idx->n_blocks : 0, depth, &P, screen_words, &screen_state));

    // Speculative thresholds of the screen (cosine_prefilter.hip, pf_spec_kernel; oi_set_screen_speculation).  Decided here because
    // the chunk schedule depends on it: with a predicted threshold after the first chunk the second can be as large as the pool takes
    // (10M rows 2.64 -> 2.54 ms, a 1.25M-row shard 0.574 -> 0.540 -> 0.523 with the short first chunk; tools/r05_spec_sched.sh), with
    // proven thresholds it must grow slowly (x 8).  Off: oi_set_screen_speculation(ctx, 0); with graph replay (the host
    // decides per call); for batches of <= 8 queries (their survivors cost next to nothing, the extra launches 11 us of 0.34 ms);
    // for spec_skip searches after a failed check.  OI_NO_SPEC=1, OI_SPEC_GROWTH (ablation builds): A/B.
    bool spec_on = false;
    uint64_t screen_growth = oi_chunk_growth(B), screen_first = oi_screen_first_chunk_rows(ctx, depth);
    if (cos_s && idx->rows && !idx->rows_bf16 && B > 8 && idx->screen_ok && oi_cosine_screen_supported(idx->dim) &&
        (ctx->cosine_mode == OI_COSINE_SCREEN || ctx->cosine_mode == OI_COSINE_SCREEN_COPY || ctx->cosine_mode == OI_COSINE_SCREEN_STREAM)) {
        static const bool spec_env_off = oi_ablation_env("OI_NO_SPEC") != nullptr;
        static const uint64_t spec_growth = oi_ablation_env("OI_SPEC_GROWTH") ? std::max(2, atoi(oi_ablation_env("OI_SPEC_GROWTH"))) : 128;
        static const uint64_t spec_first_div = oi_ablation_env("OI_SPEC_FIRST_DIV") ? std::max(1, atoi(oi_ablation_env("OI_SPEC_FIRST_DIV"))) : 4;
        if (ctx->spec_fail_host && *ctx->spec_fail_host) { // a batch since the last look failed its check: back off
            *ctx->spec_fail_host = 0;
            ++ctx->spec_failures;
            ctx->spec_backoff = ctx->spec_backoff ? std::min(1024u, 2 * ctx->spec_backoff) : 16u;
            ctx->spec_skip = ctx->spec_backoff;
        }
        spec_on = ctx->speculate && !spec_env_off && !ctx->use_graphs;
        if (spec_on && ctx->spec_skip) { --ctx->spec_skip; spec_on = false; }
        if (spec_on && !ctx->spec_fail_host) {
            if (hipHostMalloc(reinterpret_cast<void **>(&ctx->spec_fail_host), 64, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                ctx->spec_fail_host = nullptr;
                spec_on = false; // (no way to hear of a failed check: no speculation)
            } else *ctx->spec_fail_host = 0;
        }
        if (spec_on && n) { // does the rank after the FIRST chunk qualify (pf_spec_kernel is launched when 2 r <= k')?
            const uint64_t pool_max = pf_stride > pf_carry + pf_slack ? pf_stride - pf_carry - pf_slack : 0; // rows one launch may take
            // a SHORT first chunk (a quarter of the proven schedule's, >= 8192 rows, >= 8 k': its only job is the sample the
            // prediction is read from) when everything after it fits ONE launch -- a shard: 8 192 rows, then the rest
            // (1.25M rows: 0.540 -> 0.523 ms against 28 672 + the rest); a corpus that needs three launches anyway keeps the
            // regular first chunk and grows x 128 (10M: 28 672, 3.67M, 6.3M rows; a short first chunk measured the same there)
            const uint64_t first_short = std::min<uint64_t>(n, std::max<uint64_t>(std::max<uint64_t>(8192, 8ull * depth), screen_first / spec_first_div));
            if (n - first_short <= pool_max && 2 * ((3ull * depth * first_short + n - 1) / n + 12) <= depth) {
                screen_growth = spec_growth;
                screen_first = first_short;
            } else if (2 * ((3ull * depth * std::min<uint64_t>(n, screen_first) + n - 1) / n + 12) <= depth) screen_growth = spec_growth;
        }
    }

    // The two legs of a hybrid query are independent until fusion: the BM25 leg (latency-bound, 128 KB of
    // LDS per workgroup) is issued on a side stream and fills the issue slots the MFMA-bound cosine leg
    // leaves, instead of running after it.  OI_NO_OVERLAP=1 serialises them (A/B runs).
    static const bool no_overlap = oi_ablation_env("OI_NO_OVERLAP") != nullptr;
    if (cos_s && bm_s && !ctx->side_stream && !ctx->side_stream_failed && ctx->overlap_legs && !no_overlap) {
        // (default priority: at the lowest one the BM25 leg stretched over the whole cosine leg and the step was no shorter)
        if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess) { ctx->side_stream = nullptr; ctx->side_stream_failed = true; }
    }
    const bool overlap = cos_s && bm_s && ctx->side_stream && ctx->ev_fork && ctx->ev_join && ctx->overlap_legs && !no_overlap;
    // ---- BM25 list
    auto bm25_leg = [&]() -> int {
        hipStream_t st = ctx->stream; // (the side stream when the legs overlap)
        (void)st;
        OI_REQUIRE(idx->finalized, "search: index not finalized");
        // Which BM25 kernel.  Default: the stream kernel (bm25_stream.hip).  The wave-per-task kernel (bm25_wave.hip), the
        // first-generation workgroup-per-block kernel (bm25.hip) and the batch scan of the forward index
        // (bm25_scan.hip) stay selectable per index (oi_index_set_bm25_mode) or process-wide with
        // OI_BM25_MODE=stream|wave|taat|scan; all four return bit-identical lists.
        const int mode = bm25_mode_of(idx);
        const bool have_fwd = idx->fwd_terms.p && idx->total_tokens > 0;
        const bool scan = have_fwd && mode == 2;
        if (!scan && mode != 1 && mode != 3) {
            // The STREAM kernel (bm25_stream.hip, the default).  Two phases like the cosine chunks: the first eighth of the doc
            // blocks is scored with no threshold and fixes tau_q = the depth-th score so far, a lower bound of the final
            // one; the remaining blocks emit only scores >= tau_q.  A task's pool segment is SMALL and fixed (4096 keys
            // in the first phase, depth + 256 in the second): a segment that would overflow is pruned in place to its
            // top `depth` keys, so nothing can overflow whatever the data, and the pool is ~3 MB per query at 10M docs.
            const uint32_t nb = idx->n_blocks;
            if (nb == 0 || idx->n_postings == 0) {
                OI_HIP_CHECK(hipMemsetAsync(bm_c, 0, sizeof(uint32_t) * B, st));
                return OI_OK;
            }
            // (the share of the blocks scored without a threshold: 1/8 and 1/16 measure the same, 1/32 is 5 % slower)
            static const uint32_t first_div = oi_ablation_env("OI_BM25_FIRST_DIV") ? std::max(1, atoi(oi_ablation_env("OI_BM25_FIRST_DIV"))) : 8;
            // Up to 48 blocks (1.5M docs: a shard of configs[3]) ONE phase: every touched doc is a candidate (~30K keys per query,
            // the select's register path), one launch and one select fewer -- 0.052 vs 0.081 ms of kernels at 1.25M docs.
            // Round 4, second half: NO threshold-less phase at all when the index has its per-term impact floors (bm25.hip): the
            // plan kernel starts every query at max_t fl(idf_t * floor_t) -- at least `depth` docs score that much, so it is a valid
            // lower bound of the depth-th best score before a posting is read -- and ONE launch scores every block against it, with the
            // small pruned segments of the former second phase.  At 10M docs that bound is HIGHER than the first phase's (the 1024th
            // impact of one term over all docs vs the 1000th score over an eighth of them), and a launch, a select and the first
            // phase's 30 K candidates per query go away.  OI_BM25_TWO_PHASE=1 (A/B): the phases as before.
            static const bool two_phase_env = oi_ablation_env("OI_BM25_TWO_PHASE") != nullptr;
            const bool floors = idx->impact_floor.p != nullptr && !two_phase_env;
            const uint32_t first = floors ? nb : nb > 48 ? std::max<uint32_t>(8, nb / first_div) : nb;
            const uint32_t cap1 = oi_bm25_stream_seg_cap(depth, !floors), cap2 = oi_bm25_stream_seg_cap(depth, false);
            const uint64_t sstride = (uint64_t)carry_cap + std::max<uint64_t>((uint64_t)first * cap1, (uint64_t)nb * cap2);
            uint64_t pass = (2ull << 30) / 8 / sstride; // <= 2 GiB of pool (0.2 GB for 64 queries over 10M docs)
            pass = std::max<uint64_t>(1, std::min<uint64_t>(pass, std::min<uint32_t>(B, oi_bm25_stream_pass_queries())));
            DevBuf &sp = ctx->buf("pool_bm_stream"), &sc = ctx->buf("pool_bm_stream_state");
            OI_CHECK(sp.ensure(sizeof(uint64_t) * (size_t)pass * sstride));
            const size_t swords = (size_t)pass * (2 + nb);
            OI_CHECK(sc.ensure(sizeof(uint32_t) * swords));
            for (uint32_t q0 = 0; q0 < B; q0 += (uint32_t)pass) {
                const uint32_t nq = std::min<uint32_t>((uint32_t)pass, B - q0);
                uint32_t *w = sc.as<uint32_t>();
                // (the plan launch also zeroes the pass's pool state: carry_cnt[pass] tau[pass] seg_cnt[pass][nb])
                OI_CHECK(oi_launch_bm25_plan(idx, d_qt, d_qo, q0, nq, w, swords, depth, (uint32_t)pass, (uint32_t)pass, floors));
                PoolView W1{sp.as<uint64_t>(), w, w + 2 * (size_t)pass, w + pass, sstride, carry_cap, cap1, first, nb, P.bm.overflow};
                OI_CHECK(oi_launch_bm25_stream(idx, d_qt, d_qo, q0, nq, depth, W1, 0, first));
                if (first < nb) {
                    OI_CHECK(oi_launch_select(ctx, W1, nq, depth, /*compact=*/true, nullptr, nullptr, nullptr, depth));
                    PoolView W2 = W1;
                    W2.seg_cap = cap2; W2.n_segs = nb; // the first phase's segments are empty again: the same memory, cut anew
                    OI_CHECK(oi_launch_bm25_stream(idx, d_qt, d_qo, q0, nq, depth, W2, first, nb));
                    OI_CHECK(oi_launch_select(ctx, W2, nq, depth, false, bm_s + (size_t)q0 * depth, bm_d + (size_t)q0 * depth, bm_c + q0, depth));
                } else
                    OI_CHECK(oi_launch_select(ctx, W1, nq, depth, false, bm_s + (size_t)q0 * depth, bm_d + (size_t)q0 * depth, bm_c + q0, depth));
            }
            return OI_OK;
        }
        if (!scan && mode == 3) {
            // The wave-per-task kernel.  Two phases like the cosine chunks: the first eighth of the doc blocks is scored with no threshold
            // (every touched doc is a candidate) and fixes tau_q = the depth-th score so far, a lower bound of
            // the final one; the remaining blocks emit only scores >= tau_q.  A task's pool segment holds a whole
            // block, so nothing can overflow; the room is address space, not traffic (only emitted keys are
            // written).  Queries go in passes sized from a 6 GiB pool budget.
            const uint32_t nb = idx->n_blocks;
            if (nb == 0 || idx->n_postings == 0) {
                OI_HIP_CHECK(hipMemsetAsync(bm_c, 0, sizeof(uint32_t) * B, st));
                return OI_OK;
            }
            const uint64_t wstride = (uint64_t)carry_cap + (uint64_t)nb * OI_BM25_BLOCK_DOCS;
            uint64_t pass = (6ull << 30) / 8 / wstride;
            pass = std::max<uint64_t>(1, std::min<uint64_t>(pass, std::min<uint32_t>(B, oi_bm25_wave_pass_queries())));
            DevBuf &wp = ctx->buf("pool_bm_wave"), &wc = ctx->buf("pool_bm_wave_state");
            OI_CHECK(wp.ensure(sizeof(uint64_t) * (size_t)pass * wstride));
            const size_t wwords = (size_t)pass * (2 + nb);
            OI_CHECK(wc.ensure(sizeof(uint32_t) * wwords));
            const uint32_t first = nb > 16 ? std::max<uint32_t>(8, nb / 8) : nb;
            for (uint32_t q0 = 0; q0 < B; q0 += (uint32_t)pass) {
                const uint32_t nq = std::min<uint32_t>((uint32_t)pass, B - q0);
                OI_HIP_CHECK(hipMemsetAsync(wc.p, 0, sizeof(uint32_t) * wwords, st));
                uint32_t *w = wc.as<uint32_t>();
                PoolView W{wp.as<uint64_t>(), w, w + 2 * (size_t)pass, w + pass, wstride, carry_cap, OI_BM25_BLOCK_DOCS, nb, nb, P.bm.overflow};
                OI_CHECK(oi_launch_bm25_wave(idx, d_qt, d_qo, q0, nq, W, 0, first));
                if (first < nb) {
                    OI_CHECK(oi_launch_select(ctx, W, nq, depth, /*compact=*/true, nullptr, nullptr, nullptr, depth));
                    OI_CHECK(oi_launch_bm25_wave(idx, d_qt, d_qo, q0, nq, W, first, nb));
                }
                OI_CHECK(oi_launch_select(ctx, W, nq, depth, false, bm_s + (size_t)q0 * depth, bm_d + (size_t)q0 * depth, bm_c + q0, depth));
            }
            return OI_OK;
        }
        if (!scan) {
            // Term-at-a-time, one WORKGROUP per doc block (the first-generation kernel).  Two phases, like the cosine chunks: the
            // first eighth of the doc blocks fixes a per-query threshold (the depth-th score seen so far
            // is a lower bound of the final one); the remaining blocks then emit only candidates at or
            // above it, so the final selection scans little.
            const uint32_t nb = idx->n_blocks;
            const uint32_t first = nb > 16 ? std::max<uint32_t>(8, nb / 8) : nb;
            OI_CHECK(oi_launch_bm25(idx, d_qt, d_qo, B, depth, P.bm, 0, first));
            if (first < nb) {
                OI_CHECK(oi_launch_select(ctx, P.bm, B, depth, /*compact=*/true, nullptr, nullptr, nullptr, depth));
                OI_CHECK(oi_launch_bm25(idx, d_qt, d_qo, B, depth, P.bm, first, nb));
            }
            OI_CHECK(oi_launch_select(ctx, P.bm, B, depth, false, bm_s, bm_d, bm_c, depth));
        } else {
            // Batch scan of the forward index (bm25_scan.hip): the whole batch in passes of up to
            // 1024 / max_query_terms queries; docs in chunks sized from the pool's free room, the first
            // chunk (1/8 of the docs) fixing the thresholds.  Same worst-case rule as the cosine pools:
            // a chunk can append at most one entry per doc and query.
            const uint32_t segs = 2u * (uint32_t)ctx->num_cus;
            const uint64_t sslack = 1024ull * (segs + 1); // oi_bm25_scan_geometry: a workgroup's docs, rounded up by two tiles
            uint64_t sstride = carry_cap + std::min<uint64_t>(n, 1ull << 23) + sslack;
            const uint64_t sbudget = (4ull << 30) / 8 / B;
            if (sstride > sbudget) sstride = std::max<uint64_t>(sbudget, carry_cap + 4 * sslack);
            DevBuf &sp = ctx->buf("pool_bm_scan"), &sc = ctx->buf("pool_bm_scan_state");
            OI_CHECK(sp.ensure(sizeof(uint64_t) * (size_t)B * sstride));
            const size_t swords = (size_t)B * (2 + segs);
            OI_CHECK(sc.ensure(sizeof(uint32_t) * swords));
            OI_HIP_CHECK(hipMemsetAsync(sc.p, 0, sizeof(uint32_t) * swords, st));
            uint32_t *w = sc.as<uint32_t>();
            PoolView SP{sp.as<uint64_t>(), w, w + 2 * (size_t)B, w + B, sstride, carry_cap, 0, 0, segs, P.bm.overflow};
            const uint64_t max_chunk = sstride - carry_cap - sslack;
            const uint32_t pass = oi_bm25_scan_pass_queries(idx->max_query_terms);
            for (uint32_t q0 = 0; q0 < B; q0 += pass) {
                const uint32_t nq = std::min(pass, B - q0);
                PoolView V = SP;
                V.carry_cnt += q0; V.tau_keys += q0; // keys / seg_cnt are offset inside the kernel by q_begin
                uint64_t r = 0, chunk = std::max<uint64_t>(n / 8, 65536);
                bool first_chunk = true;
                while (r < n) {
                    if (chunk > max_chunk) chunk = max_chunk;
                    const uint64_t e = std::min(n, r + chunk);
                    oi_bm25_scan_geometry(ctx, e - r, &V.n_segs, &V.seg_cap);
                    SP.n_segs = V.n_segs; SP.seg_cap = V.seg_cap;
                    OI_CHECK(oi_launch_bm25_scan(idx, d_qt, d_qo, q0, nq, r, e, idx->avgdl, first_chunk, SP));
                    const bool last = e == n;
                    PoolView S2 = SP; // select works on this pass's queries only
                    S2.keys += (uint64_t)q0 * sstride; S2.carry_cnt += q0; S2.tau_keys += q0;
                    S2.seg_cnt += (uint64_t)q0 * segs;
                    OI_CHECK(oi_launch_select(ctx, S2, nq, depth, /*compact=*/!last, last ? bm_s + (size_t)q0 * depth : nullptr,
                                              last ? bm_d + (size_t)q0 * depth : nullptr, last ? bm_c + q0 : nullptr, depth));
                    r = e;
                    chunk = n; // everything that is left, as far as the pool allows
                    first_chunk = false;
                }
            }
        }
        return OI_OK;
    };
    auto fork_bm25 = [&]() -> int {
        OI_HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
        ctx->stream = ctx->side_stream;
        const int rc = bm25_leg();
        ctx->stream = st;
        if (rc != OI_OK) { (void)hipStreamSynchronize(ctx->side_stream); return rc; } // nothing of this call stays in flight
        OI_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->side_stream));
        return OI_OK;
    };
    // Round 4: beside the screen the BM25 leg starts with the LAST corpus chunk, not the first.  The screen's persistent
    // workgroups leave 1/8 of the CUs free; the BM25 kernels (enqueued AFTER the last chunk's launch, so that the screen's
    // workgroups are resident first) run there while the long chunk streams -- instead of sharing the CUs with the two short
    // first chunks, whose launches they stretched (10M rows: step 4.998 -> 4.940 ms on one box, -44 .. -58 us on three;
    // tools/r04_epilogue_probe.sh, r04_old_new_ab.sh).  Only when the last chunk is long enough to cover the leg: >= 512K rows.
    // OI_BM25_EARLY=1 (A/B): the round-3 placement.
    static const bool early_env = oi_ablation_env("OI_BM25_EARLY") != nullptr;
    bool late_pending = false;
    if (overlap) {
        bool late = false;
        if (!early_env && !idx->rows_bf16 && idx->rows &&
            (ctx->cosine_mode == OI_COSINE_SCREEN || ctx->cosine_mode == OI_COSINE_SCREEN_COPY || ctx->cosine_mode == OI_COSINE_SCREEN_STREAM) && B > 8 && idx->screen_ok &&
            oi_cosine_screen_supported(idx->dim) && pf_stride > pf_carry + pf_slack) {
            const uint64_t mc = pf_stride - pf_carry - pf_slack; // the screen's own schedule (cosine_leg below), dry
            uint64_t chunk = screen_first, r = 0, last = 0;
            while (r < n) {
                if (chunk > mc) chunk = mc;
                const uint64_t e = oi_chunk_end(r, chunk, n, mc, r == 0 ? 0 : chunk * screen_growth);
                last = e - r;
                r = e;
                chunk *= screen_growth;
            }
            late = last >= (512u << 10);
        }
        if (late) late_pending = true;
        else {
            OI_HIP_CHECK(hipEventRecord(ctx->ev_fork, st)); // pools are reset, queries staged
            OI_CHECK(fork_bm25());
        }
    }
    // ---- cosine list
    auto cosine_leg = [&]() -> int {
        if (cos_s && idx->rows_bf16) {
            // bf16 corpus: same chunk schedule; a workgroup's segment is rounded up to four tiles per wave round
            const uint64_t bslack = 128ull * ((uint64_t)ctx->num_cus + 1);
            const uint64_t room = cos_stride - carry_cap;
            const uint64_t max_chunk = room > bslack ? room - bslack : 0;
            if (max_chunk == 0) { oi_set_error("search: cosine pool too small"); return OI_ERR_STATE; }
            uint64_t chunk = oi_first_chunk_rows(depth);
            uint64_t r = 0;
            while (r < n) {
                if (chunk > max_chunk) chunk = max_chunk;
                const uint64_t e = oi_chunk_end(r, chunk, n, max_chunk, chunk * oi_chunk_growth(B));
                OI_CHECK(oi_launch_cosine_bf16_chunk(ctx, idx->rows_bf16, r, e, idx->dim, d_qv, B, idx->doc_id_base, P.cos));
                const bool last = e == n;
                OI_CHECK(oi_launch_select(ctx, P.cos, B, depth, /*compact=*/!last, last ? cos_s : nullptr,
                                          last ? cos_d : nullptr, last ? cos_c : nullptr, depth));
                r = e;
                chunk *= oi_chunk_growth(B);
            }
        } else if (cos_s) {
            OI_REQUIRE(idx->rows, "search: embeddings not set");
            const uint32_t Bp = oi_cosine_query_padding(B);
            const float *q = d_qv;
            if (Bp != B) {
                DevBuf &qp = ctx->buf("q_padded");
                OI_CHECK(qp.ensure(sizeof(float) * (size_t)Bp * idx->dim));
                OI_HIP_CHECK(hipMemsetAsync(qp.p, 0, sizeof(float) * (size_t)Bp * idx->dim, st));
                OI_HIP_CHECK(hipMemcpyAsync(qp.p, d_qv, sizeof(float) * (size_t)B * idx->dim, hipMemcpyDeviceToDevice, st));
                q = qp.as<float>();
            }
            const uint64_t max_chunk = oi_cosine_max_chunk_rows(ctx, idx->dim, B, cos_stride, carry_cap);
            if (max_chunk == 0) { oi_set_error("search: cosine pool too small"); return OI_ERR_STATE; }
            // the exact pipeline; with a gate it is the fallback behind the bf16 screen and every launch exits at
            // once unless the screen opened the gate
            // With a gate and the screen's thresholds it is the fallback behind the bf16 screen: tau~ - 2 eps is a valid
            // lower bound of the exact k'-th score even when the survivors did not fit, so the exact kernel takes all
            // rows in ONE launch (as many as the pool holds) -- two launches that exit at once when the gate is shut.
            auto exact_pipeline = [&](const uint32_t *gate, uint32_t *screen_tau) -> int {
                SelectExtra ex;
                ex.run_gate = gate;
                PoolView X = P.cos;
                if (screen_tau) X.tau_keys = screen_tau;
                uint64_t chunk = gate ? max_chunk : oi_first_chunk_rows(depth);
                uint64_t r = 0;
                while (r < n) {
                    if (chunk > max_chunk) chunk = max_chunk;
                    const uint64_t e = oi_chunk_end(r, chunk, n, max_chunk, chunk * oi_chunk_growth(B));
                    OI_CHECK(oi_launch_cosine_chunk(ctx, idx->rows, r, e, idx->dim, q, B, Bp, idx->doc_id_base, X));
                    const bool last = e == n;
                    OI_CHECK(oi_launch_select(ctx, X, B, depth, /*compact=*/!last, last ? cos_s : nullptr,
                                              last ? cos_d : nullptr, last ? cos_c : nullptr, depth, gate ? &ex : nullptr));
                    r = e;
                    chunk *= oi_chunk_growth(B);
                }
                return OI_OK;
            };
            static const bool shape16 = !(oi_ablation_env("OI_KS_SHAPE") && atoi(oi_ablation_env("OI_KS_SHAPE")) == 32);
            static const bool cos_v1 = oi_ablation_env("OI_COSINE_V1") != nullptr || oi_ablation_env("OI_SELECT_V1") != nullptr;
            // (a view never makes a copy of its own: it streams the source's if that exists, the f32 rows otherwise)
            // OI_COSINE_SCREEN streams the index's bf16 screening copy when there is one (made at finalize, budget permitting);
            // _COPY also makes a missing one now; _STREAM converts the f32 rows on the fly whatever the index holds
            const bool want_copy = ctx->cosine_mode != OI_COSINE_SCREEN_STREAM &&
                                   (idx->screen_copy.p != nullptr || (ctx->cosine_mode == OI_COSINE_SCREEN_COPY && !idx->is_view));
            // B <= 8 (configs[1]: one query): screened only when there is a copy to stream -- half the bytes of the f32 GEMV, which
            // is HBM-bound; the f32-stream screen would read what the GEMV reads.  OI_SMALL_BATCH_GEMV=1 (A/B): as before round 5.
            static const bool small_gemv = oi_ablation_env("OI_SMALL_BATCH_GEMV") != nullptr;
            const bool screen = (ctx->cosine_mode == OI_COSINE_SCREEN || ctx->cosine_mode == OI_COSINE_SCREEN_COPY ||
                                 ctx->cosine_mode == OI_COSINE_SCREEN_STREAM) && (B > 8 || (want_copy && !small_gemv)) &&
                                oi_cosine_screen_supported(idx->dim) && idx->screen_ok && shape16 && !cos_v1;
            if (!screen) {
                ctx->last_screen_gate = nullptr; // (profile "screen_gate": -1 = this search was not screened)
                return exact_pipeline(nullptr, nullptr);
            }

            // ---- bf16 screen -> margin selects -> exact rescoring -> sorted selection; then the gated exact pipeline
            // (cosine_prefilter.hip).  Its pool keeps up to 4096 keys per query between chunks.
            const uint32_t segs = (uint32_t)ctx->num_cus;
            // state, zeroed with one memset: carry_cnt[B] tau[B] rs_cnt[B] eps2[B] gate[4] seg_cnt[B][segs] spec_tau[B] spec_max[B]
            const size_t words = (size_t)B * (6 + segs) + 4;
            DevBuf &pk = ctx->buf("pool_cos"), &rk = ctx->buf("screen_rescored"), &qb = ctx->buf("screen_q_bf16");
            OI_REQUIRE(words <= screen_words, "search: screen state does not fit its reservation");
            OI_REQUIRE(pk.cap >= sizeof(uint64_t) * (size_t)B * pf_stride, "search: the shared cosine pool is too small for the screen's view");
            const uint32_t rs_cap = pf_carry + OI_LONG_ROWS_MAX; // the survivors and the index's long rows (two-class margin)
            OI_CHECK(rk.ensure(sizeof(uint64_t) * (size_t)B * rs_cap));
            const uint32_t n_padded = (B + 31u) & ~31u;
            OI_CHECK(qb.ensure(sizeof(uint16_t) * (size_t)(n_padded + 64) * idx->dim));
            uint32_t *w = screen_state; // zeroed with the pool state (prepare_pools)
            uint32_t *pf_cnt = w, *pf_tau = w + B, *rs_cnt = w + 2 * (size_t)B;
            float *eps2 = reinterpret_cast<float *>(w + 3 * (size_t)B);
            uint32_t *gate = w + 4 * (size_t)B, *pf_seg = gate + 4;
            uint32_t *spec_tau = pf_seg + (size_t)B * segs, *spec_max = spec_tau + B;
            PoolView PF{pk.as<uint64_t>(), pf_cnt, pf_seg, pf_tau, pf_stride, pf_carry, 0, 0, segs, P.cos.overflow};
            PoolView RS{rk.as<uint64_t>(), rs_cnt, pf_seg, nullptr, rs_cap, rs_cap, 0, 0, segs, P.cos.overflow};
            OI_CHECK(oi_launch_screen_stage(ctx, d_qv, B, idx->dim, idx->max_row_norm.as<uint32_t>(), qb.as<uint16_t>(),
                                            eps2, gate));
            const uint64_t pf_max_chunk = pf_stride - pf_carry - pf_slack;
            SelectExtra mx;
            mx.eps2 = eps2;
            mx.margin_gate = gate;
            if (idx->n_long) { mx.skip_bitmap = idx->long_bitmap.as<uint32_t>(); mx.skip_base = idx->doc_id_base; }
            if (want_copy && !idx->screen_copy.p) { // made once, on the first search that asks for it (n x d x 2 B of HBM)
                OI_CHECK(idx->screen_copy.ensure(sizeof(uint16_t) * (size_t)n * idx->dim + 64));
                OI_CHECK(oi_launch_make_screen_copy(ctx, idx->rows, n, idx->dim, idx->screen_copy.as<uint16_t>()));
            }
            // Speculative thresholds (decided above: spec_on, screen_growth): the next chunk is screened against the larger of
            // the proven threshold and a prediction from the rows seen so far, checked at the end (a failed check opens the gate).
            const bool spec = spec_on;
            bool spec_next = false, spec_any = false;
            uint64_t chunk = screen_first;
            uint64_t r = 0;
            while (r < n) {
                if (chunk > pf_max_chunk) chunk = pf_max_chunk;
                const uint64_t e = oi_chunk_end(r, chunk, n, pf_max_chunk, r == 0 ? 0 : chunk * screen_growth); // (r == 0: the threshold-less first chunk is never stretched)
                // the same products, the same bound: only where bf16(x) comes from differs (converted on the fly from the
                // f32 rows, 4 d bytes per row -- or read from the copy, 2 d bytes per row)
                if (late_pending && e == n) OI_HIP_CHECK(hipEventRecord(ctx->ev_fork, st)); // (before the last chunk's launch)
                uint32_t *const proven_tau = PF.tau_keys;
                if (spec_next) PF.tau_keys = spec_tau; // (this launch only: the selects keep the proven thresholds)
                const int rc_screen = want_copy ? oi_launch_cosine_screen_copy_chunk(ctx, idx->screen_copy.as<uint16_t>(), r, e, idx->dim, qb.as<uint16_t>(), B, idx->doc_id_base, PF)
                                                : oi_launch_cosine_screen_chunk(ctx, idx->rows, r, e, idx->dim, qb.as<uint16_t>(), B, idx->doc_id_base, PF);
                PF.tau_keys = proven_tau;
                OI_CHECK(rc_screen);
                if (late_pending && e == n) { late_pending = false; OI_CHECK(fork_bm25()); } // ... enqueued after it: the screen's workgroups get their CUs first
                OI_CHECK(oi_launch_select(ctx, PF, B, depth, /*compact=*/true, nullptr, nullptr, nullptr, depth, &mx));
                r = e;
                chunk *= screen_growth;
                spec_next = false;
                if (spec && r < n) {
                    // expected rank of the final k'-th among the r rows seen: depth r / n; three times that plus twelve
                    const uint64_t rank = (3ull * depth * r + n - 1) / n + 12;
                    if (2 * rank <= depth) {
                        OI_CHECK(oi_launch_spec_threshold(ctx, PF, B, (uint32_t)rank, eps2, spec_tau, spec_max));
                        spec_next = spec_any = true;
                    }
                }
            }
            if (spec_any) ++ctx->spec_searches;
            // (the check of the speculative thresholds against the proven final ones rides in the rescoring launch; the gated exact
            // pipeline is enqueued after it)
            OI_CHECK(oi_launch_rescore(ctx, idx->rows, n, idx->dim, idx->doc_id_base, d_qv, B, PF, RS,
                                       idx->n_long ? idx->long_list.as<uint32_t>() : nullptr, idx->n_long,
                                       spec_any ? spec_max : nullptr, pf_tau, gate, ctx->spec_fail_host));
            RS.n_segs = 0;
            OI_CHECK(oi_launch_select(ctx, RS, B, depth, false, cos_s, cos_d, cos_c, depth));
            ctx->run_gate = gate;
            ctx->last_screen_gate = gate;
            const int rc = exact_pipeline(gate, pf_tau);
            ctx->run_gate = nullptr;
            return rc;
        }
        return OI_OK;
    };
    {
        const int rc = cosine_leg();
        if (rc != OI_OK) { // nothing of this call stays in flight behind an error return
            if (overlap) (void)hipStreamSynchronize(ctx->side_stream);
            return rc;
        }
    }
    if (late_pending) { // (the screen was not taken after all)
        OI_HIP_CHECK(hipEventRecord(ctx->ev_fork, st));
        OI_CHECK(fork_bm25());
    }
    if (overlap) OI_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_join, 0));
    else if (bm_s) OI_CHECK(bm25_leg());
    return OI_OK;
}

struct QueryStage {
    const float *qv;
    const uint32_t *qt, *qo;
};

int stage_queries(oi_index *idx, const float *qv, const uint32_t *qt, const uint32_t *qo, uint32_t B, int location,
                  QueryStage *out) {
    oi_ctx *ctx = idx->ctx;
    if (location == OI_DEVICE) { *out = QueryStage{qv, qt, qo}; return OI_OK; }
    hipStream_t st = ctx->stream;
    // one page-locked staging buffer, one DMA: [vectors | terms | offsets], each part 16-byte aligned
    const uint32_t nt = qo[B];
    const size_t vb = sizeof(float) * (size_t)B * idx->dim, tb = sizeof(uint32_t) * (size_t)(nt ? nt : 1), ob = sizeof(uint32_t) * ((size_t)B + 1);
    const size_t off_t = (vb + 15) & ~(size_t)15, off_o = off_t + ((tb + 15) & ~(size_t)15), total = off_o + ob;
    DevBuf &a = ctx->buf("q_stage");
    OI_CHECK(a.ensure(total));
    OI_CHECK(ctx->pin_in.ensure(total));
    uint8_t *h = ctx->pin_in.as<uint8_t>();
    memcpy(h, qv, vb);
    if (nt) memcpy(h + off_t, qt, sizeof(uint32_t) * nt);
    memcpy(h + off_o, qo, ob);
    OI_HIP_CHECK(hipMemcpyAsync(a.p, h, total, hipMemcpyHostToDevice, st));
    uint8_t *d = a.as<uint8_t>();
    *out = QueryStage{reinterpret_cast<const float *>(d), reinterpret_cast<const uint32_t *>(d + off_t),
                      reinterpret_cast<const uint32_t *>(d + off_o)};
    return OI_OK;
}

// The fused result block [scores K | docs K | counts B] (contiguous in HBM) to the caller's three host arrays: one DMA into
// the context's page-locked buffer, the stream synchronised, three host copies.
int results_to_host(oi_ctx *ctx, const float *d_block, size_t K, uint32_t B, float *scores_out, uint32_t *docs_out,
                    uint32_t *counts_out) {
    const size_t bytes = (2 * K + B) * 4;
    OI_CHECK(ctx->pin_out.ensure(bytes));
    OI_HIP_CHECK(hipMemcpyAsync(ctx->pin_out.p, d_block, bytes, hipMemcpyDeviceToHost, ctx->stream));
    OI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const uint8_t *h = ctx->pin_out.as<uint8_t>();
    memcpy(scores_out, h, K * 4);
    memcpy(docs_out, h + K * 4, K * 4);
    memcpy(counts_out, h + 2 * K * 4, (size_t)B * 4);
    return OI_OK;
}

int check_search_args(oi_index *idx, const float *qv, const uint32_t *qt, const uint32_t *qo, uint32_t B,
                      uint32_t depth) {
    if (!idx) { oi_set_error("null index"); return OI_ERR_INVALID_ARG; }
    OI_REQUIRE(qv && qo && (qt || true), "search: null query buffer");
    OI_REQUIRE(B >= 1 && B <= 4096, "search: n_queries=%u outside [1,4096]", B);
    OI_REQUIRE(depth >= 1 && depth <= OI_MAX_DEPTH, "search: depth=%u outside [1,%u]", depth, OI_MAX_DEPTH);
    return OI_OK;
}

} // namespace

extern "C" int oi_search_lists(oi_index *idx, const float *qv, const uint32_t *qt, const uint32_t *qo, uint32_t B,
                               uint32_t depth, int location, float *cos_s, uint32_t *cos_d, uint32_t *cos_c,
                               float *bm_s, uint32_t *bm_d, uint32_t *bm_c) {
    OI_CHECK(check_search_args(idx, qv, qt, qo, B, depth));
    OI_REQUIRE(cos_s && cos_d && cos_c && bm_s && bm_d && bm_c, "search_lists: null output buffer");
    oi_ctx *ctx = idx->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    QueryStage q;
    OI_CHECK(stage_queries(idx, qv, qt, qo, B, location, &q));
    if (location == OI_DEVICE)
        return search_lists_device(idx, q.qv, q.qt, q.qo, B, depth, cos_s, cos_d, cos_c, bm_s, bm_d, bm_c);
    hipStream_t st = ctx->stream;
    const size_t L = (size_t)B * depth;
    DevBuf &o = ctx->buf("lists_out");
    OI_CHECK(o.ensure(L * 16 + (size_t)B * 8 + 64));
    float *d_cs = o.as<float>();
    uint32_t *d_cd = reinterpret_cast<uint32_t *>(d_cs + L);
    float *d_bs = reinterpret_cast<float *>(d_cd + L);
    uint32_t *d_bd = reinterpret_cast<uint32_t *>(d_bs + L);
    uint32_t *d_cc = d_bd + L, *d_bc = d_cc + B;
    OI_HIP_CHECK(hipMemsetAsync(o.p, 0, L * 16 + (size_t)B * 8, st));
    OI_CHECK(search_lists_device(idx, q.qv, q.qt, q.qo, B, depth, d_cs, d_cd, d_cc, d_bs, d_bd, d_bc));
    OI_HIP_CHECK(hipMemcpyAsync(cos_s, d_cs, L * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(cos_d, d_cd, L * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(bm_s, d_bs, L * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(bm_d, d_bd, L * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(cos_c, d_cc, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(bm_c, d_bc, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    return check_overflow_locked(ctx);
}

extern "C" int oi_rrf_fuse(oi_ctx *ctx, const uint32_t *docs_a, const uint32_t *counts_a, const uint32_t *docs_b,
                           const uint32_t *counts_b, uint32_t B, uint32_t depth, uint32_t k, int location,
                           float *scores_out, uint32_t *docs_out, uint32_t *counts_out) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    OI_REQUIRE(docs_a && counts_a && docs_b && counts_b && scores_out && docs_out && counts_out, "rrf: null buffer");
    if (B == 0) return OI_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    if (location == OI_DEVICE)
        return oi_launch_rrf(ctx, docs_a, counts_a, docs_b, counts_b, B, depth, k, scores_out, docs_out, counts_out);
    hipStream_t st = ctx->stream;
    const size_t L = (size_t)B * depth, K = (size_t)B * k;
    DevBuf &w = ctx->buf("rrf_io");
    OI_CHECK(w.ensure((2 * L + 2 * (size_t)B + 2 * K + B) * 4 + 64));
    uint32_t *d_a = w.as<uint32_t>(), *d_b = d_a + L, *d_ca = d_b + L, *d_cb = d_ca + B;
    float *d_so = reinterpret_cast<float *>(d_cb + B);
    uint32_t *d_do = reinterpret_cast<uint32_t *>(d_so + K), *d_co = d_do + K;
    OI_HIP_CHECK(hipMemcpyAsync(d_a, docs_a, L * 4, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemcpyAsync(d_b, docs_b, L * 4, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemcpyAsync(d_ca, counts_a, (size_t)B * 4, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemcpyAsync(d_cb, counts_b, (size_t)B * 4, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemsetAsync(d_so, 0, (2 * K + B) * 4, st));
    OI_CHECK(oi_launch_rrf(ctx, d_a, d_ca, d_b, d_cb, B, depth, k, d_so, d_do, d_co));
    OI_HIP_CHECK(hipMemcpyAsync(scores_out, d_so, K * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(docs_out, d_do, K * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(counts_out, d_co, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    return OI_OK;
}

extern "C" int oi_merge_lists(oi_ctx *ctx, const float *scores, const uint32_t *docs, const uint32_t *counts,
                              uint32_t n_shards, uint32_t B, uint32_t depth, int location, float *scores_out,
                              uint32_t *docs_out, uint32_t *counts_out) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    OI_REQUIRE(scores && docs && counts && scores_out && docs_out && counts_out, "merge: null buffer");
    OI_REQUIRE(n_shards >= 1 && n_shards <= 1024, "merge: n_shards=%u outside [1,1024]", n_shards);
    OI_REQUIRE(depth >= 1 && depth <= OI_MAX_DEPTH, "merge: depth=%u outside [1,%u]", depth, OI_MAX_DEPTH);
    if (B == 0) return OI_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t carry_cap = OI_MAX_DEPTH;
    const uint64_t mstride = (uint64_t)carry_cap + (uint64_t)n_shards * depth;
    DevBuf &flag = ctx->buf("state_flag");
    if (!flag.p) {
        OI_CHECK(flag.ensure(16));
        OI_HIP_CHECK(hipMemsetAsync(flag.p, 0, 16, st));
    }
    DevBuf &pk = ctx->buf("merge_pool"), &pc = ctx->buf("merge_counts");
    OI_CHECK(pk.ensure(sizeof(uint64_t) * (size_t)B * mstride));
    OI_CHECK(pc.ensure(sizeof(uint32_t) * (size_t)B * (1 + n_shards)));
    PoolView pool{pk.as<uint64_t>(), pc.as<uint32_t>(), pc.as<uint32_t>() + B, nullptr, mstride, carry_cap,
                  depth, n_shards, n_shards, flag.as<uint32_t>()};
    const size_t Lin = (size_t)n_shards * B * depth, Cin = (size_t)n_shards * B, L = (size_t)B * depth;
    if (location == OI_DEVICE) {
        OI_CHECK(oi_launch_lists_to_pool(ctx, scores, docs, counts, (uint64_t)B * depth, B, n_shards, B, depth, pool));
        return oi_launch_select(ctx, pool, B, depth, false, scores_out, docs_out, counts_out, depth);
    }
    DevBuf &w = ctx->buf("merge_io");
    OI_CHECK(w.ensure((2 * Lin + Cin + 2 * L + B) * 4 + 64));
    float *d_s = w.as<float>();
    uint32_t *d_d = reinterpret_cast<uint32_t *>(d_s + Lin), *d_c = d_d + Lin;
    float *d_so = reinterpret_cast<float *>(d_c + Cin);
    uint32_t *d_do = reinterpret_cast<uint32_t *>(d_so + L), *d_co = d_do + L;
    OI_HIP_CHECK(hipMemcpyAsync(d_s, scores, Lin * 4, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemcpyAsync(d_d, docs, Lin * 4, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemcpyAsync(d_c, counts, Cin * 4, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemsetAsync(d_so, 0, (2 * L + B) * 4, st));
    OI_CHECK(oi_launch_lists_to_pool(ctx, d_s, d_d, d_c, (uint64_t)B * depth, B, n_shards, B, depth, pool));
    OI_CHECK(oi_launch_select(ctx, pool, B, depth, false, d_so, d_do, d_co, depth));
    OI_HIP_CHECK(hipMemcpyAsync(scores_out, d_so, L * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(docs_out, d_do, L * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(counts_out, d_co, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    return OI_OK;
}

extern "C" int oi_search(oi_index *idx, const float *qv, const uint32_t *qt, const uint32_t *qo, uint32_t B,
                         uint32_t depth, uint32_t k, int location, float *scores_out, uint32_t *docs_out,
                         uint32_t *counts_out) {
    OI_CHECK(check_search_args(idx, qv, qt, qo, B, depth));
    OI_REQUIRE(k >= 1 && k <= OI_MAX_DEPTH, "search: k=%u outside [1,%u]", k, OI_MAX_DEPTH);
    OI_REQUIRE(scores_out && docs_out && counts_out, "search: null output buffer");
    oi_ctx *ctx = idx->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    QueryStage q;
    OI_CHECK(stage_queries(idx, qv, qt, qo, B, location, &q));
    const size_t L = (size_t)B * depth, K = (size_t)B * k;
    DevBuf &o = ctx->buf("search_lists");
    OI_CHECK(o.ensure(L * 16 + (size_t)B * 8 + 64));
    float *d_cs = o.as<float>();
    uint32_t *d_cd = reinterpret_cast<uint32_t *>(d_cs + L);
    float *d_bs = reinterpret_cast<float *>(d_cd + L);
    uint32_t *d_bd = reinterpret_cast<uint32_t *>(d_bs + L);
    uint32_t *d_cc = d_bd + L, *d_bc = d_cc + B;
    if (location == OI_DEVICE) {
        const uint64_t key[10] = {idx->uid, 3, (uint64_t)(uintptr_t)qv, (uint64_t)(uintptr_t)qt, (uint64_t)(uintptr_t)qo,
                                  ((uint64_t)B << 32) | depth, (uint64_t)(uintptr_t)scores_out,
                                  ((uint64_t)ctx->cosine_mode << 16) | ((uint64_t)idx->bm25_mode << 8) | (ctx->overlap_legs ? 1u : 0u),
                                  ((uint64_t)idx->max_query_terms << 32) | k, (uint64_t)(uintptr_t)docs_out ^ ((uint64_t)(uintptr_t)counts_out << 1)};
        return run_captured(ctx, key, [&]() -> int {
            OI_HIP_CHECK(hipMemsetAsync(d_cc, 0, (size_t)B * 8, ctx->stream));
            OI_CHECK(search_lists_device(idx, q.qv, q.qt, q.qo, B, depth, d_cs, d_cd, d_cc, d_bs, d_bd, d_bc));
            return oi_launch_rrf(ctx, d_cd, d_cc, d_bd, d_bc, B, depth, k, scores_out, docs_out, counts_out);
        });
    }
    OI_HIP_CHECK(hipMemsetAsync(d_cc, 0, (size_t)B * 8, st));
    OI_CHECK(search_lists_device(idx, q.qv, q.qt, q.qo, B, depth, d_cs, d_cd, d_cc, d_bs, d_bd, d_bc));
    DevBuf &f = ctx->buf("search_out");
    OI_CHECK(f.ensure((2 * K + B) * 4 + 64));
    float *d_so = f.as<float>();
    uint32_t *d_do = reinterpret_cast<uint32_t *>(d_so + K), *d_co = d_do + K;
    OI_HIP_CHECK(hipMemsetAsync(f.p, 0, (2 * K + B) * 4, st));
    OI_CHECK(oi_launch_rrf(ctx, d_cd, d_cc, d_bd, d_bc, B, depth, k, d_so, d_do, d_co));
    OI_CHECK(results_to_host(ctx, d_so, K, B, scores_out, docs_out, counts_out));
    return check_overflow_locked(ctx);
}

// ---------------------------------------------------------------- diagnostics of the bf16 screen
extern "C" int oi_screen_probe(oi_index *idx, const float *query_vecs, uint32_t B, uint64_t row_begin, uint32_t n_rows,
                               float *screen_scores_out, float *eps_out) {
    if (!idx || !query_vecs) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    OI_REQUIRE(B >= 1 && B <= 4096, "screen probe: n_queries=%u outside [1,4096]", B);
    OI_REQUIRE(idx->rows, "screen probe: the index holds no f32 rows");
    OI_REQUIRE(idx->dim % 16 == 0, "screen probe: dim %u is not a multiple of 16", idx->dim);
    OI_REQUIRE(row_begin <= idx->n_docs && n_rows <= idx->n_docs - row_begin && n_rows <= (1u << 20),
               "screen probe: rows [%llu, +%u) outside the index (or more than 2^20)", (unsigned long long)row_begin, n_rows);
    oi_ctx *ctx = idx->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t n_padded = (B + 31u) & ~31u;
    DevBuf &qf = ctx->buf("probe_q"), &qb = ctx->buf("probe_q_bf16"), &ws = ctx->buf("probe_state"), &out = ctx->buf("probe_out");
    OI_CHECK(qf.ensure(sizeof(float) * (size_t)B * idx->dim));
    OI_CHECK(qb.ensure(sizeof(uint16_t) * (size_t)(n_padded + 64) * idx->dim));
    OI_CHECK(ws.ensure(sizeof(float) * (size_t)B + 16));
    OI_CHECK(out.ensure(sizeof(float) * ((size_t)B * n_rows + 1)));
    OI_HIP_CHECK(hipMemcpyAsync(qf.p, query_vecs, sizeof(float) * (size_t)B * idx->dim, hipMemcpyHostToDevice, st));
    OI_HIP_CHECK(hipMemsetAsync(ws.p, 0, sizeof(float) * (size_t)B + 16, st));
    float *eps2 = ws.as<float>();
    uint32_t *gate = reinterpret_cast<uint32_t *>(eps2 + B);
    OI_CHECK(oi_launch_screen_stage(ctx, qf.as<float>(), B, idx->dim, idx->max_row_norm.as<uint32_t>(), qb.as<uint16_t>(), eps2, gate));
    if (screen_scores_out && n_rows) {
        OI_CHECK(oi_launch_screen_probe(ctx, idx->rows, row_begin, n_rows, idx->dim, qb.as<uint16_t>(), B, out.as<float>()));
        OI_HIP_CHECK(hipMemcpyAsync(screen_scores_out, out.p, sizeof(float) * (size_t)B * n_rows, hipMemcpyDeviceToHost, st));
    }
    if (eps_out) OI_HIP_CHECK(hipMemcpyAsync(eps_out, eps2, sizeof(float) * (size_t)B, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    if (eps_out)
        for (uint32_t i = 0; i < B; ++i) eps_out[i] *= 0.5f; // the kernel stores the margin 2 eps
    return OI_OK;
}

// ---------------------------------------------------------------- packed multi-GPU exchange
extern "C" int oi_search_lists_packed(oi_index *idx, const float *qv, const uint32_t *qt, const uint32_t *qo,
                                      uint32_t B, uint32_t depth, int location, uint32_t *packed_out) {
    OI_CHECK(check_search_args(idx, qv, qt, qo, B, depth));
    OI_REQUIRE(packed_out, "search_lists_packed: null output buffer");
    oi_ctx *ctx = idx->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    QueryStage q;
    OI_CHECK(stage_queries(idx, qv, qt, qo, B, location, &q));
    const size_t L = (size_t)B * depth, W = (size_t)OI_PACKED_WORDS(B, depth);
    uint32_t *d_out = packed_out;
    if (location != OI_DEVICE) {
        DevBuf &o = ctx->buf("packed_out");
        OI_CHECK(o.ensure(W * 4));
        d_out = o.as<uint32_t>();
    }
    float *sc = reinterpret_cast<float *>(d_out);
    uint32_t *dc = d_out + 2 * L, *cn = d_out + 4 * L;
    auto body = [&]() -> int {
        OI_HIP_CHECK(hipMemsetAsync(d_out + 4 * L, 0, (size_t)B * 8, ctx->stream)); // counts
        return search_lists_device(idx, q.qv, q.qt, q.qo, B, depth, sc, dc, cn, sc + L, dc + L, cn + B);
    };
    if (location == OI_DEVICE) {
        const uint64_t key[10] = {idx->uid, 1, (uint64_t)(uintptr_t)qv, (uint64_t)(uintptr_t)qt, (uint64_t)(uintptr_t)qo,
                                  ((uint64_t)B << 32) | depth, (uint64_t)(uintptr_t)packed_out,
                                  ((uint64_t)ctx->cosine_mode << 16) | ((uint64_t)idx->bm25_mode << 8) | (ctx->overlap_legs ? 1u : 0u),
                                  idx->max_query_terms, 0};
        return run_captured(ctx, key, body);
    }
    OI_CHECK(body());
    OI_HIP_CHECK(hipMemcpyAsync(packed_out, d_out, W * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    return check_overflow_locked(ctx);
}

// All shards' packed lists (device) -> global top-depth per list -> RRF top-k into device outputs.  ctx->mu held.
static int fuse_packed_device(oi_ctx *ctx, const uint32_t *d_in, uint32_t n_shards, uint32_t B, uint32_t depth, uint32_t k,
                              float *o_s, uint32_t *o_d, uint32_t *o_c) {
    hipStream_t st = ctx->stream;
    const size_t L = (size_t)B * depth, W = (size_t)OI_PACKED_WORDS(B, depth);
    DevBuf &flag = ctx->buf("state_flag");
    if (!flag.p) {
        OI_CHECK(flag.ensure(16));
        OI_HIP_CHECK(hipMemsetAsync(flag.p, 0, 16, st));
    }
    const uint32_t carry_cap = OI_MAX_DEPTH;
    const uint64_t mstride = (uint64_t)carry_cap + (uint64_t)n_shards * depth;
    DevBuf &pk = ctx->buf("merge_pool"), &pc = ctx->buf("merge_counts"), &ml = ctx->buf("merged_lists");
    OI_CHECK(pk.ensure(sizeof(uint64_t) * (size_t)2 * B * mstride));
    OI_CHECK(pc.ensure(sizeof(uint32_t) * (size_t)2 * B * (1 + n_shards)));
    OI_CHECK(ml.ensure((4 * L + 2 * (size_t)B) * 4));
    PoolView pool{pk.as<uint64_t>(), pc.as<uint32_t>(), pc.as<uint32_t>() + 2 * (size_t)B, nullptr, mstride,
                  carry_cap, depth, n_shards, n_shards, flag.as<uint32_t>()};
    float *m_s = ml.as<float>();                       // [2][B][depth]
    uint32_t *m_d = ml.as<uint32_t>() + 2 * L;          // [2][B][depth]
    uint32_t *m_c = m_d + 2 * L;                        // [2][B]
    // scores[2][B][depth] is [2B][depth]: both lists of every query are merged by ONE pair of launches
    // (2B virtual queries; list 0 = cosine, 1 = BM25)
    OI_CHECK(oi_launch_lists_to_pool(ctx, reinterpret_cast<const float *>(d_in), d_in + 2 * L, d_in + 4 * L, W, W,
                                     n_shards, 2 * B, depth, pool));
    OI_CHECK(oi_launch_select(ctx, pool, 2 * B, depth, false, m_s, m_d, m_c, depth));
    return oi_launch_rrf(ctx, m_d, m_c, m_d + L, m_c + B, B, depth, k, o_s, o_d, o_c);
}

extern "C" int oi_fuse_packed(oi_ctx *ctx, const uint32_t *packed_all, uint32_t n_shards, uint32_t B, uint32_t depth,
                              uint32_t k, int location, float *scores_out, uint32_t *docs_out, uint32_t *counts_out) {
    if (!ctx) { oi_set_error("null ctx"); return OI_ERR_INVALID_ARG; }
    OI_REQUIRE(packed_all && scores_out && docs_out && counts_out, "fuse_packed: null buffer");
    OI_REQUIRE(n_shards >= 1 && n_shards <= 1024, "fuse_packed: n_shards=%u outside [1,1024]", n_shards);
    OI_REQUIRE(depth >= 1 && depth <= OI_MAX_DEPTH && k >= 1 && k <= OI_MAX_DEPTH, "fuse_packed: depth/k out of range");
    if (B == 0) return OI_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    if (location == OI_DEVICE) {
        const uint64_t key[10] = {0xF05Eull << 32, 2, (uint64_t)(uintptr_t)packed_all, ((uint64_t)n_shards << 32) | B,
                                  ((uint64_t)depth << 32) | k, (uint64_t)(uintptr_t)scores_out, (uint64_t)(uintptr_t)docs_out,
                                  (uint64_t)(uintptr_t)counts_out, 0, 0};
        return run_captured(ctx, key, [&]() -> int {
            return fuse_packed_device(ctx, packed_all, n_shards, B, depth, k, scores_out, docs_out, counts_out);
        });
    }
    hipStream_t st = ctx->stream;
    const size_t W = (size_t)OI_PACKED_WORDS(B, depth), K = (size_t)B * k;
    DevBuf &in = ctx->buf("packed_in"), &fo = ctx->buf("fuse_out");
    OI_CHECK(in.ensure(W * 4 * n_shards));
    OI_CHECK(fo.ensure((2 * K + B) * 4 + 64));
    OI_HIP_CHECK(hipMemcpyAsync(in.p, packed_all, W * 4 * n_shards, hipMemcpyHostToDevice, st));
    float *o_s = fo.as<float>();
    uint32_t *o_d = reinterpret_cast<uint32_t *>(o_s + K), *o_c = o_d + K;
    OI_HIP_CHECK(hipMemsetAsync(o_s, 0, (2 * K + B) * 4, st));
    OI_CHECK(fuse_packed_device(ctx, in.as<uint32_t>(), n_shards, B, depth, k, o_s, o_d, o_c));
    OI_HIP_CHECK(hipMemcpyAsync(scores_out, o_s, K * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(docs_out, o_d, K * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(counts_out, o_c, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    return OI_OK;
}

// ---------------------------------------------------------------- the sharded query with RCCL inside (comm.hip)
extern "C" int oi_comm_unique_id(uint8_t *id_out) {
    if (!id_out) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    return oi_rccl_unique_id(id_out);
}

extern "C" int oi_comm_create(oi_ctx *ctx, const uint8_t *id, uint32_t rank, uint32_t world, oi_comm **out) {
    if (!ctx || !id || !out) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    *out = nullptr;
    OI_REQUIRE(world >= 1 && world <= 1024 && rank < world, "comm: rank %u of %u", rank, world);
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device)); // RCCL binds the communicator to the current device
    void *nc = nullptr;
    OI_CHECK(oi_rccl_init(&nc, world, id, rank));
    oi_comm *c = new oi_comm();
    c->ctx = ctx;
    ctx->refs.fetch_add(1);
    c->nccl = nc;
    c->rank = rank;
    c->world = world;
    *out = c;
    return OI_OK;
}

extern "C" void oi_comm_destroy(oi_comm *comm) {
    if (!comm) return;
    oi_ctx *ctx = comm->ctx;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        oi_rccl_destroy(comm->nccl);
    }
    delete comm;
    ctx_release(ctx);
}

extern "C" int oi_index_finalize_sharded(oi_index *idx, oi_comm *comm) {
    if (!idx || !comm) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    if (idx->is_view) { oi_set_error("index view: read-only (set the data on the index it was taken from)"); return OI_ERR_STATE; }
    oi_ctx *ctx = idx->ctx;
    OI_REQUIRE(comm->ctx == ctx, "finalize_sharded: the communicator belongs to another context");
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!idx->forward_set) { oi_set_error("index: set_forward has not been called"); return OI_ERR_STATE; }
    if (idx->finalized) { oi_set_error("index: already finalized"); return OI_ERR_STATE; }
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // BM25 needs the statistics of the WHOLE collection: sum over the shards of (n_docs, tokens) and of df
    DevBuf &ws = ctx->buf("sharded_stats");
    OI_CHECK(ws.ensure(16 + sizeof(uint32_t) * (size_t)idx->vocab));
    uint64_t mine[2] = {idx->n_docs, idx->total_tokens};
    OI_HIP_CHECK(hipMemcpyAsync(ws.p, mine, 16, hipMemcpyHostToDevice, st));
    uint32_t *d_df = reinterpret_cast<uint32_t *>(ws.as<uint8_t>() + 16);
    OI_HIP_CHECK(hipMemcpyAsync(d_df, idx->df_local.p, sizeof(uint32_t) * (size_t)idx->vocab, hipMemcpyDeviceToDevice, st));
    OI_CHECK(oi_rccl_all_reduce_sum(comm->nccl, ws.p, 2, /*u64=*/true, st));
    OI_CHECK(oi_rccl_all_reduce_sum(comm->nccl, d_df, idx->vocab, /*u64=*/false, st));
    uint64_t glob[2] = {0, 0};
    std::vector<uint32_t> gdf(idx->vocab);
    OI_HIP_CHECK(hipMemcpyAsync(glob, ws.p, 16, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipMemcpyAsync(gdf.data(), d_df, sizeof(uint32_t) * (size_t)idx->vocab, hipMemcpyDeviceToHost, st));
    OI_HIP_CHECK(hipStreamSynchronize(st));
    OI_CHECK(oi_bm25_finalize(idx, glob[0], glob[1], gdf.data()));
    return apply_screen_copy_policy(idx);
}

extern "C" int oi_search_sharded(oi_index *idx, oi_comm *comm, const float *qv, const uint32_t *qt, const uint32_t *qo,
                                 uint32_t B, uint32_t depth, uint32_t k, int location, float *scores_out,
                                 uint32_t *docs_out, uint32_t *counts_out) {
    OI_CHECK(check_search_args(idx, qv, qt, qo, B, depth));
    if (!comm) { oi_set_error("null communicator"); return OI_ERR_INVALID_ARG; }
    OI_REQUIRE(k >= 1 && k <= OI_MAX_DEPTH, "search: k=%u outside [1,%u]", k, OI_MAX_DEPTH);
    OI_REQUIRE(scores_out && docs_out && counts_out, "search: null output buffer");
    oi_ctx *ctx = idx->ctx;
    OI_REQUIRE(comm->ctx == ctx, "search_sharded: the communicator belongs to another context");
    std::lock_guard<std::mutex> g(ctx->mu);
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    QueryStage q;
    OI_CHECK(stage_queries(idx, qv, qt, qo, B, location, &q));
    const size_t L = (size_t)B * depth, W = (size_t)OI_PACKED_WORDS(B, depth), K = (size_t)B * k;
    DevBuf &pk = ctx->buf("sharded_packed"), &fl = ctx->buf("sharded_flat");
    OI_CHECK(pk.ensure(W * 4));
    OI_CHECK(fl.ensure(W * 4 * comm->world));
    uint32_t *d_p = pk.as<uint32_t>();
    OI_HIP_CHECK(hipMemsetAsync(d_p + 4 * L, 0, (size_t)B * 8, st)); // counts
    float *sc = reinterpret_cast<float *>(d_p);
    uint32_t *dc = d_p + 2 * L, *cn = d_p + 4 * L;
    OI_CHECK(search_lists_device(idx, q.qv, q.qt, q.qo, B, depth, sc, dc, cn, sc + L, dc + L, cn + B));
    // the ONE exchange per batch: every rank's packed lists to every rank (1 MB per rank at B=64, k'=1000)
    OI_CHECK(oi_rccl_all_gather_u32(comm->nccl, d_p, fl.as<uint32_t>(), W, st));
    if (location == OI_DEVICE)
        return fuse_packed_device(ctx, fl.as<uint32_t>(), comm->world, B, depth, k, scores_out, docs_out, counts_out);
    DevBuf &fo = ctx->buf("fuse_out");
    OI_CHECK(fo.ensure((2 * K + B) * 4 + 64));
    float *o_s = fo.as<float>();
    uint32_t *o_d = reinterpret_cast<uint32_t *>(o_s + K), *o_c = o_d + K;
    OI_HIP_CHECK(hipMemsetAsync(o_s, 0, (2 * K + B) * 4, st));
    OI_CHECK(fuse_packed_device(ctx, fl.as<uint32_t>(), comm->world, B, depth, k, o_s, o_d, o_c));
    OI_CHECK(results_to_host(ctx, o_s, K, B, scores_out, docs_out, counts_out));
    return check_overflow_locked(ctx);
}
