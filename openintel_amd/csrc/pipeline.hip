// pipeline.hip -- the pipelined (optionally row-sharded) hybrid query behind the C ABI: oi_pipeline_*.
//
// Builder-defined like the rest of the retrieval path.  The reference's port is synchronous and single-process
// (/root/reference/src/domain/ports/post_analyzer.rs:7-11: `Send + Sync`, borrowed in / owned out; composition root
// src/main.rs:17-39), so a Rust host has neither streams nor a collective library: until round 4 the overlap that gives the
// sharded step its period lived in Python only (openintel_amd/sharded.py::ShardedPipeline over torch.distributed) and a host
// on the C ABI got one synchronous lane (oi_search_sharded).  This file is that pipeline in the library, built from the
// library's own public calls:
//   * `lanes` searching lanes, each a context of its own (oi_create_like) on a stream of its own with a VIEW of the index
//     (oi_index_view: no HBM for the index, own workspaces): batch n is scored on lane n % lanes, so the next batch's
//     screen streams the corpus while the selects, the rescoring and the fusion of the previous one drain;
//   * one fusing context on a stream of its own: the ONE all-gather of a batch's packed lists (RCCL, when a communicator
//     is given) and oi_fuse_packed (merge to the global top-depth per list, THEN RRF) run there, beside the lanes;
//   * a ring of slots (packed lists, exchange buffer, staging of host queries / results) ordered with events only: no host
//     synchronisation between submit calls.
// Per batch nothing changes: the same kernels in the same order on the same data as oi_search / oi_search_sharded --
// the results are bit-identical (tests/test_gpu_pipeline.py).
// Collectives are issued from the submitting host thread in submission order on ONE stream: every rank must submit the
// same batches in the same order (as with oi_search_sharded).
#include <algorithm>
#include <chrono>
#include <vector>

#include "oi_internal.h"

// ---- which streams really run at the same time
// HIP maps streams onto a few hardware queues and two streams on one queue run one after the other, whatever their events
// allow.  Which queue a new stream gets depends on every stream the PROCESS already has: under bench.py, with a torch process
// group alive (its stream pools hold references on every queue), both lanes of a pipeline landed on ONE queue and the
// "pipelined" step was slower than the serial one (0.76 vs 0.58 ms at a 1.25M-row shard; kernel trace: every lane kernel on
// queue 8, tools/r05_native_trace.sh) while the same pipeline ran at 0.46 ms in a process of its own.  HIP has no call that
// names a stream's queue, so it is MEASURED: a kernel that spins for 500 us on stream a, an empty kernel on stream b -- if b's
// is done within 300 us of the launches, the two streams are on different queues.  oi_pipeline_create draws candidate streams
// until it has lanes + 1 that are pairwise concurrent (at most PL_CANDIDATES; then it takes what it has).
__global__ void pl_spin_kernel(uint64_t ticks) { // bounded: leaves after `ticks` of the 100 MHz wall clock whatever happens
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void pl_noop_kernel() {}

namespace {

#define PL_CANDIDATES 12

bool streams_concurrent(hipStream_t a, hipStream_t b, hipEvent_t ev) {
    // by the CLOCK: the empty kernel on b is done ~30 us after its launch if b has a queue of its own, and only after the 500 us
    // spin if it sits behind it on a's queue.  (hipStreamQuery(a) right after b's event is no test: the runtime may not have
    // noticed a's completion yet, and two streams on ONE queue then look concurrent -- the first version of this did.)
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(pl_spin_kernel, dim3(1), dim3(64), 0, a, (uint64_t)50000); // 500 us of the 100 MHz clock
    hipLaunchKernelGGL(pl_noop_kernel, dim3(1), dim3(64), 0, b);
    bool conc = false;
    if (hipEventRecord(ev, b) == hipSuccess && hipEventSynchronize(ev) == hipSuccess)
        conc = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 300.0;
    (void)hipStreamSynchronize(a);
    (void)hipGetLastError();
    return conc;
}

// `want` pairwise-concurrent streams (as many as could be found among PL_CANDIDATES fresh ones, padded with further fresh
// streams otherwise); the rest are destroyed.  *found = how many of them are pairwise concurrent.
int pick_streams(uint32_t want, int prio, std::vector<hipStream_t> *out, uint32_t *found) {
    std::vector<hipStream_t> keep, drop;
    hipEvent_t ev = nullptr;
    OI_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int c = 0; c < PL_CANDIDATES && keep.size() < want; ++c) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess) break;
        bool ok = true;
        for (hipStream_t k : keep)
            if (!streams_concurrent(k, st, ev)) { ok = false; break; }
        (ok ? keep : drop).push_back(st);
    }
    *found = (uint32_t)keep.size();
    while (keep.size() < want && !drop.empty()) { keep.push_back(drop.back()); drop.pop_back(); } // (not concurrent with all: still a stream)
    for (hipStream_t st : drop) (void)hipStreamDestroy(st);
    (void)hipEventDestroy(ev);
    while (keep.size() < want) {
        hipStream_t st = nullptr;
        OI_HIP_CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio));
        keep.push_back(st);
    }
    *out = keep;
    return OI_OK;
}

struct Slot {
    uint32_t *packed = nullptr; // device, OI_PACKED_WORDS(B_max, depth)
    uint32_t *flat = nullptr;   // device, world x the above (== packed when there is no exchange)
    uint8_t *q_dev = nullptr;   // device staging of host queries [vectors | terms | offsets]
    uint8_t *q_pin = nullptr;   // page-locked image of the same
    float *out_dev = nullptr;   // device results of a host-output batch [scores K | docs K | counts B]
    uint8_t *out_pin = nullptr; // page-locked image of the same
    hipEvent_t lists_done = nullptr, fused = nullptr;
    // the batch that occupies the slot
    uint64_t ticket = 0; // 0 = free / retired
    uint32_t B = 0;
    int location = OI_DEVICE;
    float *h_scores = nullptr;
    uint32_t *h_docs = nullptr, *h_counts = nullptr;
};

} // namespace

struct oi_pipeline {
    oi_index *src = nullptr;
    oi_comm *comm = nullptr;
    uint32_t world = 1, B_max = 0, depth = 0, k = 0, q_terms_max = 0;
    size_t q_bytes = 0, out_bytes = 0, words = 0;
    std::vector<oi_ctx *> lane_ctx;
    std::vector<oi_index *> lane_idx;
    std::vector<hipStream_t> lane_st;
    oi_ctx *fuse_ctx = nullptr;
    hipStream_t fuse_st = nullptr;
    hipEvent_t ev_in = nullptr;
    std::vector<Slot> slots;
    uint64_t n_submitted = 0;
    uint32_t concurrent_streams = 0; // of lanes + 1: how many were measured to run at the same time (oi_pipeline_create)
    std::mutex mu;
    // OI_PIPELINE_TRACE=1: host time of submit by phase (us, summed), printed to stderr at destroy -- a host-bound pipeline shows here
    bool trace = false;
    double t_phase[5] = {0, 0, 0, 0, 0}; // wait/stage, lists, exchange, fuse, rest
};

namespace {

void free_slot(Slot &s, bool own_flat) {
    if (s.packed) (void)hipFree(s.packed);
    if (own_flat && s.flat) (void)hipFree(s.flat);
    if (s.q_dev) (void)hipFree(s.q_dev);
    if (s.q_pin) (void)hipHostFree(s.q_pin);
    if (s.out_dev) (void)hipFree(s.out_dev);
    if (s.out_pin) (void)hipHostFree(s.out_pin);
    if (s.lists_done) (void)hipEventDestroy(s.lists_done);
    if (s.fused) (void)hipEventDestroy(s.fused);
    s = Slot{};
}

// The batch in `s` is complete on the host's side: host outputs delivered, the slot free.  Blocks on the batch's last event.
int retire(oi_pipeline *p, Slot &s) {
    if (!s.ticket) return OI_OK;
    OI_HIP_CHECK(hipEventSynchronize(s.fused));
    if (s.location != OI_DEVICE) {
        const size_t K = (size_t)s.B * p->k;
        memcpy(s.h_scores, s.out_pin, K * 4);
        memcpy(s.h_docs, s.out_pin + K * 4, K * 4);
        memcpy(s.h_counts, s.out_pin + 2 * K * 4, (size_t)s.B * 4);
    }
    s.ticket = 0;
    return OI_OK;
}

void destroy(oi_pipeline *p) {
    if (!p) return;
    (void)hipSetDevice(p->src->ctx->device);
    for (hipStream_t st : p->lane_st)
        if (st) (void)hipStreamSynchronize(st);
    if (p->fuse_st) (void)hipStreamSynchronize(p->fuse_st);
    for (Slot &s : p->slots) free_slot(s, p->comm != nullptr);
    for (oi_index *v : p->lane_idx)
        if (v) oi_index_destroy(v);
    for (oi_ctx *c : p->lane_ctx)
        if (c) oi_destroy(c);
    if (p->fuse_ctx) oi_destroy(p->fuse_ctx);
    for (hipStream_t st : p->lane_st)
        if (st) (void)hipStreamDestroy(st);
    if (p->fuse_st) (void)hipStreamDestroy(p->fuse_st);
    if (p->ev_in) (void)hipEventDestroy(p->ev_in);
    delete p;
}

} // namespace

extern "C" int oi_pipeline_create(oi_index *idx, oi_comm *comm, uint32_t lanes, uint32_t max_queries, uint32_t max_query_terms,
                                  uint32_t depth, uint32_t k, oi_pipeline **out) {
    if (!idx || !out) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    *out = nullptr;
    OI_REQUIRE(lanes >= 1 && lanes <= 4, "pipeline: lanes=%u outside [1,4]", lanes);
    OI_REQUIRE(max_queries >= 1 && max_queries <= 4096, "pipeline: max_queries=%u outside [1,4096]", max_queries);
    OI_REQUIRE(max_query_terms >= 1 && max_query_terms <= 1024, "pipeline: max_query_terms=%u outside [1,1024]", max_query_terms);
    OI_REQUIRE(depth >= 1 && depth <= OI_MAX_DEPTH && k >= 1 && k <= OI_MAX_DEPTH, "pipeline: depth / k outside [1,%u]", OI_MAX_DEPTH);
    if (idx->is_view) { oi_set_error("pipeline: give it the index itself, not a view"); return OI_ERR_INVALID_ARG; }
    if (comm) OI_REQUIRE(comm->ctx == idx->ctx, "pipeline: the communicator belongs to another context");
    oi_ctx *ctx = idx->ctx;
    OI_HIP_CHECK(hipSetDevice(ctx->device));
    oi_pipeline *p = new oi_pipeline();
    p->src = idx;
    p->comm = comm;
    p->world = comm ? comm->world : 1;
    p->B_max = max_queries;
    p->depth = depth;
    p->k = k;
    p->q_terms_max = max_query_terms;
    p->words = (size_t)OI_PACKED_WORDS(max_queries, depth);
    const size_t vb = (sizeof(float) * (size_t)max_queries * idx->dim + 15) & ~(size_t)15;
    const size_t tb = (sizeof(uint32_t) * (size_t)max_queries * max_query_terms + 15) & ~(size_t)15;
    p->q_bytes = vb + tb + sizeof(uint32_t) * ((size_t)max_queries + 1);
    p->out_bytes = (2 * (size_t)max_queries * k + max_queries) * 4;
    auto fail = [&](int rc) { destroy(p); return rc; };
#define PL_HIP(expr)                                                                                            \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            oi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);           \
            return fail(OI_ERR_HIP);                                                                            \
        }                                                                                                       \
    } while (0)
#define PL_OI(expr)                  \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != OI_OK) return fail(rc_); \
    } while (0)
    PL_HIP(hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming));
    // Hardware queues.  HIP spreads the streams of ONE priority over at most GPU_MAX_HW_QUEUES (default 4) queues, and two
    // streams that share a queue run one after the other whatever their events allow.  A pipeline of two lanes has seven
    // streams in play (the caller's and its BM25 side stream, two lane streams with a side stream each, the fusing stream):
    // at one priority some of them share, and which ones depends on creation order (round 4 "calibrated" placements by trial:
    // 0.73 vs 0.84 ms per batch at a shard).  The lane and fusing streams are therefore created at the HIGHEST priority: a
    // pool of queues of their own (three streams, three queues), while the BM25 side streams stay at the default priority in
    // theirs.  Measured on one box at a 1.25M-row shard, 2 lanes: 0.56 ms per batch when streams share queues, 0.45 when they do
    // not (tools/r05_pipeline_probe.py).  OI_PIPELINE_STREAM_PRIORITY=0: default priority for everything (A/B; then
    // GPU_MAX_HW_QUEUES=8 in the environment of the process has the same effect).
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const char *prio_env = getenv("OI_PIPELINE_STREAM_PRIORITY");
    const int prio = (prio_env && atoi(prio_env) != 0) ? prio_greatest : 0;
    std::vector<hipStream_t> picked;
    PL_OI(pick_streams(lanes + 1, prio, &picked, &p->concurrent_streams));
    p->fuse_st = picked[lanes]; // (owned by p from here on: destroy() frees it with the lanes')
    for (uint32_t l = 0; l < lanes; ++l) p->lane_st.push_back(picked[l]);
    for (uint32_t l = 0; l < lanes; ++l) {
        oi_ctx *c = nullptr;
        PL_OI(oi_create_like(ctx, &c));
        p->lane_ctx.push_back(c);
        hipStream_t st = p->lane_st[l];
        PL_OI(oi_set_stream(c, st));
        // With two or more lanes the BM25 leg of a batch runs IN its lane's stream, not on a side stream of its own: the other
        // lane's corpus stream is what it overlaps with, and every further stream is one more contender for the few hardware
        // queues that really dispatch at the same time.  OI_PIPELINE_LANE_OVERLAP=1 (A/B): side streams as in oi_search.
        {
            const char *lo = getenv("OI_PIPELINE_LANE_OVERLAP");
            if (lanes >= 2 && !(lo && atoi(lo) != 0)) PL_OI(oi_set_overlap(c, 0));
        }
        oi_index *v = nullptr;
        PL_OI(oi_index_view(idx, c, &v));
        p->lane_idx.push_back(v);
    }
    PL_OI(oi_create_like(ctx, &p->fuse_ctx));
    PL_OI(oi_set_stream(p->fuse_ctx, p->fuse_st));
    p->slots.resize(std::max<uint32_t>(4, 2 * lanes));
    for (Slot &s : p->slots) {
        PL_HIP(hipMalloc(reinterpret_cast<void **>(&s.packed), p->words * 4));
        PL_HIP(hipMemset(s.packed, 0, p->words * 4));
        if (comm) PL_HIP(hipMalloc(reinterpret_cast<void **>(&s.flat), p->words * 4 * p->world));
        else s.flat = s.packed;
        PL_HIP(hipMalloc(reinterpret_cast<void **>(&s.q_dev), p->q_bytes));
        PL_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.q_pin), p->q_bytes, hipHostMallocDefault));
        PL_HIP(hipMalloc(reinterpret_cast<void **>(&s.out_dev), p->out_bytes));
        PL_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.out_pin), p->out_bytes, hipHostMallocDefault));
        PL_HIP(hipEventCreateWithFlags(&s.lists_done, hipEventDisableTiming));
        PL_HIP(hipEventCreateWithFlags(&s.fused, hipEventDisableTiming));
    }
    PL_HIP(hipDeviceSynchronize());
    { const char *tr = getenv("OI_PIPELINE_TRACE"); p->trace = tr && atoi(tr) != 0; }
#undef PL_HIP
#undef PL_OI
    *out = p;
    return OI_OK;
}

extern "C" void oi_pipeline_destroy(oi_pipeline *p) {
    if (!p) return;
    if (p->trace && p->n_submitted)
        fprintf(stderr, "[oi_pipeline] %llu submits; host us per submit: wait/stage %.1f, lists %.1f, exchange %.1f, fuse %.1f; %u of %zu streams concurrent\n",
                (unsigned long long)p->n_submitted, p->t_phase[0] / p->n_submitted, p->t_phase[1] / p->n_submitted,
                p->t_phase[2] / p->n_submitted, p->t_phase[3] / p->n_submitted, p->concurrent_streams, p->lane_st.size() + 1);
    {
        std::lock_guard<std::mutex> g(p->mu);
        for (Slot &s : p->slots) (void)retire(p, s);
    }
    destroy(p);
}

extern "C" int oi_pipeline_submit(oi_pipeline *p, const float *qv, const uint32_t *qt, const uint32_t *qo, uint32_t B,
                                  int location, float *scores_out, uint32_t *docs_out, uint32_t *counts_out,
                                  uint64_t *ticket_out) {
    if (!p) { oi_set_error("null pipeline"); return OI_ERR_INVALID_ARG; }
    OI_REQUIRE(qv && qo && scores_out && docs_out && counts_out, "pipeline submit: null buffer");
    OI_REQUIRE(B >= 1 && B <= p->B_max, "pipeline submit: n_queries=%u outside [1,%u] (oi_pipeline_create's max_queries)", B, p->B_max);
    std::lock_guard<std::mutex> g(p->mu);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto t0 = now();
    oi_ctx *cctx = p->src->ctx; // the caller's context: its stream is where device inputs were produced
    OI_HIP_CHECK(hipSetDevice(cctx->device));
    const uint64_t n = p->n_submitted;
    Slot &s = p->slots[n % p->slots.size()];
    const uint32_t lane = (uint32_t)(n % p->lane_st.size());
    hipStream_t lst = p->lane_st[lane];
    // the slot's previous occupant (n_slots batches ago): its host outputs are delivered now if nobody waited for it, and
    // its buffers are free for the lane once its fusion is done (a device-side wait: the host does not stall for it)
    const bool had_host_out = s.ticket && s.location != OI_DEVICE;
    if (had_host_out) OI_CHECK(retire(p, s));
    else if (n >= p->slots.size()) OI_HIP_CHECK(hipStreamWaitEvent(lst, s.fused, 0));
    s.ticket = 0;
    const float *d_qv = qv;
    const uint32_t *d_qt = qt, *d_qo = qo;
    if (location == OI_DEVICE) {
        hipStream_t cst;
        { std::lock_guard<std::mutex> gc(cctx->mu); cst = cctx->stream; }
        OI_HIP_CHECK(hipEventRecord(p->ev_in, cst)); // the queries were produced on the caller's stream
        OI_HIP_CHECK(hipStreamWaitEvent(lst, p->ev_in, 0));
        OI_HIP_CHECK(hipStreamWaitEvent(p->fuse_st, p->ev_in, 0)); // (and earlier readers of the output buffers ran there)
    } else {
        const uint32_t nt = qo[B];
        OI_REQUIRE(nt <= (uint64_t)p->B_max * p->q_terms_max, "pipeline submit: %u query terms exceed max_queries x max_query_terms", nt);
        OI_REQUIRE(nt == 0 || qt, "pipeline submit: null term buffer");
        const size_t vb = sizeof(float) * (size_t)B * p->src->dim, tb = sizeof(uint32_t) * (size_t)(nt ? nt : 1), ob = sizeof(uint32_t) * ((size_t)B + 1);
        const size_t off_t = (vb + 15) & ~(size_t)15, off_o = off_t + ((tb + 15) & ~(size_t)15), total = off_o + ob;
        // (the slot's previous batch has been scored long before its fusion finished; a host-input batch is only staged
        // over a slot whose lists are done: wait for that on the host -- n_slots batches back, normally long past)
        if (n >= p->slots.size()) OI_HIP_CHECK(hipEventSynchronize(s.lists_done));
        memcpy(s.q_pin, qv, vb);
        if (nt) memcpy(s.q_pin + off_t, qt, sizeof(uint32_t) * nt);
        memcpy(s.q_pin + off_o, qo, ob);
        OI_HIP_CHECK(hipMemcpyAsync(s.q_dev, s.q_pin, total, hipMemcpyHostToDevice, lst));
        d_qv = reinterpret_cast<const float *>(s.q_dev);
        d_qt = reinterpret_cast<const uint32_t *>(s.q_dev + off_t);
        d_qo = reinterpret_cast<const uint32_t *>(s.q_dev + off_o);
    }
    // the shard's two lists, packed, on the lane
    const auto t1 = now();
    OI_CHECK(oi_search_lists_packed(p->lane_idx[lane], d_qv, d_qt, d_qo, B, p->depth, OI_DEVICE, s.packed));
    const auto t2 = now();
    OI_HIP_CHECK(hipEventRecord(s.lists_done, lst));
    // exchange + fusion on the fusing stream
    OI_HIP_CHECK(hipStreamWaitEvent(p->fuse_st, s.lists_done, 0));
    const size_t W = (size_t)OI_PACKED_WORDS(B, p->depth), K = (size_t)B * p->k;
    if (p->comm) OI_CHECK(oi_rccl_all_gather_u32(p->comm->nccl, s.packed, s.flat, W, p->fuse_st)); // the ONE exchange of the batch
    const auto t3 = now();
    if (location == OI_DEVICE) {
        OI_CHECK(oi_fuse_packed(p->fuse_ctx, s.flat, p->world, B, p->depth, p->k, OI_DEVICE, scores_out, docs_out, counts_out));
    } else {
        float *o_s = s.out_dev;
        uint32_t *o_d = reinterpret_cast<uint32_t *>(o_s + K), *o_c = o_d + K;
        OI_HIP_CHECK(hipMemsetAsync(o_s, 0, (2 * K + B) * 4, p->fuse_st));
        OI_CHECK(oi_fuse_packed(p->fuse_ctx, s.flat, p->world, B, p->depth, p->k, OI_DEVICE, o_s, o_d, o_c));
        OI_HIP_CHECK(hipMemcpyAsync(s.out_pin, o_s, (2 * K + B) * 4, hipMemcpyDeviceToHost, p->fuse_st));
    }
    OI_HIP_CHECK(hipEventRecord(s.fused, p->fuse_st));
    if (p->trace) {
        const auto t4 = now();
        p->t_phase[0] += us(t0, t1); p->t_phase[1] += us(t1, t2); p->t_phase[2] += us(t2, t3); p->t_phase[3] += us(t3, t4);
    }
    s.ticket = n + 1;
    s.B = B;
    s.location = location;
    s.h_scores = scores_out; s.h_docs = docs_out; s.h_counts = counts_out;
    p->n_submitted = n + 1;
    if (ticket_out) *ticket_out = n + 1;
    return OI_OK;
}

extern "C" int oi_pipeline_wait(oi_pipeline *p, uint64_t ticket, int host_sync) {
    if (!p) { oi_set_error("null pipeline"); return OI_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> g(p->mu);
    OI_REQUIRE(ticket >= 1 && ticket <= p->n_submitted, "pipeline wait: ticket %llu was never issued", (unsigned long long)ticket);
    OI_HIP_CHECK(hipSetDevice(p->src->ctx->device));
    Slot &s = p->slots[(ticket - 1) % p->slots.size()];
    if (s.ticket != ticket) return OI_OK; // retired already (waited for, or its slot was reused: its outputs were delivered then)
    if (!host_sync && s.location == OI_DEVICE) {
        // order the CALLER's stream after the batch: its results may be read by work queued there afterwards; no host stall
        hipStream_t cst;
        { std::lock_guard<std::mutex> gc(p->src->ctx->mu); cst = p->src->ctx->stream; }
        OI_HIP_CHECK(hipStreamWaitEvent(cst, s.fused, 0));
        return OI_OK;
    }
    return retire(p, s);
}

extern "C" int oi_pipeline_drain(oi_pipeline *p) {
    if (!p) { oi_set_error("null pipeline"); return OI_ERR_INVALID_ARG; }
    {
        std::lock_guard<std::mutex> g(p->mu);
        OI_HIP_CHECK(hipSetDevice(p->src->ctx->device));
        for (Slot &s : p->slots) OI_CHECK(retire(p, s));
        for (hipStream_t st : p->lane_st) OI_HIP_CHECK(hipStreamSynchronize(st));
        OI_HIP_CHECK(hipStreamSynchronize(p->fuse_st));
    }
    // a candidate-pool overflow in any lane (a bug guard) surfaces here, as at every other host-visible point
    for (oi_ctx *c : p->lane_ctx) OI_CHECK(oi_synchronize(c));
    return oi_synchronize(p->fuse_ctx);
}

extern "C" int oi_pipeline_workspace_bytes(oi_pipeline *p, uint64_t *device_bytes_out, uint64_t *pinned_host_bytes_out) {
    if (!p) { oi_set_error("null pipeline"); return OI_ERR_INVALID_ARG; }
    uint64_t d = 0, h = 0;
    for (oi_ctx *c : p->lane_ctx) {
        uint64_t a = 0, b = 0;
        OI_CHECK(oi_workspace_bytes(c, &a, &b));
        d += a; h += b;
    }
    {
        uint64_t a = 0, b = 0;
        OI_CHECK(oi_workspace_bytes(p->fuse_ctx, &a, &b));
        d += a; h += b;
    }
    const uint64_t per_slot = (uint64_t)p->words * 4 * (p->comm ? 1 + p->world : 1) + p->q_bytes + p->out_bytes;
    d += per_slot * p->slots.size();
    h += (uint64_t)(p->q_bytes + p->out_bytes) * p->slots.size();
    if (device_bytes_out) *device_bytes_out = d;
    if (pinned_host_bytes_out) *pinned_host_bytes_out = h;
    return OI_OK;
}

// Timing hooks over the lanes (bench.py): oi_profile_reset / oi_profile_read of every lane context, summed.
extern "C" int oi_pipeline_profile_reset(oi_pipeline *p, int enable) {
    if (!p) { oi_set_error("null pipeline"); return OI_ERR_INVALID_ARG; }
    for (oi_ctx *c : p->lane_ctx) OI_CHECK(oi_profile_reset(c, enable));
    return OI_OK;
}
extern "C" int oi_pipeline_profile_read(oi_pipeline *p, const char *kernel_tag, double *total_ms_out, uint64_t *launches_out) {
    if (!p || !kernel_tag) { oi_set_error("null argument"); return OI_ERR_INVALID_ARG; }
    double ms = 0.0;
    uint64_t n = 0;
    for (oi_ctx *c : p->lane_ctx) {
        double a = 0.0;
        uint64_t b = 0;
        OI_CHECK(oi_profile_read(c, kernel_tag, &a, &b));
        if (a > 0.0) ms += a;
        n += b;
    }
    if (total_ms_out) *total_ms_out = ms;
    if (launches_out) *launches_out = n;
    return OI_OK;
}

// Diagnostics: how many of the pipeline's lanes + 1 streams were measured to run at the same time when it was created.
extern "C" int oi_pipeline_concurrent_streams(oi_pipeline *p, uint32_t *concurrent_out, uint32_t *streams_out) {
    if (!p) { oi_set_error("null pipeline"); return OI_ERR_INVALID_ARG; }
    if (concurrent_out) *concurrent_out = p->concurrent_streams;
    if (streams_out) *streams_out = (uint32_t)p->lane_st.size() + 1;
    return OI_OK;
}
