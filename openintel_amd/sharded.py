"""Row-sharded hybrid retrieval over one process per GPU (torch.distributed; "nccl" = RCCL on ROCm).

Path:
  build   every rank holds rows [base, base + n_local) of the corpus.  BM25 needs GLOBAL
          statistics: one all-reduce of the (vocab) document-frequency vector and of
          (n_docs, token count) before the impacts are computed.
  query   queries are replicated.  Each rank produces its shard's two ranked lists of depth k'
          (cosine, BM25).  RRF needs GLOBAL ranks, so the exchange is ONE all-gather of the packed
          per-shard lists (B x 2 x k' x 8 B per rank ~ 1 MB at B=64, k'=1000: latency-bound over
          xGMI), then every rank merges to the global top-k' per list and fuses.  Fusing per shard
          and merging afterwards would NOT be equivalent.

The local engine and the merge/fuse steps are injected so that the collective choreography can be
exercised on CPU with gloo (tests/test_sharded_gloo.py); on the GPU they are the HIP kernels.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


def shard_bounds(n_total: int, world: int, rank: int):
    """Contiguous row range of `rank`; block-aligned to 4 rows so shard bases stay 16-byte aligned.

    Raises ValueError when the rounding would leave some rank without rows (an index shard needs >= 1 row).
    The check depends on (n_total, world) only, so EVERY rank raises before any of them enters a collective --
    a rank that failed alone would leave the others hanging in all_reduce / all_gather."""
    per = (n_total + world - 1) // world
    per = (per + 3) // 4 * 4
    if world < 1 or n_total < 1 or (world - 1) * per >= n_total:
        raise ValueError("cannot shard %d rows over %d ranks in 4-row blocks: the last rank(s) would be empty"
                         % (n_total, world))
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


class ShardedRetriever:
    """`local` needs: local_stats() -> (tokens, df ndarray), finalize(N, tokens, df),
    search_lists(qv, qt, qo, depth) -> RankedLists (torch tensors on `device`), n_docs, vocab."""

    def __init__(self, local, device, merge: Callable, fuse: Callable, group=None,
                 fuse_packed: Optional[Callable] = None):
        import torch.distributed as dist
        self.local, self.device, self.merge, self.fuse, self.group = local, device, merge, fuse, group
        # fast path: the engine emits the packed exchange format and one call merges + fuses it
        self.fuse_packed = fuse_packed if hasattr(local, "search_lists_packed") else None
        self.dist = dist
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # the collectives are skipped at world 1 -- unless a caller with an initialised process group of ONE rank sets this
        # (bench.py OI_BENCH_FORCE_DIST=1, tests/test_gpu_rccl.py: the real RCCL calls on a one-GPU box)
        self.exchange = self.world > 1

    # ---- build: global collection statistics
    def finalize(self) -> None:
        import torch
        tokens, df = self.local.local_stats()
        stats = torch.tensor([self.local.n_docs, tokens], dtype=torch.int64, device=self.device)
        gdf = torch.from_numpy(df.astype(np.int64)).to(self.device)
        if self.exchange:
            self.dist.all_reduce(stats, group=self.group)
            self.dist.all_reduce(gdf, group=self.group)
        n_global, tok_global = (int(x) for x in stats.cpu())
        self.n_global = n_global
        self.local.finalize(n_global, tok_global, gdf.cpu().numpy().astype(np.uint32))

    # ---- query
    def search(self, qv, qt, qo, k: int, depth: int, check: bool = True):
        """Returns (scores [B,k], docs [B,k], counts [B]) -- identical on every rank.

        The device path is asynchronous and the library reports a candidate-pool overflow (OI_ERR_OVERFLOW, a bug
        guard) only at a host-visible point: with check=True (default) the engine is synchronised after the fuse so
        that a truncated list can never be returned silently; a pipelined caller (bench.py's timed loop) passes
        check=False and calls `self.check()` once after the loop."""
        import torch
        if self.fuse_packed is not None:
            B = int(qv.shape[0])
            packed = self.local.search_lists_packed(qv, qt, qo, depth=depth)
            if self.exchange:
                flat = torch.empty(self.world * packed.numel(), dtype=packed.dtype, device=packed.device)
                self.dist.all_gather_into_tensor(flat, packed, group=self.group)   # the ONE exchange per batch
            else:
                flat = packed
            out = self.fuse_packed(flat, self.world, B, depth, k)
            if check:
                self.check()
            return out
        L = self.local.search_lists(qv, qt, qo, depth=depth)
        if self.world == 1:
            cos_d, cos_c, bm_d, bm_c = L.cos_docs, L.cos_counts, L.bm25_docs, L.bm25_counts
        else:
            B = int(L.cos_counts.shape[0])
            # one packed buffer per rank: [2 lists][B][depth] scores | docs, then [2][B] counts
            sc = torch.stack([L.cos_scores, L.bm25_scores]).contiguous()
            dc = torch.stack([L.cos_docs, L.bm25_docs]).contiguous()
            cn = torch.stack([L.cos_counts, L.bm25_counts]).contiguous()
            packed = torch.cat([sc.view(torch.int32).reshape(-1), dc.reshape(-1), cn.reshape(-1)])
            flat = torch.empty(self.world * packed.numel(), dtype=packed.dtype, device=packed.device)
            self.dist.all_gather_into_tensor(flat, packed, group=self.group)
            gathered = flat.view(self.world, packed.numel())
            n_list = 2 * B * depth
            g_sc = gathered[:, :n_list].contiguous().view(torch.float32).reshape(self.world, 2, B, depth)
            g_dc = gathered[:, n_list:2 * n_list].reshape(self.world, 2, B, depth)
            g_cn = gathered[:, 2 * n_list:].reshape(self.world, 2, B)
            _, cos_d, cos_c = self.merge(g_sc[:, 0].contiguous(), g_dc[:, 0].contiguous(), g_cn[:, 0].contiguous())
            _, bm_d, bm_c = self.merge(g_sc[:, 1].contiguous(), g_dc[:, 1].contiguous(), g_cn[:, 1].contiguous())
        return self.fuse(cos_d, cos_c, bm_d, bm_c, k)


    def check(self) -> None:
        """Synchronise the local engine and raise if it flagged an overflow (HybridIndex: oi_synchronize)."""
        ctx = getattr(self.local, "ctx", None)
        if ctx is not None:
            ctx.synchronize()


def calibrate_lanes(lane0, period: Callable, make_lane: Callable, drop_lane: Callable, use: Callable, placements: int,
                    max_lanes: int):
    """The decision loop of ShardedPipeline.calibrate, free of streams and devices so that its ONE invariant can be tested
    with gloo on CPU (tests/test_sharded_gloo.py): the number of period() calls -- each of which runs a fixed number of
    batches, i.e. of all-gathers -- is 1 + (max_lanes - 1) * placements on every rank, whatever the timings say.

    period() -> ms per batch with the lanes last given to use(lanes); make_lane() -> a fresh lane; drop_lane(lane) frees one.
    Returns (kept lanes, best ms, the list of trials)."""
    kept = [lane0]
    use(kept)
    tried = [{"lanes": 1, "ms": period()}]
    best_ms = tried[0]["ms"]
    searching = True
    for n_lanes in range(2, max(2, int(max_lanes)) + 1):
        best_lane = None
        for p in range(max(0, int(placements))):
            lane = make_lane()
            use(kept + [lane])
            ms = period()                                  # (run on EVERY rank, adopted or not: the collectives stay in step)
            tried.append({"lanes": n_lanes, "placement": p, "ms": ms, "considered": searching})
            if searching and ms < 0.97 * best_ms:          # another lane has to earn its keep
                if best_lane is not None:
                    drop_lane(best_lane)
                best_ms, best_lane = ms, lane
            else:
                drop_lane(lane)
        if best_lane is None:
            searching = False                              # later levels are still timed (and discarded), see above
        else:
            kept.append(best_lane)
        use(kept)
    return kept, best_ms, tried


class ShardedPipeline:
    """Throughput mode of the sharded query (GPU only).  Batches are independent, so
      * the exchange + fusion of batch i run on a second stream (`fuse_ctx`) while the shard's lists of batch i+1 are
        already being scored -- per batch nothing changes (same calls, same results, ONE all-gather); the ~0.1 ms of
        exchange + fusion stop adding to the step time, which at 8 GPUs is 10 % of it;
      * with `lane_ctxs` (further HipContexts of the same device) the lists of consecutive batches are scored through
        VIEWS of the shard (HybridIndex.view: same buffers, own stream and workspaces), one batch per lane in flight: the
        next batch's screen fills the CUs that the selects and the rescoring of the previous one leave idle
        (0.874 -> 0.813 ms per batch at 1.25M rows, tools/dual_stream_probe.py).

        pipe = ShardedPipeline(retriever, fuse_ctx, n_queries, depth, k, lane_ctxs=[HipContext(dev)])
        slot = pipe.submit(qv, qt, qo)      # asynchronous; results of this batch land in pipe.results[slot]
        ...
        pipe.drain()                        # everything submitted so far is complete (and checked for overflow)

    `fuse_ctx` is a HipContext of its own on the same device (the fusion must not queue behind the next batch's kernels
    on a scoring stream); the caller owns it (and the retriever's own ctx) -- close() closes only what the pipeline made
    or was handed as a lane.  The collectives are issued from the host in submission order on ONE stream, so every rank
    runs them in the same order whatever the lanes do.

    Ownership of a batch's tensors (all asynchronous):
      * inputs: `qv, qt, qo` must stay ALIVE AND UNMODIFIED until the batch's lists are scored.  A lane stream that is not
        the caller's gets `record_stream()` on them, so that dropping the tensors right after submit() is safe (the caching
        allocator will not hand the memory out while the lane still reads it); OVERWRITING them in place is only safe
        after `wait(slot)` (or drain()).
      * outputs: `results[slot]` is valid on the caller's stream after `wait(slot)`; the slot is written again
        `n_slots` submits later -- submit() orders that write after everything the caller's stream had queued by then, so
        a consumer that was enqueued on the caller's stream before the reuse is safe."""

    def __init__(self, retriever: "ShardedRetriever", fuse_ctx, n_queries: int, depth: int, k: int, lane_ctxs=(),
                 graphs: bool = False):
        import torch
        from .retriever import SearchResult, fuse_packed, packed_words
        self.r, self.fctx, self.B, self.depth, self.k = retriever, fuse_ctx, int(n_queries), int(depth), int(k)
        self._fuse_packed = fuse_packed
        dev = retriever.device
        self.dev = dev
        self.side = torch.cuda.Stream(device=dev)
        fuse_ctx.set_stream(self.side)
        # graphs=True: every batch is copied into its slot's staging buffers, so a slot's two C calls (lists, fusion) see
        # the same pointers every time and the library replays them as captured hipGraphs -- one launch call each instead
        # of ~35 launches / memsets / event operations (oi_set_graph_replay).  OFF by default: measured on ROCm 7.2 a
        # replayed graph runs its parallel branches (BM25 leg beside the cosine leg) one after the other and graphs of two
        # lanes do not overlap each other -- 0.84-0.98 ms per batch against 0.76 eager at a 1.25M-row shard
        self.graphs = bool(graphs)
        self.stage = {}
        for c in [retriever.local.ctx, fuse_ctx, *lane_ctxs]:
            c.set_graph_replay(self.graphs)
        # EVERY lane is a view of the retriever's index on a context and a stream of its own -- lane 0 too, on a context the
        # pipeline makes like the retriever's (HipContext.like).  The retriever's OWN context is never touched: round 3 moved it
        # onto a private stream at the first submit() and back at drain(), so a direct retriever.search between the two (or
        # after an exception skipped the drain) ran on the lane stream, unordered against the caller's, and a stream the caller
        # had set was lost.  (Round 2 ran lane 0 on the caller's stream: `lane_stream.wait_stream(main)` -- "the
        # queries are ready" -- then made every batch of lane 1 wait for the whole previous batch of lane 0, which sat in front
        # of it on that stream: consecutive batches overlapped only every other time.  rocprofv3 timeline,
        # profiles/r03_shard_timeline_before.txt.)
        from .context import HipContext
        self.lane0_ctx = HipContext.like(retriever.local.ctx)
        self.lane0_ctx.set_graph_replay(self.graphs)
        st0 = torch.cuda.Stream(device=dev)
        self.lane0_ctx.set_stream(st0)
        self.lanes = [(retriever.local.view(self.lane0_ctx), st0)]
        for c in lane_ctxs:
            st = torch.cuda.Stream(device=dev)
            c.set_stream(st)
            self.lanes.append((retriever.local.view(c), st))
        self.n_shards = retriever.world
        self.n_slots = max(6, 2 * len(self.lanes))   # (6: room for a second and a third lane added by calibrate())
        self.calibration = None
        words = packed_words(self.B, self.depth)
        mk = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
        self.packed = [mk(words, torch.int32) for _ in range(self.n_slots)]
        self.flat = ([mk(words * retriever.world, torch.int32) for _ in range(self.n_slots)] if retriever.exchange
                     else self.packed)
        self.results = [SearchResult(torch.zeros((self.B, self.k), dtype=torch.float32, device=dev),
                                     torch.zeros((self.B, self.k), dtype=torch.int32, device=dev),
                                     torch.zeros((self.B,), dtype=torch.int32, device=dev)) for _ in range(self.n_slots)]
        self.lists_done = [torch.cuda.Event() for _ in range(self.n_slots)]
        self.fused = [torch.cuda.Event() for _ in range(self.n_slots)]
        self.n = 0

    def submit(self, qv, qt, qo) -> int:
        import torch
        slot = self.n % self.n_slots
        index, st = self.lanes[self.n % len(self.lanes)]
        main = torch.cuda.current_stream(self.dev)
        st.wait_stream(main)                           # the queries were produced on the caller's stream (which carries
        for t in (qv, qt, qo):                         # nothing of the pipeline's own: no false dependency between batches)
            if hasattr(t, "record_stream"):            # ... and are read on the lane's: the allocator must know
                t.record_stream(st)
        if self.n >= self.n_slots:
            st.wait_event(self.fused[slot])            # the slot's packed buffer is free again
            self.side.wait_stream(main)                # readers of results[slot] queued on the caller's stream go first
        with torch.cuda.stream(st):
            if self.graphs:
                qv, qt, qo = self._staged(slot, qv, qt, qo)
            index.search_lists_packed(qv, qt, qo, depth=self.depth, out=self.packed[slot])
            self.lists_done[slot].record(st)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.lists_done[slot])
            self._fuse_packed(self.fctx, self._exchange(slot), self.n_shards, self.B, self.depth, self.k, out=self.results[slot])
            self.fused[slot].record(self.side)
        self.n += 1
        return slot

    def _staged(self, slot: int, qv, qt, qo):
        """Copy the batch into the slot's staging buffers (on the lane stream): stable pointers for the captured call."""
        import torch
        sg = self.stage.get(slot)
        if sg is None or sg[0].shape != qv.shape or sg[1].numel() < qt.numel() or sg[2].shape != qo.shape:
            sg = (torch.empty_like(qv), torch.empty(max(qt.numel(), 4 * self.B), dtype=qt.dtype, device=qt.device), torch.empty_like(qo))
            self.stage[slot] = sg
        sg[0].copy_(qv, non_blocking=True)
        sg[1][:qt.numel()].copy_(qt, non_blocking=True)
        sg[2].copy_(qo, non_blocking=True)
        return sg

    def _exchange(self, slot: int):
        """All shards' packed lists of the batch in `slot` (runs on the exchange stream): the ONE all-gather per batch."""
        if self.r.exchange:
            self.r.dist.all_gather_into_tensor(self.flat[slot], self.packed[slot], group=self.r.group)
        return self.flat[slot]

    def wait(self, slot: int) -> None:
        """Order the caller's current stream after the fusion of the batch in `slot`: results[slot] may be read (and the
        batch's input tensors overwritten) by work queued on that stream afterwards.  No host synchronisation."""
        import torch
        torch.cuda.current_stream(self.dev).wait_event(self.fused[slot])

    def calibrate(self, batches, make_ctx: Callable, reps: int = 16, placements: int = 4, max_lanes: int = 2) -> dict:
        """Choose the number of lanes EMPIRICALLY, and each added lane's stream with it.

        Whether another lane pays depends on which hardware queue its stream lands on (HIP streams share a handful of
        queues; measured at a 1.25M-row shard, same code: 0.745-0.82 ms per batch when the two lanes' streams do not share
        a queue with each other's corpus passes, 0.88-0.92 -- no gain -- or worse when they do; round 5:
        bench.py and the library ask for 8 hardware queues, GPU_MAX_HW_QUEUES, which makes the good placement the rule).
        So: time `reps` batches with lane 0 alone, then -- lane by lane up to `max_lanes` -- with each of `placements`
        freshly created (context, stream, view) as the next lane; a lane is kept if its best placement is at least 3 % faster
        than the set-up without it; a rank that stopped adopting lanes still RUNS every later trial (and throws it away).
        EVERY RANK RUNS THE SAME NUMBER OF BATCHES, whatever it decides: each batch is one all-gather when world > 1, so the
        collective counts must not depend on a rank's own timings (they did for max_lanes >= 3).  The choice
        itself is local to the rank.  `make_ctx()` returns a new HipContext of this device, configured like the retriever's."""
        import time

        import torch
        assert len(self.lanes) == 1, "calibrate() starts from the single-lane pipeline"

        def period():
            for i in range(2 * self.n_slots):          # every slot twice: the second call of a slot is the (slow) capture
                self.submit(*batches[i % len(batches)])
            self.drain()
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for i in range(reps):
                self.submit(*batches[i % len(batches)])
            self.drain()
            torch.cuda.synchronize(self.dev)
            return (time.perf_counter() - t0) / reps * 1e3

        def make_lane():
            c = make_ctx()
            st = torch.cuda.Stream(device=self.dev)
            c.set_stream(st)
            c.set_graph_replay(self.graphs)
            return (self.r.local.view(c), st)

        def drop_lane(lane):
            c = lane[0].ctx
            lane[0].close()
            c.close()

        def use(lanes):
            self.lanes = lanes

        kept, best_ms, tried = calibrate_lanes(self.lanes[0], period, make_lane, drop_lane, use, placements, max_lanes)
        self.lanes = kept
        self.calibration = {"chosen_lanes": len(self.lanes), "period_ms": best_ms, "tried": tried}
        return self.calibration

    def drain(self) -> None:
        import torch
        main = torch.cuda.current_stream(self.dev)
        for _, st in self.lanes:
            main.wait_stream(st)
        main.wait_stream(self.side)
        self.fctx.synchronize()
        for index, _ in self.lanes:
            index.ctx.synchronize()                    # raises if an engine flagged a pool overflow
        self.r.check()

    def close(self) -> None:
        """Drain, then close every lane: the view AND its context (each lane context owns a full set of search workspaces --
        candidate pools, screen pool, BM25 pools: HBM that would otherwise wait for the garbage collector).  `fuse_ctx` and
        the retriever (its index and its own context) stay the caller's."""
        import torch
        if self.n:
            self.drain()
        torch.cuda.synchronize(self.dev)
        for index, _ in self.lanes:
            c = index.ctx
            index.close()
            c.close()
        self.lanes = []


class ShardedAnalyzer:
    """the lexicon path over sharded posts.

    Replaces the two loops of SpeculationEngine::social_summary (src/domain/engine/speculation_engine.rs:76-97)
    when the posts of one report are spread over ranks: every rank scores ITS posts (LexiconAnalyzer::score,
    lexicon.rs:53-73) and reduces them to the raw sums (`oi_social_counters`: total, by_source[2], bullish, bearish,
    neutral, spec_count -- u64 -- and polarity_sum, f64); ONE all-gather of those 8 words per rank follows.  The
    integer fields are summed (exact); the f64 partials are added in RANK order, so the result is the same bits on
    every rank and run to run, and differs from the reference's input-order sum only by the reassociation bound of
    DESIGN.md section 5.  The global counters then go through SpeculationEngine.aggregate_counters (host scalars).

    `analyze_shard(*shard_inputs)` returns an object with those eight fields (the HIP one: make_hip_sharded_analyzer;
    tests/test_sharded_gloo.py injects the CPU oracle).  `segment_summaries` is the batch callers' form: whole tickers
    per rank, bit-identical records."""

    WORDS = 8

    def __init__(self, analyze_shard: Callable, device, group=None, scan_segments_shard: Optional[Callable] = None):
        import torch.distributed as dist
        self.analyze_shard, self.device, self.group, self.dist = analyze_shard, device, group, dist
        self.scan_segments_shard = scan_segments_shard
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def summary(self, *shard_inputs):
        """Global raw sums as a _lib.SocialCounters (identical on every rank)."""
        import torch
        from . import _lib
        c = self.analyze_shard(*shard_inputs)
        ints = [int(c.total), int(c.by_source[0]), int(c.by_source[1]), int(c.bullish), int(c.bearish), int(c.neutral),
                int(c.spec_count)]
        words = np.zeros(self.WORDS, dtype=np.int64)
        words[:7] = ints
        words[7:8] = np.array([float(c.polarity_sum)], dtype=np.float64).view(np.int64)   # the f64's bits
        mine = torch.from_numpy(words).to(self.device)
        if self.world > 1:
            allw = torch.empty(self.world * self.WORDS, dtype=torch.int64, device=self.device)
            self.dist.all_gather_into_tensor(allw, mine, group=self.group)                 # the ONE exchange
        else:
            allw = mine
        g = allw.cpu().numpy().reshape(self.world, self.WORDS)
        out = _lib.SocialCounters()
        tot = g[:, :7].sum(axis=0)
        out.total, out.bullish, out.bearish, out.neutral, out.spec_count = (int(tot[0]), int(tot[3]), int(tot[4]),
                                                                            int(tot[5]), int(tot[6]))
        out.by_source[0], out.by_source[1] = int(tot[1]), int(tot[2])
        psum = 0.0
        for r in range(self.world):                      # rank order: reproducible
            psum += float(g[r, 7:8].view(np.float64)[0])
        out.polarity_sum = psum
        return out


    @staticmethod
    def ticker_bounds(n_tickers: int, world: int, rank: int):
        """Contiguous ticker range of `rank` (the batch callers' shards: whole tickers, so no sum is ever split)."""
        per = (n_tickers + world - 1) // world
        lo = min(n_tickers, rank * per)
        return lo, min(n_tickers, lo + per)

    def segment_summaries(self, n_tickers: int, *shard_inputs):
        """The batch callers' form over ranks: the TICKERS of a scan are sharded -- rank r holds the posts
        of tickers ticker_bounds(n_tickers, world, r), pooled -- every rank runs its scan + per-ticker reduction
        (`scan_segments_shard(*shard_inputs)` -> int64 tensor of 8 words per local ticker, an `oi_social_counters` each),
        and ONE all-gather of the records follows (padded to the largest shard).  A ticker's sums are computed whole on one
        rank, in input order: the gathered records are bit-identical to the unsharded ones, polarity_sum included -- no
        reassociation, unlike `summary`.  Returns an int64 array [n_tickers, 8] (host), the same on every rank."""
        import torch
        lo, hi = self.ticker_bounds(n_tickers, self.world, self.rank)
        mine = self.scan_segments_shard(*shard_inputs)
        assert mine.numel() == (hi - lo) * self.WORDS, "one 8-word record per local ticker"
        if self.world == 1:
            return mine.cpu().numpy().reshape(n_tickers, self.WORDS).copy()
        per = (n_tickers + self.world - 1) // self.world
        padded = torch.zeros(per * self.WORDS, dtype=torch.int64, device=self.device)
        padded[: mine.numel()] = mine
        allw = torch.empty(self.world * per * self.WORDS, dtype=torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(allw, padded, group=self.group)              # the ONE exchange
        g = allw.cpu().numpy().reshape(self.world, per, self.WORDS)
        rows = [g[r, : self.ticker_bounds(n_tickers, self.world, r)[1] - self.ticker_bounds(n_tickers, self.world, r)[0]]
                for r in range(self.world)]
        return np.concatenate(rows, axis=0)


def make_hip_sharded_analyzer(ctx, device, cfg=None, group=None) -> ShardedAnalyzer:
    """Wire the HIP lexicon scan + summary reduction of `ctx` into a ShardedAnalyzer.  Shard inputs: torch CUDA
    tensors (uint8 text blob, int64 offsets[n+1], uint8 sources[n] or None)."""
    from .analyzer import HipLexiconAnalyzer
    from .domain import EngineConfig
    cfg = cfg or EngineConfig()
    an = HipLexiconAnalyzer(ctx)

    def analyze_shard(d_blob, d_offsets, d_sources=None):
        # one pass over the shard's text: the scan with the social_summary reduction fused in, nothing written per post
        return an.summary_device(d_blob, d_offsets, d_sources, tau=cfg.bull_bear_threshold)

    def scan_segments_shard(d_blob, d_offsets, d_sources, d_seg_offsets):
        # the rank's tickers, pooled: one scan + one per-ticker reduction, records left in HBM
        import torch
        out = torch.zeros((d_seg_offsets.numel() - 1) * ShardedAnalyzer.WORDS, dtype=torch.int64, device=device)
        an.scan_segments_device(d_blob, d_offsets, d_sources, d_seg_offsets, out, cfg.bull_bear_threshold)
        return out

    return ShardedAnalyzer(analyze_shard, device, group, scan_segments_shard=scan_segments_shard)


def make_hip_sharded(ctx, index, device, group=None) -> ShardedRetriever:
    """Wire the HIP merge / RRF kernels of `ctx` around a HybridIndex shard."""
    from .retriever import fuse_packed, merge_lists, rrf_fuse

    def fuse(cd, cc, bd, bc, k):
        r = rrf_fuse(ctx, cd, cc, bd, bc, k)
        return r.scores, r.docs, r.counts

    def fuse_p(flat, n_shards, B, depth, k):
        r = fuse_packed(ctx, flat, n_shards, B, depth, k)
        return r.scores, r.docs, r.counts

    return ShardedRetriever(index, device, lambda s, d, c: merge_lists(ctx, s, d, c), fuse, group, fuse_packed=fuse_p)
