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
    """Contiguous row range of `rank`; block-aligned to 4 rows so shard bases stay 16-byte aligned."""
    per = (n_total + world - 1) // world
    per = (per + 3) // 4 * 4
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

    # ---- build: global collection statistics
    def finalize(self) -> None:
        import torch
        tokens, df = self.local.local_stats()
        stats = torch.tensor([self.local.n_docs, tokens], dtype=torch.int64, device=self.device)
        gdf = torch.from_numpy(df.astype(np.int64)).to(self.device)
        if self.world > 1:
            self.dist.all_reduce(stats, group=self.group)
            self.dist.all_reduce(gdf, group=self.group)
        n_global, tok_global = (int(x) for x in stats.cpu())
        self.n_global = n_global
        self.local.finalize(n_global, tok_global, gdf.cpu().numpy().astype(np.uint32))

    # ---- query
    def search(self, qv, qt, qo, k: int, depth: int):
        """Returns (scores [B,k], docs [B,k], counts [B]) -- identical on every rank."""
        import torch
        if self.fuse_packed is not None:
            B = int(qv.shape[0])
            packed = self.local.search_lists_packed(qv, qt, qo, depth=depth)
            if self.world > 1:
                flat = torch.empty(self.world * packed.numel(), dtype=packed.dtype, device=packed.device)
                self.dist.all_gather_into_tensor(flat, packed, group=self.group)   # the ONE exchange per batch
            else:
                flat = packed
            return self.fuse_packed(flat, self.world, B, depth, k)
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
