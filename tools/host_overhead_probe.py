import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import openintel_amd as oi
import _ablation  # noqa: F401  (OI_LIB=ablation: tools only)
from openintel_amd import synth, sharded
dev = torch.device("cuda:0")
n, B, DIM, DEPTH, K = 1_250_000, 64, 768, 1000, 100
ctx = oi.HipContext(0); ctx.use_torch_current_stream()
rows = synth.embeddings_torch(n, DIM, dev); terms, offs = synth.forward_index_torch(n, dev)
idx = oi.HybridIndex(ctx, n, DIM, synth.VOCAB); idx.set_embeddings(rows, normalize=False); idx.set_forward(terms, offs); idx.set_max_query_terms(4)
sr = sharded.make_hip_sharded(ctx, idx, dev); sr.finalize()
batches = [synth.query_batch_torch(B, DIM, dev, seed=synth.SEED_QUERY + 7919 * i) for i in range(4)]
pipe = sharded.ShardedPipeline(sr, oi.HipContext(0), B, DEPTH, K)
for i in range(8): pipe.submit(*batches[i % 4])
pipe.drain(); torch.cuda.synchronize()
t0 = time.perf_counter(); host = 0.0
for i in range(200):
    h0 = time.perf_counter(); pipe.submit(*batches[i % 4]); host += time.perf_counter() - h0
pipe.drain(); torch.cuda.synchronize()
print("period_ms", (time.perf_counter() - t0) / 200 * 1e3, "host_submit_ms", host / 200 * 1e3)
