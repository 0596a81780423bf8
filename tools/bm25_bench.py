#!/usr/bin/env python3
"""BM25 kernels alone at the bench shape: 10M-doc synthetic forward index, 64 queries x 4 terms, depth 1000.
Runs the stream kernel, the two older term-at-a-time kernels and the batch scan, checks that their ranked lists are identical, and
prints one JSON line with the per-batch kernel times (HIP events inside the library).
    python tools/bm25_bench.py [n_docs] [reps] [batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import openintel_amd as oi
import _ablation  # noqa: F401  (OI_LIB=ablation: tools only)
from openintel_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
DIM, DEPTH = 32, 1000   # a narrow embedding keeps the cosine leg out of the way
dev = torch.device("cuda:0")
ctx = oi.HipContext(0)
ctx.use_torch_current_stream()
ctx.set_overlap(False)  # time the BM25 kernels alone, not beside the cosine leg
rows = synth.embeddings_torch(n, DIM, dev)
terms, offs = synth.forward_index_torch(n, dev)
n_tokens = int(offs[-1].item())
idx = oi.HybridIndex(ctx, n, DIM, synth.VOCAB)
idx.set_embeddings(rows, normalize=False)
idx.set_forward(terms, offs)
idx.set_max_query_terms(4)
idx.finalize()
qv, qt, qo = synth.query_batch_torch(B, DIM, dev)
if os.environ.get("OI_BM25_DUMP_DF"):   # df of every query term (a weight-model check beside OI_BM25_STREAM_LIGHT_CSV)
    dfv = torch.bincount(terms.to(torch.int64), minlength=synth.VOCAB)   # token counts ~ doc frequency (terms rarely repeat in a doc)
    with open(os.environ["OI_BM25_DUMP_DF"], "w") as fh:
        for b in range(B):
            tq = qt[int(qo[b]):int(qo[b + 1])].to(torch.int64)
            fh.write("%d,%s\n" % (b, ",".join(str(int(dfv[t])) for t in tq)))
res = {}
lists = {}
MODES = (("stream", idx.BM25_STREAM), ("wave", idx.BM25_WAVE), ("taat", idx.BM25_TAAT), ("scan", idx.BM25_SCAN))
if len(sys.argv) > 4 and sys.argv[4] == "wave-only":   # tools/bm25_wave_ablate.sh
    MODES = MODES[1:2]
if len(sys.argv) > 4 and sys.argv[4] == "stream-only":
    MODES = MODES[:1]
for name, mode in MODES:
    idx.set_bm25_mode(mode)
    for _ in range(2):
        r = idx.search_lists(qv, qt, qo, depth=DEPTH)
    torch.cuda.synchronize()
    ctx.profile_reset(True)
    for _ in range(reps):
        r = idx.search_lists(qv, qt, qo, depth=DEPTH)
    torch.cuda.synchronize()
    ms, launches = ctx.profile_read("bm25")
    ctx.profile_reset(False)
    res[name] = {"ms_per_batch": ms / reps, "launches_per_batch": launches / reps}
    lists[name] = (r.bm25_docs.clone(), r.bm25_scores.clone(), r.bm25_counts.clone())
if len(MODES) == 1:
    print(json.dumps({"docs": n, "batch": B, **res}))
    sys.exit(0)
same = all(all(bool(torch.equal(a, b)) for a, b in zip(lists["taat"], lists[o])) for o in ("scan", "wave", "stream"))
# algorithmic bytes of term-at-a-time: 8 B per posting of the batch's terms + 8 B per (block, term) bounds lookup
df = idx.local_stats()[1].astype("int64")
qt_h = qt.cpu().numpy()
taat_bytes = int(8 * df[qt_h].sum() + 8 * ((n + 32767) // 32768) * qt_h.size)
for k in ("stream", "wave", "taat"):
    res[k]["algorithmic_bytes"] = taat_bytes
    res[k]["algorithmic_GBs"] = taat_bytes / (res[k]["ms_per_batch"] / 1e3) / 1e9
    res[k]["frac_of_8TBs"] = res[k]["algorithmic_GBs"] / 8000.0
scan_bytes = 4 * n_tokens + 8 * (n + 1)
res["scan"]["algorithmic_GBs"] = scan_bytes / (res["scan"]["ms_per_batch"] / 1e3) / 1e9
res["scan"]["frac_of_8TBs"] = res["scan"]["algorithmic_GBs"] / 8000.0
print(json.dumps({"docs": n, "tokens": n_tokens, "batch": B, "depth": DEPTH, "lists_identical": same, **res}))
