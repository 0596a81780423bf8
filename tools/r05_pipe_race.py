#!/usr/bin/env python3
"""Round 5: hunt for the mismatch tests/test_gpu_pipeline.py showed once at 3 lanes (interleaved submit / wait, device buffers)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import openintel_amd as oi
from test_gpu_pipeline import _case, _index, _same

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
K, DEPTH = 50, 200
rows, terms, offs, batches = _case()
if "comm" in sys.argv:      # the state tests/test_gpu_native_comm.py leaves behind: RCCL loaded, a communicator made and destroyed
    c0 = oi.HipContext(0)
    cm = oi.NativeComm(c0, oi.NativeComm.unique_id(), 0, 1)
    sh = _index(c0, rows[:5000], terms[:int(offs[5000])], offs[:5001])
    sh.close(); cm.close(); c0.close()
ctx = oi.HipContext(0)
idx = _index(ctx, rows, terms, offs)
want = [idx.search(q, qt, qo, k=K, depth=DEPTH) for q, qt, qo in batches]
pipe = oi.NativePipeline(idx, lanes=lanes, max_queries=64, max_query_terms=4, depth=DEPTH, k=K)
print("concurrent streams", pipe.concurrent_streams())
dev = torch.device("cuda:0")
dbs = [[torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) for x in b] for b in batches]
torch.cuda.synchronize()
bad = 0
for rnd in range(rounds):
    subs = []
    for i, b in enumerate(dbs):
        t, out = pipe.submit(*b)
        subs.append((i, t, out))
        if i % 2:
            pipe.wait(t)
            if not _same(out, want[i]):
                bad += 1
                g, w = out.docs.cpu().numpy().view(np.uint32), want[i].docs
                gs, ws = out.scores.cpu().numpy(), want[i].scores
                nd = int((g != w).sum())
                rows_bad = np.nonzero((g != w).any(axis=1))[0]
                print("round %d batch %d (ticket %d, B=%d): %d docs differ in query rows %s; counts equal %s; first bad row got %s want %s" % (
                    rnd, i, t, w.shape[0], nd, rows_bad[:8], np.array_equal(out.counts.cpu().numpy().view(np.uint32), want[i].counts),
                    g[rows_bad[0]][:6] if rows_bad.size else None, w[rows_bad[0]][:6] if rows_bad.size else None))
    pipe.drain()
    for i, t, out in subs:
        if not _same(out, want[i]):
            bad += 1
            g, w = out.docs.cpu().numpy().view(np.uint32), want[i].docs
            gs, ws = out.scores.cpu().numpy(), want[i].scores
            gc_, wc = out.counts.cpu().numpy().view(np.uint32), want[i].counts
            rows_bad = np.nonzero((g != w).any(axis=1) | (gs.view(np.uint32) != ws.view(np.uint32)).any(axis=1))[0]
            r0 = rows_bad[0] if rows_bad.size else 0
            print("round %d after drain: batch %d (ticket %d, B=%d, lane %d, slot %d) differs: rows %s of %d; counts equal %s; row %d got docs %s scores %s | want docs %s scores %s" % (
                rnd, i, t, w.shape[0], (t - 1) % lanes, (t - 1) % max(4, 2 * lanes), rows_bad[:10], w.shape[0], np.array_equal(gc_, wc), r0, g[r0][:5], gs[r0][:5], w[r0][:5], ws[r0][:5]))
print("mismatches:", bad)
pipe.close(); idx.close(); ctx.close()
