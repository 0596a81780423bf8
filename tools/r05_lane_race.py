#!/usr/bin/env python3
"""Round 5: which LIST differs when searches through views of one index overlap in time?  Three lane contexts (own streams) run
oi_search_lists_packed on rotating batches concurrently; every packed result is compared with the serial one, part by part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import openintel_amd as oi
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import _ablation  # noqa: F401
from test_gpu_pipeline import _case, _index

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
DEPTH = 200
DIM = 768 if 'd768' in sys.argv else 384
rows, terms, offs, batches = _case(dim=DIM)
if "comm" in sys.argv:
    c0 = oi.HipContext(0)
    cm = oi.NativeComm(c0, oi.NativeComm.unique_id(), 0, 1)
    sh = _index(c0, rows[:5000], terms[:int(offs[5000])], offs[:5001])
    sh.close(); cm.close(); c0.close()
ctx = oi.HipContext(0)
from openintel_amd import _lib
mode = _lib.OI_COSINE_EXACT if "exact" in sys.argv else _lib.OI_COSINE_SCREEN_STREAM if "stream" in sys.argv else _lib.OI_COSINE_SCREEN
ctx.set_cosine_mode(mode)
idx = _index(ctx, rows, terms, offs)
dev = torch.device("cuda:0")
dbs = [[torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) for x in b] for b in batches]
torch.cuda.synchronize()
ref = [idx.search_lists_packed(*b, depth=DEPTH).clone() for b in dbs]
torch.cuda.synchronize()
L = []
for l in range(lanes):
    c = oi.HipContext.like(ctx)
    st = torch.cuda.Stream(device=dev)
    c.set_stream(st)
    if "inline" in sys.argv:
        c.set_overlap(False)
    L.append((idx.view(c), st, c))
bad = 0
parts = {"cos_scores": 0, "cos_docs": 0, "bm_scores": 0, "bm_docs": 0, "counts": 0}
for rnd in range(rounds):
    outs = []
    for i, b in enumerate(dbs):
        v, st, c = L[i % lanes]
        with torch.cuda.stream(st):
            outs.append(v.search_lists_packed(*b, depth=DEPTH))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        if not torch.equal(o, ref[i]):
            bad += 1
            B = dbs[i][0].shape[0]
            Lw = B * DEPTH
            g, w = o.cpu().numpy().view(np.uint32), ref[i].cpu().numpy().view(np.uint32)
            seg = {"cos_scores": (0, Lw), "bm_scores": (Lw, 2 * Lw), "cos_docs": (2 * Lw, 3 * Lw), "bm_docs": (3 * Lw, 4 * Lw), "counts": (4 * Lw, 4 * Lw + 2 * B)}
            msg = []
            for k, (a, e) in seg.items():
                d = np.nonzero(g[a:e] != w[a:e])[0]
                if d.size:
                    parts[k] += 1
                    msg.append("%s: %d words, first at query %d rank %d (got %s want %s)" % (k, d.size, d[0] // DEPTH if k != "counts" else d[0], d[0] % DEPTH, g[a + d[0]], w[a + d[0]]))
            print("round %d batch %d (B=%d, lane %d): %s" % (rnd, i, B, i % lanes, "; ".join(msg)))
            if bad <= 8:
                d = np.nonzero(g[2 * Lw:3 * Lw] != w[2 * Lw:3 * Lw])[0]
                if d.size == 0:
                    d = np.nonzero(g[0:Lw] != w[0:Lw])[0]
                qx, r0 = int(d[0] // DEPTH), int(d[0] % DEPTH)
                qvec = batches[i][0][qx].astype(np.float64)
                lo, hi = max(0, r0 - 2), min(DEPTH, r0 + 10)
                gd, wd = g[2 * Lw + qx * DEPTH + lo:2 * Lw + qx * DEPTH + hi], w[2 * Lw + qx * DEPTH + lo:2 * Lw + qx * DEPTH + hi]
                gsc, wsc = g[qx * DEPTH + lo:qx * DEPTH + hi].view(np.float32), w[qx * DEPTH + lo:qx * DEPTH + hi].view(np.float32)
                for k in range(hi - lo):
                    ex_g = float(rows[int(gd[k]) - 700].astype(np.float64) @ qvec)
                    ex_w = float(rows[int(wd[k]) - 700].astype(np.float64) @ qvec)
                    print("   rank %3d  got doc %6d score %.9f (exact %.9f)   want doc %6d score %.9f (exact %.9f)" % (lo + k, gd[k], gsc[k], ex_g, wd[k], wsc[k], ex_w))
                # whose score is the wrong one?  same row against another query of the batch, or another row against this query
                for k in range(hi - lo):
                    ex_g = float(rows[int(gd[k]) - 700].astype(np.float64) @ qvec)
                    if abs(ex_g - float(gsc[k])) > 1e-5:
                        allq = batches[i][0].astype(np.float64) @ rows[int(gd[k]) - 700].astype(np.float64)
                        allr = rows.astype(np.float64) @ qvec
                        mq = np.nonzero(np.abs(allq - float(gsc[k])) < 2e-7)[0]
                        mr = np.nonzero(np.abs(allr - float(gsc[k])) < 2e-7)[0]
                        other = []
                        for bi, bb in enumerate(batches):
                            aq = bb[0].astype(np.float64) @ rows[int(gd[k]) - 700].astype(np.float64)
                            for qq in np.nonzero(np.abs(aq - float(gsc[k])) < 2e-7)[0]:
                                other.append((bi, int(qq)))
                        print("   WRONG score %.9f for doc %d query %d: equals this row x query %s ; equals row %s x this query ; equals this row x (batch, query) %s" % (float(gsc[k]), gd[k], qx, mq.tolist(), (mr + 700).tolist(), other))
                gset, wset = set(g[2 * Lw + qx * DEPTH:2 * Lw + (qx + 1) * DEPTH].tolist()), set(w[2 * Lw + qx * DEPTH:2 * Lw + (qx + 1) * DEPTH].tolist())
                print("   docs only in got:", sorted(gset - wset), " only in want:", sorted(wset - gset))
print("mismatching batches:", bad, parts)
