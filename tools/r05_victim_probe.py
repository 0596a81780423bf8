#!/usr/bin/env python3
"""Round 5, the co-residency finding: a self-checking victim (tools/r05_victim.hip: small waves reading a buffer of known words
with ordinary loads) runs on its own stream while `lanes` other streams run searches through views of one index.  Wrong words
seen by the victim = a kernel of the library, resident beside it on a CU, disturbs ordinary loads.
  python tools/r05_victim_probe.py LANES ROUNDS [stream|copy|exact] [d768] [dots0..dots3]     (OI_LIB=ablation_<tag> for variant builds)
dotsF: the victim is pf_rescore_kernel's loop over small-integer rows (exact, known sums), flavor F (tools/r05_victim.hip)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import openintel_amd as oi
import _ablation  # noqa: F401
from openintel_amd import _lib
from test_gpu_pipeline import _case, _index

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
DIM = 768 if "d768" in sys.argv else 384
DEPTH = 200
vic = C.CDLL(os.path.join(ROOT, "tools", "r05_victim.so"))
vic.victim_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
vic.victim_launch.restype = C.c_int
vic.victim_dots_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                   C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
vic.victim_dots_launch.restype = C.c_int
FLAVOR = next((int(a[4:]) for a in sys.argv if a.startswith("dots")), None)

dev = torch.device("cuda:0")
rows, terms, offs, batches = _case(dim=DIM)
ctx = oi.HipContext(0)
mode = _lib.OI_COSINE_EXACT if "exact" in sys.argv else _lib.OI_COSINE_SCREEN_STREAM if "stream" in sys.argv else _lib.OI_COSINE_SCREEN
ctx.set_cosine_mode(mode)
idx = _index(ctx, rows, terms, offs)
dbs = [[torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) for x in b] for b in batches]
N_ROWS = 1 << 17                                                      # 128K rows x 2 KiB = 256 MiB of known words
i64 = torch.arange(N_ROWS * 512, dtype=torch.int64, device=dev)
buf = ((i64 * 2654435761 + 12345) & 0xFFFFFFFF).to(torch.int64)
buf = torch.where(buf >= 2 ** 31, buf - 2 ** 32, buf).to(torch.int32)
del i64
err = torch.zeros(2 + 4 * 64, dtype=torch.int32, device=dev)
if FLAVOR is not None:
    g = torch.Generator(device="cpu").manual_seed(3)
    VN = 60001
    vrows = torch.randint(0, 13, (VN, DIM), generator=g).to(torch.float32)
    vq = torch.randint(0, 5, (64, DIM), generator=g).to(torch.float32)
    vexp = (vq.double() @ vrows.double().T).to(torch.float32).contiguous().to(dev)   # exact: every sum is an integer < 2^24
    vrows, vq = vrows.to(dev), vq.to(dev)
    flog = torch.zeros(8 * 200, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
ref = [idx.search_lists_packed(*b, depth=DEPTH).clone() for b in dbs]
torch.cuda.synchronize()
L = []
for l in range(lanes):
    c = oi.HipContext.like(ctx)
    st = torch.cuda.Stream(device=dev)
    c.set_stream(st)
    c.set_overlap(False)
    L.append((idx.view(c), st, c))
vst = torch.cuda.Stream(device=dev)
seed, bad, launches = 1, 0, 0
for rnd in range(rounds):
    outs = []
    for i, b in enumerate(dbs):
        v, st, c = L[i % lanes]
        with torch.cuda.stream(st):
            outs.append(v.search_lists_packed(*b, depth=DEPTH))
        for _ in range(2):
            if FLAVOR is None:
                rc = vic.victim_launch(C.c_void_p(vst.cuda_stream), C.c_void_p(buf.data_ptr()), N_ROWS, seed, 1024, 8, C.c_void_p(err.data_ptr()))
            else:
                rc = vic.victim_dots_launch(C.c_void_p(vst.cuda_stream), FLAVOR, C.c_void_p(vrows.data_ptr()), DIM, VN, C.c_void_p(vq.data_ptr()), 64,
                                            C.c_void_p(vexp.data_ptr()), seed, 64, 2, C.c_void_p(err.data_ptr()), C.c_void_p(flog.data_ptr()))
            assert rc == 0, rc
            seed += 1; launches += 1
    torch.cuda.synchronize()
    bad += sum(1 for i, o in enumerate(outs) if not torch.equal(o, ref[i]))
e = err.cpu().numpy().view(np.uint32)
if FLAVOR is not None:
    print("victim_dots flavor %d, launches %d (64 x 64 blocks x 4 waves x 2 trips x 4 rows): wrong sums %d ; search batches whose lists differ: %d of %d"
          % (FLAVOR, launches, e[0], bad, rounds * len(dbs)))
    fl = flog.cpu().numpy()
    vr, vqq = vrows.cpu().numpy().astype(np.float64), vq.cpu().numpy().astype(np.float64)
    for s_ in range(min(int(e[1]), 8)):
        ent = fl[s_ * 200:(s_ + 1) * 200]
        q_, r_ = int(ent[0]), int(ent[1])
        hw = int(ent[4:5].view(np.uint32)[0])
        part = ent[8:72].astype(np.float64)
        prod = (vr[r_] * vqq[q_]).reshape(-1, 4).sum(1)                    # per float4
        wantp = np.zeros(64)
        for v in range(DIM // 4):
            wantp[v % 64] += prod[v]
        dl = np.nonzero(part != wantp)[0]
        print("  q %d row %d: got %.1f want %.1f (u %d trip %d, cu %d se %d hw %08x) ; lanes whose partial sum is wrong: %s" %
              (q_, r_, ent[2], ent[3], int(ent[5]), int(ent[6]), (hw >> 8) & 15, (hw >> 13) & 7, hw,
               [(int(l), float(part[l]), float(wantp[l])) for l in dl[:8]]))
        px, py = ent[72:136].astype(np.float64), ent[136:200].astype(np.float64)
        wx, wy = np.zeros(64), np.zeros(64)
        for v in range(DIM // 4):
            wx[v % 64] += vr[r_][4 * v:4 * v + 4].sum()
            wy[v % 64] += vqq[q_][4 * v:4 * v + 4].sum()
        print("     all wrong lanes %s ; lanes whose ROW-element sum is wrong %s ; lanes whose QUERY-element sum is wrong %s"
              % (dl.tolist(), np.nonzero(px != wx)[0].tolist(), np.nonzero(py != wy)[0].tolist()))
        for l in dl[:3]:
            print("     lane %d: row words %s query words %s ; dot got %.0f want %.0f ; row sum got %.0f want %.0f ; query sum got %.0f want %.0f"
                  % (l, vr[r_][4 * l:4 * l + 4].tolist(), vqq[q_][4 * l:4 * l + 4].tolist(), part[l], wantp[l], px[l], wx[l], py[l], wy[l]))
    sys.exit(0)
print("victim launches %d (1024 blocks x 4 waves x 8 rows of 2 KiB each): wrong words %d ; search batches whose lists differ: %d of %d"
      % (launches, e[0], bad, rounds * len(dbs)))
for s in range(min(int(e[1]), 64)):
    i, got, want, hw = e[2 + 4 * s:6 + 4 * s]
    inv = ((int(got) - 12345) * pow(2654435761, -1, 2 ** 32)) % 2 ** 32       # is the wrong word another index's word?
    print("  word %10d (row %6d word %3d): got %08x want %08x ; got = word_of(%d) ; f32 %.6g ; hw_id cu %d se %d xcc? %08x"
          % (i, i // 512, i % 512, got, want, inv, np.array([got], np.uint32).view(np.float32)[0], (hw >> 8) & 15, (hw >> 13) & 7, hw))
