#!/usr/bin/env python3
"""Print a steady-state window of a kernel trace of tools/r05_pipeline_probe.py with lanes side by side (stream id per line),
and the GPU-busy fraction of the window.  Usage: r05_lane_trace.py <trace dir> <n_lanes-run index: 0,1,2 for 1,2,3 lanes>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rr = [i for i, r in enumerate(rows) if "rrf_kernel" in r["Kernel_Name"]]
# the probe runs: 4 ref + steps serial, then per lane count: 8 + 3*steps pipelined batches.  Take a window in the middle of the chosen run.
steps = (len(rr) - 4 - 3 * 8) // 10
start = 4 + steps + which * (8 + 3 * steps) + 8 + steps + steps // 2
a, b = rr[start], rr[start + 4]
t0 = int(rows[a]["Start_Timestamp"])
busy = []
for r in rows[a:b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    busy.append((s, e))
    print("%8.1f +%7.1f s=%-3s q=%s %s" % (s, e - s, r["Stream_Id"], r["Queue_Id"], r["Kernel_Name"].split("(")[0][-34:]))
busy.sort()
cov, end = 0.0, busy[0][0]
for s, e in busy:
    if e > end:
        cov += e - max(s, end); end = e
print("window %.1f us, some kernel running %.1f us (%.0f %%)" % (busy[-1][1] - busy[0][0], cov, 100 * cov / (busy[-1][1] - busy[0][0])))
