#!/usr/bin/env python3
"""Timeline of the LAST `window_ms` of a rocprofv3 kernel trace: every launch with its queue, start, duration and grid, and how
much of the window had a corpus pass (screen / bf16 / exact cosine kernel) running.
    python tools/trace_timeline.py <trace dir> [window_ms] [max rows]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 3e6
maxrows = int(sys.argv[3]) if len(sys.argv) > 3 else 400
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = max(int(r["End_Timestamp"]) for r in rows)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= end - win]
t0 = int(rows[0]["Start_Timestamp"])
big = lambda n: any(k in n for k in ("cosine_screen_filter", "cosine_bf16", "cosine_ksplit"))
iv = []
for r in rows[:maxrows]:
    n = r["Kernel_Name"].split("(")[0][-44:]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f +%8.1f us  q=%-3s %-44s grid=%s wg=%s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), n, r["Grid_Size_X"], r.get("Workgroup_Size_X", "?")))
for r in rows:
    if big(r["Kernel_Name"]):
        iv.append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0))
iv.sort()
cov, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            cov += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    cov += cur_e - cur_s
span = (max(int(r["End_Timestamp"]) for r in rows) - t0)
print("window %.3f ms, corpus-pass kernels cover %.3f ms = %.1f %%; sum of their durations %.3f ms (overlap of two passes %.3f ms)" % (
    span / 1e6, cov / 1e6, 100.0 * cov / span, sum(e - s for s, e in iv) / 1e6, (sum(e - s for s, e in iv) - cov) / 1e6))
