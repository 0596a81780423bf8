#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV: per-kernel launch count / avg / total, and the last step's sequence."""
import csv
import glob
import sys
from collections import OrderedDict

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
agg = OrderedDict()
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-60:]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += d
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-62s calls=%5d avg_us=%9.1f total_ms=%8.2f" % (n, c, t / c, t / 1e3))
if len(sys.argv) > 2:
    tail = rows[-int(sys.argv[2]):]
    t0 = int(tail[0]["Start_Timestamp"])
    for r in tail:
        print("%9.1f +%8.1f us  %s grid=%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"]))
