"""Kernel time of ONE stage of one clip from a rocprofv3 kernel trace: the launches between the last launch of kernel <after> and
the first following launch of kernel <until>, for the last complete clip in the trace, grouped by kernel and launch shape.
   python tools/trace_segment.py <rocprof dir> <after substring> <until substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short

d, after, until = sys.argv[1:4]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", "")))
rows.sort()
ends = [i for i, r in enumerate(rows) if until in r[2]]
e = ends[-1]
while e > 0 and until in rows[e - 1][2]:
    e -= 1
s = max(i for i in range(e) if after in rows[i][2]) + 1
seg = rows[s:e]
agg = defaultdict(lambda: [0, 0])
for st, en, n, gx, gy in seg:
    a = agg[(n[:70], gx, gy)]
    a[0] += 1
    a[1] += en - st
busy = sum(v[1] for v in agg.values()) / 1e3
print(f"{len(seg)} launches, {busy:.0f} us of kernel time, wall {(seg[-1][1] - seg[0][0]) / 1e3:.0f} us (profiler launch overhead included)")
for (n, gx, gy), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / 1e3:8.1f} us {c:4d} x {t / c / 1e3:7.1f}  grid {gx}x{gy}  {n}")
