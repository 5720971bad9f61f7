"""Idle gaps of the GPU inside the timed steps of a rocprofv3 kernel trace: for every pair of consecutive kernels (by start time)
the gap between the end of everything launched so far and the next start, summed per (previous kernel -> next kernel) pair.
   python tools/trace_gaps.py <rocprof dir> [min_gap_us]"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short

d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])[:60]))
rows.sort()
busy_end, prev = rows[0][1], rows[0][2]
gaps = defaultdict(lambda: [0, 0.0])
total_gap = 0.0
for s, e, n in rows[1:]:
    g = (s - busy_end) / 1e3
    if g > 0:
        total_gap += g
        if g >= min_gap and g < 20000:
            a = gaps[(prev, n)]
            a[0] += 1
            a[1] += g
    if e > busy_end:
        busy_end, prev = e, n
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"{len(rows)} launches over {span / 1e3:.1f} ms; idle (all gaps < 20 ms) {sum(v[1] for v in gaps.values()) / 1e3:.1f} ms in gaps >= {min_gap} us")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t / 1e3:8.2f} ms  {c:5d} x {t / c:8.1f} us   {a}  ->  {b}")
busy = 0.0
cur_s, cur_e = rows[0][0], rows[0][1]
for s_, e_, _ in rows[1:]:
    if s_ > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
print(f"GPU busy (union of kernel intervals) {busy / 1e6:.1f} ms of {span / 1e3:.1f} ms = {busy / 1e3 / span:.3f}")
