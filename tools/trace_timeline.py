"""Per-launch timeline of ONE stage of the last complete clip in a rocprofv3 kernel trace: start offset, duration and the idle gap in
front of every launch between the last launch of kernel <after> and the first following launch of kernel <until>.
   python tools/trace_timeline.py <rocprof dir> <after substring> <until substring>"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short

d, after, until = sys.argv[1:4]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")),
                     r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
rows.sort()
ends = [i for i, r in enumerate(rows) if until in r[2]]
e = ends[-1]
while e > 0 and until in rows[e - 1][2]:
    e -= 1
s = max(i for i in range(e) if after in rows[i][2]) + 1
seg = rows[s:e]
t0, busy, gap_sum, prev_end = seg[0][0], 0, 0, rows[s - 1][1]
print(f"{len(seg)} launches, {after} -> {until}")
for st, en, n, gx, wg in seg:
    gap = max(0, st - prev_end)
    gap_sum += gap
    busy += en - st
    print(f"{(st - t0) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(en - st) / 1e3:7.1f} us  {n[:64]:64s} grid {gx} wg {wg}")
    prev_end = max(prev_end, en)
print(f"span {(seg[-1][1] - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle gaps {gap_sum / 1e3:.1f} us")
