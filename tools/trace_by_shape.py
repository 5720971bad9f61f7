"""Aggregate a rocprofv3 kernel trace by (kernel, grid, workgroup): calls, total / average duration -- the per-shape view the
--stats summary does not give.   python tools/trace_by_shape.py <rocprof dir> <out.csv> [steps]"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import demangle

d, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
agg = defaultdict(lambda: [0, 0])
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = demangle(r["Kernel_Name"]).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:110]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", "") + ("x" + r["Grid_Size_Z"] if r.get("Grid_Size_Z", "1") not in ("", "1") else ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""))
        a = agg[key]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
with open(out, "w") as fo:
    fo.write("kernel,grid_x,grid_y,wg_x,lds,calls,total_us,avg_us,pct,us_per_step\n")
    for (n, gx, gy, wx, lds), (c, t) in rows[:120]:
        fo.write(f"\"{n}\",{gx},{gy},{wx},{lds},{c},{t / 1e3:.1f},{t / c / 1e3:.1f},{100 * t / tot:.2f},{t / 1e3 / steps:.1f}\n")
print(f"{len(rows)} (kernel, launch shape) groups, {tot / 1e6:.1f} ms total -> {out}")
