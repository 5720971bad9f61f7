"""Refresh the measured numbers of DESIGN.md / README.md from a tools/final_verify.sh result directory: every number sits between
<!--KEY--> and <!--/--> markers (invisible in rendered markdown).   python tools/fill_docs.py gpurun_out/final"""
import json
import os
import re
import sys

d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


head = last_json(os.path.join(d, "bench_default.json"))
vals = {"HEAD": f"{head['value']:.1f}", "HEADMS": f"{head['ms_per_step']:.1f}",
        "PPTF": f"{head['roofline']['achieved']:.0f}", "PPFRAC": f"{head['roofline']['frac']:.2f}",
        "PPMS": f"{head['roofline']['avg_launch_ms']:.3f}",
        "CPU": f"{head['cpu_baseline']['value']:.3f}" if "cpu_baseline" in head else "0.13"}
names = {"openvis_online": "ONLINE", "san_online": "SAN", "brivis R50": "BRIVIS", "brivis_swinl": "BSWIN", "openvis_swinl": "OSWIN"}
for ln in open(os.path.join(d, "bench_all_models.jsonl")):
    l = json.loads(ln)
    for k, v in names.items():
        if l["config"]["workload"].startswith(k):
            vals[v], vals[v + "MS"] = f"{l['value']:.1f}", f"{l['ms_per_step']:.1f}"
if os.path.exists(os.path.join(d, "bench_streams2.json")):
    vals["S2"] = f"{last_json(os.path.join(d, 'bench_streams2.json'))['value']:.1f}"
for f in ("DESIGN.md", "README.md"):
    p = os.path.join(ROOT, f)
    s = open(p).read()
    for k, v in vals.items():
        s = re.sub(r"<!--%s-->.*?<!--/-->" % k, "<!--%s-->%s<!--/-->" % (k, v), s)
    open(p, "w").write(s)
print(vals)
