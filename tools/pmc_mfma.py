"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE pass into per-kernel MFMA pipe
utilisation.

usage: python tools/pmc_mfma.py <dir with *_counter_collection.csv (+ *_kernel_trace.csv)> <out.json>

SQ_VALU_MFMA_BUSY_CYCLES counts MFMA-pipe busy cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_f16,
/opt/skills/guides/MI355X_MICROARCH.md "PMC units"); the denominator is the kernel's own duration in the same pass
(dispatch timestamps) x 1024 SIMDs (256 CUs x 4) x the clock, reported for the 2.4 GHz peak clock (the clock the
2.5 PFLOP/s fp16 peak is quoted at) — so `mfma_util_at_peak_clock` is directly comparable with roofline.frac — and the
raw busy cycles are kept so that any other clock can be applied."""
import collections
import csv
import glob
import json
import sys


import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short            # noqa: E402  (demangles the names rocprofv3 leaves mangled)


if __name__ == "__main__":
    d, out = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)                    # kernel -> dispatch id -> ns
    trace = {}
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            trace[r.get("Dispatch_Id")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r.get("Start_Timestamp") and r.get("End_Timestamp") and float(r["End_Timestamp"]) > 0:
                dur[r["Kernel_Name"]][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            elif r.get("Dispatch_Id") in trace:
                dur[r["Kernel_Name"]][r["Dispatch_Id"]] = trace[r["Dispatch_Id"]]
    res = {}
    for k, c in acc.items():
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])
        if not busy or sum(busy) == 0:
            continue
        b = sum(busy) / len(busy)
        e = {"launches": len(busy), "mfma_busy_cycles_per_launch": round(b)}
        for name, key in (("GRBM_GUI_ACTIVE", "grbm_gui_active_per_launch"), ("SQ_BUSY_CU_CYCLES", "sq_busy_cu_cycles_per_launch")):
            if name in c:
                e[key] = round(sum(c[name]) / len(c[name]))
        if dur[k]:
            ns = sum(dur[k].values()) / len(dur[k])
            e["avg_launch_us_in_this_pass"] = round(ns / 1e3, 1)
            e["mfma_util_at_peak_clock"] = round(b / (ns * 2.4 * 1024), 4)
        res[short(k)] = e
    json.dump({"note": "MFMA pipe busy cycles (sum over SIMDs) per launch; util = busy / (launch duration x 2.4 GHz x 1024 "
                       "SIMDs); separate --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE + kernel trace)",
               "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"]))},
              open(out, "w"), indent=1)
    print("wrote", out, len(res), "kernels")
