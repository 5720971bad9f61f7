"""Summarise rocprofv3 --pmc passes (FETCH_SIZE pass, WRITE_SIZE pass) into per-kernel HBM traffic per launch.

usage: python tools/pmc_traffic.py <dir with *_counter_collection.csv (FETCH_SIZE)> <dir (WRITE_SIZE)> <out.json>
Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced read stream -> doubled here (stated in the output)."""
import collections
import csv
import glob
import json
import sys


def load(d, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0][:90]


if __name__ == "__main__":
    fd, wd, out = sys.argv[1:4]
    commit = sys.argv[4] if len(sys.argv) > 4 else None        # the GPU box has no .git: the caller passes HEAD
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [])
        w = write.get(k, [])
        if not f and not w:
            continue
        fb = 2.0 * 1024 * sum(f) / max(len(f), 1)          # KiB -> B, x2 gfx950 correction
        wb = 1024 * sum(w) / max(len(w), 1)
        res[short(k)] = {"launches": max(len(f), len(w)), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                         "hbm_bytes_per_launch": round(fb + wb)}
    json.dump({"note": "FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes, "
                       "average per launch; separate --pmc passes", "commit": commit, "kernels": res}, open(out, "w"), indent=1)
    print("wrote", out, len(res), "kernels")
