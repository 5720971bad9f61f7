"""Summarise rocprofv3 --pmc passes (FETCH_SIZE pass, WRITE_SIZE pass) into per-kernel HBM traffic per launch.

usage: python tools/pmc_traffic.py <dir with *_counter_collection.csv (FETCH_SIZE)> <dir (WRITE_SIZE)> <out.json>
Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced read stream -> doubled here (stated in the output)."""
import collections
import csv
import glob
import os
import json
import sys


def load(d, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def demangle(k):
    """rocprofv3 leaves kernels with _Float16 parameters mangled (binutils' c++filt does not know DF16_ either): decode the
    name and its template arguments -- integers, bools and (possibly templated) struct names are all this library uses."""
    if not k.startswith("_ZN12_GLOBAL__N_1"):
        return k
    pos = len("_ZN12_GLOBAL__N_1")

    def ident(p):
        n = 0
        while k[p].isdigit():
            n, p = n * 10 + int(k[p]), p + 1
        return k[p:p + n], p + n

    def targs(p):                                  # at 'I'
        out, p = [], p + 1
        while k[p] != "E":
            if k[p] == "L":                        # literal: Li128E / Lb1E
                q = k.index("E", p)
                out.append(("true" if k[p + 2:q] == "1" else "false") if k[p + 1] == "b" else k[p + 2:q])
                p = q + 1
            elif k[p] == "N":                      # nested name N4ovis6DenseAI...EE
                p += 1
                parts = []
                while k[p] != "E":
                    if k[p] == "I":
                        a, p = targs(p)
                        parts[-1] += "<" + ", ".join(a) + ">"
                    else:
                        nm, p = ident(p)
                        parts.append(nm)
                out.append("::".join(parts))
                p += 1
            else:
                raise ValueError(k)
        return out, p + 1

    try:
        name, pos = ident(pos)
        if k[pos] == "I":
            a, pos = targs(pos)
            name += "<" + ", ".join(a) + ">"
        return name
    except (ValueError, IndexError):
        return k


def short(k):
    k = demangle(k)
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
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_digest                              # hash of the kernel sources these counters were collected on
    json.dump({"note": "FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes, "
                       "average per launch; separate --pmc passes", "commit": commit, "csrc_sha256": csrc_digest(), "kernels": res},
              open(out, "w"), indent=1)
    print("wrote", out, len(res), "kernels")
