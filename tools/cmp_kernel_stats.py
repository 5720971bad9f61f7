import csv, glob, sys
def load(d):
    f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(sys.argv[1]), load(sys.argv[2])
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print("total kernel ms per run:", ta / 1e6, tb / 1e6)
rows = []
for k in set(a) | set(b):
    da, db = a.get(k, (0, 0.0))[1], b.get(k, (0, 0.0))[1]
    rows.append((db - da, k, a.get(k, (0, 0))[0], b.get(k, (0, 0))[0], da, db))
for d, k, ca, cb, da, db in sorted(rows, key=lambda r: -abs(r[0]))[:14]:
    print(f"{d / 1e6:+8.2f} ms  calls {ca}/{cb}  {da / 1e6:8.2f} -> {db / 1e6:8.2f}  {k[:100]}")
