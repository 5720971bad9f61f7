#!/bin/bash
# PMC pass of one python script on the GPU box:  tools/pmc_one.sh <out dir name> "<counters>" <script> [args]
#   -> gpurun_out/<name>/: the counter_collection csv summarised per kernel (sum over dispatches / dispatch count)
set -e
name="$1"; ctr="$2"; shift 2
R=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
script="$1"; shift
rc=0
# ($ctr is a space-separated counter list: word splitting intended; the script's own arguments are passed quoted)
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$R/pmc" -- python3 "$GRAFT_REPO_ROOT/$script" "$@" > "$R/stdout.txt" 2> "$R/stderr.txt" || rc=$?
cd "$GRAFT_REPO_ROOT"
if [ $rc -ne 0 ]; then echo "pmc_one.sh: rocprofv3 exited with status $rc (stderr tail follows)" >&2; tail -5 "$R/stderr.txt" >&2; fi
if ! find "$R/pmc" -name "*counter_collection.csv" | grep -q .; then echo "pmc_one.sh: no counter_collection.csv was written" >&2; exit ${rc:-1}; fi
python3 - "$R" <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(R + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        cnt[(k, row["Counter_Name"])] += 1
with open(R + "/summary.txt", "w") as o:
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write(f"   {c:32s} {v / max(cnt[(k, c)], 1):16.1f} per dispatch ({cnt[(k, c)]} dispatches)\n")
print(open(R + "/summary.txt").read())
PY
find "$R" -name "*.csv" -size +5M -delete
exit $rc
