#!/bin/bash
# PMC counters of one command (own pass: no --stats): tools/prof_pmc.sh <tag> "<C1 C2 ...>" <script> [args]  ->  gpurun_out/r03/<tag>_pmc.csv (mean per dispatch and kernel)
TAG=$1; CTRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $CTRS -d $O/pmc_$TAG -o p --output-format csv -- python3 $R/"$1" "${@:2}" > $O/${TAG}.out 2> $O/${TAG}.err ) || { tail -5 $O/${TAG}.err; exit 1; }
f=$(find $O/pmc_$TAG -name "*counter_collection.csv" | head -1)
python3 - "$f" "$O/${TAG}_pmc.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    d = (k, r["Dispatch_Id"])
    if d not in seen: seen.add(d); n[k] += 1
names = sorted({c for k in acc for c in acc[k]})
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["kernel", "dispatches"] + names)
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    w.writerow([k, n[k]] + ["%.0f" % (acc[k][c] / n[k]) for c in names])
PY
rm -rf $O/pmc_$TAG
head -${HEAD:-6} $O/${TAG}_pmc.csv
