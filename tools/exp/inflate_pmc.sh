#!/bin/bash
# PMC passes over the bgzip decoder (tools/bench_inflate.py: three launches of k_inflate_blocks over N blocks)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
N=${1:-125000}
TAG=${2:-inflate_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COUNTERS=${COUNTERS:-"FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU;SQ_INSTS_SALU;SQ_WAVES;TCC_HIT_sum TCC_MISS_sum;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;SQ_INSTS_LDS SQ_INSTS_FLAT"}
IFS=";" read -ra CLIST <<< "$COUNTERS"
for c in "${CLIST[@]}"; do
  d=$O/${TAG}_$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d $d -o pmc --output-format csv -- python3 $R/tools/bench_inflate.py $N 6 ${WAVE:-1} > $d.json 2> $d.err || { tail -3 $d.err; echo "pass $c failed"; }
done
python3 - $O $TAG <<'PY'
import csv, glob, json, sys
O, TAG = sys.argv[1:3]
tot = {}
for path in glob.glob(O + "/" + TAG + "_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "inflate" not in row["Kernel_Name"]: continue
        tot.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in tot.items()}      # per launch
out["launches_seen"] = {k: len(v) for k, v in tot.items()}
print(json.dumps(out))
open(O + "/" + TAG + ".json", "w").write(json.dumps(out) + "\n")
PY
