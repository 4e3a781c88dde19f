#!/bin/bash
# PMC passes over the one-pass stats kernel of the text path (tools/bench_text_entry.py 10000 16000 stats)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/sa_pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --pmc $c -d $d -o pmc --output-format csv -- python3 $R/tools/bench_text_entry.py 10000 16000 stats 3 > $d.json 2> $d.err || { tail -3 $d.err; echo "pass $c failed"; }
done
python3 - $O <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
tot = {}
for path in glob.glob(O + "/sa_pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("hpgv::", "")
        if "stats_all" not in k: continue
        tot.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
print(json.dumps({c: sum(v) / len(v) for c, v in tot.items()}, indent=1))
for path in glob.glob(O + "/sa_pmc_SQ_INSTS_VALU*/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "stats_all" in row["Name"]: print("k_stats_all avg ns", row["AverageNs"], "calls", row["Calls"])
PY
