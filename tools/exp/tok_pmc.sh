#!/bin/bash
# PMC passes over the tile-parallel tokenizer (tools/bench_tokenize.py 10000 16000)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$O/tok_pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d $d -o pmc --output-format csv -- python3 $R/tools/bench_tokenize.py 10000 16000 > $d.json 2> $d.err || { tail -3 $d.err; echo "pass $c failed"; }
done
python3 - $O <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
tot = {}
for path in glob.glob(O + "/tok_pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("hpgv::", "")
        if "tok" not in k: continue
        tot.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in tot.items()}
print(json.dumps(out, indent=1))
PY
