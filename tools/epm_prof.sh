#!/bin/bash
# counters of the matrix-core pair scan (k_epi_pairs_mfma): passes of their own, kernel trace only
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="${EPI_ARGS:-16384 10000 10}"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM -d $O/epm_pmc1 -o pmc1 --output-format csv -- python3 $R/tools/bench_epistasis.py $ARGS > $O/epm_pmc1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $O/epm_pmc2 -o pmc2 --output-format csv -- python3 $R/tools/bench_epistasis.py $ARGS > $O/epm_pmc2.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
for d in ("epm_pmc1","epm_pmc2"):
    acc={}
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_epi_" not in r["Kernel_Name"] or "k_epi_planes" in r["Kernel_Name"] or "k_epi_marg" in r["Kernel_Name"]: continue
            a=acc.setdefault(r["Counter_Name"],[0.0,0]); a[0]+=float(r["Counter_Value"]); a[1]+=1
    print(d,{k:(round(v[0]/v[1]),v[1]) for k,v in acc.items()})
PY
