#!/bin/bash
# PMC passes over the Fisher p-pass (bench.py --workload c3): instructions per variant, issue / wait split, cache hits.
# Run through gpurun from the repository root:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/fisher_prof.sh r02'
# Separate runs, kernel trace only beside --pmc (the pool refuses --pmc next to the API trace domains).  Everything is
# built BEFORE the first profiler line: nothing may compile under the profiler's preload.
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R && python3 __graft_entry__.py > $O/${TAG}_fisher_build.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --configs none"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES -d $O/${TAG}_fi_pmc1 -o pmc1 --output-format csv -- $CMD > $O/${TAG}_fi_pmc1.json 2> $O/${TAG}_fi_pmc1.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/${TAG}_fi_pmc2 -o pmc2 --output-format csv -- $CMD > $O/${TAG}_fi_pmc2.json 2> $O/${TAG}_fi_pmc2.err || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_fi_pmc3 -o pmc3 --output-format csv -- $CMD > $O/${TAG}_fi_pmc3.json 2> $O/${TAG}_fi_pmc3.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_fi_pmc4 -o pmc4 --output-format csv -- $CMD > $O/${TAG}_fi_pmc4.json 2> $O/${TAG}_fi_pmc4.err || exit 1
python3 $R/tools/pmc_valu.py c3 $O/${TAG}_fi_pmc1.json $O/${TAG}_pmc_valu_c3.json $(find $O/${TAG}_fi_pmc1 $O/${TAG}_fi_pmc2 $O/${TAG}_fi_pmc3 $O/${TAG}_fi_pmc4 -name '*counter_collection.csv') || exit 1
echo done
