#!/bin/bash
# HBM traffic of the scan kernels from PMC counters, one workload per call (run through gpurun from the repository root):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_traffic.sh m8 r01'
# Two SEPARATE counter passes (kernel trace only beside --pmc); writes gpurun_out/<tag>_pmc_traffic_<workload>.json.
set -o pipefail
W=${1:-c2}
TAG=${2:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
(cd $R && python3 __graft_entry__.py > $O/${TAG}_pmc_build.log 2>&1) || exit 1   # built before the profiler starts
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/${TAG}_pmc_${W}_$c -o pmc --output-format csv -- python3 $R/bench.py --workload $W --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --configs none > $O/${TAG}_pmc_${W}_$c.json 2> $O/${TAG}_pmc_${W}_$c.err || exit 1
done
F=$(find $O/${TAG}_pmc_${W}_FETCH_SIZE -name '*counter_collection.csv' | head -1)
Wf=$(find $O/${TAG}_pmc_${W}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 $R/tools/pmc_traffic.py $W $F $Wf $O/${TAG}_pmc_${W}_FETCH_SIZE.json $O/${TAG}_pmc_traffic_$W.json
