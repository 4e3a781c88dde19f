#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
# everything is built before the first profiler line: nothing may compile under the profiler's preload
(cd $R && python3 __graft_entry__.py > $R/gpurun_out/prof_build.log 2>&1) || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/epi3_pmc1 -o pmc1 --output-format csv -- python3 $R/tools/bench_epistasis.py ${EPI3_V:-1024} 10000 10 --order=3 > $R/gpurun_out/epi3_pmc1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/epi3_pmc2 -o pmc2 --output-format csv -- python3 $R/tools/bench_epistasis.py ${EPI3_V:-1024} 10000 10 --order=3 > $R/gpurun_out/epi3_pmc2.log 2>&1 || exit 1
echo done
