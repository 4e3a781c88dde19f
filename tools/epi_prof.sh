#!/bin/bash
set -o pipefail
# rocprofv3 passes over the epistasis pair scan (run on the GPU box through gpurun): kernel trace + stats, then
# PMC counters in their own runs (kernel-trace only, as the pool requires)
R=${GRAFT_REPO_ROOT:-$(pwd)}
# everything is built before the first profiler line: nothing may compile under the profiler's preload
(cd $R && python3 __graft_entry__.py > $R/gpurun_out/prof_build.log 2>&1) || exit 1
cd /tmp && export TMPDIR=/tmp
ARGS="${EPI_ARGS:-16384 10000 10}"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/epi_stats -o epi --output-format csv -- python3 $R/tools/bench_epistasis.py $ARGS > $R/gpurun_out/epi_stats.json 2> $R/gpurun_out/epi_stats.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d $R/gpurun_out/epi_pmc1 -o pmc1 --output-format csv -- python3 $R/tools/bench_epistasis.py $ARGS > $R/gpurun_out/epi_pmc1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/epi_pmc2 -o pmc2 --output-format csv -- python3 $R/tools/bench_epistasis.py $ARGS > $R/gpurun_out/epi_pmc2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY -d $R/gpurun_out/epi_pmc3 -o pmc3 --output-format csv -- python3 $R/tools/bench_epistasis.py $ARGS > $R/gpurun_out/epi_pmc3.log 2>&1 || exit 1
ls $R/gpurun_out/epi_stats
