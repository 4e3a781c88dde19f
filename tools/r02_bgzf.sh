#!/bin/bash
# Round-2 evidence for the bgzip path: both device decoders at three launch sizes, their PMC passes, and the file runners.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
(cd $R && python3 __graft_entry__.py > $O/r02_bgzf_build.log 2>&1) || exit 1
cd $R
: > $O/r02_inflate_decoders.jsonl
for mode in 4 2 0; do
  for n in 4096 32768 125000 500000; do
    timeout -k 10 200 python3 tools/bench_inflate.py $n 6 $mode >> $O/r02_inflate_decoders.jsonl || exit 1
  done
  timeout -k 10 200 python3 tools/bench_inflate.py 125000 1 $mode >> $O/r02_inflate_decoders.jsonl || exit 1
done
cat $O/r02_inflate_decoders.jsonl
C="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU;SQ_INSTS_LDS SQ_INSTS_SMEM;SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;SQ_WAVES SQ_INSTS_BRANCH"
COUNTERS="$C" WAVE=4 bash tools/exp/inflate_pmc.sh 125000 r02_inflate_pmc_wave_several_symbols > /dev/null || exit 1
COUNTERS="$C" WAVE=2 bash tools/exp/inflate_pmc.sh 125000 r02_inflate_pmc_wave_per_block > /dev/null || exit 1
COUNTERS="$C" WAVE=0 bash tools/exp/inflate_pmc.sh 125000 r02_inflate_pmc_lane_per_block > /dev/null || exit 1
cat $O/r02_inflate_pmc_wave_several_symbols.json $O/r02_inflate_pmc_wave_per_block.json $O/r02_inflate_pmc_lane_per_block.json
bash $R/tools/r02_file_runners.sh || exit 1
