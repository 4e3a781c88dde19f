#!/bin/bash
# Round-2 evidence for the bgzip path: both device decoders at three launch sizes, their PMC passes, and the file runners.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
(cd $R && python3 __graft_entry__.py > $O/r02_bgzf_build.log 2>&1) || exit 1
cd $R
: > $O/r02_inflate_decoders.jsonl
for mode in 2 0; do
  for n in 4096 32768 125000 500000; do
    timeout -k 10 200 python3 tools/bench_inflate.py $n 6 $mode >> $O/r02_inflate_decoders.jsonl || exit 1
  done
  timeout -k 10 200 python3 tools/bench_inflate.py 125000 1 $mode >> $O/r02_inflate_decoders.jsonl || exit 1
done
cat $O/r02_inflate_decoders.jsonl
C="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_SALU;SQ_INSTS_LDS SQ_INSTS_SMEM;SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;SQ_WAVES SQ_INSTS_BRANCH"
COUNTERS="$C" WAVE=2 bash tools/exp/inflate_pmc.sh 125000 r02_inflate_pmc_wave_per_block > /dev/null || exit 1
COUNTERS="$C" WAVE=0 bash tools/exp/inflate_pmc.sh 125000 r02_inflate_pmc_lane_per_block > /dev/null || exit 1
cat $O/r02_inflate_pmc_wave_per_block.json $O/r02_inflate_pmc_lane_per_block.json
cd $R
python3 tools/bench_file_runner.py 10000 200000 plain,bgzf 64 "HPGV_BGZF_HOST_TABLE=1" > $O/r02_file_runner_10k_samples.json 2> $O/r02_fr10k.err || { tail -3 $O/r02_fr10k.err; exit 1; }
cat $O/r02_file_runner_10k_samples.json
python3 tools/bench_file_runner.py 200 2000000 plain,bgzf 64 > $O/r02_file_runner_200_samples.json 2> $O/r02_fr200.err || { tail -3 $O/r02_fr200.err; exit 1; }
cat $O/r02_file_runner_200_samples.json
HPGV_RUN_TRACE=1 python3 tools/bench_file_runner.py 40000 200000 bgzf 64 "HPGV_BGZF_HOST_TABLE=1|HPGV_INFLATE_WAVE=0|HPGV_NO_GPU_INFLATE=1" > $O/r02_file_runner_40k_samples_bgzf.log 2>&1 || { tail -3 $O/r02_file_runner_40k_samples_bgzf.log; exit 1; }
tail -1 $O/r02_file_runner_40k_samples_bgzf.log > $O/r02_file_runner_40k_samples_bgzf.json
cat $O/r02_file_runner_40k_samples_bgzf.json
