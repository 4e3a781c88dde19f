#!/bin/bash
# Round-2 evidence for the file runners: 8 GB / 0.33 GB / 32 GB of VCF text, plain and bgzip (median of three runs that
# follow one another at once; *_runs_apart: 50 ms of idle time before each timed run -- the first H2D copies of a run are
# faster then, see profiles/README.md)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 tools/bench_file_runner.py 10000 200000 plain,bgzf 64 "HPGV_BGZF_HOST_TABLE=1|HPGV_ENGINE_THREADS=2" > $O/r02_file_runner_10k_samples.json 2> $O/r02_fr10k.err || { tail -3 $O/r02_fr10k.err; exit 1; }
cat $O/r02_file_runner_10k_samples.json
BENCH_PAUSE_S=0.05 python3 tools/bench_file_runner.py 10000 200000 bgzf 64 > $O/r02_file_runner_10k_samples_runs_apart.json 2> $O/r02_fr10k.err || { tail -3 $O/r02_fr10k.err; exit 1; }
cat $O/r02_file_runner_10k_samples_runs_apart.json
python3 tools/bench_file_runner.py 200 2000000 plain,bgzf 64 > $O/r02_file_runner_200_samples.json 2> $O/r02_fr200.err || { tail -3 $O/r02_fr200.err; exit 1; }
cat $O/r02_file_runner_200_samples.json
HPGV_RUN_TRACE=1 python3 tools/bench_file_runner.py 40000 200000 bgzf 64 "HPGV_BGZF_HOST_TABLE=1|HPGV_INFLATE_WAVE=0|HPGV_NO_GPU_INFLATE=1" > $O/r02_file_runner_40k_samples_bgzf.log 2>&1 || { tail -3 $O/r02_file_runner_40k_samples_bgzf.log; exit 1; }
tail -1 $O/r02_file_runner_40k_samples_bgzf.log > $O/r02_file_runner_40k_samples_bgzf.json
cat $O/r02_file_runner_40k_samples_bgzf.json
