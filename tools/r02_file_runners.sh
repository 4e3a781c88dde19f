#!/bin/bash
# Round-2 evidence for the file runners: 8 GB / 1.7 GB / 32 GB of VCF text, plain and bgzip (median of three runs that
# follow one another at once; *_runs_apart: 50 ms of idle time before each timed run -- the first H2D copies of a run are
# faster then, see profiles/README.md).  bgzip files of zlib level 6 (bgzip's default) and, *_level1, of level 1 (what
# round 1 measured: 2.3 times the symbols per byte of text).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
run() { out=$1; shift; "$@" > $O/$out.json 2> $O/$out.err || { tail -3 $O/$out.err; exit 1; }; tail -1 $O/$out.json; }
run r02_file_runner_10k_samples python3 tools/bench_file_runner.py 10000 200000 plain,bgzf 64 "HPGV_BGZF_HOST_TABLE=1|HPGV_ENGINE_THREADS=2"
BENCH_PAUSE_S=0.05 run r02_file_runner_10k_samples_runs_apart python3 tools/bench_file_runner.py 10000 200000 bgzf 64
BENCH_BGZF_LEVEL=1 run r02_file_runner_10k_samples_level1 python3 tools/bench_file_runner.py 10000 200000 bgzf 64
run r02_file_runner_200_samples python3 tools/bench_file_runner.py 200 2000000 plain,bgzf 64
HPGV_RUN_TRACE=1 python3 tools/bench_file_runner.py 40000 200000 bgzf 64 "HPGV_BGZF_HOST_TABLE=1|HPGV_INFLATE_WAVE=2|HPGV_INFLATE_WAVE=0|HPGV_NO_GPU_INFLATE=1" > $O/r02_file_runner_40k_samples_bgzf.log 2>&1 || { tail -3 $O/r02_file_runner_40k_samples_bgzf.log; exit 1; }
tail -1 $O/r02_file_runner_40k_samples_bgzf.log > $O/r02_file_runner_40k_samples_bgzf.json
cat $O/r02_file_runner_40k_samples_bgzf.json
BENCH_BGZF_LEVEL=1 run r02_file_runner_40k_samples_bgzf_level1 python3 tools/bench_file_runner.py 40000 200000 bgzf 64 "HPGV_INFLATE_WAVE=2|HPGV_INFLATE_WAVE=0"
