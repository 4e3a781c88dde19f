#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python __graft_entry__.py > $O/r02_build.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -x -q > $O/r02_batch_tests.log 2>&1; echo "batch tests rc=$?"; tail -15 $O/r02_batch_tests.log
timeout -k 10 300 python tools/bench_host_entry.py > $O/r02_host_entry_latency.jsonl 2> $O/r02_host_entry_latency.err; echo "latency rc=$?"
cat $O/r02_host_entry_latency.jsonl | cut -c1-400
