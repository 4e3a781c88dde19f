#!/bin/bash
# whole GPU suite + the per-batch latency table
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python __graft_entry__.py > $O/r02_build.log 2>&1 || { tail -20 $O/r02_build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r02_gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -8 $O/r02_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_host_entry.py > $O/r02_host_entry_latency.jsonl 2> $O/r02_host_entry_latency.err; echo "latency rc=$?"
grep '"batch_variants": 200,' $O/r02_host_entry_latency.jsonl | cut -c1-330
