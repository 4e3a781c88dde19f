#!/bin/bash
# timeline of one text-entry call: kernels, copies and the HIP calls around them (rocprofv3 traces, no counters)
# tools/exp/trace_text_entry.sh <assoc|tdt|stats>  ->  gpurun_out/r03/trace_<tool>_*.csv
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --hip-trace --memory-copy-trace -d $O/trace_$1 -o t --output-format csv -- python3 $R/tools/bench_text_entry.py 10000 16000 $1 3 > $O/trace_$1.out 2> $O/trace_$1.err ) || { tail -5 $O/trace_$1.err; exit 1; }
find $O/trace_$1 -name "*.csv" | head
