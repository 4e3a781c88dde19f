#!/bin/bash
# kernel-trace statistics of one command: tools/prof_kernels.sh <tag> <python script and args...>  ->  gpurun_out/r03/<tag>_kernel_stats.csv
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o p --output-format csv -- python3 $R/"$1" "${@:2}" > $O/${TAG}.out 2> $O/${TAG}.err ) || { tail -5 $O/${TAG}.err; exit 1; }
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $O/${TAG}_kernel_stats.csv
rm -rf $O/prof_$TAG
head -${HEAD:-12} $O/${TAG}_kernel_stats.csv | cut -c1-200
