#!/bin/bash
# round-3 evidence for profiles/: the default bench line (metric cohort + the attached configs), the rocprofv3 kernel stats of the
# metric workload's command, the PMC traffic passes, the per-workload lines, the text-entry traffic (both tokenizers).
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r03_profiles.sh'
set -o pipefail
TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 __graft_entry__.py > $O/${TAG}_build.log 2>&1 || exit 1
python3 bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || { tail -5 $O/${TAG}_bench_default.err; exit 1; }
echo "default bench done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_m -o m --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --configs none > $O/${TAG}_bench_m_under_rocprof.json 2> $O/${TAG}_prof_m.err ) || { tail -5 $O/${TAG}_prof_m.err; exit 1; }
cp $(find $O/${TAG}_prof_m -name '*kernel_stats.csv' | head -1) $O/${TAG}_bench_m_kernel_stats.csv
rm -rf $O/${TAG}_prof_m
echo "rocprof stats done"
STEPS=2 bash tools/pmc_traffic.sh m $TAG || exit 1
for w in c2 c4 stats; do STEPS=3 bash tools/pmc_traffic.sh $w $TAG || exit 1; done
echo "pmc traffic done"
cd $R
for w in c2 c3 c4 stats c5; do
    python3 bench.py --workload $w --steps 10 --warmup 2 > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err || { tail -5 $O/${TAG}_bench_$w.err; exit 1; }
done
echo "workloads done"
python3 tools/bench_tokenize.py 10000 16000 > $O/${TAG}_tokenizer_10k_samples.json 2>> $O/${TAG}_text.err || exit 1
python3 tools/bench_tokenize.py 200 800000 > $O/${TAG}_tokenizer_200_samples.json 2>> $O/${TAG}_text.err || exit 1
for tool in stats assoc; do python3 tools/bench_text_entry.py 10000 16000 $tool 5 > $O/${TAG}_text_entry_$tool.json 2>> $O/${TAG}_text.err || exit 1; done
find $O -maxdepth 1 -type d -name "${TAG}_pmc_*" -exec rm -rf {} +
echo "all done"
