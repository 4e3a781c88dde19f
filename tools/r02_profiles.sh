#!/bin/bash
# round-2 evidence for profiles/: the default bench line (metric cohort, tiled at N = 1), its rocprofv3 kernel stats,
# the PMC traffic passes of the same command, the Fisher PMC pass, and the other workloads' bench lines.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r02_profiles.sh r02'
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 __graft_entry__.py > $O/${TAG}_build.log 2>&1 || exit 1
python3 bench.py > $O/${TAG}_bench_m.json 2> $O/${TAG}_bench_m.err || { tail -5 $O/${TAG}_bench_m.err; exit 1; }
echo "default bench done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_m -o m --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_m_under_rocprof.json 2> $O/${TAG}_prof_m.err ) || { tail -5 $O/${TAG}_prof_m.err; exit 1; }
cp $(find $O/${TAG}_prof_m -name '*kernel_stats.csv' | head -1) $O/${TAG}_bench_m_kernel_stats.csv
echo "rocprof stats done"
STEPS=2 bash tools/pmc_traffic.sh m $TAG || exit 1
for w in c2 c4 stats; do STEPS=3 bash tools/pmc_traffic.sh $w $TAG || exit 1; done
echo "pmc traffic done"
bash tools/fisher_prof.sh $TAG || exit 1
for w in c2 c3 c4 stats c5; do
    python3 bench.py --workload $w --steps 10 --warmup 2 > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err || { tail -5 $O/${TAG}_bench_$w.err; exit 1; }
done
echo "all done"
# the text path: tokenizer alone (both forms), text entry points with PMC traffic, decoder, whole-file runs, per-batch latency
cd $R
python3 tools/bench_tokenize.py 10000 16000 > $O/${TAG}_tokenizer_10k_samples.json 2>> $O/${TAG}_text.err || exit 1
python3 tools/bench_tokenize.py 200 800000 > $O/${TAG}_tokenizer_200_samples.json 2>> $O/${TAG}_text.err || exit 1
HPGV_TOKENIZER_TILES=0 python3 tools/bench_tokenize.py 10000 16000 > $O/${TAG}_tokenizer_10k_samples_line_by_line.json 2>> $O/${TAG}_text.err || exit 1
HPGV_TOKENIZER_TILES=0 python3 tools/bench_tokenize.py 200 800000 > $O/${TAG}_tokenizer_200_samples_line_by_line.json 2>> $O/${TAG}_text.err || exit 1
bash tools/r02_text_traffic.sh > $O/${TAG}_text_traffic.log 2>&1 || { tail -5 $O/${TAG}_text_traffic.log; exit 1; }
cd $R
# the bgzip path (both decoders, their PMC passes, the file runners): tools/r02_bgzf.sh
bash $R/tools/r02_bgzf.sh > $O/${TAG}_bgzf.log 2>&1 || { tail -5 $O/${TAG}_bgzf.log; exit 1; }
python3 tools/bench_host_entry.py > $O/${TAG}_host_entry_latency.jsonl 2>> $O/${TAG}_text.err || exit 1
echo "text path done"
