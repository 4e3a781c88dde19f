#!/bin/bash
# Regenerates the evidence under profiles/ on the GPU box (run through gpurun from the repository root):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r01'
# Writes into gpurun_out/<tag>_*; copy what is to be kept into profiles/.
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${TAG}_bench_c2.json 2> $O/${TAG}_bench_c2.err || exit 1
for w in m8 c3 c4 c5; do
    python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err || exit 1
done
python tools/bench_epistasis.py 16384 10000 10 --cpu > $O/${TAG}_epi_bench_16k.json 2> $O/${TAG}_epi.err || exit 1
python tools/bench_epistasis.py 16384 10000 5 > $O/${TAG}_epi_bench_16k_5folds.json 2>> $O/${TAG}_epi.err || exit 1
python tools/bench_epistasis.py 8192 10000 10 > $O/${TAG}_epi_bench_8k.json 2>> $O/${TAG}_epi.err || exit 1
python tools/bench_epistasis.py 32768 10000 10 > $O/${TAG}_epi_bench_32k.json 2>> $O/${TAG}_epi.err || exit 1
python tools/bench_epistasis.py 16384 10000 10 --complete > $O/${TAG}_epi_bench_16k_complete.json 2>> $O/${TAG}_epi.err || exit 1
python tools/bench_epistasis.py 16384 10000 5 --complete > $O/${TAG}_epi_bench_16k_complete_5folds.json 2>> $O/${TAG}_epi.err || exit 1
for v in 512 1024 2048; do
    python tools/bench_epistasis.py $v 10000 10 --order=3 > $O/${TAG}_epi3_bench_$v.json 2>> $O/${TAG}_epi.err || exit 1
done
python tools/bench_epistasis.py 1024 10000 10 --order=3 --option=epi_triples_1pass=0 > $O/${TAG}_epi3_bench_1024_two_pass.json 2>> $O/${TAG}_epi.err || exit 1
python tools/bench_file_runner.py 10000 200000 plain,bgzf 64 > $O/${TAG}_file_runner_10k_samples.json 2> $O/${TAG}_fr.err || exit 1
python tools/bench_file_runner.py 200 2000000 plain,bgzf 64 > $O/${TAG}_file_runner_200_samples.json 2>> $O/${TAG}_fr.err || exit 1
python tools/bench_file_runner.py 40000 200000 bgzf 64 > $O/${TAG}_file_runner_40k_samples_bgzf.json 2>> $O/${TAG}_fr.err || exit 1
HPGV_NO_GPU_INFLATE=1 python tools/bench_file_runner.py 40000 200000 bgzf 64 > $O/${TAG}_file_runner_40k_samples_bgzf_cpu_inflate.json 2>> $O/${TAG}_fr.err || exit 1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_file_runner -o fr --output-format csv -- python3 tools/bench_file_runner.py 10000 200000 plain,bgzf 64 > $O/${TAG}_file_runner_under_rocprof.json 2>> $O/${TAG}_fr.err || exit 1
python tools/bench_inflate.py 125000 6 > $O/${TAG}_inflate_gpu_125k_blocks.json 2>> $O/${TAG}_fr.err || exit 1
python tools/bench_tokenize.py 10000 20000 > $O/${TAG}_tokenizer_10k_samples.json 2>> $O/${TAG}_fr.err || exit 1
# rocprofv3: kernel trace + stats of the headline command and of the epistasis scan (the program itself after --)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_c2 -o c2 --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_c2_under_rocprof.json 2> $O/${TAG}_prof_c2.err || exit 1
for w in m8 c3 c4; do
    rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_$w -o $w --output-format csv -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_${w}_under_rocprof.json 2> $O/${TAG}_prof_$w.err || exit 1
done
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_epi -o epi --output-format csv -- python3 $R/tools/bench_epistasis.py 16384 10000 10 > $O/${TAG}_epi_under_rocprof.json 2> $O/${TAG}_prof_epi.err || exit 1
echo refreshed
