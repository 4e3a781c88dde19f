#!/bin/bash
# Copies what tools/refresh_profiles.sh left under gpurun_out/ into profiles/ (run here, from the repository root).
TAG=${1:-r01}
O=gpurun_out
P=profiles
for w in c2 m8 c3 c4 c5; do cp $O/${TAG}_bench_$w.json $P/; done
cp $O/${TAG}_bench_c2_under_rocprof.json $P/
cp $O/${TAG}_prof_c2/c2_kernel_stats.csv $P/${TAG}_bench_c2_kernel_stats.csv
for w in m8 c3 c4; do
    [ -f $O/${TAG}_prof_$w/${w}_kernel_stats.csv ] && cp $O/${TAG}_prof_$w/${w}_kernel_stats.csv $P/${TAG}_bench_${w}_kernel_stats.csv
done
cp $O/${TAG}_prof_epi/epi_kernel_stats.csv $P/${TAG}_epi_pairs_kernel_stats_16k.csv
for f in epi_bench_16k epi_bench_16k_5folds epi_bench_8k epi_bench_32k epi_bench_16k_complete epi_bench_16k_complete_5folds \
         epi3_bench_512 epi3_bench_1024 epi3_bench_2048 epi3_bench_1024_two_pass file_runner_10k_samples file_runner_200_samples \
         file_runner_40k_samples_bgzf file_runner_40k_samples_bgzf_cpu_inflate file_runner_under_rocprof inflate_gpu_125k_blocks tokenizer_10k_samples; do
    [ -f $O/${TAG}_$f.json ] && cp $O/${TAG}_$f.json $P/
done
[ -f $O/${TAG}_prof_file_runner/fr_kernel_stats.csv ] && cp $O/${TAG}_prof_file_runner/fr_kernel_stats.csv $P/${TAG}_file_runner_kernel_stats.csv
echo collected
