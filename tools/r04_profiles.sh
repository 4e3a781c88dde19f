#!/bin/bash
# round 4: the epistasis bench lines and counters that profiles/r04_epi_* hold (run on the GPU box through gpurun)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
P=$O/r04_epi_pairs_mfma.jsonl; : > $P
for args in "16384 10000 10" "16384 10000 10 --option=epi_pairs_mfma=0" "16384 10000 5" "16384 10000 5 --option=epi_pairs_mfma=0" \
            "16384 10000 16" "16384 10000 16 --option=epi_pairs_mfma=0" "16384 10000 10 --unbalanced" "16384 10000 10 --unbalanced --option=epi_pairs_mfma=0" "16384 10000 10 --affected=4987" "16384 10000 10 --affected=4987 --option=epi_pairs_mfma=0" \
            "4096 100000 10" "4096 100000 10 --option=epi_pairs_mfma=0" "16384 10000 10 --complete" "16384 10000 10 --complete --option=epi_pairs_mfma=0" \
            "16384 10000 5 --complete" "16384 10000 5 --complete --option=epi_pairs_mfma=0"; do
  timeout -k 10 300 python3 tools/bench_epistasis.py $args >> $P || exit 1
done
T=$O/r04_epi3_bench.jsonl; : > $T
for args in "1024 10000 10 --order=3 --option=epi_triples_mfma=0" "1024 10000 5 --order=3 --option=epi_triples_mfma=0" "512 10000 10 --order=3 --unbalanced --option=epi_triples_mfma=0" \
            "1024 10000 10 --order=3 --option=epi_triples_mfma=0 --option=epi_triples_1pass=0"; do
  timeout -k 10 600 python3 tools/bench_epistasis.py $args >> $T || exit 1
done
T3=$O/r04_epi3_mfma_bench.jsonl; : > $T3
for args in "1024 10000 10 --order=3" "1024 10000 5 --order=3" "512 10000 10 --order=3 --unbalanced" "2048 10000 10 --order=3" "1024 100000 10 --order=3"; do
  timeout -k 10 600 python3 tools/bench_epistasis.py $args >> $T3 || exit 1
done
cat $T3 | cut -c1-220
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r04_epm_stats -o epm --output-format csv -- python3 $R/tools/bench_epistasis.py 16384 10000 10 > $O/r04_epm_stats.json 2> $O/r04_epm_stats.err || exit 1
rocprofv3 --kernel-trace --stats -d $O/r04_epm3_stats -o epm3 --output-format csv -- python3 $R/tools/bench_epistasis.py 1024 10000 10 --order=3 > $O/r04_epm3_stats.json 2> $O/r04_epm3_stats.err || exit 1
bash $R/tools/epm_prof.sh > $O/r04_epm_counters.txt 2>&1 || exit 1
EPI_ARGS="1024 10000 10 --order=3" bash $R/tools/epm_prof.sh > $O/r04_epm3_counters.txt 2>&1 || exit 1
tail -2 $O/r04_epm_counters.txt
cat $P | cut -c1-220
cat $T | cut -c1-220
