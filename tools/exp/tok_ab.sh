cd $GRAFT_REPO_ROOT
for tt in 1 0; do
  export HPGV_TOKENIZER_TILES=$tt
  echo "tokenizer_tiles=$tt"
  python tools/bench_tokenize.py 10000 16000 | cut -c1-300
  python tools/bench_tokenize.py 200 800000 | cut -c1-300
  python tools/bench_text_entry.py 10000 16000 stats 5 | cut -c1-300
  python tools/bench_text_entry.py 10000 16000 assoc 5 | cut -c1-300
done
cd /tmp && export TMPDIR=/tmp
HPGV_TOKENIZER_TILES=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tok2 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_tokenize.py 10000 16000 > /dev/null 2>&1
grep -h "k_tok" $GRAFT_REPO_ROOT/gpurun_out/tok2/*kernel_stats.csv | awk -F'","' '{print $1, $2, $4}' | cut -c1-200
