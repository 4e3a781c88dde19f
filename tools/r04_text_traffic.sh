#!/bin/bash
# HBM traffic of the tokenizer on a window of text the bgzip decoder left on the device: with the decoder's tile records
# against the ordinary two-sweep call (tools/bench_tokenize_tiles.py).  Counter passes of their own, kernel trace only.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
N=${1:-10000}; LN=${2:-16000}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_tokenize_tiles.py $N $LN both > $O/r04_tok_tiles_$N.json 2> $O/r04_tok_tiles_$N.err || { tail -3 $O/r04_tok_tiles_$N.err; exit 1; }
for mode in tiles plain; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/r04_tt_${N}_${mode}_$c -o pmc --output-format csv -- python3 $R/tools/bench_tokenize_tiles.py $N $LN $mode > $O/r04_tt_${N}_${mode}_$c.json 2> $O/r04_tt_${N}_${mode}_$c.err || { tail -3 $O/r04_tt_${N}_${mode}_$c.err; exit 1; }
  done
  python3 $R/tools/pmc_tok_tiles.py $O/r04_tt_${N}_${mode}_FETCH_SIZE.json $mode $(find $O/r04_tt_${N}_${mode}_FETCH_SIZE $O/r04_tt_${N}_${mode}_WRITE_SIZE -name '*counter_collection.csv') > $O/r04_text_traffic_${N}_$mode.json || exit 1
  python3 -c "import json;d=json.load(open('$O/r04_text_traffic_${N}_$mode.json'));print({k:d[k] for k in ('mode','read_over_text_plus_matrix','read_plus_write_over_text_plus_matrix','tokenizer_kernels_GB_per_call')})"
done
rocprofv3 --kernel-trace --stats -d $O/r04_tt_${N}_stats -o st --output-format csv -- python3 $R/tools/bench_tokenize_tiles.py $N $LN both > $O/r04_tt_${N}_stats.json 2> $O/r04_tt_${N}_stats.err || exit 1
cat $O/r04_tok_tiles_$N.json
