#!/bin/bash
# HBM traffic of one text-entry call, one-pass kernels (default) against the kernel chains (HPGV_BATCH_FUSED=0)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
(cd $R && python3 __graft_entry__.py > $O/r02_tt_build.log 2>&1) || exit 1
cd /tmp && export TMPDIR=/tmp
for tool in stats assoc; do
  for fused in 1 0; do
    export HPGV_BATCH_FUSED=$fused
    python3 $R/tools/bench_text_entry.py 10000 16000 $tool 5 > $O/r02_text_${tool}_f$fused.json 2> $O/r02_text_${tool}_f$fused.err || { tail -3 $O/r02_text_${tool}_f$fused.err; exit 1; }
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c -d $O/r02_tt_${tool}_f${fused}_$c -o pmc --output-format csv -- python3 $R/tools/bench_text_entry.py 10000 16000 $tool 2 > $O/r02_tt_${tool}_f${fused}_$c.json 2> $O/r02_tt_${tool}_f${fused}_$c.err || exit 1
    done
    # (the profiled run's own timing is not quoted: ms_per_call of the unprofiled run above replaces it when the profiles are filed)
    python3 $R/tools/pmc_sum.py $O/r02_tt_${tool}_f${fused}_FETCH_SIZE.json $(find $O/r02_tt_${tool}_f${fused}_FETCH_SIZE $O/r02_tt_${tool}_f${fused}_WRITE_SIZE -name '*counter_collection.csv') > $O/r02_text_traffic_${tool}_f$fused.json || exit 1
    cat $O/r02_text_${tool}_f$fused.json
    python3 -c "import json;d=json.load(open('$O/r02_text_traffic_${tool}_f$fused.json'));print({k:d[k] for k in ('tool','fused','read_over_text_plus_matrix','hbm_read_bytes_per_call','hbm_write_bytes_per_call','text_plus_matrix_bytes')})"
  done
done
