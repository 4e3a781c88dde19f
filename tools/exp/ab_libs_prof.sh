#!/bin/bash
# per-kernel durations of library variants: tools/exp/ab_libs_prof.sh "<A B ...>" <script> [args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
L=$R/hpg-variant_amd/lib
cp $L/libhpgv.so $L/libhpgv_keep.so
for v in $1; do
  cp $L/libhpgv_$v.so $L/libhpgv.so
  echo "== $v: $(python3 $R/$2 "${@:3}" | tail -1 | cut -c1-120)"
  HEAD=4 $R/tools/prof_kernels.sh ab_$v $2 "${@:3}" | awk -F'",' 'NR>1 {split($1,a,"("); print "   " a[1] "  " $2}' | cut -c1-120
done
cp $L/libhpgv_keep.so $L/libhpgv.so
