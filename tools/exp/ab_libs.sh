#!/bin/bash
# same-box A/B of library variants: tools/exp/ab_libs.sh "<A B ...>" <script> [args]  (variants = hpg-variant_amd/lib/libhpgv_<X>.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
L=$R/hpg-variant_amd/lib
cp $L/libhpgv.so $L/libhpgv_keep.so
for round in 1 2; do
  for v in $1; do
    cp $L/libhpgv_$v.so $L/libhpgv.so
    echo "$v: $(python3 $R/$2 "${@:3}" | tail -1 | cut -c1-140)"
  done
done
cp $L/libhpgv_keep.so $L/libhpgv.so
