#!/bin/bash
# random (members, variants, samples) through tests/c/group_scan (bit-identity of the group scan with a single context): a soak run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
L=$R/hpg-variant_amd/lib
gcc -O1 -g -std=gnu99 -I $R/include $R/tests/c/group_scan.c -o /tmp/group_scan -L $L -lhpgv -Wl,-rpath,$L -lm || exit 1
RANDOM=${1:-7}
for i in $(seq 1 ${2:-25}); do
  G=$(( RANDOM % 5 + 1 )); V=$(( (RANDOM * 7 + RANDOM) % 60000 + 1 )); N=$(( RANDOM % 6000 + 3 ))
  S=""; [ $(( RANDOM % 3 )) = 0 ] && S=self
  out=$(timeout -k 5 120 /tmp/group_scan $G $V $N $S 2>&1) || { echo "FAILED: $G $V $N $S"; echo "$out" | tail -5; exit 1; }
  echo "$out" | grep -q "group scan ok" || { echo "NOT OK: $G $V $N $S"; exit 1; }
done
echo "group scan soak ok (${2:-25} shapes)"
