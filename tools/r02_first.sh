#!/bin/bash
# round 2, first GPU pass: new bench.py paths (resident c2, tiled m at N=1 on a short run, gloo self-launch rehearsal)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python __graft_entry__.py > $O/r02_build.log 2>&1 || exit 1
python bench.py --workload c2 --steps 10 --warmup 2 > $O/r02_bench_c2.json 2> $O/r02_bench_c2.err || { tail -5 $O/r02_bench_c2.err; exit 1; }
echo c2 done
python bench.py --steps 3 --warmup 1 > $O/r02_bench_m_n1.json 2> $O/r02_bench_m_n1.err || { tail -5 $O/r02_bench_m_n1.err; exit 1; }
echo m done
for w in c3 c4 stats; do
  python bench.py --workload $w --steps 10 --warmup 2 > $O/r02_bench_$w.json 2> $O/r02_bench_$w.err || { tail -5 $O/r02_bench_$w.err; exit 1; }
  echo $w done
done
python bench.py --gpus 2 --backend gloo --variants 400000 --samples 10000 --steps 4 --warmup 1 --tile-gb 1.2 > $O/r02_rehearsal_n2_gloo.json 2> $O/r02_rehearsal_n2_gloo.err || { tail -5 $O/r02_rehearsal_n2_gloo.err; exit 1; }
echo gloo2 done
python bench.py --gpus 2 --backend gloo --variants 400000 --samples 10000 --steps 4 --warmup 1 --tile-gb 1.2 --resident no > $O/r02_rehearsal_n2_gloo_tiled.json 2> $O/r02_rehearsal_n2_gloo_tiled.err || { tail -5 $O/r02_rehearsal_n2_gloo_tiled.err; exit 1; }
echo gloo2 tiled done
python bench.py --gpus 2 > $O/r02_n2_refused.out 2> $O/r02_n2_refused.err; echo "gpus 2 on one GPU: rc=$?" | tee -a $O/r02_n2_refused.err
