# PMC passes over the Fisher p-pass (bench.py --workload c3), separate runs, kernel trace only beside --pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES -d $R/gpurun_out/fi_pmc1 -o pmc1 --output-format csv -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/fi_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/fi_pmc2 -o pmc2 --output-format csv -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/fi_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/fi_pmc3 -o pmc3 --output-format csv -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/fi_pmc3.log 2>&1
ls $R/gpurun_out/fi_pmc3
