cd $GRAFT_REPO_ROOT
python __graft_entry__.py > gpurun_out/b.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_text.py tests/test_file_runner_gpu.py tests/test_multi_device_gpu.py -x -q 2>&1 | tail -3
for parts in none sm me ce groups sm,me,ce,groups; do
  STATS_PARTS=$parts rocprofv3 --kernel-trace --stats -d gpurun_out/parts_$parts -o p --output-format csv -- python3 tools/bench_text_entry.py 10000 16000 stats 3 > gpurun_out/parts_$parts.json 2>/dev/null
  echo "$parts: $(grep k_stats_all gpurun_out/parts_$parts/*kernel_stats.csv | cut -d, -f2-4)"
done
grep -h "k_tok" gpurun_out/parts_none/*kernel_stats.csv | cut -d'(' -f1,3- | cut -c1-120
cat gpurun_out/parts_sm,me,ce,groups.json
