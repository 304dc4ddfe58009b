cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_iteration.py tests/test_gpu_bench_workloads.py tests/test_gpu_config3.py tests/test_gpu_iter_push.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py tests/test_gpu_comm.py -x -q 2>&1 | tail -2 || exit 1
for spec in "1 1000000" "3 10000000" "1 1000000" "3 10000000"; do
  set -- $spec
line=$(timeout -k 10 300 python3 bench.py --config $1 --reads $2 --steps 40 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
echo "cfg$1 $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('%.4f ms/step' % d['ms_per_step'], d['certificate']['alignments_sha256'][:12], d['certificate']['consensus_sha256'][:12])" "$line")"
done
bash tools/quick_timeline.sh 3 10000000 r6cull | grep "cull_records\|slot_count\|tally_binned\|step span"
bash tools/quick_timeline.sh 1 1000000 r6cull1 | grep "cull_records\|slot_count\|tally_binned\|step span"
