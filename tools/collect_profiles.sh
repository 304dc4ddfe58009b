set -x
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; RND=${ROUND:-r06}; O=$R/gpurun_out/$RND; mkdir -p $O
PART=${PART:-all}
if [ $PART != pmc ]; then
for c in 1 2 3 4; do
  reads=1000000; [ $c = 3 ] && reads=10000000; [ $c = 4 ] && reads=5000000
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_cfg$c -o p -- python3 $R/bench.py --config $c --reads $reads --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $O/ks_cfg$c.json 2> $O/ks_cfg$c.err || exit 1
  python3 $R/tools/timeline.py $O/ks_cfg$c > $O/cfg${c}_timeline.txt
  rm -f $O/ks_cfg$c/*kernel_trace.csv $O/ks_cfg$c/*agent_info.csv
done
fi
cd $R
if [ $PART != stats ]; then
for c in 1 2 4 3; do
  reads=1000000; [ $c = 3 ] && reads=10000000; [ $c = 4 ] && reads=5000000
  timeout -k 10 700 bash tools/pmc_collect.sh $c $reads gpurun_out/$RND/pmc > $O/pmc_cfg$c.log 2>&1 || { tail -5 $O/pmc_cfg$c.log; exit 1; }
  find gpurun_out/$RND/pmc -name "*.csv" -delete
done
fi
if [ $PART != pmc ]; then
# the first iteration (against mt311 itself) of configs[1..3], 1 M reads, and of configs[3] at its 10 M
cd /tmp
for c in 1 2 3; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/fi_cfg$c -o p -- python3 $R/tools/first_iter_probe.py $c 1000000 > $O/fi_cfg$c.txt 2>&1 && python3 $R/tools/timeline.py $O/fi_cfg$c > $O/first_iteration_cfg${c}_timeline.txt
  rm -rf $O/fi_cfg$c
done
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/fi_cfg3_10m -o p -- python3 $R/tools/first_iter_probe.py 3 10000000 > $O/fi_cfg3_10m.txt 2>&1 && python3 $R/tools/timeline.py $O/fi_cfg3_10m > $O/first_iteration_cfg3_10M_timeline.txt
rm -rf $O/fi_cfg3_10m
cd $R
fi
ls -la gpurun_out/$RND gpurun_out/$RND/pmc 2>/dev/null | head -60
