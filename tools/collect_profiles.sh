set -x
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
for c in 1 2 3 4; do
  reads=1000000; [ $c = 3 ] && reads=10000000; [ $c = 4 ] && reads=500000
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_cfg$c -o p -- python3 $R/bench.py --config $c --reads $reads --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $O/ks_cfg$c.json 2> $O/ks_cfg$c.err || exit 1
  python3 $R/tools/timeline.py $O/ks_cfg$c > $O/cfg${c}_timeline.txt
  rm -f $O/ks_cfg$c/*kernel_trace.csv $O/ks_cfg$c/*agent_info.csv
done
cd $R
for c in 1 2 4 3; do
  reads=1000000; [ $c = 3 ] && reads=10000000; [ $c = 4 ] && reads=500000
  timeout -k 10 500 bash tools/pmc_collect.sh $c $reads gpurun_out/r03/pmc > $O/pmc_cfg$c.log 2>&1 || { tail -5 $O/pmc_cfg$c.log; exit 1; }
  find gpurun_out/r03/pmc -name "*.csv" -delete
done
ls -la gpurun_out/r03 gpurun_out/r03/pmc | head -40
