# usage (on the GPU box): bash tools/quick_timeline.sh <config> <reads> <tag> [first]  -- kernel timeline of one steady step (or of the first iteration) under rocprofv3
set -e
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; mkdir -p $O
c=$1; reads=$2; tag=$3
if [ "$4" = first ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o p -- python3 $R/tools/first_iter_probe.py $c $reads > $O/$tag.txt 2>&1
else
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o p -- python3 $R/bench.py --config $c --reads $reads --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $O/$tag.json 2> $O/$tag.err
fi
python3 $R/tools/timeline.py $O/$tag > $O/${tag}_timeline.txt
rm -rf $O/$tag
cat $O/${tag}_timeline.txt
