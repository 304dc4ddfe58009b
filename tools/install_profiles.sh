#!/bin/bash
# gpurun_out/<round> (tools/collect_profiles.sh on the GPU box) -> profiles/<round>; ROUND=r06 by default
cd "$(dirname "$0")/.."
RND=${ROUND:-r06}; mkdir -p profiles/$RND/pmc
for c in 1 2 3 4; do
  cp gpurun_out/$RND/ks_cfg$c/p_kernel_stats.csv profiles/$RND/cfg${c}_kernel_stats.csv
  tail -1 gpurun_out/$RND/ks_cfg$c.json > profiles/$RND/cfg${c}_bench_under_rocprofv3.json
  cp gpurun_out/$RND/cfg${c}_timeline.txt profiles/$RND/cfg${c}_timeline.txt
  cp gpurun_out/$RND/pmc/cfg$c.json profiles/$RND/pmc/cfg$c.json
done
cp gpurun_out/$RND/first_iteration_*_timeline.txt profiles/$RND/ 2>/dev/null
[ -f gpurun_out/$RND/bench_default.json ] && tail -1 gpurun_out/$RND/bench_default.json > profiles/$RND/bench_default.json
python3 -c "
import json,sys
sys.path.insert(0,'.')
import bench
print('source hash', bench.source_hash(), [json.load(open('profiles/'+__import__('os').environ.get('ROUND','r06')+'/pmc/cfg%d.json'%c))['_meta']['source_hash'] for c in (1,2,3,4)])"
