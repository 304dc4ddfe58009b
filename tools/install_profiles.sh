#!/bin/bash
# gpurun_out/r03 (tools/collect_profiles.sh on the GPU box) -> profiles/r03
cd "$(dirname "$0")/.."
for c in 1 2 3 4; do
  cp gpurun_out/r03/ks_cfg$c/p_kernel_stats.csv profiles/r03/cfg${c}_kernel_stats.csv
  tail -1 gpurun_out/r03/ks_cfg$c.json > profiles/r03/cfg${c}_bench_under_rocprofv3.json
  cp gpurun_out/r03/cfg${c}_timeline.txt profiles/r03/cfg${c}_timeline.txt
  cp gpurun_out/r03/pmc/cfg$c.json profiles/r03/pmc/cfg$c.json
done
[ -f gpurun_out/r03/bench_default.json ] && tail -1 gpurun_out/r03/bench_default.json > profiles/r03/bench_default.json
python3 -c "
import json,sys
sys.path.insert(0,'.')
import bench
print('source hash', bench.source_hash(), [json.load(open('profiles/r03/pmc/cfg%d.json'%c))['_meta']['source_hash'] for c in (1,2,3,4)])"
