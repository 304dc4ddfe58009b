#!/bin/bash
# A/B of environment switches on one box: the headline step, 200 timed steps each.  usage: tools/ab_bench.sh "" "MIA_HIP_X=1" ...
cd "$(dirname "$0")/.."
for e in "$@"; do
  line=$(env $e timeout -k 10 300 python3 bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
  echo "[$e] $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d.get('value_first_iteration'))" "$line")"
done
