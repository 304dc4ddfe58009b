#!/bin/bash
# A/B of environment switches on one box: the headline step, ${STEPS:-200} timed steps each.  usage: tools/ab_bench.sh "" "MIA_HIP_X=1" ...
cd "$(dirname "$0")/.."
for e in "$@"; do
  line=$(env $e timeout -k 10 300 python3 bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
  echo "[$e] $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); r=d.get('roofline') or {}; print('%.4f ms/step; dominant %s %.4f ms (%s launches)' % (d['ms_per_step'], r.get('kernel'), r.get('kernel_ms') or 0, r.get('launches')))" "$line")"
done
