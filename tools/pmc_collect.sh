#!/bin/bash
# Three rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; the SQ set is the third) of the
# timed steps of one BASELINE config, summarised per kernel by tools/pmc_summary.py.  Run on the GPU box:
#   tools/pmc_collect.sh <config 1|2|4> <reads> <outdir under gpurun_out/>
# then copy <outdir>/cfg<k>.json to profiles/<round>/pmc/.  Counter runs carry --kernel-trace only (no other trace domain).
set -e
cfg=$1; reads=$2; out=$(realpath -m "$3"); mkdir -p "$out"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU"; do
  name=cfg${cfg}_$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$out/$name" -o p -- python3 "$R/bench.py" --pmc-run cfg$cfg --config $cfg --reads $reads --steps 4 --warmup 2 > "$out/$name.log" 2>&1
  echo "pass $name done"
done
python3 "$R/tools/pmc_summary.py" "$out/cfg$cfg.json" $reads $(find "$out" -name "*counter_collection.csv" -path "*cfg${cfg}_*")
