# a second set of band campaigns with other seeds (usage on the GPU box: SEED=610000 bash tools/run_band_campaigns_extra.sh)
cd $GRAFT_REPO_ROOT
S=${SEED:-610000}
O=gpurun_out/${ROUND:-r06}_campaigns_extra.log; : > $O
run() { echo "== $*" >> $O; timeout -k 10 900 python3 "$@" 2>&1 | tail -1 >> $O; echo "rc $?" >> $O; tail -2 $O; }
run tools/band_campaign.py 30 $((S + 1000)) MIA_HIP_NO_DIAG_FILTER flat
run tools/band_campaign.py 20 $((S + 2000)) MIA_HIP_NO_DIAG_FILTER flat fewn
run tools/band_campaign.py 30 $((S + 3000)) MIA_HIP_NO_DIAG_FILTER ancient
run tools/band_campaign.py 20 $((S + 4000)) MIA_HIP_NO_DIAG_FILTER ancient fewn
run tools/band_campaign.py 25 $((S + 5000)) MIA_HIP_NO_DIAG_FILTER solexa
run tools/band_campaign.py 15 $((S + 6000)) MIA_HIP_NO_DIAG_FILTER solexa fewn
run tools/band_campaign.py 20 $((S + 7000)) MIA_HIP_NO_DIAG_FILTER flat nrich
run tools/tally_campaign.py 40 $((S + 8000))
run tools/tally_campaign.py 30 $((S + 9000)) ancient MIA_HIP_NO_BINNED_TALLY 200000
