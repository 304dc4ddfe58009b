cd $GRAFT_REPO_ROOT
O=gpurun_out/${ROUND:-r06}_campaigns.log; : > $O
run() { echo "== $*" >> $O; timeout -k 10 900 python3 "$@" 2>&1 | tail -1 >> $O; echo "rc $?" >> $O; tail -2 $O; }
run tools/band_campaign.py 40 501000 MIA_HIP_NO_DIAG_FILTER flat
run tools/band_campaign.py 40 502000 MIA_HIP_NO_DIAG_FILTER flat nrich
run tools/band_campaign.py 40 503000 MIA_HIP_NO_DIAG_FILTER ancient
run tools/band_campaign.py 40 504000 MIA_HIP_NO_DIAG_FILTER ancient nrich
run tools/band_campaign.py 30 505000 MIA_HIP_NO_DIAG_FILTER solexa
run tools/band_campaign.py 30 506000 MIA_HIP_NO_DIAG_FILTER solexa nrich
# round 6: a handful of N columns (the table spells them out, the quick plan answers for the windows without one)
run tools/band_campaign.py 20 531000 MIA_HIP_NO_DIAG_FILTER flat fewn
run tools/band_campaign.py 20 532000 MIA_HIP_NO_DIAG_FILTER ancient fewn
run tools/band_campaign.py 15 533000 MIA_HIP_NO_DIAG_FILTER solexa fewn
run tools/tally_campaign.py 80 507000
# round 5: the position-specific tally (strand split, second sort, runs of equal starts; reads of 150 and 200 bases take the RALL
# instance) against the plain global-atomic tally, 200 000 .. 400 000 reads per configuration
run tools/tally_campaign.py 60 517000 ancient MIA_HIP_NO_BINNED_TALLY 200000
run tools/tally_campaign.py 40 527000 solexa MIA_HIP_NO_BINNED_TALLY 400000
run tools/pass1_campaign.py 20 508000 plain flat
run tools/pass1_campaign.py 20 509000 nrich flat
run tools/pass1_campaign.py 30 510000 plain ancient
run tools/pass1_campaign.py 30 511000 nrich ancient
run tools/pass1_campaign.py 20 512000 plain solexa
run tools/pass1_campaign.py 20 513000 nrich solexa
