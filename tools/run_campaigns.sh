cd $GRAFT_REPO_ROOT
O=gpurun_out/${ROUND:-r05}_campaigns.log; : > $O
run() { echo "== $*" >> $O; timeout -k 10 900 python3 "$@" 2>&1 | tail -1 >> $O; echo "rc $?" >> $O; tail -2 $O; }
run tools/band_campaign.py 60 301000 MIA_HIP_NO_DIAG_FILTER flat
run tools/band_campaign.py 60 302000 MIA_HIP_NO_DIAG_FILTER flat nrich
run tools/band_campaign.py 60 303000 MIA_HIP_NO_DIAG_FILTER ancient
run tools/band_campaign.py 60 304000 MIA_HIP_NO_DIAG_FILTER ancient nrich
run tools/band_campaign.py 40 305000 MIA_HIP_NO_DIAG_FILTER solexa
run tools/band_campaign.py 40 306000 MIA_HIP_NO_DIAG_FILTER solexa nrich
run tools/tally_campaign.py 120 407000
run tools/pass1_campaign.py 40 308000 plain flat
run tools/pass1_campaign.py 40 309000 nrich flat
run tools/pass1_campaign.py 60 310000 plain ancient
run tools/pass1_campaign.py 60 311000 nrich ancient
run tools/pass1_campaign.py 40 312000 plain solexa
run tools/pass1_campaign.py 40 313000 nrich solexa
