#!/usr/bin/env python3
"""First failing round of tools/band_campaign.py for a matrix, under the environment as it is (usage: band_bisect.py rounds seed0 matrix [nrich])"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import band_campaign
rounds, seed0, matrix = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nrich = len(sys.argv) > 4
bad = []
for k in range(rounds):
    try:
        band_campaign.run(1, seed0 + k, "MIA_HIP_NO_DIAG_FILTER", matrix, nrich, quiet=True)
    except AssertionError as e:
        bad.append((seed0 + k, str(e)))
        print("seed", seed0 + k, "differs:", e, flush=True)
        if len(bad) >= 3:
            break
print("env", {k: v for k, v in os.environ.items() if k.startswith("MIA_HIP")}, "rounds", rounds, "bad", bad)
