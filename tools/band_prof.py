"""Where k_band_align's time goes: the bench workload with parts of the kernel switched off (MIA_HIP_DEBUG_SKIP bits
256 = stop after the plan, 512 = stop before the traceback; results are then wrong, only the timing means something)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import gen_data  # noqa: E402
import mia_amd  # noqa: E402

n = 1_000_000
ref0, stored, soff, strand, as0, ae0 = bench.make_workload(n, 1)
ref = gen_data.resolve_individual(ref0)
for bits in (0, 256, 512):
    os.environ["MIA_HIP_DEBUG_SKIP"] = str(bits)
    hip = mia_amd.MiaHip(0)
    hip.set_pssm(mia_amd.flat_pssm())
    hip.upload_reads(stored.reshape(-1), soff, strand, np.ones(n, np.uint8), as0, ae0)
    for it in range(3):
        hip.realign(ref, True)
    hip.band_stats(reset=True)
    for it in range(5):
        hip.realign(ref, True)
    done, ms, k = hip.band_stats()
    print("dbg", bits, "band kernel ms", ms / k, "finished per launch", done / k, flush=True)
    hip.close()
