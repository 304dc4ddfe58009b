"""Wall time of mia_hip_tally on the bench workload under debug switches (MIA_HIP_DEBUG_SKIP bit 32 = no one-read-per-
wavefront path in k_tally_binned: wrong sums, only the timing means something) and without the banded DP."""
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
for label, env in (("default", {}), ("no general path", {"MIA_HIP_DEBUG_SKIP": "32"}), ("no banded DP", {"MIA_HIP_NO_BAND_DP": "1"}), ("one-gap reads dropped", {"MIA_HIP_DEBUG_SKIP": "2048"}), ("neither", {"MIA_HIP_DEBUG_SKIP": "2080"})):
    for k, v in env.items():
        os.environ[k] = v
    hip = mia_amd.MiaHip(0)
    for k in env:
        os.environ.pop(k)
    hip.set_pssm(mia_amd.flat_pssm())
    hip.upload_reads(stored.reshape(-1), soff, strand, np.ones(n, np.uint8), as0, ae0)
    hip.realign(ref, True)
    sc, a, e = hip.alignments()
    hip.cull(0, 100.0, 0.0, 0)
    ts = []
    for it in range(6):
        hip.sync()
        t0 = time.perf_counter()
        hip.tally()
        hip.sync()
        ts.append(time.perf_counter() - t0)
    cols, rstart = hip.scripts()
    ngap = ((cols[:, 1:] >= 0) & (cols[:, :-1] >= 0) & (cols[:, 1:] - cols[:, :-1] > 1)).sum(axis=1) + ((cols[:, 1:] == -1) & (cols[:, :-1] != -1)).sum(axis=1)
    print(label, "tally ms", round(min(ts) * 1e3, 3), "reads with 0/1/2+ gaps", int((ngap == 0).sum()), int((ngap == 1).sum()), int((ngap > 1).sum()), flush=True)
    hip.close()
