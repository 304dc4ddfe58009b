"""k_myers_ond: kernel time (HIP events) of one 16.6 kb pair against the number of differences -- the per-row cost and the fixed part."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mia_amd  # noqa: E402

rng = np.random.default_rng(7)
bases = np.frombuffer(b"ACGT", np.uint8)
hip = mia_amd.MiaHip(0)
for length in (16_600, 2_000):
    big = bases[rng.integers(0, 4, length)].copy()
    for subs in (0, 1, 10, 40, 80, 160, 320, 640):
        b2 = big.copy()
        for p in rng.integers(0, len(big), subs):
            b2[p] = bases[(np.searchsorted(bases, b2[p]) + 1) % 4]
        ms = []
        for _ in range(5):
            d = hip.myers([big.tobytes()], [b2.tobytes()], np.zeros(1, np.int32), np.full(1, length // 10, np.int32))
            ms.append(hip.myers_time())
        print("len %6d subs %4d distance %10d kernel ms: min %.4f median %.4f" % (length, subs, int(d[0]), min(ms), sorted(ms)[2]))
hip.close()
