#!/usr/bin/env python3
"""Steady-state step time of a config (converged reference) under a few switches.  usage: steady_probe.py [config] [reads]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import mia_amd  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
w = bench.make_workload(cfg, n, 1 if cfg == 1 else 3)
hip = mia_amd.MiaHip(0)
pipe = bench.Pipeline(hip, w)
cur = w["ref"]
for _ in range(6):
    cur = pipe.step(cur)
hip.close()
for env in ({}, {"MIA_HIP_BX_SERIAL": "1"}, {"MIA_HIP_NO_LANES": "1"}, {"MIA_HIP_NO_AUTO_PLAIN": "1"}):
    os.environ.update(env)
    hip = mia_amd.MiaHip(0)
    for k in env:
        os.environ.pop(k)
    pipe = bench.Pipeline(hip, w)
    for _ in range(3):
        pipe.step(cur)
    hip.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        pipe.step(cur)
    hip.sync()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    st = hip.stage_stats()
    print(env, "step %.3f ms" % ms, {k: round(v[0] / max(v[1], 1), 3) for k, v in st.items() if v[1]}, flush=True)
    hip.close()
