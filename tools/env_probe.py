#!/usr/bin/env python3
"""Steady and first-iteration step time of a config under a list of environment settings (alt build).
usage: env_probe.py <config> <reads> "A=1 B=2" "C=3" ...   ("" = the default)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import mia_amd  # noqa: E402

cfg = int(sys.argv[1])
n = int(sys.argv[2])
w = bench.make_workload(cfg, n, 1 if cfg == 1 else 3)
hip = mia_amd.MiaHip(0)
pipe = bench.Pipeline(hip, w)
cur = w["ref"]
for _ in range(6):
    cur = pipe.step(cur)
hip.close()
reps = 20 if n <= 2_000_000 else 6


def timed(pipe, hip, ref):
    for _ in range(3):
        pipe.step(ref)
    hip.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            pipe.step(ref)
        hip.sync()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


for spec in sys.argv[3:]:
    env = dict(kv.split("=", 1) for kv in spec.split())
    os.environ.update(env)
    hip = mia_amd.MiaHip(0)
    for k in env:
        os.environ.pop(k)
    pipe = bench.Pipeline(hip, w)
    t_steady = timed(pipe, hip, cur)
    st = {k: round(v[0] / max(v[1], 1), 4) for k, v in hip.stage_stats().items() if v[1]} if os.environ.get("PROBE_STAGES") else {}
    print("[%s] cfg %d %d reads: steady %.4f ms, first %.4f ms" % (spec, cfg, n, t_steady, timed(pipe, hip, w["ref"])), st, flush=True)
    hip.close()
