#!/usr/bin/env python3
"""Where the band kernels' time goes (profiling only): the steady step of configs[1] / [2] with parts of the kernels switched
off through MIA_HIP_BX_DEBUG (1 no traceback, 2 one DP row, 4 no values launch, 8 no trace launch).  Results of such runs
are wrong by construction; only the stage times are read.  usage: bxl_probe.py [config]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import mia_amd  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = bench.make_workload(cfg, 1_000_000, 1 if cfg == 1 else 3)
# the converged reference, from a clean context
hip = mia_amd.MiaHip(0)
pipe = bench.Pipeline(hip, w)
cur = w["ref"]
for _ in range(5):
    cur = pipe.step(cur)
hip.close()
for env in ({}, {"MIA_HIP_NO_LANES": "1"}, {"MIA_HIP_BX_DEBUG": "1"}, {"MIA_HIP_BX_DEBUG": "2"}, {"MIA_HIP_BX_DEBUG": "3"}, {"MIA_HIP_BX_DEBUG": "4"}, {"MIA_HIP_BX_DEBUG": "8"},
            {"MIA_HIP_BX_DEBUG": "5"}):
    os.environ.update(env)
    hip = mia_amd.MiaHip(0)
    for k in env:
        os.environ.pop(k)
    pipe = bench.Pipeline(hip, w)
    pipe.step(cur)
    pipe.reset_stats()
    for _ in range(6):
        pipe.step(cur)                  # (always against the converged reference: the outputs of a crippled run are not fed back)
    hip.sync()
    st = hip.stage_stats()
    print(env, {k: round(v[0] / max(v[1], 1), 4) for k, v in st.items() if v[1]}, flush=True)
    hip.close()
