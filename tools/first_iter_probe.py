#!/usr/bin/env python3
"""The iteration against the starting reference itself (mt311: an ambiguity code in every tenth column), four times over, for
a kernel timeline (rocprofv3 --kernel-trace -- python3 tools/first_iter_probe.py <config>; tools/timeline.py prints the last)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import mia_amd  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
w = bench.make_workload(cfg, n_reads, 1 if cfg == 1 else 3)
hip = mia_amd.MiaHip(0)
pipe = bench.Pipeline(hip, w)
import time
for _ in range(3):
    pipe.step(w["ref"])
hip.sync()
t0 = time.perf_counter()
for _ in range(4):
    pipe.step(w["ref"])
hip.sync()
print("config", cfg, "first-iteration step: %.3f ms" % ((time.perf_counter() - t0) / 4 * 1e3), {k: os.environ[k] for k in os.environ if k.startswith("MIA_HIP")})
c = hip.bx_counters()
print("values lists", list(c[:5]), "trace lists", list(c[5:10]), "late", list(c[24:29]), "plan gave up by reason (READ WINDOW BLOCKS SPAN PATH BUDGET WIDTH)", list(c[17:24]), pipe.stages(1, None, None, True)[1])
hip.close()
