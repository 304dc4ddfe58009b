#!/usr/bin/env python3
"""Which route of k_tally_binned do the reads of a steady step take?  Counts, from the scripts of one iteration: gap-free reads,
reads with one gap (an insert or a deletion run), reads with more (the one-read-per-wavefront path)."""
import os
import sys

import numpy as np

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
ref = w["ref"]
for _ in range(3):
    out = pipe.step(ref)
    if isinstance(out, (bytes, str)):
        ref = out
hip.sync()
cols, rs = hip.scripts()
lens = np.diff(np.asarray(w["offsets"])).astype(np.int64)
n = len(lens)
INS = -1
runs_ins = np.zeros(n, np.int64); runs_del = np.zeros(n, np.int64); soft = np.zeros(n, np.int64)
c = cols.astype(np.int32)
valid = np.arange(c.shape[1])[None, :] < lens[:, None]
aligned = (c >= 0) & valid
ins = (c == INS) & valid
# insert runs: an insert row whose predecessor is not an insert row
prev_ins = np.zeros_like(ins); prev_ins[:, 1:] = ins[:, :-1]
runs_ins = (ins & ~prev_ins).sum(1)
# deletion runs: two consecutive aligned rows whose columns differ by more than one
d = np.zeros_like(c); d[:, 1:] = c[:, 1:] - c[:, :-1]
both = np.zeros_like(aligned); both[:, 1:] = aligned[:, 1:] & aligned[:, :-1]
runs_del = (both & (d > 1)).sum(1)
other = ((c < 0) & (c != INS) & valid).sum(1)
gaps = runs_ins + runs_del
print("reads", n, "gap-free", int((gaps == 0).sum()), "one gap", int((gaps == 1).sum()), "two", int((gaps == 2).sum()), "more", int((gaps > 2).sum()),
      "rows that are neither aligned nor inserted (soft ends)", int((other > 0).sum()))
c = hip.bx_counters()
print("bx counters: values lists", list(c[:5]), "trace lists", list(c[5:10]), "cur/done/seen", list(c[10:16]), "plan gave up by reason", list(c[16:24]), "late", list(c[24:30]), "cand", list(c[30:32]))
try:
    print("stages", pipe.stages(1, None, None, True)[1])
except Exception as e:
    print("stages: n/a", e)
if os.environ.get("MIA_HIP_DEBUG_SKIP"):
    import ctypes as C
    k = np.zeros(16, np.uint64)
    hip._l.mia_hip_debug_tally_kinds.argtypes = [C.c_void_p, C.c_void_p]
    hip._l.mia_hip_debug_tally_kinds(hip._h, k.ctypes.data_as(C.c_void_p))      # (resets the counts)
    pipe.step(ref); hip.sync()
    hip._l.mia_hip_debug_tally_kinds(hip._h, k.ctypes.data_as(C.c_void_p))
    print("k_tally_binned routes of one step: gap-free lane", int(k[0]), "(bit-sliced", int(k[1]), ") over the origin", int(k[2]), "one gap", int(k[3]),
          "(bit-sliced", int(k[4]), ") one per wavefront", int(k[5]), "(of them marked one-gap", int(k[6]), ", marked diagonal", int(k[7]), ")")
    if k[15]:
        print("k_tally_binned, shader-clock cycles per workgroup (thread 0): init %.0f, reads %.0f, counters to window %.0f, barrier %.0f, flush %.0f, slab %.0f; workgroups %d"
              % tuple([float(k[8 + q]) / float(k[15]) for q in range(6)] + [int(k[15])]))
hip.close()
