#!/usr/bin/env python3
"""How well does the GPU overlap the iterations of several contexts?  W contexts with n / W reads each, one host thread each, all
iterating at once (no exchange between them: an upper bound on what slices of one read set could gain), against one
context with all n reads.  usage: python tools/slice_probe.py <config> <reads> <W> [stagger_us]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import mia_amd  # noqa: E402

cfg, n, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
stagger = float(sys.argv[4]) * 1e-6 if len(sys.argv) > 4 else 0.0
w = bench.make_workload(cfg, n, 1 if cfg == 1 else 3)
L = w["read_len"]


def ctx(lo, hi):
    h = mia_amd.MiaHip(0)
    h.set_pssm(w["pssm"])
    h.upload_reads(w["stored"][lo:hi].reshape(-1), np.arange(hi - lo + 1, dtype=np.int64) * L, w["rc"][lo:hi], np.ones(hi - lo, np.uint8), w["as_"][lo:hi], w["ae"][lo:hi])
    return h


whole = ctx(0, n)
cur = w["ref"]
for _ in range(3):
    cur = whole.iterate(cur, w["circular"])
whole.sync()
t0 = time.perf_counter()
for _ in range(20):
    cur = whole.iterate(cur, w["circular"])
whole.sync()
one = (time.perf_counter() - t0) / 20 * 1e3
cuts = [n * k // W for k in range(W + 1)]
parts = [ctx(cuts[k], cuts[k + 1]) for k in range(W)]
K = 20
bar = threading.Barrier(W + 1)
times = []


def work(k):
    c = w["ref"]
    for _ in range(3):
        c = parts[k].iterate(c, w["circular"])
    c = cur
    for _ in range(K):
        bar.wait()
        if stagger:
            time.sleep(stagger * k)
        parts[k].iterate(c, w["circular"])
        bar.wait()


th = [threading.Thread(target=work, args=(k,)) for k in range(W)]
for t in th:
    t.start()
for _ in range(K):
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    times.append((time.perf_counter() - t0) * 1e3)
for t in th:
    t.join()
times.sort()
print("config %d, %d reads: one context %.3f ms per step; %d contexts of %d reads at once: median %.3f ms, best %.3f ms (stagger %.0f us)"
      % (cfg, n, one, W, n // W, times[len(times) // 2], times[0], stagger * 1e6))
