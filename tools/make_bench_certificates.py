#!/usr/bin/env python3
"""GPU box: the digests bench.py prints for its four workloads (bench.certificate: sha256 of the converged consensus and of every
read's (score, as, ae)), written to gpurun_out/bench_certificates.json -- copy to tests/golden/ once the tests that check samples
of the SAME workloads against the oracle are green (tests/test_gpu_bench_workloads.py, test_gpu_config3.py, test_gpu_config4_full.py).
usage: python3 tools/make_bench_certificates.py [cfg ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import mia_amd  # noqa: E402

WORKLOADS = {1: (1_000_000, 1), 2: (1_000_000, 3), 3: (10_000_000, 4), 4: (5_000_000, 5)}      # bench.py: timed_job (seed 1 + rank), section_converge calls


def converge(cfg, n, seed):
    w = bench.make_workload(cfg, n, seed)
    hip = mia_amd.MiaHip(0)
    pipe = bench.Pipeline(hip, w)
    cur, rounds = w["ref"], 0
    while rounds < 12:
        nxt = pipe.step(cur)
        rounds += 1
        if nxt == cur:
            break
        cur = nxt
    c = bench.certificate(hip, nxt, fixed_point=nxt == cur)
    c["iterations_to_convergence"] = rounds
    c["workload"] = "make_workload(%d, %d, seed=%d)" % (cfg, n, seed)
    hip.close()
    return c


if __name__ == "__main__":
    which = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]
    path = os.path.join(ROOT, "gpurun_out", "bench_certificates.json")
    out = {}
    if os.path.exists(os.path.join(ROOT, "tests", "golden", "bench_certificates.json")):
        out = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_certificates.json")))
    for cfg in which:
        out["cfg%d" % cfg] = converge(cfg, *WORKLOADS[cfg])
        print("cfg%d" % cfg, out["cfg%d" % cfg], flush=True)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)
