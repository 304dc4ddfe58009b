#!/usr/bin/env python3
"""Pass-1 micro-benchmark: N synthetic 100 bp reads against mt311, kernel time from HIP events.
usage: python tools/p1_bench.py [n_reads] [kmer_len] [resolved|mt311] [flat|ancient|solexa]      (env: MIA_HIP_P1_CPL, MIA_HIP_P1_PLAIN, MIA_HIP_NO_DIAG_FILTER)
"resolved": reference and reads come from mt311 with its ambiguity codes resolved to plain bases (the usual kind of
reference; against mt311 itself half of the columns are N for the aligner and the diagonal filter cannot decide anything)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_data  # noqa: E402
import mia_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    ref = open(os.path.join(ROOT, "tests", "golden", "mt311.fa")).read().split("\n", 1)[1].replace("\n", "")
    if len(sys.argv) > 3 and sys.argv[3] == "resolved":
        ref = gen_data.resolve_individual(ref)
    rng = np.random.default_rng(5)
    L = len(ref)
    refa = np.frombuffer(ref.encode(), dtype=np.uint8)
    pos = rng.integers(0, L, n)
    idx = (pos[:, None] + np.arange(100)[None, :]) % L
    seq = refa[idx].copy()
    mut = rng.random(seq.shape) < 0.02
    seq[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(mut.sum()))]
    rc = rng.random(n) < 0.5
    seq[rc] = gen_data._COMP[seq[rc][:, ::-1]]
    offsets = (np.arange(n + 1, dtype=np.int64) * 100)
    hip = mia_amd.MiaHip(0)
    matrix = sys.argv[4] if len(sys.argv) > 4 else "flat"
    golden = os.path.join(ROOT, "tests", "golden")
    hip.set_pssm(mia_amd.flat_pssm() if matrix == "flat" else mia_amd.read_pssm(os.path.join(golden, {"ancient": "ancient.submat.txt", "solexa": "ancient.submat.solexa.pe.txt"}[matrix])))
    hip.pass1(ref, True, seq[:256].reshape(-1), offsets[:257], k)
    t0 = time.perf_counter()
    sc, rcs, as_, ae, fl = hip.pass1(ref, True, seq.reshape(-1), offsets, k)
    wall = time.perf_counter() - t0
    ms = hip.pass1_time()
    print("reads %d k %d: kernel %.2f ms (%.0f reads/s), wall %.1f ms (%.0f reads/s), kept %d, mean score %.1f, strand agreement %.4f, "
          "decided by the diagonal filter %d, by anchored windows %d"
          % (n, k, ms, n / ms * 1e3, wall * 1e3, n / wall, int(((fl & 2) != 0).sum()), float(sc.mean()), float((rcs.astype(bool) == rc).mean()),
             hip.pass1_filtered(), hip.pass1_anchored()))


if __name__ == "__main__":
    main()
