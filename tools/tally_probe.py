#!/usr/bin/env python3
"""What the tally costs with and without indel reads, and with debug switches (MIA_HIP_DEBUG_SKIP bits): a probe, not a test."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_data, mia_amd
_, _, mt = gen_data.read_fasta_one(os.path.join(ROOT, "tests", "golden", "mt311.fa"))
indiv = gen_data.resolve_individual(mt)
n = 1_000_000
for indel in (0.001, 0.0):
    d = gen_data.make_reads(indiv, n, 100, 1, circular=True, indel_rate=indel)
    stored = gen_data.stored_orientation(d)
    as_ = d["start"].astype(np.int32); ae = (as_ + 99).astype(np.int32)
    off = np.arange(n + 1, dtype=np.int64) * 100
    hip = mia_amd.MiaHip(0)
    hip.set_pssm(mia_amd.flat_pssm())
    hip.upload_reads(stored.reshape(-1), off, d["strand"].astype(np.uint8), np.ones(n, np.uint8), as_, ae)
    cur = indiv
    for k in range(4):
        cur = hip.iterate(cur, True, 0, None, 1)
    hip.stage_stats(reset=True)
    t0 = time.perf_counter()
    for k in range(6):
        cur = hip.iterate(cur, True, 0, None, 1)
    hip.sync()
    dt = (time.perf_counter() - t0) / 6 * 1e3
    st = hip.stage_stats()
    print("indel_rate", indel, "ms/iter %.3f" % dt, {k: round(v[0] / max(v[1], 1), 3) for k, v in st.items() if v[1]}, flush=True)
    hip.close()
