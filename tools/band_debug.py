#!/usr/bin/env python3
"""One round of tools/band_campaign.py (flat matrix, given round index k of seed0) under two settings of the environment: the reads that differ.
usage: band_debug.py seed0 k "ENV_A=.." "ENV_B=.." """
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import mia_amd
from test_gpu_band import damaged_reads
from test_gpu_filter_stress import adversarial_reference
seed0, k = int(sys.argv[1]), int(sys.argv[2])
n = 100_000
seed = seed0 + k
rng = np.random.default_rng(seed)
read_len = int(rng.choice([30, 33, 41, 50, 59, 60, 64, 77, 90, 100, 101, 128, 150, 200, 250]))
L = int(rng.integers(2000, 20000))
ref = adversarial_reference(rng, L) if k % 3 == 0 else rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
reads, start = damaged_reads(rng, ref, n, read_len, float(rng.choice([0.2, 0.5, 0.8])), int(rng.integers(1, min(13, read_len // 4))),
                             int(rng.integers(1, 8)), two_share=float(rng.choice([0.0, 0.1, 0.4])), junk_share=float(rng.choice([0.0, 0.05, 0.2])))
jitter = rng.integers(-10, 11, n) * (rng.random(n) < 0.3)
as0 = ((start + jitter) % L).astype(np.int32)
ae0 = (as0 + read_len - 1).astype(np.int32)
off = np.arange(n + 1, dtype=np.int64) * read_len
out = []
for spec in sys.argv[3:5]:
    env = dict(kv.split("=", 1) for kv in spec.split())
    os.environ.update(env)
    hip = mia_amd.MiaHip(0)
    for q in env: os.environ.pop(q)
    hip.set_pssm(mia_amd.flat_pssm())
    hip.upload_reads(reads.reshape(-1), off, np.zeros(n, np.uint8), np.ones(n, np.uint8), as0, ae0)
    hip.realign(ref.tobytes().decode(), True)
    sc, a, e = hip.alignments()
    out.append((sc.copy(), a.copy(), e.copy()))
    print(spec, "bx stats", hip.bx_stats()[0][:12])
    hip.close()
d = np.nonzero((out[0][0] != out[1][0]) | (out[0][1] != out[1][1]) | (out[0][2] != out[1][2]))[0]
print("len", read_len, "L", L, "differing reads", len(d), d[:10])
for i in d[:6]:
    print(i, "A:", out[0][0][i], out[0][1][i], out[0][2][i], " B:", out[1][0][i], out[1][1][i], out[1][2][i], "as0", as0[i], "read", reads[i].tobytes().decode())
    w0 = as0[i] - 60
    print("   ref window", ref[max(0, w0):as0[i] + read_len + 60].tobytes().decode())
