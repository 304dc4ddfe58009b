"""Differential campaign for the tally's one-read-per-lane paths (proven diagonal, one gap, over the origin; flat matrix
only): many seeds of tests/test_gpu_band.py's generators through realign + cull + tally + consensus, against the same run
with MIA_HIP_NO_LINEAR_TALLY=1 (gapped and over-the-origin reads walk their scripts one per wavefront, every base adds its
four score words).  Column tallies, ref->gaps, consensus and insert tallies must be equal.  usage: tally_campaign.py [rounds [first seed]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mia_amd  # noqa: E402
from test_gpu_band import damaged_reads  # noqa: E402



def run(rounds=40, seed0=3000, n=60_000, matrix="flat", switch="MIA_HIP_NO_LINEAR_TALLY", quiet=False):
    """matrix: flat (the one-read-per-lane and bit-sliced paths against per-base score adds) or ancient / solexa with
    switch MIA_HIP_NO_BINNED_TALLY (the LDS-window tally against the plain global-atomic one); returns reads compared"""
    golden = os.path.join(ROOT, "tests", "golden")
    pssm = mia_amd.flat_pssm() if matrix == "flat" else mia_amd.read_pssm(os.path.join(golden, {"ancient": "ancient.submat.txt", "solexa": "ancient.submat.solexa.pe.txt"}[matrix]))
    t0 = time.time()
    for k in range(rounds):
        rng = np.random.default_rng(seed0 + k)
        read_len = int(rng.choice([36, 50, 64, 77, 100, 101, 128, 150, 200]))
        L = int(rng.integers(600, 12000))                      # short references: a large share of the reads runs over the origin
        ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
        reads, start = damaged_reads(rng, ref, n, read_len, float(rng.choice([0.2, 0.6])), int(rng.integers(1, 6)), int(rng.integers(0, 5)),
                                     two_share=float(rng.choice([0.0, 0.2])), junk_share=float(rng.choice([0.0, 0.1])))
        strand = (rng.random(n) < 0.5).astype(np.uint8)
        sk = (rng.random(n) < 0.98).astype(np.uint8)
        off = np.arange(n + 1, dtype=np.int64) * read_len
        as0 = (start % L).astype(np.int32)
        ae0 = (as0 + read_len - 1).astype(np.int32)
        refs = ref.tobytes().decode()
        out = []
        for env in (None, switch):
            if env:
                os.environ[env] = "1"
            hip = mia_amd.MiaHip(0)
            if env:
                os.environ.pop(env)
            hip.set_pssm(pssm)
            hip.upload_reads(reads.reshape(-1), off, strand, sk, as0, ae0)
            hip.realign(refs, True)
            sc, a, e = hip.alignments()
            cut = hip.score_cut(sc, np.full(n, read_len, np.int32))
            hip.cull(0, cut[0] if cut[0] > 0 else 100.0, cut[1], 0)
            hip.tally()
            t, g = hip.get_tally()
            cons = hip.consensus(1)
            it = hip.ins_tally()
            out.append((t, g, cons, it))
            hip.close()
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), ("tally", seed0 + k)
        assert out[0][2] == out[1][2], ("consensus", seed0 + k)
        for x, y in zip(out[0][3], out[1][3]):
            assert np.array_equal(x, y), ("insert tally", seed0 + k)
        if not quiet:
            print("round", k, "len", read_len, "L", L, "ok", round(time.time() - t0, 1), "s", flush=True)
    print("campaign done:", rounds, "configurations, no difference")
    return rounds * n


if __name__ == "__main__":
    # usage: tally_campaign.py [rounds [first seed [matrix [switch [reads]]]]]
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 3000,
        n=int(sys.argv[5]) if len(sys.argv) > 5 else 60_000,
        matrix=sys.argv[3] if len(sys.argv) > 3 else "flat", switch=sys.argv[4] if len(sys.argv) > 4 else "MIA_HIP_NO_LINEAR_TALLY")
