"""Differential campaign for pass 1 without k-mer filter: the diagonal filter and the anchored windows (flat matrix, both
strands of the whole reference) against the whole-strand DP kernel alone (MIA_HIP_NO_DIAG_FILTER=1): score, end points,
strand and flags of every read must be equal.  References with repeated and reverse-complemented blocks, circular and
linear, reads of 60-200 bases with 0-9 substitutions and indels of 1-30 bases.  usage: pass1_campaign.py [rounds [first seed
[nrich]]] -- nrich: after the reads are drawn, 1-20 % of the reference's columns and a few stretches become ambiguity codes
(mt311's kind of reference: the anchored windows work from a table that lists such 10-mers under every spelling).
matrix: flat (default), ancient, solexa -- with a position-specific matrix there is no diagonal filter; the anchored windows
work in losses (mia_pass1_kernels.h, GEN), and the reads carry aDNA damage (C->T at the 5' end, G->A at the 3' end)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mia_amd  # noqa: E402
from test_gpu_band import damaged_reads  # noqa: E402
from test_gpu_filter_stress import COMP  # noqa: E402



GOLDEN = os.path.join(ROOT, "tests", "golden")


def run(rounds=30, seed0=7000, nrich=False, n=12_000, quiet=False, matrix="flat"):
    """returns (reads compared, reads decided by the diagonal filter, reads decided by the anchored windows)"""
    total_decided = [0, 0]
    PSSM = mia_amd.flat_pssm() if matrix == "flat" else mia_amd.read_pssm(os.path.join(GOLDEN, {"ancient": "ancient.submat.txt", "solexa": "ancient.submat.solexa.pe.txt"}[matrix]))
    t0 = time.time()
    for k in range(rounds):
        rng = np.random.default_rng(seed0 + k)
        read_len = int(rng.choice([60, 64, 75, 90, 100, 120, 150, 200]))
        L = int(rng.integers(1500, 9000))
        base = rng.choice(np.frombuffer(b"ACGT", np.uint8), L).astype(np.uint8)
        for _ in range(int(rng.integers(0, 14))):               # repeated blocks, some reverse-complemented
            ln = int(rng.integers(15, 140))
            src, dst = int(rng.integers(0, L - ln)), int(rng.integers(0, L - ln))
            blk = base[src:src + ln].copy()
            if rng.random() < 0.4:
                blk = COMP[blk[::-1]]
            base[dst:dst + ln] = blk
        circular = bool(rng.random() < 0.6)
        reads, start = damaged_reads(rng, base, n, read_len, float(rng.choice([0.1, 0.4])), int(rng.integers(1, 31)), int(rng.integers(0, 10)),
                                     two_share=float(rng.choice([0.0, 0.2])), junk_share=float(rng.choice([0.0, 0.1])))
        if matrix != "flat":                                      # deamination towards the ends of the sequenced strand
            for col, (frm, to) in ((0, (ord("C"), ord("T"))), (1, (ord("C"), ord("T"))), (2, (ord("C"), ord("T"))), (read_len - 1, (ord("G"), ord("A"))), (read_len - 2, (ord("G"), ord("A")))):
                hit = (reads[:, col] == frm) & (rng.random(n) < 0.3)
                reads[hit, col] = to
        flip = rng.random(n) < 0.5
        reads[flip] = COMP[reads[flip][:, ::-1]]
        off = np.arange(n + 1, dtype=np.int64) * read_len
        if nrich:
            hit = rng.random(L) < float(rng.choice([0.01, 0.05, 0.1, 0.1, 0.2]))
            base[hit] = rng.choice(np.frombuffer(b"YRYRMWVHDSBKN", np.uint8), int(hit.sum()))
            for _ in range(int(rng.integers(0, 8))):
                at = int(rng.integers(0, L - 13))
                base[at:at + int(rng.integers(2, 13))] = ord("N")
        refs = base.tobytes().decode()
        out = []
        for env in (None, "MIA_HIP_NO_DIAG_FILTER"):
            if env:
                os.environ[env] = "1"
            hip = mia_amd.MiaHip(0)
            if env:
                os.environ.pop(env)
            hip.set_pssm(PSSM)
            out.append(hip.pass1(refs, circular, reads.reshape(-1), off, -1))
            stat = (hip.pass1_filtered(), hip.pass1_anchored())
            hip.close()
            if env is None:
                decided = stat
                total_decided[0] += stat[0]; total_decided[1] += stat[1]
        for j, (x, y) in enumerate(zip(out[0], out[1])):
            assert np.array_equal(x, y), ("output", j, "seed", seed0 + k)
        if not quiet:
            print("round", k, "len", read_len, "L", L, "circular", circular, "filter/anchored", decided, "ok", round(time.time() - t0, 1), "s", flush=True)
    print("campaign done:", rounds, "configurations", "with N-rich references" if nrich else "", "no difference")
    return rounds * n, total_decided[0], total_decided[1]


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 7000, len(sys.argv) > 3 and sys.argv[3] == "nrich",
        matrix=sys.argv[4] if len(sys.argv) > 4 else "flat")
