"""A longer differential campaign for the banded DP than the test suite has time for: many seeds of
tests/test_gpu_band.py's generators (random and adversarial references, read lengths 30-250, indels up to 12, jittered
pass-1 coordinates), banded DP on against off, every read's score, end points and script.
usage: band_campaign.py [rounds [first seed [MIA_HIP_NO_DIAG_FILTER [matrix]]]]   (third argument: the switch that defines
"off"; MIA_HIP_NO_DIAG_FILTER takes every shortcut out, i.e. compares the band pipeline with the full-window DP kernels;
fourth: flat (default), ancient, solexa -- the position-specific matrices, both strands mixed, aDNA damage on the reads;
fifth: nrich -- after the reads are drawn, 3-20 % of the reference columns (and a few stretches of 2-12) become
ambiguity codes, as in mt311; fewn -- a handful of columns only (fewer than one in 500: an assembly's consensus where coverage is thin),
the case in which the table spells out N columns AND the quick plan answers for the windows that hold none)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mia_amd  # noqa: E402
from test_gpu_band import damaged_reads, run_both  # noqa: E402
from test_gpu_filter_stress import adversarial_reference  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def run(rounds=40, seed0=1000, switch="MIA_HIP_NO_BAND_DP", matrix="flat", nrich=False, n=100_000, quiet=False):
    """returns (reads compared, reads the band pipeline finished or placed)"""
    planned = [0, 0]
    PSSM = mia_amd.flat_pssm() if matrix == "flat" else mia_amd.read_pssm(os.path.join(GOLDEN, {"ancient": "ancient.submat.txt", "solexa": "ancient.submat.solexa.pe.txt"}[matrix]))

    def compare(refs, reads, read_len, as0, ae0, rc=None):
        n = len(reads)
        rc = np.zeros(n, np.uint8) if rc is None else rc
        off = np.arange(n + 1, dtype=np.int64) * read_len
        out = []
        for env in (None, switch):
            if env:
                os.environ[env] = "1"
            elif matrix != "flat":
                os.environ["MIA_HIP_QUICK_PLAN"] = "2"      # (with the fine blocks on, the quick plan starts at two million reads by itself: here always)
            hip = mia_amd.MiaHip(0)
            if env:
                os.environ.pop(env)
            os.environ.pop("MIA_HIP_QUICK_PLAN", None)
            hip.set_pssm(PSSM)
            hip.upload_reads(reads.reshape(-1), off, rc, np.ones(n, np.uint8), as0, ae0)
            hip.realign(refs, True)
            if not env:
                bx = hip.bx_stats()[0]
                planned[0] += int(sum(bx[1:4])); planned[1] += n
            sc, a, e = hip.alignments()
            cols, rstart = hip.scripts()
            out.append((sc, a, e, np.where(cols >= 0, cols.astype(np.int32) + rstart[:, None], cols.astype(np.int32))))
            hip.close()
        for name, x, y in zip(("score", "start", "end", "script"), out[0], out[1]):
            assert np.array_equal(x, y), name

    t0 = time.time()
    reads_total = 0
    for k in range(rounds):
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
        if nrich == "fewn":
            ref = ref.copy()
            codes = np.frombuffer(b"YRMWSKNNNN", np.uint8)
            k = int(rng.integers(1, max(2, L // 500)))
            ref[rng.integers(0, L, k)] = rng.choice(codes, k)
        elif nrich:
            ref = ref.copy()
            codes = np.frombuffer(b"YRYRMWVHDSBKN", np.uint8)
            hit = rng.random(L) < float(rng.choice([0.03, 0.1, 0.1, 0.2]))
            ref[hit] = rng.choice(codes, int(hit.sum()))
            for _ in range(int(rng.integers(0, 12))):
                at = int(rng.integers(0, L - 13))
                ref[at:at + int(rng.integers(2, 13))] = ord("N")
        if matrix != "flat":
            # aDNA damage on the stored read (C->T towards one end, G->A towards the other, by strand) and a random strand flag
            rc = (rng.random(n) < 0.5).astype(np.uint8)
            pos = np.arange(read_len)
            p5 = 0.30 * np.exp(-0.35 * pos)[None, :]
            p3 = p5[:, ::-1]
            u, v = rng.random(reads.shape), rng.random(reads.shape)
            fw = rc[:, None] == 0
            reads = np.where((reads == ord("C")) & (u < np.where(fw, p5, 0)), ord("T"), reads)
            reads = np.where((reads == ord("G")) & (v < np.where(fw, p3, 0)), ord("A"), reads)
            reads = np.where((reads == ord("G")) & (u < np.where(~fw, p3, 0)), ord("A"), reads)
            reads = np.where((reads == ord("C")) & (v < np.where(~fw, p5, 0)), ord("T"), reads).astype(np.uint8)
            compare(ref.tobytes().decode(), reads, read_len, as0, ae0, rc)
        elif switch == "MIA_HIP_NO_BAND_DP" and not nrich:
            run_both(mia_amd, ref.tobytes().decode(), reads, read_len, as0, ae0, 0.0)
        else:
            compare(ref.tobytes().decode(), reads, read_len, as0, ae0)
        reads_total += n
        if not quiet:
            print("round", k, "seed", seed, "len", read_len, "L", L, "ok", round(time.time() - t0, 1), "s", flush=True)
    print("campaign done:", matrix, "matrix,", ("a few N columns," if nrich == "fewn" else "N-rich references,") if nrich else "", reads_total, "reads, no difference;", planned[0], "of", planned[1],
          "finished or placed by the band pipeline")
    return reads_total, planned[0]


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1000,
        sys.argv[3] if len(sys.argv) > 3 else "MIA_HIP_NO_BAND_DP", sys.argv[4] if len(sys.argv) > 4 else "flat",
        (len(sys.argv) > 5 and sys.argv[5] in ("nrich", "fewn")) and (sys.argv[5] if sys.argv[5] == "fewn" else True))
