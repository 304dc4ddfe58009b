"""Differential test of the banded DP (csrc/band_body.h, k_band_align) on the GPU.  The reads the diagonal filter leaves
over -- three and more substitutions, insertions, deletions -- are aligned inside a band of at most 32 diagonals when
their ten-mer anchors allow it.  A context with the banded DP switched off (the same reads through the full-window
kernels, which the other suites pin to the oracle) must return the same score, end points and script for every read:
on plain random references, on the adversarial material of test_gpu_filter_stress (tandem repeats, two-letter
stretches, copied blocks: gap placements tie), with indels of 1..12 bases anywhere including next to the read ends,
with two indels per read, with junk heads (late starts) and with pass-1 coordinates that are a few columns off."""
import os

import numpy as np
import pytest

from test_gpu_filter_stress import adversarial_reference

pytestmark = pytest.mark.gpu


def damaged_reads(rng, ref, n, read_len, indel_share, max_gap, subs_max, two_share=0.1, junk_share=0.05):
    L = len(ref)
    start = rng.integers(0, L, n)
    idx = (start[:, None] + np.arange(read_len + 2 * max_gap + 8)[None, :]) % L
    tmpl = ref[idx]
    reads = tmpl[:, :read_len].copy()
    has_gap = rng.random(n) < indel_share
    two = rng.random(n) < two_share
    for i in np.nonzero(has_gap)[0]:
        row = tmpl[i]
        for _ in range(2 if two[i] else 1):
            edge = rng.random() < 0.2
            p = int(rng.choice([1, 2, 3, 9, 10, 11, read_len - 12, read_len - 11, read_len - 3, read_len - 2])) if edge else int(rng.integers(1, read_len - 1))
            g = int(rng.integers(1, max_gap + 1))
            if rng.random() < 0.5:
                row = np.concatenate([row[:p], row[p + g:]])
            else:
                row = np.concatenate([row[:p], rng.choice(np.frombuffer(b"ACGT", np.uint8), g), row[p:]])
        reads[i] = row[:read_len]
    k = rng.integers(0, subs_max + 1, n)
    for i in np.nonzero(k)[0]:
        pos = rng.choice(read_len, k[i], replace=False)
        reads[i, pos] = rng.choice(np.frombuffer(b"ACGT", np.uint8), k[i])
    for i in np.nonzero(rng.random(n) < junk_share)[0]:
        j = int(rng.integers(1, 15))
        reads[i, :j] = rng.choice(np.frombuffer(b"ACGT", np.uint8), j)
    return reads.astype(np.uint8), start


def run_both(mod, refs, reads, read_len, as0, ae0, min_band):
    n = len(reads)
    off = np.arange(n + 1, dtype=np.int64) * read_len
    out, share = [], []
    for band_off in (False, True):
        if band_off:
            os.environ["MIA_HIP_NO_BAND_DP"] = "1"
        try:
            hip = mod.MiaHip(0)
        finally:
            os.environ.pop("MIA_HIP_NO_BAND_DP", None)
        hip.set_pssm(mod.flat_pssm())
        hip.upload_reads(reads.reshape(-1), off, np.zeros(n, np.uint8), np.ones(n, np.uint8), as0, ae0)
        hip.realign(refs, True)
        sc, a, e = hip.alignments()
        cols, rstart = hip.scripts()
        absolute = np.where(cols >= 0, cols.astype(np.int32) + rstart[:, None], cols.astype(np.int32))
        done = hip.band_stats()[0] + sum(hip.bx_stats()[0][1:])       # round 1's k_band_align, or the band pipeline for any matrix
        won = hip.filter_stats()[1]
        share.append((done, won))
        out.append((sc, a, e, absolute))
        hip.close()
    for name, x, y in zip(("score", "start", "end", "script"), out[0], out[1]):
        bad = np.nonzero((x != y).reshape(n, -1).any(axis=1))[0]
        assert len(bad) == 0, (name, len(bad), bad[:5], reads[bad[0]].tobytes(), int(as0[bad[0]]))
    assert share[1][0] == 0 and share[0][0] >= min_band * (n - share[0][1]), share


@pytest.mark.parametrize("seed,read_len,max_gap", [(21, 100, 3), (22, 64, 6), (23, 150, 12), (24, 61, 2), (25, 250, 8), (26, 36, 2), (27, 45, 3), (28, 52, 4)])
def test_random_reference(seed, read_len, max_gap):
    import mia_amd
    rng = np.random.default_rng(seed)
    L = 16000
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    n = 120_000
    reads, start = damaged_reads(rng, ref, n, read_len, 0.5, max_gap, 6)
    jitter = rng.integers(-8, 9, n) * (rng.random(n) < 0.3)
    as0 = ((start + jitter) % L).astype(np.int32)
    ae0 = (as0 + read_len - 1).astype(np.int32)
    run_both(mia_amd, ref.tobytes().decode(), reads, read_len, as0, ae0, 0.25 if read_len >= 60 else 0.05)


@pytest.mark.parametrize("seed,read_len", [(31, 100), (32, 72), (33, 130), (34, 40)])
def test_adversarial_reference(seed, read_len):
    import mia_amd
    rng = np.random.default_rng(seed)
    L = 12000
    ref = adversarial_reference(rng, L)
    n = 150_000
    reads, start = damaged_reads(rng, ref, n, read_len, 0.4, 5, 4)
    jitter = rng.integers(-6, 7, n) * (rng.random(n) < 0.3)
    as0 = ((start + jitter) % L).astype(np.int32)
    ae0 = (as0 + read_len - 1).astype(np.int32)
    run_both(mia_amd, ref.tobytes().decode(), reads, read_len, as0, ae0, 0.0)


def test_reads_at_the_origin_of_a_circular_reference():
    """windows that are cut at the reference start (column 0 is a real column) and reads that run over the wrap"""
    import mia_amd
    rng = np.random.default_rng(41)
    L = 3000
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    n = 60_000
    reads, start = damaged_reads(rng, ref, n, 90, 0.5, 4, 4)
    start = np.where(rng.random(n) < 0.7, rng.integers(-95, 60, n) % L, start)
    idx = (start[:, None] + np.arange(90)[None, :]) % L
    fresh = rng.random(n) < 0.5
    reads[fresh] = ref[idx[fresh]]
    k = rng.integers(3, 7, n)
    for i in np.nonzero(fresh)[0]:
        pos = rng.choice(90, k[i], replace=False)
        reads[i, pos] = rng.choice(np.frombuffer(b"ACGT", np.uint8), k[i])
    as0 = start.astype(np.int32)
    ae0 = (as0 + 89).astype(np.int32)
    run_both(mia_amd, ref.tobytes().decode(), reads, 90, as0, ae0, 0.0)


@pytest.mark.parametrize("seed,read_len", [(51, 100), (52, 70), (53, 140)])
def test_tally_of_one_gap_reads(seed, read_len):
    """k_band_align tells the tally where the single gap of a read is (ST_ONEGAP), and the binned tally then takes such
    reads one per lane instead of one per wavefront.  Column tallies, ref->gaps, the consensus (with its inserts) and the
    insert tallies must be what the script-walking path gives with the banded DP switched off: reads with one deletion,
    one insertion, at all distances from the ends, both strands, dropped and kept, reads over the origin."""
    import mia_amd
    rng = np.random.default_rng(seed)
    L = 9000
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    n = 200_000
    reads, start = damaged_reads(rng, ref, n, read_len, 0.6, 5, 3, two_share=0.1, junk_share=0.02)
    # some inserts shared by many reads, so that the consensus calls them
    for p in rng.integers(200, L - 200, 12):
        ins = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(1, 4)))
        hit = np.nonzero((start < p - 15) & (start + read_len > p + 15 + len(ins)))[0]
        for i in hit[: len(hit) * 3 // 4]:
            o = int(p - start[i])
            reads[i] = np.concatenate([ref[start[i]: start[i] + o], ins, ref[start[i] + o: start[i] + read_len]])[:read_len]
    strand = (rng.random(n) < 0.5).astype(np.uint8)
    off = np.arange(n + 1, dtype=np.int64) * read_len
    as0 = (start % L).astype(np.int32)
    ae0 = (as0 + read_len - 1).astype(np.int32)
    refs = ref.tobytes().decode()
    out = []
    # the one-read-per-lane paths for one-gap and over-the-origin reads exist for the count-only ("linear") tally; with
    # MIA_HIP_NO_LINEAR_TALLY=1 those reads walk their scripts one per wavefront and every base adds its four score words
    # (round 2: the count-only tally takes pure-diagonal and one-gap reads through bit-sliced vertical counters;
    # MIA_HIP_DEBUG_SKIP=4096 is the same tally with one LDS atomic per base)
    for env, val in ((None, ""), ("MIA_HIP_NO_BAND_DP", "1"), ("MIA_HIP_NO_LINEAR_TALLY", "1"), ("MIA_HIP_DEBUG_SKIP", "4096")):
        if env:
            os.environ[env] = val
        try:
            hip = mia_amd.MiaHip(0)
        finally:
            if env:
                os.environ.pop(env, None)
        hip.set_pssm(mia_amd.flat_pssm())
        hip.upload_reads(reads.reshape(-1), off, strand, np.ones(n, np.uint8), as0, ae0)
        hip.realign(refs, True)
        sc, a, e = hip.alignments()
        cut = hip.score_cut(sc, np.full(n, read_len, np.int32))
        hip.cull(0, cut[0] if cut[0] > 0 else 100.0, cut[1], 0)
        hip.tally()
        t, g = hip.get_tally()
        cons = hip.consensus(1)
        it = hip.ins_tally()
        out.append((t, g, cons, it, hip.band_stats()[0] + sum(hip.bx_stats()[0][1:])))
        hip.close()
    assert out[0][4] > 0.2 * n and out[1][4] == 0 and out[3][4] == out[0][4]
    for other in out[1:]:
        assert np.array_equal(out[0][0], other[0]) and np.array_equal(out[0][1], other[1])
        assert out[0][2] == other[2] and out[0][1].max() > 0          # (the same consensus; insert events did reach ref->gaps)
        for x, y in zip(out[0][3], other[3]):
            assert np.array_equal(x, y)
