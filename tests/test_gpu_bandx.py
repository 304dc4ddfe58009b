"""Differential test of the band pipeline for any matrix (csrc/bandx_body.h: k_bx_plan, k_bx_values, k_bx_trace) on
the GPU.  A context with every shortcut switched off (MIA_HIP_NO_DIAG_FILTER=1: the full-window DP kernels, which the
other suites pin to the reference's own answers) must return the same score, end points and script for every read --
with the flat matrix and with both aDNA matrices (reference matrices/ancient.submat.txt, ancient.submat.solexa.pe.txt;
depth-dependent, strand-specific: src/pssm.c:6-46, src/mia_main.c:179-184), on reads of both strands with C->T / G->A
damage at the ends, substitutions, indels of 1..12 bases, two indels, junk heads, jittered coordinates, random and
adversarial references, and around the origin of a circular reference."""
import os

import numpy as np
import pytest

from test_gpu_band import damaged_reads
from test_gpu_filter_stress import adversarial_reference

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MATS = ["flat", "ancient.submat.txt", "ancient.submat.solexa.pe.txt"]


def pssm(mod, spec):
    return mod.flat_pssm() if spec == "flat" else mod.read_pssm(os.path.join(GOLDEN, spec))


def deaminate(rng, reads, p0=0.3, lam=0.35):
    """C->T at distance i from the 5' end with probability p0 * exp(-lam * i), G->A mirrored at the 3' end"""
    n, m = reads.shape
    for i in range(min(m, 20)):
        p = p0 * np.exp(-lam * i)
        hit = (reads[:, i] == ord("C")) & (rng.random(n) < p)
        reads[hit, i] = ord("T")
        j = m - 1 - i
        hit = (reads[:, j] == ord("G")) & (rng.random(n) < p)
        reads[hit, j] = ord("A")
    return reads


def run_both(mod, spec, refs, reads, read_len, strand, as0, ae0, min_share):
    n = len(reads)
    off = np.arange(n + 1, dtype=np.int64) * read_len
    out, stats = [], []
    for shortcuts_off in (False, True):
        if shortcuts_off:
            os.environ["MIA_HIP_NO_DIAG_FILTER"] = "1"
        try:
            hip = mod.MiaHip(0)
        finally:
            os.environ.pop("MIA_HIP_NO_DIAG_FILTER", None)
        hip.set_pssm(pssm(mod, spec))
        hip.upload_reads(reads.reshape(-1), off, strand, np.ones(n, np.uint8), as0, ae0)
        hip.realign(refs, True)
        sc, a, e = hip.alignments()
        cols, rstart = hip.scripts()
        absolute = np.where(cols >= 0, cols.astype(np.int32) + rstart[:, None], cols.astype(np.int32))
        stats.append(hip.bx_stats()[0])
        out.append((sc, a, e, absolute))
        hip.close()
    for name, x, y in zip(("score", "start", "end", "script"), out[0], out[1]):
        bad = np.nonzero((x != y).reshape(n, -1).any(axis=1))[0]
        assert len(bad) == 0, (spec, name, len(bad), bad[:5], reads[bad[0]].tobytes(), int(as0[bad[0]]), int(strand[bad[0]]))
    assert sum(stats[1]) == 0 and sum(stats[0][1:]) >= min_share * n, stats
    return stats[0]


@pytest.mark.parametrize("spec", MATS)
@pytest.mark.parametrize("seed,read_len,max_gap", [(21, 100, 3), (22, 64, 6), (23, 150, 12), (25, 250, 8), (26, 36, 2)])
def test_random_reference(spec, seed, read_len, max_gap):
    import mia_amd
    rng = np.random.default_rng(seed)
    L = 16000
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    n = 100_000
    reads, start = damaged_reads(rng, ref, n, read_len, 0.3, max_gap, 5)
    reads = deaminate(rng, reads)
    jitter = rng.integers(-8, 9, n) * (rng.random(n) < 0.3)
    as0 = ((start + jitter) % L).astype(np.int32)
    ae0 = (as0 + read_len - 1).astype(np.int32)
    strand = (rng.random(n) < 0.5).astype(np.uint8)
    run_both(mia_amd, spec, ref.tobytes().decode(), reads, read_len, strand, as0, ae0, 0.4 if read_len >= 60 else 0.05)


@pytest.mark.parametrize("spec", MATS)
@pytest.mark.parametrize("seed,read_len", [(31, 100), (33, 130)])
def test_adversarial_reference(spec, seed, read_len):
    import mia_amd
    rng = np.random.default_rng(seed)
    L = 12000
    ref = adversarial_reference(rng, L)
    n = 120_000
    reads, start = damaged_reads(rng, ref, n, read_len, 0.4, 5, 4)
    reads = deaminate(rng, reads, p0=0.5)
    jitter = rng.integers(-6, 7, n) * (rng.random(n) < 0.3)
    as0 = ((start + jitter) % L).astype(np.int32)
    ae0 = (as0 + read_len - 1).astype(np.int32)
    strand = (rng.random(n) < 0.5).astype(np.uint8)
    run_both(mia_amd, spec, ref.tobytes().decode(), reads, read_len, strand, as0, ae0, 0.0)


@pytest.mark.parametrize("spec", MATS)
def test_reads_at_the_origin_of_a_circular_reference(spec):
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
    k = rng.integers(0, 5, n)
    for i in np.nonzero(fresh)[0]:
        pos = rng.choice(90, k[i], replace=False)
        reads[i, pos] = rng.choice(np.frombuffer(b"ACGT", np.uint8), k[i])
    reads = deaminate(rng, reads)
    as0 = start.astype(np.int32)
    ae0 = (as0 + 89).astype(np.int32)
    strand = (rng.random(n) < 0.5).astype(np.uint8)
    run_both(mia_amd, spec, ref.tobytes().decode(), reads, 90, strand, as0, ae0, 0.0)


@pytest.mark.parametrize("spec", MATS)
def test_typical_resequencing_batch_is_mostly_finished_early(spec):
    """1 % substitutions, 0.1 % indels, damage: the plan alone must finish the bulk, the band DPs nearly all the rest"""
    import mia_amd
    rng = np.random.default_rng(5)
    L = 16569
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    n = 200_000
    reads, start = damaged_reads(rng, ref, n, 100, 0.1, 3, 0, two_share=0.0, junk_share=0.0)
    sub = rng.random(reads.shape) < 0.01
    reads[sub] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(sub.sum()))
    reads = deaminate(rng, reads)
    as0 = (start % L).astype(np.int32)
    ae0 = (as0 + 99).astype(np.int32)
    strand = (rng.random(n) < 0.5).astype(np.uint8)
    st = run_both(mia_amd, spec, ref.tobytes().decode(), reads, 100, strand, as0, ae0, 0.9)
    assert st[1] > 0.4 * n, st
