"""Adversarial differential test of the diagonal filter (csrc/diag_filter.h) on the GPU: references made of the material
the filter's rules exist for -- two-letter alphabets, tandem repeats with units of 1..13 bases, blocks copied a few
columns off, long homopolymers -- where many diagonals are nearly as good as the true one and where one short gap can
buy back two mismatches.  Reads carry 0..3 substitutions at structured positions (adjacent, at the ends, one per
repeat unit) and some carry an indel.  A context with the filter switched off (every read through the DP kernels) must
return the same score, end points and script for every one of the reads, in realign and in pass 1."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b


def adversarial_reference(rng, L):
    parts = []
    total = 0
    while total < L:
        kind = int(rng.integers(0, 6))
        ln = int(rng.integers(40, 400))
        if kind == 0:      # two letters only
            seg = rng.choice(np.frombuffer(b"AC", np.uint8), ln)
        elif kind == 1:    # tandem repeat
            unit = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(1, 14)))
            seg = np.tile(unit, ln // len(unit) + 1)[:ln]
        elif kind == 2:    # tandem repeat with sparse substitutions
            unit = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(2, 9)))
            seg = np.tile(unit, ln // len(unit) + 1)[:ln].copy()
            hit = rng.random(ln) < 0.04
            seg[hit] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(hit.sum()))
        elif kind == 3 and parts:   # a block seen before, again a few columns further on
            src = parts[int(rng.integers(0, len(parts)))]
            seg = src[: min(len(src), ln)].copy()
        elif kind == 4:    # homopolymer with rare interruptions
            seg = np.full(ln, rng.choice(np.frombuffer(b"ACGT", np.uint8)), np.uint8)
            hit = rng.random(ln) < 0.05
            seg[hit] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(hit.sum()))
        else:
            seg = rng.choice(np.frombuffer(b"ACGT", np.uint8), ln)
        parts.append(seg.astype(np.uint8))
        total += len(seg)
    return np.concatenate(parts)[:L]


def make_reads(rng, ref, n, read_len):
    L = len(ref)
    start = rng.integers(0, L, n)
    idx = (start[:, None] + np.arange(read_len + 4)[None, :]) % L
    tmpl = ref[idx]
    reads = tmpl[:, :read_len].copy()
    k = rng.integers(0, 4, n)                       # substitutions per read
    mode = rng.integers(0, 4, n)
    for i in range(n):
        if k[i] == 0:
            continue
        if mode[i] == 0:                            # adjacent
            p0 = int(rng.integers(0, read_len - 4))
            pos = p0 + np.arange(k[i])
        elif mode[i] == 1:                          # at the ends
            pos = rng.choice(np.array([0, 1, 2, 3, read_len - 4, read_len - 3, read_len - 2, read_len - 1]), k[i], replace=False)
        elif mode[i] == 2:                          # a few columns apart
            p0 = int(rng.integers(0, read_len - 12))
            pos = p0 + np.sort(rng.choice(12, k[i], replace=False))
        else:
            pos = rng.choice(read_len, k[i], replace=False)
        for p in pos:
            reads[i, p] = rng.choice([b for b in b"ACGT" if b != reads[i, p]])
    # every eighth read carries an indel
    for i in range(0, n, 8):
        p = int(rng.integers(3, read_len - 6))
        g = int(rng.integers(1, 4))
        if i % 16 == 0:
            reads[i] = np.concatenate([tmpl[i, :p], tmpl[i, p + g:]])[:read_len]
        else:
            ins = rng.choice(np.frombuffer(b"ACGT", np.uint8), g)
            reads[i] = np.concatenate([tmpl[i, :p], ins, tmpl[i, p:]])[:read_len]
    return reads.astype(np.uint8), start


def contexts(mod):
    for off in (False, True):
        if off:
            os.environ["MIA_HIP_NO_DIAG_FILTER"] = "1"
        try:
            hip = mod.MiaHip(0)
        finally:
            os.environ.pop("MIA_HIP_NO_DIAG_FILTER", None)
        hip.set_pssm(mod.flat_pssm())
        yield off, hip
        hip.close()


@pytest.mark.parametrize("seed,read_len", [(1, 100), (2, 64), (3, 65), (4, 36), (5, 150)])
def test_realign_on_adversarial_references(seed, read_len):
    import mia_amd
    rng = np.random.default_rng(seed)
    L = 12000
    ref = adversarial_reference(rng, L)
    if seed % 2:                                     # a sprinkle of ambiguity codes (N for the aligner), under the 2 % at which
        ref = ref.copy()                             # the library stops trying the filter
        ref[rng.choice(L, L // 150, replace=False)] = ord("N")
    n = 250_000
    reads, start = make_reads(rng, ref, n, read_len)
    reads[reads == ord("N")] = ord("A")
    off = np.arange(n + 1, dtype=np.int64) * read_len
    # pass-1 coordinates a few columns off the truth now and then: the window is what the filter sees
    jitter = rng.integers(-6, 7, n) * (rng.random(n) < 0.3)
    as0 = ((start + jitter) % L).astype(np.int32)
    ae0 = (as0 + read_len - 1).astype(np.int32)
    refs = ref.tobytes().decode()
    out = []
    for is_off, hip in contexts(mia_amd):
        hip.upload_reads(reads.reshape(-1), off, np.zeros(n, np.uint8), np.ones(n, np.uint8), as0, ae0)
        hip.realign(refs, True)
        sc, a, e = hip.alignments()
        cols, rstart = hip.scripts()
        absolute = np.where(cols >= 0, cols.astype(np.int32) + rstart[:, None], cols.astype(np.int32))
        seen, won = hip.filter_stats()[:2]
        assert won == 0 if is_off else won > 0.05 * n, (seen, won)
        out.append((sc, a, e, absolute))
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("seed", [11, 12])
def test_pass1_on_adversarial_references(seed):
    import mia_amd
    rng = np.random.default_rng(seed)
    L = 5000
    ref = adversarial_reference(rng, L)
    if seed % 2:
        ref = ref.copy()
        ref[rng.choice(L, L // 150, replace=False)] = ord("N")
    n = 30_000
    reads, _ = make_reads(rng, ref, n, 80)
    reads[reads == ord("N")] = ord("A")
    flip = rng.random(n) < 0.5
    reads[flip] = COMP[reads[flip][:, ::-1]]
    off = np.arange(n + 1, dtype=np.int64) * 80
    refs = ref.tobytes().decode()
    out = []
    for is_off, hip in contexts(mia_amd):
        out.append(hip.pass1(refs, True, reads.reshape(-1), off, -1))
        decided = hip.pass1_filtered()
        assert decided == 0 if is_off else decided >= 0, decided
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y)
