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
        early = won + sum(hip.bx_stats()[0][2:])       # finished without the full-window kernels: filter / plan, band DPs
        assert early == 0 if is_off else early > 0.05 * n, (seen, won, early)
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


@pytest.mark.parametrize("seed,read_len,circular,n_rate", [(21, 100, True, 0.0), (22, 64, True, 0.0), (23, 150, False, 0.0), (24, 90, True, 0.0),
                                                          (25, 100, True, 0.10), (26, 64, True, 0.05), (27, 150, False, 0.10), (28, 100, True, 0.01)])
def test_pass1_anchored_windows_at_their_limits(seed, read_len, circular, n_rate):
    """anchored pass 1 (k_pass1_anchor / k_pass1_select): reads with 0..9 substitutions (the budget of nine 10-mer blocks
    ends at eight defects), deletions and insertions of 1..40 bases (a path may stray 29 diagonals from its anchor),
    reads that hang over the ends of a linear reference, reads across the origin of a circular one, a reference with
    repeated blocks (several anchor clusters, stray 10-mers on the other strand).  n_rate > 0: after the reads are drawn
    that share of the reference's columns (and a few stretches) become ambiguity codes, as in mt311 -- the anchors then come
    from a table that lists such 10-mers under every spelling (bandx_body.h, N COLUMNS).  Same answers as the whole-strand DP."""
    import mia_amd
    rng = np.random.default_rng(seed)
    L = 6000
    base = rng.choice(np.frombuffer(b"ACGT", np.uint8), L).astype(np.uint8)
    # repeated blocks, some reverse-complemented: anchors in several places and on both strands
    for _ in range(12):
        ln = int(rng.integers(15, 140))
        src, dst = int(rng.integers(0, L - ln)), int(rng.integers(0, L - ln))
        blk = base[src:src + ln].copy()
        if rng.random() < 0.4:
            blk = COMP[blk[::-1]]
        hit = rng.random(ln) < 0.03
        blk[hit] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(hit.sum()))
        base[dst:dst + ln] = blk
    n = 40_000
    ext = np.concatenate([base, base]) if circular else np.concatenate([rng.choice(np.frombuffer(b"ACGT", np.uint8), 60), base,
                                                                        rng.choice(np.frombuffer(b"ACGT", np.uint8), 160)]).astype(np.uint8)
    start = rng.integers(0, L if circular else L + 120 - read_len, n)
    reads = np.empty((n, read_len), np.uint8)
    for i in range(n):
        t = ext[start[i]: start[i] + read_len + 45].copy()
        kind = i % 5
        if kind == 1:                                   # deletion from the read's point of view
            p, g = int(rng.integers(5, read_len - 5)), int(rng.choice([1, 2, 3, 5, 10, 17, 28, 29, 30, 40]))
            t = np.concatenate([t[:p], t[p + g:]])
        elif kind == 2:                                 # insertion
            p, g = int(rng.integers(5, read_len - 5)), int(rng.choice([1, 2, 3, 5, 10, 17, 28, 29, 30]))
            t = np.concatenate([t[:p], rng.choice(np.frombuffer(b"ACGT", np.uint8), g), t[p:]])
        r = t[:read_len].copy()
        k = int(rng.integers(0, 10)) if kind != 4 else int(rng.integers(0, 3))
        for p in rng.choice(read_len, k, replace=False):
            r[p] = rng.choice([b for b in b"ACGT" if b != r[p]])
        reads[i] = r
    flip = rng.random(n) < 0.5
    reads[flip] = COMP[reads[flip][:, ::-1]]
    off = np.arange(n + 1, dtype=np.int64) * read_len
    if n_rate > 0:
        hit = rng.random(L) < n_rate
        base[hit] = rng.choice(np.frombuffer(b"YRYRMWVHDSBKN", np.uint8), int(hit.sum()))
        for _ in range(6):
            at = int(rng.integers(0, L - 13))
            base[at:at + int(rng.integers(2, 13))] = ord("N")
    refs = base.tobytes().decode()
    out = []
    for is_off, hip in contexts(mia_amd):
        out.append(hip.pass1(refs, circular, reads.reshape(-1), off, -1))
        if not is_off:
            assert hip.pass1_anchored() > (0.15 if n_rate == 0 else 0.10) * n, (hip.pass1_filtered(), hip.pass1_anchored())
    names = ("score", "rc", "as", "ae", "flags")
    for nm, x, y in zip(names, out[0], out[1]):
        bad = np.nonzero(x != y)[0]
        assert len(bad) == 0, (nm, bad[:5], x[bad[:5]], y[bad[:5]])
