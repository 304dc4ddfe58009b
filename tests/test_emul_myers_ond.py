"""csrc/myers_ond_body.h -- the D-path recurrence k_myers_ond runs per lane (packed 32-character snake steps, the cell rule,
the walk back through the table) -- driven row by row on the CPU (tests/emul/emu_myers_ond.cpp) against the REAL reference's
myers_diff answers: tests/golden/myers_vectors.txt (distance and bt_a, dumped by oracle/_ref/ref_myers_driver).  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def emu(oracle_build):
    out = os.path.join(oracle_build, "libmia_emul_myers_ond.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", "mapping-iterative-assembler_amd/csrc", "-o", out,
                    "tests/emul/emu_myers_ond.cpp"], cwd=ROOT, check=True)
    lib = C.CDLL(out)
    lib.emu_myers_ond.restype = C.c_uint32
    lib.emu_myers_ond.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    return lib


def vectors():
    lines = [l.rstrip("\n") for l in open(os.path.join(GOLDEN, "myers_vectors.txt"))]
    for inp, out in zip(lines[0::2], lines[1::2]):
        m, d, a, b = inp.split(" ")
        want_d, want_a = out.split(" ")
        yield int(m), int(d), a.encode(), b.encode(), int(want_d), want_a


def run(emu, a, mode, b, maxd, cap):
    ra, rb = C.create_string_buffer(len(a) + len(b) + 8), C.create_string_buffer(len(a) + len(b) + 8)
    d = emu.emu_myers_ond(a, mode, b, maxd, cap, ra, rb)
    return d, ra.value.decode(), rb.value.decode()


def test_reference_vectors(emu):
    n = 0
    for mode, maxd, a, b, want_d, want_a in vectors():
        d, ra, _rb = run(emu, a, mode, b, maxd, 1 << 20)
        assert d == want_d, (mode, maxd, len(a), len(b))
        if want_d != 0xFFFFFFFF:
            assert ra == want_a
            n += 1
    assert n > 100


def test_cap_below_the_distance_hands_the_pair_on(emu):
    """a cap below maxd: the distance if it lies under the cap, else 0xFFFFFFFE (the bit-vector kernel's business)"""
    seen = [0, 0]
    for mode, maxd, a, b, want_d, _ in vectors():
        if want_d == 0xFFFFFFFF or want_d < 2:
            continue
        d, _, _ = run(emu, a, mode, b, maxd, want_d)            # rows 0 .. want_d - 1 only
        assert d == 0xFFFFFFFE
        d, _, _ = run(emu, a, mode, b, maxd, want_d + 1)
        assert d == want_d
        seen[0] += 1
    assert seen[0] > 50


def _host_distance(a, b, mode):
    """plain dynamic programme over bitmaps (D[i][0] = i, D[0][j] = j)"""
    bits = {c: v for c, v in zip(b"ACGTUSWRYKMBDHVN", [1, 2, 4, 8, 8, 6, 9, 5, 10, 12, 3, 14, 13, 11, 7, 15])}
    A = np.array([bits.get(c & ~32, 0) for c in a], np.int64)
    Bv = [bits.get(c & ~32, 0) for c in b]
    col = np.arange(len(a) + 1)
    best_last_row = col[-1]
    for j, bj in enumerate(Bv, 1):
        new = np.empty_like(col)
        new[0] = j
        sub = col[:-1] + ((A & bj) == 0)
        up = col[1:] + 1
        m = np.minimum(sub, up)
        # the in-column dependency new[i] = min(m[i-1], new[i-1] + 1): a running minimum of (m - i) + i
        run = np.minimum.accumulate(np.concatenate(([new[0]], m)) - np.arange(len(a) + 1)) + np.arange(len(a) + 1)
        col = run
        best_last_row = min(best_last_row, col[-1])
    return int(col[-1]) if mode == 0 else int(col.min()) if mode == 1 else int(best_last_row)


def test_long_pairs_against_a_plain_dp(emu):
    """pairs of a few thousand characters (snakes of hundreds of characters, word-unaligned starts on both sides, IUPAC codes,
    characters no bitmap knows): the distance against a plain O(len^2) programme, the rows against the distance"""
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGTACGTACGTACGTNRYKMSWBDHVX", np.uint8)
    bits = {c: v for c, v in zip(b"ACGTUSWRYKMBDHVN", [1, 2, 4, 8, 8, 6, 9, 5, 10, 12, 3, 14, 13, 11, 7, 15])}
    for it in range(24):
        la = int(rng.integers(300, 3000))
        a = alpha[rng.integers(0, 16 if it % 2 else len(alpha), la)].copy()
        b = list(a)
        for _ in range(int(rng.integers(0, 40))):
            p = int(rng.integers(0, len(b)))
            u = rng.random()
            if u < 0.4:
                b[p] = alpha[rng.integers(0, 16)]
            elif u < 0.7:
                del b[p:p + int(rng.integers(1, 4))]
            else:
                b[p:p] = list(alpha[rng.integers(0, 16, int(rng.integers(1, 4)))])
        mode = it % 3
        if mode == 1:
            a = np.concatenate([a, alpha[rng.integers(0, 16, 50)]])      # seq_b a prefix of seq_a
        if mode == 2:
            b = b + list(alpha[rng.integers(0, 16, 50)])                 # seq_a a prefix of seq_b
        a, b = bytes(a), bytes(np.array(b, np.uint8))
        want = _host_distance(a, b, mode)
        d, ra, rb = run(emu, a, mode, b, 100000, 1 << 20)
        assert d == want, (it, mode, len(a), len(b))
        if len(ra) == len(rb):
            cost = sum(1 for x, y in zip(ra.encode(), rb.encode()) if x == 45 or y == 45 or not (bits.get(x & ~32, 0) & bits.get(y & ~32, 0)))
            assert cost == d
