"""GPU Myers bit-vector edit distance (mia_hip_myers) against the REAL reference's
myers_diff answers (tests/golden/myers_vectors.txt, dumped by oracle/_ref/ref_myers_driver)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_myers_vectors():
    import mia_amd
    lines = [l.rstrip("\n") for l in open(os.path.join(GOLDEN, "myers_vectors.txt"))]
    A, B, mode, maxd, exp = [], [], [], [], []
    for inp, out in zip(lines[0::2], lines[1::2]):
        m, d, a, b = inp.split(" ")
        A.append(a); B.append(b); mode.append(int(m)); maxd.append(int(d)); exp.append(int(out.split(" ")[0]))
    hip = mia_amd.MiaHip(0)
    got = hip.myers(A, B, mode, maxd)
    bad = [(i, int(got[i]), exp[i], mode[i], maxd[i], len(A[i]), len(B[i])) for i in range(len(exp)) if int(got[i]) != exp[i]]
    assert not bad, bad[:10]
    assert len(exp) >= 200 and any(e != 0xFFFFFFFF and e > 20 for e in exp)
    hip.close()


def _bits(c):
    return {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "S": 6, "W": 9, "R": 5, "Y": 10, "K": 12, "M": 3, "B": 14, "D": 13, "H": 11, "V": 7,
            "N": 15}.get(c.upper(), 0)


def test_myers_align_backtrace():
    """mia_hip_myers_align: distance from the kernel, rows from the host walk.  The row over seq_a must equal the
    reference's bt_a (golden); the row over seq_b -- which the reference leaves unterminated and the golden driver
    therefore does not print -- must spell seq_b, pair up with bt_a column by column, and cost exactly the distance."""
    import mia_amd
    lines = [l.rstrip("\n") for l in open(os.path.join(GOLDEN, "myers_vectors.txt"))]
    hip = mia_amd.MiaHip(0)
    n = 0
    for inp, out in zip(lines[0::2], lines[1::2]):
        m, d, a, b = inp.split(" ")
        want_d, want_a = out.split(" ")
        got_d, ra, rb = hip.myers_align(a, int(m), b, int(d))
        if int(want_d) == 0xFFFFFFFF:
            assert got_d is None
            continue
        assert (got_d, ra) == (int(want_d), want_a), inp[:80]
        if len(ra) != len(rb):
            # prefix modes only: a D-path that ran past the end of the sequence that need not be consumed puts that
            # sequence's terminator into its row (src/myers_align.c:26-32 has no bound there)
            assert int(m) != 0
            continue
        assert b.startswith(rb.replace("-", "")) and (int(m) == 2 or rb.replace("-", "") == b)   # mode 1: all of seq_b, a prefix of seq_a
        assert a.startswith(ra.replace("-", "")) and (int(m) == 1 or ra.replace("-", "") == a)   # mode 2: all of seq_a, a prefix of seq_b
        cost = sum(1 for x, y in zip(ra, rb) if x == "-" or y == "-" or not (_bits(x) & _bits(y)))
        assert cost == got_d, inp[:80]
        n += 1
    assert n > 100
    hip.close()
