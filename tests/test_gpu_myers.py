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
