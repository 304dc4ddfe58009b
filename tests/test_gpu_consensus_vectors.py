"""find_consensus (reference src/map_align.c:294-391) on the GPU against the reference's own answers: the 300 BaseCounts
vectors of tests/golden/cons_vectors.txt (dumped from the real find_consensus by tools/make_goldens.py: ties between
bases, the 50 % gap ratio, the code-1 and code-2 score thresholds, empty columns) are written into the tally buffer
through mia_hip_set_tally, one vector per column, and called by k_call_columns through mia_hip_consensus."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def vectors():
    with open(os.path.join(GOLDEN, "cons_vectors.txt")) as f:
        lines = [l.rstrip("\n") for l in f]
    out = []
    for inp, exp in zip(lines[0::2], lines[1::2]):
        v = [int(x) for x in inp.split()[1:]]
        out.append((v[0], v[1:], exp.split(" ")[1]))
    return out


@pytest.mark.parametrize("code", [1, 2])
def test_reference_vectors_through_k_call_columns(code):
    import mia_amd
    vec = [(v, e) for c, v, e in vectors() if c == code]
    assert len(vec) >= 100
    # every vector is followed by two sentinel columns that call 'A' and 'C', so that a '-' call (which the consensus
    # string skips, src/mia.c:597-599) shows as a missing character at a known place
    L = 3 * len(vec)
    t = np.zeros((mia_amd.TALLY_WORDS, L + 1), np.int32)
    expect = []
    for k, (v, e) in enumerate(vec):
        t[0:10, 3 * k] = v                      # As Cs Gs Ts gaps cov scoreA scoreC scoreG scoreT
        for j, b in ((1, 0), (2, 1)):
            t[b, 3 * k + j] = 1
            t[5, 3 * k + j] = 1
            t[6:10, 3 * k + j] = -600
            t[6 + b, 3 * k + j] = 200
        expect.append(("" if e == "-" else e) + "AC")
    hip = mia_amd.MiaHip(0)
    hip.set_pssm(mia_amd.flat_pssm())
    hip.set_tally(t)
    got = hip.consensus(code)
    hip.close()
    want = "".join(expect)
    if got != want:                              # name the first vector that differs
        p = 0
        for k, x in enumerate(expect):
            assert got[p:p + len(x)] == x, (k, vec[k], got[p:p + 3])
            p += len(x)
    assert got == want
    assert any(e == "-" for _, e in vec) and any(e == "N" for _, e in vec)
