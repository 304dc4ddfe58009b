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


def test_lane_kernel_equals_systolic_kernel():
    """k_myers_lanes (one pair per lane, seq_a up to 320 characters, four bit planes instead of a Peq table) against k_myers
    (one pair per wavefront; MIA_HIP_MYERS_NO_LANES=1), which the reference's vectors pin: 30 000 random pairs -- lengths 0..340
    on either side (around the 64-row block edges and the 320-row limit), IUPAC codes and characters no bitmap knows, edits
    from none to everything, all three modes, maxd from 1 to beyond the sum of the lengths."""
    import mia_amd
    rng = np.random.default_rng(99)
    alpha = np.frombuffer(b"ACGTACGTACGTNRYKMSWBDHVacgtnX-", np.uint8)
    A, B, mode, maxd = [], [], [], []
    edge = [0, 1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 255, 256, 257, 319, 320, 321, 340]
    for i in range(30_000):
        la = int(rng.choice(edge)) if i % 3 == 0 else int(rng.integers(0, 341))
        a = alpha[rng.integers(0, len(alpha), la)].copy()
        kind = i % 5
        if kind == 0:
            b = alpha[rng.integers(0, len(alpha), int(rng.integers(0, 341)))].copy()          # unrelated
        else:
            b = list(a)
            for _ in range(int(rng.integers(0, 1 + la // (3 if kind == 1 else 12)))):
                if not b:
                    break
                p = int(rng.integers(0, len(b)))
                u = rng.random()
                if u < 0.4:
                    b[p] = alpha[rng.integers(0, len(alpha))]
                elif u < 0.7:
                    del b[p]
                else:
                    b.insert(p, alpha[rng.integers(0, len(alpha))])
            if kind == 2 and len(b) > 20:
                b = b[int(rng.integers(0, 10)):len(b) - int(rng.integers(0, 10))]                 # overhangs (modes 1 and 2)
            b = np.array(b[:400], np.uint8)
        A.append(a.tobytes().decode("latin1")); B.append(b.tobytes().decode("latin1"))
        mode.append(int(rng.integers(0, 3)))
        maxd.append(int(rng.choice([1, 2, 5, 30, 100, 700, 100000])))
    hip = mia_amd.MiaHip(0)
    got = hip.myers(A, B, mode, maxd)
    hip.close()
    os.environ["MIA_HIP_MYERS_NO_LANES"] = "1"
    try:
        ref = mia_amd.MiaHip(0)
    finally:
        os.environ.pop("MIA_HIP_MYERS_NO_LANES")
    want = ref.myers(A, B, mode, maxd)
    ref.close()
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, [(int(i), int(got[i]), int(want[i]), mode[i], maxd[i], len(A[i]), len(B[i])) for i in bad[:8]]
    assert (want != 0xFFFFFFFF).sum() > 5000 and (want == 0xFFFFFFFF).sum() > 1000


def test_packed_entry_point_on_the_reference_vectors():
    """mia_hip_myers_packed (round 4: the pre-packed batch form, no strlen / packing inside the call) on every golden pair whose
    seq_a fits one lane (up to 320 characters): the reference's myers_diff answers again; a longer seq_a is refused."""
    import mia_amd
    lines = [l.rstrip("\n") for l in open(os.path.join(GOLDEN, "myers_vectors.txt"))]
    A, B, mode, maxd, exp = [], [], [], [], []
    for inp, out in zip(lines[0::2], lines[1::2]):
        m, d, a, b = inp.split(" ")
        if len(a) > 320:
            continue
        A.append(a); B.append(b); mode.append(int(m)); maxd.append(int(d)); exp.append(int(out.split(" ")[0]))
    assert len(exp) >= 150
    hip = mia_amd.MiaHip(0)
    got = hip.myers_packed(mia_amd.pack_myers_pairs(A, B), mode, maxd)
    bad = [(i, int(got[i]), exp[i], mode[i], maxd[i], len(A[i]), len(B[i])) for i in range(len(exp)) if int(got[i]) != exp[i]]
    assert not bad, bad[:10]
    with pytest.raises(mia_amd.MiaHipError):
        hip.myers_packed(mia_amd.pack_myers_pairs(["A" * 400], ["A" * 400]), [0], [10])
    hip.close()


def _edited_copy(rng, a, subs, dels, ins):
    alpha = np.frombuffer(b"ACGT", np.uint8)
    b = list(a)
    for _ in range(subs):
        b[int(rng.integers(0, len(b)))] = alpha[rng.integers(0, 4)]
    for _ in range(dels):
        p = int(rng.integers(0, len(b) - 4))
        del b[p:p + int(rng.integers(1, 4))]
    for _ in range(ins):
        p = int(rng.integers(0, len(b)))
        b[p:p] = list(alpha[rng.integers(0, 4, int(rng.integers(1, 4)))])
    return np.array(b, np.uint8)


def test_dpath_kernel_equals_bitvector_kernel_on_long_pairs():
    """k_myers_ond (round 5: furthest-reaching D-paths, one row of diagonals per step, for the pairs too long for a lane)
    against k_myers (MIA_HIP_MYERS_NO_OND=1: every long pair through the bit-vector sweep the reference's vectors pin):
    400 pairs of 330 .. 6 000 characters -- near-identical (the D-path kernel finishes them), a tenth apart (its cap decides),
    unrelated (it hands them on: 0xFFFFFFFE never reaches the caller), overhangs for the prefix modes, IUPAC codes and
    characters no bitmap knows, maxd from 1 to beyond the sum of the lengths."""
    import mia_amd
    rng = np.random.default_rng(2024)
    alpha = np.frombuffer(b"ACGTACGTACGTACGTNRYKMSWBDHVacgtX", np.uint8)
    A, B, mode, maxd = [], [], [], []
    for i in range(400):
        la = int(rng.integers(330, 6000 if i % 8 == 0 else 1500))
        a = alpha[rng.integers(0, len(alpha) if i % 2 else 16, la)].copy()
        kind = i % 5
        if kind == 0:
            b = alpha[rng.integers(0, 16, int(rng.integers(330, 1500)))].copy()
        elif kind == 1:
            b = _edited_copy(rng, a, la // 10, la // 40, la // 40)
        else:
            b = _edited_copy(rng, a, int(rng.integers(0, 12)), int(rng.integers(0, 4)), int(rng.integers(0, 4)))
        m = int(rng.integers(0, 3))
        if kind == 2 and m == 1:
            a = np.concatenate([a, alpha[rng.integers(0, 16, 40)]])
        if kind == 2 and m == 2:
            b = np.concatenate([b, alpha[rng.integers(0, 16, 40)]])
        A.append(a.tobytes().decode("latin1")); B.append(b.tobytes().decode("latin1"))
        mode.append(m)
        maxd.append(int(rng.choice([1, 3, 20, 150, 700, 100000])))
    hip = mia_amd.MiaHip(0)
    got = hip.myers(A, B, mode, maxd)
    hip.close()
    os.environ["MIA_HIP_MYERS_NO_OND"] = "1"
    try:
        ref = mia_amd.MiaHip(0)
    finally:
        os.environ.pop("MIA_HIP_MYERS_NO_OND")
    want = ref.myers(A, B, mode, maxd)
    ref.close()
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, [(int(i), int(got[i]), int(want[i]), mode[i], maxd[i], len(A[i]), len(B[i])) for i in bad[:8]]
    assert (want < 0xFFFFFFFE).sum() > 150 and (want == 0xFFFFFFFF).sum() > 50 and not (got == 0xFFFFFFFE).any()


def test_ccheck_sized_pair_rows_from_the_device_table():
    """The one call ccheck makes (src/ccheck.cc:477-480): 16.6 kb against 16.6 kb, maxd = len/10.  mia_hip_myers_align now
    walks back through the table k_myers_ond wrote; the rows must spell both sequences and cost exactly the distance, and the
    distance must be what the bit-vector kernel says."""
    import mia_amd
    rng = np.random.default_rng(16)
    a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 16_600)].copy()
    a[rng.integers(0, len(a), 1600)] = ord("N")                  # mt311 carries an ambiguity code in every tenth column
    b = _edited_copy(rng, np.where(a == ord("N"), ord("A"), a).astype(np.uint8), 90, 5, 5)
    sa, sb = a.tobytes().decode(), b.tobytes().decode()
    hip = mia_amd.MiaHip(0)
    for mode in (0, 1, 2):
        d, ra, rb = hip.myers_align(sa, mode, sb, len(sa) // 10)
        assert d is not None and 20 < d < 140
        os.environ["MIA_HIP_MYERS_NO_OND"] = "1"
        try:
            ref = mia_amd.MiaHip(0)
        finally:
            os.environ.pop("MIA_HIP_MYERS_NO_OND")
        d2, ra2, rb2 = ref.myers_align(sa, mode, sb, len(sa) // 10)      # bit-vector distance, D-paths on the host
        ref.close()
        assert (d, ra, rb) == (d2, ra2, rb2)
        if len(ra) == len(rb):
            assert sb.startswith(rb.replace("-", "")) and sa.startswith(ra.replace("-", ""))
            cost = sum(1 for x, y in zip(ra, rb) if x == "-" or y == "-" or not (_bits(x) & _bits(y)))
            assert cost == d
    hip.close()


def test_rows_of_a_pair_beyond_the_dpath_kernels_cap():
    """a long pair whose distance lies BETWEEN k_myers_ond's cap (about sqrt(64 * sweep): ~260 for 1 000 characters,
    ~360 for 2 000) and maxd is finished by the bit-vector kernel; the device table then holds no row for it (it was sized for
    the cap) and the rows must come from the host's D-path walk -- the same rows, the same distance, as with the D-path kernel
    switched off altogether.  A pair below the cap rides along (rows from the device table), and one with maxd = 0 (no distance
    is admissible: the reference returns UINT_MAX without looking at the sequences)."""
    import mia_amd
    rng = np.random.default_rng(77)
    hip = mia_amd.MiaHip(0)
    os.environ["MIA_HIP_MYERS_NO_OND"] = "1"
    try:
        ref = mia_amd.MiaHip(0)
    finally:
        os.environ.pop("MIA_HIP_MYERS_NO_OND")
    seen_beyond = 0
    for la, subs, dels, ins in ((1000, 260, 15, 15), (2000, 380, 25, 25), (1500, 330, 10, 10), (1200, 20, 2, 2)):
        a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, la)].copy()
        b = _edited_copy(rng, a, subs, dels, ins)
        sa, sb = a.tobytes().decode(), b.tobytes().decode()
        for mode in (0, 1, 2):
            maxd = la          # far above the cap
            d, ra, rb = hip.myers_align(sa, mode, sb, maxd)
            d2, ra2, rb2 = ref.myers_align(sa, mode, sb, maxd)
            assert d is not None and (d, ra, rb) == (d2, ra2, rb2), (la, mode, d, d2)
            if d > 300:
                seen_beyond += 1
            if len(ra) == len(rb):
                cost = sum(1 for x, y in zip(ra, rb) if x == "-" or y == "-" or not (_bits(x) & _bits(y)))
                assert cost == d
    assert seen_beyond >= 3
    # maxd = 0 beside a D-path pair in one call: the pair without a cap must not be read as if it had been packed
    a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 900)].copy()
    b = _edited_copy(rng, a, 5, 1, 1)
    sa, sb = a.tobytes().decode(), b.tobytes().decode()
    got = hip.myers([sa, sa, sa], [sb, sb, sb], [0, 0, 0], [0, 50, -3])
    want = ref.myers([sa, sa, sa], [sb, sb, sb], [0, 0, 0], [0, 50, -3])
    assert got[0] == 0xFFFFFFFF and got[2] == 0xFFFFFFFF and got[1] < 50 and (got == want).all()
    hip.close()
    ref.close()
