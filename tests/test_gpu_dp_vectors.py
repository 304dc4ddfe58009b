"""The reference's own DP answers (tests/golden/dp_vectors.txt, dumped from the real
dyn_prog/max_sg_score/find_align_begin/populate_pwaln_to_begin) replayed through the C ABI:
each vector becomes a one-read realign call whose window is the whole (linear) reference.
tests/golden/dp_vectors_wide.txt adds what the first file is thin on: windows of 257-512, 513-768 and more than 768
columns (the three one-read-per-wavefront classes, the exact scalar kernel) and gaps of 63 and more on the best path
(byte-trace escape -> exact kernel); the counts at the end make sure each of them was really exercised."""
import numpy as np
import pytest

from test_oracle_vs_golden import dp_cases
from test_emul_align import script_to_strings

pytestmark = pytest.mark.gpu


def test_dp_vectors_through_abi():
    import mia_amd
    flat = mia_amd.flat_pssm()
    anc = mia_amd.read_pssm(__import__("os").path.join(__import__("conftest").GOLDEN, "ancient.submat.txt"))
    n_run = n_long = 0
    n_class = [0, 0, 0, 0]
    hips = {}
    for inp, exp in dp_cases():
        _, spec, rc, sg5, _sg3, s1, s2, mask = inp.split(" ")
        if mask != "*" or sg5 != "1" or len(s1) < len(s2):
            continue          # reiterate_assembly always aligns with an all-ones mask and sg5 = 1
        e = exp.split(" ")
        key = (spec, rc)
        if key not in hips:
            h = mia_amd.MiaHip(0)
            p = flat if spec == "flat" else anc
            fwd = mia_amd.revcom_pssm(p) if rc == "1" else p
            h.set_pssm(fwd, fwd)          # the vector's matrix is used whatever the read's strand flag
            hips[key] = h
        h = hips[key]
        # window of reiterate_assembly: [max(0, as-50), min(wrap, ae+50)) -> as = 0, ae = len(s1) gives [0, len(s1))
        seq = np.frombuffer(s2.encode(), dtype=np.uint8)
        h.upload_reads(seq, np.array([0, len(s2)], np.int64), [0], [1], [0], [len(s1)])
        h.realign(s1, False)
        score, as_, ae = h.alignments()
        cols, rstart = h.scripts()
        assert (int(score[0]), int(ae[0]), int(as_[0])) == (int(e[1]), int(e[2]), int(e[4])), inp[:90]
        if e[8]:
            abr, aer = int(e[5]), int(e[3])
            c = cols[0].astype(np.int64)
            c = np.where(c >= 0, c + int(rstart[0]), c)
            r, f = script_to_strings(s1, s2, c, abr, aer)
            assert (r, f) == (e[8], e[9]), inp[:90]
            n_long += ("-" * 63 in e[8]) or ("-" * 63 in e[9])
        n_class[0 if len(s1) <= 256 else 1 if len(s1) <= 512 else 2 if len(s1) <= 768 else 3] += 1
        n_run += 1
    assert n_run > 150 and n_long >= 6 and min(n_class) >= 8, (n_run, n_long, n_class)
    for h in hips.values():
        h.close()
