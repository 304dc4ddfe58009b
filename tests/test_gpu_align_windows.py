"""mia_hip_align_windows -- ccheck's per-read re-alignment (reference src/ccheck.cc:569-604): every read of the store
against its own reference string, no margin, no whole-reference fallback.

* the reference's own DP answers (tests/golden/dp_vectors.txt) replayed as ONE batch per matrix, the vectors whose
  window is shorter than the read included (mia_hip_realign cannot express those);
* a ccheck-shaped batch (windows one column longer than the read's span, substitutions, indels, N columns,
  reads up to 256 bases) against the oracle's dyn_prog, read by read: score, end points and the two gapped strings."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import oracle_ctypes as oc
from conftest import GOLDEN
from test_emul_align import script_to_strings
from test_oracle_vs_golden import _pssm, dp_cases

pytestmark = pytest.mark.gpu


def run_batch(mod, pssm_words, pairs):
    """pairs: (window, read) strings -> per read (score, abc, aec, cols relative to its window)"""
    h = mod.MiaHip(0)
    h.set_pssm(pssm_words, pssm_words)
    n = len(pairs)
    reads = "".join(r for _, r in pairs)
    roff = np.concatenate([[0], np.cumsum([len(r) for _, r in pairs])]).astype(np.int64)
    woff = np.concatenate([[0], np.cumsum([len(w) for w, _ in pairs])]).astype(np.int64)
    h.upload_reads(np.frombuffer(reads.encode(), dtype=np.uint8), roff, np.zeros(n, np.uint8), np.ones(n, np.uint8),
                   np.zeros(n, np.int32), np.zeros(n, np.int32))
    h.align_windows("".join(w for w, _ in pairs), woff[:-1], np.diff(woff))
    score, as_, ae = h.alignments()
    cols, rstart = h.scripts()
    out = []
    for i in range(n):
        c = cols[i].astype(np.int64)
        c = np.where(c >= 0, c + int(rstart[i]) - int(woff[i]), c)
        out.append((int(score[i]), int(as_[i] - woff[i]), int(ae[i] - woff[i]), c))
    with pytest.raises(RuntimeError):
        h.cull(0, 100.0, 0.0, 0)          # the consensus path refuses to run on caller-supplied windows
    h.close()
    return out


def test_dp_vectors_as_one_batch():
    import mia_amd
    flat = mia_amd.flat_pssm()
    anc = mia_amd.read_pssm(os.path.join(GOLDEN, "ancient.submat.txt"))
    groups = {}
    for inp, exp in dp_cases():
        _, spec, rc, sg5, _sg3, s1, s2, mask = inp.split(" ")
        if mask != "*" or sg5 != "1" or len(s2) > 256:
            continue
        groups.setdefault((spec, rc), []).append((s1, s2, exp.split(" ")))
    n_run = n_short = 0
    for (spec, rc), vecs in groups.items():
        p = flat if spec == "flat" else anc
        pm = mia_amd.revcom_pssm(p) if rc == "1" else p
        res = run_batch(mia_amd, pm, [(s1, s2) for s1, s2, _ in vecs])
        for (s1, s2, e), (score, abc, aec, cols) in zip(vecs, res):
            assert (score, aec, abc) == (int(e[1]), int(e[2]), int(e[4])), (spec, rc, s1[:60], s2[:60])
            if e[8]:
                r, f = script_to_strings(s1, s2, cols, int(e[5]), int(e[3]))
                assert (r, f) == (e[8], e[9]), (spec, rc, s1[:60], s2[:60])
            n_short += len(s1) < len(s2)
            n_run += 1
    assert n_run > 120 and n_short >= 1


def ccheck_shaped_pairs(n, seed):
    rnd = random.Random(seed)
    genome = "".join(rnd.choice("ACGT") for _ in range(20000))
    pairs = []
    for i in range(n):
        len2 = rnd.choice([25, 36, 50, 60, 76, 100, 100, 100, 150, 200, 256]) if i % 50 else rnd.randint(1, 256)
        st = rnd.randrange(0, len(genome) - 300)
        read = list(genome[st:st + len2])
        win = list(genome[st:st + len2 + 1])              # lift_over(.., start, end + 2)
        k = rnd.random()
        if k < 0.5:
            for _ in range(rnd.randint(0, 3)):
                read[rnd.randrange(len(read))] = rnd.choice("ACGTN")
        if 0.3 < k < 0.45 and len2 > 12:
            p = rnd.randrange(3, len2 - 3); del read[p:p + rnd.randint(1, 3)]
        if 0.45 < k < 0.6 and len2 > 12:
            p = rnd.randrange(3, len2 - 3); read[p:p] = [rnd.choice("ACGT") for _ in range(rnd.randint(1, 4))]
        if 0.6 < k < 0.7:
            for _ in range(rnd.randint(1, 4)):
                win[rnd.randrange(len(win))] = "N"        # ambiguity codes of the contaminant become N
        if 0.7 < k < 0.75 and len(win) > 8:
            del win[:rnd.randint(1, 5)]                    # window shorter than the read
        if 0.75 < k < 0.8:
            win[0:0] = [rnd.choice("ACGT") for _ in range(rnd.randint(1, 30))]
        if 0.8 < k < 0.83:
            p = rnd.randrange(len(win)); win[p:p] = [rnd.choice("ACGT") for _ in range(70)]   # a gap the byte trace cannot hold
        pairs.append(("".join(win), "".join(read)[:256]))
    return pairs


@pytest.mark.parametrize("spec", ["flat", "ancient.submat.txt"])
def test_ccheck_shaped_batch(oracle, spec):
    import mia_amd
    pm = mia_amd.flat_pssm() if spec == "flat" else mia_amd.read_pssm(os.path.join(GOLDEN, spec))
    opm = _pssm(oracle, spec, 0)
    pairs = ccheck_shaped_pairs(20000, 3 if spec == "flat" else 4)
    res = run_batch(mia_amd, pm, pairs)
    a = oc.Aln()
    rg = C.create_string_buffer(1100)
    fg = C.create_string_buffer(1100)
    n_gap = n_clip = 0
    for (w, r), (score, abc, aec, cols) in zip(pairs, res):
        assert oracle.ora_align(w.encode(), len(w), r.encode(), len(r), None, C.byref(opm), 1, C.byref(a), rg, fg, None, None) == 0
        assert (score, abc, aec) == (a.best, a.abc, a.aec), (w, r)
        if not rg.value:
            continue                                    # more than 512 alignment columns: undefined in the reference
        gr, gf = script_to_strings(w, r, cols, a.abr, a.aer)
        assert (gr, gf) == (rg.value.decode(), fg.value.decode()), (w, r)
        assert all(int(cols[q]) == -2 for q in range(a.abr))
        n_gap += "-" in gr or "-" in gf
        n_clip += a.abr > 0
    assert n_gap > 1000 and n_clip > 100
