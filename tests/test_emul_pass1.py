"""Pass-1 kernel body (csrc/pass1_body.h) on the CPU lock-step emulation vs the oracle:
both-strand whole-reference DP with k-mer style column masks, strand choice, traceback.
CPU only."""
import ctypes as C
import random

import numpy as np
import pytest

import oracle_ctypes as oc
from test_emul_align import codes, emul  # noqa: F401  (fixture)
from test_oracle_vs_golden import _pssm

RUNS = {}
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP.get(ch, "N") for ch in reversed(s))


def run_emul(emul, fw, rc, read, pssm, fmask, rmask, max_abs=1100, cpl=4):
    out = (C.c_int32 * 8)()
    pm = np.ctypeslib.as_array(pssm.sm).reshape(-1).astype(np.int32)
    cf, cr, c2 = codes(fw), codes(rc), codes(read)
    fm = None if fmask is None else np.frombuffer(bytes(fmask), dtype=np.uint8).copy()
    rm = None if rmask is None else np.frombuffer(bytes(rmask), dtype=np.uint8).copy()
    emul.emu_pass1.restype = C.c_int
    rcode = emul.emu_pass1(cpl, cf.ctypes.data_as(C.c_void_p), cr.ctypes.data_as(C.c_void_p), len(fw), c2.ctypes.data_as(C.c_void_p),
                           len(read), pm.ctypes.data_as(C.c_void_p), max_abs,
                           None if fm is None else fm.ctypes.data_as(C.c_void_p),
                           None if rm is None else rm.ctypes.data_as(C.c_void_p), out)
    assert rcode == 0
    return list(out)


def oracle_pass1(oracle, fw, rc, read, pssm, fmask, rmask):
    res = []
    for s, m in ((fw, fmask), (rc, rmask)):
        a = oc.Aln()
        oracle.ora_align(s.encode(), len(s), read.encode(), len(read), None if m is None else bytes(m), C.byref(pssm), 1,
                         C.byref(a), None, None, None, None)
        res.append(a)
    st = 0 if res[0].best > res[1].best else 1
    return res, st


def check(emul, oracle, fw, read, pssm, fmask=None, rmask=None, cpls=(4, 12, 112)):
    rc = revcomp(fw)
    exp, st = oracle_pass1(oracle, fw, rc, read, pssm, fmask, rmask)
    pm = np.ctypeslib.as_array(pssm.sm).reshape(-1)
    n = len(read)
    for cpl in cpls:
        # the wide chunks keep a 256-column candidate horizon: exact only under the host's range check (mia_hip_pass1)
        if cpl == 112 and fmask is not None:
            continue
        if cpl in (12, 112) and n * int(pm.max()) + 1000 + 200 * (n + 1) >= 1000 + 200 * 255:
            continue
        RUNS[cpl] = RUNS.get(cpl, 0) + 1
        _check_one(run_emul(emul, fw, rc, read, pssm, fmask, rmask, cpl=cpl), exp, st)


def _check_one(got, exp, st):
    assert got[0] == exp[0].best and got[1] == exp[1].best, (got, exp[0].best, exp[1].best)
    assert got[2] == st
    e = exp[st]
    assert (got[3], got[4], got[5], got[6]) == (e.best, e.aec, e.abc, e.abr), (got, e.best, e.aec, e.abc, e.abr)
    assert got[7] == 0


def mutate(rnd, s, nsub=2, indel=True):
    s = list(s)
    for _ in range(nsub):
        s[rnd.randrange(len(s))] = rnd.choice("ACGTN")
    if indel and len(s) > 12:
        p = rnd.randrange(3, len(s) - 3)
        if rnd.random() < 0.5:
            del s[p:p + rnd.randint(1, 3)]
        else:
            s[p:p] = [rnd.choice("ACGT") for _ in range(rnd.randint(1, 3))]
    return "".join(s)[:256]


def test_pass1_unmasked(emul, oracle):
    rnd = random.Random(17)
    flat = _pssm(oracle, "flat", 0)
    anc = _pssm(oracle, "ancient.submat.txt", 0)
    for i in range(60):
        L = rnd.choice([90, 255, 256, 257, 300, 511, 600, 767, 768, 1000, 1300, 1700, 2500, 4000])
        ref = "".join(rnd.choice("ACGT") for _ in range(L))
        wrap = ref + ref[: min(L, 256)]
        n = rnd.choice([20, 45, 100, 100, 150, 256])
        p = rnd.randrange(0, L)
        frag = (ref + ref)[p:p + n]
        if i % 2:
            frag = revcomp(frag)
        read = mutate(rnd, frag, rnd.randint(0, 3), i % 3 == 0)
        if i % 7 == 0:
            wrap = wrap[:50] + "N" + wrap[51:]
        check(emul, oracle, wrap, read, [flat, anc][i % 2])
    assert RUNS.get(4, 0) >= 40 and RUNS.get(12, 0) >= 15 and RUNS.get(112, 0) >= 15, RUNS


def test_pass1_chunk_boundaries(emul, oracle):
    """reads (with indels) straddling the 256- and 768-column chunk boundaries, both strands: the retrace has to
    cross into the chunk on the left, and the plain sweep's placeholder carries must never surface"""
    rnd = random.Random(99)
    flat = _pssm(oracle, "flat", 0)
    anc = _pssm(oracle, "ancient.submat.txt", 0)
    before = dict(RUNS)
    L = 3300
    for i in range(36):
        ref = "".join(rnd.choice("ACGT") for _ in range(L))
        wrap = ref + ref[:256]
        n = rnd.choice([40, 80, 100, 120])
        edge = rnd.choice([256, 512, 768, 1536, 2304, 3072])
        p = max(0, edge - rnd.randint(1, n - 1))
        frag = ref[p:p + n]
        if i % 2:                       # reverse strand: the alignment sits at L - p - n on the reverse complement
            p2 = max(0, L - edge - rnd.randint(1, n - 1))
            frag = revcomp(ref[p2:p2 + n])
        read = mutate(rnd, frag, rnd.randint(0, 2), i % 3 != 2)
        check(emul, oracle, wrap, read, [flat, anc][i % 2])
    assert RUNS.get(112, 0) - before.get(112, 0) >= 20, RUNS


def test_pass1_masked(emul, oracle):
    """k-mer style masks: a few open intervals per strand, possibly several chunks apart,
    one strand sometimes fully masked"""
    rnd = random.Random(23)
    flat = _pssm(oracle, "flat", 0)
    anc = _pssm(oracle, "ancient.submat.txt", 0)
    for i in range(60):
        L = rnd.choice([300, 700, 1100, 1500, 2400])
        ref = "".join(rnd.choice("ACGT") for _ in range(L))
        wrap = ref + ref[:256]
        n = rnd.choice([30, 60, 100, 140])
        p = rnd.randrange(0, L)
        frag = (ref + ref)[p:p + n]
        strand = i % 2
        read = mutate(rnd, revcomp(frag) if strand else frag, rnd.randint(0, 2), i % 4 == 0)
        W = len(wrap)

        def mk(hit_at, extra):
            m = bytearray(W)
            spans = []
            if hit_at is not None:
                spans.append((max(0, hit_at - 10 - rnd.randint(0, 5)), min(W - 1, hit_at + n + 10)))
            for _ in range(extra):
                a = rnd.randrange(0, W)
                spans.append((a, min(W - 1, a + rnd.randint(1, n + 20))))
            for a, b in spans:
                for c in range(a, b + 1):
                    m[c] = 1
            return m
        rc_pos = (L - (p + n)) % L
        fm = mk(p if strand == 0 else None, rnd.randint(0, 2))
        rm = mk(rc_pos if strand == 1 else None, rnd.randint(0, 2))
        if sum(fm) == 0 and sum(rm) == 0:
            continue
        if i % 9 == 0:
            fm = bytearray([1]) * W      # saturated (>=128 hits): everything open on one strand
        check(emul, oracle, wrap, read, [flat, anc][i % 2], fm, rm)
    assert RUNS.get(12, 0) >= 30, RUNS
