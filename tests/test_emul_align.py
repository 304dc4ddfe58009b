"""The windowed-DP kernel body (csrc/align_body.h -- the same source hipcc
compiles for gfx950) run on the CPU lock-step wave emulation (tests/emul) and
compared with the oracle: score, end points and the full alignment (rebuilt
from the per-row column script).  CPU only; catches logic errors in the
packed-word DP / ballot traceback before any GPU time is spent."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
import oracle_ctypes as oc
from test_oracle_vs_golden import dp_cases, _pssm

CODE = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}


@pytest.fixture(scope="session")
def emul(oracle_build):
    out = os.path.join(oracle_build, "libmia_emul.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", "mapping-iterative-assembler_amd/csrc", "-I",
                    "tests/emul", "-o", out, "tests/emul/emu_align.cpp"], cwd=ROOT, check=True)
    lib = C.CDLL(out)
    lib.emu_align_window.restype = C.c_int
    return lib


def codes(s):
    return np.array([CODE.get(ch, 4) for ch in s.encode()], dtype=np.uint8)


def script_to_strings(s1, s2, cols, abr, aer):
    """Rebuild populate_pwaln_to_begin's two gapped strings from the column script."""
    ref, frag = [], []
    prev = None
    for r in range(abr, aer + 1):
        c = int(cols[r])
        if c == -1:
            ref.append("-"); frag.append(s2[r])
            continue
        assert c >= 0
        if prev is not None:
            for k in range(prev + 1, c):
                ref.append(s1[k]); frag.append("-")
        ref.append(s1[c]); frag.append(s2[r])
        prev = c
    return "".join(ref), "".join(frag)


def run_emul(emul, cpl, s1, s2, pssm, sg5, max_abs):
    c1, c2 = codes(s1), codes(s2)
    out5 = (C.c_int32 * 5)()
    cols = np.full(len(s2) + 8, -9, dtype=np.int16)
    pm = np.ctypeslib.as_array(pssm.sm).reshape(-1).astype(np.int32)
    rc = emul.emu_align_window(cpl, c1.ctypes.data_as(C.c_void_p), 0, len(s1), c2.ctypes.data_as(C.c_void_p), len(s2),
                               pm.ctypes.data_as(C.c_void_p), sg5, max_abs, out5, cols.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return list(out5), cols


def check(emul, oracle, s1, s2, pssm, sg5, max_abs=1100, cpl=None):
    res = oc.Aln()
    rg = C.create_string_buffer(520)
    fg = C.create_string_buffer(520)
    oracle.ora_align(s1.encode(), len(s1), s2.encode(), len(s2), None, C.byref(pssm), sg5, C.byref(res), rg, fg, None, None)
    cpls = [cpl] if cpl else [c for c in (4, 8, 12) if len(s1) <= 64 * c]
    for c in cpls:
        out5, cols = run_emul(emul, c, s1, s2, pssm, sg5, max_abs)
        if out5[4] & 1:       # saturated gap on the path: legitimately deferred to the wide kernel
            assert max(len(x) for x in (rg.value.split(b"-") + fg.value.split(b"-"))) >= 0
            continue
        assert out5[0] == res.best and out5[3] == res.aec, (c, s1[:40], s2[:40])
        assert out5[1] == res.abc and out5[2] == res.abr, (c, s1[:40], s2[:40])
        if len(rg.value) == 0:     # > 512 columns: undefined in the reference
            assert out5[4] & 2
            continue
        r, f = script_to_strings(s1, s2, cols, res.abr, res.aer)
        assert r == rg.value.decode() and f == fg.value.decode(), (c, s1[:40], s2[:40])
        assert all(int(cols[i]) == -2 for i in range(res.abr))
    return res


def test_emul_golden_vectors(emul, oracle):
    n = 0
    for inp, _ in dp_cases():
        _, spec, rc, sg5, _sg3, s1, s2, mask = inp.split(" ")
        if mask != "*" or len(s1) > 768:
            continue
        check(emul, oracle, s1, s2, _pssm(oracle, spec, int(rc)), int(sg5))
        n += 1
    assert n > 100


def test_emul_random_windows(emul, oracle):
    rnd = random.Random(5)
    anc = _pssm(oracle, "ancient.submat.txt", 0)
    anc_rc = _pssm(oracle, "ancient.submat.txt", 1)
    flat = _pssm(oracle, "flat", 0)
    for i in range(150):
        len2 = rnd.choice([1, 2, 3, 17, 30, 31, 64, 100, 100, 100, 150, 255, 256])
        L = rnd.randint(len2, len2 + 100)
        ref = "".join(rnd.choice("ACGT") for _ in range(L + 120))
        st = rnd.randint(0, 100)
        read = list(ref[st:st + len2])
        for _ in range(rnd.randint(0, 4)):
            p = rnd.randrange(len(read)); read[p] = rnd.choice("ACGTN")
        if i % 3 == 0 and len(read) > 10:
            p = rnd.randrange(2, len(read) - 2); del read[p:p + rnd.randint(1, 3)]
        if i % 3 == 1 and len(read) > 10:
            p = rnd.randrange(2, len(read) - 2); read[p:p] = [rnd.choice("ACGT") for _ in range(rnd.randint(1, 4))]
        read = "".join(read)[:256]
        s1 = ref[max(0, st - 50): st + len2 + 50]
        if i % 7 == 0:
            s1 = s1.replace("A", "N", 2)
        if len(s1) > 768:
            continue
        check(emul, oracle, s1, read, [flat, anc, anc_rc][i % 3], 1)


def test_emul_tie_heavy(emul, oracle):
    """repeats / homopolymers: every tie-break rule is exercised"""
    rnd = random.Random(11)
    flat = _pssm(oracle, "flat", 0)
    for i in range(120):
        unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 3)))
        n1, n2 = rnd.randint(5, 200), rnd.randint(1, 90)
        s1 = (unit * 300)[:n1]
        s2 = (unit * 300)[rnd.randint(0, 2):][:n2]
        if i % 2:
            s2 = list(s2)
            for _ in range(rnd.randint(1, 3)):
                s2[rnd.randrange(len(s2))] = rnd.choice("ACGT")
            s2 = "".join(s2)
        check(emul, oracle, s1, s2, flat, 1 if i % 5 else 0)


def test_emul_quad_kernel(emul, oracle):
    """four reads per wavefront (16 lanes x 13 columns each): every read must come out exactly as
    the one-read-per-wave body / the oracle gives it, whatever its three neighbours are"""
    rnd = random.Random(41)
    flat = _pssm(oracle, "flat", 0)
    anc = _pssm(oracle, "ancient.submat.txt", 0)
    anc_rc = _pssm(oracle, "ancient.submat.txt", 1)
    emul.emu_align_quad.restype = C.c_int
    n_band_escapes = 0
    for it in range(80):
        len2 = rnd.choice([1, 2, 17, 50, 100, 100, 100, 101, 150])
        ng = rnd.choice([1, 2, 3, 4, 4, 4])
        use_anc = it % 2
        fwd = np.ctypeslib.as_array((anc if use_anc else flat).sm).reshape(-1)
        rcm = np.ctypeslib.as_array((anc_rc if use_anc else flat).sm).reshape(-1)
        pssm2 = np.concatenate([fwd, rcm]).astype(np.int32)
        ref = "".join(rnd.choice("ACGT") for _ in range(600))
        refc = codes(ref)
        starts, len1s, reads, rcs, dexps = [], [], [], [], []
        for g in range(ng):
            st = rnd.randint(0, 300)
            l1 = rnd.randint(max(len2, 2), 208) if len2 <= 150 else 208
            l1 = max(l1, min(208, len2))
            frag = list(ref[st + min(50, max(0, l1 - len2)):][:len2])
            for _ in range(rnd.randint(0, 3)):
                frag[rnd.randrange(len(frag))] = rnd.choice("ACGTN")
            if len2 > 20 and rnd.random() < 0.4:
                p = rnd.randrange(3, len2 - 3)
                frag[p:p + rnd.randint(1, 2)] = []
                frag += [rnd.choice("ACGT") for _ in range(len2 - len(frag))]
            starts.append(st); len1s.append(l1); reads.append("".join(frag)[:len2].ljust(len2, "A")); rcs.append(rnd.randint(0, 1))
            dexps.append(min(50, max(0, l1 - len2)) + (rnd.choice([0, 0, 0, 40, -35]) if it % 5 == 4 else 0))
        rd = np.concatenate([codes(r) for r in reads]).astype(np.uint8)
        out = (C.c_int32 * (5 * ng))()
        cols = np.full(4 * 256, -9, dtype=np.int16)
        band = it % 2          # odd rounds: only the trace band around the expected diagonal is stored
        rcode = emul.emu_align_quad(ng, refc.ctypes.data_as(C.c_void_p), (C.c_int32 * ng)(*starts), (C.c_int32 * ng)(*len1s),
                                    rd.ctypes.data_as(C.c_void_p), len2, pssm2.ctypes.data_as(C.c_void_p), (C.c_int32 * ng)(*rcs), 1100,
                                    out, cols.ctypes.data_as(C.c_void_p), band, (C.c_int32 * ng)(*dexps))
        assert rcode == 0
        for g in range(ng):
            if out[g * 5 + 4] & 8:      # ST_BAND: the path left the stored band -> the read is re-run with a full trace
                assert band
                n_band_escapes += 1
                continue
            s1 = ref[starts[g]:starts[g] + len1s[g]]
            pm = (anc_rc if rcs[g] else anc) if use_anc else flat
            res = oc.Aln(); rg = C.create_string_buffer(520); fg = C.create_string_buffer(520)
            oracle.ora_align(s1.encode(), len(s1), reads[g].encode(), len2, None, C.byref(pm), 1, C.byref(res), rg, fg, None, None)
            got = list(out[g * 5:(g + 1) * 5])
            if got[4] & 1:
                continue
            assert (got[0], got[1], got[2], got[3]) == (res.best, res.abc, res.abr, res.aec), (it, g, got, res.best, res.abc, res.abr, res.aec)
            r, f = script_to_strings(s1, reads[g], cols[g * 256:(g + 1) * 256], res.abr, res.aer)
            assert r == rg.value.decode() and f == fg.value.decode(), (it, g)
    assert 0 < n_band_escapes < 40


def test_emul_quad_plain_kernel(emul, oracle):
    """values-only quad body (align_body_quad_plain.h): score and end column always equal the oracle's; whenever it
    claims the diagonal proof, begin point and script equal the oracle's too, and a gap-free, clip-free oracle
    alignment whose diagonal prefix never drops to the start level must be proven"""
    rnd = random.Random(77)
    flat = _pssm(oracle, "flat", 0)
    anc = _pssm(oracle, "ancient.submat.txt", 0)
    anc_rc = _pssm(oracle, "ancient.submat.txt", 1)
    emul.emu_align_quad_plain.restype = C.c_int
    n_proven = n_gapped = n_clip = n_quirk = 0
    for it in range(140):
        len2 = rnd.choice([1, 2, 17, 50, 100, 100, 100, 101, 150, 200])
        ng = rnd.choice([1, 2, 3, 4, 4, 4])
        use_anc = it % 2
        fwd = np.ctypeslib.as_array((anc if use_anc else flat).sm).reshape(-1)
        rcm = np.ctypeslib.as_array((anc_rc if use_anc else flat).sm).reshape(-1)
        pssm2 = np.concatenate([fwd, rcm]).astype(np.int32)
        ref = "".join(rnd.choice("ACGT") for _ in range(1000))
        if it % 9 == 0:
            ref = ref[:200] + "N" + ref[201:330] + "R" + ref[331:]
        refc = codes(ref)
        starts, len1s, reads, rcs = [], [], [], []
        for g in range(ng):
            st = rnd.randint(0, 300)
            l1 = min(208, max(len2, rnd.randint(max(len2, 2), 208)))
            lead = rnd.choice([min(50, max(0, l1 - len2)), 0, rnd.randint(0, max(0, l1 - len2))])
            frag = list(ref[st + lead:][:len2])
            kind = rnd.random()
            for _ in range(rnd.randint(0, 3)):
                frag[rnd.randrange(len(frag))] = rnd.choice("ACGTN")
            if len2 > 20 and kind < 0.25:                        # indel
                p = rnd.randrange(3, len2 - 3)
                if rnd.random() < 0.5:
                    frag[p:p + rnd.randint(1, 2)] = []
                    frag += [rnd.choice("ACGT") for _ in range(len2 - len(frag))]
                else:
                    frag[p:p] = [rnd.choice("ACGT") for _ in range(rnd.randint(1, 3))]
                    frag = frag[:len2]
            elif len2 > 30 and kind < 0.4:                       # junk at the 5' end: soft clip / bad prefix
                k = rnd.randint(3, 15)
                frag[:k] = [rnd.choice("ACGT") for _ in range(k)]
            starts.append(st); len1s.append(l1); reads.append("".join(frag)[:len2].ljust(len2, "A")); rcs.append(rnd.randint(0, 1))
        rd = np.concatenate([codes(r) for r in reads]).astype(np.uint8)
        out = (C.c_int32 * (6 * ng))()
        cols = np.full(4 * 256, -9, dtype=np.int16)
        assert emul.emu_align_quad_plain(ng, refc.ctypes.data_as(C.c_void_p), (C.c_int32 * ng)(*starts), (C.c_int32 * ng)(*len1s),
                                         rd.ctypes.data_as(C.c_void_p), len2, pssm2.ctypes.data_as(C.c_void_p), (C.c_int32 * ng)(*rcs),
                                         out, cols.ctypes.data_as(C.c_void_p)) == 0
        for g in range(ng):
            s1 = ref[starts[g]:starts[g] + len1s[g]]
            pm = (anc_rc if rcs[g] else anc) if use_anc else flat
            res = oc.Aln(); rg = C.create_string_buffer(520); fg = C.create_string_buffer(520)
            oracle.ora_align(s1.encode(), len(s1), reads[g].encode(), len2, None, C.byref(pm), 1, C.byref(res), rg, fg, None, None)
            score, abc, abr, aec, proven, _ = list(out[g * 6:(g + 1) * 6])
            assert (score, aec) == (res.best, res.aec), (it, g)
            gapped = b"-" in rg.value or b"-" in fg.value
            n_gapped += gapped
            n_clip += res.abr > 0 and res.abc > 0
            if proven:
                n_proven += 1
                assert not gapped and (abc, abr) == (res.abc, res.abr), (it, g)
                c = cols[g * 256: g * 256 + len2]
                assert (c[:abr] == -2).all() and np.array_equal(c[abr:], np.arange(abr, len2) + (aec - (len2 - 1))), (it, g)
            elif not gapped and not (res.abr > 0 and res.abc > 0):
                # unproven although the oracle's path LOOKS like a pure diagonal: allowed only if the diagonal prefix sank
                # below the level of a new start somewhere (condition (ii) of the proof), or if the score is not the
                # diagonal's own sum -- a gap whose source index is 0 has T == 0 and is read back as a diagonal step
                # (src/mia.c:619), so the reported path can be gap-free while the score came through the gap
                d = res.aec - (len2 - 1)
                r0 = max(0, -d)
                sm = np.ctypeslib.as_array(pm.sm).reshape(31, 5, 5)
                c1, c2 = codes(s1), codes(reads[g])
                D = 0
                sank = False
                for r in range(r0, len2):
                    dep = r if r < 15 else (30 - (len2 - r - 1) if len2 - (r + 1) < 15 else 15)
                    D += int(sm[dep][c1[r + d]][c2[r]]) + (-(1000 + 200 * (r0 + 1)) if (r == r0 and d < 0) else 0)
                    if r < len2 - 1 and D < -(1000 + 200 * (r + 2)):
                        sank = True
                n_quirk += D != res.best
                assert sank or D != res.best, (it, g, s1, reads[g])
    assert n_proven > 150 and n_gapped > 40 and n_clip > 5, (n_proven, n_gapped, n_clip)


def test_trim_lastcol_mode(emul, oracle):
    """the LASTCOL end condition of the window aligner (trim_frag) on the reference's own trim vectors and on
    random pairs against the oracle; reads whose path holds a gap of 63 or more report ST_ESCAPE (the GPU then
    re-runs them in the exact scalar kernel)"""
    import mia_amd
    from test_oracle_vs_golden import trim_vectors
    flat = mia_amd.flat_pssm().reshape(-1).astype(np.int32)
    out = (C.c_int32 * 6)()
    n_ok = n_esc = 0
    for ad, read, exp in trim_vectors():
        cr, ca = codes(read), codes(ad)
        assert emul.emu_trim(cr.ctypes.data_as(C.c_void_p), len(read), ca.ctypes.data_as(C.c_void_p), len(ad),
                             flat.ctypes.data_as(C.c_void_p), 600, out) == 0
        score, aec, aer, abc, abr, st = list(out)
        tr, tp, a = C.c_int(), C.c_int(), oc.Aln()
        oracle.ora_trim(read.encode(), len(read), ad.encode(), C.byref(tr), C.byref(tp), C.byref(a))
        assert (score, aec, aer) == (a.best, a.aec, a.aer), (ad, read)
        if st & 1:            # ST_ESCAPE
            n_esc += 1
            continue
        assert st == 0 and (abc, abr) == (a.abc, a.abr), (ad, read, list(out), exp)
        trimmed = int(score >= 1000 or score >= (aer - abr + 1) * 200)
        assert [trimmed, abc - 1 if trimmed else -999] == exp[:2]
        n_ok += 1
    assert n_ok > 450
