"""The banded DP (csrc/band_body.h -- the source hipcc compiles into k_band_align) on the CPU against the oracle's
full-window DP.  Whenever band_plan accepts a read and band_align finishes it, score, end points, begin row and the
whole gapped alignment must be what dyn_prog / max_sg_score / find_align_begin / populate_pwaln_to_begin give over the
whole window.  The cases aim at what a band can get wrong: indels of every length that fits, two indels, indels next to
the read ends (the reference's index-0 quirk), tandem repeats where equal-scoring gap placements compete (ties decide
the script), late starts, windows clipped at the reference start (column 0 rule)."""
import ctypes as C
import random

import numpy as np
import pytest

import oracle_ctypes as oc
from test_emul_align import codes, emul, script_to_strings  # noqa: F401  (fixture)
from test_emul_diag_filter import mutate
from test_oracle_vs_golden import _pssm


@pytest.fixture(scope="module")
def flat(oracle):
    return _pssm(oracle, "flat", 0)


def run_band(emul, ref, s, len1, read, widen=0):
    rc, c2 = codes(ref), codes(read)
    out5 = (C.c_int32 * 5)()
    plan = (C.c_int32 * 4)()
    cols = np.full(len(read) + 8, -9, dtype=np.int16)
    emul.emu_band.restype = C.c_int
    done = emul.emu_band(rc.ctypes.data_as(C.c_void_p), C.c_int64(len(ref)), s, len1, c2.ctypes.data_as(C.c_void_p), len(read), out5,
                         cols.ctypes.data_as(C.c_void_p), plan, widen)
    return done, list(out5), cols, list(plan)


def check(emul, oracle, flat, ref, s, len1, read, stats):
    done, out5, cols, plan = run_band(emul, ref, s, len1, read, stats[0] % 7)     # a wavefront's band is its widest read's
    stats[0] += 1
    if not done:
        return False
    stats[1] += 1
    win = ref[s:s + len1]
    res = oc.Aln()
    rg = C.create_string_buffer(1100)
    fg = C.create_string_buffer(1100)
    assert oracle.ora_align(win.encode(), len(win), read.encode(), len(read), None, C.byref(flat), 1, C.byref(res), rg, fg, None, None) == 0
    assert (out5[0], out5[1], out5[2], out5[3]) == (res.best, res.abc, res.aec, res.abr), (win, read, plan, out5, (res.best, res.abc, res.aec, res.abr))
    r, f = script_to_strings(win, read, cols, res.abr, res.aer)
    assert r == rg.value.decode() and f == fg.value.decode(), (win, read, plan)
    assert all(int(cols[i]) == -2 for i in range(res.abr))
    assert out5[4] == len([x for x in rg.value.split(b"-") if x]) - 1 + len([x for x in fg.value.split(b"-") if x]) - 1, (win, read)
    return True


def window(ref, pos, length, margin=50):
    s = max(0, pos - margin)
    e = min(len(ref), pos + length + margin)
    return s, e - s


def test_single_indels_everywhere(emul, oracle, flat):
    rnd = random.Random(7)
    ref = "".join(rnd.choice("ACGT") for _ in range(4000))
    stats = [0, 0]
    for i in range(600):
        n = rnd.randint(30, 180)
        pos = rnd.randint(0, len(ref) - n - 40)
        src = ref[pos:pos + n + 30]
        at = rnd.choice([1, 2, 3, 5, 9, 10, 11, n // 2, n - 12, n - 10, n - 3, n - 2, rnd.randint(1, n - 2)])
        k = rnd.choice([1, 1, 1, 2, 3, 5, 8])
        if i % 2:
            read = src[:at] + src[at + k:][:n - at]                     # deletion from the reference's point of view
        else:
            ins = "".join(rnd.choice("ACGT") for _ in range(k))
            read = (src[:at] + ins + src[at:])[:n]
        read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([0, 0, 1, 2, 3])))
        s, l1 = window(ref, pos, len(read))
        check(emul, oracle, flat, ref, s, l1, read, stats)
    assert stats[1] > stats[0] // 2, stats


def test_two_indels_and_heavy_damage(emul, oracle, flat):
    rnd = random.Random(8)
    ref = "".join(rnd.choice("ACGT") for _ in range(4000))
    stats = [0, 0]
    for i in range(500):
        n = rnd.randint(70, 200)
        pos = rnd.randint(0, len(ref) - n - 60)
        read = ref[pos:pos + n + 40]
        for _ in range(2):
            at = rnd.randint(1, len(read) - 30)
            k = rnd.randint(1, 4)
            if rnd.random() < 0.5:
                read = read[:at] + read[at + k:]
            else:
                read = read[:at] + "".join(rnd.choice("ACGT") for _ in range(k)) + read[at:]
        read = read[:n]
        read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([0, 1, 3, 5, 8])))
        s, l1 = window(ref, pos, len(read))
        check(emul, oracle, flat, ref, s, l1, read, stats)
    assert stats[1] > 50, stats


def test_repeats_where_gap_placements_tie(emul, oracle, flat):
    rnd = random.Random(9)
    stats = [0, 0]
    for i in range(400):
        unit = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([1, 2, 3, 4, 7])))
        reps = rnd.randint(3, 12)
        left = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(150, 300)))
        right = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(150, 300)))
        ref = left + unit * reps + right
        a = rnd.randint(40, 110)
        b = rnd.randint(40, 110)
        delta = rnd.choice([-2, -1, 1, 2])                       # the read has more / fewer copies of the unit
        read = left[-a:] + unit * max(0, reps + delta) + right[:b]
        if len(read) > 250:
            continue
        read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([0, 0, 1, 2])))
        pos = len(left) - a
        s, l1 = window(ref, pos, a + len(unit) * reps + b)
        check(emul, oracle, flat, ref, s, l1, read, stats)
    assert stats[1] > 100, stats


def test_clipped_windows_and_late_starts(emul, oracle, flat):
    rnd = random.Random(10)
    ref = "".join(rnd.choice("ACGT") for _ in range(1500))
    stats = [0, 0]
    for i in range(400):
        n = rnd.randint(32, 150)
        pos = rnd.choice([0, 0, 1, 2, 5, len(ref) - n, len(ref) - n - 1, rnd.randint(0, 30)])
        read = ref[pos:pos + n]
        junk = rnd.choice([0, 0, 3, 6, 12])                       # a junk head: the alignment starts late
        read = "".join(rnd.choice("ACGT") for _ in range(junk)) + read[junk:]
        if i % 3 == 0:
            at = rnd.randint(8, n - 8)
            read = read[:at] + read[at + 1:]
        read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([0, 1, 3])))
        s, l1 = window(ref, pos, len(read), margin=rnd.choice([50, 50, 10, 0]))
        if l1 < len(read):
            continue
        check(emul, oracle, flat, ref, s, l1, read, stats)
    assert stats[1] > 100, stats


def test_band_never_wider_than_registers(emul):
    rnd = random.Random(11)
    ref = "".join(rnd.choice("ACGT") for _ in range(2000))
    for i in range(200):
        n = rnd.randint(60, 200)
        pos = rnd.randint(0, len(ref) - n - 40)
        at = rnd.randint(10, n - 10)
        k = rnd.randint(1, 30)
        read = (ref[pos:pos + at] + ref[pos + at + k:])[:n]
        s, l1 = window(ref, pos, n + k)
        done, out5, cols, plan = run_band(emul, ref, s, l1, read)
        if done:
            assert 1 <= plan[1] <= 32 and plan[2] <= plan[3]
