"""The matrix-agnostic band pipeline (csrc/bandx_body.h -- the source hipcc compiles into k_bx_plan / k_bx_values /
k_bx_trace) on the CPU against the oracle's full-window DP, for the flat matrix and both strands of the two aDNA
matrices.  Whenever a stage finishes a read -- the plan alone (one-diagonal band), the values-only DP with the plan's
diagonal check, or the trace DP -- score, end points, begin row and the whole gapped alignment must be what dyn_prog /
max_sg_score / find_align_begin / populate_pwaln_to_begin give over the whole window, and the stages must agree with
each other wherever more than one of them can finish the read (the harness returns a negative code if not)."""
import ctypes as C
import math
import random

import numpy as np
import pytest

import oracle_ctypes as oc
from test_emul_align import codes, emul, script_to_strings  # noqa: F401  (fixture)
from test_emul_diag_filter import mutate
from test_oracle_vs_golden import _pssm

MATS = [("flat", 0), ("ancient.submat.txt", 0), ("ancient.submat.txt", 1), ("ancient.submat.solexa.pe.txt", 0),
        ("ancient.submat.solexa.pe.txt", 1)]


def pssm_pair(oracle, spec):
    f, r = _pssm(oracle, spec, 0), _pssm(oracle, spec, 1)
    return (np.ctypeslib.as_array(f.sm).astype(np.int32).reshape(-1).copy(), np.ctypeslib.as_array(r.sm).astype(np.int32).reshape(-1).copy())


def run_bandx(emul, oracle, spec, strand, ref, s, len1, read, opts=0):
    fwd, rc = pssm_pair(oracle, spec)
    rcd, c2 = codes(ref), codes(read)
    out6 = (C.c_int32 * 6)()
    plan = (C.c_int32 * 5)()
    cols = np.full(len(read) + 8, -9, dtype=np.int16)
    emul.emu_bandx.restype = C.c_int
    mode = emul.emu_bandx(rcd.ctypes.data_as(C.c_void_p), C.c_int64(len(ref)), s, len1, c2.ctypes.data_as(C.c_void_p), len(read),
                          fwd.ctypes.data_as(C.c_void_p), rc.ctypes.data_as(C.c_void_p), strand, opts, out6, cols.ctypes.data_as(C.c_void_p), plan)
    assert mode >= 0, (mode, ref[s:s + len1], read, list(plan))
    return mode, list(out6), cols, list(plan)


def check(emul, oracle, spec, strand, ref, s, len1, read, stats, opts=None):
    if opts is None:
        k = stats["n"]
        opts = (k & 3) | (((k >> 2) % 3) << 4)          # trace forced or not, edge form or not, class widened by 0..2
    stats["n"] += 1
    if "N" not in ref[s:s + len1] and "N" not in read:
        check_quick(emul, oracle, spec, strand, ref, s, len1, read, stats)
    mode, out6, cols, plan = run_bandx(emul, oracle, spec, strand, ref, s, len1, read, opts)
    stats["mode%d" % mode] = stats.get("mode%d" % mode, 0) + 1
    if mode == 0:
        stats["why%d" % plan[3]] = stats.get("why%d" % plan[3], 0) + 1
    if out6[5] == 0:
        return False
    stats["how%d" % out6[5]] = stats.get("how%d" % out6[5], 0) + 1
    win = ref[s:s + len1]
    res = oc.Aln()
    rg = C.create_string_buffer(1100)
    fg = C.create_string_buffer(1100)
    assert oracle.ora_align(win.encode(), len(win), read.encode(), len(read), None, C.byref(_pssm(oracle, spec, strand)), 1, C.byref(res), rg, fg,
                            None, None) == 0
    ctx = (spec, strand, win, read, plan, out6, (res.best, res.abc, res.aec, res.abr))
    assert (out6[0], out6[1], out6[2], out6[3]) == (res.best, res.abc, res.aec, res.abr), ctx
    r, f = script_to_strings(win, read, cols, res.abr, res.aer)
    assert r == rg.value.decode() and f == fg.value.decode(), ctx
    assert all(int(cols[i]) == -2 for i in range(res.abr)) or out6[5] != 3, ctx
    assert out6[4] == len([x for x in rg.value.split(b"-") if x]) - 1 + len([x for x in fg.value.split(b"-") if x]) - 1, ctx
    return True


def check_quick(emul, oracle, spec, strand, ref, s, len1, read, stats):
    """The QUICK plan (bx_quick, round 6: what k_bx_plan<NW, 4> runs with d = the read's diagonal of the iteration before) asked about
    EVERY diagonal of the window on which the read has at most BX_QUICK_MAX mismatches -- the right one, and any wrong one a second
    copy or a repeat offers: whenever it plans the read, the stages that plan leads to must deliver dyn_prog's alignment over the
    whole window, exactly as for the full plan."""
    win = ref[s:s + len1]
    n = len(read)
    res = None
    for d in range(0, len1 - n + 1):
        # (the one-diagonal form wants few mismatches on d; the one-indel form -- bx_quick2 -- a clean stretch at the read's head)
        if sum(1 for i in range(n) if read[i] != win[d + i]) > 8 and sum(1 for i in range(min(n, 24)) if read[i] != win[d + i]) > 1:
            continue
        stats["quick_asked"] = stats.get("quick_asked", 0) + 1
        k = stats["quick_asked"]
        opts = ((d + 1) << 16) | (k & 3) | (((k >> 2) % 3) << 4)
        mode, out6, cols, plan = run_bandx(emul, oracle, spec, strand, ref, s, len1, read, opts)
        if mode == 0 or out6[5] == 0:
            continue
        stats["quick_mode%d" % mode] = stats.get("quick_mode%d" % mode, 0) + 1
        if res is None:
            res = oc.Aln()
            rg = C.create_string_buffer(1100)
            fg = C.create_string_buffer(1100)
            assert oracle.ora_align(win.encode(), len(win), read.encode(), n, None, C.byref(_pssm(oracle, spec, strand)), 1, C.byref(res), rg, fg, None, None) == 0
        ctx = ("quick", d, spec, strand, win, read, plan, out6, (res.best, res.abc, res.aec, res.abr))
        assert (out6[0], out6[1], out6[2], out6[3]) == (res.best, res.abc, res.aec, res.abr), ctx
        r, f = script_to_strings(win, read, cols, res.abr, res.aer)
        assert r == rg.value.decode() and f == fg.value.decode(), ctx


def window(ref, pos, length, margin=50):
    s = max(0, pos - margin)
    e = min(len(ref), pos + length + margin)
    return s, e - s


def damage(rnd, read, p0=0.30, lam=0.35):
    """C->T near the 5' end, G->A near the 3' end."""
    r = list(read)
    n = len(r)
    for i in range(min(n, 25)):
        if r[i] == "C" and rnd.random() < p0 * math.exp(-lam * i):
            r[i] = "T"
        j = n - 1 - i
        if r[j] == "G" and rnd.random() < p0 * math.exp(-lam * i):
            r[j] = "A"
    return "".join(r)


@pytest.mark.parametrize("spec,strand", MATS)
def test_damaged_reads_with_substitutions(emul, oracle, spec, strand):
    rnd = random.Random(21 + strand)
    ref = "".join(rnd.choice("ACGT") for _ in range(5000))
    stats = {"n": 0}
    for i in range(500):
        n = rnd.choice([100, 100, 100, 150, rnd.randint(30, 250)])
        pos = rnd.randint(0, len(ref) - n)
        read = damage(rnd, ref[pos:pos + n], p0=0.6)
        read = mutate(rnd, read, rnd.sample(range(n), rnd.choice([0, 0, 1, 1, 2, 3, 4, 6])))
        s, l1 = window(ref, pos + rnd.randint(-3, 3) if 3 <= pos < len(ref) - n - 3 else pos, n)
        check(emul, oracle, spec, strand, ref, s, l1, read, stats)
    assert stats.get("how1", 0) > 60 and stats.get("how2", 0) + stats.get("how3", 0) > 60, stats


@pytest.mark.parametrize("spec,strand", MATS)
def test_single_indels_everywhere(emul, oracle, spec, strand):
    rnd = random.Random(7 + strand)
    ref = "".join(rnd.choice("ACGT") for _ in range(4000))
    stats = {"n": 0}
    for i in range(400):
        n = rnd.randint(30, 180)
        pos = rnd.randint(0, len(ref) - n - 40)
        src = ref[pos:pos + n + 30]
        at = rnd.choice([1, 2, 3, 5, 9, 10, 11, n // 2, n - 12, n - 10, n - 3, n - 2, rnd.randint(1, n - 2)])
        k = rnd.choice([1, 1, 1, 2, 3, 5, 8])
        if i % 2:
            read = src[:at] + src[at + k:][:n - at]
        else:
            ins = "".join(rnd.choice("ACGT") for _ in range(k))
            read = (src[:at] + ins + src[at:])[:n]
        read = damage(rnd, read)
        read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([0, 0, 1, 2, 3])))
        s, l1 = window(ref, pos, len(read))
        check(emul, oracle, spec, strand, ref, s, l1, read, stats)
    assert stats.get("how3", 0) > 100, stats


@pytest.mark.parametrize("spec,strand", MATS[:3])
def test_two_indels_and_heavy_damage(emul, oracle, spec, strand):
    rnd = random.Random(8 + strand)
    ref = "".join(rnd.choice("ACGT") for _ in range(4000))
    stats = {"n": 0}
    for i in range(350):
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
        read = damage(rnd, read[:n], p0=0.5)
        read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([0, 1, 3, 5, 8])))
        s, l1 = window(ref, pos, len(read))
        check(emul, oracle, spec, strand, ref, s, l1, read, stats)
    assert stats.get("how3", 0) > 5, stats


@pytest.mark.parametrize("spec,strand", MATS[:3])
def test_repeats_where_gap_placements_tie(emul, oracle, spec, strand):
    rnd = random.Random(9 + strand)
    stats = {"n": 0}
    for i in range(300):
        unit = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([1, 2, 3, 4, 7])))
        reps = rnd.randint(3, 12)
        left = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(150, 300)))
        right = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(150, 300)))
        ref = left + unit * reps + right
        a = rnd.randint(40, 110)
        b = rnd.randint(40, 110)
        delta = rnd.choice([-2, -1, 0, 1, 2])                    # the read has more / fewer copies of the unit
        read = left[-a:] + unit * max(0, reps + delta) + right[:b]
        if len(read) > 250:
            continue
        read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([0, 0, 1, 2])))
        pos = len(left) - a
        s, l1 = window(ref, pos, a + len(unit) * reps + b)
        check(emul, oracle, spec, strand, ref, s, l1, read, stats)
    assert stats.get("how3", 0) + stats.get("how2", 0) + stats.get("how1", 0) > 80, stats


@pytest.mark.parametrize("spec,strand", MATS[:3])
def test_clipped_windows_and_late_starts(emul, oracle, spec, strand):
    rnd = random.Random(10 + strand)
    ref = "".join(rnd.choice("ACGT") for _ in range(1500))
    stats = {"n": 0}
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
        check(emul, oracle, spec, strand, ref, s, l1, read, stats)
    assert stats.get("how3", 0) + stats.get("how2", 0) + stats.get("how1", 0) > 100, stats


def test_second_copies_compete(emul, oracle):
    """A second, slightly worse copy of the read's target a few columns away: anchors on two diagonals, ties between
    diagonals, the first maximum of the last row."""
    rnd = random.Random(12)
    stats = {"n": 0}
    for spec, strand in MATS[:3]:
        for i in range(150):
            n = rnd.randint(60, 120)
            core = "".join(rnd.choice("ACGT") for _ in range(n))
            gap = rnd.randint(0, 25)
            copy2 = mutate(rnd, core, rnd.sample(range(n), rnd.choice([0, 0, 1, 2])))
            left = "".join(rnd.choice("ACGT") for _ in range(200))
            ref = left + core + "".join(rnd.choice("ACGT") for _ in range(gap)) + copy2 + left[::-1]
            read = mutate(rnd, core, rnd.sample(range(n), rnd.choice([0, 1, 2])))
            s, l1 = window(ref, 200, n, margin=rnd.choice([50, 50 + gap + n]))
            if l1 > 760:
                continue
            check(emul, oracle, spec, strand, ref, s, l1, read, stats)
    assert stats["n"] > 300, stats


def test_stray_anchors_are_set_aside_soundly(emul, oracle):
    """Ten-mers of the read that occur a second time inside the window (bx_anchors clusters the anchors around their median
    and sets the others aside -- only when every path through none of the kept ones provably loses more): single planted
    ten-mers, second copies of long stretches of the read 13-90 columns away (which may well hold the better alignment),
    with the true locus damaged more or less."""
    rnd = random.Random(77)
    stats = {"n": 0}
    finished = 0
    for spec, strand in MATS[:3]:
        for i in range(260):
            n = rnd.choice([100, 100, 150, rnd.randint(60, 200)])
            core = "".join(rnd.choice("ACGT") for _ in range(n))
            left = "".join(rnd.choice("ACGT") for _ in range(60))
            right = "".join(rnd.choice("ACGT") for _ in range(60))
            locus = mutate(rnd, core, rnd.sample(range(n), rnd.choice([0, 0, 1, 2, 4, 6, 9])))
            kind = i % 4
            if kind == 0:                                   # one or two ten-mers of the read, planted in the flanks
                for _ in range(rnd.choice([1, 2])):
                    at = rnd.randint(0, n - 10)
                    p = rnd.randint(0, 50)
                    if rnd.random() < 0.5:
                        left = left[:p] + core[at:at + 10] + left[p + 10:]
                    else:
                        right = right[:p] + core[at:at + 10] + right[p + 10:]
                ref = left + locus + right
            elif kind == 1:                                 # a second copy of a stretch of the read right behind the locus
                a = rnd.randint(0, n - 30)
                ln = rnd.randint(20, min(90, n - a))
                seg = mutate(rnd, core[a:a + ln], rnd.sample(range(ln), rnd.choice([0, 0, 1, 3])))
                ref = left + locus + seg + right
            elif kind == 2:                                 # ... or in front of it
                a = rnd.randint(0, n - 30)
                ln = rnd.randint(20, min(90, n - a))
                seg = mutate(rnd, core[a:a + ln], rnd.sample(range(ln), rnd.choice([0, 0, 1, 3])))
                ref = left + seg + locus + right
            else:                                           # the locus itself torn apart by an insertion of unrelated bases
                cut = rnd.randint(15, n - 15)
                ref = left + locus[:cut] + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(9, 40))) + locus[cut:] + right
            read = damage(rnd, core) if spec != "flat" else core
            read = mutate(rnd, read, rnd.sample(range(n), rnd.choice([0, 1, 2])))
            pos = ref.find(locus[:25]) if kind != 2 else len(left)
            s, l1 = window(ref, max(pos, 0), n, margin=rnd.choice([50, 50, 80]))
            if l1 > 760 or l1 < n:
                continue
            finished += 1 if check(emul, oracle, spec, strand, ref, s, l1, read, stats) else 0
    assert stats["n"] > 600 and finished > 200, (stats, finished)


def test_tight_stray_bound_on_low_complexity(emul, oracle):
    """The band is as narrow as the loss budget allows once every block that occurs nowhere in the window is paid for
    (bx_finish, stray tables): cases where alignments a few diagonals off compete -- microsatellites and homopolymer runs
    inside the read, tandem duplications of 1-9 bases between read and reference at every distance from the ends,
    substitutions spread one per block (the budget left for straying is then nil) or clustered."""
    rnd = random.Random(99)
    stats = {"n": 0}
    finished = 0

    def low_complexity(n):
        out = []
        while sum(map(len, out)) < n:
            kind = rnd.random()
            if kind < 0.35:
                out.append(rnd.choice("ACGT") * rnd.randint(3, 14))
            elif kind < 0.7:
                unit = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([2, 2, 3, 4, 5])))
                out.append(unit * rnd.randint(2, 8))
            else:
                out.append("".join(rnd.choice("ACGT") for _ in range(rnd.randint(5, 25))))
        return "".join(out)[:n]

    for spec, strand in MATS[:3]:
        for i in range(420):
            n = rnd.choice([100, 100, 150, rnd.randint(40, 220)])
            core = low_complexity(n) if i % 3 else "".join(rnd.choice("ACGT") for _ in range(n))
            locus = core
            if i % 2:                                       # a tandem duplication / deletion between read and reference
                at = rnd.choice([1, 2, 4, 9, 10, 11, n // 2, n - 11, n - 9, n - 3, rnd.randint(1, n - 2)])
                k = rnd.randint(1, 9)
                if rnd.random() < 0.5:
                    locus = core[:at] + core[max(0, at - k):at] + core[at:]        # the reference repeats k bases
                else:
                    locus = core[:at] + core[at + k:]                              # the reference lacks k bases
            left = low_complexity(70) if i % 5 == 0 else "".join(rnd.choice("ACGT") for _ in range(70))
            right = low_complexity(70) if i % 7 == 0 else "".join(rnd.choice("ACGT") for _ in range(70))
            ref = left + locus + right
            read = damage(rnd, core) if spec != "flat" else core
            nsub = rnd.choice([0, 1, 2, 3, 4, 5, 6, 8])
            if i % 4 == 0:                                  # one substitution per ten-mer block
                rows = [min(n - 1, 11 * b + rnd.randint(0, 9)) for b in rnd.sample(range(max(1, n // 11)), min(nsub, max(1, n // 11)))]
            else:
                rows = rnd.sample(range(n), nsub)
            read = mutate(rnd, read, rows)
            s, l1 = window(ref, 70, len(locus), margin=rnd.choice([50, 50, 20]))
            if l1 < len(read) or l1 > 760:
                continue
            finished += 1 if check(emul, oracle, spec, strand, ref, s, l1, read, stats) else 0
    assert stats["n"] > 1000 and finished > 500, (stats, finished)


def test_new_start_quirk_under_the_written_path(emul, oracle):
    """dyn_prog's "new start" branch drops the substitution score of the row it starts in (src/mia.c:916-917): along a
    diagonal whose first rows mismatch heavily the recurrence reaches LESS than the path's own value, so a deletion right
    behind a head of 2-5 rows can win although its loss exceeds the diagonal's (found by tools/band_campaign.py, ancient
    matrix: rows 0 and 1 mismatch, a start in row 2 forfeits 214, a six-column gap wins by 30).  The band must be sized
    from the value the recurrence reaches (bx_rows_loss: nfail).  Heads whose bases face transversions on the body's
    diagonal, every gap length from 3 to 12 -- the critical one, whose cost just exceeds the diagonal's loss, among them."""
    rnd = random.Random(123)
    stats = {"n": 0}
    finished = 0
    worst = {"A": "C", "C": "A", "G": "T", "T": "G"}
    for spec, strand in MATS:
        for i in range(60):
            n = rnd.choice([100, 150, 250])
            head = rnd.randint(2, 5)
            left = "".join(rnd.choice("ACGT") for _ in range(80))
            body = "".join(rnd.choice("ACGT") for _ in range(n + 60))
            core_head = "".join(rnd.choice("ACGT") for _ in range(head))
            for k in range(max(head, 3), 13):
                filler = "".join(rnd.choice("ACGT") for _ in range(k - head)) + "".join(worst[c] for c in core_head)
                # reference: head, k unrelated bases (the last `head` of them as unlike the head as can be), body;
                # read: head + body -- a deletion of k right behind the head
                ref = left + core_head + filler + body
                read = (core_head + body)[:n]
                if i % 3 == 0:
                    read = read[:head] + damage(rnd, read)[head:]
                s, l1 = window(ref, len(left) + k, n, margin=50)
                if l1 < n or l1 > 760:
                    continue
                finished += 1 if check(emul, oracle, spec, strand, ref, s, l1, read, stats) else 0
    assert stats["n"] > 2000 and finished > 1500, (stats, finished)


def sprinkle_n(rnd, seq, rate, runs=0):
    """Ambiguity codes over a sequence as in mt311 (src/mt311.c: every tenth column a Y, R, M ...): single columns at
    `rate`, plus `runs` stretches of 2-12 of them."""
    s = list(seq)
    for i in range(len(s)):
        if rnd.random() < rate:
            s[i] = rnd.choice("YRYRMWVHDSBKN")
    for _ in range(runs):
        at = rnd.randint(0, len(s) - 13)
        for i in range(at, at + rnd.randint(2, 12)):
            s[i] = rnd.choice("YRN")
    return "".join(s)


@pytest.mark.parametrize("spec,strand", MATS)
def test_reference_with_ambiguity_codes(emul, oracle, spec, strand):
    """A reference whose columns are one in ten ambiguity codes (mt311 itself, the start of every run): an N column is a
    known small loss under any base, the table lists the 10-mers that hold up to three of them under every spelling, and
    the band follows from the same pigeonhole (bandx_body.h, N COLUMNS).  Substitutions, indels next to and across the N
    columns, damage; the plan must still finish or place most reads, and whatever it finishes must be dyn_prog's answer."""
    rnd = random.Random(31 + strand)
    stats = {"n": 0}
    finished = 0
    for rate, runs in ((0.10, 6), (0.03, 2), (0.20, 10)):
        truth = "".join(rnd.choice("ACGT") for _ in range(3000))
        ref = sprinkle_n(rnd, truth, rate, runs)
        for i in range(200):
            n = rnd.choice([100, 100, 60, 150, rnd.randint(30, 250)])
            pos = rnd.randint(0, len(ref) - n - 40)
            src = truth[pos:pos + n + 30]
            kind = i % 4
            if kind == 0:
                read = src[:n]
            elif kind == 1:
                at, k = rnd.randint(1, n - 2), rnd.choice([1, 1, 2, 3, 5, 8])
                read = src[:at] + src[at + k:][:n - at]
            elif kind == 2:
                at, k = rnd.randint(1, n - 2), rnd.choice([1, 1, 2, 3, 5])
                read = (src[:at] + "".join(rnd.choice("ACGT") for _ in range(k)) + src[at:])[:n]
            else:
                read = damage(rnd, src[:n], p0=0.6)
            read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([0, 0, 1, 1, 2, 3, 5])))
            s, l1 = window(ref, pos + rnd.randint(-3, 3) if 3 <= pos else pos, len(read))
            if l1 < len(read):
                continue
            finished += 1 if check(emul, oracle, spec, strand, ref, s, l1, read, stats) else 0
    assert stats["n"] > 550 and finished > 150, (sorted(stats.items()), finished)


def test_gaps_through_ambiguity_columns(emul, oracle):
    """With the flat matrix a column gap crosses an N column for 200 (GEP) where a row aligned to it loses 210: next to a
    stretch of N columns a deletion and the diagonal are nearly level, and which of them wins (or ties -- dyn_prog's
    cascade decides) hangs on single substitutions.  Reads over such stretches, with the stretch deleted, shortened or kept
    and 0-3 substitutions just beside it; every matrix."""
    rnd = random.Random(77)
    stats = {"n": 0}
    finished = 0
    for spec, strand in MATS:
        for i in range(120):
            n = rnd.choice([100, 100, 150, 64])
            left = "".join(rnd.choice("ACGT") for _ in range(60 + rnd.randint(0, 9)))
            a = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(12, n - 12)))
            k = rnd.randint(1, 9)
            gap_truth = "".join(rnd.choice("ACGT") for _ in range(k))
            b = "".join(rnd.choice("ACGT") for _ in range(n + 40))
            ref = sprinkle_n(rnd, left + a, 0.05) + "N" * k + sprinkle_n(rnd, b, 0.05)
            keep = rnd.choice([0, 0, k, k, rnd.randint(0, k)])                # columns of the stretch the read holds
            read = (a + gap_truth[:keep] + b)[:n]
            near = [len(a) - 1 - rnd.randint(0, 4), len(a) + keep + rnd.randint(0, 4)]
            read = mutate(rnd, read, [r for r in rnd.sample(near, rnd.choice([0, 1, 2])) if 0 <= r < len(read)])
            if i % 3 == 0:
                read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([1, 2, 3])))
            s, l1 = window(ref, len(left), len(read), margin=rnd.choice([50, 50, 30]))
            if l1 < len(read):
                continue
            finished += 1 if check(emul, oracle, spec, strand, ref, s, l1, read, stats) else 0
    assert stats["n"] > 500 and finished > 200, (stats, finished)


@pytest.mark.parametrize("spec,strand", MATS)
def test_end_indels_are_rescued(emul, oracle, spec, strand):
    """An indel within a dozen rows of a read end: the end has no block of its own, all anchors lie on one diagonal and the
    pure diagonal pays a substitution for most of the shifted rows -- over the budget.  bx_rescue tries the unanchored end
    on the diagonals one to three off and, where that explains it, writes the path down with a gap there (the band then
    follows from a B0 of one gap instead of ten substitutions).  Indels of 1-5 at 2-16 rows from either end, with and
    without a substitution beside them; sizes above three are not rescued and must simply stay correct."""
    rnd = random.Random(91 + strand)
    ref = "".join(rnd.choice("ACGT") for _ in range(4000))
    stats = {"n": 0}
    finished = small = 0
    for i in range(360):
        n = rnd.choice([100, 100, 150, 64, 128])
        pos = rnd.randint(10, len(ref) - n - 60)
        src = ref[pos:pos + n + 30]
        dist = rnd.randint(2, 16)
        k = rnd.choice([1, 1, 2, 2, 3, 3, 4, 5])
        at = dist if i % 2 else n - dist - (k if i % 4 < 2 else 0)
        if i % 4 < 2:
            read = (src[:at] + "".join(rnd.choice("ACGT") for _ in range(k)) + src[at:])[:n]      # inserted bases
        else:
            read = src[:at] + src[at + k:][:n - at]                                                # deleted columns
        if i % 3 == 0:
            read = mutate(rnd, read, [rnd.randint(0, 9) if i % 2 else rnd.randint(n - 10, n - 1)])
        if i % 5 == 0:
            read = damage(rnd, read)
        s, l1 = window(ref, pos, len(read))
        ok = check(emul, oracle, spec, strand, ref, s, l1, read, stats)
        if k <= 3:
            small += 1
            finished += 1 if ok else 0
    assert stats["n"] == 360 and finished > 0.8 * small, (sorted(stats.items()), finished, small)


@pytest.mark.parametrize("spec,strand", MATS)
def test_window_wide_n_credit(emul, oracle, spec, strand):
    """Round 4: against a reference with an ambiguity code in every tenth column B0 is mostly N columns, and three
    substitutions used to exhaust the pigeonhole's budget (a fifth of the damaged reads of configs[2] in their first
    iteration).  Every other path crosses N columns as well: the fewest any stretch of len2 window columns holds, at min(GEP,
    lambda) each, is added to the budget (bx_window_nmin / BX_LOSS_NCRED).  Reads with 2-8 substitutions; and the case the
    credit must NOT be given for -- a second copy of the locus a few columns away that is free of N columns, so that a path
    over it pays nothing of the kind: whatever the plan finishes must still be dyn_prog's answer."""
    rnd = random.Random(404 + strand)
    stats = {"n": 0}
    finished = planned = 0
    for rep in range(3):
        truth = "".join(rnd.choice("ACGT") for _ in range(3000))
        ref = sprinkle_n(rnd, truth, 0.10, 0)
        for i in range(150):
            n = rnd.choice([100, 100, 100, 150, 70])
            pos = rnd.randint(60, len(ref) - n - 60)
            read = truth[pos:pos + n]
            if i % 3 == 0:
                read = damage(rnd, read, p0=0.5)
            read = mutate(rnd, read, rnd.sample(range(n), rnd.choice([2, 2, 3, 3, 3, 4, 5, 7])))
            s, l1 = window(ref, pos + rnd.randint(-3, 3), n)
            ok = check(emul, oracle, spec, strand, ref, s, l1, read, stats)
            finished += 1 if ok else 0
    # a clean second copy beside an N-rich locus: locus at `at`, copy `shift` columns further on, both inside the window
    for i in range(150):
        n = rnd.choice([100, 100, 64])
        shift = rnd.randint(11, 40)
        core = "".join(rnd.choice("ACGT") for _ in range(n))
        left = "".join(rnd.choice("ACGT") for _ in range(80))
        right = "".join(rnd.choice("ACGT") for _ in range(120))
        locus = sprinkle_n(rnd, core, rnd.choice([0.10, 0.15]), 0)
        # the copy: every block of the read broken by one substitution, no N column at all
        copy = mutate(rnd, core, [min(n - 1, b * 11 + rnd.randint(0, 9)) for b in range(n // 11 + 1)])
        ref = left + locus[:shift] + copy if i % 2 else left + locus + right
        if i % 2:
            # (overlapping placement: the copy starts `shift` columns into the locus -- a path can leave the locus for it)
            ref = left + locus[:shift] + copy + right
        read = mutate(rnd, core, rnd.sample(range(n), rnd.choice([1, 2, 3, 4, 5])))
        s, l1 = window(ref, len(left), n, margin=50)
        if l1 < n:
            continue
        check(emul, oracle, spec, strand, ref, s, l1, read, stats)
    print("n_credit", spec, strand, sorted(stats.items()), finished)
    assert stats["n"] > 550 and finished > 150, (str(sorted(stats.items())), finished)


def test_quick_plan_on_wrapped_references(emul, oracle):
    """bx_quick's bitmaps count the start positions 0 .. L - 1 of a circular reference ONCE (the 256 codes behind L are the wrap's copies
    of the first ones): a window over the origin holds a place or its copy, never both -- unless the window is longer than the reference,
    which the quick plan must refuse.  References of 300 .. 3 000 bases, wrapped; reads anywhere, many over the origin; windows in
    wrapped coordinates; substitutions, single indels; the repeats that a short circular reference makes of itself."""
    rnd = random.Random(2026)
    stats = {"n": 0}
    for spec, strand in MATS[:3]:
        for i in range(240):
            L = rnd.choice([300, 420, 700, 1500, 3000])
            core = "".join(rnd.choice("ACGT") for _ in range(L))
            if i % 5 == 0:                                            # a tandem duplication inside the circle: repeated 10-mers
                a = rnd.randint(0, L - 80)
                core = core[:a + 40] + core[a:a + 40] + core[a + 80:]
            ref = core + core[:256]
            n = rnd.choice([60, 100, 100, 130])
            pos = rnd.choice([L - rnd.randint(1, n), L - n - rnd.randint(0, 40), rnd.randint(0, L - 1), rnd.randint(0, 50)])
            read = ref[pos:pos + n]
            if len(read) < n:
                continue
            if i % 3 == 0:
                at = rnd.randint(5, n - 5)
                read = read[:at] + read[at + 1:] if i % 2 else read[:at] + rnd.choice("ACGT") + read[at:]
            read = damage(rnd, read) if spec != "flat" else read
            read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([0, 0, 1, 2, 4])))
            s, l1 = window(ref, pos, len(read), margin=rnd.choice([50, 50, 120]))
            if l1 < len(read) or l1 > 760:
                continue
            win = ref[s:s + l1]
            res = None
            for d in range(0, l1 - len(read) + 1):
                if sum(1 for k in range(len(read)) if read[k] != win[d + k]) > 8 and sum(1 for k in range(min(len(read), 24)) if read[k] != win[d + k]) > 1:
                    continue
                stats["n"] += 1
                opts = ((d + 1) << 16) | 512 | (stats["n"] & 3)
                mode, out6, cols, plan = run_bandx(emul, oracle, spec, strand, ref, s, l1, read, opts)
                if l1 > L:
                    assert mode == 0, ("a window longer than the reference", L, l1)
                if mode == 0 or out6[5] == 0:
                    continue
                stats["planned"] = stats.get("planned", 0) + 1
                if res is None:
                    res = oc.Aln()
                    rg = C.create_string_buffer(1100)
                    fg = C.create_string_buffer(1100)
                    assert oracle.ora_align(win.encode(), len(win), read.encode(), len(read), None, C.byref(_pssm(oracle, spec, strand)), 1, C.byref(res), rg, fg, None, None) == 0
                ctx = ("quick, wrapped", L, d, spec, strand, win, read, plan, out6, (res.best, res.abc, res.aec, res.abr))
                assert (out6[0], out6[1], out6[2], out6[3]) == (res.best, res.abc, res.aec, res.abr), ctx
                r, f = script_to_strings(win, read, cols, res.abr, res.aer)
                assert r == rg.value.decode() and f == fg.value.decode(), ctx
    assert stats["n"] > 500 and stats.get("planned", 0) > 300, stats


def test_quick_plan_beside_n_columns(emul, oracle):
    """A reference with a FEW N columns (an assembly's consensus where coverage is thin; configs[4]'s has two in 100 kb): the table
    spells them out (kh.wild), and the quick plan answers for the windows that hold none -- the bitmaps count the places made of plain
    bases, which are all the places such a window has.  Windows with an N column are refused (the full plan and its N credit take
    them).  Adversarial part: a stretch copied elsewhere WITH an N put into the copy, so that a 10-mer that is unique among the plain
    places has a second home under one spelling of the N -- outside every window the quick plan answers for."""
    rnd = random.Random(606)
    stats = {"n": 0}
    for spec, strand in MATS[:3]:
        for i in range(260):
            L = rnd.choice([1200, 2000, 4000])
            core = [rnd.choice("ACGT") for _ in range(L)]
            a = rnd.randint(100, L // 2 - 200)
            b = rnd.randint(L // 2 + 100, L - 300)
            if i % 2 == 0:                                            # the copy with an N inside
                core[b:b + 120] = core[a:a + 120]
                core[b + rnd.randint(20, 100)] = "N"
            for _ in range(rnd.choice([1, 2, 4])):
                core[rnd.randint(0, L - 1)] = rnd.choice("NNRY")
            ref = "".join(core)
            n = rnd.choice([60, 100, 100, 130])
            pos = rnd.choice([a + rnd.randint(-30, 60), b + rnd.randint(-30, 60), rnd.randint(0, L - n)])
            pos = max(0, min(L - n, pos))
            read = ref[pos:pos + n]
            if any(c not in "ACGT" for c in read):
                read = "".join(c if c in "ACGT" else rnd.choice("ACGT") for c in read)
            if i % 3 == 0:
                at = rnd.randint(5, n - 5)
                read = read[:at] + read[at + 1:] if i % 2 else read[:at] + rnd.choice("ACGT") + read[at:]
            read = damage(rnd, read) if spec != "flat" else read
            read = mutate(rnd, read, rnd.sample(range(len(read)), rnd.choice([0, 0, 1, 2, 4])))
            s, l1 = window(ref, pos, len(read), margin=rnd.choice([20, 50, 50]))
            if l1 < len(read):
                continue
            win = ref[s:s + l1]
            clear = all(c in "ACGT" for c in win)
            res = None
            for d in range(0, l1 - len(read) + 1):
                if sum(1 for k in range(len(read)) if read[k] != win[d + k]) > 8 and sum(1 for k in range(min(len(read), 24)) if read[k] != win[d + k]) > 1:
                    continue
                stats["n"] += 1
                opts = ((d + 1) << 16) | (stats["n"] & 3)
                mode, out6, cols, plan = run_bandx(emul, oracle, spec, strand, ref, s, l1, read, opts)
                if not clear:
                    stats["refused"] = stats.get("refused", 0) + 1
                    assert mode == 0, ("a window with an N column", win)
                if mode == 0 or out6[5] == 0:
                    continue
                stats["planned"] = stats.get("planned", 0) + 1
                if res is None:
                    res = oc.Aln()
                    rg = C.create_string_buffer(1100)
                    fg = C.create_string_buffer(1100)
                    assert oracle.ora_align(win.encode(), len(win), read.encode(), len(read), None, C.byref(_pssm(oracle, spec, strand)), 1, C.byref(res), rg, fg, None, None) == 0
                ctx = ("quick, N columns elsewhere", L, d, spec, strand, win, read, plan, out6, (res.best, res.abc, res.aec, res.abr))
                assert (out6[0], out6[1], out6[2], out6[3]) == (res.best, res.abc, res.aec, res.abr), ctx
                r, f = script_to_strings(win, read, cols, res.abr, res.aer)
                assert r == rg.value.decode() and f == fg.value.decode(), ctx
    print("quick beside N", sorted(stats.items()))
    assert stats["n"] > 600 and stats.get("planned", 0) > 300 and stats.get("refused", 0) > 30, stats
