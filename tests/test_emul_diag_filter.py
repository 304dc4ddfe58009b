"""The diagonal filter (csrc/diag_filter.h -- the source hipcc compiles into k_diag_filter) on the CPU against the
oracle's full DP.  Soundness is the whole point: whenever the filter says "this read's alignment is the gap-free
diagonal delta with K mismatches", dyn_prog / max_sg_score / find_align_begin / populate_pwaln_to_begin must say
exactly that (score 200 len - 800 K, abc = delta, aec = delta + len - 1, abr = 0, no gap in either row).  The cases
are built to sit on the filter's decision boundaries: 0-3 mismatches at chosen distances, mismatches at the read ends,
tandem repeats and homopolymers (other diagonals nearly as good), a second copy of the read's prefix / suffix elsewhere
in the window (one gap away from a better path), indels, N in read and window, windows clipped at the reference ends."""
import ctypes as C
import random

import numpy as np
import pytest

import oracle_ctypes as oc
from test_emul_align import codes, emul  # noqa: F401  (fixture)
from test_oracle_vs_golden import _pssm


def run_filter(emul, ref, s, len1, read):
    rc, c2 = codes(ref), codes(read)
    d, k = C.c_int(0), C.c_int(0)
    won = emul.emu_diag_filter(rc.ctypes.data_as(C.c_void_p), C.c_int64(len(ref)), s, len1, c2.ctypes.data_as(C.c_void_p), len(read),
                               C.byref(d), C.byref(k))
    assert won in (0, 1), "step 1 through the 10-mer table and step 1 by sliding disagree"
    return won, d.value, k.value


def check(emul, oracle, flat, ref, s, len1, read, stats):
    won, delta, k = run_filter(emul, ref, s, len1, read)
    stats[0] += 1
    if not won:
        return False
    stats[1] += 1
    win = ref[s:s + len1]
    res = oc.Aln()
    rg = C.create_string_buffer(1100)
    fg = C.create_string_buffer(1100)
    assert oracle.ora_align(win.encode(), len(win), read.encode(), len(read), None, C.byref(flat), 1, C.byref(res), rg, fg, None, None) == 0
    want = (200 * len(read) - 800 * k, delta, delta + len(read) - 1, 0, len(read) - 1)
    assert (res.best, res.abc, res.aec, res.abr, res.aer) == want, (win, read, delta, k)
    assert b"-" not in rg.value and b"-" not in fg.value and len(rg.value) == len(read), (win, read)
    return True


def mutate(rnd, read, positions):
    read = list(read)
    for p in positions:
        read[p] = rnd.choice([b for b in "ACGT" if b != read[p]])
    return "".join(read)


@pytest.fixture(scope="module")
def flat(oracle):
    return _pssm(oracle, "flat", 0)


def test_random_reads_with_planted_mismatches(emul, oracle, flat):
    rnd = random.Random(41)
    ref = "".join(rnd.choice("ACGT") for _ in range(3000))
    stats = [0, 0]
    for i in range(1500):
        len2 = rnd.choice([20, 33, 50, 64, 65, 100, 100, 128, 129, 150, 200, 256])
        st = rnd.randrange(0, len(ref) - len2)
        s = max(0, st - 50)
        len1 = min(len(ref), st + len2 - 1 + 50) - s
        k = rnd.choice([0, 0, 1, 1, 2, 2, 2, 3])
        if i % 4 == 0 and k >= 2:      # neighbours
            p0 = rnd.randrange(0, len2 - 6)
            pos = [p0, p0 + rnd.choice([1, 2, 3, 4])] + ([rnd.randrange(len2)] if k == 3 else [])
            pos = sorted(set(pos))
        elif i % 4 == 1:               # at the ends
            pos = sorted(set(rnd.choice([0, 1, 2, len2 - 1, len2 - 2, len2 - 3]) for _ in range(k)))
        else:
            pos = sorted(rnd.sample(range(len2), k))
        read = mutate(rnd, ref[st:st + len2], pos)
        won = check(emul, oracle, flat, ref, s, len1, read, stats)
        if len(pos) >= 3:
            assert not won
    assert stats[1] > 0.6 * stats[0], stats      # the filter must also be worth having


def test_repeats_and_low_complexity(emul, oracle, flat):
    rnd = random.Random(43)
    stats = [0, 0]
    for i in range(1200):
        unit = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([1, 2, 3, 5, 7, 11, 17])))
        left = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 120)))
        right = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 120)))
        ref = left + (unit * 400)[: rnd.randint(10, 180)] + right
        len2 = rnd.randint(8, min(120, len(ref)))
        st = rnd.randrange(0, len(ref) - len2 + 1)
        s = max(0, st - rnd.choice([0, 5, 50]))
        len1 = min(len(ref), st + len2 + rnd.choice([0, 5, 50])) - s
        read = mutate(rnd, ref[st:st + len2], sorted(rnd.sample(range(len2), rnd.choice([0, 1, 2, 2]))))
        check(emul, oracle, flat, ref, s, len1, read, stats)
    assert stats[1] > 50, stats


def test_second_copies_one_gap_away(emul, oracle, flat):
    """the read's prefix (suffix) also occurs a few columns off the true diagonal: with two mismatches on the diagonal a
    path that switches diagonals through one short gap can tie or win -- rule (c) must hand those reads to the DP"""
    rnd = random.Random(47)
    stats = [0, 0]
    for i in range(1500):
        len2 = rnd.randint(30, 110)
        body = "".join(rnd.choice("ACGT") for _ in range(len2))
        cut = rnd.randint(3, len2 - 3)
        gap = rnd.choice([1, 2, 3, 4, 5])
        filler = "".join(rnd.choice("ACGT") for _ in range(gap))
        if i % 2:
            true_ref = body[:cut] + filler + body[cut:]            # reference has extra columns: the read needs a column gap
        else:
            true_ref = body[:cut] + body[cut + gap:] if cut + gap < len2 - 3 else body   # reference lacks columns: row gap
        lead = "".join(rnd.choice("ACGT") for _ in range(60))
        tail = "".join(rnd.choice("ACGT") for _ in range(60))
        ref = lead + true_ref + tail
        # the read is the body with 0-2 further substitutions; diagonals left and right of the indel compete
        read = mutate(rnd, body, sorted(rnd.sample(range(len2), rnd.choice([0, 1, 2]))))
        s = max(0, 60 - 50)
        len1 = min(len(ref), 60 + len(true_ref) + 50) - s
        if len1 < len2:
            continue
        check(emul, oracle, flat, ref, s, len1, read, stats)
    assert stats[0] > 1000


def test_n_and_window_edges(emul, oracle, flat):
    rnd = random.Random(53)
    stats = [0, 0]
    ref = list("".join(rnd.choice("ACGT") for _ in range(600)))
    for p in rnd.sample(range(600), 25):
        ref[p] = "N"
    ref = "".join(ref)
    n_read_n = 0
    for i in range(1500):
        len2 = rnd.randint(5, 140)
        st = rnd.randrange(0, len(ref) - len2 + 1)
        s = max(0, st - 50)
        len1 = min(len(ref), st + len2 - 1 + 50) - s
        src = ref[st:st + len2].replace("N", rnd.choice("ACGT"))
        read = mutate(rnd, src, sorted(rnd.sample(range(len2), rnd.choice([0, 1, 2]))))
        if i % 10 == 0:
            p = rnd.randrange(len2)
            read = read[:p] + "N" + read[p + 1:]
            won, _, _ = run_filter(emul, ref, s, len1, read)
            assert not won          # reads with N are never decided by the filter
            n_read_n += 1
            continue
        check(emul, oracle, flat, ref, s, len1, read, stats)
    assert stats[1] > 150 and n_read_n > 100, stats


def test_pass1_form_both_strands(emul, oracle, flat):
    """pass 1: the read against every diagonal of both strands of a circular (wrapped) reference.  Whenever the filter
    decides, the strand choice of sg_align (reverse on a tie), score and end points of the full DP must agree -- reads
    from either strand, reads across the origin (seen twice, L columns apart: the first copy must win), palindromic
    stretches (both strands equally good: never decided), repeats."""
    from test_emul_pass1 import oracle_pass1, revcomp
    rnd = random.Random(59)
    decided = total = near_origin = 0
    for rep in range(6):
        L = rnd.choice([700, 900, 1300])
        genome = "".join(rnd.choice("ACGT") for _ in range(L))
        if rep % 2:
            p = rnd.randrange(100, L - 200)          # a reverse-complement palindrome and a tandem repeat
            half = genome[p:p + 40]
            genome = genome[:p + 40] + revcomp(half) + genome[p + 80:]
            q = rnd.randrange(0, 60)
            genome = genome[:q] + ("ACG" * 30)[:70] + genome[q + 70:]
        fw = genome + genome[:256]                   # add_ref_wrap
        rc = revcomp(fw)
        cf, cr = codes(fw), codes(rc)
        for i in range(160):
            len2 = rnd.choice([30, 50, 64, 65, 100, 130])
            st = rnd.randrange(0, L) if i % 5 else rnd.choice([0, 1, 5, L - 40, L - 10, L - 1])
            src = (genome + genome)[st:st + len2]
            if i % 2:
                src = revcomp(src)
            k = rnd.choice([0, 0, 1, 2, 2, 3])
            read = mutate(rnd, src, sorted(rnd.sample(range(len2), k)))
            if i % 9 == 0:
                p = rnd.randrange(5, len2 - 5)
                read = read[:p] + read[p + 2:]        # a deletion: never decidable
            c2 = codes(read)
            strand, delta = C.c_int(0), C.c_int(0)
            kk = emul.emu_pass1_filter(cf.ctypes.data_as(C.c_void_p), cr.ctypes.data_as(C.c_void_p), len(fw), c2.ctypes.data_as(C.c_void_p),
                                       len(read), C.byref(strand), C.byref(delta))
            total += 1
            if kk < 0:
                continue
            decided += 1
            near_origin += st < 256 or st > L - 130
            exp, want_strand = oracle_pass1(oracle, fw, rc, read, flat, None, None)
            e = exp[want_strand]
            got = (strand.value, 200 * len(read) - 800 * kk, delta.value, delta.value + len(read) - 1, 0)
            assert got == (want_strand, e.best, e.abc, e.aec, e.abr), (L, st, i, got, (want_strand, e.best, e.abc, e.aec, e.abr))
    assert decided > 0.5 * total and near_origin > 30, (decided, total, near_origin)


def test_kmer_route_of_rule_c_never_says_more_than_the_slide(emul):
    """rule (c) has a shortcut: long clean prefixes / suffixes are looked up in a 10-mer table of the reference instead of
    sliding over every diagonal, all diagonals the table does not name counting as 9.  That bound is conservative, so the
    shortcut may only prove what the slide proves -- on random, repetitive and N-holding references, clipped windows,
    second copies of the read's ends nearby -- and it must prove most of them (or it would not be worth having)."""
    rnd = random.Random(61)
    both = only_scan = cand = 0
    for i in range(2500):
        kind = i % 4
        if kind == 0:
            ref = "".join(rnd.choice("ACGT") for _ in range(700))
        elif kind == 1:
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([2, 3, 5, 9, 12, 21])))
            ref = "".join(rnd.choice("ACGT") for _ in range(200)) + (unit * 100)[:250] + "".join(rnd.choice("ACGT") for _ in range(250))
        elif kind == 2:
            ref = list("".join(rnd.choice("ACGT") for _ in range(700)))
            for p in rnd.sample(range(700), 6):
                ref[p] = "N"
            ref = "".join(ref)
        else:
            body = "".join(rnd.choice("ACGT") for _ in range(300))
            ref = body + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 40))) + body[: rnd.randint(20, 200)] + \
                "".join(rnd.choice("ACGT") for _ in range(120))
        len2 = rnd.choice([24, 30, 50, 64, 65, 100, 128, 140])
        st = rnd.randrange(0, len(ref) - len2)
        s = max(0, st - rnd.choice([0, 3, 50]))
        len1 = min(len(ref), st + len2 + rnd.choice([0, 3, 50])) - s
        src = ref[st:st + len2].replace("N", "A")
        gap = rnd.choice([1, 2, 3, 4, 8, 20, 40])
        p0 = rnd.randrange(0, max(1, len2 - gap))
        read = mutate(rnd, src, sorted({p0, min(len2 - 1, p0 + gap)}))
        rc, c2 = codes(ref), codes(read)
        v = emul.emu_step2_both(rc.ctypes.data_as(C.c_void_p), C.c_int64(len(ref)), s, len1, c2.ctypes.data_as(C.c_void_p), len(read))
        if v < 0:
            continue
        cand += 1
        assert v != 2, (ref, s, len1, read)         # table route yes, slide no: the bound would be unsound
        both += v == 3
        only_scan += v == 1
    assert cand > 800 and both > 0.6 * (both + only_scan), (cand, both, only_scan)
