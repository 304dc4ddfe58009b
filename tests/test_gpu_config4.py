"""GPU parity on BASELINE.json configs[4]'s shape, one GPU's share: 150 bp aDNA-damaged reads against a 100 kb synthetic
region (seed 5), linear (no -c: src/mia_main.c:645 appends no wrap), matrices/ancient.submat.txt, -k 12
(new_kmer_filter over 100 kb, src/kmer.c:239-331; windows of 250 columns, src/mia_main.c:190-217).

* pass 1 with the k-mer filter, the first realignment (against the region itself) and the second (against the batch's
  consensus) of a random sample are compared with the oracle read by read;
* a context with every shortcut switched off must return the same score, end points and script for every read;
* re-aligning is idempotent, the tally is linear over a split of the read set, the iteration reaches a fixed point;
* 4 000 reads of the batch through both of its realignments against the oracle, and a 20 000-read subset iterated side by
  side with the oracle (every read, dropped marks, tallies, gaps, consensus).
The whole-run identity against the reference's own mia on 2 000 such reads is tests/test_gpu_g2.py::g2_c4.

MIA_CONFIG4_READS overrides the read count."""
import ctypes as C
import os

import numpy as np
import pytest

import gen_data
import oracle_ctypes as oc
from conftest import GOLDEN
from oracle_sample import PushedOracle, check_subset_iterations

pytestmark = pytest.mark.gpu

N = int(os.environ.get("MIA_CONFIG4_READS", "400000"))
SAMPLE = 120
KMER = 12
RLEN = 150


class Full:
    pass


def iterate(hip, ref, lens):
    hip.realign(ref, False)
    al = hip.alignments()
    s, ic = hip.score_cut(al[0], lens)
    hip.cull(0, s if s > 0 else 100.0, ic, 0)
    hip.tally()
    return al, hip.consensus(1)


@pytest.fixture(scope="module")
def full():
    import mia_amd
    f = Full()
    f.mod = mia_amd
    f.pssm = mia_amd.read_pssm(os.path.join(GOLDEN, "ancient.submat.txt"))
    f.ref = gen_data.random_reference(100_000, seed=5)
    d = gen_data.make_reads(f.ref, N, RLEN, seed=41, circular=False, damage=True)
    f.seq = d["reads"]                                              # as sequenced
    f.offsets = np.arange(N + 1, dtype=np.int64) * RLEN
    hip = mia_amd.MiaHip(0)
    hip.set_pssm(f.pssm)
    f.p1 = hip.pass1(f.ref, False, f.seq.reshape(-1), f.offsets, KMER)      # score, rc, as, ae, flags
    score, rc, as_, ae, fl = f.p1
    f.kept = np.nonzero((fl & mia_amd.P1_KEPT) != 0)[0]
    assert len(f.kept) > 0.9 * N
    k = f.kept
    f.stored = np.where(rc[k, None] == 1, gen_data._COMP[f.seq[k][:, ::-1]], f.seq[k]).astype(np.uint8)
    f.n = len(k)
    f.soff = np.arange(f.n + 1, dtype=np.int64) * RLEN
    f.rc, f.sk = rc[k].astype(np.uint8), ((fl[k] & mia_amd.P1_STRAND_KNOWN) != 0).astype(np.uint8)
    f.as0, f.ae0 = as_[k].astype(np.int32), ae[k].astype(np.int32)
    f.lens = np.full(f.n, RLEN, np.int32)
    hip.upload_reads(f.stored.reshape(-1), f.soff, f.rc, f.sk, f.as0, f.ae0)
    hip.bx_stats(reset=True)
    f.al1, f.cons1 = iterate(hip, f.ref, f.lens)                   # against the region itself
    f.bx1 = hip.bx_stats(reset=True)[0]
    f.al2, f.cons2 = iterate(hip, f.cons1, f.lens)                 # against the batch's consensus
    f.cols2, f.rstart2 = hip.scripts()
    f.tally2, f.gaps2 = hip.get_tally()
    f.hip = hip
    yield f
    hip.close()


def absolute(cols, rstart):
    return np.where(cols >= 0, cols.astype(np.int32) + rstart[:, None], cols.astype(np.int32))


def test_band_pipeline_takes_the_bulk(full):
    seen, by_plan, by_values, by_trace = full.bx1
    assert seen >= 0.99 * full.n and by_plan + by_values + by_trace > 0.9 * full.n, full.bx1


def test_sample_against_oracle(full, oracle):
    """pass 1 (-k 12), the first and the second realignment of SAMPLE reads, read by read"""
    f = full
    rng = np.random.default_rng(6)
    pick = np.sort(rng.choice(N, SAMPLE, replace=False))
    o = oc.Opts()
    oracle.ora_opts_default(C.byref(o))
    o.circular, o.kmer_len = 0, KMER
    anc = oc.Pssm()
    assert oracle.ora_pssm_read(os.path.join(GOLDEN, "ancient.submat.txt").encode(), C.byref(anc)) == 1
    st = oracle.ora_new(C.byref(o), C.byref(anc))
    oracle.ora_set_ref(st, b"region100k", b"synthetic", f.ref.encode())
    oracle.ora_prepare_ref(st)
    for i in pick:
        oracle.ora_pass1_read(st, b"r%d" % i, b"", f.seq[i].tobytes())
    oracle.ora_finish_pass1(st)
    score, rc, as_, ae, fl = f.p1
    kept_pick = [int(i) for i in pick if fl[i] & f.mod.P1_KEPT]
    assert oracle.ora_num_frags(st) == len(kept_pick)
    pos_in_store = {int(g): j for j, g in enumerate(f.kept)}
    for j, i in enumerate(kept_pick):
        fr = oracle.ora_frag_at(st, j).contents
        assert fr.id.decode() == "r%d" % i
        assert (fr.score, fr.rc, fr.as_, fr.ae, fr.strand_known) == (score[i], rc[i], as_[i], ae[i], 1 if fl[i] & f.mod.P1_STRAND_KNOWN else 0), i
    L0 = oracle.ora_ref_len(st)
    ref = oracle.ora_ref_seq(st)[:L0]
    for it, (new_ref, al) in enumerate(((ref, f.al1), (f.cons1.encode(), f.al2)), 1):
        oracle.ora_iterate(st, new_ref, it)
        sc, a, e = al
        for j, i in enumerate(kept_pick):
            fr = oracle.ora_frag_at(st, j).contents
            if not fr.strand_known:
                continue
            s = pos_in_store[i]
            assert (fr.score, fr.as_, fr.ae) == (sc[s], a[s], e[s]), (it, i)
    oracle.ora_free(st)


def test_shortcuts_change_nothing(full):
    """the second realignment again in a context that runs the full-window DP kernels only"""
    f = full
    os.environ["MIA_HIP_NO_DIAG_FILTER"] = "1"
    try:
        hip = f.mod.MiaHip(0)
    finally:
        os.environ.pop("MIA_HIP_NO_DIAG_FILTER", None)
    hip.set_pssm(f.pssm)
    sc1, as1, ae1 = f.al1
    hip.upload_reads(f.stored.reshape(-1), f.soff, f.rc, f.sk, as1, ae1)
    hip.realign(f.cons1, False)
    assert sum(hip.bx_stats()[0]) == 0
    for x, y in zip(hip.alignments(), f.al2):
        assert np.array_equal(x, y)
    cols, rstart = hip.scripts()
    sk = f.sk.astype(bool)
    assert np.array_equal(absolute(cols, rstart)[sk], absolute(f.cols2, f.rstart2)[sk])
    s, ic = hip.score_cut(f.al2[0], f.lens)
    hip.cull(0, s if s > 0 else 100.0, ic, 0)
    hip.tally()
    t, g = hip.get_tally()
    assert np.array_equal(t, f.tally2) and np.array_equal(g, f.gaps2)
    assert hip.consensus(1) == f.cons2
    hip.close()


def test_idempotent_linear_and_convergent(full):
    f = full
    f.hip.realign(f.cons1, False)
    for a, b in zip(f.al2, f.hip.alignments()):
        assert np.array_equal(a, b)
    cols, rstart = f.hip.scripts()
    assert np.array_equal(absolute(cols, rstart), absolute(f.cols2, f.rstart2))
    # the two halves of the read set, each in its own context (what two ranks would hold): tallies add, gaps combine by maximum
    s, ic = f.hip.score_cut(f.al2[0], f.lens)
    h = f.n // 2
    parts, base = [], 0
    for lo, hi in ((0, h), (h, f.n)):
        hip = f.mod.MiaHip(0)
        hip.set_pssm(f.pssm)
        hip.upload_reads(f.stored[lo:hi].reshape(-1), f.soff[: hi - lo + 1], f.rc[lo:hi], f.sk[lo:hi], f.al1[1][lo:hi], f.al1[2][lo:hi])
        hip.realign(f.cons1, False)
        sc, a, e = hip.alignments()
        assert np.array_equal(sc, f.al2[0][lo:hi]) and np.array_equal(a, f.al2[1][lo:hi]) and np.array_equal(e, f.al2[2][lo:hi])
        hip.cull(0, s if s > 0 else 100.0, ic, base)
        base += hip.num_records()
        hip.tally()
        parts.append(hip.get_tally())
        hip.close()
    assert np.array_equal(parts[0][0] + parts[1][0], f.tally2)
    assert np.array_equal(np.maximum(parts[0][1], parts[1][1]), f.gaps2)
    ref, cons, rounds = f.cons1, f.cons2, 0
    while cons != ref and rounds < 8:
        ref = cons
        _, cons = iterate(f.hip, ref, f.lens)
        rounds += 1
    assert cons == ref, rounds
    assert abs(len(cons) - len(f.ref)) < 100


def test_fused_iteration_identical(full):
    """mia_hip_iterate from the same starting point gives what the four separate entry points gave (150 bp reads: the
    one-read-per-wavefront window kernel takes what the band pipeline leaves, its range read from the device)"""
    f = full
    hip = f.mod.MiaHip(0)
    hip.set_pssm(f.pssm)
    hip.upload_reads(f.stored.reshape(-1), f.soff, f.rc, f.sk, f.as0, f.ae0)
    assert hip.iterate(f.ref, False) == f.cons1
    for x, y in zip(hip.alignments(), f.al1):
        assert np.array_equal(x, y)
    assert hip.iterate(f.cons1, False) == f.cons2
    for x, y in zip(hip.alignments(), f.al2):
        assert np.array_equal(x, y)
    cols, rstart = hip.scripts()
    sk = f.sk.astype(bool)
    assert np.array_equal(absolute(cols, rstart)[sk], absolute(f.cols2, f.rstart2)[sk])
    t, g = hip.get_tally()
    assert np.array_equal(t, f.tally2) and np.array_equal(g, f.gaps2)
    hip.close()


def test_window_tally_equals_plain_atomic_tally(full):
    """the LDS-window tally (per-lane paths, bases of depth code 15 only counted and scored at the flush, one-gap and
    over-the-origin reads one per lane) against the plain global-atomic tally that walks every script (MIA_HIP_NO_BINNED_TALLY=1):
    all twelve words of every column, ref->gaps, the consensus and the insert-column tallies"""
    f = full
    os.environ["MIA_HIP_NO_BINNED_TALLY"] = "1"
    try:
        hip = f.mod.MiaHip(0)
    finally:
        os.environ.pop("MIA_HIP_NO_BINNED_TALLY", None)
    hip.set_pssm(f.pssm)
    hip.upload_reads(f.stored.reshape(-1), f.soff, f.rc, f.sk, f.al1[1], f.al1[2])
    hip.realign(f.cons1, False)
    s, ic = hip.score_cut(f.al2[0], f.lens)
    hip.cull(0, s if s > 0 else 100.0, ic, 0)
    hip.tally()
    t, g = hip.get_tally()
    assert np.array_equal(g, f.gaps2)
    for w in range(11):
        assert np.array_equal(t[w], f.tally2[w]), w
    assert hip.consensus(1) == f.cons2
    hip.close()


def test_big_sample_of_the_batch_against_oracle(full, oracle):
    """4 000 reads of the batch (pass-1 coordinates from the GPU, themselves pinned by the 120-read sample above) through
    both realignments -- against the region and against the batch's consensus -- read by read against the oracle"""
    f = full
    pick = np.sort(np.random.default_rng(31).choice(f.n, min(4000, f.n), replace=False))
    po = PushedOracle(oracle, f.ref, False, "ancient.submat.txt", f.stored[pick], f.rc[pick], f.as0[pick], f.ae0[pick], sk=f.sk[pick])
    known = f.sk[pick].astype(bool)
    for it, (ref, al) in enumerate(((f.ref, f.al1), (f.cons1, f.al2)), 1):
        po.iterate(ref)
        o = po.alignments()
        for k in range(3):
            bad = np.nonzero((al[k][pick] != o[k]) & known)[0]
            assert len(bad) == 0, (it, k, len(bad), pick[bad[:5]].tolist())
    po.close()


def test_subset_iterations_against_oracle(full, oracle):
    f = full
    pick = np.sort(np.random.default_rng(32).choice(f.n, min(20_000, f.n), replace=False))
    first = tuple(a[pick] for a in f.al1)
    done, _ = check_subset_iterations(f.mod, oracle, f.ref, False, "ancient.submat.txt", f.pssm, f.stored[pick], f.rc[pick], f.sk[pick], f.as0[pick],
                                      f.ae0[pick], iters=3, expect_first=first)
    assert done >= 2
