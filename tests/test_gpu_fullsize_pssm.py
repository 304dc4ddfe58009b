"""GPU parity at BASELINE.json's full size with the aDNA matrices (configs[2] and [3]: 1 M synthetic 100 bp reads with
C->T / G->A damage against mt311, reference matrices/ancient.submat.txt and ancient.submat.solexa.pe.txt -- sub_mat_score /
find_sm_depth, src/pssm.c:6-46; the strand picks the matrix, src/mia_main.c:179-184).

* pass 1 (forward matrix on both strands, src/mia_main.c:788-789), the first realignment (against mt311 itself) and the
  SECOND realignment (against the consensus of the whole batch -- plain bases, where the band pipeline of
  csrc/bandx_body.h takes nearly every read) of a random sample are compared with the oracle read by read;
* a context with every shortcut switched off must return the same score, end points and script for every read;
* re-aligning is idempotent, the tally is linear over a split of the read set, the iteration reaches a fixed point;
* a 100 000-read subset in a context of its own, iterated from mt311 side by side with the oracle: every read's result,
  the dropped marks, all ten tally words of every column, ref->gaps and the consensus, every iteration -- and, for 50 000
  reads of the full batch, the second realignment (against the full batch's consensus) read by read.

MIA_FULLSIZE_READS overrides the read count."""
import ctypes as C
import os

import numpy as np
import pytest

import gen_data
import oracle_ctypes as oc
from conftest import GOLDEN
from oracle_sample import PushedOracle, check_subset_iterations

pytestmark = pytest.mark.gpu

N = int(os.environ.get("MIA_FULLSIZE_READS", "1000000"))
SAMPLE = 4000
KMER = 12
MATS = ["ancient.submat.txt", "ancient.submat.solexa.pe.txt"]


class Full:
    pass


def iterate(hip, ref, lens):
    hip.realign(ref, True)
    al = hip.alignments()
    s, ic = hip.score_cut(al[0], lens)
    hip.cull(0, s if s > 0 else 100.0, ic, 0)
    hip.tally()
    return al, hip.consensus(1)


@pytest.fixture(scope="module", params=MATS)
def full(request):
    import mia_amd
    f = Full()
    f.mod = mia_amd
    f.spec = request.param
    f.pssm = mia_amd.read_pssm(os.path.join(GOLDEN, f.spec))
    _, _, mt = gen_data.read_fasta_one(os.path.join(GOLDEN, "mt311.fa"))
    f.ref = mt.upper()
    indiv = gen_data.resolve_individual(mt)
    d = gen_data.make_reads(indiv, N, 100, seed=13, circular=True, damage=True)
    f.seq = d["reads"]                                              # as sequenced
    f.offsets = np.arange(N + 1, dtype=np.int64) * 100
    hip = mia_amd.MiaHip(0)
    hip.set_pssm(f.pssm)
    f.p1 = hip.pass1(f.ref, True, f.seq.reshape(-1), f.offsets, KMER)      # score, rc, as, ae, flags
    score, rc, as_, ae, fl = f.p1
    f.kept = np.nonzero((fl & mia_amd.P1_KEPT) != 0)[0]
    assert len(f.kept) > 0.9 * N
    k = f.kept
    f.stored = np.where(rc[k, None] == 1, gen_data._COMP[f.seq[k][:, ::-1]], f.seq[k]).astype(np.uint8)
    f.n = len(k)
    f.soff = np.arange(f.n + 1, dtype=np.int64) * 100
    f.rc, f.sk = rc[k].astype(np.uint8), ((fl[k] & mia_amd.P1_STRAND_KNOWN) != 0).astype(np.uint8)
    f.as0, f.ae0 = as_[k].astype(np.int32), ae[k].astype(np.int32)
    f.lens = np.full(f.n, 100, np.int32)
    hip.upload_reads(f.stored.reshape(-1), f.soff, f.rc, f.sk, f.as0, f.ae0)
    f.al1, f.cons1 = iterate(hip, f.ref, f.lens)                   # against mt311 itself
    hip.bx_stats(reset=True)
    f.al2, f.cons2 = iterate(hip, f.cons1, f.lens)                 # against the batch's consensus: plain bases
    f.bx2 = hip.bx_stats()[0]
    f.cols2, f.rstart2 = hip.scripts()
    f.tally2, f.gaps2 = hip.get_tally()
    f.hip = hip
    yield f
    hip.close()


def absolute(cols, rstart):
    return np.where(cols >= 0, cols.astype(np.int32) + rstart[:, None], cols.astype(np.int32))


def test_band_pipeline_takes_the_bulk(full):
    seen, by_plan, by_values, by_trace = full.bx2
    assert seen >= 0.99 * full.n and by_plan + by_values + by_trace > 0.9 * full.n, full.bx2


def test_sample_against_oracle(full, oracle):
    """pass 1, the first and the second realignment of SAMPLE reads, read by read"""
    f = full
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(N, SAMPLE, replace=False))
    o = oc.Opts()
    oracle.ora_opts_default(C.byref(o))
    o.circular, o.kmer_len = 1, KMER
    anc = oc.Pssm()
    assert oracle.ora_pssm_read(os.path.join(GOLDEN, f.spec).encode(), C.byref(anc)) == 1
    st = oracle.ora_new(C.byref(o), C.byref(anc))
    assert oracle.ora_load_ref_fasta(st, os.path.join(GOLDEN, "mt311.fa").encode()) == 1
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
    hip.realign(f.cons1, True)
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
    # once more against the same reference: nothing moves
    f.hip.realign(f.cons1, True)
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
        hip.realign(f.cons1, True)
        sc, a, e = hip.alignments()
        assert np.array_equal(sc, f.al2[0][lo:hi]) and np.array_equal(a, f.al2[1][lo:hi]) and np.array_equal(e, f.al2[2][lo:hi])
        hip.cull(0, s if s > 0 else 100.0, ic, base)
        base += hip.num_records()
        hip.tally()
        parts.append(hip.get_tally())
        hip.close()
    assert np.array_equal(parts[0][0] + parts[1][0], f.tally2)
    assert np.array_equal(np.maximum(parts[0][1], parts[1][1]), f.gaps2)
    # consensus -> realign -> consensus ... stops changing within a few rounds (src/mia_main.c:905-940)
    ref, cons, rounds = f.cons1, f.cons2, 0
    while cons != ref and rounds < 8:
        ref = cons
        _, cons = iterate(f.hip, ref, f.lens)
        rounds += 1
    assert cons == ref, rounds
    assert abs(len(cons) - len(f.ref)) < 50


def test_fused_iteration_identical(full):
    """mia_hip_iterate (one call per iteration) from the same starting point: alignments, scripts, tallies, gaps and the
    consensus of both iterations equal what the four separate entry points gave"""
    f = full
    hip = f.mod.MiaHip(0)
    hip.set_pssm(f.pssm)
    hip.upload_reads(f.stored.reshape(-1), f.soff, f.rc, f.sk, f.as0, f.ae0)
    assert hip.iterate(f.ref, True) == f.cons1
    for x, y in zip(hip.alignments(), f.al1):
        assert np.array_equal(x, y)
    assert hip.iterate(f.cons1, True) == f.cons2
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
    hip.realign(f.cons1, True)
    s, ic = hip.score_cut(f.al2[0], f.lens)
    hip.cull(0, s if s > 0 else 100.0, ic, 0)
    hip.tally()
    t, g = hip.get_tally()
    assert np.array_equal(g, f.gaps2)
    for w in range(11):
        assert np.array_equal(t[w], f.tally2[w]), w
    assert hip.consensus(1) == f.cons2
    hip.close()


def test_subset_iterations_against_oracle(full, oracle):
    """100 000 reads of the batch, three iterations from mt311: alignments of every read, dropped marks, tallies, gaps
    and consensus against the oracle; the first iteration also against what the same reads got inside the 1 M batch"""
    f = full
    pick = np.sort(np.random.default_rng(21).choice(f.n, min(100_000, f.n), replace=False))
    first = tuple(a[pick] for a in f.al1)
    done, _ = check_subset_iterations(f.mod, oracle, f.ref, True, f.spec, f.pssm, f.stored[pick], f.rc[pick], f.sk[pick], f.as0[pick], f.ae0[pick],
                                      iters=3, expect_first=first)
    assert done >= 2


def test_big_sample_of_the_batch_against_oracle(full, oracle):
    """50 000 reads of the 1 M batch through both of ITS realignments -- against mt311 and against the batch's own
    consensus (plain bases: the band pipeline's home ground) -- read by read against the oracle"""
    f = full
    pick = np.sort(np.random.default_rng(22).choice(f.n, min(50_000, f.n), replace=False))
    po = PushedOracle(oracle, f.ref, True, f.spec, f.stored[pick], f.rc[pick], f.as0[pick], f.ae0[pick], sk=f.sk[pick])
    known = f.sk[pick].astype(bool)
    for it, (ref, al) in enumerate(((f.ref, f.al1), (f.cons1, f.al2)), 1):
        po.iterate(ref)
        o = po.alignments()
        for k in range(3):
            bad = np.nonzero((al[k][pick] != o[k]) & known)[0]
            assert len(bad) == 0, (it, k, len(bad), pick[bad[:5]].tolist())
    po.close()
