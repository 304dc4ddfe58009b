"""GPU parity at BASELINE.json's full size (1 M synthetic 100 bp reads against mt311):

* pass 1 (k-mer filter + whole-reference DP, both strands) and the first realignment of a random
  sample of reads are compared with the oracle read by read (per-read results do not depend on the
  other reads of the batch, so the sample pins the full-size run);
* size-independent properties of the whole batch: re-aligning is idempotent, results do not depend
  on the batch a read travels in, the tally is linear over a split of the read set (the multi-GPU
  decomposition: sum of the tallies, maximum of the gaps) and the iteration reaches a fixed point.

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


class Full:
    pass


@pytest.fixture(scope="module")
def full():
    import mia_amd
    f = Full()
    f.mod = mia_amd
    _, _, mt = gen_data.read_fasta_one(os.path.join(GOLDEN, "mt311.fa"))
    f.ref = mt.upper()
    indiv = gen_data.resolve_individual(mt)
    d = gen_data.make_reads(indiv, N, 100, seed=11, circular=True)
    f.seq = d["reads"]                                              # as sequenced
    f.offsets = np.arange(N + 1, dtype=np.int64) * 100
    hip = mia_amd.MiaHip(0)
    hip.set_pssm(mia_amd.flat_pssm())
    f.p1 = hip.pass1(f.ref, True, f.seq.reshape(-1), f.offsets, KMER)      # score, rc, as, ae, flags
    score, rc, as_, ae, fl = f.p1
    f.kept = np.nonzero((fl & mia_amd.P1_KEPT) != 0)[0]
    assert len(f.kept) > 0.95 * N
    k = f.kept
    # the read store keeps reverse-strand reads reverse-complemented (src/fsdb.c:209-227)
    f.stored = np.where(rc[k, None] == 1, gen_data._COMP[f.seq[k][:, ::-1]], f.seq[k]).astype(np.uint8)
    f.n = len(k)
    f.soff = np.arange(f.n + 1, dtype=np.int64) * 100
    f.rc, f.sk = rc[k].astype(np.uint8), ((fl[k] & mia_amd.P1_STRAND_KNOWN) != 0).astype(np.uint8)
    f.as0, f.ae0 = as_[k].astype(np.int32), ae[k].astype(np.int32)
    f.lens = np.full(f.n, 100, np.int32)
    hip.upload_reads(f.stored.reshape(-1), f.soff, f.rc, f.sk, f.as0, f.ae0)
    hip.realign(f.ref, True)
    f.al = hip.alignments()
    f.cols, f.rstart = hip.scripts()
    f.cut = hip.score_cut(f.al[0], f.lens)
    hip.cull(0, f.cut[0] if f.cut[0] > 0 else 100.0, f.cut[1], 0)
    hip.tally()
    f.tally, f.gaps = hip.get_tally()
    f.hip = hip
    yield f
    hip.close()


def absolute(cols, rstart):
    return np.where(cols >= 0, cols.astype(np.int32) + rstart[:, None], cols.astype(np.int32))


def test_sample_against_oracle(full, oracle):
    """pass 1 and the first realignment of SAMPLE reads, read by read"""
    f = full
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(N, SAMPLE, replace=False))
    o = oc.Opts()
    oracle.ora_opts_default(C.byref(o))
    o.circular, o.kmer_len = 1, KMER
    anc = oc.Pssm()
    oracle.ora_pssm_flat(C.byref(anc))
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
    oracle.ora_iterate(st, ref, 1)
    sc1, as1, ae1 = f.al
    for j, i in enumerate(kept_pick):
        fr = oracle.ora_frag_at(st, j).contents
        if not fr.strand_known:
            continue
        s = pos_in_store[i]
        assert (fr.score, fr.as_, fr.ae) == (sc1[s], as1[s], ae1[s]), i
    oracle.ora_free(st)


def test_unfiltered_pass1_sample(full, oracle):
    """the default mia run has no k-mer filter: whole-reference DP of 200 000 reads on the wide plain-key kernel,
    150 of them checked against the oracle (30 ms per read there)"""
    f = full
    m = min(N, 200_000)
    score, rc, as_, ae, fl = f.hip.pass1(f.ref, True, f.seq[:m].reshape(-1), f.offsets[: m + 1], -1)
    assert ((fl & f.mod.P1_PASSED) != 0).all()
    rng = np.random.default_rng(9)
    pick = np.sort(rng.choice(m, 150, replace=False))
    o = oc.Opts()
    oracle.ora_opts_default(C.byref(o))
    o.circular, o.kmer_len = 1, -1
    anc = oc.Pssm()
    oracle.ora_pssm_flat(C.byref(anc))
    st = oracle.ora_new(C.byref(o), C.byref(anc))
    assert oracle.ora_load_ref_fasta(st, os.path.join(GOLDEN, "mt311.fa").encode()) == 1
    oracle.ora_prepare_ref(st)
    for i in pick:
        oracle.ora_pass1_read(st, b"r%d" % i, b"", f.seq[i].tobytes())
    oracle.ora_finish_pass1(st)
    kept_pick = [int(i) for i in pick if fl[i] & f.mod.P1_KEPT]
    assert oracle.ora_num_frags(st) == len(kept_pick) and len(kept_pick) > 100
    for j, i in enumerate(kept_pick):
        fr = oracle.ora_frag_at(st, j).contents
        assert (fr.score, fr.rc, fr.as_, fr.ae) == (score[i], rc[i], as_[i], ae[i]), i
    oracle.ora_free(st)


def test_score_sums_on_device(full):
    """pass 1 of find_fsdb_score_cut reduced on the device equals numpy's sums, and with equally long reads the
    regression from the sums equals the host regression over the downloaded scores (NaN pattern included)"""
    f = full
    sc = f.hip.scores()
    used = sc >= 2000
    s5 = f.hip.score_sums()
    assert s5.tolist() == [int(f.lens[used].sum()), int(sc[used].astype(np.int64).sum()), int(used.sum()), 100, 100]
    a = f.hip.score_cut_from_sums(s5)
    b = f.hip.score_cut(sc, f.lens)
    assert a is not None and all((x == y) or (np.isnan(x) and np.isnan(y)) for x, y in zip(a, b)), (a, b)
    s5[4] += 1                       # reads of different lengths: the host regression is needed
    assert f.hip.score_cut_from_sums(s5) is None
    # by-products of the same sweep: the record count and the number of links the next cull emits
    n_rec, n_lnk = f.hip.pre_cull_counts()
    assert n_rec == f.hip.num_records()
    f.hip.cull(0, f.cut[0] if f.cut[0] > 0 else 100.0, f.cut[1], 0)
    assert f.hip.links()[1] == n_lnk


def test_realign_idempotent(full):
    f = full
    f.hip.realign(f.ref, True)
    again = f.hip.alignments()
    for a, b in zip(f.al, again):
        assert np.array_equal(a, b)
    # scripts are relative to the window, and the window follows the updated coordinates: compare reference columns
    cols, rstart = f.hip.scripts()
    assert np.array_equal(absolute(cols, rstart), absolute(f.cols, f.rstart))


def test_batch_independence_and_tally_linearity(full):
    """the two halves of the read set, each in its own context (what two ranks would hold): identical per-read
    results, tallies add up, gaps combine by maximum"""
    f = full
    h = f.n // 2
    parts = []
    base = 0
    for lo, hi in ((0, h), (h, f.n)):
        hip = f.mod.MiaHip(0)
        hip.set_pssm(f.mod.flat_pssm())
        hip.upload_reads(f.stored[lo:hi].reshape(-1), f.soff[: hi - lo + 1], f.rc[lo:hi], f.sk[lo:hi], f.as0[lo:hi], f.ae0[lo:hi])
        hip.realign(f.ref, True)
        sc, a, e = hip.alignments()
        assert np.array_equal(sc, f.al[0][lo:hi]) and np.array_equal(a, f.al[1][lo:hi]) and np.array_equal(e, f.al[2][lo:hi])
        cols, rstart = hip.scripts()
        assert np.array_equal(cols, f.cols[lo:hi]) and np.array_equal(rstart, f.rstart[lo:hi])
        hip.cull(0, f.cut[0] if f.cut[0] > 0 else 100.0, f.cut[1], base)
        base += hip.num_records()
        hip.tally()
        parts.append(hip.get_tally())
        hip.close()
    assert np.array_equal(parts[0][0] + parts[1][0], f.tally)
    assert np.array_equal(np.maximum(parts[0][1], parts[1][1]), f.gaps)


def test_iteration_reaches_fixed_point(full):
    """consensus -> realign -> consensus ... stops changing within a few rounds (src/mia_main.c:905-940)"""
    f = full
    ref, cons, rounds = None, f.ref, 0
    while cons != ref and rounds < 8:
        ref = cons
        f.hip.realign(ref, True)
        sc, _, _ = f.hip.alignments()
        s, ic = f.hip.score_cut(sc, f.lens)
        f.hip.cull(0, s if s > 0 else 100.0, ic, 0)
        f.hip.tally()
        cons = f.hip.consensus(1)
        rounds += 1
    assert cons == ref, rounds
    assert rounds >= 2 and abs(len(cons) - len(f.ref)) < 50


def test_diag_filter_changes_nothing(full):
    """the diagonal filter (csrc/diag_filter.h) finishes most reads before any DP runs; a context with the filter
    switched off must deliver the same scores, end points and scripts for every read of the full-size batch, and for
    a second batch whose reads carry more substitutions and sit in a reference with planted tandem repeats"""
    f = full

    def both(ref, stored, soff, rc, sk, as0, ae0, min_share):
        out = []
        for off in (False, True):
            if off:
                os.environ["MIA_HIP_NO_DIAG_FILTER"] = "1"
            try:
                hip = f.mod.MiaHip(0)
            finally:
                os.environ.pop("MIA_HIP_NO_DIAG_FILTER", None)
            hip.set_pssm(f.mod.flat_pssm())
            hip.upload_reads(stored.reshape(-1), soff, rc, sk, as0, ae0)
            hip.realign(ref, True)
            sc, a, e = hip.alignments()
            cols, rstart = hip.scripts()
            st = hip.filter_stats()
            assert st[1] == 0 if off else st[1] >= min_share * len(rc), st
            out.append((sc, a, e, absolute(cols, rstart)))
            hip.close()
        for x, y in zip(out[0], out[1]):
            assert np.array_equal(x, y)

    # mt311 itself: half of its columns carry ambiguity codes (N for the aligner), few reads can be decided early
    both(f.ref, f.stored, f.soff, f.rc, f.sk, f.as0, f.ae0, 0.0)
    # the usual case, a consensus of plain bases: the filter must take the bulk
    both(gen_data.resolve_individual(f.ref), f.stored, f.soff, f.rc, f.sk, f.as0, f.ae0, 0.6)
    # a harder batch: 1.5 % substitutions, 0.3 % indels, reference with tandem repeats and homopolymer runs
    rng = np.random.default_rng(77)
    g = list(gen_data.resolve_individual(f.ref))
    for _ in range(60):
        p = int(rng.integers(0, len(g) - 200))
        unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 7))))
        run = (unit * 100)[: int(rng.integers(12, 90))]
        g[p:p + len(run)] = list(run)
    ref2 = "".join(g)
    m = 300_000
    d = gen_data.make_reads(ref2, m, 100, seed=12, circular=True, sub_rate=0.015, indel_rate=0.003)
    start, strand = d["start"], d["strand"]
    stored = gen_data.stored_orientation(d)
    soff = np.arange(m + 1, dtype=np.int64) * 100
    as0 = start.astype(np.int32)
    ae0 = (start + 99).astype(np.int32)
    both(ref2, stored, soff, strand.astype(np.uint8), np.ones(m, np.uint8), as0, ae0, 0.3)


def test_subset_iterations_against_oracle(full, oracle):
    """100 000 reads of the batch in a context of their own, three iterations from mt311 side by side with the oracle:
    every read's (score, as, ae), the dropped marks, all ten tally words of every column, ref->gaps and the consensus;
    the first iteration also against what the same reads got inside the 1 M batch"""
    f = full
    pick = np.sort(np.random.default_rng(21).choice(f.n, min(100_000, f.n), replace=False))
    first = tuple(a[pick] for a in f.al)
    done, _ = check_subset_iterations(f.mod, oracle, f.ref, True, None, f.mod.flat_pssm(), f.stored[pick], f.rc[pick], f.sk[pick], f.as0[pick], f.ae0[pick],
                                      iters=3, expect_first=first)
    assert done >= 2


def test_big_sample_of_the_batch_against_oracle(full, oracle):
    """the whole 1 M batch iterated twice in a fresh context (mt311, then its own consensus: plain bases, where the plan
    finishes four reads in five without a DP); 50 000 of its reads through both realignments against the oracle"""
    f = full
    hip = f.mod.MiaHip(0)
    hip.set_pssm(f.mod.flat_pssm())
    hip.upload_reads(f.stored.reshape(-1), f.soff, f.rc, f.sk, f.as0, f.ae0)
    cons1 = hip.iterate(f.ref, True)
    al1 = hip.alignments()
    hip.iterate(cons1, True)
    al2 = hip.alignments()
    hip.close()
    for a, b in zip(al1, f.al):
        assert np.array_equal(a, b)
    pick = np.sort(np.random.default_rng(22).choice(f.n, min(50_000, f.n), replace=False))
    po = PushedOracle(oracle, f.ref, True, None, f.stored[pick], f.rc[pick], f.as0[pick], f.ae0[pick], sk=f.sk[pick])
    known = f.sk[pick].astype(bool)
    for it, (ref, al) in enumerate(((f.ref, al1), (cons1, al2)), 1):
        po.iterate(ref)
        o = po.alignments()
        for k in range(3):
            bad = np.nonzero((al[k][pick] != o[k]) & known)[0]
            assert len(bad) == 0, (it, k, len(bad), pick[bad[:5]].tolist())
    po.close()
