"""GPU parity of pass 1 (new_kmer_filter + sg_align, reference src/mia_main.c:759-805)
through the C ABI, against the oracle's pass 1 on the committed inputs."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_ctypes as oc
from conftest import GOLDEN
from mia_flow import fsdb_arrays, oracle_after_pass1, pssm_array

pytestmark = pytest.mark.gpu

CASES = {
    # name: (ref, reads, circular, kmer, soft_mask, pssm)
    "fixture_c": ("tr1.fna", "tf.fna", True, -1, False, None),
    "fixture_lin": ("tr1.fna", "tf.fna", False, -1, False, None),
    "fixture_c_k8M": ("tr1.fna", "tf.fna", True, 8, True, None),
    "fixture_c_anc": ("tr1.fna", "tf.fna", True, -1, False, "ancient.submat.txt"),
    "s150_k12": ("mt311.fa", "s150.fa", True, 12, False, None),
    "s150_full": ("mt311.fa", "s150.fa", True, -1, False, None),
    "d150_anc_k12": ("mt311.fa", "d150.fa", True, 12, False, "ancient.submat.txt"),
    "indel_anc_k10": ("mt311.fa", "indel.fa", True, 10, False, "ancient.submat.txt"),
}


def read_fasta(path):
    """id, sequence as read_fasta leaves it (upper case, truncated to 256; src/io.c:246-278)"""
    out = []
    for rec in open(path).read().split(">")[1:]:
        lines = rec.split("\n")
        out.append((lines[0].split()[0] if lines[0].split() else "", "".join(lines[1:]).replace(" ", "").upper()[:256]))
    return out


def ref_fasta(path):
    rec = open(path).read().split(">")[1]
    return "".join(rec.split("\n")[1:])


@pytest.mark.parametrize("name", sorted(CASES))
def test_pass1_matches_oracle(name, oracle):
    import mia_amd
    ref_fa, reads_fa, circ, kmer, soft, pfile = CASES[name]
    if name == "s150_full" and os.environ.get("MIA_SLOW", "0") != "1":
        # the GPU needs milliseconds; the CPU oracle 25 s for the unseeded 150-read pass 1
        pass
    o = oc.Opts()
    oracle.ora_opts_default(C.byref(o))
    o.circular, o.kmer_len, o.soft_mask = int(circ), kmer, int(soft)
    anc = oc.Pssm()
    if pfile:
        assert oracle.ora_pssm_read(os.path.join(GOLDEN, pfile).encode(), C.byref(anc)) == 1
    else:
        oracle.ora_pssm_flat(C.byref(anc))
    st = oracle.ora_new(C.byref(o), C.byref(anc))
    assert oracle.ora_load_ref_fasta(st, os.path.join(GOLDEN, ref_fa).encode()) == 1
    oracle.ora_prepare_ref(st)
    oracle.ora_pass1_file(st, os.path.join(GOLDEN, reads_fa).encode())   # no finish_pass1: raw pass-1 state
    exp = {}
    for i in range(oracle.ora_num_frags(st)):
        f = oracle.ora_frag_at(st, i).contents
        exp[f.id.decode()] = (f.score, f.rc, f.as_, f.ae, f.strand_known, f.back >= 0)

    reads = [(i, s) for i, s in read_fasta(os.path.join(GOLDEN, reads_fa)) if len(s) > 0]
    bases = np.frombuffer("".join(s for _, s in reads).encode(), dtype=np.uint8)
    offsets = np.zeros(len(reads) + 1, np.int64)
    offsets[1:] = np.cumsum([len(s) for _, s in reads])
    hip = mia_amd.MiaHip(0)
    hip.set_pssm(pssm_array(anc))
    score, rc, as_, ae, flags = hip.pass1(ref_fasta(os.path.join(GOLDEN, ref_fa)), circ, bases, offsets, kmer, soft)
    n_kept = 0
    for k, (rid, _) in enumerate(reads):
        kept = bool(flags[k] & mia_amd.P1_KEPT)
        assert kept == (rid in exp), (name, rid, int(flags[k]), int(score[k]))
        if kept:
            n_kept += 1
            e = exp[rid]
            got = (int(score[k]), int(rc[k]), int(as_[k]), int(ae[k]), int(bool(flags[k] & mia_amd.P1_STRAND_KNOWN)),
                   bool(flags[k] & mia_amd.P1_SPLIT))
            assert got == (e[0], e[1], e[2], e[3], e[4], e[5]), (name, rid, got, e)
    assert n_kept == len(exp) and n_kept > 0
    hip.close()
    oracle.ora_free(st)


def test_pass1_diag_filter_changes_nothing():
    """unfiltered pass 1 (no k-mer mask, flat matrix): the diagonal filter decides most reads without the whole-reference
    DP; a context with the filter switched off must return the same score / strand / as / ae / flags for every read --
    reads from both strands, reads across the origin of the circular reference, reads with indels, reads with N, a
    reference with ambiguity codes (mt311) as well as a resolved one, and a linear reference that holds only half of the reads"""
    import gen_data
    import mia_amd
    _, _, mt = gen_data.read_fasta_one(os.path.join(GOLDEN, "mt311.fa"))
    indiv = gen_data.resolve_individual(mt)
    n = 40_000
    d = gen_data.make_reads(indiv, n, 100, seed=31, circular=True, sub_rate=0.012, indel_rate=0.002)
    seq = d["reads"].copy()
    rng = np.random.default_rng(5)
    seq[rng.integers(0, n, 300), rng.integers(0, 100, 300)] = ord("N")
    seq[:200] = gen_data.make_reads(indiv[-150:] + indiv[:150], 200, 100, seed=32, circular=False)["reads"]   # across the origin
    offsets = np.arange(n + 1, dtype=np.int64) * 100
    for ref, circular, min_share in ((indiv, True, 0.6), (mt.upper(), True, 0.0), (indiv[:9000], False, 0.25)):
        out = []
        for off in (False, True):
            if off:
                os.environ["MIA_HIP_NO_DIAG_FILTER"] = "1"
            try:
                hip = mia_amd.MiaHip(0)
            finally:
                os.environ.pop("MIA_HIP_NO_DIAG_FILTER", None)
            hip.set_pssm(mia_amd.flat_pssm())
            out.append(hip.pass1(ref, circular, seq.reshape(-1), offsets, -1))
            decided = hip.pass1_filtered()
            assert decided == 0 if off else decided >= min_share * n, decided
            if not off and ref is not indiv and circular:       # mt311 itself: the filter decides nothing, the anchored windows most
                assert hip.pass1_anchored() > 0.5 * n, hip.pass1_anchored()
            hip.close()
        for x, y in zip(out[0], out[1]):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("pfile", ["ancient.submat.txt", "ancient.submat.solexa.pe.txt"])
def test_pass1_anchored_windows_with_a_position_specific_matrix(pfile, oracle, tmp_path):
    """pass 1 without a k-mer mask and with a position-specific matrix against a reference of plain bases: nearly every
    read is decided by the anchored windows in losses (mia_pass1_kernels.h, GEN) instead of the whole-strand DP.  The
    oracle's sg_align (reference src/mia.c:1500-1665: both strands of the whole wrapped reference) on a sample of damaged
    reads -- substitutions, indels, both strands, reads across the origin -- must give the same score, strand and end points."""
    import gen_data
    import mia_amd
    _, _, mt = gen_data.read_fasta_one(os.path.join(GOLDEN, "mt311.fa"))
    indiv = gen_data.resolve_individual(mt)[:6000]           # (the oracle's whole-strand DP is 0.1 s per read and kilobase)
    n = 48
    d = gen_data.make_reads(indiv, n, 100, seed=77, circular=True, sub_rate=0.02, indel_rate=0.004, damage=True)
    seq = d["reads"].copy()
    seq[:6] = gen_data.make_reads(indiv[-150:] + indiv[:150], 6, 100, seed=78, circular=False)["reads"]     # across the origin
    ref_fa, reads_fa = tmp_path / "ref.fa", tmp_path / "reads.fa"
    ref_fa.write_text(">r\n" + indiv + "\n")
    reads_fa.write_text("".join(">q%d\n%s\n" % (i, bytes(seq[i]).decode()) for i in range(n)))
    o = oc.Opts()
    oracle.ora_opts_default(C.byref(o))
    o.circular, o.kmer_len, o.soft_mask = 1, -1, 0
    anc = oc.Pssm()
    assert oracle.ora_pssm_read(os.path.join(GOLDEN, pfile).encode(), C.byref(anc)) == 1
    st = oracle.ora_new(C.byref(o), C.byref(anc))
    assert oracle.ora_load_ref_fasta(st, str(ref_fa).encode()) == 1
    oracle.ora_prepare_ref(st)
    oracle.ora_pass1_file(st, str(reads_fa).encode())
    exp = {}
    for i in range(oracle.ora_num_frags(st)):
        f = oracle.ora_frag_at(st, i).contents
        exp[f.id.decode()] = (f.score, f.rc, f.as_, f.ae)
    hip = mia_amd.MiaHip(0)
    hip.set_pssm(pssm_array(anc))
    offsets = np.arange(n + 1, dtype=np.int64) * 100
    score, rc, as_, ae, flags = hip.pass1(indiv, True, seq.reshape(-1), offsets, -1, False)
    assert hip.pass1_anchored() >= n // 2, hip.pass1_anchored()
    kept = 0
    for k in range(n):
        if flags[k] & mia_amd.P1_KEPT:
            kept += 1
            assert (int(score[k]), int(rc[k]), int(as_[k]), int(ae[k])) == exp["q%d" % k], (k, exp["q%d" % k])
        else:
            assert "q%d" % k not in exp
    assert kept == len(exp) and kept >= n - 2
    hip.close()
    oracle.ora_free(st)
