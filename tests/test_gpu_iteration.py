"""GPU parity: the HIP per-iteration path (through the C ABI) against the oracle on
the committed inputs, iteration by iteration until convergence."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_ctypes as oc
from conftest import GOLDEN
from mia_flow import fsdb_arrays, hip_iteration, oracle_after_pass1, pssm_array, records_from_script

pytestmark = pytest.mark.gpu

CASES = {
    # name: (ref, reads, circular, pssm file, hard_cut, cons_code, (slope, intercept))
    "s150_flat": ("mt311.fa", "s150.fa", True, None, 0, 1, None),
    "d150_anc": ("mt311.fa", "d150.fa", True, "ancient.submat.txt", 0, 1, None),
    "indel_anc": ("mt311.fa", "indel.fa", True, "ancient.submat.txt", 0, 1, None),
    "indel_anc_SN": ("mt311.fa", "indel.fa", True, "ancient.submat.txt", 0, 1, (150.0, 100.0)),
    "indel_anc_H": ("mt311.fa", "indel.fa", True, "ancient.submat.txt", 17000, 2, None),
    "fixture_c": ("tr1.fna", "tf.fna", True, None, 0, 1, None),
    # linear reference; tf11-adapt scores exactly 2000 there => strand_known == 0: it is never re-aligned and both its
    # AlnSeq pointers stay on the pass-1 slots, whose later occupants are listed twice (HISTORY.md 3.4)
    "fixture_lin": ("tr1.fna", "tf.fna", False, None, 0, 1, None),
    "fixture_lin_minus": ("tr1.fna", "tf.fna:-tf11-adapt", False, None, 0, 1, None),
    "fixture_c_anc_H": ("tr1.fna", "tf.fna", True, "ancient.submat.txt", 4000, 1, None),
    # adapter-trimmed reads (-T -a GTCAGACACGCAACAGG); two reads that are split at the origin after pass 1 are not
    # in iteration 2: their stale back_asp makes the reference list two unrelated records twice (HISTORY.md 3.4)
    "adapt_T_user": ("mt311.fa", "adapt.fa", True, None, 0, 1, None, "GTCAGACACGCAACAGG"),
}


@pytest.fixture(scope="module")
def hipmod():
    import mia_amd
    return mia_amd


@pytest.mark.parametrize("fused", [False, True, "redo"], ids=["stepwise", "iterate", "iterate-second-round"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_iterations_match_oracle(name, fused, oracle, hipmod, tmp_path):
    """fused: the whole iteration through mia_hip_iterate (planner, cut line and event count stay on the device) instead of
    mia_hip_realign + _cull + _tally + _consensus; everything below is checked the same way after either.
    "redo": mia_hip_iterate queues cull, tally and consensus without waiting for the alignment's counters; their kernels
    return at once if reads are waiting for the exact one-read-per-thread kernel, and the chain is queued a second time.
    MIA_HIP_SPEC_TEST=1 makes every iteration take that second round (where the cut line leaves the host nothing to wait for)."""
    ref_fa, reads_fa, circ, pfile, hard, cc, sn = CASES[name][:7]
    adapter = CASES[name][7] if len(CASES[name]) > 7 else None
    if ":-" in reads_fa:            # drop one record from a committed FASTA
        src, drop = reads_fa.split(":-")
        recs = open(os.path.join(GOLDEN, src)).read().split(">")[1:]
        reads_fa = str(tmp_path / "filtered.fa")
        with open(reads_fa, "w") as f:
            f.write("".join(">" + r for r in recs if r.split()[0] != drop))
    kmer = 12 if ref_fa == "mt311.fa" else -1
    st, opts, anc = oracle_after_pass1(oracle, ref_fa, reads_fa, circ, kmer, pfile, hard, cc,
                                       sn[0] if sn else None, sn[1] if sn else None, adapter)
    fs = fsdb_arrays(oracle, st)
    assert fs["n"] > 0
    if name == "fixture_lin":
        assert not fs["sk"].all()
    if fused == "redo":
        os.environ["MIA_HIP_SPEC_TEST"] = "1"
    try:
        hip = hipmod.MiaHip(0)
    finally:
        os.environ.pop("MIA_HIP_SPEC_TEST", None)
    fused = bool(fused)
    hip.set_pssm(pssm_array(anc))
    hip.upload_reads(fs["bases"], fs["offsets"], fs["rc"], fs["sk"], fs["as_"], fs["ae"])
    n_slots1 = oracle.ora_num_culled(st)
    hip.set_slot_dropped(np.array([oracle.ora_slot_at(st, i).contents.dropped for i in range(n_slots1)], np.uint8))
    hip.set_pass1_state(fs["front"], fs["back"], fs["score"])   # fs->front_asp, back_asp (-1 = NULL), score after pass 1
    lens = (fs["offsets"][1:] - fs["offsets"][:-1]).astype(np.int32)

    L0 = oracle.ora_ref_len(st)
    ref = oracle.ora_ref_seq(st)[:L0].decode()
    max_mult = 1
    for it in range(1, 8):
        oracle.ora_iterate(st, ref.encode(), it)
        score, as_, ae, cons = hip_iteration(hip, ref, circ, lens, hard, sn, cc, fused=fused)
        prm, _ = hip.record_params()
        max_mult = max(max_mult, int(prm[:, 3].max()), int(prm[:, 7].max()))
        L = len(ref)
        wrapped = ref + (ref[: min(L, 256)] if circ else "")
        cols, rstart = hip.scripts()
        dF, dB = hip.dropped()
        # per-read alignment results (fs->score/as/ae)
        for i in range(fs["n"]):
            f = oracle.ora_frag_at(st, i).contents
            if not f.strand_known:
                continue
            assert (score[i], as_[i], ae[i]) == (f.score, f.as_, f.ae), (name, it, i)
            recs = records_from_script(fs["seqs"][i], cols[i], int(rstart[i]), int(as_[i]), int(ae[i]), L, wrapped)
            slots = [f.front] + ([f.back] if len(recs) == 2 else [])
            for rec, s, d in zip(recs, slots, (dF[i], dB[i])):
                a = oracle.ora_slot_at(st, s).contents
                assert (rec["start"], rec["end"], rec["seq"]) == (a.start, a.end, a.seq.decode()), (name, it, i)
                assert rec["segment"] == a.segment.decode()
                assert bool(d) == bool(a.dropped), (name, it, i)
                for p in range(len(rec["seq"])):
                    want = a.ins[p].decode() if a.ins[p] else None
                    assert rec["ins"].get(p) == want, (name, it, i, p)
        # tallies, gaps, consensus
        exp = (C.c_int * (L * 10))()
        oracle.ora_column_tallies(st, exp)
        exp = np.ctypeslib.as_array(exp).reshape(L, 10)
        t, g = hip.get_tally()
        assert np.array_equal(t[:10, :L].T, exp), (name, it)
        eg = np.ctypeslib.as_array(oracle.ora_ref_gaps(st), shape=(L,))
        assert np.array_equal(g[:L], eg), (name, it)
        ocons = oc.consensus_string(oracle, st)
        assert cons == ocons, (name, it)
        if cons == ref:
            break
        ref = cons
    if name in ("adapt_T_user", "fixture_lin"):
        assert max_mult == 2      # the stale back_asp path was taken: some record was listed twice
    hip.close()
    oracle.ora_free(st)
