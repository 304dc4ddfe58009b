"""Shared test plumbing: drive the oracle (pass 1 + reference iterations) and the
HIP library through the same MIA flow (reference src/mia_main.c:742-964)."""
import ctypes as C
import math
import os

import numpy as np

import oracle_ctypes as oc
from conftest import GOLDEN


def oracle_after_pass1(oracle, ref_fa, reads_fa, circular=True, kmer=12, pssm_file=None, hard_cut=0, cons_code=1,
                       slope=None, intercept=None, adapter=None):
    """ora_new + load ref + pass 1 over a FASTA + finish_pass1; returns (state, opts, anc pssm)."""
    o = oc.Opts()
    oracle.ora_opts_default(C.byref(o))
    o.circular = 1 if circular else 0
    o.kmer_len = kmer
    o.hard_cut = hard_cut
    o.cons_code = cons_code
    if slope is not None:
        o.score_cut_set, o.slope, o.intercept = 1, slope, intercept
    if adapter is not None:         # -T -a <adapter>
        o.do_trim = 1
        o.adapter = adapter.encode()
    anc = oc.Pssm()
    if pssm_file:
        assert oracle.ora_pssm_read(os.path.join(GOLDEN, pssm_file).encode(), C.byref(anc)) == 1
    else:
        oracle.ora_pssm_flat(C.byref(anc))
    st = oracle.ora_new(C.byref(o), C.byref(anc))
    assert oracle.ora_load_ref_fasta(st, os.path.join(GOLDEN, ref_fa).encode()) == 1
    oracle.ora_prepare_ref(st)
    assert oracle.ora_pass1_file(st, os.path.join(GOLDEN, reads_fa).encode()) > 0
    oracle.ora_finish_pass1(st)
    return st, o, anc


def fsdb_arrays(oracle, st):
    n = oracle.ora_num_frags(st)
    seqs, rc, sk, as_, ae, score, ids, front, back = [], [], [], [], [], [], [], [], []
    for i in range(n):
        f = oracle.ora_frag_at(st, i).contents
        seqs.append(bytes(f.seq))
        rc.append(f.rc); sk.append(f.strand_known); as_.append(f.as_); ae.append(f.ae); score.append(f.score)
        ids.append(f.id.decode()); front.append(f.front); back.append(f.back)
    offsets = np.zeros(n + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    return dict(n=n, bases=bases, offsets=offsets, seqs=seqs, rc=np.array(rc, np.uint8), sk=np.array(sk, np.uint8),
                as_=np.array(as_, np.int32), ae=np.array(ae, np.int32), score=np.array(score, np.int32), ids=ids,
                front=np.array(front), back=np.array(back))


def pssm_array(p):
    return np.ctypeslib.as_array(p.sm).reshape(31, 5, 5).astype(np.int32).copy()


def hip_iteration(hip, ref_seq, circular, lens, hard_cut=0, score_cut=None, cons_code=1, slot_base=0, fused=False):
    """One pass of the per-iteration path on the GPU: reiterate_assembly + cull + tally + consensus.
    fused: through mia_hip_iterate (one call, no host round trips between the stages) instead of the four entry points."""
    if fused:
        cons = hip.iterate(ref_seq, circular, hard_cut, score_cut, cons_code)
        score, as_, ae = hip.alignments()
        return score, as_, ae, cons
    hip.realign(ref_seq, circular)
    score, as_, ae = hip.alignments()
    if hard_cut > 0:
        hip.cull(hard_cut, 0.0, 0.0, slot_base)
    else:
        if score_cut is not None:
            slope, intercept = score_cut
        else:
            slope, intercept = hip.score_cut(score, lens)      # find_fsdb_score_cut
        if slope <= 0:                                          # src/mia.c:440-442
            slope = 100.0
        hip.cull(0, slope, intercept, slot_base)
    hip.tally()
    cons = hip.consensus(cons_code)
    return score, as_, ae, cons


def records_from_script(seq, cols, ref_start, as_, ae, L, ref_wrapped):
    """Rebuild the AlnSeq records (SEQ + inserts) that merge_pwaln_into_maln would store
    (reference src/map_align.c:866-954, split as src/mia.c:1376-1438) from a column script.
    Returns list of dicts(start,end,seq,ins{pos:str},segment)."""
    rows = [r for r in range(len(seq)) if cols[r] != -2]
    ref_g, frag_g = [], []
    prev = None
    for r in rows:
        c = int(cols[r])
        if c == -1:
            ref_g.append("-"); frag_g.append(chr(seq[r])); continue
        g = ref_start + c
        if prev is not None:
            for k in range(prev + 1, g):
                ref_g.append(ref_wrapped[k]); frag_g.append("-")
        ref_g.append(ref_wrapped[g]); frag_g.append(chr(seq[r]))
        prev = g
    start, end = as_, ae
    if end > L:
        end -= L

    def build(rg, fg, start, end, seg):
        s, ins, pos, cur = [], {}, 0, None
        for a, b in zip(rg, fg):
            if a == "-":
                cur = (cur or "") + b
            else:
                if cur is not None:
                    ins[pos] = cur
                    cur = None
                s.append(b); pos += 1
        return dict(start=start, end=end, seq="".join(s), ins=ins, segment=seg)

    if start > end:
        rp, ap = start, 0
        while rp < L:
            if ref_g[ap] != "-":
                rp += 1
            ap += 1
        return [build(ref_g[:ap], frag_g[:ap], start, L - 1, "f"), build(ref_g[ap:], frag_g[ap:], 0, end, "b")]
    return [build(ref_g, frag_g, start, end, "a")]
