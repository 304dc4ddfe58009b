"""Test plumbing: the oracle's per-iteration loop on a read store whose strand and coordinates are set directly
(ora_push_frag; pinned to the reference's own loop by tests/golden/iter_push.json), with the alignments of an iteration
spread over the host's cores -- large samples of a full-size GPU run in seconds.  TEST INFRASTRUCTURE."""
import ctypes as C
import os

import numpy as np

import oracle_ctypes as oc
from conftest import GOLDEN


def threads():
    return max(1, min(os.cpu_count() or 1, 32))


class PushedOracle:
    """an oracle state with `stored` reads (uint8 [n, len] or a list of bytes) pushed at (rc, as, ae)"""

    def __init__(self, oracle, ref, circular, matrix_file, stored, rc, as_, ae, sk=None, lens=None):
        self.lib = oracle
        o = oc.Opts()
        oracle.ora_opts_default(C.byref(o))
        o.circular = 1 if circular else 0
        anc = oc.Pssm()
        if matrix_file:
            assert oracle.ora_pssm_read(os.path.join(GOLDEN, matrix_file).encode(), C.byref(anc)) == 1
        else:
            oracle.ora_pssm_flat(C.byref(anc))
        self.st = oracle.ora_new(C.byref(o), C.byref(anc))
        oracle.ora_set_ref(self.st, b"ref", b"", ref.encode() if isinstance(ref, str) else ref)
        oracle.ora_prepare_ref(self.st)
        self.n = len(rc)
        for i in range(self.n):
            s = stored[i].tobytes() if hasattr(stored[i], "tobytes") else stored[i]
            if lens is not None:
                s = s[: int(lens[i])]
            oracle.ora_push_frag(self.st, b"r%d" % i, s, int(rc[i]), int(as_[i]), int(ae[i]), 2001, 1 if sk is None else int(sk[i]))
        oracle.ora_set_threads(self.st, threads())
        self.it = 0

    def iterate(self, ref):
        self.it += 1
        self.lib.ora_iterate(self.st, ref.encode() if isinstance(ref, str) else ref, self.it)

    def alignments(self):
        """(score, as, ae) of every read, int32 [n] each: one pass over the structs"""
        out = np.empty((self.n, 3), np.int32)
        for i in range(self.n):
            f = self.lib.ora_frag_at(self.st, i).contents
            out[i] = (f.score, f.as_, f.ae)
        return out[:, 0], out[:, 1], out[:, 2]

    def tallies(self):
        L = self.lib.ora_ref_len(self.st)
        buf = (C.c_int * (L * 10))()
        self.lib.ora_column_tallies(self.st, buf)
        gaps = np.ctypeslib.as_array(self.lib.ora_ref_gaps(self.st), shape=(L,)).copy()
        return np.ctypeslib.as_array(buf).reshape(L, 10).copy(), gaps

    def dropped(self):
        """dropped flag of every read's front record"""
        out = np.zeros(self.n, np.uint8)
        for i in range(self.n):
            f = self.lib.ora_frag_at(self.st, i).contents
            if f.front >= 0:
                out[i] = self.lib.ora_slot_at(self.st, f.front).contents.dropped
        return out

    def consensus(self):
        return oc.consensus_string(self.lib, self.st)

    def close(self):
        self.lib.ora_free(self.st)


def check_subset_iterations(mod, oracle, ref, circular, matrix_file, pssm, stored, rc, sk, as0, ae0, iters=3, expect_first=None, lens=None):
    """A read set in a HIP context of its own and in a PushedOracle, iterated side by side from `ref`: after every
    iteration EVERY read's (score, as, ae), the dropped marks, all ten tally words of every column
    (consensus_assembly_string's BaseCounts, reference src/mia.c:576-595), ref->gaps and the consensus string must be
    equal.  expect_first: (score, as, ae) these reads got in iteration 1 inside a larger batch -- per-read results do not
    depend on the company, so this pins the larger batch's reads to the oracle as well.  Returns the number of
    iterations run and the final consensus."""
    n = len(rc)
    width = stored.shape[1]
    hip = mod.MiaHip(0)
    hip.set_pssm(pssm)
    if lens is None:
        offsets = np.arange(n + 1, dtype=np.int64) * width
        hip.upload_reads(stored.reshape(-1), offsets, rc, sk, as0, ae0)
    else:
        offsets = np.zeros(n + 1, np.int64)
        offsets[1:] = np.cumsum(lens)
        flat = np.concatenate([stored[i, : lens[i]] for i in range(n)])
        hip.upload_reads(flat, offsets, rc, sk, as0, ae0)
    po = PushedOracle(oracle, ref, circular, matrix_file, stored, rc, as0, ae0, sk=sk, lens=lens)
    known = sk.astype(bool)
    cur, done = ref, 0
    for it in range(1, iters + 1):
        cons = hip.iterate(cur, circular)
        po.iterate(cur)
        h = hip.alignments()
        o = po.alignments()
        for k, name in enumerate(("score", "as", "ae")):
            bad = np.nonzero((h[k] != o[k]) & known)[0]
            assert len(bad) == 0, (it, name, len(bad), bad[:5].tolist(), h[k][bad[:5]].tolist(), o[k][bad[:5]].tolist())
        if it == 1 and expect_first is not None:
            for k in range(3):
                assert np.array_equal(h[k][known], expect_first[k][known]), (it, k)
        assert np.array_equal(hip.dropped()[0][known].astype(bool), po.dropped()[known].astype(bool)), it
        L = len(cur)
        t, g = hip.get_tally()
        et, eg = po.tallies()
        assert np.array_equal(g[:L], eg), it
        bad = np.nonzero((t[:10, :L].T != et).any(axis=1))[0]
        assert len(bad) == 0, (it, "tally columns", len(bad), bad[:5].tolist())
        assert cons == po.consensus(), it
        done = it
        if cons == cur:
            break
        cur = cons
    po.close()
    hip.close()
    return done, cur
