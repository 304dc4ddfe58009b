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
