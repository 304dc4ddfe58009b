"""The HIP iteration against the REFERENCE's own loop, directly: tests/golden/iter_push.json holds what
oracle/_ref/ref_iter_driver (reiterate_assembly + pop_smp_from_FSDB + cull_maln_from_fsdb + sort_aln_frags +
consensus_assembly_string of /root/reference/src/mia_main.c:931-963, linked from the reference's objects) returned for
3 000 seeded reads per set -- flat matrix, ancient.submat.txt on damaged reads, ancient.submat.solexa.pe.txt on PAIRED
damaged reads (BASELINE configs[3]'s recipe: two reads per 300 +- 30 bp fragment) -- pushed into its read store at jittered
true positions.  mia_hip_iterate on the same inputs must return the same per-read (score, as, ae) and the same consensus
in each of the three iterations, with one context and with the read store split over two and three loopback ranks."""
import hashlib
import json
import os
import threading

import numpy as np
import pytest

from conftest import GOLDEN
import make_goldens

pytestmark = pytest.mark.gpu


def sets():
    with open(os.path.join(GOLDEN, "iter_push.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("ranks", [1, 2, 3])
@pytest.mark.parametrize("name", sorted(sets().keys()))
def test_iterations_equal_the_references(name, ranks):
    import mia_amd
    want = sets()[name]
    ref, stored, rc, as_, ae = make_goldens.iter_push_inputs(name)
    assert hashlib.sha256(stored.tobytes() + rc.tobytes() + as_.tobytes()).hexdigest() == want["inputs_sha256"]
    n, L = stored.shape
    pssm = mia_amd.read_pssm(os.path.join(GOLDEN, want["matrix"])) if want["matrix"] else mia_amd.flat_pssm()
    cuts = [n * k // ranks for k in range(ranks + 1)]
    parts = []
    for k in range(ranks):
        lo, hi = cuts[k], cuts[k + 1]
        h = mia_amd.MiaHip(0)
        h.set_pssm(pssm)
        h.upload_reads(stored[lo:hi].reshape(-1), np.arange(hi - lo + 1, dtype=np.int64) * L, rc[lo:hi], np.ones(hi - lo, np.uint8), as_[lo:hi], ae[lo:hi])
        h.set_read_base(lo)
        parts.append(h)
    grp = None
    if ranks > 1:
        grp = mia_amd.LoopbackGroup(ranks)
        for k, h in enumerate(parts):
            grp.attach(h, k)
    cons = ref
    for k, it in enumerate(want["iterations"], 1):
        out = [None] * ranks

        def work(r, cur=cons):
            out[r] = parts[r].iterate(cur, True)
        th = [threading.Thread(target=work, args=(r,)) for r in range(ranks)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert all(c is not None and c == out[0] for c in out), (name, k)
        cons = out[0]
        got = np.stack([np.concatenate([h.alignments()[j] for h in parts]) for j in range(3)], axis=1).astype(np.int32)
        assert got[:8].tolist() == it["first_reads"], (name, k)
        assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == it["reads_sha256"], (name, k)
        assert len(cons) == it["cons_len"] and hashlib.sha256(cons.encode()).hexdigest() == it["cons_sha256"], (name, k)
    for h in parts:
        if grp:
            h.comm_destroy()
    if grp:
        grp.close()
    for h in parts:
        h.close()
