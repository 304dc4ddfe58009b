"""BASELINE.json configs[3] end to end on one GPU: 10 M paired 100 bp reads (two per 300 +- 30 bp fragment, ids /1 /2,
aDNA damage: tools/gen_data.make_paired_reads) against mt311, circular,
matrices/ancient.submat.solexa.pe.txt -- 4.2 GB of read store in HBM.  The reference reads one sequence at a time and
never looks at the ids (/root/reference/src/mia_main.c:759-805), so the mates are independent reads; its own loop
cannot hold this many (5.4 KB of host memory per AlnSeq), hence samples and size-independent properties:

* two iterations from mt311 (one call each, mia_hip_iterate); 20 000 sampled reads through both realignments against the
  oracle read by read (pinned to the reference's loop on the same recipe by tests/golden/iter_push.json::ip_pe);
* a 1 M-read slice with every shortcut off (full-window DP kernels only): same score, end points and script;
* the 8-GPU decomposition of north_star -- the read store split into eight contiguous shards, eight contexts on eight
  host threads joined by the library's loopback transport, mia_hip_iterate's own sharded path (pre-cull all-gather,
  all-reduce of tallies and gaps, event exchange): every rank returns the single context's consensus in both iterations,
  the concatenated per-read results and the reduced tallies are identical;
* the consensus is a fixed point after a few more rounds.
MIA_CONFIG3_READS overrides the read count."""
import os
import threading

import numpy as np
import pytest

import gen_data
from conftest import GOLDEN
from oracle_sample import PushedOracle

pytestmark = pytest.mark.gpu

N = int(os.environ.get("MIA_CONFIG3_READS", "10000000"))
SPEC = "ancient.submat.solexa.pe.txt"
L = 100


class Full:
    pass


def upload(mod, f, lo, hi, as_, ae, env=None):
    if env:
        os.environ[env] = "1"
    try:
        hip = mod.MiaHip(0)
    finally:
        if env:
            os.environ.pop(env, None)
    hip.set_pssm(f.pssm)
    hip.upload_reads(f.stored[lo:hi].reshape(-1), np.arange(hi - lo + 1, dtype=np.int64) * L, f.rc[lo:hi], np.ones(hi - lo, np.uint8), as_[lo:hi], ae[lo:hi])
    hip.set_read_base(lo)
    return hip


@pytest.fixture(scope="module")
def full():
    import mia_amd
    f = Full()
    f.mod = mia_amd
    f.pssm = mia_amd.read_pssm(os.path.join(GOLDEN, SPEC))
    _, _, mt = gen_data.read_fasta_one(os.path.join(GOLDEN, "mt311.fa"))
    f.ref = mt.upper()
    indiv = gen_data.resolve_individual(mt)
    d = gen_data.make_paired_reads(indiv, N, L, seed=4, damage=True)        # (seed 4: bench.py's make_workload(3, 10 000 000, 4) -- the bytes configs3 times)
    assert (d["strand"][0::2] != d["strand"][1::2]).all()                  # mates face each other
    f.stored = gen_data.stored_orientation(d)
    f.rc = d["strand"].astype(np.uint8)
    f.as0 = d["start"].astype(np.int32)
    f.ae0 = (f.as0 + L - 1).astype(np.int32)
    hip = upload(mia_amd, f, 0, N, f.as0, f.ae0)
    f.cons1 = hip.iterate(f.ref, True)
    f.al1 = hip.alignments()
    f.cons2 = hip.iterate(f.cons1, True)
    f.al2 = hip.alignments()
    f.tally2, f.gaps2 = hip.get_tally()
    f.hip = hip
    yield f
    hip.close()


def test_sample_against_oracle(full, oracle):
    f = full
    pick = np.sort(np.random.default_rng(41).choice(N, min(20_000, N), replace=False))
    po = PushedOracle(oracle, f.ref, True, SPEC, f.stored[pick], f.rc[pick], f.as0[pick], f.ae0[pick])
    for it, (ref, al) in enumerate(((f.ref, f.al1), (f.cons1, f.al2)), 1):
        po.iterate(ref)
        o = po.alignments()
        for k in range(3):
            bad = np.nonzero(al[k][pick] != o[k])[0]
            assert len(bad) == 0, (it, k, len(bad), pick[bad[:5]].tolist())
    po.close()


def test_shortcuts_change_nothing_on_a_slice(full):
    f = full
    m = min(1_000_000, N)
    hip = upload(f.mod, f, 0, m, f.al1[1], f.al1[2], env="MIA_HIP_NO_DIAG_FILTER")
    hip.realign(f.cons1, True)
    assert sum(hip.bx_stats()[0]) == 0
    for x, y in zip(hip.alignments(), f.al2):
        assert np.array_equal(x, y[:m])
    cols, rstart = hip.scripts()
    hip.close()
    own = upload(f.mod, f, 0, m, f.al1[1], f.al1[2])
    own.realign(f.cons1, True)
    c2, r2 = own.scripts()
    own.close()
    absolute = lambda c, r: np.where(c >= 0, c.astype(np.int32) + r[:, None], c.astype(np.int32))   # noqa: E731
    assert np.array_equal(absolute(cols, rstart), absolute(c2, r2))


def test_eight_shards_through_the_library(full):
    """north_star's partition: contiguous fsdb blocks on 8 ranks, tallies all-reduced before each consensus call"""
    f = full
    W = 8
    cuts = [N * k // W for k in range(W + 1)]
    parts = [upload(f.mod, f, cuts[k], cuts[k + 1], f.as0, f.ae0) for k in range(W)]
    grp = f.mod.LoopbackGroup(W)
    for k, h in enumerate(parts):
        grp.attach(h, k)
    for it, (ref, want_cons, want_al) in enumerate(((f.ref, f.cons1, f.al1), (f.cons1, f.cons2, f.al2)), 1):
        out, err = [None] * W, [None] * W

        def work(r):
            try:
                out[r] = parts[r].iterate(ref, True)
            except BaseException as e:    # noqa: BLE001
                err[r] = e
        th = [threading.Thread(target=work, args=(r,)) for r in range(W)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert err == [None] * W, (it, err)
        assert all(c == want_cons for c in out), it
        for k in range(3):
            assert np.array_equal(np.concatenate([h.alignments()[k] for h in parts]), want_al[k]), (it, k)
    for h in parts:                                  # every rank holds the reduced tallies of the whole job
        t, g = h.get_tally()
        assert np.array_equal(t, f.tally2) and np.array_equal(g, f.gaps2)
    for h in parts:
        h.comm_destroy()
    grp.close()
    for h in parts:
        h.close()


def test_fixed_point(full):
    f = full
    ref, cons, rounds = f.cons1, f.cons2, 0
    while cons != ref and rounds < 8:
        ref = cons
        cons = f.hip.iterate(ref, True)
        rounds += 1
    assert cons == ref, rounds
    assert abs(len(cons) - len(f.ref)) < 50
    if N == 10_000_000:
        # the digests bench.py prints for configs3 (bench.certificate; tests/test_gpu_bench_workloads.py): same generator, same seed, same size
        import json
        import bench
        want = json.load(open(os.path.join(GOLDEN, "bench_certificates.json")))["cfg3"]
        got = bench.certificate(f.hip, cons, True)
        assert {k: got[k] for k in ("consensus_sha256", "alignments_sha256", "consensus_len")} == {k: want[k] for k in ("consensus_sha256", "alignments_sha256", "consensus_len")}
