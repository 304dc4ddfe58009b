"""Every alternative route through the product path that an environment switch still reaches must give the same iteration:
round 1's flat-matrix pipeline (diagonal filter + k_band_align, MIA_HIP_NO_BANDX=1), the one-lane band kernels
(MIA_HIP_NO_LANES=1), round 2's stream order (MIA_HIP_BX_SERIAL=1), and this round's host-side orders switched off one by one
(MIA_HIP_NO_SPEC: wait for the alignment's counters before the cull is queued; MIA_HIP_SPEC_TEST: always take the second
round; MIA_HIP_NO_PREP_FUSE: six launches instead of k_ref_prep; MIA_HIP_NO_SIDE_BUCKETS: counting sort behind the cull; MIA_HIP_BX_DEBUG=64: the values DP in its aged form instead of
the ageing-free coordinates of bxl_values_star; MIA_HIP_NO_PLAN_SPLIT: k_bx_plan in one launch).
Two calls of mia_hip_iterate on 200 000 reads (first against mt311, then against the consensus): scores, end points,
scripts, all tally words, ref->gaps and the consensus string of both iterations must be identical to the default build's
(reference loop body: /root/reference/src/mia_main.c:915-964)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

SWITCHES = ["MIA_HIP_NO_BANDX", "MIA_HIP_NO_LANES", "MIA_HIP_BX_SERIAL", "MIA_HIP_NO_SPEC", "MIA_HIP_SPEC_TEST", "MIA_HIP_NO_PREP_FUSE",
            "MIA_HIP_NO_SIDE_BUCKETS", "MIA_HIP_BX_DEBUG=64", "MIA_HIP_NO_PLAN_SPLIT", "MIA_HIP_NO_ZERO_COPY", "MIA_HIP_NO_EXT_EVENTS",
            "MIA_HIP_EVENT_DEVICE_SCOPE", "MIA_HIP_SPIN_WAIT=0", "MIA_HIP_CULL_SCAN", "MIA_HIP_TAIL_SCANS",
            # round 4: the plan's third launch (fine blocks, bandx_body.h: bx_fine_anchors) never / in every iteration
            "MIA_HIP_NO_FINE", "MIA_HIP_FINE=2",
            # ... and the early tally experiment (the plan's reads tallied beside the band DPs, k_rec_early / k_tally_fix) on
            "MIA_HIP_EARLY_TALLY",
            # the reads no one-read-per-lane route of k_tally_binned takes: inside it, one per wavefront (default: by extra workgroups of k_tally_reduce)
            "MIA_HIP_TALLY_INLINE",
            # position-specific matrices: the tally's buckets by column and strand, the rows of depth code 15 through the vertical counters
            # (an experiment; default: by column only); no vertical counters at all, either matrix
            # (round 5: split by strand, sorted by start, is the default -- MIA_HIP_STRAND_SPLIT=0 is round 4's tally; MIA_HIP_NO_TALLY_RUNS=1
            # keeps the split and the sort but adds the rows at either end of a read one read at a time instead of one run at a time)
            "MIA_HIP_STRAND_SPLIT=0", "MIA_HIP_NO_TALLY_RUNS", "MIA_HIP_SORT2_UNPACKED", "MIA_HIP_DEBUG_SKIP=4096",
            # round 5: the band DPs in two rounds, the first beside the plan's second and third launch (measured, no gain: off by default)
            "MIA_HIP_SPLIT_DP=1",
            # round 5: the tally records written by k_rec_params in every iteration (default: by k_cull_records, k_rec_params only where a link exists)
            "MIA_HIP_NO_CULL_RECORDS",
            # round 4, second half: every wavefront at priority 0 (default: the step's chain ahead of k_bxl_trace); smaller persistent grids
            "MIA_HIP_BX_DEBUG=128", "MIA_HIP_BX_VALUES_PCT=50", "MIA_HIP_BX_TRACE_PCT=44",
            # round 6: the planner's chain (count / scan / fill, quad kernels, window classes) instead of the plan's own open list
            "MIA_HIP_NO_DIRECT_OPEN",
            # round 6: the full plan (bx_anchors: every block looked up) for every read instead of the quick plan on the read's old diagonal first
            # (... and the quick plan in front of the fine blocks' three launches below two million reads too, where the size rule leaves it out)
            "MIA_HIP_NO_QUICK_PLAN", "MIA_HIP_QUICK_PLAN=2"]


def two_iterations(mod, w, env):
    name, _, val = (env or "").partition("=")
    if env:
        os.environ[name] = val or "1"
    try:
        hip = mod.MiaHip(0)
    finally:
        if env:
            os.environ.pop(name, None)
    hip.set_pssm(w["pssm"])
    n = w["n"]
    hip.upload_reads(w["stored"].reshape(-1), w["offsets"], w["rc"], np.ones(n, np.uint8), w["as_"], w["ae"])
    out = []
    ref = w["ref"]
    for _ in range(2):
        cons = hip.iterate(ref, w["circular"])
        sc, a, e = hip.alignments()
        cols, rstart = hip.scripts()
        t, g = hip.get_tally()
        out.append((cons, sc.copy(), a.copy(), e.copy(), np.where(cols >= 0, cols.astype(np.int32) + rstart[:, None], cols.astype(np.int32)), t.copy(), g.copy()))
        ref = cons
    hip.close()
    return out


@pytest.mark.parametrize("config", [1, 2], ids=["flat", "ancient"])
def test_every_switch_gives_the_same_iterations(config):
    import bench
    import mia_amd
    w = bench.make_workload(config, 200_000, 3)
    base = two_iterations(mia_amd, w, None)
    assert len(base[0][0]) > 16000 and base[0][0] != w["ref"]
    for env in SWITCHES:
        got = two_iterations(mia_amd, w, env)
        for it in range(2):
            for k, name in enumerate(("consensus", "score", "as", "ae", "script", "tally", "gaps")):
                x, y = base[it][k], got[it][k]
                assert (x == y) if isinstance(x, str) else np.array_equal(x, y), (env, it, name)


def test_rall_tally_on_reads_with_n_and_shared_starts(oracle):
    """the position-specific tally's instance for reads of 129-256 bases (k_tally_binned<false, *, true>: EVERY row through
    the runs of equal starts) had only the campaigns' coverage.  60 000 reads of 150 bases against a 3 kb circular reference (twenty
    reads per start and strand: long runs; a twentieth of the reads runs over the origin), a fifth of them carrying N (the planes_ok ==
    false route), both strands, ancient matrix: three iterations side by side with the ORACLE (every tally word of every column), then
    the routing switches against the default, word for word."""
    import gen_data
    import mia_amd
    from oracle_sample import check_subset_iterations
    rng = np.random.default_rng(150)
    ref = gen_data.random_reference(3000, seed=150)
    n = 60_000
    d = gen_data.make_reads(ref, n, 150, seed=151, circular=True, damage=True)
    stored = gen_data.stored_orientation(d)
    with_n = rng.random(n) < 0.2
    for i in np.nonzero(with_n)[0]:
        stored[i, rng.integers(0, 150, int(rng.integers(1, 4)))] = ord("N")
    rc = d["strand"].astype(np.uint8)
    as0 = d["start"].astype(np.int32)
    ae0 = (as0 + 149).astype(np.int32)
    pssm = mia_amd.read_pssm(os.path.join(GOLDEN, "ancient.submat.txt"))
    done, _ = check_subset_iterations(mia_amd, oracle, ref, True, "ancient.submat.txt", pssm, stored, rc, np.ones(n, np.uint8), as0, ae0, iters=3)
    assert done >= 1            # (the reference is the reads' own genome: the consensus may repeat it at once)
    w = {"pssm": pssm, "n": n, "stored": stored, "offsets": np.arange(n + 1, dtype=np.int64) * 150, "rc": rc, "as_": as0, "ae": ae0, "ref": ref, "circular": True}
    base = two_iterations(mia_amd, w, None)
    for env in ("MIA_HIP_NO_TALLY_RALL", "MIA_HIP_NO_TALLY_RUNS", "MIA_HIP_STRAND_SPLIT=0", "MIA_HIP_NO_BINNED_TALLY", "MIA_HIP_SORT2_UNPACKED"):
        got = two_iterations(mia_amd, w, env)
        for it in range(2):
            for k, name in enumerate(("consensus", "score", "as", "ae", "script", "tally", "gaps")):
                x, y = base[it][k], got[it][k]
                assert (x == y) if isinstance(x, str) else np.array_equal(x, y), (env, it, name)


@pytest.mark.parametrize("config", [1, 2], ids=["flat", "ancient"])
def test_direct_open_list(config, oracle):
    """Round 6: where the plan gives up on FEW reads it lists them itself and k_align_open takes them one per wavefront (align_all:
    direct_open) -- no planner, no quad kernels.  That is every steady-state iteration, but never the first two of a run from mt311
    (ambiguity codes, then the first iteration's reject count), which is all most tests run.  Here the run starts from a reference of
    plain bases, so iteration 1 already takes the new route, with everything the planner used to sort out on the list: 1 500 reads of
    100 bases with 20 % substitutions (the plan gives up: windows of class 0), 300 reads of 230 bases with as many (windows of 330
    columns: class 1).  Against the oracle (every read, every tally word, three iterations), and -- with 200 reads of unknown strand added
    (ST_SKIPPED, never re-aligned) -- against the planner's route (MIA_HIP_NO_DIRECT_OPEN), word for word."""
    import bench
    import gen_data
    import mia_amd
    from oracle_sample import check_subset_iterations
    w = bench.make_workload(config, 1000, 7)                  # (for the reference, the matrix and its file name)
    rng = np.random.default_rng(66)
    ref = w["plain_ref"]
    # a LINEAR run (no -c): nothing is split at the origin, so no read can be split in one iteration and whole in the next -- the corner the
    # library refuses by name (MIA_HIP_ERR_RANGE) and that is not what this test is about
    n0, long_n, long_len = 60_000, 300, 230
    d0 = gen_data.make_reads(ref, n0, 100, seed=68, circular=False, damage=config != 1)
    d = gen_data.make_reads(ref, long_n, long_len, seed=67, circular=False, damage=config != 1)
    n = n0 + long_n
    stored = np.full((n, long_len), ord("A"), np.uint8)
    stored[:n0, :100] = gen_data.stored_orientation(d0)
    stored[n0:] = gen_data.stored_orientation(d)
    lens = np.concatenate([np.full(n0, 100, np.int32), np.full(long_n, long_len, np.int32)])
    rc = np.concatenate([d0["strand"].astype(np.uint8), d["strand"].astype(np.uint8)])
    as0 = np.concatenate([d0["start"].astype(np.int32), d["start"].astype(np.int32)])
    ae0 = (as0 + lens - 1).astype(np.int32)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for i in np.concatenate([rng.choice(n0, 1500, replace=False), np.arange(n0, n)]):
        k = int(lens[i]) // 5
        stored[i, rng.choice(int(lens[i]), k, replace=False)] = acgt[rng.integers(0, 4, k)]
    done, _ = check_subset_iterations(mia_amd, oracle, ref, False, w["matrix_file"], w["pssm"], stored, rc, np.ones(n, np.uint8), as0, ae0, iters=3, lens=lens)
    assert done >= 1            # (the reference IS the reads' genome: the consensus may repeat it at once)
    # (strand-unknown reads only in the comparison of the two routes: the oracle's pushed read store has no pass-1 records for them to point
    # at -- tests/test_gpu_iteration.py::fixture_lin is such a read from pass 1 on, against the oracle)
    sk = np.ones(n, np.uint8)
    sk[rng.choice(n0, 200, replace=False)] = 0
    offsets = np.zeros(n + 1, np.int64)
    offsets[1:] = np.cumsum(lens)
    flat = np.concatenate([stored[:n0, :100].reshape(-1), stored[n0:].reshape(-1)])

    def run(env):
        name, _, val = (env or "").partition("=")
        if env:
            os.environ[name] = val or "1"
        try:
            hip = mia_amd.MiaHip(0)
        finally:
            if env:
                os.environ.pop(name, None)
        hip.set_pssm(w["pssm"])
        hip.upload_reads(flat, offsets, rc, sk, as0, ae0)
        out, cur, fates = [], ref, []
        for _ in range(3):
            cons = hip.iterate(cur, False)
            sc, a, e = hip.alignments()
            cols, rstart = hip.scripts()
            t, g = hip.get_tally()
            out.append((cons, sc.copy(), a.copy(), e.copy(), np.where(cols >= 0, cols.astype(np.int32) + rstart[:, None], cols.astype(np.int32)), t.copy(), g.copy()))
            cur = cons
            fates.append(hip.bx_counters())
        hip.close()
        return out, fates

    base, fate = run(None)
    other, _ = run("MIA_HIP_NO_DIRECT_OPEN")
    known = sk.astype(bool)
    for it in range(3):
        for k, name in enumerate(("consensus", "score", "as", "ae", "script", "tally", "gaps")):
            x, y = base[it][k], other[it][k]
            if name == "script":            # (a strand-unknown read is never aligned: its script is whatever the buffer held)
                x, y = x[known], y[known]
            assert (x == y) if isinstance(x, str) else np.array_equal(x, y), (it, name)
    assert max(f[29] for f in fate) >= 1500, [f[29] for f in fate]      # BXC_OPEN: the list was in use (fewer than a twentieth of the reads: more, and the planner's quad kernels take them)


def test_quick_plan_with_eight_stretches_per_workgroup():
    """From four million reads the quick plan's workgroups take eight stretches of 256 reads (BxDev::qch; their per-read arrays sized in the
    launch's dynamic LDS): 4 M flat reads, two iterations, against the full plan for every read (MIA_HIP_NO_QUICK_PLAN) -- consensus, every
    read's score / start / end, tally and gaps."""
    import bench
    import mia_amd
    w = bench.make_workload(1, 4_000_000, 7)

    def run(env):
        if env:
            os.environ[env] = "1"
        try:
            hip = mia_amd.MiaHip(0)
        finally:
            if env:
                os.environ.pop(env, None)
        hip.set_pssm(w["pssm"])
        hip.upload_reads(w["stored"].reshape(-1), w["offsets"], w["rc"], np.ones(w["n"], np.uint8), w["as_"], w["ae"])
        out, ref = [], w["ref"]
        for _ in range(2):
            cons = hip.iterate(ref, w["circular"])
            sc, a, e = hip.alignments()
            t, g = hip.get_tally()
            out.append((cons, sc.copy(), a.copy(), e.copy(), t.copy(), g.copy(), list(hip.bx_counters())))
            ref = cons
        hip.close()
        return out

    base, other = run(None), run("MIA_HIP_NO_QUICK_PLAN")
    assert base[1][6][10] > 0 and other[1][6][10] == 0        # (reads the quick plan left to the full plan: it ran / it did not)
    for it in range(2):
        for k, name in enumerate(("consensus", "score", "as", "ae", "tally", "gaps")):
            x, y = base[it][k], other[it][k]
            assert (x == y) if isinstance(x, str) else np.array_equal(x, y), (it, name)
