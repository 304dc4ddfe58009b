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
            "MIA_HIP_BX_DEBUG=128", "MIA_HIP_BX_VALUES_PCT=50", "MIA_HIP_BX_TRACE_PCT=44"]


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
