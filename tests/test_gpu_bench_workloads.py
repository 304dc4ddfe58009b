"""The bytes bench.py times, against the oracle.

bench.py's headline runs make_workload(1, 1 000 000, seed=1) -- and configs[2] make_workload(2, 1 000 000, seed=3) -- to the fixed
point and prints `certificate`: sha256 of the consensus and of every read's (score, as, ae).  Here the SAME workloads are built by
the same function and run the same way (bench.Pipeline.step = mia_hip_iterate), and

* 4 000 sampled reads go through every realignment of the run in the oracle, read by read (the plan finishes four reads in five
  without any DP on this data: this is the check of that shortcut on the benched bytes, not on another seed);
* 50 000 reads of the batch run three iterations side by side with the oracle in a context of their own: every read's result, the
  dropped marks, all ten tally words of every column, ref->gaps, the consensus;
* the digests equal tests/golden/bench_certificates.json -- what BENCH_r06.json's line must show as well.
configs[3] and configs[4] at their bench seeds and sizes: test_gpu_config3.py, test_gpu_config4_full.py (same file of digests)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle_sample import PushedOracle, check_subset_iterations

pytestmark = pytest.mark.gpu

PINNED = json.load(open(os.path.join(GOLDEN, "bench_certificates.json")))


def pinned(cert, key):
    want = PINNED[key]
    got = {k: cert[k] for k in ("consensus_sha256", "alignments_sha256", "consensus_len")}
    assert got == {k: want[k] for k in got}, (key, got, want)


@pytest.mark.parametrize("cfg,seed", [(1, 1), (2, 3)])
def test_benched_workload_against_oracle(cfg, seed, oracle):
    import bench
    import mia_amd
    n = 1_000_000
    w = bench.make_workload(cfg, n, seed)
    hip = mia_amd.MiaHip(0)
    pipe = bench.Pipeline(hip, w)
    refs, als, cur = [], [], w["ref"]
    for _ in range(12):
        nxt = pipe.step(cur)
        refs.append(cur)
        if len(als) < 3:
            als.append(hip.alignments())
        if nxt == cur:
            break
        cur = nxt
    assert nxt == cur and 2 <= len(refs) <= 6
    cert = bench.certificate(hip, nxt, True)
    hip.close()
    pinned(cert, "cfg%d" % cfg)
    assert PINNED["cfg%d" % cfg]["iterations_to_convergence"] == len(refs)
    # the sample, through the run's own sequence of references
    pick = np.sort(np.random.default_rng(100 + cfg).choice(n, 4000, replace=False))
    po = PushedOracle(oracle, w["ref"], True, w["matrix_file"], w["stored"][pick], w["rc"][pick], w["as_"][pick], w["ae"][pick])
    for it, (ref, al) in enumerate(zip(refs, als), 1):
        po.iterate(ref)
        o = po.alignments()
        for k in range(3):
            bad = np.nonzero(al[k][pick] != o[k])[0]
            assert len(bad) == 0, (cfg, it, k, len(bad), pick[bad[:5]].tolist())
    po.close()
    # a subset with everything the consensus is made of
    sub = np.sort(np.random.default_rng(200 + cfg).choice(n, 50_000, replace=False))
    first = tuple(a[sub] for a in als[0])
    done, _ = check_subset_iterations(mia_amd, oracle, w["ref"], True, w["matrix_file"], w["pssm"], w["stored"][sub], w["rc"][sub], np.ones(len(sub), np.uint8),
                                      w["as_"][sub], w["ae"][sub], iters=3, expect_first=first)
    assert done >= 2
