"""CPU-side checks of the C ABI: the library loads, exports every symbol that
include/mia_hip.h declares, refuses to run without a GPU (no CPU fallback), and its
host-only helper (score-cut regression) follows the reference arithmetic."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def mia():
    import mia_amd
    if not os.path.exists(mia_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return mia_amd


def test_every_declared_symbol_is_exported(mia):
    hdr = open(os.path.join(ROOT, "include", "mia_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(mia_hip_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = mia.lib()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(mia.exported_symbols()) == declared


def test_release_library_reads_five_environment_variables(mia):
    """the wrong-results switches (MIA_HIP_DEBUG_SKIP ...) and the alternative routes are not in the release
    library at all -- its text holds only the documented names; the alt build (-DMIA_HIP_ALT_PATHS) holds them and exports
    the same ABI."""
    def names(path):
        with open(path, "rb") as f:
            return set(m.decode() for m in re.findall(rb"MIA_HIP_[A-Z0-9_]+", f.read()))
    rel = names(mia.LIB_PATH) - {"MIA_HIP_OP_MAX", "MIA_HIP_OP_SUM"}
    assert rel <= {"MIA_HIP_SPIN_WAIT", "MIA_HIP_LOOPBACK_TIMEOUT", "MIA_HIP_THREADS", "MIA_HIP_TIMING"}, rel
    assert os.path.exists(mia.ALT_LIB_PATH)
    alt = names(mia.ALT_LIB_PATH)
    assert {"MIA_HIP_DEBUG_SKIP", "MIA_HIP_NO_LANES", "MIA_HIP_NO_DIAG_FILTER", "MIA_HIP_NO_FINE"} <= alt
    hdr = open(os.path.join(ROOT, "include", "mia_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(mia_hip_[a-z_0-9]+)\s*\(", hdr)))
    lib = mia.alt_lib()
    assert not [s for s in declared if not hasattr(lib, s)]


def test_no_cpu_fallback(mia):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(mia.MiaHipError):
        mia.MiaHip(0)


def _score_cut_py(score, length):
    """find_fsdb_score_cut (reference src/fsdb.c:269-383) in sequential float64."""
    used = [i for i in range(len(score)) if score[i] >= 2000]
    j = len(used)
    xbar = ybar = 0.0
    for i in used:
        xbar += length[i]; ybar += score[i]
    xbar = xbar / j if j else float("nan")
    ybar = ybar / j if j else float("nan")
    ssxy = ssxx = 0.0
    for i in used:
        ssxy += (length[i] - xbar) * (score[i] - ybar)
        ssxx += (length[i] - xbar) * (length[i] - xbar)
    slope_bf = ssxy / ssxx if ssxx != 0 else (float("nan") if ssxy == 0 or math.isnan(ssxy) else math.copysign(float("inf"), ssxy))
    icpt = ybar - slope_bf * xbar
    md = 0.0
    for i in used:
        d = (score[i] - ((slope_bf * length[i]) + icpt)) / length[i]
        if d > md:
            md = d
    slope = slope_bf - md * 2.0 if (slope_bf - md) > 0 else slope_bf * (80 / 100.0)
    return slope, icpt


def test_score_cut_matches_reference_arithmetic(mia):
    rng = np.random.default_rng(3)
    lib = mia.lib()

    def call(score, length):
        s, i = C.c_double(), C.c_double()
        sc = np.ascontiguousarray(score, np.int32); ln = np.ascontiguousarray(length, np.int32)
        lib.mia_hip_score_cut(sc.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p), None, len(sc), C.byref(s), C.byref(i))
        return s.value, i.value

    for n in (5, 200, 5000):
        length = rng.integers(30, 200, size=n)
        score = (length * rng.integers(120, 200, size=n) + rng.integers(-3000, 500, size=n)).astype(np.int64)
        got = call(score, length)
        exp = _score_cut_py(score.tolist(), length.tolist())
        assert got == exp, (n, got, exp)
    # all reads the same length: 0/0 -> NaN slope and intercept, nothing is ever dropped
    s, i = call(np.full(100, 19000), np.full(100, 100))
    assert math.isnan(s) and math.isnan(i)


def test_pssm_helpers_match_oracle(mia, oracle):
    import oracle_ctypes as oc
    p = oc.Pssm(); q = oc.Pssm()
    oracle.ora_pssm_flat(C.byref(p)); oracle.ora_pssm_revcom(C.byref(p), C.byref(q))
    assert np.array_equal(mia.flat_pssm(), np.ctypeslib.as_array(p.sm).reshape(31, 5, 5))
    assert np.array_equal(mia.revcom_pssm(mia.flat_pssm()), np.ctypeslib.as_array(q.sm).reshape(31, 5, 5))
    from conftest import GOLDEN
    path = os.path.join(GOLDEN, "ancient.submat.txt")
    assert oracle.ora_pssm_read(path.encode(), C.byref(p)) == 1
    oracle.ora_pssm_revcom(C.byref(p), C.byref(q))
    a = mia.read_pssm(path)
    assert np.array_equal(a, np.ctypeslib.as_array(p.sm).reshape(31, 5, 5))
    assert np.array_equal(mia.revcom_pssm(a), np.ctypeslib.as_array(q.sm).reshape(31, 5, 5))
