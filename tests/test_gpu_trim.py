"""GPU parity of mia_hip_trim (trim_frag, reference src/mia.c:1318-1368): the reference's own answers
(tests/golden/trim_vectors.txt, dumped from the real trim_frag by oracle/ref_dp_driver.c) and a larger random
batch against the oracle, including reads built to need the exact scalar path (a gap of 63+ on the best path)."""
import ctypes as C
import random

import numpy as np
import pytest

import oracle_ctypes as oc
from test_oracle_vs_golden import trim_vectors

pytestmark = pytest.mark.gpu

NEAND = "GTCAGACACGCAACAGGGGATAGGCAAGGCACACAGGGGATAGG"


def run_batch(hip, adapter, reads):
    offs = np.zeros(len(reads) + 1, np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    return hip.trim(adapter, bases, offs)


def test_trim_reference_vectors():
    import mia_amd
    hip = mia_amd.MiaHip(0)
    by_adapter = {}
    for ad, read, exp in trim_vectors():
        by_adapter.setdefault(ad, []).append((read, exp))
    n = 0
    for ad, items in by_adapter.items():
        trimmed, point = run_batch(hip, ad, [r for r, _ in items])
        for k, (read, exp) in enumerate(items):
            got = [int(trimmed[k]), int(point[k]) if trimmed[k] else -999]
            assert got == exp[:2], (ad, read, got, exp)
            n += 1
    assert n >= 400
    hip.close()


def test_trim_random_batch_vs_oracle(oracle):
    import mia_amd
    rnd = random.Random(4)
    reads = []
    for i in range(20000):
        body = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(20, 180)))
        kind = i % 6
        if kind == 0:
            r = body
        elif kind in (1, 2):
            r = body + NEAND[: rnd.randint(1, len(NEAND))]
        elif kind == 3:
            r = body + NEAND[: rnd.randint(8, len(NEAND))] + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 40)))
        elif kind == 4:       # adapter prefix, 63+ unrelated bases, rest of the adapter: the long-gap path
            k = rnd.randint(33, 40)      # bridging the gap beats a fresh start only when > 31 adapter bases precede it
            r = body[:30] + NEAND[:k] + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(63, 110))) + NEAND[k:]
        else:
            lst = list(body + NEAND)
            for _ in range(3):
                lst[rnd.randrange(len(lst))] = rnd.choice("ACGTN")
            r = "".join(lst)
        reads.append(r[:256])
    hip = mia_amd.MiaHip(0)
    import time
    t0 = time.perf_counter()
    trimmed, point = run_batch(hip, NEAND, reads)
    dt = time.perf_counter() - t0
    reruns = hip.trim_exact_reruns()
    print(f"trim: {len(reads) / dt:.0f} reads/s wall, {reruns} exact re-runs")
    hip.close()
    pick = list(range(0, len(reads), 7)) + [i for i in range(len(reads)) if i % 6 == 4][:600]
    n_t = 0
    for i in pick:
        tr, tp = C.c_int(), C.c_int()
        oracle.ora_trim(reads[i].encode(), len(reads[i]), NEAND.encode(), C.byref(tr), C.byref(tp), None)
        assert (int(trimmed[i]), int(point[i]) if trimmed[i] else 0) == (tr.value, tp.value if tr.value else 0), (i, reads[i])
        n_t += tr.value
    assert n_t > 500


def test_trim_long_gap_exact_path(oracle):
    """a 120-base user adapter split by 63+ unrelated read bases: bridging the gap is the best path, the byte trace
    saturates, and the read is re-run by the exact scalar kernel -- same answers as the oracle"""
    import mia_amd
    rnd = random.Random(8)
    adapter = "".join(rnd.choice("ACGT") for _ in range(120))
    reads = []
    for i in range(400):
        k = rnd.randint(36, 55)
        junk = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(63, 70)))
        head = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 10)))
        reads.append((head + adapter[:k] + junk + adapter[k:])[:256])
    hip = mia_amd.MiaHip(0)
    trimmed, point = run_batch(hip, adapter, reads)
    reruns = hip.trim_exact_reruns()
    hip.close()
    assert reruns > 100, reruns
    for i, r in enumerate(reads):
        tr, tp = C.c_int(), C.c_int()
        oracle.ora_trim(r.encode(), len(r), adapter.encode(), C.byref(tr), C.byref(tp), None)
        assert (int(trimmed[i]), int(point[i]) if trimmed[i] else 0) == (tr.value, tp.value if tr.value else 0), (i, r)
