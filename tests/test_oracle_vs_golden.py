"""Pin the oracle restatement (oracle/*.c) to the REAL reference.

Every expected value under tests/golden/ was produced by the reference itself
(oracle/_ref/*, built by oracle/Makefile.ref, driven by tools/make_goldens.py).
CPU only."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import pytest

from conftest import GOLDEN
import oracle_ctypes as oc


def fnv(vals):
    h = 1469598103934665603
    for v in vals:
        u = v & 0xFFFFFFFF
        for i in range(4):
            h ^= (u >> (8 * i)) & 0xFF
            h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _pssm(lib, spec, rc, cache={}):
    key = (spec, rc)
    if key not in cache:
        p = oc.Pssm()
        if spec == "flat":
            lib.ora_pssm_flat(C.byref(p))
        else:
            assert lib.ora_pssm_read(os.path.join(GOLDEN, spec).encode(), C.byref(p)) == 1
        if rc:
            q = oc.Pssm()
            lib.ora_pssm_revcom(C.byref(p), C.byref(q))
            p = q
        cache[key] = p
    return cache[key]


def dp_cases(files=("dp_vectors.txt", "dp_vectors_wide.txt")):
    out = []
    for name in files:
        with open(os.path.join(GOLDEN, name)) as f:
            lines = [l.rstrip("\n") for l in f if not l.startswith("#")]
        out += list(zip(lines[0::2], lines[1::2]))
    return out


def test_dp_vectors(oracle):
    """dyn_prog / max_sg_score / find_align_begin / populate_pwaln_to_begin:
    full S and T matrices (hashed), end points and both gapped strings."""
    cases = dp_cases()
    assert len(cases) >= 400
    for inp, exp in cases:
        _, spec, rc, sg5, _sg3, s1, s2, mask = inp.split(" ")
        e = exp.split(" ")
        n1, n2 = len(s1), len(s2)
        S = (C.c_int * (n1 * n2))()
        T = (C.c_int * (n1 * n2))()
        res = oc.Aln()
        rg = C.create_string_buffer(520)
        fg = C.create_string_buffer(520)
        m = None if mask == "*" else bytes(1 if ch == "1" else 0 for ch in mask)
        oracle.ora_align(s1.encode(), n1, s2.encode(), n2, m, C.byref(_pssm(oracle, spec, int(rc))), int(sg5),
                         C.byref(res), rg, fg, S, T)
        got = [res.best, res.aec, res.aer, res.abc, res.abr]
        assert got == [int(x) for x in e[1:6]], inp[:80]
        assert "%016x" % fnv(S) == e[6], inp[:80]
        assert "%016x" % fnv(T) == e[7], inp[:80]
        assert rg.value.decode() == e[8] and fg.value.decode() == e[9], inp[:80]


def test_consensus_vectors(oracle):
    with open(os.path.join(GOLDEN, "cons_vectors.txt")) as f:
        lines = [l.rstrip("\n") for l in f]
    for inp, exp in zip(lines[0::2], lines[1::2]):
        v = [int(x) for x in inp.split()[1:]]
        bc = oc.Counts(As=v[1], Cs=v[2], Gs=v[3], Ts=v[4], gaps=v[5], cov=v[6], scoreA=v[7], scoreC=v[8],
                       scoreG=v[9], scoreT=v[10])
        got = oracle.ora_find_consensus(C.byref(bc), v[0]).decode()
        assert got == exp.split(" ")[1], inp


def test_myers_vectors(oracle):
    with open(os.path.join(GOLDEN, "myers_vectors.txt")) as f:
        lines = [l.rstrip("\n") for l in f]
    n = 0
    for inp, exp in zip(lines[0::2], lines[1::2]):
        mode, maxd, a, b = inp.split(" ")
        d_exp, bt_exp = exp.split(" ")
        bt = C.create_string_buffer(len(a) + len(b) + int(maxd) + 8)
        d = oracle.ora_myers_diff(a.encode(), int(mode), b.encode(), int(maxd), bt)
        assert d == int(d_exp), inp[:60]
        if d != 0xFFFFFFFF:
            assert bt.value.decode() == bt_exp, inp[:60]
        n += 1
    assert n >= 200


def maln_cases():
    with open(os.path.join(GOLDEN, "maln", "cases.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(maln_cases().keys()))
def test_whole_run_maln(name, oracle_build, tmp_path):
    """Run the oracle's mia front end on the committed inputs; every .maln it
    writes must equal the reference's byte for byte (line 1 = timestamp excluded)."""
    if name == "s150_full" and os.environ.get("MIA_SLOW", "0") != "1":
        pytest.skip("unseeded pass 1 over 150 reads takes ~25 s on the CPU oracle (set MIA_SLOW=1)")
    args = maln_cases()[name]
    root = str(tmp_path / name)
    subprocess.run([os.path.join(oracle_build, "ora_mia")] + args + ["-m", root], cwd=GOLDEN, check=True,
                   stderr=subprocess.DEVNULL)
    with open(os.path.join(GOLDEN, "maln", "hashes.json")) as f:
        hashes = json.load(f)          # iterations beyond the fourth are pinned by sha256
    it = 1
    while os.path.exists(os.path.join(GOLDEN, "maln", f"{name}.{it}")) or f"{name}.{it}" in hashes:
        with open(f"{root}.{it}") as f:
            got = "".join(f.readlines()[1:])
        if f"{name}.{it}" in hashes:
            assert hashlib.sha256(got.encode()).hexdigest() == hashes[f"{name}.{it}"], f"{name}.{it}"
        else:
            with open(os.path.join(GOLDEN, "maln", f"{name}.{it}")) as f:
                assert got == f.read(), f"{name}.{it}"
        it += 1
    assert it > 1
    assert not os.path.exists(f"{root}.{it}")


def trim_vectors():
    lines = open(os.path.join(GOLDEN, "trim_vectors.txt")).read().splitlines()
    out = []
    for k in range(0, len(lines), 2):
        ad, read = lines[k].split()
        f = lines[k + 1].split()
        assert f[0] == "T"
        out.append((ad, read, [int(x) for x in f[1:]]))   # trimmed trim_point aec aer abc abr
    return out


def test_trim_vectors(oracle):
    """ora_trim against trim_frag of the real reference (oracle/ref_dp_driver.c, `T` lines)"""
    vecs = trim_vectors()
    assert len(vecs) >= 400 and sum(v[2][0] for v in vecs) > 100 and sum(1 - v[2][0] for v in vecs) > 30
    for ad, read, exp in vecs:
        tr, tp, a = C.c_int(), C.c_int(), oc.Aln()
        oracle.ora_trim(read.encode(), len(read), ad.encode(), C.byref(tr), C.byref(tp), C.byref(a))
        got = [tr.value, tp.value if tr.value else -999, a.aec, a.aer, a.abc, a.abr]
        assert got == exp, (ad, read, got, exp)


# ---- the per-iteration loop on a read store filled with post-pass-1 fields (ora_push_frag), alignments on host threads ----
def _iter_push_sets():
    import json
    with open(os.path.join(GOLDEN, "iter_push.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(_iter_push_sets().keys()))
def test_pushed_reads_iterate_like_the_reference(oracle, name):
    """tests/golden/iter_push.json: what the reference's own loop (oracle/ref_iter_driver.c around reiterate_assembly,
    pop_smp_from_FSDB, cull_maln_from_fsdb, sort_aln_frags, consensus_assembly_string -- src/mia_main.c:931-963) gives for
    3 000 seeded reads whose strand and coordinates are set directly.  The oracle must give the same per-read
    (score, as, ae) and the same consensus in every iteration -- through ora_push_frag and with the alignments spread over
    four host threads, the way the full-size GPU tests drive it."""
    import hashlib
    import ctypes as C
    import numpy as np
    import make_goldens
    import oracle_ctypes as oc
    want = _iter_push_sets()[name]
    ref, stored, rc, as_, ae = make_goldens.iter_push_inputs(name)
    assert hashlib.sha256(stored.tobytes() + rc.tobytes() + as_.tobytes()).hexdigest() == want["inputs_sha256"]
    o = oc.Opts()
    oracle.ora_opts_default(C.byref(o))
    o.circular = 1
    anc = oc.Pssm()
    if want["matrix"]:
        assert oracle.ora_pssm_read(os.path.join(GOLDEN, want["matrix"]).encode(), C.byref(anc)) == 1
    else:
        oracle.ora_pssm_flat(C.byref(anc))
    st = oracle.ora_new(C.byref(o), C.byref(anc))
    oracle.ora_set_ref(st, b"ref", b"", ref.encode())
    oracle.ora_prepare_ref(st)
    for i in range(len(rc)):
        oracle.ora_push_frag(st, b"r%d" % i, stored[i].tobytes(), int(rc[i]), int(as_[i]), int(ae[i]), 2001, 1)
    oracle.ora_set_threads(st, 4)
    cons = ref
    for k, it in enumerate(want["iterations"], 1):
        oracle.ora_iterate(st, cons.encode(), k)
        cons = oc.consensus_string(oracle, st)
        got = np.array([[f.score, f.as_, f.ae] for f in (oracle.ora_frag_at(st, i).contents for i in range(len(rc)))], dtype=np.int32)
        assert got[:8].tolist() == it["first_reads"], (name, k)
        assert hashlib.sha256(got.tobytes()).hexdigest() == it["reads_sha256"], (name, k)
        assert len(cons) == it["cons_len"] and hashlib.sha256(cons.encode()).hexdigest() == it["cons_sha256"], (name, k)
    oracle.ora_free(st)
