"""ccheck_hip (host/ccheck_main.cpp over mia_hip_myers_align + mia_hip_align_windows) must print what the reference's
own ccheck prints -- stdout, stderr at every verbosity, exit code -- on assemblies made by the reference's own mia.
Goldens: tests/golden/ccheck, written by tools/make_goldens.py from oracle/_ref/ccheck (inputs as .maln.gz, the
reports of every run of runs.json, streams above 64 KB pinned by sha256)."""
import gzip
import hashlib
import json
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

CC = os.path.join(ROOT, "mapping-iterative-assembler_amd", "ccheck_hip")
GC = os.path.join(GOLDEN, "ccheck")
HEADER = "/* map_alignment [V1.0] */ golden\n"


def runs():
    with open(os.path.join(GC, "runs.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("ccheck")
    for name in ("cc_flat", "cc_anc"):
        with gzip.open(os.path.join(GC, name + ".maln.gz"), "rb") as f, open(d / (name + ".maln"), "wb") as g:
            g.write(f.read())
    with open(d / "cc_small.maln", "w") as f:
        f.write(HEADER + open(os.path.join(GOLDEN, "maln", "s150_k12.2")).read())
    shutil.copy(os.path.join(GC, "cc_contam.fa"), d)
    shutil.copy(d / "cc_small.maln", d / "sib.1")
    shutil.copy(d / "cc_flat.maln", d / "sib.3")
    return d


@pytest.fixture(scope="module")
def hashes():
    with open(os.path.join(GC, "hashes.json")) as f:
        return json.load(f)


def compare(key, r, hashes):
    assert r.returncode == int(open(os.path.join(GC, key + ".rc")).read()), (key, r.stderr[-400:])
    for ext, data in (("out", r.stdout), ("err", r.stderr)):
        path = os.path.join(GC, f"{key}.{ext}")
        if os.path.exists(path + ".gz"):
            # -vvvvvv prints the two rows of the global alignment; the reference's assembly row has no terminator
            # (src/myers_align.c:44-45) and drags heap remains along: exactly one line may carry such a tail
            want = gzip.open(path + ".gz", "rb").read().split(b"\n")
            got = data.split(b"\n")
            assert len(got) == len(want), f"{key}.{ext}"
            odd = [i for i in range(len(got)) if got[i] != want[i]]
            assert len(odd) <= 1 and all(want[i].startswith(got[i]) for i in odd), (f"{key}.{ext}", odd[:5])
        elif os.path.exists(path):
            want = open(path, "rb").read()
            if data != want:
                a, b = data.split(b"\n"), want.split(b"\n")
                k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
                raise AssertionError(f"{key}.{ext}: first difference in line {k + 1}: got {a[k:k + 2]!r} want {b[k:k + 2]!r}")
        else:
            h = hashes[f"{key}.{ext}"]
            assert (len(data), hashlib.sha256(data).hexdigest()) == (h["bytes"], h["sha256"]), f"{key}.{ext}"


@pytest.mark.parametrize("key", sorted(runs()))
def test_ccheck_reports_identical(key, workdir, hashes):
    args, files = runs()[key]
    r = subprocess.run([CC, "-f"] + args + files, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    compare(key, r, hashes)


def test_ccheck_picks_newest_iteration(workdir, hashes):
    """without -f the highest-numbered sibling of the file named is read (find_maln, src/ccheck.cc:206-236)"""
    r = subprocess.run([CC, "sib.1"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    compare("find_maln", r, hashes)


def test_ccheck_against_reference_binary_when_present(workdir):
    """where the real ccheck was built (oracle/_ref, this container and the snapshot that travels to the GPU box):
    a run that is not among the goldens"""
    ref = os.path.join(ROOT, "oracle", "_ref", "ccheck")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/ccheck not built")
    args = ["-f", "-a", "-n", "3", "-s", "500-15000", "-vv", "cc_anc.maln"]
    want = subprocess.run([ref] + args, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    got = subprocess.run([CC] + args, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert (got.returncode, got.stdout, got.stderr) == (want.returncode, want.stdout, want.stderr)
