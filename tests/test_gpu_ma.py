"""ma_hip (host/ma_main.cpp over mia_hip_ma_tally) must print what the reference's own `ma` prints for the
report formats computed from the column tallies: -f 5 (assembled FASTA), -f 41 / -f 4 (column table), both
consensus codes, on every committed .maln.  Goldens: tests/golden/ma, written by tools/make_goldens.py from
oracle/_ref/ma; outputs above 40 KB are pinned by sha256."""
import glob
import hashlib
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

MA = os.path.join(ROOT, "mapping-iterative-assembler_amd", "ma_hip")
HEADER = "/* map_alignment [V1.0] */ golden\n"
RUNS = [(5, 1), (5, 2), (41, 1), (41, 2), (4, 1)]


def malns():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "maln", "*.[0-9]")))


@pytest.fixture(scope="module")
def hashes():
    with open(os.path.join(GOLDEN, "ma", "hashes.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", malns())
def test_ma_reports_identical(name, hashes, tmp_path):
    full = str(tmp_path / name)
    with open(full, "w") as f:
        f.write(HEADER + open(os.path.join(GOLDEN, "maln", name)).read())
    checked = 0
    for fmt, code in RUNS:
        key = f"{name}.f{fmt}c{code}"
        out = subprocess.run([MA, "-M", full, "-f", str(fmt), "-c", str(code)], check=True, stdout=subprocess.PIPE, timeout=300).stdout
        path = os.path.join(GOLDEN, "ma", key)
        if os.path.exists(path):
            assert out == open(path, "rb").read(), key
        else:
            assert key in hashes, key
            assert (len(out), hashlib.sha256(out).hexdigest()) == (hashes[key]["bytes"], hashes[key]["sha256"]), key
        checked += 1
    assert checked == len(RUNS)


def test_ma_assigned_id(tmp_path):
    name = "fix_c.2"
    full = str(tmp_path / name)
    with open(full, "w") as f:
        f.write(HEADER + open(os.path.join(GOLDEN, "maln", name)).read())
    out = subprocess.run([MA, "-M", full, "-f", "5", "-I", "my_assembly"], check=True, stdout=subprocess.PIPE, timeout=300).stdout
    assert out == open(os.path.join(GOLDEN, "ma", name + ".f5c1.I"), "rb").read()


def test_ma_rejects_other_formats(tmp_path):
    name = "fix_c.1"
    full = str(tmp_path / name)
    with open(full, "w") as f:
        f.write(HEADER + open(os.path.join(GOLDEN, "maln", name)).read())
    r = subprocess.run([MA, "-M", full, "-f", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"outside the MI355X-accelerated path" in r.stderr
