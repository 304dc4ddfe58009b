"""The mia_hip command line (host C++ over the C ABI, GPU kernels underneath) must write
the same .maln files as the reference's mia, byte for byte from line 2, on every
committed whole-run case (tests/golden/maln, produced by the real reference)."""
import hashlib
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "mapping-iterative-assembler_amd", "mia_hip")


def cases():
    with open(os.path.join(GOLDEN, "maln", "cases.json")) as f:
        return json.load(f)


SKIP = {}


@pytest.mark.parametrize("name", sorted(cases().keys()))
def test_cli_maln_identical(name, tmp_path):
    if name in SKIP:
        pytest.skip(SKIP[name])
    args = cases()[name]
    root = str(tmp_path / name)
    env = dict(os.environ, MIA_DATA_PATH=GOLDEN)
    subprocess.run([CLI] + args + ["-m", root], cwd=GOLDEN, check=True, stderr=subprocess.DEVNULL, env=env, timeout=600)
    with open(os.path.join(GOLDEN, "maln", "hashes.json")) as f:
        hashes = json.load(f)          # iterations beyond the fourth are pinned by sha256
    it = 1
    while os.path.exists(os.path.join(GOLDEN, "maln", f"{name}.{it}")) or f"{name}.{it}" in hashes:
        got = "".join(open(f"{root}.{it}").readlines()[1:])
        if f"{name}.{it}" in hashes:
            assert hashlib.sha256(got.encode()).hexdigest() == hashes[f"{name}.{it}"], f"{name}.{it}"
        else:
            assert got == open(os.path.join(GOLDEN, "maln", f"{name}.{it}")).read(), f"{name}.{it}"
        it += 1
    assert it > 1 and not os.path.exists(f"{root}.{it}")
