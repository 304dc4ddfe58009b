"""whole runs of 2 000 seeded 100 bp reads against mt311 -- flat matrix, matrices/ancient.submat.txt and
matrices/ancient.submat.solexa.pe.txt (aDNA-damaged reads), no k-mer filter, iterated to convergence -- and, as g2_c4,
BASELINE.json configs[4]'s shape: 2 000 damaged 150 bp reads against a 100 kb linear region (seed 5), ancient matrix,
-k 12, no -c (src/mia_main.c:645: no wrap; src/kmer.c:239-331 over 100 kb; windows of 250 columns, src/mia_main.c:190-217).  mia_hip must write
the .maln files the reference's own mia wrote (tools/make_goldens.py g2, oracle/_ref/mia), byte for byte from line 2:
each is pinned by the sha256 of that text (tests/golden/g2_runs.json); the reads are regenerated from their seed."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_goldens  # noqa: E402

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "mapping-iterative-assembler_amd", "mia_hip")


def runs():
    with open(os.path.join(GOLDEN, "g2_runs.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(runs().keys()))
def test_g2_whole_run_identical(name, tmp_path):
    want = runs()[name]
    fa = str(tmp_path / (name + ".fa"))
    assert make_goldens.g2_reads(name, fa) == want["reads_sha256"]          # the same reads the reference saw
    root = str(tmp_path / name)
    env = dict(os.environ, MIA_DATA_PATH=GOLDEN)
    ref_arg, _ = make_goldens.g2_ref(name, fa + ".ref.fa")
    subprocess.run([CLI, "-r", ref_arg, "-f", fa] + want["args"] + ["-m", root], cwd=GOLDEN, check=True, stderr=subprocess.DEVNULL, env=env, timeout=900)
    assert len(want["maln_sha256"]) >= 2
    for it, (h, hs) in enumerate(zip(want["maln_sha256"], want["ref_seq"]), 1):
        body = open(f"{root}.{it}").readlines()[1:]
        seq = next(l for l in body if l.startswith("SEQ "))
        assert hashlib.sha256(seq.encode()).hexdigest()[:16] == hs, (name, it, "the reference sequence of this iteration differs")
        assert hashlib.sha256("".join(body).encode()).hexdigest() == h, (name, it)
    assert not os.path.exists(f"{root}.{len(want['maln_sha256']) + 1}")
