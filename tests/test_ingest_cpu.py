"""The product's fragment reader (mapping-iterative-assembler_amd/host/ingest.h: the reference's character-level state
machine run over the mapped file on all host threads, with validated starting points) against the REFERENCE's own reader
(find_input_type + read_next_seq, /root/reference/src/io.c:11-281): tests/golden/ingest.json holds the sha256 of the
records and of the messages oracle/_ref/ref_read_driver produced for the deliberately awkward inputs of
tools/ingest_cases.py -- folded and lower-case sequences, CRLF, over-long reads / ids / descriptions, '>' inside
sequence lines, FASTQ quality lines that begin with '@', inputs the reference gives up on half way.  Every input is read
with 1, 3 and 16 threads and with stretches as short as 2 KB, so that the guessed starting points land everywhere."""
import hashlib
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT
import ingest_cases


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ingest") / "ingest_dump")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "emul", "ingest_dump.cpp")], check=True)
    return exe


def golden():
    with open(os.path.join(GOLDEN, "ingest.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(golden().keys()))
def test_reader_equals_the_references(name, dump, tmp_path):
    want = golden()[name]
    text = ingest_cases.cases()[name].encode()
    assert hashlib.sha256(text).hexdigest() == want["input_sha256"]
    path = str(tmp_path / name)
    with open(path, "wb") as f:
        f.write(text)
    for threads, stretch in ((1, 1 << 20), (3, 20_000), (16, 2_000), (16, 300), (7, 64)):
        r = subprocess.run([dump, path, str(threads), str(stretch)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        recs = r.stdout.split(b"\n")[:-1]
        ctx = (name, threads, stretch)
        assert len(recs) == want["records"], ctx
        assert recs[0].decode("latin1") == want["first"] and recs[-1].decode("latin1") == want["last"], ctx
        assert hashlib.sha256(r.stdout).hexdigest() == want["records_sha256"], ctx
        assert len(r.stderr.split(b"\n")) - 1 == want["stderr_lines"], ctx
        assert hashlib.sha256(r.stderr).hexdigest() == want["stderr_sha256"], ctx


def test_large_input_in_parallel(dump, tmp_path):
    """200 000 reads (22 MB): every thread count gives the one-thread record list"""
    import random
    rnd = random.Random(5)
    chunk = "".join(">r%d\n%s\n" % (i, "".join(rnd.choice("ACGT") for _ in range(100))) for i in range(2000))
    path = str(tmp_path / "big.fa")
    with open(path, "w") as f:
        for k in range(100):
            f.write(chunk.replace(">r", ">b%d_" % k))
    base = subprocess.run([dump, path, "1"], stdout=subprocess.PIPE, check=True).stdout
    assert base.count(b"\n") == 200_000
    for threads in (2, 8, 21):
        assert subprocess.run([dump, path, str(threads)], stdout=subprocess.PIPE, check=True).stdout == base
