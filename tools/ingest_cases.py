"""Deliberately awkward FASTA / FASTQ inputs for the fragment reader, regenerated from a seed.
tools/make_goldens.py ingest runs the REFERENCE's reader (oracle/_ref/ref_read_driver: find_input_type + read_next_seq,
/root/reference/src/io.c:11-281) over each of them and commits the sha256 of what it read (tests/golden/ingest.json);
tests/test_ingest_cpu.py holds the product's multi-threaded reader (host/ingest.h) against that."""
import random


def _seq(rnd, n, alphabet="ACGT"):
    return "".join(rnd.choice(alphabet) for _ in range(n))


def cases():
    out = {}
    rnd = random.Random(20261004)
    # 1. the usual: one line per sequence, some descriptions
    t = []
    for i in range(1500):
        t.append(">r%d%s\n%s\n" % (i, "" if i % 3 else " sample=%d lane 7" % i, _seq(rnd, 100)))
    out["fa_plain"] = "".join(t)
    # 2. folded lines, lower case, CRLF, blank lines, tabs in the header, no newline at the end
    t = []
    for i in range(700):
        s = _seq(rnd, rnd.randint(1, 250), "ACGTacgtNn")
        lines = [s[k:k + 60] for k in range(0, len(s), 60)]
        eol = "\r\n" if i % 4 == 1 else "\n"
        t.append(">read_%d\tdesc with\ttabs%s" % (i, eol) + eol.join(lines) + eol + ("\n" if i % 7 == 0 else ""))
    out["fa_folded"] = "".join(t).rstrip("\n")
    # 3. over-long reads (cut at 256, the rest skipped), empty records
    t = []
    for i in range(400):
        n = rnd.choice([0, 1, 255, 256, 257, 300, 1000, 100])
        s = _seq(rnd, n)
        t.append(">long%d\n%s\n" % (i, "\n".join(s[k:k + 70] for k in range(0, len(s), 70))))
    out["fa_long"] = "".join(t)
    # 4. what nobody writes: '>' inside a sequence line and inside a description, over-long ids and descriptions (the tail of
    #    the header line is then read as sequence), a bare '>', white space before the description
    t = []
    for i in range(600):
        k = i % 12
        s = _seq(rnd, 80)
        if k == 0:
            t.append(">x%d\n%s>y%d_in_line\n%s\n" % (i, s[:40], i, s[40:]))
        elif k == 1:
            t.append(">x%d has > in its description\n%s\n" % (i, s))
        elif k == 2:
            t.append(">%s\n%s\n" % ("I" * 150 + str(i), s))
        elif k == 3:
            t.append(">x%d %s ACGTACGT\n%s\n" % (i, "d" * 300, s))
        elif k == 4:
            t.append(">\n%s\n" % s)
        elif k == 5:
            t.append(">x%d   \t  spaced out\n%s\n" % (i, s))
        elif k == 6:
            t.append(">x%d %s>z%d\n%s\n" % (i, "e" * 200, i, s))
        elif k == 7:
            t.append(">%s %s\n%s\n" % ("J" * 100, "f" * 128, s))
        else:
            t.append(">x%d\n%s\n" % (i, s))
    out["fa_odd"] = "".join(t)
    # 5. FASTQ: quality lines that begin with '@' or '+', over-long reads, CRLF, lower case
    t = []
    for i in range(1200):
        n = rnd.choice([36, 100, 100, 100, 256, 300])
        s = _seq(rnd, n, "ACGTacgtN")
        q = "".join(rnd.choice("@+IIIIFFF#5:") for _ in range(n))
        if i % 5 == 0:
            q = "@" + q[1:]
        if i % 11 == 0:
            q = "+" + q[1:]
        eol = "\r\n" if i % 9 == 4 else "\n"
        t.append("@q%d%s%s%s%s+%s%s%s%s" % (i, " 1:N:0" if i % 2 else "", eol, s, eol, "q%d" % i if i % 6 == 0 else "", eol, q, eol))
    out["fq_plain"] = "".join(t)
    # 6. FASTQ that ends early: a record whose quality line is shorter than its sequence (the reference stops there)
    t = []
    for i in range(900):
        s = _seq(rnd, 100)
        q = "I" * (100 if i != 613 else 97)
        t.append("@e%d\n%s\n+\n%s\n" % (i, s, q))
    out["fq_bad_qual"] = "".join(t)
    # 7. FASTQ with a record that lacks its '+' line, and one that does not begin with '@'
    t = []
    for i in range(700):
        s = _seq(rnd, 90)
        if i == 211:
            t.append("@m%d\n%s\n%s\n" % (i, s, "I" * 90))
        elif i == 555:
            t.append("m%d\n%s\n+\n%s\n" % (i, s, "I" * 90))
        else:
            t.append("@m%d\n%s\n+\n%s\n" % (i, s, "I" * 90))
    out["fq_missing_plus"] = "".join(t)
    return out
