#!/usr/bin/env python3
"""Generate tests/golden/* by running the REAL reference (oracle/_ref/*, built
by oracle/Makefile.ref from /root/reference).  Runs only in the build
container; the outputs (data: inputs + expected outputs) are committed so that
the GPU box, which has no /root/reference, can replay them.

  tests/golden/mt311.fa              sequence data recovered from src/mt311.c
  tests/golden/tr1.fna, tf.fna       the reference's own test fixtures (data)
  tests/golden/ancient.submat.txt    the reference's PSSM table (data)
  tests/golden/dp_vectors.txt        DP unit vectors  (ref_dp_driver)
  tests/golden/cons_vectors.txt      find_consensus vectors (ref_dp_driver)
  tests/golden/myers_vectors.txt     Myers vectors    (ref_myers_driver)
  tests/golden/maln/<case>.<iter>    whole-run .maln files, line 1 (timestamp) removed
  tests/golden/maln/cases.json       the command line of each case
  tests/golden/ccheck/*              ccheck inputs (.maln.gz) and the reference's reports on them
"""
import json, os, random, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_data  # noqa: E402

REF = "/root/reference"
RB = os.path.join(ROOT, "oracle", "_ref")
G = os.path.join(ROOT, "tests", "golden")


def sh(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def dp_vectors():
    rnd = random.Random(20261003)
    anc = os.path.join(G, "ancient.submat.txt")
    lines = []

    def rseq(n, alphabet="ACGT"):
        return "".join(rnd.choice(alphabet) for _ in range(n))

    def mutate(s, p=0.1):
        out = []
        for ch in s:
            u = rnd.random()
            if u < p / 3:
                continue
            if u < 2 * p / 3:
                out.append(rnd.choice("ACGT"))
            if u < p:
                out.append(rnd.choice("ACGT"))
            else:
                out.append(ch)
        return "".join(out) or "A"

    for i in range(400):
        kind = i % 8
        len2 = rnd.randint(1, 70) if i % 5 else rnd.randint(90, 256)
        len1 = rnd.randint(max(2, len2 // 2), len2 + 140)
        refalpha = "ACGT" if kind < 5 else "ACGTNRYacgt"
        s1 = rseq(len1, refalpha)
        if kind in (0, 1, 2, 5):           # read derived from the window
            st = rnd.randint(0, max(0, len1 - len2))
            s2 = mutate(s1[st:st + len2].upper().replace("R", "A").replace("Y", "C"), 0.08 if kind != 2 else 0.3)[:256]
        elif kind == 3:                    # repeats -> ties
            unit = rseq(rnd.randint(1, 4))
            s1 = (unit * (len1 // len(unit) + 1))[:len1]
            s2 = (unit * (len2 // len(unit) + 1))[:len2]
        elif kind == 4:                    # homopolymers -> ties in gap placement
            s1 = "".join(rnd.choice("AC") * rnd.randint(1, 6) for _ in range(len1))[:len1]
            st = rnd.randint(0, max(0, len1 - len2))
            s2 = mutate(s1[st:st + len2], 0.15)
        else:                              # unrelated / N-rich read
            s2 = rseq(len2, "ACGTN" if kind == 6 else "ACGT")
        s2 = s2[:256] or "A"
        matrix = ["flat", anc][i % 2]
        rc = 1 if (i % 7 == 3) else 0
        sg5 = 0 if (i % 11 == 5) else 1
        if i % 3 == 0:
            mask = "*"
        else:                              # k-mer style column mask: a few open intervals
            m = ["0"] * len(s1)
            for _ in range(rnd.randint(1, 3)):
                a = rnd.randint(0, len(s1) - 1)
                b = min(len(s1), a + rnd.randint(1, len(s2) + 30))
                for j in range(a, b):
                    m[j] = "1"
            mask = "".join(m)
        lines.append(f"D {matrix} {rc} {sg5} 1 {s1} {s2} {mask}")
    # long-gap cases (trace escape paths): read = left part + right part far apart
    for g in (40, 70, 120, 200):
        left, right = rseq(60), rseq(60)
        s1 = rseq(20) + left + rseq(g) + right + rseq(20)
        lines.append(f"D flat 0 1 1 {s1} {left + right} *")
        s2 = left[:30] + rseq(g // 2) + left[30:]          # long insertion in the read
        lines.append(f"D flat 0 1 1 {rseq(30) + left + rseq(30)} {s2[:256]} *")
    inp = "\n".join(lines) + "\n"
    out = subprocess.run([os.path.join(RB, "ref_dp_driver")], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode()
    outs = out.strip().split("\n")
    assert len(outs) == len(lines)
    with open(os.path.join(G, "dp_vectors.txt"), "w") as f:
        f.write("# input line (see oracle/ref_dp_driver.c), then the reference's answer; matrix path is relative to tests/golden\n")
        for a, b in zip(lines, outs):
            f.write(a.replace(anc, "ancient.submat.txt") + "\n" + b + "\n")


def dp_vectors_wide():
    """More of the reference's own dyn_prog answers where dp_vectors.txt is thin: windows of 257-512, 513-768 and more than
    768 columns (the three one-read-per-wavefront classes and the exact scalar kernel of the GPU path), gaps of 63 and
    more on the optimal path in either direction (the byte trace's escape), both matrices and both strands; all-ones
    mask and sg5 = 1 as reiterate_assembly aligns (src/mia_main.c:190-252).  Alignments stay below the 512 characters
    populate_pwaln_to_begin's buffers hold."""
    rnd = random.Random(20261004)
    anc = os.path.join(G, "ancient.submat.txt")
    lines = []

    def rseq(n, alphabet="ACGT"):
        return "".join(rnd.choice(alphabet) for _ in range(n))

    def noisy(s, p):
        out = []
        for ch in s:
            u = rnd.random()
            if u < p / 4:
                continue
            if u < p / 2:
                out.append(rnd.choice("ACGT"))
            out.append(rnd.choice("ACGT") if u < p else ch)
        return "".join(out) or "A"

    k = 0
    for lo, hi in ((257, 512), (513, 768), (769, 1500)):
        for _ in range(12):
            len1 = rnd.randint(lo, hi)
            len2 = rnd.choice([36, 60, 100, 100, 150, 200, 250])
            s1 = rseq(len1, "ACGT" if k % 4 else "ACGTNRY")
            st = rnd.randint(0, len1 - len2)
            s2 = noisy(s1[st:st + len2].replace("R", "A").replace("Y", "C").replace("N", "G"), rnd.choice([0.02, 0.08, 0.2]))[:256]
            lines.append(f"D {['flat', anc][k % 2]} {1 if k % 3 == 0 else 0} 1 1 {s1} {s2} *")
            k += 1
    for g in (63, 64, 80, 130, 220, 300):       # a deletion of g columns / an insertion of g read bases on the best path
        left, right = rseq(70), rseq(70)
        pad_l, pad_r = rseq(rnd.randint(10, 300)), rseq(rnd.randint(10, 300))
        lines.append(f"D {['flat', anc][g % 2]} 0 1 1 {pad_l + left + rseq(g) + right + pad_r} {left + right} *")
        if g <= 110:
            s2 = left + rseq(g) + right
            lines.append(f"D flat {g % 2} 1 1 {pad_l + left + right + pad_r} {s2[:256]} *")
    inp = "\n".join(lines) + "\n"
    out = subprocess.run([os.path.join(RB, "ref_dp_driver")], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode()
    outs = out.strip().split("\n")
    assert len(outs) == len(lines)
    with open(os.path.join(G, "dp_vectors_wide.txt"), "w") as f:
        f.write("# as dp_vectors.txt: input line (oracle/ref_dp_driver.c), then the reference's answer\n")
        for a, b in zip(lines, outs):
            f.write(a.replace(anc, "ancient.submat.txt") + "\n" + b + "\n")
    print(len(lines), "wide DP vectors")


def cons_vectors():
    rnd = random.Random(7)
    lines = []
    for i in range(300):
        cc = 1 + (i % 2)
        cov = rnd.choice([0, 1, 2, 3, 4, 7, 10, 100])
        gaps = rnd.randint(0, cov) if cov else 0
        if i % 9 == 0 and cov:
            gaps = cov // 2
        cnt = [rnd.randint(0, max(0, cov - gaps)) for _ in range(4)]
        if i % 4 == 0:
            base = rnd.randint(-3000, 3000)
            sc = [base, base, rnd.randint(-3000, 3000), base][: 4]
            rnd.shuffle(sc)
        else:
            sc = [rnd.randint(-5000, 5000) for _ in range(4)]
        if i % 13 == 0:
            sc = [-399, -400, -399, -5000]
        if i % 17 == 0:
            sc = [-1, -2401, -2402, -9000]
        lines.append(f"C {cc} {cnt[0]} {cnt[1]} {cnt[2]} {cnt[3]} {gaps} {cov} {sc[0]} {sc[1]} {sc[2]} {sc[3]}")
    inp = "\n".join(lines) + "\n"
    out = subprocess.run([os.path.join(RB, "ref_dp_driver")], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().strip().split("\n")
    with open(os.path.join(G, "cons_vectors.txt"), "w") as f:
        for a, b in zip(lines, out):
            f.write(a + "\n" + b + "\n")


def myers_vectors(mt311):
    rnd = random.Random(99)
    lines = []
    iupac = "ACGTRYSWKMBDHVN"

    def mut(s, nsub, nins, ndel):
        s = list(s)
        for _ in range(nsub):
            p = rnd.randrange(len(s)); s[p] = rnd.choice("ACGT")
        for _ in range(nins):
            p = rnd.randrange(len(s)); s.insert(p, rnd.choice("ACGT"))
        for _ in range(ndel):
            if len(s) > 1:
                del s[rnd.randrange(len(s))]
        return "".join(s)

    for i in range(220):
        n = rnd.randint(1, 400)
        alpha = "ACGT" if i % 3 else iupac
        a = "".join(rnd.choice(alpha) for _ in range(n))
        b = mut(a, rnd.randint(0, 6), rnd.randint(0, 3), rnd.randint(0, 3))
        if i % 10 == 0:
            b = b[: max(1, len(b) // 2)]
        if i % 10 == 1:
            a = a[: max(1, len(a) // 2)]
        mode = i % 3
        maxd = rnd.choice([1, 2, 5, 10, 20, 40])
        maxd = min(maxd, len(a), len(b))     # stay inside the reference's defined domain
        if maxd < 1:
            maxd = 1
        lines.append(f"{mode} {maxd} {a} {b}")
    # the ccheck-shaped call: mt311 vs a mutated assembly, maxd = len/10 (src/ccheck.cc:477)
    indiv = gen_data.resolve_individual(mt311)
    asm = mut(indiv, 25, 4, 4)
    lines.append(f"0 {max(len(mt311), len(asm)) // 10} {mt311} {asm}")
    inp = "\n".join(lines) + "\n"
    out = subprocess.run([os.path.join(RB, "ref_myers_driver")], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().strip().split("\n")
    assert len(out) == len(lines)
    with open(os.path.join(G, "myers_vectors.txt"), "w") as f:
        for a, b in zip(lines, out):
            f.write(a + "\n" + b + "\n")


def maln_cases(mt311_path):
    out_dir = os.path.join(G, "maln")
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    tmp = tempfile.mkdtemp()
    _, _, mt = gen_data.read_fasta_one(mt311_path)
    indiv = gen_data.resolve_individual(mt)
    sets = {
        "s150": dict(n=150, seed=1, damage=False),
        "d150": dict(n=150, seed=2, damage=True),
    }
    for name, kw in sets.items():
        d = gen_data.make_reads(indiv, kw["n"], 100, kw["seed"], circular=True, damage=kw["damage"])
        gen_data.write_fasta_reads(os.path.join(G, f"{name}.fa"), d["reads"])
    # planted indels: consensus must change length across iterations
    ind2 = indiv[:5000] + "ACG" + indiv[5000:9000] + indiv[9002:]
    d = gen_data.make_reads(ind2, 1200, 100, 3, circular=True, damage=True)
    # keep only reads near the planted events and the origin so the set stays small
    keep = [i for i in range(1200) if (4800 <= d["start"][i] <= 5100) or (8800 <= d["start"][i] <= 9100) or d["start"][i] > 16480 or d["start"][i] < 60]
    gen_data.write_fasta_reads(os.path.join(G, "indel.fa"), d["reads"][keep])
    # reads that run into the sequencing adapter (-T): 40-90 genome bases, then none / part / all of the adapter
    rnd = random.Random(515)
    d = gen_data.make_reads(indiv, 140, 100, 4, circular=True, damage=False)
    with open(os.path.join(G, "adapt.fa"), "w") as f:
        for i in range(140):
            body = d["reads"][i].tobytes().decode()[: rnd.randint(40, 90)]
            ad = [NEAND_ADAPT, STAND_ADAPT][i % 2]
            tail = ["", ad[: rnd.randint(1, 12)], ad[: rnd.randint(13, len(ad))], ad][i % 4]
            f.write(f">a{i}\n{body + tail}\n")
    # the fixture reads as FASTQ (read_fastq, src/io.c:35-175): same bases, dummy qualities, one over-long record and
    # one record with a description
    recs = open(os.path.join(G, "tf.fna")).read().split(">")[1:]
    with open(os.path.join(G, "tf.fq"), "w") as f:
        for k, r in enumerate(recs):
            head, seq = r.split("\n", 1)
            seq = seq.replace("\n", "")
            if k == 3:
                head = head.split()[0] + " a description"
            f.write(f"@{head}\n{seq}\n+\n{'I' * len(seq)}\n")
        long_seq = (recs[0].split("\n", 1)[1].replace("\n", "") * 4)[:300]
        f.write(f"@toolong\n{long_seq}\n+\n{'#' * len(long_seq)}\n")
    A = "ancient.submat.txt"
    cases = {
        "fix_c_fq": ["-r", "tr1.fna", "-f", "tf.fq", "-c"],
        "adapt_T_k12": ["-r", "mt311.fa", "-f", "adapt.fa", "-c", "-k", "12", "-T"],
        "adapt_T_aS_k12": ["-r", "mt311.fa", "-f", "adapt.fa", "-c", "-k", "12", "-T", "-a", "S"],
        "adapt_T_user_k12": ["-r", "mt311.fa", "-f", "adapt.fa", "-c", "-k", "12", "-T", "-a", "GTCAGACACGCAACAGG"],
        "fix_c_T": ["-r", "tr1.fna", "-f", "tf.fna", "-c", "-T"],
        "fix_c": ["-r", "tr1.fna", "-f", "tf.fna", "-c"],
        "fix_c_n": ["-r", "tr1.fna", "-f", "tf.fna", "-c", "-n"],
        "fix_lin": ["-r", "tr1.fna", "-f", "tf.fna"],
        "fix_c_p2": ["-r", "tr1.fna", "-f", "tf.fna", "-c", "-p", "2"],
        "fix_c_H": ["-r", "tr1.fna", "-f", "tf.fna", "-c", "-H", "4000"],
        "fix_c_k8M": ["-r", "tr1.fna", "-f", "tf.fna", "-c", "-k", "8", "-M"],
        "fix_c_anc": ["-r", "tr1.fna", "-f", "tf.fna", "-c", "-s", A],
        "s150_k12": ["-r", "mt311.fa", "-f", "s150.fa", "-c", "-k", "12"],
        "s150_full": ["-r", "mt311.fa", "-f", "s150.fa", "-c", "-n"],
        "d150_anc_k12": ["-r", "mt311.fa", "-f", "d150.fa", "-c", "-k", "12", "-s", A],
        "indel_anc_k12": ["-r", "mt311.fa", "-f", "indel.fa", "-c", "-k", "12", "-s", A],
        "indel_anc_k12_SN": ["-r", "mt311.fa", "-f", "indel.fa", "-c", "-k", "12", "-s", A, "-S", "150", "-N", "100"],
    }
    hashes = {}
    for name, args in cases.items():
        root = os.path.join(tmp, name)
        sh([os.path.join(RB, "mia")] + args + ["-m", root], cwd=G, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        it = 1
        while os.path.exists(f"{root}.{it}"):
            with open(f"{root}.{it}") as f:
                body = f.readlines()[1:]
            if it <= 4:
                with open(os.path.join(out_dir, f"{name}.{it}"), "w") as f:
                    f.writelines(body)
            else:       # long runs (a case that oscillates until MAX_ITER): later iterations are pinned by hash
                import hashlib
                hashes[f"{name}.{it}"] = hashlib.sha256("".join(body).encode()).hexdigest()
            it += 1
        print(f"{name}: {it - 1} iteration file(s)")
    with open(os.path.join(out_dir, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    with open(os.path.join(out_dir, "hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)


def g2_sets():
    """2 000 seeded reads against mt311 -- whole runs, no k-mer filter, iterated to convergence.  The
    read files are regenerated from the seed where they are needed (gen_data), only their hash is committed."""
    return {
        "g2_flat": dict(seed=21, damage=False, args=["-c", "-i"]),
        "g2_anc": dict(seed=22, damage=True, args=["-c", "-i", "-s", "ancient.submat.txt"]),
        "g2_pe": dict(seed=23, damage=True, args=["-c", "-i", "-s", "ancient.submat.solexa.pe.txt"]),
        # BASELINE.json configs[4]'s shape: 150 bp damaged reads against a 100 kb linear region (seed 5), ancient matrix, -k 12
        "g2_c4": dict(seed=24, damage=True, args=["-i", "-s", "ancient.submat.txt", "-k", "12"], read_len=150, ref="rand100k"),
    }


def g2_ref(name, out_path):
    """the reference FASTA of a G2 set: mt311 (committed) or the seeded 100 kb region; returns (path to hand to -r, genome the reads come from)"""
    kw = g2_sets()[name]
    if kw.get("ref") == "rand100k":
        g = gen_data.random_reference(100_000, seed=5)
        gen_data.write_fasta(out_path, "region100k synthetic", g)
        return out_path, g
    _, _, mt = gen_data.read_fasta_one(os.path.join(G, "mt311.fa"))
    return "mt311.fa", gen_data.resolve_individual(mt)


def g2_reads(name, out_path):
    kw = g2_sets()[name]
    _, genome = g2_ref(name, out_path + ".ref.fa")
    d = gen_data.make_reads(genome, 2000, kw.get("read_len", 100), kw["seed"], circular="-c" in kw["args"], damage=kw["damage"])
    gen_data.write_fasta_reads(out_path, d["reads"])
    import hashlib
    return hashlib.sha256(open(out_path, "rb").read()).hexdigest()


def g2_cases(only=None):
    """The reference's own mia on the G2 sets (two minutes of CPU each: pass 1 is the whole-reference DP).  Every .maln of
    a run is pinned by the sha256 of its text from line 2 on; the SEQ line of each iteration's reference is kept readable.
    only: names to (re)generate, the others keep their committed entries."""
    import hashlib
    shutil.copy(os.path.join(REF, "matrices", "ancient.submat.solexa.pe.txt"), os.path.join(G, "ancient.submat.solexa.pe.txt"))
    tmp = tempfile.mkdtemp()
    out = {}
    if only and os.path.exists(os.path.join(G, "g2_runs.json")):
        with open(os.path.join(G, "g2_runs.json")) as f:
            out = json.load(f)
    procs = {}
    for name, kw in g2_sets().items():
        if only and name not in only:
            continue
        fa = os.path.join(tmp, name + ".fa")
        out[name] = {"reads_sha256": g2_reads(name, fa), "args": kw["args"], "maln_sha256": [], "ref_seq": []}
        ref_arg, _ = g2_ref(name, fa + ".ref.fa")
        procs[name] = subprocess.Popen([os.path.join(RB, "mia"), "-r", ref_arg, "-f", fa] + kw["args"] + ["-m", os.path.join(tmp, name)], cwd=G,
                                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for name, pr in procs.items():
        assert pr.wait() == 0, name
        it = 1
        while os.path.exists(os.path.join(tmp, f"{name}.{it}")):
            with open(os.path.join(tmp, f"{name}.{it}")) as f:
                body = f.readlines()[1:]
            out[name]["maln_sha256"].append(hashlib.sha256("".join(body).encode()).hexdigest())
            out[name]["ref_seq"].append(hashlib.sha256(next(l for l in body if l.startswith("SEQ ")).encode()).hexdigest()[:16])
            it += 1
        print(f"{name}: {it - 1} iteration file(s)")
    with open(os.path.join(G, "g2_runs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)


def ingest_cases_golden():
    """what the reference's own reader (oracle/_ref/ref_read_driver) makes of tools/ingest_cases.py's inputs"""
    import hashlib
    import ingest_cases
    tmp = tempfile.mkdtemp()
    out = {}
    for name, text in ingest_cases.cases().items():
        path = os.path.join(tmp, name)
        with open(path, "wb") as f:
            f.write(text.encode())
        r = subprocess.run([os.path.join(RB, "ref_read_driver"), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        recs = r.stdout.split(b"\n")[:-1]
        out[name] = {"input_sha256": hashlib.sha256(text.encode()).hexdigest(), "records": len(recs),
                     "records_sha256": hashlib.sha256(r.stdout).hexdigest(), "stderr_sha256": hashlib.sha256(r.stderr).hexdigest(),
                     "stderr_lines": len(r.stderr.split(b"\n")) - 1, "first": recs[0].decode("latin1") if recs else "", "last": recs[-1].decode("latin1") if recs else ""}
        print(name, out[name]["records"], "records,", out[name]["stderr_lines"], "message lines")
    with open(os.path.join(G, "ingest.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)


def iter_push_sets():
    """The reference's own per-iteration loop (oracle/ref_iter_driver.c around reiterate_assembly + pop_smp + cull + sort +
    consensus_assembly_string) on a read store filled with post-pass-1 fields -- what bench.py and the full-size parity
    tests hand to the GPU path and, through ora_push_frag, to the oracle.  3 000 seeded reads each, true positions jittered
    by up to +-12 columns so that windows are not centred, three iterations."""
    return {
        "ip_flat": dict(seed=31, damage=False, matrix=None, paired=False),
        "ip_anc": dict(seed=32, damage=True, matrix="ancient.submat.txt", paired=False),
        "ip_pe": dict(seed=33, damage=True, matrix="ancient.submat.solexa.pe.txt", paired=True),
    }


def iter_push_inputs(name):
    """(reference string, stored reads [n, 100], rc, as, ae) of an iter_push set, regenerated from its seed"""
    import numpy as np
    kw = iter_push_sets()[name]
    _, _, mt = gen_data.read_fasta_one(os.path.join(G, "mt311.fa"))
    indiv = gen_data.resolve_individual(mt)
    n, L = 3000, 100
    d = (gen_data.make_paired_reads(indiv, n, L, kw["seed"], damage=True) if kw["paired"]
         else gen_data.make_reads(indiv, n, L, kw["seed"], circular=True, damage=kw["damage"], indel_rate=0.003))
    stored = gen_data.stored_orientation(d)
    jit = np.random.default_rng(kw["seed"] + 1000).integers(-12, 13, size=n)
    as_ = ((d["start"] + jit) % len(indiv)).astype(np.int32)
    return mt.upper(), stored, d["strand"].astype(np.uint8), as_, (as_ + L - 1).astype(np.int32)


def iter_push_cases():
    import hashlib
    import numpy as np
    tmp = tempfile.mkdtemp()
    out = {}
    for name, kw in iter_push_sets().items():
        ref, stored, rc, as_, ae = iter_push_inputs(name)
        gen_data.write_fasta(os.path.join(tmp, "ref.fa"), "ref", ref)
        with open(os.path.join(tmp, "reads.txt"), "w") as f:
            for i in range(len(rc)):
                f.write(f"{int(rc[i])} {int(as_[i])} {int(ae[i])} {stored[i].tobytes().decode()}\n")
        matrix = os.path.join(REF, "matrices", kw["matrix"]) if kw["matrix"] else "flat"
        dump = os.path.join(tmp, "dump.txt")
        sh([os.path.join(RB, "ref_iter_driver"), os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.txt"), "1", matrix, "3", dump])
        its, cur = [], None
        for line in open(dump):
            t = line.split()
            if t[0] == "I":
                cur = {"cons_len": len(t[2]), "cons_sha256": hashlib.sha256(t[2].encode()).hexdigest(), "reads": []}
                its.append(cur)
            else:
                cur["reads"].append([int(x) for x in t[1:4]])
        for it in its:
            a = np.array(it.pop("reads"), dtype=np.int32)
            it["n_reads"] = len(a)
            it["reads_sha256"] = hashlib.sha256(a.tobytes()).hexdigest()          # [n, 3] int32: score, as, ae
            it["first_reads"] = a[:8].tolist()
        out[name] = {"matrix": kw["matrix"], "inputs_sha256": hashlib.sha256(stored.tobytes() + rc.tobytes() + as_.tobytes()).hexdigest(), "iterations": its}
        print(name, [it["cons_len"] for it in its])
    with open(os.path.join(G, "iter_push.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)


NEAND_ADAPT = "GTCAGACACGCAACAGGGGATAGGCAAGGCACACAGGGGATAGG"     # src/mia_main.c:462-463 (data, quoted for the inputs)
STAND_ADAPT = "CTGAGACACGCAACAGGGGATAGGCAAGGCACACAGGGGATAGG"


def trim_inputs(n=500, seed=77):
    """(adapter, read) pairs: no adapter, whole / partial / mutated / gapped adapter at the 3' end, adapter in the
    middle followed by junk, one-base matches at the very end, reads with N, user-defined adapters."""
    rnd = random.Random(seed)

    def rseq(k, alphabet="ACGT"):
        return "".join(rnd.choice(alphabet) for _ in range(k))

    def mut(s, p):
        out = []
        for ch in s:
            u = rnd.random()
            if u < p / 3:
                continue
            if u < 2 * p / 3:
                out.append(rnd.choice("ACGT"))
            out.append(rnd.choice("ACGT") if u < p else ch)
        return "".join(out)

    pairs = []
    for i in range(n):
        ad = [NEAND_ADAPT, STAND_ADAPT, rseq(rnd.randint(2, 127)), rseq(rnd.randint(8, 30))][i % 4]
        kind = i % 10
        body = rseq(rnd.randint(1, 200), "ACGTN" if kind == 9 else "ACGT")
        if kind == 0:
            read = body
        elif kind in (1, 2):
            read = body + ad[: rnd.randint(1, len(ad))]
        elif kind == 3:
            read = body + (mut(ad, 0.1) or "A")[: rnd.randint(min(3, len(ad)), len(ad))]
        elif kind == 4:                                    # adapter, then unrelated sequence
            read = body + ad[: rnd.randint(min(5, len(ad)), len(ad))] + rseq(rnd.randint(1, 90))
        elif kind == 5:                                    # adapter prefix early, long junk, adapter continues
            k = rnd.randint(min(4, len(ad)), max(min(4, len(ad)), len(ad) // 2))
            read = body[:40] + ad[:k] + rseq(rnd.randint(20, 120)) + ad[k:]
        elif kind == 6:                                    # starts inside the adapter
            read = body + ad[rnd.randint(1, max(1, len(ad) - 2)):]
        elif kind == 7:                                    # the read IS adapter
            read = ad[: rnd.randint(1, len(ad))]
        elif kind == 8:                                    # single-base coincidences at the end
            read = body + ad[0]
        else:
            read = body + mut(ad, 0.05)
        pairs.append((ad, read[:256] or "A"))
    return pairs


def trim_vectors():
    pairs = trim_inputs()
    out = subprocess.run([os.path.join(RB, "ref_dp_driver")], input="".join(f"T {a} {r}\n" for a, r in pairs).encode(), check=True,
                         stdout=subprocess.PIPE).stdout.decode().splitlines()
    assert len(out) == len(pairs)
    with open(os.path.join(G, "trim_vectors.txt"), "w") as f:
        for (a, r), o in zip(pairs, out):
            f.write(f"{a} {r}\n{o}\n")
    print("trim vectors:", len(pairs), "trimmed:", sum(1 for o in out if o.split()[1] == "1"))


MA_HEADER = "/* map_alignment [V1.0] */ golden\n"   # line 1 of a .maln carries a timestamp and is not stored
MA_RUNS = [(5, 1), (5, 2), (41, 1), (41, 2), (4, 1)]


def ma_cases():
    """Reports of the reference's own `ma` on every committed .maln: -f 5 (FASTA), -f 41 / -f 4 (column table),
    consensus codes 1 and 2.  Outputs above 40 KB are pinned by sha256 (tests/golden/ma/hashes.json)."""
    import glob
    import hashlib
    out_dir = os.path.join(G, "ma")
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    tmp = tempfile.mkdtemp()
    hashes = {}
    for path in sorted(glob.glob(os.path.join(G, "maln", "*.[0-9]"))):
        name = os.path.basename(path)
        full = os.path.join(tmp, name)
        with open(full, "w") as f:
            f.write(MA_HEADER + open(path).read())
        for fmt, code in MA_RUNS:
            out = subprocess.run([os.path.join(RB, "ma"), "-M", full, "-f", str(fmt), "-c", str(code)], check=True,
                                 stdout=subprocess.PIPE).stdout
            key = f"{name}.f{fmt}c{code}"
            if len(out) <= 40 * 1024:
                with open(os.path.join(out_dir, key), "wb") as f:
                    f.write(out)
            else:
                hashes[key] = {"sha256": hashlib.sha256(out).hexdigest(), "bytes": len(out)}
    # -I: a user-assigned id for the FASTA header
    name = "fix_c.2"
    out = subprocess.run([os.path.join(RB, "ma"), "-M", os.path.join(tmp, name), "-f", "5", "-I", "my_assembly"], check=True,
                         stdout=subprocess.PIPE).stdout
    with open(os.path.join(out_dir, name + ".f5c1.I"), "wb") as f:
        f.write(out)
    with open(os.path.join(out_dir, "hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)
    print("ma goldens:", len(os.listdir(out_dir)) - 1, "files,", len(hashes), "hashed")


CCHECK_RUNS = {
    # name -> (arguments in front of the file names, file names)
    "plain": ([], ["cc_flat.maln"]),
    "ancient": (["-a"], ["cc_anc.maln"]),
    "anc_plain": ([], ["cc_anc.maln"]),
    "tv": (["-t"], ["cc_flat.maln"]),
    "span": (["-s", "2000-9000"], ["cc_flat.maln"]),
    "n2": (["-n", "2"], ["cc_flat.maln"]),
    "table": (["-T"], ["cc_flat.maln"]),
    "table_a_two_files": (["-T", "-a"], ["cc_anc.maln", "cc_flat.maln"]),
    "two_files": ([], ["cc_flat.maln", "cc_anc.maln"]),
    "v1": (["-v"], ["cc_flat.maln"]),
    "v2": (["-vv"], ["cc_flat.maln"]),
    "v3": (["-vvv"], ["cc_flat.maln"]),
    "v4": (["-vvvv", "-a"], ["cc_anc.maln"]),
    "v5": (["-vvvvv"], ["cc_small.maln"]),
    "v6": (["-vvvvvv", "-F"], ["cc_small.maln"]),
    "maxd_too_small": (["-d", "20"], ["cc_flat.maln"]),
    "maxd_large": (["-d", "4000"], ["cc_flat.maln"]),
    "other_ref": (["-r", "cc_contam.fa"], ["cc_flat.maln"]),
    "few_positions": ([], ["cc_small.maln"]),
    "few_positions_F": (["-F"], ["cc_small.maln"]),
    "few_positions_F_table": (["-F", "-T", "-a", "-n", "2"], ["cc_small.maln"]),
}


def ccheck_cases(mt311_path):
    """The reference's own ccheck (oracle/_ref/ccheck) on assemblies made by the reference's own mia:
       cc_flat   an individual 170 substitutions + 3 indels away from mt311, 2 750 reads of 100 and 60 bases,
                 13 % of them from a second (contaminating) human, flat matrix
       cc_anc    the same with deaminated reads and the ancient matrix
       cc_small  the 150-read set of the maln goldens (too few diagnostic positions: needs -F)
    Committed: the final .maln of each (gzip, line 1 replaced by a fixed header), cc_contam.fa, and for every run of
    CCHECK_RUNS stdout / stderr / exit code (streams above 64 KB as sha256)."""
    import gzip
    import hashlib
    out_dir = os.path.join(G, "ccheck")
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    tmp = tempfile.mkdtemp()
    _, _, mt = gen_data.read_fasta_one(mt311_path)
    rnd = random.Random(909)
    endo = list(gen_data.resolve_individual(mt, seed=311))
    for p in rnd.sample(range(len(endo)), 170):
        endo[p] = rnd.choice([b for b in "ACGT" if b != endo[p]])
    endo = "".join(endo)
    endo = endo[:3000] + endo[3002:9000] + "TT" + endo[9000:12000] + endo[12001:]
    contam = gen_data.resolve_individual(mt, seed=77)
    gen_data.write_fasta(os.path.join(out_dir, "cc_contam.fa"), "contaminant one resolved human", contam)
    for name, damage in (("cc_flat", False), ("cc_anc", True)):
        parts = [(endo, 1500, 100, 21), (endo, 900, 60, 22), (contam, 250, 100, 23), (contam, 100, 60, 24)]
        recs = []
        for k, (genome, n, ln, seed) in enumerate(parts):
            d = gen_data.make_reads(genome, n, ln, seed + (10 if damage else 0), circular=True, damage=damage and k < 2)
            recs += [(f"{'e' if k < 2 else 'c'}{k}_{i}", d["reads"][i].tobytes().decode()) for i in range(n)]
        rnd.shuffle(recs)
        with open(os.path.join(tmp, name + ".fa"), "w") as f:
            for rid, seq in recs:
                f.write(f">{rid}\n{seq}\n")
        args = ["-r", mt311_path, "-f", name + ".fa", "-c", "-k", "12", "-i"]
        if damage:
            args += ["-s", os.path.join(G, "ancient.submat.txt")]
        sh([os.path.join(RB, "mia")] + args + ["-m", name], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        it = 1
        while os.path.exists(os.path.join(tmp, f"{name}.{it + 1}")):
            it += 1
        body = open(os.path.join(tmp, f"{name}.{it}")).readlines()[1:]
        with open(os.path.join(tmp, name + ".maln"), "w") as f:
            f.write(MA_HEADER + "".join(body))
        print(f"{name}: iteration {it}")
    with open(os.path.join(tmp, "cc_small.maln"), "w") as f:
        f.write(MA_HEADER + open(os.path.join(G, "maln", "s150_k12.2")).read())
    shutil.copy(os.path.join(out_dir, "cc_contam.fa"), tmp)
    for name in ("cc_flat", "cc_anc"):
        with open(os.path.join(tmp, name + ".maln"), "rb") as f, gzip.GzipFile(os.path.join(out_dir, name + ".maln.gz"), "wb", mtime=0) as g:
            g.write(f.read())
    # find_maln (src/ccheck.cc:206-236): without -f the highest-numbered sibling is read
    shutil.copy(os.path.join(tmp, "cc_small.maln"), os.path.join(tmp, "sib.1"))
    shutil.copy(os.path.join(tmp, "cc_flat.maln"), os.path.join(tmp, "sib.3"))
    runs = dict(CCHECK_RUNS)
    hashes = {}
    for key, (args, files) in list(runs.items()) + [("find_maln", (None, ["sib.1"]))]:
        argv = [os.path.join(RB, "ccheck")] + (["-f"] + args if args is not None else []) + files
        r = subprocess.run(argv, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        for ext, data in (("out", r.stdout), ("err", r.stderr)):
            if len(data) <= 64 * 1024:
                with open(os.path.join(out_dir, f"{key}.{ext}"), "wb") as f:
                    f.write(data)
            elif f"{key}.{ext}" == "v6.err":
                # print_aln prints the assembly row of myers_diff, which the reference leaves without its terminator
                # (src/myers_align.c:44-45): the bytes after it are heap remains.  Kept in full so that the test can
                # compare everything but that tail.
                with gzip.GzipFile(os.path.join(out_dir, f"{key}.{ext}.gz"), "wb", mtime=0) as g:
                    g.write(data)
            else:
                hashes[f"{key}.{ext}"] = {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data)}
        with open(os.path.join(out_dir, f"{key}.rc"), "w") as f:
            f.write(f"{r.returncode}\n")
        print(f"ccheck {key}: rc {r.returncode}, {len(r.stdout)} B out, {len(r.stderr)} B err")
    with open(os.path.join(out_dir, "runs.json"), "w") as f:
        json.dump(runs, f, indent=1)
    with open(os.path.join(out_dir, "hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)


def main():
    os.makedirs(G, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "trim":        # only the trim_frag vectors
        sh(["make", "-s", "-f", "oracle/Makefile.ref"], cwd=ROOT)
        trim_vectors()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ccheck":      # only the ccheck inputs and reports
        sh(["make", "-s", "-f", "oracle/Makefile.ref"], cwd=ROOT)
        ccheck_cases(os.path.join(G, "mt311.fa"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ingest":      # only the fragment reader's cases
        ingest_cases_golden()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "dp_wide":     # only the wide-window / long-gap DP vectors
        dp_vectors_wide()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "iter_push":   # only the per-iteration loop on pushed reads
        iter_push_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g2":          # only the 2 000-read whole runs
        sh(["make", "-s", "-f", "oracle/Makefile.ref"], cwd=ROOT)
        g2_cases(sys.argv[2:] or None)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ma":          # only the ma reports (the .maln files stay as they are)
        sh(["make", "-s", "-f", "oracle/Makefile.ref"], cwd=ROOT)
        ma_cases()
        return
    sh(["make", "-s", "-f", "oracle/Makefile.ref"], cwd=ROOT)
    shutil.copy(os.path.join(RB, "mt311.fa"), os.path.join(G, "mt311.fa"))
    shutil.copy(os.path.join(REF, "test", "tr1.fna"), os.path.join(G, "tr1.fna"))
    shutil.copy(os.path.join(REF, "test", "tf.fna"), os.path.join(G, "tf.fna"))
    shutil.copy(os.path.join(REF, "matrices", "ancient.submat.txt"), os.path.join(G, "ancient.submat.txt"))
    _, _, mt = gen_data.read_fasta_one(os.path.join(G, "mt311.fa"))
    dp_vectors()
    cons_vectors()
    myers_vectors(mt)
    maln_cases(os.path.join(G, "mt311.fa"))
    g2_cases()
    ma_cases()
    trim_vectors()
    ccheck_cases(os.path.join(G, "mt311.fa"))


if __name__ == "__main__":
    main()
