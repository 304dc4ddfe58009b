#!/usr/bin/env python3
"""Seeded synthetic inputs for MIA (SURVEY.md section 8(d) recipe).

  individual genome : mt311 with every IUPAC code resolved uniformly at random
                      among its bases, upper-cased (seed 311)
  reads             : uniform start on the circle, fixed length, fair strand
                      coin, 1 % substitutions, 0.1 % indels (length 1-3)
  aDNA damage       : C->T with p = 0.30*exp(-0.35*i) at distance i from the
                      5' end, G->A mirrored at the 3' end, applied BEFORE the
                      strand flip (configs 3-5)

Pure numpy so that 1-10 M reads generate in seconds; used by tests, bench.py
and the golden-fixture generator.  Nothing here touches /root/reference.
"""
import numpy as np

IUPAC = {
    "A": "A", "C": "C", "G": "G", "T": "T", "U": "T",
    "R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT", "M": "AC",
    "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT",
}
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


def read_fasta_one(path):
    """First record of a FASTA file -> (id, desc, sequence with case kept)."""
    name, seq = None, []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    break
                name = line[1:].rstrip("\n")
            else:
                seq.append(line.strip())
    parts = name.split(None, 1)
    return parts[0], (parts[1] if len(parts) > 1 else ""), "".join(seq)


def resolve_individual(ref_seq, seed=311):
    """Resolve IUPAC codes uniformly at random, upper-case everything."""
    rng = np.random.default_rng(seed)
    out = []
    for ch in ref_seq.upper():
        opts = IUPAC.get(ch, "ACGT")
        out.append(opts[rng.integers(len(opts))] if len(opts) > 1 else opts)
    return "".join(out)


def random_reference(length, seed=5):
    rng = np.random.default_rng(seed)
    return "".join(np.array(list("ACGT"))[rng.integers(0, 4, size=length)])


def make_reads(genome, n_reads, read_len=100, seed=1, circular=True, sub_rate=0.01,
               indel_rate=0.001, damage=False, start=None, strand=None, fast_indels=False):
    """Returns dict with
         reads   : uint8 array [n_reads, read_len] of ASCII bases as sequenced
         start   : int64 true 0-based start on the genome (forward strand)
         strand  : uint8 0 = forward, 1 = reverse complement
       Indels are applied on a slightly longer template so every read comes out
       at exactly read_len bases.  start / strand: place the reads there instead of
       drawing positions and strands (make_paired_reads).  fast_indels: inserted bases in one
       draw (another random stream than the seeded goldens were made with)."""
    rng = np.random.default_rng(seed)
    g = np.frombuffer(genome.encode(), dtype=np.uint8)
    L = len(g)
    pad = 8
    if start is not None:
        start = np.asarray(start, dtype=np.int64)
    elif circular:
        start = rng.integers(0, L, size=n_reads)
    else:
        start = rng.integers(0, L - read_len - pad, size=n_reads)
    idx = (start[:, None] + np.arange(read_len + pad)[None, :])
    if circular:
        idx %= L
    tmpl = g[idx]                                            # [n, read_len+pad]
    # --- indels: at most one event per read keeps this vectorisable and is the
    #     dominant case at 0.1 %/base (P(two events in 100 bp) ~ 0.5 %)
    has_indel = rng.random(n_reads) < (1.0 - (1.0 - indel_rate) ** read_len)
    is_ins = rng.random(n_reads) < 0.5
    ilen = rng.integers(1, 4, size=n_reads)
    ipos = rng.integers(5, read_len - 5, size=n_reads)
    reads = tmpl[:, :read_len].copy()
    rows = np.nonzero(has_indel)[0]
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    if len(rows):                                            # ~10 % of reads, all at once
        p, k, ins_row = ipos[rows][:, None], ilen[rows][:, None], is_ins[rows][:, None]
        j = np.arange(read_len)[None, :]
        # where each base of the read comes from in the template: an insertion pushes the tail right, a deletion pulls it left
        src = np.where(ins_row, np.where(j >= p + k, j - k, j), np.where(j >= p, j + k, j))
        fixed = np.take_along_axis(tmpl[rows], src, axis=1)
        irows = np.nonzero(ins_row[:, 0])[0]
        insb = np.zeros((len(rows), 3), dtype=np.uint8)
        if fast_indels:
            insb[irows] = bases[rng.integers(0, 4, size=(len(irows), 3))]
        else:                                                # the stream of the first version, draw by draw (seeded goldens)
            for q in irows:
                kk = int(k[q, 0])
                insb[q, :kk] = bases[rng.integers(0, 4, size=kk)]
        inside = ins_row & (j >= p) & (j < p + k)
        fixed = np.where(inside, np.take_along_axis(insb, np.clip(j - p, 0, 2), axis=1), fixed)
        reads[rows] = fixed
    # --- substitutions
    sub = rng.random(reads.shape) < sub_rate
    shift = rng.integers(1, 4, size=reads.shape)
    code = np.zeros(256, dtype=np.int64)
    for i, b in enumerate(b"ACGT"):
        code[b] = i
    rc = code[reads]
    rc = np.where(sub, (rc + shift) % 4, rc)
    reads = bases[rc]
    # --- ancient-DNA deamination, before the strand flip
    if damage:
        i = np.arange(read_len)
        p5 = 0.30 * np.exp(-0.35 * i)[None, :]
        p3 = 0.30 * np.exp(-0.35 * (read_len - 1 - i))[None, :]
        u = rng.random(reads.shape)
        reads = np.where((reads == ord("C")) & (u < p5), ord("T"), reads)
        u = rng.random(reads.shape)
        reads = np.where((reads == ord("G")) & (u < p3), ord("A"), reads).astype(np.uint8)
    strand = (rng.random(n_reads) < 0.5).astype(np.uint8) if strand is None else np.asarray(strand, dtype=np.uint8)
    flipped = _COMP[reads[:, ::-1]]
    reads = np.where(strand[:, None] == 1, flipped, reads).astype(np.uint8)
    return {"reads": reads, "start": start.astype(np.int64), "strand": strand}


def make_reads_chunked(genome, n_reads, read_len=100, seed=1, chunk=500_000, **kw):
    """make_reads for millions of reads without its multi-gigabyte temporaries: chunks of `chunk` reads, chunk c seeded
    with (seed, c).  A different random stream than make_reads(seed) -- say which one made a data set."""
    parts = _pool_map(lambda cl: make_reads(genome, min(chunk, n_reads - cl[1]), read_len, seed=[seed, cl[0]], fast_indels=True, **kw),
                      list(enumerate(range(0, n_reads, chunk))))
    return {k: np.concatenate([p[k] for p in parts]) for k in ("reads", "start", "strand")}


def _pool_map(fn, items):
    """chunks on host threads (numpy drops the GIL inside its array operations); results in order"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(len(items), (os.cpu_count() or 2) - 1, 16))
    if workers == 1:
        return [fn(x) for x in items]
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, items))


def make_paired_reads(genome, n_reads, read_len=100, seed=1, frag_mean=300.0, frag_sd=30.0, damage=True, sub_rate=0.01,
                      indel_rate=0.001, chunk=500_000):
    """SURVEY.md section 8(d), BASELINE configs[3]: two reads per fragment of 300 +- 30 bp on the circular genome, ids
    /1 and /2 -- mate 1 reads the fragment's first read_len bases on the fragment's strand, mate 2 its last read_len bases
    on the other strand (the usual inward-facing pair); sequencing errors, indels and deamination are drawn for each mate
    on its own, exactly as make_reads draws them.  MIA treats the mates as independent reads (reference
    src/mia_main.c:759-805 reads one sequence at a time; nothing looks at the ids).  Reads come out interleaved:
    pair p = reads 2p, 2p+1.  n_reads must be even.  Extra keys: frag_start, frag_len, ids()."""
    assert n_reads % 2 == 0
    L = len(genome)
    def one(cl):
        c, lo = cl
        m = min(chunk // 2, n_reads // 2 - lo)
        rng = np.random.default_rng([seed, c, 77])
        fstart = rng.integers(0, L, size=m)
        flen = np.maximum(np.rint(rng.normal(frag_mean, frag_sd, size=m)).astype(np.int64), read_len)
        fstrand = (rng.random(m) < 0.5).astype(np.uint8)
        far = (fstart + flen - read_len) % L
        s1 = np.where(fstrand == 0, fstart, far)          # forward-strand coordinate of each mate's first base
        s2 = np.where(fstrand == 0, far, fstart)
        d1 = make_reads(genome, m, read_len, seed=[seed, c, 1], circular=True, sub_rate=sub_rate, indel_rate=indel_rate,
                        damage=damage, start=s1, strand=fstrand, fast_indels=True)
        d2 = make_reads(genome, m, read_len, seed=[seed, c, 2], circular=True, sub_rate=sub_rate, indel_rate=indel_rate,
                        damage=damage, start=s2, strand=1 - fstrand, fast_indels=True)
        res = {"frag_start": fstart, "frag_len": flen}
        for k in ("reads", "start", "strand"):
            both = np.empty((2 * m,) + d1[k].shape[1:], dtype=d1[k].dtype)
            both[0::2], both[1::2] = d1[k], d2[k]
            res[k] = both
        return res
    parts = _pool_map(one, list(enumerate(range(0, n_reads // 2, chunk // 2))))
    out = {k: [p[k] for p in parts] for k in ("reads", "start", "strand", "frag_start", "frag_len")}
    d = {k: np.concatenate(v) for k, v in out.items()}
    d["ids"] = lambda prefix="r": ["%s%d/%d" % (prefix, i // 2, i % 2 + 1) for i in range(n_reads)]
    return d


def stored_orientation(d):
    """Reads as MIA keeps them in its read store after pass 1: reverse-strand reads are
    reverse-complemented once the strand is known (reference src/fsdb.c:209-227)."""
    reads, strand = d["reads"], d["strand"]
    flipped = _COMP[reads[:, ::-1]]
    return np.where(strand[:, None] == 1, flipped, reads).astype(np.uint8)


def write_fasta_reads(path, reads, prefix="r", ids=None):
    with open(path, "w") as f:
        for i in range(reads.shape[0]):
            f.write(f">{ids[i] if ids is not None else prefix + str(i)}\n{reads[i].tobytes().decode()}\n")


def write_fasta(path, name, seq, width=60):
    with open(path, "w") as f:
        f.write(f">{name}\n")
        for i in range(0, len(seq), width):
            f.write(seq[i:i + width] + "\n")


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", required=True, help="reference FASTA (e.g. mt311.fa)")
    ap.add_argument("--out", required=True)
    ap.add_argument("-n", type=int, default=1000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--damage", action="store_true")
    ap.add_argument("--linear", action="store_true")
    ap.add_argument("--paired", action="store_true", help="configs[3]: two reads per 300 +- 30 bp fragment, ids /1 and /2")
    a = ap.parse_args()
    _, _, ref = read_fasta_one(a.ref)
    indiv = resolve_individual(ref)
    if a.paired:
        d = make_paired_reads(indiv, a.n, a.len, a.seed, damage=a.damage)
        write_fasta_reads(a.out, d["reads"], ids=d["ids"]())
    else:
        d = make_reads(indiv, a.n, a.len, a.seed, circular=not a.linear, damage=a.damage)
        write_fasta_reads(a.out, d["reads"])
