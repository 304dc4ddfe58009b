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
               indel_rate=0.001, damage=False):
    """Returns dict with
         reads   : uint8 array [n_reads, read_len] of ASCII bases as sequenced
         start   : int64 true 0-based start on the genome (forward strand)
         strand  : uint8 0 = forward, 1 = reverse complement
       Indels are applied on a slightly longer template so every read comes out
       at exactly read_len bases."""
    rng = np.random.default_rng(seed)
    g = np.frombuffer(genome.encode(), dtype=np.uint8)
    L = len(g)
    pad = 8
    if circular:
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
    for r in rows:                                           # ~10 % of reads; cheap loop
        p, k = int(ipos[r]), int(ilen[r])
        t = tmpl[r]
        if is_ins[r]:
            ins = bases[rng.integers(0, 4, size=k)]
            reads[r] = np.concatenate([t[:p], ins, t[p:]])[:read_len]
        else:
            reads[r] = np.concatenate([t[:p], t[p + k:]])[:read_len]
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
    strand = (rng.random(n_reads) < 0.5).astype(np.uint8)
    flipped = _COMP[reads[:, ::-1]]
    reads = np.where(strand[:, None] == 1, flipped, reads).astype(np.uint8)
    return {"reads": reads, "start": start.astype(np.int64), "strand": strand}


def stored_orientation(d):
    """Reads as MIA keeps them in its read store after pass 1: reverse-strand reads are
    reverse-complemented once the strand is known (reference src/fsdb.c:209-227)."""
    reads, strand = d["reads"], d["strand"]
    flipped = _COMP[reads[:, ::-1]]
    return np.where(strand[:, None] == 1, flipped, reads).astype(np.uint8)


def write_fasta_reads(path, reads, prefix="r"):
    with open(path, "w") as f:
        for i in range(reads.shape[0]):
            f.write(f">{prefix}{i}\n{reads[i].tobytes().decode()}\n")


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
    a = ap.parse_args()
    _, _, ref = read_fasta_one(a.ref)
    indiv = resolve_individual(ref)
    d = make_reads(indiv, a.n, a.len, a.seed, circular=not a.linear, damage=a.damage)
    write_fasta_reads(a.out, d["reads"])
