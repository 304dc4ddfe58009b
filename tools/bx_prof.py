#!/usr/bin/env python3
"""Stage times of one MIA iteration (realign / score sums / cull / tally / consensus) for a matrix and a read set, and
what each realignment stage finished.  usage: bx_prof.py [--matrix flat|ancient|solexa] [--reads N] [--len 100] [--steps K]
[--ref mt311|random:<len>] [--linear]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_data  # noqa: E402
import mia_amd  # noqa: E402

MATS = {"ancient": "ancient.submat.txt", "solexa": "ancient.submat.solexa.pe.txt"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="flat")
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--ref", default="mt311")
    ap.add_argument("--linear", action="store_true")
    ap.add_argument("--no-damage", action="store_true")
    ap.add_argument("--again", action="store_true", help="one more step from the starting reference at the end (the first step pays for the allocations)")
    a = ap.parse_args()
    if a.ref == "mt311":
        _, _, mt = gen_data.read_fasta_one(os.path.join(ROOT, "tests", "golden", "mt311.fa"))
        ref0, indiv = mt.upper(), gen_data.resolve_individual(mt)
    else:
        indiv = gen_data.random_reference(int(a.ref.split(":")[1]), seed=5)
        ref0 = indiv
    n = a.reads
    d = gen_data.make_reads(indiv, n, a.len, 1, circular=not a.linear, damage=(a.matrix != "flat" and not a.no_damage))
    stored = gen_data.stored_orientation(d)
    as_ = d["start"].astype(np.int32)
    ae = (as_ + a.len - 1).astype(np.int32)
    off = np.arange(n + 1, dtype=np.int64) * a.len
    hip = mia_amd.MiaHip(0)
    hip.set_pssm(mia_amd.flat_pssm() if a.matrix == "flat" else mia_amd.read_pssm(os.path.join(ROOT, "tests", "golden", MATS[a.matrix])))
    hip.upload_reads(stored.reshape(-1), off, d["strand"].astype(np.uint8), np.ones(n, np.uint8), as_, ae)
    phase = {}

    def tick(name, t0):
        hip.sync()
        phase.setdefault(name, []).append(time.perf_counter() - t0)
        return time.perf_counter()

    def step(cur):
        t0 = time.perf_counter()
        hip.realign(cur, not a.linear)
        t0 = tick("realign", t0)
        sums = hip.score_sums()
        cut = hip.score_cut_from_sums(sums)
        if cut is None:
            cut = hip.score_cut(hip.scores(), np.full(n, a.len, np.int32))
        t0 = tick("score", t0)
        hip.cull(0, cut[0] if cut[0] > 0 else 100.0, cut[1], 0)
        t0 = tick("cull", t0)
        hip.tally()
        t0 = tick("tally", t0)
        c = hip.consensus(1)
        tick("consensus", t0)
        return c

    cur = ref0
    out = {"matrix": a.matrix, "reads": n, "len": a.len, "steps": []}
    for k in range(a.steps + (1 if a.again else 0)):
        if k == a.steps:
            cur = ref0
        for f in (hip.bx_stats, hip.filter_stats, hip.band_stats, hip.plain_stats, hip.kernel_time):
            f(reset=True)
        t0 = time.perf_counter()
        nxt = step(cur)
        wall = time.perf_counter() - t0
        bx = hip.bx_stats()
        out["steps"].append({"wall_ms": wall * 1e3, "phases_ms": {p: v[-1] * 1e3 for p, v in phase.items()},
                             "bx_reads": bx[0], "bx_counters": hip.bx_counters(), "bx_ms": bx[1], "filter": hip.filter_stats(), "band": hip.band_stats(),
                             "plain": hip.plain_stats(), "trace_full": hip.kernel_time(), "changed": nxt != cur, "cons_len": len(nxt)})
        cur = nxt
    print(json.dumps(out))


if __name__ == "__main__":
    main()
