#!/usr/bin/env python3
"""Whole-run comparison: the reference's own `mia` binary (oracle/_ref/mia, built in place from
/root/reference by oracle/Makefile.ref) against the mia_hip command line on the same synthetic FASTA.
Checks that every .maln iteration is byte-identical from line 2 and reports both wall times as JSON.

usage: python tools/whole_run.py [-n READS] [--kmer K] [--matrix FILE] [--plain-ref] [--damage] [--keep DIR]
Needs a GPU (mia_hip) and the prebuilt reference binary; reads nothing under /root/reference at run time."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=20000)
    ap.add_argument("--kmer", type=int, default=0)
    ap.add_argument("--keep", default=None)
    ap.add_argument("--skip-reference", action="store_true")
    ap.add_argument("--plain-ref", action="store_true", help="reference = mt311 with its ambiguity codes resolved (the usual kind of "
                    "reference; lets the diagonal filter of pass 1 work)")
    ap.add_argument("--matrix", default=None, help="substitution matrix file under tests/golden (-s), e.g. ancient.submat.txt; default: the flat matrix")
    ap.add_argument("--damage", action="store_true", help="aDNA damage on the synthetic reads")
    ap.add_argument("--ccheck", action="store_true", help="also run ccheck (reference and ccheck_hip) on the final .maln")
    ap.add_argument("--ccheck-reference", action="store_true", help="with --skip-reference: still time the reference's ccheck on mia_hip's .maln")
    a = ap.parse_args()
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mia")
    hip_bin = os.path.join(ROOT, "mapping-iterative-assembler_amd", "mia_hip")
    work = a.keep or tempfile.mkdtemp(prefix="mia_whole_")
    os.makedirs(work, exist_ok=True)
    reads = os.path.join(work, "reads.fa")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_data.py"), "--ref", os.path.join(GOLDEN, "mt311.fa"),
                    "--out", reads, "-n", str(a.n), "--len", "100", "--seed", "7"] + (["--damage"] if a.damage else []), check=True)
    ref_fa = os.path.join(GOLDEN, "mt311.fa")
    if a.plain_ref:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_data
        _, _, mt = gen_data.read_fasta_one(ref_fa)
        ref_fa = os.path.join(work, "plain_ref.fa")
        gen_data.write_fasta(ref_fa, "mt311_resolved", gen_data.resolve_individual(mt, seed=99))   # not the reads' own individual
    args = ["-r", ref_fa, "-f", reads, "-c"]
    if a.kmer > 0:
        args += ["-k", str(a.kmer)]
    if a.matrix:
        args += ["-s", os.path.join(GOLDEN, a.matrix)]
    env = dict(os.environ, MIA_DATA_PATH=GOLDEN)
    out = {"reads": a.n, "kmer": a.kmer, "plain_ref": bool(a.plain_ref), "matrix": a.matrix or "flat", "damage": bool(a.damage)}
    runs = [("mia_hip", hip_bin)] + ([] if a.skip_reference else [("reference", ref_bin)])
    for label, exe in runs:
        root = os.path.join(work, label)
        t0 = time.perf_counter()
        subprocess.run([exe] + args + ["-m", root], check=True, env=env, cwd=work, stdout=subprocess.DEVNULL,
                       stderr=(None if (label == "mia_hip" and os.environ.get("MIA_HIP_TIMING")) else subprocess.DEVNULL))
        out[label + "_s"] = round(time.perf_counter() - t0, 3)
        it = 0
        while os.path.exists("%s.%d" % (root, it + 1)):
            it += 1
        out[label + "_iterations"] = it
    if not a.skip_reference:
        same = out["mia_hip_iterations"] == out["reference_iterations"] and out["reference_iterations"] > 0
        for it in range(1, out["reference_iterations"] + 1):
            if not same:
                break
            x = open(os.path.join(work, "reference.%d" % it)).readlines()[1:]
            y = open(os.path.join(work, "mia_hip.%d" % it)).readlines()[1:]
            same = x == y
        out["maln_identical"] = bool(same)
        out["speedup"] = round(out["reference_s"] / out["mia_hip_s"], 1)
    ok = out.get("maln_identical", True)
    if a.ccheck:
        # the contamination check of the final assembly: the reference's ccheck against ccheck_hip, same file
        final = os.path.join(work, "mia_hip.%d" % out["mia_hip_iterations"])
        reports = {}
        for label, exe in (("ccheck_hip", os.path.join(ROOT, "mapping-iterative-assembler_amd", "ccheck_hip")),
                           ("ccheck_reference", os.path.join(ROOT, "oracle", "_ref", "ccheck"))):
            if label == "ccheck_reference" and a.skip_reference and not a.ccheck_reference:
                continue
            t0 = time.perf_counter()
            r = subprocess.run([exe, "-f", "-F", "-a", os.path.basename(final)], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            out[label + "_s"] = round(time.perf_counter() - t0, 3)
            reports[label] = (r.returncode, r.stdout, r.stderr)
        if len(reports) == 2:
            out["ccheck_identical"] = reports["ccheck_hip"] == reports["ccheck_reference"]
            out["ccheck_speedup"] = round(out["ccheck_reference_s"] / out["ccheck_hip_s"], 1)
            ok = ok and out["ccheck_identical"]
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
