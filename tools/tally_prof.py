"""Where k_tally_binned's time goes (profiling only): the steady step of configs[1] / [2] with parts of the kernel switched off
through MIA_HIP_DEBUG_SKIP (8: no one-read-per-lane paths, 32: no one-read-per-wavefront path, 2048: one-gap reads dropped,
4096: no bit-sliced counts, 16: position-specific matrix: no per-base adds).  The sums of such runs are wrong by construction;
only the stage time is read.  usage: tally_prof.py [config]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import mia_amd  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
w = bench.make_workload(cfg, n_reads, 1 if cfg == 1 else 3)
hip = mia_amd.MiaHip(0)
pipe = bench.Pipeline(hip, w)
cur = w["ref"]
for _ in range(5):
    cur = pipe.step(cur)
hip.close()
for label, skip in (("default", 0), ("no lane paths", 8), ("no wavefront path", 32), ("one-gap reads dropped", 2048), ("no bit slices", 4096),
                    ("no per-base adds (PSSM)", 16), ("lanes and wavefront paths off", 40), ("no adds to the vertical counters", 1 << 18),
                    ("counters not folded into the window", 1 << 19), ("no slab store", 1 << 20), ("records fetched, no read taken", 1 << 21),
                    ("no counters, no fold, no slab", (1 << 18) | (1 << 19) | (1 << 20)),
                    ("rows at either end read by read (MIA_HIP_NO_TALLY_RUNS)", "MIA_HIP_NO_TALLY_RUNS"), ("round 4's tally (MIA_HIP_STRAND_SPLIT=0)", "MIA_HIP_STRAND_SPLIT=0")):
    name, val = "MIA_HIP_DEBUG_SKIP", str(skip)
    if isinstance(skip, str):
        name, _, val = skip.partition("=")
        val = val or "1"
    if skip:
        os.environ[name] = val
    hip = mia_amd.MiaHip(0)
    os.environ.pop(name, None)
    pipe = bench.Pipeline(hip, w)
    try:
        pipe.step(cur)
        pipe.reset_stats()
        for _ in range(6):
            pipe.step(cur)
        hip.sync()
        st = hip.stage_stats()
        print(label, {k: round(v[0] / max(v[1], 1), 4) for k, v in st.items() if v[1] and k == "k_tally_binned"}, flush=True)
    except Exception as ex:
        print(label, "failed:", repr(ex)[:120], flush=True)
    hip.close()
