"""Where a wavefront of k_bx_plan's first launch spends its cycles (alt build, MIA_HIP_BX_DEBUG=512: shader-clock stamps between the
stretches of the kernel, each stretch run over the whole wavefront before the next starts).  usage: plan_clk_probe.py <config> <reads> [first]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MIA_HIP_BX_DEBUG"] = "512"
import bench  # noqa: E402
import mia_amd  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
first = len(sys.argv) > 3
w = bench.make_workload(cfg, n_reads, 1 if cfg == 1 else 3)
hip = mia_amd.MiaHip(0)
pipe = bench.Pipeline(hip, w)
cur = w["ref"]
for _ in range(3):
    cur = pipe.step(cur)
out = (C.c_uint64 * 12)()
hip._l.mia_hip_debug_plan_clk.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
hip._l.mia_hip_debug_plan_clk(hip._h, out)              # reset
K = 4
pipe.reset_stats()
for _ in range(K):
    pipe.step(w["ref"] if first else cur)
hip.sync()
hip._l.mia_hip_debug_plan_clk(hip._h, out)
v = list(out)
names = ["set-up", "fetch + window + plannable", "planes", "anchors", "finish", "emit", "hand-over"]
if not first and not os.environ.get("MIA_HIP_NO_QUICK_PLAN"):        # the quick plan is in front (steady state): its stretches
    names = ["set-up", "fetch + window + planes (x4)", "one diagonal, bx_quick (x4)", "results and marks (x4)", "barrier + one indel, bx_quick2", "list reservations", "list entries"]
tot = sum(v[:7])
waves = max(v[11], 1)
print("config", cfg, "reads", n_reads, "first iteration" if first else "steady", "| wavefronts", waves // K, "per launch; cycles per wavefront (s_memtime, 100 MHz):")
for k, nm in enumerate(names):
    print("  %-28s %10.1f  %5.1f %%" % (nm, v[k] / waves, 100.0 * v[k] / max(tot, 1)))
st = hip.stage_stats()
print("  k_bx_plan stage ms per step (instrumented):", st["k_bx_plan"][0] / K)
hip.close()
hip2 = mia_amd.MiaHip(0)
pipe2 = bench.Pipeline(hip2, w)
c2 = w["ref"]
for _ in range(4):
    c2 = pipe2.step(c2)
ctr = hip2.bx_counters()
print("  last step's counters: seen %d, finished by the plan %d, values lists %s, trace lists %s, late lists %s, quick -> one-indel form %d, quick -> full plan %d, open %d" %
      (ctr[15], ctr[12], ctr[0:5], ctr[5:10], ctr[24:29], ctr[16], ctr[10], ctr[29]))
hip2.close()
