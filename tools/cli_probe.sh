# usage (GPU box): bash tools/cli_probe.sh -- the phases of a 1 M-read mia_hip run (MIA_HIP_TIMING=1), final .maln only
python3 - <<'PY'
import os, sys, subprocess, time, tempfile, re
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tools"); sys.path.insert(0, ROOT + "/tests")
import bench
w = bench.make_workload(1, 1_000_000, 1)
import numpy as np, gen_data
tmp = tempfile.mkdtemp()
m = w["n"]; stored, rc = w["stored"], w["rc"]
seq = np.where(rc[:, None] == 1, gen_data._COMP[stored[:, ::-1]], stored).astype(np.uint8)
L = seq.shape[1]
rec = np.empty((m, 10 + L + 1), np.uint8)
rec[:, 0], rec[:, 1], rec[:, 9], rec[:, -1] = ord(">"), ord("r"), ord("\n"), ord("\n")
idx = np.arange(m)
for k in range(7): rec[:, 8 - k] = ord("0") + (idx // 10 ** k) % 10
rec[:, 10:10 + L] = seq
fa = os.path.join(tmp, "reads.fa"); rec.tofile(fa)
gen_data.write_fasta(os.path.join(tmp, "ref.fa"), "ref", w["ref"])
exe = os.path.join(ROOT, "mapping-iterative-assembler_amd", "mia_hip")
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([exe, "-r", os.path.join(tmp, "ref.fa"), "-f", fa, "-c", "-i", "-m", os.path.join(tmp, "out"), "-F"], env=dict(os.environ, MIA_HIP_TIMING="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print("wall %.3f s rc %d" % (time.perf_counter() - t0, r.returncode))
    for mm in re.finditer(r"\[mia_hip timing\]\s+(.*?)\s+([0-9.]+) ms", r.stderr.decode(errors="replace")): print("   %-50s %8.1f" % (mm.group(1), float(mm.group(2))))
PY
