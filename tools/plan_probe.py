import os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, mia_amd
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = bench.make_workload(cfg, 1_000_000, 1 if cfg == 1 else 3)
hip = mia_amd.MiaHip(0); pipe = bench.Pipeline(hip, w); cur = w["ref"]
for _ in range(5): cur = pipe.step(cur)
hip.close()
for env in ({}, {"MIA_HIP_BX_DEBUG": "32"}):
    os.environ.update(env); hip = mia_amd.MiaHip(0)
    for k in env: os.environ.pop(k)
    pipe = bench.Pipeline(hip, w); pipe.step(cur); pipe.reset_stats()
    for _ in range(6): pipe.step(cur)
    hip.sync(); st = hip.stage_stats()
    print(env, {k: round(v[0] / max(v[1], 1), 4) for k, v in st.items() if v[1]}, flush=True); hip.close()
