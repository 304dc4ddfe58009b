import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tools"); sys.path.insert(0, ROOT + "/tests")
import numpy as np, bench, gen_data, mia_amd
w = bench.make_workload(1, 60_000, 7)
rng = np.random.default_rng(66)
n0, ref = w["n"], w["plain_ref"]
long_n, long_len = 300, 230
d = gen_data.make_reads(ref, long_n, long_len, seed=67, circular=True, damage=False)
n = n0 + long_n
stored = np.full((n, long_len), ord("A"), np.uint8)
stored[:n0, :100] = w["stored"]
stored[n0:] = gen_data.stored_orientation(d)
lens = np.concatenate([np.full(n0, 100, np.int32), np.full(long_n, long_len, np.int32)])
rc = np.concatenate([w["rc"], d["strand"].astype(np.uint8)])
as0 = np.concatenate([w["as_"], d["start"].astype(np.int32)])
ae0 = (as0 + lens - 1).astype(np.int32)
keep = np.nonzero((as0 > 400) & (as0 < len(ref) - 700))[0]
stored, lens, rc, as0, ae0 = stored[keep], lens[keep], rc[keep], as0[keep], ae0[keep]
n0, n = int((keep < n0).sum()), len(keep)
acgt = np.frombuffer(b"ACGT", np.uint8)
for i in np.concatenate([rng.choice(n0, 1500, replace=False), np.arange(n0, n)]):
    k = int(lens[i]) // 5
    stored[i, rng.choice(int(lens[i]), k, replace=False)] = acgt[rng.integers(0, 4, k)]
sk = np.ones(n, np.uint8)
offsets = np.zeros(n + 1, np.int64); offsets[1:] = np.cumsum(lens)
flat = np.concatenate([stored[:n0, :100].reshape(-1), stored[n0:].reshape(-1)])
for variant in ("sk_all", "sk_some"):
    if variant == "sk_some": sk[rng.choice(n0, 200, replace=False)] = 0
    hip = mia_amd.MiaHip(0); hip.set_pssm(w["pssm"]); hip.upload_reads(flat, offsets, rc, sk, as0, ae0)
    cur = ref
    for it in range(3):
        cons = hip.iterate(cur, True)
        c = hip.bx_counters()
        print(variant, it, "seen", c[15], "done", c[12], "open", c[29], "fails", c[16:24], "qleft", c[10], "cons==cur", cons == cur, flush=True)
        cur = cons
    hip.close()
