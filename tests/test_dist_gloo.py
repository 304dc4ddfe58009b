"""world_size-2 gloo test of the sharded iteration's host logic: reads
sharded in contiguous fsdb blocks, column tallies all-reduced (sum), gaps (max), scores
gathered in fsdb order.  The per-shard compute is done by the oracle here (no GPU in
this container); the collectives are mapping-iterative-assembler_amd/dist.py, the same
code bench.py runs over RCCL."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

WORKER = r'''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mia_amd
from mia_amd import dist as mdist
import oracle_ctypes as oc
from mia_flow import oracle_after_pass1, fsdb_arrays

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
oracle = oc.load(os.path.join(ROOT, "oracle", "_build", "libmia_oracle.so"))

def shard_file(path, lo, hi, out):
    recs = open(path).read().split(">")[1:]
    open(out, "w").write("".join(">" + r for r in recs[lo:hi]))
    return len(recs)

src = os.path.join(GOLDEN, "d150.fa")
n_total = len(open(src).read().split(">")) - 1
per = (n_total + world - 1) // world
mine = os.path.join(TMP, f"shard{rank}.fa")
shard_file(src, rank * per, min(n_total, (rank + 1) * per), mine)

def run(reads):
    st, o, anc = oracle_after_pass1(oracle, "mt311.fa", reads, True, 12, "ancient.submat.txt", hard_cut=17500)
    L = oracle.ora_ref_len(st); ref = oracle.ora_ref_seq(st)[:L]
    oracle.ora_iterate(st, ref, 1)
    t = (C.c_int * (L * 10))(); oracle.ora_column_tallies(st, t)
    gaps = np.ctypeslib.as_array(oracle.ora_ref_gaps(st), shape=(L,)).copy()
    fs = fsdb_arrays(oracle, st)
    return np.ctypeslib.as_array(t).reshape(L, 10).copy(), gaps, fs, oracle.ora_num_culled(st)

tally, gaps, fs, nrec = run(mine)
tt, tg = torch.from_numpy(tally.astype(np.int32)), torch.from_numpy(gaps.astype(np.int32))
mdist.allreduce_tallies(tt, tg)
scores = mdist.all_gather_concat(torch.from_numpy(np.pad(fs["score"], (0, per - fs["n"]), constant_values=-1)))
base = mdist.exclusive_rank_sum(nrec, "cpu")
ev = mdist.all_gather_ragged(torch.arange(rank * 100, rank * 100 + 3 + rank, dtype=torch.int64))
# the same with the counts riding on the gaps all-reduce (no collective of their own)
tt2, tg2 = torch.from_numpy(tally.astype(np.int32)), torch.from_numpy(gaps.astype(np.int32))
cnts = mdist.allreduce_tallies_with_counts(tt2, tg2, 3 + rank)
assert cnts == [3 + r for r in range(world)], cnts
assert torch.equal(tt2, tt) and torch.equal(tg2, tg)
ev2 = mdist.all_gather_ragged(torch.arange(rank * 100, rank * 100 + 3 + rank, dtype=torch.int64), cnts)
assert torch.equal(ev2, ev)

s5 = mdist.allreduce_score_sums(np.array([10 + rank, 100 * (rank + 1), 3, 50 - rank, 60 + rank], dtype=np.int64), "cpu")
assert s5.tolist() == [21, 300, 6, 49, 61], s5

g5, sb = mdist.gather_pre_cull(np.array([10 + rank, 100 * (rank + 1), 3, 50 - rank, 60 + rank], dtype=np.int64), 7 + rank, "cpu")
assert g5.tolist() == [21, 300, 6, 49, 61] and sb == (0 if rank == 0 else 7), (g5, sb)
g5, sb, lc = mdist.gather_pre_cull(np.array([10 + rank, 100 * (rank + 1), 3, 50 - rank, 60 + rank], dtype=np.int64), 7 + rank, "cpu", 2 - rank)
assert g5.tolist() == [21, 300, 6, 49, 61] and sb == (0 if rank == 0 else 7) and lc == [2, 1], (g5, sb, lc)

# the link exchange of formerly split reads (stale back_asp): rank 0 has two links, rank 1 one; a link is 4 int64
# {reader, slot, flen<<32|actf, low}.  A stand-in with the library's five calls checks the protocol of exchange_links.
class FakeHip:
    def __init__(self):
        n = 2 - rank
        self.own = torch.tensor([[1000 * rank + k, 7 + 10 * k + rank, (50 << 32) | 40, k & 1] for k in range(n)], dtype=torch.int64).reshape(-1)
        self.done = False
    def links(self):
        return self.own.data_ptr(), self.own.numel() // 4
    def set_links(self, ptr, n):
        self.all = np.ctypeslib.as_array((C.c_int64 * (4 * n)).from_address(ptr)).reshape(n, 4).copy()
        # "my" slots are those with slot % 2 == rank: their record length is known here, the others are -1
        self.lens = torch.tensor([200 + int(s) if int(s) % 2 == rank else -1 for s in self.all[:, 1]], dtype=torch.int32)
        self.acts = self.lens.clone()
    def link_lengths(self):
        return self.lens.data_ptr(), self.acts.data_ptr(), self.lens.numel()
    def finish_links(self):
        self.done = True

def cpu_tensor(ptr, n, ts):
    ct = {"<i8": C.c_int64, "<i4": C.c_int32}[ts]
    return torch.from_numpy(np.ctypeslib.as_array((ct * n).from_address(ptr)))

fh = FakeHip()
mdist.exchange_links(fh, cpu_tensor)
assert fh.done and fh.all[:, 0].tolist() == [0, 1, 1000], fh.all          # all links, rank order
assert fh.lens.tolist() == [200 + int(s) for s in fh.all[:, 1]], fh.lens     # every length resolved by its owner
assert fh.acts.tolist() == fh.lens.tolist()
# with the link counts known beforehand (mia_hip_pre_cull_counts through gather_pre_cull): same result without the count
# gather; and when nobody has a link nothing is called at all
fh2 = FakeHip()
mdist.exchange_links(fh2, cpu_tensor, [2, 1])
assert fh2.done and np.array_equal(fh2.all, fh.all) and fh2.lens.tolist() == fh.lens.tolist()
class NoLinks:
    def links(self):
        raise AssertionError("links() must not be called when every rank reported zero links")
mdist.exchange_links(NoLinks(), cpu_tensor, [0, 0])
if rank == 0:
    full_t, full_g, full_fs, full_nrec = run("d150.fa")
    assert np.array_equal(tt.numpy(), full_t), "summed shard tallies != unsharded tallies"
    assert np.array_equal(tg.numpy(), full_g), "max of shard gaps != unsharded gaps"
    got = scores.numpy(); got = got[got >= 0]
    assert np.array_equal(got, full_fs["score"]), "gathered scores are not in fsdb order"
    assert base == 0
    assert ev.tolist() == [0, 1, 2, 100, 101, 102, 103]
    print("DIST_OK")
else:
    assert base > 0
dist.destroy_process_group()
'''


def test_two_rank_sharded_tallies(oracle_build, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\nGOLDEN = {GOLDEN!r}\nTMP = {str(tmp_path)!r}\n" + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert "DIST_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
