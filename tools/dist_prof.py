"""Where the time of a sharded iteration goes on the host side: wraps the collectives of mia_amd.dist and the library calls
of bench.py's step with synchronised timers (MIA_BENCH_FORCE_DIST=1: the RCCL path on one GPU).  usage: python tools/dist_prof.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MIA_BENCH_FORCE_DIST"]="1"
sys.argv=["bench.py","--no-cpu-baseline","--steps","10","--warmup","3"]
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import mia_amd
from mia_amd import dist as mdist
acc={}
def wrap(mod,name):
    f=getattr(mod,name)
    def g(*a,**k):
        torch.cuda.synchronize(); t=time.perf_counter(); r=f(*a,**k); torch.cuda.synchronize(); acc.setdefault(name,[]).append(time.perf_counter()-t); return r
    setattr(mod,name,g)
for n in ("gather_pre_cull","exchange_links","allreduce_tallies_with_counts","all_gather_ragged"): wrap(mdist,n)
for n in ("num_records","links","tally_buffers","ins_events","set_ins_events","score_sums","cull","tally","consensus","realign"): wrap(mia_amd.MiaHip,n)
import runpy
runpy.run_path(os.path.join(ROOT,"bench.py"), run_name="__main__")
for k in sorted(acc):
    v = sorted(acc[k])
    print("%-32s median %8.1f us  (min %.1f, %d calls)" % (k, v[len(v) // 2] * 1e6, v[0] * 1e6, len(v)))
