"""Host-side collectives of the sharded iteration, device agnostic:
the same functions run on CUDA tensors over RCCL (bench.py, backend "nccl") and on CPU
tensors over gloo (tests/test_dist_gloo.py).

Reads are sharded in contiguous fsdb blocks, rank r owning block r, so "concatenate in
rank order" == fsdb order."""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_initialized() else 1


def _active():
    return dist.is_initialized()


def all_gather_concat(t):
    """Concatenate equally sized 1-D tensors of all ranks in rank (= fsdb) order: the
    global score array the reference's score-cut regression runs over (src/fsdb.c:269-383)."""
    if not _active():
        return t
    parts = [torch.empty_like(t) for _ in range(world())]
    dist.all_gather(parts, t)
    return torch.cat(parts)


def exclusive_rank_sum(value, device):
    """Sum of `value` over all lower ranks: the AlnSeq slot index of this shard's first
    record (records are numbered in merge order across the whole fsdb, src/map_align.c:882)."""
    if not _active():
        return 0
    v = torch.tensor([int(value)], dtype=torch.int64, device=device)
    parts = [torch.empty_like(v) for _ in range(world())]
    dist.all_gather(parts, v)
    return int(sum(int(p.item()) for p in parts[: dist.get_rank()]))


def allreduce_tallies(tally, gaps):
    """In place: integer column tallies add up, ref->gaps is a maximum (src/mia.c:486-504)."""
    if not _active():
        return
    dist.all_reduce(tally, op=dist.ReduceOp.SUM)
    dist.all_reduce(gaps, op=dist.ReduceOp.MAX)


def allreduce_tallies_with_counts(tally, gaps, n_mine):
    """allreduce_tallies plus every rank's n_mine (its number of insert events) WITHOUT a collective of its own: the
    counts ride on the MAX all-reduce of the gaps in `world` extra slots, rank r filling slot r.  Returns the list of
    counts in rank order (what all_gather_ragged wants)."""
    if not _active():
        return [int(n_mine)]
    w, r = world(), dist.get_rank()
    dist.all_reduce(tally, op=dist.ReduceOp.SUM)
    ext = torch.zeros(gaps.numel() + w, dtype=gaps.dtype, device=gaps.device)
    ext[: gaps.numel()] = gaps
    ext[gaps.numel() + r] = int(n_mine)
    dist.all_reduce(ext, op=dist.ReduceOp.MAX)
    gaps.copy_(ext[: gaps.numel()])
    return ext[gaps.numel():].cpu().tolist()


def gather_pre_cull(sums5, n_records, device, n_links=None):
    """One small all-gather before the cull instead of four collectives: every rank's score sums (mia_hip_score_sums),
    record count and -- if given -- the number of links its cull will emit (mia_hip_pre_cull_counts).  Returns
    (global sums5, slot_base) or, with n_links, (global sums5, slot_base, every rank's link count)."""
    if not _active():
        return (sums5, 0) if n_links is None else (sums5, 0, [int(n_links)])
    mine = torch.tensor(list(map(int, sums5)) + [int(n_records), int(n_links or 0)], dtype=torch.int64, device=device)
    allt = torch.empty(7 * world(), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allt, mine)
    allv = allt.cpu().numpy().reshape(world(), 7)
    import numpy as np
    g = np.array([allv[:, 0].sum(), allv[:, 1].sum(), allv[:, 2].sum(), allv[:, 3].min(), allv[:, 4].max()], dtype=np.int64)
    base = int(allv[: dist.get_rank(), 5].sum())
    return (g, base) if n_links is None else (g, base, [int(x) for x in allv[:, 6]])


def allreduce_score_sums(sums5, device):
    """{sum len, sum score, count, min len, max len} of mia_hip_score_sums over all ranks (integers: exact)."""
    if not _active():
        return sums5
    t = torch.as_tensor(sums5, dtype=torch.int64).to(device)
    add, lo, hi = t[:3].clone(), t[3:4].clone(), t[4:5].clone()
    dist.all_reduce(add, op=dist.ReduceOp.SUM)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return torch.cat([add, lo, hi]).cpu().numpy()


def _gather_counts(n, device):
    """every rank's n as a python list, one collective and one device-to-host copy"""
    mine = torch.tensor([int(n)], dtype=torch.int64, device=device)
    allc = torch.empty(world(), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allc, mine)
    return allc.cpu().tolist()


def all_gather_ragged(t, counts=None):
    """Concatenate variable-length 1-D int64 tensors (insert events, links) of all ranks in rank order."""
    if not _active():
        return t
    if counts is None:
        counts = _gather_counts(t.numel(), t.device)
    mx = max(max(counts), 1)
    mine = torch.zeros(mx, dtype=t.dtype, device=t.device)
    mine[: t.numel()] = t
    allt = torch.empty(mx * world(), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(allt, mine)
    if all(c == mx for c in counts):
        return allt
    return torch.cat([allt[r * mx: r * mx + counts[r]] for r in range(world())]).contiguous()


def exchange_links(hip, as_tensor, link_counts=None):
    """Between cull() and tally() of a sharded iteration: the links of formerly split reads (stale fs->back_asp,
    include/mia_hip.h) may point at AlnSeq slots of another rank.  Every rank gets all links, applies those that hit
    its own slots, and the record lengths the readers need come back with a max-reduce.
    as_tensor(ptr, n, typestr) wraps a device pointer of the library as a tensor (bench.DevArray on CUDA).
    link_counts: see below."""
    if not _active():
        return
    # almost always no rank has a link.  link_counts (every rank's number of links, known before the cull from
    # gather_pre_cull) settles that without a collective or a device round trip; otherwise one tiny all-gather does.
    if link_counts is not None and sum(link_counts) == 0:
        return
    ptr, n = hip.links()
    counts = [4 * c for c in link_counts] if link_counts is not None else _gather_counts(4 * n, _device())
    if sum(counts) == 0:
        return
    mine = as_tensor(ptr, 4 * n, "<i8") if n else torch.zeros(0, dtype=torch.int64, device=_device())
    alll = all_gather_ragged(mine, counts)
    total = int(alll.numel()) // 4
    if total == 0:
        return
    _sync(alll)
    hip.set_links(alll.data_ptr(), total)
    lp, ap, ln = hip.link_lengths()
    for ptr in (lp, ap):
        t = as_tensor(ptr, ln, "<i4")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        _sync(t)
    hip.finish_links()


def _device():
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def _sync(t):
    if t.is_cuda:
        torch.cuda.synchronize()
