#!/usr/bin/env python3
"""bench.py -- reads aligned per second per iteration on MI355X.

One "step" = one MIA iteration over all stored reads (reference
src/mia_main.c:931-963): reiterate_assembly + pop_smp_from_FSDB +
cull_maln_from_fsdb + consensus_assembly_string, inputs resident in HBM.

Headline (`value`): BASELINE.json configs[1] -- 1 M synthetic 100 bp reads (1 % subs, 0.1 % indels, both strands)
against the 16 619 bp mt311 reference, circular, flat matrix.  Pass-1 coordinates are the generator's true positions
(pass 1 is reported separately under "pass1"); every iteration re-aligns every read in its +-50 window exactly as the
reference does.  At N = 1 the same JSON line also carries
  "configs2"  configs[2]: 1 M aDNA-damaged reads, matrices/ancient.submat.txt, iterated from mt311 itself to convergence
  "configs4"  configs[4] at its size on one GPU: 5 M reads of 150 bp against a 100 kb linear reference, ancient matrix
              ("configs4_share": 625 k of them, one GPU's share of the 8-GPU job)
  "peaks"     the two ceilings measured on this device: streaming-copy GB/s and int32 VALU wave-instructions/s
  "myers"     the bit-vector edit distance (reference src/myers_align.c) on a batch of pairs
  "cpu_baseline"  the reference's own loop (oracle/_ref/ref_iter_driver) on the host cores, flat and ancient matrix

  "configs3"  configs[3] on one GPU: 10 M paired damaged reads, matrices/ancient.submat.solexa.pe.txt (the N = 1 point of the
              strong-scaling curve the N > 1 lines continue)
  "value_first_iteration"  configs[1] read literally: the iteration against mt311 itself (`value` is the steady state)

N > 1 (one process per GPU; `python bench.py --gpus N` starts the N ranks itself through torch.distributed.run when no
launcher did, and fails loudly when the box has fewer GPUs): reads are sharded in contiguous fsdb blocks; per iteration the
int32 column tallies are all-reduced (sum) and the gap lengths (max), insert events are all-gathered -- inside libmia_hip
(mia_hip_iterate over its RCCL communicator; the line carries the rank count RCCL itself reports).  The default job at
N > 1 is north_star's: STRONG scaling of configs[3] (10 M paired reads in total, split over the ranks) as `value`, and the
weak figure (configs[1], 1 M reads per GPU) under "weak".  --scaling weak: --reads is PER GPU.  --scaling strong: --reads
is the whole job, split over the ranks.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s nominal (about 6.3 TB/s achievable)
# SURVEY.md section 8(d), algorithmic HBM bytes per read per iteration: ceil(len/2) packed bases + 16 B meta in,
# 16 B result + len bytes of edit script out -> 182 B at 100 bp, 257 B at 150 bp
def bytes_per_read(read_len):
    return (read_len + 1) // 2 + 16 + 16 + read_len


PROFILE_ROUND = "r06"
PMC_DIR = os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc")
CSRC = os.path.join(ROOT, "mapping-iterative-assembler_amd", "csrc")
STAGES = ["k_diag_filter", "k_band_align", "k_bx_plan", "k_bx_values", "k_bx_trace", "k_align_quad_plain", "k_align_quad", "k_tally_binned"]
# the full-window stage is timed as a whole: k_align_quad (four reads per wavefront, windows up to 208 columns) and the
# k_align_window<CPL> launches (one read per wavefront: wider windows, reads whose path left the quad kernel's trace band)
# the band DPs are timed by stream: stream2 carries the values DP and, behind it, the trace DP of its left-overs
STAGE_KERNELS = {"k_align_quad": ["k_align_quad", "k_align_window"], "k_bx_values": ["k_bxl_values", "k_bxl_trace_late"], "k_bx_trace": ["k_bxl_trace"],
                 "k_bx_plan": ["k_bx_plan"]}           # (the plan is two launches of one kernel per step)
# what `roofline.kernel` may name: stages that are ONE kernel of the rocprofv3 kernel trace (profiles/r0N/cfgK_kernel_stats.csv),
# under the name it carries there; stage groups (k_bx_values = values DP + late trace, k_align_quad = quad + window classes)
# are reported under roofline.groups in bench_extras.json, never as the dominant kernel
ROCPROF_NAME = {"k_bx_trace": "k_bxl_trace", "k_tally_binned": "k_tally_binned", "k_diag_filter": "k_diag_filter",
                "k_band_align": "k_band_align", "k_align_quad_plain": "k_align_quad_plain"}
EXTRAS_FILE = "bench_extras.json"
HEADLINE_MAX = 4096


def source_hash():
    """what the kernels were built from: the PMC summaries name the build they were collected on"""
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.startswith(".") or not os.path.isfile(os.path.join(CSRC, f)):
            continue
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def load_pmc(tag):
    """profiles/r02/pmc/<tag>.json, written by tools/pmc_summary.py from the three rocprofv3 --pmc passes of
    `bench.py --pmc-run <tag>`.  Returns (per-kernel counters, stale?) -- counters of another build are not replayed:
    the caller then reports traffic and VALU fractions as null.  A summary that lacks a kernel this run timed is an error."""
    path = os.path.join(PMC_DIR, tag + ".json")
    if not os.path.exists(path):
        return None, True
    with open(path) as f:
        d = json.load(f)
    return d, d.get("_meta", {}).get("source_hash") != source_hash()


def pmc_usable(pmc, stale, reads_per_gpu):
    return pmc is not None and not stale and pmc.get("_meta", {}).get("reads_per_gpu") == reads_per_gpu


class DevArray:
    """Wrap a raw device pointer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


# ---- workloads (SURVEY.md section 8(d) recipe; tools/gen_data.py) -------------------------------------------------
def make_workload(cfg, n_reads, seed):
    import gen_data
    import mia_amd
    w = {"cfg": cfg, "n": n_reads}
    if cfg == 4:
        w["read_len"], w["circular"] = 150, False
        indiv = gen_data.random_reference(100_000, seed=5)
        w["ref"] = indiv
        w["ref_name"] = "100 kb synthetic region (seed 5), linear"
    else:
        w["read_len"], w["circular"] = 100, True
        _, _, mt = gen_data.read_fasta_one(os.path.join(GOLDEN, "mt311.fa"))
        w["ref"] = mt.upper()                              # make_ref_upper (src/mia.c:642-648)
        indiv = gen_data.resolve_individual(mt)
        w["ref_name"] = "mt311 (16619 bp, circular)"
    w["plain_ref"] = indiv
    w["matrix_file"] = {1: None, 2: "ancient.submat.txt", 3: "ancient.submat.solexa.pe.txt", 4: "ancient.submat.txt"}[cfg]
    w["pssm"] = mia_amd.flat_pssm() if w["matrix_file"] is None else mia_amd.read_pssm(os.path.join(GOLDEN, w["matrix_file"]))
    L = w["read_len"]
    if cfg == 3:           # two reads per 300 +- 30 bp fragment, ids /1 and /2 (an odd share gets one single read more)
        d = gen_data.make_paired_reads(indiv, n_reads + (n_reads & 1), L, seed, damage=True)
        d = {k: d[k][:n_reads] for k in ("reads", "start", "strand")}
    elif n_reads > 2_000_000:
        d = gen_data.make_reads_chunked(indiv, n_reads, L, seed, circular=w["circular"], damage=cfg != 1)
    else:
        d = gen_data.make_reads(indiv, n_reads, L, seed, circular=w["circular"], damage=cfg != 1)
    w["stored"] = gen_data.stored_orientation(d)
    w["rc"] = d["strand"].astype(np.uint8)
    w["as_"] = d["start"].astype(np.int32)
    w["ae"] = (w["as_"] + L - 1).astype(np.int32)
    w["offsets"] = np.arange(n_reads + 1, dtype=np.int64) * L
    w["lens"] = np.full(n_reads, L, np.int32)
    w["bytes_per_read"] = bytes_per_read(L)
    return w


def cpu_baseline(w, sample_per_proc=3000, iters=2):
    """The reference's own per-iteration path (oracle/_ref/ref_iter_driver, built from /root/reference by
    oracle/Makefile.ref) on the host cores, P independent processes on disjoint samples of the same workload."""
    import gen_data
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_iter_driver")
    if not os.path.exists(drv):
        return {"value": None, "unit": "reads/s per iteration", "cores": 0, "kind": "reference", "sample": "oracle/_ref/ref_iter_driver missing"}
    P = max(1, min(os.cpu_count() or 1, 32))
    n = w["n"]
    P = max(1, min(P, n // 500))
    S = min(sample_per_proc, n // P)
    tmp = tempfile.mkdtemp()
    gen_data.write_fasta(os.path.join(tmp, "ref.fa"), "ref", w["ref"])
    matrix = os.path.join(GOLDEN, w["matrix_file"]) if w["matrix_file"] else "flat"
    procs = []
    for p in range(P):
        path = os.path.join(tmp, f"reads{p}.txt")
        with open(path, "w") as f:
            for i in range(p * S, (p + 1) * S):
                f.write(f"{int(w['rc'][i])} {int(w['as_'][i])} {int(w['ae'][i])} {w['stored'][i].tobytes().decode()}\n")
        procs.append(subprocess.Popen([drv, os.path.join(tmp, "ref.fa"), path, "1" if w["circular"] else "0", matrix, str(iters)], stdout=subprocess.PIPE))
    secs = []
    for pr in procs:
        out = pr.communicate()[0].decode().split()
        secs.append(float(out[out.index("seconds") + 1]))
    return {"value": P * S * iters / max(secs), "unit": "reads/s per iteration", "cores": P, "kind": "reference",
            "sample": f"{P} processes x {S} reads x {iters} iterations of oracle/_ref/ref_iter_driver (reference reiterate_assembly+"
                      f"pop_smp+cull+consensus, matrix {w['matrix_file'] or 'flat'}), slowest process {max(secs):.2f} s"}


def certificate(hip, cons, fixed_point=None):
    """What the timed steps computed, in 64 bytes each: sha256 of the consensus string the last step returned
    and of every read's (score, as, ae) after it -- int32, little endian, the three arrays one after the other.
    tests/test_gpu_bench_workloads.py builds THIS workload with THIS seed, checks samples and subset tallies of it against the
    oracle and pins the same digests (tests/golden/bench_certificates.json): the number and the parity evidence are about the
    same bytes."""
    import hashlib
    sc, a, e = hip.alignments()
    h = hashlib.sha256()
    for x in (sc, a, e):
        h.update(np.ascontiguousarray(x, dtype="<i4").tobytes())
    c = {"consensus_sha256": hashlib.sha256(cons.encode()).hexdigest(), "alignments_sha256": h.hexdigest(), "consensus_len": len(cons)}
    if fixed_point is not None:
        c["consensus_is_fixed_point"] = bool(fixed_point)
    return c


# ---- one iteration ------------------------------------------------------------------------------------------------
class Pipeline:
    def __init__(self, hip, w, world=1, rank=0, force_dist=False, breakdown=False, c_comm=False):
        self.hip, self.w, self.world, self.rank = hip, w, world, rank
        self.sharded = world > 1 or force_dist
        self.c_comm = c_comm                    # the library's own RCCL communicator: iterate() does the exchanges itself
        self.phase = {} if breakdown else None
        hip.set_pssm(w["pssm"])
        hip.upload_reads(w["stored"].reshape(-1), w["offsets"], w["rc"], np.ones(w["n"], np.uint8), w["as_"], w["ae"])
        if world > 1:
            hip.set_read_base(w["read_base"])       # contiguous fsdb blocks: global index of this rank's first read

    def _tick(self, name, t0):
        if self.phase is not None:
            self.hip.sync()
            self.phase[name] = self.phase.get(name, 0.0) + (time.perf_counter() - t0)
        return time.perf_counter()

    def step(self, cur_ref):
        hip, w = self.hip, self.w
        if (not self.sharded or self.c_comm) and self.phase is None and not os.environ.get("MIA_BENCH_STEPWISE"):
            # the product's own iteration call (mia_hip_iterate): planner, cut line and insert-event count stay on the device
            return hip.iterate(cur_ref, w["circular"])
        t0 = time.perf_counter()
        hip.realign(cur_ref, w["circular"])
        t0 = self._tick("realign", t0)
        slot_base = 0
        # find_fsdb_score_cut: pass 1 (integer sums, length range) on the device; with equally long reads that is the
        # whole regression.  Only reads of different lengths need the sequential double sums over the scores on the host.
        sums = hip.score_sums()
        if self.sharded:
            import torch
            from mia_amd import dist as mdist
            n_rec, n_lnk = hip.pre_cull_counts()           # by-products of score_sums: no further round trip
            sums, slot_base, link_counts = mdist.gather_pre_cull(sums, n_rec, "cuda", n_lnk)
        t0 = self._tick("score_sums", t0)
        cut = hip.score_cut_from_sums(sums)
        if cut is None:
            score = hip.scores()
            if self.sharded:
                # the regression runs over ALL reads in fsdb order (src/fsdb.c:269-383): scores and lengths in rank order
                all_scores = mdist.all_gather_concat(torch.from_numpy(score).cuda()).cpu().numpy()
                all_lens = mdist.all_gather_concat(torch.from_numpy(w["lens"]).cuda()).cpu().numpy()
                cut = hip.score_cut(all_scores, all_lens)
            else:
                cut = hip.score_cut(score, w["lens"])
        slope, intercept = cut
        t0 = self._tick("score_cut", t0)
        if slope <= 0:
            slope = 100.0
        hip.cull(0, slope, intercept, slot_base)
        if self.sharded:
            mdist.exchange_links(hip, lambda ptr, n, ts: torch.as_tensor(DevArray(ptr, n, ts), device="cuda"), link_counts)
        t0 = self._tick("cull", t0)
        hip.tally()
        t0 = self._tick("tally", t0)
        if self.sharded:
            pt, nt, pg, ng = hip.tally_buffers()
            pe, ne = hip.ins_events()
            counts = mdist.allreduce_tallies_with_counts(torch.as_tensor(DevArray(pt, nt, "<i4"), device="cuda"),
                                                         torch.as_tensor(DevArray(pg, ng, "<i4"), device="cuda"), ne)
            mine = torch.as_tensor(DevArray(pe, ne, "<i8"), device="cuda") if ne else torch.zeros(0, dtype=torch.int64, device="cuda")
            allev = mdist.all_gather_ragged(mine, counts)
            torch.cuda.synchronize()
            hip.set_ins_events(allev.data_ptr() if allev.numel() else 0, int(allev.numel()))
        c = hip.consensus(1)
        self._tick("consensus", t0)
        return c

    def reset_stats(self):
        h = self.hip
        h.stage_stats(reset=True)
        h.plain_stats(reset=True)
        h.filter_stats(reset=True)
        h.band_stats(reset=True)
        h.bx_stats(reset=True)

    def stages(self, steps, peaks, pmc, pmc_stale):
        """Per kernel of the realignment and the tally: HIP-event time over the steps just run, the reads it worked on,
        algorithmic GB/s against the HBM roof, cells actually evaluated, and -- from the PMC summary of THIS build, if
        there is one -- memory-side traffic and VALU issue against the measured issue peak."""
        h, w = self.hip, self.w
        st = h.stage_stats()
        (seen, by_plan, by_values, by_trace), _, bx_launches = h.bx_stats()
        f_seen, f_done, _, f_launches = h.filter_stats()
        band_done, _, _ = h.band_stats()
        _, _, plain_in, plain_retried = h.plain_stats()
        ctr = h.bx_counters()                      # of the last realign: list lengths by band class
        widths = (8, 16, 24, 32, 64)                # band classes (csrc/bandx_body.h: BX_NCLS); counters: values lists, trace lists
        n, L2 = w["n"], w["read_len"]
        win = L2 + 100                             # +-50 window (src/mia_main.c:190-212)
        per_launch = {
            "k_diag_filter": (n if f_launches else 0, None),
            "k_band_align": ((f_seen - f_done) / max(st["k_band_align"][1], 1), 32 * L2),
            "k_bx_plan": (seen / max(bx_launches, 1), None),
            "k_bx_values": (sum(ctr[0:5]), sum(c * wd for c, wd in zip(ctr[0:5], widths)) * L2),
            "k_bx_trace": (sum(ctr[5:10]), sum(c * wd for c, wd in zip(ctr[5:10], widths)) * L2),
            "k_align_quad_plain": (plain_in / max(st["k_align_quad_plain"][1], 1), win * L2),
            "k_tally_binned": (n, None),
        }
        left = f_seen - f_done - band_done - by_values - by_trace if (f_launches or bx_launches) else n * steps
        per_launch["k_align_quad"] = ((plain_retried if st["k_align_quad_plain"][1] else left) / max(st["k_align_quad"][1], 1), win * L2)
        out = []
        for name in STAGES:
            ms, k = st[name]
            if k <= 0 or ms <= 0:
                continue
            if name in STAGE_KERNELS:
                k = steps                          # several launches per step (quad, window classes, retries; values DP + late trace): one figure per step
            reads, cells = per_launch[name]
            if name == "k_align_quad":
                reads = reads * st[name][1] / steps
            if cells is not None and name in ("k_band_align", "k_align_quad_plain", "k_align_quad"):
                cells = cells * reads              # per read -> per launch
            k_ms = ms / k
            ach = w["bytes_per_read"] * reads / (k_ms * 1e-3) / 1e9
            s = {"kernel": name, "kernel_ms": k_ms, "ms_per_step": ms / steps, "launches": k, "reads_per_launch": reads,
                 "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                 "gcups": cells / (k_ms * 1e-3) / 1e9 if cells else None, "traffic": None, "valu_frac": None}
            if pmc_usable(pmc, pmc_stale, n):
                kept = pmc["_meta"].get("steps_kept", 4)
                tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "SQ_INSTS_VALU": 0.0}
                # (a kernel the counter passes never met -- the route a few open reads take depends on last iteration's reject count, which
                # sits at its threshold for configs[4] and differs with the seed: no counter figures for that stage, and the line says so)
                missing = [kn for kn in STAGE_KERNELS.get(name, [name]) if kn not in pmc or not all(c in pmc[kn] for c in tot)]
                if missing:
                    s["pmc_missing"] = missing
                    out.append(s)
                    continue
                for kn in STAGE_KERNELS.get(name, [name]):
                    for c in tot:                   # average per dispatch x dispatches per step
                        per_step = pmc[kn][c] * pmc[kn]["dispatches_" + c] / kept
                        tot[c] += per_step if name in STAGE_KERNELS else pmc[kn][c]
                # MI355X_MICROARCH.md, HBM: FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE counts 128-byte read
                # requests as 64 bytes for wide streaming reads -> doubled (calibrated on k_peak_copy in the same passes,
                # see _meta.fetch_calibration); per launch, like `achieved`
                s["traffic"] = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024
                if peaks and peaks.get("valu_ginst_s"):
                    s["valu_insts"] = tot["SQ_INSTS_VALU"]
                    s["valu_frac"] = tot["SQ_INSTS_VALU"] / (k_ms * 1e-3) / (peaks["valu_ginst_s"] * 1e9)
            out.append(s)
        return out, {"reads_seen": seen or f_seen, "finished_by_plan_or_filter": f_done, "finished_by_values_dp": by_values,
                     "finished_by_trace_dp": by_trace + band_done, "to_full_window_kernels": max(left, 0)}


def use_timed_region(stages, dominant, dom_ms, dom_launches, steps):
    """Replace the dominant stage's duration by the one measured inside the timed region (only that stage carried event
    pairs there) and rescale the figures derived from it.  A grouped stage (STAGE_KERNELS) stays one figure per step."""
    if not dominant or not dom_launches:
        return
    for sg in stages:
        if sg["kernel"] != dominant:
            continue
        per = steps if dominant in STAGE_KERNELS else dom_launches
        scale = (dom_ms / per) / sg["kernel_ms"]
        sg["kernel_ms_instrumented_steps"] = sg["kernel_ms"]
        sg["kernel_ms"], sg["ms_per_step"], sg["launches"], sg["timed_region"] = dom_ms / per, dom_ms / steps, per, True
        if dominant in STAGE_KERNELS:
            sg["device_launches"] = dom_launches
        for key in ("achieved", "frac", "gcups", "valu_frac"):
            if sg.get(key) is not None:
                sg[key] = sg[key] / scale


def dp_phase(stages, peaks):
    """The DP kernels of one step run SIDE BY SIDE on three streams (values DP and the trace launch behind it; the trace DP of
    the plan's own lists; planner + full-window kernels), so one kernel's instructions over its own elapsed time says little
    about the chip: here all their VALU instructions (PMC summary of this build) over the wall time of the phase -- the HIP
    events around values DP + late trace on the context's stream, which is the critical path the others hide behind."""
    by = {s["kernel"]: s for s in stages}
    v = by.get("k_bx_values")
    names = [k for k in ("k_bx_values", "k_bx_trace", "k_align_quad", "k_align_quad_plain") if k in by]
    if not v or any(by[k].get("valu_insts") is None for k in names) or not (peaks and peaks.get("valu_ginst_s")):
        return None
    insts = sum(by[k]["valu_insts"] * (1 if k in STAGE_KERNELS else by[k]["launches"] / max(v["launches"], 1)) for k in names)
    ms = v["kernel_ms"]
    cells = sum((by[k]["gcups"] or 0) * by[k]["kernel_ms"] * 1e6 * (1 if k in STAGE_KERNELS else by[k]["launches"] / max(v["launches"], 1)) for k in names)
    return {"kernels": names, "wall_ms": ms, "valu_insts": insts, "achieved": insts / (ms * 1e-3) / 1e9, "peak": peaks["valu_ginst_s"],
            "unit": "1e9 wave64 instructions/s", "frac": insts / (ms * 1e-3) / (peaks["valu_ginst_s"] * 1e9),
            "gcups": cells / (ms * 1e-3) / 1e9 if cells else None}


def step_traffic(pmc, pmc_stale, reads_per_gpu, bytes_per_read_):
    """memory-side bytes of ONE whole step (every kernel of the PMC summary of this build: FETCH_SIZE doubled as the guide's gfx950 correction
    prescribes, + WRITE_SIZE) over the step's algorithmic bytes; None when the committed summary is from another build"""
    if not pmc_usable(pmc, pmc_stale, reads_per_gpu):
        return None
    kept = pmc["_meta"].get("steps_kept", 4)
    tot = 0.0
    for kn, c in pmc.items():
        if kn.startswith("_") or "k_peak" in kn or not all(k in c for k in ("FETCH_SIZE", "WRITE_SIZE", "dispatches_FETCH_SIZE", "dispatches_WRITE_SIZE")):
            continue
        if c["dispatches_FETCH_SIZE"] < kept:       # not a kernel of the step: set-up (k_bx_umax, k_read_planes) or the warm-up's first iteration
            continue
        tot += (2.0 * c["FETCH_SIZE"] * c["dispatches_FETCH_SIZE"] + c["WRITE_SIZE"] * c["dispatches_WRITE_SIZE"]) / kept * 1024
    algo = float(reads_per_gpu) * bytes_per_read_
    return {"bytes_per_step": tot, "algorithmic_bytes_per_step": algo, "ratio": tot / algo if algo else None}


def roofline(stages, peaks, pmc_tag, pmc_stale):
    timed = [s for s in stages if s.get("timed_region")]
    single = [s for s in stages if s["kernel"] in ROCPROF_NAME]
    dom = timed[0] if timed else max(single or stages, key=lambda s: s["ms_per_step"])
    r = {"bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": dom["traffic"],
         "kernel": ROCPROF_NAME.get(dom["kernel"], dom["kernel"]), "stage": dom["kernel"],
         "kernel_ms": dom["kernel_ms"], "launches": dom["launches"], "reads_per_launch": dom["reads_per_launch"],

         "primary_bound": "valu",
         "valu": {"bound": "valu", "achieved": (dom["valu_insts"] / (dom["kernel_ms"] * 1e-3) / 1e9) if dom.get("valu_insts") else None,
                  "peak": peaks.get("valu_ginst_s") if peaks else None, "unit": "1e9 wave64 instructions/s", "frac": dom["valu_frac"]},
         "peak_measured_copy": peaks.get("hbm_copy_gbs") if peaks else None,
         "frac_of_measured_copy": dom["achieved"] / peaks["hbm_copy_gbs"] if peaks and peaks.get("hbm_copy_gbs") else None,
         "pmc": {"summary": f"profiles/{PROFILE_ROUND}/pmc/{pmc_tag}.json", "from_this_build": not pmc_stale},
         "dp_phase": dp_phase(stages, peaks),
         "groups": [s for s in stages if s["kernel"] in STAGE_KERNELS and s["kernel"] not in ROCPROF_NAME],
         "largest_stage": (lambda m: {"stage": m["kernel"], "ms_per_step": m["ms_per_step"], "kernels": STAGE_KERNELS.get(m["kernel"], [m["kernel"]])})(
             max(stages, key=lambda s: s["ms_per_step"])),
         "stages": stages,
         "note": "integer DP: the kernels are bound by VALU issue, not HBM -- `frac` prices the SURVEY 8(d) algorithmic bytes of the "
                 "largest SINGLE kernel (a name of the rocprofv3 trace; `largest_stage` names the largest stage or group of kernels beside it) "
                 "against the nominal 8 TB/s as the contract asks; `valu.frac` = its VALU "
                 "instructions (rocprofv3 SQ_INSTS_VALU of this build) over its live HIP-event time, against the issue rate "
                 "measured by k_peak_valu in this run; dp_phase = the same for all DP kernels of the step together, which share the chip "
                 "on three streams (their instructions over the phase's wall time); traffic/valu are null when the committed PMC "
                 "summary is from another build"}
    return r


def run_steps(pipe, cur, steps):
    for _ in range(steps):
        cur = pipe.step(cur)
    pipe.hip.sync()
    return cur


def section_loopback(hip_mod, device, w, W, K=6):
    """The sharded iteration (mia_hip_iterate over a communicator: pre-cull all-gather, tally all-reduce, gaps max-reduce, insert
    events) with W ranks on ONE GPU through the library's in-process loopback transport: W contexts of n / W reads each, one host
    thread each.  The GPU does all W shards' work, so this is no speed-up figure -- it is the sharded code path timed at full
    size, and its difference to one context with all n reads bounds what W ranks' exchanges and extra launches cost."""
    import threading
    n, L = w["n"], w["read_len"]
    cuts = [n * k // W for k in range(W + 1)]
    parts = []
    for k in range(W):
        lo, hi = cuts[k], cuts[k + 1]
        h = hip_mod.MiaHip(device)
        h.set_pssm(w["pssm"])
        h.upload_reads(w["stored"][lo:hi].reshape(-1), np.arange(hi - lo + 1, dtype=np.int64) * L, w["rc"][lo:hi], np.ones(hi - lo, np.uint8),
                       w["as_"][lo:hi], w["ae"][lo:hi])
        h.set_read_base(lo)
        parts.append(h)
    grp = hip_mod.LoopbackGroup(W)
    for k, h in enumerate(parts):
        grp.attach(h, k)
    plan = [w["ref"]] * 1 + [None] * 3 + [None] * K + [w["ref"]] * 2        # None: the consensus of the step before
    bar = threading.Barrier(W + 1)
    cons, errs = [None] * W, [None] * W

    def work(k):
        cur = w["ref"]
        try:
            for ref in plan:
                bar.wait()
                cur = parts[k].iterate(ref if ref is not None else cur, w["circular"])
                parts[k].sync()
                bar.wait()
            cons[k] = cur
        except Exception as ex:      # noqa: BLE001
            errs[k] = repr(ex)
            bar.abort()

    th = [threading.Thread(target=work, args=(k,)) for k in range(W)]
    for t in th:
        t.start()
    ms = []
    try:
        for _ in plan:
            bar.wait()
            t0 = time.perf_counter()
            bar.wait()
            ms.append((time.perf_counter() - t0) * 1e3)
    except threading.BrokenBarrierError:
        pass
    for t in th:
        t.join()
    for h in parts:
        h.comm_destroy()
    grp.close()
    for h in parts:
        h.close()
    if any(errs) or len(ms) < len(plan):
        return {"error": [e for e in errs if e]}
    steady = sorted(ms[4:4 + K])
    return {"ranks": W, "reads_per_rank": n // W, "transport": "loopback (one GPU, W contexts, host barriers + device copies)",
            "steady_ms_per_step": steady[len(steady) // 2], "steady_ms_min": steady[0], "steady_ms_max": steady[-1],
            "first_iteration_ms": min(ms[-2:]), "all_ranks_same_consensus": len(set(cons)) == 1}


def section_converge(hip_mod, device, cfg, n, seed, peaks, no_cpu, max_iters=12, w=None):
    """configs[2] / configs[4]: from the starting reference to convergence, every iteration timed on its own (the first
    one runs against the reference itself -- mt311 carries an ambiguity code in every tenth column --, the later ones
    against consensus sequences); then the converged state is stepped a few times for the stage table."""
    w = w or make_workload(cfg, n, seed)
    hip = hip_mod.MiaHip(device)
    pipe = Pipeline(hip, w)
    cur, it_ms, rounds = w["ref"], [], 0
    while rounds < max_iters:
        hip.sync()
        t0 = time.perf_counter()
        nxt = pipe.step(cur)
        hip.sync()
        it_ms.append((time.perf_counter() - t0) * 1e3)
        rounds += 1
        if nxt == cur:                              # src/mia_main.c:905-940: stop when the consensus repeats
            break
        cur = nxt
    converged = nxt == cur
    K = 6
    pipe.reset_stats()
    cur = run_steps(pipe, cur, 2)                   # which kernel takes the most time in the converged state
    st = hip.stage_stats()
    dominant = max((k for k in STAGES if k in ROCPROF_NAME), key=lambda k: st[k][0])
    hip.set_timed_stages([dominant])
    pipe.reset_stats()
    hip.sync()
    t0 = time.perf_counter()
    cur = run_steps(pipe, cur, K)
    steady_ms = (time.perf_counter() - t0) * 1e3 / K
    dom_ms, dom_launches = hip.stage_stats()[dominant]
    hip.set_timed_stages(None)
    pipe.reset_stats()
    before = cur
    cur = run_steps(pipe, cur, K)                   # the stage table: every stage timed
    # the converged state's digest (taken here: the iteration against the starting reference further down moves every read's window)
    cert = dict(certificate(hip, cur, fixed_point=converged and cur == before), workload="make_workload(%d, %d, seed=%d)" % (cfg, n, seed))
    tag = f"cfg{cfg}"
    pmc, stale = load_pmc(tag)
    stages, counts = pipe.stages(K, peaks, pmc, stale)
    use_timed_region(stages, dominant, dom_ms, dom_launches, K)
    # the first iteration once more, now that every buffer exists: the starting reference (mt311: an ambiguity code in
    # every tenth column) against reads whose coordinates are where the first pass would have put them
    pipe.reset_stats()
    hip.sync()
    t0 = time.perf_counter()
    pipe.step(w["ref"])
    hip.sync()
    first_warm_ms = (time.perf_counter() - t0) * 1e3
    _, first_counts = pipe.stages(1, None, None, True)
    out = {"reads": n, "workload": f"configs[{cfg}]: {n} synthetic {w['read_len']} bp {'aDNA-damaged ' if cfg != 1 else ''}reads vs {w['ref_name']}, matrix {w['matrix_file'] or 'flat'}; "
                       "pass-1 coordinates = true positions",
           "iterations_to_convergence": rounds, "converged": converged, "ms_per_iteration": it_ms,
           "reads_per_s_per_iteration": n * rounds / (sum(it_ms) * 1e-3),
           "steady_state_ms_per_iteration": steady_ms, "steady_state_reads_per_s": n / (steady_ms * 1e-3),
           "first_iteration_again_ms": first_warm_ms, "first_iteration_over_steady": first_warm_ms / steady_ms,
           "first_iteration_read_fate": first_counts, "certificate": cert,
           "bytes_per_read": w["bytes_per_read"], "consensus_len": len(cur), "read_fate": counts,
           "roofline": roofline(stages, peaks, tag, stale), "step_traffic": step_traffic(pmc, stale, n, w["bytes_per_read"])}
    if not no_cpu:
        out["cpu_baseline"] = cpu_baseline(w, sample_per_proc=3000 if cfg != 4 else 1500)
    hip.close()
    return out


def section_myers(hip, no_cpu=False):
    """myers_diff (reference src/myers_align.c:10-99) as a batch: 100 000 pairs of 100-300 characters with ~3 % edits (what a
    read-against-read use holds; one pair per lane), and the ccheck-sized pair (16.6 kb, maxd = len/10 as src/ccheck.cc:477;
    one pair per wavefront).  Kernel figures from HIP events (mia_hip_myers_time), wall figures including the packing of the
    strings on the host and the copies; the reference's own myers_diff on one host core beside them."""
    rng = np.random.default_rng(7)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    A, B = [], []
    for _ in range(100_000):
        n = int(rng.integers(100, 301))
        a = bases[rng.integers(0, 4, n)].copy()
        b = a.copy()
        for p in rng.integers(0, n, max(1, n // 33)):
            b[p] = bases[rng.integers(0, 4)]
        A.append(a.tobytes())
        B.append(b.tobytes())
    mode = np.zeros(len(A), np.int32)
    maxd = np.full(len(A), 64, np.int32)
    hip.myers(A[:64], B[:64], mode[:64], maxd[:64])
    t0 = time.perf_counter()
    d = hip.myers(A, B, mode, maxd)
    dt = time.perf_counter() - t0
    k_ms = hip.myers_time()
    call_s = hip.myers_call_s
    cells = sum(len(a) * len(b) for a, b in zip(A, B))
    # algorithmic bytes: both sequences as 4-bit codes in, one distance out
    algo_bytes = sum((len(a) + len(b) + 1) // 2 + 4 for a, b in zip(A, B))
    # the pre-packed batch form (mia_hip_myers_packed: no strlen, no packing inside the call)
    import mia_amd as _m
    packed = _m.pack_myers_pairs(A, B)
    hip.myers_packed(packed, mode, maxd)                                   # warm-up
    d_packed = hip.myers_packed(packed, mode, maxd)
    packed_call_s = hip.myers_call_s
    packed_kernel_ms = hip.myers_time()
    big = bases[rng.integers(0, 4, 16_600)].copy()
    big2 = big.copy()
    for p in rng.integers(0, len(big), 160):
        big2[p] = bases[rng.integers(0, 4)]
    hip.myers([big.tobytes()], [big2.tobytes()], np.zeros(1, np.int32), np.full(1, 1660, np.int32))      # warm-up (LDS attribute, pool)
    t1 = time.perf_counter()
    dbig = hip.myers([big.tobytes()], [big2.tobytes()], np.zeros(1, np.int32), np.full(1, 1660, np.int32))
    dt_big = time.perf_counter() - t1
    big_kernel_ms = hip.myers_time()
    # the call ccheck makes (src/ccheck.cc:477-480): distance AND both rows (mia_hip_myers_align: D-path table from the device,
    # walked back on the host)
    hip.myers_align(big.tobytes(), 0, big2.tobytes(), 1660)
    t1 = time.perf_counter()
    d_al, ra_al, _rb_al = hip.myers_align(big.tobytes(), 0, big2.tobytes(), 1660)
    dt_align = time.perf_counter() - t1
    align_kernel_ms = hip.myers_time()
    out = {"pairs": len(A), "pairs_per_s": len(A) / dt, "gcups": cells / dt / 1e9, "mean_distance": float(d[d != 0xFFFFFFFF].mean()),
           "c_abi_call_pairs_per_s": len(A) / call_s, "c_abi_call_gcups": cells / call_s / 1e9,
           "kernel_ms": k_ms, "kernel_pairs_per_s": len(A) / (k_ms * 1e-3), "kernel_gcups": cells / (k_ms * 1e-3) / 1e9,
           "packed_c_abi_call_pairs_per_s": len(A) / packed_call_s, "packed_kernel_ms": packed_kernel_ms, "packed_equal": bool(np.array_equal(d, d_packed)),
           "roofline": {"bound": "hbm", "achieved": algo_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "note": "bit-vector DP: 64 cells per 64-bit operation, bound by integer issue; bytes = packed sequences in + distances out"},
           "pair_16k6_ms": dt_big * 1e3, "pair_16k6_kernel_ms": big_kernel_ms, "pair_16k6_distance": int(dbig[0]),
           "pair_16k6_align_ms": dt_align * 1e3, "pair_16k6_align_kernel_ms": align_kernel_ms, "pair_16k6_align_distance": d_al,
           "note": "pairs_per_s: the Python call (list -> char** included); c_abi_call: mia_hip_myers alone (strlen, packing on the host threads, "
                   "PCIe, kernel); kernel: HIP events.  k_myers_lanes: one pair per lane, "
                   "k_myers: one pair per wavefront, 64-bit lanes; pair_16k6: ccheck's one call (16.6 kb, maxd 1660) through k_myers_ond "
                   "(furthest-reaching D-paths, one row of diagonals per step over 256 threads); _align_: with both rows"}
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_myers_driver")
    if not no_cpu and os.path.exists(drv):
        m = 4000
        inp = "".join("0 64 %s %s\n" % (a.decode(), b.decode()) for a, b in zip(A[:m], B[:m]))
        t2 = time.perf_counter()
        r = subprocess.run([drv], input=inp.encode(), stdout=subprocess.PIPE)
        cpu_dt = time.perf_counter() - t2
        ok = [int(x.split()[0]) for x in r.stdout.decode().strip().split("\n")]
        out["cpu_baseline"] = {"value": m / cpu_dt, "unit": "pairs/s", "cores": 1, "kind": "reference",
                               "sample": f"{m} of the same pairs through oracle/_ref/ref_myers_driver (the reference's myers_diff with backtrace), "
                                         f"{cpu_dt:.2f} s; distances equal: {ok == [int(x) for x in d[:m]]}"}
        # the 16.6 kb pair on one host core: the reference's myers_diff with backtrace, R calls in one process (start-up divided out
        # by a second run of one call)
        R = 20
        one = "0 1660 %s %s\n" % (big.tobytes().decode(), big2.tobytes().decode())
        t2 = time.perf_counter(); r1 = subprocess.run([drv], input=one.encode(), stdout=subprocess.PIPE); t_one = time.perf_counter() - t2
        t2 = time.perf_counter(); rR = subprocess.run([drv], input=(one * (R + 1)).encode(), stdout=subprocess.PIPE); t_R = time.perf_counter() - t2
        first = rR.stdout.decode().split("\n")[0].split(" ")
        out["pair_16k6_cpu_reference_ms"] = max(t_R - t_one, 0.0) / R * 1e3
        out["pair_16k6_cpu_reference_equal"] = bool(int(first[0]) == int(dbig[0]) == d_al and first[1] == ra_al)
    return out


def section_cli(w, n=1_000_000):
    """The host program end to end (mia_hip: FASTA in, .maln out) on the headline workload: ingest (host/ingest.h: the
    reference's reader on all host threads, beside the GPU start-up), pass 1 without a k-mer mask, read store, iterations
    to convergence, .maln files -- wall time of the process and the phases it reports itself (MIA_HIP_TIMING=1)."""
    import re
    exe = os.path.join(ROOT, "mapping-iterative-assembler_amd", "mia_hip")
    if not os.path.exists(exe):
        return {"error": "mia_hip not built"}
    import gen_data
    m = min(n, w["n"])
    tmp = tempfile.mkdtemp()
    stored, rc = w["stored"][:m], w["rc"][:m]
    seq = np.where(rc[:, None] == 1, gen_data._COMP[stored[:, ::-1]], stored).astype(np.uint8)       # as sequenced
    L = seq.shape[1]
    rec = np.empty((m, 10 + L + 1), np.uint8)                                                         # ">r0000000\n" + bases + "\n"
    rec[:, 0], rec[:, 1], rec[:, 9], rec[:, -1] = ord(">"), ord("r"), ord("\n"), ord("\n")
    idx = np.arange(m)
    for k in range(7):
        rec[:, 8 - k] = ord("0") + (idx // 10 ** k) % 10
    rec[:, 10:10 + L] = seq
    fa = os.path.join(tmp, "reads.fa")
    rec.tofile(fa)
    gen_data.write_fasta(os.path.join(tmp, "ref.fa"), "ref", w["ref"])
    out = {"reads": m, "input_bytes": int(rec.size)}
    for label, extra in (("every_iteration_written", []), ("final_maln_only", ["-F"])):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-r", os.path.join(tmp, "ref.fa"), "-f", fa, "-c", "-i", "-m", os.path.join(tmp, "out_" + label)] + extra,
                           env=dict(os.environ, MIA_HIP_TIMING="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.perf_counter() - t0
        phases = {}
        for mm in re.finditer(r"\[mia_hip timing\]\s+(.*?)\s+([0-9.]+) ms", r.stderr.decode(errors="replace")):
            phases.setdefault(mm.group(1).strip(), []).append(float(mm.group(2)))
        files = [f for f in os.listdir(tmp) if f.startswith("out_" + label)]
        out[label] = {"wall_s": wall, "exit": r.returncode, "maln_files": len(files), "maln_bytes": sum(os.path.getsize(os.path.join(tmp, f)) for f in files),
                      "phases_ms": {k: (v if len(v) > 1 else v[0]) for k, v in phases.items()}}
        if "read input alone" in phases:
            out["ingest_reads_per_s"] = m / (phases["read input alone"][0] * 1e-3)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out


def write_extras(out):
    """Everything the run measured (stage tables, configs2/3/4 sections, pass 1, Myers, CLI, peaks, notes) goes to
    bench_extras.json beside bench.py -- and to gpurun_out/ when that exists, so it comes back from a GPU box; the one
    stdout line is the compact headline (a 21 KB line was more than the driver parses)."""
    path = os.path.join(ROOT, EXTRAS_FILE)
    for p in (path, os.path.join(ROOT, "gpurun_out", EXTRAS_FILE)):
        if os.path.isdir(os.path.dirname(p)):
            try:
                with open(p, "w") as f:
                    json.dump(out, f, indent=1)
            except OSError as e:                       # (a read-only tree must not cost the run its line)
                sys.stderr.write("bench.py: %s not written: %s\n" % (p, e))
    return EXTRAS_FILE


def _r(x, nd=4):
    """shorten floats for the headline: 4 significant decimals are more than the measurement holds"""
    if isinstance(x, float):
        return float("%.*g" % (nd + 2, x))
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, list):
        return [_r(v, nd) for v in x]
    return x


def headline(out, extras_path):
    """The driver's line: the contract's keys, the roofline of the dominant KERNEL (a name of the rocprofv3 kernel trace),
    the CPU baseline, and the product-speed figures (first iteration, to convergence); the rest is in `extras`."""
    ro = out["roofline"]
    h = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "blocks", "higher_is_better", "scaling", "vs_baseline",
                             "dtype", "data", "config") if k in out}
    rl = {k: ro.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "launches", "reads_per_launch",
                                 "primary_bound", "peak_measured_copy", "frac_of_measured_copy")}
    rl["bytes_per_read"] = out["config"].get("bytes_per_read")
    if ro.get("valu"):
        rl["valu"] = {k: ro["valu"].get(k) for k in ("achieved", "peak", "unit", "frac")}
    if ro.get("dp_phase"):
        rl["dp_phase"] = {k: ro["dp_phase"].get(k) for k in ("wall_ms", "frac", "gcups")}
    rl["pmc"] = ro.get("pmc")
    h["roofline"] = rl
    if "cpu_baseline" in out:
        cb = dict(out["cpu_baseline"])
        cb["sample"] = str(cb.get("sample", ""))[:200]
        h["cpu_baseline"] = cb
    if "certificate" in out:
        h["certificate"] = {k: out["certificate"].get(k) for k in ("consensus_sha256", "alignments_sha256", "consensus_is_fixed_point", "workload")}
    for k in ("value_first_iteration", "collectives", "communicator", "collectives_note", "library", "step_traffic"):
        if k in out:
            h[k] = out[k]
    if "first_iteration" in out:
        h["first_iteration_ms"] = out["first_iteration"]["ms"]
        h["first_iteration_over_steady"] = out["first_iteration"]["over_steady"]
    conv = {}
    for k in ("configs1", "configs2", "configs3", "configs3_share", "configs4", "configs4_share"):
        c = out.get(k)
        if c:
            # (what the line has room for; first_over_steady, steady reads/s, the section's roofline and CPU figure: bench_extras.json)
            conv[k] = {"reads": c.get("reads"), "iterations": c["iterations_to_convergence"], "value_to_convergence": c["reads_per_s_per_iteration"],
                       "steady_ms": c["steady_state_ms_per_iteration"], "first_ms": c["first_iteration_again_ms"],
                       "traffic_over_algorithmic": (c.get("step_traffic") or {}).get("ratio"),
                       "consensus_sha256": (c.get("certificate") or {}).get("consensus_sha256", "")[:12]}
    lb = out.get("configs3_loopback_w8")
    if lb and "error" not in lb:
        conv["configs3_loopback_w8"] = {k: lb[k] for k in ("ranks", "reads_per_rank", "steady_ms_per_step", "first_iteration_ms")}
    if conv:
        h["to_convergence"] = conv
        h["value_to_convergence"] = conv.get("configs2", {}).get("value_to_convergence")
    if "weak" in out:
        h["weak"] = {k: out["weak"].get(k) for k in ("value", "ms_per_step", "reads_per_gpu", "total_reads")}
    h["extras"] = extras_path
    return _r(h)


def self_launch(a):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks here -- BEFORE anything touches a GPU (a
    child process per rank through torch.distributed.run) -- pass their output through and leave with their exit code.
    Fewer than N GPUs on this box: say so and fail; nothing ever reports a rank count it did not run."""
    import socket
    import torch                                     # (device_count does not initialise the GPU)
    have = torch.cuda.device_count()
    if have < a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} asked for, this box shows {have} GPU(s): not run (no line is printed for a rank count that did not run)\n")
        sys.exit(3)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def timed_job(a, env, cfg, reads, scaling, peaks=None):
    """One sharded (or single-GPU) job: workload, context, communicator, warm-up, then EXACTLY a.steps iterations between
    barriers; the longest rank counts.  Returns what the JSON line is made of, plus the live objects for the stage table."""
    import mia_amd
    torch, dist, rank, world, local, force_dist = env["torch"], env["dist"], env["rank"], env["world"], env["local"], env["force_dist"]
    n = reads if scaling == "weak" else (reads // world + (1 if rank < reads % world else 0))
    w = make_workload(cfg, n, seed=1 + rank)
    w["read_base"] = rank * n if scaling == "weak" else rank * (reads // world) + min(rank, reads % world)
    hip = mia_amd.MiaHip(local)
    # several GPUs: the exchanges of a sharded iteration run inside the library (RCCL communicator made from an id that rank 0
    # hands out through torch.distributed); --coll torch keeps them in Python over torch.distributed (dist.py) instead
    c_comm, coll_note, comm_info = False, None, None
    if (world > 1 or force_dist) and a.coll == "rccl":
        # no silent fallback: if the library's communicator cannot be made the run ends non-zero -- the
        # Python path over torch.distributed runs only when asked for by name (--coll torch)
        try:
            box = [mia_amd.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            hip.comm_init(box[0], world, rank)
            c_comm = True
            comm_info = hip.comm_info()                 # what RCCL itself says: (ncclCommCount, ncclCommUserRank, "rccl")
        except Exception as e:              # noqa: BLE001
            sys.stderr.write("bench.py: rank %d: the library's RCCL communicator failed (%s); not falling back -- run with "
                             "--coll torch to time the torch.distributed path instead\n" % (rank, e))
            sys.stderr.flush()
            os._exit(4)                     # (the other ranks sit in a collective: the launcher takes them down with this one)
        if comm_info and comm_info[0] != world:
            sys.stderr.write("bench.py: communicator reports %d ranks, launched %d\n" % (comm_info[0], world))
            os._exit(4)
    pipe = Pipeline(hip, w, world, rank, force_dist, breakdown=bool(os.environ.get("MIA_BENCH_BREAKDOWN")), c_comm=c_comm)
    cur = w["ref"]
    dominant = None
    for k in range(a.warmup):
        if k == a.warmup - 1 and a.warmup >= 2:
            pipe.reset_stats()                      # the last warm-up step alone says which kernel takes the most time
        cur = pipe.step(cur)
    if a.warmup >= 2:
        st = hip.stage_stats()
        dominant = max((k for k in STAGES if k in ROCPROF_NAME), key=lambda k: st[k][0])
        # HIP events cost the stream a few microseconds each: over the timed region only the dominant kernel carries them;
        # the other stages are timed in as many instrumented steps after it
        hip.set_timed_stages([dominant])
    # one block of a.steps iterations is 16 ms of GPU time -- a.blocks such blocks, each timed EXACTLY as the
    # contract says (barrier + synchronize on both sides, the longest rank counts); `value` is the MEDIAN block, min / max ride along
    block_dt, dom_ms, dom_launches = [], 0.0, 0
    step_in = cur
    for _blk in range(max(1, a.blocks)):
        pipe.reset_stats()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step_in = cur
            cur = pipe.step(cur)
        hip.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        block_dt.append(dt)
        if dominant:
            dm, dl = hip.stage_stats()[dominant]
            dom_ms, dom_launches = dom_ms + dm, dom_launches + dl
    if dominant and len(block_dt) > 1:             # the dominant kernel's events: the average over all blocks, as launches of ONE block
        dom_ms, dom_launches = dom_ms / len(block_dt), dom_launches // len(block_dt)
    dt = sorted(block_dt)[len(block_dt) // 2]
    cert = certificate(hip, cur, fixed_point=(cur == step_in) if a.steps > 0 else None)      # (of this rank's reads; the consensus is every rank's)
    if world > 1:
        tot = torch.tensor([n], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_reads = int(tot.item())
    else:
        total_reads = n
    if dominant:                                    # the stage table: the same steps again, every stage timed
        hip.set_timed_stages(None)
        pipe.reset_stats()
        cur = run_steps(pipe, cur, a.steps)
    job = {"value": total_reads * a.steps / dt, "ms_per_step": dt / a.steps * 1e3, "scaling": scaling, "reads_per_gpu": n, "total_reads": total_reads,
           "blocks": {"n": len(block_dt), "steps_each": a.steps, "ms_per_step_median": dt / a.steps * 1e3, "ms_per_step_min": min(block_dt) / a.steps * 1e3,
                      "ms_per_step_max": max(block_dt) / a.steps * 1e3,
                      # (the contract's single timed region is the FIRST block; the median is what `value` reports)
                      "ms_per_step_first_block": block_dt[0] / a.steps * 1e3, "value_first_block": total_reads * a.steps / block_dt[0]},
           "certificate": cert,
           "workload": "configs[%d]: %d synthetic %d bp %sreads %s vs %s, matrix %s; step = reiterate_assembly + pop_smp + cull + "
                       "consensus; pass-1 coordinates = true positions"
                       % (cfg, reads, w["read_len"], "paired (two per 300 +- 30 bp fragment, ids /1 /2) aDNA-damaged " if cfg == 3 else ("aDNA-damaged " if cfg != 1 else ""),
                          "per GPU" if scaling == "weak" else "in total, split over the GPUs", w["ref_name"], w["matrix_file"] or "flat"),
           "consensus_len": len(cur), "bytes_per_read": w["bytes_per_read"]}
    if world > 1 or force_dist:
        job["collectives"] = "libmia_hip (RCCL communicator, mia_hip_iterate)" if c_comm else "torch.distributed (mapping-iterative-assembler_amd/dist.py)"
        if comm_info:
            job["communicator"] = {"ranks": comm_info[0], "rank_of_this_line": comm_info[1], "transport": comm_info[2]}
        if coll_note:
            job["collectives_note"] = coll_note
    return job, {"pipe": pipe, "hip": hip, "w": w, "cur": cur, "dominant": dominant, "dom_ms": dom_ms, "dom_launches": dom_launches, "c_comm": c_comm, "n": n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps iterations each; the line reports the median block (min / max beside it)")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU (weak) or in the whole job (strong); default 1 M on one GPU, 10 M (strong) on several")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None, help="default: weak on one GPU, strong (north_star's target) on several")
    ap.add_argument("--config", type=int, default=None, choices=(1, 2, 3, 4), help="BASELINE.json configs[k] for the timed steps; default 1 on one GPU, 3 on several")
    ap.add_argument("--coll", choices=("rccl", "torch"), default="rccl", help="N > 1: exchanges inside libmia_hip (RCCL) or in Python over torch.distributed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="the headline line only (no configs2/configs3/configs4/pass1/myers sections, no weak figure)")
    ap.add_argument("--pmc-run", default=None, help="reduced run under rocprofv3 --pmc: only the timed steps of this config, plus k_peak_copy for the FETCH_SIZE calibration")
    ap.add_argument("--allow-alt-build", action="store_true", help="run although a non-release MIA_HIP_* switch is set (the alt build is then timed, and the line says so)")
    a = ap.parse_args()

    # a leftover MIA_HIP_* switch silently selects libmia_hip_alt.so (debug branches, other launch paths): refuse, or say so
    sys.path.insert(0, ROOT)
    import mia_amd as _mia
    stray = _mia.alt_switches_set()
    if stray and not (a.allow_alt_build or a.pmc_run):
        sys.stderr.write("bench.py: non-release switches are set (%s): the alt build would be timed; unset them or pass --allow-alt-build\n" % ", ".join(stray))
        sys.exit(5)

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)                                 # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): not run\n")
        sys.exit(3)

    # stdout carries ONE line, the JSON of rank 0: whatever libraries print while they start up (RCCL's version banner ...)
    # goes to stderr -- file descriptor 1 points there until the line is written
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    force_dist = os.environ.get("MIA_BENCH_FORCE_DIST") == "1"   # exercise the RCCL code path on one GPU
    if force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        if torch.cuda.device_count() <= local:
            sys.stderr.write(f"bench.py: rank {rank} has no GPU {local} on this box\n")
            sys.exit(3)
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import mia_amd
    env = {"torch": torch, "dist": dist, "rank": rank, "world": world, "local": local, "force_dist": force_dist}

    if a.pmc_run:
        a.no_cpu_baseline = a.no_extras = True
        a.blocks = 1
    # one GPU: configs[1] as BASELINE.json quotes the metric.  Several GPUs: north_star's target is STRONG scaling of the
    # 10 M-read job (configs[3]) -- the whole job split over the ranks; the weak figure (configs[1], 1 M reads per GPU)
    # rides along in the same line.
    cfg = a.config if a.config else (1 if world == 1 else 3)
    scaling = a.scaling if a.scaling else ("weak" if world == 1 else "strong")
    reads = a.reads if a.reads else (1_000_000 if world == 1 else 10_000_000)
    peaks = None
    if rank == 0 and world == 1 and not a.pmc_run:
        hp = mia_amd.MiaHip(local)
        gbs, ginst = hp.measure_peaks(1 << 30)
        issue, mhz = hp.measure_issue()
        hp.close()
        # the guide's issue bound (MI355X_MICROARCH.md): a SIMD issues one wave64 VALU instruction every 2 cycles at best -> SIMDs x clock / 2
        peaks = {"hbm_copy_gbs": gbs, "valu_ginst_s": ginst, "valu_issue_ginst_s": issue, "shader_clock_mhz": mhz,
                 "issue_bound_2cycle_ginst_s": 1024 * mhz * 1e6 / 2 / 1e9 if mhz else None,
                 "valu_mix_over_issue": ginst / issue if issue else None,
                 "note": "k_peak_copy: 1 GiB streamed in and out (16 bytes per lane and step), best of 5; k_peak_valu: v_max3_i32/v_add_u32 chains, 8 waves "
                         "per SIMD, wave64 instructions per second over the whole chip; k_peak_issue: sixteen independent v_add_u32 per round (nothing waits for "
                         "anything), shader_clock_mhz = s_memtime / s_memrealtime inside that kernel (csrc/mia_peak_kernels.h)"}
    job, live = timed_job(a, env, cfg, reads, scaling, peaks)
    pipe, hip, w, cur, n = live["pipe"], live["hip"], live["w"], live["cur"], live["n"]
    weak = None
    if world > 1 and not a.no_extras and not (cfg == 1 and scaling == "weak"):
        hip.close()
        weak, live2 = timed_job(a, env, 1, 1_000_000, "weak")
        live2["hip"].close()

    if rank == 0:
        tag = f"cfg{cfg}"
        pmc, stale = load_pmc(tag)
        stages, counts = pipe.stages(a.steps, peaks, pmc, stale)
        use_timed_region(stages, live["dominant"], live["dom_ms"], live["dom_launches"], a.steps)
        out = {
            "metric": "reads aligned/sec per iteration (16.5kb mito ref, 100bp reads)",
            "value": job["value"], "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": job["ms_per_step"], "blocks": job["blocks"], "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": job["workload"], "reads_per_gpu": n, "total_reads": job["total_reads"], "consensus_len": job["consensus_len"],
                       "bytes_per_read": job["bytes_per_read"]},
            "roofline": roofline(stages, peaks, tag, stale),
            "read_fate": counts,
            "certificate": dict(job["certificate"], workload="make_workload(%d, %d, seed=%d)" % (cfg, n, 1 + rank)),
            "step_traffic": step_traffic(pmc, stale, n, job["bytes_per_read"]),
            "library": {"path": os.path.relpath(hip.lib_path, ROOT), "alt_build": hip.is_alt_build, "switches": mia_amd.alt_switches_set(),
                        "source_hash": source_hash()},
        }
        for k in ("collectives", "communicator", "collectives_note"):
            if k in job:
                out[k] = job[k]
        if weak:
            out["weak"] = weak
        if peaks:
            out["peaks"] = peaks
        if pipe.phase:
            out["phase_ms_per_step"] = {k: v / (a.steps + a.warmup) * 1e3 for k, v in pipe.phase.items()}
        if world == 1 and not a.no_extras and not a.pmc_run:
            # configs[1] read literally ("vs mt311 ... 1 iteration"): the iteration against the starting reference itself (for
            # mt311 an ambiguity code in every tenth column), with every buffer in place; `value` is the steady state
            pipe.reset_stats()
            ms = []
            for _ in range(3):
                hip.sync()
                t1 = time.perf_counter()
                pipe.step(w["ref"])
                hip.sync()
                ms.append((time.perf_counter() - t1) * 1e3)
            first_ms = min(ms)
            out["value_first_iteration"] = n / (first_ms * 1e-3)
            out["first_iteration"] = {"ms": first_ms, "over_steady": first_ms / out["ms_per_step"], "reference": w["ref_name"],
                                      "read_fate": pipe.stages(3, None, None, True)[1]}
        if world == 1 and not a.no_extras:      # single-GPU line only: the other ranks of a sharded run would sit waiting
            # pass 1 (new_kmer_filter + sg_align over the whole wrapped reference, both strands), reported separately
            import gen_data
            m = min(n, 200_000)
            stored, rc, offsets = w["stored"], w["rc"], w["offsets"]
            seq = np.where(rc[:m, None] == 1, gen_data._COMP[stored[:m, ::-1]], stored[:m]).astype(np.uint8)   # as sequenced
            p1 = {}
            # mt311 itself carries an ambiguity code in every tenth column (N for the aligner); "plain_ref" is the same
            # sequence with those resolved, the usual kind of reference, where the diagonal filter decides most reads
            for label, k, r1 in (("k12", 12, w["ref"]), ("no_kmer", -1, w["ref"]), ("no_kmer_plain_ref", -1, w["plain_ref"])):
                hip.pass1(r1, w["circular"], seq[:256].reshape(-1), offsets[:257], k)          # warm-up
                t1 = time.perf_counter()
                sc, _, _, _, fl = hip.pass1(r1, w["circular"], seq.reshape(-1), offsets[: m + 1], k)
                p1[label] = {"reads_per_s": m / (time.perf_counter() - t1), "kernel_reads_per_s": m / (hip.pass1_time() * 1e-3),
                             "reads": m, "kept": int((fl & 2).astype(bool).sum()), "decided_by_diag_filter": hip.pass1_filtered(),
                             "decided_by_anchored_windows": hip.pass1_anchored()}
            # the same with a position-specific matrix (ancient.submat.txt, damaged reads of configs[2]): no diagonal filter, the
            # anchored windows work in losses; against mt311 itself the N columns alone (340 each against 640 for the
            # cheapest substitution) use up the pigeonhole's budget and the whole-strand DP takes nearly every read
            try:
                w2 = make_workload(2, m, 3)
                hip2 = mia_amd.MiaHip(int(os.environ.get("LOCAL_RANK", "0")))
                hip2.set_pssm(w2["pssm"])
                seq2 = np.where(w2["rc"][:m, None] == 1, gen_data._COMP[w2["stored"][:m, ::-1]], w2["stored"][:m]).astype(np.uint8)
                for label, r1 in (("ancient_no_kmer", w2["ref"]), ("ancient_no_kmer_plain_ref", w2["plain_ref"])):
                    hip2.pass1(r1, w2["circular"], seq2[:256].reshape(-1), w2["offsets"][:257], -1)
                    t1 = time.perf_counter()
                    sc, _, _, _, fl = hip2.pass1(r1, w2["circular"], seq2.reshape(-1), w2["offsets"][: m + 1], -1)
                    p1[label] = {"reads_per_s": m / (time.perf_counter() - t1), "kernel_reads_per_s": m / (hip2.pass1_time() * 1e-3),
                                 "reads": m, "kept": int((fl & 2).astype(bool).sum()), "decided_by_diag_filter": hip2.pass1_filtered(),
                                 "decided_by_anchored_windows": hip2.pass1_anchored()}
                hip2.close()
            except Exception as ex:                               # (an extra: never costs the line its headline)
                p1["ancient_error"] = repr(ex)
            out["pass1"] = p1
            out["myers"] = section_myers(hip, a.no_cpu_baseline)
        if a.pmc_run:
            hip.measure_peaks(1 << 28)           # k_peak_copy in the counter passes: the FETCH_SIZE calibration
        if not a.no_cpu_baseline and world == 1:      # the CPU comparator is timed beside the single-GPU line only
            out["cpu_baseline"] = cpu_baseline(w)
        if not weak:
            hip.close()
        if world == 1 and not a.no_extras and cfg == 1:
            # configs[1] from mt311 to convergence, every iteration timed ("iterations: 2" used to be assumed)
            out["configs1"] = section_converge(mia_amd, local, 1, n, 1, peaks, True, w=w)
            out["configs2"] = section_converge(mia_amd, local, 2, 1_000_000, 3, peaks, a.no_cpu_baseline)
            w3 = make_workload(3, 10_000_000, 4)
            out["configs3"] = section_converge(mia_amd, local, 3, 10_000_000, 4, peaks, a.no_cpu_baseline, w=w3)
            # north_star's 8-GPU job as far as one GPU can show it (VERDICT r04 missing #2): the 1.25 M reads one rank would hold,
            # on their own; and all eight shards through the sharded code path on this one GPU (loopback transport)
            out["configs3_share"] = section_converge(mia_amd, local, 3, 1_250_000, 4, peaks, True)
            try:
                out["configs3_loopback_w8"] = section_loopback(mia_amd, local, w3, 8)
            except Exception as ex:                           # (an extra: never costs the line its headline)
                out["configs3_loopback_w8"] = {"error": repr(ex)}
            del w3
            # configs[4] at its size (5 M reads of 150 bp against the 100 kb region: what BASELINE.json spreads over 8 GPUs, on one), and one
            # GPU's share of it (625 k reads: the N = 8 point of that job as far as one GPU can show it)
            out["configs4"] = section_converge(mia_amd, local, 4, 5_000_000, 5, peaks, a.no_cpu_baseline)
            out["configs4_share"] = section_converge(mia_amd, local, 4, 625_000, 5, peaks, True)
        if world == 1 and not a.no_extras:
            # the host program end to end, LAST: every context of this process is closed by now.  (Rounds 4 and 5 ran it while the bench still
            # held a context with a million reads on the same GPU: the child's first mia_hip_iterate then took 30-60 ms -- two processes
            # taking turns on one device -- against 2.6-2.7 ms when it has the GPU to itself, which is how a user runs it.)
            out["cli"] = section_cli(w)
        extras_path = write_extras(out)
        hl = headline(out, extras_path)
        line = json.dumps(hl, separators=(",", ":"))
        # never die for a long line after everything was measured -- drop optional keys (all of them are in the extras file)
        for victim in ("collectives_note", "library", "cpu_baseline.sample", "roofline.pmc", "roofline.dp_phase", "step_traffic", "weak", "roofline.valu", "to_convergence"):
            if len(line) < HEADLINE_MAX:
                break
            top, _, sub = victim.partition(".")
            if sub:
                if isinstance(hl.get(top), dict):
                    hl[top].pop(sub, None)
            else:
                hl.pop(top, None)
            hl["headline_shortened"] = True
            line = json.dumps(hl, separators=(",", ":"))
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(line, flush=True)                       # the LAST (and only) stdout line
        os.dup2(2, 1)
    if rank != 0 and not weak:
        hip.close()                         # (the communicator goes with the context)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
