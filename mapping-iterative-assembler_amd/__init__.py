"""mapping-iterative-assembler_amd -- MI355X-native per-iteration path of MIA.

Python is plumbing only: this module is a ctypes binding of libmia_hip.so (the
C ABI in include/mia_hip.h; HIP kernels in csrc/).  It fails loudly when the
library or a GPU is missing -- there is no CPU fallback.  Import it as
`mia_amd` (see mia_amd.py at the repository root; the directory name contains
a hyphen).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIA_HIP_LIB") or os.path.join(_HERE, "libmia_hip.so")    # (MIA_HIP_LIB: another build of the same library, for A/B timing)
# libmia_hip_alt.so: the same sources with -DMIA_HIP_ALT_PATHS -- every alternative route and debug switch the differential tests and
# the profiling tools reach through MIA_HIP_* variables.  The release library does not read them (csrc/mia_hip.hip: alt_env): a context
# made while one of them is set comes from the alt build.
ALT_LIB_PATH = os.path.join(_HERE, "libmia_hip_alt.so")
RELEASE_ENV = {"MIA_HIP_THREADS", "MIA_HIP_TIMING", "MIA_HIP_LOOPBACK_TIMEOUT", "MIA_HIP_SPIN_WAIT", "MIA_HIP_LIB"}


def alt_switches_set():
    return sorted(k for k in os.environ if k.startswith("MIA_HIP_") and k not in RELEASE_ENV)

PSSM_WORDS = 31 * 5 * 5
TALLY_WORDS = 12
COL_INSERT, COL_CLIP = -1, -2
P1_PASSED, P1_KEPT, P1_STRAND_KNOWN, P1_SPLIT = 1, 2, 4, 8


class MiaHipError(RuntimeError):
    pass


def _load(path=None):
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise MiaHipError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(path)
    P, vp = C.POINTER, C.c_void_p
    lib.mia_hip_create.argtypes = [P(vp), C.c_int]
    lib.mia_hip_destroy.argtypes = [vp]
    lib.mia_hip_destroy.restype = None
    lib.mia_hip_last_error.argtypes = [vp]
    lib.mia_hip_last_error.restype = C.c_char_p
    lib.mia_hip_sync.argtypes = [vp]
    lib.mia_hip_set_pssm.argtypes = [vp, vp, vp]
    lib.mia_hip_upload_reads.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    lib.mia_hip_pass1.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
    lib.mia_hip_realign.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int]
    lib.mia_hip_align_windows.argtypes = [vp, vp, C.c_int64, vp, vp]
    lib.mia_hip_get_alignments.argtypes = [vp, vp, vp, vp]
    lib.mia_hip_get_scripts.argtypes = [vp, vp, C.c_int32, vp]
    lib.mia_hip_cull.argtypes = [vp, C.c_int32, C.c_double, C.c_double, C.c_int64]
    lib.mia_hip_get_dropped.argtypes = [vp, vp, vp]
    lib.mia_hip_set_slot_dropped.argtypes = [vp, vp, C.c_int64]
    lib.mia_hip_score_cut.argtypes = [vp, vp, vp, C.c_int64, P(C.c_double), P(C.c_double)]
    lib.mia_hip_score_cut.restype = None
    lib.mia_hip_num_records.argtypes = [vp, P(C.c_int64)]
    lib.mia_hip_tally.argtypes = [vp]
    lib.mia_hip_tally_buffers.argtypes = [vp, P(vp), P(C.c_int64), P(vp), P(C.c_int64)]
    lib.mia_hip_ins_events.argtypes = [vp, P(vp), P(C.c_int64)]
    lib.mia_hip_set_ins_events.argtypes = [vp, vp, C.c_int64]
    lib.mia_hip_get_tally.argtypes = [vp, vp, vp]
    lib.mia_hip_set_tally.argtypes = [vp, C.c_int32, vp, vp]
    lib.mia_hip_iterate.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int, C.c_int32, vp, C.c_int, vp, C.c_int64, P(C.c_int64)]
    lib.mia_hip_consensus.argtypes = [vp, C.c_int, vp, C.c_int64, P(C.c_int64)]
    lib.mia_hip_myers.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp]
    if hasattr(lib, "mia_hip_myers_packed"):      # (MIA_HIP_LIB may name an older build, for A/B timing)
        lib.mia_hip_myers_packed.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
    lib.mia_hip_pre_cull_counts.argtypes = [vp, vp, vp]
    lib.mia_hip_filter_stats.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    lib.mia_hip_band_stats.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.mia_hip_bx_stats.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.mia_hip_bx_counters.argtypes = [vp, vp]
    lib.mia_hip_myers_align.argtypes = [vp, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, vp, vp, vp]
    lib.mia_hip_pass1_time.argtypes = [vp, P(C.c_double)]
    lib.mia_hip_myers_time.argtypes = [vp, P(C.c_double)]
    lib.mia_hip_ma_tally.argtypes = [vp, C.c_int32, vp, C.c_int64, vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp]
    lib.mia_hip_trim.argtypes = [vp, C.c_char_p, C.c_int64, vp, vp, vp, vp]
    lib.mia_hip_set_back_slots.argtypes = [vp, vp]
    lib.mia_hip_set_pass1_state.argtypes = [vp, vp, vp, vp]
    lib.mia_hip_get_record_params.argtypes = [vp, vp, vp]
    lib.mia_hip_set_read_base.argtypes = [vp, C.c_int64]
    lib.mia_hip_links.argtypes = [vp, P(vp), P(C.c_int64)]
    lib.mia_hip_set_links.argtypes = [vp, vp, C.c_int64]
    lib.mia_hip_link_lengths.argtypes = [vp, P(vp), P(vp), P(C.c_int64)]
    lib.mia_hip_finish_links.argtypes = [vp]
    lib.mia_hip_plain_stats.argtypes = [vp, C.c_int, P(C.c_double), P(C.c_int64), P(C.c_int64), P(C.c_int64)]
    lib.mia_hip_score_sums.argtypes = [vp, vp]
    lib.mia_hip_score_cut_from_sums.argtypes = [vp, P(C.c_double), P(C.c_double)]
    lib.mia_hip_score_cut_from_sums.restype = C.c_int
    lib.mia_hip_trim_stats.argtypes = [vp, P(C.c_int64)]
    lib.mia_hip_get_ins_tally.argtypes = [vp, vp, vp, C.c_int64, P(C.c_int64)]
    lib.mia_hip_kernel_time.argtypes = [vp, C.c_int, P(C.c_double), P(C.c_int64)]
    lib.mia_hip_stage_stats.argtypes = [vp, C.c_int, C.c_int32, vp, vp, vp, P(C.c_int32)]
    lib.mia_hip_set_stage_mask.argtypes = [vp, C.c_uint32]
    lib.mia_hip_comm_unique_id.argtypes = [vp]
    lib.mia_hip_comm_init.argtypes = [vp, vp, C.c_int32, C.c_int32]
    lib.mia_hip_comm_destroy.argtypes = [vp]
    lib.mia_hip_comm_attach.argtypes = [vp, vp]
    lib.mia_hip_comm_info.argtypes = [vp, P(C.c_int32), P(C.c_int32), P(C.c_char_p)]
    lib.mia_hip_loopback_create.argtypes = [C.c_int32, P(vp)]
    lib.mia_hip_loopback_table.argtypes = [vp, C.c_int32, vp]
    lib.mia_hip_loopback_destroy.argtypes = [vp]
    lib.mia_hip_loopback_destroy.restype = None
    lib.mia_hip_measure_peaks.argtypes = [vp, C.c_int64, P(C.c_double), P(C.c_double)]
    lib.mia_hip_measure_issue.argtypes = [vp, P(C.c_double), P(C.c_double)]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


_alt_lib = None


def alt_lib():
    """the build with the alternative routes and debug switches (see ALT_LIB_PATH)"""
    global _alt_lib
    if _alt_lib is None:
        _alt_lib = _load(ALT_LIB_PATH)
    return _alt_lib


def exported_symbols():
    """Every entry point include/mia_hip.h declares (used by the CPU-side ABI test)."""
    return ["mia_hip_create", "mia_hip_destroy", "mia_hip_last_error", "mia_hip_sync", "mia_hip_set_pssm",
            "mia_hip_upload_reads", "mia_hip_pass1", "mia_hip_realign", "mia_hip_align_windows", "mia_hip_get_alignments", "mia_hip_get_scripts", "mia_hip_cull",
            "mia_hip_get_dropped", "mia_hip_set_slot_dropped", "mia_hip_score_cut", "mia_hip_num_records",
            "mia_hip_tally", "mia_hip_tally_buffers", "mia_hip_ins_events", "mia_hip_set_ins_events",
            "mia_hip_get_tally", "mia_hip_consensus", "mia_hip_myers", "mia_hip_myers_packed", "mia_hip_myers_align", "mia_hip_filter_stats", "mia_hip_band_stats", "mia_hip_bx_stats", "mia_hip_bx_counters", "mia_hip_kernel_time", "mia_hip_pass1_time", "mia_hip_myers_time", "mia_hip_pass1_filtered", "mia_hip_pass1_anchored", "mia_hip_pre_cull_counts", "mia_hip_ma_tally", "mia_hip_get_ins_tally", "mia_hip_trim", "mia_hip_trim_stats", "mia_hip_set_back_slots", "mia_hip_set_pass1_state",
            "mia_hip_get_record_params", "mia_hip_set_read_base", "mia_hip_links", "mia_hip_set_links", "mia_hip_link_lengths",
            "mia_hip_finish_links", "mia_hip_plain_stats", "mia_hip_score_sums",
            "mia_hip_score_cut_from_sums", "mia_hip_stage_stats", "mia_hip_measure_peaks", "mia_hip_measure_issue", "mia_hip_set_tally", "mia_hip_iterate", "mia_hip_set_stage_mask", "mia_hip_comm_unique_id", "mia_hip_comm_init", "mia_hip_comm_destroy",
            "mia_hip_comm_attach", "mia_hip_comm_info", "mia_hip_loopback_create", "mia_hip_loopback_table", "mia_hip_loopback_destroy"]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def flat_pssm():
    """init_flatsubmat (reference src/pssm.c:96-126)."""
    p = np.zeros((31, 5, 5), dtype=np.int32)
    for i in range(5):
        for j in range(4):
            p[:, i, j] = 200 if i == j else -600
        p[:, i, 4] = -100
    p[:, 4, :] = -10
    return p


def revcom_pssm(p):
    """revcom_submat (reference src/pssm.c:53-91): rc[30-d][3-i][3-j] = sm[d][i][j], index 4 fixed."""
    idx = np.array([3, 2, 1, 0, 4])
    return np.ascontiguousarray(p[::-1][:, idx][:, :, idx]).astype(np.int32)


def read_pssm(path):
    """read_pssm (reference src/io.c:408-503)."""
    p = np.zeros((31, 5, 5), dtype=np.int32)
    with open(path) as f:
        lines = f.read().split("\n")
    k = 0
    for d in range(31):
        if "# Matrix for position" not in lines[k]:
            raise ValueError(f"Problem parsing matrix file: {path}")
        for i in range(4):
            p[d, i, :4] = [int(x) for x in lines[k + 1 + i].split("\t")[:4]]
            p[d, i, 4] = -100
        p[d, 4, :] = -10
        k += 6
    return p


class Collectives(C.Structure):
    """mia_hip_collectives (include/mia_hip.h): the table of collectives a sharded iterate() speaks through"""
    _fields_ = [("user", C.c_void_p), ("n_ranks", C.c_int32), ("rank", C.c_int32),
                ("all_gather", C.c_void_p), ("all_reduce_i32", C.c_void_p), ("query", C.c_void_p), ("abort", C.c_void_p),
                ("destroy", C.c_void_p), ("error", C.c_void_p), ("name", C.c_char_p)]


class LoopbackGroup:
    """mia_hip_loopback_*: n_ranks contexts of this process, one host thread each, exchange through host barriers and
    device copies -- the sharded path on a single GPU (RCCL refuses two ranks on one device)"""

    def __init__(self, n_ranks, like=None):
        # the release and the alt build are two libraries with two sets of static state (the loopback registry among them):
        # the group lives in the library of the contexts it serves (`like`: one of them; default: the release build's)
        self._l = like._l if like is not None else lib()
        self._g = C.c_void_p()
        if self._l.mia_hip_loopback_create(n_ranks, C.byref(self._g)) != 0:
            raise MiaHipError("mia_hip_loopback_create failed")
        self.n_ranks = n_ranks

    def attach(self, hip, rank):
        if hip._l is not self._l:
            raise MiaHipError("LoopbackGroup: this context comes from another build of the library than the group (pass like=<a context> when the group is made)")
        t = Collectives()
        if self._l.mia_hip_loopback_table(self._g, rank, C.byref(t)) != 0:
            raise MiaHipError("mia_hip_loopback_table failed")
        hip.comm_attach(t)

    def close(self):
        if self._g:
            self._l.mia_hip_loopback_destroy(self._g)
            self._g = C.c_void_p()


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0 calls it and passes the 128 bytes to the other ranks)"""
    buf = C.create_string_buffer(128)
    if lib().mia_hip_comm_unique_id(buf) != 0:
        raise MiaHipError("mia_hip_comm_unique_id failed: librccl could not be opened")
    return buf.raw


_IUPAC_BITS = np.zeros(256, np.uint8)
for _c, _v in zip("ACGTUSWRYKMBDHVN", (1, 2, 4, 8, 8, 6, 9, 5, 10, 12, 3, 14, 13, 11, 7, 15)):
    _IUPAC_BITS[ord(_c)] = _v
    _IUPAC_BITS[ord(_c.lower())] = _v


def pack_myers_pairs(seq_a, seq_b):
    """The layout mia_hip_myers_packed takes (include/mia_hip.h): 4-bit IUPAC bitmaps, eight to a word, one spare word behind
    every sequence.  Returns (codes uint32, a_off, b_off, la, lb)."""
    n = len(seq_a)
    la = np.array([len(s) for s in seq_a], np.int32)
    lb = np.array([len(s) for s in seq_b], np.int32)
    wa, wb = (la + 7) // 8 + 1, (lb + 7) // 8 + 1
    tot = np.cumsum(np.stack([wa, wb], 1).reshape(-1).astype(np.int64))
    a_off = np.concatenate([[0], tot[1:-1:2]]).astype(np.uint32) if n else np.zeros(0, np.uint32)
    b_off = tot[0::2].astype(np.uint32)
    nib = np.zeros(int(tot[-1]) * 8 if n else 0, np.uint8)
    for i in range(n):
        a = np.frombuffer(seq_a[i] if isinstance(seq_a[i], bytes) else seq_a[i].encode(), np.uint8)
        b = np.frombuffer(seq_b[i] if isinstance(seq_b[i], bytes) else seq_b[i].encode(), np.uint8)
        nib[int(a_off[i]) * 8: int(a_off[i]) * 8 + len(a)] = _IUPAC_BITS[a]
        nib[int(b_off[i]) * 8: int(b_off[i]) * 8 + len(b)] = _IUPAC_BITS[b]
    codes = (nib.reshape(-1, 8).astype(np.uint32) << (4 * np.arange(8, dtype=np.uint32))).sum(axis=1).astype(np.uint32)
    return codes, a_off, b_off, la, lb


class MiaHip:
    """One context = one GPU.  Mirrors the call order of the reference's main loop
    (src/mia_main.c:931-963): realign -> cull -> tally -> consensus."""

    def __init__(self, device=0):
        # (switches are read once, when the context is made: a context made while a non-release switch is set needs the alt build)
        self.is_alt_build = bool(alt_switches_set() and not os.environ.get("MIA_HIP_LIB"))
        self._l = alt_lib() if self.is_alt_build else lib()
        self.lib_path = ALT_LIB_PATH if self.is_alt_build else LIB_PATH
        self._h = C.c_void_p()
        rc = self._l.mia_hip_create(C.byref(self._h), device)
        if rc != 0:
            raise MiaHipError(f"mia_hip_create failed ({rc}): no usable gfx950 device -- there is no CPU fallback")
        self.n = 0
        self.max_len = 0
        self.L = 0

    def close(self):
        if self._h:
            self._l.mia_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MiaHipError(f"libmia_hip error {rc}: {self._l.mia_hip_last_error(self._h).decode()}")

    def set_pssm(self, fwd, rc=None):
        fwd = np.ascontiguousarray(fwd, dtype=np.int32)
        rc = np.ascontiguousarray(revcom_pssm(fwd) if rc is None else rc, dtype=np.int32)
        self._chk(self._l.mia_hip_set_pssm(self._h, _ptr(fwd), _ptr(rc)))

    def upload_reads(self, bases, offsets, rc, strand_known, as_, ae):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        rc = np.ascontiguousarray(rc, dtype=np.uint8)
        sk = np.ascontiguousarray(strand_known, dtype=np.uint8)
        as_ = np.ascontiguousarray(as_, dtype=np.int32)
        ae = np.ascontiguousarray(ae, dtype=np.int32)
        self.n = len(offsets) - 1
        self.lens = (offsets[1:] - offsets[:-1]).astype(np.int32)
        self.max_len = int(self.lens.max()) if self.n else 0
        self._chk(self._l.mia_hip_upload_reads(self._h, self.n, _ptr(bases), _ptr(offsets), _ptr(rc), _ptr(sk), _ptr(as_), _ptr(ae)))

    def pass1(self, ref, circular, bases, offsets, kmer_len=-1, soft_mask=False):
        """new_kmer_filter + sg_align for a batch of reads (as sequenced, upper case).
        Returns score, rc, as, ae, flags (P1_* bits)."""
        if isinstance(ref, str):
            ref = ref.encode()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        score = np.zeros(n, np.int32); as_ = np.zeros(n, np.int32); ae = np.zeros(n, np.int32)
        rc = np.zeros(n, np.uint8); flags = np.zeros(n, np.uint8)
        self._chk(self._l.mia_hip_pass1(self._h, ref, len(ref), 1 if circular else 0, kmer_len, 1 if soft_mask else 0, n,
                                        _ptr(bases), _ptr(offsets), _ptr(score), _ptr(rc), _ptr(as_), _ptr(ae), _ptr(flags)))
        return score, rc, as_, ae, flags

    def realign(self, ref, circular):
        if isinstance(ref, str):
            ref = ref.encode()
        self.L = len(ref)
        self._chk(self._l.mia_hip_realign(self._h, ref, len(ref), 1 if circular else 0))

    def align_windows(self, ref, win_start, win_len):
        """ccheck's per-read re-alignment (reference src/ccheck.cc:569-604): read i against its own window
        ref[win_start[i] : win_start[i] + win_len[i]], no margin; results through alignments() / scripts()."""
        w = np.frombuffer(ref.encode() if isinstance(ref, str) else bytes(ref), dtype=np.uint8)
        st = np.ascontiguousarray(win_start, dtype=np.int64)
        ln = np.ascontiguousarray(win_len, dtype=np.int32)
        assert len(st) == self.n and len(ln) == self.n
        self._chk(self._l.mia_hip_align_windows(self._h, _ptr(w), len(w), _ptr(st), _ptr(ln)))

    def alignments(self):
        s = np.empty(self.n, dtype=np.int32)
        a = np.empty(self.n, dtype=np.int32)
        e = np.empty(self.n, dtype=np.int32)
        self._chk(self._l.mia_hip_get_alignments(self._h, _ptr(s), _ptr(a), _ptr(e)))
        return s, a, e

    def scores(self):
        """fs->score only (4 bytes per read over PCIe instead of 12)."""
        s = np.empty(self.n, dtype=np.int32)
        self._chk(self._l.mia_hip_get_alignments(self._h, _ptr(s), None, None))
        return s

    def scripts(self):
        cols = np.empty((self.n, max(self.max_len, 1)), dtype=np.int16)
        rs = np.empty(self.n, dtype=np.int32)
        self._chk(self._l.mia_hip_get_scripts(self._h, _ptr(cols), cols.shape[1], _ptr(rs)))
        return cols, rs

    def score_cut(self, score, seq_len, unique_best=None):
        score = np.ascontiguousarray(score, dtype=np.int32)
        seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
        ub = None if unique_best is None else np.ascontiguousarray(unique_best, dtype=np.uint8)
        s, i = C.c_double(), C.c_double()
        self._l.mia_hip_score_cut(_ptr(score), _ptr(seq_len), _ptr(ub), len(score), C.byref(s), C.byref(i))
        return s.value, i.value

    def score_sums(self):
        """{sum len, sum score, count, min len, max len} of the reads the score-cut regression uses, reduced on the device."""
        s5 = np.zeros(5, dtype=np.int64)
        self._chk(self._l.mia_hip_score_sums(self._h, _ptr(s5)))
        return s5

    def score_cut_from_sums(self, sums5):
        """(slope, intercept) if the regression follows from the sums alone (all used reads equally long), else None."""
        s5 = np.ascontiguousarray(sums5, dtype=np.int64)
        s, i = C.c_double(), C.c_double()
        if self._l.mia_hip_score_cut_from_sums(_ptr(s5), C.byref(s), C.byref(i)) != 0:
            return None
        return s.value, i.value

    def cull(self, hard_cut=0, slope=0.0, intercept=0.0, slot_base=0):
        self._chk(self._l.mia_hip_cull(self._h, hard_cut, slope, intercept, slot_base))

    def dropped(self):
        f = np.empty(self.n, dtype=np.uint8)
        b = np.empty(self.n, dtype=np.uint8)
        self._chk(self._l.mia_hip_get_dropped(self._h, _ptr(f), _ptr(b)))
        return f, b

    def set_slot_dropped(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self._chk(self._l.mia_hip_set_slot_dropped(self._h, _ptr(flags), len(flags)))

    def num_records(self):
        n = C.c_int64()
        self._chk(self._l.mia_hip_num_records(self._h, C.byref(n)))
        return n.value

    def tally(self):
        self._chk(self._l.mia_hip_tally(self._h))

    def tally_buffers(self):
        """(device ptr, words) of the tally and gaps arrays, for an in-place RCCL all-reduce."""
        pt, nt, pg, ng = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        self._chk(self._l.mia_hip_tally_buffers(self._h, C.byref(pt), C.byref(nt), C.byref(pg), C.byref(ng)))
        return pt.value, nt.value, pg.value, ng.value

    def ins_events(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self._l.mia_hip_ins_events(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_ins_events(self, dptr, n):
        self._chk(self._l.mia_hip_set_ins_events(self._h, C.c_void_p(dptr), n))

    def get_tally(self):
        t = np.empty((TALLY_WORDS, self.L + 1), dtype=np.int32)
        g = np.empty(self.L + 1, dtype=np.int32)
        self._chk(self._l.mia_hip_get_tally(self._h, _ptr(t), _ptr(g)))
        return t, g

    def set_tally(self, tally, gaps=None):
        """BaseCounts of every column from the caller: tally[12][L+1] (word-major, see include/mia_hip.h), gaps[L+1] or None"""
        t = np.ascontiguousarray(tally, dtype=np.int32)
        assert t.ndim == 2 and t.shape[0] == TALLY_WORDS
        self.L = t.shape[1] - 1
        g = None if gaps is None else np.ascontiguousarray(gaps, dtype=np.int32)
        self._chk(self._l.mia_hip_set_tally(self._h, self.L, _ptr(t), _ptr(g)))

    def consensus(self, cons_code=1):
        cap = self.L * 2 + 1024 * 1024
        buf = C.create_string_buffer(cap)
        n = C.c_int64()
        self._chk(self._l.mia_hip_consensus(self._h, cons_code, buf, cap, C.byref(n)))
        return buf.raw[: n.value].decode()

    def iterate(self, ref, circular, hard_cut=0, score_cut=None, cons_code=1):
        """One whole iteration (reference src/mia_main.c:931-963) as one call: realign + score cut + cull + tally +
        consensus without host round trips in between.  score_cut: (slope, intercept) of -S/-N, or None for the
        regression of find_fsdb_score_cut.  Returns the consensus string."""
        if isinstance(ref, str):
            ref = ref.encode()
        self.L = len(ref)
        cap = len(ref) * 2 + 65536
        if getattr(self, "_cons_buf_cap", 0) < cap:
            self._cons_buf, self._cons_buf_cap = C.create_string_buffer(cap), cap
        n = C.c_int64()
        sc = None if score_cut is None else np.ascontiguousarray(score_cut, dtype=np.float64)
        self._chk(self._l.mia_hip_iterate(self._h, ref, len(ref), 1 if circular else 0, hard_cut, _ptr(sc), cons_code, self._cons_buf,
                                          self._cons_buf_cap, C.byref(n)))
        return C.string_at(self._cons_buf, n.value).decode()

    def myers_time(self):
        """kernel time (ms, HIP events) of the last myers() call"""
        v = C.c_double(0)
        self._chk(self._l.mia_hip_myers_time(self._h, C.byref(v)))
        return v.value

    def pass1_anchored(self):
        """reads of the last pass1() call decided by windowed alignment around their 10-mer anchors"""
        v = C.c_int64(0)
        self._chk(self._l.mia_hip_pass1_anchored(self._h, C.byref(v)))
        return v.value

    def pre_cull_counts(self):
        """(AlnSeq records of this context, links the cull will emit): by-products of the last score_sums() call"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self._l.mia_hip_pre_cull_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pass1_filtered(self):
        """reads of the last pass1() call decided by the diagonal filter instead of the whole-reference DP"""
        v = C.c_int64(0)
        self._chk(self._l.mia_hip_pass1_filtered(self._h, C.byref(v)))
        return v.value

    def filter_stats(self, reset=False):
        """(reads examined, reads finished, kernel ms, launches) of the diagonal filter in front of the DP kernels (flat matrix only)"""
        a, b, ms, k = C.c_int64(0), C.c_int64(0), C.c_double(0), C.c_int64(0)
        self._chk(self._l.mia_hip_filter_stats(self._h, 1 if reset else 0, C.byref(a), C.byref(b), C.byref(ms), C.byref(k)))
        return a.value, b.value, ms.value, k.value

    def band_stats(self, reset=False):
        """(reads finished, kernel ms, launches) of the banded DP behind the filter (csrc/band_body.h)"""
        b, ms, k = C.c_int64(0), C.c_double(0), C.c_int64(0)
        self._chk(self._l.mia_hip_band_stats(self._h, 1 if reset else 0, C.byref(b), C.byref(ms), C.byref(k)))
        return b.value, ms.value, k.value

    def bx_stats(self, reset=False):
        """((reads planned on, finished by the plan, by the values DP, by the trace DP), (ms of k_bx_plan, k_bx_values,
        k_bx_trace), launches) of the band pipeline for any matrix (csrc/bandx_body.h)"""
        r = (C.c_int64 * 4)()
        ms = (C.c_double * 3)()
        k = C.c_int64(0)
        self._chk(self._l.mia_hip_bx_stats(self._h, 1 if reset else 0, r, ms, C.byref(k)))
        return tuple(r), tuple(ms), k.value

    def bx_counters(self):
        """the 32 device counters of the band pipeline after the last realign (include/mia_hip.h)"""
        out = (C.c_uint32 * 32)()
        self._chk(self._l.mia_hip_bx_counters(self._h, out))
        return list(out)

    def myers_align(self, seq_a, mode, seq_b, maxd):
        """myers_diff with its backtrace (reference src/myers_align.h:35): (distance or None, row over seq_a, row over seq_b)"""
        a = seq_a.encode() if isinstance(seq_a, str) else seq_a
        b = seq_b.encode() if isinstance(seq_b, str) else seq_b
        d = C.c_uint32(0)
        cap = len(a) + len(b) + 4
        ra, rb = C.create_string_buffer(cap), C.create_string_buffer(cap)
        self._chk(self._l.mia_hip_myers_align(self._h, a, mode, b, maxd, C.byref(d), ra, rb))
        if d.value == 0xFFFFFFFF:
            return None, None, None
        return d.value, ra.value.decode(), rb.value.decode()

    def myers_packed(self, packed, mode, maxd):
        """mia_hip_myers_packed: distances for pairs packed by pack_myers_pairs() (no strlen, no packing inside the call)"""
        codes, a_off, b_off, la, lb = packed
        n = len(la)
        mode = np.ascontiguousarray(mode, dtype=np.int32)
        maxd = np.ascontiguousarray(maxd, dtype=np.int32)
        out = np.zeros(n, dtype=np.uint32)
        import time
        t0 = time.perf_counter()
        self._chk(self._l.mia_hip_myers_packed(self._h, n, _ptr(codes), len(codes), _ptr(a_off), _ptr(b_off), _ptr(la), _ptr(lb), _ptr(mode), _ptr(maxd), _ptr(out)))
        self.myers_call_s = time.perf_counter() - t0
        return out

    def myers(self, seq_a, seq_b, mode, maxd):
        """Batch of myers_diff calls (reference src/myers_align.h:35): distances, 0xFFFFFFFF if >= maxd."""
        n = len(seq_a)
        A = (C.c_char_p * n)(*[s.encode() if isinstance(s, str) else s for s in seq_a])
        B = (C.c_char_p * n)(*[s.encode() if isinstance(s, str) else s for s in seq_b])
        mode = np.ascontiguousarray(mode, dtype=np.int32)
        maxd = np.ascontiguousarray(maxd, dtype=np.int32)
        out = np.zeros(n, dtype=np.uint32)
        import time
        t0 = time.perf_counter()
        self._chk(self._l.mia_hip_myers(self._h, n, A, B, _ptr(mode), _ptr(maxd), _ptr(out)))
        self.myers_call_s = time.perf_counter() - t0          # the C call alone (the lists above are Python's business)
        return out

    def kernel_time(self, reset=False):
        ms, k = C.c_double(), C.c_int64()
        self._chk(self._l.mia_hip_kernel_time(self._h, 1 if reset else 0, C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def stage_stats(self, reset=False):
        """{stage name: (milliseconds, launches)} of every timed kernel since the last reset (HIP events on the context's stream)"""
        cap = 32
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        k = (C.c_int64 * cap)()
        n = C.c_int32(0)
        self._chk(self._l.mia_hip_stage_stats(self._h, 1 if reset else 0, cap, names, ms, k, C.byref(n)))
        return {names[i].decode(): (ms[i], k[i]) for i in range(min(n.value, cap))}

    STAGES = ["k_align_quad", "k_align_quad_plain", "k_diag_filter", "k_band_align", "k_bx_plan", "k_bx_values", "k_bx_trace", "k_tally_binned", "k_pass1"]

    def comm_init(self, unique_id, n_ranks, rank):
        """attach an RCCL communicator (ncclCommInitRank on this context's GPU); unique_id: the 128 bytes of comm_unique_id()
        of rank 0.  iterate() then does the exchanges of a sharded run itself."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._l.mia_hip_comm_init(self._h, buf, n_ranks, rank))

    def comm_attach(self, table):
        """attach a caller-made table of collectives (Collectives); the library copies it"""
        self._chk(self._l.mia_hip_comm_attach(self._h, C.byref(table)))

    def comm_info(self):
        """(ranks, rank, transport) as the attached transport itself reports them"""
        n, r, t = C.c_int32(), C.c_int32(), C.c_char_p()
        self._chk(self._l.mia_hip_comm_info(self._h, C.byref(n), C.byref(r), C.byref(t)))
        return n.value, r.value, (t.value or b"").decode()

    def comm_destroy(self):
        self._chk(self._l.mia_hip_comm_destroy(self._h))

    def set_timed_stages(self, names=None):
        """time only these stages (None: all); see mia_hip_set_stage_mask"""
        mask = 0xFFFFFFFF if names is None else sum(1 << self.STAGES.index(n) for n in names)
        self._chk(self._l.mia_hip_set_stage_mask(self._h, mask))

    def measure_peaks(self, copy_bytes=1 << 30):
        """(HBM copy GB/s, 10^9 wave64 VALU instructions/s) measured on this device"""
        a, b = C.c_double(0), C.c_double(0)
        self._chk(self._l.mia_hip_measure_peaks(self._h, copy_bytes, C.byref(a), C.byref(b)))
        return a.value, b.value

    def measure_issue(self):
        """(10^9 independent wave64 v_add_u32 per second over the chip, shader clock in MHz during that kernel)"""
        a, b = C.c_double(0), C.c_double(0)
        self._chk(self._l.mia_hip_measure_issue(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def ma_tally(self, ref_len, gaps, start, revcom, col_off, seq, smp, ins_record=(), ins_pos=(), ins_off=(0,), ins_bases=b""):
        """show_consensus / find_ins_cons tallies over stored AlnSeq records (reference src/map_alignment.c:107-170)."""
        gaps = np.ascontiguousarray(gaps, dtype=np.int32)
        start = np.ascontiguousarray(start, dtype=np.int32)
        revcom = np.ascontiguousarray(revcom, dtype=np.uint8)
        col_off = np.ascontiguousarray(col_off, dtype=np.int64)
        seq = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
        smp = np.frombuffer(bytes(smp), dtype=np.uint8) if not isinstance(smp, np.ndarray) else smp
        ir = np.ascontiguousarray(ins_record, dtype=np.int32)
        ip = np.ascontiguousarray(ins_pos, dtype=np.int32)
        io = np.ascontiguousarray(ins_off, dtype=np.int64)
        ib = np.frombuffer(bytes(ins_bases), dtype=np.uint8) if not isinstance(ins_bases, np.ndarray) else ins_bases
        self.L = int(ref_len)
        self._chk(self._l.mia_hip_ma_tally(self._h, int(ref_len), _ptr(gaps), len(start), _ptr(start), _ptr(revcom), _ptr(col_off),
                                           _ptr(seq) if len(seq) else None, _ptr(smp) if len(smp) else None, len(ir),
                                           _ptr(ir) if len(ir) else None, _ptr(ip) if len(ip) else None, _ptr(io),
                                           _ptr(ib) if len(ib) else None))

    def ins_tally(self):
        """(ins_off[L+1], ins_tally[slots][9]) of the insert columns, after consensus()."""
        n = C.c_int64()
        off = np.empty(self.L + 1, dtype=np.int32)
        self._chk(self._l.mia_hip_get_ins_tally(self._h, _ptr(off), None, 0, C.byref(n)))
        t = np.zeros((max(n.value, 1), 9), dtype=np.int32)
        if n.value:
            self._chk(self._l.mia_hip_get_ins_tally(self._h, None, _ptr(t), n.value, None))
        return off, t[: n.value]

    def trim(self, adapter, bases, offsets):
        """trim_frag for a batch (reference src/mia.c:1318-1368): returns trimmed[n], trim_point[n]."""
        if isinstance(adapter, str):
            adapter = adapter.encode()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        trimmed = np.zeros(n, np.uint8)
        point = np.zeros(n, np.int32)
        self._chk(self._l.mia_hip_trim(self._h, adapter, n, _ptr(bases), _ptr(offsets), _ptr(trimmed), _ptr(point)))
        return trimmed, point

    def set_back_slots(self, back_slot):
        b = np.ascontiguousarray(back_slot, dtype=np.int64)
        assert len(b) == self.n
        self._chk(self._l.mia_hip_set_back_slots(self._h, _ptr(b)))

    def set_pass1_state(self, front_slot=None, back_slot=None, score=None):
        f = None if front_slot is None else np.ascontiguousarray(front_slot, dtype=np.int64)
        b = None if back_slot is None else np.ascontiguousarray(back_slot, dtype=np.int64)
        sc = None if score is None else np.ascontiguousarray(score, dtype=np.int32)
        self._chk(self._l.mia_hip_set_pass1_state(self._h, _ptr(f), _ptr(b), _ptr(sc)))

    def record_params(self):
        """(params[n][8], back_slot[n]) after cull(): see mia_hip_get_record_params."""
        p = np.empty((self.n, 8), dtype=np.int32)
        b = np.empty(self.n, dtype=np.int64)
        self._chk(self._l.mia_hip_get_record_params(self._h, _ptr(p), _ptr(b)))
        return p, b

    def set_read_base(self, base):
        self._chk(self._l.mia_hip_set_read_base(self._h, int(base)))

    def links(self):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self._l.mia_hip_links(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_links(self, dptr, n):
        self._chk(self._l.mia_hip_set_links(self._h, C.c_void_p(dptr), int(n)))

    def link_lengths(self):
        p, q, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        self._chk(self._l.mia_hip_link_lengths(self._h, C.byref(p), C.byref(q), C.byref(n)))
        return p.value, q.value, n.value

    def finish_links(self):
        self._chk(self._l.mia_hip_finish_links(self._h))

    def plain_stats(self, reset=False):
        """(ms, launches, reads in, reads re-run with a trace) of the values-only first pass."""
        ms, a, b, c = C.c_double(), C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self._l.mia_hip_plain_stats(self._h, 1 if reset else 0, C.byref(ms), C.byref(a), C.byref(b), C.byref(c)))
        return ms.value, a.value, b.value, c.value

    def trim_exact_reruns(self):
        n = C.c_int64()
        self._chk(self._l.mia_hip_trim_stats(self._h, C.byref(n)))
        return n.value

    def pass1_time(self):
        ms = C.c_double()
        self._chk(self._l.mia_hip_pass1_time(self._h, C.byref(ms)))
        return ms.value

    def sync(self):
        self._chk(self._l.mia_hip_sync(self._h))
