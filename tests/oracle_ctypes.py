"""ctypes binding of oracle/_build/libmia_oracle.so -- TEST INFRASTRUCTURE.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it."""
import ctypes as C


class Pssm(C.Structure):
    _fields_ = [("sm", C.c_int * 5 * 5 * 31)]


class Aln(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("best", "aec", "aer", "abc", "abr")]


class Counts(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("As", "scoreA", "Cs", "scoreC", "Gs", "scoreG", "Ts", "scoreT", "gaps", "cov")]


class Opts(C.Structure):
    _fields_ = [("circular", C.c_int), ("iterate", C.c_int), ("cons_code", C.c_int), ("hard_cut", C.c_int),
                ("score_cut_set", C.c_int), ("slope", C.c_double), ("intercept", C.c_double),
                ("kmer_len", C.c_int), ("soft_mask", C.c_int), ("final_only", C.c_int), ("do_trim", C.c_int),
                ("adapter", C.c_char * 128)]


class AlnSeq(C.Structure):
    _fields_ = [("id", C.c_char * 104), ("desc", C.c_char * 132), ("seq", C.c_char * 513), ("smp", C.c_char * 513),
                ("ins", C.c_char_p * 513), ("start", C.c_int), ("end", C.c_int), ("score", C.c_int),
                ("num_inputs", C.c_int), ("segment", C.c_char), ("revcom", C.c_int), ("trimmed", C.c_int),
                ("dropped", C.c_int)]


class Frag(C.Structure):
    _fields_ = [("id", C.c_char * 104), ("desc", C.c_char * 132), ("seq", C.c_char * 257), ("seq_len", C.c_int),
                ("trimmed", C.c_int), ("trim_point", C.c_int), ("strand_known", C.c_int), ("rc", C.c_int),
                ("as_", C.c_int), ("ae", C.c_int), ("score", C.c_int), ("front", C.c_int), ("back", C.c_int),
                ("unique_best", C.c_int), ("num_inputs", C.c_int)]


def load(path):
    lib = C.CDLL(path)
    P = C.POINTER
    lib.ora_pssm_flat.argtypes = [P(Pssm)]
    lib.ora_pssm_revcom.argtypes = [P(Pssm), P(Pssm)]
    lib.ora_pssm_read.argtypes = [C.c_char_p, P(Pssm)]
    lib.ora_pssm_read.restype = C.c_int
    lib.ora_sm_depth.argtypes = [C.c_int, C.c_int]
    lib.ora_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, P(Pssm), C.c_int, P(Aln),
                              C.c_char_p, C.c_char_p, P(C.c_int), P(C.c_int)]
    lib.ora_align.restype = C.c_int
    lib.ora_trim.argtypes = [C.c_char_p, C.c_int, C.c_char_p, P(C.c_int), P(C.c_int), P(Aln)]
    lib.ora_trim.restype = None
    lib.ora_find_consensus.argtypes = [P(Counts), C.c_int]
    lib.ora_find_consensus.restype = C.c_char
    lib.ora_myers_diff.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p]
    lib.ora_myers_diff.restype = C.c_uint
    lib.ora_opts_default.argtypes = [P(Opts)]
    lib.ora_new.argtypes = [P(Opts), P(Pssm)]
    lib.ora_new.restype = C.c_void_p
    lib.ora_free.argtypes = [C.c_void_p]
    lib.ora_load_ref_fasta.argtypes = [C.c_void_p, C.c_char_p]
    lib.ora_set_ref.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.ora_prepare_ref.argtypes = [C.c_void_p]
    lib.ora_pass1_read.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.ora_pass1_file.argtypes = [C.c_void_p, C.c_char_p]
    lib.ora_finish_pass1.argtypes = [C.c_void_p]
    lib.ora_push_frag.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.ora_push_frag.restype = None
    lib.ora_set_threads.argtypes = [C.c_void_p, C.c_int]
    lib.ora_set_threads.restype = None
    lib.ora_iterate.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.ora_consensus.argtypes = [C.c_void_p]
    lib.ora_consensus.restype = C.c_void_p
    lib.ora_write_maln.argtypes = [C.c_void_p, C.c_char_p]
    lib.ora_run.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    lib.ora_num_frags.argtypes = [C.c_void_p]
    lib.ora_frag_at.argtypes = [C.c_void_p, C.c_int]
    lib.ora_frag_at.restype = P(Frag)
    lib.ora_num_culled.argtypes = [C.c_void_p]
    lib.ora_culled_at.argtypes = [C.c_void_p, C.c_int]
    lib.ora_culled_at.restype = P(AlnSeq)
    lib.ora_slot_at.argtypes = [C.c_void_p, C.c_int]
    lib.ora_slot_at.restype = P(AlnSeq)
    lib.ora_ref_len.argtypes = [C.c_void_p]
    lib.ora_ref_seq.argtypes = [C.c_void_p]
    lib.ora_ref_seq.restype = C.c_char_p
    lib.ora_ref_gaps.argtypes = [C.c_void_p]
    lib.ora_ref_gaps.restype = P(C.c_int)
    lib.ora_column_tallies.argtypes = [C.c_void_p, P(C.c_int)]
    lib.libc_free = C.CDLL(None).free
    lib.libc_free.argtypes = [C.c_void_p]
    return lib


def consensus_string(lib, st):
    p = lib.ora_consensus(st)
    s = C.string_at(p).decode()
    lib.libc_free(p)
    return s
