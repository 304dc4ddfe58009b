/* mia_hip.h -- C ABI of libmia_hip.so: the MI355X (gfx950) implementation of
 * MIA's per-iteration align-and-consensus path.
 *
 * The reference (mpieva/mapping-iterative-assembler) has no plugin or FFI
 * layer: the seam is ordinary C linkage inside one executable (SURVEY.md
 * section 8(b)).  Each entry point below is the batch, device-resident equivalent of
 * one of those C functions and cites the prototype it replaces.  Signatures
 * carry only plain pointers and sizes; all pointers are HOST pointers unless
 * the parameter name starts with d_ (device pointer, used for multi-GPU
 * all-reduce of tallies by the caller).
 *
 * Threading: a context is not re-entrant (neither is the reference: all its
 * scratch is shared mutable state).  One context per GPU per process.
 * Errors: every function returns MIA_HIP_OK (0) or a negative code;
 * mia_hip_last_error() gives the text.  Nothing falls back to the CPU: if the
 * GPU or the code object is missing, creation fails.
 */
#ifndef MIA_HIP_H
#define MIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIA_HIP_OK 0
#define MIA_HIP_ERR_DEVICE (-1)   /* no gfx950 device / HIP runtime error */
#define MIA_HIP_ERR_ARG (-2)      /* bad argument */
#define MIA_HIP_ERR_RANGE (-3)    /* PSSM values outside the packed-score range */
#define MIA_HIP_ERR_STATE (-4)    /* call order (e.g. realign before upload) */
#define MIA_HIP_ERR_NOMEM (-5)

#define MIA_HIP_PSSM_WORDS (31 * 5 * 5) /* PSSM.sm[31][5][5], src/types.h:155-158 */
#define MIA_HIP_MAX_READ 256            /* INIT_ALN_SEQ_LEN, src/params.h:68 */
#define MIA_HIP_TALLY_WORDS 12          /* per column: As,Cs,Gs,Ts,gaps,cov,scoreA,scoreC,scoreG,scoreT,span,pad */

/* alignment script entry for read row r (see mia_hip_get_scripts) */
#define MIA_HIP_COL_INSERT (-1) /* read base aligned to '-' (insert relative to the reference) */
#define MIA_HIP_COL_CLIP (-2)   /* read base before the alignment start (5' soft clip) */

typedef struct mia_hip_ctx mia_hip_ctx;

/* ---- life cycle ------------------------------------------------------- */
int mia_hip_create(mia_hip_ctx **ctx, int device_index);
void mia_hip_destroy(mia_hip_ctx *ctx);
const char *mia_hip_last_error(const mia_hip_ctx *ctx);
int mia_hip_sync(mia_hip_ctx *ctx); /* wait for all queued work of this context */

/* ---- inputs ----------------------------------------------------------- */

/* The two PSSMs reiterate_assembly receives as `PSSMP ancsubmat, PSSMP
 * rcancsubmat` (src/mia_main.c:29-30) and consensus_assembly_string reads as
 * maln->fpsm / maln->rpsm (src/mia.c:584-589).  Layout sm[depth][ref][read]. */
int mia_hip_set_pssm(mia_hip_ctx *ctx, const int32_t *fwd, const int32_t *rc);

/* The read store: replaces `FSDB fsdb` (src/mia_main.c:26) -- the fields
 * fss[i]->{seq, seq_len, rc, strand_known, as, ae} after pass 1 / clean_FSDB.
 * bases: concatenated ASCII (already reverse-complemented for rc reads, as
 * add_virgin_fs2fsdb leaves them, src/fsdb.c:209-227); offsets[n+1].
 * Reads are uploaded once and stay resident (4-bit packed) for all iterations. */
int mia_hip_upload_reads(mia_hip_ctx *ctx, int64_t n_reads, const char *bases, const int64_t *offsets,
                         const uint8_t *rc, const uint8_t *strand_known, const int32_t *as, const int32_t *ae);

/* ---- pass 1 ------------------------------------------------------------ */

#define MIA_HIP_P1_PASSED 1        /* new_kmer_filter returned non-zero (always set without -k) */
#define MIA_HIP_P1_KEPT 2          /* score >= FIRST_ROUND_SCORE_CUTOFF: the read enters the fsdb (src/mia.c:1614) */
#define MIA_HIP_P1_STRAND_KNOWN 4  /* score > cutoff (src/mia.c:1653) */
#define MIA_HIP_P1_SPLIT 8         /* the alignment crosses the origin: front + back AlnSeq (src/mia.c:1619) */

/* The body of main()'s read loop, src/mia_main.c:759-805, for a batch of reads:
 *   int new_kmer_filter(FragSeqP, KPL*, KPL*, int, AlignmentP, AlignmentP)      -- src/kmer.h:83
 *   int sg_align(MapAlignmentP, FragSeqP, FSDB, AlignmentP, AlignmentP, ...)    -- src/mia.h:212-215
 * ref: reference sequence as read from the FASTA file (case kept: -M soft masking looks at it,
 * populate_kpa runs before make_ref_upper, src/mia_main.c:659-676); circular as -c; kmer_len as -k
 * (-1 = no filter); soft_mask as -M.  bases/offsets: reads as sequenced (upper-cased, <= 256).
 * Uses the forward PSSM of mia_hip_set_pssm for both strands (src/mia_main.c:788-789).
 * Outputs, n_reads each: fs->score, fs->rc, fs->as, fs->ae and MIA_HIP_P1_* flags. */
int mia_hip_pass1(mia_hip_ctx *ctx, const char *ref, int32_t ref_len, int circular, int kmer_len, int soft_mask,
                  int64_t n_reads, const char *bases, const int64_t *offsets, int32_t *score, uint8_t *rc, int32_t *as,
                  int32_t *ae, uint8_t *flags);

/* ---- the per-iteration path ------------------------------------------- */

/* void reiterate_assembly(char* new_ref_seq, int iter_num, MapAlignmentP, FSDB,
 * AlignmentP, PWAlnFragP, PWAlnFragP, PSSMP, PSSMP)  -- src/mia_main.c:24-30.
 * new_ref: the unwrapped consensus (ASCII, only upper-case ACGT are bases);
 * circular != 0 appends the wrap as add_ref_wrap does (src/mia.c:657-689).
 * Re-aligns every strand_known read in its window and updates as/ae/score. */
int mia_hip_realign(mia_hip_ctx *ctx, const char *new_ref, int32_t ref_len, int circular);

/* fs->{score, as, ae} after the call above (src/mia_main.c:254-257); any
 * pointer may be NULL. */
int mia_hip_get_alignments(mia_hip_ctx *ctx, int32_t *score, int32_t *as, int32_t *ae);

/* ccheck's per-read re-alignment (src/ccheck.cc:569-604: init_alignment, pop_s1c_in_a/pop_s2c_in_a, dyn_prog,
 * max_sg_score, find_align_begin, populate_pwaln_to_begin per AlnSeq): read i of the store against ITS OWN
 * window ref[win_start[i] .. win_start[i] + win_len[i]) of a caller-supplied string (ASCII, only upper-case ACGT
 * are bases; for ccheck the contaminant consensus, whose lifted-over stretch is the window) with sg5 = sg3 = 1,
 * no margin around the window and no whole-reference fallback; windows may be shorter than the read and may
 * overlap.  rc[i] of the upload selects the matrix as in mia_hip_realign.  Empty windows are an argument error.
 * Results come back through mia_hip_get_alignments (as/ae = win_start[i] + abc/aec) and mia_hip_get_scripts
 * (columns relative to ref_start[i]).  The consensus path (cull, tally) needs a mia_hip_realign afterwards. */
int mia_hip_align_windows(mia_hip_ctx *ctx, const char *ref, int64_t ref_len, const int64_t *win_start, const int32_t *win_len);

/* The alignment itself, i.e. what populate_pwaln_to_begin (src/mia.c:1440-1497)
 * hands to merge_pwaln_into_maln: for read i and read row r,
 * cols[i*stride + r] = wrapped reference column aligned to that base minus
 * ref_start[i], or MIA_HIP_COL_INSERT / MIA_HIP_COL_CLIP.  stride >= longest read. */
int mia_hip_get_scripts(mia_hip_ctx *ctx, int16_t *cols, int32_t stride, int32_t *ref_start);

/* cull_maln_from_fsdb (src/mia.c:418-506), device part: marks `dropped` for
 * reads with score < hard_cut (hard_cut > 0) or score < intercept + slope*len
 * (IEEE double, as the reference).  `dropped` is sticky per AlnSeq slot, like
 * the reference's never-cleared bit field; slot_base = number of AlnSeq records
 * that precede this context's reads (0 on a single GPU). */
int mia_hip_cull(mia_hip_ctx *ctx, int32_t hard_cut, double slope, double intercept, int64_t slot_base);
int mia_hip_get_dropped(mia_hip_ctx *ctx, uint8_t *front_dropped, uint8_t *back_dropped);
/* Seed the sticky per-slot `dropped` bits, e.g. with the marks the pass-1 cull
 * left on the AlnSeq slots (src/mia_main.c:848); slot s of this context is
 * flags[s - slot_base]. */
int mia_hip_set_slot_dropped(mia_hip_ctx *ctx, const uint8_t *flags, int64_t n_flags);

/* fs->back_asp is set when a read is split at the origin and never cleared (src/mia_main.c:259-276): a read that
 * is no longer split keeps pointing at its old AlnSeq slot, and cull_maln_from_fsdb / pop_smp_from_FSDB /
 * consensus_assembly_string see the record that now sits there a second time through it.  mia_hip_cull reproduces
 * this (duplicate listing, dropped mark by the reader's score, depth codes overwritten by later readers).
 * back_slot[n]: slot index of each read's back record after pass 1, -1 if it was not split (global slot numbers,
 * as fs->back_asp would address them). */
int mia_hip_set_back_slots(mia_hip_ctx *ctx, const int64_t *back_slot);
/* The same plus what a strand-unknown read (pass-1 score exactly FIRST_ROUND_SCORE_CUTOFF, src/mia.c:1653) keeps for
 * ever because reiterate_assembly skips it (src/mia_main.c:178): its pass-1 front_asp slot and its pass-1 score (which
 * find_fsdb_score_cut and cull_maln_from_fsdb keep using).  front_slot / back_slot / score: n entries each, any may be
 * NULL; front_slot is only looked at for reads uploaded with strand_known == 0. */
int mia_hip_set_pass1_state(mia_hip_ctx *ctx, const int64_t *front_slot, const int64_t *back_slot, const int32_t *score);
/* After mia_hip_cull: per read  params[i*8 + {0..3}] = front record {dffBase, actOffset, total, multiplicity},
 * params[i*8 + {4..7}] = back record: the depth code of a column reached after `a` bases of that record is
 * depth(dffBase + actOffset + a, total - (actOffset + a) - 1) (src/fsdb.c:568-581,597-610); multiplicity = how often
 * cull_maln_from_fsdb lists the record.  back_slot[i] = fs->back_asp as a slot index (-1 = NULL). */
int mia_hip_get_record_params(mia_hip_ctx *ctx, int32_t *params, int64_t *back_slot);
/* Sharded runs (one context per GPU): global index of this context's first read, and the exchange of the links
 * between mia_hip_cull and mia_hip_tally:  all-gather the buffers of mia_hip_links (4 int64 per link), hand the
 * concatenation to mia_hip_set_links on every rank, all-reduce(max) the two buffers of mia_hip_link_lengths (record length and
 * read bases of the addressed record, -1 where the slot belongs to another rank), then
 * mia_hip_finish_links.  Not needed on a single GPU. */
int mia_hip_set_read_base(mia_hip_ctx *ctx, int64_t read_base);
int mia_hip_links(mia_hip_ctx *ctx, int64_t **d_links, int64_t *n_links);
int mia_hip_set_links(mia_hip_ctx *ctx, const int64_t *d_links_all, int64_t n_all);
int mia_hip_link_lengths(mia_hip_ctx *ctx, int32_t **d_len, int32_t **d_act, int64_t *n);
int mia_hip_finish_links(mia_hip_ctx *ctx);

/* find_fsdb_score_cut (src/fsdb.c:269-383) -- HOST helper, no device work: the
 * reference's least-squares line through (seq_len, score) of all unique_best
 * reads with score >= FIRST_ROUND_SCORE_CUTOFF, evaluated in IEEE double in fsdb
 * order (the sums of products are order dependent, so this stays sequential).
 * unique_best may be NULL (= all 1). */
void mia_hip_score_cut(const int32_t *score, const int32_t *seq_len, const uint8_t *unique_best, int64_t n,
                       double *slope, double *intercept);
/* By-products of the last mia_hip_score_sums call (same sweep over the reads, no further device round trip), for a
 * sharded run: the number of AlnSeq records this context will hold (what mia_hip_num_records returns) and the number
 * of links mia_hip_cull will emit (mia_hip_links).  With both known on every rank BEFORE the cull, one small
 * all-gather settles the slot bases and whether any links have to be exchanged at all. */
int mia_hip_pre_cull_counts(mia_hip_ctx *ctx, int64_t *n_records, int64_t *n_links);

/* Pass 1 of find_fsdb_score_cut on the device: sums5 = {sum of seq_len, sum of score, count, min seq_len, max seq_len}
 * over the reads with score >= FIRST_ROUND_SCORE_CUTOFF (integers: exact, all-reducible).  When all those reads have the
 * same length the whole regression follows from the sums (slope_bf = 0/0): mia_hip_score_cut_from_sums (host helper)
 * returns 0 and fills slope/intercept exactly as the reference's arithmetic would; it returns 1 when the lengths
 * differ and the sequential double sums need the scores on the host (mia_hip_get_alignments + mia_hip_score_cut). */
int mia_hip_score_sums(mia_hip_ctx *ctx, int64_t *sums5);
int mia_hip_score_cut_from_sums(const int64_t *sums5, double *slope, double *intercept);
/* number of AlnSeq records (1 per read, 2 if split at the origin) of this context */
int mia_hip_num_records(mia_hip_ctx *ctx, int64_t *n_records);

/* pop_smp_from_FSDB (src/fsdb.c:542-619) + the add_base loop of
 * consensus_assembly_string (src/mia.c:576-595) + ref->gaps (src/mia.c:486-504),
 * as scatter-adds into  tally[(L+1)][MIA_HIP_TALLY_WORDS]  and  gaps[L+1];
 * insert columns (find_ins_cons, src/map_align.c:444-510) go to an event list. */
int mia_hip_tally(mia_hip_ctx *ctx);
/* device views for the caller's RCCL all-reduce (sum on tally, max on gaps) */
int mia_hip_tally_buffers(mia_hip_ctx *ctx, int32_t **d_tally, int64_t *n_tally_words, int32_t **d_gaps,
                          int64_t *n_gaps_words);
/* insert events (8 bytes each) for all-gather across GPUs */
int mia_hip_ins_events(mia_hip_ctx *ctx, uint64_t **d_events, int64_t *n_events);
int mia_hip_set_ins_events(mia_hip_ctx *ctx, const uint64_t *d_events, int64_t n_events);
int mia_hip_get_tally(mia_hip_ctx *ctx, int32_t *tally, int32_t *gaps); /* host copies, (L+1)*12 and L+1 words */
/* The reverse: BaseCounts of every column handed in by the caller (same layout: tally[word][L+1], word = As, Cs, Gs, Ts,
 * gaps, cov, scoreA, scoreC, scoreG, scoreT, span, pad; gaps[L+1] = ref->gaps or NULL for none), no insert events --
 * e.g. tallies reduced outside the library, or find_consensus (src/map_align.c:294-391) on its own.  mia_hip_consensus
 * may follow. */
int mia_hip_set_tally(mia_hip_ctx *ctx, int32_t ref_len, const int32_t *tally, const int32_t *gaps);

/* char* consensus_assembly_string(MapAlignmentP) -- src/mia.h:74, src/mia.c:515-603.
 * Calls find_consensus / find_ins_cons per column on the (all-reduced) tallies.
 * out must hold ref_len + sum(gaps) + 1 bytes; *out_len = strlen(out). */
int mia_hip_consensus(mia_hip_ctx *ctx, int cons_code, char *out, int64_t out_cap, int64_t *out_len);

/* One whole iteration of the main loop, src/mia_main.c:931-963 -- reiterate_assembly, pop_smp_from_FSDB,
 * cull_maln_from_fsdb (with find_fsdb_score_cut unless hard_cut > 0 or a line is given) and consensus_assembly_string --
 * as one call.  Same kernels and same results as mia_hip_realign + mia_hip_cull + mia_hip_tally + mia_hip_consensus, but
 * the numbers those calls pass through the host in between (planner bins, cut line, insert-event count, result arrays)
 * stay on the device; the host waits once behind the alignment and once for the consensus string -- or, with one context
 * and a cut line that is given, hard, or needs no scores (reads of one length), only for the consensus string: cull, tally
 * and consensus are then queued without a look at the alignment's counters and queued again in the rare case that reads
 * were still waiting for the exact one-read-per-thread kernel (their kernels return untouched in that case).
 *   new_ref/ref_len/circular   as mia_hip_realign
 *   hard_cut                   -H (> 0), else
 *   slope_intercept            -S / -N as {slope, intercept}, or NULL: the regression of find_fsdb_score_cut
 *   cons_code, out, out_cap, out_len   as mia_hip_consensus
 * Afterwards every getter (mia_hip_get_alignments, _scripts, _dropped, _record_params, _tally, _ins_tally) answers as
 * after the four separate calls.
 * SHARDED (a communicator attached by mia_hip_comm_init, the read store split in contiguous fsdb blocks, this context's
 * first read announced by mia_hip_set_read_base): the call also does the exchanges between the GPUs, on its own stream --
 * one all-gather of eight integers per rank before the cull (score sums, AlnSeq records, links, reads still waiting for
 * the exact kernel), the link exchange if any rank has links, all-reduce(sum) of the tallies, all-reduce(max) of ref->gaps
 * with the ranks' insert-event counts riding on it, all-gather of the insert events -- and every rank returns the same
 * consensus.  Every rank must call it.  The host waits as often as without a communicator (twice); a rank that fails
 * calls the transport's abort so that the others return an error too. */
int mia_hip_iterate(mia_hip_ctx *ctx, const char *new_ref, int32_t ref_len, int circular, int32_t hard_cut,
                    const double *slope_intercept, int cons_code, char *out, int64_t out_cap, int64_t *out_len);

/* ---- several GPUs: RCCL over xGMI -------------------------------------------
 * One context per GPU (one process or one host thread each).  Rank 0 makes an id, hands the 128 bytes to the others by
 * whatever channel the host program has (a pipe, MPI, a file, an in-process variable), and every rank calls
 * mia_hip_comm_init -- together, as ncclCommInitRank wants.  librccl is opened at that moment (dlopen), not before.
 * The reference has no counterpart (src/ is single-threaded); the contract is SURVEY.md section 8(e). */
#define MIA_HIP_COMM_ID_BYTES 128
int mia_hip_comm_unique_id(void *id128);
int mia_hip_comm_init(mia_hip_ctx *ctx, const void *id128, int32_t n_ranks, int32_t rank);
int mia_hip_comm_destroy(mia_hip_ctx *ctx);

/* The exchanges of a sharded mia_hip_iterate go through a table of two collectives and nothing else; mia_hip_comm_init
 * attaches the RCCL one.  A host program with a transport of its own (MPI, a test harness) attaches its table with
 * mia_hip_comm_attach.  All buffers are DEVICE pointers of the context's GPU, `hip_stream` is the context's hipStream_t:
 * the collective is ordered behind the work queued there and the work queued afterwards is ordered behind it (it may
 * wait for the stream on the host).  Every function returns 0 or a negative MIA_HIP_ERR_* code.
 *   all_gather      every rank contributes `bytes` bytes; d_recv gets n_ranks * bytes in rank order
 *   all_reduce_i32  in place over `count` int32, op = MIA_HIP_OP_SUM / MIA_HIP_OP_MAX
 *   query           what the transport itself says about the group (may be NULL)
 *   abort           called when this rank fails inside a sharded call, so that the peers' collectives return with an
 *                   error instead of waiting for a rank that will not come (may be NULL: then they wait)
 *   destroy         releases `user` (mia_hip_comm_destroy / mia_hip_destroy; may be NULL)
 *   error           text of the last failure (may be NULL) */
#define MIA_HIP_OP_SUM 0
#define MIA_HIP_OP_MAX 1
typedef struct mia_hip_collectives {
  void *user;
  int32_t n_ranks, rank;
  int (*all_gather)(void *user, const void *d_send, void *d_recv, size_t bytes, void *hip_stream);
  int (*all_reduce_i32)(void *user, void *d_buf, size_t count, int op, void *hip_stream);
  int (*query)(void *user, int32_t *n_ranks, int32_t *rank);
  void (*abort)(void *user);
  void (*destroy)(void *user);
  const char *(*error)(void *user);
  const char *name;
} mia_hip_collectives;
int mia_hip_comm_attach(mia_hip_ctx *ctx, const mia_hip_collectives *table);
/* ranks and rank as the attached transport reports them (RCCL: ncclCommCount / ncclCommUserRank), its name
 * ("rccl", "loopback", the caller's); n_ranks = 1, transport = "none" without a communicator */
int mia_hip_comm_info(mia_hip_ctx *ctx, int32_t *n_ranks, int32_t *rank, const char **transport);

/* In-process loopback: n_ranks contexts of ONE process, one host thread each, on one GPU (or GPUs with peer access).
 * The ranks meet at a host barrier and copy each other's device buffers.  RCCL refuses two ranks on one device; this is
 * how the sharded path runs (and is tested) on a single GPU, and what `mia_hip -g 0,0` uses.  A rank that does not
 * arrive within MIA_HIP_LOOPBACK_TIMEOUT seconds (default 120) fails the collective on every rank. */
int mia_hip_loopback_create(int32_t n_ranks, void **group);
int mia_hip_loopback_table(void *group, int32_t rank, mia_hip_collectives *table);
void mia_hip_loopback_destroy(void *group); /* after every context of the group dropped its table */

/* ---- adapter trimming ---------------------------------------------------- */

/* void trim_frag(FragSeqP, char* adapter, AlignmentP) -- src/mia.h, src/mia.c:1318-1368, as main() sets it up for -T
 * (src/mia_main.c:694-717,771-775): flat matrix, adapter (ASCII, 1..127 bases) = rows with sg5 = 1, read = columns,
 * best score over the last column.  bases/offsets: reads as sequenced (1..256 bases each).
 * Out: trimmed[n] = fs->trimmed, trim_point[n] = fs->trim_point (0 when not trimmed). */
int mia_hip_trim(mia_hip_ctx *ctx, const char *adapter, int64_t n_reads, const char *bases, const int64_t *offsets,
                 uint8_t *trimmed, int32_t *trim_point);

/* reads of the most recent mia_hip_trim call whose best path held a gap of 63 or more and that were therefore
 * re-run by the exact scalar kernel (diagnostic; results are identical either way) */
int mia_hip_trim_stats(mia_hip_ctx *ctx, int64_t *exact_reruns);

/* ---- ma: reports from a .maln ------------------------------------------ */

/* The tally loop of show_consensus / get_consensus (src/map_alignment.c:107-170,222-262) and of
 * find_ins_cons (src/map_align.c:444-510) over the AlnSeq records of a .maln file as read_ma
 * (src/map_alignment.c:384-607) leaves them -- EVERY record counts here, dropped or not, unlike
 * consensus_assembly_string.  Uses the matrices of mia_hip_set_pssm (the file's FPSM / RPSM).
 *   gaps[ref_len]                         ref->gaps from the file's GAPS line
 *   start[n], revcom[n]                   per record
 *   col_off[n+1], seq, smp                record r owns seq/smp[col_off[r] .. col_off[r+1]) = columns start..end
 *   ins_record/ins_pos[n_ins], ins_off[n_ins+1], ins_bases   the INS_POS pairs: inserted bases before column
 *                                         start+ins_pos of that record
 * Afterwards mia_hip_get_tally, mia_hip_consensus (the sequence -f 5 prints) and mia_hip_get_ins_tally work. */
int mia_hip_ma_tally(mia_hip_ctx *ctx, int32_t ref_len, const int32_t *gaps, int64_t n_records, const int32_t *start,
                     const uint8_t *revcom, const int64_t *col_off, const char *seq, const char *smp, int64_t n_ins,
                     const int32_t *ins_record, const int32_t *ins_pos, const int64_t *ins_off, const char *ins_bases);

/* BaseCounts of the insert columns (find_ins_cons), valid after mia_hip_consensus: ins_off[ref_len+1] = index of
 * the first insert column before each reference column, ins_tally[slot*9 + {A,C,G,T,bases,scoreA,scoreC,scoreG,scoreT}]
 * (bases = records with a base in that column; cov = T_SPAN word of the column, gaps = cov - bases). */
int mia_hip_get_ins_tally(mia_hip_ctx *ctx, int32_t *ins_off, int32_t *ins_tally, int64_t cap_slots, int64_t *n_slots);

/* ---- Myers edit distance -------------------------------------------------- */

/* unsigned myers_diff(const char *seq_a, enum myers_align_mode mode, const char* seq_b, int maxd,
 *                     char *bt_a, char *bt_b)                                -- src/myers_align.h:35
 * for a batch of pairs: unit-cost edit distance with IUPAC-compatible matching, modes 0/1/2 as
 * the enum (global / only seq_b must be consumed / only seq_a must be consumed, src/myers_align.c:39-40).
 * dist[i] = distance, or 0xFFFFFFFF when it is >= maxd[i] (maxd clamped to len_a+len_b, :13).
 * The backtrace strings of the reference are not produced (mia_hip_myers_align does that for one pair).  seq_a up to 32768
 * characters.  Pairs of up to 320 characters go one to a lane, longer ones one to a workgroup (D-paths while the distance
 * stays under the kernel's cap, the bit-vector sweep beyond it). */
int mia_hip_myers(mia_hip_ctx *ctx, int64_t n_pairs, const char *const *seq_a, const char *const *seq_b, const int32_t *mode,
                  const int32_t *maxd, uint32_t *dist);

/* The same for pairs the caller has packed already -- the batch form a read-against-read use would keep its sequences in (no
 * strlen, no packing on the way in: the host side of mia_hip_myers was seventeen times its kernel).  codes[n_words]: every
 * sequence as 4-bit IUPAC bitmaps (A = 1, C = 2, G = 4, T = 8, the codes of src/myers_align.h:40-67 OR-ed; 0 = matches
 * nothing), character c of a sequence in bits 4 (c % 8) .. 4 (c % 8) + 3 of word off + c / 8, and ONE spare word behind the
 * last word of every sequence.  a_off / b_off: first word of seq_a / seq_b of pair i; la / lb: their lengths in characters.
 * seq_a of up to 320 characters (one pair per lane); longer ones are refused with MIA_HIP_ERR_ARG: mia_hip_myers takes
 * those.  mode, maxd, dist: as mia_hip_myers. */
int mia_hip_myers_packed(mia_hip_ctx *ctx, int64_t n_pairs, const uint32_t *codes, int64_t n_words, const uint32_t *a_off, const uint32_t *b_off,
                         const int32_t *la, const int32_t *lb, const int32_t *mode, const int32_t *maxd, uint32_t *dist);

/* duration of the kernels of the most recent mia_hip_myers call, from HIP events on the context's stream (diagnostic:
 * bench.py prices the kernel apart from the packing of the strings and the copies) */
int mia_hip_myers_time(mia_hip_ctx *ctx, double *kernel_ms);

/* One myers_diff call with its backtrace (what ccheck asks for, src/ccheck.cc:478-480).  A long pair at a small distance
 * (ccheck's: 16.6 kb each, maxd = len / 10) runs the reference's own recurrence on the device -- furthest-reaching D-paths,
 * one row of diagonals per step (k_myers_ond) -- and the table of those rows comes back for the walk; any other pair gets
 * its distance from the bit-vector kernels above and its D-paths on the host.  Either way the two rows of the alignment
 * (bt_a over seq_a, bt_b over seq_b, '-' for a gap) follow the reference's preferences (src/myers_align.c:47-83).
 * Buffers of len + dist + 2 bytes each, as the reference wants them (len_a + maxd + 2 is enough).  BOTH rows are
 * NUL-terminated (the reference leaves bt_b without its terminator, src/myers_align.c:44-45).  dist = 0xFFFFFFFF and
 * untouched buffers when the distance is >= maxd.  bt_a / bt_b may be NULL. */
int mia_hip_myers_align(mia_hip_ctx *ctx, const char *seq_a, int32_t mode, const char *seq_b, int32_t maxd, uint32_t *dist,
                        char *bt_a, char *bt_b);

/* ---- timing hooks for bench.py (HIP events on the context's stream) ------ */
/* milliseconds spent in, and launches of, the windowed DP kernel since the last reset */
int mia_hip_kernel_time(mia_hip_ctx *ctx, int reset, double *align_ms, int64_t *align_launches);
/* The values-only first pass of mia_hip_realign (k_align_quad_plain: score and end column of every read, plus the proof
 * that its alignment is the pure diagonal; the rest goes on to the trace kernel that mia_hip_kernel_time reports):
 * accumulated kernel time, launches, reads that entered and reads that had to be re-run with a trace. */
int mia_hip_plain_stats(mia_hip_ctx *ctx, int reset, double *plain_ms, int64_t *plain_launches, int64_t *reads_in,
                        int64_t *reads_retried);

/* The diagonal filter in front of the DP kernels of mia_hip_realign / mia_hip_align_windows (csrc/diag_filter.h: flat
 * matrix only; a read whose alignment is provably one gap-free diagonal with at most two mismatches is finished by
 * bit-parallel comparison and never reaches the DP): reads examined and reads finished WITHOUT any DP since the last reset
 * (by this filter, or by the plan of the band pipeline below when its band is a single diagonal -- mia_hip_bx_stats);
 * kernel_ms / launches: accumulated time of k_diag_filter (HIP events on the context's stream).  Any pointer may be NULL. */
int mia_hip_filter_stats(mia_hip_ctx *ctx, int reset, int64_t *reads_seen, int64_t *reads_finished, double *kernel_ms, int64_t *launches);
/* The banded DP behind the filter (csrc/band_body.h: flat matrix, windows without N).  A left-over read whose ten-mer
 * anchors confine every alignment that can win to at most 32 diagonals is aligned inside that band, one read per thread,
 * with the same recurrence, tie rules and traceback as dyn_prog / find_align_begin / populate_pwaln_to_begin
 * (reference src/mia.c:740-981, 612-637, 1440-1497); the rest goes on to the full-window kernels.  reads_finished /
 * kernel_ms / launches of k_band_align since the last reset.  Any pointer may be NULL. */
int mia_hip_band_stats(mia_hip_ctx *ctx, int reset, int64_t *reads_finished, double *kernel_ms, int64_t *launches);
/* The band pipeline of mia_hip_realign / mia_hip_align_windows for ANY substitution matrix (csrc/bandx_body.h: the
 * position-specific matrices of src/pssm.c:6-46 included).  Per read: a band of diagonals from the 10-mer anchors that
 * provably holds every alignment that can win or tie (pigeonhole over the losses against the best score each row can
 * have); reads whose band is one diagonal are finished by the plan itself, the others by a values-only banded DP plus the
 * diagonal proof, the rest by a banded DP with trace and the reference's traceback (src/mia.c:740-981, 612-637,
 * 1440-1497).  What none of them can take (N in read or window, too many defects, the trace-0 quirk) goes on to the
 * full-window kernels.  reads4 = {reads planned on, finished by the plan, by the values DP, by the trace DP};
 * kernel_ms3 = accumulated time of k_bx_plan, k_bx_values, k_bx_trace (HIP events on the context's stream);
 * launches = realign calls that used the pipeline.  Any pointer may be NULL. */
int mia_hip_bx_stats(mia_hip_ctx *ctx, int reset, int64_t *reads4, double *kernel_ms3, int64_t *launches);
/* Diagnostic: the first 32 device counters of the band pipeline after the last realign -- [0..4] reads listed for the values DP
 * by band class (8, 16, 24, 32, 64 diagonals), [5..9] for the trace DP, [10] reads the quick plan (round 6: every read asked on
 * the diagonal it was aligned on before) left to the full plan, [12..14] finished by plan / values DP / trace DP, [15] reads
 * planned on, [16] reads the quick plan's one-diagonal form handed to its one-indel form, [17..23] reads not planned, by reason
 * (N in the read, window, too few anchored blocks, anchors too far apart, written-down path outside the window, loss over the
 * pigeonhole budget, band wider than 64), [24..28] reads the values DP left to the late trace launch by class, [29] reads the
 * plan listed for the full-window kernels itself (k_align_open), [30], [31] reads handed to the plan's second / third launch. */
int mia_hip_bx_counters(mia_hip_ctx *ctx, uint32_t *out32);
/* Every timed stage at once: names[k] (static strings: k_align_quad, k_align_quad_plain, k_diag_filter, k_band_align,
 * k_bx_plan, k_bx_values, k_bx_trace, k_tally_binned, k_pass1), accumulated milliseconds and launches since the last
 * reset; *n_stages = how many there are, at most cap are written.  Any pointer may be NULL. */
int mia_hip_stage_stats(mia_hip_ctx *ctx, int reset, int32_t cap, const char **names, double *ms, int64_t *launches, int32_t *n_stages);
/* Which stages are timed: bit k = stage k of mia_hip_stage_stats (default: all).  An event pair costs the stream a few
 * microseconds of idle time; a caller that wants the duration of one kernel over a timed region switches the others off. */
int mia_hip_set_stage_mask(mia_hip_ctx *ctx, uint32_t mask);
/* The two ceilings the roofline is priced against, measured on this device: a streaming copy of
 * copy_bytes (read + written bytes per second, GB/s) and the issue rate of the DP kernels' own instruction mix
 * (v_max3_i32 / v_add_u32 chains, 10^9 wave64 instructions per second over the whole chip).  Either may be NULL. */
int mia_hip_measure_peaks(mia_hip_ctx *ctx, int64_t copy_bytes, double *hbm_copy_gbs, double *valu_ginst_s);
/* The pure issue ceiling beside it: independent v_add_u32 (nothing waits for anything), 10^9 wave64 instructions per second over the
 * whole chip, and the shader clock in MHz the kernel itself ran at (s_memtime over the constant 100 MHz s_memrealtime), so that a
 * kernel's instruction rate can be read against the 2-cycle issue bound at the clock of the day.  Either may be NULL. */
int mia_hip_measure_issue(mia_hip_ctx *ctx, double *add_ginst_s, double *shader_clock_mhz);
/* milliseconds the k_pass1 kernel of the most recent mia_hip_pass1 call took (HIP events) */
int mia_hip_pass1_time(mia_hip_ctx *ctx, double *kernel_ms);
/* reads of the last mia_hip_pass1 call decided by the diagonal filter (csrc/diag_filter.h: flat matrix, no k-mer mask)
 * instead of the whole-reference DP */
int mia_hip_pass1_filtered(mia_hip_ctx *ctx, int64_t *reads);
/* ... and those decided by windowed alignment around their 10-mer anchors (csrc/mia_pass1_kernels.h: k_pass1_anchor) */
int mia_hip_pass1_anchored(mia_hip_ctx *ctx, int64_t *reads);

#ifdef __cplusplus
}
#endif
#endif
