/* mia_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the MIA align-and-consensus path, written from
 * SURVEY.md section 8 and from reading the reference; every function cites the
 * reference file:line it follows.  It exists so that the HIP kernels can be
 * checked on the GPU box, where /root/reference does not exist.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link, import or execute anything in oracle/.  The product
 * (mapping-iterative-assembler_amd/) never does.
 *
 * Parity status: PINNED.  The reference owns no golden vectors or tests for
 * this path, so the pin is the reference itself, compiled
 * by oracle/Makefile.ref into oracle/_ref/ and run in the build container:
 *   - DP matrices / traceback:  tests/golden/dp_vectors.txt   (ref_dp_driver)
 *   - whole runs:               tests/golden/maln/                  (oracle/_ref/mia)
 *   - Myers:                    tests/golden/myers_vectors.txt (ref_myers_driver)
 * tests/test_oracle_vs_golden.py replays all of them against this library.
 */
#ifndef MIA_ORACLE_H
#define MIA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* src/params.h:15-78 */
#define ORA_PSSM_DEPTH 15
#define ORA_GOP 1000
#define ORA_GEP 200
#define ORA_MAX_READ 256   /* INIT_ALN_SEQ_LEN */
#define ORA_MAX_ALN 512    /* 2*INIT_ALN_SEQ_LEN */
#define ORA_MAX_ID 100
#define ORA_MAX_DESC 128
#define ORA_FIRST_ROUND_CUTOFF 2000
#define ORA_REALIGN_BUFFER 50
#define ORA_MAX_ITER 30
#define ORA_HIM (-1073741824) /* INT_MIN/2, src/mia.c:751 */

/* src/types.h:155-158: sm[depth][ref base][read base], A,C,G,T,other = 0..4 */
typedef struct { int sm[2 * ORA_PSSM_DEPTH + 1][5][5]; } ora_pssm;

void ora_pssm_flat(ora_pssm *p);                          /* src/pssm.c:96-126 */
void ora_pssm_revcom(const ora_pssm *in, ora_pssm *out);  /* src/pssm.c:53-91  */
int ora_pssm_read(const char *path, ora_pssm *p);         /* src/io.c:408-503; 0 on failure */
int ora_sm_depth(int row, int len);                       /* src/pssm.c:38-46  */
int ora_base_code(char b);                                /* src/map_align.c:16-29 */
char ora_revcom_char(char b);                             /* src/map_align.c:418-432 */

/* ---- one alignment: dyn_prog + max_sg_score + find_align_begin +
 *      populate_pwaln_to_begin (src/mia.c:740-981,1278-1302,612-637,1440-1497) */
typedef struct {
  int best, aec, aer, abc, abr;
} ora_aln;

/* seq1/seq2 are ASCII (only upper-case ACGT map to 0-3, src/mia.c:1054-1082).
 * mask: len1 bytes or NULL (= all ones).  ref_gapped/frag_gapped: >= 513 bytes
 * or NULL.  S_out/T_out: len2*len1 ints (row-major) or NULL.
 * Returns 0 on success, -1 if len2 == 0 (max_sg_score's INT_MIN case). */
int ora_align(const char *seq1, int len1, const char *seq2, int len2,
              const unsigned char *mask, const ora_pssm *pm, int sg5,
              ora_aln *res, char *ref_gapped, char *frag_gapped,
              int *S_out, int *T_out);

/* ---- adapter trimming: trim_frag (src/mia.c:1318-1368) as main() sets it up (src/mia_main.c:694-717):
 *      read = seq1 (columns), adapter = seq2 (rows), flat matrix, sg5 = 1, sg3 = 0; best score over the LAST COLUMN
 *      (first maximum), walk back to the alignment start; trimmed iff best >= TRIM_SCORE_CUT (1000) or
 *      best >= (aer - abr + 1) * FLAT_MATCH (200); trim_point = abc - 1.  res may be NULL. */
void ora_trim(const char *read, int read_len, const char *adapter, int *trimmed, int *trim_point, ora_aln *res);

/* ---- consensus primitives (src/map_align.c:229-391) */
typedef struct {
  int As, scoreA, Cs, scoreC, Gs, scoreG, Ts, scoreT, gaps, cov;
} ora_counts;
void ora_add_base(char b, ora_counts *bc, const ora_pssm *pm, int pssm_code);
char ora_find_consensus(const ora_counts *bc, int cons_code);

/* ---- Myers O(ND) diff (src/myers_align.c:10-99).  Returns distance or
 *      0xFFFFFFFF when d >= maxd.  bt_a may be NULL. */
unsigned ora_myers_diff(const char *seq_a, int mode, const char *seq_b, int maxd, char *bt_a);

/* ---- the whole assembler state: MapAlignment + FSDB as flat arrays --------
 * AlnSeq slots are addressed by index so that the reference's pointer reuse
 * (slots are recycled every iteration, `dropped` is never cleared,
 * src/map_align.c:866-954, src/mia.c:469-479) is reproduced literally. */
typedef struct {
  char id[ORA_MAX_ID + 4];
  char desc[ORA_MAX_DESC + 4];
  char seq[ORA_MAX_ALN + 1];
  char smp[ORA_MAX_ALN + 1];
  char *ins[ORA_MAX_ALN + 1];
  int start, end, score, num_inputs;
  char segment;
  int revcom, trimmed, dropped;
} ora_alnseq;

typedef struct {
  char id[ORA_MAX_ID + 4];
  char desc[ORA_MAX_DESC + 4];
  char seq[ORA_MAX_READ + 1];
  int seq_len, trimmed, trim_point;
  int strand_known, rc, as, ae, score;
  int front, back; /* slot index, -1 = NULL */
  int unique_best, num_inputs;
} ora_frag;

typedef struct ora_state ora_state;

typedef struct {
  int circular;        /* -c */
  int iterate;         /* -i (1, default) / -n (0) */
  int cons_code;       /* -p */
  int hard_cut;        /* -H, 0 = unset */
  int score_cut_set;   /* -S / -N given */
  double slope, intercept;
  int kmer_len;        /* -k, -1 = off */
  int soft_mask;       /* -M */
  int final_only;      /* -F */
  int do_trim;         /* -T: trim_frag every read before the k-mer filter (src/mia_main.c:771-775) */
  char adapter[128];   /* -a (src/mia_main.c:558-578); default = the Neandertal adapter (:462,466) */
} ora_opts;

void ora_opts_default(ora_opts *o);

ora_state *ora_new(const ora_opts *o, const ora_pssm *anc);
void ora_free(ora_state *st);

/* reference: read_fasta_ref (src/io.c:287-386); 1 on success */
int ora_load_ref_fasta(ora_state *st, const char *path);
/* set the reference from memory (id/desc may be ""), case preserved */
void ora_set_ref(ora_state *st, const char *id, const char *desc, const char *seq);
/* add_ref_wrap / gaps / kmer tables / make_ref_upper (src/mia_main.c:644-676) */
void ora_prepare_ref(ora_state *st);

/* pass 1 on one read: new_kmer_filter + sg_align (src/mia_main.c:759-805).
 * seq is upper-cased/truncated to 256 as read_fasta does (src/io.c:246-278). */
void ora_pass1_read(ora_state *st, const char *id, const char *desc, const char *seq);
/* whole FASTA/FASTQ file through read_next_seq (src/io.c:35-281); returns #records seen */
int ora_pass1_file(ora_state *st, const char *path);

/* a read whose strand and coordinates are already known (add_fs2fsdb, src/fsdb.c:628-663, without the sg_align in front
 * of it): seq in alignment orientation; the first ora_iterate merges its records */
void ora_push_frag(ora_state *st, const char *id, const char *seq, int rc, int as, int ae, int score, int strand_known);
/* host threads for the alignments of ora_iterate (the records are still merged in fsdb order); default 1 */
void ora_set_threads(ora_state *st, int n);

/* src/mia_main.c:812-875: pop_smp, cull, sort, mask reset, clean_FSDB */
void ora_finish_pass1(ora_state *st);
/* one iteration: reiterate_assembly + pop_smp + cull + sort (src/mia_main.c:931-955) */
void ora_iterate(ora_state *st, const char *new_ref, int iter_num);
/* consensus_assembly_string (src/mia.c:515-603); malloc'd, caller frees */
char *ora_consensus(ora_state *st);
/* write_ma (src/map_alignment.c:283-382) */
int ora_write_maln(ora_state *st, const char *path);
/* full main() flow (src/mia_main.c:394-989); returns number of iterations done */
int ora_run(ora_state *st, const char *frag_path, const char *maln_root);

/* accessors for tests */
int ora_num_frags(const ora_state *st);
const ora_frag *ora_frag_at(const ora_state *st, int i);
int ora_num_culled(const ora_state *st);
const ora_alnseq *ora_culled_at(const ora_state *st, int i);
const ora_alnseq *ora_slot_at(const ora_state *st, int i);
int ora_ref_len(const ora_state *st);
const char *ora_ref_seq(const ora_state *st);
const int *ora_ref_gaps(const ora_state *st);
/* per-column tallies exactly as consensus_assembly_string builds them
 * (10 ints per column: As,Cs,Gs,Ts,gaps,cov,scoreA,scoreC,scoreG,scoreT) */
void ora_column_tallies(ora_state *st, int *out /* seq_len*10 */);

#ifdef __cplusplus
}
#endif
#endif
