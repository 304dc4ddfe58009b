/* ref_iter_driver.c -- TEST/BENCH INFRASTRUCTURE (ours, not reference code).
 *
 * Times the REAL reference's per-iteration path -- the statements of
 * /root/reference/src/mia_main.c:931-963:
 *     reiterate_assembly + pop_smp_from_FSDB + cull_maln_from_fsdb +
 *     sort_aln_frags + consensus_assembly_string
 * -- on a read store that is filled directly with post-pass-1 fields
 * (seq, rc, as, ae), exactly the inputs bench.py hands to the GPU path.  Linked
 * against the reference's own objects by oracle/Makefile.ref; the binary lands
 * in oracle/_ref/ and is used by bench.py's cpu_baseline leg (kind "reference").
 *
 * usage: ref_iter_driver <ref.fa> <reads.txt> <circular 0|1> <matrix|flat> <iterations> [dump.txt]
 *   dump.txt (golden fixtures, tools/make_goldens.py iter_push): after every iteration one line "I <k> <consensus>" and
 *   one line "R <score> <as> <ae>" per read in fsdb order
 *   reads.txt: one read per line "rc as ae SEQUENCE" (sequence already in
 *   alignment orientation, as add_virgin_fs2fsdb leaves it)
 * stdout: "reads N iterations K seconds S cons_len L" (S = time inside the path only)
 */
#include "mia.h"
#include <time.h>

/* defined in src/mia_main.c:24-30 (no prototype in the headers) */
void reiterate_assembly(char *new_ref_seq, int iter_num, MapAlignmentP maln, FSDB fsdb, AlignmentP a,
                        PWAlnFragP front_pwaln, PWAlnFragP back_pwaln, PSSMP ancsubmat, PSSMP rcancsubmat);

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s ref.fa reads.txt circular matrix iterations\n", argv[0]); return 2; }
  int circular = atoi(argv[3]), iters = atoi(argv[5]), i;
  PSSMP anc = strcmp(argv[4], "flat") ? read_pssm(argv[4]) : init_flatsubmat();
  PSSMP rcanc = revcom_submat(anc);
  MapAlignmentP maln = init_map_alignment();
  maln->cons_code = 1;
  maln->distant_ref = 0;
  FSDB fsdb = init_FSDB();
  if (read_fasta_ref(maln->ref, argv[1]) != 1) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  if (circular) add_ref_wrap(maln->ref); else maln->ref->wrap_seq_len = maln->ref->seq_len;
  maln->ref->gaps = (int *)calloc(maln->ref->wrap_seq_len + 1, sizeof(int));
  make_ref_upper(maln->ref);
  AlignmentP a = init_alignment(INIT_ALN_SEQ_LEN, maln->ref->wrap_seq_len + 2 * INIT_ALN_SEQ_LEN + 4096, 0, 0);
  a->submat = anc; a->sg5 = 1; a->sg3 = 1;      /* as sg_align leaves them, src/mia.c:1535-1538 */
  PWAlnFragP front = (PWAlnFragP)calloc(1, sizeof(PWAlnFrag)), back = (PWAlnFragP)calloc(1, sizeof(PWAlnFrag));
  FragSeqP fs = (FragSeqP)calloc(1, sizeof(FragSeq));
  FILE *f = fopen(argv[2], "r");
  if (!f) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
  int rc, as, ae, n = 0;
  static char seq[4096];
  while (fscanf(f, "%d %d %d %4095s", &rc, &as, &ae, seq) == 4) {
    memset(fs, 0, sizeof *fs);
    sprintf(fs->id, "r%d", n);
    strncpy(fs->seq, seq, INIT_ALN_SEQ_LEN);
    fs->seq_len = strlen(fs->seq);
    fs->rc = rc; fs->as = as; fs->ae = ae; fs->strand_known = 1; fs->unique_best = 1; fs->num_inputs = 1;
    fs->score = 2001; fs->front_asp = NULL; fs->back_asp = NULL;
    add_fs2fsdb(fs, fsdb);
    n++;
  }
  fclose(f);
  MapAlignmentP culled = (MapAlignmentP)malloc(sizeof(MapAlignment));
  culled->ref = maln->ref;
  culled->AlnSeqArray = (AlnSeqP *)malloc(sizeof(AlnSeqP) * (2 * (size_t)n + 16));
  culled->num_aln_seqs = 0; culled->size = 2 * n + 16; culled->cons_code = 1; culled->distant_ref = 0;
  char *last = (char *)malloc(maln->ref->seq_len + 1);
  strncpy(last, maln->ref->seq, maln->ref->seq_len);
  last[maln->ref->seq_len] = 0;
  char *cons = last;
  double t = 0;
  FILE *dump = argc > 6 ? fopen(argv[6], "w") : NULL;
  for (i = 1; i <= iters; i++) {
    double t0 = now();
    reiterate_assembly(cons, i, maln, fsdb, a, front, back, anc, rcanc);
    pop_smp_from_FSDB(fsdb, PSSM_DEPTH);
    cull_maln_from_fsdb(culled, fsdb, 0, 0, DEF_S, DEF_N);
    culled->fpsm = anc; culled->rpsm = rcanc;
    sort_aln_frags(culled);
    cons = consensus_assembly_string(culled);
    t += now() - t0;
    if (dump) {
      int k;
      fprintf(dump, "I %d %s\n", i, cons);
      for (k = 0; k < (int)fsdb->num_fss; k++) fprintf(dump, "R %d %d %d\n", fsdb->fss[k]->score, fsdb->fss[k]->as, fsdb->fss[k]->ae);
    }
  }
  if (dump) fclose(dump);
  printf("reads %d iterations %d seconds %.6f cons_len %d\n", n, iters, t, (int)strlen(cons));
  return 0;
}
