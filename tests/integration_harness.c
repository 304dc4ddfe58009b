/* integration_harness.c -- TEST INFRASTRUCTURE for tests/test_integration_snippets.py.
 *
 * INTEGRATION.md section 2 shows, as C fragments, the patch a maintainer of the reference would make to
 * src/mia_main.c:875,878,931-963.  This file is the translation unit those fragments are pasted into VERBATIM (the
 * test replaces every @SNIPPET n@ marker by the n-th ```c block of that section), so the document cannot drift from
 * include/mia_hip.h unnoticed: it is compiled with gcc -std=c99 -Wall -Werror, linked against libmia_hip.so, and --
 * on a GPU box -- run on the committed fixture to the oracle's consensus.
 *
 * The structs are stand-ins for the reference's, holding only the fields the patch touches, with the reference's
 * types (/root/reference/src/types.h:61-143,155-158,183-196: `dropped` IS a one-bit signed bit-field there).
 *
 * State file (written by the test from the oracle's state after pass 1):
 *   circular cons_code hard_cut score_cut_set slope intercept n n_slots iterations
 *   <reference string>
 *   31*25 ints (forward matrix), 31*25 ints (reverse-complement matrix)
 *   n lines:  rc strand_known as ae score front_slot back_slot <stored bases>
 *   n_slots ints: dropped
 * Output: one line per iteration, "<mode> <iteration> <consensus>".
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mia_hip.h"

#define PSSM_DEPTH 15
#define INIT_ALN_SEQ_LEN 256

typedef struct alnseq {
  int start, end, score;
  char revcom : 1;
  char trimmed : 1;
  char dropped : 1;
} AlnSeq;
typedef struct alnseq* AlnSeqP;

typedef struct refseq {
  char* seq;
  int seq_len;
  int* gaps;
  int circular;
} RefSeq;
typedef struct refseq* RefSeqP;

typedef struct fragseq {
  char seq[INIT_ALN_SEQ_LEN + 1];
  int seq_len;
  int strand_known;
  int rc;
  int as;
  int ae;
  int score;
  AlnSeqP front_asp;
  AlnSeqP back_asp;
} FragSeq;
typedef struct fragseq* FragSeqP;

typedef struct fragseqdb {
  FragSeqP* fss;
  size_t size;
  size_t num_fss;
} FragSeqDB;
typedef struct fragseqdb* FSDB;

typedef struct pssm {
  int sm[2 * PSSM_DEPTH + 1][5][5];
  int depth;
} PSSM;
typedef struct pssm* PSSMP;

typedef struct map_alignment {
  RefSeqP ref;
  PSSMP fpsm, rpsm;
  int num_aln_seqs;
  int size;
  int cons_code;
  AlnSeqP* AlnSeqArray;
} MapAlignment;
typedef struct map_alignment* MapAlignmentP;

/* position of the pointer in maln->AlnSeqArray before sort_aln_frags, -1 for NULL */
static int64_t slot_index_of(MapAlignmentP maln, AlnSeqP p) {
  int k;
  if (!p) return -1;
  for (k = 0; k < maln->num_aln_seqs; k++) if (maln->AlnSeqArray[k] == p) return k;
  return -1;
}

struct state {
  int circular, Hard_cut, SCORE_CUT_SET, iterations;
  double slope, intercept;
  char* cons;
  PSSMP ancsubmat, rcancsubmat;
  FSDB fsdb;
  MapAlignmentP maln;
};

static void die(const char* what) { fprintf(stderr, "integration_harness: %s\n", what); exit(2); }

static void load(const char* path, struct state* s) {
  FILE* f = fopen(path, "r");
  long n, n_slots, k;
  int cons_code, j;
  size_t cap = 1 << 22;
  if (!f) die("cannot open the state file");
  s->ancsubmat = calloc(1, sizeof(PSSM)); s->rcancsubmat = calloc(1, sizeof(PSSM));
  s->fsdb = calloc(1, sizeof(FragSeqDB)); s->maln = calloc(1, sizeof(MapAlignment));
  s->maln->ref = calloc(1, sizeof(RefSeq));
  if (fscanf(f, "%d %d %d %d %lf %lf %ld %ld %d", &s->circular, &cons_code, &s->Hard_cut, &s->SCORE_CUT_SET, &s->slope, &s->intercept, &n, &n_slots,
             &s->iterations) != 9) die("header");
  s->cons = malloc(cap);
  if (fscanf(f, "%4194000s", s->cons) != 1) die("reference");
  for (j = 0; j < 31 * 25; j++) if (fscanf(f, "%d", &s->ancsubmat->sm[0][0][0] + j) != 1) die("matrix");
  for (j = 0; j < 31 * 25; j++) if (fscanf(f, "%d", &s->rcancsubmat->sm[0][0][0] + j) != 1) die("matrix");
  s->ancsubmat->depth = s->rcancsubmat->depth = PSSM_DEPTH;
  s->maln->cons_code = cons_code;
  s->maln->num_aln_seqs = (int)n_slots;
  s->maln->AlnSeqArray = calloc((size_t)n_slots + 1, sizeof(AlnSeqP));
  for (k = 0; k < n_slots; k++) s->maln->AlnSeqArray[k] = calloc(1, sizeof(AlnSeq));
  s->fsdb->fss = calloc((size_t)n + 1, sizeof(FragSeqP));
  s->fsdb->num_fss = s->fsdb->size = (size_t)n;
  for (k = 0; k < n; k++) {
    FragSeqP fs = calloc(1, sizeof(FragSeq));
    long front, back;
    if (fscanf(f, "%d %d %d %d %d %ld %ld %256s", &fs->rc, &fs->strand_known, &fs->as, &fs->ae, &fs->score, &front, &back, fs->seq) != 8) die("read");
    fs->seq_len = (int)strlen(fs->seq);
    fs->front_asp = front >= 0 ? s->maln->AlnSeqArray[front] : NULL;
    fs->back_asp = back >= 0 ? s->maln->AlnSeqArray[back] : NULL;
    s->fsdb->fss[k] = fs;
  }
  for (k = 0; k < n_slots; k++) { if (fscanf(f, "%d", &j) != 1) die("dropped"); s->maln->AlnSeqArray[k]->dropped = j ? 1 : 0; }
  fclose(f);
}

/* the loop of src/mia_main.c:878-963 with the step-wise replacements of INTEGRATION.md section 2 */
static int run_stepwise(struct state* st) {
  PSSMP ancsubmat = st->ancsubmat, rcancsubmat = st->rcancsubmat;
  FSDB fsdb = st->fsdb;
  MapAlignmentP maln = st->maln;
  char* cons = st->cons;
  const int circular = st->circular, Hard_cut = st->Hard_cut, SCORE_CUT_SET = st->SCORE_CUT_SET;
  const double slope = st->slope, intercept = st->intercept;
  size_t i;
  int iter;
/*@SNIPPET 0@*/
  for (iter = 1; iter <= st->iterations; iter++) {
/*@SNIPPET 1@*/
    {
/*@SNIPPET 2@*/
    }
    {
/*@SNIPPET 3@*/
      printf("stepwise %d %s\n", iter, assembly_cons);
      if (clen != (int64_t)strlen(assembly_cons)) die("consensus length");
      cons = assembly_cons;
    }
  }
  mia_hip_destroy(gpu);
  return 0;
}

/* ... and with the whole loop body as one call */
static int run_onecall(struct state* st) {
  PSSMP ancsubmat = st->ancsubmat, rcancsubmat = st->rcancsubmat;
  FSDB fsdb = st->fsdb;
  MapAlignmentP maln = st->maln;
  char* cons = st->cons;
  const int circular = st->circular, Hard_cut = st->Hard_cut, SCORE_CUT_SET = st->SCORE_CUT_SET;
  const double slope = st->slope, intercept = st->intercept;
  size_t i;
  int iter;
/*@SNIPPET 0@*/
  for (iter = 1; iter <= st->iterations; iter++) {
/*@SNIPPET 4@*/
    printf("onecall %d %s\n", iter, assembly_cons);
    if (clen != (int64_t)strlen(assembly_cons)) die("consensus length");
    cons = assembly_cons;
  }
  {
/*@SNIPPET 5@*/
    for (i = 0; i < n; i++) if (fsdb->fss[i]->strand_known && (rstart[i] < 0 || dF[i] > 1 || dB[i] > 1)) die("scripts / dropped marks");
    (void)cols;
  }
  (void)len;
  mia_hip_destroy(gpu);
  return 0;
}

int main(int argc, char** argv) {
  struct state s;
  if (argc != 3) die("usage: integration_harness stepwise|onecall <state file>");
  load(argv[2], &s);
  return strcmp(argv[1], "onecall") == 0 ? run_onecall(&s) : run_stepwise(&s);
}
