/* ref_dp_driver.c -- TEST INFRASTRUCTURE (ours, not reference code).
 *
 * A line-oriented driver linked against the REAL reference objects
 * (/root/reference/src/{mia,pssm,map_align,...}.c, see oracle/Makefile.ref).
 * It exposes the reference's finest-grained seam:
 *   dyn_prog / max_sg_score / find_align_begin / populate_pwaln_to_begin
 *   (src/mia.h:84,114,181,209) and find_consensus (src/map_align.h)
 * so that golden unit vectors can be dumped for the oracle restatement and
 * the HIP kernels.
 *
 * stdin lines:
 *   D <matrix> <rc> <sg5> <sg3> <seq1> <seq2> <mask|*>
 *       matrix = "flat" or a PSSM file path; rc=1 -> revcom_submat of it
 *       mask   = string of '0'/'1' of length len1, or '*' for all ones
 *     -> "D best aec aer abc abr hashS hashT ref_gapped frag_gapped"
 *   C <cons_code> As Cs Gs Ts gaps cov scoreA scoreC scoreG scoreT
 *     -> "C <char>"
 *   T <adapter> <read>
 *       trim_frag (src/mia.c:1318-1368) set up as main() does (src/mia_main.c:694-717):
 *       flat matrix, adapter = seq2 (rows), sg5 = 1, sg3 = 0
 *     -> "T trimmed trim_point aec aer abc abr"   (trim_point is -999 when not trimmed)
 */
#include "mia.h"
#include <stdint.h>

static uint64_t fnv(uint64_t h, int v) {
  unsigned u = (unsigned)v; int i;
  for (i = 0; i < 4; i++) { h ^= (u >> (8 * i)) & 0xff; h *= 1099511628211ULL; }
  return h;
}

#define MAXL1 (1 << 17)
static char line[4 * MAXL1], mspec[2048], s1[MAXL1 + 1], s2[1024], mask[MAXL1 + 1];

int main(void) {
  PSSMP flat = init_flatsubmat();
  PSSMP cur = NULL, currc = NULL; char curspec[2048] = "";
  AlignmentP a = init_alignment(INIT_ALN_SEQ_LEN, MAXL1, 0, 0);
  PWAlnFragP pw = (PWAlnFragP)malloc(sizeof(PWAlnFrag));
  while (fgets(line, sizeof line, stdin)) {
    if (line[0] == 'C') {
      BaseCounts bc; int cc; char ch;
      sscanf(line + 1, "%d %d %d %d %d %d %d %d %d %d %d", &cc, &bc.As, &bc.Cs, &bc.Gs, &bc.Ts,
             &bc.gaps, &bc.cov, &bc.scoreA, &bc.scoreC, &bc.scoreG, &bc.scoreT);
      ch = find_consensus(&bc, cc);
      printf("C %c\n", ch);
      continue;
    }
    if (line[0] == 'T') {
      static char ad[256]; static FragSeq fsq; static AlignmentP ta = NULL;
      if (sscanf(line + 1, "%255s %300s", ad, fsq.seq) != 2) { fprintf(stderr, "bad T line\n"); return 1; }
      if (!ta) ta = init_alignment(INIT_ALN_SEQ_LEN, INIT_ALN_SEQ_LEN, 0, 0);
      ta->submat = flat; ta->seq2 = ad; ta->len2 = strlen(ad); pop_s2c_in_a(ta);
      ta->sg5 = 1; ta->sg3 = 0;
      fsq.trimmed = 0; fsq.trim_point = -999;
      trim_frag(&fsq, ad, ta);
      printf("T %d %d %d %d %d %d\n", fsq.trimmed, fsq.trimmed ? fsq.trim_point : -999, ta->aec, ta->aer, ta->abc, ta->abr);
      continue;
    }
    if (line[0] != 'D') continue;
    int rc, sg5, sg3;
    if (sscanf(line + 1, "%2047s %d %d %d %131072s %1023s %131072s", mspec, &rc, &sg5, &sg3, s1, s2, mask) != 7) {
      fprintf(stderr, "bad line\n"); return 1;
    }
    PSSMP m;
    if (!strcmp(mspec, "flat")) m = flat;
    else {
      if (strcmp(mspec, curspec)) { cur = read_pssm(mspec); strcpy(curspec, mspec); currc = NULL; }
      m = cur;
    }
    if (rc) { if (m == flat) m = revcom_submat(flat); else { if (!currc) currc = revcom_submat(cur); m = currc; } }
    a->seq1 = s1; a->len1 = strlen(s1);
    a->seq2 = s2; a->len2 = strlen(s2);
    a->submat = m; a->sg5 = sg5; a->sg3 = sg3; a->rc = rc;
    if (mask[0] == '*') memset(a->align_mask, 1, a->len1);
    else { int i; for (i = 0; i < a->len1; i++) a->align_mask[i] = (mask[i] == '1'); }
    pop_s1c_in_a(a); pop_s2c_in_a(a);
    dyn_prog(a);
    int best = max_sg_score(a);
    find_align_begin(a);
    populate_pwaln_to_begin(a, pw);
    uint64_t hS = 1469598103934665603ULL, hT = hS; int r, c;
    for (r = 0; r < a->len2; r++) for (c = 0; c < a->len1; c++) {
      hS = fnv(hS, a->m->mat[r][c].score); hT = fnv(hT, a->m->mat[r][c].trace);
    }
    printf("D %d %d %d %d %d %016llx %016llx %s %s\n", best, a->aec, a->aer, a->abc, a->abr,
           (unsigned long long)hS, (unsigned long long)hT, pw->ref_seq, pw->frag_seq);
  }
  return 0;
}
