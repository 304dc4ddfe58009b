/* mia_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see mia_oracle.h).
 *
 * Plain-C restatement of the MIA hot path.  Data lives in flat arrays and
 * slot indices instead of the reference's pointer graph; the arithmetic,
 * tie-breaking and the reference's slot-recycling quirks are reproduced
 * exactly (each function cites the reference lines it follows).
 */
#include "mia_oracle.h"

#include <ctype.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ */
/* small helpers                                                        */
/* ------------------------------------------------------------------ */

static int gap_pen(int len) { return ORA_GOP + ORA_GEP * len; } /* GOP+GEP*len, src/params.h:26-27 */

/* src/map_align.c:16-29, src/mia.c:1054-1082,1243-1268: only upper-case ACGT are bases */
int ora_base_code(char b) {
  switch (b) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 4;
  }
}

/* src/map_align.c:418-432: IUPAC-aware complement, lower case kept, '-' kept,
 * anything unknown -> 'N' (with a message on stderr in the reference) */
char ora_revcom_char(char b) {
  static const char comp[26] = {'T', 'V', 'G', 'H', 0,   0,   'C', 'D', 0,   0,   'M', 0,   'K',
                                'N', 0,   0,   0,   'Y', 'S', 'A', 'A', 'B', 'W', 'X', 'R', 0};
  char r = 0;
  if (b == '-') return '-';
  if (b >= 'A' && b <= 'Z') r = comp[b - 'A'];
  else if (b >= 'a' && b <= 'z') {
    r = comp[b - 'a'];
    if (r) r = (char)(r + 32);
    else r = 32; /* tbl[..] + 32 with tbl == 0 gives ' ' (non-zero) in the reference */
  }
  if (r) return r;
  return 'N';
}

/* src/pssm.c:38-46 (and sub_mat_score :6-27): first test wins for short reads */
int ora_sm_depth(int row, int len) {
  if (row < ORA_PSSM_DEPTH) return row;
  if (len - (row + 1) < ORA_PSSM_DEPTH) return 2 * ORA_PSSM_DEPTH - (len - (row + 1));
  return ORA_PSSM_DEPTH;
}

/* src/pssm.c:96-126: 200 / -600, read-N column -100, then ref-other row -10 overrides */
void ora_pssm_flat(ora_pssm *p) {
  int d, i, j;
  for (d = 0; d <= 2 * ORA_PSSM_DEPTH; d++) {
    for (i = 0; i < 5; i++) {
      for (j = 0; j < 4; j++) p->sm[d][i][j] = (i == j) ? 200 : -600;
      p->sm[d][i][4] = -100;
    }
    for (j = 0; j < 5; j++) p->sm[d][4][j] = -10;
  }
}

/* src/pssm.c:53-91: rc.sm[30-d][3-i][3-j] = sm[d][i][j]; index 4 stays put */
void ora_pssm_revcom(const ora_pssm *in, ora_pssm *out) {
  int d, i, j;
  for (d = 0; d <= 2 * ORA_PSSM_DEPTH; d++) {
    int rd = 2 * ORA_PSSM_DEPTH - d;
    for (i = 0; i < 5; i++)
      for (j = 0; j < 5; j++) {
        int si = (i < 4) ? 3 - i : 4, sj = (j < 4) ? 3 - j : 4;
        out->sm[rd][i][j] = in->sm[d][si][sj];
      }
  }
}

/* src/io.c:408-503: 31 blocks "# Matrix for position..." + 4 rows of 4 tab ints
 * + blank; column 4 := -100 (N_SCORE), row 4 := -10 (NR_SCORE) */
int ora_pssm_read(const char *path, ora_pssm *p) {
  FILE *f = fopen(path, "r");
  char line[4096];
  int d, i, j;
  if (!f) return 0;
  for (d = 0; d <= 2 * ORA_PSSM_DEPTH; d++) {
    if (!fgets(line, sizeof line, f) || !strstr(line, "# Matrix for position")) { fclose(f); return 0; }
    if (d == ORA_PSSM_DEPTH && !strstr(line, "# Matrix for position: MIDDLE")) { fclose(f); return 0; }
    for (i = 0; i < 4; i++) {
      if (!fgets(line, sizeof line, f)) { fclose(f); return 0; }
      sscanf(line, "%d\t%d\t%d\t%d", &p->sm[d][i][0], &p->sm[d][i][1], &p->sm[d][i][2], &p->sm[d][i][3]);
      p->sm[d][i][4] = -100;
    }
    for (j = 0; j < 5; j++) p->sm[d][4][j] = -10;
    if (!fgets(line, sizeof line, f)) line[0] = 0; /* blank separator */
  }
  fclose(f);
  return 1;
}

/* ------------------------------------------------------------------ */
/* a1/a5/a6: the DP                                                     */
/* ------------------------------------------------------------------ */

/* src/mia.c:740-981.  S,T are len2 x len1 row-major.  The two running
 * arg-maxima of the reference (best_gap_col, reset per row; best_gap_row[],
 * persistent) are kept as "keys" S + GEP*index, which orders candidates exactly
 * as the reference's pairwise comparisons do (strictly-greater replaces, so the
 * earliest candidate wins ties). */
static void dp_fill(const unsigned char *c1, int n1, const unsigned char *c2, int n2,
                    const unsigned char *mask, const ora_pssm *pm, int sg5, int *S, int *T) {
  int r, c;
  int *colkey = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
  int *colrow = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
  /* row 0: plain substitution score, no penalty (src/mia.c:769-785) */
  for (c = 0; c < n1; c++) {
    int open = mask ? mask[c] : 1;
    S[c] = open ? pm->sm[0][c1[c]][c2[0]] : ORA_HIM;
    T[c] = 0;
    colkey[c] = S[c]; /* best_gap_row[c] = 0 */
    colrow[c] = 0;
  }
  for (r = 1; r < n2; r++) {
    const int d = ora_sm_depth(r, n2);
    const int *prev = S + (size_t)(r - 1) * n1;
    const int *prev2 = (r >= 2) ? S + (size_t)(r - 2) * n1 : NULL;
    int *cur = S + (size_t)r * n1, *tr = T + (size_t)r * n1;
    const int fresh = sg5 ? -gap_pen(r + 1) : 0; /* src/mia.c:877-880 */
    int rowkey, rowcol;
    if (n1 <= 0) continue;
    /* column 0 (src/mia.c:799-822) */
    if (mask ? mask[0] : 1) {
      cur[0] = pm->sm[d][c1[0]][c2[r]];
      if (sg5) cur[0] -= gap_pen(r + 1);
    } else cur[0] = ORA_HIM;
    tr[0] = 0;
    rowkey = prev[0]; /* best_gap_col = 0 (src/mia.c:825) */
    rowcol = 0;
    for (c = 1; c < n1; c++) {
      int sub, diag, gapc, gapr, best;
      if (mask && !mask[c]) { cur[c] = ORA_HIM; tr[c] = 0; continue; } /* :967-970 */
      sub = pm->sm[d][c1[c]][c2[r]];
      if (c >= 2) { /* :838-847 */
        int k = prev[c - 2] + ORA_GEP * (c - 2);
        if (k > rowkey) { rowkey = k; rowcol = c - 2; }
        gapc = rowkey - ORA_GOP - ORA_GEP * (c - 1);
      } else gapc = ORA_HIM;
      if (r >= 2) { /* :856-865 */
        int k = prev2[c - 1] + ORA_GEP * (r - 2);
        if (k > colkey[c - 1]) { colkey[c - 1] = k; colrow[c - 1] = r - 2; }
        gapr = colkey[c - 1] - ORA_GOP - ORA_GEP * (r - 1);
      } else gapr = ORA_HIM;
      diag = prev[c - 1];
      /* :910-948 (hp branch is out of scope, both hp scores stay HIM) */
      if (fresh > diag && fresh > gapc && fresh > gapr && fresh > ORA_HIM) {
        cur[c] = fresh; /* substitution score is NOT added */
        tr[c] = c;
      } else if (diag >= gapc && diag >= gapr && diag >= ORA_HIM) {
        cur[c] = sub + diag;
        tr[c] = 0;
      } else if (gapc >= gapr && gapc >= ORA_HIM) {
        cur[c] = sub + gapc;
        tr[c] = rowcol;
      } else if (gapr >= ORA_HIM) {
        cur[c] = sub + gapr;
        tr[c] = -colrow[c - 1];
      } else {
        /* everything below HIM: the reference falls into the hp branches,
         * which add HIM and store a trace from NULL hp arrays when hp==0
         * (undefined there).  Unreachable for in-range scores. */
        best = ORA_HIM;
        cur[c] = sub + best;
        tr[c] = 0;
      }
    }
  }
  free(colkey);
  free(colrow);
}

/* trim_frag, src/mia.c:1318-1368 */
void ora_trim(const char *read, int read_len, const char *adapter, int *trimmed, int *trim_point, ora_aln *res) {
  ora_pssm flat;
  int len1 = read_len, len2 = (int)strlen(adapter), i, r, c, best = INT_MIN, aer = 0, aec, abr, abc;
  unsigned char *c1, *c2;
  int *S, *T;
  *trimmed = 0;
  *trim_point = 0;
  if (len1 <= 0 || len2 <= 0) return;
  ora_pssm_flat(&flat);
  c1 = (unsigned char *)malloc((size_t)len1);
  c2 = (unsigned char *)malloc((size_t)len2);
  for (i = 0; i < len1; i++) c1[i] = (unsigned char)ora_base_code(read[i]);
  for (i = 0; i < len2; i++) c2[i] = (unsigned char)ora_base_code(adapter[i]);
  S = (int *)malloc(sizeof(int) * (size_t)len1 * len2);
  T = (int *)malloc(sizeof(int) * (size_t)len1 * len2);
  dp_fill(c1, len1, c2, len2, NULL, &flat, 1, S, T);
  /* last column, all rows, first maximum (src/mia.c:1346-1353) */
  aec = len1 - 1;
  for (r = 0; r < len2; r++)
    if (S[(size_t)r * len1 + aec] > best) { best = S[(size_t)r * len1 + aec]; aer = r; }
  /* find_align_begin, src/mia.c:612-637 */
  r = aer; c = aec;
  for (;;) {
    int t = T[(size_t)r * len1 + c];
    if (t == c || t == -r) break;
    if (t == 0) { r--; c--; }
    else if (t < 0) { r = -t; c--; }
    else { c = t; r--; }
  }
  abr = r; abc = c;
  if (best >= 1000 || best >= (aer - abr + 1) * 200) { *trimmed = 1; *trim_point = abc - 1; }
  if (res) { res->best = best; res->aec = aec; res->aer = aer; res->abc = abc; res->abr = abr; }
  free(S); free(T); free(c1); free(c2);
}

int ora_align(const char *seq1, int len1, const char *seq2, int len2, const unsigned char *mask,
              const ora_pssm *pm, int sg5, ora_aln *res, char *ref_gapped, char *frag_gapped,
              int *S_out, int *T_out) {
  unsigned char *c1, *c2;
  int *S, *T;
  int i, r, c, best;
  if (len2 <= 0 || len1 <= 0) {
    res->best = INT_MIN; res->aec = res->aer = res->abc = res->abr = 0;
    return -1;
  }
  c1 = (unsigned char *)malloc((size_t)len1);
  c2 = (unsigned char *)malloc((size_t)len2);
  for (i = 0; i < len1; i++) c1[i] = (unsigned char)ora_base_code(seq1[i]);
  for (i = 0; i < len2; i++) c2[i] = (unsigned char)ora_base_code(seq2[i]);
  S = S_out ? S_out : (int *)malloc(sizeof(int) * (size_t)len1 * len2);
  T = T_out ? T_out : (int *)malloc(sizeof(int) * (size_t)len1 * len2);
  dp_fill(c1, len1, c2, len2, mask, pm, sg5, S, T);

  /* max_sg_score, src/mia.c:1278-1302: last row, first maximum wins */
  r = len2 - 1;
  best = INT_MIN;
  res->aec = 0;
  for (c = 0; c < len1; c++)
    if (S[(size_t)r * len1 + c] > best) { best = S[(size_t)r * len1 + c]; res->aec = c; }
  res->aer = r;
  res->best = best;

  /* find_align_begin + populate_pwaln_to_begin, src/mia.c:612-637,1440-1497.
   * T==0 is always read as a diagonal step (also when it encodes a gap that
   * started in column/row 0 -- the reference's overload). */
  {
    char rbuf[2 * ORA_MAX_ALN + 8], fbuf[2 * ORA_MAX_ALN + 8];
    int n = 0, overflow = 0;
    c = res->aec;
    for (;;) {
      int t = T[(size_t)r * len1 + c];
      if (t == c || t == -r) break;
      if (n < 2 * ORA_MAX_ALN) { rbuf[n] = seq1[c]; fbuf[n] = seq2[r]; n++; } else overflow = 1;
      if (t == 0) { r--; c--; }
      else if (t < 0) {
        int nr = -t;
        r--; c--;
        while (r > nr) {
          if (n < 2 * ORA_MAX_ALN) { rbuf[n] = '-'; fbuf[n] = seq2[r]; n++; } else overflow = 1;
          r--;
        }
      } else {
        int nc = t;
        r--; c--;
        while (c > nc) {
          if (n < 2 * ORA_MAX_ALN) { rbuf[n] = seq1[c]; fbuf[n] = '-'; n++; } else overflow = 1;
          c--;
        }
      }
    }
    if (n < 2 * ORA_MAX_ALN) { rbuf[n] = seq1[c]; fbuf[n] = seq2[r]; n++; } else overflow = 1;
    res->abc = c;
    res->abr = r;
    if (n > ORA_MAX_ALN) overflow = 1; /* reference buffer is 512 chars: UB beyond */
    if (ref_gapped && frag_gapped) {
      int m = overflow ? 0 : n;
      for (i = 0; i < m; i++) { ref_gapped[i] = rbuf[m - 1 - i]; frag_gapped[i] = fbuf[m - 1 - i]; }
      ref_gapped[m] = 0;
      frag_gapped[m] = 0;
    }
  }
  if (!S_out) free(S);
  if (!T_out) free(T);
  free(c1);
  free(c2);
  return 0;
}

/* ------------------------------------------------------------------ */
/* a15/a16: consensus primitives                                        */
/* ------------------------------------------------------------------ */

/* src/map_align.c:229-263 */
void ora_add_base(char b, ora_counts *bc, const ora_pssm *pm, int pssm_code) {
  int bi, d;
  switch (b) {
    case 'A': bc->As++; break;
    case 'C': bc->Cs++; break;
    case 'G': bc->Gs++; break;
    case 'T': bc->Ts++; break;
    case '-': bc->gaps++; break;
    default: break;
  }
  bc->cov++;
  if (b == '-') return;
  bi = ora_base_code(b);
  d = pssm_code - 'A';
  bc->scoreA += pm->sm[d][0][bi];
  bc->scoreC += pm->sm[d][1][bi];
  bc->scoreG += pm->sm[d][2][bi];
  bc->scoreT += pm->sm[d][3][bi];
}

/* src/map_align.c:294-391: later base wins ties (>=), A<C<G<T */
char ora_find_consensus(const ora_counts *bc, int cons_code) {
  int top, second;
  char base = 'A';
  if (bc->cov == 0) return 'N';
  if (((double)bc->gaps / (double)bc->cov) >= (double)(50 / 100.0)) return '-';
  top = bc->scoreA;
  second = INT_MIN;
  if (bc->scoreC >= top) { second = top; top = bc->scoreC; base = 'C'; } else second = bc->scoreC;
  if (bc->scoreG >= top) { second = top; top = bc->scoreG; base = 'G'; }
  else if (bc->scoreG >= second) second = bc->scoreG;
  if (bc->scoreT >= top) { second = top; top = bc->scoreT; base = 'T'; }
  else if (bc->scoreT >= second) second = bc->scoreT;
  if (cons_code == 2) return (top >= 0 || (top - 2400) > second) ? base : 'N';
  return (top >= -399) ? base : 'N';
}

/* ------------------------------------------------------------------ */
/* a9: Myers O(ND)                                                      */
/* ------------------------------------------------------------------ */

/* src/myers_align.h:40-67 */
static int iupac_bits(char x) {
  switch (x & ~32) {
    case 'A': return 1;
    case 'C': return 2;
    case 'G': return 4;
    case 'T': case 'U': return 8;
    case 'S': return 6;
    case 'W': return 9;
    case 'R': return 5;
    case 'Y': return 10;
    case 'K': return 12;
    case 'M': return 3;
    case 'B': return 14;
    case 'D': return 13;
    case 'H': return 11;
    case 'V': return 7;
    case 'N': return 15;
    default: return 0;
  }
}

/* src/myers_align.c:10-99.  Furthest-reaching x on diagonal k = x - y
 * (x indexes seq_b, y indexes seq_a); substitutions, insertions and deletions
 * all cost 1.  Diagonals outside [-len_a, len_b] are never computed by the
 * reference and would be read uninitialised when d exceeds a sequence length;
 * here they count as "unreachable" (the reference's callers keep
 * maxd <= len/10, src/ccheck.cc:477). */
unsigned ora_myers_diff(const char *seq_a, int mode, const char *seq_b, int maxd, char *bt_a) {
  const int la = (int)strlen(seq_a), lb = (int)strlen(seq_b);
  const int NEG = -(1 << 29);
  int **V;
  int d, k, x, y, found_d = -1, found_k = 0;
  unsigned result = 0xFFFFFFFFu;
  if (maxd > la + lb) maxd = la + lb;
  if (maxd <= 0) return result;
  V = (int **)calloc((size_t)maxd, sizeof(int *));
  for (d = 0; d < maxd && found_d < 0; d++) {
    int lo = (-d > -la) ? -d : -la, hi = (d < lb) ? d : lb;
    int *row = (int *)malloc(sizeof(int) * (size_t)(2 * d + 1));
    V[d] = row;
    for (k = -d; k <= d; k++) row[k + d] = NEG;
    for (k = lo; k <= hi; k++) {
      if (d == 0) x = 0;
      else {
        const int *p = V[d - 1];
        int pd = d - 1;
        int sub = (k >= -pd && k <= pd) ? p[k + pd] + 1 : NEG;       /* mismatch: same diagonal */
        int ins = (k - 1 >= -pd && k - 1 <= pd) ? p[k - 1 + pd] + 1 : NEG; /* consume seq_b */
        int del = (k + 1 >= -pd && k + 1 <= pd) ? p[k + 1 + pd] : NEG;     /* consume seq_a */
        x = sub;
        if (ins > x) x = ins;
        if (del > x) x = del;
      }
      y = x - k;
      while (x >= 0 && y >= 0 && x < lb && y < la && (iupac_bits(seq_b[x]) & iupac_bits(seq_a[y]))) { x++; y++; }
      row[k + d] = x;
      if ((mode == 1 || y == la) && (mode == 2 || x == lb)) { found_d = d; found_k = k; break; }
    }
  }
  if (found_d >= 0) {
    result = (unsigned)found_d;
    if (bt_a) {
      /* backtrace of seq_a's row only (the reference leaves bt_b unterminated,
       * src/myers_align.c:44-45); preference order sub, ins, del as :47-73 */
      int dd = found_d;
      char *buf = (char *)malloc((size_t)(la + lb + found_d + 4));
      int n = 0, i;
      k = found_k;
      x = V[dd][k + dd];
      y = x - k;
      while (dd != 0) {
        const int *p = V[dd - 1];
        int pd = dd - 1;
        if (k != -dd && k != dd && x == p[k + pd] + 1) { dd--; x--; y--; buf[n++] = seq_a[y]; }
        else if (k > -dd + 1 && x == p[k - 1 + pd] + 1) { x--; k--; dd--; buf[n++] = '-'; }
        else if (k < dd - 1 && x == p[k + 1 + pd]) { k++; y--; dd--; buf[n++] = seq_a[y]; }
        else { x--; y--; buf[n++] = seq_a[y]; }
      }
      while (x > 0) { x--; buf[n++] = seq_a[x]; }
      for (i = 0; i < n; i++) bt_a[i] = buf[n - 1 - i];
      bt_a[n] = 0;
      free(buf);
    }
  }
  for (d = 0; d < maxd; d++) free(V[d]);
  free(V);
  return result;
}
