/* mia_oracle_flow.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see mia_oracle.h).
 *
 * The assembler state machine around the DP: pass 1 (sg_align), the
 * per-iteration driver (reiterate_assembly), pop_smp, cull, consensus, the
 * .maln writer and the FASTA/FASTQ readers, restated over flat arrays.
 */
#include "mia_oracle.h"

#include <ctype.h>
#include <limits.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAX_KMER_POS 128  /* src/params.h:75 */
#define KMER_SATURATE 128 /* src/params.h:77 */
#define MASK_BUFFER 10    /* src/params.h:78 */

typedef struct { unsigned n; unsigned pos[MAX_KMER_POS]; } kpl; /* src/types.h:276-282 */

struct ora_state {
  ora_opts opt;
  ora_pssm anc, rcanc;
  /* RefSeq (src/types.h:82-96) */
  char ref_id[ORA_MAX_ID + 4], ref_desc[ORA_MAX_DESC + 4];
  char *seq, *rcseq;
  int seq_len, size, wrap_seq_len, circular;
  int *gaps;
  /* maln->AlnSeqArray: slots are recycled from 0 every iteration */
  ora_alnseq **slots;
  int slots_size, num_aln_seqs;
  /* culled_maln */
  int *culled;
  int culled_n, culled_cap, culled_size_reported;
  /* fsdb */
  ora_frag *fss;
  int num_fss, fss_cap;
  /* k-mer tables over the wrapped fwd / rc reference */
  kpl **fk, **rk;
  /* pass-1 column masks + lengths of the two alignment objects */
  unsigned char *fw_mask, *rc_mask;
  int fw_len1;
  int ref_prepared;
  int threads; /* ora_set_threads: host threads for the alignments of an iteration (results are merged in fsdb order) */
};

void ora_opts_default(ora_opts *o) {
  memset(o, 0, sizeof *o);
  o->iterate = 1;
  o->cons_code = 1;
  o->slope = 200.0;   /* DEF_S */
  o->intercept = 0.0; /* DEF_N */
  o->kmer_len = -1;
  strcpy(o->adapter, "GTCAGACACGCAACAGGGGATAGGCAAGGCACACAGGGGATAGG");   /* neand_adapt, src/mia_main.c:462,466 */
}

ora_state *ora_new(const ora_opts *o, const ora_pssm *anc) {
  ora_state *st = (ora_state *)calloc(1, sizeof *st);
  st->opt = *o;
  st->anc = *anc;
  ora_pssm_revcom(&st->anc, &st->rcanc);
  return st;
}

static void free_slot(ora_alnseq *a) {
  int j;
  if (!a) return;
  for (j = 0; j <= ORA_MAX_ALN; j++) free(a->ins[j]);
  free(a);
}

void ora_free(ora_state *st) {
  int i;
  if (!st) return;
  for (i = 0; i < st->slots_size; i++) free_slot(st->slots[i]);
  free(st->slots);
  free(st->culled);
  free(st->fss);
  free(st->seq);
  free(st->rcseq);
  free(st->gaps);
  free(st->fw_mask);
  free(st->rc_mask);
  if (st->fk) {
    size_t n = (size_t)1 << (2 * st->opt.kmer_len), k;
    for (k = 0; k < n; k++) { free(st->fk[k]); free(st->rk[k]); }
    free(st->fk);
    free(st->rk);
  }
  free(st);
}

/* ------------------------------------------------------------------ */
/* reference handling                                                   */
/* ------------------------------------------------------------------ */

void ora_set_ref(ora_state *st, const char *id, const char *desc, const char *seq) {
  int n = (int)strlen(seq), i;
  strncpy(st->ref_id, id, ORA_MAX_ID); st->ref_id[ORA_MAX_ID] = 0;
  strncpy(st->ref_desc, desc, ORA_MAX_DESC); st->ref_desc[ORA_MAX_DESC] = 0;
  /* read_fasta_ref: buffer starts at INIT_REF_SEQ_LEN and doubles while !(len < size), src/io.c:296-373 */
  st->size = 32768;
  while (!(n < st->size)) st->size *= 2;
  free(st->seq); free(st->rcseq);
  st->seq = (char *)calloc((size_t)st->size * 2 + 1024, 1);
  st->rcseq = (char *)calloc((size_t)st->size * 2 + 1024, 1);
  memcpy(st->seq, seq, (size_t)n);
  st->seq_len = n;
  for (i = 0; i < n; i++) st->rcseq[i] = ora_revcom_char(seq[n - 1 - i]); /* src/io.c:388-399 */
}

/* src/io.c:287-386 */
int ora_load_ref_fasta(ora_state *st, const char *path) {
  FILE *f = fopen(path, "r");
  char id[ORA_MAX_ID + 4], desc[ORA_MAX_DESC + 4];
  char *buf;
  int c, n = 0, cap = 1 << 16, head_done = 0, len = 0;
  if (!f) return 0;
  c = fgetc(f);
  if (c != '>') { fclose(f); return 0; }
  while (!isspace(c = fgetc(f)) && !head_done) {
    if (c == EOF) { fclose(f); return 0; }
    id[len++] = (char)c;
    if (len == ORA_MAX_ID) head_done = 1;
  }
  id[len] = 0;
  len = 0; head_done = 0;
  if (c == '\n') head_done = 1; else c = fgetc(f);
  while (c != '\n' && !head_done) {
    if (c == EOF) { fclose(f); return 0; }
    desc[len++] = (char)c;
    if (len == ORA_MAX_DESC) head_done = 1;
    c = fgetc(f);
  }
  desc[len] = 0;
  buf = (char *)malloc((size_t)cap);
  c = fgetc(f);
  while (c != '>' && c != EOF) {
    if (!isspace(c)) {
      if (n + 2 >= cap) { cap *= 2; buf = (char *)realloc(buf, (size_t)cap); }
      buf[n++] = (char)c;
    }
    c = fgetc(f);
  }
  buf[n] = 0;
  fclose(f);
  ora_set_ref(st, id, desc, buf);
  free(buf);
  return 1;
}

/* add_ref_wrap, src/mia.c:657-689 */
static void add_wrap(ora_state *st) {
  int w = st->seq_len < ORA_MAX_READ ? st->seq_len : ORA_MAX_READ;
  while (st->seq_len + w >= st->size) {
    st->seq = (char *)realloc(st->seq, (size_t)st->size * 4 + 1024);
    if (st->rcseq) st->rcseq = (char *)realloc(st->rcseq, (size_t)st->size * 4 + 1024);
    st->size *= 2;
  }
  memcpy(st->seq + st->seq_len, st->seq, (size_t)w);
  st->seq[st->seq_len + w] = 0;
  if (st->rcseq) { memcpy(st->rcseq + st->seq_len, st->rcseq, (size_t)w); st->rcseq[st->seq_len + w] = 0; }
  st->wrap_seq_len = st->seq_len + w;
  st->circular = 1;
}

/* kmer2inx, src/kmer.c:18-49 (case-insensitive, ACGT only) */
static int kmer_index(const char *s, int k, size_t *out) {
  size_t v = 0;
  int i;
  for (i = 0; i < k; i++) {
    int ch = toupper((unsigned char)s[i]);
    v <<= 2;
    if (ch == 'A') v += 0; else if (ch == 'C') v += 1; else if (ch == 'G') v += 2; else if (ch == 'T') v += 3;
    else return 0;
  }
  *out = v;
  return 1;
}

/* populate_kpa + add_kmer, src/kmer.c:65-107,153-168: at most 128 positions per k-mer, rest dropped */
static void fill_kmers(kpl **tab, const char *seq, int n, int k, int soft_mask) {
  int i, j;
  for (i = 0; i + k <= n; i++) {
    size_t inx;
    if (soft_mask) {
      int up = 1;
      for (j = 0; j < k; j++) if (islower((unsigned char)seq[i + j])) { up = 0; break; }
      if (!up) continue;
    }
    if (!kmer_index(seq + i, k, &inx)) continue;
    if (!tab[inx]) tab[inx] = (kpl *)calloc(1, sizeof(kpl));
    if (tab[inx]->n == MAX_KMER_POS) continue;
    tab[inx]->pos[tab[inx]->n++] = (unsigned)i;
  }
}

/* src/mia_main.c:644-690,719-728 */
void ora_prepare_ref(ora_state *st) {
  int i, msz;
  if (st->opt.circular) add_wrap(st);
  else { st->wrap_seq_len = st->seq_len; st->circular = 0; }
  free(st->gaps);
  st->gaps = (int *)calloc((size_t)st->wrap_seq_len + 1, sizeof(int));
  if (st->opt.kmer_len > 0) {
    size_t n = (size_t)1 << (2 * st->opt.kmer_len);
    st->fk = (kpl **)calloc(n, sizeof(kpl *));
    st->rk = (kpl **)calloc(n, sizeof(kpl *));
    fill_kmers(st->fk, st->seq, st->wrap_seq_len, st->opt.kmer_len, st->opt.soft_mask);
    fill_kmers(st->rk, st->rcseq, st->wrap_seq_len, st->opt.kmer_len, st->opt.soft_mask);
  }
  for (i = 0; i < st->wrap_seq_len; i++) { /* make_ref_upper, src/mia.c:642-648 */
    st->seq[i] = (char)toupper((unsigned char)st->seq[i]);
    st->rcseq[i] = (char)toupper((unsigned char)st->rcseq[i]);
  }
  msz = st->wrap_seq_len + 2 * ORA_MAX_READ;
  st->fw_mask = (unsigned char *)malloc((size_t)msz);
  st->rc_mask = (unsigned char *)malloc((size_t)msz);
  memset(st->fw_mask, 1, (size_t)msz); /* init_alignment, src/mia.c:1007 */
  memset(st->rc_mask, 1, (size_t)msz);
  st->fw_len1 = st->opt.circular ? st->wrap_seq_len : st->seq_len;
  st->ref_prepared = 1;
}

/* ------------------------------------------------------------------ */
/* maln slots                                                           */
/* ------------------------------------------------------------------ */

static ora_alnseq *slot_for_write(ora_state *st) {
  if (st->num_aln_seqs >= st->slots_size) {
    int ns = st->slots_size ? st->slots_size * 2 : 16000, i;
    st->slots = (ora_alnseq **)realloc(st->slots, sizeof(ora_alnseq *) * (size_t)ns);
    for (i = st->slots_size; i < ns; i++) st->slots[i] = NULL;
    st->slots_size = ns;
  }
  if (!st->slots[st->num_aln_seqs]) {
    ora_alnseq *a = (ora_alnseq *)calloc(1, sizeof(ora_alnseq)); /* init: dropped = 0, segment 'n' */
    a->segment = 'n';
    st->slots[st->num_aln_seqs] = a;
  }
  return st->slots[st->num_aln_seqs];
}

typedef struct {
  char frag_id[ORA_MAX_ID + 4], frag_desc[ORA_MAX_DESC + 4];
  char ref_seq[ORA_MAX_ALN + 1], frag_seq[ORA_MAX_ALN + 1];
  int start, end, revcom, trimmed, score, num_inputs;
  char segment;
} pwaln;

/* merge_pwaln_into_maln, src/map_align.c:866-954.  `dropped` is NOT reset. */
static void merge_record(ora_state *st, const pwaln *pw) {
  ora_alnseq *a = slot_for_write(st);
  int i, n = (int)strlen(pw->frag_seq), pos = 0, in_gap = 0, j = 0, span;
  int own_gaps[ORA_MAX_ALN + 2];
  char *ins = NULL;
  strcpy(a->id, pw->frag_id);
  strcpy(a->desc, pw->frag_desc);
  a->score = pw->score; a->start = pw->start; a->end = pw->end;
  a->revcom = pw->revcom; a->trimmed = pw->trimmed; a->segment = pw->segment;
  a->num_inputs = pw->num_inputs;
  own_gaps[0] = 0;
  for (i = 0; i < n; i++) {
    char rc = pw->ref_seq[i], fc = pw->frag_seq[i];
    if (rc == '-') {
      own_gaps[pos]++;
      if (!in_gap) { ins = (char *)malloc(ORA_MAX_ALN + 1); j = 0; }
      ins[j++] = fc;
      in_gap = 1;
    } else {
      if (in_gap) { ins[j] = 0; a->ins[pos] = ins; }
      else a->ins[pos] = NULL;
      a->seq[pos++] = fc;
      own_gaps[pos] = 0;
      in_gap = 0;
    }
  }
  if (in_gap) free(ins); /* trailing insert is never attached (leaked in the reference) */
  a->seq[pos] = 0;
  span = a->end - a->start + 1;
  for (i = 0; i < span && i <= ORA_MAX_ALN; i++) {
    int rp = a->start + i;
    if (rp >= 0 && rp <= st->wrap_seq_len && own_gaps[i] > st->gaps[rp]) st->gaps[rp] = own_gaps[i];
  }
  st->num_aln_seqs++;
}

/* split_pwaln, src/mia.c:1376-1438 */
static void split_record(pwaln *front, pwaln *back, int wrap_point) {
  int ap = 0, rp = front->start, fp = 0, i = 0;
  while (i <= ORA_MAX_ID - 2 && front->frag_id[i]) i++;
  front->frag_id[i] = '_'; front->frag_id[i + 1] = 'f'; front->frag_id[i + 2] = 0;
  strcpy(back->frag_id, front->frag_id);
  back->frag_id[i + 1] = 'b';
  while (rp < wrap_point) {
    if (front->ref_seq[ap] != '-') rp++;
    if (front->frag_seq[ap] != '-') fp++;
    ap++;
  }
  strcpy(back->ref_seq, front->ref_seq + ap);
  strcpy(back->frag_seq, front->frag_seq + ap);
  front->ref_seq[ap] = 0;
  front->frag_seq[ap] = 0;
  back->start = 0;
  back->end = front->end;
  front->end = wrap_point - 1;
  back->segment = 'b';
  front->segment = 'f';
  strcpy(back->frag_desc, front->frag_desc);
  back->revcom = front->revcom; back->trimmed = front->trimmed;
  back->score = front->score; back->num_inputs = front->num_inputs;
  (void)fp;
}

/* ------------------------------------------------------------------ */
/* pass 1                                                               */
/* ------------------------------------------------------------------ */

/* new_kmer_filter, src/kmer.c:239-331 */
static unsigned kmer_filter(ora_state *st, const char *seq, int frag_len) {
  const int k = st->opt.kmer_len, len1 = st->fw_len1;
  unsigned nf = 0, nr = 0;
  int fp;
  if (k < 0) { memset(st->fw_mask, 1, (size_t)len1); return 1; } /* :251-254 (rc mask keeps its init 1s) */
  memset(st->fw_mask, 0, (size_t)len1);
  memset(st->rc_mask, 0, (size_t)len1);
  if (frag_len < k) return 0;
  for (fp = 0; fp + k <= frag_len; fp++) {
    size_t inx;
    unsigned i;
    if (!kmer_index(seq + fp, k, &inx)) continue;
    if (st->fk[inx]) {
      nf += st->fk[inx]->n;
      if (nf >= KMER_SATURATE) memset(st->fw_mask, 1, (size_t)len1);
      for (i = 0; i < st->fk[inx]->n; i++) {
        int rp = (int)st->fk[inx]->pos[i];
        int lo = rp - fp - MASK_BUFFER, hi = rp + (frag_len - fp) + MASK_BUFFER; /* :290-297 */
        if (lo < 0) lo = 0;
        if (hi >= len1) hi = len1 - 1;
        if (hi >= lo) memset(st->fw_mask + lo, 1, (size_t)(hi - lo + 1));
      }
    }
    if (st->rk[inx]) {
      nr += st->rk[inx]->n;
      if (nr >= KMER_SATURATE) memset(st->rc_mask, 1, (size_t)len1);
      for (i = 0; i < st->rk[inx]->n; i++) {
        int rp = (int)st->rk[inx]->pos[i];
        int lo = rp - fp - MASK_BUFFER, hi = rp + frag_len - fp - 1 + MASK_BUFFER; /* :314-322 (one less than fwd) */
        if (lo < 0) lo = 0;
        if (hi >= len1) hi = len1 - 1;
        if (hi >= lo) memset(st->rc_mask + lo, 1, (size_t)(hi - lo + 1));
      }
    }
  }
  return nf + nr;
}

static ora_frag *fsdb_push(ora_state *st) {
  if (st->num_fss == st->fss_cap) {
    st->fss_cap = st->fss_cap ? st->fss_cap * 2 : 16000;
    st->fss = (ora_frag *)realloc(st->fss, sizeof(ora_frag) * (size_t)st->fss_cap);
  }
  memset(&st->fss[st->num_fss], 0, sizeof(ora_frag));
  return &st->fss[st->num_fss++];
}

static void revcom_string(char *s) {
  int n = (int)strlen(s), i;
  for (i = 0; i < n / 2; i++) {
    char a = s[i], b = s[n - 1 - i];
    s[i] = ora_revcom_char(b);
    s[n - 1 - i] = ora_revcom_char(a);
  }
  if (n % 2) s[n / 2] = ora_revcom_char(s[n / 2]);
}

/* c2rcc, src/mia.c:26-30 */
static int to_rc_coord(int c, int len) { return len - (c % len) - 1; }

/* sg_align, src/mia.c:1500-1665 (both strands scored with the forward matrix,
 * src/mia_main.c:788-789) + add_virgin_fs2fsdb, src/fsdb.c:194-231 */
void ora_pass1_read(ora_state *st, const char *id, const char *desc, const char *seq_in) {
  char seq[ORA_MAX_READ + 1];
  char rg_f[ORA_MAX_ALN + 1], fg_f[ORA_MAX_ALN + 1], rg_r[ORA_MAX_ALN + 1], fg_r[ORA_MAX_ALN + 1];
  ora_aln fw, rc, *best;
  pwaln front, back;
  int n = 0, is_rc, as, ae, score, L = st->seq_len, front_slot, back_slot = -1, trimmed = 0, trim_point = 0;
  ora_frag *fs;
  while (seq_in[n] && n < ORA_MAX_READ) { seq[n] = (char)toupper((unsigned char)seq_in[n]); n++; }
  seq[n] = 0;
  if (st->opt.do_trim && n > 0) {
    /* trim_frag (src/mia.c:1318-1368); from here on the read ends at the trim point: the k-mer filter and sg_align
     * use trim_point + 1 bases (src/kmer.c:262-263, src/mia.c:1514-1516), add_virgin_fs2fsdb cuts the string (src/fsdb.c:199-203) */
    ora_trim(seq, n, st->opt.adapter, &trimmed, &trim_point, NULL);
    if (trimmed) {
      if (trim_point + 1 <= 0) return;   /* adapter only: len2 = 0 is undefined in the reference (see the n == 0 note below) */
      n = trim_point + 1;
      seq[n] = 0;
    }
  }
  if (!kmer_filter(st, seq, n)) return;
  if (n == 0) {
    /* dyn_prog on an empty read leaves both scores INT_MIN; sg_align then reads
     * stale matrix cells (src/mia.c:1278-1286).  Not reproducible: skip. */
    return;
  }
  ora_align(st->seq, st->fw_len1, seq, n, st->fw_mask, &st->anc, 1, &fw, rg_f, fg_f, NULL, NULL);
  ora_align(st->rcseq, st->fw_len1, seq, n, st->rc_mask, &st->anc, 1, &rc, rg_r, fg_r, NULL, NULL);
  is_rc = !(fw.best > rc.best); /* :1549-1554 */
  best = is_rc ? &rc : &fw;
  memset(&front, 0, sizeof front);
  memset(&back, 0, sizeof back);
  strncpy(front.frag_id, id, ORA_MAX_ID);
  strncpy(front.frag_desc, desc, ORA_MAX_DESC);
  strcpy(front.ref_seq, is_rc ? rg_r : rg_f);
  strcpy(front.frag_seq, is_rc ? fg_r : fg_f);
  front.start = best->abc; front.end = best->aec;
  front.trimmed = trimmed; front.segment = 'a'; front.score = best->best;
  front.num_inputs = 0; /* PWAlnFrag.num_inputs is never set in pass 1; malloc'd memory (0 in practice) */
  score = best->best;
  if (is_rc) {
    revcom_string(front.ref_seq);
    revcom_string(front.frag_seq);
    front.revcom = 1;
    front.start = to_rc_coord(best->aec, L);
    front.end = to_rc_coord(best->abc, L);
    as = front.start; ae = front.end;
  } else {
    front.revcom = 0;
    as = best->abc; ae = best->aec;
  }
  if (as > ae) ae = L + as;               /* :1600-1604 (sic) */
  if (front.end > L) front.end -= L;      /* :1606-1610 */
  if (score < ORA_FIRST_ROUND_CUTOFF) return; /* :1614 (distant_ref is out of scope) */
  if (front.start > front.end) {
    split_record(&front, &back, L);
    merge_record(st, &front); front_slot = st->num_aln_seqs - 1;
    merge_record(st, &back); back_slot = st->num_aln_seqs - 1;
  } else {
    merge_record(st, &front); front_slot = st->num_aln_seqs - 1;
  }
  fs = fsdb_push(st);
  strncpy(fs->id, id, ORA_MAX_ID);
  strncpy(fs->desc, desc, ORA_MAX_DESC);
  strcpy(fs->seq, seq);
  fs->seq_len = n;
  fs->trimmed = trimmed; fs->trim_point = trim_point;
  fs->rc = is_rc; fs->as = as; fs->ae = ae; fs->score = score;
  fs->front = front_slot; fs->back = back_slot;
  fs->unique_best = 1; fs->num_inputs = 1;
  fs->strand_known = score > ORA_FIRST_ROUND_CUTOFF; /* :1653 */
  if (fs->rc && fs->strand_known) revcom_string(fs->seq);
}

/* read_fasta / read_fastq, src/io.c:35-281.  Returns 1 if a record was read. */
static int next_record(FILE *f, int fastq, char *id, char *desc, char *seq) {
  int c, i;
  c = fgetc(f);
  if (c == EOF) return 0;
  if (c != (fastq ? '@' : '>')) return 0;
  i = 0;
  while (!isspace(c = fgetc(f)) && i < ORA_MAX_ID) {
    if (c == EOF) return 0;
    id[i++] = (char)c;
  }
  id[i] = 0;
  if (c == '\n') desc[0] = 0;
  else {
    while (c != '\n' && isspace(c)) c = fgetc(f);
    i = 0;
    if (!fastq) {
      /* read_fasta pushes the first description character back and then also
       * stores it: the character is doubled (src/io.c:228-234) */
      if (c != '\n' && i < ORA_MAX_DESC) desc[i++] = (char)c;
    }
    while (c != '\n' && c != EOF && i < ORA_MAX_DESC) { desc[i++] = (char)c; c = fgetc(f); }
    desc[i] = 0;
    if (!fastq && c != '\n' && c != EOF) { /* description truncated: the rest of the line is read as sequence */ }
  }
  i = 0;
  c = fgetc(f);
  if (!fastq) {
    while (c != '>' && c != EOF && i < ORA_MAX_READ) {
      if (!isspace(c)) seq[i++] = (char)toupper(c);
      c = fgetc(f);
    }
    seq[i] = 0;
    if (c == '>') { ungetc('>', f); return 1; }
    if (i == ORA_MAX_READ) {
      while (c != '>' && c != EOF) c = fgetc(f);
      if (c == '>') ungetc('>', f);
    }
    return 1;
  }
  while (c != '\n' && c != EOF && i < ORA_MAX_READ) {
    if (!isspace(c)) seq[i++] = (char)toupper(c);
    c = fgetc(f);
  }
  seq[i] = 0;
  if (i == ORA_MAX_READ) while (c != '\n' && c != EOF) c = fgetc(f);
  c = fgetc(f);
  if (c != '+') return 1; /* "Problem reading quality line": record is still used */
  c = fgetc(f);
  while (c != '\n' && c != EOF) c = fgetc(f);
  {
    int q = 0;
    c = fgetc(f);
    while (c != '\n' && c != EOF && q < ORA_MAX_READ) { if (!isspace(c)) q++; c = fgetc(f); }
    if (q == ORA_MAX_READ) while (c != '\n' && c != EOF) c = fgetc(f);
    if (q != i) return 0; /* unequal lengths ends the input loop (src/io.c:166-170) */
  }
  return 1;
}

int ora_pass1_file(ora_state *st, const char *path) {
  FILE *f = fopen(path, "r");
  char id[ORA_MAX_ID + 4], desc[ORA_MAX_DESC + 4], seq[ORA_MAX_READ + 4];
  int fastq, c, seen = 0;
  if (!f) return -1;
  c = fgetc(f);
  if (c != EOF) ungetc(c, f);
  fastq = (c == '@'); /* find_input_type, src/io.c:11-26 */
  while (next_record(f, fastq, id, desc, seq)) {
    seen++;
    ora_pass1_read(st, id, desc, seq);
  }
  fclose(f);
  return seen;
}

/* ------------------------------------------------------------------ */
/* pop_smp, cull, sort                                                  */
/* ------------------------------------------------------------------ */

/* asp_len, src/fsdb.c:518-530 */
static int slot_total_len(const ora_alnseq *a) {
  int span = a->end - a->start + 1, tot = span, i;
  for (i = 0; i < span && i <= ORA_MAX_ALN; i++)
    if (a->ins[i]) tot += (int)strlen(a->ins[i]);
  return tot;
}

/* pop_smp_from_FSDB, src/fsdb.c:542-619 */
static void pop_smp(ora_state *st) {
  const int depth = ORA_PSSM_DEPTH;
  int i;
  for (i = 0; i < st->num_fss; i++) {
    ora_frag *fs = &st->fss[i];
    ora_alnseq *fa = st->slots[fs->front], *ba = fs->back >= 0 ? st->slots[fs->back] : NULL;
    int flen = slot_total_len(fa), blen = ba ? slot_total_len(ba) : 0;
    int act = 0, p, span = fa->end - fa->start + 1;
    for (p = 0; p < span; p++) {
      int dff, dfb;
      if (fa->ins[p]) act += (int)strlen(fa->ins[p]);
      dff = act;
      dfb = (flen + blen) - act - 1;
      if (dff <= depth) fa->smp[p] = (char)('A' + dff);
      else if (dfb < depth) fa->smp[p] = (char)('A' + depth * 2 - dfb);
      else fa->smp[p] = (char)('A' + depth);
      if (fa->seq[p] != '-') act++;
    }
    if (span < 0) span = 0;
    fa->smp[span] = 0;
    if (ba) {
      span = ba->end - ba->start + 1;
      for (p = 0; p < span; p++) {
        int dff, dfb;
        if (ba->ins[p]) act += (int)strlen(ba->ins[p]);
        dff = flen + act; /* sic: act already includes the front bases */
        dfb = (flen + blen) - act - 1;
        if (dff <= depth) ba->smp[p] = (char)('A' + dff);
        else if (dfb < depth) ba->smp[p] = (char)('A' + depth * 2 - dfb);
        else ba->smp[p] = (char)('A' + depth);
        if (ba->seq[p] != '-') act++;
      }
      if (span < 0) span = 0;
      ba->smp[span] = 0;
    }
  }
}

/* find_fsdb_score_cut, src/fsdb.c:269-383 (IEEE double, evaluated in fsdb order) */
static void score_cut_line(const ora_state *st, double *slope, double *intercept) {
  double xbar = 0, ybar = 0, ssxy = 0, ssxx = 0, slope_bf, intercept_bf, max_delta = 0;
  size_t j = 0;
  int i;
  for (i = 0; i < st->num_fss; i++) {
    const ora_frag *fs = &st->fss[i];
    if (fs->unique_best && fs->score >= ORA_FIRST_ROUND_CUTOFF) { xbar += fs->seq_len; ybar += fs->score; j++; }
  }
  xbar /= j;
  ybar /= j;
  for (i = 0; i < st->num_fss; i++) {
    const ora_frag *fs = &st->fss[i];
    if (fs->unique_best && fs->score >= ORA_FIRST_ROUND_CUTOFF) {
      ssxy += (fs->seq_len - xbar) * (fs->score - ybar);
      ssxx += (fs->seq_len - xbar) * (fs->seq_len - xbar);
    }
  }
  slope_bf = ssxy / ssxx;
  intercept_bf = ybar - slope_bf * xbar;
  for (i = 0; i < st->num_fss; i++) {
    const ora_frag *fs = &st->fss[i];
    if (fs->unique_best && fs->score >= ORA_FIRST_ROUND_CUTOFF) {
      double delta = (fs->score - ((slope_bf * fs->seq_len) + intercept_bf)) / fs->seq_len;
      if (delta > max_delta) max_delta = delta;
    }
  }
  *intercept = intercept_bf;
  if ((slope_bf - max_delta) > 0) *slope = slope_bf - (max_delta * 2.0);
  else *slope = (double)(slope_bf * (80 / 100.0));
}

/* cull_maln_from_fsdb, src/mia.c:418-506 */
static void cull(ora_state *st) {
  double slope, intercept;
  int i, j, n = 0;
  if (st->opt.score_cut_set) { slope = st->opt.slope; intercept = st->opt.intercept; }
  else score_cut_line(st, &slope, &intercept);
  if (slope <= 0) slope = 100.0;
  if (st->culled_cap < 2 * st->num_fss + 16) {
    st->culled_cap = 2 * st->num_fss + 16;
    st->culled = (int *)realloc(st->culled, sizeof(int) * (size_t)st->culled_cap);
  }
  for (i = 0; i < st->num_fss; i++) {
    ora_frag *fs = &st->fss[i];
    double min_score = st->opt.hard_cut > 0 ? (double)st->opt.hard_cut : (double)(intercept + (slope * fs->seq_len));
    if (!fs->unique_best) continue;
    st->culled[n++] = fs->front;
    if (fs->score < min_score) st->slots[fs->front]->dropped = 1;
    if (fs->back >= 0) {
      st->culled[n++] = fs->back;
      if (fs->score < min_score) st->slots[fs->back]->dropped = 1;
    }
  }
  st->culled_n = n;
  for (i = 0; i < st->seq_len; i++) {
    int g = 0;
    if (st->gaps[i] <= 0) continue;
    for (j = 0; j < n; j++) {
      const ora_alnseq *a = st->slots[st->culled[j]];
      if (a->start < i && a->end >= i) {
        const char *ins = (i - a->start <= ORA_MAX_ALN) ? a->ins[i - a->start] : NULL;
        if (ins && (int)strlen(ins) > g) g = (int)strlen(ins);
      }
    }
    st->gaps[i] = g;
  }
}

/* sort_aln_frags + alnSeqCmp, src/map_align.c:393-414, src/map_alignment.c:630-633.
 * glibc qsort is a merge sort here => stable; a stable merge sort of our own
 * keeps the oracle independent of libc. */
static void sort_culled(ora_state *st) {
  int n = st->culled_n, w, i;
  int *a = st->culled, *b = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)), *t;
  for (w = 1; w < n; w *= 2) {
    for (i = 0; i < n; i += 2 * w) {
      int l = i, m = i + w < n ? i + w : n, r = i + 2 * w < n ? i + 2 * w : n, p = l, q = m, o = l;
      while (p < m && q < r) {
        const ora_alnseq *x = st->slots[a[p]], *y = st->slots[a[q]];
        int less_eq = (x->start < y->start) || (x->start == y->start && x->end <= y->end);
        b[o++] = less_eq ? a[p++] : a[q++];
      }
      while (p < m) b[o++] = a[p++];
      while (q < r) b[o++] = a[q++];
    }
    t = a; a = b; b = t;
  }
  if (a != st->culled) { memcpy(st->culled, a, sizeof(int) * (size_t)n); free(a); }
  else free(b);
}

/* src/mia_main.c:812-875 */
void ora_finish_pass1(ora_state *st) {
  int i, j;
  pop_smp(st);
  st->culled_size_reported = st->num_aln_seqs; /* init_culled_map_alignment, src/mia.c:54 */
  cull(st);
  sort_culled(st);
  memset(st->fw_mask, 1, (size_t)st->fw_len1);
  for (i = 0, j = 0; i < st->num_fss; i++) /* clean_FSDB, src/mia.c:400-406 */
    if (st->fss[i].score > 0) st->fss[j++] = st->fss[i];
  st->num_fss = j;
}

/* ------------------------------------------------------------------ */
/* the per-iteration driver                                             */
/* ------------------------------------------------------------------ */

/* one read of reiterate_assembly's loop up to the alignment itself, src/mia_main.c:179-252 */
typedef struct { pwaln front; ora_aln res; int ref_start; } realigned;
typedef struct { ora_state *st; realigned *out; int lo, hi; } realign_job;

static void *realign_range(void *arg) {
  realign_job *job = (realign_job *)arg;
  ora_state *st = job->st;
  int i;
  for (i = job->lo; i < job->hi; i++) {
    const ora_frag *fs = &st->fss[i];
    realigned *r = &job->out[i];
    const ora_pssm *pm;
    int len2, ref_start, ref_end;
    if (!fs->strand_known) continue; /* :178 */
    pm = fs->rc ? &st->rcanc : &st->anc;
    len2 = (int)strlen(fs->seq);
    ref_start = (fs->as - ORA_REALIGN_BUFFER) < 0 ? 0 : fs->as - ORA_REALIGN_BUFFER;       /* :191-196 */
    ref_end = (fs->ae + ORA_REALIGN_BUFFER + 1) > st->wrap_seq_len ? st->wrap_seq_len
                                                                    : fs->ae + ORA_REALIGN_BUFFER; /* :197-203 */
    if (ref_start + len2 > ref_end) { ref_start = 0; ref_end = st->wrap_seq_len; }          /* :209-212 */
    r->ref_start = ref_start;
    ora_align(st->seq + ref_start, ref_end - ref_start, fs->seq, len2, NULL, pm, 1, &r->res,
              r->front.ref_seq, r->front.frag_seq, NULL, NULL);
  }
  return NULL;
}

static void realign_run(realign_job *all) {
  ora_state *st = all->st;
  int T = st->threads > 1 ? st->threads : 1, t;
  if (T > 64) T = 64;
  if (T == 1 || st->num_fss < 2 * T) { all->lo = 0; all->hi = st->num_fss; realign_range(all); return; }
  {
    pthread_t th[64];
    realign_job jobs[64];
    int started = 0;
    for (t = 0; t < T; t++) {
      jobs[t] = *all;
      jobs[t].lo = (int)((long long)st->num_fss * t / T);
      jobs[t].hi = (int)((long long)st->num_fss * (t + 1) / T);
      if (pthread_create(&th[t], NULL, realign_range, &jobs[t]) != 0) break;
      started++;
    }
    for (t = started; t < T; t++) realign_range(&jobs[t]); /* (a thread that could not be started: do its share here) */
    for (t = 0; t < started; t++) pthread_join(th[t], NULL);
  }
}

/* reiterate_assembly, src/mia_main.c:24-280 (distant_ref and hp are out of scope) */
static void realign_all(ora_state *st, const char *new_ref, int iter_num) {
  int i, j, L = (int)strlen(new_ref);
  free(st->seq); free(st->rcseq); free(st->gaps);
  st->seq = (char *)calloc((size_t)(L + 1) * 2 + 1024, 1);
  memcpy(st->seq, new_ref, (size_t)L);
  st->rcseq = NULL;
  if (iter_num > 1) { sprintf(st->ref_id, "ConsAssem.%d", iter_num); strcpy(st->ref_desc, "iteration assembly"); }
  st->seq_len = L;
  st->size = L + 1;
  if (st->circular) add_wrap(st); else st->wrap_seq_len = L;
  st->gaps = (int *)calloc((size_t)st->wrap_seq_len + 1, sizeof(int));
  for (i = 0; i < st->num_aln_seqs; i++) { /* :81-92 */
    ora_alnseq *a = st->slots[i];
    int n = (int)strlen(a->seq);
    for (j = 0; j < n; j++) if (a->ins[j]) { free(a->ins[j]); a->ins[j] = NULL; }
  }
  st->num_aln_seqs = 0;
  /* The alignment of a read depends on the new reference and on the read's own previous coordinates only, so the
   * alignments may be computed ahead of the merge loop -- on several host threads when ora_set_threads asked for them --
   * as long as the records are merged in fsdb order, one read after the other, exactly as the reference's single loop
   * does (:176-278). */
  {
    realign_job job;
    job.st = st;
    job.out = (realigned *)calloc((size_t)(st->num_fss ? st->num_fss : 1), sizeof(realigned));
    realign_run(&job);
    for (i = 0; i < st->num_fss; i++) {
      ora_frag *fs = &st->fss[i];
      realigned *r = &job.out[i];
      pwaln *front = &r->front, back;
      if (!fs->strand_known) continue; /* :178 */
      memset(&back, 0, sizeof back);
      strcpy(front->frag_id, fs->id);
      strcpy(front->frag_desc, fs->desc);
      front->trimmed = fs->trimmed; front->revcom = fs->rc; front->num_inputs = fs->num_inputs;
      front->segment = 'a'; front->score = r->res.best;
      front->start = r->res.abc + r->ref_start;
      front->end = r->res.aec + r->ref_start;
      fs->as = front->start; fs->ae = front->end; fs->unique_best = 1; fs->score = r->res.best;
      if (front->end > st->seq_len) front->end -= st->seq_len; /* :259-263 */
      if (front->start > front->end) {
        split_record(front, &back, st->seq_len);
        merge_record(st, front); fs->front = st->num_aln_seqs - 1;
        merge_record(st, &back); fs->back = st->num_aln_seqs - 1;
      } else {
        merge_record(st, front); fs->front = st->num_aln_seqs - 1;
        /* fs->back is left untouched (stale index if the read was split before), :273-276 */
      }
    }
    free(job.out);
  }
}

void ora_iterate(ora_state *st, const char *new_ref, int iter_num) {
  realign_all(st, new_ref, iter_num);
  pop_smp(st);
  cull(st);
  sort_culled(st);
}

/* The reference scans ALL records for every column (src/mia.c:576-595, src/map_align.c:463-495: O(L*N)).  The records
 * that can cover column `pos` are those with start <= pos <= end; when the list is sorted by start (sort_aln_frags ran:
 * every caller here) they lie in the index range [lo, hi) with hi = first record starting beyond pos and lo = first
 * record starting at or after pos - (longest record span).  Scanning that range in list order adds the same bases in
 * the same order as the full scan -- the records left out fail the reference's own test.  An unsorted list is scanned
 * in full. */
typedef struct { int sorted, maxspan, lo, hi; } col_window;

static void window_init(const ora_state *st, col_window *w) {
  int j;
  w->sorted = 1; w->maxspan = 0; w->lo = 0; w->hi = 0;
  for (j = 0; j < st->culled_n; j++) {
    const ora_alnseq *a = st->slots[st->culled[j]];
    if (j > 0 && st->slots[st->culled[j - 1]]->start > a->start) w->sorted = 0;
    if (a->end - a->start > w->maxspan) w->maxspan = a->end - a->start;
  }
}

/* columns are visited in ascending order: both ends of the range only move forward */
static void window_at(const ora_state *st, col_window *w, int pos, int *lo, int *hi) {
  if (!w->sorted) { *lo = 0; *hi = st->culled_n; return; }
  while (w->hi < st->culled_n && st->slots[st->culled[w->hi]]->start <= pos) w->hi++;
  while (w->lo < w->hi && st->slots[st->culled[w->lo]]->start < pos - w->maxspan) w->lo++;
  *lo = w->lo; *hi = w->hi;
}

/* find_ins_cons, src/map_align.c:444-510 (no `dropped` test) */
static void insert_consensus(const ora_state *st, int pos, char *out, int lo, int hi) {
  int n = st->gaps[pos], i, j;
  ora_counts *bc = (ora_counts *)calloc((size_t)n, sizeof(ora_counts));
  for (i = lo; i < hi; i++) {
    const ora_alnseq *a = st->slots[st->culled[i]];
    const ora_pssm *pm;
    const char *ins;
    int ilen;
    if (!(a->start < pos && a->end >= pos)) continue;
    pm = a->revcom ? &st->rcanc : &st->anc;
    ins = a->ins[pos - a->start];
    ilen = ins ? (int)strlen(ins) : 0;
    for (j = 0; j < n; j++) ora_add_base(j < ilen ? ins[j] : '-', &bc[j], pm, a->smp[pos - a->start]);
  }
  for (j = 0; j < n; j++) out[j] = ora_find_consensus(&bc[j], st->opt.cons_code);
  free(bc);
}

/* consensus_assembly_string, src/mia.c:515-603 */
char *ora_consensus(ora_state *st) {
  int num_gaps = 0, j, pos, o = 0;
  char *cons, ins_cons[ORA_MAX_ALN + 2];
  for (j = 0; j < st->seq_len; j++) num_gaps += st->gaps[j];
  col_window w;
  cons = (char *)malloc((size_t)st->seq_len + num_gaps + 1);
  window_init(st, &w);
  for (pos = 0; pos < st->seq_len; pos++) {
    ora_counts bc;
    char b;
    int lo, hi;
    window_at(st, &w, pos, &lo, &hi);
    if (st->gaps[pos] > 0 && pos > 0) {
      insert_consensus(st, pos, ins_cons, lo, hi);
      for (j = 0; j < st->gaps[pos]; j++)
        if (ins_cons[j] != '-' && ins_cons[j] != ' ') cons[o++] = ins_cons[j];
    }
    memset(&bc, 0, sizeof bc);
    for (j = lo; j < hi; j++) {
      const ora_alnseq *a = st->slots[st->culled[j]];
      if (a->start <= pos && a->end >= pos && !a->dropped)
        ora_add_base(a->seq[pos - a->start], &bc, a->revcom ? &st->rcanc : &st->anc, a->smp[pos - a->start]);
    }
    b = ora_find_consensus(&bc, st->opt.cons_code);
    if (b != '-' && b != ' ') cons[o++] = b;
  }
  cons[o] = 0;
  return cons;
}

void ora_column_tallies(ora_state *st, int *out) {
  int pos, j;
  col_window w;
  window_init(st, &w);
  for (pos = 0; pos < st->seq_len; pos++) {
    ora_counts bc;
    int lo, hi;
    window_at(st, &w, pos, &lo, &hi);
    memset(&bc, 0, sizeof bc);
    for (j = lo; j < hi; j++) {
      const ora_alnseq *a = st->slots[st->culled[j]];
      if (a->start <= pos && a->end >= pos && !a->dropped)
        ora_add_base(a->seq[pos - a->start], &bc, a->revcom ? &st->rcanc : &st->anc, a->smp[pos - a->start]);
    }
    out[pos * 10 + 0] = bc.As; out[pos * 10 + 1] = bc.Cs; out[pos * 10 + 2] = bc.Gs; out[pos * 10 + 3] = bc.Ts;
    out[pos * 10 + 4] = bc.gaps; out[pos * 10 + 5] = bc.cov;
    out[pos * 10 + 6] = bc.scoreA; out[pos * 10 + 7] = bc.scoreC; out[pos * 10 + 8] = bc.scoreG; out[pos * 10 + 9] = bc.scoreT;
  }
}

/* write_ma, src/map_alignment.c:283-382 */
int ora_write_maln(ora_state *st, const char *path) {
  FILE *f = fopen(path, "w");
  time_t t = time(NULL);
  int i, j, d, row;
  if (!f) return 0;
  fprintf(f, "/* map_alignment [V%s] */ %s", "1.0", asctime(localtime(&t)));
  fprintf(f, "MALN_NAS %d\nMALN_SIZ %d\nMALN_COC %d\n", st->culled_n, st->culled_size_reported, st->opt.cons_code);
  fprintf(f, "__REFERENCE__\nID %s\nDESC %s\nLEN %d\nSIZE %d\nSEQ ", st->ref_id, st->ref_desc, st->seq_len, st->size);
  for (i = 0; i < st->seq_len; i++) fputc(st->seq[i], f);
  fprintf(f, "\nGAPS");
  for (i = 0; i < st->seq_len; i++) fprintf(f, " %d", st->gaps[i]);
  fprintf(f, "\n__PSSM__\nDEPTH %d\nFPSM:\n", ORA_PSSM_DEPTH);
  for (d = 0; d <= 2 * ORA_PSSM_DEPTH; d++) {
    for (row = 0; row < 5; row++)
      fprintf(f, "%d %d %d %d %d\n", st->anc.sm[d][row][0], st->anc.sm[d][row][1], st->anc.sm[d][row][2],
              st->anc.sm[d][row][3], st->anc.sm[d][row][4]);
    fprintf(f, "\n");
  }
  fprintf(f, "RPSM:\n");
  for (d = 0; d <= 2 * ORA_PSSM_DEPTH; d++) {
    for (row = 0; row < 5; row++)
      fprintf(f, "%d %d %d %d %d\n", st->rcanc.sm[d][row][0], st->rcanc.sm[d][row][1], st->rcanc.sm[d][row][2],
              st->rcanc.sm[d][row][3], st->rcanc.sm[d][row][4]);
    fprintf(f, "\n");
  }
  fprintf(f, "__ALNSEQS__\n");
  for (i = 0; i < st->culled_n; i++) {
    const ora_alnseq *a = st->slots[st->culled[i]];
    int n = (int)strlen(a->seq);
    fprintf(f, "ID %s\nDESC %s\nSCORE %d\nNUM_INPUTS %d\nSTART %d\nEND %d\nRC %d\nTR %d\nDR %d\nSEG %c\nSEQ %s\nSMP %s\nINS_POS",
            a->id, a->desc, a->score, a->num_inputs, a->start, a->end, !!a->revcom, !!a->trimmed, !!a->dropped,
            a->segment, a->seq, a->smp);
    for (j = 0; j < n; j++) if (a->ins[j]) fprintf(f, " %d %s", j, a->ins[j]);
    fprintf(f, "\n");
  }
  fclose(f);
  return 1;
}

/* main(), src/mia_main.c:742-976 */
int ora_run(ora_state *st, const char *frag_path, const char *maln_root) {
  char fn[2048], *last, *cons;
  int iter = 1;
  if (!st->ref_prepared) ora_prepare_ref(st);
  if (ora_pass1_file(st, frag_path) < 0) return -1;
  ora_finish_pass1(st);
  last = (char *)malloc((size_t)st->seq_len + 1);
  memcpy(last, st->seq, (size_t)st->seq_len);
  last[st->seq_len] = 0;
  ora_iterate(st, last, iter);
  snprintf(fn, sizeof fn, "%s.%d", maln_root, iter);
  if (!st->opt.iterate || !st->opt.final_only) ora_write_maln(st, fn);
  if (st->opt.iterate) {
    cons = ora_consensus(st);
    while (strcmp(cons, last) != 0 && iter < ORA_MAX_ITER) {
      iter++;
      free(last);
      last = cons;
      ora_iterate(st, cons, iter);
      snprintf(fn, sizeof fn, "%s.%d", maln_root, iter);
      if (!st->opt.final_only) ora_write_maln(st, fn);
      cons = ora_consensus(st);
    }
    snprintf(fn, sizeof fn, "%s.%d", maln_root, iter);
    if (st->opt.final_only) ora_write_maln(st, fn);
    if (cons != last) free(cons);
  }
  free(last);
  return iter;
}

/* accessors */
/* A read with its post-pass-1 fields set directly -- add_fs2fsdb (src/fsdb.c:628-663) without sg_align before it: what a
 * caller holds who already knows strand and coordinates (bench.py's inputs; oracle/ref_iter_driver.c fills the
 * reference's FSDB the same way).  front/back slots are empty until the first ora_iterate merges records. */
void ora_push_frag(ora_state *st, const char *id, const char *seq, int rc, int as, int ae, int score, int strand_known) {
  ora_frag *fs = fsdb_push(st);
  int n = 0;
  strncpy(fs->id, id, ORA_MAX_ID);
  while (seq[n] && n < ORA_MAX_READ) { fs->seq[n] = seq[n]; n++; }
  fs->seq[n] = 0;
  fs->seq_len = n;
  fs->rc = rc; fs->as = as; fs->ae = ae; fs->score = score; fs->strand_known = strand_known;
  fs->unique_best = 1; fs->num_inputs = 1; fs->front = -1; fs->back = -1;
}

void ora_set_threads(ora_state *st, int n) { st->threads = n; }

int ora_num_frags(const ora_state *st) { return st->num_fss; }
const ora_frag *ora_frag_at(const ora_state *st, int i) { return &st->fss[i]; }
int ora_num_culled(const ora_state *st) { return st->culled_n; }
const ora_alnseq *ora_culled_at(const ora_state *st, int i) { return st->slots[st->culled[i]]; }
const ora_alnseq *ora_slot_at(const ora_state *st, int i) { return st->slots[i]; }
int ora_ref_len(const ora_state *st) { return st->seq_len; }
const char *ora_ref_seq(const ora_state *st) { return st->seq; }
const int *ora_ref_gaps(const ora_state *st) { return st->gaps; }
