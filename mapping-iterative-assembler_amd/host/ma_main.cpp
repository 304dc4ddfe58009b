// ma_hip -- the reference's `ma` report tool (/root/reference/src/map_assembler.c) for the formats that
// are computed from the column tallies: -f 5 (assembled sequence as FASTA), -f 41 and -f 4 (per-column
// table).  The .maln text is parsed here exactly as read_ma does (src/map_alignment.c:384-607); the
// add_base loops of show_consensus / find_ins_cons run on the GPU (mia_hip_ma_tally, every record counts,
// dropped or not); calling, phred score and printing follow src/map_alignment.c:107-220,
// src/map_align.c:152-227,294-391 and src/io.c:929-951.  No CPU fallback.
#include <ctype.h>
#include <float.h>
#include <getopt.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mia_hip.h"
#include "maln_text.h"

namespace {

constexpr int PSSM_DEPTH = 15, FASTA_LINE_WIDTH = 60;   // src/params.h:20-21
using namespace maln_text;

void help() {
  printf("ma_hip -M <maln input file>\n   -c <consensus code>\n   -f <output format: 5, 41 or 4>\n   -I <ID to assign to assembly sequence>\n"
         "ma_hip reports the assembled sequence (-f 5) or the per-column table (-f 41 all positions, -f 4 positions that\n"
         "differ from the reference) of a .maln file written by mia, as the reference's ma does; the tallies are\n"
         "computed on the MI355X.  The other ma formats are outside the accelerated path.\n");
}

struct Maln {
  std::string ref_id, ref_seq;
  int L = 0;
  std::vector<int32_t> gaps;
  int32_t fpsm[31][5][5], rpsm[31][5][5];
  std::vector<int32_t> start, ins_record, ins_pos;
  std::vector<uint8_t> revcom;
  std::vector<int64_t> col_off, ins_off;
  std::string seq, smp, ins_bases;
};

void bad(const char* what, const char* fn) { fprintf(stderr, what, fn); exit(1); }

void read_ma(const char* fn, Maln* m) {
  std::string buf;
  if (!slurp(fn, &buf)) { fprintf(stderr, "Cannot open %s\n", fn); exit(1); }
  Cursor c{buf.data(), buf.data() + buf.size()};
  std::string line, tok;
  c.line(&line);
  if (line.find("/* map_alignment") == std::string::npos) bad("%s does not look like a map_alignment input file\n", fn);
  int nas = 0, tmp = 0;
  c.line(&line); field_int(line, "MALN_NAS", &nas);
  c.line(&line);                       // MALN_SIZ: only sizes an array
  c.line(&line);                       // MALN_COC: overridden by -c (src/map_assembler.c:191)
  c.line(&line);
  if (line.find("__REFERENCE__") == std::string::npos) bad("Do not see reference sequence header in %s\n", fn);
  c.line(&line); field(line, "ID", &m->ref_id);
  c.line(&line);                       // DESC
  c.line(&line); field_int(line, "LEN", &m->L);
  c.line(&line);                       // SIZE
  c.line(&line); field(line, "SEQ", &m->ref_seq);
  if ((int)m->ref_seq.size() != m->L) {
    fprintf(stderr, "Reported length of reference sequence %d is not observed length %d\n", m->L, (int)m->ref_seq.size());
    exit(1);
  }
  c.literal("GAPS");
  m->gaps.assign((size_t)m->L, 0);
  for (int i = 0; i < m->L; i++) { long v = 0; if (c.integer(&v)) m->gaps[(size_t)i] = (int32_t)v; }
  while (c.p < c.end && *c.p != '\n') c.p++;
  if (c.p < c.end) c.p++;
  c.line(&line);
  if (line.find("__PSSM__") == std::string::npos) { fprintf(stderr, "Do not see __PSSM__ line in %s\n", fn); exit(2); }
  int depth = PSSM_DEPTH;
  c.line(&line); field_int(line, "DEPTH", &depth);
  c.line(&line);
  if (line.find("FPSM:") == std::string::npos) { fprintf(stderr, "Do not see the FPSM: in %s\n", fn); exit(2); }
  memset(m->fpsm, 0, sizeof m->fpsm);
  memset(m->rpsm, 0, sizeof m->rpsm);
  read_matrices(c, depth, m->fpsm);
  c.line(&line);
  if (line.find("RPSM:") == std::string::npos) { fprintf(stderr, "Do not see the RPSM: in %s\n", fn); exit(2); }
  read_matrices(c, depth, m->rpsm);
  c.line(&line);
  if (line.find("__ALNSEQS__") == std::string::npos) bad("Do not see __ALNSEQS__ line in %s\n", fn);
  m->col_off.push_back(0);
  m->ins_off.push_back(0);
  for (int r = 0; r < nas; r++) {
    int start = 0, end = 0, rc = 0;
    std::string seq, smp;
    c.line(&line);                                  // ID
    c.line(&line);                                  // DESC
    c.line(&line);                                  // SCORE
    c.line(&line);                                  // NUM_INPUTS, if there
    if (field_int(line, "NUM_INPUTS", &tmp)) c.line(&line);
    field_int(line, "START", &start);
    c.line(&line); field_int(line, "END", &end);
    c.line(&line); field_int(line, "RC", &rc);
    c.line(&line);                                  // TR
    c.line(&line);                                  // DR, if there
    if (field_int(line, "DR", &tmp)) c.line(&line);
    // SEG
    c.line(&line); field(line, "SEQ", &seq);
    c.line(&line); field(line, "SMP", &smp);
    const int ncols = end - start + 1;
    if (ncols < 0 || (int)seq.size() < ncols || (int)smp.size() < ncols || start < 0) {
      fprintf(stderr, "record %d of %s: SEQ/SMP shorter than START..END\n", r, fn);
      exit(1);
    }
    m->start.push_back(start);
    m->revcom.push_back(rc ? 1 : 0);
    m->seq.append(seq, 0, (size_t)ncols);
    m->smp.append(smp, 0, (size_t)ncols);
    m->col_off.push_back((int64_t)m->seq.size());
    c.literal("INS_POS");
    for (;;) {
      const char* save = c.p;
      long pos = 0;
      if (!c.integer(&pos)) break;                   // (white space already consumed, as fscanf does)
      if (!c.token(&tok)) { c.p = save; break; }
      m->ins_record.push_back(r);
      m->ins_pos.push_back((int32_t)pos);
      m->ins_bases += tok;
      m->ins_off.push_back((int64_t)m->ins_bases.size());
    }
  }
}

// find_phred_qscore, src/map_align.c:152-205
int phred(int sA, int sC, int sG, int sT) {
  int best, nb[3];
  if (sA >= sC && sA >= sG && sA >= sT) { best = sA; nb[0] = sC; nb[1] = sG; nb[2] = sT; }
  else if (sC >= sG && sC >= sT) { best = sC; nb[0] = sA; nb[1] = sG; nb[2] = sT; }
  else if (sG >= sT) { best = sG; nb[0] = sA; nb[1] = sC; nb[2] = sT; }
  else { best = sT; nb[0] = sA; nb[1] = sC; nb[2] = sG; }
  double p_best = pow(2, ((double)best / 100));
  double p_nbs[3];
  for (int i = 0; i < 3; i++) p_nbs[i] = pow(2, ((double)nb[i] / 100));
  double p_correct = p_best / (p_nbs[0] + p_nbs[1] + p_nbs[2]);
  if (p_correct >= DBL_MAX) p_correct = DBL_MAX;
  return 10 * log10(p_correct);
}

struct Counts { int As, Cs, Gs, Ts, gaps, cov, sA, sC, sG, sT; };

// find_consensus, src/map_align.c:294-391 (the call AND frac_agree)
char find_consensus(const Counts& b, int cons_code, double* frac) {
  if (b.cov == 0) { *frac = 0.0; return 'N'; }
  if (((double)b.gaps / (double)b.cov) >= (double)(50 / 100.0)) { *frac = ((double)b.gaps / (double)b.cov); return '-'; }
  int top = b.sA, second = INT_MIN;
  char base = 'A';
  *frac = ((double)b.As / (double)b.cov);
  if (b.sC >= top) { second = top; top = b.sC; base = 'C'; *frac = ((double)b.Cs / (double)b.cov); } else second = b.sC;
  if (b.sG >= top) { second = top; top = b.sG; base = 'G'; *frac = ((double)b.Gs / (double)b.cov); } else if (b.sG >= second) second = b.sG;
  if (b.sT >= top) { second = top; top = b.sT; base = 'T'; *frac = ((double)b.Ts / (double)b.cov); } else if (b.sT >= second) second = b.sT;
  if (cons_code == 2) return (top >= 0 || (top - 2400) > second) ? base : 'N';
  return (top >= -399) ? base : 'N';
}

void show_single_pos(int ref_pos, char ref_base, char cons_base, const Counts& b, double frac) {   // src/map_align.c:208-227
  printf("%d %c %c %d %d %d %d %d %d %d %d %d %d %d %0.3f\n", ref_pos, ref_base, cons_base, b.cov, b.As, b.Cs, b.Gs, b.Ts, b.gaps, b.sA, b.sC,
         b.sG, b.sT, phred(b.sA, b.sC, b.sG, b.sT), frac);
}

void die(mia_hip_ctx* g, const char* what) {
  fprintf(stderr, "%s: %s\n", what, g ? mia_hip_last_error(g) : "no context");
  exit(1);
}

}  // namespace

int main(int argc, char* argv[]) {
  std::string ma_in_fn, assign_id;
  bool id_assigned = false, in_ma = false, any_arg = false;
  int cons_scheme = 1, out_format = 1, gpu = 0;
  double score_int = -1.0, score_slo = -1.0;
  int ich;
  // the reference's option string (src/map_assembler.c:113) plus -g <gpu>
  while ((ich = getopt(argc, argv, "I:c:i:f:R:s:m:M:Cb:s:dg:")) != -1) {
    switch (ich) {
      case 'I': assign_id = optarg; id_assigned = true; break;
      case 'c': cons_scheme = atoi(optarg); any_arg = true; break;
      case 'i': any_arg = true; break;                 // parsed and never used by the reference either
      case 'f': out_format = atoi(optarg); any_arg = true; break;
      case 'R': any_arg = true; break;
      case 's': score_slo = atof(optarg); any_arg = true; break;
      case 'b': score_int = atof(optarg); any_arg = true; break;
      case 'C': break;
      case 'm': fprintf(stderr, "option -m (rewrite the .maln) is outside the MI355X-accelerated path and is not supported by ma_hip\n"); exit(1);
      case 'M': ma_in_fn = optarg; in_ma = true; any_arg = true; break;
      case 'd': any_arg = true; break;
      case 'g': gpu = atoi(optarg); break;
      default: help(); exit(0);
    }
  }
  if (!any_arg || ((score_slo == -1) && (score_int != -1)) || ((score_slo != -1) && (score_int == -1)) || !in_ma) { help(); exit(0); }
  if (out_format != 5 && out_format != 4 && out_format != 41) {
    fprintf(stderr, "output format %d is outside the MI355X-accelerated path (formats 5, 41 and 4 are); use the reference's ma\n", out_format);
    exit(1);
  }
  Maln m;
  read_ma(ma_in_fn.c_str(), &m);
  if (id_assigned) m.ref_id = assign_id.substr(0, 256);

  mia_hip_ctx* g = nullptr;
  if (mia_hip_create(&g, gpu) != MIA_HIP_OK) { fprintf(stderr, "ma_hip: no usable MI355X (gfx950) device %d; there is no CPU fallback\n", gpu); exit(1); }
  if (mia_hip_set_pssm(g, &m.fpsm[0][0][0], &m.rpsm[0][0][0]) != MIA_HIP_OK) die(g, "set_pssm");
  const int64_t n = (int64_t)m.start.size(), n_ins = (int64_t)m.ins_record.size();
  if (mia_hip_ma_tally(g, m.L, m.gaps.data(), n, m.start.data(), m.revcom.data(), m.col_off.data(), m.seq.data(), m.smp.data(), n_ins,
                       m.ins_record.data(), m.ins_pos.data(), m.ins_off.data(), m.ins_bases.data()) != MIA_HIP_OK)
    die(g, "ma_tally");
  const int L = m.L;
  if (out_format == 5) {
    // fasta_print_cons of the called columns (src/io.c:929-951); '-' calls are not printed
    int64_t total_gaps = 0;
    for (int p = 0; p < L; p++) total_gaps += m.gaps[(size_t)p];
    std::string cons((size_t)L + (size_t)total_gaps + 16, '\0');
    int64_t clen = 0;
    if (mia_hip_consensus(g, cons_scheme, &cons[0], (int64_t)cons.size(), &clen) != MIA_HIP_OK) die(g, "consensus");
    printf(">%s\n", m.ref_id.c_str());
    int64_t i = 0;
    for (; i + FASTA_LINE_WIDTH <= clen; i += FASTA_LINE_WIDTH) { fwrite(&cons[(size_t)i], 1, FASTA_LINE_WIDTH, stdout); fputc('\n', stdout); }
    fwrite(&cons[(size_t)i], 1, (size_t)(clen - i), stdout);
    fputc('\n', stdout);
    mia_hip_destroy(g);
    return 0;
  }
  // formats 4 / 41: BaseCounts of every column from the device, calls and the double-valued columns here
  std::vector<int32_t> tally((size_t)MIA_HIP_TALLY_WORDS * (size_t)(L + 1)), dgaps((size_t)L + 1), ins_off((size_t)L + 1);
  {
    std::string scratch((size_t)L * 2 + (1 << 20), '\0');
    int64_t clen = 0;
    if (mia_hip_consensus(g, cons_scheme, &scratch[0], (int64_t)scratch.size(), &clen) != MIA_HIP_OK) die(g, "consensus");
  }
  if (mia_hip_get_tally(g, tally.data(), dgaps.data()) != MIA_HIP_OK) die(g, "get_tally");
  int64_t slots = 0;
  if (mia_hip_get_ins_tally(g, ins_off.data(), nullptr, 0, &slots) != MIA_HIP_OK) die(g, "get_ins_tally");
  std::vector<int32_t> ins_tally((size_t)slots * 9 + 9);
  if (slots > 0 && mia_hip_get_ins_tally(g, nullptr, ins_tally.data(), slots, nullptr) != MIA_HIP_OK) die(g, "get_ins_tally");
  const size_t Lp = (size_t)L + 1;
  auto word = [&](int w, int p) { return tally[(size_t)w * Lp + (size_t)p]; };
  for (int p = 0; p < L; p++) {
    if (m.gaps[(size_t)p] > 0 && p > 0) {           // find_ins_cons (src/map_align.c:444-510)
      const int span = word(10 /* T_SPAN */, p);
      for (int j = 0; j < m.gaps[(size_t)p]; j++) {
        const int32_t* t = &ins_tally[(size_t)(ins_off[(size_t)p] + j) * 9];
        Counts b{t[0], t[1], t[2], t[3], span - t[4], span, t[5], t[6], t[7], t[8]};
        double frac = 0.0;
        const char cb = find_consensus(b, cons_scheme, &frac);
        if (out_format == 41 || cb != '-') show_single_pos(p, '-', cb, b, frac);
      }
    }
    Counts b{word(0, p), word(1, p), word(2, p), word(3, p), word(4, p), word(5, p), word(6, p), word(7, p), word(8, p), word(9, p)};
    double frac = 0.0;
    const char cb = find_consensus(b, cons_scheme, &frac);
    if (out_format == 41 || m.ref_seq[(size_t)p] != cb) show_single_pos(p, m.ref_seq[(size_t)p], cb, b, frac);
  }
  mia_hip_destroy(g);
  return 0;
}
