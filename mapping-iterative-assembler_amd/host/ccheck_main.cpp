// ccheck_hip -- the reference's contamination checker (/root/reference/src/ccheck.cc) with its two heavy steps on the
// MI355X: the global alignment of the contaminant consensus to the assembly (mia_hip_myers_align, src/ccheck.cc:480)
// and the re-alignment of EVERY assembled read to the contaminant stretch under it (src/ccheck.cc:569-604), which
// the reference does one dyn_prog at a time and which runs here as one batch (mia_hip_align_windows: the window of a
// read is a substring of the contaminant, so the device holds one 16 kb string and a (start, length) pair per read).
// What is left on the host is bookkeeping that is linear in the file: the lift-over index, the two walks over each
// read's columns and the report.  Same command line, same stdout, same stderr (-v ... -vvvvvv), same exit codes.
// No CPU fallback: without the GPU library the program stops.
#include <dirent.h>
#include <getopt.h>
#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/mia_hip.h"
#include "maln_text.h"

namespace {
using namespace maln_text;

constexpr int PSSM_DEPTH = 15;   // src/params.h:20

// ---- IUPAC arithmetic (src/myers_align.h:40-69, src/ccheck.cc:104-129,178-183) ----------------------------------
int iupac(char x) {
  switch (x & ~32) {
    case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8;
    case 'S': return 6; case 'W': return 9; case 'R': return 5; case 'Y': return 10; case 'K': return 12; case 'M': return 3;
    case 'B': return 14; case 'D': return 13; case 'H': return 11; case 'V': return 7; case 'N': return 15;
    default: return 0;
  }
}
bool differs(char con, char ass) { return con != '-' && ass != '-' && toupper((unsigned char)con) != toupper((unsigned char)ass); }   // "weakly diagnostic"
bool excludes(char con, char ass) { return con != '-' && ass != '-' && (iupac(con) & iupac(ass)) == 0; }                               // "strongly diagnostic"
bool transversion(char a, char b) {
  const char u = a & ~32, v = b & ~32;
  switch (u) {
    case 'A': return v != 'G';
    case 'C': return v != 'T';
    case 'G': return v != 'A';
    case 'T': case 'U': return v != 'C';
    default: return false;
  }
}
// could a read showing y have come from a molecule carrying x?  (deaminated: G may read as A, C as T)
bool consistent(bool adna, char x, char y) {
  const char xd = x == 'G' ? 'R' : x == 'C' ? 'Y' : x == 'g' ? 'r' : x == 'c' ? 'y' : x;
  return x == '-' || y == '-' || (iupac(adna ? xd : x) & iupac(y)) != 0;
}

enum Strength { WEAK, EFFECTIVE, STRONG };
struct Site { char consensus, assembly, contaminant; Strength strength; };
typedef std::map<int, Site> Sites;   // keyed by assembly coordinate

enum Verdict { UNKNOWN, CLEAN, DIRT, CONFLICT, NONSENSE, N_VERDICTS };
const char* const LABEL[] = {"unclassified", "clean", "polluting", "conflicting", "nonsensical", "LB", "ML", "UB"};

Verdict merge(Verdict a, Verdict b) {
  if (a == b) return a;
  if (a == UNKNOWN) return b;
  if (b == UNKNOWN) return a;
  if (a == NONSENSE || b == NONSENSE) return NONSENSE;
  return CONFLICT;
}
void vote(Verdict& v, int& votes, bool maybe_clean, bool maybe_dirt) {   // src/ccheck.cc:298-306
  if (maybe_clean && !maybe_dirt && v == UNKNOWN) v = CLEAN;
  if (maybe_clean && !maybe_dirt && v == DIRT) v = CONFLICT;
  if (!maybe_clean && maybe_dirt && v == UNKNOWN) v = DIRT;
  if (!maybe_clean && maybe_dirt && v == CLEAN) v = CONFLICT;
  if (!maybe_clean && !maybe_dirt) v = NONSENSE;
  if (maybe_clean != maybe_dirt) votes++;
}

void print_sites(FILE* out, Sites::const_iterator i, Sites::const_iterator e, char tail, bool strong_only = false) {
  for (; i != e; ++i) {
    const Site& s = i->second;
    const char who[] = {'(', s.contaminant, ')', 0};
    if (strong_only) {
      if (s.strength >= STRONG) fprintf(out, "<%d:%c%s,%c>, ", i->first, s.consensus, s.strength == EFFECTIVE ? who : "", s.assembly);
    } else {
      fprintf(out, "<%d%c:%c%s,%c>, ", i->first, "wes"[s.strength], s.consensus, s.strength == EFFECTIVE ? who : "", s.assembly);
    }
  }
  if (tail) putc(tail, out);
}

// Wilson score interval of the polluting share (src/ccheck.cc:329-367)
void print_summary(const int* summary, bool table) {
  const double z = 1.96;
  const double k = summary[DIRT], n = k + summary[CLEAN];
  const double p = k / n, c = p + 0.5 * z * z / n, w = z * sqrt(p * (1 - p) / n + 0.25 * z * z / (n * n)), d = 1 + z * z / n;
  double lb = 100.0 * (c - w) / d, ml = 100.0 * p, ub = 100.0 * (c + w) / d;
  const int nn = summary[DIRT] + summary[CLEAN];
  int width = 0;
  for (int v = UNKNOWN; v != N_VERDICTS; v++) width = std::max(width, (int)strlen(LABEL[v]));
  if (lb < 0) lb = 0;
  if (ub > 100) ub = 100;
  for (int v = UNKNOWN; v != N_VERDICTS; v++) {
    if (table) { printf("%d\t", summary[v]); continue; }
    printf("  %*s fragments: %d", width, LABEL[v], summary[v]);
    if (v == DIRT && nn) printf(" (%.1f .. %.1f .. %.1f%%)", lb, ml, ub);
    putchar('\n');
  }
  if (table) {
    if (nn) printf("%.1f\t%.1f\t%.1f\t", lb, ml, ub);
    else fputs("N/A\tN/A\tN/A\t", stdout);
  } else putchar('\n');
}

// ---- inputs ---------------------------------------------------------------------------------------------------------
struct Contaminant { std::string id, desc, seq; };

// read_fasta_ref, src/io.c:288-386: first record; every non-blank character up to the next '>' is sequence
bool read_fasta_first(const char* fn, Contaminant* c) {
  std::string buf;
  if (!slurp(fn, &buf) || buf.empty() || buf[0] != '>') return false;
  size_t i = 1;
  c->id.clear(); c->desc.clear(); c->seq.clear();
  while (i < buf.size() && !isspace((unsigned char)buf[i])) c->id.push_back(buf[i++]);
  if (i >= buf.size()) return false;
  if (buf[i] != '\n') { i++; while (i < buf.size() && buf[i] != '\n') c->desc.push_back(buf[i++]); }
  if (i < buf.size()) i++;
  for (; i < buf.size() && buf[i] != '>'; i++) if (!isspace((unsigned char)buf[i])) c->seq.push_back(buf[i]);
  return true;
}

bool only_iupac(const std::string& s) {   // sanity_check_sequence, src/ccheck.cc:198-204
  for (char ch : s) if (!strchr("ACGTBDHVMKYRSWUN", toupper((unsigned char)ch))) return false;
  return true;
}

struct Record {
  std::string id, seq;                 // SEQ as written: one character per assembly column, '-' for a deletion
  std::vector<std::string> ins;        // ins[p]: bases inserted after column p (a later INS_POS entry replaces an earlier one)
  char segment = 'a';
  int start = 0, end = 0;
};
struct Assembly {
  std::string seq;
  int32_t fpsm[31][5][5];
  std::vector<Record> recs;
};

void read_assembly(const char* fn, Assembly* m) {   // read_ma, src/map_alignment.c:384-607
  std::string buf, line, tok;
  if (!slurp(fn, &buf)) { fprintf(stderr, "Cannot open %s\n", fn); exit(1); }
  Cursor c{buf.data(), buf.data() + buf.size()};
  c.line(&line);
  if (line.find("/* map_alignment") == std::string::npos) { fprintf(stderr, "%s does not look like a map_alignment input file\n", fn); exit(1); }
  int nas = 0, len = 0, tmp = 0;
  c.line(&line); field_int(line, "MALN_NAS", &nas);
  c.line(&line);
  c.line(&line);
  c.line(&line);
  if (line.find("__REFERENCE__") == std::string::npos) { fprintf(stderr, "Do not see reference sequence header in %s\n", fn); exit(1); }
  c.line(&line);                       // ID
  c.line(&line);                       // DESC
  c.line(&line); field_int(line, "LEN", &len);
  c.line(&line);                       // SIZE
  c.line(&line); field(line, "SEQ", &m->seq);
  if ((int)m->seq.size() != len) {
    fprintf(stderr, "Reported length of reference sequence %d is not observed length %d\n", len, (int)m->seq.size());
    exit(1);
  }
  c.literal("GAPS");
  for (int i = 0; i < len; i++) { long v; if (!c.integer(&v)) break; }
  while (c.p < c.end && *c.p != '\n') c.p++;
  if (c.p < c.end) c.p++;
  c.line(&line);
  if (line.find("__PSSM__") == std::string::npos) { fprintf(stderr, "Do not see __PSSM__ line in %s\n", fn); exit(2); }
  int depth = PSSM_DEPTH;
  c.line(&line); field_int(line, "DEPTH", &depth);
  c.line(&line);
  if (line.find("FPSM:") == std::string::npos) { fprintf(stderr, "Do not see the FPSM: in %s\n", fn); exit(2); }
  int32_t rpsm[31][5][5];
  memset(m->fpsm, 0, sizeof m->fpsm);
  read_matrices(c, depth, m->fpsm);
  c.line(&line);
  if (line.find("RPSM:") == std::string::npos) { fprintf(stderr, "Do not see the RPSM: in %s\n", fn); exit(2); }
  read_matrices(c, depth, rpsm);
  c.line(&line);
  if (line.find("__ALNSEQS__") == std::string::npos) { fprintf(stderr, "Do not see __ALNSEQS__ line in %s\n", fn); exit(1); }
  m->recs.resize((size_t)(nas > 0 ? nas : 0));
  for (Record& r : m->recs) {
    c.line(&line); field(line, "ID", &r.id);
    c.line(&line);                                  // DESC
    c.line(&line);                                  // SCORE
    c.line(&line);
    if (field_int(line, "NUM_INPUTS", &tmp)) c.line(&line);
    field_int(line, "START", &r.start);
    c.line(&line); field_int(line, "END", &r.end);
    c.line(&line);                                  // RC
    c.line(&line);                                  // TR
    c.line(&line);
    if (field_int(line, "DR", &tmp)) c.line(&line);
    if (field(line, "SEG", &tok)) r.segment = tok[0];
    c.line(&line); field(line, "SEQ", &r.seq);
    c.line(&line);                                  // SMP
    r.ins.assign(r.seq.size(), std::string());
    c.literal("INS_POS");
    for (;;) {
      const char* save = c.p;
      long pos = 0;
      if (!c.integer(&pos)) break;
      if (!c.token(&tok)) { c.p = save; break; }
      if (pos >= 0 && pos < (long)r.ins.size()) r.ins[(size_t)pos] = tok;
    }
  }
}

// the highest-numbered sibling "<base><n>" of the file named (src/ccheck.cc:206-236)
std::string newest_iteration(std::string fn) {
  const size_t slash = fn.rfind('/');
  const std::string dir = slash == std::string::npos ? std::string(".") : fn.substr(0, slash);
  std::string base = slash == std::string::npos ? fn : fn.substr(slash + 1);
  while (!base.empty() && isdigit((unsigned char)base.back())) base.pop_back();
  int best = 1;
  DIR* d = opendir(dir.c_str());
  if (!d) return fn;
  while (struct dirent* de = readdir(d)) {
    const char* name = de->d_name;
    if (strlen(name) <= base.size() || base.compare(0, base.size(), name, base.size()) != 0) continue;
    const char* digits = name + base.size();
    const char* q = digits;
    while (*q && isdigit((unsigned char)*q)) q++;
    if (*q) continue;
    const int n = atoi(digits);
    if (n > best) { best = n; fn = slash == std::string::npos ? std::string(name) : dir + "/" + name; }
  }
  closedir(d);
  return fn;
}

void strip_segment_suffix(std::string* id) {   // fixup_name, src/ccheck.cc:240-248
  const size_t n = id->size();
  if (n > 3 && ((*id)[n - 1] == 'b' || (*id)[n - 1] == 'f') && (*id)[n - 2] == '_') id->resize((*id)[n - 3] == ',' ? n - 3 : n - 2);
}

void usage(const char* pname) {
  fputs("Usage: ", stdout);
  fputs(pname, stdout);
  fputs(" [-r <ref.fa>] [-a] [-t] [-s M-N] [-v] <aln.maln> \n\n"
        "Reads a maln file and tries to quantify contained contamination.\n"
        "Options:\n"
        "  -r, --reference FILE     FASTA file with the likely contaminant (default: builtin mt311)\n"
        "  -a, --ancient            Treat DNA as ancient (i.e. likely deaminated)\n"
        "  -t, --transversions      Treat only transversions as diagnostic\n"
        "  -s, --span M-N           Look only at range from M to N\n"
        "  -n, --numpos N           Require N diagnostic sites in a single read (default: 1)\n"
        "  -f, --force              Do not look for a higher numbered .maln\n"
        "  -T, --table              Output as tables (easier for scripts, harder on the eyes)\n"
        "  -v, --verbose            Increase verbosity level (can be repeated)\n"
        "  -h, --help               Print this help message\n\n", stdout);
}

[[noreturn]] void die(mia_hip_ctx* g, const char* what) {
  fprintf(stderr, "ccheck_hip: %s: %s\n", what, g ? mia_hip_last_error(g) : "no context");
  exit(3);
}

// ---- the pairwise alignment of contaminant (row `con`) and assembly (row `ass`), indexed ---------------------------
struct Pairing {
  std::string con, ass;          // the two rows, equally long
  std::vector<int> col_at;       // col_at[p]: first column at which p assembly bases lie to the left (p = 0 .. len)
  std::vector<int> con_before;   // con_before[c]: contaminant bases in columns [0, c)
  int ass_len = 0;
  void index() {
    const int A = (int)con.size();
    con_before.assign((size_t)A + 1, 0);
    col_at.clear();
    col_at.push_back(0);
    for (int c = 0; c < A; c++) {
      con_before[(size_t)c + 1] = con_before[(size_t)c] + (con[(size_t)c] != '-');
      if (ass[(size_t)c] != '-') col_at.push_back(c + 1);
    }
    ass_len = (int)col_at.size() - 1;
  }
  int column(int p) const { return p <= ass_len ? col_at[(size_t)(p < 0 ? 0 : p)] : (int)con.size(); }
  // contaminant bases opposite assembly positions [s, e): lift_over, src/ccheck.cc:166-176, as an index range into the
  // ungapped contaminant
  void lifted(int s, int e, int* from, int* count) const {
    const int c0 = column(s), c1 = e <= s ? c0 : column(e);
    *from = con_before[(size_t)c0];
    *count = c1 > c0 ? con_before[(size_t)c1] - *from : 0;
  }
};

struct Pairwise { int start; std::string ref_row, frag_row; };   // cached_pwaln, src/ccheck.cc:292-296

struct Walker {   // the column walk both passes share (src/ccheck.cc:606-613,632-689 and 749-822)
  const Pairing& pg;
  const std::string& assembly;
  const Record& rec;
  std::string in_ref;
  const std::string& frag_vs_ref;
  int col, ass_pos;
  size_t ri = 0, fi = 0, ai, gi = 0;
  Walker(const Pairing& p, const std::string& a, const Record& r, const std::string& lifted_prefix, const Pairwise& pw)
      : pg(p), assembly(a), rec(r), in_ref(lifted_prefix.substr(0, (size_t)std::max(pw.start, 0)) + pw.ref_row), frag_vs_ref(pw.frag_row) {
    col = pg.column(rec.start);
    ass_pos = std::min(rec.start, pg.ass_len);
    ai = (size_t)rec.start;
  }
  char con() const { return pg.con[(size_t)col]; }
  char ass() const { return pg.ass[(size_t)col]; }
  char ref0() const { return ri < in_ref.size() ? in_ref[ri] : '\0'; }
  char frag_ref() const { return fi < frag_vs_ref.size() ? frag_vs_ref[fi] : '\0'; }
  char in_ass() const { return ai < assembly.size() ? assembly[ai] : '\0'; }
  char frag_ass() const { return gi < rec.seq.size() ? rec.seq[gi] : '\0'; }
  bool more() const {
    return ass_pos != rec.end + 1 && col < (int)pg.con.size() && ri < in_ref.size() && in_ass() && frag_ass() && frag_ref();
  }
  void step() {
    if (con() != '-') do { ri++; fi++; } while (ref0() == '-');
    if (ass() != '-') { ass_pos++; do { ai++; gi++; } while (in_ass() == '-'); }
    col++;
  }
};

void print_rows(const std::string& a, const std::string& b) {   // print_aln, src/ccheck.cc:73-89
  size_t i = 0;
  const size_t n = std::min(a.size(), b.size());
  while (i < n) {
    fwrite(a.data() + i, 1, std::min<size_t>(72, a.size() - i), stderr); putc('\n', stderr);
    fwrite(b.data() + i, 1, std::min<size_t>(72, b.size() - i), stderr); putc('\n', stderr);
    for (size_t p = 0; i < n && p != 72; p++, i++) putc(a[i] == b[i] ? '*' : ' ', stderr);
    putc('\n', stderr);
    putc('\n', stderr);
  }
}

struct Options {
  bool adna = false, clever = true, table = false, really = false;
  int min_sites = 1, verbose = 0, maxd = 0, span_from = 0, span_to = INT_MAX;
};

int check_file(mia_hip_ctx* g, const Options& o, int* maxd, const Contaminant& hum, const std::string& infile) {
  int summary[N_VERDICTS] = {0}, summary2[N_VERDICTS] = {0};
  const int verbose = o.verbose;
  if (o.table) { fputs(infile.c_str(), stdout); putchar('\t'); }
  else { puts(infile.c_str()); putchar('\n'); }
  Assembly m;
  read_assembly(infile.c_str(), &m);
  const bool hum_ok = only_iupac(hum.seq), ass_ok = only_iupac(m.seq);
  if (!ass_ok) fputs("FUBAR'ed maln file: consensus sequence contains gap symbols.\n", stderr);
  if (!hum_ok || !ass_ok) { fputs("Problem might exist between keyboard and chair.  I give up.\n", stderr); return 1; }

  // ---- contaminant against assembly, globally
  if (!*maxd) *maxd = (int)std::max(hum.seq.size(), m.seq.size()) / 10;
  Pairing pg;
  {
    std::vector<char> row_a(hum.seq.size() + (size_t)*maxd + 2), row_b(m.seq.size() + (size_t)*maxd + 2);
    uint32_t d = 0;
    if (mia_hip_myers_align(g, hum.seq.c_str(), 0, m.seq.c_str(), *maxd, &d, row_a.data(), row_b.data()) != MIA_HIP_OK) die(g, "myers_align");
    if (d == 0xFFFFFFFFu) {
      fprintf(stderr, "\n *** Could not align references with up to %d mismatches.\n"
                      " *** This is usually a sign of trouble, but\n"
                      " *** IF AND ONLY IF YOU KNOW WHAT YOU ARE DOING, you can\n"
                      " *** try the -d N option with N > %d.\n\n", *maxd, *maxd);
      return 1;
    }
    if (o.table) printf("%d\t", (int)d);
    else printf("  %d alignment distance between reference and assembly.\n", (int)d);
    pg.con = row_a.data();
    pg.ass = row_b.data();
    pg.index();
  }
  if (verbose >= 6) print_rows(pg.con, pg.ass);

  // ---- every column where the two differ (mk_dp_list, src/ccheck.cc:132-154)
  Sites sites;
  {
    int c = pg.column(o.span_from), index = std::min(o.span_from, pg.ass_len);
    if (o.span_from > pg.ass_len) c = (int)pg.con.size();
    for (; index != o.span_to && c < (int)pg.con.size(); c++) {
      const char x = pg.con[(size_t)c], y = pg.ass[(size_t)c];
      if (differs(x, y)) { Site& s = sites[index]; s.consensus = x; s.assembly = y; s.contaminant = 0; s.strength = excludes(x, y) ? STRONG : WEAK; }
      if (y != '-') index++;
    }
  }
  if (o.table) printf("%u\t", (unsigned)sites.size());
  else printf("  %u total differences between reference and assembly.\n", (unsigned)sites.size());
  int num_strong = 0;
  for (const auto& kv : sites) if (kv.second.strength > WEAK) num_strong++;
  if (o.table) printf("%d\t", (int)sites.size());
  else {
    printf("  %d diagnostic positions", (int)sites.size());
    if (o.span_from != 0 || o.span_to != INT_MAX) printf(" in range [%d,%d)", o.span_from, o.span_to);
    printf(", %d of which are strongly diagnostic.\n", num_strong);
  }
  if (verbose >= 3) {
    print_sites(stderr, sites.begin(), sites.end(), '\n', false);
    print_sites(stderr, sites.begin(), sites.end(), '\n', true);
  }
  if (num_strong < 40 && !o.really) {
    fprintf(stderr, "\n *** Low number (%d) of diagnostic positions found.\n"
                    " *** I will stop now for your own safety.\n"
                    " *** If you are sure you want to shoot yourself\n"
                    " *** in the foot, read the man page to learn\n"
                    " *** how to lift this restriction.\n\n", num_strong);
    return 1;
  }

  // ---- every record against the contaminant stretch under it, as one batch on the device (src/ccheck.cc:549-604)
  const size_t n = m.recs.size();
  std::string mia_ref = hum.seq;                      // ref_for_mia: ACGT upper-cased, everything else N
  for (char& ch : mia_ref) {
    const char u = (char)toupper((unsigned char)ch);
    ch = (u == 'A' || u == 'C' || u == 'G' || u == 'T') ? u : 'N';
  }
  std::vector<std::string> reads(n);
  std::vector<int64_t> roff(n + 1, 0), wstart(n);
  std::vector<int32_t> wlen(n);
  std::string bases;
  for (size_t i = 0; i < n; i++) {
    Record& r = m.recs[i];
    strip_segment_suffix(&r.id);
    std::string& rd = reads[i];
    for (size_t p = 0; p < r.seq.size(); p++) {
      if (r.seq[p] != '-') rd.push_back(r.seq[p]);
      rd += r.ins[p];
    }
    int from, count;
    pg.lifted(r.start, r.end + 2, &from, &count);
    if (rd.empty() || rd.size() > MIA_HIP_MAX_READ || count <= 0) {
      fprintf(stderr, "ccheck_hip: record %s/%c: %s\n", r.id.c_str(), r.segment,
              rd.empty() ? "no bases" : count <= 0 ? "no contaminant bases under it" : "longer than 256 bases");
      exit(3);
    }
    wstart[i] = from; wlen[i] = count;
    bases += rd;
    roff[i + 1] = (int64_t)bases.size();
  }
  std::vector<int32_t> score(n), as(n), ae(n), rstart(n);
  const int stride = MIA_HIP_MAX_READ;
  std::vector<int16_t> cols(n * (size_t)stride);
  if (n) {
    std::vector<uint8_t> zeros(n, 0), ones(n, 1);
    std::vector<int32_t> z32(n, 0);
    if (mia_hip_set_pssm(g, &m.fpsm[0][0][0], &m.fpsm[0][0][0]) != MIA_HIP_OK) die(g, "set_pssm");
    if (mia_hip_upload_reads(g, (int64_t)n, bases.data(), roff.data(), zeros.data(), ones.data(), z32.data(), z32.data()) != MIA_HIP_OK) die(g, "upload_reads");
    if (mia_hip_align_windows(g, mia_ref.data(), (int64_t)mia_ref.size(), wstart.data(), wlen.data()) != MIA_HIP_OK) die(g, "align_windows");
    if (mia_hip_get_alignments(g, score.data(), as.data(), ae.data()) != MIA_HIP_OK) die(g, "get_alignments");
    if (mia_hip_get_scripts(g, cols.data(), stride, rstart.data()) != MIA_HIP_OK) die(g, "get_scripts");
  }
  // the two gapped rows of populate_pwaln_to_begin (src/mia.c:1440-1497) from the column script
  std::vector<Pairwise> pw(n);
  for (size_t i = 0; i < n; i++) {
    const int16_t* cs = &cols[i * (size_t)stride];
    const std::string& rd = reads[i];
    Pairwise& q = pw[i];
    q.start = as[i] - (int)wstart[i];
    int prev = -1;
    for (size_t r = 0; r < rd.size(); r++) {
      if (cs[r] == MIA_HIP_COL_CLIP) continue;
      if (cs[r] == MIA_HIP_COL_INSERT) { q.ref_row.push_back('-'); q.frag_row.push_back(rd[r]); continue; }
      const int gc = rstart[i] + cs[r];
      if (prev >= 0) for (int k = prev + 1; k < gc; k++) { q.ref_row.push_back(mia_ref[(size_t)k]); q.frag_row.push_back('-'); }
      q.ref_row.push_back(mia_ref[(size_t)gc]); q.frag_row.push_back(rd[r]);
      prev = gc;
    }
  }

  // ---- pass one: which of the weak sites does some read actually show as the contaminant's base?
  if (verbose >= 2) fputs("Pass one: finding actually diagnostic positions.\n", stderr);
  for (size_t i = 0; i < n; i++) {
    const Record& r = m.recs[i];
    const Pairwise& q = pw[i];
    const std::string lifted = hum.seq.substr((size_t)wstart[i], (size_t)wlen[i]);
    const std::string the_ass = r.start <= (int)m.seq.size() && r.end >= r.start - 1 ? m.seq.substr((size_t)r.start, (size_t)(r.end - r.start + 1)) : std::string();
    if (verbose >= 3) {
      fprintf(stderr, "%s/%c:\n  %d potentially diagnostic positions", r.id.c_str(), r.segment,
              (int)std::distance(sites.lower_bound(r.start), sites.lower_bound(r.end + 1)));
      if (verbose >= 4) { putc(':', stderr); putc(' ', stderr); print_sites(stderr, sites.lower_bound(r.start), sites.lower_bound(r.end + 1), 0); }
      fprintf(stderr, "; range:  %d..%d\n", r.start, r.end);
    }
    if (verbose >= 5) {
      fprintf(stderr, "\nraw read: %s\nlifted:   %s\nassembly: %s\n\naln.read: %s\naln.assm: %s\nmatches:  ", reads[i].c_str(), lifted.c_str(),
              the_ass.c_str(), r.seq.c_str(), the_ass.c_str());
      for (size_t k = 0; k < the_ass.size() && k < r.seq.size(); k++) putc(the_ass[k] == r.seq[k] ? '*' : ' ', stderr);
      fprintf(stderr, "\n\naln.read: %s\naln.ref:  %s\nmatches:  ", q.frag_row.c_str(), q.ref_row.c_str());
      for (size_t k = 0; k < q.ref_row.size() && k < q.frag_row.size(); k++) putc(q.ref_row[k] == q.frag_row[k] ? '*' : ' ', stderr);
      putc('\n', stderr);
      putc('\n', stderr);
    }
    Walker w(pg, m.seq, r, lifted, q);
    if (verbose) {
      const int A = (int)pg.con.size();
      const char c1 = w.col < A ? w.con() : '\0', c2 = w.col < A ? w.ass() : '\0';
      if (c1 != w.ref0() || c1 == '-')
        fprintf(stderr, "huh? (R+%d) %.10s %.10s\n", q.start, w.col < A ? pg.con.c_str() + w.col : "", w.in_ref.c_str());
      if (c2 != w.in_ass() && c2 != '-')
        fprintf(stderr, "huh? (A+%d) %.10s %.10s\n", q.start, w.col < A ? pg.ass.c_str() + w.col : "", w.ai < m.seq.size() ? m.seq.c_str() + w.ai : "");
    }
    for (; w.more(); w.step()) {
      if (!differs(w.con(), w.ass())) continue;
      Sites::iterator it = sites.find(w.ass_pos);
      if (it == sites.end()) {
        fprintf(stderr, "diagnostic site not found: %d\n", w.ass_pos);
      } else {
        Site& s = it->second;
        if (verbose >= 4) fprintf(stderr, "diagnostic pos.: %d %c(%c)/%c %c/%c", w.ass_pos, s.consensus, w.ref0(), w.frag_ref(), w.in_ass(), w.frag_ass());
        if (w.frag_ref() != w.frag_ass()) {
          if (verbose >= 4) fputs(" in disagreement.", stderr);
        } else {
          const bool maybe_clean = consistent(o.adna, s.assembly, w.frag_ass()), maybe_dirt = consistent(o.adna, s.consensus, w.frag_ref());
          if (!maybe_clean && maybe_dirt && s.strength == WEAK) {
            if (verbose >= 4) fputs(" possible contaminant, upgraded to `effective'.", stderr);
            s.contaminant = w.frag_ref();
            s.strength = EFFECTIVE;
          }
        }
      }
      if (verbose >= 4) putc('\n', stderr);
    }
    if (verbose >= 4) fprintf(stderr, "\n");
  }
  for (Sites::iterator i = sites.begin(); i != sites.end();) {
    if (i->second.strength == WEAK) i = sites.erase(i); else ++i;
  }
  {
    int t = 0;
    for (const auto& kv : sites) if (transversion(kv.second.consensus, kv.second.assembly)) t++;
    if (o.table) printf("%d\t%d\t", t, num_strong);
    else {
      printf("  %d effectively diagnostic positions", (int)sites.size());
      if (o.span_from != 0 || o.span_to != INT_MAX) printf(" in range [%d,%d)", o.span_from, o.span_to);
      printf(", %d of which are transversions.\n\n", t);
    }
  }
  if (verbose >= 3) print_sites(stderr, sites.begin(), sites.end(), '\n');

  // ---- pass two: a verdict per fragment
  if (verbose >= 2) fputs("Pass two: classifying fragments.\n", stderr);
  typedef std::map<std::string, std::pair<Verdict, int>> Backs;
  Backs backs, backs2;
  for (size_t i = 0; i < n; i++) {
    const Record& r = m.recs[i];
    const Pairwise& q = pw[i];
    Verdict klass = UNKNOWN, klass2 = UNKNOWN;
    int votes = 0, votes2 = 0;
    const Sites::const_iterator lo = sites.lower_bound(r.start), hi = sites.lower_bound(r.end + 1);
    if (std::distance(lo, hi) < o.min_sites) {
      if (verbose >= 3) { fputs(r.id.c_str(), stderr); putc('/', stderr); putc(r.segment, stderr); fputs(": no diagnostic positions\n", stderr); }
    } else {
      if (verbose >= 3) {
        fprintf(stderr, "%s/%c: %d diagnostic positions", r.id.c_str(), r.segment, (int)std::distance(lo, hi));
        if (verbose >= 4) { putc(':', stderr); putc(' ', stderr); print_sites(stderr, lo, hi, 0); }
        fprintf(stderr, "; range:  %d..%d\n", r.start, r.end);
      }
      int from, count;
      pg.lifted(r.start, r.end + 1, &from, &count);
      Walker w(pg, m.seq, r, hum.seq.substr((size_t)from, (size_t)count), q);
      for (; w.more(); w.step()) {
        if (!differs(w.con(), w.ass())) continue;
        const Sites::const_iterator it = sites.find(w.ass_pos);
        if (it == sites.end()) continue;
        const Site& s = it->second;
        if (verbose >= 4)
          fprintf(stderr, "diagnostic pos. %s: %d %c(%c)/%c %c/%c", s.strength == STRONG ? "(strong)" : "  (weak)", w.ass_pos, s.consensus, w.ref0(),
                  w.frag_ref(), w.in_ass(), w.frag_ass());
        if (w.frag_ref() != w.frag_ass()) {
          if (verbose >= 4) fputs(" in disagreement.\n", stderr);
        } else {
          const bool maybe_clean = consistent(o.adna, s.assembly, w.frag_ass()), maybe_dirt = consistent(o.adna, s.consensus, w.frag_ref());
          if (verbose >= 4) {
            fputs(maybe_dirt ? " " : " in", stderr);
            fputs("consistent/", stderr);
            fputs(maybe_clean ? "" : "in", stderr);
            fputs("consistent\n", stderr);
          }
          vote(klass2, votes2, maybe_clean, maybe_dirt && !maybe_clean);
          if (s.strength == STRONG) vote(klass, votes, maybe_clean, maybe_dirt);
        }
      }
      if (verbose >= 4) putc('\n', stderr);
    }
    const Backs::const_iterator b1 = backs.find(r.id), b2 = backs2.find(r.id);
    switch (r.segment) {
      case 'b':
        backs[r.id] = std::make_pair(klass, votes);
        backs2[r.id] = std::make_pair(klass2, votes2);
        if (verbose >= 3) putc('\n', stderr);
        break;
      case 'f':
        if (b1 == backs.end()) { fputs(r.id.c_str(), stderr); fputs("/f is missing its back.\n", stderr); }
        else { votes += b1->second.second; klass = merge(klass, b1->second.first); }
        if (b2 == backs2.end()) { fputs(r.id.c_str(), stderr); fputs("/f is missing its back.\n", stderr); }
        else { votes2 += b1->second.second; klass2 = merge(klass2, b1->second.first); }   // the first table again, as the reference has it (src/ccheck.cc:848-849)
        // a front counts like a whole fragment
      case 'a':
        if (verbose >= 2) fprintf(stderr, "%s is %s (%d votes)\n", r.id.c_str(), LABEL[klass], votes);
        if (verbose >= 2) fprintf(stderr, "%s is %s (%d votes)\n", r.id.c_str(), LABEL[klass2], votes2);
        if (verbose >= 3) putc('\n', stderr);
        summary[klass]++;
        summary2[klass2]++;
        break;
      default:
        fputs("don't know how to handle fragment type ", stderr);
        putc(r.segment, stderr);
        putc('\n', stderr);
    }
  }
  if (!o.table) {
    int t = 0;
    for (const auto& kv : sites) if (kv.second.strength == STRONG) t++;
    printf("  strongly diagnostic positions: %d\n", t);
  }
  print_summary(summary, o.table);
  if (!o.table) printf("  effectively diagnostic positions: %d\n", (int)sites.size());
  else printf("%d\t", (int)sites.size());
  print_summary(summary2, o.table);
  putc('\n', stdout);
  return 0;
}

std::string exe_dir() {
  char buf[4096];
  const ssize_t k = readlink("/proc/self/exe", buf, sizeof buf - 1);
  if (k <= 0) return ".";
  buf[k] = 0;
  char* slash = strrchr(buf, '/');
  if (slash) *slash = 0;
  return buf;
}

}  // namespace

int main(int argc, char* const argv[]) {
  static const struct option longopts[] = {
      {"reference", required_argument, 0, 'r'}, {"ancient", no_argument, 0, 'a'}, {"verbose", no_argument, 0, 'v'},
      {"help", no_argument, 0, 'h'}, {"transversions", no_argument, 0, 't'}, {"span", required_argument, 0, 's'},
      {"maxd", required_argument, 0, 'd'}, {"table", no_argument, 0, 'T'}, {"shoot", no_argument, 0, 'F'},
      {"foot", no_argument, 0, 'F'}, {0, 0, 0, 0}};
  Options o;
  Contaminant hum;
  bool have_ref = false;
  if (argc == 0) { usage("ccheck_hip"); return 0; }
  int opt;
  do {
    opt = getopt_long(argc, argv, "r:avhts:d:n:MfTF", longopts, 0);
    switch (opt) {
      case 'r':
        if (!read_fasta_first(optarg, &hum)) { fprintf(stderr, "Cannot read a FASTA record from %s\n", optarg); return 1; }
        have_ref = true;
        break;
      case 'a': o.adna = true; break;
      case 'v': ++o.verbose; break;
      case ':': fputs("missing option argument\n", stderr); break;
      case '?': fputs("unknown option\n", stderr); break;
      case 'h': usage(argv[0]); return 1;
      case 't': break;                       // accepted and, as in the reference, without effect
      case 's':
        sscanf(optarg, "%u-%u", (unsigned*)&o.span_from, (unsigned*)&o.span_to);
        if (o.span_from) o.span_from--;
        break;
      case 'n': o.min_sites = atoi(optarg); break;
      case 'd': o.maxd = atoi(optarg); break;
      case 'M': break;
      case 'f': o.clever = false; break;
      case 'T': o.table = true; break;
      case 'F': o.really = true; break;
    }
  } while (opt != -1);
  if (optind == argc) { usage(argv[0]); return 1; }
  if (!have_ref) {
    // the reference links its contaminant consensus in (src/mt311.c); here it is a data file beside the program
    const std::string fn = exe_dir() + "/share/mt311.fa";
    if (!read_fasta_first(fn.c_str(), &hum)) { fprintf(stderr, "Cannot read the built-in contaminant consensus %s\n", fn.c_str()); return 1; }
  }
  if (!only_iupac(hum.seq)) fputs("FUBAR'ed FastA file: contaminant sequence contains gap symbols.\n", stderr);

  mia_hip_ctx* g = nullptr;
  if (mia_hip_create(&g, 0) != MIA_HIP_OK) die(g, "no MI355X context (there is no CPU fallback)");

  if (o.table) {
    fputs("#Filename\tAln.dist\t#diff\t#weak\t#tv", stdout);
    for (int i = 0; i != 2; ++i) {
      fputs(i ? "\t#eff" : "\t#strong", stdout);
      for (size_t v = 0; v != sizeof(LABEL) / sizeof(LABEL[0]); ++v) {
        putchar('\t');
        fputs(LABEL[v], stdout);
        if (i) putchar('\'');
      }
    }
    putchar('\n');
  }
  int maxd = o.maxd;   // once derived from the first file it stays (src/ccheck.cc:477)
  for (; optind != argc; ++optind) {
    const std::string infile = o.clever ? newest_iteration(argv[optind]) : std::string(argv[optind]);
    const int rc = check_file(g, o, &maxd, hum, infile);
    if (rc) { mia_hip_destroy(g); return rc; }
  }
  mia_hip_destroy(g);
  return 0;
}
