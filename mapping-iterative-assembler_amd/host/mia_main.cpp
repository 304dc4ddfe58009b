// mia_main.cpp -- `mia_hip`: the mia command line on top of libmia_hip.so.
//
// Host side of the boundary: same flags, same .maln text, same
// read-store semantics as the reference's main() (/root/reference/src/mia_main.c:394-989),
// but every alignment, tally and consensus call runs on the GPU through the C ABI of
// include/mia_hip.h.  What stays here is what the reference also does on the host
// around the hot path: option parsing, FASTA/FASTQ parsing (src/io.c:35-386), the
// read store, and formatting the .maln (src/map_alignment.c:283-382).
//
// Adapter trimming (-T, -a) runs on the GPU (mia_hip_trim).  Options of the reference that are
// outside the accelerated path ("OUT OF SCOPE": -u -U -A -C -h -D -I -q)
// are rejected with a message instead of being silently ignored.
#include <ctype.h>
#include <fcntl.h>
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <errno.h>
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mia_hip.h"
#include "ingest.h"

namespace {

const int MAX_ID_LEN = 100, MAX_DESC_LEN = 128, MAX_READ = 256, PSSM_DEPTH = 15, MAX_ITER = 30;

struct Pssm { int32_t sm[31][5][5]; };

void flat_pssm(Pssm* p) {   // init_flatsubmat, src/pssm.c:96-126
  for (int d = 0; d < 31; d++) {
    for (int i = 0; i < 5; i++) {
      for (int j = 0; j < 4; j++) p->sm[d][i][j] = (i == j) ? 200 : -600;
      p->sm[d][i][4] = -100;
    }
    for (int j = 0; j < 5; j++) p->sm[d][4][j] = -10;
  }
}
void revcom_pssm(const Pssm* in, Pssm* out) {   // revcom_submat, src/pssm.c:53-91
  for (int d = 0; d < 31; d++)
    for (int i = 0; i < 5; i++)
      for (int j = 0; j < 5; j++) out->sm[30 - d][i][j] = in->sm[d][i < 4 ? 3 - i : 4][j < 4 ? 3 - j : 4];
}
bool read_pssm_file(const char* fn, Pssm* p) {   // read_pssm, src/io.c:408-503
  FILE* f = fopen(fn, "r");
  if (!f) return false;
  char line[4096];
  for (int d = 0; d < 31; d++) {
    if (!fgets(line, sizeof line, f) || !strstr(line, "# Matrix for position")) { fclose(f); fprintf(stderr, "Problem parsing matrix file: %s\n", fn); exit(2); }
    for (int i = 0; i < 4; i++) {
      if (!fgets(line, sizeof line, f)) line[0] = 0;
      sscanf(line, "%d\t%d\t%d\t%d", &p->sm[d][i][0], &p->sm[d][i][1], &p->sm[d][i][2], &p->sm[d][i][3]);
      p->sm[d][i][4] = -100;
    }
    for (int j = 0; j < 5; j++) p->sm[d][4][j] = -10;
    if (!fgets(line, sizeof line, f)) line[0] = 0;
  }
  fclose(f);
  return true;
}
// find_read_pssm, src/mia_main.c:299-328: the path itself, then ./, then DATA_PATH/matrices/
void find_read_pssm(const char* fn, Pssm* p) {
  if (read_pssm_file(fn, p)) return;
  const char* base = strrchr(fn, '/');
  base = base ? base + 1 : fn;
  std::string here = std::string("./") + base;
  if (read_pssm_file(here.c_str(), p)) return;
  const char* dp = getenv("MIA_DATA_PATH");
  if (dp) {
    std::string d = std::string(dp) + "/matrices/" + base;
    if (read_pssm_file(d.c_str(), p)) return;
  }
  fprintf(stderr, "Sadly, the substitution matrix cannot be read. Bye bye.\n");
  exit(10);
}

char revcom_char(char b) {   // src/map_align.c:418-432
  static const char tbl[] = "TVGH\0\0CD\0\0M\0KN\0\0\0YSAABWXR\0";
  char r = 0;
  if (b == '-') return '-';
  if (b >= 'A' && b <= 'Z') r = tbl[b - 'A'];
  else if (b >= 'a' && b <= 'z') r = (char)(tbl[b - 'a'] + 32);
  if (r) return r;
  fprintf(stderr, "Do not know how to revcom \"%c\"\n", b);
  return 'N';
}

struct Ref { std::string id, desc, seq; };

bool read_fasta_ref(const char* fn, Ref* r) {   // src/io.c:287-386
  FILE* f = fopen(fn, "r");
  if (!f) return false;
  int c = fgetc(f);
  if (c != '>') { fclose(f); return false; }
  bool done = false;
  while (!isspace(c = fgetc(f)) && !done) {
    if (c == EOF) { fclose(f); return false; }
    r->id.push_back((char)c);
    if ((int)r->id.size() == MAX_ID_LEN) done = true;
  }
  done = false;
  if (c == '\n') done = true; else c = fgetc(f);
  while (c != '\n' && !done) {
    if (c == EOF) { fclose(f); return false; }
    r->desc.push_back((char)c);
    if ((int)r->desc.size() == MAX_DESC_LEN) done = true;
    c = fgetc(f);
  }
  c = fgetc(f);
  while (c != '>' && c != EOF) {
    if (!isspace(c)) r->seq.push_back((char)c);
    c = fgetc(f);
  }
  fclose(f);
  return true;
}

using ingest::Read;      // one input record (host/ingest.h: the reference's reader, on all host threads)

// One fsdb entry (FragSeq, src/types.h:110-143) -- only what the path needs.
struct Frag {
  std::string id, desc, seq;   // seq already reverse-complemented when rc && strand_known (src/fsdb.c:209-227)
  int rc, strand_known, as, ae, score;
  int trimmed = 0;             // adapter found by trim_frag: seq is already cut at the trim point (src/fsdb.c:199-203)
};

// a buffer that is written before it is read: no zero fill (std::vector value-initialises, a page fault per 4 KB at first touch)
template <class T>
struct HostBuf {
  T* p = nullptr; size_t n = 0;
  explicit HostBuf(size_t count) : p((T*)malloc(count * sizeof(T) + 64)), n(count) { if (!p) { fprintf(stderr, "mia_hip: out of memory\n"); _exit(1); } }
  HostBuf(const HostBuf&) = delete; HostBuf& operator=(const HostBuf&) = delete;
  ~HostBuf() { free(p); }
  T* data() { return p; } const T* data() const { return p; }
  size_t bytes() const { return n * sizeof(T); }
  T& operator[](size_t i) { return p[i]; } const T& operator[](size_t i) const { return p[i]; }
};

template <class F>
void run_parallel(int T, F&& fn) {
  if (T <= 1) { fn(0); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) th.emplace_back([&fn, t] { fn(t); });
  fn(0);
  for (auto& x : th) x.join();
}

void die(mia_hip_ctx* g, const char* what) {
  fprintf(stderr, "%s: %s\n", what, g ? mia_hip_last_error(g) : "no context");
  fflush(stderr);
  _exit(1);      // (may be called from one of several per-GPU threads: no static destructors under the others' feet)
}

void help() {
  printf("\n\nMIA -- Mapping Iterativ Assembler V 1.0 (MI355X build, libmia_hip)\n"
         "usage: mia_hip -r <reference fasta> -f <fasta/fastq reads> [-m maln root] [-s matrix] [-c] [-i|-n]\n"
         "               [-p cons code] [-H hard score cut] [-S slope -N intercept] [-k kmer] [-M] [-F] [-g gpu[,gpu...]]\n");
}

}  // namespace

int main(int argc, char** argv) {
  std::string maln_root = "assembly.maln.iter", ref_fn, frag_fn;
  int hard_cut = 0, circular = 0, iterate = 1, final_only = 0, score_cut_set = 0, kmer = -1, soft_mask = 0, cc = 1, any = 0;
  std::vector<int> gpus{0};             // -g 0  or  -g 0,1,2,3: the read store is split over these GPUs (contiguous fsdb blocks)
  double slope = 200.0, intercept = 0.0;
  int do_adapter_trimming = 0;
  // src/mia_main.c:462-466: the two built-in adapters, Neandertal by default
  const std::string neand_adapt = "GTCAGACACGCAACAGGGGATAGGCAAGGCACACAGGGGATAGG", stand_adapt = "CTGAGACACGCAACAGGGGATAGGCAAGGCACACAGGGGATAGG";
  std::string adapter = neand_adapt;
  Pssm anc, rcanc;
  flat_pssm(&anc);
  int ich;
  while ((ich = getopt(argc, argv, "s:r:f:m:a:p:H:I:S:N:k:q:g:FTcinuhDMUAC::")) != -1) {
    switch (ich) {
      case 'c': circular = 1; break;
      case 'n': iterate = 0; break;
      case 'i': iterate = 1; break;
      case 'p': cc = atoi(optarg); any = 1; break;
      case 'H':
        hard_cut = atoi(optarg);
        if (hard_cut <= 0) { fprintf(stderr, "Hard cutoff (-H) must be positive\n"); help(); exit(0); }
        any = 1;
        break;
      case 'M': soft_mask = 1; break;
      case 's': find_read_pssm(optarg, &anc); any = 1; break;
      case 'r': ref_fn = optarg; any = 1; break;
      case 'k': kmer = atoi(optarg); any = 1; break;
      case 'f': frag_fn = optarg; any = 1; break;
      case 'm': maln_root = optarg; any = 1; break;
      case 'S': slope = atof(optarg); score_cut_set = 1; break;
      case 'N': intercept = atof(optarg); score_cut_set = 1; break;
      case 'F': final_only = 1; break;
      case 'g': {
        // a comma-separated list of device numbers, nothing else: no empty fields, no signs, no blanks
        gpus.clear();
        bool ok = *optarg != 0;
        for (const char* q = optarg; ok && *q;) {
          if (!isdigit((unsigned char)*q)) { ok = false; break; }
          char* e = nullptr;
          const long v = strtol(q, &e, 10);
          if (v > 1023 || (*e && *e != ',') || (*e == ',' && !e[1])) { ok = false; break; }
          gpus.push_back((int)v);
          q = *e ? e + 1 : e;
        }
        if (!ok || gpus.empty()) { fprintf(stderr, "mia_hip: -g wants a list of GPU numbers such as 0 or 0,1,2,3 (got \"%s\")\n", optarg); exit(1); }
        break;
      }
      case 'T': do_adapter_trimming = 1; break;
      case 'a':                                        // src/mia_main.c:558-578
        if (strlen(optarg) > 127) { fprintf(stderr, "That adapter is too big!\nMIA will use the standard adapter.\n"); adapter = stand_adapt; }
        else if (strlen(optarg) > 1) adapter = optarg;
        else adapter = (optarg[0] == 'n' || optarg[0] == 'N') ? neand_adapt : stand_adapt;
        break;
      case 'u': case 'U': case 'A': case 'C': case 'h': case 'D': case 'I': case 'q':
        fprintf(stderr, "option -%c is outside the MI355X-accelerated path (see DESIGN.md, section 7) and is not supported by mia_hip\n", ich);
        exit(2);
      default: help(); exit(0);
    }
  }
  if (!any) { help(); exit(0); }
  if (optind != argc) { fprintf(stderr, "There seems to be some extra cruff on the command line that mia does not understand.\n"); exit(0); }
  revcom_pssm(&anc, &rcanc);

  // MIA_HIP_TIMING=1: wall time of each phase on stderr
  const bool timing = getenv("MIA_HIP_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[mia_hip timing] %-22s %9.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  time_t now = time(NULL);
  fprintf(stderr, "Starting assembly of %s\nusing %s\nas reference at %s\n", frag_fn.c_str(), ref_fn.c_str(), asctime(localtime(&now)));

  Ref ref;
  if (!read_fasta_ref(ref_fn.c_str(), &ref)) { fprintf(stderr, "Problem reading reference sequence file %s\n", ref_fn.c_str()); exit(1); }

  // one context per GPU; every per-read step below runs on all of them at once, each on its contiguous share of the reads
  const int NG = (int)gpus.size();
  // The same GPU several times (-g 0,0): as many contexts on that one device, one host thread each, joined by the library's
  // in-process transport instead of RCCL (which refuses two ranks on one device) -- the sharded driver on a single GPU.
  // Mixing that with other devices is refused: the in-process transport copies between the contexts' buffers directly.
  bool same_gpu = NG > 1;
  for (int k = 1; k < NG; k++) same_gpu = same_gpu && gpus[(size_t)k] == gpus[0];
  if (!same_gpu)
    for (int k = 0; k < NG; k++)
      for (int j = 0; j < k; j++)
        if (gpus[(size_t)j] == gpus[(size_t)k]) { fprintf(stderr, "mia_hip: -g lists GPU %d twice beside other GPUs (list every GPU once, or one GPU several times)\n", gpus[(size_t)k]); exit(1); }
  std::vector<mia_hip_ctx*> G((size_t)NG, nullptr);
  // (the HIP runtime takes a third of a second to come up: that happens beside the parsing of the input)
  std::thread gpu_init([&] {
    for (int k = 0; k < NG; k++) {
      if (mia_hip_create(&G[(size_t)k], gpus[(size_t)k]) != MIA_HIP_OK) { fprintf(stderr, "mia_hip: no usable MI355X (gfx950) device %d; there is no CPU fallback\n", gpus[(size_t)k]); fflush(stderr); _exit(1); }
      if (mia_hip_set_pssm(G[(size_t)k], &anc.sm[0][0][0], &rcanc.sm[0][0][0]) != MIA_HIP_OK) die(G[(size_t)k], "set_pssm");
    }
  });
  auto on_gpus = [&](const std::function<void(int)>& fn) {      // fn(k) for every GPU k, concurrently (the collectives need that)
    if (NG == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int k = 0; k < NG; k++) th.emplace_back(fn, k);
    for (auto& t : th) t.join();
  };
  auto share = [&](int64_t total, int k) { return total * k / NG; };    // first item of GPU k's contiguous share

  // ---- read the fragments (read_next_seq loop, src/mia_main.c:759): the reference's character-level reader, run over the
  //      mapped file on all host threads at once (host/ingest.h); MIA_HIP_THREADS=1 is the plain sequential walk
  std::vector<Read> reads;
  bool fastq = false;
  {
    int T = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("MIA_HIP_THREADS")) T = atoi(e);
    if (T > 64) T = 64;
    std::string msgs;
    const auto ti = std::chrono::steady_clock::now();
    if (!ingest::read_all(frag_fn.c_str(), T, &reads, &fastq, &msgs)) { fprintf(stderr, "Cannot open %s\n", frag_fn.c_str()); fflush(stderr); _exit(1); }   // (the GPU start-up thread is still running: no atexit handlers under its feet)
    fputs(msgs.c_str(), stderr);
    if (timing) {
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti).count();
      fprintf(stderr, "[mia_hip timing] read input alone       %9.1f ms  (%zu records, %.1f M records/s, beside the GPU start-up)\n", ms, reads.size(), reads.size() / ms / 1e3);
    }
  }
  (void)fastq;
  gpu_init.join();
  mia_hip_ctx* g = G[0];
  lap("init + read input");
  fprintf(stderr, "Starting to align sequences to the reference...\n");

  // ---- adapter trimming (trim_frag, src/mia_main.c:771-775): the read ends at the trim point from here on
  if (do_adapter_trimming && !reads.empty()) {
    std::vector<int64_t> toff;
    std::string tb;
    std::vector<size_t> tsrc;
    toff.push_back(0);
    for (size_t i = 0; i < reads.size(); i++)
      if (!reads[i].seq.empty()) { tb += reads[i].seq; toff.push_back((int64_t)tb.size()); tsrc.push_back(i); }
    std::vector<uint8_t> tr(tsrc.size());
    std::vector<int32_t> tp(tsrc.size());
    const int64_t nt = (int64_t)tsrc.size();
    on_gpus([&](int k) {
      const int64_t lo = share(nt, k), hi = share(nt, k + 1);
      if (hi <= lo) return;
      std::vector<int64_t> o2((size_t)(hi - lo) + 1);
      for (int64_t q = lo; q <= hi; q++) o2[(size_t)(q - lo)] = toff[(size_t)q] - toff[(size_t)lo];
      if (mia_hip_trim(G[(size_t)k], adapter.c_str(), hi - lo, tb.data() + toff[(size_t)lo], o2.data(), tr.data() + lo, tp.data() + lo) != MIA_HIP_OK) die(G[(size_t)k], "trim");
    });
    int emptied = 0;
    for (size_t k = 0; k < tsrc.size(); k++) {
      if (!tr[k]) continue;
      Read& r = reads[tsrc[k]];
      r.trimmed = 1;
      if (tp[k] + 1 <= 0) { r.seq.clear(); emptied++; }    // nothing but adapter: the reference's len2 = 0 case is undefined, the read is left out
      else r.seq.resize((size_t)tp[k] + 1);
    }
    if (emptied) fprintf(stderr, "mia_hip: %d read(s) consist of adapter only and are left out\n", emptied);
    lap("adapter trimming");
  }

  // ---- pass 1 on the GPU (new_kmer_filter + sg_align)
  std::vector<int64_t> off;
  std::string bases;
  std::vector<size_t> src;   // reads with at least one base (an empty read cannot be aligned: src/mia.c:1283-1286)
  off.push_back(0);
  for (size_t i = 0; i < reads.size(); i++)
    if (!reads[i].seq.empty()) { bases += reads[i].seq; off.push_back((int64_t)bases.size()); src.push_back(i); }
  const int64_t n1 = (int64_t)src.size();
  std::vector<int32_t> p_score((size_t)n1), p_as((size_t)n1), p_ae((size_t)n1);
  std::vector<uint8_t> p_rc((size_t)n1), p_fl((size_t)n1);
  on_gpus([&](int k) {
    const int64_t lo = share(n1, k), hi = share(n1, k + 1);
    if (hi <= lo) return;
    std::vector<int64_t> o2((size_t)(hi - lo) + 1);
    for (int64_t q = lo; q <= hi; q++) o2[(size_t)(q - lo)] = off[(size_t)q] - off[(size_t)lo];
    if (mia_hip_pass1(G[(size_t)k], ref.seq.c_str(), (int32_t)ref.seq.size(), circular, kmer, soft_mask, hi - lo, bases.data() + off[(size_t)lo], o2.data(),
                      p_score.data() + lo, p_rc.data() + lo, p_as.data() + lo, p_ae.data() + lo, p_fl.data() + lo) != MIA_HIP_OK) die(G[(size_t)k], "pass1");
  });
  lap("pass 1");
  fprintf(stderr, "\n");

  // ---- fsdb (add_virgin_fs2fsdb, src/fsdb.c:194-231) and the pass-1 AlnSeq slots
  std::vector<Frag> fsdb;
  std::vector<uint8_t> slot_dropped;   // pass-1 slots, merge order
  std::vector<int> first_slot, n_slots;
  int n_unknown = 0;
  // (which reads were kept and where their slots start: one cheap pass; the strings are moved into the store by all threads)
  std::vector<int64_t> kept_k;
  kept_k.reserve((size_t)n1);
  {
    int slots = 0;
    for (int64_t k = 0; k < n1; k++) {
      if (!(p_fl[k] & MIA_HIP_P1_KEPT)) continue;
      kept_k.push_back(k);
      first_slot.push_back(slots);
      const int ns = (p_fl[k] & MIA_HIP_P1_SPLIT) ? 2 : 1;
      n_slots.push_back(ns);
      slots += ns;
      if (!(p_fl[k] & MIA_HIP_P1_STRAND_KNOWN)) n_unknown++;
    }
    slot_dropped.assign((size_t)slots, 0);
  }
  const int n = (int)kept_k.size();
  fsdb.resize((size_t)n);
  {
    // reverse-strand reads are kept reverse-complemented from here on (add_virgin_fs2fsdb, src/fsdb.c:209-227)
    int T = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("MIA_HIP_THREADS")) T = atoi(e);
    T = std::max(1, std::min(std::min(T, 64), n / 8192 + 1));
    run_parallel(T, [&](int t) {
      for (int i = (int)((int64_t)n * t / T), hi = (int)((int64_t)n * (t + 1) / T); i < hi; i++) {
        const int64_t k = kept_k[(size_t)i];
        Read& r = reads[src[(size_t)k]];      // each input read is looked at once: move its strings into the store
        Frag& f = fsdb[(size_t)i];
        f.id = std::move(r.id); f.desc = std::move(r.desc); f.seq = std::move(r.seq); f.trimmed = r.trimmed;
        f.rc = p_rc[k]; f.strand_known = (p_fl[k] & MIA_HIP_P1_STRAND_KNOWN) ? 1 : 0;
        f.as = p_as[k]; f.ae = p_ae[k]; f.score = p_score[k];
        if (f.rc && f.strand_known) {
          std::reverse(f.seq.begin(), f.seq.end());
          for (auto& ch : f.seq) ch = revcom_char(ch);
        }
      }
    });
  }
  const int pass1_records = (int)slot_dropped.size();   // culled_maln->size, frozen here (src/mia.c:54)
  if (n == 0) { fprintf(stderr, "No sequence aligned to the reference with a score of at least 2000.\n"); for (auto* c : G) mia_hip_destroy(c); exit(0); }
  if (NG > n) { fprintf(stderr, "mia_hip: more GPUs (%d) than reads in the store (%d)\n", NG, n); exit(1); }
  (void)n_unknown;   // strand-unknown reads (score exactly 2000) stay in the fsdb: their stale pointers are followed as the reference does
  std::vector<int32_t> len((size_t)n), score((size_t)n), as((size_t)n), ae((size_t)n);
  std::vector<int64_t> f0((size_t)n, -1);     // pass-1 front slot of every read
  for (int i = 0; i < n; i++) { len[i] = (int32_t)fsdb[i].seq.size(); score[i] = fsdb[i].score; }

  bool on_device = false;    // true once the read store is resident and re-aligned (the first cull runs on pass-1 scores)
  auto score_cut = [&](double* s, double* ic) {   // cull_maln_from_fsdb, src/mia.c:429-442
    *s = slope; *ic = intercept;
    if (hard_cut <= 0 && !score_cut_set) {
      // the regression's first pass (integer sums) runs on the device; with equally long reads that is all of it
      mia_hip_score_cut(score.data(), len.data(), NULL, n, s, ic);
    }
    if (*s <= 0) *s = 100.0;
  };
  // first cull on the pass-1 records: only its `dropped` marks survive (src/mia_main.c:848)
  fprintf(stderr, "Repeat and score filtering\n");
  {
    double s, ic;
    score_cut(&s, &ic);
    for (int i = 0; i < n; i++) {
      const double min_score = hard_cut > 0 ? (double)hard_cut : (double)(ic + (s * len[i]));
      if (score[i] < min_score)
        for (int k = 0; k < n_slots[i]; k++) slot_dropped[first_slot[i] + k] = 1;   // front_asp and back_asp (src/mia.c:471-478)
    }
  }
  // clean_FSDB (src/mia.c:400-406) cannot remove anything: every kept read scores >= 2000

  lap("read store");
  // ---- upload the read store: GPU k holds the contiguous fsdb block [lo_k, hi_k); AlnSeq slot numbers stay global
  std::vector<int> lo_of((size_t)NG + 1);
  for (int k = 0; k <= NG; k++) lo_of[(size_t)k] = (int)share(n, k);
  {
    std::vector<int64_t> b0((size_t)n, -1);
    for (int i = 0; i < n; i++) {
      as[i] = fsdb[i].as; ae[i] = fsdb[i].ae;
      f0[(size_t)i] = first_slot[i];
      if (n_slots[i] == 2) b0[(size_t)i] = (int64_t)first_slot[i] + 1;
    }
    on_gpus([&](int k) {
      const int lo = lo_of[(size_t)k], hi = lo_of[(size_t)k + 1], m = hi - lo;
      std::vector<int64_t> o2((size_t)m + 1, 0);
      std::vector<uint8_t> rc((size_t)m), sk((size_t)m);
      for (int i = lo; i < hi; i++) o2[(size_t)(i - lo) + 1] = o2[(size_t)(i - lo)] + (int64_t)fsdb[i].seq.size();
      HostBuf<char> b2((size_t)o2[(size_t)m] + 1);
      {
        int T = (int)std::thread::hardware_concurrency();
        if (const char* e = getenv("MIA_HIP_THREADS")) T = atoi(e);
        T = std::max(1, std::min(std::min(T, 64) / NG, m / 8192 + 1));
        run_parallel(T, [&](int t) {
          for (int i = lo + (int)((int64_t)m * t / T), e2 = lo + (int)((int64_t)m * (t + 1) / T); i < e2; i++) {
            memcpy(b2.data() + o2[(size_t)(i - lo)], fsdb[i].seq.data(), fsdb[i].seq.size());
            rc[(size_t)(i - lo)] = (uint8_t)fsdb[i].rc; sk[(size_t)(i - lo)] = (uint8_t)fsdb[i].strand_known;
          }
        });
      }
      mia_hip_ctx* c = G[(size_t)k];
      if (mia_hip_upload_reads(c, m, b2.data(), o2.data(), rc.data(), sk.data(), as.data() + lo, ae.data() + lo) != MIA_HIP_OK) die(c, "upload_reads");
      if (mia_hip_set_slot_dropped(c, slot_dropped.data(), (int64_t)slot_dropped.size()) != MIA_HIP_OK) die(c, "set_slot_dropped");
      // fs->front_asp / back_asp and fs->score after pass 1 (src/mia.c:1619-1653): the back slot of every split read, and
      // for strand-unknown reads (never re-aligned, src/mia_main.c:178) also the front slot and the score
      if (mia_hip_set_pass1_state(c, f0.data() + lo, b0.data() + lo, score.data() + lo) != MIA_HIP_OK) die(c, "set_pass1_state");
      if (mia_hip_set_read_base(c, lo) != MIA_HIP_OK) die(c, "set_read_base");
    });
    if (same_gpu) {
      void* group = nullptr;
      if (mia_hip_loopback_create(NG, &group) != MIA_HIP_OK) { fprintf(stderr, "mia_hip: the in-process transport could not be made\n"); exit(1); }
      for (int k = 0; k < NG; k++) {
        mia_hip_collectives t;
        if (mia_hip_loopback_table(group, k, &t) != MIA_HIP_OK || mia_hip_comm_attach(G[(size_t)k], &t) != MIA_HIP_OK) die(G[(size_t)k], "comm_attach");
      }
    } else if (NG > 1) {
      // RCCL over xGMI: rank k = GPU k of the list; the id goes from here to every thread
      char id[MIA_HIP_COMM_ID_BYTES];
      if (mia_hip_comm_unique_id(id) != MIA_HIP_OK) { fprintf(stderr, "mia_hip: RCCL (librccl.so.1) is needed for more than one GPU and could not be opened\n"); exit(1); }
      on_gpus([&](int k) { if (mia_hip_comm_init(G[(size_t)k], id, NG, k) != MIA_HIP_OK) die(G[(size_t)k], "comm_init"); });
    }
  }
  lap("upload");

  // make_ref_upper (src/mia.c:642-648): iteration 1 aligns to the upper-cased input reference
  std::string cons = ref.seq;
  for (auto& ch : cons) ch = (char)toupper((unsigned char)ch);
  std::string ref_id = ref.id, ref_desc = ref.desc;
  int stride = 4;                        // columns of the script table: longest stored read, rounded up
  for (int i = 0; i < n; i++) stride = std::max(stride, (len[i] + 3) & ~3);
  // what comes back from the device for the .maln writer: 230 bytes per read.  Plain allocations (every byte is written by a copy
  // before it is read) whose pages are touched here by all host threads at once -- a zero-filled std::vector of 200 MB, faulted
  // in page by page by the first copy from the device, was 80 ms in front of the first iteration
  HostBuf<int16_t> cols((size_t)n * stride);
  HostBuf<int32_t> rstart((size_t)n), rparams((size_t)n * 8);
  HostBuf<uint8_t> dF((size_t)n), dB((size_t)n);
  HostBuf<int64_t> back_slot((size_t)n);
  std::vector<int32_t> gaps;
  {
    int T = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("MIA_HIP_THREADS")) T = atoi(e);
    T = std::max(1, std::min(T, 64));
    struct Span { char* p; size_t bytes; };
    const Span spans[] = {{(char*)cols.data(), cols.bytes()}, {(char*)rstart.data(), rstart.bytes()}, {(char*)rparams.data(), rparams.bytes()},
                          {(char*)dF.data(), dF.bytes()}, {(char*)dB.data(), dB.bytes()}, {(char*)back_slot.data(), back_slot.bytes()}};
    run_parallel(T, [&](int t) {
      for (const Span& sp : spans)
        for (size_t o = sp.bytes * (size_t)t / (size_t)T & ~(size_t)4095, hi = sp.bytes * (size_t)(t + 1) / (size_t)T; o < hi; o += 4096) sp.p[o] = 0;
    });
  }

  lap("host buffers");
  std::string next_cons;     // consensus_assembly_string of the iteration just run
  auto iteration = [&](int iter_num) {
    // reiterate_assembly + cull_maln_from_fsdb + consensus_assembly_string (src/mia_main.c:24-280, 931-963) as one call per
    // GPU; with several GPUs the call also exchanges tallies, gaps, links and insert events (RCCL)
    const double sn[2] = {slope, intercept};
    std::vector<std::string> outs((size_t)NG);
    on_gpus([&](int k) {
      std::string& out = outs[(size_t)k];
      out.assign(cons.size() * 2 + (1 << 20), '\0');
      int64_t clen = 0;
      if (mia_hip_iterate(G[(size_t)k], cons.c_str(), (int32_t)cons.size(), circular, hard_cut, score_cut_set ? sn : nullptr, cc, &out[0], (int64_t)out.size(),
                          &clen) != MIA_HIP_OK) die(G[(size_t)k], "iterate");
      out.resize((size_t)clen);
      if (k == 0) lap("    mia_hip_iterate");
      const int lo = lo_of[(size_t)k];
      if (mia_hip_get_alignments(G[(size_t)k], score.data() + lo, as.data() + lo, ae.data() + lo) != MIA_HIP_OK) die(G[(size_t)k], "get_alignments");
    });
    for (int k = 1; k < NG; k++) if (outs[(size_t)k] != outs[0]) { fprintf(stderr, "mia_hip: the GPUs disagree on the consensus (internal error)\n"); exit(1); }
    next_cons = std::move(outs[0]);
    lap("  iteration (realign, cull, tally, consensus)");
    on_device = true;
    if (iter_num > 1) { ref_id = "ConsAssem." + std::to_string(iter_num); ref_desc = "iteration assembly"; }
    fprintf(stderr, "Repeat and score filtering\n");
  };

  // write_ma (src/map_alignment.c:283-382) from the device results
  auto write_maln = [&](const std::string& fn) {
    const int L = (int)cons.size();
    const int wl = circular ? std::min(L, MAX_READ) : 0;
    std::string wrapped = cons + cons.substr(0, (size_t)wl);
    on_gpus([&](int k) {
      mia_hip_ctx* c = G[(size_t)k];
      const size_t lo = (size_t)lo_of[(size_t)k];
      if (mia_hip_get_scripts(c, cols.data() + lo * (size_t)stride, stride, rstart.data() + lo) != MIA_HIP_OK) die(c, "get_scripts");
      if (mia_hip_get_dropped(c, dF.data() + lo, dB.data() + lo) != MIA_HIP_OK) die(c, "get_dropped");
      if (mia_hip_get_record_params(c, rparams.data() + lo * 8, back_slot.data() + lo) != MIA_HIP_OK) die(c, "get_record_params");
    });
    gaps.assign((size_t)L + 1, 0);
    if (mia_hip_get_tally(g, NULL, gaps.data()) != MIA_HIP_OK) die(g, "get_tally");
    // ---- the AlnSeq records of this iteration, by slot (merge order: front record, then back record of every strand-known
    //      read, src/map_align.c:866-954).  Start and end follow from the alignment's end points alone, so the records can be
    //      listed, culled and sorted before a single character of them is formatted.
    std::vector<int32_t> rec_read, rec_start, rec_end;
    std::vector<uint8_t> rec_half;                 // 0: whole read ('a') or front ('f'), 1: back ('b')
    std::vector<int32_t> first_rec((size_t)n + 1, 0);
    rec_read.reserve((size_t)n + 1024); rec_start.reserve((size_t)n + 1024); rec_end.reserve((size_t)n + 1024); rec_half.reserve((size_t)n + 1024);
    for (int i = 0; i < n; i++) {
      first_rec[(size_t)i] = (int32_t)rec_read.size();
      if (!fsdb[i].strand_known) continue;
      int start = as[i], end = ae[i];
      if (end > L) end -= L;                      // src/mia_main.c:259-263
      if (start > end) {                          // split_pwaln, src/mia.c:1376-1438
        rec_read.push_back(i); rec_half.push_back(0); rec_start.push_back(start); rec_end.push_back(L - 1);
        rec_read.push_back(i); rec_half.push_back(1); rec_start.push_back(0); rec_end.push_back(end);
      } else {
        rec_read.push_back(i); rec_half.push_back(0); rec_start.push_back(start); rec_end.push_back(end);
      }
    }
    first_rec[(size_t)n] = (int32_t)rec_read.size();
    const size_t n_recs = rec_read.size();
    // cull_maln_from_fsdb (src/mia.c:451-481): front_asp of every read, then back_asp if it is not NULL.  back_asp is
    // never cleared (src/mia_main.c:259-276): for a read that is not split any more it addresses the record that
    // sits in that slot NOW, which is thereby listed twice.
    std::vector<int32_t> order;
    order.reserve(n_recs + 16);
    for (int i = 0; i < n; i++) {
      if (!fsdb[i].strand_known) {               // both pointers still address the pass-1 slots (src/mia_main.c:178)
        if (f0[(size_t)i] >= 0 && f0[(size_t)i] < (int64_t)n_recs) order.push_back((int32_t)f0[(size_t)i]);
        if (back_slot[(size_t)i] >= 0 && back_slot[(size_t)i] < (int64_t)n_recs) order.push_back((int32_t)back_slot[(size_t)i]);
        continue;
      }
      const int32_t k = first_rec[(size_t)i];
      const bool split = first_rec[(size_t)i + 1] - k == 2;
      order.push_back(k);
      if (split) order.push_back(k + 1);
      else if (back_slot[(size_t)i] >= 0 && back_slot[(size_t)i] < (int64_t)n_recs) order.push_back((int32_t)back_slot[(size_t)i]);
    }
    // sort_aln_frags: stable by (start, end) over cull order (src/map_align.c:393-414; glibc's qsort is a merge sort here).
    // Two stable counting sorts -- by end, then by start -- give the same order in O(n + L).
    {
      std::vector<int32_t> tmp(order.size());
      auto pass = [&](const std::vector<int32_t>& key, const std::vector<int32_t>& in, std::vector<int32_t>& out) {
        std::vector<int64_t> cnt((size_t)L + 3, 0);
        for (int32_t k : in) cnt[(size_t)std::min(std::max(key[(size_t)k], 0), L + 1) + 1]++;
        for (size_t v = 1; v < cnt.size(); v++) cnt[v] += cnt[v - 1];
        for (int32_t k : in) out[(size_t)cnt[(size_t)std::min(std::max(key[(size_t)k], 0), L + 1)]++] = k;
      };
      pass(rec_end, order, tmp);
      pass(rec_start, tmp, order);
    }
    // ---- the header
    std::string head;
    {
      time_t t = time(NULL);
      head += "/* map_alignment [V1.0] */ ";
      head += asctime(localtime(&t));
      int size = L + 1;                             // src/mia_main.c:66, src/mia.c:669-675
      if (circular) while (L + wl >= size) size *= 2;
      char b[256];
      snprintf(b, sizeof b, "MALN_NAS %d\nMALN_SIZ %d\nMALN_COC %d\n__REFERENCE__\nID ", (int)order.size(), pass1_records, cc);
      head += b; head += ref_id; head += "\nDESC "; head += ref_desc;
      snprintf(b, sizeof b, "\nLEN %d\nSIZE %d\nSEQ ", L, size);
      head += b; head += cons; head += "\nGAPS";
      for (int p = 0; p < L; p++) { char nb[16]; auto r = std::to_chars(nb, nb + sizeof nb, gaps[(size_t)p]); head += ' '; head.append(nb, r.ptr); }
      snprintf(b, sizeof b, "\n__PSSM__\nDEPTH %d\nFPSM:\n", PSSM_DEPTH);
      head += b;
      for (int pass2 = 0; pass2 < 2; pass2++) {
        const Pssm& m = pass2 ? rcanc : anc;
        if (pass2) head += "RPSM:\n";
        for (int d = 0; d < 31; d++) {
          for (int r = 0; r < 5; r++) { snprintf(b, sizeof b, "%d %d %d %d %d\n", m.sm[d][r][0], m.sm[d][r][1], m.sm[d][r][2], m.sm[d][r][3], m.sm[d][r][4]); head += b; }
          head += "\n";
        }
      }
      head += "__ALNSEQS__\n";
    }
    // ---- one record as text (write_ma, src/map_alignment.c:283-382), straight from the device's script of its read:
    //      the two gapped strings of populate_pwaln_to_begin (src/mia.c:1440-1497), split_pwaln at the origin
    //      (src/mia.c:1376-1438), merge_pwaln_into_maln (src/map_align.c:866-954: one character per reference column, inserts
    //      before the column that follows them, a trailing insert run is lost as in the reference) and the depth codes of
    //      pop_smp_from_FSDB (src/fsdb.c:542-619).  The depth code of a column reached after `a` bases of its record is
    //      depth(q[0] + q[1] + a, q[2] - (q[1] + a) - 1); q = {0, 0, flen+blen} for a front record and {flen, bases in the
    //      front, flen+blen} for a back record, unless a formerly split read further down the fsdb overwrote the codes
    //      through its stale back_asp (mia_hip_get_record_params).
    struct Scratch { std::string rg, fg, seq, smp; std::vector<std::pair<int, int>> ins; std::string ins_text; };
    auto emit = [&](int32_t k, Scratch& w, std::string& o) {
      const int i = rec_read[(size_t)k];
      const int half = rec_half[(size_t)k];
      const Frag& f = fsdb[i];
      const int16_t* cs = &cols[(size_t)i * stride];
      const int len2 = (int)f.seq.size();
      w.rg.clear(); w.fg.clear();
      int prev = -1;
      for (int r = 0; r < len2; r++) {
        if (cs[r] == MIA_HIP_COL_CLIP) continue;
        if (cs[r] == MIA_HIP_COL_INSERT) { w.rg.push_back('-'); w.fg.push_back(f.seq[(size_t)r]); continue; }
        const int gc = rstart[i] + cs[r];
        if (prev >= 0) for (int c = prev + 1; c < gc; c++) { w.rg.push_back(wrapped[(size_t)c]); w.fg.push_back('-'); }
        w.rg.push_back(wrapped[(size_t)gc]); w.fg.push_back(f.seq[(size_t)r]);
        prev = gc;
      }
      const bool split = first_rec[(size_t)i + 1] - first_rec[(size_t)i] == 2;
      size_t lo = 0, hi = w.rg.size();
      if (split) {
        int rp = rec_start[(size_t)first_rec[(size_t)i]];
        size_t ap = 0;
        while (rp < L && ap < w.rg.size()) { if (w.rg[ap] != '-') rp++; ap++; }
        if (half) lo = ap; else hi = ap;
      }
      w.seq.clear(); w.ins.clear(); w.ins_text.clear();
      {
        bool in = false;
        size_t run0 = 0;
        for (size_t c = lo; c < hi; c++) {
          if (w.rg[c] == '-') { if (!in) { run0 = w.ins_text.size(); in = true; } w.ins_text.push_back(w.fg[c]); }
          else {
            if (in) { w.ins.push_back({(int)w.seq.size(), (int)run0}); in = false; }
            w.seq.push_back(w.fg[c]);
          }
        }
        if (in) w.ins_text.resize(run0);          // (a run of inserted bases at the very end of a record is not stored)
      }
      const int st = rec_start[(size_t)k], en = rec_end[(size_t)k];
      const int32_t* q = &rparams[(size_t)i * 8 + (half ? 4 : 0)];
      w.smp.clear();
      {
        const int span = en - st + 1;
        size_t ii = 0;
        int act = 0;
        for (int p2 = 0; p2 < span; p2++) {
          while (ii < w.ins.size() && w.ins[ii].first < p2) ii++;
          if (ii < w.ins.size() && w.ins[ii].first == p2) {
            const size_t e = ii + 1 < w.ins.size() ? (size_t)w.ins[ii + 1].second : w.ins_text.size();
            act += (int)(e - (size_t)w.ins[ii].second);
          }
          const int dff = q[0] + q[1] + act, dfb = q[2] - (q[1] + act) - 1;
          w.smp.push_back(dff <= PSSM_DEPTH ? (char)('A' + dff) : (dfb < PSSM_DEPTH ? (char)('A' + 2 * PSSM_DEPTH - dfb) : (char)('A' + PSSM_DEPTH)));
          if (p2 < (int)w.seq.size() && w.seq[(size_t)p2] != '-') act++;
        }
      }
      auto num = [&](int v) { char b[16]; auto r = std::to_chars(b, b + sizeof b, v); o.append(b, r.ptr); };
      o += "ID ";
      if (split) { o.append(f.id, 0, (size_t)MAX_ID_LEN - 1); o += half ? "_b" : "_f"; } else o += f.id;
      o += "\nDESC "; o += f.desc; o += "\nSCORE "; num(score[i]);
      o += "\nNUM_INPUTS 1\nSTART "; num(st); o += "\nEND "; num(en);
      o += f.rc ? "\nRC 1\nTR " : "\nRC 0\nTR "; o += f.trimmed ? '1' : '0'; o += "\nDR "; o += (half ? dB[i] : dF[i]) ? '1' : '0';
      o += "\nSEG "; o += split ? (half ? 'b' : 'f') : 'a'; o += "\nSEQ "; o += w.seq; o += "\nSMP "; o += w.smp; o += "\nINS_POS";
      for (size_t x = 0; x < w.ins.size(); x++) {
        const size_t e = x + 1 < w.ins.size() ? (size_t)w.ins[x + 1].second : w.ins_text.size();
        o += ' '; num(w.ins[x].first); o += ' '; o.append(w.ins_text, (size_t)w.ins[x].second, e - (size_t)w.ins[x].second);
      }
      o += '\n';
    };
    // ---- format on all host threads (contiguous runs of the sorted order), then every thread writes its own stretch of
    //      the file (pwrite at the offset the sizes before it add up to)
    const size_t nr = order.size();
    int TF = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("MIA_HIP_THREADS")) TF = atoi(e);
    TF = std::max(1, std::min(std::min(TF, 64), (int)(nr / 4096) + 1));
    std::vector<std::string> text((size_t)TF);
    const int fd = open(fn.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) { fprintf(stderr, "Cannot write %s\n", fn.c_str()); exit(1); }
    std::vector<size_t> sizes((size_t)TF, 0), at((size_t)TF + 1, 0);
    run_parallel(TF, [&](int t) {
      std::string& o = text[(size_t)t];
      Scratch w;
      const size_t lo = nr * (size_t)t / (size_t)TF, hi = nr * (size_t)(t + 1) / (size_t)TF;
      o.reserve((hi - lo) * 330 + 4096);
      for (size_t q = lo; q < hi; q++) emit(order[q], w, o);
      sizes[(size_t)t] = o.size();
    });
    at[0] = head.size();
    for (int t = 0; t < TF; t++) at[(size_t)t + 1] = at[(size_t)t] + sizes[(size_t)t];
    // positioned writes from all threads -- or, where the target cannot seek (a pipe, a FIFO: ESPIPE), the pieces one after the
    // other with write(); an interrupted call (EINTR) is repeated
    std::atomic<bool> seekable{true};
    auto put = [&](const char* p2, size_t len, size_t off) -> bool {
      while (len) {
        const ssize_t w = pwrite(fd, p2, len, (off_t)off);
        if (w < 0 && errno == EINTR) continue;
        if (w < 0 && errno == ESPIPE) { seekable.store(false); return false; }
        if (w <= 0) { fprintf(stderr, "Cannot write %s\n", fn.c_str()); _exit(1); }
        p2 += w; len -= (size_t)w; off += (size_t)w;
      }
      return true;
    };
    auto put_seq = [&](const char* p2, size_t len) {
      while (len) {
        const ssize_t w = write(fd, p2, len);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) { fprintf(stderr, "Cannot write %s\n", fn.c_str()); _exit(1); }
        p2 += w; len -= (size_t)w;
      }
    };
    if (put(head.data(), head.size(), 0))
      run_parallel(TF, [&](int t) { if (seekable.load()) put(text[(size_t)t].data(), text[(size_t)t].size(), at[(size_t)t]); });
    if (!seekable.load()) {
      put_seq(head.data(), head.size());
      for (int t = 0; t < TF; t++) put_seq(text[(size_t)t].data(), text[(size_t)t].size());
    }
    for (int t = 0; t < TF; t++) std::string().swap(text[(size_t)t]);
    close(fd);
    lap("write .maln");
  };

  auto consensus = [&]() { return next_cons; };     // mia_hip_iterate has called it already

  // ---- main loop (src/mia_main.c:878-976)
  int iter_num = 1;
  iteration(iter_num);
  if (!iterate || !final_only) write_maln(maln_root + "." + std::to_string(iter_num));
  if (iterate) {
    fprintf(stderr, "Generating new assembly consensus\n");
    std::string next = consensus();
    while (next != cons && iter_num < MAX_ITER) {
      iter_num++;
      cons = next;
      fprintf(stderr, "Starting assembly iteration %d\n", iter_num);
      iteration(iter_num);
      if (!final_only) { fprintf(stderr, "Writing maln file for iteration %d\n", iter_num); write_maln(maln_root + "." + std::to_string(iter_num)); }
      next = consensus();
    }
    if (next == cons) fprintf(stderr, "Assembly convergence - writing final maln\n");
    else fprintf(stderr, "Assembly did not converge after %d rounds, quitting\n", iter_num);
    if (final_only) write_maln(maln_root + "." + std::to_string(iter_num));
  }
  now = time(NULL);
  fprintf(stderr, "Assembly finished at %s\n", asctime(localtime(&now)));
  // (one context: the process ends here and the driver takes the device memory back with it -- handing every buffer back one by one
  // first was 30-40 ms; several contexts leave their communicator in an orderly way)
  if (NG > 1) for (auto* c : G) mia_hip_destroy(c);
  // every file is closed; the read store's millions of small strings need not be handed back one by one
  fflush(stdout); fflush(stderr);
  _exit(0);
}
