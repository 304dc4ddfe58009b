// ingest.h -- FASTA / FASTQ input at memory speed.
//
// The reference reads its fragments one character at a time (read_fasta / read_fastq / read_next_seq,
// /root/reference/src/io.c:35-281) and the records that come out depend on that state machine's quirks: the id ends at
// the first white space or after MAX_ID_LEN characters, the first description character is stored twice (FASTA only,
// src/io.c:228-234), a description ends at MAX_DESC_LEN characters and whatever is left of that header line is then read
// as SEQUENCE, sequences are upper-cased and cut at 256 bases (src/io.c:246,267-278), any '>' outside a header line begins
// a record, a FASTQ record whose quality line has another length than its sequence ends the input, and so on.  Here the
// SAME machine (next_record below, character for character what host/mia_main.cpp ran on a FILE* before) walks a memory
// image of the file, and it does so on all host threads at once:
//   * the file is cut into as many stretches as there are threads; a thread looks for the first place in its stretch where
//     a record can only BEGIN ("\n>" for FASTA -- after a header's newline the machine reads sequence, and a '>' there
//     always starts a record; for FASTQ an '@' line whose second successor starts with '+') and parses from there up to
//     the next thread's starting point;
//   * the starting points are then VALIDATED in file order: stretch t is accepted only if stretch t-1's machine stopped
//     exactly on stretch t's starting point, in the "between records" state.  If not (a file nobody writes: '>' inside a
//     sequence, an over-long header ...), the machine simply runs on from where it really stood until it meets a later
//     thread's starting point; the records of the stretches it ran over are thrown away.
// The record list is therefore the sequential machine's for EVERY input, and its messages come out in file order.
#pragma once
#include <ctype.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

namespace ingest {

constexpr int MAX_ID_LEN = 100;     // src/params.h
constexpr int MAX_DESC_LEN = 128;
constexpr int MAX_READ = 256;       // INIT_ALN_SEQ_LEN

struct Read { std::string id, desc, seq; int trimmed = 0; };

// a thread's stretch is at least this long (tests shrink it to cut small files into many stretches)
inline size_t& min_stretch_bytes() { static size_t v = (size_t)1 << 20; return v; }

inline bool& use_fast_path() { static bool v = true; return v; }      // (tests switch it off: the character machine alone)

struct Cursor {
  const unsigned char* p;
  const unsigned char* e;
  int get() { return p < e ? *p++ : EOF; }
  void unget() { --p; }
};

// read_fasta / read_fastq, src/io.c:35-281.  false: no further record (end of input, or a record the reference gives up on).
// Messages go to `log` (the caller prints them in file order).
inline bool next_record(Cursor& f, bool fastq, Read* r, std::string* log) {
  r->id.clear(); r->desc.clear(); r->seq.clear();
  int c = f.get();
  if (c == EOF) return false;
  if (c != (fastq ? '@' : '>')) {
    if (fastq) *log += "While reading fastq file, saw record not beginning with @\nMaybe badly formed input? Continuing, anyway...\n";
    return false;
  }
  while (!isspace(c = f.get()) && (int)r->id.size() < MAX_ID_LEN) {
    if (c == EOF) return false;
    r->id.push_back((char)c);
  }
  if (c != '\n') {
    while (c != '\n' && isspace(c)) c = f.get();
    if (!fastq && c != '\n') r->desc.push_back((char)c);
    while (c != '\n' && c != EOF && (int)r->desc.size() < MAX_DESC_LEN) { r->desc.push_back((char)c); c = f.get(); }
  }
  c = f.get();
  if (!fastq) {
    while (c != '>' && c != EOF && (int)r->seq.size() < MAX_READ) {
      if (!isspace(c)) r->seq.push_back((char)toupper(c));
      c = f.get();
    }
    if (c == '>') { f.unget(); return true; }
    if ((int)r->seq.size() == MAX_READ) {
      while (c != '>' && c != EOF) c = f.get();
      if (c == '>') f.unget();
      *log += r->id + " is longer than allowed length: " + std::to_string(MAX_READ) + "\n";
    }
    return true;
  }
  while (c != '\n' && c != EOF && (int)r->seq.size() < MAX_READ) {
    if (!isspace(c)) r->seq.push_back((char)toupper(c));
    c = f.get();
  }
  if ((int)r->seq.size() == MAX_READ) while (c != '\n' && c != EOF) c = f.get();
  c = f.get();
  if (c != '+') { *log += "Problem reading quality line for " + r->id + "\n"; return true; }
  c = f.get();
  while (c != '\n' && c != EOF) c = f.get();
  int q = 0;
  c = f.get();
  while (c != '\n' && c != EOF && q < MAX_READ) { if (!isspace(c)) q++; c = f.get(); }
  if (q == MAX_READ) while (c != '\n' && c != EOF) c = f.get();
  if (q != (int)r->seq.size()) { *log += r->id + " has unequal sequence and qual line lengths\n"; return false; }
  return true;
}

// ---- the common record, without the character loop ----------------------------------------------------------------------
// A record whose header fits the id / description limits and whose sequence stays below 256 bases takes the machine
// through the same states whatever its bytes are: the header ends at its newline, the sequence is every non-blank
// character up to the next '>' (FASTA) or newline (FASTQ).  Those records are cut out with memchr and one table-driven
// copy; the result is what next_record would have stored.  Anything else -- and anything near a limit -- returns false
// with the cursor untouched, and next_record walks the record.  (isspace / toupper of the "C" locale, as the reference
// runs them: main() never calls setlocale.)
struct Tables {
  unsigned char up[256];     // 0 for white space, else toupper
  Tables() {
    for (int c = 0; c < 256; c++) up[c] = (unsigned char)((c == ' ' || (c >= 9 && c <= 13)) ? 0 : (c >= 'a' && c <= 'z' ? c - 32 : c));
  }
};
inline const Tables& tables() { static const Tables t; return t; }

inline bool fast_record(Cursor& f, bool fastq, Read* r) {
  const unsigned char* p = f.p;
  const unsigned char* e = f.e;
  const unsigned char* up = tables().up;
  if (p >= e || *p != (fastq ? '@' : '>')) return false;
  const unsigned char* nl = (const unsigned char*)memchr(p, '\n', (size_t)(e - p));
  if (!nl) return false;
  // id: up to the first white space of the header line, at most MAX_ID_LEN characters
  const unsigned char* q = p + 1;
  while (q < nl && up[*q]) q++;
  if (q - (p + 1) > MAX_ID_LEN) return false;
  const unsigned char* id_end = q;
  while (q < nl && !up[*q]) q++;                       // blanks between id and description
  const size_t rem = (size_t)(nl - q);
  if (rem + (fastq ? 0 : 1) > (size_t)MAX_DESC_LEN) return false;
  // sequence
  unsigned char buf[MAX_READ];
  int n = 0;
  const unsigned char* s = nl + 1;
  const unsigned char* after;
  if (!fastq) {
    const unsigned char* gt = (const unsigned char*)memchr(s, '>', (size_t)(e - s));
    const unsigned char* stop = gt ? gt : e;
    if ((size_t)(stop - s) >= (size_t)MAX_READ) {      // long stretch: count as we go, give up at the limit
      for (const unsigned char* x = s; x < stop; x++) { const unsigned char t = up[*x]; if (t) { if (n == MAX_READ - 1) return false; buf[n++] = t; } }
    } else {
      for (const unsigned char* x = s; x < stop; x++) { const unsigned char t = up[*x]; buf[n] = t; n += t != 0; }
    }
    after = stop;
  } else {
    const unsigned char* nl1 = (const unsigned char*)memchr(s, '\n', (size_t)(e - s));
    if (!nl1 || nl1 + 1 >= e || nl1[1] != '+') return false;
    if ((size_t)(nl1 - s) >= (size_t)MAX_READ) return false;
    for (const unsigned char* x = s; x < nl1; x++) { const unsigned char t = up[*x]; buf[n] = t; n += t != 0; }
    const unsigned char* nl2 = (const unsigned char*)memchr(nl1 + 1, '\n', (size_t)(e - nl1 - 1));
    if (!nl2) return false;
    const unsigned char* qs = nl2 + 1;
    const unsigned char* nl3 = (const unsigned char*)memchr(qs, '\n', (size_t)(e - qs));
    const unsigned char* qe = nl3 ? nl3 : e;
    if ((size_t)(qe - qs) >= (size_t)MAX_READ) return false;
    int qn = 0;
    for (const unsigned char* x = qs; x < qe; x++) qn += up[*x] != 0;
    if (qn != n) return false;                         // (the reference stops reading there: the machine says so)
    after = nl3 ? nl3 + 1 : e;
  }
  r->id.assign((const char*)p + 1, (size_t)(id_end - (p + 1)));
  r->desc.clear();
  if (rem) {
    if (!fastq) r->desc.push_back((char)*q);           // read_fasta stores the first description character twice (src/io.c:228-234)
    r->desc.append((const char*)q, rem);
  }
  r->seq.assign((const char*)buf, (size_t)n);
  f.p = after;
  return true;
}

struct Stretch {
  size_t start = 0, end = 0;      // first byte parsed; where the machine stood when it left off ("between records")
  bool stopped = false;           // the machine returned false: nothing follows
  std::vector<Read> reads;
  std::string log;
};

// records whose first byte lies in [from, limit): the machine starts at `from` and leaves off at the first record boundary
// at or beyond `limit` (or where it gives up)
inline void parse_stretch(const unsigned char* base, size_t size, size_t from, size_t limit, bool fastq, Stretch* out) {
  Cursor cur{base + from, base + size};
  out->start = from;
  Read r;
  for (;;) {
    const size_t at = (size_t)(cur.p - base);
    if (at >= limit) { out->end = at; return; }
    if (!(use_fast_path() && fast_record(cur, fastq, &r)) && !next_record(cur, fastq, &r, &out->log)) { out->end = (size_t)(cur.p - base); out->stopped = true; return; }
    out->reads.emplace_back();
    std::swap(out->reads.back(), r);
  }
}

// first place at or after `from` where a record can only begin; `size` if there is none
inline size_t record_start_after(const unsigned char* base, size_t size, size_t from, bool fastq) {
  size_t p = from;
  while (p < size) {
    const void* nl = memchr(base + p, '\n', size - p);
    if (!nl) return size;
    p = (size_t)((const unsigned char*)nl - base) + 1;
    if (p >= size) return size;
    if (!fastq) { if (base[p] == '>') return p; continue; }
    if (base[p] != '@') continue;
    // a header line: the next line is sequence (never starts with '@'), the one after it starts with '+'
    const void* n1 = memchr(base + p, '\n', size - p);
    if (!n1) return size;
    const size_t l1 = (size_t)((const unsigned char*)n1 - base) + 1;
    if (l1 >= size || base[l1] == '@') continue;
    const void* n2 = memchr(base + l1, '\n', size - l1);
    if (!n2) return size;
    const size_t l2 = (size_t)((const unsigned char*)n2 - base) + 1;
    if (l2 < size && base[l2] == '+') return p;
  }
  return size;
}

struct Mapped {
  const unsigned char* data = nullptr;
  size_t size = 0;
  bool mapped = false;
  std::vector<unsigned char> owned;      // input that cannot be mapped (a pipe): read to the end
  ~Mapped() { if (mapped && data) munmap(const_cast<unsigned char*>(data), size); }
};

inline bool load(const char* path, Mapped* m) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
    m->size = (size_t)st.st_size;
    if (m->size == 0) { close(fd); return true; }
    void* p = mmap(nullptr, m->size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p != MAP_FAILED) {
      (void)madvise(p, m->size, MADV_SEQUENTIAL);      // (advice values are enumerators, not flags: one call each)
      (void)madvise(p, m->size, MADV_WILLNEED);
      m->data = (const unsigned char*)p; m->mapped = true;
      close(fd);
      return true;
    }
  }
  unsigned char buf[1 << 16];
  for (;;) {
    const ssize_t k = read(fd, buf, sizeof buf);
    if (k <= 0) break;
    m->owned.insert(m->owned.end(), buf, buf + k);
  }
  close(fd);
  m->data = m->owned.data(); m->size = m->owned.size();
  return true;
}

// The whole input: the reference's record list and its messages.  threads <= 1: the plain sequential walk.
inline bool read_all(const char* path, int threads, std::vector<Read>* reads, bool* fastq_out, std::string* log) {
  Mapped m;
  if (!load(path, &m)) return false;
  const unsigned char* base = m.data;
  const size_t size = m.size;
  const bool fastq = size > 0 && base[0] == '@';      // find_input_type, src/io.c:11-26
  if (fastq_out) *fastq_out = fastq;
  if (size == 0) return true;
  int T = threads < 1 ? 1 : threads;
  if ((size_t)T > size / min_stretch_bytes()) T = (int)(size / min_stretch_bytes());      // a megabyte per thread at least
  if (T < 1) T = 1;
  std::vector<size_t> start((size_t)T + 1, size);
  start[0] = 0;
  for (int t = 1; t < T; t++) start[(size_t)t] = record_start_after(base, size, size / (size_t)T * (size_t)t, fastq);
  for (int t = 1; t < T; t++) if (start[(size_t)t] < start[(size_t)t - 1]) start[(size_t)t] = start[(size_t)t - 1];
  std::vector<Stretch> st((size_t)T);
  {
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++)
      th.emplace_back([&, t] { if (start[(size_t)t] < start[(size_t)t + 1]) parse_stretch(base, size, start[(size_t)t], start[(size_t)t + 1], fastq, &st[(size_t)t]); });
    parse_stretch(base, size, 0, start[1], fastq, &st[0]);
    for (auto& x : th) x.join();
  }
  // stitch in file order, validating every guessed starting point against where the machine really stood
  std::vector<Stretch*> order;
  std::vector<Stretch> extras;
  extras.reserve((size_t)T + 1);
  order.push_back(&st[0]);
  size_t pos = st[0].end;
  bool stopped = st[0].stopped;
  int t = 1;
  while (!stopped && pos < size) {
    while (t < T && (start[(size_t)t] < pos || start[(size_t)t] >= start[(size_t)t + 1])) t++;      // stretches the machine ran over, empty stretches
    if (t < T && start[(size_t)t] == pos) {
      order.push_back(&st[(size_t)t]);
      pos = st[(size_t)t].end; stopped = st[(size_t)t].stopped;
      t++;
      continue;
    }
    // the machine stands somewhere no thread started from: walk on until the next starting point (or the end)
    extras.emplace_back();
    parse_stretch(base, size, pos, t < T ? start[(size_t)t] : size, fastq, &extras.back());
    order.push_back(&extras.back());
    pos = extras.back().end; stopped = extras.back().stopped;
  }
  // the accepted stretches, moved into one list (each by a thread of its own: a million records are 100 MB of string headers)
  std::vector<size_t> first(order.size() + 1, reads->size());
  for (size_t k = 0; k < order.size(); k++) first[k + 1] = first[k] + order[k]->reads.size();
  reads->resize(first.back());
  {
    std::vector<std::thread> th;
    auto move_in = [&](size_t k) { Read* dst = reads->data() + first[k]; for (auto& r : order[k]->reads) std::swap(*dst++, r); };
    for (size_t k = 1; k < order.size(); k++) th.emplace_back(move_in, k);
    move_in(0);
    for (auto& x : th) x.join();
  }
  for (Stretch* s2 : order) *log += s2->log;
  return true;
}

}  // namespace ingest
