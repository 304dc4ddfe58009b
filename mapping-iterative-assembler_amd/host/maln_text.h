// maln_text.h -- reading the text of a .maln file the way the reference's read_ma does (src/map_alignment.c:384-607:
// fgets with MAX_LINE_LEN, sscanf "KEY %s" / "KEY %d", fscanf " %d %s" for the insert list).  Shared by ma_hip and
// ccheck_hip; each keeps its own record layout.
#pragma once
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

namespace maln_text {

constexpr int MAX_LINE_LEN = 1000000;   // src/params.h:22

struct Cursor {
  const char* p;
  const char* end;
  // fgets(line, MAX_LINE_LEN, f): at most MAX_LINE_LEN-1 characters, newline included
  bool line(std::string* out) {
    if (p >= end) { out->clear(); return false; }
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* stop = nl ? nl + 1 : end;
    if (stop - p > MAX_LINE_LEN - 1) stop = p + (MAX_LINE_LEN - 1);
    out->assign(p, stop);
    p = stop;
    return true;
  }
  void skip_ws() { while (p < end && isspace((unsigned char)*p)) p++; }
  bool literal(const char* lit) {   // fscanf(f, "LITERAL"): stops at the first mismatch
    for (; *lit; lit++) { if (p < end && *p == *lit) p++; else return false; }
    return true;
  }
  bool integer(long* v) {           // fscanf " %d"
    skip_ws();
    const char* q = p;
    if (q < end && (*q == '+' || *q == '-')) q++;
    if (q >= end || !isdigit((unsigned char)*q)) return false;
    char* e = nullptr;
    *v = strtol(p, &e, 10);
    p = e;
    return true;
  }
  bool token(std::string* t) {      // fscanf " %s"
    skip_ws();
    const char* q = p;
    while (q < end && !isspace((unsigned char)*q)) q++;
    if (q == p) return false;
    t->assign(p, q);
    p = q;
    return true;
  }
};

// sscanf(line, "KEY %s") / "KEY %d"
inline bool field(const std::string& line, const char* key, std::string* tok) {
  const size_t k = strlen(key);
  if (line.compare(0, k, key) != 0) return false;
  size_t i = k;
  while (i < line.size() && isspace((unsigned char)line[i])) i++;
  size_t j = i;
  while (j < line.size() && !isspace((unsigned char)line[j])) j++;
  if (j == i) return false;
  tok->assign(line, i, j - i);
  return true;
}
inline bool field_int(const std::string& line, const char* key, int* v) {
  std::string t;
  if (!field(line, key, &t)) return false;
  char* e = nullptr;
  long x = strtol(t.c_str(), &e, 10);
  if (e == t.c_str()) return false;
  *v = (int)x;
  return true;
}

inline void read_matrices(Cursor& c, int depth, int32_t sm[31][5][5]) {
  std::string line;
  for (int i = 0; i <= depth * 2 && i < 31; i++) {
    for (int row = 0; row <= 4; row++) {
      c.line(&line);
      int v[5] = {0, 0, 0, 0, 0};
      sscanf(line.c_str(), "%d %d %d %d %d", &v[0], &v[1], &v[2], &v[3], &v[4]);
      for (int k = 0; k < 5; k++) sm[i][row][k] = v[k];
    }
    c.line(&line);   // blank line between matrices
  }
}


inline bool slurp(const char* fn, std::string* buf) {
  FILE* f = fopen(fn, "r");
  if (!f) return false;
  char chunk[1 << 16];
  size_t n;
  while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) buf->append(chunk, n);
  fclose(f);
  return true;
}

}  // namespace maln_text
