// myers_ond_body.h -- the furthest-reaching D-path recurrence of /root/reference/src/myers_align.c:10-99 (myers_diff) in the
// shape the device runs it: one row of diagonals per step, every diagonal of a row independent of the others (each reads
// three cells of the row before), snakes followed 32, then 128 characters at a time on sequences packed as 4-bit IUPAC bitmaps
// (src/myers_align.h:40-67: two characters match when their bitmaps intersect).
//
// Where the long pairs come from: ccheck aligns the assembly with the contaminant consensus once per file, 16.6 kb against
// 16.6 kb, maxd = max(len)/10 (src/ccheck.cc:477-480).  The work of this algorithm is O((len_a + len_b) * d), the memory
// for the walk back (d+1)^2 cells; the bit-vector kernels (mia_myers_kernels.h) walk len_a/64 words for every character of
// seq_b whatever the distance, which is the better trade only for pairs whose distance is a large part of their length.
//
// Plain C++ (no HIP headers): csrc/mia_myers_kernels.h runs ond_cell / ond_snake per lane, tests/emul/emu_myers_ond.cpp
// runs the same functions row by row on the CPU against the reference's answers (tests/test_emul_myers_ond.py).
#pragma once
#include <stdint.h>

#include <string>

#ifndef MIA_HD
#define MIA_HD
#endif

namespace mia {

constexpr int32_t OND_NONE = INT32_MIN / 2;    // a cell no D-path reaches (outside -d..d, or cut off by the ends of the sequences)
constexpr int OND_PAD_WORDS = 18;              // zero words behind a packed sequence: a snake step reads seventeen words from any start inside it

// cell (d, k) of the table: row d starts at d*d and holds the diagonals -d .. d
MIA_HD inline size_t ond_at(int d, int k) { return (size_t)d * (size_t)d + (size_t)(k + d); }

// how far the snake runs from (x on seq_b, y on seq_a): the number of leading positions whose bitmaps intersect.  Positions
// at and behind the end of a sequence hold the bitmap 0 (the padding), which intersects nothing -- the loop bound of
// src/myers_align.c:35 without a comparison.  A, B: eight bitmaps per word, lowest nibble first.
// {hi, lo} >> s (0 <= s < 32), low word: one v_alignbit_b32 on the device
MIA_HD inline uint32_t ond_funnel(uint32_t lo, uint32_t hi, int s) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}

// CHUNKS * 8 characters from (x, y): the number that match before the first that does not, -1 if all of them match.  No early
// exit on purpose: with one the compiler fetches every word where it is first needed -- CHUNKS dependent LDS round trips
// instead of one (measured on the ccheck pair: 1.9 us per row of D-paths).
template <int CHUNKS, class Words>
MIA_HD inline int ond_snake_step(const Words& A, const Words& B, int y, int x) {
  const int wa = y >> 3, sa = (y & 7) * 4, wb = x >> 3, sb = (x & 7) * 4;
  uint32_t a[CHUNKS + 1], b[CHUNKS + 1];
#pragma unroll
  for (int i = 0; i <= CHUNKS; i++) { a[i] = A[wa + i]; b[i] = B[wb + i]; }
  int r = -1;
#pragma unroll
  for (int i = CHUNKS - 1; i >= 0; i--) {
    const uint32_t t = ond_funnel(a[i], a[i + 1], sa) & ond_funnel(b[i], b[i + 1], sb);
    const uint32_t miss = ~(t | (t >> 1) | (t >> 2) | (t >> 3)) & 0x11111111u;      // bit 4q set: nibble q is empty
    r = miss ? i * 8 + (__builtin_ctz(miss) >> 2) : r;
  }
  return r;
}

// The snake of ONE diagonal (the host's statement, and what the emulation checks the shared one against): 32 characters,
// then 128 at a time.
template <class Words>
MIA_HD inline int ond_snake(const Words& A, const Words& B, int y, int x, int la, int lb) {
  if (x < 0 || y < 0 || x >= lb || y >= la) return 0;
  int r = ond_snake_step<4>(A, B, y, x);
  if (r >= 0) return r;
  int run = 32;
  for (;;) {
    r = ond_snake_step<16>(A, B, y + run, x + run);
    if (r >= 0) return run + r;
    run += 128;
  }
}

// A snake shared by the 64 lanes of a wavefront.  The snakes along the alignment's own path add up to the length of the
// sequences whatever the distance, and one lane following them 128 characters per step was the floor of the whole kernel
// (85 us for a 16.6 kb pair at distance 0).  Here lane l of a round looks at the eight characters from `off + 8 l` on: 512
// characters per round.  ond_shared_chunk: what one lane sees -- 0..7 characters that match before one that does not, 8: all.
constexpr int OND_SHARED_SPAN = 8 * 64;
template <class Words>
MIA_HD inline int ond_shared_chunk(const Words& A, const Words& B, int y, int x, int la, int lb) {
  if (x >= lb || y >= la) return 0;                     // (at or behind an end: nothing matches, and nothing is read)
  const int r = ond_snake_step<1>(A, B, y, x);
  return r < 0 ? 8 : r;
}

// x of cell (d, k) before its snake, from the row before (src/myers_align.c:26-32; a cell that does not exist counts as
// OND_NONE, which is what the seven cases of the reference amount to).  prev(k') = cell (d-1, k') or OND_NONE.
template <class Prev>
MIA_HD inline int32_t ond_cell(int d, int k, const Prev& prev) {
  if (d == 0) return 0;
  const int32_t keep = prev(k), from_left = prev(k - 1), from_right = prev(k + 1);
  int32_t x = keep == OND_NONE ? OND_NONE : keep + 1;
  if (from_left != OND_NONE && from_left + 1 > x) x = from_left + 1;
  if (from_right != OND_NONE && from_right > x) x = from_right;
  return x;
}

// has the path on diagonal k, standing at x, arrived?  (src/myers_align.c:39-40)
MIA_HD inline bool ond_arrived(int mode, int x, int k, int la, int lb) {
  const int y = x - k;
  return (mode == 1 || y == la) && (mode == 2 || x == lb);
}

// The walk back through a finished table (host side; the table is rows 0 .. dist of the device's, or the host's own): the
// reference's preferences -- mismatch, then a seq_b-only column, then a seq_a-only column, else one step down the snake
// (src/myers_align.c:47-83).  In the prefix modes a D-path may run past the end of the sequence that need not be consumed
// (the recurrence has no bound there): the reference then copies that sequence's terminator into the row, which ends the
// C string early.  Same here, without reading past the terminator.
inline bool ond_walk_back(const int32_t* v, const char* a, int la, const char* b, int lb, int dist, int end_k, std::string* row_a, std::string* row_b) {
  auto at = [&](int d, int k) -> int32_t { return (k < -d || k > d) ? OND_NONE : v[ond_at(d, k)]; };
  auto ca = [&](int y) { return y < la ? a[y] : '\0'; };
  auto cb = [&](int x) { return x < lb ? b[x] : '\0'; };
  std::string ra, rb;   // built back to front
  int k = end_k, x = at(dist, k), y = x - k;
  if (x == OND_NONE) return false;
  for (int d = dist; d != 0;) {
    if (k != -d && k != d && x == at(d - 1, k) + 1) { d--; x--; y--; rb.push_back(cb(x)); ra.push_back(ca(y)); }
    else if (k > -d + 1 && x == at(d - 1, k - 1) + 1) { x--; k--; d--; rb.push_back(cb(x)); ra.push_back('-'); }
    else if (k < d - 1 && x == at(d - 1, k + 1)) { k++; y--; d--; rb.push_back('-'); ra.push_back(ca(y)); }
    else { x--; y--; rb.push_back(cb(x)); ra.push_back(ca(y)); }
    if (x < 0 || y < 0) return false;
  }
  while (x > 0) { x--; rb.push_back(cb(x)); ra.push_back(ca(x)); }
  row_a->assign(ra.rbegin(), ra.rend());
  row_b->assign(rb.rbegin(), rb.rend());
  return true;
}

}  // namespace mia
