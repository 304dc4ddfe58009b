// diag_filter.h -- decides, WITHOUT running the DP, that the alignment dyn_prog / max_sg_score / find_align_begin
// (/root/reference/src/mia.c:740-981,1278-1302,612-637) would produce for a read in its window is one gap-free
// diagonal, and which.  Bit-parallel: read and reference are 2-bit planes, one shift of the read against the window
// costs a few 64-bit XOR / AND / popcount operations instead of len2 DP cells.  Only for the FLAT matrix
// (src/pssm.c:96-126: match 200, mismatch -600 at every depth) and reads without N; everything else goes to the DP
// kernels.  Shared by the gfx950 kernel (mia_kernels.h: k_diag_filter) and the CPU tests (tests/emul).
//
// Why the verdict is exact.  Write R = len2-1, P(g) = GOP + GEP*g = 1000 + 200 g.  Every cell value S(r,c) of the
// recurrence is the value of a PATH: rows aligned to columns (each scoring 200 / -600 / the N score <= 200), joined
// by events -- a column gap or a row gap of g >= 1 (cost P(g); skipped rows score nothing), or a late start at row
// r > 0 (cost P(r+1), rows before it skipped; src/mia.c:905-948).  A path that starts in row 0 and has no event is a
// diagonal.  Define loss = 200*len2 - value >= 0: 800 per definite mismatch (both ACGT, different), >= 0 per N
// column, 200 per skipped row, P(g) >= 1200 per event.  Let D be the diagonal at offset delta with K definite
// mismatches and no N under it: loss(D) = 800 K.  Checked here:
//   (a) every other diagonal that fits the window has >= K+1 definite mismatches        -> loss >= 800(K+1) > loss(D)
//   (b) K <= 2.  Paths with one event and >= 1 definite mismatch lose >= 1200 + 800 = 2000 > 1600 >= loss(D); paths
//       with >= 2 events lose >= 2400.  With K <= 1 one event alone (1200) already exceeds loss(D) <= 800.
//   (c) K == 2: a path with ONE event and NO definite mismatch loses 1000 + 200 g (+ 200 per skipped ordinary row),
//       which is <= 1600 only for g <= 3.  It is a mismatch-free prefix of the read on one diagonal followed by a
//       mismatch-free suffix on another (or nothing but a suffix after a late start), together covering all but
//       <= 3 rows.  LP = longest mismatch-free prefix over ALL diagonals, LS = longest mismatch-free suffix; the test
//       LP + LS <= R - 3 rules every such path out.  A prefix (suffix) counts only as far as its diagonal stays
//       inside the window; N columns and positions outside the reference are treated as matching anything, which
//       can only make LP / LS larger (more fall-backs, never a wrong verdict).  LP and LS need not be exact either,
//       only not too small: where the reference stretch holds no N, a clean prefix (suffix) of 10 or more rows on a
//       diagonal means that the read's first (last) 10-mer occurs there, so a table of the reference's 10-mers names
//       the only diagonals that can exceed 9 and all others are counted as 9 (diag_step2_kmer) -- a look-up instead
//       of a slide over every diagonal.
// Hence every path other than D ends in row R with a value STRICTLY below D's: max_sg_score's first maximum of the
// last row is D's end cell (ties impossible), and by extending any better path to a diagonal cell along D one sees
// that at every cell of D the diagonal candidate is >= both gap candidates, and D(r-1) >= -600 > -P(r+1), so the
// "start a new alignment" branch (strictly greater only) never fires: trace 0 all the way, find_align_begin stops
// at row 0.  Result: score 200 len2 - 800 K, abc = delta, aec = delta + R, abr = 0, script = consecutive columns --
// exactly what k_align_quad_plain's proof or the trace kernels would deliver.
#pragma once
#include <stdint.h>

#include "mia_layout.h"

namespace mia {

constexpr int PLANE_LEAD = 320;      // wild-card bits in front of reference position 0 (multiple of 64, >= MAX_READ)
constexpr int PLANE_TAIL = 704;      // ... and behind the last code (>= 2*MAX_READ + 3*64)
constexpr int DF_MAX_LEN1 = 768;     // windows wider than the widest DP class are not examined
constexpr int DF_GAP_HINT = 6;       // this many mismatches on the best gap-free diagonal: the read almost surely carries a gap

MIA_HD inline int64_t plane_words(int64_t n_codes) { return (PLANE_LEAD + n_codes + PLANE_TAIL) / 64 + 1; }

struct RefPlanes {
  const uint64_t* lo;   // bit p + PLANE_LEAD: low bit of the code at reference position p
  const uint64_t* hi;   //                     high bit
  const uint64_t* ok;   //                     1 iff the code is A, C, G or T (0: N, or outside the reference)
};

// Where each 10-mer of the reference starts (first DF_KCAP occurrences; cnt beyond that = "too many, ask the planes").
// Rule (c) only cares about clean prefixes / suffixes that are LONG; a clean stretch of 10 or more rows on a diagonal
// means the read's first (last) 10-mer occurs there, so the few diagonals the table names are the only ones that can
// exceed 9 -- no need to slide over all of them.  Sound only where the reference stretch has no N (a wild card is not
// in the table): the caller checks the ok plane.
constexpr int DF_K = 10, DF_KCAP = 4;
constexpr int64_t DF_KTAB = (int64_t)1 << (2 * DF_K);
struct KmerOcc {
  const int32_t* cnt;   // [4^10] occurrences of the 10-mer in the reference
  const int32_t* pos;   // [4^10][DF_KCAP] start positions of the first DF_KCAP of them
};

// position p of the reference: its 10-mer index (code of p+t in bits 2t, 2t+1), or -1 if it holds an N / runs off the end
MIA_HD inline int64_t kmer_at(const uint8_t* codes, int64_t n_codes, int64_t p) {
  if (p < 0 || p + DF_K > n_codes) return -1;
  int64_t idx = 0;
  for (int t = 0; t < DF_K; t++) {
    const uint32_t c = codes[p + t];
    if (c > 3) return -1;
    idx |= (int64_t)c << (2 * t);
  }
  return idx;
}

// The 10-mer at reference position p with its N columns: *idx = the bases (N as 0), *npos = the places of the N, four bits
// each; returns how many there are, -1 off the end of the reference.
MIA_HD inline int kmer_wild_at(const uint8_t* codes, int64_t n_codes, int64_t p, uint32_t* idx, uint64_t* npos) {
  if (p < 0 || p + DF_K > n_codes) return -1;
  uint32_t x = 0;
  uint64_t np = 0;
  int k = 0;
  for (int t = 0; t < DF_K; t++) {
    const uint32_t c = codes[p + t];
    if (c > 3) { np |= (uint64_t)t << (4 * k); k++; }
    else x |= c << (2 * t);
  }
  *idx = x; *npos = np;
  return k;
}
// spelling number x (0 .. 4^k - 1) of such a 10-mer
MIA_HD inline uint32_t kmer_wild_key(uint32_t idx, uint64_t npos, int k, uint32_t x) {
  for (int j = 0; j < k; j++) idx |= ((x >> (2 * j)) & 3u) << (2 * (int)((npos >> (4 * j)) & 15u));
  return idx;
}
// one word of the three planes from 64 consecutive reference codes (0..3 bases, anything else N)
MIA_HD inline void plane_word(const uint8_t* codes, int64_t n_codes, int64_t word, uint64_t* lo, uint64_t* hi, uint64_t* ok) {
  uint64_t l = 0, h = 0, k = 0;
  for (int b = 0; b < 64; b++) {
    const int64_t p = word * 64 + b - PLANE_LEAD;
    if (p < 0 || p >= n_codes) continue;
    const uint32_t c = codes[p];
    if (c > 3) continue;
    l |= (uint64_t)(c & 1) << b;
    h |= (uint64_t)(c >> 1) << b;
    k |= 1ull << b;
  }
  *lo = l; *hi = h; *ok = k;
}

MIA_HD inline int df_popc(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(x);
#else
  return __builtin_popcountll(x);
#endif
}
MIA_HD inline int df_ctz(uint64_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
  return __ffsll((long long)x) - 1;
#else
  return __builtin_ctzll(x);
#endif
}
MIA_HD inline int df_clz(uint64_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)x);
#else
  return __builtin_clzll(x);
#endif
}

MIA_HD inline int df_popc32(uint32_t x) { return df_popc((uint64_t)x); }
MIA_HD inline int df_ctz32(uint32_t x) { return df_ctz((uint64_t)x); }            // x != 0
MIA_HD inline int df_clz32(uint32_t x) { return df_clz((uint64_t)x) - 32; }      // x != 0

// The read against every diagonal of a range: planes of the window slide past the read one bit per step.  NW = 64-bit
// words per read plane, a template parameter so that every array index is static (registers, not scratch).
template <int NW>
struct DiagScan {
  uint64_t rlo[NW], rhi[NW], rows[NW];         // read planes, valid-row mask
  uint64_t clo[NW + 1], chi[NW + 1], cok[NW + 1];
  int len2, left;                              // bits left in the top word before a refill

  // false if the read holds an N (such reads are left to the DP)
  MIA_HD bool load_read(const uint8_t* packed, int n) {
    len2 = n;
    bool acgt = true;
    const uint32_t* pw = (const uint32_t*)packed;        // reads start on 4-byte boundaries, 8 codes per word
#pragma unroll
    for (int j = 0; j < NW; j++) {
      uint64_t l = 0, h = 0, m = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int r0 = j * 64 + q * 8;
        const int cnt = n - r0 < 8 ? n - r0 : 8;
        if (cnt <= 0) continue;
        uint32_t x = pw[r0 >> 3];
        if (cnt < 8) x &= (1u << (4 * cnt)) - 1u;
        if (x & 0xCCCCCCCCu) acgt = false;               // a code above 3
        // gather bit 0 (bit 1) of each nibble into one byte
        uint32_t a = x & 0x11111111u, b = (x >> 1) & 0x11111111u;
        a = (a | (a >> 3)) & 0x03030303u; a = (a | (a >> 6)) & 0x000F000Fu; a = (a | (a >> 12)) & 0xFFu;
        b = (b | (b >> 3)) & 0x03030303u; b = (b | (b >> 6)) & 0x000F000Fu; b = (b | (b >> 12)) & 0xFFu;
        l |= (uint64_t)a << (q * 8);
        h |= (uint64_t)b << (q * 8);
        m |= (uint64_t)((1u << cnt) - 1u) << (q * 8);
      }
      rlo[j] = l; rhi[j] = h; rows[j] = m;
    }
    return acgt;
  }

  // the same from planes made earlier (k_read_planes: NWP words of lo, then NWP of hi); the caller knows there is no N
  MIA_HD void set_read(const uint64_t* lo, const uint64_t* hi, int n) {
    len2 = n;
#pragma unroll
    for (int j = 0; j < NW; j++) {
      const int left = n - 64 * j;
      rlo[j] = lo[j]; rhi[j] = hi[j];
      rows[j] = left >= 64 ? ~0ull : (left > 0 ? (1ull << left) - 1ull : 0ull);
    }
  }

  // planes of reference positions [pos, pos + 64 (NW+1)); pos >= -PLANE_LEAD
  MIA_HD void seek(const RefPlanes& rp, int64_t pos) {
    const int64_t bit = pos + PLANE_LEAD;
    const int64_t q = bit >> 6;
    const int b = (int)(bit & 63);
    uint64_t l0 = rp.lo[q], h0 = rp.hi[q], k0 = rp.ok[q];
#pragma unroll
    for (int j = 0; j <= NW; j++) {
      const uint64_t l1 = rp.lo[q + j + 1], h1 = rp.hi[q + j + 1], k1 = rp.ok[q + j + 1];
      // (x << 1) << (63 - b) == x << (64 - b) for b in 1..63 and 0 for b == 0, without a 64-bit shift by 64
      clo[j] = (l0 >> b) | ((l1 << 1) << (63 - b));
      chi[j] = (h0 >> b) | ((h1 << 1) << (63 - b));
      cok[j] = (k0 >> b) | ((k1 << 1) << (63 - b));
      l0 = l1; h0 = h1; k0 = k1;
    }
    left = 64;
  }
  // one reference position further
  MIA_HD void advance(const RefPlanes& rp, int64_t next_pos) {
    if (--left == 0) { seek(rp, next_pos); return; }
#pragma unroll
    for (int j = 0; j < NW; j++) {
      clo[j] = (clo[j] >> 1) | (clo[j + 1] << 63);
      chi[j] = (chi[j] >> 1) | (chi[j + 1] << 63);
      cok[j] = (cok[j] >> 1) | (cok[j + 1] << 63);
    }
    clo[NW] >>= 1; chi[NW] >>= 1; cok[NW] >>= 1;
  }
  // definite mismatches of the rows in word j on the current diagonal
  MIA_HD uint64_t mis(int j) const { return ((rlo[j] ^ clo[j]) | (rhi[j] ^ chi[j])) & cok[j] & rows[j]; }
  MIA_HD int mismatches() const {
    int n = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) n += df_popc(mis(j));
    return n;
  }
  MIA_HD bool all_acgt() const {   // no N (and nothing outside the reference) under the read
    bool full = true;
#pragma unroll
    for (int j = 0; j < NW; j++) full = full && ((cok[j] & rows[j]) == rows[j]);
    return full;
  }
  MIA_HD int clean_prefix() const {   // rows 0 .. without a definite mismatch
    int p = len2;
#pragma unroll
    for (int j = NW - 1; j >= 0; j--) {
      const uint64_t m = mis(j);
      if (m) p = j * 64 + df_ctz(m);
    }
    return p;
  }
  MIA_HD int clean_suffix() const {   // rows .. R without a definite mismatch
    int q = len2;
#pragma unroll
    for (int j = 0; j < NW; j++) {
      const uint64_t m = mis(j);
      if (m) q = len2 - 1 - (j * 64 + 63 - df_clz(m));
    }
    return q;
  }
};

// 64 reference positions of the three planes, sliding one position per step (a second word feeds the first)
struct Slide {
  uint64_t lo0, lo1, hi0, hi1, ok0, ok1;
  int left;
  MIA_HD void seek(const RefPlanes& rp, int64_t pos) {
    const int64_t bit = pos + PLANE_LEAD;
    const int64_t q = bit >> 6;
    const int b = (int)(bit & 63);
    const uint64_t l0 = rp.lo[q], l1 = rp.lo[q + 1], l2 = rp.lo[q + 2];
    const uint64_t h0 = rp.hi[q], h1 = rp.hi[q + 1], h2 = rp.hi[q + 2];
    const uint64_t k0 = rp.ok[q], k1 = rp.ok[q + 1], k2 = rp.ok[q + 2];
    lo0 = (l0 >> b) | ((l1 << 1) << (63 - b)); lo1 = (l1 >> b) | ((l2 << 1) << (63 - b));
    hi0 = (h0 >> b) | ((h1 << 1) << (63 - b)); hi1 = (h1 >> b) | ((h2 << 1) << (63 - b));
    ok0 = (k0 >> b) | ((k1 << 1) << (63 - b)); ok1 = (k1 >> b) | ((k2 << 1) << (63 - b));
    left = 64;
  }
  MIA_HD void advance(const RefPlanes& rp, int64_t next_pos) {
    if (--left == 0) { seek(rp, next_pos); return; }
    lo0 = (lo0 >> 1) | (lo1 << 63); lo1 >>= 1;
    hi0 = (hi0 >> 1) | (hi1 << 63); hi1 >>= 1;
    ok0 = (ok0 >> 1) | (ok1 << 63); ok1 >>= 1;
  }
  MIA_HD uint64_t mis(uint64_t rlo, uint64_t rhi, uint64_t rows) const { return ((rlo ^ lo0) | (rhi ^ hi0)) & ok0 & rows; }
};

// Step 1, rules (a) and (b): the best diagonal among those that hold the whole read.  Returns its mismatch count K
// (0..2) if it is the only diagonal that good and has no N under it, -1 if the read has to go to the DP.  K <= 1 settles
// the read; K == 2 needs step 2.  Only the first 64 rows are compared on every diagonal: three mismatches there already
// put a diagonal out of the race (all that is asked of the others is ">= K+1", K <= 2), the few diagonals that pass
// are compared in full.  *best_out: a lower bound of the mismatches on the best diagonal, exact when it is below 3.
//
// RIGHT_TIES: diagonals to the RIGHT of the best one may be equally good (pass 1 on a circular reference sees every
// read near the origin twice, L columns apart).  Such a diagonal ends in a later column of the last row with the same
// value, and max_sg_score keeps the FIRST maximum; nothing else in the argument needs strictness on that side.  The
// scan runs left to right and replaces the best only by a strictly better diagonal, so everything left of the final
// best is strictly worse by construction.
template <int NW, bool RIGHT_TIES>
MIA_HD inline int diag_step1(const RefPlanes& rp, int s, int len1, const uint8_t* read_packed, int len2, int* delta_out, int* best_out) {
  DiagScan<NW> sc;
  *best_out = -1;
  if (!sc.load_read(read_packed, len2)) return -1;
  const int fit = len1 - len2;                        // diagonals 0 .. fit hold the whole read
  int best = 1 << 20, second = 1 << 20, delta = -1;
  bool clean = false;
  Slide head;
  head.seek(rp, s);
  for (int d = 0; d <= fit; d++) {
    int m = df_popc(head.mis(sc.rlo[0], sc.rhi[0], sc.rows[0]));
    bool acgt = (head.ok0 & sc.rows[0]) == sc.rows[0];
    if (NW > 1 && m < 3) {                            // a contender: all rows
      sc.seek(rp, (int64_t)s + d);
      m = sc.mismatches();
      acgt = sc.all_acgt();
    }
    if (m < best) { second = best; best = m; delta = d; clean = acgt; }
    else if (m < second) second = m;
    if (d < fit) head.advance(rp, (int64_t)s + d + 1);
  }
  *best_out = best;                                   // a hint for the planner: many mismatches on the best diagonal = a gap
  if (best > 2 || (!RIGHT_TIES && second <= best) || !clean) return -1;
  *delta_out = delta;
  return best;
}

// Step 2, rule (c), for K == 2: prefixes start in row 0 at a window column (diagonals 0 .. len1-1), suffixes end in row
// R at a window column (diagonals -R .. len1-len2).  A prefix is almost always cut within the first 64 rows and a suffix
// within the last word of rows: two sliding words; the full planes are fetched only when one of them shows no mismatch.
template <int NW>
MIA_HD inline bool diag_step2(const RefPlanes& rp, int s, int len1, const uint8_t* read_packed, int len2) {
  DiagScan<NW> sc;
  sc.load_read(read_packed, len2);
  const int R = len2 - 1, fit = len1 - len2;
  constexpr int TOP = 64 * (NW - 1);                  // first row of the last word
  int lp = 0, ls = 0;
  Slide head, tail;
  head.seek(rp, (int64_t)s - R);
  if (NW > 1) tail.seek(rp, (int64_t)s - R + TOP);
  for (int d = -R; d <= len1 - 1; d++) {
    const uint64_t m0 = head.mis(sc.rlo[0], sc.rhi[0], sc.rows[0]);
    const uint64_t mt = NW > 1 ? tail.mis(sc.rlo[NW - 1], sc.rhi[NW - 1], sc.rows[NW - 1]) : m0;
    int p, q;
    if (NW > 1 && (m0 == 0 || mt == 0)) {             // a long clean stretch: all rows
      sc.seek(rp, (int64_t)s + d);
      p = sc.clean_prefix();
      q = sc.clean_suffix();
    } else {
      p = m0 ? df_ctz(m0) : len2;
      q = mt ? len2 - 1 - (TOP + 63 - df_clz(mt)) : len2;
    }
    // a prefix on diagonal d cannot run past the last window column, a suffix cannot begin before the first one
    if (p > len1 - d) p = len1 - d;
    if (d < 0 && q > len2 + d) q = len2 + d;
    if (d >= 0 && p > lp) lp = p;
    if (d <= fit && q > ls) ls = q;
    if (d < len1 - 1) {
      head.advance(rp, (int64_t)s + d + 1);
      if (NW > 1) tail.advance(rp, (int64_t)s + d + 1 + TOP);
    }
  }
  return lp + ls <= R - 3;
}

// are reference positions [lo, hi) all A/C/G/T?  (positions outside the reference count as "not": their ok bits are 0)
MIA_HD inline bool all_bases(const RefPlanes& rp, int64_t lo, int64_t hi) {
  for (int64_t b = lo + PLANE_LEAD; b < hi + PLANE_LEAD;) {
    const int64_t w = b >> 6;
    const int o = (int)(b & 63);
    const int64_t room = 64 - o, left = hi + PLANE_LEAD - b;
    const int n = (int)(left < room ? left : room);
    const uint64_t mask = (n == 64 ? ~0ull : ((1ull << n) - 1ull)) << o;
    if ((rp.ok[w] & mask) != mask) return false;
    b += n;
  }
  return true;
}

// Rule (c) through the 10-mer table: 1 = proven (LP + LS <= R - 3 with every diagonal the table does not name counted
// as 9); 0 = not proven -- the read then simply goes to the DP (sliding over all diagonals would settle a few more of
// these, but one such read per wavefront would cost the whole wavefront the slide); -1 = the table cannot be used here
// (N in reach, a 10-mer with too many occurrences, a short read, no table): the caller slides.  n_ref = reference positions.
template <int NW>
MIA_HD inline int diag_step2_kmer(const RefPlanes& rp, const KmerOcc& ko, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2) {
  if (!ko.cnt || len2 < 2 * DF_K + 4) return -1;
  const int R = len2 - 1, fit = len1 - len2;
  // every position a prefix or suffix inside the window can touch must be a plain base
  int64_t lo = (int64_t)s - R, hi = (int64_t)s + len1 + R;
  if (lo < 0) lo = 0;
  if (hi > n_ref) hi = n_ref;
  if (!all_bases(rp, lo, hi)) return -1;
  DiagScan<NW> sc;
  sc.load_read(read_packed, len2);
  // the read's first and last 10-mer
  int64_t first = 0, last = 0;
  for (int t = 0; t < DF_K; t++) {
    first |= (int64_t)(((sc.rlo[0] >> t) & 1ull) | (((sc.rhi[0] >> t) & 1ull) << 1)) << (2 * t);
    const int r = R - DF_K + 1 + t;
    const int c = (read_packed[r >> 1] >> ((r & 1) * 4)) & 3;
    last |= (int64_t)c << (2 * t);
  }
  const int nf = ko.cnt[first], nl = ko.cnt[last];
  if (nf > DF_KCAP || nl > DF_KCAP) return -1;
  int lp = DF_K - 1, ls = DF_K - 1;
  for (int k = 0; k < DF_KCAP; k++) {
    if (k < nf) {
      const int d = ko.pos[first * DF_KCAP + k] - s;            // the prefix starts in window column d
      if (d >= 0 && d <= len1 - 1) {
        sc.seek(rp, (int64_t)s + d);
        int p = sc.clean_prefix();
        if (p > len1 - d) p = len1 - d;
        if (p > lp) lp = p;
      }
    }
    if (k < nl) {
      const int d = ko.pos[last * DF_KCAP + k] + DF_K - 1 - R - s;   // row R sits in window column d + R
      if (d >= -R && d <= fit) {
        sc.seek(rp, (int64_t)s + d);
        int q = sc.clean_suffix();
        if (d < 0 && q > len2 + d) q = len2 + d;
        if (q > ls) ls = q;
      }
    }
  }
  return lp + ls <= R - 3 ? 1 : 0;
}
MIA_HD inline int diag_step2_kmer(const RefPlanes& rp, const KmerOcc& ko, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2) {
  switch ((len2 + 63) >> 6) {
    case 1: return diag_step2_kmer<1>(rp, ko, n_ref, s, len1, read_packed, len2);
    case 2: return diag_step2_kmer<2>(rp, ko, n_ref, s, len1, read_packed, len2);
    case 3: return diag_step2_kmer<3>(rp, ko, n_ref, s, len1, read_packed, len2);
    default: return diag_step2_kmer<4>(rp, ko, n_ref, s, len1, read_packed, len2);
  }
}

// Step 1 through the 10-mer table.  Pigeonhole: cut B disjoint 10-mers out of the read; a diagonal with fewer than B
// mismatches leaves at least one of them intact, i.e. that 10-mer occurs in the reference at the matching place.  So the
// occurrences the table lists for the B blocks name every diagonal that can have fewer than B mismatches; those few are
// compared in full, all others have >= B.  B = 6 where the read is long enough (the gap hint wants "best >= 6"), at
// least 3 (rules (a), (b) ask for ">= K+1", K <= 2).  Returns K / -1 like diag_step1, or -2 if the table cannot be used
// for this read (N in the window, a block whose 10-mer has more occurrences than the table keeps, a short read, no
// table): the caller slides.  n_ref = reference positions; the window is [s, s + len1).
constexpr int DF_BLOCKS = 6;
template <int NW, bool RIGHT_TIES>
MIA_HD inline int diag_step1_kmer(const RefPlanes& rp, const KmerOcc& ko, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2,
                                  int* delta_out, int* best_out) {
  *best_out = -1;
  const int nblk = len2 / DF_K < DF_BLOCKS ? len2 / DF_K : DF_BLOCKS;
  if (!ko.cnt || nblk < 3 || s < 0 || (int64_t)s + len1 > n_ref) return -2;
  if (!all_bases(rp, s, (int64_t)s + len1)) return -2;
  DiagScan<NW> sc;
  if (!sc.load_read(read_packed, len2)) return -1;
  const int fit = len1 - len2;
  int best = 1 << 20, second = nblk, delta = -1;        // diagonals the table does not name: >= nblk mismatches
  for (int b = 0; b < DF_BLOCKS; b++) {
    if (b >= nblk) break;
    const int o = (int)((int64_t)b * (len2 - DF_K) / (nblk - 1));       // first row of block b; blocks do not overlap (len2 >= 10 nblk)
    int64_t idx = 0;
    for (int t = 0; t < DF_K; t++) {
      const int r = o + t;
      idx |= (int64_t)((read_packed[r >> 1] >> ((r & 1) * 4)) & 3) << (2 * t);
    }
    const int n = ko.cnt[idx];
    if (n > DF_KCAP) return -2;
    for (int k = 0; k < DF_KCAP; k++) {
      if (k >= n) break;
      const int d = ko.pos[idx * DF_KCAP + k] - o - s;
      if (d < 0 || d > fit || d == delta) continue;
      sc.seek(rp, (int64_t)s + d);
      const int m = sc.mismatches();
      if (m < best || (m == best && d < delta)) { if (best < second) second = best; best = m; delta = d; }
      else if (m < second) second = m;
    }
  }
  *best_out = best < nblk ? best : nblk;               // a lower bound of the fewest mismatches on any diagonal, exact below nblk
  if (best > 2 || (!RIGHT_TIES && second <= best)) return -1;
  *delta_out = delta;
  return best;
}

MIA_HD inline bool diag_examined(int len1, int len2) { return len2 >= 1 && len2 <= MAX_READ && len1 >= len2 && len1 <= DF_MAX_LEN1; }

// the table first, the slide where the table cannot be used
MIA_HD inline int diag_step1(const RefPlanes& rp, const KmerOcc& ko, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2,
                             int* delta_out, int* best_out);
MIA_HD inline int diag_step1(const RefPlanes& rp, int s, int len1, const uint8_t* read_packed, int len2, int* delta_out, int* best_out) {
  *best_out = -1;
  if (!diag_examined(len1, len2)) return -1;
  switch ((len2 + 63) >> 6) {
    case 1: return diag_step1<1, false>(rp, s, len1, read_packed, len2, delta_out, best_out);
    case 2: return diag_step1<2, false>(rp, s, len1, read_packed, len2, delta_out, best_out);
    case 3: return diag_step1<3, false>(rp, s, len1, read_packed, len2, delta_out, best_out);
    default: return diag_step1<4, false>(rp, s, len1, read_packed, len2, delta_out, best_out);
  }
}
MIA_HD inline int diag_step1(const RefPlanes& rp, const KmerOcc& ko, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2,
                             int* delta_out, int* best_out) {
  *best_out = -1;
  if (!diag_examined(len1, len2)) return -1;
  int k;
  switch ((len2 + 63) >> 6) {
    case 1: k = diag_step1_kmer<1, false>(rp, ko, n_ref, s, len1, read_packed, len2, delta_out, best_out); break;
    case 2: k = diag_step1_kmer<2, false>(rp, ko, n_ref, s, len1, read_packed, len2, delta_out, best_out); break;
    case 3: k = diag_step1_kmer<3, false>(rp, ko, n_ref, s, len1, read_packed, len2, delta_out, best_out); break;
    default: k = diag_step1_kmer<4, false>(rp, ko, n_ref, s, len1, read_packed, len2, delta_out, best_out); break;
  }
  return k != -2 ? k : diag_step1(rp, s, len1, read_packed, len2, delta_out, best_out);
}
// the same against a whole strand of the (wrapped) reference, columns 0 .. len1-1: pass 1
MIA_HD inline int strand_step1(const RefPlanes& rp, const KmerOcc& ko, int len1, const uint8_t* read_packed, int len2, int* delta_out, int* best_out) {
  *best_out = -1;
  if (len2 < 1 || len2 > MAX_READ || len1 < len2) return -1;
  int k;
  switch ((len2 + 63) >> 6) {
    case 1: k = diag_step1_kmer<1, true>(rp, ko, len1, 0, len1, read_packed, len2, delta_out, best_out); break;
    case 2: k = diag_step1_kmer<2, true>(rp, ko, len1, 0, len1, read_packed, len2, delta_out, best_out); break;
    case 3: k = diag_step1_kmer<3, true>(rp, ko, len1, 0, len1, read_packed, len2, delta_out, best_out); break;
    default: k = diag_step1_kmer<4, true>(rp, ko, len1, 0, len1, read_packed, len2, delta_out, best_out); break;
  }
  if (k != -2) return k;
  switch ((len2 + 63) >> 6) {
    case 1: return diag_step1<1, true>(rp, 0, len1, read_packed, len2, delta_out, best_out);
    case 2: return diag_step1<2, true>(rp, 0, len1, read_packed, len2, delta_out, best_out);
    case 3: return diag_step1<3, true>(rp, 0, len1, read_packed, len2, delta_out, best_out);
    default: return diag_step1<4, true>(rp, 0, len1, read_packed, len2, delta_out, best_out);
  }
}

// Pass 1 (sg_align, src/mia.c:1500-1665: both strands of the whole reference, the better one wins, the reverse strand
// on a tie, :1549) for a read that the flat matrix aligns gap-free.  fw / rc: planes of the two strands, len1 columns
// each.  Step 1 of both strands: *strand = the strand whose best diagonal has fewer mismatches, K of them (returned;
// -1 = leave the read to the DP).  The loser's best diagonal must have >= K+1 mismatches: every path there then loses
// more than 800 K (a diagonal >= 800 (K+1); anything with an event >= 1200, and for K == 2 pass1_step2 on the loser
// excludes the event paths that lose <= 1600), so the winner's score 200 len - 800 K is strictly the larger one.
MIA_HD inline int pass1_step1(const RefPlanes& fw, const RefPlanes& rc, const KmerOcc& kf, const KmerOcc& kr, int len1, const uint8_t* read_packed,
                              int len2, int* strand, int* delta) {
  int d[2] = {0, 0}, best[2], k[2];
  k[0] = strand_step1(fw, kf, len1, read_packed, len2, &d[0], &best[0]);
  if (best[0] < 0) return -1;                            // a read with N
  k[1] = strand_step1(rc, kr, len1, read_packed, len2, &d[1], &best[1]);
  const int x = best[0] < best[1] ? 0 : 1, y = 1 - x;     // best[] below 3 is exact, from 3 on a lower bound
  if (k[x] < 0 || best[y] <= k[x]) return -1;
  *strand = x;
  *delta = d[x];
  return k[x];
}
MIA_HD inline bool diag_step2(const RefPlanes& rp, int s, int len1, const uint8_t* read_packed, int len2);
MIA_HD inline int diag_step2_kmer(const RefPlanes& rp, const KmerOcc& ko, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2);
// K == 2: rule (c) on both strands, each through its 10-mer table where that can be used
MIA_HD inline bool pass1_step2(const RefPlanes& fw, const RefPlanes& rc, const KmerOcc& kf, const KmerOcc& kr, int len1, const uint8_t* read_packed,
                               int len2) {
  const int a = diag_step2_kmer(fw, kf, len1, 0, len1, read_packed, len2);
  if (a == 0 || (a < 0 && !diag_step2(fw, 0, len1, read_packed, len2))) return false;
  const int b = diag_step2_kmer(rc, kr, len1, 0, len1, read_packed, len2);
  return b > 0 || (b < 0 && diag_step2(rc, 0, len1, read_packed, len2));
}
MIA_HD inline bool diag_step2(const RefPlanes& rp, int s, int len1, const uint8_t* read_packed, int len2) {
  switch ((len2 + 63) >> 6) {
    case 1: return diag_step2<1>(rp, s, len1, read_packed, len2);
    case 2: return diag_step2<2>(rp, s, len1, read_packed, len2);
    case 3: return diag_step2<3>(rp, s, len1, read_packed, len2);
    default: return diag_step2<4>(rp, s, len1, read_packed, len2);
  }
}

struct DiagVerdict { int delta, mismatches; };

// Window = reference positions [s, s + len1).  True iff the alignment is provably the diagonal out->delta (offset
// inside the window) with out->mismatches definite mismatches.  The caller has established the flat matrix.
MIA_HD inline bool diag_filter(const RefPlanes& rp, int s, int len1, const uint8_t* read_packed, int len2, DiagVerdict* out) {
  int delta = 0, best = 0;
  const int k = diag_step1(rp, s, len1, read_packed, len2, &delta, &best);
  if (k < 0 || (k == 2 && !diag_step2(rp, s, len1, read_packed, len2))) return false;
  out->delta = delta;
  out->mismatches = k;
  return true;
}
// the same with the reference's 10-mer table at hand (what k_diag_filter does)
MIA_HD inline bool diag_filter(const RefPlanes& rp, const KmerOcc& ko, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2,
                               DiagVerdict* out) {
  int delta = 0, best = 0;
  const int k = diag_step1(rp, ko, n_ref, s, len1, read_packed, len2, &delta, &best);
  if (k < 0) return false;
  if (k == 2) {
    const int via_table = diag_step2_kmer(rp, ko, n_ref, s, len1, read_packed, len2);
    if (via_table == 0 || (via_table < 0 && !diag_step2(rp, s, len1, read_packed, len2))) return false;
  }
  out->delta = delta;
  out->mismatches = k;
  return true;
}

// the flat matrix, both strands, every depth; N columns may score anything up to a match (src/pssm.c:96-126)
inline bool pssm_is_flat(const int32_t* fwd, const int32_t* rc) {
  const int32_t* tabs[2] = {fwd, rc};
  for (const int32_t* t : tabs)
    for (int d = 0; d < 2 * PSSM_DEPTH + 1; d++)
      for (int a = 0; a < 5; a++)
        for (int b = 0; b < 4; b++) {
          const int32_t v = t[(d * 5 + a) * 5 + b];
          if (a < 4 ? v != (a == b ? FLAT_MATCH : FLAT_MISMATCH) : v > FLAT_MATCH) return false;
        }
  return true;
}

}  // namespace mia
