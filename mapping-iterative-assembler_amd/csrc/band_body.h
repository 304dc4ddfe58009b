// band_body.h -- the windowed DP of dyn_prog / max_sg_score / find_align_begin / populate_pwaln_to_begin
// (/root/reference/src/mia.c:740-981,1278-1302,612-637,1440-1497) confined to a BAND of diagonals that provably holds
// every alignment that matters, one read per thread.  Flat matrix, read and window free of N (diag_filter.h's premises).
//
// Which band.  Cut the read into nb = min(9, len / 10) >= 3 disjoint 10-mers (diag_filter.h: pigeonhole; a substitution
// costs 800 and breaks one of them, a gap of n columns 1000 + 200 n and breaks at most one, n inserted or clipped rows at
// least 1000 + 400 n and touch at most n / 10 + 2: never less than 800 per 10-mer broken).  Some valid path is cheap to write down -- the
// read on the diagonal of its first anchor up to a switch row, one gap, the rest on the diagonal of its last anchor -- and
// its loss B0 (against 200 x len) bounds the optimum's.  If B0 <= 800 nb - 400, every path that loses no more than B0 runs
// through an anchor and strays at most g = (B0 - 1000) / 200 diagonals from it (0 if B0 < 1200), so all of them, ties
// included, lie on diagonals [lowest anchor - g, highest anchor + g]: W of them, not the window's hundred.
//
// Why the band is enough.  Run the recurrence with every cell outside the band "absent".  A cell ON an optimal path has the
// same value as in the full matrix (its best path is an optimal prefix, inside the band), and the candidates that lose
// against the chosen one in the full matrix are not larger here; a candidate that ties is itself optimal, hence inside
// with its full value, and the running maxima keep the earliest index among equals in both.  So values, choices and gap
// sources agree on every cell the traceback visits, and the first maximum of the last row is the same cell.  The one
// thing that can lead the reference's traceback OFF the optimal cells is its own quirk -- a gap whose source index is 0
// reads back as a diagonal step (src/mia.c:619) -- so a traceback that meets such a gap gives up and the read goes to the
// full-window kernels, as does any read whose band would be wider than BAND_W.
#pragma once
#include <stdint.h>

#include "diag_filter.h"
#include "mia_layout.h"

namespace mia {

constexpr int BAND_W = 32;            // diagonals a thread keeps in registers
constexpr int BAND_BLOCKS = 9;        // 10-mers cut out of the read: len / 10 of them, at most 9
constexpr int BAND_MIN_BLOCKS = 3;    // ... and at least 3 (budget 2000: two substitutions, or one short indel and one substitution)
constexpr int BAND_NEG = -(1 << 22);  // "no such cell": far below any score a read of 256 bases can have, small enough to pack

struct BandPlan { int d0, w, budget, b0; };   // diagonals d0 .. d0 + w - 1 (column minus row, window coordinates)

// 10-mer index (diag_filter.h: kmer_at's packing) of read rows o .. o+9, from the packed nibbles
MIA_HD inline int64_t band_kmer(const uint32_t* pw, int len2, int o) {
  const int w = o >> 3, last = (len2 - 1) >> 3;
  const uint64_t w0 = pw[w], w1 = w + 1 <= last ? pw[w + 1] : 0u, w2 = w + 2 <= last ? pw[w + 2] : 0u;
  const int sh = 4 * (o & 7);
  uint64_t y = ((w0 | (w1 << 32)) >> sh) | (sh ? (w2 << 32) << (32 - sh) : 0ull);          // 40 bits: ten nibbles
  y &= 0x3333333333ull;
  y = (y | (y >> 2)) & 0x0F0F0F0F0Full;
  y = (y | (y >> 4)) & 0x00FF00FF00FFull;
  y = (y | (y >> 8)) & 0x0000FFFF0000FFFFull;
  return (int64_t)((y & 0xFFFFull) | ((y >> 16) & 0xF0000ull));
}

// the band of a read, or false if the read is left to the full-window kernels
MIA_HD inline bool band_plan(const RefPlanes& rp, const KmerOcc& ko, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2, BandPlan* out) {
  if (!ko.cnt || len2 < BAND_MIN_BLOCKS * DF_K || len2 > MAX_READ || len1 < len2 || len1 > DF_MAX_LEN1 || s < 0 || (int64_t)s + len1 > n_ref) return false;
  if (!all_bases(rp, s, (int64_t)s + len1)) return false;
  DiagScan<4> sc;
  if (!sc.load_read(read_packed, len2)) return false;             // a read with N
  const int R = len2 - 1, nb_cut = len2 / DF_K < BAND_BLOCKS ? len2 / DF_K : BAND_BLOCKS;
  const uint32_t* pw = reinterpret_cast<const uint32_t*>(read_packed);
  // all look-ups first, then their use: nine independent loads in flight instead of nine round trips
  int32_t cn[BAND_BLOCKS], ps[BAND_BLOCKS][DF_KCAP];
#pragma unroll
  for (int b = 0; b < BAND_BLOCKS; b++) {
    cn[b] = DF_KCAP + 1;
    if (b < nb_cut) {
      const int64_t idx = band_kmer(pw, len2, (int)((int64_t)b * (len2 - DF_K) / (nb_cut - 1)));
      cn[b] = ko.cnt[idx];
#pragma unroll
      for (int k = 0; k < DF_KCAP; k++) ps[b][k] = ko.pos[idx * DF_KCAP + k];
    }
  }
  int nb = 0, a_lo = 1 << 20, a_hi = -(1 << 20), d_first = 0, d_last = 0;
  bool any = false;
#pragma unroll
  for (int b = 0; b < BAND_BLOCKS; b++) {
    if (cn[b] > DF_KCAP) continue;                   // no such block, or an overloaded 10-mer: not part of the pigeonhole
    nb++;
    const int o = (int)((int64_t)b * (len2 - DF_K) / (nb_cut - 1));
#pragma unroll
    for (int k = 0; k < DF_KCAP; k++) {
      if (k >= cn[b]) continue;
      const int d = ps[b][k] - o - s;                         // diagonal in window coordinates
      if (d < -R || d > len1 - 1) continue;                   // not a place inside this window
      if (!any) { d_first = d; any = true; }
      d_last = d;
      if (d < a_lo) a_lo = d;
      if (d > a_hi) a_hi = d;
    }
  }
  if (nb < BAND_MIN_BLOCKS || !any || a_hi - a_lo >= BAND_W) return false;
  // the loss of one valid path: rows [0, t) on d_first, one gap, the rest on d_last (or the plain diagonal if they agree)
  if (d_first < 0 || d_first > len1 - len2 || d_last < 0 || d_last > len1 - len2) return false;   // keep the written-down path inside the window
  int b0;
  {
    uint64_t m1[4], m2[5];                           // mismatch masks of the read on the two diagonals
    sc.seek(rp, (int64_t)s + d_first);
#pragma unroll
    for (int j = 0; j < 4; j++) m1[j] = sc.mis(j);
    if (d_first == d_last) {
      b0 = 800 * (df_popc(m1[0]) + df_popc(m1[1]) + df_popc(m1[2]) + df_popc(m1[3]));
    } else {
      sc.seek(rp, (int64_t)s + d_last);
#pragma unroll
      for (int j = 0; j < 4; j++) m2[j] = sc.mis(j);
      m2[4] = 0;
      // column gap (d_last > d_first: the rows from t on continue d_last - d_first columns further right) or row gap (the read
      // skips d_first - d_last rows): prefix mismatches on d_first plus suffix mismatches on d_last, best switch row t
      const int shift = d_last - d_first, skip = shift < 0 ? -shift : 0;            // |shift| < BAND_W
      if (skip) {                                    // bit t of m2 := its row t + skip
#pragma unroll
        for (int j = 0; j < 4; j++) m2[j] = (m2[j] >> skip) | (m2[j + 1] << (64 - skip));
      }
      int prefix = 0, suffix = df_popc(m2[0]) + df_popc(m2[1]) + df_popc(m2[2]) + df_popc(m2[3]), best = 1 << 20;
      // rows [0, t) on d_first, `skip` rows inserted, rows [t + skip, len2) on d_last; both stretches non-empty
#pragma unroll
      for (int j = 0; j < 4; j++) {
        for (int q = 0; q < 64; q++) {
          const int t = j * 64 + q + 1;
          if (t + skip > R) break;
          prefix += (int)((m1[j] >> q) & 1);
          suffix -= (int)((m2[j] >> q) & 1);
          if (prefix + suffix < best) best = prefix + suffix;
        }
      }
      if (best == (1 << 20)) return false;
      b0 = 800 * best + (shift > 0 ? GOP + GEP * shift : GOP + GEP * skip + FLAT_MATCH * skip);
    }
  }
  const int budget = 800 * nb - 400;
  if (b0 > budget) return false;
  // One diagonal more only where the window's first column is within reach: a path that starts late THERE gets its
  // substitution score back (src/mia.c:838-846) and may spend those 200 on one more gap column; it starts on a
  // negative diagonal, so with a_lo - g - 1 >= 0 no such path can come near an anchor.
  int g = b0 < GOP + GEP ? 0 : (b0 - GOP) / GEP;
  if (a_lo - g - 1 < 0) g++;
  const int d0 = a_lo - g, w = a_hi - a_lo + 2 * g + 1;
  if (w > BAND_W) return false;
  out->d0 = d0; out->w = w; out->budget = budget; out->b0 = b0;
  return true;
}

struct BandResult { int score, abc, aec, abr, gaps; uint32_t gap_desc; };   // gap_desc: the last gap met by the traceback, as ST_ONEGAP's upper bits >> 8

// 32 bits of a plane from bit position `bit` on
MIA_HD inline uint64_t band_bits(const uint64_t* plane, int64_t bit) {
  const int64_t q = bit >> 6;
  const int o = (int)(bit & 63);
  return o ? (plane[q] >> o) | (plane[q + 1] << (64 - o)) : plane[q];
}

// true if no cell of the band [d0, d0 + wmax) leaves the window in any row and none but row 0's touches column 0: the
// plain form of band_align applies
MIA_HD inline bool band_interior(const BandPlan& bp, int wmax, int len1, int len2) { return bp.d0 >= 0 && len2 - 1 + bp.d0 + wmax <= len1; }

// The DP over the band [d0, d0 + wmax), wmax >= the plan's width and a multiple of 4 (a wavefront uses the widest of its
// 64 reads: a scalar loop bound).  trace: 8 words (BAND_W bytes) per row, row r at trace + r * row_words -- private to the thread (the
// kernel interleaves the rows of a wavefront's 64 reads so that its stores coalesce).  cols_out: the script (window column
// per read row, COL_INSERT, COL_CLIP) as the other kernels write it.  false: the traceback met the reference's index-0
// quirk -- the caller sends the read to the full-window kernels.  EDGE: the band may leave the window (see band_interior).
//
// Everything a cell compares is packed value * 256 + code, so that one integer maximum applies the reference's tie rules:
//   diagonal   S(r-1, c-1) * 256 + 0xFF                wins ties against both gaps          (src/mia.c:848-935)
//   column gap (best - GOP - GEP n) * 256 + 0x40 + n   n = gap length; beats the row gap on ties
//   row gap    (best - GOP - GEP n) * 256 + n
// and within one kind the LARGER n among equal values, i.e. the earliest source, as the reference's running maxima keep
// the first of equals.  The winning code byte is the trace.  A new start (code 0x80) needs fresh > all three, strictly.
// byte k of word w := the low byte of v
MIA_HD inline uint32_t band_put(uint32_t w, uint32_t v, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(v, w, 0x03020100u ^ ((uint32_t)(4 ^ k) << (8 * k)));
#else
  return (w & ~(0xFFu << (8 * k))) | ((v & 0xFFu) << (8 * k));
#endif
}

template <bool EDGE>
MIA_HD inline bool band_align(const RefPlanes& rp, int s, int len1, const uint8_t* read_packed, int len2, const BandPlan& bp, int wmax, uint32_t* trace,
                              int64_t row_words, int16_t* cols_out, BandResult* res, bool no_walk = false) {
  const int R = len2 - 1, d0 = bp.d0;
  constexpr int DEAD = BAND_NEG * 256;
  constexpr int STEP = 1 - GEP * 256;                 // a running maximum ages by one position: value - GEP, length + 1
  constexpr int CAND = -GOP * 256 + STEP - 0xFF;      // a cell (packed as a diagonal source) becomes a gap source: value - GOP - GEP, length 1
  // P[j]: S(r-1, c-1) * 256 + (its trace code) for the cell on diagonal d0 + j of the current row (its diagonal predecessor).
  // H[j]: the packed row-gap candidate of that cell (best over rows <= r-2 of its left-hand column).
  // wmax is a multiple of 4 here: cells are computed in groups of four behind one scalar branch.
  int32_t P[BAND_W], H[BAND_W];
  const uint32_t* rwords = reinterpret_cast<const uint32_t*>(read_packed);     // reads start on 4-byte boundaries
  uint64_t wl, wh;
  uint32_t rw = rwords[0];
  {
    // row 0: every column may start the alignment (src/mia.c:781-800)
    const int64_t bit = (int64_t)s + d0 + PLANE_LEAD;
    wl = band_bits(rp.lo, bit); wh = band_bits(rp.hi, bit);
    const int c2 = (int)(rw & 3u);
    const uint32_t match = ~((((uint32_t)wl) ^ ((c2 & 1) ? ~0u : 0u)) | (((uint32_t)wh) ^ ((c2 & 2) ? ~0u : 0u)));
    wl >>= 1; wh >>= 1;
    const int jlo = d0 < 0 ? -d0 : 0, jhi = (len1 - d0) < wmax ? (len1 - d0) : wmax;
    const uint32_t live = jhi > jlo ? ((jhi >= 32 ? ~0u : ((1u << jhi) - 1u)) & ~((1u << jlo) - 1u)) : 0u;
#pragma unroll
    for (int j = 0; j < BAND_W; j++) {
      H[j] = DEAD;
      P[j] = ((live >> j) & 1u) ? (((match >> j) & 1u) ? FLAT_MATCH : FLAT_MISMATCH) * 256 + 0xFF : DEAD;
    }
#pragma unroll
    for (int k = 0; k < BAND_W / 4; k++) trace[k] = 0xFFFFFFFFu;
  }
  for (int r = 1; r < len2; r++) {
    const int c0 = r + d0;                        // column of j = 0
    if ((r & 31) == 0) {                          // 64 plane bits serve 32 rows (the band slides one column per row)
      const int64_t bit = (int64_t)s + c0 + PLANE_LEAD;
      wl = band_bits(rp.lo, bit); wh = band_bits(rp.hi, bit);
    }
    if ((r & 7) == 0) rw = rwords[r >> 3];
    const int c2 = (int)((rw >> (4 * (r & 7))) & 3u);
    const uint32_t match = ~((((uint32_t)wl) ^ ((c2 & 1) ? ~0u : 0u)) | (((uint32_t)wh) ^ ((c2 & 2) ? ~0u : 0u)));
    wl >>= 1; wh >>= 1;
    uint32_t live = ~0u, col0 = 0u;
    if (EDGE) {
      const int jlo = c0 < 0 ? -c0 : 0, jhi = (len1 - c0) < wmax ? (len1 - c0) : wmax;      // cells jlo .. jhi-1 are inside the window
      live = jhi > jlo ? ((jhi >= 32 ? ~0u : ((1u << jhi) - 1u)) & ~((1u << jlo) - 1u)) : 0u;
      col0 = (c0 <= 0 && c0 > -BAND_W) ? (1u << (-c0)) : 0u;                               // the cell in the window's first column
    }
    const int fresh = -(GOP + GEP * (r + 1));
    const int f0 = fresh * 256, f0s = f0 | 0x80;
    int G = DEAD;                                 // the packed column-gap candidate of the cell about to be computed
    uint32_t tw[BAND_W / 4];
#pragma unroll
    for (int k = 0; k < BAND_W / 4; k++) tw[k] = 0;
#pragma unroll
    for (int j0 = 0; j0 < BAND_W; j0 += 4) {
      if (j0 < wmax) {                            // (the same for all reads of a wavefront: a scalar branch)
#pragma unroll
        for (int j = j0; j < j0 + 4; j++) {
          const int pd = P[j] | 0xFF, h = H[j], gc = G | 0x40;
          const int x = pd > gc ? (pd > h ? pd : h) : (gc > h ? gc : h);
          const int sub256 = (int)((match >> j) & 1u) * ((FLAT_MATCH - FLAT_MISMATCH) * 256) + FLAT_MISMATCH * 256;
          // value and trace code of the cell in one word; fresh must beat all three strictly (its code byte is 0 in f0)
          int cur = f0 > x ? f0s : x + sub256;
          if (EDGE) {
            if ((col0 >> j) & 1u) cur = (sub256 + f0) | 0xFF;                          // src/mia.c:838-846
            if (!((live >> j) & 1u)) cur = DEAD;
          }
          tw[j >> 2] = band_put(tw[j >> 2], (uint32_t)cur, j & 3);
          // S(r-1, c-1) = P[j] becomes a gap source: for this row's cells further right, and for column c-1 (which the
          // next row reaches from index j-1)
          const int cand = pd + CAND;
          G = G + STEP > cand ? G + STEP : cand;
          if (j >= 1) H[j - 1] = h + STEP > cand ? h + STEP : cand;
          P[j] = cur;
        }
      } else if (j0 == wmax) {
        H[j0 - 1] = DEAD;                         // (wmax >= 4)
      }
    }
    if (wmax == BAND_W) H[BAND_W - 1] = DEAD;
    uint32_t* tr = trace + (int64_t)r * row_words;
#pragma unroll
    for (int k = 0; k < BAND_W / 4; k++) tr[k] = tw[k];
  }
  // max_sg_score: first maximum of the last row (src/mia.c:1278-1302)
  int best = BAND_NEG, bj = -1;
#pragma unroll
  for (int j = 0; j < BAND_W; j++) if (j < wmax && (P[j] >> 8) > best) { best = P[j] >> 8; bj = j; }
  if (bj < 0 || best <= BAND_NEG / 2 || no_walk) return false;
  // find_align_begin + populate_pwaln_to_begin (src/mia.c:612-637, 1440-1497).  Nearly every step is a diagonal one and
  // stays on band index j: the trace bytes of eight rows are fetched at once, so that the walk waits for memory once per
  // eight rows (and once per gap) instead of once per row.
  int r = R, c = R + d0 + bj, gaps = 0;
  uint32_t gap_desc = 0;
  const int aec = c;
  bool stop = false;
  while (!stop) {
    const int j = c - r - d0;
    if (j < 0 || j >= wmax) return false;
    uint32_t wv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) wv[k] = trace[(int64_t)(r - k > 0 ? r - k : 0) * row_words + (j >> 2)];
    bool moved = false;                      // left index j: fetch again
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (stop || moved) continue;
      cols_out[r] = (int16_t)c;
      if (r == 0 || c == 0) { stop = true; continue; }
      const int code = (int)((wv[k] >> (8 * (j & 3))) & 255u);
      if (code == 0x80) { stop = true; continue; }
      if (code == 0xFF) { r--; c--; continue; }
      moved = true;
      gaps++;
      if (code & 0x40) {
        const int sc = c - 1 - (code & 63);
        if (sc <= 0) return false;           // a gap from column 0 reads back as a diagonal step in the reference: not followed here
        gap_desc = 0u | ((uint32_t)r << 1) | ((uint32_t)(code & 63) << 10);
        r--; c = sc;
      } else {
        const int sr = r - 1 - code;
        if (sr <= 0) return false;           // a gap from row 0: the same quirk
        for (int q = r - 1; q > sr; q--) cols_out[q] = COL_INSERT;
        gap_desc = 1u | ((uint32_t)(sr + 1) << 1) | ((uint32_t)code << 10);
        r = sr; c--;
      }
    }
  }
  for (int q = 0; q < r; q++) cols_out[q] = COL_CLIP;
  res->score = best; res->abc = c; res->aec = aec; res->abr = r; res->gaps = gaps; res->gap_desc = gap_desc;
  return true;
}

}  // namespace mia
