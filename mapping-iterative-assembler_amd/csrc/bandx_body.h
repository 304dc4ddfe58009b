// bandx_body.h -- the windowed DP of dyn_prog / max_sg_score / find_align_begin / populate_pwaln_to_begin
// (/root/reference/src/mia.c:740-981,1278-1302,612-637,1440-1497) for ANY substitution matrix (sub_mat_score /
// find_sm_depth, src/pssm.c:6-46), confined to a band of diagonals that provably holds every alignment that matters.
// One read per thread.  Three stages, all exact:
//
//   bx_plan    the band from 10-mer anchors (pigeonhole over losses); reads whose band is ONE diagonal are finished here
//   bx_values  band DP, values only; with the plan's diagonal proof most reads are finished without a trace
//   bx_trace   band DP with a one-byte trace per cell and the reference's traceback, for everything with a gap
//
// LOSSES.  st = strand matrix of the read (src/mia_main.c:179-184), d(r) = find_sm_depth(r, len), b_r = read base,
// M(r) = max over reference bases i of sm[st][d(r)][i][b_r].  No path scores more than U = sum_r M(r); loss = U - value.
// A row aligned to column c loses M(r) - sm[d][ref_c][b_r] >= 0; the tables are only used when the identical base is the
// best one (loss 0 on identity, checked on the host) and every M > 0, so a skipped row (insert, soft clip) loses M(r) > 0.
// Events: a column gap of n costs GOP + GEP n; n skipped rows cost GOP + GEP n (+ GEP for a late start) + their M.
//
// PIGEONHOLE.  Cut nb disjoint 10-mers ("blocks") out of the read.  A path crosses block b CLEANLY if its ten rows sit
// on one diagonal over identical bases: the reference then holds that 10-mer right there (an ANCHOR; the table lists
// them all).  Crossing it any other way costs at least
//   dl_b = min( min over its rows of delta[st][d(r)],   a substitution: delta = least loss of a non-identical base
//               GOP + GEP,                                a column gap inside it
//               E )                                       skipped rows: n of them touch <= ceil((n-1)/10)+1 blocks and cost
//                                                         >= GOP + (GEP + min M) n;  E = the least quotient
// so a path that loses less than sum_b dl_b crosses some block cleanly, i.e. runs through an anchor; and from an anchor it
// strays at most g = (loss - GOP) / GEP diagonals (every change of diagonal is paid for by a gap).  B0, the loss of one
// path that is cheap to write down, bounds the optimum's loss: every alignment that can win OR TIE lies on the diagonals
// [lowest anchor - g, highest anchor + g].  The band argument itself (band_body.h) never looked at the matrix.
//
// With one anchor diagonal and g = 0 the band is that diagonal: the pure diagonal path D is the only path that loses
// <= B0 (a late start on it costs > GOP + GEP), hence the unique optimum, and with D(r-1) >= -P(r+1) for every row the
// reference's traceback walks it (align_body_quad_plain.h's proof) -- no DP at all.
#pragma once
#include <stdint.h>

#include "diag_filter.h"
#include "mia_layout.h"

namespace mia {

constexpr int BX_BLOCKS = 9;          // 10-mers cut out of a read: len / 10 of them, at most 9
constexpr int BX_MIN_BLOCKS = 3;
constexpr int BX_MAXW = 32;           // widest band
constexpr int BX_NEG = -(1 << 22);    // "no such cell"
constexpr int BX_SUB_ROW = 8;         // words per (strand, depth, read base) row of the substitution table: codes 0..4
constexpr int BX_SUB_WORDS = 2 * 31 * 4 * BX_SUB_ROW;
constexpr int BX_NIB_LEAD = 320;      // nibbles in front of reference position 0 (multiple of 8, >= MAX_READ)
constexpr int BX_NIB_TAIL = 704;
constexpr int BX_NCLS = 4;            // band classes: 8, 16, 24, 32 diagonals
MIA_HD inline int bx_class_of(int w) { return w <= 8 ? 0 : (w <= 16 ? 1 : (w <= 24 ? 2 : 3)); }

// what the host derives from the two matrices (mia_hip_set_pssm) -- see bx_make_tables
struct BxTab {
  const int32_t* sub;      // [2][31][4][8]: sm[strand][depth][ref code 0..4][read base], read base major
  const int32_t* mrow;     // [2][31][4]:    M
  const int32_t* delta;    // [2][31]:       least loss of a non-identical base at that depth
  int32_t ev_block;        // E
  int32_t min_m, max_m;
};

MIA_HD inline int64_t bx_nib_words(int64_t n_codes) { return (BX_NIB_LEAD + n_codes + BX_NIB_TAIL) / 8 + 2; }

// host side of BxTab; false: the band pipeline cannot be used with these matrices
inline bool bx_make_tables(const int32_t* fwd, const int32_t* rc, int32_t* sub, int32_t* mrow, int32_t* delta, int32_t* ev_block, int32_t* min_m,
                           int32_t* max_m) {
  const int32_t* tabs[2] = {fwd, rc};
  int mn = 1 << 30, mx = -(1 << 30);
  for (int st = 0; st < 2; st++)
    for (int d = 0; d < 31; d++) {
      int dl = 1 << 30;
      for (int b = 0; b < 4; b++) {
        int32_t* row = sub + ((st * 31 + d) * 4 + b) * BX_SUB_ROW;
        for (int i = 0; i < BX_SUB_ROW; i++) row[i] = tabs[st][(d * 5 + (i < 5 ? i : 4)) * 5 + b];
        int m = row[0];
        for (int i = 1; i < 4; i++) if (row[i] > m) m = row[i];
        if (row[b] != m || m <= 0 || row[4] > m) return false;      // identity is the best base; skipping a row never pays; N is no better
        for (int i = 0; i < 5; i++) if (row[i] > 4000 || row[i] < -4000) return false;   // (value * 256 + code must fit a word, with room)
        mrow[(st * 31 + d) * 4 + b] = m;
        for (int i = 0; i < 4; i++) if (i != b && m - row[i] < dl) dl = m - row[i];
        if (m < mn) mn = m;
        if (m > mx) mx = m;
      }
      delta[st * 31 + d] = dl;
    }
  int e = 1 << 30;
  for (int n = 1; n <= 2 * MAX_READ; n++) {
    const int q = (GOP + (GEP + mn) * n) / ((n - 1 + 9) / 10 + 1);
    if (q < e) e = q;
  }
  if (e > GOP + GEP) e = GOP + GEP;
  *ev_block = e; *min_m = mn; *max_m = mx;
  return e > 0;
}

// 10-mer index (diag_filter.h: kmer_at's packing) of read rows o .. o+9, from the packed nibbles
MIA_HD inline int64_t bx_kmer(const uint32_t* pw, int len2, int o) {
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

// mode of a planned read
constexpr int BX_NONE = 0;     // not planned: the full-window kernels take it
constexpr int BX_DONE = 1;     // finished by the plan: pure diagonal dstar, score = U - b0
constexpr int BX_VALUES = 2;   // values-only DP, then the check  best == U - b0 at diagonal dstar
constexpr int BX_TRACE = 3;    // straight to the trace DP
struct BxPlan { int mode, d0, w, dstar, b0, edge; };

// loss of read row r (base b, depth d) aligned to reference code i
MIA_HD inline int bx_loss(const BxTab& T, int st, int d, int b, int i) {
  return T.mrow[(st * 31 + d) * 4 + b] - T.sub[((st * 31 + d) * 4 + b) * BX_SUB_ROW + i];
}

template <int NW>
MIA_HD inline void bx_plan_nw(const RefPlanes& rp, const KmerOcc& ko, const uint8_t* codes, int64_t n_ref, int s, int len1, const uint8_t* read_packed,
                              int len2, int st, const BxTab& T, BxPlan* out) {
  out->mode = BX_NONE;
  DiagScan<NW> sc;
  if (!sc.load_read(read_packed, len2)) return;              // a read with N
  const int R = len2 - 1, nb_cut = len2 / DF_K < BX_BLOCKS ? len2 / DF_K : BX_BLOCKS;
  const uint32_t* pw = reinterpret_cast<const uint32_t*>(read_packed);
  // all look-ups first, then their use: nine independent loads in flight instead of nine round trips
  int32_t cn[BX_BLOCKS], ps[BX_BLOCKS][DF_KCAP];
#pragma unroll
  for (int b = 0; b < BX_BLOCKS; b++) {
    cn[b] = DF_KCAP + 1;
    if (b < nb_cut) {
      const int64_t idx = bx_kmer(pw, len2, (int)((int64_t)b * (len2 - DF_K) / (nb_cut - 1)));
      cn[b] = ko.cnt[idx];
#pragma unroll
      for (int k = 0; k < DF_KCAP; k++) ps[b][k] = ko.pos[idx * DF_KCAP + k];
    }
  }
  int nb = 0, a_lo = 1 << 20, a_hi = -(1 << 20), d_first = 0, d_last = 0, budget = -1;
  bool any = false;
#pragma unroll
  for (int b = 0; b < BX_BLOCKS; b++) {
    if (cn[b] > DF_KCAP) continue;                   // no such block, or an overloaded 10-mer: not part of the pigeonhole
    nb++;
    const int o = (int)((int64_t)b * (len2 - DF_K) / (nb_cut - 1));
    // what breaking this block costs at least (depths are monotone along the read: the ends of the block bound them)
    {
      const int dlo = sm_depth(o, len2), dhi = sm_depth(o + DF_K - 1, len2);
      int dl = T.ev_block;
      for (int d = dlo; d <= dhi; d++) { const int v = T.delta[st * 31 + d]; if (v < dl) dl = v; }
      budget += dl;
    }
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
  if (nb < BX_MIN_BLOCKS || !any || a_hi - a_lo >= BX_MAXW) return;
  if (d_first < 0 || d_first > len1 - len2 || d_last < 0 || d_last > len1 - len2) return;   // keep the written-down path inside the window
  // the loss of one valid path: rows [0, t) on d_first, one gap, the rest on d_last (or the plain diagonal if they agree),
  // row by row from the mismatch masks (identical bases lose nothing)
  auto row_loss = [&](int r, int dg) {
    const int b = (int)((read_packed[r >> 1] >> ((r & 1) * 4)) & 3u), i = codes[(int64_t)s + dg + r];
    return bx_loss(T, st, sm_depth(r, len2), b, i);
  };
  uint64_t m1[NW];
  sc.seek(rp, (int64_t)s + d_first);
#pragma unroll
  for (int j = 0; j < NW; j++) m1[j] = sc.mis(j);
  int b0 = 0;
  bool proof = true;      // D(r-1) >= -P(r+1) for every row r >= 1 of the pure diagonal (only asked for when d_first == d_last)
  if (d_first == d_last) {
#pragma unroll
    for (int j = 0; j < NW; j++) {
      uint64_t m = m1[j];
      while (m) {
        const int q = j * 64 + df_ctz(m);
        m &= m - 1;
        b0 += row_loss(q, d_first);
        // D(q) >= (q+1) min M - loss so far; the bound -P(q+2) falls with q while D only drops at mismatches
        if (b0 > (q + 1) * T.min_m + GOP + GEP * (q + 2)) proof = false;
      }
    }
  } else {
    uint64_t m2[NW + 1];
    sc.seek(rp, (int64_t)s + d_last);
#pragma unroll
    for (int j = 0; j < NW; j++) m2[j] = sc.mis(j);
    m2[NW] = 0;
    // column gap (d_last > d_first: the rows from t on continue d_last - d_first columns further right) or row gap (the read
    // skips d_first - d_last rows): the switch row t with the fewest mismatches, then that path's loss
    const int shift = d_last - d_first, skip = shift < 0 ? -shift : 0;            // |shift| < BX_MAXW
    uint64_t m2s[NW + 1];
#pragma unroll
    for (int j = 0; j <= NW; j++) m2s[j] = m2[j];
    if (skip) {                                    // bit t of m2s := row t + skip
#pragma unroll
      for (int j = 0; j < NW; j++) m2s[j] = (m2[j] >> skip) | (m2[j + 1] << (64 - skip));
    }
    int prefix = 0, suffix = 0, best = 1 << 20, tbest = -1;
#pragma unroll
    for (int j = 0; j < NW; j++) suffix += df_popc(m2s[j]);
#pragma unroll
    for (int j = 0; j < NW; j++) {
      for (int q = 0; q < 64; q++) {
        const int t = j * 64 + q + 1;
        if (t + skip > R) break;
        prefix += (int)((m1[j] >> q) & 1);
        suffix -= (int)((m2s[j] >> q) & 1);
        if (prefix + suffix < best) { best = prefix + suffix; tbest = t; }
      }
    }
    if (tbest < 0) return;
    b0 = shift > 0 ? GOP + GEP * shift : GOP + (GEP + T.max_m) * skip;
#pragma unroll
    for (int j = 0; j < NW; j++) {
      uint64_t m = m1[j];
      while (m) { const int q = j * 64 + df_ctz(m); m &= m - 1; if (q < tbest) b0 += row_loss(q, d_first); }
      m = m2[j];
      while (m) { const int q = j * 64 + df_ctz(m); m &= m - 1; if (q >= tbest + skip) b0 += row_loss(q, d_last); }
    }
  }
  if (b0 > budget) return;
  // (one diagonal more where the window's first column is within reach, as band_body.h)
  int g = b0 < GOP + GEP ? 0 : (b0 - GOP) / GEP;
  if (a_lo - g - 1 < 0) g++;
  const int d0 = a_lo - g, w = a_hi - a_lo + 2 * g + 1;
  if (w > BX_MAXW) return;
  out->d0 = d0; out->w = w; out->b0 = b0; out->dstar = d_first;
  if (d_first != d_last || !proof) { out->mode = BX_TRACE; }
  else out->mode = w == 1 ? BX_DONE : BX_VALUES;
  // the widest band of the read's class must not leave the window anywhere for the plain form of the recurrence
  const int wc = 8 * (bx_class_of(w) + 1);
  out->edge = !(d0 >= 0 && len2 - 1 + d0 + wc <= len1);
}

MIA_HD inline void bx_plan(const RefPlanes& rp, const KmerOcc& ko, const uint8_t* codes, int64_t n_ref, int s, int len1, const uint8_t* read_packed,
                           int len2, int st, const BxTab& T, BxPlan* out) {
  out->mode = BX_NONE;
  if (!ko.cnt || len2 < BX_MIN_BLOCKS * DF_K || len2 > MAX_READ || len1 < len2 || len1 > DF_MAX_LEN1 || s < 0 || (int64_t)s + len1 > n_ref) return;
  if (!all_bases(rp, s, (int64_t)s + len1)) return;
  switch ((len2 + 63) >> 6) {
    case 1: bx_plan_nw<1>(rp, ko, codes, n_ref, s, len1, read_packed, len2, st, T, out); break;
    case 2: bx_plan_nw<2>(rp, ko, codes, n_ref, s, len1, read_packed, len2, st, T, out); break;
    case 3: bx_plan_nw<3>(rp, ko, codes, n_ref, s, len1, read_packed, len2, st, T, out); break;
    default: bx_plan_nw<4>(rp, ko, codes, n_ref, s, len1, read_packed, len2, st, T, out); break;
  }
}

// sum of M over the rows of a read (U); -1 for a read with N
MIA_HD inline int bx_umax(const int32_t* mrow, const uint8_t* read_packed, int len2, int st) {
  int u = 0;
  for (int r = 0; r < len2; r++) {
    const int b = (read_packed[r >> 1] >> ((r & 1) * 4)) & 15;
    if (b > 3) return -1;
    u += mrow[(st * 31 + sm_depth(r, len2)) * 4 + b];
  }
  return u;
}

// ---- the band DP ---------------------------------------------------------------------------------------------------
// W nibbles of reference codes from nibble index `nib` on (BX_NIB_LEAD + reference position), as W/8 words
template <int NWD>
MIA_HD inline void bx_codes(const uint32_t* refnib, int64_t nib, uint32_t* cw) {
  const int64_t q = nib >> 3;
  const int sh = (int)(nib & 7) * 4;
  uint32_t lo = refnib[q];
#pragma unroll
  for (int k = 0; k < NWD; k++) {
    const uint32_t hi = refnib[q + k + 1];
    cw[k] = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
    lo = hi;
  }
}

struct BxResult { int score, abc, aec, abr, gaps; uint32_t gap_desc; };

// Values only.  sub: the substitution table of the read's strand (31 x 4 x BX_SUB_ROW words; LDS on the device).
// Returns the first maximum of the last row: *best and its band index *bj (-1: none).
template <int W, bool EDGE>
MIA_HD inline void bx_values(const uint32_t* refnib, int s, int len1, const uint32_t* rwords, int len2, int d0, const int32_t* sub, int* best_out, int* bj_out) {
  constexpr int NWD = W / 8;
  int32_t P[W], H[W];
  uint32_t cw[NWD];
  uint32_t rw = rwords[0];
  auto live_mask = [&](int c0) -> uint32_t {             // cells whose column c0 + j lies inside the window
    const int jlo = c0 < 0 ? -c0 : 0, jhi = (len1 - c0) < W ? (len1 - c0) : W;
    return jhi > jlo ? ((jhi >= 32 ? ~0u : ((1u << jhi) - 1u)) & ~((1u << jlo) - 1u)) : 0u;
  };
  {
    bx_codes<NWD>(refnib, (int64_t)s + d0 + BX_NIB_LEAD, cw);
    const int32_t* row = sub + ((0 * 4) + (int)(rw & 3u)) * BX_SUB_ROW;          // depth 0
    const uint32_t live = EDGE ? live_mask(d0) : ~0u;
#pragma unroll
    for (int j = 0; j < W; j++) {
      H[j] = BX_NEG;
      const int v = row[(cw[j >> 3] >> (4 * (j & 7))) & 7u];
      P[j] = ((live >> j) & 1u) ? v : BX_NEG;
    }
  }
  for (int r = 1; r < len2; r++) {
    const int c0 = r + d0;
    bx_codes<NWD>(refnib, (int64_t)s + c0 + BX_NIB_LEAD, cw);
    if ((r & 7) == 0) rw = rwords[r >> 3];
    const int32_t* row = sub + (sm_depth(r, len2) * 4 + (int)((rw >> (4 * (r & 7))) & 3u)) * BX_SUB_ROW;
    uint32_t live = ~0u, col0 = 0u;
    if (EDGE) {
      live = live_mask(c0);
      col0 = (c0 <= 0 && c0 > -W) ? (1u << (-c0)) : 0u;
    }
    const int fresh = -(GOP + GEP * (r + 1));
    int G = BX_NEG;
#pragma unroll
    for (int j = 0; j < W; j++) {
      const int pd = P[j], h = H[j];
      const int sb = row[(cw[j >> 3] >> (4 * (j & 7))) & 7u];
      const int x = pd > G ? (pd > h ? pd : h) : (G > h ? G : h);
      int cur = fresh > x ? fresh : x + sb;
      if (EDGE) {
        if ((col0 >> j) & 1u) cur = sb + fresh;                                       // src/mia.c:805-822
        if (!((live >> j) & 1u)) cur = BX_NEG;
      }
      const int cand = pd - (GOP + GEP);
      G = G - GEP > cand ? G - GEP : cand;
      if (j >= 1) H[j - 1] = h - GEP > cand ? h - GEP : cand;
      P[j] = cur;
    }
    H[W - 1] = BX_NEG;
  }
  int best = BX_NEG, bj = -1;
#pragma unroll
  for (int j = 0; j < W; j++) if (P[j] > best) { best = P[j]; bj = j; }
  if (best <= BX_NEG / 2) bj = -1;
  *best_out = best; *bj_out = bj;
}

// byte k of word w := the low byte of v
MIA_HD inline uint32_t bx_put(uint32_t w, uint32_t v, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(v, w, 0x03020100u ^ ((uint32_t)(4 ^ k) << (8 * k)));
#else
  return (w & ~(0xFFu << (8 * k))) | ((v & 0xFFu) << (8 * k));
#endif
}

// With a trace.  Everything a cell compares is packed value * 256 + code (band_body.h): diagonal 0xFF beats column gap
// 0x40 + n beats row gap n on equal values, the larger n (the earlier source) within a kind; the winning code byte is the
// trace; a new start (0x80) only if strictly better.  sub256: the table times 256.  trace: W/4 words per row, row r at
// trace + r * row_words.  false: the traceback met the reference's index-0 quirk (or left the band): not finished here.
template <int W, bool EDGE>
MIA_HD inline bool bx_trace(const uint32_t* refnib, int s, int len1, const uint32_t* rwords, int len2, int d0, const int32_t* sub256, uint32_t* trace,
                            int64_t row_words, int16_t* cols_out, BxResult* res) {
  constexpr int NWD = W / 8;
  constexpr int DEAD = BX_NEG * 256;
  constexpr int STEP = 1 - GEP * 256;                 // a running maximum ages by one position: value - GEP, length + 1
  constexpr int CAND = -GOP * 256 + STEP - 0xFF;      // a cell (packed as a diagonal source) becomes a gap source
  const int R = len2 - 1;
  int32_t P[W], H[W];
  uint32_t cw[NWD];
  uint32_t rw = rwords[0];
  auto live_mask = [&](int c0) -> uint32_t {
    const int jlo = c0 < 0 ? -c0 : 0, jhi = (len1 - c0) < W ? (len1 - c0) : W;
    return jhi > jlo ? ((jhi >= 32 ? ~0u : ((1u << jhi) - 1u)) & ~((1u << jlo) - 1u)) : 0u;
  };
  {
    bx_codes<NWD>(refnib, (int64_t)s + d0 + BX_NIB_LEAD, cw);
    const int32_t* row = sub256 + ((0 * 4) + (int)(rw & 3u)) * BX_SUB_ROW;
    const uint32_t live = EDGE ? live_mask(d0) : ~0u;
#pragma unroll
    for (int j = 0; j < W; j++) {
      H[j] = DEAD;
      const int v = row[(cw[j >> 3] >> (4 * (j & 7))) & 7u];
      P[j] = ((live >> j) & 1u) ? (v | 0xFF) : DEAD;
    }
#pragma unroll
    for (int k = 0; k < W / 4; k++) trace[k] = 0xFFFFFFFFu;
  }
  for (int r = 1; r < len2; r++) {
    const int c0 = r + d0;
    bx_codes<NWD>(refnib, (int64_t)s + c0 + BX_NIB_LEAD, cw);
    if ((r & 7) == 0) rw = rwords[r >> 3];
    const int32_t* row = sub256 + (sm_depth(r, len2) * 4 + (int)((rw >> (4 * (r & 7))) & 3u)) * BX_SUB_ROW;
    uint32_t live = ~0u, col0 = 0u;
    if (EDGE) {
      live = live_mask(c0);
      col0 = (c0 <= 0 && c0 > -W) ? (1u << (-c0)) : 0u;
    }
    const int f0 = -(GOP + GEP * (r + 1)) * 256, f0s = f0 | 0x80;
    int G = DEAD;
    uint32_t tw[W / 4];
#pragma unroll
    for (int k = 0; k < W / 4; k++) tw[k] = 0;
#pragma unroll
    for (int j = 0; j < W; j++) {
      const int pd = P[j] | 0xFF, h = H[j], gc = G | 0x40;
      const int sb = row[(cw[j >> 3] >> (4 * (j & 7))) & 7u];
      const int x = pd > gc ? (pd > h ? pd : h) : (gc > h ? gc : h);
      int cur = f0 > x ? f0s : x + sb;                  // fresh must beat all three strictly (its code byte is 0 in f0)
      if (EDGE) {
        if ((col0 >> j) & 1u) cur = (sb + f0) | 0xFF;
        if (!((live >> j) & 1u)) cur = DEAD;
      }
      tw[j >> 2] = bx_put(tw[j >> 2], (uint32_t)cur, j & 3);
      const int cand = pd + CAND;
      G = G + STEP > cand ? G + STEP : cand;
      if (j >= 1) H[j - 1] = h + STEP > cand ? h + STEP : cand;
      P[j] = cur;
    }
    H[W - 1] = DEAD;
    uint32_t* tr = trace + (int64_t)r * row_words;
#pragma unroll
    for (int k = 0; k < W / 4; k++) tr[k] = tw[k];
  }
  // max_sg_score: first maximum of the last row (src/mia.c:1278-1302)
  int best = BX_NEG, bj = -1;
#pragma unroll
  for (int j = 0; j < W; j++) if ((P[j] >> 8) > best) { best = P[j] >> 8; bj = j; }
  if (bj < 0 || best <= BX_NEG / 2) return false;
  // find_align_begin + populate_pwaln_to_begin (src/mia.c:612-637, 1440-1497); eight rows of one band index per fetch
  int r = R, c = R + d0 + bj, gaps = 0;
  uint32_t gap_desc = 0;
  const int aec = c;
  bool stop = false;
  while (!stop) {
    const int j = c - r - d0;
    if (j < 0 || j >= W) return false;
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
