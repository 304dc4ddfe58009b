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
// N COLUMNS.  A reference column that holds an ambiguity code scores sm[d][N][b] whatever the read base (base2inx,
// src/mia.c:1521-1541): a known loss lambda(r) = M(r) - sm[d][N][b_r] >= 0, small next to a substitution (210 against 800 with
// the flat matrix).  Such a column cannot break a block: "cleanly" means ten rows on one diagonal over columns that are
// identical OR N, and the table lists those places too -- a reference 10-mer with k <= BX_WILD ambiguity codes is entered
// under all 4^k spellings (k_kmer_hash).  A clean crossing over more than BX_WILD of them is in no table, so it must
// cost at least dl_b like any other way past the block: dl_b is capped at (BX_WILD + 1) min lambda over the block's
// rows.  Everything else is unchanged: B0 counts the N columns of the written-down path, gaps cost what they cost.
//
// With one anchor diagonal and g = 0 the band is that diagonal: the pure diagonal path D is the only path that loses
// <= B0 (a late start on it costs > GOP + GEP), hence the unique optimum, and with D(r-1) >= -P(r+1) for every row the
// reference's traceback walks it (align_body_quad_plain.h's proof) -- no DP at all.
#pragma once
#include <stdint.h>

#include "diag_filter.h"
#include "mia_layout.h"

namespace mia {

constexpr int BX_BLOCKS = 12;         // 10-mers cut out of a read: len / 10 of them, at most 9 (reads up to 128 bases) or 12 (longer ones)
template <int NW> constexpr int bx_nb_max() { return NW <= 2 ? 9 : 12; }   // NW: 64-row words of the longest read
constexpr int BX_MIN_BLOCKS = 3;
constexpr int BX_MAXW = 64;           // widest band (BxTab::maxw: what the kernels in use hold -- 64 spread over eight lanes, 32 in one lane)
constexpr int BX_CLUSTER_TRIGGER = 8;  // anchors further apart than this are clustered around their median ...
constexpr int BX_CLUSTER_RADIUS = 12; // ... keeping those within this many diagonals of it
constexpr int BX_NEG = -(1 << 22);    // "no such cell"
constexpr int BX_SUB_ROW = 8;         // words per (strand, depth, read base) row of the substitution table: codes 0..4
constexpr int BX_SUB_WORDS = 2 * 31 * 4 * BX_SUB_ROW;
constexpr int BX_NIB_LEAD = 320;      // nibbles in front of reference position 0 (multiple of 8, >= MAX_READ)
constexpr int BX_NIB_TAIL = 704;
constexpr int BX_NCLS = 5;            // band classes: 8, 16, 24, 32, 64 diagonals (the last one only with a read spread over lanes, bandx_lanes.h)
constexpr int BX_WILD = 3;            // most N columns in a reference 10-mer that is still entered in the table (4^k spellings)
MIA_HD inline int bx_class_of(int w) { return w <= 8 ? 0 : (w <= 16 ? 1 : (w <= 24 ? 2 : (w <= 32 ? 3 : 4))); }
MIA_HD inline int bx_class_width(int c) { return c < 4 ? 8 * (c + 1) : 64; }

// what the host derives from the two matrices (mia_hip_set_pssm) -- see bx_make_tables
struct BxTab {
  const int32_t* sub;      // [2][31][4][8]: sm[strand][depth][ref code 0..4][read base], read base major
  const int32_t* mrow;     // [2][31][4]:    M
  const int16_t* loss;     // [2][31][4][4]: M - sm, by (strand, depth, read base, reference base); behind it [2][31][4]: the
                           // same over an N column (BX_LOSS_N), and the N-column credit (BX_LOSS_KAP, see bx_finish)
  const int16_t* dl;       // [2][MAX_READ+1][BX_BLOCKS]: what breaking block b of a read of that length costs at least;
                           // behind it the stray tables (bx_stray_off): what straying n diagonals costs at least, net of the
                           // blocks the gaps themselves break
  int32_t min_m, max_m;
  int32_t maxw;            // widest band the DP kernels in use hold (32: one read per lane; BX_MAXW: a read over eight lanes)
};
constexpr int BX_LOSS_N = 2 * 31 * 4 * 4;
constexpr int BX_LOSS_KAP = BX_LOSS_N + 2 * 31 * 4;       // [2][31][31]: what crossing an N column costs at least, by strand and depth range
constexpr int BX_LOSS_NCRED = BX_LOSS_KAP + 2 * 31 * 31;    // [2][32]: what ANY path pays at least for k N columns it spans (bx_window_nmin), all 0 = not usable
constexpr int BX_NCRED_K = 32;
constexpr int BX_LOSS_KAP2 = BX_LOSS_NCRED + 2 * BX_NCRED_K;  // [2][31][31]: the least a ROW over an N column costs, by strand and depth range (not capped at GEP)
constexpr int BX_LOSS_FDL = BX_LOSS_KAP2 + 2 * 31 * 31;  // [2][31]: what breaking a FINE block costs at least, by strand and depth of its cheapest row (bx_fine_anchors); 0 = not usable
constexpr int BX_LOSS_WORDS = BX_LOSS_FDL + 2 * 31;
constexpr int BX_FQ = 6;               // rows of a fine block
constexpr int BX_FINE_RADIUS = 3;      // fine anchors this close to the 10-mer anchors' diagonals are kept, the others set aside
constexpr int BX_GMAX = 32;            // stray tables: phi / psi for 0..BX_GMAX diagonals
constexpr int BX_DL_BLOCKS = 2 * (MAX_READ + 1) * BX_BLOCKS;                       // dl proper
constexpr int BX_DL_FINE = BX_DL_BLOCKS + 2 * (MAX_READ + 1) * 2 * (BX_GMAX + 1);   // + [strand][len][down | up][0..BX_GMAX]
constexpr int BX_DL_WORDS = BX_DL_FINE + 2 * 2 * (BX_GMAX + 1);                    // + the stray tables of the fine blocks: [strand][down | up][0..BX_GMAX]
MIA_HD inline int bx_stray_off(int st, int len2, int up) { return BX_DL_BLOCKS + ((st * (MAX_READ + 1) + len2) * 2 + up) * (BX_GMAX + 1); }
MIA_HD inline int bx_stray_off_fine(int st, int up) { return BX_DL_FINE + (st * 2 + up) * (BX_GMAX + 1); }
// fine blocks: len2 / BX_FQ of them, spread like the 10-mers
MIA_HD inline int bx_fine_blocks_of(int len2) { return len2 / BX_FQ; }
MIA_HD inline int bx_fine_block_row(int b, int len2, int nb_cut) { return (int)((int64_t)b * (len2 - BX_FQ) / (nb_cut - 1)); }

MIA_HD inline int64_t bx_nib_words(int64_t n_codes) { return (BX_NIB_LEAD + n_codes + BX_NIB_TAIL) / 8 + 2; }
// first row of block b of a read of len2 bases cut into nb_cut blocks
MIA_HD inline int bx_block_row(int b, int len2, int nb_cut) { return (int)((int64_t)b * (len2 - DF_K) / (nb_cut - 1)); }
MIA_HD inline int bx_blocks_of(int len2) { const int cap = len2 > 128 ? BX_BLOCKS : 9; return len2 / DF_K < cap ? len2 / DF_K : cap; }

// host side of BxTab; false: the band pipeline cannot be used with these matrices
inline bool bx_make_tables(const int32_t* fwd, const int32_t* rc, int32_t* sub, int32_t* mrow, int16_t* loss, int16_t* dl, int32_t* min_m, int32_t* max_m) {
  const int32_t* tabs[2] = {fwd, rc};
  int mn = 1 << 30, mx = -(1 << 30);
  int delta[2][31], lamn[2][31];
  for (int st = 0; st < 2; st++)
    for (int d = 0; d < 31; d++) {
      int dmin = 1 << 30, lmin = 1 << 30;
      for (int b = 0; b < 4; b++) {
        int32_t* row = sub + ((st * 31 + d) * 4 + b) * BX_SUB_ROW;
        for (int i = 0; i < BX_SUB_ROW; i++) row[i] = tabs[st][(d * 5 + (i < 5 ? i : 4)) * 5 + b];
        int m = row[0];
        for (int i = 1; i < 4; i++) if (row[i] > m) m = row[i];
        if (row[b] != m || m <= 0 || row[4] > m) return false;      // identity is the best base; skipping a row never pays; N is no better
        for (int i = 0; i < 5; i++) if (row[i] > 4000 || row[i] < -4000) return false;   // (value * 256 + code must fit a word, with room)
        mrow[(st * 31 + d) * 4 + b] = m;
        loss[BX_LOSS_N + (st * 31 + d) * 4 + b] = (int16_t)(m - row[4]);
        if (m - row[4] < lmin) lmin = m - row[4];
        for (int i = 0; i < 4; i++) {
          loss[((st * 31 + d) * 4 + b) * 4 + i] = (int16_t)(m - row[i]);
          if (i != b && m - row[i] < dmin) dmin = m - row[i];
        }
        if (m < mn) mn = m;
        if (m > mx) mx = m;
      }
      delta[st][d] = dmin;
      lamn[st][d] = lmin;
    }
  // kap[st][dlo][dhi]: an N column crossed by a row of depth dlo..dhi, or inside a column gap, costs at least this
  for (int st = 0; st < 2; st++)
    for (int a = 0; a < 31; a++)
      for (int b = 0; b < 31; b++) {
        int v = GEP;
        for (int d = a; d <= b; d++) if (lamn[st][d] < v) v = lamn[st][d];
        loss[BX_LOSS_KAP + (st * 31 + a) * 31 + b] = (int16_t)(a <= b && v > 0 ? v : 0);
        int v2 = 1 << 14;
        for (int d = a; d <= b; d++) if (lamn[st][d] < v2) v2 = lamn[st][d];
        loss[BX_LOSS_KAP2 + (st * 31 + a) * 31 + b] = (int16_t)(a <= b && v2 > 0 && v2 < (1 << 14) ? v2 : 0);
      }
  // skipped rows (an insert, a soft clip): n of them touch at most ceil((n-1)/10)+1 blocks and cost at least GOP + (GEP + min M) n
  int e = GOP + GEP;                    // (a column gap inside a block)
  for (int n = 1; n <= 2 * MAX_READ; n++) {
    const int q = (GOP + (GEP + mn) * n) / ((n - 1 + 9) / 10 + 1);
    if (q < e) e = q;
  }
  if (e <= 0) return false;
  for (int st = 0; st < 2; st++)
    for (int len2 = 0; len2 <= MAX_READ; len2++)
      for (int b = 0; b < BX_BLOCKS; b++) {
        int v = 0;
        const int nb_cut = bx_blocks_of(len2);
        if (nb_cut >= BX_MIN_BLOCKS && b < nb_cut) {
          const int o = bx_block_row(b, len2, nb_cut);
          v = e;
          int lam = 1 << 20;
          for (int r = o; r < o + DF_K; r++) {
            const int dd = delta[st][sm_depth(r, len2)], ll = lamn[st][sm_depth(r, len2)];
            if (dd < v) v = dd;
            if (ll < lam) lam = ll;
          }
          if ((BX_WILD + 1) * lam < v) v = (BX_WILD + 1) * lam;      // a clean crossing over more N columns than the table knows
        }
        dl[(st * (MAX_READ + 1) + len2) * BX_BLOCKS + b] = (int16_t)v;
      }
  // STRAY TABLES.  A path that reaches a diagonal n above (below) its anchors holds column gaps (skipped rows) that add up
  // to n or more.  An event costs c_e and spoils at most the blocks it touches, which the pigeonhole sum would otherwise
  // charge dl each: a column gap touches one block at most, a run of k skipped rows t(k) blocks (exact, from the block
  // positions of this read length).  psi(n) / phi(n) = the least of sum(c_e - dl_max * touched) over all ways to make up n,
  // never below 0 per event (dl <= e <= every event's cost per touched block, see above).
  for (int st = 0; st < 2; st++)
    for (int len2 = 0; len2 <= MAX_READ; len2++) {
      int16_t* dn = dl + bx_stray_off(st, len2, 0);
      int16_t* up = dl + bx_stray_off(st, len2, 1);
      for (int n = 0; n <= BX_GMAX; n++) { dn[n] = 0; up[n] = 0; }
      const int nb_cut = bx_blocks_of(len2);
      if (nb_cut < BX_MIN_BLOCKS) continue;
      int dmax = 0;
      for (int b = 0; b < nb_cut; b++) { const int v = dl[(st * (MAX_READ + 1) + len2) * BX_BLOCKS + b]; if (v > dmax) dmax = v; }
      constexpr int NMAX = 3 * BX_GMAX;
      int fdn[NMAX + 1], fup[NMAX + 1];
      int orow[BX_BLOCKS];
      for (int b = 0; b < nb_cut; b++) orow[b] = bx_block_row(b, len2, nb_cut);
      for (int n = 1; n <= NMAX; n++) {
        // most blocks a run of n rows can meet: block b meets rows q .. q + n - 1 iff q - DF_K < o_b < q + n -- the most block starts
        // in an open interval of n + DF_K - 1 integers, found with two pointers over the ascending starts (the first version tried
        // every q and every block: 100 ms of every mia_hip_set_pssm)
        int t = 0;
        for (int lo = 0, hi = 0; lo < nb_cut; lo++) {
          if (hi < lo) hi = lo;
          while (hi + 1 < nb_cut && orow[hi + 1] - orow[lo] <= n + DF_K - 2) hi++;
          if (hi - lo + 1 > t) t = hi - lo + 1;
        }
        const int vd = GOP + (GEP + mn) * n - dmax * t, vu = GOP + GEP * n - dmax;
        fdn[n] = vd > 0 ? vd : 0;
        fup[n] = vu > 0 ? vu : 0;
      }
      int phi[BX_GMAX + 1], psi[BX_GMAX + 1];
      phi[0] = psi[0] = 0;
      for (int d = 1; d <= BX_GMAX; d++) {
        int bd = 1 << 30, bu = 1 << 30;
        for (int n = 1; n <= NMAX; n++) {
          const int rest = d - n > 0 ? d - n : 0;
          if (fdn[n] + phi[rest] < bd) bd = fdn[n] + phi[rest];
          if (fup[n] + psi[rest] < bu) bu = fup[n] + psi[rest];
        }
        phi[d] = bd; psi[d] = bu;
      }
      for (int d = 0; d <= BX_GMAX; d++) { dn[d] = (int16_t)(phi[d] > 32000 ? 32000 : phi[d]); up[d] = (int16_t)(psi[d] > 32000 ? 32000 : psi[d]); }
    }
  // WINDOW-WIDE N CREDIT (bx_window_ncredit).  Every path spans at least len2 - (skipped rows) consecutive window columns, each
  // of them under a row or inside a column gap; an N column under a row costs lambda(r), inside a gap GEP: at least
  // kap = min(GEP, min lambda) either way.  The pigeonhole sum may be raised by kap x (fewest N columns in any stretch of len2
  // columns of the window) if nothing is counted twice:
  //   * a mismatch row sits on a column that is not N: disjoint;
  //   * a column gap of n columns inside block b pays GOP + GEP n >= dl_b + kap n             needs dl_b <= GOP;
  //   * a run of n skipped rows pays GOP + (GEP + min M) n, breaks at most t(n) blocks and shortens the span by n: no column
  //     carries more credit than (GOP + (GEP + min M) n - dl_max t(n)) / n for any n (and never more than GEP);
  //   * a clean crossing over more than BX_WILD N columns is in no table and counts as a break costing (BX_WILD + 1) min lambda --
  //     the same N columns again: the columns of such places carry no credit (bx_window_nmin).
  // FINE BLOCKS (bx_fine_anchors): the read cut into blocks of BX_FQ rows, their clean places found by comparing every
  // diagonal of the window directly (no table, so any number of N columns under a block is a clean place).  Breaking one costs
  // at least fdl(depth) = min(delta, GOP, E_q) with E_q the skipped-row quotient for t_q(n) = ceil((n-1)/q) + 1 touched blocks;
  // the stray tables are made with the largest fdl and that t_q for every read length (never above the exact ones).
  int fdmax[2] = {0, 0};
  {
    int eq = GOP + GEP;
    for (int n = 1; n <= 2 * MAX_READ; n++) {
      const int q = (GOP + (GEP + mn) * n) / ((n - 1 + BX_FQ - 1) / BX_FQ + 1);
      if (q < eq) eq = q;
    }
    if (eq > GOP) eq = GOP;
    for (int st = 0; st < 2; st++) {
      for (int d = 0; d < 31; d++) {
        const int v = delta[st][d] < eq ? delta[st][d] : eq;
        loss[BX_LOSS_FDL + st * 31 + d] = (int16_t)(eq > 0 && v > 0 ? v : 0);
        if (v > fdmax[st]) fdmax[st] = v;
      }
      int16_t* dn = dl + bx_stray_off_fine(st, 0);
      int16_t* up = dl + bx_stray_off_fine(st, 1);
      constexpr int NMAX = 3 * BX_GMAX;
      int fdn[NMAX + 1], fup[NMAX + 1];
      for (int n = 1; n <= NMAX; n++) {
        const int t = (n - 1 + BX_FQ - 1) / BX_FQ + 1;
        const int vd = GOP + (GEP + mn) * n - fdmax[st] * t, vu = GOP + GEP * n - fdmax[st];
        fdn[n] = vd > 0 ? vd : 0;
        fup[n] = vu > 0 ? vu : 0;
      }
      int phi[BX_GMAX + 1], psi[BX_GMAX + 1];
      phi[0] = psi[0] = 0;
      for (int d = 1; d <= BX_GMAX; d++) {
        int bd = 1 << 30, bu = 1 << 30;
        for (int n = 1; n <= NMAX; n++) {
          const int rest = d - n > 0 ? d - n : 0;
          if (fdn[n] + phi[rest] < bd) bd = fdn[n] + phi[rest];
          if (fup[n] + psi[rest] < bu) bu = fup[n] + psi[rest];
        }
        phi[d] = bd; psi[d] = bu;
      }
      for (int d = 0; d <= BX_GMAX; d++) { dn[d] = (int16_t)(phi[d] > 32000 ? 32000 : phi[d]); up[d] = (int16_t)(psi[d] > 32000 ? 32000 : psi[d]); }
    }
  }
  // Which row crosses an N column is not known, but every row crosses one column at most: k spanned N columns cost at least the
  // k smallest of { min(GEP, lambda_min(depth of r)) : rows r } -- the end depths (0..14, 16..30) occur once each, depth 15 as
  // often as the read is long (a read under 31 bases has a subset of the depths: the sum of ITS k smallest is no less).
  for (int st = 0; st < 2; st++) {
    int kap = GEP;
    for (int d = 0; d < 31; d++) if (lamn[st][d] < kap) kap = lamn[st][d];
    int dmax = 0;
    for (int len2 = 0; len2 <= MAX_READ; len2++)
      for (int b = 0; b < BX_BLOCKS; b++) { const int v = dl[(st * (MAX_READ + 1) + len2) * BX_BLOCKS + b]; if (v > dmax) dmax = v; }
    // skipped rows: a run of n pays GOP + (GEP + min M) n, breaks at most t(n) blocks (dl_max each) and costs the span n columns:
    // no column may carry more credit than what is left of that per row, for any n
    bool ok = kap > 0 && dmax <= GOP && mn > 0;
    int ccap = GEP;
    for (int n = 1; n <= 2 * MAX_READ; n++) {
      const int left = (GOP + (GEP + mn) * n - dmax * ((n - 1 + 9) / 10 + 1)) / n;
      const int leftq = (GOP + (GEP + mn) * n - fdmax[st] * ((n - 1 + BX_FQ - 1) / BX_FQ + 1)) / n;      // (the fine blocks' geometry)
      if (left < ccap) ccap = left;
      if (leftq < ccap) ccap = leftq;
    }
    if (ccap <= 0) ok = false;
    int v[31];
    for (int d = 0; d < 31; d++) v[d] = lamn[st][d] < ccap ? lamn[st][d] : ccap;
    const int mid = v[15];
    for (int a = 0; a < 31; a++)
      for (int b = a + 1; b < 31; b++) if (v[b] < v[a]) { const int t = v[a]; v[a] = v[b]; v[b] = t; }
    int cum = 0;
    for (int k = 0; k < BX_NCRED_K; k++) {
      loss[BX_LOSS_NCRED + st * BX_NCRED_K + k] = (int16_t)(ok ? cum : 0);
      const int next = k < 31 ? (v[k] < mid ? v[k] : mid) : mid;
      cum += next;
    }
  }
  *min_m = mn; *max_m = mx;
  return true;
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

// The same index from the read's bit planes (which the planner holds in registers anyway: no trip to the packed read)
template <int NW>
MIA_HD inline uint32_t bx_kmer_planes(const DiagScan<NW>& sc, int o) {
  const int j = o >> 6, sh = o & 63;
  uint64_t l0 = sc.rlo[0], h0 = sc.rhi[0], l1 = 0, h1 = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) {
    if (k == j) { l0 = sc.rlo[k]; h0 = sc.rhi[k]; }
    if (k == j + 1) { l1 = sc.rlo[k]; h1 = sc.rhi[k]; }
  }
  uint32_t lo = (uint32_t)((l0 >> sh) | ((l1 << 1) << (63 - sh))) & 0x3FFu;
  uint32_t hi = (uint32_t)((h0 >> sh) | ((h1 << 1) << (63 - sh))) & 0x3FFu;
  lo = (lo | (lo << 8)) & 0x00FF00FFu; hi = (hi | (hi << 8)) & 0x00FF00FFu;
  lo = (lo | (lo << 4)) & 0x0F0F0F0Fu; hi = (hi | (hi << 4)) & 0x0F0F0F0Fu;
  lo = (lo | (lo << 2)) & 0x33333333u; hi = (hi | (hi << 2)) & 0x33333333u;
  lo = (lo | (lo << 1)) & 0x55555555u; hi = (hi | (hi << 1)) & 0x55555555u;
  return lo | (hi << 1);
}

// The reference's 10-mers as a compact hash table (open addressing, 16-byte slots {key, pos0, count - 1, pos1}, load <= 1/4):
// a megabyte for a mitochondrion, so that the nine look-ups of a read stay in the L2 (the direct-addressed table of
// diag_filter.h is 20 MB and every look-up a trip to memory).  Positions 3 and 4 of a repeated 10-mer sit in `ovf`.
struct KmerHash {
  const uint32_t* slot;    // [4 * (mask + 1)], all ones = empty
  const int32_t* ovf;      // [2 * (mask + 1)]
  uint32_t mask;
  int32_t shift;           // 32 - log2(slots)
  int32_t wild;            // reference 10-mers with up to this many N are in the table under every spelling (0: none with N)
};
constexpr uint32_t KH_EMPTY = 0xFFFFFFFFu;
// behind a batch of independent loads: nothing is scheduled across, so that the loads are all issued before the first is waited for
#if defined(__HIP_DEVICE_COMPILE__)
#define BX_LOADS_ISSUED() __builtin_amdgcn_sched_barrier(0)
#else
#define BX_LOADS_ISSUED() do { } while (0)
#endif
MIA_HD inline uint32_t kh_slots_for(int64_t n_codes) { uint32_t s = 1024; while ((int64_t)s < 4 * n_codes) s <<= 1; return s; }
MIA_HD inline int kh_shift_for(uint32_t slots) { int b = 0; while ((1u << b) < slots) b++; return 32 - b; }
MIA_HD inline uint32_t kh_home(const KmerHash& kh, uint32_t idx) { return (idx * 2654435761u) >> kh.shift; }
// occurrences of 10-mer idx (DF_KCAP + 1 = more than the table keeps), the first DF_KCAP positions in ps; e = its home slot, already loaded
MIA_HD inline int kh_resolve(const KmerHash& kh, uint32_t idx, uint32_t h, uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, int32_t* ps) {
  for (int probe = 0; probe < 32; probe++) {
    if (e0 == KH_EMPTY) return 0;
    if (e0 == idx) {
      const int cnt = (int)(e2 + 1u);
      ps[0] = (int32_t)e1; ps[1] = (int32_t)e3;
      if (cnt > 2) { ps[2] = kh.ovf[2 * h]; ps[3] = kh.ovf[2 * h + 1]; }
      return cnt > DF_KCAP ? DF_KCAP + 1 : cnt;
    }
    h = (h + 1) & kh.mask;
    const uint32_t* e = kh.slot + 4 * (size_t)h;
    e0 = e[0]; e1 = e[1]; e2 = e[2]; e3 = e[3];
  }
  return DF_KCAP + 1;                                // a crowded neighbourhood: not part of the pigeonhole
}
// table entries a reference makes (host; the slots must outnumber them two to one)
inline int64_t kh_wild_entries(const uint8_t* codes, int64_t n_codes, int wild) {
  int64_t e = 0;
  int k = 0;
  for (int64_t p = 0; p < n_codes; p++) {
    k += codes[p] > 3;
    if (p >= DF_K) k -= codes[p - DF_K] > 3;
    if (p >= DF_K - 1 && k <= wild) e += (int64_t)1 << (2 * k);
  }
  return e;
}
inline uint32_t kh_slots_for_entries(int64_t n_codes, int64_t entries) {
  uint32_t s = kh_slots_for(n_codes);
  while ((int64_t)s < 2 * entries && s < (1u << 30)) s <<= 1;
  return s;
}
// host-side insert (the tests); the device inserts with atomics (k_kmer_hash)
inline void kh_insert_host(uint32_t* slot, int32_t* ovf, uint32_t mask, int shift, uint32_t idx, int32_t p) {
  uint32_t h = (idx * 2654435761u) >> shift;
  while (slot[4 * (size_t)h] != KH_EMPTY && slot[4 * (size_t)h] != idx) h = (h + 1) & mask;
  uint32_t* e = slot + 4 * (size_t)h;
  e[0] = idx;
  const uint32_t c = ++e[2];                        // (starts at all ones)
  if (c == 0) e[1] = (uint32_t)p; else if (c == 1) e[3] = (uint32_t)p; else if (c < 4) ovf[2 * (size_t)h + c - 2] = p;
}
// every spelling of the 10-mer at p (the tests; k_kmer_hash on the device)
inline void kh_insert_wild_host(uint32_t* slot, int32_t* ovf, uint32_t mask, int shift, const uint8_t* codes, int64_t n_codes, int64_t p, int wild) {
  uint32_t idx;
  uint64_t npos;
  const int k = kmer_wild_at(codes, n_codes, p, &idx, &npos);
  if (k < 0 || k > wild) return;
  for (uint32_t x = 0; x < (1u << (2 * k)); x++) kh_insert_host(slot, ovf, mask, shift, kmer_wild_key(idx, npos, k, x), (int32_t)p);
}

// mode of a planned read
constexpr int BX_NONE = 0;     // not planned: the full-window kernels take it
constexpr int BX_DONE = 1;     // finished by the plan: pure diagonal dstar, score = U - b0
constexpr int BX_VALUES = 2;   // values-only DP, then the check  best == U - b0 at diagonal dstar
constexpr int BX_TRACE = 3;    // straight to the trace DP
struct BxPlan { int mode, d0, w, dstar, b0, edge; };
// why a read was not planned (statistics only)
enum { BXF_READ = 1, BXF_WINDOW, BXF_BLOCKS, BXF_SPAN, BXF_PATH, BXF_BUDGET, BXF_WIDTH, BXF_KINDS };

// the anchors of a read and what they imply, before any loss is summed
struct BxAnchors { int fail, a_lo, a_hi, d_first, d_last, budget, t_lo, t_hi, l_out, s_un, r_head, r_tail, rescue, fine, fc1, fc2, fmax; };
// fine blocks only: fc1 / fc2 = diagonals outside the kept range on which at least one / two blocks are clean (fc2 < 0: some diagonal
// holds three or more, or the window is too wide to tell); fmax = the dearest block
// r_head: first row of the first block with a kept anchor; r_tail: first row behind the last such block; rescue: bx_rescue is to be tried   // l_out < 0: every anchor counts; s_un: dl of the blocks that occur nowhere in the window

// Where the read's blocks occur inside the window.  sc holds the read's planes.
template <int NW>
MIA_HD inline void bx_anchors(const DiagScan<NW>& sc, const KmerHash& kh, const uint32_t* pw, int s, int len1, int len2, int st, const BxTab& T, BxAnchors* an) {
  constexpr int NB = bx_nb_max<NW>();      // (a read of NW words has at most this many blocks)
  (void)pw;                                // (the 10-mers come from the planes in sc)
  const int R = len2 - 1, nb_cut = bx_blocks_of(len2);
  // Every trip to the table is made for all blocks at once: the home slots of the nine blocks are nine loads in flight, and so is
  // every further round of the open-addressing walk (the blocks whose home slot holds another 10-mer move on together), and the
  // overflow positions of the repeated 10-mers.  One block after the other -- each with its own probe loop -- a WAVEFRONT walked
  // nine loops of dependent loads whenever any of its lanes had a collision in that block, i.e. always: a few dozen trips to
  // the L2 per read, most of k_bx_plan's time.  The blocks' dl come in with the first round (they are wanted by every scan below).
  // Rounds in straight-line code (selects, no branches; a block that is through loads its own slot again, which leaves its entry
  // where it is): with a branch per block the loads of a round were issued one by one, each behind a wait for the one before.
  const int16_t* dl = T.dl + (st * (MAX_READ + 1) + len2) * BX_BLOCKS;
  int32_t cn[NB], ps[NB][DF_KCAP], dlv[NB];
  uint32_t kidx[NB], hcur[NB], ke[NB][4];
  uint32_t open = 0, found = 0;                      // bit b: block b still walks the table / its 10-mer is in the table (entry in ke[b])
#pragma unroll
  for (int b = 0; b < NB; b++) {
    kidx[b] = 0; hcur[b] = 0;
    if (b < nb_cut) {
      kidx[b] = bx_kmer_planes<NW>(sc, bx_block_row(b, len2, nb_cut));
      hcur[b] = kh_home(kh, kidx[b]);
      open |= 1u << b;
    }
  }
  for (int probe = 0; ; probe++) {                  // (kh_resolve's walk: 32 slots looked at, then "a crowded neighbourhood")
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const uint32_t* e = kh.slot + 4 * (size_t)hcur[b];
#pragma unroll
      for (int k = 0; k < 4; k++) ke[b][k] = e[k];
    }
    BX_LOADS_ISSUED();
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const bool o = ((open >> b) & 1u) != 0u, emp = ke[b][0] == KH_EMPTY, hit = ke[b][0] == kidx[b];
      const bool fin = o && (emp || hit);
      found |= (o && hit && !emp) ? 1u << b : 0u;
      open &= fin ? ~(1u << b) : ~0u;
      hcur[b] = (o && !fin) ? ((hcur[b] + 1) & kh.mask) : hcur[b];
    }
    if (!open || probe == 31) break;
  }
  uint32_t more = 0;                                 // bit b: overflow positions to fetch
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const bool got = ((found >> b) & 1u) != 0u;
    const int cnt = (int)(ke[b][2] + 1u);
    cn[b] = got ? (cnt > DF_KCAP ? DF_KCAP + 1 : cnt) : ((b < nb_cut && !((open >> b) & 1u)) ? 0 : DF_KCAP + 1);
    ps[b][0] = got ? (int32_t)ke[b][1] : 0; ps[b][1] = got ? (int32_t)ke[b][3] : 0; ps[b][2] = 0; ps[b][3] = 0;
    more |= (got && cnt > 2) ? 1u << b : 0u;
  }
#pragma unroll
  for (int b = 0; b < NB; b++) dlv[b] = dl[b];
  if (more) {
#pragma unroll
    for (int b = 0; b < NB; b++)
      if ((more >> b) & 1u) { ps[b][2] = kh.ovf[2 * (size_t)hcur[b]]; ps[b][3] = kh.ovf[2 * (size_t)hcur[b] + 1]; }
  }
  BX_LOADS_ISSUED();
  int nb = 0, a_lo = 1 << 20, a_hi = -(1 << 20), d_first = 0, d_last = 0, budget = -1, b_first = 0, l_out = -1, s_un = 0, b_lo_any = 0, b_hi_any = 0;
  bool any = false;
  // the anchors whose diagonal lies in [m_lo, m_hi]: their extent, the first and the last in block order; l_out (if asked
  // for) = what a path loses at least that crosses NONE of them cleanly: dl of every block without an anchor outside
  auto scan = [&](int m_lo, int m_hi, bool want_out) {
    nb = 0; a_lo = 1 << 20; a_hi = -(1 << 20); d_first = 0; d_last = 0; budget = -1; b_first = 0; any = false; s_un = 0;
    int lo_sum = 0;
#pragma unroll
    for (int b = 0; b < NB; b++) {
      if (cn[b] > DF_KCAP) continue;                   // no such block, or an overloaded 10-mer: not part of the pigeonhole
      nb++;
      budget += dlv[b];
      const int o = bx_block_row(b, len2, nb_cut);
      bool outside = false, nowhere = true;
#pragma unroll
      for (int k = 0; k < DF_KCAP; k++) {
        if (k >= cn[b]) continue;
        const int d = ps[b][k] - o - s;                         // diagonal in window coordinates
        if (d < -R || d > len1 - 1) continue;                   // not a place inside this window
        nowhere = false;
        if (d < m_lo || d > m_hi) { outside = true; continue; }
        if (!any) { d_first = d; any = true; b_lo_any = b; }
        b_hi_any = b;
        if (d == d_first) b_first = b;                          // (the last block that has an anchor on d_first)
        d_last = d;
        if (d < a_lo) a_lo = d;
        if (d > a_hi) a_hi = d;
      }
      if (!outside) lo_sum += dlv[b];
      if (nowhere) s_un += dlv[b];                       // no path crosses this block cleanly
    }
    l_out = want_out ? lo_sum : -1;
  };
  scan(-(1 << 20), 1 << 20, false);
  if (any && nb >= BX_MIN_BLOCKS && a_hi - a_lo > BX_CLUSTER_TRIGGER) {
    // Anchors far apart: a 10-mer of the read that also occurs elsewhere in the window (one read in a few hundred), or a long
    // indel.  Keep the cluster around the MEDIAN of the blocks' first anchors (the true diagonals hold the majority) and
    // drop the rest -- soundly: a path through a kept anchor strays at most g diagonals from it, whatever else it visits,
    // and a path that crosses no kept anchor cleanly breaks every block that has no anchor elsewhere, i.e. loses >= l_out,
    // which bx_finish requires to exceed the loss B0 of the path it writes down.
    int v[NB];
    int c = 0;
#pragma unroll
    for (int b = 0; b < NB; b++) {
      v[b] = 1 << 20;
      if (cn[b] > DF_KCAP) continue;
      const int o = bx_block_row(b, len2, nb_cut);
      bool got = false;
#pragma unroll
      for (int k = 0; k < DF_KCAP; k++) {
        if (k >= cn[b] || got) continue;
        const int d = ps[b][k] - o - s;
        if (d < -R || d > len1 - 1) continue;
        v[b] = d; got = true;
      }
      c += got ? 1 : 0;
    }
#pragma unroll
    for (int pass = 0; pass < NB; pass++)
#pragma unroll
      for (int k = pass & 1; k + 1 < NB; k += 2) { const int lo = v[k] < v[k + 1] ? v[k] : v[k + 1], hi = v[k] < v[k + 1] ? v[k + 1] : v[k]; v[k] = lo; v[k + 1] = hi; }
    int med = v[0];
    const int mi = (c - 1) >> 1;
#pragma unroll
    for (int k = 1; k < NB; k++) if (k == mi) med = v[k];
    scan(med - BX_CLUSTER_RADIUS, med + BX_CLUSTER_RADIUS, true);
  }
  an->fail = 0;
  if (nb < BX_MIN_BLOCKS || !any) { an->fail = BXF_BLOCKS; return; }
  if (a_hi - a_lo >= T.maxw) { an->fail = BXF_SPAN; return; }
  // keep the written-down path inside the window
  if (d_first < 0 || d_first > len1 - len2 || d_last < 0 || d_last > len1 - len2) { an->fail = BXF_PATH; return; }
  an->a_lo = a_lo; an->a_hi = a_hi; an->d_first = d_first; an->d_last = d_last; an->budget = budget; an->l_out = l_out; an->s_un = s_un;
  an->r_head = bx_block_row(b_lo_any, len2, nb_cut); an->r_tail = bx_block_row(b_hi_any, len2, nb_cut) + DF_K; an->rescue = 0; an->fine = 0;
  an->fc1 = 0; an->fc2 = 0; an->fmax = 0;
  an->t_lo = 1; an->t_hi = R;
  if (d_first != d_last) {
    // the switch row is looked for between the last block anchored on d_first and the first one after it anchored on d_last
    int b_last = nb_cut - 1;
    bool found = false;
#pragma unroll
    for (int b = 0; b < NB; b++) {
      if (cn[b] > DF_KCAP || b <= b_first || found) continue;
      const int o = bx_block_row(b, len2, nb_cut);
#pragma unroll
      for (int k = 0; k < DF_KCAP; k++)
        if (k < cn[b] && ps[b][k] - o - s == d_last) { b_last = b; found = true; }
    }
    an->t_lo = bx_block_row(b_first, len2, nb_cut) + DF_K;
    an->t_hi = found ? bx_block_row(b_last, len2, nb_cut) : R;
    if (an->t_lo < 1) an->t_lo = 1;
  }
}

// bit q of a multiword mask
template <int NW>
MIA_HD inline int bx_bit(const uint64_t* m, int q) {
  uint64_t w = m[0];
#pragma unroll
  for (int j = 1; j < NW; j++) if ((q >> 6) == j) w = m[j];
  return (int)((w >> (q & 63)) & 1ull);
}
// mismatches in rows [from, to) of a mask
template <int NW>
MIA_HD inline int bx_count(const uint64_t* m, int from, int to) {
  int n = 0;
#pragma unroll
  for (int j = 0; j < NW; j++) {
    const int lo = from - 64 * j, hi = to - 64 * j;
    if (hi <= 0 || lo >= 64) continue;
    uint64_t w = m[j];
    if (lo > 0) w &= ~0ull << lo;
    if (hi < 64) w &= (1ull << hi) - 1ull;
    n += df_popc(w);
  }
  return n;
}

// loss of the rows in [from, to) that lose anything on the diagonal sc is on (m = those rows: mismatches and N columns,
// bx_loss_rows).  *nfail counts the
// rows q among them behind which dyn_prog's "new start" branch can fire on this path: D(q) < -P(q+2), with D(q) >= (q+1) min M
// - loss so far.  That branch DROPS the substitution score of the row it starts in (src/mia.c:916-917), so the value the
// recurrence reaches along the path is lower than the path's own by less than max M each time it fires (and it can only
// fire behind a mismatch row; the test counts earlier firings against the later ones).  nfail == 0: the diagonal proof of
// align_body_quad_plain.h.
template <int NW>
MIA_HD inline int bx_rows_loss(const DiagScan<NW>& sc, const uint64_t* m, int from, int to, int len2, int st, const BxTab& T, int b0, int* nfail) {
#pragma unroll
  for (int j = 0; j < NW; j++) {
    uint64_t w = m[j];
    const int lo = from - 64 * j, hi = to - 64 * j;
    if (hi <= 0 || lo >= 64) continue;
    if (lo > 0) w &= ~0ull << lo;
    if (hi < 64) w &= (1ull << hi) - 1ull;
    while (w) {
      const int k = df_ctz(w), q = j * 64 + k;
      w &= w - 1;
      const int b = (int)(((sc.rlo[j] >> k) & 1ull) | (((sc.rhi[j] >> k) & 1ull) << 1));
      const int i = (int)(((sc.clo[j] >> k) & 1ull) | (((sc.chi[j] >> k) & 1ull) << 1));
      const int at = (st * 31 + sm_depth(q, len2)) * 4 + b;
      b0 += ((sc.cok[j] >> k) & 1ull) ? T.loss[at * 4 + i] : T.loss[BX_LOSS_N + at];
      if (b0 + *nfail * T.max_m > (q + 1) * T.min_m + GOP + GEP * (q + 2)) (*nfail)++;
    }
  }
  return b0;
}

// the most set bits of m[] that lie in one stretch holding at most `zeros` clear ones between its first and last set bit
template <int NW>
MIA_HD inline int bx_ones_span(const uint64_t* m, int zeros) {
  uint64_t hi[NW], lo[NW];                 // bits still to come at the front / not yet dropped at the back
#pragma unroll
  for (int j = 0; j < NW; j++) { hi[j] = m[j]; lo[j] = m[j]; }
  auto first = [&](const uint64_t* w) -> int {
    int p = -1;
#pragma unroll
    for (int j = NW - 1; j >= 0; j--) if (w[j]) p = j * 64 + df_ctz(w[j]);
    return p;
  };
  auto drop = [&](uint64_t* w) {
    bool done = false;
#pragma unroll
    for (int j = 0; j < NW; j++) if (!done && w[j]) { w[j] &= w[j] - 1; done = true; }
  };
  int best = 0, cnt = 0;
  for (int pb = first(hi); pb >= 0; pb = first(hi)) {
    drop(hi);
    cnt++;
    while ((pb - first(lo)) - (cnt - 1) > zeros) { drop(lo); cnt--; }
    if (cnt > best) best = cnt;
  }
  return best;
}

// END INDELS.  An indel within a dozen rows of a read end leaves that end without a block of its own: all anchors lie on
// one diagonal, the written-down path is that diagonal, and it pays a substitution for most of the shifted rows -- B0
// exceeds the budget and the read (three in a hundred with the synthetic indel rate: nearly all the plan gives up on) went
// to the full-window kernels.  B0 only has to be the loss of SOME valid path: look at the unanchored head and tail on
// the diagonals one to three off, and if one of them explains that end much better, write the path down with a gap there
// (bx_finish's two-diagonal form finds the switch row).  The anchors, and with them the band's centre, stay what they
// are: the proof is the same, only its bound got better.  false: nothing to gain.
template <int NW>
MIA_HD inline bool bx_rescue(DiagScan<NW>& sc, const RefPlanes& rp, BxAnchors& an, int s, int len1, int len2) {
  const int R = len2 - 1, dstar = an.d_first;
  sc.seek(rp, (int64_t)s + dstar);
  uint64_t m0[NW];
#pragma unroll
  for (int j = 0; j < NW; j++) m0[j] = sc.mis(j);
  const int head = an.r_head > 0 ? bx_count<NW>(m0, 0, an.r_head) : 0, tail = an.r_tail < len2 ? bx_count<NW>(m0, an.r_tail, len2) : 0;
  const bool at_tail = tail >= head;
  const int here = at_tail ? tail : head;
  if (here < 3) return false;
  // which diagonal?  The six rows at the very end of the read (its start) sit on it: one seek three columns below d*, then
  // the planes slide along (DiagScan::advance) -- the mismatches of those rows on d* - 3 .. d* + 3
  const int e_lo = at_tail ? (len2 - 6 > an.r_tail ? len2 - 6 : an.r_tail) : 0, e_hi = at_tail ? len2 : (an.r_head < 6 ? an.r_head : 6);
  if (e_hi - e_lo < 4) return false;
  const int crude0 = bx_count<NW>(m0, e_lo, e_hi);
  int pick = 0, pick_c = crude0;
  {
    DiagScan<NW> s2 = sc;
    s2.seek(rp, (int64_t)s + dstar - 3);
    for (int k = 0; k < 7; k++) {
      const int sh = k - 3, d = dstar + sh;
      if (sh != 0 && d >= 0 && d <= len1 - len2) {                // (the written-down path stays inside the window)
        uint64_t m[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) m[j] = s2.mis(j);
        const int c = bx_count<NW>(m, e_lo, e_hi);
        if (c < pick_c) { pick_c = c; pick = sh; }
      }
      if (k < 6) s2.advance(rp, (int64_t)s + dstar - 3 + k + 1);
    }
  }
  if (pick == 0 || pick_c + 2 > crude0) return false;
  // that end's mismatches with the best switch row: rows below it on the first diagonal, `skip` rows left out where the
  // second diagonal is the lower one (an insert), the rest on the second
  int best = here, best_sh = 0;
  {
    const int sh = pick;
    DiagScan<NW> s2 = sc;
    s2.seek(rp, (int64_t)s + dstar + sh);
    uint64_t m[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) m[j] = s2.mis(j);
    int c = here;
    if (at_tail) {
      const int skip = sh < 0 ? -sh : 0, t0 = an.r_tail, t1 = len2 - skip;
      if (t1 >= t0) {
        int cur = bx_count<NW>(m, t0 + skip, len2);
        c = cur;
        for (int t = t0 + 1; t <= t1; t++) { cur += bx_bit<NW>(m0, t - 1) - bx_bit<NW>(m, t - 1 + skip); if (cur < c) c = cur; }
      }
    } else {
      const int skip = sh > 0 ? sh : 0, t1 = an.r_head - skip;
      if (t1 >= 1) {
        int cur = bx_bit<NW>(m, 0) + bx_count<NW>(m0, 1 + skip, an.r_head);
        c = cur;
        for (int t = 2; t <= t1; t++) { cur += bx_bit<NW>(m, t - 1) - bx_bit<NW>(m0, t - 1 + skip); if (cur < c) c = cur; }
      }
    }
    if (c < best) { best = c; best_sh = sh; }
  }
  if (best_sh == 0 || best + 2 > here) return false;
  if (at_tail) { an.d_last = dstar + best_sh; an.t_lo = an.r_tail > 1 ? an.r_tail : 1; an.t_hi = R; }
  else { an.d_first = dstar + best_sh; an.t_lo = 1; an.t_hi = an.r_head > 1 ? an.r_head : 1; }
  return true;
}

// WINDOW-WIDE N CREDIT: the fewest credited N columns in any stretch of len2 consecutive columns of the window [s, s + len1) --
// what every path crosses at least, by a row or inside a column gap (bx_make_tables: when that may be added to the pigeonhole
// sum).  Credited = not part of any ten consecutive columns that hold more than BX_WILD N columns: a clean crossing of a
// block over such a place is in no 10-mer table, counts as a break (dl_b <= (BX_WILD + 1) min lambda) and must not be paid
// for twice (mt311: 481 such places, one window in two holds one).  Windows beyond 64 BX_NWW columns: 0 (no credit).
constexpr int BX_NWW = 5;
// all_count: every N column is credited (the caller's clean places are found by comparison, not looked up: bx_fine_anchors).
MIA_HD inline int bx_window_nmin(const RefPlanes& rp, int s, int len1, int len2, bool all_count = false) {
  if (len1 > 64 * BX_NWW || len2 > len1) return 0;
  uint64_t n[BX_NWW + 1];
  bool any = false;
#pragma unroll
  for (int k = 0; k < BX_NWW; k++) {
    const int lo = 64 * k;
    uint64_t w = 0;
    if (lo < len1) {
      const int64_t bit = (int64_t)s + lo + PLANE_LEAD;
      const int64_t q = bit >> 6;
      const int b = (int)(bit & 63);
      w = ~((rp.ok[q] >> b) | ((rp.ok[q + 1] << 1) << (63 - b)));
      if (len1 - lo < 64) w &= (1ull << (len1 - lo)) - 1ull;
    }
    n[k] = w;
    any = any || w != 0;
  }
  n[BX_NWW] = 0;
  if (!any) return 0;
  // N columns in [p, p + 10) for every p at once: a bit-sliced counter over the ten shifted masks; over = "four or more"
  uint64_t c0[BX_NWW], c1[BX_NWW], over[BX_NWW];
#pragma unroll
  for (int k = 0; k < BX_NWW; k++) { c0[k] = 0; c1[k] = 0; over[k] = 0; }
#pragma unroll
  for (int t = 0; t < DF_K; t++) {
#pragma unroll
    for (int k = 0; k < BX_NWW; k++) {
      const uint64_t x = t ? (n[k] >> t) | (n[k + 1] << (64 - t)) : n[k];
      const uint64_t k0 = c0[k] & x;
      c0[k] ^= x;
      const uint64_t k1 = c1[k] & k0;
      c1[k] ^= k0;
      over[k] |= k1;                                    // a carry out of the two-bit counter: the count reached four (and stays flagged)
    }
  }
  // a column belongs to an overloaded place if one starts at most nine columns before it
  uint64_t cr[BX_NWW];
#pragma unroll
  for (int k = 0; k < BX_NWW; k++) {
    uint64_t e = over[k];
#pragma unroll
    for (int t = 1; t < DF_K; t++) e |= (over[k] << t) | (k ? over[k - 1] >> (64 - t) : 0ull);
    cr[k] = all_count ? n[k] : n[k] & ~e;
  }
  auto below = [&](int end) -> int {                     // credited columns at positions < end
    int c = 0;
#pragma unroll
    for (int k = 0; k < BX_NWW; k++) {
      const int hi = end - 64 * k;
      if (hi <= 0) continue;
      c += df_popc(hi >= 64 ? cr[k] : cr[k] & ((1ull << hi) - 1ull));
    }
    return c;
  };
  // the count over [c, c + len2) changes only where c passes a credited column: the minimum is at c = 0 or just behind one
  int best = below(len2), seen = 0;
#pragma unroll
  for (int k = 0; k < BX_NWW; k++) {
    uint64_t w = cr[k];
    while (w) {
      const int c = 64 * k + df_ctz(w) + 1;
      w &= w - 1;
      seen++;
      if (c + len2 <= len1) {
        const int v = below(c + len2) - seen;
        if (v < best) best = v;
      }
    }
  }
  return best;
}
static_assert(BX_WILD == 3, "bx_window_nmin flags four N columns within ten");

// FINE BLOCKS -- a second chance for the reads whose loss B0 exceeds what the nine 10-mers can vouch for (sum of dl: 4 500 with
// the ancient matrix -- three ordinary substitutions at 1 000 each beside the N columns of mt311; a fifth of configs[2]'s reads in
// their first iteration, one in fifty later on).  The pigeonhole holds for ANY disjoint blocks: with blocks of BX_FQ rows
// there are len2 / BX_FQ of them (16 for 100 bases, sum of fdl about 8 300).  Their clean places are not looked up but
// found: the read's planes against every diagonal of the window (one shift per diagonal, DiagScan::advance), a block being
// clean where none of its rows is a definite mismatch (N columns are clean by definition) and all of them lie inside the
// window.  Six-mers occur by chance (one block in twenty somewhere in a 200-column window): clean places within
// BX_FINE_RADIUS of the 10-mer anchors' diagonals are kept (the band must hold them), all others are set aside the way
// bx_anchors sets stray anchors aside -- a path that crosses no kept place cleanly breaks every block that has none
// elsewhere and loses at least l_out, which bx_finish requires to exceed B0.  The written-down path (d_first, d_last, switch
// rows) stays what the 10-mers gave.  an: in/out.  false: no gain possible (tables not usable, too few blocks).
// How the clean places are found: the window as four bit planes over its COLUMNS (bit p of plane x: column p holds base x or an
// N; nothing beyond the window), and for block b the columns p at which it is clean, M_b(p) = AND over its six rows k of
// plane[read base of row o_b + k](p + k) -- six shift-and-AND steps over the window's words per block, whatever the number of
// diagonals (walking the ~290 diagonals of a window one by one with the read's row planes cost ten times as much).
// In parts, so that a kernel can deal the blocks of one read to several lanes: bx_fine_scan for blocks b0, b0 + step, ...,
// the lanes' results combined (OR, min, max), bx_fine_sums on one of them.
constexpr int BX_FWW = 5;               // words of a window's column planes: windows up to 320 columns
template <int NW>
MIA_HD inline bool bx_fine_usable(int len1, int len2, int st, const BxTab& T) {
  const int nbq = bx_fine_blocks_of(len2);
  return nbq >= 2 * BX_MIN_BLOCKS && nbq <= 64 && len1 <= 64 * BX_FWW && T.loss[BX_LOSS_FDL + st * 31 + 15] > 0;
}
struct BxWinPlanes { uint64_t p[4][BX_FWW]; };
MIA_HD inline void bx_win_planes(const RefPlanes& rp, int s, int len1, BxWinPlanes* wp) {
#pragma unroll
  for (int k = 0; k < BX_FWW; k++) {
    const int lo = 64 * k;
    uint64_t l = 0, h = 0, ok = ~0ull, in = 0;
    if (lo < len1) {
      const int64_t bit = (int64_t)s + lo + PLANE_LEAD;
      const int64_t q = bit >> 6;
      const int b = (int)(bit & 63);
      l = (rp.lo[q] >> b) | ((rp.lo[q + 1] << 1) << (63 - b));
      h = (rp.hi[q] >> b) | ((rp.hi[q + 1] << 1) << (63 - b));
      ok = (rp.ok[q] >> b) | ((rp.ok[q + 1] << 1) << (63 - b));
      in = len1 - lo < 64 ? (1ull << (len1 - lo)) - 1ull : ~0ull;
    }
    wp->p[0][k] = ((~l & ~h) | ~ok) & in;
    wp->p[1][k] = ((l & ~h) | ~ok) & in;
    wp->p[2][k] = ((~l & h) | ~ok) & in;
    wp->p[3][k] = ((l & h) | ~ok) & in;
  }
}
// blocks b0, b0 + step, ... of the read in sc: bit b of *in / *out = block b is clean somewhere on the kept diagonals k_lo .. k_hi /
// somewhere else in the window; *a_lo / *a_hi take in the kept diagonals that hold a clean block
constexpr int BX_FDW = 8;               // words of the per-diagonal counters: diagonals -(len2 - 6) .. len1 - 6, at most 512 of them
// ge[0..BX_FDW) / ge[BX_FDW..2 BX_FDW) / ge[2 BX_FDW..3 BX_FDW): diagonals outside the kept range on which at least one / two / three of
// the blocks seen so far are clean (bit d + len2 - BX_FQ); nullptr: not kept
template <int NW>
MIA_HD inline void bx_fine_scan(const DiagScan<NW>& sc, const BxWinPlanes& wp, int len1, int len2, int k_lo, int k_hi, int b0, int step,
                                uint64_t* in, uint64_t* out, int* a_lo, int* a_hi, uint64_t* ge = nullptr) {
  const int nbq = bx_fine_blocks_of(len2);
  for (int b = b0; b < nbq; b += step) {
    const int o = bx_fine_block_row(b, len2, nbq);
    uint64_t A[BX_FWW];
#pragma unroll
    for (int k = BX_FQ - 1; k >= 0; k--) {
      const int row = o + k;
      uint64_t wl = sc.rlo[0], wh = sc.rhi[0];
#pragma unroll
      for (int j = 1; j < NW; j++) if ((row >> 6) == j) { wl = sc.rlo[j]; wh = sc.rhi[j]; }
      const int x = (int)(((wl >> (row & 63)) & 1ull) | (((wh >> (row & 63)) & 1ull) << 1));
#pragma unroll
      for (int w = 0; w < BX_FWW; w++) {
        const uint64_t pl = x == 0 ? wp.p[0][w] : (x == 1 ? wp.p[1][w] : (x == 2 ? wp.p[2][w] : wp.p[3][w]));
        if (k == BX_FQ - 1) A[w] = pl;
        else A[w] = pl & ((A[w] >> 1) | (w + 1 < BX_FWW ? A[w + 1] << 63 : 0ull));      // (words ascend: A[w + 1] still holds the previous step)
      }
    }
    // kept: columns k_lo + o .. k_hi + o
    const int p_lo = k_lo + o < 0 ? 0 : k_lo + o, p_hi = k_hi + o;
    bool is_in = false, is_out = false;
    int first = 1 << 20, last = -(1 << 20);
#pragma unroll
    for (int w = 0; w < BX_FWW; w++) {
      const int lo = p_lo - 64 * w, hi = p_hi + 1 - 64 * w;               // bits lo .. hi - 1 of this word are kept columns
      uint64_t R = 0;
      if (hi > 0 && lo < 64) {
        R = ~0ull;
        if (lo > 0) R &= ~0ull << lo;
        if (hi < 64) R &= (1ull << hi) - 1ull;
      }
      const uint64_t ai = A[w] & R, ao = A[w] & ~R;
      A[w] = ao;                                          // (from here on: the clean places outside the kept range)
      if (ai) {
        is_in = true;
        const int f = 64 * w + df_ctz(ai), l = 64 * w + 63 - df_clz(ai);
        if (f < first) first = f;
        if (l > last) last = l;
      }
      if (ao) is_out = true;
    }
    if (is_in) {
      *in |= 1ull << b;
      if (first - o < *a_lo) *a_lo = first - o;
      if (last - o > *a_hi) *a_hi = last - o;
    }
    if (is_out) *out |= 1ull << b;
    if (ge && is_out) {
      // the block's outside places by DIAGONAL (column p is diagonal p - o: bit p + len2 - BX_FQ - o), added to the saturating
      // per-diagonal counters: first the bit shift, then the shift by whole words
      const int sh = len2 - BX_FQ - o, bs = sh & 63, ws = sh >> 6;
      uint64_t S[BX_FWW + 1];
#pragma unroll
      for (int w = 0; w <= BX_FWW; w++) {
        const uint64_t lo = w < BX_FWW ? A[w] : 0ull, below = w > 0 ? A[w - 1] : 0ull;
        S[w] = (lo << bs) | (bs ? below >> (64 - bs) : 0ull);
      }
#pragma unroll
      for (int dw = 0; dw < BX_FDW; dw++) {
        uint64_t D = 0;
#pragma unroll
        for (int w = 0; w <= BX_FWW; w++) if (dw - ws == w) D = S[w];
        const uint64_t c1 = ge[dw] & D, c2 = ge[BX_FDW + dw] & c1;
        ge[dw] |= D; ge[BX_FDW + dw] |= c1; ge[2 * BX_FDW + dw] |= c2;
      }
    }
  }
}
template <int NW>
MIA_HD inline void bx_fine_sums(uint64_t in, uint64_t out, int a_lo, int a_hi, int len2, int st, const BxTab& T, BxAnchors* an, const uint64_t* ge = nullptr) {
  const int nbq = bx_fine_blocks_of(len2);
  int budget = -1, l_out = 0, s_un = 0, fmax = 0;
  an->fc1 = 0; an->fc2 = -1;
  if (ge) {
    int c1 = 0, c2 = 0;
    bool three = false;
#pragma unroll
    for (int dw = 0; dw < BX_FDW; dw++) { c1 += df_popc(ge[dw]); c2 += df_popc(ge[BX_FDW + dw]); three = three || ge[2 * BX_FDW + dw] != 0; }
    an->fc1 = c1; an->fc2 = three ? -1 : c2;
  }
  for (int b = 0; b < nbq; b++) {
    const int o = bx_fine_block_row(b, len2, nbq);
    int v = 1 << 20;
    for (int r = o; r < o + BX_FQ; r++) { const int f = T.loss[BX_LOSS_FDL + st * 31 + sm_depth(r, len2)]; if (f < v) v = f; }
    const int is_in = (int)((in >> b) & 1ull), is_out = (int)((out >> b) & 1ull);
    budget += v;
    if (v > fmax) fmax = v;
    if (!is_out) l_out += v;
    if (!is_in && !is_out) s_un += v;
  }
  an->budget = budget; an->l_out = l_out; an->s_un = s_un; an->a_lo = a_lo; an->a_hi = a_hi; an->fine = 1; an->fmax = fmax;
}
template <int NW>
MIA_HD inline bool bx_fine_anchors(DiagScan<NW>& sc, const RefPlanes& rp, int s, int len1, int len2, int st, const BxTab& T, BxAnchors* an) {
  if (!bx_fine_usable<NW>(len1, len2, st, T)) return false;
  BxWinPlanes wp;
  bx_win_planes(rp, s, len1, &wp);
  uint64_t in = 0, out = 0;
  int a_lo = an->a_lo, a_hi = an->a_hi;
  uint64_t ge[3 * BX_FDW];
#pragma unroll
  for (int k = 0; k < 3 * BX_FDW; k++) ge[k] = 0;
  const bool counted = len1 + len2 <= 64 * BX_FDW;
  bx_fine_scan<NW>(sc, wp, len1, len2, an->a_lo - BX_FINE_RADIUS, an->a_hi + BX_FINE_RADIUS, 0, 1, &in, &out, &a_lo, &a_hi, counted ? ge : nullptr);
  bx_fine_sums<NW>(in, out, a_lo, a_hi, len2, st, T, an, counted ? ge : nullptr);
  return true;
}

// the rows of word j that lose anything on the diagonal sc is on: definite mismatches and N columns
template <int NW>
MIA_HD inline uint64_t bx_loss_rows(const DiagScan<NW>& sc, int j) { return sc.mis(j) | (~sc.cok[j] & sc.rows[j]); }

#ifdef BX_DIAG
static int bx_diag[8];       // (tests / tools only: what the last bx_finish saw)
#endif
// B0 (the loss of one valid path) and what follows from it
// PATHS: 0 = whatever the anchors say, 1 = the caller knows d_first == d_last, 2 = the caller knows they differ
template <int NW, int PATHS = 0>
MIA_HD inline void bx_finish(DiagScan<NW>& sc, const RefPlanes& rp, const BxAnchors& an, int s, int len1, int len2, int st, const BxTab& T, BxPlan* out) {
  out->mode = BX_NONE;
  const int R = len2 - 1, d_first = an.d_first, d_last = PATHS == 1 ? an.d_first : an.d_last;
  uint64_t m1[NW];
  sc.seek(rp, (int64_t)s + d_first);
#pragma unroll
  for (int j = 0; j < NW; j++) m1[j] = bx_loss_rows<NW>(sc, j);
  int b0 = 0, nfail = 0;
  if (PATHS == 1 || (PATHS == 0 && d_first == d_last)) {
    b0 = bx_rows_loss<NW>(sc, m1, 0, len2, len2, st, T, 0, &nfail);
  } else {
    // rows [0, t) on d_first, one gap, the rest on d_last: a column gap (d_last > d_first) or `skip` inserted rows
    const int shift = d_last - d_first, skip = shift < 0 ? -shift : 0;            // |shift| < BX_MAXW
    int t_lo = an.t_lo, t_hi = an.t_hi - skip;
    if (t_hi > R - skip) t_hi = R - skip;
    if (t_lo > t_hi) t_lo = t_hi;
    if (t_lo < 1) { out->b0 = BXF_PATH; return; }
    uint64_t m2[NW];
    DiagScan<NW> s2 = sc;
    s2.seek(rp, (int64_t)s + d_last);
#pragma unroll
    for (int j = 0; j < NW; j++) m2[j] = bx_loss_rows<NW>(s2, j);
    // the switch row with the fewest mismatches in [t_lo, t_hi]
    int cur = bx_count<NW>(m1, 0, t_lo) + bx_count<NW>(m2, t_lo + skip, len2), best = cur, tbest = t_lo;
    for (int t = t_lo + 1; t <= t_hi; t++) {
      cur += bx_bit<NW>(m1, t - 1) - bx_bit<NW>(m2, t - 1 + skip);
      if (cur < best) { best = cur; tbest = t; }
    }
    b0 = shift > 0 ? GOP + GEP * shift : GOP + (GEP + T.max_m) * skip;
    b0 = bx_rows_loss<NW>(sc, m1, 0, tbest, len2, st, T, b0, &nfail);
    b0 = bx_rows_loss<NW>(s2, m2, tbest + skip, len2, len2, st, T, b0, &nfail);
  }
  const bool proof = nfail == 0;
  // what bounds the optimum is the value the RECURRENCE reaches along the written-down path, which the new-start quirk
  // can push below the path's own (found by tools/band_campaign.py with the ancient matrix: two heavy substitutions in
  // rows 0 and 1, a start in row 2 that forfeits 214, and a six-column gap that then wins by 30)
  const int b0x = b0 + nfail * T.max_m;
  out->b0 = BXF_BUDGET;
  // the pigeonhole: a path that crosses no block cleanly loses more than B0.  Against a reference full of ambiguity codes B0
  // is mostly N columns (ten of them under a 100-base read, 210 each) -- but so is every other path's loss: the window-wide
  // credit (bx_make_tables, bx_window_nmin) is worked out only for the reads the plain sum turns away
  int ncredit = 0;
  if (b0x > an.budget || (an.l_out >= 0 && an.l_out <= b0x)) {
    if (T.loss[BX_LOSS_NCRED + st * BX_NCRED_K + 1] > 0) {
      const int nm = bx_window_nmin(rp, s, len1, len2, an.fine != 0);
      const int16_t* cum = T.loss + BX_LOSS_NCRED + st * BX_NCRED_K;
      // (beyond the table every further column carries what the last one did: the interior rows' credit)
      if (nm > 0) ncredit = nm < BX_NCRED_K ? cum[nm] : cum[BX_NCRED_K - 1] + (nm - (BX_NCRED_K - 1)) * (cum[BX_NCRED_K - 1] - cum[BX_NCRED_K - 2]);
    }
  }
#ifdef BX_DIAG
  bx_diag[0] = b0x; bx_diag[1] = an.budget; bx_diag[2] = an.l_out; bx_diag[3] = ncredit; bx_diag[4] = an.fine; bx_diag[5] = an.s_un; bx_diag[6] = an.a_hi - an.a_lo;
#endif
  if (b0x > an.budget + ncredit) return;
  // Anchors were set aside (bx_anchors, bx_fine_anchors): a path that crosses none of the kept ones cleanly must lose more than B0.
  // l_out charges it every block that is clean nowhere outside the kept range.  Six-mers are clean somewhere by chance -- against a
  // reference with an N in every tenth column half the blocks are, and l_out collapsed -- but a path does not get to USE them all:
  // it has at most J = B0 / (GOP + GEP) events (each costs that much), so its rows lie on at most J + 1 diagonals, and the blocks it
  // crosses cleanly are clean on one of those: at most the J + 1 largest per-diagonal counts together (fc2 diagonals hold two, fc1
  // hold one or more, none holds three).  Every other block is broken -- by a row that is no match (fdl_b) or by an event, which
  // pays for the blocks it touches (the netting of the stray tables).
  int l_out = an.l_out;
  if (an.fine && l_out >= 0 && an.fc2 >= 0) {
    const int m = b0x / (GOP + GEP) + 1;
    const int two = m < an.fc2 ? m : an.fc2, rest = m - two, ones = an.fc1 - an.fc2;
    const int t = 2 * two + (rest < ones ? rest : ones);
    const int alt = an.budget + 1 - an.fmax * t;
    if (alt > l_out) l_out = alt;
  }
  if (l_out >= 0 && l_out + ncredit <= b0x) { out->b0 = BXF_SPAN; return; }
  // how far a path that loses no more than b0 can stray from the anchors: all of b0 spent on one gap (band_body.h) --
  // or, tighter, what is left of b0 once every block that occurs nowhere in the window has been paid for (an.s_un: such
  // a block costs dl wherever it is crossed, unless a gap of the path itself breaks it -- the stray tables are net of
  // that).  In full: loss(P) >= s_un + sum over P's events of (cost - dl of the unanchored blocks the event touches), every
  // term >= 0; to be n diagonals off an anchor, the events between that place and the anchor add up to n in one direction.
  int g_dn = b0x < GOP + GEP ? 0 : (b0x - GOP) / GEP, g_up = g_dn;
  if (g_dn > 0) {
    const int x = b0x - an.s_un;
    if (x < 0) { out->b0 = BXF_PATH; return; }         // (cannot happen: every path pays for the blocks that occur nowhere)
    const int16_t* dn = T.dl + (an.fine ? bx_stray_off_fine(st, 0) : bx_stray_off(st, len2, 0));
    const int16_t* up = T.dl + (an.fine ? bx_stray_off_fine(st, 1) : bx_stray_off(st, len2, 1));
    if (g_dn > BX_GMAX) g_dn = g_up = BX_GMAX + 1;     // (beyond the tables: the band is too wide anyway)
    else {
      // either side can be reached either way: below the anchors by skipped rows behind them or by a column gap in front
      // of them (the path starts low and comes up), above them the other way round
      while (g_dn > 0 && dn[g_dn] > x && up[g_dn] > x) g_dn--;
      g_up = g_dn;
    }
  }
  // N CREDIT.  A band of [a_lo - G, a_hi + G] is already proven, the written-down path P0 lies in it.  Every N column c
  // with a_hi + G <= c <= R + a_lo - G (window columns) is crossed by EVERY path P of that band that starts in row 0: by a row
  // r = c - d for a diagonal d of the band (cost >= kap of that depth range) or inside a column gap (GEP >= kap).  Take that
  // much (credit = the sum of kap over those columns) out of both sides: P's events must fit into y = B0 - credit, where an
  // event costs, net of the credit it may consume, at least GOP -- a column gap GOP + GEP per column that is NOT one
  // of those, skipped rows GOP + (GEP + min M) each (they cross nothing), a late start of r rows GOP + GEP (r + 1) +
  // r min M (it misses at most r of the columns).  P runs through an anchor somewhere (the pigeonhole), so it never is
  // further from the anchors than its column gaps add up to, or its skipped rows:
  //   y < GOP      no event at all: a pure diagonal through an anchor.
  //   j gaps       hold m_j = (y - j GOP) / GEP columns without credit between them, and each of them at most H(m_j) credited
  //                ones, H(m) = the most credited columns in a stretch with at most m others (bx_ones_span): together no
  //                more than m_j + min(all credited, j H(m_j)) columns;
  //   skipped rows number at most (y - GOP) / (GEP + min M).
  // The band this gives is proven in turn, so the argument can be repeated with it (more columns count, y shrinks).
  // (Against mt311, every tenth column an ambiguity code, B0 is mostly such columns: without the credit the band would
  // be 20-30 diagonals wide.)  sc sits on d_first: bit q of its planes is window column d_first + q.
  if (g_dn + g_up > 0) {
    int G = b0x < GOP + GEP ? 0 : (b0x - GOP) / GEP;
    const int gt = g_dn > g_up ? g_dn : g_up;
    if (gt <= BX_GMAX && gt < G) G = gt;
    for (int pass = 0; pass < 2 && G > 0; pass++) {
      const int q_lo = an.a_hi + G - d_first, q_hi = R + an.a_lo - G - d_first;      // (0 <= q_lo, q_hi <= R)
      int credit = 0, k = 0, dmin = 1 << 14;              // dmin: the least a credited column costs MORE when a row crosses it than the credit it carries
      uint64_t cm[NW];
#pragma unroll
      for (int j = 0; j < NW; j++) {
        uint64_t w = ~sc.cok[j] & sc.rows[j];
        cm[j] = 0;
        while (w) {
          const int t = df_ctz(w), q = j * 64 + t;
          w &= w - 1;
          if (q < q_lo || q > q_hi) continue;
          int r_lo = q + d_first - an.a_hi - G, r_hi = q + d_first - an.a_lo + G;
          if (r_lo < 0) r_lo = 0;
          if (r_hi > R) r_hi = R;
          const int kv = T.loss[BX_LOSS_KAP + (st * 31 + sm_depth(r_lo, len2)) * 31 + sm_depth(r_hi, len2)];
          if (kv <= 0) continue;
          const int kv2 = T.loss[BX_LOSS_KAP2 + (st * 31 + sm_depth(r_lo, len2)) * 31 + sm_depth(r_hi, len2)];
          if (kv2 - kv < dmin) dmin = kv2 - kv;
          credit += kv;
          k++;
          cm[j] |= 1ull << t;
        }
      }
      const int y = b0x - credit;
      if (credit <= 0 || y >= 4 * GOP) break;
      // A credited column under a ROW costs lambda, which may exceed the credit it carries (min(GEP, lambda): 210 against 200 with the
      // flat matrix) by dmin or more; only the columns inside the path's gaps get away with GEP.  A path with j gaps that hold
      // h of the k credited columns therefore needs  j GOP + GEP m + (k - h) dmin <= y  -- with ten N columns under a read that is
      // what tells "one substitution, no room for any gap" (the plan finishes the read) from "one substitution and a gap of one".
      if (dmin < 0 || dmin >= (1 << 14)) dmin = 0;
      int gn = 0;
      if (y >= GOP) {
        const int rows_room = y - GOP - k * dmin;                   // skipped rows cross nothing: every credited column is under a row
        gn = rows_room >= 0 ? rows_room / (GEP + T.min_m) : 0;
        // (H is worked out once, for the one-gap case's m -- the largest: H grows with m, so the same value bounds the cases of two
        // and more gaps from above; walking the bit masks once per case was a third of the planner's time against mt311)
        const int h1 = bx_ones_span<NW>(cm, (y - GOP) / GEP);
        for (int j = 1; j * GOP <= y; j++) {
          const int h = j * h1 < k ? j * h1 : k, room = y - j * GOP - (k - h) * dmin;
          if (room < 0) continue;
          const int tot = room / GEP + h;
          if (tot > gn) gn = tot;
        }
      }
      if (gn >= G) break;
      G = gn;
      if (gn < g_dn) g_dn = gn;
      if (gn < g_up) g_up = gn;
    }
  }
  // (one diagonal more where the window's first column is within reach, as band_body.h)
  if (an.a_lo - g_dn - 1 < 0) { g_dn++; g_up++; }
  const int d0 = an.a_lo - g_dn, w = an.a_hi - an.a_lo + g_dn + g_up + 1;
  out->b0 = BXF_WIDTH;
  if (w > T.maxw) return;
  out->d0 = d0; out->w = w; out->b0 = b0; out->dstar = d_first;
  if (d_first != d_last || !proof) out->mode = BX_TRACE;
  else out->mode = w == 1 ? BX_DONE : BX_VALUES;
  // the widest band of the read's class must not leave the window anywhere for the plain form of the recurrence
  const int wc = bx_class_width(bx_class_of(w));
  out->edge = !(d0 >= 0 && len2 - 1 + d0 + wc <= len1);
}

// THE QUICK PLAN (round 6).  bx_anchors asks the table where EACH of the read's nine blocks occurs -- a walk of dependent loads through
// the open-addressed table --, sorts the answers into diagonals and clusters them around their median: some 1 500 instructions and a few
// trips to the L2 per read, 44 % of k_bx_plan's cycles -- to find out, for nine reads in ten of a steady-state iteration, that every
// block sits where the read was aligned before: on diagonal d = as - s of its window.  This is the same proof with the question
// turned round: look at diagonal d FIRST (one seek of the planes, the mismatch mask), and ask about each block's 10-mer the two
// questions that settle its place in the pigeonhole WITHOUT knowing positions -- two bitmaps over all 4^10 10-mers (KmerBits, 256 KB,
// made with the table): does it occur in the reference at all, does it occur more than once (start positions 0 .. L - 1 of the
// wrapped string: the wrap's copies are the same places).
//   * a block whose ten rows are free of loss on d carries the reference's own 10-mer of that place; if that 10-mer occurs ONCE in the
//     reference its only place inside the window is on d (a window no longer than the reference holds a place and its wrap copy never
//     both): the block is in the family, anchored on d and nowhere else;
//   * a block whose 10-mer occurs NOWHERE in the reference is in the family with no anchor at all: every path breaks it (budget and
//     s_un take its dl, exactly as bx_anchors' "nowhere");
//   * every other block -- a repeated 10-mer, or a broken block whose 10-mer happens to exist somewhere -- is left out of the family as
//     an overloaded 10-mer is there: no budget from it, no anchor of it, nothing is claimed about it.
// The theorem is the one bx_finish already applies: a path that crosses no block of the family cleanly loses at least the sum of
// their dl > B0, so every path that can win or tie crosses one -- on d, the only place there is -- and strays from it by what B0 less
// s_un leaves.  The family is at most bx_anchors' own, so the band is at least as wide as the full plan's: the values DP may have to
// decide where the full plan would have -- never a wrong answer.  A reference with N columns (kh.wild: the table spells them out): only
// for the windows that hold NONE.  The bitmaps count the places made of plain bases alone, all over the reference; a window of plain
// bases holds no other kind of place, so "once in the reference" and "nowhere in the reference" say about the window what they say
// without any N, and no N credit is due.  (An assembly's consensus keeps a handful of N columns where coverage is thin -- configs[4]'s
// has two in 100 kb -- and the whole run would go without the quick plan for their sake; a reference that is N all over, every run's
// first iteration, is not asked: align_all.)  Not for a window longer than the reference.  false: nothing decided, the full plan
// (bx_anchors ...) takes the read.
// (the two bitmaps word by word side by side: one 8-byte load answers both questions about a 10-mer)
struct alignas(8) KbPair { uint32_t present, repeated; };
struct KmerBits { const KbPair* w; int32_t ref_len; };      // ref_len = L (the places the bitmaps count); w == nullptr: none
constexpr int64_t KB_WORDS = (int64_t)1 << (2 * DF_K - 5);      // pairs (32 10-mers each)
constexpr int BX_QUICK_MAX = 8;        // more rows with a loss on d than this: an indel, or a read that belongs elsewhere -- not worth the look-ups
template <int NW>
MIA_HD inline bool bx_quick(DiagScan<NW>& sc, const RefPlanes& rp, const KmerHash& kh, const KmerBits& kb, int s, int len1, int len2, int st, int d, const BxTab& T,
                            BxPlan* out) {
  constexpr int NB = bx_nb_max<NW>();
  out->mode = BX_NONE; out->b0 = 0;
  if (!kb.w || d < 0 || d > len1 - len2 || len1 > kb.ref_len) return false;
  if (kh.wild > 0 && !all_bases(rp, s, (int64_t)s + len1)) return false;      // (a reference with N columns: only the windows that hold none -- see above)
  const int R = len2 - 1, nb_cut = bx_blocks_of(len2);
  sc.seek(rp, (int64_t)s + d);
  uint64_t m1[NW];
  int nm = 0;
#pragma unroll
  for (int j = 0; j < NW; j++) { m1[j] = bx_loss_rows<NW>(sc, j); nm += df_popc(m1[j]); }
  if (nm > BX_QUICK_MAX) return false;
  const int16_t* dl = T.dl + (st * (MAX_READ + 1) + len2) * BX_BLOCKS;
  uint32_t kidx[NB], w1[NB], w2[NB];
  int32_t dlv[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    kidx[b] = 0; dlv[b] = 0; w1[b] = 0; w2[b] = 0;
    if (b < nb_cut) {
      kidx[b] = bx_kmer_planes<NW>(sc, bx_block_row(b, len2, nb_cut));
      const KbPair pr = kb.w[kidx[b] >> 5];
      w1[b] = pr.present; w2[b] = pr.repeated;
      dlv[b] = dl[b];
    }
  }
  BX_LOADS_ISSUED();
  int budget = -1, nbv = 0, s_un = 0, b_lo = -1, b_hi = -1;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    if (b >= nb_cut) continue;
    const int o = bx_block_row(b, len2, nb_cut);
    const bool clean = bx_count<NW>(m1, o, o + DF_K) == 0;
    const bool present = ((w1[b] >> (kidx[b] & 31u)) & 1u) != 0u, repeated = ((w2[b] >> (kidx[b] & 31u)) & 1u) != 0u;
    if (!present) {                           // occurs nowhere: every path breaks this block
      budget += dlv[b]; s_un += dlv[b]; nbv++;
    } else if (clean && !repeated) {          // the reference's own 10-mer of this place, and its only one: anchored on d
      budget += dlv[b]; nbv++;
      if (b_lo < 0) b_lo = b;
      b_hi = b;
    }
  }
  if (nbv < BX_MIN_BLOCKS || b_lo < 0) return false;
  BxAnchors an;
  an.fail = 0; an.a_lo = d; an.a_hi = d; an.d_first = d; an.d_last = d; an.budget = budget; an.t_lo = 1; an.t_hi = R; an.l_out = -1; an.s_un = s_un;
  an.r_head = bx_block_row(b_lo, len2, nb_cut); an.r_tail = bx_block_row(b_hi, len2, nb_cut) + DF_K; an.rescue = 0; an.fine = 0; an.fc1 = 0; an.fc2 = 0; an.fmax = 0;
  bx_finish<NW, 1>(sc, rp, an, s, len1, len2, st, T, out);
  if (out->mode == BX_NONE) { out->b0 = 0; return false; }
  return true;
}
// THE QUICK PLAN, SECOND FORM: one short indel.  A read that the one-diagonal form cannot plan -- in a steady-state iteration that is a
// read with an indel, one in ten -- leaves the read's own diagonal d behind the indel for d2 = d + shift, |shift| <= BX_QUICK_SHIFT
// (a deletion in the read moves it up, an insertion down).  The blocks in front of the indel are clean on d, the blocks behind it on
// d2; a 10-mer that occurs once in the reference has ONE place, so a block that is found clean on either diagonal and is unique is
// anchored there and nowhere else -- however it was found.  The family: those blocks, and the blocks whose 10-mer occurs nowhere; the
// anchors: d in front, d2 behind (all of d's blocks before all of d2's, or this is not one indel and the full plan takes the read);
// bx_finish's two-diagonal form writes the path down (the switch row between the last block on d and the first on d2) and proves the
// band [min - g, max + g] exactly as for anchors that bx_anchors found in the table.  Every shift is tried, the one with the most blocks
// behind the indel is taken (a wrong choice costs a wider band or the budget test, never a wrong answer: the anchors are true places).
constexpr int BX_QUICK_SHIFT = 3;
template <int NW>
MIA_HD inline bool bx_quick2(DiagScan<NW>& sc, const RefPlanes& rp, const KmerHash& kh, const KmerBits& kb, int s, int len1, int len2, int st, int d, const BxTab& T,
                             BxPlan* out) {
  constexpr int NB = bx_nb_max<NW>();
  out->mode = BX_NONE; out->b0 = 0;
  if (!kb.w || d < 0 || d > len1 - len2 || len1 > kb.ref_len) return false;
  if (kh.wild > 0 && !all_bases(rp, s, (int64_t)s + len1)) return false;      // (a reference with N columns: only the windows that hold none -- see above)
  const int R = len2 - 1, nb_cut = bx_blocks_of(len2);
  const int16_t* dl = T.dl + (st * (MAX_READ + 1) + len2) * BX_BLOCKS;
  uint32_t kidx[NB], w1[NB], w2[NB];
  int32_t dlv[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    kidx[b] = 0; dlv[b] = 0; w1[b] = 0; w2[b] = 0;
    if (b < nb_cut) {
      kidx[b] = bx_kmer_planes<NW>(sc, bx_block_row(b, len2, nb_cut));
      const KbPair pr = kb.w[kidx[b] >> 5];
      w1[b] = pr.present; w2[b] = pr.repeated;
      dlv[b] = dl[b];
    }
  }
  BX_LOADS_ISSUED();
  uint32_t uniq = 0, absent = 0;               // bit b: the block's 10-mer occurs once / nowhere in the reference
#pragma unroll
  for (int b = 0; b < NB; b++) {
    if (b >= nb_cut) continue;
    const bool present = ((w1[b] >> (kidx[b] & 31u)) & 1u) != 0u, repeated = ((w2[b] >> (kidx[b] & 31u)) & 1u) != 0u;
    if (!present) absent |= 1u << b; else if (!repeated) uniq |= 1u << b;
  }
  // the unique blocks that are clean on diagonal x (bit b), for x = d - SHIFT .. d + SHIFT: one seek, then a column at a time
  uint32_t on[2 * BX_QUICK_SHIFT + 1];
  sc.seek(rp, (int64_t)s + d - BX_QUICK_SHIFT);
#pragma unroll
  for (int k = 0; k <= 2 * BX_QUICK_SHIFT; k++) {
    if (k) sc.advance(rp, (int64_t)s + d - BX_QUICK_SHIFT + k);
    uint64_t m[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) m[j] = bx_loss_rows<NW>(sc, j);
    uint32_t c = 0;
#pragma unroll
    for (int b = 0; b < NB; b++) {
      if (b >= nb_cut) continue;
      const int o = bx_block_row(b, len2, nb_cut);
      if (bx_count<NW>(m, o, o + DF_K) == 0) c |= 1u << b;
    }
    const int x = d - BX_QUICK_SHIFT + k;
    on[k] = (x >= 0 && x <= len1 - len2) ? (c & uniq) : 0u;       // (the written-down path must stay inside the window: bx_anchors' BXF_PATH)
  }
  const uint32_t cd = on[BX_QUICK_SHIFT];
  if (!cd) return false;                       // no anchor on the read's own diagonal (an indel in its first rows: the full plan's end-indel rescue)
  const int b_first = 31 - df_clz32(cd);       // the last block anchored on d
  int best = -1, best_n = 0;
#pragma unroll
  for (int k = 0; k <= 2 * BX_QUICK_SHIFT; k++) {
    if (k == BX_QUICK_SHIFT) continue;
    const uint32_t c2 = on[k];
    if (!c2 || (c2 & ((2u << b_first) - 1u))) continue;            // nothing there, or a block of it in front of d's last: not "d, one indel, d2"
    const int nn = df_popc32(c2);
    if (nn > best_n) { best_n = nn; best = k; }
  }
  if (best < 0) return false;
  const uint32_t c2 = on[best];
  const int d2 = d - BX_QUICK_SHIFT + best;
  const uint32_t fam = cd | c2 | absent;
  int budget = -1, s_un = 0, nbv = 0;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    if (!((fam >> b) & 1u)) continue;
    budget += dlv[b]; nbv++;
    if ((absent >> b) & 1u) s_un += dlv[b];
  }
  if (nbv < BX_MIN_BLOCKS) return false;
  const int b_lo = df_ctz32(cd), b_last = df_ctz32(c2), b_hi = 31 - df_clz32(c2);
  BxAnchors an;
  an.fail = 0; an.a_lo = d < d2 ? d : d2; an.a_hi = d < d2 ? d2 : d; an.d_first = d; an.d_last = d2; an.budget = budget; an.l_out = -1; an.s_un = s_un;
  an.t_lo = bx_block_row(b_first, len2, nb_cut) + DF_K; an.t_hi = bx_block_row(b_last, len2, nb_cut);
  if (an.t_lo < 1) an.t_lo = 1;
  an.r_head = bx_block_row(b_lo, len2, nb_cut); an.r_tail = bx_block_row(b_hi, len2, nb_cut) + DF_K; an.rescue = 0; an.fine = 0; an.fc1 = 0; an.fc2 = 0; an.fmax = 0;
  (void)R;
  bx_finish<NW, 2>(sc, rp, an, s, len1, len2, st, T, out);
  if (out->mode == BX_NONE) { out->b0 = 0; return false; }
  return true;
}
// one reference place into the bitmaps (device: atomics; host: the tests)
MIA_HD inline void kmer_bits_insert(const uint8_t* codes, int64_t n_codes, int64_t p, KbPair* w) {
  uint32_t idx;
  uint64_t npos;
  if (kmer_wild_at(codes, n_codes, p, &idx, &npos) != 0) return;                  // (off the end, or an N inside: no path crosses such a place cleanly)
  const uint32_t bit = 1u << (idx & 31u);
  KbPair* e = w + (idx >> 5);
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t old = atomicOr(&e->present, bit);
  if (old & bit) atomicOr(&e->repeated, bit);
#else
  if (e->present & bit) e->repeated |= bit;
  e->present |= bit;
#endif
}

// a read the 10-mers could not vouch for (budget, or anchors set aside that l_out could not cover): the fine blocks may
MIA_HD inline bool bx_wants_fine(const BxPlan& p) { return p.mode == BX_NONE && (p.b0 == BXF_BUDGET || p.b0 == BXF_SPAN); }
MIA_HD inline bool a_hi_ok(const BxAnchors& an, const BxTab& T) { return an.fail == 0 && an.a_hi - an.a_lo + 2 * BX_FINE_RADIUS < T.maxw; }

MIA_HD inline bool bx_plannable(const RefPlanes& rp, const KmerHash& ko, int64_t n_ref, int s, int len1, int len2) {
  if (!ko.slot || len2 < BX_MIN_BLOCKS * DF_K || len2 > MAX_READ || len1 < len2 || len1 > DF_MAX_LEN1 || s < 0 || (int64_t)s + len1 > n_ref) return false;
  return ko.wild > 0 || all_bases(rp, s, (int64_t)s + len1);     // (N columns: only with a table that lists them)
}

// everything in one go (the kernel does the same in two phases: reads with anchors on two diagonals are collected and
// finished by the block's first threads).  out->mode == BX_NONE: out->b0 holds the reason (BXF_*).
template <int NW>
MIA_HD inline void bx_plan_nw(const RefPlanes& rp, const KmerHash& ko, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2, int st,
                              const BxTab& T, BxPlan* out) {
  out->mode = BX_NONE;
  DiagScan<NW> sc;
  out->b0 = BXF_READ;
  if (!sc.load_read(read_packed, len2)) return;              // a read with N
  BxAnchors an;
  bx_anchors<NW>(sc, ko, reinterpret_cast<const uint32_t*>(read_packed), s, len1, len2, st, T, &an);
  out->b0 = an.fail;
  if (an.fail) return;
  bx_finish<NW>(sc, rp, an, s, len1, len2, st, T, out);
  if (out->mode == BX_NONE && (out->b0 == BXF_BUDGET || out->b0 == BXF_WIDTH) && an.d_first == an.d_last && an.a_lo == an.a_hi) {
    const int why = out->b0;
    if (bx_rescue<NW>(sc, rp, an, s, len1, len2)) bx_finish<NW, 2>(sc, rp, an, s, len1, len2, st, T, out);
    else out->b0 = why;
  }
  if (bx_wants_fine(*out) && a_hi_ok(an, T)) {
    const int why = out->b0;
    if (bx_fine_anchors<NW>(sc, rp, s, len1, len2, st, T, &an)) bx_finish<NW>(sc, rp, an, s, len1, len2, st, T, out);
    if (out->mode == BX_NONE && out->b0 != BXF_WIDTH) out->b0 = why;
  }
}

// the quick plan alone, for a caller that has a diagonal to ask about (tests/emul: emu_bandx with opts & 256); false: not decided
MIA_HD inline bool bx_plan_quick(const RefPlanes& rp, const KmerHash& ko, const KmerBits& kb, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2, int st, int d,
                                 const BxTab& T, BxPlan* out) {
  out->mode = BX_NONE; out->b0 = 0;
  if (!bx_plannable(rp, ko, n_ref, s, len1, len2)) return false;
  switch ((len2 + 63) >> 6) {
    case 1: { DiagScan<1> sc; return sc.load_read(read_packed, len2) && (bx_quick<1>(sc, rp, ko, kb, s, len1, len2, st, d, T, out) || bx_quick2<1>(sc, rp, ko, kb, s, len1, len2, st, d, T, out)); }
    case 2: { DiagScan<2> sc; return sc.load_read(read_packed, len2) && (bx_quick<2>(sc, rp, ko, kb, s, len1, len2, st, d, T, out) || bx_quick2<2>(sc, rp, ko, kb, s, len1, len2, st, d, T, out)); }
    case 3: { DiagScan<3> sc; return sc.load_read(read_packed, len2) && (bx_quick<3>(sc, rp, ko, kb, s, len1, len2, st, d, T, out) || bx_quick2<3>(sc, rp, ko, kb, s, len1, len2, st, d, T, out)); }
    default: { DiagScan<4> sc; return sc.load_read(read_packed, len2) && (bx_quick<4>(sc, rp, ko, kb, s, len1, len2, st, d, T, out) || bx_quick2<4>(sc, rp, ko, kb, s, len1, len2, st, d, T, out)); }
  }
}

MIA_HD inline void bx_plan(const RefPlanes& rp, const KmerHash& ko, int64_t n_ref, int s, int len1, const uint8_t* read_packed, int len2, int st,
                           const BxTab& T, BxPlan* out) {
  out->mode = BX_NONE;
  out->b0 = BXF_WINDOW;
  if (!bx_plannable(rp, ko, n_ref, s, len1, len2)) return;
  switch ((len2 + 63) >> 6) {
    case 1: bx_plan_nw<1>(rp, ko, n_ref, s, len1, read_packed, len2, st, T, out); break;
    case 2: bx_plan_nw<2>(rp, ko, n_ref, s, len1, read_packed, len2, st, T, out); break;
    case 3: bx_plan_nw<3>(rp, ko, n_ref, s, len1, read_packed, len2, st, T, out); break;
    default: bx_plan_nw<4>(rp, ko, n_ref, s, len1, read_packed, len2, st, T, out); break;
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

// The same W nibbles for a position that advances by ONE per read row (the band slides along the reference): NWD + 2 raw
// words stay in registers, a new one is fetched every eighth row and one group ahead of its use, so that the row loop
// never waits for the load (one read per thread: a few wavefronts per SIMD do not hide a trip to the L2 per row).
template <int NWD>
struct BxSlide {
  uint32_t raw[NWD + 2];
  int64_t q;               // word index of raw[0]
  MIA_HD inline void init(const uint32_t* refnib, int64_t nib) {
    q = nib >> 3;
#pragma unroll
    for (int k = 0; k < NWD + 2; k++) raw[k] = refnib[q + k];
  }
  // codes for nibble index nib (>= the previous one, at most 8 further on)
  MIA_HD inline void get(const uint32_t* refnib, int64_t nib, uint32_t* cw) {
    if ((nib >> 3) != q) {
#pragma unroll
      for (int k = 0; k < NWD + 1; k++) raw[k] = raw[k + 1];
      q++;
      raw[NWD + 1] = refnib[q + NWD + 1];
    }
    const int sh = (int)(nib & 7) * 4;
#pragma unroll
    for (int k = 0; k < NWD; k++) cw[k] = sh ? (raw[k] >> sh) | (raw[k + 1] << (32 - sh)) : raw[k];
  }
};

struct BxResult { int score, abc, aec, abr, gaps; uint32_t gap_desc; };

// Values only.  sub: the substitution table of the read's strand (31 x 4 x BX_SUB_ROW words; LDS on the device).
// Returns the first maximum of the last row: *best and its band index *bj (-1: none).
template <int W, bool EDGE>
MIA_HD inline void bx_values(const uint32_t* refnib, int s, int len1, const uint32_t* rwords, int len2, int d0, const int32_t* sub, int* best_out, int* bj_out) {
  constexpr int NWD = W / 8;
  int32_t P[W], H[W];
  uint32_t cw[NWD];
  uint32_t rw = rwords[0];
  auto live_mask = [&](int c0) -> uint64_t {             // cells whose column c0 + j lies inside the window
    const int jlo = c0 < 0 ? -c0 : 0, jhi = (len1 - c0) < W ? (len1 - c0) : W;
    return jhi > jlo ? ((jhi >= 64 ? ~0ull : ((1ull << jhi) - 1ull)) & ~((1ull << jlo) - 1ull)) : 0ull;
  };
  BxSlide<NWD> slide;
  slide.init(refnib, (int64_t)s + d0 + BX_NIB_LEAD);
  {
    slide.get(refnib, (int64_t)s + d0 + BX_NIB_LEAD, cw);
    const int32_t* row = sub + ((0 * 4) + (int)(rw & 3u)) * BX_SUB_ROW;          // depth 0
    const uint64_t live = EDGE ? live_mask(d0) : ~0ull;
#pragma unroll
    for (int j = 0; j < W; j++) {
      H[j] = BX_NEG;
      const int v = row[(cw[j >> 3] >> (4 * (j & 7))) & 7u];
      P[j] = ((live >> j) & 1ull) ? v : BX_NEG;
    }
  }
  for (int r = 1; r < len2; r++) {
    const int c0 = r + d0;
    slide.get(refnib, (int64_t)s + c0 + BX_NIB_LEAD, cw);
    if ((r & 7) == 0) rw = rwords[r >> 3];
    const int32_t* row = sub + (sm_depth(r, len2) * 4 + (int)((rw >> (4 * (r & 7))) & 3u)) * BX_SUB_ROW;
    uint64_t live = ~0ull, col0 = 0ull;
    if (EDGE) {
      live = live_mask(c0);
      col0 = (c0 <= 0 && c0 > -W) ? (1ull << (-c0)) : 0ull;
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
        if ((col0 >> j) & 1ull) cur = sb + fresh;                                       // src/mia.c:805-822
        if (!((live >> j) & 1ull)) cur = BX_NEG;
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
  auto live_mask = [&](int c0) -> uint64_t {
    const int jlo = c0 < 0 ? -c0 : 0, jhi = (len1 - c0) < W ? (len1 - c0) : W;
    return jhi > jlo ? ((jhi >= 64 ? ~0ull : ((1ull << jhi) - 1ull)) & ~((1ull << jlo) - 1ull)) : 0ull;
  };
  BxSlide<NWD> slide;
  slide.init(refnib, (int64_t)s + d0 + BX_NIB_LEAD);
  {
    slide.get(refnib, (int64_t)s + d0 + BX_NIB_LEAD, cw);
    const int32_t* row = sub256 + ((0 * 4) + (int)(rw & 3u)) * BX_SUB_ROW;
    const uint64_t live = EDGE ? live_mask(d0) : ~0ull;
#pragma unroll
    for (int j = 0; j < W; j++) {
      H[j] = DEAD;
      const int v = row[(cw[j >> 3] >> (4 * (j & 7))) & 7u];
      P[j] = ((live >> j) & 1ull) ? (v | 0xFF) : DEAD;
    }
#pragma unroll
    for (int k = 0; k < W / 4; k++) trace[k] = 0xFFFFFFFFu;
  }
  for (int r = 1; r < len2; r++) {
    const int c0 = r + d0;
    slide.get(refnib, (int64_t)s + c0 + BX_NIB_LEAD, cw);
    if ((r & 7) == 0) rw = rwords[r >> 3];
    const int32_t* row = sub256 + (sm_depth(r, len2) * 4 + (int)((rw >> (4 * (r & 7))) & 3u)) * BX_SUB_ROW;
    uint64_t live = ~0ull, col0 = 0ull;
    if (EDGE) {
      live = live_mask(c0);
      col0 = (c0 <= 0 && c0 > -W) ? (1ull << (-c0)) : 0ull;
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
        if ((col0 >> j) & 1ull) cur = (sb + f0) | 0xFF;
        if (!((live >> j) & 1ull)) cur = DEAD;
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
