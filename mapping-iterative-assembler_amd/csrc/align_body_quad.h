// align_body_quad.h -- the windowed DP of align_body.h for FOUR reads per wavefront.
//
// A 100 bp read has a 200-column window; on 64 lanes x 4 columns a quarter of the lanes
// idle and the cross-lane work (prefix scan, neighbour shifts) is paid once per 4 cells.
// Here each read gets one 16-lane DPP row and every lane owns 13 columns (208 per read):
// 96 % of the lanes carry real columns, the row_shr DPP forms never leave a 16-lane row
// (no masking needed between the four reads), the scan needs 4 steps instead of 6, and
// its cost is spread over 13 cells.  The four reads of a quad must have the same length
// (the planner pads each length bin to a multiple of four with empty slots); window
// lengths, strands and positions are per read.  Arithmetic, packed words and tie rules
// are exactly those of align_body.h (IB = 8: all indices < 256).
#pragma once
#include "align_body.h"
#include "mia_layout.h"

namespace mia {

constexpr int Q_CPL = 13, Q_LPR = 16, Q_G = 4, Q_COLS = Q_CPL * Q_LPR;   // 208 columns per read
constexpr int Q_TRACE_STRIDE = Q_LPR * 32;   // per row: 16 lanes x 32 bytes = 13 cells x 16 bit [prio:2][len:8] (+pad)
constexpr int Q_BAND = 20;   // half width (columns) of the stored trace band around the expected diagonal

struct QuadArgs {                 // wave-uniform; [g] = read of lane row g, len1 == 0 marks an empty slot
  const uint8_t* ref_codes;
  const uint8_t* packed;          // base of the packed read store
  const int32_t* pssm2;           // forward PSSM followed by the reverse-complemented one
  int32_t len2;                   // common read length of the quad
  int32_t ref_start[Q_G], len1[Q_G];
  uint32_t roff[Q_G], rc[Q_G];
  PackParams pk;                  // ib = 8
  uint32_t lds_sub;               // LDS: 4 tables int16 sub[5][rows padded to even], Q_SUB_BYTES apart
  uint32_t slab_group;            // bytes between the trace areas of two reads inside the workgroup slab
  int16_t* cols_out[Q_G];
  // Trace band: only lanes whose 13 columns come within Q_BAND columns of the diagonal the read followed in
  // the previous iteration (column dexp + row) store their trace bytes; the live trace of all wavefronts then
  // fits the Infinity Cache instead of streaming to HBM.  A path that leaves the band is detected in the
  // traceback (ST_BAND) and the read is re-run by the one-read kernel with a full trace -- never guessed.
  int32_t band;                   // 0 = store everything
  int32_t dexp[Q_G];
  uint32_t dbg;
};
// LDS substitution tables: sub[code][row] int16.  The row stride is padded so that the five code rows of a read
// start >= 4 banks apart and consecutive reads are shifted by one bank: the 20 (read, code) rows of a quad then
// hit 20 different banks (measured before: half of the LDS cycles were bank conflicts).
MIA_HD inline uint32_t q_sub_stride(int len2) {
  uint32_t dw = (uint32_t)((len2 + 1) / 2);
  for (;; dw++) {
    uint32_t k = dw & 31u, ok = 1;
    for (uint32_t c1 = 0; c1 < 5 && ok; c1++)
      for (uint32_t c2 = c1 + 1; c2 < 5; c2++) {
        uint32_t d = (k * c2 - k * c1) & 31u;
        if (d < 4u || d > 28u) { ok = 0; break; }
      }
    if (ok) return dw * 4u;
  }
}
// bytes between the tables of two reads of a quad: the padded table plus one bank of shift
MIA_HD inline uint32_t q_sub_bytes(int len2) { return 5u * q_sub_stride(len2) + 4u; }

template <class P>
struct QuadAligner {
  typedef typename P::U U;
  typedef typename P::M M;
  static constexpr int CPL = Q_CPL, IB = 8, SH = 10;
  static constexpr uint32_t IDXM = 255u;

  // does the 13-column block starting at column blk_lo come within Q_BAND columns of the expected diagonal in row r?
  MIA_HD static inline M in_band(P& w, const QuadArgs& a, const U& dbias, const U& blk_lo, const U& r) {
    if (!a.band) return w.lane() < 64u;
    U dg = dbias + r;                                  // biased diagonal column of this row
    return (blk_lo + 4096u <= dg + (uint32_t)Q_BAND) & (blk_lo + (4096u + (uint32_t)CPL - 1u + (uint32_t)Q_BAND) >= dg);
  }

  MIA_HD static inline __attribute__((always_inline)) void run(P& w, const QuadArgs& a, AlignResult* res /* [Q_G] */) {
    const uint32_t OFF = (uint32_t)a.pk.off;
    const uint32_t UNAV = ((uint32_t)a.pk.unavail << SH) | IDXM;
    const U lane = w.lane();
    const U grp = lane >> 4, gl = lane & 15u;
    const int len2 = a.len2;
    const uint32_t RS2 = q_sub_stride(len2);
    // per-lane copies of the per-read scalars
    U len1v = U(0u), startv = U(0u), roffv = U(0u), pmoff = U(0u), dbias = U(4096u);
    for (int g = 0; g < Q_G; g++) {
      M mine = grp == (uint32_t)g;
      len1v = w.sel(mine, U((uint32_t)a.len1[g]), len1v);
      startv = w.sel(mine, U((uint32_t)a.ref_start[g]), startv);
      roffv = w.sel(mine, U(a.roff[g]), roffv);
      pmoff = w.sel(mine, U(a.rc[g] ? (uint32_t)PSSM_WORDS : 0u), pmoff);
      dbias = w.sel(mine, U((uint32_t)(a.dexp[g] + 4096)), dbias);   // biased so that band arithmetic stays unsigned
    }
    const U subbase = U(a.lds_sub) + grp * q_sub_bytes(len2);
    const U trbase = grp * a.slab_group + gl * 32u;

    // ---- substitution tables, 16 lanes per read
    for (int e0 = 0; e0 < len2 * 5; e0 += Q_LPR) {
      U e = gl + (uint32_t)e0;
      M ok = e < (uint32_t)(len2 * 5);
      U r = w.udiv5(e);
      U c1 = e - r * 5u;
      U byte = w.gload_u8(a.packed, roffv + (r >> 1), ok);
      U c2 = (byte >> ((r & 1u) << 2)) & 15u;
      U d = w.depth(r, (uint32_t)len2);
      U v = w.gload_i32(a.pssm2, pmoff + (d * 5u + c1) * 5u + c2, ok);
      w.lds_w16(subbase + c1 * RS2 + r * 2u, v, ok);
    }
    w.lds_fence();

    // Register diet (3 -> 4 waves per SIMD): the column-gap constant is affine in the column
    // (KC[j] = KC0 + j*KCD), the best_gap_row key of row r-1 is rebuilt from the old score when
    // it is needed, substitution scores are read as sign-extended 16-bit LDS loads, and the
    // cells of a row are visited right to left so that every column is updated in place.
    U col[CPL], sub_addr[CPL], QC[CPL];
    for (int j = 0; j < CPL; j++) {
      col[j] = gl * (uint32_t)CPL + (uint32_t)j;
      M in = col[j] < len1v;
      U code = w.sel(in, w.gload_u8(a.ref_codes, startv + col[j], in), U(4u));
      sub_addr[j] = subbase + code * RS2;
      QC[j] = ((col[j] * (uint32_t)GEP) << SH) + (U(IDXM) - col[j]);
    }
    // key -> column-gap candidate: value -= GOP + GEP*(c-1); prio = 2; idx -> len = c-1-k
    const U KC0 = (U(0u) - ((U((uint32_t)GOP) + (col[0] - 1u) * (uint32_t)GEP) << SH)) + (TR_COLGAP << IB) + (col[0] - 1u - IDXM);
    const uint32_t KCD = 1u - ((uint32_t)GEP << SH);
    const uint32_t WDC = TR_DIAG << IB;
    const U unav = U(UNAV);

    // The score state is kept SHIFTED and already carries the diagonal priority bits:
    // Sd = ((S + off) << SH) | (TR_DIAG << IB) IS the diagonal candidate word of the cell below-right.  The two
    // key words (next row's best_gap_col key, best_gap_row key of the previous row) are one add each, and the new
    // state is rebuilt from the winning word with one AND-OR and one shift-add of the substitution score.
    const uint32_t HI = ~((1u << SH) - 1u);
    U Sd[CPL], q[CPL], rrun[CPL];
    const U wdcv = U(WDC);
    for (int j = 0; j < CPL; j++) {
      Sd[j] = ((w.lds_ri16(sub_addr[j]) + OFF) << SH) | WDC;
      q[j] = Sd[j] + (QC[j] - WDC);
      rrun[j] = unav;
    }
    {
      const U d4 = U((TR_DIAG << IB) * 0x00010001u);
      const M ib0 = in_band(w, a, dbias, gl * (uint32_t)CPL, 0u);
      w.tr_w128m(trbase, d4, d4, d4, d4, ib0);
      w.tr_w128m(trbase + 16u, d4, d4, d4, d4, ib0);
    }

    // one DP row; `off` (0 or 2) is the immediate LDS offset of this row inside the current row pair
    auto do_row = [&](const int r, const uint32_t off) __attribute__((always_inline)) {
      const int32_t fresh = -(GOP + GEP * (r + 1));
      const uint32_t freshb = (uint32_t)(fresh + (int32_t)OFF);
      const uint32_t WS = freshb << SH;
      const uint32_t KR = (0u - ((uint32_t)(GOP + GEP * (r - 1)) << SH)) + (TR_ROWGAP << IB) + ((uint32_t)(r - 1) - IDXM);
      const uint32_t RKP = ((uint32_t)(GEP * (r - 1)) << SH) + (IDXM - (uint32_t)(r - 1)) - WDC;   // key constant of row r-1

      // the row's substitution scores: all LDS reads issued before anything waits for one (interleaved with the cells by the
      // compiler they were thirteen LDS round trips per row)
      U subv[CPL];
      for (int j = 0; j < CPL; j++) subv[j] = w.lds_ri16o(sub_addr[j], off);
      w.sched_fence();
      U dleft = w.rshr1(Sd[CPL - 1], U(WS | WDC));
      U rleft = w.rshr1_max(rrun[CPL - 1], unav);
      U u0 = w.rshr1_max(q[CPL - 2], unav);
      U u1 = w.rshr1_max(q[CPL - 1], unav);
      // lane total by a max3 tree (6 ops), then -- with the prefix of the lanes to the left folded in --
      // cand[j] = best key over all columns <= c_j - 2 as one running chain
      U tot = w.umax3(w.umax3(u0, u1, q[0]), w.umax3(q[1], q[2], q[3]),
                      w.umax3(w.umax3(q[4], q[5], q[6]), w.umax3(q[7], q[8], q[9]), q[10]));
      U excl = w.rshr1_max(w.rscan_max(tot), unav);
      U cand[CPL];
      cand[0] = w.umax(excl, u0);
      cand[1] = w.umax(cand[0], u1);
      for (int j = 2; j < CPL; j++) cand[j] = w.umax(cand[j - 1], q[j - 2]);

      // trace cell = low 16 bits of the winning word ([..|prio:2|len:8]); two cells per dword, no saturation.
      // Cells are visited right to left; each 16-byte half of the lane's trace row is stored as soon as it is complete.
      const M ib = in_band(w, a, dbias, gl * (uint32_t)CPL, (uint32_t)r);
      const U trrow = trbase + (uint32_t)r * Q_TRACE_STRIDE;
      U bodd = U(0u), pk[4];
      pk[3] = U(0u);
      for (int j = CPL - 1; j >= 0; j--) {
        U Wd = (j == 0) ? dleft : Sd[j - 1];
        U rl = (j == 0) ? rleft : rrun[j - 1];
        U Wc = w.add3(cand[j], KC0, (uint32_t)j * KCD);
        U Wr = rl + KR;
        U m3 = w.umax3(Wd, Wc, Wr);
        U best = w.umax(m3, U(WS));
        U sub = subv[j];
        // start (only if strictly better than the other three) drops the substitution score (src/mia.c:910-917)
        U snew = w.shl_add(w.sel(m3 < WS, U(0u), sub), SH, w.bfi(HI, best, wdcv));
        rrun[j] = w.umax(rrun[j], Sd[j] + RKP);   // row r-1 becomes a best_gap_row candidate for row r+1
        Sd[j] = snew;
        q[j] = snew + (QC[j] - WDC);
        if (j == CPL - 1) pk[2] = best & 0xFFFFu;            // cell 12 (+ pad)
        else if (j & 1) bodd = best;
        else pk[(j >> 1) & 3] = w.pack16(best, bodd);          // cells (j, j+1)
        if (j == 8 && !(a.dbg & 1u)) w.tr_w128m(trrow + 16u, pk[0], pk[1], pk[2], pk[3], ib);   // cells 8..12
      }
      if (!(a.dbg & 1u)) w.tr_w128m(trrow, pk[0], pk[1], pk[2], pk[3], ib);                      // cells 0..7
    };
    // rows are processed in pairs so that the substitution-table address of a column is bumped once per two rows
    const int rows = (a.dbg & 4u) ? 1 : len2;
    for (int j = 0; j < CPL; j++) sub_addr[j] = sub_addr[j] + 2u;      // -> row 1
    {
      int r = 1;
      for (; r + 1 < rows; r += 2) {        // straight-line row pairs (no copies of the column state between rows)
        do_row(r, 0u);
        do_row(r + 1, 2u);
        for (int j = 0; j < CPL; j++) sub_addr[j] = sub_addr[j] + 4u;
      }
      if (r < rows) do_row(r, 0u);
    }

    // ---- max_sg_score per read (16-lane row); the state words order like scores
    U m = U(0u);
    for (int j = 0; j < CPL; j++) m = w.umax(m, w.sel(col[j] < len1v, Sd[j], U(0u)));
    const U bestv = w.row_last(w.rscan_max(m));
    U cmin = U(0x7FFFFFFFu);
    for (int j = CPL - 1; j >= 0; j--) cmin = w.sel((col[j] < len1v) & (Sd[j] == bestv), col[j], cmin);
    const U aecv = ~w.row_last(w.rscan_max(~cmin));
    w.tr_fence();

    // ---- traceback, one read after the other with all 64 lanes (same walk as align_body.h;
    //      trace rows here are [16 lanes][16 bytes], column c lives at (c/13)*16 + c%13)
    for (int gi = 0; gi < Q_G; gi++) {
      AlignResult& rs = res[gi];
      rs.score = 0; rs.abc = 0; rs.abr = 0; rs.aec = 0; rs.status = ST_SKIPPED;
      if (a.len1[gi] <= 0) continue;
      rs.score = (int32_t)((w.lane_val(bestv, gi * Q_LPR) >> SH) - OFF);
      rs.aec = (int32_t)w.lane_val(aecv, gi * Q_LPR);
      const uint32_t tb0 = (uint32_t)gi * a.slab_group;
      int r = len2 - 1, c = rs.aec, aln_cols = 0;
      uint32_t status = ST_OK;
      for (int guard = 0; guard < ((a.dbg & 2u) ? 0 : 4 * MAX_READ + 8); guard++) {
        U ri = U((uint32_t)r) - lane, ci = U((uint32_t)c) - lane;
        M inside = (lane <= (uint32_t)r) & (lane <= (uint32_t)c);
        U cl = w.udiv13(ci);
        // a cell whose lane block lies outside the stored band has no trace byte
        M covered = in_band(w, a, U(w.lane_val(dbias, gi * Q_LPR)), cl * (uint32_t)CPL, ri);
        U tb = w.tr_r16(U(tb0) + ri * Q_TRACE_STRIDE + cl * 32u + (ci - cl * 13u) * 2u, inside & covered);
        U ty = (tb >> IB) & 3u, ln = tb & IDXM;
        M colgap0 = (ty == U(TR_COLGAP)) & (ln + 1u == ci);
        M rowgap0 = (ty == U(TR_ROWGAP)) & (ln + 1u == ri);
        M plain_diag = (ty == U(TR_DIAG)) | colgap0 | rowgap0;
        M terminal = (ri == U(0u)) | (ci == U(0u)) | (ty == U(TR_START));
        M stop_here = (!inside) | (!covered) | terminal | (!plain_diag);
        uint64_t bal = w.ballot(stop_here);
        int f = bal ? __builtin_ctzll(bal) : WAVE;
        if (f < WAVE && w.lane_bit(inside, f) && !w.lane_bit(covered, f)) { status |= ST_BAND; rs.abr = r - f; rs.abc = c - f; break; }
        int npairs = f < WAVE ? f + 1 : WAVE;
        if (f < WAVE && !w.lane_bit(inside, f)) npairs = f;
        w.gstore_i16(a.cols_out[gi], ri, ci, lane < (uint32_t)npairs);
        aln_cols += npairs;
        if (f == WAVE) { r -= WAVE; c -= WAVE; continue; }
        const uint32_t fty = w.lane_val(ty, f), fln = w.lane_val(ln, f);
        const bool fterm = w.lane_bit(terminal, f);
        const int fr = r - f, fc = c - f;
        if (fterm) { rs.abr = fr; rs.abc = fc; break; }
        if (fty == TR_COLGAP) { r = fr - 1; c = fc - 1 - (int)fln; aln_cols += (int)fln; }
        else {
          for (int k0 = 0; k0 < (int)fln; k0 += WAVE) {   // lengths are exact here (no saturation): up to 255 rows
            U rr = U((uint32_t)(fr - 1 - k0)) - lane;
            w.gstore_i16(a.cols_out[gi], rr, U((uint32_t)(uint16_t)COL_INSERT), lane + (uint32_t)k0 < fln);
          }
          r = fr - 1 - (int)fln; c = fc - 1;
          aln_cols += (int)fln;
        }
      }
      for (int r0 = 0; r0 < rs.abr; r0 += WAVE) {
        U rr = lane + (uint32_t)r0;
        w.gstore_i16(a.cols_out[gi], rr, U((uint32_t)(uint16_t)COL_CLIP), rr < (uint32_t)rs.abr);
      }
      if (aln_cols > 2 * MAX_READ) status |= ST_TOO_LONG;
      rs.status = status;
    }
  }
};

}  // namespace mia
