// align_body.h -- the windowed semi-global DP + traceback for ONE read on ONE
// 64-lane wavefront, written against a small "wave policy" P so that the very
// same source is compiled (a) by hipcc for gfx950 with P = DevWave (DPP
// cross-lane moves, LDS, one value per lane) and (b) by g++ with P = EmuWave
// (tests/emul: 64-wide arrays stepped in lock-step) to unit-test the logic on
// a CPU-only machine.  Nothing in the product calls the emulation.
//
// What it computes (bit-exact): dyn_prog + max_sg_score + find_align_begin +
// populate_pwaln_to_begin of the reference for an all-ones column mask
// (/root/reference/src/mia.c:740-981, 1278-1302, 612-637, 1440-1497), i.e. the
// body of the read loop of reiterate_assembly (src/mia_main.c:186-257).
//
// Mapping to the wavefront (row-parallel; a cell needs row r-1 and the history
// of column c-1 up to row r-2, never cell (r,c-1)):
//   * lane l owns columns l*CPL .. l*CPL+CPL-1 of the window, all rows;
//   * best_gap_col (running arg-max along the row)  -> wave prefix-max of
//     packed key words (lane-local chain + 6 DPP steps);
//   * best_gap_row[c] (running arg-max down a column) -> one register per owned
//     column, handed to the right neighbour with one DPP shift;
//   * scores are carried biased (S + off) so that every packed word is built
//     with one shift-add;
//   * the trace is one byte per cell, written through the policy's tr_* hooks (a
//     per-workgroup slab in global memory that stays in L2 / Infinity Cache on the
//     GPU, so that LDS does not limit occupancy); the traceback follows diagonal
//     runs 64 cells at a time with a ballot.
#pragma once
#include "mia_layout.h"

namespace mia {

struct AlignArgs {            // everything here is wave-uniform
  const uint8_t* ref_codes;   // wrapped reference, one code (0..4) per byte   [global]
  int32_t ref_start;          // first window column in ref_codes
  int32_t len1;               // window columns (<= 64*CPL)
  const uint8_t* read_packed; // this read, 4-bit codes, low nibble first      [global]
  int32_t len2;               // read length (1..256)
  const int32_t* pssm;        // sm[31][5][5] of this read's strand            [global]
  int32_t sg5;                // pay for unaligned 5' read bases (always 1 in mia)
  PackParams pk;
  uint32_t lds_sub;           // LDS byte offset: int16 sub[5][rows padded to even]
  uint32_t trace_stride;      // bytes per trace row, multiple of 4, >= len1
  int16_t* cols_out;          // script: window column per read row            [global]
  uint32_t dbg;               // timing experiments only (0 in production): 1 = no trace stores, 2 = no traceback, 4 = no row loop
};

struct AlignResult {          // wave-uniform
  int32_t score, abc, abr, aec;
  uint32_t status;
  int32_t aer;                // last row of the alignment: len2-1, or the arg-max row of the last column (LASTCOL)
};

template <int CPL>
struct PackBits {
  static constexpr int IB = (CPL == 4) ? 8 : 10;   // must agree with make_pack_params(64*CPL, ..)
  static constexpr int SH = IB + 2;
  static constexpr uint32_t IDXM = (1u << IB) - 1u;
};

// LASTCOL = the end condition of trim_frag (src/mia.c:1346-1353): best score over the LAST COLUMN (all rows,
// first maximum) instead of max_sg_score's last row; everything else is the same DP.
template <class P, int CPL, bool LASTCOL = false>
struct WindowAligner {
  typedef typename P::U U;
  typedef typename P::M M;
  static constexpr int IB = PackBits<CPL>::IB, SH = PackBits<CPL>::SH;
  static constexpr uint32_t IDXM = PackBits<CPL>::IDXM;

  MIA_HD static inline __attribute__((always_inline)) AlignResult run(P& w, const AlignArgs& a) {
    const uint32_t OFF = (uint32_t)a.pk.off;
    const uint32_t UNAV = ((uint32_t)a.pk.unavail << SH) | IDXM;
    const U lane = w.lane();
    const int len1 = a.len1, len2 = a.len2;
    const uint32_t RS2 = (uint32_t)((len2 + 1) & ~1) * 2u;   // bytes per code row of the sub table

    // ---- 1. substitution table sub[code1][r] = sm[depth(r)][code1][read[r]] (src/mia.c:792-795),
    //         int16, rows r and r+1 adjacent so that one 32-bit LDS read serves two DP rows
    for (int e0 = 0; e0 < len2 * 5; e0 += WAVE) {
      U e = lane + (uint32_t)e0;
      M ok = e < (uint32_t)(len2 * 5);
      U r = w.udiv5(e);
      U c1 = e - r * 5u;
      U byte = w.gload_u8(a.read_packed, r >> 1, ok);
      U c2 = (byte >> ((r & 1u) << 2)) & 15u;
      U d = w.depth(r, (uint32_t)len2);
      U v = w.gload_i32(a.pssm, (d * 5u + c1) * 5u + c2, ok);
      w.lds_w16(U(a.lds_sub) + c1 * RS2 + r * 2u, v, ok);
    }
    w.lds_fence();

    // ---- 2. per-lane column constants
    U col[CPL], sub_addr[CPL], KC[CPL], QC[CPL];
    for (int j = 0; j < CPL; j++) {
      col[j] = lane * (uint32_t)CPL + (uint32_t)j;
      M in = col[j] < (uint32_t)len1;
      U code = w.sel(in, w.gload_u8(a.ref_codes, U((uint32_t)a.ref_start) + col[j], in), U(4u));
      sub_addr[j] = U(a.lds_sub) + code * RS2;
      // key -> column-gap candidate: value -= GOP + GEP*(c-1); prio = 2; idx -> len = c-1-k
      KC[j] = (U(0u) - ((U((uint32_t)GOP) + (col[j] - 1u) * (uint32_t)GEP) << SH)) + (TR_COLGAP << IB) + (col[j] - 1u - IDXM);
      // biased S -> key of this column for the next row: value += GEP*c, idx = IDXM - c
      QC[j] = ((col[j] * (uint32_t)GEP) << SH) + (U(IDXM) - col[j]);
    }
    const uint32_t WDC = TR_DIAG << IB;
    const U unav = U(UNAV);

    // ---- 3. row 0 (src/mia.c:769-785): S = sub, T = 0.  Sb = S + off throughout.
    U Sb[CPL], q[CPL], rrun[CPL], pend[CPL], pw[CPL];
    {
      const uint32_t rk = IDXM;   // GEP*0, idx = IDXM - 0
      for (int j = 0; j < CPL; j++) {
        pw[j] = w.lds_r32(sub_addr[j]);                 // rows 0 and 1
        Sb[j] = w.sext_lo(pw[j]) + OFF;
        q[j] = (Sb[j] << SH) + QC[j];
        pend[j] = (Sb[j] << SH) + rk;
        rrun[j] = unav;
      }
      for (int j4 = 0; j4 < CPL; j4 += 4)
        w.tr_w32(col[j4], U((TR_DIAG << 6) * 0x01010101u), col[j4] < (uint32_t)a.trace_stride);
    }
    // LASTCOL: running (first) maximum of column len1-1 over the rows, kept in the lane that owns that column
    U colbest = U(0u), colrow = U(0u);
    auto last_col_update = [&](int r) __attribute__((always_inline)) {
      U v = U(0u);
      for (int j = 0; j < CPL; j++) v = w.sel(col[j] == (uint32_t)(len1 - 1), Sb[j], v);
      M up = v > colbest;                       // biased scores are >= 1: the first row always enters
      colbest = w.sel(up, v, colbest);
      colrow = w.sel(up, U((uint32_t)r), colrow);
    };
    if (LASTCOL) last_col_update(0);

    // ---- 4. rows 1 .. len2-1
    for (int r = 1; r < ((a.dbg & 4u) ? 1 : len2); r++) {
      const int32_t fresh = a.sg5 ? -(GOP + GEP * (r + 1)) : 0;            // src/mia.c:877-880
      const uint32_t freshb = (uint32_t)(fresh + (int32_t)OFF);
      const uint32_t WS = freshb << SH;                                     // prio 0, len 0
      const uint32_t KR = (0u - ((uint32_t)(GOP + GEP * (r - 1)) << SH)) + (TR_ROWGAP << IB) + ((uint32_t)(r - 1) - IDXM);
      const uint32_t RK = ((uint32_t)(GEP * r) << SH) + (IDXM - (uint32_t)r);
      const uint32_t row_tr = (uint32_t)r * a.trace_stride;
      const bool hi = (r & 1) != 0;
      if (!hi)
        for (int j = 0; j < CPL; j++) pw[j] = w.lds_r32(sub_addr[j] + (uint32_t)r * 2u);   // rows r, r+1

      // neighbours: diag of the first owned column (column 0: "diag" = fresh, so S = sub + fresh
      // and T = 0, src/mia.c:805-822), its best_gap_row state, and the two keys left of the
      // lane for the shifted prefix scan.  Keys never go below UNAV, so max(., unav) is the fill.
      U dleft = w.shr1(Sb[CPL - 1], U(freshb));
      U rleft = w.shr1_max(rrun[CPL - 1], unav);
      U u0 = w.shr1_max(q[CPL - 2], unav);
      U u1 = w.shr1_max(q[CPL - 1], unav);
      // g[j] = max key over columns <= c_j - 2 inside {two left keys, own keys}
      U g[CPL];
      g[0] = u0;
      g[1] = w.umax(u0, u1);
      for (int j = 2; j < CPL; j++) g[j] = w.umax(g[j - 1], q[j - 2]);
      U excl = w.shr1_max(w.scan_max(g[CPL - 1]), unav);

      U best[CPL], Snew[CPL];
      for (int j = 0; j < CPL; j++) {
        U diag = (j == 0) ? dleft : Sb[j - 1];
        U rl = (j == 0) ? rleft : rrun[j - 1];
        U Wd = (diag << SH) + WDC;
        U Wc = w.umax(excl, g[j]) + KC[j];
        U Wr = rl + KR;
        U m3 = w.umax3(Wd, Wc, Wr);
        best[j] = w.umax(m3, U(WS));
        U sub = hi ? w.sext_hi(pw[j]) : w.sext_lo(pw[j]);
        // start (only if strictly better than the other three) drops the substitution score (src/mia.c:910-917)
        Snew[j] = (best[j] >> SH) + w.sel(m3 < WS, U(0u), sub);
      }
      if (!(a.dbg & 1u)) for (int j4 = 0; j4 < CPL; j4 += 4)
        w.tr_w32(U(row_tr) + col[j4], w.template trace_pack4<IB>(best[j4], best[j4 + 1], best[j4 + 2], best[j4 + 3]),
                 col[j4] < (uint32_t)a.trace_stride);
      for (int j = 0; j < CPL; j++) {
        rrun[j] = w.umax(rrun[j], pend[j]);       // rows <= r-1 become candidates for row r+1
        pend[j] = (Snew[j] << SH) + RK;
        q[j] = (Snew[j] << SH) + QC[j];
        Sb[j] = Snew[j];
      }
      if (LASTCOL) last_col_update(r);
    }

    // ---- 5. max_sg_score (src/mia.c:1278-1302): last row, first maximum
    AlignResult res;
    res.abr = 0;
    res.abc = 0;
    res.aer = len2 - 1;
    if (LASTCOL) {
      const int owner = (len1 - 1) / CPL;
      res.score = (int32_t)(w.lane_val(colbest, owner) - OFF);
      res.aer = (int32_t)w.lane_val(colrow, owner);
      res.aec = len1 - 1;
    } else {
      U m = U(0u);
      for (int j = 0; j < CPL; j++) m = w.umax(m, w.sel(col[j] < (uint32_t)len1, Sb[j], U(0u)));
      const uint32_t bestb = w.reduce_max(m);
      U cmin = U(0xFFFFFFFFu);
      for (int j = CPL - 1; j >= 0; j--) cmin = w.sel((col[j] < (uint32_t)len1) & (Sb[j] == bestb), col[j], cmin);
      res.score = (int32_t)(bestb - OFF);
      res.aec = (int32_t)w.reduce_min(cmin);
    }
    w.tr_fence();

    // ---- 6. traceback (src/mia.c:612-637,1440-1497), diagonal runs of up to 64 cells per step.
    // cols_out[row] = window column aligned to that read base, COL_INSERT, or COL_CLIP.
    int r = res.aer, c = res.aec;
    uint32_t status = ST_OK;
    int aln_cols = 0;
    for (int guard = 0; guard < ((a.dbg & 2u) ? 0 : 4 * MAX_READ + 8); guard++) {
      // lane i inspects cell (r-i, c-i)
      U ri = U((uint32_t)r) - lane, ci = U((uint32_t)c) - lane;
      M inside = (lane <= (uint32_t)r) & (lane <= (uint32_t)c);
      U tb = w.tr_r8(ri * a.trace_stride + ci, inside);
      U ty = tb >> 6, ln = tb & 63u;
      // the reference reads T==0 as a diagonal step even when it encodes a gap that
      // started in column/row 0 (src/mia.c:619,1460): gap source index 0 <=> len == c-1 / r-1
      M colgap0 = (ty == U(TR_COLGAP)) & (ln + 1u == ci) & (ln < U(TR_LEN_SAT));
      M rowgap0 = (ty == U(TR_ROWGAP)) & (ln + 1u == ri) & (ln < U(TR_LEN_SAT));
      M plain_diag = (ty == U(TR_DIAG)) | colgap0 | rowgap0;
      M terminal = (ri == U(0u)) | (ci == U(0u)) | (ty == U(TR_START));   // T==col or T==-row
      M stop_here = (!inside) | terminal | (!plain_diag);
      uint64_t bal = w.ballot(stop_here);
      int f = bal ? __builtin_ctzll(bal) : WAVE;    // first lane that is not a plain diagonal step
      // lanes before f (and f itself if it is inside) are aligned pairs on the path
      int npairs = f < WAVE ? f + 1 : WAVE;
      if (f < WAVE && !w.lane_bit(inside, f)) npairs = f;   // cannot happen (terminal fires first), kept for safety
      w.gstore_i16(a.cols_out, ri, ci, lane < (uint32_t)npairs);
      aln_cols += npairs;
      if (f == WAVE) { r -= WAVE; c -= WAVE; continue; }
      const uint32_t fty = w.lane_val(ty, f), fln = w.lane_val(ln, f);
      const bool fterm = w.lane_bit(terminal, f);
      const int fr = r - f, fc = c - f;
      if (fterm) { res.abr = fr; res.abc = fc; break; }
      if (fln >= TR_LEN_SAT) { status |= ST_ESCAPE; res.abr = fr; res.abc = fc; break; }
      if (fty == TR_COLGAP) {              // rows consecutive, fln reference columns skipped (deletion in the read)
        r = fr - 1; c = fc - 1 - (int)fln;
        aln_cols += (int)fln;
      } else {                             // TR_ROWGAP: fln read bases have no column (insert)
        U rr = U((uint32_t)(fr - 1)) - lane;
        w.gstore_i16(a.cols_out, rr, U((uint32_t)(uint16_t)COL_INSERT), lane < fln);
        r = fr - 1 - (int)fln; c = fc - 1;
        aln_cols += (int)fln;
      }
    }
    // rows before the alignment start are soft-clipped
    for (int r0 = 0; r0 < res.abr; r0 += WAVE) {
      U rr = lane + (uint32_t)r0;
      w.gstore_i16(a.cols_out, rr, U((uint32_t)(uint16_t)COL_CLIP), rr < (uint32_t)res.abr);
    }
    if (aln_cols > 2 * MAX_READ) status |= ST_TOO_LONG;
    res.status = status;
    return res;
  }
};

}  // namespace mia
