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
//   * the trace is one byte per cell in LDS; the traceback follows diagonal
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
  uint32_t lds_sub;           // LDS byte offset: int16 sub[len2][5]
  uint32_t lds_trace;         // LDS byte offset: trace bytes [len2][trace_stride]
  uint32_t trace_stride;      // bytes per trace row, multiple of 4, >= len1
  int16_t* cols_out;          // script: window column per read row            [global]
};

struct AlignResult {          // wave-uniform
  int32_t score, abc, abr, aec;
  uint32_t status;
};

template <class P, int CPL>
struct WindowAligner {
  typedef typename P::U U;
  typedef typename P::M M;

  MIA_HD static inline __attribute__((always_inline)) AlignResult run(P& w, const AlignArgs& a) {
    const int SH = a.pk.sh, IB = a.pk.ib;
    const uint32_t IDXM = a.pk.idxm;
    const uint32_t OFF = (uint32_t)a.pk.off;
    const uint32_t UNAV = ((uint32_t)a.pk.unavail << SH) | IDXM;
    const U lane = w.lane();
    const int len1 = a.len1, len2 = a.len2;

    // ---- 1. substitution table sub[r][code1] = sm[depth(r)][code1][read[r]] (src/mia.c:792-795)
    for (int e0 = 0; e0 < len2 * 5; e0 += WAVE) {
      U e = lane + (uint32_t)e0;
      M ok = e < (uint32_t)(len2 * 5);
      U r = w.udiv5(e);
      U c1 = e - r * 5u;
      U byte = w.gload_u8(a.read_packed, r >> 1, ok);
      U c2 = (byte >> ((r & 1u) << 2)) & 15u;
      U d = w.depth(r, (uint32_t)len2);
      U v = w.gload_i32(a.pssm, (d * 5u + c1) * 5u + c2, ok);
      w.lds_w16(U(a.lds_sub) + e * 2u, v, ok);
    }
    w.lds_fence();

    // ---- 2. per-lane column constants
    U col[CPL], sub_addr[CPL], KC[CPL], QC[CPL];
    for (int j = 0; j < CPL; j++) {
      col[j] = lane * (uint32_t)CPL + (uint32_t)j;
      M in = col[j] < (uint32_t)len1;
      U code = w.sel(in, w.gload_u8(a.ref_codes, U((uint32_t)a.ref_start) + col[j], in), U(4u));
      sub_addr[j] = U(a.lds_sub) + code * 2u;
      // key -> column-gap candidate: value -= GOP + GEP*(c-1); prio = 2; idx -> len = c-1-k
      KC[j] = (U(0u) - ((U((uint32_t)GOP) + (col[j] - 1u) * (uint32_t)GEP) << SH)) + (2u << IB) + (col[j] - 1u - IDXM);
      // S -> key of this column for the next row: value = S + GEP*c + off, idx = IDXM - c
      QC[j] = ((col[j] * (uint32_t)GEP + OFF) << SH) + (U(IDXM) - col[j]);
    }
    const uint32_t WDC = (OFF << SH) + (TR_DIAG << IB);

    // ---- 3. row 0 (src/mia.c:769-785): S = sub, T = 0
    U S[CPL], q[CPL], rrun[CPL], pend[CPL];
    {
      const uint32_t rk = ((0u * GEP + OFF) << SH) + (IDXM - 0u);
      U tw = U(0u);
      for (int j = 0; j < CPL; j++) {
        S[j] = w.lds_ri16(sub_addr[j]);
        q[j] = (S[j] << SH) + QC[j];
        pend[j] = (S[j] << SH) + rk;
        rrun[j] = U(UNAV);
        tw = tw | (U(TR_DIAG << 6) << (8 * (j & 3)));
        if ((j & 3) == 3 || j == CPL - 1) {
          w.lds_w32(U(a.lds_trace) + col[j & ~3] , tw, col[j & ~3] < (uint32_t)a.trace_stride);
          tw = U(0u);
        }
      }
    }

    // ---- 4. rows 1 .. len2-1
    for (int r = 1; r < len2; r++) {
      const int32_t fresh = a.sg5 ? -(GOP + GEP * (r + 1)) : 0;            // src/mia.c:877-880
      const uint32_t WS = ((uint32_t)(fresh + (int32_t)OFF)) << SH;         // prio 0, len 0
      const uint32_t KR = (0u - ((uint32_t)(GOP + GEP * (r - 1)) << SH)) + (TR_ROWGAP << IB) + ((uint32_t)(r - 1) - IDXM);
      const uint32_t RK = (((uint32_t)(GEP * r) + OFF) << SH) + (IDXM - (uint32_t)r);
      const uint32_t row_sub = (uint32_t)r * 10u;
      const uint32_t row_tr = a.lds_trace + (uint32_t)r * a.trace_stride;

      // neighbours: diag of the first owned column, its best_gap_row state, and the
      // two keys left of the lane for the shifted prefix scan
      U dleft = w.shr1(S[CPL - 1], U((uint32_t)fresh));   // column 0: "diag" = fresh  => S = sub + fresh, T = 0 (src/mia.c:805-822)
      U rleft = w.shr1(rrun[CPL - 1], U(UNAV));
      U u0 = w.shr1(q[CPL - 2 >= 0 ? CPL - 2 : 0], U(UNAV));
      U u1 = w.shr1(q[CPL - 1], U(UNAV));
      // g[j] = max key over columns <= c_j - 2 inside {two left keys, own keys}
      U g[CPL];
      g[0] = u0;
      g[1] = w.umax(u0, u1);
      for (int j = 2; j < CPL; j++) g[j] = w.umax(g[j - 1], q[j - 2]);
      U incl = w.scan_max(g[CPL - 1]);
      U excl = w.shr1(incl, U(UNAV));

      U Snew[CPL];
      U tw = U(0u);
      for (int j = 0; j < CPL; j++) {
        U diag = (j == 0) ? dleft : S[j - 1];
        U rl = (j == 0) ? rleft : rrun[j - 1];
        U cand = w.umax(excl, g[j]);
        U Wd = (diag << SH) + WDC;
        U Wc = cand + KC[j];
        U Wr = rl + KR;
        U best = w.umax(w.umax3(Wd, Wc, Wr), U(WS));
        U sub = w.lds_ri16(sub_addr[j] + row_sub);
        M is_start = (best >> IB & 3u) == U(TR_START);
        Snew[j] = (best >> SH) - OFF + w.sel(is_start, U(0u), sub);       // start drops the substitution score (src/mia.c:916-917)
        U tb = ((best >> IB & 3u) << 6) | w.umin(best & IDXM, U(TR_LEN_SAT));
        tw = tw | (tb << (8 * (j & 3)));
        if ((j & 3) == 3 || j == CPL - 1) {
          w.lds_w32(U(row_tr) + col[j & ~3], tw, col[j & ~3] < (uint32_t)a.trace_stride);
          tw = U(0u);
        }
      }
      for (int j = 0; j < CPL; j++) {
        rrun[j] = w.umax(rrun[j], pend[j]);       // rows <= r-1 become candidates for row r+1
        pend[j] = (Snew[j] << SH) + RK;
        q[j] = (Snew[j] << SH) + QC[j];
        S[j] = Snew[j];
      }
    }

    // ---- 5. max_sg_score (src/mia.c:1278-1302): last row, first maximum
    AlignResult res;
    {
      U m = U(0u);  // biased signed max
      for (int j = 0; j < CPL; j++) m = w.umax(m, w.sel(col[j] < (uint32_t)len1, S[j] ^ 0x80000000u, U(0u)));
      uint32_t bestb = w.reduce_max(m);
      U cmin = U(0xFFFFFFFFu);
      for (int j = CPL - 1; j >= 0; j--)
        cmin = w.sel((col[j] < (uint32_t)len1) & ((S[j] ^ 0x80000000u) == U(bestb)), col[j], cmin);
      res.score = (int32_t)(bestb ^ 0x80000000u);
      res.aec = (int32_t)w.reduce_min(cmin);
    }
    w.lds_fence();

    // ---- 6. traceback (src/mia.c:612-637,1440-1497), diagonal runs of up to 64 cells per step.
    // cols_out[row] = window column aligned to that read base, COL_INSERT, or COL_CLIP.
    int r = len2 - 1, c = res.aec;
    uint32_t status = ST_OK;
    int aln_cols = 0;
    for (int guard = 0; guard < 4 * MAX_READ + 8; guard++) {
      // lane i inspects cell (r-i, c-i)
      U ri = U((uint32_t)r) - lane, ci = U((uint32_t)c) - lane;
      M inside = (lane <= (uint32_t)r) & (lane <= (uint32_t)c);
      U tb = w.lds_r8(U(a.lds_trace) + ri * a.trace_stride + ci, inside);
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
