// align_body_quad_plain.h -- the windowed DP for FOUR reads per wavefront, VALUES ONLY, plus an exact proof that the
// reference's traceback is the pure diagonal; reads for which the proof does not hold are handed to the packed
// kernel of align_body_quad.h (which stores a trace and walks it).
//
// Why: most reads of an iteration re-align without gaps or soft clips.  For those the whole answer is (score, end
// column): the path is the diagonal through the end cell.  Computing values only lets the two arg-max registers of
// dyn_prog collapse into one key per cell (pass1_body.h, run_plain):  kk(r,c) = S(r,c) + GEP*(r+c),
//     kk(r,c) = max3(kk(r-1,c-1) + GOP, CB, RB) + sub(r,c) + (2*GEP - GOP),
//     a new start iff max3 < GEP*(c-3), with key GEP*(c-1) - GOP                      (both row independent),
// 7.5 vector operations per cell instead of 14.5 + trace packing, and no trace traffic at all.
//
// The proof (R = last row, d = aec - R, cells c_r = r + d, r0 = max(0,-d); D(r) = score of the pure diagonal path
// down to row r: D(r0) = sub (row 0) or sub - P(r0+1) (column 0, src/mia.c:805-822), D(r) = D(r-1) + sub(r,c_r)):
//   (i)  best == D(R)   and   (ii)  D(r-1) >= -P(r+1) for every r in (r0, R]
// imply that find_align_begin walks exactly that diagonal:
//   * (ii) gives S(r,c_r) >= D(r) for all r by induction: the diagonal candidate S(r-1,c_{r-1}) >= D(r-1) is not
//     below the start value -P(r+1), a start needs to be STRICTLY better (src/mia.c:910-917), so the cell is not a
//     start and S(r,c_r) >= sub + S(r-1,c_{r-1}) >= D(r);
//   * let k be the largest row whose cell did not take the diagonal.  Below it every cell took it, so
//     best = S(k,c_k) + sum_{r>k} sub and with (i) S(k,c_k) = D(k).  A start at k is excluded by (ii).  A gap is taken
//     only if strictly better than the diagonal (diag >= gapc >= gapr on ties): S(k,c_k) > sub + S(k-1,c_{k-1}) >=
//     D(k), a contradiction.  Hence no such k above r0: every cell on the diagonal has T == 0 and the walk ends at
//     row 0 or column 0 (T == col there).
// (i) and (ii) cost one pass of table look-ups over the rows per read, with a wave prefix sum.
#pragma once
#include "align_body.h"
#include "align_body_quad.h"
#include "mia_layout.h"

namespace mia {

struct QuadPlainArgs {            // wave-uniform; [g] = read of lane row g, len1 == 0 marks an empty slot
  const uint8_t* ref_codes;
  const uint8_t* packed;          // base of the packed read store
  const int32_t* pssm2;           // forward PSSM followed by the reverse-complemented one
  int32_t len2;                   // common read length of the quad
  int32_t ref_start[Q_G], len1[Q_G];
  uint32_t roff[Q_G], rc[Q_G];
  uint32_t lds_sub;               // LDS: 4 tables int16 sub[5][rows], q_sub_bytes apart
  int16_t* cols_out[Q_G];
};

struct QuadPlainResult {          // wave-uniform
  int32_t score, aec, abc, abr;
  int32_t proven;                 // 1: the alignment is the pure diagonal (abr, abc) .. (len2-1, aec); 0: needs the trace kernel
};

template <class P>
struct QuadPlainAligner {
  typedef typename P::U U;
  typedef typename P::M M;
  static constexpr int CPL = Q_CPL;
  static constexpr uint32_t PB = 1u << 30;                       // key bias: 0 = no candidate
  static constexpr uint32_t CK = (uint32_t)(2 * GEP - GOP);

  MIA_HD static inline __attribute__((always_inline)) void run(P& w, const QuadPlainArgs& a, QuadPlainResult* res /* [Q_G] */) {
    const U lane = w.lane();
    const U grp = lane >> 4, gl = lane & 15u;
    const int len2 = a.len2;
    const uint32_t RS2 = q_sub_stride(len2);
    U len1v = U(0u), startv = U(0u), roffv = U(0u), pmoff = U(0u);
    for (int g = 0; g < Q_G; g++) {
      M mine = grp == (uint32_t)g;
      len1v = w.sel(mine, U((uint32_t)a.len1[g]), len1v);
      startv = w.sel(mine, U((uint32_t)a.ref_start[g]), startv);
      roffv = w.sel(mine, U(a.roff[g]), roffv);
      pmoff = w.sel(mine, U(a.rc[g] ? (uint32_t)PSSM_WORDS : 0u), pmoff);
    }
    const U subbase = U(a.lds_sub) + grp * q_sub_bytes(len2);

    // ---- substitution tables, 16 lanes per read (as align_body_quad.h)
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

    U sub_addr[CPL], STc[CPL], STK[CPL];
    const U col0 = gl * (uint32_t)CPL;
    for (int j = 0; j < CPL; j++) {
      U col = col0 + (uint32_t)j;
      M in = col < len1v;
      U code = w.sel(in, w.gload_u8(a.ref_codes, startv + col, in), U(4u));
      sub_addr[j] = w.lds_abs(subbase + code * RS2);
      STc[j] = col * (uint32_t)GEP + (PB - 3u * (uint32_t)GEP);    // a new start iff max3 < this
      STK[j] = STc[j] + CK;                                        // key of a new start
    }
    // columns at or beyond len1 (only at the right end of the window) never feed a column to their left; they are
    // computed like any other and left out of the final maximum
    U kk[CPL], rrun[CPL];
    for (int j = 0; j < CPL; j++) {     // row 0 (src/mia.c:769-785): S = sub
      kk[j] = w.lds_ri16a(sub_addr[j], 0u) + (STc[j] + 3u * (uint32_t)GEP);
      rrun[j] = U(0u);
    }
    const U zero = U(0u);
    // column 0 takes "diag" = -P(r+1) (src/mia.c:805-822): as a key of the virtual cell (r-1,-1) that is a constant
    const U fillD = U(PB - (uint32_t)GOP - 3u * (uint32_t)GEP);

    auto do_row = [&](const uint32_t off) __attribute__((always_inline)) {
      U dleft = w.rshr1(kk[CPL - 1], fillD);
      U rleft = w.rshr1_max(rrun[CPL - 1], zero);
      U u0 = w.rshr1_max(kk[CPL - 2], zero);
      U u1 = w.rshr1_max(kk[CPL - 1], zero);
      U tot = w.umax3(w.umax3(u0, u1, kk[0]), w.umax3(kk[1], kk[2], kk[3]),
                      w.umax3(w.umax3(kk[4], kk[5], kk[6]), w.umax3(kk[7], kk[8], kk[9]), kk[10]));
      U excl = w.rshr1_max(w.rscan_max(tot), zero);
      U subv[CPL];
      for (int j = 0; j < CPL; j++) subv[j] = w.lds_ri16a(sub_addr[j], off);
      U cand[CPL];
      cand[0] = w.umax(excl, u0);
      cand[1] = w.umax(cand[0], u1);
      for (int j = 2; j < CPL; j++) cand[j] = w.umax(cand[j - 1], kk[j - 2]);
      for (int j = CPL - 1; j >= 0; j--) {
        U kd = (j == 0) ? dleft : kk[j - 1];
        U rl = (j == 0) ? rleft : rrun[j - 1];
        U m3 = w.umax3(kd + (uint32_t)GOP, cand[j], rl);
        U kn = w.sel(m3 < STc[j], STK[j], w.add3(m3, subv[j], CK));
        rrun[j] = w.umax(rrun[j], kk[j]);
        kk[j] = kn;
      }
    };
    for (int j = 0; j < CPL; j++) sub_addr[j] = sub_addr[j] + 2u;      // -> row 1
    {
      int r = 1;
      for (; r + 1 < len2; r += 2) {        // straight-line row pairs: every column is updated in place, no copies between rows
        do_row(0u);
        do_row(2u);
        for (int j = 0; j < CPL; j++) { sub_addr[j] = sub_addr[j] + 4u; w.keep(sub_addr[j]); }
      }
      if (r < len2) do_row(0u);
    }

    // ---- max_sg_score (src/mia.c:1278-1302): last row, first maximum.  S + PB = kk - GEP*(len2-1) - GEP*c
    U m = zero, sc[CPL];
    for (int j = 0; j < CPL; j++) {
      sc[j] = w.sel((col0 + (uint32_t)j) < len1v, kk[j] - (STc[j] - (PB - 3u * (uint32_t)GEP)) - (uint32_t)((len2 - 1) * GEP), zero);
      m = w.umax(m, sc[j]);
    }
    const U bestv = w.row_last(w.rscan_max(m));
    U cmin = U(0x7FFFFFFFu);
    for (int j = CPL - 1; j >= 0; j--) cmin = w.sel(((col0 + (uint32_t)j) < len1v) & (sc[j] == bestv), col0 + (uint32_t)j, cmin);
    const U aecv = ~w.row_last(w.rscan_max(~cmin));

    // ---- the diagonal proof, one read after the other, 64 rows at a time
    const int R = len2 - 1;
    for (int gi = 0; gi < Q_G; gi++) {
      QuadPlainResult& rs = res[gi];
      rs.score = 0; rs.aec = 0; rs.abc = 0; rs.abr = 0; rs.proven = 0;
      if (a.len1[gi] <= 0) continue;
      const int32_t best = (int32_t)(w.lane_val(bestv, gi * Q_LPR) - PB);
      const int aec = (int32_t)w.lane_val(aecv, gi * Q_LPR);
      const int d = aec - R, r0 = d < 0 ? -d : 0;
      rs.score = best; rs.aec = aec; rs.abr = r0; rs.abc = d < 0 ? 0 : d;
      const uint32_t tb = a.lds_sub + (uint32_t)gi * q_sub_bytes(len2);
      int32_t carry = 0;          // D(r) of the last row of the previous block of 64
      bool ok = true;
      for (int rb = r0; rb <= R && ok; rb += WAVE) {
        U r = lane + (uint32_t)rb;
        M in = r <= (uint32_t)R;
        U c = r + (uint32_t)d;                                   // (wraps are confined to lanes with !in)
        U code = w.gload_u8(a.ref_codes, U((uint32_t)a.ref_start[gi]) + c, in);
        U s = w.sel(in, w.lds_ri16o(U(tb) + code * RS2 + r * 2u, 0u), zero);
        if (rb == r0 && d < 0) s = s + w.sel(lane == 0u, U((uint32_t)(-(GOP + GEP * (r0 + 1)))), zero);   // column 0: sub - P(r+1)
        U D = w.scan_add(s) + (uint32_t)carry;                   // D(r), inclusive
        // (ii) D(r-1) >= -P(r+1) for r in (r0, R]: lane holds D(r); test it against the NEXT row's start value
        M bad = in & (r < (uint32_t)R) & ((D + 0x80000000u) < (U((uint32_t)(-(GOP + 2 * GEP))) - r * (uint32_t)GEP + 0x80000000u));
        if (w.ballot(bad)) ok = false;
        const int last = (R - rb) < (WAVE - 1) ? (R - rb) : (WAVE - 1);
        carry = (int32_t)w.lane_val(D, last);
      }
      if (ok && carry == best) {
        rs.proven = 1;
        for (int rb = 0; rb <= R; rb += WAVE) {                  // the script: clipped rows, then one column per row
          U r = lane + (uint32_t)rb;
          U v = w.sel(r < (uint32_t)r0, U((uint32_t)(uint16_t)COL_CLIP), (r + (uint32_t)d) & 0xFFFFu);
          w.gstore_i16(a.cols_out[gi], r, v, r <= (uint32_t)R);
        }
      }
    }
  }
};

}  // namespace mia
