// pass1_body.h -- pass 1 of MIA for ONE read on ONE wavefront: semi-global DP of the
// read against the WHOLE wrapped reference on both strands, with the k-mer column
// masks, strand choice and traceback -- sg_align of the reference
// (/root/reference/src/mia.c:1500-1599, dyn_prog :740-981 with align_mask,
// max_sg_score :1278-1302, find_align_begin :612-637).  Same policy scheme as
// align_body.h: compiled for gfx950 (DevWave) and, for tests, for the CPU emulation.
//
// The reference fills a len2 x wrap matrix per strand (13.5 MB for mt311) and walks it
// back.  Here the columns are swept in chunks of 256 (lane l owns 4 columns, as in the
// window kernel) and only five words per row cross a chunk boundary:
//     D  biased score of the chunk's last column (diagonal input of the next chunk)
//     R  best_gap_row state of that column
//     U0,U1  keys of its last two columns (the "k <= c-2" shift of best_gap_col)
//     E  running best_gap_col key over everything further left
// Phase A sweeps all chunks without a trace, records (best, first arg-max) of the last
// row and checkpoints the five carry arrays of every chunk in global memory.  Phase B
// picks the strand (forward only if strictly better, :1549).  Phase C recomputes just
// the chunks the optimal path visits (1-3) from their checkpoints, now with a 16-bit
// trace, and follows the path with the ballot walk.  Every cell value is identical to
// the reference's; nothing is approximated.
//
// Masked columns (align_mask == 0, k-mer filter) hold HIM = INT_MIN/2 in the reference.
// Anything derived from HIM loses against "start a new alignment" (>= -52 400), so such
// cells are simply absent here: biased score 0, keys UNAV.  A column k is offered to
// best_gap_col only while column k+2 is processed (src/mia.c:838-843), i.e. only if
// mask[k+2] -- the `qen` flag.  Candidates further left than 768 columns can never beat
// a new start (P(767) > len2*max_sub + P(len2+1), checked by the host) and are dropped,
// which keeps column indices relative to the chunk in a 10-bit field.
#pragma once
#include "mia_layout.h"

namespace mia {

constexpr int P1_IB = 10, P1_SH = 12;
constexpr uint32_t P1_IDXM = 1023u;
// A chunk is 64*CPL columns (lane l owns CPL adjacent ones).  Two widths are compiled:
//   CPL = 4   256-column chunks, keys remember candidates up to 768 columns left of the chunk  (k-mer masks: fine skipping)
//   CPL = 12  768-column chunks, horizon 256 columns                                           (unmasked sweep: 3x less per-row overhead)
// chunk + horizon always fill the 10-bit index field.
MIA_HD constexpr int p1_ch(int cpl) { return WAVE * cpl; }
MIA_HD constexpr int p1_rel(int cpl) { return (int)P1_IDXM + 1 - WAVE * cpl; }
constexpr int P1_CPL_NARROW = 4, P1_CPL_WIDE = 12;
constexpr int P1_CH_MAX = WAVE * P1_CPL_WIDE;

struct Pass1Args {               // wave-uniform
  const uint8_t* ref_codes[2];   // [0] forward wrapped reference, [1] reverse complement, wrapped   [global]
  int32_t len1;                  // columns (wrap_seq_len, or seq_len when linear)
  const uint8_t* read_packed;    // 4-bit codes                                                     [global]
  int32_t len2;
  const int32_t* pssm;           // ancsubmat for BOTH strands in pass 1 (src/mia_main.c:788-789)    [global]
  PackParams pk;                 // from make_pack_params(1024, ...): ib = 10
  uint32_t lds_sub;              // LDS: int16 sub[5][rows padded to even]
  uint32_t lds_carry;            // LDS: 5 arrays [P1_ROWS] u32 : D, R, U0, U1, E
  uint32_t lds_mask[2];          // LDS: column bit masks per strand (only if masked != 0)
  int32_t masked;                // 0 = all columns open
  int32_t plain;                 // unmasked strands: sweep with plain keys (run_plain)
  uint32_t* ckpt;                // global: [2 strands][nchunks][5][rows_p] u32 checkpoints
  int32_t rows_p;                // row stride of the checkpoint arrays (>= len2)
};

struct Pass1Result {             // wave-uniform
  int32_t best[2];               // per strand; INT32_MIN/2 if the strand has no open column
  int32_t strand, score, aec, abc, abr;
  uint32_t status;
};

template <class P, int CPL_>
struct Pass1Aligner {
  typedef typename P::U U;
  typedef typename P::M M;
  static constexpr int CPL = CPL_, IB = P1_IB, SH = P1_SH, CH = WAVE * CPL_, REL = (int)P1_IDXM + 1 - CH;
  static constexpr uint32_t IDXM = P1_IDXM;
  static constexpr uint32_t WDC = TR_DIAG << IB;

  // move a carried key to a chunk base that is `cols` columns further right
  MIA_HD static inline U rebase(P& w, const U& e, uint32_t cols, uint32_t unav) {
    U idx = (e & IDXM) + cols;
    U moved = e - ((cols * (uint32_t)GEP) << SH) + cols;
    return w.sel((idx > IDXM) | (e <= unav), U(unav), moved);
  }

  MIA_HD static inline M maskbit(P& w, const Pass1Args& a, int strand, const U& gcol) {
    M in = gcol < (uint32_t)a.len1;
    if (!a.masked) return in;
    U word = w.lds_r32m(U(a.lds_mask[strand]) + (gcol >> 5) * 4u, in);
    return in & (((word >> (gcol & 31u)) & 1u) != 0u);
  }

  // One chunk, all rows.  Carries are consumed from / produced into the LDS arrays in place.
  // Returns through (cbest, ccol) the maximum state word of the last row and its first column.
  //
  // The score state is kept SHIFTED with the diagonal priority already in place, exactly as in
  // align_body_quad.h: Sd = ((S + off) << SH) | (TR_DIAG << IB) is the diagonal candidate of the
  // cell below-right; an absent (masked) cell holds 0, which loses against everything.
  // MASKED = false is the fast path for chunks in which every column exists and is open.
  template <bool TRACE, bool MASKED>
  MIA_HD static inline void chunk(P& w, const Pass1Args& a, int strand, int ch, uint32_t& cbest, uint32_t& ccol) {
    const uint32_t OFF = (uint32_t)a.pk.off;
    const uint32_t UNAV = ((uint32_t)a.pk.unavail << SH) | IDXM;
    const uint32_t HI = ~((1u << SH) - 1u);
    const U unav = U(UNAV), wdcv = U(WDC);
    const U lane = w.lane();
    const int len2 = a.len2;
    const uint32_t base = (uint32_t)ch * CH;
    const uint32_t RS2 = (uint32_t)((len2 + 1) & ~1) * 2u;
    const uint32_t cD = a.lds_carry, cR = cD + MAX_READ * 4, cU0 = cR + MAX_READ * 4, cU1 = cU0 + MAX_READ * 4, cE = cU1 + MAX_READ * 4;
    const M last_lane = lane == 63u;
    const M all = lane < 64u;

    U sub_addr[CPL];
    M valid[CPL], qen[CPL];
    const U crel0 = lane * (uint32_t)CPL, cp0 = crel0 + (uint32_t)REL;   // cp: shifted column index used inside keys
    for (int j = 0; j < CPL; j++) {
      U gcol = crel0 + (base + (uint32_t)j);
      if (MASKED) {
        valid[j] = maskbit(w, a, strand, gcol);
        qen[j] = valid[j] & ((gcol == 0u) | maskbit(w, a, strand, gcol + 2u));
      } else {
        valid[j] = all; qen[j] = all;
      }
      M in = gcol < (uint32_t)a.len1;
      U code = w.sel(in, w.gload_u8(a.ref_codes[strand], gcol, in), U(4u));
      sub_addr[j] = w.lds_abs(U(a.lds_sub) + code * RS2);
    }
    // key -> column-gap candidate: value -= GOP + GEP*(cp-1); prio = 2; idx -> len = cp-1-k.  Affine in the column.
    const U KC0 = (U(0u) - ((U((uint32_t)GOP) + (cp0 - 1u) * (uint32_t)GEP) << SH)) + (TR_COLGAP << IB) + (cp0 - 1u - IDXM);
    const uint32_t KCD = 1u - ((uint32_t)GEP << SH);
    // state -> best_gap_col key of its column: value += GEP*cp, idx = IDXM - cp, diagonal bits removed.  Affine too.
    const U QC0 = ((cp0 * (uint32_t)GEP) << SH) + (U(IDXM) - cp0) - WDC;
    const uint32_t QCD = ((uint32_t)GEP << SH) - 1u;

    U Sd[CPL], q[CPL], rrun[CPL];
    for (int j = 0; j < CPL; j++) {     // row 0 (src/mia.c:769-785)
      U s0 = ((w.lds_ri16a(sub_addr[j], 0u) + OFF) << SH) | WDC;
      Sd[j] = MASKED ? w.sel(valid[j], s0, U(0u)) : s0;
      U k0 = w.add3(s0, QC0, (uint32_t)j * QCD);
      q[j] = MASKED ? w.sel(qen[j], k0, unav) : k0;
      rrun[j] = unav;
    }
    if (TRACE) {
      const U d2 = U(WDC * 0x00010001u);
      for (int j = 0; j < CPL; j += 2) w.tr_w32(lane * (uint32_t)(CPL * 2) + (uint32_t)(j * 2), d2, all);
    }

    auto do_row = [&](const int r, const uint32_t off) __attribute__((always_inline)) {
      const int32_t fresh = -(GOP + GEP * (r + 1));                         // sg5 = 1 (src/mia.c:1535)
      const uint32_t freshb = (uint32_t)(fresh + (int32_t)OFF);
      const uint32_t WS = w.sconst(freshb << SH);
      const uint32_t KR = w.sconst((0u - ((uint32_t)(GOP + GEP * (r - 1)) << SH)) + (TR_ROWGAP << IB) + ((uint32_t)(r - 1) - IDXM));
      const uint32_t RKP = w.sconst(((uint32_t)(GEP * (r - 1)) << SH) + (IDXM - (uint32_t)(r - 1)) - WDC);   // key constant of row r-1

      // carries of this row from the chunk on the left (same value in every lane) ...
      const U inD = w.lds_r32(U(cD + (uint32_t)r * 4u)), inR = w.lds_r32(U(cR + (uint32_t)r * 4u));
      const U inU0 = w.lds_r32(U(cU0 + (uint32_t)r * 4u)), inU1 = w.lds_r32(U(cU1 + (uint32_t)r * 4u));
      const U inE = w.lds_r32(U(cE + (uint32_t)r * 4u));
      // ... and ours for the chunk on the right (lane 63; state BEFORE this row is computed)
      w.lds_w32(U(cD + (uint32_t)r * 4u), Sd[CPL - 1], last_lane);
      w.lds_w32(U(cR + (uint32_t)r * 4u), rrun[CPL - 1], last_lane);
      w.lds_w32(U(cU0 + (uint32_t)r * 4u), q[CPL - 2], last_lane);
      w.lds_w32(U(cU1 + (uint32_t)r * 4u), q[CPL - 1], last_lane);

      U dleft = w.shr1(Sd[CPL - 1], inD);
      U rleft = w.shr1(rrun[CPL - 1], inR);
      U u0 = w.shr1(q[CPL - 2], inU0);
      U u1 = w.shr1(q[CPL - 1], inU1);
      // lane total of the keys it offers to the lanes on its right (columns <= own last column - 2 ... plus the two shifted in)
      U tot = w.umax(u0, u1);
      {
        int j = 0;
        for (; j + 1 < CPL - 2; j += 2) tot = w.umax3(tot, q[j], q[j + 1]);
        if (j < CPL - 2) tot = w.umax(tot, q[j]);
      }
      U incl = w.scan_max(tot);
      U excl = w.umax(w.shr1_max(incl, unav), inE);
      w.lds_w32(U(cE + (uint32_t)r * 4u), w.umax(incl, inE), last_lane);
      U subv[CPL];    // (loaded in the block that uses them: the sign extension then folds into ds_read_i16)
      for (int j = 0; j < CPL; j++) subv[j] = w.lds_ri16a(sub_addr[j], off);
      // cand[j] = best key over all columns <= c_j - 2, one running chain with the prefix folded in
      U cand[CPL];
      cand[0] = w.umax(excl, u0);
      cand[1] = w.umax(cand[0], u1);
      for (int j = 2; j < CPL; j++) cand[j] = w.umax(cand[j - 1], q[j - 2]);

      const U trrow = lane * (uint32_t)(CPL * 2) + (uint32_t)r * (uint32_t)(CH * 2);
      U bodd = U(0u);
      for (int j = CPL - 1; j >= 0; j--) {     // right to left: every column is updated in place
        U Wd = (j == 0) ? dleft : Sd[j - 1];
        U rl = (j == 0) ? rleft : rrun[j - 1];
        U Wc = w.add3(cand[j], KC0, (uint32_t)j * KCD);
        U Wr = rl + KR;
        U m3 = w.umax3(Wd, Wc, Wr);
        U sub = subv[j];
        // start (only if strictly better than the other three) drops the substitution score (src/mia.c:910-917)
        U best, snew;
        if (TRACE) {
          best = w.umax(m3, U(WS));
          snew = w.template shl_addc<SH>(w.sel(m3 < WS, U(0u), sub), w.bfi(HI, best, wdcv));
        } else {   // one operation less when the winning word itself is not needed
          best = U(0u);
          snew = w.bfi(HI, w.sel(m3 < WS, U(WS), w.template shl_addc<SH>(sub, m3)), wdcv);
        }
        U rk = Sd[j] + RKP;                    // row r-1 becomes a best_gap_row candidate for row r+1
        U qk = w.add3(snew, QC0, (uint32_t)j * QCD);
        if (MASKED) {
          snew = w.sel(valid[j], snew, U(0u));
          rk = w.sel(valid[j], rk, unav);
          qk = w.sel(qen[j], qk, unav);
        }
        rrun[j] = w.umax(rrun[j], rk);
        Sd[j] = snew;
        q[j] = qk;
        if (TRACE) {
          // trace cell = low 12 bits of the winning word ([prio:2|len:10]) in a 16-bit cell
          if (j & 1) bodd = best & 0xFFFu;
          else w.tr_w32(trrow + (uint32_t)(j * 2), (best & 0xFFFu) | (bodd << 16), all);
        }
      }
    };
    // rows in pairs: the substitution-table address of a column is bumped once per two rows
    for (int j = 0; j < CPL; j++) sub_addr[j] = sub_addr[j] + 2u;      // -> row 1
    {
      int r = 1;
      for (; r + 1 < len2; r += 2) {        // straight-line row pairs (no copies of the column state between rows)
        do_row(r, 0u);
        do_row(r + 1, 2u);
        for (int j = 0; j < CPL; j++) { sub_addr[j] = sub_addr[j] + 4u; w.keep(sub_addr[j]); }
      }
      if (r < len2) do_row(r, 0u);
    }
    // last row: maximum and its first column inside this chunk (state words order like scores; absent cells are 0)
    U m = U(0u);
    for (int j = 0; j < CPL; j++) m = w.umax(m, Sd[j]);
    cbest = w.reduce_max(m);
    U cmin = U(0xFFFFFFFFu);
    for (int j = CPL - 1; j >= 0; j--) cmin = w.sel(valid[j] & (Sd[j] == cbest), crel0 + (base + (uint32_t)j), cmin);
    ccol = w.reduce_min(cmin);
    w.lds_fence();
  }

  template <bool TRACE>
  MIA_HD static inline void chunk_any(P& w, const Pass1Args& a, int strand, int ch, int nch, uint32_t& cbest, uint32_t& ccol) {
    // the fast path needs every column of the chunk to exist: all but the last chunk of an unmasked strand
    if (a.masked || ch == nch - 1) chunk<TRACE, true>(w, a, strand, ch, cbest, ccol);
    else chunk<TRACE, false>(w, a, strand, ch, cbest, ccol);
  }

  MIA_HD static inline bool chunk_open(P& w, const Pass1Args& a, int strand, int ch) {
    if (!a.masked) return true;
    const U lane = w.lane();
    U widx = U((uint32_t)ch * (CH / 32)) + lane;
    M in = (lane < (uint32_t)(CH / 32)) & (widx * 32u < (uint32_t)a.len1);
    U word = w.lds_r32m(U(a.lds_mask[strand]) + widx * 4u, in);
    return w.ballot(in & (word != 0u)) != 0;
  }

  MIA_HD static inline uint32_t* ckpt_of(const Pass1Args& a, int strand, int ch, int nch) {
    return a.ckpt + ((size_t)strand * nch + ch) * 5u * (uint32_t)a.rows_p;
  }

  // substitution table (as in the window kernel)
  MIA_HD static inline void build_sub_table(P& w, const Pass1Args& a) {
    const U lane = w.lane();
    const int len2 = a.len2;
    const uint32_t RS2 = (uint32_t)((len2 + 1) & ~1) * 2u;
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
  }

  // Packed carries entering chunk `ch` when the previous computed chunk was `prev` (< 0: none): taken from the
  // carry-outs still in LDS, moved to the new chunk base, written back to LDS and (optionally) checkpointed.
  MIA_HD static inline void enter_chunk(P& w, const Pass1Args& a, int strand, int ch, int prev, int nch, bool checkpoint) {
    const uint32_t OFF = (uint32_t)a.pk.off;
    const uint32_t UNAV = ((uint32_t)a.pk.unavail << SH) | IDXM;
    const U lane = w.lane();
    const int len2 = a.len2;
    const uint32_t cD = a.lds_carry;
    for (int r0 = 0; r0 < len2; r0 += WAVE) {
      U r = lane + (uint32_t)r0;
      M ok = r < (uint32_t)len2;
      U d, rr, u0, u1, e;
      if (prev < 0) {
        // nothing to the left: global column 0 takes "diag" = fresh (src/mia.c:805-822), otherwise absent
        U fr = ((U(OFF - (uint32_t)GOP) - (r + 1u) * (uint32_t)GEP) << SH) | WDC;
        d = (ch == 0) ? fr : U(0u);
        rr = U(UNAV); u0 = U(UNAV); u1 = U(UNAV); e = U(UNAV);
      } else {
        const uint32_t adv = (uint32_t)(ch - prev) * CH;
        e = rebase(w, w.lds_r32m(U(cD + 4u * MAX_READ * 4u) + r * 4u, ok), adv, UNAV);
        if (ch - prev == 1) {
          d = w.lds_r32m(U(cD) + r * 4u, ok);
          rr = w.lds_r32m(U(cD + 1u * MAX_READ * 4u) + r * 4u, ok);
          u0 = rebase(w, w.lds_r32m(U(cD + 2u * MAX_READ * 4u) + r * 4u, ok), adv, UNAV);
          u1 = rebase(w, w.lds_r32m(U(cD + 3u * MAX_READ * 4u) + r * 4u, ok), adv, UNAV);
        } else {
          // masked chunks in between: their columns are absent, but the two last keys of the
          // previous open chunk are still candidates further left -> fold them into E
          U pu0 = rebase(w, w.lds_r32m(U(cD + 2u * MAX_READ * 4u) + r * 4u, ok), adv, UNAV);
          U pu1 = rebase(w, w.lds_r32m(U(cD + 3u * MAX_READ * 4u) + r * 4u, ok), adv, UNAV);
          e = w.umax3(e, pu0, pu1);
          d = U(0u); rr = U(UNAV); u0 = U(UNAV); u1 = U(UNAV);
        }
      }
      w.lds_w32(U(cD) + r * 4u, d, ok);
      w.lds_w32(U(cD + 1u * MAX_READ * 4u) + r * 4u, rr, ok);
      w.lds_w32(U(cD + 2u * MAX_READ * 4u) + r * 4u, u0, ok);
      w.lds_w32(U(cD + 3u * MAX_READ * 4u) + r * 4u, u1, ok);
      w.lds_w32(U(cD + 4u * MAX_READ * 4u) + r * 4u, e, ok);
      if (checkpoint) {
        uint32_t* ck = ckpt_of(a, strand, ch, nch);
        w.gstore_u32(ck, r, d, ok);
        w.gstore_u32(ck + a.rows_p, r, rr, ok);
        w.gstore_u32(ck + 2 * a.rows_p, r, u0, ok);
        w.gstore_u32(ck + 3 * a.rows_p, r, u1, ok);
        w.gstore_u32(ck + 4 * a.rows_p, r, e, ok);
      }
    }
    w.lds_fence();
  }

  // phase B: strand choice (src/mia.c:1549-1554); an all-masked strand scores HIM, arg-max column 0
  MIA_HD static inline void choose_strand(const int32_t* best, const bool* open, const uint32_t* scol, int len2, Pass1Result& res) {
    const int32_t HIM = INT32_MIN / 2;
    res.best[0] = open[0] ? best[0] : HIM;
    res.best[1] = open[1] ? best[1] : HIM;
    res.strand = (res.best[0] > res.best[1]) ? 0 : 1;
    const int st = res.strand;
    res.score = res.best[st];
    res.aec = open[st] ? (int32_t)scol[st] : 0;
    res.abc = res.aec;
    res.abr = len2 - 1;
    if (!open[st]) res.status |= ST_TOO_LONG;   // both strands fully masked cannot happen (the k-mer filter rejects the read first)
  }

  // phase C: traceback on the chosen strand.  `trace_chunk(ch)` leaves the 16-bit trace of chunk ch in the slab.
  template <class F>
  MIA_HD static inline void walk_back(P& w, const Pass1Args& a, Pass1Result& res, F&& trace_chunk) {
    const U lane = w.lane();
    const int len2 = a.len2;
    int r = len2 - 1, c = res.aec, cur = -1, aln_cols = 0;
    for (int guard = 0; guard < 8 * MAX_READ + 64; guard++) {
      const int ch = c / CH;
      if (ch != cur) {
        trace_chunk(ch);
        w.tr_fence();
        cur = ch;
      }
      const uint32_t base = (uint32_t)ch * CH;
      U ri = U((uint32_t)r) - lane, ci = U((uint32_t)c) - lane;
      M inside = (lane <= (uint32_t)r) & (lane <= (uint32_t)c) & (ci >= base);
      U tc = w.tr_r16(ri * (uint32_t)(CH * 2) + (ci - base) * 2u, inside);
      U ty = tc >> IB, ln = tc & IDXM;
      M colgap0 = (ty == U(TR_COLGAP)) & (ln + 1u == ci);   // source column 0 -> T == 0 -> read as diagonal (src/mia.c:619)
      M rowgap0 = (ty == U(TR_ROWGAP)) & (ln + 1u == ri);
      M plain_diag = (ty == U(TR_DIAG)) | colgap0 | rowgap0;
      M terminal = (ri == U(0u)) | (ci == U(0u)) | (ty == U(TR_START));
      M stop_here = (!inside) | terminal | (!plain_diag);
      uint64_t bal = w.ballot(stop_here);
      int f = bal ? __builtin_ctzll(bal) : WAVE;
      if (f == WAVE) { r -= WAVE; c -= WAVE; aln_cols += WAVE; continue; }
      if (!w.lane_bit(inside, f)) {          // left the chunk (or the matrix edge is handled by `terminal` first)
        if (f == 0) { res.status |= ST_TOO_LONG; break; }   // cannot happen: c / CH == ch
        r -= f; c -= f; aln_cols += f;
        continue;
      }
      aln_cols += f + 1;
      const uint32_t fty = w.lane_val(ty, f), fln = w.lane_val(ln, f);
      const int fr = r - f, fc = c - f;
      if (w.lane_bit(terminal, f)) { res.abr = fr; res.abc = fc; break; }
      if (fty == TR_COLGAP) { r = fr - 1; c = fc - 1 - (int)fln; }
      else { r = fr - 1 - (int)fln; c = fc - 1; }
      aln_cols += (int)fln;
    }
    if (aln_cols > 2 * MAX_READ) res.status |= ST_TOO_LONG;
  }

  // ---- the packed flow: k-mer masked strands, and unmasked ones whose PSSM does not allow the wide chunks
  MIA_HD static inline Pass1Result run(P& w, const Pass1Args& a) {
    const uint32_t OFF = (uint32_t)a.pk.off;
    const U lane = w.lane();
    const int len2 = a.len2, nch = (a.len1 + CH - 1) / CH;
    const uint32_t cD = a.lds_carry;
    Pass1Result res;
    res.status = ST_OK;
    build_sub_table(w, a);

    // phase A: score sweep of both strands
    uint32_t sbest[2] = {0u, 0u}, scol[2] = {0u, 0u};
    for (int strand = 0; strand < 2; strand++) {
      int prev = -1;
      for (int ch = 0; ch < nch; ch++) {
        if (!chunk_open(w, a, strand, ch)) continue;
        enter_chunk(w, a, strand, ch, prev, nch, true);
        uint32_t cb, cc;
        chunk_any<false>(w, a, strand, ch, nch, cb, cc);
        if (cb > sbest[strand]) { sbest[strand] = cb; scol[strand] = cc; }   // first maximum wins (src/mia.c:1293-1299)
        prev = ch;
      }
    }
    // note: U1 of the previous chunk enters E only with mask[k+2]; qen already encodes that.
    const int32_t best[2] = {(int32_t)((sbest[0] >> SH) - OFF), (int32_t)((sbest[1] >> SH) - OFF)};
    const bool open[2] = {sbest[0] != 0u, sbest[1] != 0u};
    choose_strand(best, open, scol, len2, res);
    if (res.status & ST_TOO_LONG) return res;
    const int st = res.strand;
    // phase C: recompute just the visited chunks from their checkpoints, now with a trace
    walk_back(w, a, res, [&](int ch) {
      const uint32_t* ck = ckpt_of(a, st, ch, nch);
      for (int r0 = 0; r0 < len2; r0 += WAVE) {
        U rr = lane + (uint32_t)r0;
        M ok = rr < (uint32_t)len2;
        for (int k = 0; k < 5; k++) w.lds_w32(U(cD + (uint32_t)k * MAX_READ * 4u) + rr * 4u, w.gload_u32(ck + k * a.rows_p, rr, ok), ok);
      }
      w.lds_fence();
      uint32_t cb, cc;
      chunk_any<true>(w, a, st, ch, nch, cb, cc);
    });
    return res;
  }

  // =====================================================================================================
  // The plain sweep (unmasked strands): phase A without index fields.
  //
  // Only VALUES matter for the best score and its column, and for those the two arg-max registers of dyn_prog
  // collapse into one key per cell:   kk(r,c) = S(r,c) + GEP*(r + c).   Within a row the GEP*r term is a common
  // constant (best_gap_col order unchanged), within a column the GEP*c term is (best_gap_row unchanged), and
  //     through a column gap:  max_{k<=c-2} kk(r-1,k) - GOP - GEP*(r-1) - GEP*(c-1)
  //     through a row gap:     max_{k<=r-2} kk(k,c-1) - GOP - GEP*(r-1) - GEP*(c-1)        (the same constant)
  //     diagonal:              kk(r-1,c-1) + GOP       - GOP - GEP*(r-1) - GEP*(c-1)
  // so   kk(r,c) = max3(kk(r-1,c-1) + GOP, CB, RB) + sub(r,c) + (2*GEP - GOP),
  // a new start is taken iff  max3 < GEP*(c-3)  and has key  GEP*(c-1) - GOP  -- neither depends on the row.
  // That is 7.5 vector operations per cell instead of 14.5.  Keys are biased by 2^30 (0 = absent / no candidate) so
  // the unsigned DPP scans apply; c is relative to the chunk, carries drop GEP*CH per chunk.
  // Checkpoints hold plain carries (D, R, U0, E; U1 == D).  Phase C turns the checkpoint of the chunk LEFT of
  // the one it needs into packed carries whose unknown arg-max indices are placeholders (values are exact, see
  // plain_to_packed), recomputes that chunk without a trace and then the wanted one with a trace, all carries of
  // the latter now exact.  A placeholder can only surface for a gap longer than a whole chunk (> horizon), which
  // never beats a new start.
  // =====================================================================================================
  static constexpr uint32_t PB = 1u << 30;
  static constexpr uint32_t CK = (uint32_t)(2 * GEP - GOP);

  template <bool EDGE>   // EDGE: some columns of the chunk lie beyond len1 (the last chunk)
  MIA_HD static inline void chunk_plain(P& w, const Pass1Args& a, int strand, int ch, uint32_t& cbest, uint32_t& ccol) {
    const U lane = w.lane();
    const int len2 = a.len2;
    const uint32_t base = (uint32_t)ch * CH;
    const uint32_t RS2 = (uint32_t)((len2 + 1) & ~1) * 2u;
    const uint32_t cD = a.lds_carry, cR = cD + MAX_READ * 4, cU0 = cR + MAX_READ * 4, cE = cU0 + 2 * MAX_READ * 4;
    const M last_lane = lane == 63u;
    const M all = lane < 64u;
    const U zero = U(0u);

    U sub_addr[CPL], STc[CPL], STK[CPL];
    M valid[CPL];
    const U crel0 = lane * (uint32_t)CPL;
    for (int j = 0; j < CPL; j++) {
      U gcol = crel0 + (base + (uint32_t)j);
      M in = gcol < (uint32_t)a.len1;
      valid[j] = EDGE ? in : all;
      U code = w.sel(in, w.gload_u8(a.ref_codes[strand], gcol, in), U(4u));
      sub_addr[j] = w.lds_abs(U(a.lds_sub) + code * RS2);
      STc[j] = (crel0 + (uint32_t)j) * (uint32_t)GEP + (PB - 3u * (uint32_t)GEP);    // start iff max3 < this
      STK[j] = STc[j] + CK;                                                          // key of a new start
    }
    U kk[CPL], rrun[CPL];
    for (int j = 0; j < CPL; j++) {     // row 0 (src/mia.c:769-785)
      U k0 = w.lds_ri16a(sub_addr[j], 0u) + (STc[j] + 3u * (uint32_t)GEP);
      kk[j] = EDGE ? w.sel(valid[j], k0, zero) : k0;
      rrun[j] = zero;
    }
    auto do_row = [&](const int r, const uint32_t off) __attribute__((always_inline)) {
      const U inD = w.lds_r32(U(cD + (uint32_t)r * 4u)), inR = w.lds_r32(U(cR + (uint32_t)r * 4u));
      const U inU0 = w.lds_r32(U(cU0 + (uint32_t)r * 4u)), inE = w.lds_r32(U(cE + (uint32_t)r * 4u));
      w.lds_w32(U(cD + (uint32_t)r * 4u), kk[CPL - 1], last_lane);
      w.lds_w32(U(cR + (uint32_t)r * 4u), rrun[CPL - 1], last_lane);
      w.lds_w32(U(cU0 + (uint32_t)r * 4u), kk[CPL - 2], last_lane);
      U dleft = w.shr1(kk[CPL - 1], inD);
      U rleft = w.shr1(rrun[CPL - 1], inR);
      U u0 = w.shr1(kk[CPL - 2], inU0);
      U tot = w.umax(u0, dleft);
      {
        int j = 0;
        for (; j + 1 < CPL - 2; j += 2) tot = w.umax3(tot, kk[j], kk[j + 1]);
        if (j < CPL - 2) tot = w.umax(tot, kk[j]);
      }
      U incl = w.scan_max(tot);
      U excl = w.umax(w.shr1_max(incl, zero), inE);
      w.lds_w32(U(cE + (uint32_t)r * 4u), w.umax(incl, inE), last_lane);
      U subv[CPL];
      for (int j = 0; j < CPL; j++) subv[j] = w.lds_ri16a(sub_addr[j], off);
      U cand[CPL];
      cand[0] = w.umax(excl, u0);
      cand[1] = w.umax(cand[0], dleft);
      for (int j = 2; j < CPL; j++) cand[j] = w.umax(cand[j - 1], kk[j - 2]);
      for (int j = CPL - 1; j >= 0; j--) {
        U kd = (j == 0) ? dleft : kk[j - 1];
        U rl = (j == 0) ? rleft : rrun[j - 1];
        U m3 = w.umax3(kd + (uint32_t)GOP, cand[j], rl);
        U kn = w.sel(m3 < STc[j], STK[j], w.add3(m3, subv[j], CK));
        if (EDGE) kn = w.sel(valid[j], kn, zero);
        rrun[j] = w.umax(rrun[j], kk[j]);
        kk[j] = kn;
      }
    };
    for (int j = 0; j < CPL; j++) sub_addr[j] = sub_addr[j] + 2u;
    {
      int r = 1;
      for (; r + 1 < len2; r += 2) {        // straight-line row pairs (no copies of the column state between rows)
        do_row(r, 0u);
        do_row(r + 1, 2u);
        for (int j = 0; j < CPL; j++) { sub_addr[j] = sub_addr[j] + 4u; w.keep(sub_addr[j]); }
      }
      if (r < len2) do_row(r, 0u);
    }
    // last row: back to scores (still biased by PB): S = kk - GEP*(len2-1) - GEP*c
    U m = zero, sc[CPL];
    for (int j = 0; j < CPL; j++) {
      sc[j] = w.sel(valid[j], kk[j] - (STc[j] - (PB - 3u * (uint32_t)GEP)) - (uint32_t)((len2 - 1) * GEP), zero);
      m = w.umax(m, sc[j]);
    }
    cbest = w.reduce_max(m);
    U cmin = U(0xFFFFFFFFu);
    for (int j = CPL - 1; j >= 0; j--) cmin = w.sel(valid[j] & (sc[j] == cbest), crel0 + (base + (uint32_t)j), cmin);
    ccol = w.reduce_min(cmin);
    w.lds_fence();
  }

  // plain carries entering chunk ch (its left neighbour was just computed, or nothing if ch == 0); checkpointed
  MIA_HD static inline void enter_chunk_plain(P& w, const Pass1Args& a, int strand, int ch, int nch) {
    const U lane = w.lane();
    const int len2 = a.len2;
    const uint32_t cD = a.lds_carry;
    const uint32_t step = (uint32_t)GEP * (uint32_t)CH;
    for (int r0 = 0; r0 < len2; r0 += WAVE) {
      U r = lane + (uint32_t)r0;
      M ok = r < (uint32_t)len2;
      U v[4];
      if (ch == 0) {
        // global column 0 takes "diag" = fresh(r) (src/mia.c:805-822): kk = fresh(r) + GEP*(r-1 + -1) = -GOP - 3*GEP
        v[0] = U(PB - (uint32_t)GOP - 3u * (uint32_t)GEP);
        v[1] = U(0u); v[2] = U(0u); v[3] = U(0u);
      } else {
        for (int k = 0; k < 4; k++) {
          U x = w.lds_r32m(U(cD + (uint32_t)(k == 3 ? 4 : k) * MAX_READ * 4u) + r * 4u, ok);
          v[k] = x - w.umin(x, U(step));       // one chunk further right: c drops by CH; 0 (absent) stays 0
        }
      }
      uint32_t* ck = ckpt_of(a, strand, ch, nch);
      for (int k = 0; k < 4; k++) {
        w.lds_w32(U(cD + (uint32_t)(k == 3 ? 4 : k) * MAX_READ * 4u) + r * 4u, v[k], ok);
        w.gstore_u32(ck + k * a.rows_p, r, v[k], ok);
      }
    }
    w.lds_fence();
  }

  // plain checkpoint of chunk ch -> packed carries in LDS.  Values are exact; the arg-max indices that a plain
  // key does not carry (E: some column <= -3, R: some row <= r-2) are placeholders, see above.
  MIA_HD static inline void plain_to_packed(P& w, const Pass1Args& a, int strand, int ch, int nch) {
    const uint32_t OFF = (uint32_t)a.pk.off;
    const uint32_t UNAVV = (uint32_t)a.pk.unavail, UNAV = (UNAVV << SH) | IDXM;
    const U lane = w.lane();
    const int len2 = a.len2;
    const uint32_t cD = a.lds_carry;
    const uint32_t* ck = ckpt_of(a, strand, ch, nch);
    const uint32_t SGN = 0x80000000u;
    for (int r0 = 0; r0 < len2; r0 += WAVE) {
      U r = lane + (uint32_t)r0;
      M ok = r < (uint32_t)len2;
      U pd = w.gload_u32(ck, r, ok), pr = w.gload_u32(ck + a.rows_p, r, ok);
      U pu0 = w.gload_u32(ck + 2 * a.rows_p, r, ok), pe = w.gload_u32(ck + 3 * a.rows_p, r, ok);
      const uint32_t HALF = PB / 2u;
      // D = kk(r-1, -1): S = D - PB - GEP*(r-2)
      U sD = pd - PB - (r - 2u) * (uint32_t)GEP + OFF;                       // biased score
      U d = w.sel(pd < HALF, U(0u), (sD << SH) | WDC);
      U u1v = sD + (uint32_t)(GEP * (REL - 1));
      U u1 = w.sel(pd < HALF, U(UNAV), (u1v << SH) + (IDXM - (uint32_t)(REL - 1)));
      // U0 = kk(r-1, -2): S = U0 - PB - GEP*(r-3)
      U u0v = pu0 - PB - (r - 3u) * (uint32_t)GEP + OFF + (uint32_t)(GEP * (REL - 2));
      U u0 = w.sel(pu0 < HALF, U(UNAV), (u0v << SH) + (IDXM - (uint32_t)(REL - 2)));
      // E = max_{c <= -3} kk(r-1, c): S + GEP*c = E - PB - GEP*(r-1); packed value adds GEP*REL; placeholder column
      U ev = pe - PB - (r - 1u) * (uint32_t)GEP + OFF + (uint32_t)(GEP * REL);
      U e = w.sel((pe < HALF) | ((ev + SGN) <= (UNAVV + SGN)), U(UNAV), (ev << SH) | IDXM);
      // R = max_{k <= r-2} kk(k, -1): S + GEP*k = R - PB + GEP; placeholder row
      U rv = pr - PB + (uint32_t)GEP + OFF;
      U rr = w.sel((pr < HALF) | ((rv + SGN) <= (UNAVV + SGN)), U(UNAV), (rv << SH) | IDXM);
      w.lds_w32(U(cD) + r * 4u, d, ok);
      w.lds_w32(U(cD + 1u * MAX_READ * 4u) + r * 4u, rr, ok);
      w.lds_w32(U(cD + 2u * MAX_READ * 4u) + r * 4u, u0, ok);
      w.lds_w32(U(cD + 3u * MAX_READ * 4u) + r * 4u, u1, ok);
      w.lds_w32(U(cD + 4u * MAX_READ * 4u) + r * 4u, e, ok);
    }
    w.lds_fence();
  }

  MIA_HD static inline Pass1Result run_plain(P& w, const Pass1Args& a) {
    const int len2 = a.len2, nch = (a.len1 + CH - 1) / CH;
    Pass1Result res;
    res.status = ST_OK;
    build_sub_table(w, a);
    uint32_t sbest[2] = {0u, 0u}, scol[2] = {0u, 0u};
    for (int strand = 0; strand < 2; strand++) {
      for (int ch = 0; ch < nch; ch++) {
        enter_chunk_plain(w, a, strand, ch, nch);
        uint32_t cb, cc;
        if (ch == nch - 1) chunk_plain<true>(w, a, strand, ch, cb, cc);
        else chunk_plain<false>(w, a, strand, ch, cb, cc);
        if (cb > sbest[strand]) { sbest[strand] = cb; scol[strand] = cc; }
      }
    }
    const int32_t best[2] = {(int32_t)(sbest[0] - PB), (int32_t)(sbest[1] - PB)};
    const bool open[2] = {true, true};
    choose_strand(best, open, scol, len2, res);
    const int st = res.strand;
    // Most alignments are gap-free: the proof of align_body_quad_plain.h -- (i) best == D(R), (ii) D(r-1) >= -P(r+1) on the
    // diagonal through the end cell -- then gives the begin point without recomputing any chunk with a trace.
    {
      const U lane = w.lane();
      const int R = len2 - 1, d = res.aec - R, r0 = d < 0 ? -d : 0;
      const uint32_t RS2 = (uint32_t)((len2 + 1) & ~1) * 2u;
      int32_t carry = 0;
      bool ok = true;
      for (int rb = r0; rb <= R && ok; rb += WAVE) {
        U r = lane + (uint32_t)rb;
        M in = r <= (uint32_t)R;
        U c = r + (uint32_t)d;
        U code = w.gload_u8(a.ref_codes[st], c, in);
        U sv = w.sel(in, w.lds_ri16o(U(a.lds_sub) + code * RS2 + r * 2u, 0u), U(0u));
        if (rb == r0 && d < 0) sv = sv + w.sel(lane == 0u, U((uint32_t)(-(GOP + GEP * (r0 + 1)))), U(0u));   // column 0: sub - P(r+1)
        U D = w.scan_add(sv) + (uint32_t)carry;
        M bad = in & (r < (uint32_t)R) & ((D + 0x80000000u) < (U((uint32_t)(-(GOP + 2 * GEP))) - r * (uint32_t)GEP + 0x80000000u));
        if (w.ballot(bad)) ok = false;
        const int last = (R - rb) < (WAVE - 1) ? (R - rb) : (WAVE - 1);
        carry = (int32_t)w.lane_val(D, last);
      }
      if (ok && carry == res.score) {
        res.abr = r0;
        res.abc = d < 0 ? 0 : d;
        return res;
      }
    }
    walk_back(w, a, res, [&](int ch) {
      uint32_t cb, cc;
      if (ch == 0) {
        enter_chunk(w, a, st, 0, -1, nch, false);
      } else {
        if (ch == 1) enter_chunk(w, a, st, 0, -1, nch, false);
        else plain_to_packed(w, a, st, ch - 1, nch);
        chunk_any<false>(w, a, st, ch - 1, nch, cb, cc);
        enter_chunk(w, a, st, ch, ch - 1, nch, false);
      }
      chunk_any<true>(w, a, st, ch, nch, cb, cc);
    });
    return res;
  }
};

}  // namespace mia
