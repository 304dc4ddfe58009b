// emu_align.cpp -- TEST INFRASTRUCTURE: runs csrc/align_body.h (the exact source
// the HIP kernel instantiates) on the CPU lock-step wave emulation, so the
// packed-word DP and the ballot traceback can be checked against the oracle
// without a GPU.  Built by tests/test_emul_align.py into oracle/_build/.
#include <stdint.h>
#include <string.h>
#include <vector>

#include "wave_emu.h"
#include "align_body.h"
#include "pass1_body.h"
#include "align_body_quad.h"
#include "align_body_quad_plain.h"
#include "diag_filter.h"
#include "band_body.h"

using namespace mia;

template <int CPL>
static int run_one(const uint8_t* ref_codes, int ref_start, int len1, const uint8_t* read_codes, int len2,
                   const int32_t* pssm, int sg5, int max_abs, int32_t* out5, int16_t* cols) {
  PackParams pk;
  if (!make_pack_params(64 * CPL, max_abs, &pk)) return -1;
  std::vector<uint8_t> packed((len2 + 1) / 2 + 4, 0);
  for (int i = 0; i < len2; i++) packed[i >> 1] |= (uint8_t)(read_codes[i] << ((i & 1) * 4));
  AlignArgs a;
  a.ref_codes = ref_codes; a.ref_start = ref_start; a.len1 = len1;
  a.read_packed = packed.data(); a.len2 = len2; a.pssm = pssm; a.sg5 = sg5; a.pk = pk;
  a.lds_sub = 0;
  a.trace_stride = (uint32_t)((len1 + 3) & ~3);
  a.cols_out = cols;
  a.dbg = 0;
  EmuWave w((size_t)len2 * 10 + 16, (size_t)len2 * a.trace_stride + 16);
  AlignResult r = WindowAligner<EmuWave, CPL>::run(w, a);
  out5[0] = r.score; out5[1] = r.abc; out5[2] = r.abr; out5[3] = r.aec; out5[4] = (int32_t)r.status;
  return 0;
}

// trim_frag (src/mia.c:1318-1368): the read is the "reference" (columns), the adapter the "read" (rows), last-column end
extern "C" int emu_trim(const uint8_t* read_codes, int len1, const uint8_t* adapter_codes, int len2, const int32_t* flat_pssm,
                        int max_abs, int32_t* out6) {
  PackParams pk;
  if (!make_pack_params(256, max_abs, &pk)) return -1;
  std::vector<uint8_t> packed((len2 + 1) / 2 + 4, 0);
  for (int i = 0; i < len2; i++) packed[i >> 1] |= (uint8_t)(adapter_codes[i] << ((i & 1) * 4));
  std::vector<int16_t> cols(MAX_READ + 8);
  AlignArgs a;
  a.ref_codes = read_codes; a.ref_start = 0; a.len1 = len1;
  a.read_packed = packed.data(); a.len2 = len2; a.pssm = flat_pssm; a.sg5 = 1; a.pk = pk;
  a.lds_sub = 0;
  a.trace_stride = (uint32_t)((len1 + 3) & ~3);
  a.cols_out = cols.data();
  a.dbg = 0;
  EmuWave w((size_t)len2 * 10 + 16, (size_t)len2 * a.trace_stride + 16);
  AlignResult r = WindowAligner<EmuWave, 4, true>::run(w, a);
  out6[0] = r.score; out6[1] = r.aec; out6[2] = r.aer; out6[3] = r.abc; out6[4] = r.abr; out6[5] = (int32_t)r.status;
  return 0;
}

extern "C" int emu_align_window(int cpl, const uint8_t* ref_codes, int ref_start, int len1, const uint8_t* read_codes,
                                int len2, const int32_t* pssm, int sg5, int max_abs, int32_t* out5, int16_t* cols) {
  switch (cpl) {
    case 4: return run_one<4>(ref_codes, ref_start, len1, read_codes, len2, pssm, sg5, max_abs, out5, cols);
    case 8: return run_one<8>(ref_codes, ref_start, len1, read_codes, len2, pssm, sg5, max_abs, out5, cols);
    case 12: return run_one<12>(ref_codes, ref_start, len1, read_codes, len2, pssm, sg5, max_abs, out5, cols);
    default: return -2;
  }
}

// pass 1 of one read: both strands vs the whole (wrapped) reference, optional column masks
template <int CPL>
static int emu_pass1_t(const uint8_t* fw_codes, const uint8_t* rc_codes, int len1, const uint8_t* read_codes, int len2,
                       const int32_t* pssm, int max_abs, const uint8_t* fw_mask, const uint8_t* rc_mask, int32_t* out8, int plain) {
  constexpr int P1_CH = p1_ch(CPL);
  PackParams pk;
  if (!make_pack_params(1024, max_abs, &pk)) return -1;
  std::vector<uint8_t> packed((len2 + 1) / 2 + 4, 0);
  for (int i = 0; i < len2; i++) packed[i >> 1] |= (uint8_t)(read_codes[i] << ((i & 1) * 4));
  const int nch = (len1 + P1_CH - 1) / P1_CH, words = nch * (P1_CH / 32) + 4;
  Pass1Args a;
  a.ref_codes[0] = fw_codes; a.ref_codes[1] = rc_codes; a.len1 = len1;
  a.read_packed = packed.data(); a.len2 = len2; a.pssm = pssm; a.pk = pk;
  a.lds_sub = 0;
  a.lds_carry = MAX_READ * 10;
  a.lds_mask[0] = a.lds_carry + 5 * MAX_READ * 4;
  a.lds_mask[1] = a.lds_mask[0] + (uint32_t)words * 4;
  a.masked = (fw_mask && rc_mask) ? 1 : 0;
  a.rows_p = (len2 + 3) & ~3;
  std::vector<uint32_t> ckpt((size_t)2 * nch * 5 * a.rows_p + 16, 0xDEADBEEFu);
  a.ckpt = ckpt.data();
  EmuWave w(a.lds_mask[1] + (size_t)words * 4 + 64, (size_t)MAX_READ * P1_CH * 2 + 64);
  if (a.masked) {
    for (int s = 0; s < 2; s++) {
      const uint8_t* m = s ? rc_mask : fw_mask;
      for (int c = 0; c < len1; c++)
        if (m[c]) { uint32_t x; memcpy(&x, &w.lds[a.lds_mask[s] + (c >> 5) * 4], 4); x |= 1u << (c & 31); memcpy(&w.lds[a.lds_mask[s] + (c >> 5) * 4], &x, 4); }
    }
  }
  a.plain = (plain && !a.masked) ? 1 : 0;
  Pass1Result r = a.plain ? Pass1Aligner<EmuWave, CPL>::run_plain(w, a) : Pass1Aligner<EmuWave, CPL>::run(w, a);
  out8[0] = r.best[0]; out8[1] = r.best[1]; out8[2] = r.strand; out8[3] = r.score; out8[4] = r.aec; out8[5] = r.abc;
  out8[6] = r.abr; out8[7] = (int32_t)r.status;
  return 0;
}

// cpl: 4 or 12; cpl = 112 selects the plain-key sweep on 768-column chunks (unmasked only)
extern "C" int emu_pass1(int cpl, const uint8_t* fw_codes, const uint8_t* rc_codes, int len1, const uint8_t* read_codes, int len2,
                         const int32_t* pssm, int max_abs, const uint8_t* fw_mask, const uint8_t* rc_mask, int32_t* out8) {
  if (cpl == P1_CPL_NARROW) return emu_pass1_t<P1_CPL_NARROW>(fw_codes, rc_codes, len1, read_codes, len2, pssm, max_abs, fw_mask, rc_mask, out8, 0);
  if (cpl == P1_CPL_WIDE || cpl == 100 + P1_CPL_WIDE)
    return emu_pass1_t<P1_CPL_WIDE>(fw_codes, rc_codes, len1, read_codes, len2, pssm, max_abs, fw_mask, rc_mask, out8, cpl >= 100);
  return -2;
}

// quad kernel body: up to four reads of equal length, one per 16-lane row
extern "C" int emu_align_quad(int ng, const uint8_t* ref_codes, const int32_t* ref_start, const int32_t* len1,
                              const uint8_t* read_codes /* ng x len2 */, int len2, const int32_t* pssm2, const int32_t* rc,
                              int max_abs, int32_t* out5 /* ng x 5 */, int16_t* cols /* ng x 256 */, int band, const int32_t* dexp) {
  PackParams pk;
  if (!make_pack_params(256, max_abs, &pk)) return -1;
  const uint32_t stride = (uint32_t)(((len2 + 1) / 2 + 3) & ~3);
  std::vector<uint8_t> packed((size_t)stride * 4 + 8, 0);
  QuadArgs a;
  a.ref_codes = ref_codes; a.packed = packed.data(); a.pssm2 = pssm2; a.len2 = len2; a.pk = pk;
  a.lds_sub = 0; a.slab_group = MAX_READ * Q_TRACE_STRIDE; a.dbg = 0; a.band = band;
  for (int g = 0; g < Q_G; g++) {
    a.ref_start[g] = 0; a.len1[g] = 0; a.roff[g] = 0; a.rc[g] = 0; a.cols_out[g] = cols + g * 256; a.dexp[g] = (band && g < ng) ? dexp[g] : 0;
    if (g < ng) {
      a.ref_start[g] = ref_start[g]; a.len1[g] = len1[g]; a.roff[g] = (uint32_t)g * stride; a.rc[g] = (uint32_t)rc[g];
      for (int i = 0; i < len2; i++) packed[a.roff[g] + (i >> 1)] |= (uint8_t)(read_codes[g * len2 + i] << ((i & 1) * 4));
    }
  }
  EmuWave w(Q_G * q_sub_bytes(len2) + 64, (size_t)Q_G * a.slab_group + 64);
  AlignResult res[Q_G];
  QuadAligner<EmuWave>::run(w, a, res);
  for (int g = 0; g < ng; g++) {
    out5[g * 5 + 0] = res[g].score; out5[g * 5 + 1] = res[g].abc; out5[g * 5 + 2] = res[g].abr; out5[g * 5 + 3] = res[g].aec;
    out5[g * 5 + 4] = (int32_t)res[g].status;
  }
  return 0;
}

// values-only quad body with the diagonal proof: out6 per read = score, abc, abr, aec, proven, 0
extern "C" int emu_align_quad_plain(int ng, const uint8_t* ref_codes, const int32_t* ref_start, const int32_t* len1,
                                    const uint8_t* read_codes /* ng x len2 */, int len2, const int32_t* pssm2, const int32_t* rc,
                                    int32_t* out6 /* ng x 6 */, int16_t* cols /* ng x 256 */) {
  const uint32_t stride = (uint32_t)(((len2 + 1) / 2 + 3) & ~3);
  std::vector<uint8_t> packed((size_t)stride * 4 + 8, 0);
  QuadPlainArgs a;
  a.ref_codes = ref_codes; a.packed = packed.data(); a.pssm2 = pssm2; a.len2 = len2; a.lds_sub = 0;
  for (int g = 0; g < Q_G; g++) {
    a.ref_start[g] = 0; a.len1[g] = 0; a.roff[g] = 0; a.rc[g] = 0; a.cols_out[g] = cols + g * 256;
    if (g < ng) {
      a.ref_start[g] = ref_start[g]; a.len1[g] = len1[g]; a.roff[g] = (uint32_t)g * stride; a.rc[g] = (uint32_t)rc[g];
      for (int i = 0; i < len2; i++) packed[a.roff[g] + (i >> 1)] |= (uint8_t)(read_codes[g * len2 + i] << ((i & 1) * 4));
    }
  }
  EmuWave w(Q_G * q_sub_bytes(len2) + 64, 64);
  QuadPlainResult res[Q_G];
  QuadPlainAligner<EmuWave>::run(w, a, res);
  for (int g = 0; g < ng; g++) {
    out6[g * 6 + 0] = res[g].score; out6[g * 6 + 1] = res[g].abc; out6[g * 6 + 2] = res[g].abr; out6[g * 6 + 3] = res[g].aec;
    out6[g * 6 + 4] = res[g].proven; out6[g * 6 + 5] = 0;
  }
  return 0;
}

// 10-mer table of a code string in static buffers (sparse reset: a table is 20 MB, the tests call this thousands of times)
struct EmuTable {
  std::vector<int32_t> cnt, pos;
  std::vector<int64_t> touched;
  EmuTable() : cnt((size_t)mia::DF_KTAB, 0), pos((size_t)mia::DF_KTAB * mia::DF_KCAP, 0) {}
  mia::KmerOcc build(const uint8_t* codes, int64_t n) {
    for (int64_t idx : touched) cnt[(size_t)idx] = 0;
    touched.clear();
    for (int64_t p = 0; p < n; p++) {
      const int64_t idx = mia::kmer_at(codes, n, p);
      if (idx < 0) continue;
      const int c = cnt[(size_t)idx]++;
      if (c < mia::DF_KCAP) pos[(size_t)(idx * mia::DF_KCAP + c)] = (int32_t)p;
      touched.push_back(idx);
    }
    return mia::KmerOcc{cnt.data(), pos.data()};
  }
};
static EmuTable g_tab_a, g_tab_b;

// The diagonal filter (csrc/diag_filter.h) as the kernel runs it: planes of the whole reference, the read as 4-bit codes
// on a 4-byte boundary.  Returns 1 if the read's alignment in [ref_start, ref_start+len1) is proven to be the gap-free
// diagonal *delta with *mismatches mismatches, 0 if the read is left to the DP.
extern "C" int emu_diag_filter(const uint8_t* ref_codes, int64_t n_codes, int ref_start, int len1, const uint8_t* read_codes, int len2,
                               int* delta, int* mismatches) {
  using namespace mia;
  const int64_t words = plane_words(n_codes);
  std::vector<uint64_t> lo((size_t)words), hi((size_t)words), ok((size_t)words);
  for (int64_t w = 0; w < words; w++) plane_word(ref_codes, n_codes, w, &lo[(size_t)w], &hi[(size_t)w], &ok[(size_t)w]);
  std::vector<uint32_t> packed((size_t)(len2 / 8 + 2), 0);
  uint8_t* pb = (uint8_t*)packed.data();
  for (int r = 0; r < len2; r++) pb[r >> 1] |= (uint8_t)((read_codes[r] & 15) << ((r & 1) * 4));
  RefPlanes rp{lo.data(), hi.data(), ok.data()};
  DiagVerdict v{0, 0};
  const KmerOcc ko = g_tab_a.build(ref_codes, n_codes);
  const bool won = diag_filter(rp, ko, n_codes, ref_start, len1, pb, len2, &v);
  *delta = v.delta; *mismatches = v.mismatches;
  // where the table applies, step 1 through it must see what the slide over all diagonals sees
  int d1 = 0, b1 = 0, d2 = 0, b2 = 0;
  const int k_table = diag_step1(rp, ko, n_codes, ref_start, len1, pb, len2, &d1, &b1);
  const int k_slide = diag_step1(rp, ref_start, len1, pb, len2, &d2, &b2);
  if (k_table != k_slide || (k_table >= 0 && d1 != d2)) return -7;
  return won ? 1 : 0;
}

// The pass-1 form of the filter (both strands of the whole reference): returns K (0..2) with *strand / *delta when the
// read is decided, -1 when it is left to the DP.
extern "C" int emu_pass1_filter(const uint8_t* fw_codes, const uint8_t* rc_codes, int len1, const uint8_t* read_codes, int len2, int* strand,
                                int* delta) {
  using namespace mia;
  const int64_t words = plane_words(len1);
  std::vector<uint64_t> pl((size_t)words * 6);
  for (int64_t w = 0; w < words; w++) {
    plane_word(fw_codes, len1, w, &pl[(size_t)w], &pl[(size_t)(words + w)], &pl[(size_t)(2 * words + w)]);
    plane_word(rc_codes, len1, w, &pl[(size_t)(3 * words + w)], &pl[(size_t)(4 * words + w)], &pl[(size_t)(5 * words + w)]);
  }
  std::vector<uint32_t> packed((size_t)(len2 / 8 + 2), 0);
  uint8_t* pb = (uint8_t*)packed.data();
  for (int r = 0; r < len2; r++) pb[r >> 1] |= (uint8_t)((read_codes[r] & 15) << ((r & 1) * 4));
  RefPlanes fw{pl.data(), pl.data() + words, pl.data() + 2 * words}, rc{pl.data() + 3 * words, pl.data() + 4 * words, pl.data() + 5 * words};
  const KmerOcc kf0 = g_tab_a.build(fw_codes, len1), kr0 = g_tab_b.build(rc_codes, len1);
  const int k = pass1_step1(fw, rc, kf0, kr0, len1, pb, len2, strand, delta);
  if (k == 2 && !pass1_step2(fw, rc, kf0, kr0, len1, pb, len2)) return -1;
  return k;
}

// Rule (c) of the filter both ways for one read: bit 0 = verdict of the slide over all diagonals (diag_step2), bit 1 =
// verdict of the 10-mer table route (diag_step2_kmer).  The table route counts every diagonal it does not look at as 9,
// so it may only say yes where the slide says yes.  -1: step 1 does not make the read a K = 2 candidate.
extern "C" int emu_step2_both(const uint8_t* ref_codes, int64_t n_codes, int ref_start, int len1, const uint8_t* read_codes, int len2) {
  using namespace mia;
  const int64_t words = plane_words(n_codes);
  std::vector<uint64_t> lo((size_t)words), hi((size_t)words), ok((size_t)words);
  for (int64_t w = 0; w < words; w++) plane_word(ref_codes, n_codes, w, &lo[(size_t)w], &hi[(size_t)w], &ok[(size_t)w]);
  std::vector<uint32_t> packed((size_t)(len2 / 8 + 2), 0);
  uint8_t* pb = (uint8_t*)packed.data();
  for (int r = 0; r < len2; r++) pb[r >> 1] |= (uint8_t)((read_codes[r] & 15) << ((r & 1) * 4));
  RefPlanes rp{lo.data(), hi.data(), ok.data()};
  int delta = 0, best = 0;
  if (diag_step1(rp, ref_start, len1, pb, len2, &delta, &best) != 2) return -1;
  static std::vector<int32_t> cnt((size_t)DF_KTAB, 0), pos((size_t)DF_KTAB * DF_KCAP, 0);
  std::vector<int64_t> touched;
  for (int64_t p = 0; p < n_codes; p++) {
    const int64_t idx = kmer_at(ref_codes, n_codes, p);
    if (idx < 0) continue;
    const int c = cnt[(size_t)idx]++;
    if (c < DF_KCAP) pos[(size_t)(idx * DF_KCAP + c)] = (int32_t)p;
    touched.push_back(idx);
  }
  KmerOcc ko{cnt.data(), pos.data()};
  const int scan = diag_step2(rp, ref_start, len1, pb, len2) ? 1 : 0;
  const int km = diag_step2_kmer(rp, ko, n_codes, ref_start, len1, pb, len2) > 0 ? 2 : 0;
  for (int64_t idx : touched) cnt[(size_t)idx] = 0;
  return scan | km;
}

// The banded DP (csrc/band_body.h) for one read: returns 1 with out5 = {score, abc, aec, abr, gaps} and the script in cols
// (window columns), 0 if the read is left to the full-window kernels.  out_plan = {d0, w, b0, budget}.  widen: extra
// diagonals on the band's right-hand side.
extern "C" int emu_band(const uint8_t* ref_codes, int64_t n_codes, int ref_start, int len1, const uint8_t* read_codes, int len2, int32_t* out5,
                        int16_t* cols, int32_t* out_plan, int widen) {
  using namespace mia;
  const int64_t words = plane_words(n_codes);
  std::vector<uint64_t> lo((size_t)words), hi((size_t)words), ok((size_t)words);
  for (int64_t w = 0; w < words; w++) plane_word(ref_codes, n_codes, w, &lo[(size_t)w], &hi[(size_t)w], &ok[(size_t)w]);
  std::vector<uint32_t> packed((size_t)(len2 / 8 + 2), 0);
  uint8_t* pb = (uint8_t*)packed.data();
  for (int r = 0; r < len2; r++) pb[r >> 1] |= (uint8_t)((read_codes[r] & 15) << ((r & 1) * 4));
  RefPlanes rp{lo.data(), hi.data(), ok.data()};
  const KmerOcc ko = g_tab_a.build(ref_codes, n_codes);
  BandPlan bp;
  if (!band_plan(rp, ko, n_codes, ref_start, len1, pb, len2, &bp)) return 0;
  out_plan[0] = bp.d0; out_plan[1] = bp.w; out_plan[2] = bp.b0; out_plan[3] = bp.budget;
  std::vector<uint32_t> trace((size_t)len2 * (BAND_W / 4));
  BandResult res;
  // the kernel runs every read of a wavefront with the widest band among them: any width from the plan's up must do
  const int wmax = ((bp.w + widen > BAND_W ? BAND_W : bp.w + widen) + 3) & ~3;
  const bool got = (band_interior(bp, wmax, len1, len2) && !(widen & 1))   // (odd widen: the edge form on interior reads too)
                      ? band_align<false>(rp, ref_start, len1, pb, len2, bp, wmax, trace.data(), BAND_W / 4, cols, &res)
                      : band_align<true>(rp, ref_start, len1, pb, len2, bp, wmax, trace.data(), BAND_W / 4, cols, &res);
  if (!got) return 0;
  out5[0] = res.score; out5[1] = res.abc; out5[2] = res.aec; out5[3] = res.abr; out5[4] = res.gaps;
  return 1;
}

// The matrix-agnostic band pipeline (csrc/bandx_body.h) as k_bx_plan / k_bx_values / k_bx_trace run it.
//   opts bit 0: run the trace DP even where the values-only check (or the plan alone) would finish the read
//   opts bit 1: use the EDGE form of the recurrence even for interior bands
//   opts >> 4 : widen the band class (0..3 classes up)
// Returns the plan's mode (0 = not planned), negated if the stages disagree among themselves; out6 = score, abc, aec,
// abr, gaps, how the read was finished (1 plan, 2 values, 3 trace, 0 not finished); plan5 = d0, w, dstar, b0, edge.
#define BX_DIAG 1
#include "bandx_body.h"
template <int W>
static int emu_bandx_w(const uint32_t* nib, int s, int len1, const uint32_t* rw, int len2, const mia::BxPlan& bp, bool edge, const int32_t* sub,
                       const int32_t* sub256, bool want_values, bool want_trace, int expect, int32_t* out6, int16_t* cols) {
  using namespace mia;
  int how = 0;
  if (want_values) {
    int best, bj;
    if (edge) bx_values<W, true>(nib, s, len1, rw, len2, bp.d0, sub, &best, &bj);
    else bx_values<W, false>(nib, s, len1, rw, len2, bp.d0, sub, &best, &bj);
    if (bj >= 0 && best == expect && bj == bp.dstar - bp.d0) {
      how = 2;
      out6[0] = best; out6[1] = bp.dstar; out6[2] = bp.dstar + len2 - 1; out6[3] = 0; out6[4] = 0;
      for (int r = 0; r < len2; r++) cols[r] = (int16_t)(bp.dstar + r);
    }
  }
  if (want_trace || (want_values && !how)) {
    std::vector<uint32_t> trace((size_t)len2 * (W / 4));
    std::vector<int16_t> c2((size_t)len2 + 8, -9);
    BxResult res;
    const bool got = edge ? bx_trace<W, true>(nib, s, len1, rw, len2, bp.d0, sub256, trace.data(), W / 4, c2.data(), &res)
                          : bx_trace<W, false>(nib, s, len1, rw, len2, bp.d0, sub256, trace.data(), W / 4, c2.data(), &res);
    if (got) {
      if (how) {      // both stages finished the read: they must agree
        if (res.score != out6[0] || res.abc != out6[1] || res.aec != out6[2] || res.abr != out6[3] || res.gaps != 0) return -100;
        for (int r = 0; r < len2; r++) if (c2[r] != cols[r]) return -101;
      } else {
        how = 3;
        out6[0] = res.score; out6[1] = res.abc; out6[2] = res.aec; out6[3] = res.abr; out6[4] = res.gaps;
        for (int r = 0; r < len2; r++) cols[r] = c2[r];
      }
    } else if (how) return -102;     // the values stage proved a pure diagonal the traceback could not walk
  }
  return how;
}

extern "C" int emu_bandx(const uint8_t* ref_codes, int64_t n_codes, int ref_start, int len1, const uint8_t* read_codes, int len2, const int32_t* fwd,
                         const int32_t* rc, int strand, int opts, int32_t* out6, int16_t* cols, int32_t* plan5) {
  using namespace mia;
  const int64_t words = plane_words(n_codes);
  std::vector<uint64_t> lo((size_t)words), hi((size_t)words), ok((size_t)words);
  for (int64_t w = 0; w < words; w++) plane_word(ref_codes, n_codes, w, &lo[(size_t)w], &hi[(size_t)w], &ok[(size_t)w]);
  std::vector<uint32_t> packed((size_t)(len2 / 8 + 2), 0);
  uint8_t* pb = (uint8_t*)packed.data();
  for (int r = 0; r < len2; r++) pb[r >> 1] |= (uint8_t)((read_codes[r] & 15) << ((r & 1) * 4));
  RefPlanes rp{lo.data(), hi.data(), ok.data()};
  // (opts & 8: the table without the N columns' spellings, as for a reference that has none)
  bool has_n = false;
  for (int64_t p = 0; p < n_codes; p++) has_n |= ref_codes[p] > 3;
  const int wild = has_n && !(opts & 8) ? BX_WILD : 0;
  const uint32_t kslots = kh_slots_for_entries(n_codes, wild ? kh_wild_entries(ref_codes, n_codes, wild) : 0);
  std::vector<uint32_t> kslot((size_t)kslots * 4, KH_EMPTY);
  std::vector<int32_t> kovf((size_t)kslots * 2, 0);
  const KmerHash ko{kslot.data(), kovf.data(), kslots - 1, kh_shift_for(kslots), wild};
  for (int64_t p = 0; p < n_codes; p++) kh_insert_wild_host(kslot.data(), kovf.data(), kslots - 1, ko.shift, ref_codes, n_codes, p, wild);
  // the tables of a matrix pair are made once (the stray tables take a moment) and kept
  static std::vector<int32_t> sub, mrow, key;
  static std::vector<int16_t> loss, dl;
  static int32_t s_min_m = 0, s_max_m = 0;
  static bool s_ok = false;
  std::vector<int32_t> k2(fwd, fwd + PSSM_WORDS);
  k2.insert(k2.end(), rc, rc + PSSM_WORDS);
  if (k2 != key) {
    key = k2;
    sub.assign(BX_SUB_WORDS, 0); mrow.assign(2 * 31 * 4, 0); loss.assign(BX_LOSS_WORDS, 0); dl.assign(BX_DL_WORDS, 0);
    s_ok = bx_make_tables(fwd, rc, sub.data(), mrow.data(), loss.data(), dl.data(), &s_min_m, &s_max_m);
  }
  if (!s_ok) return 0;
  std::vector<int32_t> sub256(BX_SUB_WORDS, 0);
  BxTab T{sub.data(), mrow.data(), loss.data(), dl.data(), s_min_m, s_max_m, BX_MAXW};
  for (int k = 0; k < BX_SUB_WORDS; k++) sub256[(size_t)k] = sub[(size_t)k] * 256;
  std::vector<uint32_t> nib((size_t)bx_nib_words(n_codes), 0x44444444u);
  for (int64_t p = 0; p < n_codes; p++) {
    const int64_t q = p + BX_NIB_LEAD;
    nib[(size_t)(q >> 3)] = (nib[(size_t)(q >> 3)] & ~(0xFu << (4 * (q & 7)))) | ((uint32_t)(ref_codes[p] > 4 ? 4 : ref_codes[p]) << (4 * (q & 7)));
  }
  BxPlan bp;
  // (opts >> 16 = d + 1: the QUICK plan for diagonal d alone -- bx_quick, what k_bx_plan<NW, 4> runs -- instead of the full plan)
  const int quick_d = (opts >> 16) - 1;
  if (quick_d >= 0) {
    // (the reference as it stands is the unwrapped one: every start position counts; opts & 512: a circular reference whose last 256 codes
    // are the wrap -- the places the bitmaps count are the ones in front of it)
    const int64_t L = (opts & 512) && n_codes > 2 * 256 ? n_codes - 256 : n_codes;
    std::vector<KbPair> bits((size_t)KB_WORDS, KbPair{0u, 0u});
    for (int64_t p = 0; p < L; p++) kmer_bits_insert(ref_codes, n_codes, p, bits.data());
    const KmerBits kb{bits.data(), (int32_t)L};
    if (!bx_plan_quick(rp, ko, kb, n_codes, ref_start, len1, pb, len2, strand, quick_d, T, &bp)) { bp.mode = BX_NONE; bp.b0 = 0; }
  }
  else bx_plan(rp, ko, n_codes, ref_start, len1, pb, len2, strand, T, &bp);
  out6[5] = 0;
  if (bp.mode == BX_NONE) { plan5[3] = bp.b0; return 0; }      // (b0 = the reason, BXF_*)
  plan5[0] = bp.d0; plan5[1] = bp.w; plan5[2] = bp.dstar; plan5[3] = bp.b0; plan5[4] = bp.edge;
  const int u = bx_umax(mrow.data(), pb, len2, strand);
  const int expect = u - bp.b0;
  int how = 0;
  if (bp.mode == BX_DONE) {
    how = 1;
    out6[0] = expect; out6[1] = bp.dstar; out6[2] = bp.dstar + len2 - 1; out6[3] = 0; out6[4] = 0;
    for (int r = 0; r < len2; r++) cols[r] = (int16_t)(bp.dstar + r);
  }
  int cls = bx_class_of(bp.w) + ((opts >> 4) & 3);
  if (cls > 4) cls = 4;
  const int wc = bx_class_width(cls);
  const bool interior = bp.d0 >= 0 && len2 - 1 + bp.d0 + wc <= len1;
  const bool edge = !interior || (opts & 2);
  const bool force_trace = (opts & 1) != 0;
  if (bp.mode == BX_DONE && !force_trace) { out6[5] = 1; return bp.mode; }
  const int32_t* st_sub = sub.data() + strand * 31 * 4 * BX_SUB_ROW;
  const int32_t* st_sub256 = sub256.data() + strand * 31 * 4 * BX_SUB_ROW;
  int32_t o2[6] = {0, 0, 0, 0, 0, 0};
  std::vector<int16_t> c2((size_t)len2 + 8, -9);
  const bool want_values = bp.mode == BX_VALUES || bp.mode == BX_DONE;
  const bool want_trace = bp.mode == BX_TRACE || force_trace;
  const uint32_t* rw = packed.data();
  int h2;
  switch (cls) {
    case 0: h2 = emu_bandx_w<8>(nib.data(), ref_start, len1, rw, len2, bp, edge, st_sub, st_sub256, want_values, want_trace, expect, o2, c2.data()); break;
    case 1: h2 = emu_bandx_w<16>(nib.data(), ref_start, len1, rw, len2, bp, edge, st_sub, st_sub256, want_values, want_trace, expect, o2, c2.data()); break;
    case 2: h2 = emu_bandx_w<24>(nib.data(), ref_start, len1, rw, len2, bp, edge, st_sub, st_sub256, want_values, want_trace, expect, o2, c2.data()); break;
    case 3: h2 = emu_bandx_w<32>(nib.data(), ref_start, len1, rw, len2, bp, edge, st_sub, st_sub256, want_values, want_trace, expect, o2, c2.data()); break;
    default: h2 = emu_bandx_w<64>(nib.data(), ref_start, len1, rw, len2, bp, edge, st_sub, st_sub256, want_values, want_trace, expect, o2, c2.data()); break;
  }
  if (h2 < 0) return h2;
  if (how == 1) {
    // the plan finished the read; whatever the DP stages found must be the same alignment
    if (h2 == 0) return -103;
    if (o2[0] != out6[0] || o2[1] != out6[1] || o2[2] != out6[2] || o2[3] != out6[3] || o2[4] != 0) return -104;
    for (int r = 0; r < len2; r++) if (c2[r] != cols[r]) return -105;
  } else {
    how = h2;
    for (int k = 0; k < 5; k++) out6[k] = o2[k];
    for (int r = 0; r < len2; r++) cols[r] = c2[r];
  }
  out6[5] = how;
  return bp.mode;
}

extern "C" void emu_bandx_diag(int32_t* out8) { for (int k = 0; k < 8; k++) out8[k] = mia::bx_diag[k]; }
