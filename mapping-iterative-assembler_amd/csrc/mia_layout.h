// mia_layout.h -- constants and packed-word layout shared by the HIP kernels,
// the C-ABI host code and the CPU lock-step emulation used by the tests.
//
// Scoring constants follow /root/reference/src/params.h:26-27,36,68.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MIA_HD __host__ __device__
#else
#define MIA_HD
#endif

namespace mia {

constexpr int GOP = 1000;           // gap open   (src/params.h:26)
constexpr int GEP = 200;            // gap extend (src/params.h:27)
constexpr int PSSM_DEPTH = 15;      // src/params.h:22
constexpr int FLAT_MATCH = 200;     // src/params.h:28
constexpr int FLAT_MISMATCH = -600; // src/params.h:29
constexpr int MAX_READ = 256;       // INIT_ALN_SEQ_LEN (src/params.h:68)
constexpr int REALIGN_BUFFER = 50;  // src/params.h:36
constexpr int PSSM_WORDS = 31 * 25;
constexpr int WAVE = 64;

// ---------------------------------------------------------------------------
// Packed 32-bit score words of the windowed DP kernel.
//
//   [ value + off : 32-SH bits ][ prio : 2 ][ idx/len : IB bits ]      SH = IB+2
//
// * "key" words feed the two running arg-maxima of dyn_prog (best_gap_col per
//   row, best_gap_row per column; src/mia.c:838-865): value = S + GEP*index,
//   idx = IDXM - index, prio = 0.  An unsigned max keeps the largest key and,
//   among equal keys, the smallest index -- the reference's "strictly greater
//   replaces" rule.
// * adding one per-column (or per-row) constant turns a key word into a
//   candidate word: value = score of arriving through that gap, prio = 2
//   (column gap) or 1 (row gap), idx field = gap length.
// * the diagonal candidate has prio 3, starting a new alignment prio 0, so one
//   unsigned max over the four words reproduces the cascade of
//   src/mia.c:910-948 (start only if strictly better; diag >= gapc >= gapr).
// ---------------------------------------------------------------------------
struct PackParams {
  int ib;        // index bits (8 for windows <= 256 columns, 10 up to 640)
  int sh;        // ib + 2
  uint32_t idxm; // (1<<ib)-1
  int32_t off;   // value offset making every real field positive
  int32_t unavail; // value field of the "no candidate yet" key
};

// M = largest |PSSM entry|; ncols = 64*CPL; returns false if the value field
// would overflow (caller routes such reads to the exact wide kernel).
inline bool make_pack_params(int ncols, int max_abs_sub, PackParams* p) {
  const int ib = ncols <= 256 ? 8 : 10;
  const int ncm = ncols > 256 ? ncols : 256;
  const int64_t u0 = GOP + (int64_t)GEP * ncm;
  const int64_t off = u0 + GOP + (int64_t)GEP * (MAX_READ + 1) + max_abs_sub + 16;
  const int64_t vmax = (int64_t)MAX_READ * max_abs_sub + (int64_t)GEP * ncm + off + max_abs_sub;
  p->ib = ib;
  p->sh = ib + 2;
  p->idxm = (1u << ib) - 1u;
  p->off = (int32_t)off;
  p->unavail = (int32_t)u0;
  return vmax < ((int64_t)1 << (32 - (ib + 2)));
}

// trace byte written per DP cell: [type:2][len:6]; len 63 = "63 or more"
// (such a read is re-run by the exact wide kernel).
constexpr uint32_t TR_START = 0, TR_ROWGAP = 1, TR_COLGAP = 2, TR_DIAG = 3;
constexpr uint32_t TR_LEN_SAT = 63;

// per-read status bits written by the align kernels
constexpr uint32_t ST_OK = 0;
constexpr uint32_t ST_ESCAPE = 1;    // saturated gap length met on the path: needs the wide kernel
constexpr uint32_t ST_TOO_LONG = 2;  // alignment longer than 512 columns (undefined in the reference, src/mia.c:1442-1450)
constexpr uint32_t ST_SKIPPED = 4;   // strand_known == 0 (src/mia_main.c:178)
constexpr uint32_t ST_BAND = 8;      // the path left the stored trace band of the quad kernel: re-run with a full trace

// exactly one gap, described in the upper bits (k_band_align): bit 8 = 1 for inserted read rows / 0 for deleted reference
// columns, bits 9-17 = first inserted row / first row after the deletion, bits 18-23 = length of the gap
constexpr uint32_t ST_ONEGAP = 32;
constexpr uint32_t ST_DIAG = 16;     // proven pure diagonal (k_align_quad_plain): no insert, no deletion, script = consecutive columns

constexpr int16_t COL_INSERT = -1;
constexpr int16_t COL_CLIP = -2;

// depth of the PSSM for read row `row` of a read of length `len` (src/pssm.c:38-46)
MIA_HD static inline __attribute__((always_inline)) int sm_depth(int row, int len) {
  if (row < PSSM_DEPTH) return row;
  if (len - (row + 1) < PSSM_DEPTH) return 2 * PSSM_DEPTH - (len - (row + 1));
  return PSSM_DEPTH;
}

constexpr int TALLY_WORDS = 12;
enum { T_A = 0, T_C, T_G, T_T, T_GAP, T_COV, T_SA, T_SC, T_SG, T_ST, T_SPAN, T_PAD };

}  // namespace mia
