// bandx_lanes.h -- the band DP of bandx_body.h with a read spread over W/8 lanes (gfx950 only: DPP).
//
// bx_values / bx_trace keep a whole band row -- W = 8 ... 32 cells -- in the registers of ONE lane and walk it cell by
// cell: a wavefront issues one instruction per four cycles whatever it does, so a chunk of 64 such reads lasts
// rows x W x 13-15 instructions x 4+ cycles, 0.1-0.3 ms for the wide classes, and with fewer chunks than the chip has wave
// slots each band kernel lasted exactly as long as its widest chunk (round 2: k_bx_trace 0.29 ms at 18 % of the issue
// rate).  Here a read of class W owns LPR = W / 8 neighbouring lanes of one 16-lane DPP row, eight cells each: every
// chunk -- 64, 32, 20 or 16 reads -- walks eight cells per row, and the wide classes simply make more wavefronts.
//
// What crosses a lane boundary (recurrence: /root/reference/src/mia.c:740-981, restated in bandx_body.h):
//   * the column-gap maximum G(j) = max over j' < j of prev[j'] - GOP - GEP (j - j'): a prefix maximum over the PREVIOUS
//     row.  Each lane folds its own eight cells first (g_loc), the lanes to its left hand theirs over by row_shr, aged by
//     eight positions per lane in between; the cell loop then starts from that G instead of "none".
//   * the row-gap candidates H slide down one band index per row (the band moves one column to the right): what a lane
//     computes for its cell 0 becomes its left neighbour's H[7] (row_shl:1); the rightmost lane's H[7] is "none".
//   * the best cell of the last row (first maximum: the lower band index wins ties, src/mia.c:1278-1302) is reduced
//     across the read's lanes; the traceback is walked by the read's first lane, which reads the other lanes' trace bytes
//     from the slab.
// Tie rules, packed words (value * 256 + code), the reference's index-0 quirk: exactly as bx_values / bx_trace, which stay
// in bandx_body.h as the one-lane statement of the same recurrence (tests/test_emul_bandx.py runs them on the CPU;
// MIA_HIP_NO_LANES=1 runs them on the GPU; tests/test_gpu_bandx.py and the campaigns hold these kernels against the
// full-window kernels and the oracle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "bandx_body.h"

namespace mia {

constexpr int BXL_CELLS = 8;                       // band cells per lane
template <int LPR> constexpr int bxl_reads_per_row() { return 16 / LPR; }
template <int LPR> constexpr int bxl_reads_per_wave() { return 4 * (16 / LPR); }
__host__ __device__ constexpr int bxl_chunk_reads(int cls) { return cls == 0 ? 64 : (cls == 1 ? 32 : (cls == 2 ? 20 : (cls == 3 ? 16 : 8))); }

// value of the lane K places to the left (right) in the 16-lane row; lanes without such a neighbour get `fill`
template <int K>
__device__ __forceinline__ int bxl_from_left(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x110 + K, 0xF, 0xF, false); }
__device__ __forceinline__ int bxl_from_right1(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x101, 0xF, 0xF, false); }
template <int K>
__device__ __forceinline__ int bxl_from_right(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x100 + K, 0xF, 0xF, false); }

// The column-gap maximum a lane's cells start from: the best of the lanes to its left within the read, the lane k places
// away aged by (k - 1) * age (age = what a running maximum loses over one lane's eight cells; 0 in the coordinates of
// bxl_values_star).  Up to four lanes: one DPP move per neighbour; eight lanes: an inclusive scan in three doubling steps
// (S(u) = max over k >= 0 of v(u - k) + k age), handed on by one lane.  u: this lane's place in its read; none: "no cell".
template <int LPR>
__device__ __forceinline__ int bxl_prefix_left(int v, int u, int age, int none) {
  int G = none;
  if (LPR <= 1) return G;
  if (LPR <= 4) {
    const int g1 = bxl_from_left<1>(v, none);
    if (u >= 1) G = g1;
    if (LPR > 2) { const int g2 = bxl_from_left<2>(v, none) + age; if (u >= 2 && g2 > G) G = g2; }
    if (LPR > 3) { const int g3 = bxl_from_left<3>(v, none) + 2 * age; if (u >= 3 && g3 > G) G = g3; }
    return G;
  }
  static_assert(LPR <= 4 || LPR == 8, "a read owns 1, 2, 3, 4 or 8 lanes of a 16-lane row");
  int s = v;
  { const int t = bxl_from_left<1>(s, none) + age; if (u >= 1 && t > s) s = t; }
  { const int t = bxl_from_left<2>(s, none) + 2 * age; if (u >= 2 && t > s) s = t; }
  { const int t = bxl_from_left<4>(s, none) + 4 * age; if (u >= 4 && t > s) s = t; }
  const int g = bxl_from_left<1>(s, none);
  return u >= 1 ? g : none;
}
// First maximum of the last row over the read's lanes (the lower band index wins ties: a lane further right must be strictly
// better, src/mia.c:1278-1302); valid on the read's first lane.
template <int LPR>
__device__ __forceinline__ void bxl_first_max(int& best, int& bj) {
  if (LPR <= 1) return;
  if (LPR <= 4) {
    const int b1 = bxl_from_right<1>(best, BX_NEG), j1 = bxl_from_right<1>(bj, -1);
    int bb = best, jj = bj;
    if (LPR > 3) {
      const int b2 = bxl_from_right<2>(best, BX_NEG), j2 = bxl_from_right<2>(bj, -1);
      const int b3 = bxl_from_right<3>(best, BX_NEG), j3 = bxl_from_right<3>(bj, -1);
      if (b1 > bb) { bb = b1; jj = j1; }
      if (b2 > bb) { bb = b2; jj = j2; }
      if (b3 > bb) { bb = b3; jj = j3; }
    } else if (LPR > 2) {
      const int b2 = bxl_from_right<2>(best, BX_NEG), j2 = bxl_from_right<2>(bj, -1);
      if (b1 > bb) { bb = b1; jj = j1; }
      if (b2 > bb) { bb = b2; jj = j2; }
    } else {
      if (b1 > bb) { bb = b1; jj = j1; }
    }
    best = bb; bj = jj;
    return;
  }
  // eight lanes: three halving steps towards the first lane (it only ever takes in lanes of its own read)
  { const int b = bxl_from_right<1>(best, BX_NEG), j = bxl_from_right<1>(bj, -1); if (b > best) { best = b; bj = j; } }
  { const int b = bxl_from_right<2>(best, BX_NEG), j = bxl_from_right<2>(bj, -1); if (b > best) { best = b; bj = j; } }
  { const int b = bxl_from_right<4>(best, BX_NEG), j = bxl_from_right<4>(bj, -1); if (b > best) { best = b; bj = j; } }
}

// eight nibbles of reference codes at a position that advances by one per row (BxSlide<1> of bandx_body.h)
// The operands of a lane's rows -- eight nibbles of reference codes at a position that advances by one per row, and the read's own
// code of the row -- eight rows at a time.  A block of eight rows needs 8 + 7 + 7 nibbles of reference: three words, held in
// registers; the word the NEXT block needs on top is asked for at the head of the block and taken over behind its last row, in a
// different basic block, so that the wait for it falls eight rows after the load.  (With one `if (position crossed a word) load`
// per row the compiler put the wait right behind the load: ~1 us of L2 latency every eight rows of every chunk, twice -- the
// read's words came the same way.)
struct BxlFeed {
  uint32_t raw[3], rw;       // reference words q, q + 1, q + 2 of this block; the read's word of this block
  uint32_t raw_n, rw_n;      // in flight: reference word q + 3, the read's next word
  const uint32_t* rp;        // refnib + q + 3 of this block
  const uint32_t* wp;        // the read's words, from this block's next one on
  int a;                     // nibble offset of row 8 b in raw[0]
  int more;                  // words of the read left behind rw_n's (packed reads are padded: one word beyond the last is readable)
  __device__ __forceinline__ void init(const uint32_t* refnib, int64_t nib, const uint32_t* rwords, int len2) {
    const int64_t q = nib >> 3;
    a = (int)(nib & 7);
    raw[0] = refnib[q]; raw[1] = refnib[q + 1]; raw[2] = refnib[q + 2];
    rp = refnib + q + 3;
    rw = rwords[0];
    wp = rwords + 1;
    more = ((len2 - 1) >> 3);
  }
  __device__ __forceinline__ void block_begin() { raw_n = *rp; rw_n = *wp; }
  __device__ __forceinline__ void block_end() {
    raw[0] = raw[1]; raw[1] = raw[2]; raw[2] = raw_n; rp++;
    rw = rw_n; if (more > 0) { wp++; more--; }
  }
  // row 8 b + k of the block: the eight reference nibbles from the lane's position on / the read's code
  __device__ __forceinline__ uint32_t cw(int k) const {
    const int t = a + k;       // 0 .. 14
    const bool hi = t >= 8;
    return __builtin_amdgcn_alignbit(hi ? raw[2] : raw[1], hi ? raw[1] : raw[0], (uint32_t)(4 * t) & 31u);
  }
  __device__ __forceinline__ int code(int k) const { return (int)((rw >> (4 * k)) & 3u); }
};
// for (r = 1; r < len2; r++) body(r, cw, code), fed block by block (row 0 is the caller's: feed.cw(0), feed.code(0) before this)
template <class F>
__device__ __forceinline__ void bxl_rows(BxlFeed& feed, int len2, F&& body) {
  for (int r0 = 0; r0 < len2; r0 += 8) {
    feed.block_begin();
    const int k_hi = len2 - r0 < 8 ? len2 - r0 : 8;
    for (int k = r0 ? 0 : 1; k < k_hi; k++) body(r0 + k, feed.cw(k), feed.code(k));
    feed.block_end();
  }
}
// the row's eight substitution scores: all eight LDS reads issued before anything waits for one (left to itself the compiler
// interleaved them with the recurrence, one read and one wait per cell: eight LDS round trips per row)
#define BXL_SUB_ROW(sb, row, cw)                                                              \
  int sb[BXL_CELLS];                                                                          \
  _Pragma("unroll") for (int j_ = 0; j_ < BXL_CELLS; j_++) sb[j_] = (row)[__builtin_amdgcn_ubfe((cw), 4 * j_, 3)]; \
  __builtin_amdgcn_sched_barrier(0)

// cells of this lane (band indices 8u .. 8u + 7) whose column lies inside the window [0, len1)
__device__ __forceinline__ uint32_t bxl_live(int c0_lane, int len1) {
  const int jlo = c0_lane < 0 ? -c0_lane : 0, jhi = (len1 - c0_lane) < BXL_CELLS ? (len1 - c0_lane) : BXL_CELLS;
  return jhi > jlo ? (((1u << jhi) - 1u) & ~((1u << jlo) - 1u)) : 0u;
}

// ---- values only ---------------------------------------------------------------------------------------------------------
// u: this lane's place among the read's LPR lanes.  Every lane of a read is given the same read.  *best_out / *bj_out
// (band index, -1: none) are valid on the read's first lane.
template <int LPR, bool EDGE>
__device__ __forceinline__ void bxl_values(const uint32_t* refnib, int s, int len1, const uint32_t* rwords, int len2, int d0, const int32_t* sub, int u,
                                           int* best_out, int* bj_out) {
  const int dl = d0 + BXL_CELLS * u;                       // this lane's first diagonal
  int32_t P[BXL_CELLS], H[BXL_CELLS];
  BxlFeed feed;
  feed.init(refnib, (int64_t)s + dl + BX_NIB_LEAD, rwords, len2);
  {
    const uint32_t cw = feed.cw(0);
    const int32_t* row = sub + feed.code(0) * BX_SUB_ROW;              // depth 0
    const uint32_t live = EDGE ? bxl_live(dl, len1) : 0xFFu;
#pragma unroll
    for (int j = 0; j < BXL_CELLS; j++) {
      H[j] = BX_NEG;
      const int v = row[(cw >> (4 * j)) & 7u];
      P[j] = ((live >> j) & 1u) ? v : BX_NEG;
    }
  }
  bxl_rows(feed, len2, [&](int r, uint32_t cw, int code) {
    const int c0 = r + dl;
    const int32_t* row = sub + (sm_depth(r, len2) * 4 + code) * BX_SUB_ROW;
    BXL_SUB_ROW(sbv, row, cw);
    uint32_t live = 0xFFu, col0 = 0u;
    if (EDGE) {
      live = bxl_live(c0, len1);
      col0 = (c0 <= 0 && c0 > -BXL_CELLS) ? (1u << (-c0)) : 0u;
    }
    const int fresh = -(GOP + GEP * (r + 1));
    // the column-gap maximum this lane's cells start from: the lanes to the left, each aged by the positions in between
    int cand[BXL_CELLS];
    int G = BX_NEG;
    if (LPR > 1) {
      int gloc = BX_NEG;
#pragma unroll
      for (int j = 0; j < BXL_CELLS; j++) {
        cand[j] = P[j] - (GOP + GEP);
        gloc = gloc - GEP > cand[j] ? gloc - GEP : cand[j];
      }
      G = bxl_prefix_left<LPR>(gloc, u, -BXL_CELLS * GEP, BX_NEG);
    } else {
#pragma unroll
      for (int j = 0; j < BXL_CELLS; j++) cand[j] = P[j] - (GOP + GEP);
    }
    int nh0 = BX_NEG;
#pragma unroll
    for (int j = 0; j < BXL_CELLS; j++) {
      const int pd = P[j], h = H[j];
      const int sb = sbv[j];
      const int x = pd > G ? (pd > h ? pd : h) : (G > h ? G : h);
      int cur = fresh > x ? fresh : x + sb;
      if (EDGE) {
        if ((col0 >> j) & 1u) cur = sb + fresh;                                       // src/mia.c:805-822
        if (!((live >> j) & 1u)) cur = BX_NEG;
      }
      G = G - GEP > cand[j] ? G - GEP : cand[j];
      const int nh = h - GEP > cand[j] ? h - GEP : cand[j];
      if (j >= 1) H[j - 1] = nh; else nh0 = nh;
      P[j] = cur;
    }
    if (LPR > 1) {
      const int hr = bxl_from_right1(nh0, BX_NEG);
      H[BXL_CELLS - 1] = u < LPR - 1 ? hr : BX_NEG;
    } else H[BXL_CELLS - 1] = BX_NEG;
  });
  int best = BX_NEG, bj = -1;
#pragma unroll
  for (int j = 0; j < BXL_CELLS; j++) if (P[j] > best) { best = P[j]; bj = j; }
  bj = bj < 0 ? -1 : bj + BXL_CELLS * u;
  bxl_first_max<LPR>(best, bj);                            // (meaningful on the read's first lane, u == 0)
  if (best <= BX_NEG / 2) bj = -1;
  *best_out = best; *bj_out = bj;
}

// ---- values only, windows that hold the whole band (no EDGE), in coordinates that need no ageing ---------------------------
// bxl_values ages two running maxima per cell (G - GEP per band index, H - GEP per row and index) and decides "new start"
// with a compare and a select: 11 vector instructions per cell, and with the position-specific matrices' long values lists
// (configs[2]: 333 k reads) this kernel runs at 69 % of the chip's issue rate -- instructions are what it costs.  Here
// every value is carried as  S*(r, j) = S(r, j) + 2 GEP r + GEP j  (row r, band index j, src/mia.c:905-948 restated):
//   * the column-gap maximum of cell (r, j) over the previous row, max over j' < j of S(r-1, j') - GOP - GEP (j - j'),
//     becomes max over j' < j of Q(j') - GOP with Q(j') = S*(r-1, j') + 2 GEP: a plain prefix maximum, across lanes too;
//   * the row-gap candidate that slides from (r, j) to (r+1, j-1) and loses GEP on the way keeps its value
//     (2 GEP (r+1) + GEP (j-1) = 2 GEP r + GEP j + GEP): H(r+1, j-1) = max(H(r, j), Q(j) - GOP), the SAME Q(j) - GOP;
//   * the diagonal predecessor of (r, j) is Q(j) itself, and Q of the new row is max3(Q, G, H) + sub + 2 GEP.
// Seven instructions per cell.  The "new start" branch (score -(GOP + GEP (r+1)), which forfeits the row's substitution
// score) is only kept for rows r < r_cut, as max(x + sub, fresh): (1) any cell may come out HIGHER than dyn_prog's without
// harm -- the caller accepts a read only if the best of the last row equals the plan's diagonal's score `expect` at the
// plan's band index and first: values never below the true ones, a true optimum >= expect (the diagonal is a path), so
// equality with expect at the first maximum carries over; (2) a path that starts in row r scores at most
// U - sum of M(0..r) - GOP - GEP (r+1), below expect = U - b0 from r_cut on (the first r where sum M(0..r) + GOP + GEP (r+1)
// > b0): cells that would take the branch there never reach a last-row value >= expect, with or without it.
// sub: the unshifted table (the EDGE form of the same chunk class reads it too).  b0 < 0: not known, the branch stays in every row.
template <int LPR>
__device__ __forceinline__ void bxl_values_star(const uint32_t* refnib, int s, const uint32_t* rwords, int len2, int d0, const int32_t* sub, int u, int b0,
                                                int* best_out, int* bj_out) {
  const int dl = d0 + BXL_CELLS * u;
  int32_t Q[BXL_CELLS], H[BXL_CELLS];
  BxlFeed feed;
  feed.init(refnib, (int64_t)s + dl + BX_NIB_LEAD, rwords, len2);
  // rows below r_cut keep the new-start branch (this lane's read; the wavefront runs the largest)
  int r_cut = len2;
  if (b0 >= 0) {
    int acc = GOP, q = 0;
    uint32_t w = feed.rw;
    for (; q < len2; q++) {
      if (q && (q & 7) == 0) w = rwords[q >> 3];
      const int b = (int)((w >> (4 * (q & 7))) & 3u);
      acc += sub[(sm_depth(q, len2) * 4 + b) * BX_SUB_ROW + b] + GEP;
      if (acc > b0) break;
    }
    r_cut = q < len2 ? q : len2;
  }
  for (int o = 32; o; o >>= 1) { const int t = __shfl_xor(r_cut, o); r_cut = t > r_cut ? t : r_cut; }
  r_cut = __builtin_amdgcn_readfirstlane(r_cut);            // (the same in every lane: a scalar, so that `early` below is a branch)
  {
    const uint32_t cw = feed.cw(0);
    const int32_t* row = sub + feed.code(0) * BX_SUB_ROW;              // depth 0
#pragma unroll
    for (int j = 0; j < BXL_CELLS; j++) {
      H[j] = BX_NEG;
      Q[j] = row[__builtin_amdgcn_ubfe(cw, 4 * j, 3)] + 2 * GEP + GEP * (BXL_CELLS * u + j);
    }
  }
  int fl = -GOP - GEP + 2 * GEP + GEP * BXL_CELLS * u + GEP;           // fresh*(r, 8 u) + 2 GEP at r = 1
  // one row; EARLY: with the new-start branch (two loops below: a flag tested per cell would be a select per cell)
  auto row_step = [&](int r, uint32_t cw, int code, auto early_tag) {
    constexpr bool EARLY = decltype(early_tag)::value;
    const int32_t* row = sub + (sm_depth(r, len2) * 4 + code) * BX_SUB_ROW;
    BXL_SUB_ROW(sbv, row, cw);
    int cand[BXL_CELLS];
#pragma unroll
    for (int j = 0; j < BXL_CELLS; j++) cand[j] = Q[j] - GOP;
    int G = BX_NEG;
    if (LPR > 1) {
      int gloc = cand[0];
#pragma unroll
      for (int j = 1; j < BXL_CELLS; j++) gloc = gloc > cand[j] ? gloc : cand[j];
      G = bxl_prefix_left<LPR>(gloc, u, 0, BX_NEG);
    }
    int nh0 = BX_NEG;
#pragma unroll
    for (int j = 0; j < BXL_CELLS; j++) {
      const int pd = Q[j], h = H[j];
      const int sb = sbv[j];
      const int x = pd > G ? (pd > h ? pd : h) : (G > h ? G : h);
      int q = x + sb + 2 * GEP;
      if (EARLY) { const int f = fl + GEP * j; q = q > f ? q : f; }
      G = G > cand[j] ? G : cand[j];
      const int nh = h > cand[j] ? h : cand[j];
      if (j >= 1) H[j - 1] = nh; else nh0 = nh;
      Q[j] = q;
    }
    if (LPR > 1) {
      const int hr = bxl_from_right1(nh0, BX_NEG);
      H[BXL_CELLS - 1] = u < LPR - 1 ? hr : BX_NEG;
    } else H[BXL_CELLS - 1] = BX_NEG;
    fl += GEP;
  };
  // (blocks of eight rows, see BxlFeed: a block is of one kind or the other, the one r_cut falls into keeps the branch)
  for (int r0 = 0; r0 < len2; r0 += 8) {
    feed.block_begin();
    const int k_hi = len2 - r0 < 8 ? len2 - r0 : 8;
    if (r0 < r_cut) { for (int k = r0 ? 0 : 1; k < k_hi; k++) row_step(r0 + k, feed.cw(k), feed.code(k), std::true_type{}); }
    else { for (int k = 0; k < k_hi; k++) row_step(r0 + k, feed.cw(k), feed.code(k), std::false_type{}); }
    feed.block_end();
  }
  // back to scores: S(R, j) = Q(j) - 2 GEP - 2 GEP R - GEP j
  const int back = 2 * GEP + 2 * GEP * (len2 - 1) + GEP * BXL_CELLS * u;
  int best = BX_NEG, bj = -1;
#pragma unroll
  for (int j = 0; j < BXL_CELLS; j++) { const int v = Q[j] - back - GEP * j; if (v > best) { best = v; bj = j; } }
  bj = bj < 0 ? -1 : bj + BXL_CELLS * u;
  bxl_first_max<LPR>(best, bj);
  if (best <= BX_NEG / 2) bj = -1;
  *best_out = best; *bj_out = bj;
}

// ---- with a trace ----------------------------------------------------------------------------------------------------------
// trace: this WAVEFRONT's slab.  A trace word holds the codes of ONE cell (band index) in FOUR consecutive rows: rows 4q ..
// 4q + 3 of lane l's cell jj sit in word q * 512 + (jj >> 2) * 256 + l * 4 + (jj & 3): each of a lane's two 16-byte stores per four
// rows lies in a stretch of 1 KB that the wavefront's 64 lanes fill completely (whole cache lines: a line written in two halves by
// two instructions was fetched first -- 210 MB per launch, the size of the slabs), and the traceback -- which follows one band index down the rows -- finds sixteen rows in four
// words instead of sixteen (round 2's layout had a row's eight cells side by side: sixteen cache lines per fetch and lane,
// a third of the kernel's time).  lane_in_wave: this lane; the read's first lane walks the traceback and writes cols_out / res.  Returns (on the first lane) whether the read is
// finished here; false: the reference's index-0 quirk, or nothing alive in the last row.
template <int LPR, bool EDGE>
__device__ __forceinline__ bool bxl_trace(const uint32_t* refnib, int s, int len1, const uint32_t* rwords, int len2, int d0, const int32_t* sub256, int u,
                                          int lane_in_wave, uint32_t* trace, int16_t* cols_out, BxResult* res, bool lazy_diag, bool no_traceback = false) {
  constexpr int DEAD = BX_NEG * 256;
  constexpr int STEP = 1 - GEP * 256;                 // a running maximum ages by one position: value - GEP, length + 1
  constexpr int CAND = -GOP * 256 + STEP - 0xFF;      // a cell (packed as a diagonal source) becomes a gap source
  constexpr int BLK_WORDS = 64 * BXL_CELLS;           // words of four rows of the wavefront
  const int R = len2 - 1;
  const int dl = d0 + BXL_CELLS * u;
  int32_t P[BXL_CELLS], H[BXL_CELLS];
  uint32_t tb[BXL_CELLS];                             // the four rows' codes of each of this lane's cells
  BxlFeed feed;
  feed.init(refnib, (int64_t)s + dl + BX_NIB_LEAD, rwords, len2);
  uint32_t* mine = trace + 4 * lane_in_wave;          // this lane's four words of cells 0..3; cells 4..7 sit 256 words further
  {
    const uint32_t cw = feed.cw(0);
    const int32_t* row = sub256 + feed.code(0) * BX_SUB_ROW;
    const uint32_t live = EDGE ? bxl_live(dl, len1) : 0xFFu;
#pragma unroll
    for (int j = 0; j < BXL_CELLS; j++) {
      H[j] = DEAD;
      const int v = row[(cw >> (4 * j)) & 7u];
      P[j] = ((live >> j) & 1u) ? (v | 0xFF) : DEAD;
      tb[j] = 0xFFFFFFFFu;                            // row 0 reads as a diagonal step (the walk stops there anyway)
    }
    if (R == 0) {
#pragma unroll
      for (int j = 0; j < BXL_CELLS; j += 4) *reinterpret_cast<uint4*>(mine + (j >> 2) * 256) = make_uint4(tb[j], tb[j + 1], tb[j + 2], tb[j + 3]);
    }
  }
  bxl_rows(feed, len2, [&](int r, uint32_t cw, int code) {
    const int c0 = r + dl;
    const int32_t* row = sub256 + (sm_depth(r, len2) * 4 + code) * BX_SUB_ROW;
    BXL_SUB_ROW(sbv, row, cw);
    uint32_t live = 0xFFu, col0 = 0u;
    if (EDGE) {
      live = bxl_live(c0, len1);
      col0 = (c0 <= 0 && c0 > -BXL_CELLS) ? (1u << (-c0)) : 0u;
    }
    const int f0 = -(GOP + GEP * (r + 1)) * 256, f0s = f0 | 0x80;
    int cand[BXL_CELLS];
    int G = DEAD;
    if (LPR > 1) {
      int gloc = DEAD;
#pragma unroll
      for (int j = 0; j < BXL_CELLS; j++) {
        cand[j] = (P[j] | 0xFF) + CAND;
        gloc = gloc + STEP > cand[j] ? gloc + STEP : cand[j];
      }
      G = bxl_prefix_left<LPR>(gloc, u, BXL_CELLS * STEP, DEAD);
    } else {
#pragma unroll
      for (int j = 0; j < BXL_CELLS; j++) cand[j] = (P[j] | 0xFF) + CAND;
    }
    const int rk = r & 3;
    int nh0 = DEAD;
#pragma unroll
    for (int j = 0; j < BXL_CELLS; j++) {
      const int pd = P[j] | 0xFF, h = H[j], gc = G | 0x40;
      const int sb = sbv[j];
      const int x = pd > gc ? (pd > h ? pd : h) : (gc > h ? gc : h);
      int cur = f0 > x ? f0s : x + sb;                  // fresh must beat all three strictly (its code byte is 0 in f0)
      if (EDGE) {
        if ((col0 >> j) & 1u) cur = (sb + f0) | 0xFF;
        if (!((live >> j) & 1u)) cur = DEAD;
      }
      tb[j] = bx_put(tb[j], (uint32_t)cur, rk);
      G = G + STEP > cand[j] ? G + STEP : cand[j];
      const int nh = h + STEP > cand[j] ? h + STEP : cand[j];
      if (j >= 1) H[j - 1] = nh; else nh0 = nh;
      P[j] = cur;
    }
    if (LPR > 1) {
      const int hr = bxl_from_right1(nh0, DEAD);
      H[BXL_CELLS - 1] = u < LPR - 1 ? hr : DEAD;
    } else H[BXL_CELLS - 1] = DEAD;
    if (rk == 3 || r == R) {                            // (the bytes of rows beyond the read's last are never looked at)
      uint32_t* tr = mine + (uint32_t)(r >> 2) * BLK_WORDS;
#pragma unroll
      for (int j = 0; j < BXL_CELLS; j += 4) *reinterpret_cast<uint4*>(tr + (j >> 2) * 256) = make_uint4(tb[j], tb[j + 1], tb[j + 2], tb[j + 3]);
    }
  });
  // max_sg_score: first maximum of the last row (src/mia.c:1278-1302)
  int best = BX_NEG, bj = -1;
#pragma unroll
  for (int j = 0; j < BXL_CELLS; j++) if ((P[j] >> 8) > best) { best = P[j] >> 8; bj = j; }
  bj = bj < 0 ? -1 : bj + BXL_CELLS * u;
  bxl_first_max<LPR>(best, bj);
  // the traceback is the first lane's business.  It reads trace bytes its neighbours stored: same wavefront, same memory
  // pipe, the loads are issued after the stores -- the fence only keeps the compiler from moving them.
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  if (u != 0 || no_traceback) return false;
  if (bj < 0 || best <= BX_NEG / 2) return false;
  // find_align_begin + populate_pwaln_to_begin (src/mia.c:612-637, 1440-1497).  The path is a handful of diagonal
  // stretches: the walk only notes where they break (sixteen rows of one band index per fetch -- nearly every step is
  // diagonal and stays on it), the script is written afterwards, four rows per 8-byte store.  (One 2-byte store per row and
  // lane -- 64 different cache lines per instruction -- was a third of this kernel's time.)
  constexpr int W = BXL_CELLS * LPR;
  constexpr int EV = 4;                                  // breaks kept in registers; a path with more is walked again, storing as it goes
  int ev_row[EV], ev_delta[EV];                          // rows above ev_row sit ev_delta columns further right than the diagonal below says
#pragma unroll
  for (int k = 0; k < EV; k++) { ev_row[k] = -1; ev_delta[k] = 0; }
  int r = R, c = R + d0 + bj, gaps = 0;
  uint32_t gap_desc = 0;
  const int aec = c;
  // Nearly every step is diagonal and stays on its band index j: the four words that hold sixteen rows of that index are
  // fetched at once, turned into a bit mask of "not a diagonal step" and the whole run is taken in one go.  Only where a
  // run breaks is the byte itself looked at.
  for (;;) {
    if (r == 0 || c == 0) break;
    const int j = c - r - d0;
    if (j < 0 || j >= W) return false;
    const uint32_t col = (uint32_t)(((j >> 2) & 1) * 256 + (lane_in_wave + (j >> 3)) * 4 + (j & 3));      // (band index j: cell j & 7 of the lane j >> 3 places to the right)
    const int q0 = r >> 2;
    uint32_t nd = 0;                         // bit i: row 4 (q0 - 3) + i holds something else than a diagonal step
    uint32_t wq[4];                          // (all four loads before the first wait: one after the other they were four trips to the L2 per step of the walk)
#pragma unroll
    for (int t = 0; t < 4; t++) { const int q = q0 - t; wq[t] = trace[(uint32_t)(q > 0 ? q : 0) * BLK_WORDS + col]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int q = q0 - t;
      const uint32_t wv = wq[t];
      const uint32_t x = ~wv, y = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;      // bit 8 k + 7: byte k is not 0xFF
      const uint32_t z = y >> 7, nib = (z | (z >> 7) | (z >> 14) | (z >> 21)) & 15u;
      if (q >= 0) nd |= nib << (4 * (3 - t));  // (blocks above the matrix: all diagonal; the walk never gets there: kmax)
    }
    const int rel = r - 4 * (q0 - 3);        // 12 .. 15: this row's bit
    const uint32_t low = nd & ((2u << rel) - 1u);
    const int run = low ? rel - (31 - __builtin_clz(low)) : rel + 1;
    const int kmax = r < c ? r : c;
    const int k = run < kmax ? run : kmax;
    r -= k; c -= k;
    if (k == kmax) break;                    // row 0 or column 0
    if (!low) continue;
    const int code = (int)((trace[(uint32_t)(r >> 2) * BLK_WORDS + col] >> (8 * (r & 3))) & 255u);
    if (code == 0x80) break;
    int delta;
    if (code & 0x40) {
      const int sc = c - 1 - (code & 63);
      if (sc <= 0) return false;             // a gap from column 0 reads back as a diagonal step in the reference: not followed here
      gap_desc = 0u | ((uint32_t)r << 1) | ((uint32_t)(code & 63) << 10);
      delta = -(code & 63);
    } else {
      const int sr = r - 1 - code;
      if (sr <= 0) return false;             // a gap from row 0: the same quirk
      gap_desc = 1u | ((uint32_t)(sr + 1) << 1) | ((uint32_t)code << 10);
      delta = code;
    }
#pragma unroll
    for (int e = 0; e < EV; e++) if (e == gaps) { ev_row[e] = r; ev_delta[e] = delta; }
    gaps++;
    if (delta < 0) { r--; c = c - 1 + delta; } else { r = r - 1 - delta; c--; }
  }
  const int abr = r, abc = c;
  res->score = best; res->abc = abc; res->aec = aec; res->abr = abr; res->gaps = gaps; res->gap_desc = gap_desc;
  if (gaps == 0 && abr == 0 && lazy_diag) return true;   // a pure diagonal from row 0: k_diag_scripts writes it when somebody asks
  if (gaps <= EV) {
    // row q (>= abr) sits on column q + off, off = aec - R plus the shifts of every break above it; the rows a row gap
    // skipped are inserts
    const int off0 = aec - R;
    for (int q0 = 0; q0 < len2; q0 += 4) {
      // a group of four rows that no break touches -- nearly all of them -- is four consecutive columns
      int off = off0;
      bool clean = q0 >= abr;
#pragma unroll
      for (int e = 0; e < EV; e++) {
        const int er = ev_row[e], d = ev_delta[e];       // (unused entries: row -1, never above anything)
        if (er > q0) {
          off += d;
          if (er - (d > 0 ? d : 0) <= q0 + 3) clean = false;       // the break itself, or the rows a row gap skipped, fall into the group
        }
      }
      uint2 w2;
      if (clean) {
        const uint32_t c0 = (uint32_t)(q0 + off);
        w2.x = (c0 & 0xFFFFu) | ((c0 + 1u) << 16);
        w2.y = ((c0 + 2u) & 0xFFFFu) | ((c0 + 3u) << 16);
      } else {
        uint32_t v[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int q = q0 + t;
          int col = q + off0;
          bool ins = false;
#pragma unroll
          for (int e = 0; e < EV; e++) {
            const bool above = ev_row[e] > q;
            const int d = ev_delta[e];
            if (above) col += d;                         // a column gap of n: the rows above it lie n columns further LEFT (d = -n); a row gap of n rows: n further RIGHT
            ins = ins || (above && d > 0 && q >= ev_row[e] - d);
          }
          v[t] = (uint32_t)(uint16_t)(int16_t)(q < abr ? COL_CLIP : (ins ? COL_INSERT : col));
        }
        w2.x = v[0] | (v[1] << 16);
        w2.y = v[2] | (v[3] << 16);
      }
      *reinterpret_cast<uint2*>(cols_out + q0) = w2;     // (the script row is a multiple of four entries long and 8-byte aligned)
    }
    return true;
  }
  // more breaks than the registers hold (rare): the same walk again, storing every row as it goes
  r = R; c = aec;
  for (;;) {
    const int j = c - r - d0;
    cols_out[r] = (int16_t)c;
    if (r == 0 || c == 0) break;
    const int code = (int)((trace[(uint32_t)(r >> 2) * BLK_WORDS + (uint32_t)(((j >> 2) & 1) * 256 + (lane_in_wave + (j >> 3)) * 4 + (j & 3))] >> (8 * (r & 3))) & 255u);
    if (code == 0x80) break;
    if (code == 0xFF) { r--; c--; continue; }
    if (code & 0x40) { r--; c = c - 1 - (code & 63); }
    else { const int sr = r - 1 - code; for (int q = r - 1; q > sr; q--) cols_out[q] = COL_INSERT; r = sr; c--; }
  }
  for (int q = 0; q < abr; q++) cols_out[q] = COL_CLIP;
  return true;
}

}  // namespace mia
