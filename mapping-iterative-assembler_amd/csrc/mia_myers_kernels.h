// mia_myers_kernels.h -- unit-cost edit distance with IUPAC-compatible matching, the
// quantity /root/reference/src/myers_align.c:10-99 (myers_diff) returns, computed with
// Myers' bit-vector algorithm (1999) in the block formulation of Hyyro: 64 pattern rows
// per 64-bit word.  One pair per wavefront; lane l owns K consecutive 64-row blocks of
// seq_a and the lanes run as a systolic array: at step t lane l consumes character
// t - l of seq_b, taking the horizontal carry (-1, 0, +1) of lane l-1 from the
// previous step through a DPP shift.
//
// The reference walks furthest-reaching diagonals for d = 0 .. maxd-1 and returns d, or
// UINT_MAX when d >= maxd (loop `d != maxd`, :20).  Both procedures return the same
// number whenever it is below maxd:
//   mode 0 (global)      D[len_a][len_b]
//   mode 1 (:39 needs only x == len_b)   min_i D[i][len_b]
//   mode 2 (:40 needs only y == len_a)   min_j D[len_a][j]
// with D[i][0] = i, D[0][j] = j (the walk always starts at (0,0)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wave_dev.h"

namespace mia {

constexpr int MYERS_MAX_K = 8;   // blocks per lane -> seq_a up to 64*64*8 = 32768 characters

__host__ __device__ __forceinline__ uint32_t iupac_bits(char x) {   // src/myers_align.h:40-67
  switch (x & ~32) {
    case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8;
    case 'S': return 6; case 'W': return 9; case 'R': return 5; case 'Y': return 10; case 'K': return 12; case 'M': return 3;
    case 'B': return 14; case 'D': return 13; case 'H': return 11; case 'V': return 7; case 'N': return 15;
    default: return 0;
  }
}

struct MyersPair {
  const char* a;
  const char* b;
  int32_t la, lb, mode, maxd;
};

// LDS: peq[16][nblk] u64  (bit i of peq[t][blk] = seq_a[blk*64+i] is compatible with a seq_b symbol of bitmap t)
__global__ __launch_bounds__(64) void k_myers(const MyersPair* pairs, int32_t n_pairs, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  uint64_t* peq = reinterpret_cast<uint64_t*>(lds_raw);
  const int lane = threadIdx.x;
  for (int p = blockIdx.x; p < n_pairs; p += gridDim.x) {
    const MyersPair pr = pairs[p];
    const int m = pr.la, n = pr.lb;
    int maxd = pr.maxd;
    if (maxd > m + n) maxd = m + n;                                 // :13
    uint32_t result = 0xFFFFFFFFu;
    if (maxd <= 0) { if (lane == 0) out[p] = result; continue; }
    if (m == 0 || n == 0) {                                         // degenerate: the walk only takes gaps
      int d = (pr.mode == 0) ? m + n : (pr.mode == 1 ? n : m);      // mode 1 stops when x == len_b, mode 2 when y == len_a
      if (m == 0 && pr.mode == 2) d = 0;
      if (n == 0 && pr.mode == 1) d = 0;
      if (lane == 0) out[p] = d < maxd ? (uint32_t)d : 0xFFFFFFFFu;
      continue;
    }
    const int nblk = (m + 63) / 64, K = (nblk + 63) / 64;
    // ---- Peq table
    for (int e = lane; e < 16 * nblk; e += 64) peq[e] = 0;
    __syncthreads();
    for (int blk = lane; blk < nblk; blk += 64) {
      uint64_t w[16];
      for (int t = 0; t < 16; t++) w[t] = 0;
      for (int i = 0; i < 64 && blk * 64 + i < m; i++) {
        const uint32_t ab = iupac_bits(pr.a[blk * 64 + i]);
        for (int t = 1; t < 16; t++) if (ab & t) w[t] |= 1ull << i;
      }
      for (int t = 0; t < 16; t++) peq[t * nblk + blk] = w[t];
    }
    __syncthreads();
    // ---- systolic sweep
    uint64_t Pv[MYERS_MAX_K], Mv[MYERS_MAX_K];
    for (int k = 0; k < MYERS_MAX_K; k++) { Pv[k] = ~0ull; Mv[k] = 0; }
    const int last_blk = nblk - 1, last_lane = last_blk / K, last_k = last_blk % K, last_bit = (m - 1) & 63;
    int score = m, best_row = m;          // D[m][j] (lane last_lane); best over j for mode 2 (j = 0 counts: D[m][0] = m)
    int carry = 0;                        // hout of this lane's last block at the previous step
    for (int t = 0; t < n + 64; t++) {
      const int hin_left = __builtin_amdgcn_update_dpp(1, carry, DPP_WAVE_SHR1, 0xF, 0xF, false);   // lane 0: D[0][j]-D[0][j-1] = +1
      const int j = t - lane;
      const bool act = j >= 0 && j < n;
      int hin = hin_left, hout = 0;
      if (act) {
        const uint32_t tb = iupac_bits(pr.b[j]);
#pragma unroll
        for (int k = 0; k < MYERS_MAX_K; k++) {
          const int blk = lane * K + k;
          if (k < K && blk < nblk) {
            uint64_t Eq = peq[tb * nblk + blk];
            const uint64_t pv = Pv[k], mv = Mv[k];
            const uint64_t neg = hin < 0 ? 1ull : 0ull, pos = hin > 0 ? 1ull : 0ull;
            const uint64_t Xv = Eq | mv;
            Eq |= neg;
            const uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            uint64_t Ph = mv | ~(Xh | pv), Mh = pv & Xh;
            if (blk == last_blk) score += (int)((Ph >> last_bit) & 1) - (int)((Mh >> last_bit) & 1);
            hout = (int)(Ph >> 63) - (int)(Mh >> 63);
            Ph = (Ph << 1) | pos;
            Mh = (Mh << 1) | neg;
            Pv[k] = Mh | ~(Xv | Ph);
            Mv[k] = Ph & Xv;
            hin = hout;
          }
        }
        if (lane == last_lane && score < best_row) best_row = score;
      }
      carry = act ? hout : 0;
    }
    (void)last_k;
    // ---- result
    const int d_global = __shfl(score, last_lane), d_mode2 = __shfl(best_row, last_lane);
    int d;
    if (pr.mode == 0) d = d_global;
    else if (pr.mode == 2) d = d_mode2;
    else {
      // mode 1: min_i D[i][n], D[0][n] = n, D[i][n] = n + sum of vertical deltas of rows 1..i
      int tot = 0, mn = 0;                       // this lane's blocks: total delta and minimum running prefix
      for (int k = 0; k < MYERS_MAX_K; k++) {
        const int blk = lane * K + k;
        if (k < K && blk < nblk)
          for (int i = 0; i < 64 && blk * 64 + i < m; i++) {
            tot += (int)((Pv[k] >> i) & 1) - (int)((Mv[k] >> i) & 1);
            if (tot < mn) mn = tot;
          }
      }
      // exclusive prefix of lane totals, then global minimum
      int incl = tot;
      for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
      int cand = (incl - tot) + mn;
      for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(cand, o); if (v < cand) cand = v; }
      d = n + (cand < 0 ? cand : 0);
    }
    if (lane == 0) out[p] = d < maxd ? (uint32_t)d : 0xFFFFFFFFu;
    __syncthreads();
  }
}

}  // namespace mia
