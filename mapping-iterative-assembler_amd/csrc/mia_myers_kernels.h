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
#define MIA_HD __host__ __device__
#include "myers_ond_body.h"

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
  int32_t cap;                 // k_myers_ond: rows of D-paths it may walk for this pair (0: the pair is not its business)
  uint32_t a_off, b_off;       // ... and where its sequences lie in `codes` as 4-bit bitmaps, eight to a word (the host packs them)
  int32_t pad;
};

constexpr uint32_t MYERS_BEYOND_CAP = 0xFFFFFFFEu;   // k_myers_ond's answer for "not within my cap, and the cap is below maxd": k_myers takes the pair
constexpr int MYERS_OND_THREADS = 256;
constexpr int MYERS_OND_MAX_CAP = 4096;
constexpr int MYERS_OND_LDS_BUDGET = 150 * 1024;

// LDS bytes k_myers_ond needs for a pair: both sequences packed (eight characters a word) with their padding, two rows of
// 2 cap + 3 cells
__host__ __device__ inline size_t myers_ond_lds(int la, int lb, int cap) {
  return 4 * ((size_t)(la + 7) / 8 + (size_t)(lb + 7) / 8 + 2 * OND_PAD_WORDS) + 2 * 4 * (2 * (size_t)cap + 3);
}

// ---- long pairs at a small distance: furthest-reaching D-paths, one row of diagonals per step (myers_ond_body.h) -------------
// One pair per workgroup; thread t owns the diagonals -d + t, -d + t + 256, ... of row d.  Both sequences sit in LDS as
// 4-bit bitmaps (packed by the host); the row before and the row being written are two LDS arrays (swapped every step), and when the caller
// wants the alignment every row also goes to `table` (cell (d, k) at d*d + k + d) for the walk back on the host.  One
// barrier per row, waiting for LDS only: the table's stores drain behind the kernel's back.  The first diagonal in
// ascending order that has arrived ends the search, as the reference's loop does (src/myers_align.c:24,39-40).
__global__ __launch_bounds__(MYERS_OND_THREADS) void k_myers_ond(const MyersPair* pairs, const uint32_t* codes, int32_t n_pairs, uint32_t* out, int32_t* table, int32_t* end_k_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int found_k;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int p = blockIdx.x; p < n_pairs; p += gridDim.x) {
    const MyersPair pr = pairs[p];
    const int la = pr.la, lb = pr.lb, mode = pr.mode;
    int maxd = pr.maxd;
    if (maxd > la + lb) maxd = la + lb;                              // src/myers_align.c:13
    // a pair without a cap was not packed for this kernel (a_off = b_off = 0, no LDS counted for it): never touch its codes.
    // maxd <= 0 admits no distance at all (d < maxd, src/myers_align.c:20): the answer is "none" here and now
    if (pr.cap <= 0) { if (tid == 0) out[p] = maxd > 0 ? MYERS_BEYOND_CAP : 0xFFFFFFFFu; continue; }
    const int cap = pr.cap < maxd ? pr.cap : maxd;
    const int wa = (la + 7) / 8 + OND_PAD_WORDS, wb = (lb + 7) / 8 + OND_PAD_WORDS;
    uint32_t* A = reinterpret_cast<uint32_t*>(lds_raw);
    uint32_t* B = A + wa;
    int32_t* row0 = reinterpret_cast<int32_t*>(B + wb);
    int32_t* row1 = row0 + (2 * cap + 3);
    if (tid == 0) found_k = INT32_MAX;
    __syncthreads();
    for (int side = 0; side < 2; side++) {
      const uint32_t* src = codes + (side ? pr.b_off : pr.a_off);
      const int full = ((side ? lb : la) + 7) / 8, words = side ? wb : wa;
      uint32_t* dst = side ? B : A;
      for (int w = tid; w < words; w += MYERS_OND_THREADS) dst[w] = w < full ? src[w] : 0u;      // (rows beyond the end are packed as 0 by the host)
    }
    __syncthreads();
    int dist = -1;
    for (int d = 0; d < cap; d++) {
      int32_t* cur = (d & 1) ? row1 : row0;
      const int32_t* prow = (d & 1) ? row0 : row1;
      const int klo = -d > -la ? -d : -la, khi = d < lb ? d : lb;
      auto prev = [&](int kk) -> int32_t { return (kk < -(d - 1) || kk > d - 1) ? OND_NONE : prow[kk + cap + 1]; };
      for (int k0 = -d; k0 <= d; k0 += MYERS_OND_THREADS) {          // (every lane of a wavefront goes round together: the shared snakes below)
        const int k = k0 + tid;
        const bool cell = k <= d;
        int32_t x = OND_NONE;
        bool more = false;
        if (cell && k >= klo && k <= khi) {
          x = ond_cell(d, k, prev);
          // the first eight characters on its own: on all but a few diagonals of a row the snake ends there
          if (x != OND_NONE && x >= 0 && x - k >= 0) {
            const int r = ond_shared_chunk(A, B, x - k, x, la, lb);
            x += r;
            more = r == 8;
          }
        }
        // ... and the few that go on, one after the other, by the whole wavefront: lane l the eight characters from 8 l on
        unsigned long long todo = __ballot(more);
        while (todo) {
          const int src = __builtin_ctzll(todo);
          todo &= todo - 1;
          const int xs = __builtin_amdgcn_readlane(x, src), ks = __builtin_amdgcn_readlane(k, src);
          int total = 0;
          for (int off = 0;; off += OND_SHARED_SPAN) {
            const int xx = xs + off + 8 * lane;
            const int c = ond_shared_chunk(A, B, xx - ks, xx, la, lb);
            const unsigned long long stop = __ballot(c < 8);
            if (stop) { const int f = __builtin_ctzll(stop); total = off + 8 * f + __builtin_amdgcn_readlane(c, f); break; }
          }
          if (lane == src) x += total;
        }
        if (cell) {
          if (x != OND_NONE && k >= klo && k <= khi && ond_arrived(mode, x, k, la, lb)) atomicMin(&found_k, k);
          cur[k + cap + 1] = x;
          if (table) table[ond_at(d, k)] = x;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS only (see above)
      if (found_k != INT32_MAX) { dist = d; break; }
    }
    if (tid == 0) {
      out[p] = dist >= 0 ? (uint32_t)dist : (cap < maxd ? MYERS_BEYOND_CAP : 0xFFFFFFFFu);
      if (end_k_out) end_k_out[p] = dist >= 0 ? found_k : 0;
    }
    __syncthreads();
  }
}

// LDS: peq[16][nblk] u64  (bit i of peq[t][blk] = seq_a[blk*64+i] is compatible with a seq_b symbol of bitmap t)
__global__ __launch_bounds__(64) void k_myers(const MyersPair* pairs, int32_t n_pairs, uint32_t* out, int only_beyond_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  uint64_t* peq = reinterpret_cast<uint64_t*>(lds_raw);
  const int lane = threadIdx.x;
  for (int p = blockIdx.x; p < n_pairs; p += gridDim.x) {
    if (only_beyond_cap && out[p] != MYERS_BEYOND_CAP) continue;     // (wave-uniform) k_myers_ond has this pair's answer already
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

// ---- short patterns: one pair per LANE --------------------------------------------------------------------------------------
// A pair of reads (100-300 characters: what ccheck's batches and any read-against-read use hold) fills five of the
// systolic kernel's 64 lanes.  Here every lane runs its own pair: seq_a as up to MYERS_LANE_K 64-row blocks in registers,
// one column of the same Myers / Hyyro recurrence per character of seq_b.  The match vector of a column is not looked up
// in a 16-entry table per pair (640 bytes of LDS each) but OR-ed together from four bit planes of seq_a -- "which rows
// hold a code compatible with A / C / G / T" (IUPAC bitmaps, src/myers_align.h:40-67): Eq = the planes of the bases the
// column's character may be.  Both sequences arrive as 4-bit IUPAC bitmaps, eight per word (the host packs them).
constexpr int MYERS_LANE_K = 5;              // 320 rows

struct MyersLanePair {
  uint32_t a_off, b_off;                     // word offsets of the packed sequences
  int32_t la, lb, mode, maxd;
};

__global__ __launch_bounds__(64) void k_myers_lanes(const MyersLanePair* pairs, const uint32_t* codes, int32_t n_pairs, const int32_t* index, uint32_t* out) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  const bool live = p < n_pairs;
  MyersLanePair pr{0, 0, 0, 0, 0, 0};
  if (live) pr = pairs[p];
  const int m = pr.la, n = pr.lb;
  int maxd = pr.maxd;
  if (maxd > m + n) maxd = m + n;                                   // src/myers_align.c:13
  const int nblk = (m + 63) / 64;
  uint64_t P0[MYERS_LANE_K], P1[MYERS_LANE_K], P2[MYERS_LANE_K], P3[MYERS_LANE_K], Pv[MYERS_LANE_K], Mv[MYERS_LANE_K];
#pragma unroll
  for (int k = 0; k < MYERS_LANE_K; k++) { P0[k] = P1[k] = P2[k] = P3[k] = 0; Pv[k] = ~0ull; Mv[k] = 0; }
  // bit planes of seq_a, eight rows per packed word
  const uint32_t* ca = codes + pr.a_off;
  for (int w = 0; w < (m + 7) / 8; w++) {
    const uint32_t v = ca[w];
    uint64_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t c = (v >> (4 * q)) & 15u;                       // rows beyond m are packed as 0: compatible with nothing
      b0 |= (uint64_t)(c & 1u) << q; b1 |= (uint64_t)((c >> 1) & 1u) << q; b2 |= (uint64_t)((c >> 2) & 1u) << q; b3 |= (uint64_t)((c >> 3) & 1u) << q;
    }
    const int blk = w >> 3, sh = (w & 7) * 8;
#pragma unroll
    for (int k = 0; k < MYERS_LANE_K; k++)
      if (k == blk) { P0[k] |= b0 << sh; P1[k] |= b1 << sh; P2[k] |= b2 << sh; P3[k] |= b3 << sh; }
  }
  const int last_blk = nblk - 1, last_bit = (m - 1) & 63;
  int score = m, best_row = m;            // D[m][j]; its minimum over j for mode 2 (j = 0 counts: D[m][0] = m)
  const uint32_t* cb = codes + pr.b_off;
  uint32_t bw = 0;
  int n_max = n;
  for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(n_max, o); n_max = v > n_max ? v : n_max; }
  for (int j = 0; j < n_max; j++) {
    if (j < n && m > 0) {
      if ((j & 7) == 0) bw = cb[j >> 3];
      const uint32_t tb = (bw >> (4 * (j & 7))) & 15u;
      const uint64_t s0 = (tb & 1u) ? ~0ull : 0ull, s1 = (tb & 2u) ? ~0ull : 0ull, s2 = (tb & 4u) ? ~0ull : 0ull, s3 = (tb & 8u) ? ~0ull : 0ull;
      int hin = 1;                                                   // D[0][j] - D[0][j-1]
#pragma unroll
      for (int k = 0; k < MYERS_LANE_K; k++) {
        if (k < nblk) {
          uint64_t Eq = (P0[k] & s0) | (P1[k] & s1) | (P2[k] & s2) | (P3[k] & s3);
          const uint64_t pv = Pv[k], mv = Mv[k];
          const uint64_t neg = hin < 0 ? 1ull : 0ull, pos = hin > 0 ? 1ull : 0ull;
          const uint64_t Xv = Eq | mv;
          Eq |= neg;
          const uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
          uint64_t Ph = mv | ~(Xh | pv), Mh = pv & Xh;
          if (k == last_blk) score += (int)((Ph >> last_bit) & 1) - (int)((Mh >> last_bit) & 1);
          hin = (int)(Ph >> 63) - (int)(Mh >> 63);
          Ph = (Ph << 1) | pos;
          Mh = (Mh << 1) | neg;
          Pv[k] = Mh | ~(Xv | Ph);
          Mv[k] = Ph & Xv;
        }
      }
      if (score < best_row) best_row = score;
    }
  }
  if (!live) return;
  uint32_t result = 0xFFFFFFFFu;
  if (maxd > 0) {
    int d;
    if (m == 0 || n == 0) {                                          // degenerate: the walk only takes gaps (k_myers)
      d = (pr.mode == 0) ? m + n : (pr.mode == 1 ? n : m);
      if (m == 0 && pr.mode == 2) d = 0;
      if (n == 0 && pr.mode == 1) d = 0;
    } else if (pr.mode == 0) d = score;
    else if (pr.mode == 2) d = best_row;
    else {
      // mode 1: min_i D[i][n], D[0][n] = n, D[i][n] = n + the vertical deltas of rows 1 .. i
      int tot = 0, mn = 0;
#pragma unroll
      for (int k = 0; k < MYERS_LANE_K; k++)
        if (k < nblk)
          for (int i = 0; i < 64 && k * 64 + i < m; i++) {
            tot += (int)((Pv[k] >> i) & 1) - (int)((Mv[k] >> i) & 1);
            if (tot < mn) mn = tot;
          }
      d = n + mn;
    }
    if (d < maxd) result = (uint32_t)d;
  }
  out[index ? index[p] : p] = result;        // (index == nullptr: the pairs in the caller's order, mia_hip_myers_packed)
}

}  // namespace mia
