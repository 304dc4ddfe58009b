// mia_pass1_kernels.h -- pass 1 on gfx950: the body of the read loop of main()
// (/root/reference/src/mia_main.c:759-805): new_kmer_filter (src/kmer.c:239-331)
// followed by sg_align (src/mia.c:1500-1665), one read per wavefront, persistent grid.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bandx_body.h"
#include "diag_filter.h"
#include "mia_layout.h"
#include "pass1_body.h"
#include "wave_dev.h"

namespace mia {

constexpr int MAX_KMER_POS = 128;   // src/params.h:75
constexpr int KMER_SATURATE = 128;  // src/params.h:77
constexpr int MASK_BUFFER = 10;     // src/params.h:78

constexpr uint8_t P1_PASSED = 1, P1_KEPT = 2, P1_STRAND_KNOWN = 4, P1_SPLIT = 8;

struct Pass1Reads {
  int64_t n;
  const uint8_t* packed;   // 4-bit codes, reads as sequenced
  const uint32_t* roff;
  const uint16_t* len;
  int32_t* score;
  int32_t* as;
  int32_t* ae;
  uint8_t* rc;
  uint8_t* flags;
  uint32_t* status;
};

struct KmerIndex {          // dense table over the 4^k k-mers of each strand of the wrapped reference
  const uint32_t* tab[2];   // [0] forward, [1] reverse complement: (first index into pos << 8) | number of positions (<= 128)
  const int32_t* pos[2];    // positions, ascending per k-mer
  int32_t k;                // < 0: no filter
};

// the table is zeroed with a memset; only the k-mers that occur in the reference are written (at most one per position)
__global__ void k_kmer_fill(int32_t n, const uint32_t* kmer, const uint32_t* entry, uint32_t* table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) table[kmer[i]] = entry[i];
}

__device__ __forceinline__ void mask_or_range(uint32_t* m, int lo, int hi) {   // set bits lo..hi (inclusive) of an LDS bit mask
  for (int w = lo >> 5; w <= (hi >> 5); w++) {
    const int a = (w == (lo >> 5)) ? (lo & 31) : 0, b = (w == (hi >> 5)) ? (hi & 31) : 31;
    const uint32_t bits = (b == 31 ? 0xFFFFFFFFu : ((1u << (b + 1)) - 1u)) & ~((1u << a) - 1u);
    atomicOr(&m[w], bits);
  }
}

// what sg_align leaves in the FragSeq for the winning strand's alignment (src/mia.c:1568-1610,1614,1619,1653)
__device__ __forceinline__ void pass1_store(const Pass1Reads& rs, int64_t i, int32_t L, int strand, int32_t score, int32_t abc, int32_t aec,
                                            uint32_t status) {
  int start, end, as, ae;
  if (strand) {
    start = L - (aec % L) - 1;   // c2rcc (src/mia.c:26-30)
    end = L - (abc % L) - 1;
  } else { start = abc; end = aec; }
  as = start; ae = end;
  if (as > ae) ae = L + as;        // sic
  if (end > L) end -= L;
  uint8_t fl = P1_PASSED;
  if (score >= 2000) fl |= P1_KEPT;
  if (score > 2000) fl |= P1_STRAND_KNOWN;
  if (start > end) fl |= P1_SPLIT;
  rs.flags[i] = fl; rs.score[i] = score; rs.as[i] = as; rs.ae[i] = ae; rs.rc[i] = (uint8_t)strand; rs.status[i] = status;
}

// ---- the diagonal filter in front of the whole-reference DP (diag_filter.h; flat matrix, no k-mer mask): one read per
// thread against every diagonal of both strands.  Reads it cannot decide are collected in `todo` for k_pass1.
__global__ __launch_bounds__(256) void k_pass1_filter(Pass1Reads rs, RefPlanes fw, RefPlanes rc, KmerOcc kf, KmerOcc kr, int32_t len1, int32_t L,
                                                      int32_t* todo, uint32_t* n_todo) {
  __shared__ int16_t cand[256];
  __shared__ int32_t cand_delta[256];
  __shared__ uint8_t cand_strand[256];
  __shared__ int16_t left[256];
  __shared__ int32_t n_cand, n_left;
  __shared__ uint32_t base;
  if (threadIdx.x == 0) { n_cand = 0; n_left = 0; }
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * 256;
  {
    const int64_t i = i0 + threadIdx.x;
    if (i < rs.n) {
      const int len2 = rs.len[i];
      int strand = 0, delta = 0;
      const int k = pass1_step1(fw, rc, kf, kr, len1, rs.packed + rs.roff[i], len2, &strand, &delta);
      if (k == 2) {
        const int slot = atomicAdd(&n_cand, 1);
        cand[slot] = (int16_t)threadIdx.x; cand_delta[slot] = delta; cand_strand[slot] = (uint8_t)strand;
      } else if (k >= 0) {
        pass1_store(rs, i, L, strand, FLAT_MATCH * len2 - (FLAT_MATCH - FLAT_MISMATCH) * k, delta, delta + len2 - 1, ST_DIAG);
      } else {
        left[atomicAdd(&n_left, 1)] = (int16_t)threadIdx.x;
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n_cand) {               // rule (c) for the reads with two mismatches, packed into the first threads
    const int t = cand[threadIdx.x], strand = cand_strand[threadIdx.x];
    const int64_t i = i0 + t;
    const int len2 = rs.len[i], delta = cand_delta[threadIdx.x];
    if (pass1_step2(fw, rc, kf, kr, len1, rs.packed + rs.roff[i], len2))
      pass1_store(rs, i, L, strand, FLAT_MATCH * len2 - (FLAT_MATCH - FLAT_MISMATCH) * 2, delta, delta + len2 - 1, ST_DIAG);
    else
      left[atomicAdd(&n_left, 1)] = (int16_t)t;
  }
  __syncthreads();
  if (threadIdx.x == 0 && n_left) base = atomicAdd(n_todo, (uint32_t)n_left);
  __syncthreads();
  if ((int)threadIdx.x < n_left) todo[base + threadIdx.x] = (int32_t)(i0 + left[threadIdx.x]);
}

// ---- anchored pass 1 for the reads the filter left over (flat matrix, both strands free of N, no k-mer mask) -----------
// Pigeonhole again, now for ALIGNMENTS: a path through the DP matrix that loses no more than B against 200 x len has at
// most B/800 defects (a mismatch costs 800; a gap at least 1200 and 200 + 200 per skipped row), so if the read is cut into
// six 10-mers and B < 4600 at least one 10-mer is crossed without any defect -- the path runs through a place where that
// 10-mer occurs in the reference, and the table lists all of those ("anchors").  From an anchor on diagonal d the path
// cannot stray further than (B - 1000) / 200 < 18 diagonals.  So every alignment that can compete with one of score
// 200 len - B lies inside a +-50 window around a cluster of anchors, and the best over those few windows IS the best
// over the whole strand: the windowed kernels (mia_hip_align_windows' pipeline) do in ~10 ns what k_pass1 does in 900.
// k_pass1_anchor finds at most two clusters per read and writes their windows (absolute positions in the string
// [forward strand | reverse strand]); k_pass1_select checks the budget on the result, applies max_sg_score's first
// maximum and sg_align's strand rule (src/mia.c:1549) and hands everything it cannot vouch for to k_pass1.
constexpr int P1A_SLOTS = 2;       // windows per read that are aligned
constexpr int P1A_CLUSTERS = 24;   // clusters of anchors a read may have (most of them one stray 10-mer)
constexpr int P1A_BLOCKS = 9;      // 10-mers cut out of the read (fewer for reads under 90 bases, at least 6)
constexpr int P1A_JOIN = 20;       // anchors this close (in diagonals) share a window
constexpr int P1A_MARGIN = 50;     // columns around a cluster, as reiterate_assembly's REALIGN_BUFFER
constexpr int P1A_APART = 8;       // weak clusters at least this many diagonals apart cannot both be used by one path
// With nb blocks the budget is 800 nb - 400: fewer than nb defects, and 200 in hand because the first column of a window
// may give a late start its substitution score back (src/mia.c: column 0 against the "new start" branch elsewhere).  A path
// that loses no more strays at most (800*9 - 400 - 1000) / 200 = 29 diagonals from its anchor: inside the margin.
// Clusters in which a single block occurs are not aligned: a path through them alone keeps at most as many blocks intact
// as occur in such clusters, which bounds it by 200 len - 800 (nb - that many) + 200; k_pass1_select wants more (w_bound).

//
// GEN (any matrix the band pipeline has tables for -- bandx_body.h: identity is the best base at every depth): the same
// argument in LOSSES against U = sum of the rows' best scores (sg_align scores both strands with the forward matrix,
// src/mia.c:1500-1560, so strand 0's tables).  Breaking block b costs at least dl_b (BxTab::dl: the cheapest other base
// over its rows, never more than a gap's GOP + GEP or a run of skipped rows' cost per touched block; capped where a clean
// crossing could hold more N columns than the table spells out), so a path that loses less than the sum of dl over the
// usable blocks crosses one of them cleanly.  The window quirk's slack (one substitution score) is max M instead of 200:
//   budget = sum dl - 2 max M        bound = U - (sum dl - the `kept` largest dl) + max M
// and a path within the budget strays at most (9 x 1200 - 1000) / 200 = 49 columns from its anchor: inside the margin.
// Blocks are cut as the band pipeline cuts them (bx_blocks_of / bx_block_row: twelve for reads beyond 128 bases), so that
// dl and the stray tables apply as they stand.  Two weak clusters P1A_APART diagonals apart count as one kept block only if
// straying that far costs at least the dearest block (BxTab's stray tables: net of the blocks the gaps break themselves).
template <bool GEN>
__global__ __launch_bounds__(256) void k_pass1_anchor(Pass1Reads rs, const int32_t* todo, int64_t n_todo, KmerOcc kf, KmerOcc kr, int32_t len1,
                                                      uint32_t* w_roff, uint16_t* w_len, uint8_t* w_sk, int32_t* w_as, int32_t* w_ae,
                                                      int32_t* w_bound, int32_t* w_budget, int32_t* w_u, BxTab tab) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_todo) return;
  const int64_t i = todo[t];
  const int len2 = rs.len[i];
  const uint8_t* rp = rs.packed + rs.roff[i];
  const int nb_cut = GEN ? bx_blocks_of(len2) : (len2 / DF_K < P1A_BLOCKS ? len2 / DF_K : P1A_BLOCKS);
  int n_cl = 0, c_strand[P1A_CLUSTERS], c_lo[P1A_CLUSTERS], c_hi[P1A_CLUSTERS], c_mask[P1A_CLUSTERS];
  bool usable = nb_cut >= 6;
  int why = usable ? 0 : 2;          // (statistics: 1 a read with N, 2 too few blocks, 3 too many clusters, 4 a cluster too wide, 5 too many strong ones, 6 none, 7 no room)
  for (int r = 0; usable && r < len2; r++) if (((rp[r >> 1] >> ((r & 1) * 4)) & 15) > 3) { usable = false; why = 1; }     // a read with N
  // A block whose 10-mer has more occurrences (on either strand) than the table keeps cannot be used -- the pigeonhole
  // then runs over the nb blocks that can: still "fewer than nb defects leave one of them intact".
  int blocks = 0;
  if (usable)
    for (int b = 0; b < nb_cut; b++) {
      const int o = (int)((int64_t)b * (len2 - DF_K) / (nb_cut - 1));
      int64_t idx = 0;
      for (int q = 0; q < DF_K; q++) { const int r = o + q; idx |= (int64_t)((rp[r >> 1] >> ((r & 1) * 4)) & 3) << (2 * q); }
      if (kf.cnt[idx] <= DF_KCAP && kr.cnt[idx] <= DF_KCAP) blocks |= 1 << b;
    }
  const int nb = __popc((unsigned)blocks);
  if (usable && nb < 6) { usable = false; why = 2; }
  for (int st = 0; usable && st < 2; st++) {
    const KmerOcc& ko = st ? kr : kf;
    for (int b = 0; usable && b < nb_cut; b++) {
      if (!((blocks >> b) & 1)) continue;
      const int o = (int)((int64_t)b * (len2 - DF_K) / (nb_cut - 1));
      int64_t idx = 0;
      for (int q = 0; q < DF_K; q++) { const int r = o + q; idx |= (int64_t)((rp[r >> 1] >> ((r & 1) * 4)) & 3) << (2 * q); }
      const int n = ko.cnt[idx];
      for (int k = 0; k < n; k++) {
        const int d = ko.pos[idx * DF_KCAP + k] - o;          // diagonal of this anchor (may hang over either end of the strand)
        int hit = -1;
        for (int c = 0; c < n_cl; c++) if (c_strand[c] == st && d >= c_lo[c] - P1A_JOIN && d <= c_hi[c] + P1A_JOIN) hit = c;
        if (hit >= 0) {
          if (d < c_lo[hit]) c_lo[hit] = d;
          if (d > c_hi[hit]) c_hi[hit] = d;
          c_mask[hit] |= 1 << b;
          if (c_hi[hit] - c_lo[hit] > 100) { usable = false; why = 4; }
        } else if (n_cl < P1A_CLUSTERS) { c_strand[n_cl] = st; c_lo[n_cl] = d; c_hi[n_cl] = d; c_mask[n_cl] = 1 << b; n_cl++; }
        else { usable = false; why = 3; }
      }
    }
  }
  // (two clusters that grew towards each other stay two overlapping windows: harmless)
  int n_slot = 0, weak = 0;
  int s_strand[P1A_SLOTS] = {0, 0}, s_lo[P1A_SLOTS] = {0, 0}, s_hi[P1A_SLOTS] = {0, 0};
  for (int c = 0; usable && c < n_cl; c++) {
    if (__popc((unsigned)c_mask[c]) >= 2) {
      if (n_slot < P1A_SLOTS) { s_strand[n_slot] = c_strand[c]; s_lo[n_slot] = c_lo[c]; s_hi[n_slot] = c_hi[c]; n_slot++; }
      else { usable = false; why = 5; }
    } else weak |= c_mask[c];
  }
  // A path that runs through no aligned cluster keeps intact only blocks that occur in the weak clusters: at most
  // popc(weak) of the nb, every other block costs it 800.  To keep the blocks of TWO weak clusters it must change diagonal
  // between them, and where they lie P1A_APART or more diagonals apart that costs more than the second block saves: an
  // event pays for the blocks it touches itself (a column gap of g: 1000 + 200 g, one block at most; n skipped rows:
  // 1000 + 400 n, at most ceil((n-1)/10) + 1 blocks) and has at least 100 per diagonal to spare -- 800 over eight
  // diagonals, the price of a block.  So with all weak clusters that far apart one block is all such a path keeps.
  // (A table with the N columns' spellings hands every read two or three stray 10-mers somewhere on the two strands.)
  bool apart = true;
  for (int c = 0; c < n_cl; c++)
    for (int e = c + 1; e < n_cl; e++)
      if (__popc((unsigned)c_mask[c]) < 2 && __popc((unsigned)c_mask[e]) < 2 && c_strand[c] == c_strand[e] &&
          c_lo[e] - c_hi[c] < P1A_APART && c_lo[c] - c_hi[e] < P1A_APART) apart = false;
  int kept = apart ? 1 : __popc((unsigned)weak);
  int bound = weak ? FLAT_MATCH * len2 - (FLAT_MATCH - FLAT_MISMATCH) * (nb - kept) + FLAT_MATCH : INT32_MIN;
  int budget = (FLAT_MATCH - FLAT_MISMATCH) * nb - 2 * FLAT_MATCH, u_all = FLAT_MATCH * len2;
  if (GEN) {
    u_all = 0;
    for (int r = 0; r < len2; r++) u_all += tab.mrow[sm_depth(r, len2) * 4 + (int)((rp[r >> 1] >> ((r & 1) * 4)) & 3)];     // (a read with N is not usable anyway)
    const int16_t* dlb = tab.dl + (int64_t)len2 * BX_BLOCKS;                  // strand 0
    int sum = 0, dmax = 0, top[BX_BLOCKS];
    for (int b = 0; b < BX_BLOCKS; b++) top[b] = 0;
    for (int b = 0; b < nb_cut; b++) {
      if (!((blocks >> b) & 1)) continue;
      const int v = dlb[b];
      sum += v;
      if (v > dmax) dmax = v;
      // (insertion into the descending list of the blocks' prices)
      int x = v;
      for (int q = 0; q < BX_BLOCKS; q++) if (x > top[q]) { const int y = top[q]; top[q] = x; x = y; }
    }
    const int16_t* dn = tab.dl + bx_stray_off(0, len2, 0);
    const int16_t* up = tab.dl + bx_stray_off(0, len2, 1);
    const int stray = dn[P1A_APART] < up[P1A_APART] ? dn[P1A_APART] : up[P1A_APART];
    kept = (apart && stray >= dmax) ? 1 : __popc((unsigned)weak);
    int keep_sum = 0;
    for (int q = 0; q < kept && q < BX_BLOCKS; q++) keep_sum += top[q];
    budget = sum - 2 * tab.max_m;
    bound = weak ? u_all - (sum - keep_sum) + tab.max_m : INT32_MIN;
    if (usable && (budget <= 0 || dmax <= 0)) { usable = false; why = 2; }
  }
  if (usable && n_slot == 0) { usable = false; why = 6; }                           // nothing strong to align: nothing bounds the optimum from below
  for (int c = 0; c < P1A_SLOTS; c++) {
    const int64_t slot = t * P1A_SLOTS + c;
    w_roff[slot] = rs.roff[i];
    w_len[slot] = (uint16_t)len2;
    int ws = 0, we = 0;
    if (usable && c < n_slot) {
      ws = s_lo[c] - P1A_MARGIN; if (ws < 0) ws = 0;
      we = s_hi[c] + len2 + P1A_MARGIN; if (we > len1) we = len1;     // exclusive
      if (we - ws < len2) { usable = false; why = 7; }               // the read does not fit: leave it to k_pass1
    }
    w_as[slot] = (c < n_slot && s_strand[c] ? len1 : 0) + ws;        // the reverse strand follows the forward one
    w_ae[slot] = (c < n_slot && s_strand[c] ? len1 : 0) + we - 1;
  }
  for (int c = 0; c < P1A_SLOTS; c++) w_sk[t * P1A_SLOTS + c] = (uint8_t)(usable && c < n_slot);
  w_bound[t] = bound;
  w_budget[t] = usable ? budget : -why;
  w_u[t] = u_all;
}

// after the windowed alignment of the slots: the read's result, or its place in the list of k_pass1
__global__ __launch_bounds__(256) void k_pass1_select(Pass1Reads rs, const int32_t* todo, int64_t n_todo, int32_t len1, int32_t L, const uint8_t* w_sk,
                                                      const int32_t* w_score, const int32_t* w_as, const int32_t* w_ae, const int16_t* w_abr,
                                                      const uint32_t* w_status, const int32_t* w_bound, const int32_t* w_budget, const int32_t* w_u,
                                                      int32_t* rest, uint32_t* n_rest) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_todo) return;
  const int64_t i = todo[t];
  int best[2] = {INT32_MIN, INT32_MIN}, abc[2] = {0, 0}, aec[2] = {0, 0};
  bool good = false, bad = false, clipped = false;
  for (int c = 0; c < P1A_SLOTS; c++) {
    const int64_t slot = t * P1A_SLOTS + c;
    if (!w_sk[slot]) continue;
    good = true;
    const uint32_t stt = w_status[slot];
    if (stt & (ST_TOO_LONG | ST_ESCAPE | ST_BAND)) bad = true;
    if (w_abr[slot] != 0) clipped = true;                                              // (a clipped start is left to the whole-strand DP)
    const int st = w_as[slot] >= len1 ? 1 : 0, base = st ? len1 : 0;
    const int sc = w_score[slot], a = w_as[slot] - base, e = w_ae[slot] - base;
    if (sc > best[st] || (sc == best[st] && e < aec[st])) { best[st] = sc; abc[st] = a; aec[st] = e; }     // first maximum of the last row
  }
  const int st = best[0] > best[1] ? 0 : 1;                        // src/mia.c:1549: the reverse strand on a tie
  // within the budget of its blocks, and better than anything the clusters that were not aligned could hold
  if (good && !bad && !clipped && w_u[t] - best[st] <= w_budget[t] && best[st] > w_bound[t]) {      // (w_u: 200 x len with the flat matrix)
    pass1_store(rs, i, L, st, best[st], abc[st], aec[st], ST_OK);
  } else {
    rest[atomicAdd(n_rest, 1u)] = (int32_t)i;
    // why (n_rest[1..5], statistics): no usable cluster; a window the kernels did not finish; a clipped start; over budget; a weak cluster could hold better
    const int why = !good ? 1 : (bad ? 2 : (clipped ? 3 : (w_u[t] - best[st] > w_budget[t] ? 4 : 5)));
    atomicAdd(n_rest + why, 1u);
    if (!good && w_budget[t] < 0 && w_budget[t] >= -7) atomicAdd(n_rest + 8 - w_budget[t], 1u);
  }
}

template <int CPL>
__global__ __launch_bounds__(64, 4) void k_pass1(Pass1Reads rs, const uint8_t* ref_fw, const uint8_t* ref_rc, int32_t len1, int32_t L,
                                               const int32_t* pssm_fwd, PackParams pk, KmerIndex kx, unsigned char* trace_slabs,
                                               int64_t trace_bytes, uint32_t* ckpt_slabs, int64_t ckpt_words, int32_t rows_p,
                                               int32_t mask_words, int32_t plain, const int32_t* todo, int64_t n_todo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  DevWave wave(lds_raw, trace_slabs + (int64_t)blockIdx.x * trace_bytes);
  const int lane = (int)wave.lane();
  Pass1Args a;
  a.ref_codes[0] = ref_fw;
  a.ref_codes[1] = ref_rc;
  a.len1 = len1;
  a.pssm = pssm_fwd;
  a.pk = pk;
  a.lds_sub = 0;
  a.lds_carry = MAX_READ * 10;
  a.lds_mask[0] = a.lds_carry + 5 * MAX_READ * 4;
  a.lds_mask[1] = a.lds_mask[0] + (uint32_t)mask_words * 4;
  a.masked = kx.k > 0;
  a.plain = (CPL == P1_CPL_WIDE && plain && !a.masked) ? 1 : 0;
  a.ckpt = ckpt_slabs + (int64_t)blockIdx.x * ckpt_words;
  a.rows_p = rows_p;
  uint32_t* mask[2] = {reinterpret_cast<uint32_t*>(lds_raw + a.lds_mask[0]), reinterpret_cast<uint32_t*>(lds_raw + a.lds_mask[1])};

  const int64_t n_work = todo ? n_todo : rs.n;     // todo: the reads the diagonal filter left over
  for (int64_t w = blockIdx.x; w < n_work; w += gridDim.x) {
    const int64_t i = todo ? (int64_t)todo[w] : w;
    const int len2 = rs.len[i];
    const uint8_t* rp = rs.packed + rs.roff[i];
    bool pass = true;
    if (kx.k > 0) {
      // ---- new_kmer_filter (src/kmer.c:239-331)
      for (int w = lane; w < 2 * mask_words; w += 64) mask[0][w] = 0;
      wave.lds_fence();
      int nf = 0, nr = 0;
      for (int fp0 = 0; fp0 + kx.k <= len2; fp0 += 64) {
        const int fp = fp0 + lane;
        bool ok = fp + kx.k <= len2;
        uint32_t inx = 0;
        if (ok)
          for (int t = 0; t < kx.k; t++) {
            const int c = (rp[(fp + t) >> 1] >> (((fp + t) & 1) * 4)) & 15;
            if (c > 3) { ok = false; break; }          // kmer2inx: anything but ACGT invalidates the k-mer (src/kmer.c:28-44)
            inx = (inx << 2) | (uint32_t)c;
          }
        int cf = 0, cr = 0;
        if (ok) {
          const uint32_t ef = kx.tab[0][inx], er = kx.tab[1][inx];
          cf = (int)(ef & 255u); cr = (int)(er & 255u);
          const int f0 = (int)(ef >> 8), f1 = f0 + cf, r0 = (int)(er >> 8), r1 = r0 + cr;
          for (int t = f0; t < f1; t++) {              // src/kmer.c:287-298
            const int p = kx.pos[0][t];
            int lo = p - fp - MASK_BUFFER, hi = p + (len2 - fp) + MASK_BUFFER;
            if (lo < 0) lo = 0;
            if (hi >= len1) hi = len1 - 1;
            if (hi >= lo) mask_or_range(mask[0], lo, hi);
          }
          for (int t = r0; t < r1; t++) {              // src/kmer.c:311-323 (one column less than forward)
            const int p = kx.pos[1][t];
            int lo = p - fp - MASK_BUFFER, hi = p + len2 - fp - 1 + MASK_BUFFER;
            if (lo < 0) lo = 0;
            if (hi >= len1) hi = len1 - 1;
            if (hi >= lo) mask_or_range(mask[1], lo, hi);
          }
        }
        for (int o = 32; o > 0; o >>= 1) { cf += __shfl_xor(cf, o); cr += __shfl_xor(cr, o); }
        nf += cf; nr += cr;
      }
      wave.lds_fence();
      // >= 128 hits on a strand unmask it completely; the running count only grows, so the final count decides (src/kmer.c:283-285)
      if (nf >= KMER_SATURATE) for (int w = lane; w < mask_words; w += 64) mask[0][w] = 0xFFFFFFFFu;
      if (nr >= KMER_SATURATE) for (int w = lane; w < mask_words; w += 64) mask[1][w] = 0xFFFFFFFFu;
      wave.lds_fence();
      pass = (len2 >= kx.k) && (nf + nr) > 0;
    }
    if (!pass) {
      if (lane == 0) { rs.flags[i] = 0; rs.score[i] = 0; rs.as[i] = 0; rs.ae[i] = 0; rs.rc[i] = 0; rs.status[i] = ST_SKIPPED; }
      continue;
    }
    a.read_packed = rp;
    a.len2 = len2;
    Pass1Result r;
    if (CPL == P1_CPL_WIDE && a.plain) r = Pass1Aligner<DevWave, CPL>::run_plain(wave, a);
    else r = Pass1Aligner<DevWave, CPL>::run(wave, a);
    if (lane == 0) pass1_store(rs, i, L, r.strand, r.score, r.abc, r.aec, r.status);
    wave.lds_fence();
  }
}

}  // namespace mia
