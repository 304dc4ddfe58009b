// mia_kernels.h -- gfx950 kernels of the per-iteration path.  Included once by
// mia_hip.hip.  Device memory layout (all SoA, resident across iterations):
//
//   reads     packed[ ]  4-bit base codes, each read starts on a 4-byte boundary
//             roff[n] (u32 byte offset), len[n] (u16), rc[n], sk[n] (strand_known)
//   per read  as[n], ae[n] (wrapped coords, in/out), score[n], refstart[n], abr[n],
//             status[n], cols[n][stride] (int16 alignment script)
//   reference ref[wrap+pad] one code per byte (0..4), wrap = L + min(L,256) if circular
//   consensus tally[12][L+1] int32 (word-major so that a wave's atomics are 256 B
//             contiguous), gaps[L+1], insert events (u64 list)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "align_body.h"
#include "align_body_quad.h"
#include "align_body_quad_plain.h"
#include "diag_filter.h"
#include "band_body.h"
#include "mia_layout.h"
#include "wave_dev.h"

namespace mia {

constexpr int N_CPL = 3;                 // CPL 4, 8, 12  -> windows of up to 256 / 512 / 768 columns
constexpr int BIN_WIDE = N_CPL;          // exact int32 kernel (whole-reference windows, escapes, overflow)
constexpr int BIN_QUAD0 = BIN_WIDE + 1;  // + read length: windows <= 208 columns, four equally long reads per wavefront
constexpr int N_BINS = BIN_QUAD0 + MAX_READ + 1;
constexpr int SUB_LDS_BYTES = MAX_READ * 10;   // int16 sub[len2][5]

struct ReadSet {
  int64_t n;
  const uint8_t* packed;
  const uint32_t* roff;
  const uint16_t* len;
  const uint8_t* rc;
  const uint8_t* sk;
  int32_t* as;
  int32_t* ae;
  int32_t* score;
  int32_t* refstart;
  int16_t* abr;
  uint32_t* status;
  int16_t* cols;
  int32_t stride;
};

struct RefInfo {
  const uint8_t* codes;
  int32_t L, wrap;
  int32_t explicit_win;   // mia_hip_align_windows: read i is aligned to codes[as[i] .. ae[i]] as it stands
};

// window of reiterate_assembly (src/mia_main.c:191-212)
MIA_HD inline void realign_window(int as, int ae, int len2, int wrap, int* ref_start, int* len1) {
  int rs = (as - REALIGN_BUFFER) < 0 ? 0 : as - REALIGN_BUFFER;
  int re = (ae + REALIGN_BUFFER + 1) > wrap ? wrap : ae + REALIGN_BUFFER;
  if (rs + len2 > re) { rs = 0; re = wrap; }
  *ref_start = rs;
  *len1 = re - rs;
}

// the window of read i: reiterate_assembly's, or the caller's own (src/ccheck.cc:556,589-592: no margin, no fallback)
MIA_HD inline void read_window(const RefInfo& ref, int as, int ae, int len2, int* ref_start, int* len1) {
  if (ref.explicit_win) { *ref_start = as; *len1 = ae - as + 1; return; }
  realign_window(as, ae, len2, ref.wrap, ref_start, len1);
}

struct PackSet { PackParams p[N_CPL]; int ok[N_CPL]; };

MIA_HD inline int classify(int len2, int len1, const PackSet& ps, int use_quad) {
  if (use_quad && len1 <= Q_COLS && ps.ok[0]) return BIN_QUAD0 + len2;
  int ci = len1 <= 256 ? 0 : (len1 <= 512 ? 1 : (len1 <= 768 ? 2 : -1));
  if (ci < 0 || !ps.ok[ci]) return BIN_WIDE;
  return ci;
}

// Histogram update with one LDS atomic per (wavefront, distinct bin): the reads of a wavefront almost always share
// their bin (same kernel class, same read length), so 64 same-address atomics collapse into one.  Returns the rank of
// the lane among the lanes of its wavefront with the same bin plus the old bin count (as atomicAdd would).
__device__ __forceinline__ int hist_add_aggregated(int32_t* hist, int b) {
  int rank = 0;
  unsigned long long todo = __ballot(b >= 0);
  while (todo) {
    const int leader = __builtin_ctzll(todo);
    const int b0 = __shfl(b, leader);
    const unsigned long long same = __ballot(b == b0);
    int base = 0;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&hist[b0], (int)__popcll(same));
    base = __shfl(base, leader);
    if (b == b0) rank = base + (int)__popcll(same & ((1ull << (threadIdx.x & 63)) - 1ull));
    todo &= ~same;
  }
  return rank;
}

// ---- plan: bin every read by kernel variant and LDS footprint ----------------
// Counts go through an LDS histogram per block, so global memory sees one atomic per
// (block, non-empty bin) instead of one per read on a single hot address.
constexpr int PLAN_DEFERRED = -1000;   // bin_of = PLAN_DEFERRED - bin: a quad-class read that waits for the trace stage
constexpr int PLAN_PER = 8;   // reads per thread of the planner kernels: a block bins 2048 reads per LDS histogram

// ---- the diagonal filter (diag_filter.h), ahead of everything else: one read per thread.  A read whose alignment
// is provably one gap-free diagonal is finished here (score, end points, script, ST_DIAG) and marked bin_of = -2 so
// that the planner leaves it out; all others are marked 0.  Flat matrix only (the host checks).
// one plane word per wavefront and step, one code per lane, three ballots (a thread that builds a word alone walks 64
// bytes); any grid: the wavefronts stride over the words
__device__ __forceinline__ void ref_planes_body(const uint8_t* codes, int64_t n_codes, int64_t words, uint64_t* lo, uint64_t* hi, uint64_t* ok) {
  const int lane = threadIdx.x & 63;
  for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < words; w += (int64_t)gridDim.x * 4) {
    const int64_t p = w * 64 + lane - PLANE_LEAD;
    const uint32_t c = (p >= 0 && p < n_codes) ? codes[p] : 4u;
    const bool base = c <= 3u;
    const uint64_t l = __ballot(base && (c & 1u)), h = __ballot(base && (c >> 1)), v = __ballot(base);
    if (lane == 0) { lo[w] = l; hi[w] = h; ok[w] = v; }
  }
}
__global__ __launch_bounds__(256) void k_ref_planes(const uint8_t* codes, int64_t n_codes, int64_t words, uint64_t* lo, uint64_t* hi, uint64_t* ok) {
  ref_planes_body(codes, n_codes, words, lo, hi, ok);
}

// the 10-mer table of the reference for rule (c) (diag_filter.h: KmerOcc); cnt is zeroed before.  wild > 0: a 10-mer with
// up to that many N columns is entered under every spelling (bandx_body.h, N COLUMNS) -- for the anchors of pass 1, not
// for rule (c).
__global__ __launch_bounds__(256) void k_kmer_occ(const uint8_t* codes, int64_t n_codes, int32_t* cnt, int32_t* pos, int32_t wild) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t base;
  uint64_t npos;
  const int k = kmer_wild_at(codes, n_codes, p, &base, &npos);
  if (k < 0 || k > wild) return;
  for (uint32_t x = 0; x < (1u << (2 * k)); x++) {
    const int64_t idx = kmer_wild_key(base, npos, k, x);
    const int c = atomicAdd(&cnt[idx], 1);
    if (c < DF_KCAP) pos[idx * DF_KCAP + c] = (int32_t)p;
  }
}

// reads of a list get a status instead of an alignment (pass 1's windows: the whole-strand DP takes them)
__global__ __launch_bounds__(256) void k_mark_status(const int32_t* list, int32_t count, uint32_t* status, uint32_t flag) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < count) status[list[t]] = flag;
}

__global__ __launch_bounds__(256) void k_iota(int64_t n, int32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_diag_filter(ReadSet rs, RefInfo ref, RefPlanes rp, KmerOcc ko, int64_t n_ref, int32_t* bin_of, uint32_t dbg,
                                                      int32_t* left_list, uint32_t* n_left) {
  __shared__ int16_t verdict[256];      // per read of the block: diagonal of a finished read, -1 otherwise
  __shared__ int16_t cand[256];         // reads with exactly two mismatches on their diagonal: rule (c) is still open
  __shared__ int16_t cand_delta[256];
  __shared__ int32_t n_cand;
  if (threadIdx.x == 0) n_cand = 0;
  verdict[threadIdx.x] = -1;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * 256;
  auto finish = [&](int64_t i, int s, int len2, int delta, int k) {
    rs.score[i] = FLAT_MATCH * len2 - (FLAT_MATCH - FLAT_MISMATCH) * k;
    rs.refstart[i] = s;
    rs.abr[i] = 0;
    rs.as[i] = s + delta;                   // src/mia_main.c:254-255
    rs.ae[i] = s + delta + len2 - 1;
    rs.status[i] = ST_DIAG;
    bin_of[i] = -2;
  };
  {
    const int64_t i = i0 + threadIdx.x;
    if (i < rs.n) {
      int k = -1, delta = 0, s = 0, l1 = 0, best = -1;
      const int len2 = rs.len[i];
      if (rs.sk[i]) {
        read_window(ref, rs.as[i], rs.ae[i], len2, &s, &l1);
        // (step 1 by sliding: over the ~110 diagonals of a window that is cheaper than the table's dependent look-ups,
        // which pay off against the 2 x 16.8 k diagonals of pass 1)
        k = diag_step1(rp, s, l1, rs.packed + rs.roff[i], len2, &delta, &best);
      }
      if (k == 2) {                         // one of the block's later threads takes it through step 2
        const int slot = atomicAdd(&n_cand, 1);
        cand[slot] = (int16_t)threadIdx.x;
        cand_delta[slot] = (int16_t)delta;
      } else if (k >= 0) {
        finish(i, s, len2, delta, k);
        verdict[threadIdx.x] = (int16_t)delta;
      } else {
        // not decided here.  With DF_GAP_HINT or more mismatches on its best gap-free diagonal the values-only DP pass
        // would only find out that the read needs a trace: the planner sends it to the trace kernel directly (-3).
        bin_of[i] = best >= DF_GAP_HINT ? -3 : 0;
      }
    }
  }
  __syncthreads();
  // step 2 is three times as long as step 1 and concerns one read in five: run it on the candidates packed into the
  // block's first threads instead of leaving four in five lanes idle
  if ((int)threadIdx.x < n_cand) {
    const int t = cand[threadIdx.x], delta = cand_delta[threadIdx.x];
    const int64_t i = i0 + t;
    int s, l1;
    const int len2 = rs.len[i];
    read_window(ref, rs.as[i], rs.ae[i], len2, &s, &l1);       // as / ae of a candidate are still untouched
    const int via_table = (dbg & 64u) ? 0 : diag_step2_kmer(rp, ko, n_ref, s, l1, rs.packed + rs.roff[i], len2);
    if (via_table > 0 || (via_table < 0 && diag_step2(rp, s, l1, rs.packed + rs.roff[i], len2))) {
      finish(i, s, len2, delta, 2);
      verdict[t] = (int16_t)delta;
    } else {
      bin_of[i] = 0;
    }
  }
  __syncthreads();
  // the reads left over, as a list for the banded DP (k_band_align): one atomic per wavefront
  if (left_list) {
    const int64_t i = i0 + threadIdx.x;
    const bool left = i < rs.n && rs.sk[i] && verdict[threadIdx.x] < 0;
    const unsigned long long m = __ballot(left);
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(n_left, (uint32_t)__popcll(m));
    base = __shfl(base, 0);
    if (left) left_list[base + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (int32_t)i;
  }
  // the scripts of the finished reads: eight threads per read, four consecutive columns (8 bytes) per store
  for (int k = 0; k < 8; k++) {
    const int t = k * 32 + (int)(threadIdx.x >> 3);
    const int d = verdict[t];
    if (d < 0) continue;
    const int64_t j = i0 + t;
    const int len2 = rs.len[j];
    int16_t* cols = rs.cols + j * rs.stride;        // stride is a multiple of 4 and >= len2
    for (int base = (int)(threadIdx.x & 7) * 4; base < len2; base += 32) {
      const uint32_t c0 = (uint32_t)(d + base);
      uint2 v;
      v.x = (c0 & 0xFFFFu) | ((c0 + 1u) << 16);
      v.y = ((c0 + 2u) & 0xFFFFu) | ((c0 + 3u) << 16);
      *reinterpret_cast<uint2*>(cols + base) = v;
    }
  }
}

// ---- the banded DP (band_body.h) for the reads the filter left over: one read per thread, persistent grid of
// wavefronts, each with a private trace slab [row][lane][BAND_W bytes] so that a row's 64 stores are one 2 KB stretch.
// A finished read is marked bin_of = -4 (the planner leaves it out); every other read keeps its mark and its window.
constexpr int BAND_ROW_WORDS = 64 * (BAND_W / 4);
__global__ __launch_bounds__(64) void k_band_align(ReadSet rs, RefInfo ref, RefPlanes rp, KmerOcc ko, int64_t n_ref, const int32_t* left_list,
                                                   const uint32_t* n_left, uint32_t* trace_slabs, int64_t slab_words, int32_t* bin_of,
                                                   uint32_t* n_done, uint32_t* next_chunk, uint32_t dbg) {
  const uint32_t total = *n_left;
  uint32_t* trace = trace_slabs + (int64_t)blockIdx.x * slab_words + threadIdx.x * (BAND_W / 4);
  uint32_t done = 0;
  for (;;) {
    // chunks of 64 reads are handed out as wavefronts come free: their cost differs by the width of their bands
    uint32_t chunk = 0;
    if (threadIdx.x == 0) chunk = atomicAdd(next_chunk, 1u);
    chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
    if ((uint64_t)chunk * 64 >= total) break;
    const uint32_t t = chunk * 64 + threadIdx.x;
    const bool live = t < total;
    const int64_t i = live ? left_list[t] : 0;
    const int len2 = live ? rs.len[i] : 0;
    int s = 0, l1 = 0;
    if (live) read_window(ref, rs.as[i], rs.ae[i], len2, &s, &l1);
    const uint8_t* rd = rs.packed + (live ? rs.roff[i] : 0);
    BandPlan bp;
    const bool ok = live && band_plan(rp, ko, n_ref, s, l1, rd, len2, &bp);
    // the whole wavefront runs the widest band among its reads (a scalar loop bound), and the plain form of the
    // recurrence only if no read's band leaves its window
    int w = ok ? bp.w : 0;
    for (int o = 32; o; o >>= 1) { const int v = __shfl_xor(w, o); w = v > w ? v : w; }
    const int wmax = (__builtin_amdgcn_readfirstlane(w) + 3) & ~3;
    if (wmax == 0 || (dbg & 256u)) continue;
    const bool edge = __ballot(ok && !band_interior(bp, wmax, l1, len2)) != 0ull;
    if (!ok) continue;
    BandResult res;
    const bool got = edge ? band_align<true>(rp, s, l1, rd, len2, bp, wmax, trace, BAND_ROW_WORDS, rs.cols + i * rs.stride, &res, (dbg & 512u) != 0)
                          : band_align<false>(rp, s, l1, rd, len2, bp, wmax, trace, BAND_ROW_WORDS, rs.cols + i * rs.stride, &res, (dbg & 512u) != 0);
    if (!got) continue;
    rs.score[i] = res.score;
    rs.refstart[i] = s;
    rs.abr[i] = (int16_t)res.abr;
    rs.as[i] = res.abc + s;                   // src/mia_main.c:254-255
    rs.ae[i] = res.aec + s;
    rs.status[i] = res.gaps == 0 ? ST_DIAG : (res.gaps == 1 ? (ST_ONEGAP | (res.gap_desc << 8)) : ST_OK);
    bin_of[i] = -4;
    done++;
  }
  for (int o = 32; o; o >>= 1) done += __shfl_xor(done, o);
  if (threadIdx.x == 0 && done) atomicAdd(n_done, done);
}

__global__ __launch_bounds__(256) void k_plan_count(ReadSet rs, RefInfo ref, PackSet ps, int use_quad, int filtered, int defer, int32_t* bin_of, int32_t* bin_count,
                                                     uint32_t* n_filtered) {
  __shared__ int32_t hist[N_BINS];
  __shared__ uint32_t done;     // reads of this block that k_diag_filter finished (one global atomic per block, not per wave)
  if (threadIdx.x == 0) done = 0;
  for (int b = threadIdx.x; b < N_BINS; b += blockDim.x) hist[b] = 0;
  __syncthreads();
  for (int k = 0; k < PLAN_PER; k++) {
    const int64_t i = ((int64_t)blockIdx.x * PLAN_PER + k) * 256 + threadIdx.x;
    int b = -1;
    bool was_done = false;
    if (i < rs.n) {
      const int mark = filtered ? bin_of[i] : 0;
      if (mark == -2) {
        was_done = true;
      } else if (mark == -4 || mark == -5) {   // finished by a band kernel (counted there), or on one of its lists (it may be finishing
                                               // the read this very moment: the planner runs beside the band kernels and must not touch the mark)
      } else if (rs.sk[i]) {
        int s, l1;
        read_window(ref, rs.as[i], rs.ae[i], rs.len[i], &s, &l1);
        b = classify(rs.len[i], l1, ps, use_quad);
        // gap hint of the filter: skip the values-only pass; k_plan_recount brings the read back for the trace kernel
        if (mark == -3 && defer && b >= BIN_QUAD0) b = PLAN_DEFERRED - b;
      } else {
        rs.status[i] = ST_SKIPPED;
      }
      if (mark != -4 && mark != -5) bin_of[i] = b;
    }
    hist_add_aggregated(hist, b);
    const unsigned long long dm = __ballot(was_done);
    if ((threadIdx.x & 63) == 0 && dm) atomicAdd(&done, (uint32_t)__popcll(dm));
  }
  __syncthreads();
  for (int b = threadIdx.x; b < N_BINS; b += blockDim.x) if (hist[b]) atomicAdd(&bin_count[b], hist[b]);
  if (threadIdx.x == 0 && done) atomicAdd(n_filtered, done);
}

__global__ __launch_bounds__(256) void k_plan_fill(int64_t n, const int32_t* bin_of, const int32_t* bin_off, int32_t* bin_cursor,
                                                    int32_t* list) {
  __shared__ int32_t hist[N_BINS], base[N_BINS];
  for (int b = threadIdx.x; b < N_BINS; b += blockDim.x) hist[b] = 0;
  __syncthreads();
  int bb[PLAN_PER], rank[PLAN_PER];
  for (int k = 0; k < PLAN_PER; k++) {
    const int64_t i = ((int64_t)blockIdx.x * PLAN_PER + k) * 256 + threadIdx.x;
    bb[k] = i < n ? bin_of[i] : -1;
    rank[k] = hist_add_aggregated(hist, bb[k]);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < N_BINS; b += blockDim.x) if (hist[b]) base[b] = atomicAdd(&bin_cursor[b], hist[b]);
  __syncthreads();
  for (int k = 0; k < PLAN_PER; k++) {
    const int64_t i = ((int64_t)blockIdx.x * PLAN_PER + k) * 256 + threadIdx.x;
    if (bb[k] >= 0) list[bin_off[bb[k]] + base[bb[k]] + rank[k]] = (int32_t)i;
  }
}

// ---- the planner's answer without a host round trip (mia_hip_iterate) ---------------------------------------------
// What the host computes between k_plan_count and k_plan_fill in mia_hip_realign -- bin offsets with every quad bin padded
// to whole quads -- by one wavefront, and the ranges the DP kernels then read themselves (their dev_range argument):
//   hdr[PH_WIN + 2 ci] = {offset, count} of window class ci,  hdr[PH_QUAD] = {offset, quads},  hdr[PH_WIDE0] = {offset, count}
//   hdr[PH_RETRY] = {0, reads whose path left the quad kernel's trace band} (k_align_quad counts into the second word)
//   hdr[PH_TOTAL] = list entries in use (padding included)
enum { PH_WIN = 0, PH_QUAD = 2 * N_CPL, PH_WIDE0 = PH_QUAD + 2, PH_RETRY = PH_WIDE0 + 2, PH_TOTAL = PH_RETRY + 2, PH_RETRIED_PLAIN, PH_RETRY2, PH_WORDS = 16 };
static_assert(PH_RETRY2 + 2 <= PH_WORDS && (PH_RETRY2 & 1) == 0, "hdr[PH_RETRY2] = {0, reads the band kernels put on the retry list}: an aligned pair");
// list (or nullptr): the padding slots of the quad bins (each bin is rounded up to whole quads) are set to -1 = "no read" here -- up
// to three words per bin -- instead of clearing the whole list in front of this launch (4 MB per 1 M reads: a 65 us fill on the
// planner's stream, beside the band DPs)
__global__ __launch_bounds__(512) void k_plan_scan(const int32_t* count, int32_t* off, int32_t* hdr, int32_t quads_only, int32_t* list = nullptr) {
  __shared__ int32_t sh[512];
  const int b = threadIdx.x;
  const int c = b < N_BINS ? count[b] : 0;
  const int take = b >= N_BINS ? 0 : (b >= BIN_QUAD0 ? ((c + 3) & ~3) : (quads_only ? 0 : c));
  sh[b] = take;
  __syncthreads();
  for (int o = 1; o < 512; o <<= 1) {
    const int v = b >= o ? sh[b - o] : 0;
    __syncthreads();
    sh[b] += v;
    __syncthreads();
  }
  const int excl = sh[b] - take, run = sh[511];
  if (b < N_BINS) off[b] = excl;
  if (list && b >= BIN_QUAD0 && b < N_BINS) for (int q = c; q < take; q++) list[excl + q] = -1;
  if (!quads_only) {
    if (b < N_CPL) { hdr[PH_WIN + 2 * b] = excl; hdr[PH_WIN + 2 * b + 1] = c; }
    if (b == BIN_WIDE) { hdr[PH_WIDE0] = excl; hdr[PH_WIDE0 + 1] = c; }
  }
  if (b == BIN_QUAD0) { hdr[PH_QUAD] = excl; hdr[PH_QUAD + 1] = (run - excl) / 4; hdr[PH_TOTAL] = run; }
  if (quads_only) {      // reads the values-only pass handed on: everything still in a quad bin
    __syncthreads();
    sh[b] = (b >= BIN_QUAD0 && b < N_BINS) ? c : 0;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) { if (b < o) sh[b] += sh[b + o]; __syncthreads(); }
    if (b == 0) hdr[PH_RETRIED_PLAIN] = sh[0];
  }
}
// reads that need the exact kernel from the start: the head of wide_list (the DP kernels append their escapes behind)
// zero_a / zero_b (or nullptr): the planner's bin counts and cursors, used up by the k_plan_fill in front of this launch, cleared for the
// re-plan behind the values-only quad pass (two memsets there were five fill kernels, ~30 us on the chain a first iteration waits for)
__global__ __launch_bounds__(256) void k_wide_seed(const int32_t* list, const int32_t* hdr, int32_t* wide_list, int32_t* wide_count,
                                                    int32_t* zero_a = nullptr, int32_t* zero_b = nullptr) {
  const int off = hdr[PH_WIDE0], n = hdr[PH_WIDE0 + 1];
  for (int t = threadIdx.x; t < n; t += 256) wide_list[t] = list[off + t];
  if (threadIdx.x == 0) *wide_count = n;
  if (zero_a) for (int t = threadIdx.x; t < N_BINS; t += 256) { zero_a[t] = 0; zero_b[t] = 0; }
}

// ---- the windowed DP: one read at a time per 64-lane workgroup, persistent grid -------------
// The grid is sized to the machine (waves/CU x CUs); workgroup w takes reads w, w+G, w+2G ...
// of its bin.  LDS holds only the 5-column substitution table, the byte trace goes to the
// workgroup's private slab `trace_slabs + w*slab_bytes` which is re-used for every read.
template <int CPL>
__global__ __launch_bounds__(64) void k_align_window(ReadSet rs, RefInfo ref, const int32_t* pssm2, PackParams pk,
                                                      const int32_t* list, int32_t count, unsigned char* trace_slabs,
                                                      int64_t slab_bytes, int32_t* wide_list, int32_t* wide_count, uint32_t dbg,
                                                      const int32_t* dev_range) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SUB_LDS_BYTES];
  if (dev_range) { list += dev_range[0]; count = dev_range[1]; }     // the planner's answer stayed on the device (k_plan_scan)
  __builtin_amdgcn_s_setprio(3);    // (a handful of reads, each a chain of its window's rows: ahead of the persistent band grids on its SIMD)
  DevWave wave(lds_raw, trace_slabs + (int64_t)blockIdx.x * slab_bytes);
  for (int w = blockIdx.x; w < count; w += gridDim.x) {
    const int i = list[w];
    AlignArgs a;
    int s, l1;
    const int len2 = rs.len[i];
    read_window(ref, rs.as[i], rs.ae[i], len2, &s, &l1);
    a.ref_codes = ref.codes;
    a.ref_start = s;
    a.len1 = l1;
    a.read_packed = rs.packed + rs.roff[i];
    a.len2 = len2;
    a.pssm = pssm2 + (rs.rc[i] ? PSSM_WORDS : 0);   // src/mia_main.c:179-184
    a.sg5 = 1;                                      // sg_align leaves sg5 = sg3 = 1 (src/mia.c:1535-1538)
    a.pk = pk;
    a.lds_sub = 0;
    a.trace_stride = (uint32_t)((l1 + 3) & ~3);
    a.cols_out = rs.cols + (int64_t)i * rs.stride;
    a.dbg = dbg;
    AlignResult r = WindowAligner<DevWave, CPL>::run(wave, a);
    if (wave.lane() == 0) {
      if (r.status & ST_ESCAPE) {
        // a gap of >= 63 on the optimal path: the byte trace cannot represent it; hand the
        // read to the exact kernel (as/ae are left untouched so that it sees the same window)
        int p = atomicAdd(wide_count, 1);
        wide_list[p] = i;
      } else {
        rs.score[i] = r.score;
        rs.refstart[i] = s;
        rs.abr[i] = (int16_t)r.abr;
        rs.as[i] = r.abc + s;   // src/mia_main.c:254-255
        rs.ae[i] = r.aec + s;
      }
      rs.status[i] = r.status;
    }
    wave.lds_fence();   // the next read overwrites the substitution table
  }
}

// ---- the reads the band plan left open, straight from its own list (BxDev::open), one per wavefront: what k_plan_count would do with
// each of them (strand unknown: ST_SKIPPED; the window's class; the exact kernel's list for what no class holds) and then the window DP
// of that class.  One launch instead of the planner's count / scan / fill / seed, the quad kernel and four window launches -- for the
// few hundred reads per million a steady-state iteration leaves open (align_all: direct_open).  slab_bytes: the widest class's.
// list_b / count_b_p (or nullptr): a second list behind the first -- the reads the band DPs could not finish (their retry list): the last
// launch of the alignment takes both, a read of either kind per wavefront, in the time one of them takes.
__global__ __launch_bounds__(64) void k_align_open(ReadSet rs, RefInfo ref, const int32_t* pssm2, PackSet ps, const int32_t* list, const uint32_t* count_p,
                                                    unsigned char* trace_slabs, int64_t slab_bytes, int32_t* wide_list, int32_t* wide_count, uint32_t dbg,
                                                    const int32_t* list_b, const uint32_t* count_b_p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SUB_LDS_BYTES];
  const int count_a = (int)*count_p, count = count_a + (list_b ? (int)*count_b_p : 0);
  __builtin_amdgcn_s_setprio(3);
  DevWave wave(lds_raw, trace_slabs + (int64_t)blockIdx.x * slab_bytes);
  for (int w = blockIdx.x; w < count; w += gridDim.x) {
    const int i = w < count_a ? list[w] : list_b[w - count_a];
    if (!rs.sk[i]) { if (wave.lane() == 0) rs.status[i] = ST_SKIPPED; continue; }      // (k_plan_count's mark for a read reiterate_assembly skips: src/mia_main.c:178)
    AlignArgs a;
    int s, l1;
    const int len2 = rs.len[i];
    read_window(ref, rs.as[i], rs.ae[i], len2, &s, &l1);
    const int ci = classify(len2, l1, ps, 0);
    if (ci == BIN_WIDE) {
      if (wave.lane() == 0) { const int q = atomicAdd(wide_count, 1); wide_list[q] = i; }
      continue;
    }
    a.ref_codes = ref.codes;
    a.ref_start = s;
    a.len1 = l1;
    a.read_packed = rs.packed + rs.roff[i];
    a.len2 = len2;
    a.pssm = pssm2 + (rs.rc[i] ? PSSM_WORDS : 0);
    a.sg5 = 1;
    a.pk = ps.p[ci];
    a.lds_sub = 0;
    a.trace_stride = (uint32_t)((l1 + 3) & ~3);
    a.cols_out = rs.cols + (int64_t)i * rs.stride;
    a.dbg = dbg;
    AlignResult r;
    if (ci == 0) r = WindowAligner<DevWave, 4>::run(wave, a);
    else if (ci == 1) r = WindowAligner<DevWave, 8>::run(wave, a);
    else r = WindowAligner<DevWave, 12>::run(wave, a);
    if (wave.lane() == 0) {
      if (r.status & ST_ESCAPE) {
        const int q = atomicAdd(wide_count, 1);
        wide_list[q] = i;
      } else {
        rs.score[i] = r.score;
        rs.refstart[i] = s;
        rs.abr[i] = (int16_t)r.abr;
        rs.as[i] = r.abc + s;   // src/mia_main.c:254-255
        rs.ae[i] = r.aec + s;
      }
      rs.status[i] = r.status;
    }
    wave.lds_fence();
  }
}

// ---- four reads per wavefront (align_body_quad.h): windows <= 208 columns, equal read lengths ----
__global__ __launch_bounds__(64, 4) void k_align_quad(ReadSet rs, RefInfo ref, const int32_t* pssm2, PackParams pk, const int32_t* list,
                                                    int32_t n_quads, unsigned char* trace_slabs, int64_t slab_bytes,
                                                    int32_t* wide_list, int32_t* wide_count, int32_t* retry_list, int32_t* retry_count,
                                                    int32_t band, uint32_t dbg, const int32_t* dev_range) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // Q_G * q_sub_bytes(longest read)
  if (dev_range) { list += dev_range[0]; n_quads = dev_range[1]; }
  __builtin_amdgcn_s_setprio(1);    // (between the step's chain -- values DP, late trace -- and k_bxl_trace, which has slack)
  DevWave wave(lds_raw, trace_slabs + (int64_t)blockIdx.x * slab_bytes);
  for (int qd = blockIdx.x; qd < n_quads; qd += gridDim.x) {
    QuadArgs a;
    int idx[Q_G];
    a.ref_codes = ref.codes;
    a.packed = rs.packed;
    a.pssm2 = pssm2;
    a.pk = pk;
    a.lds_sub = 0;
    a.slab_group = (uint32_t)(slab_bytes / Q_G);
    a.dbg = dbg;
    a.band = band;
    a.len2 = 1;
    for (int g = 0; g < Q_G; g++) {
      const int i = list[4 * qd + g];
      idx[g] = i;
      a.ref_start[g] = 0; a.len1[g] = 0; a.roff[g] = 0; a.rc[g] = 0; a.cols_out[g] = rs.cols; a.dexp[g] = 0;
      if (i >= 0) {
        int s, l1;
        const int len2 = rs.len[i];
        read_window(ref, rs.as[i], rs.ae[i], len2, &s, &l1);
        a.len2 = len2;      // equal for the whole quad: the planner pads every length bin to a multiple of four
        a.ref_start[g] = s; a.len1[g] = l1; a.roff[g] = rs.roff[i]; a.rc[g] = rs.rc[i];
        a.cols_out[g] = rs.cols + (int64_t)i * rs.stride;
        // expected diagonal: where the read started last time, minus its soft-clipped rows (0 after pass 1)
        a.dexp[g] = (rs.as[i] - s) - rs.abr[i];
      }
    }
    AlignResult res[Q_G];
    QuadAligner<DevWave>::run(wave, a, res);
    if (wave.lane() == 0) {
      for (int g = 0; g < Q_G; g++) {
        const int i = idx[g];
        if (i < 0) continue;
        if (res[g].status & ST_BAND) {          // path outside the stored band: one-read kernel, full trace
          int p = atomicAdd(retry_count, 1);
          retry_list[p] = i;
        } else if (res[g].status & ST_ESCAPE) {
          int p = atomicAdd(wide_count, 1);
          wide_list[p] = i;
        } else {
          rs.score[i] = res[g].score;
          rs.refstart[i] = a.ref_start[g];
          rs.abr[i] = (int16_t)res[g].abr;
          rs.as[i] = res[g].abc + a.ref_start[g];
          rs.ae[i] = res[g].aec + a.ref_start[g];
        }
        rs.status[i] = res[g].status;
      }
    }
    wave.lds_fence();
  }
}

// ---- first pass over the quad bins: values only + the diagonal proof (align_body_quad_plain.h).  A proven read is
// finished here and leaves its bin (bin_of = -1); the others keep their bin and are re-planned into quads for k_align_quad.
__global__ __launch_bounds__(64, 4) void k_align_quad_plain(ReadSet rs, RefInfo ref, const int32_t* pssm2, const int32_t* list,
                                                          int32_t n_quads, int32_t* bin_of, const int32_t* dev_range) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // Q_G * q_sub_bytes(longest read)
  if (dev_range) { list += dev_range[0]; n_quads = dev_range[1]; }
  DevWave wave(lds_raw, nullptr);
  for (int qd = blockIdx.x; qd < n_quads; qd += gridDim.x) {
    QuadPlainArgs a;
    int idx[Q_G];
    a.ref_codes = ref.codes;
    a.packed = rs.packed;
    a.pssm2 = pssm2;
    a.lds_sub = 0;
    a.len2 = 1;
    for (int g = 0; g < Q_G; g++) {
      const int i = list[4 * qd + g];
      idx[g] = i;
      a.ref_start[g] = 0; a.len1[g] = 0; a.roff[g] = 0; a.rc[g] = 0; a.cols_out[g] = rs.cols;
      if (i >= 0) {
        int s, l1;
        const int len2 = rs.len[i];
        read_window(ref, rs.as[i], rs.ae[i], len2, &s, &l1);
        a.len2 = len2;
        a.ref_start[g] = s; a.len1[g] = l1; a.roff[g] = rs.roff[i]; a.rc[g] = rs.rc[i];
        a.cols_out[g] = rs.cols + (int64_t)i * rs.stride;
      }
    }
    QuadPlainResult res[Q_G];
    QuadPlainAligner<DevWave>::run(wave, a, res);
    if (wave.lane() == 0) {
      for (int g = 0; g < Q_G; g++) {
        const int i = idx[g];
        if (i < 0 || !res[g].proven) continue;               // unproven: as/ae untouched, the trace kernel sees the same window
        rs.score[i] = res[g].score;
        rs.refstart[i] = a.ref_start[g];
        rs.abr[i] = (int16_t)res[g].abr;
        rs.as[i] = res[g].abc + a.ref_start[g];               // src/mia_main.c:254-255
        rs.ae[i] = res[g].aec + a.ref_start[g];
        rs.status[i] = ST_DIAG;
        bin_of[i] = -1;
      }
    }
    wave.lds_fence();
  }
}

// reads still in a quad bin after the first pass (everything else is done): counts per bin
__global__ __launch_bounds__(256) void k_plan_recount(int64_t n, int32_t* bin_of, int32_t* bin_count) {
  __shared__ int32_t hist[N_BINS];
  for (int b = threadIdx.x; b < N_BINS; b += blockDim.x) hist[b] = 0;
  __syncthreads();
  for (int k = 0; k < PLAN_PER; k++) {
    const int64_t i = ((int64_t)blockIdx.x * PLAN_PER + k) * 256 + threadIdx.x;
    int b = -1;
    if (i < n) {
      b = bin_of[i];
      if (b <= PLAN_DEFERRED) { b = PLAN_DEFERRED - b; bin_of[i] = b; }
      else if (b >= 0 && b < BIN_QUAD0) { bin_of[i] = -1; b = -1; }
    }
    hist_add_aggregated(hist, b);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < N_BINS; b += blockDim.x) if (hist[b]) atomicAdd(&bin_count[b], hist[b]);
}

// ---- exact wide kernel: one read per thread, int32 scores and trace in global scratch.
// Handles whole-reference windows (src/mia_main.c:209-212), gaps >= 63 and anything the
// packed kernel cannot represent.  Slow by design; it is the rare path.
struct WideJob { int64_t scratch_off; };  // in int32 words

__global__ void k_align_wide(ReadSet rs, RefInfo ref, const int32_t* pssm2, const int32_t* list, int32_t count,
                             const int64_t* scratch_off, int32_t* scratch) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int i = list[t];
  const int len2 = rs.len[i];
  int s, n1;
  read_window(ref, rs.as[i], rs.ae[i], len2, &s, &n1);
  const uint8_t* c1 = ref.codes + s;
  const uint8_t* rp = rs.packed + rs.roff[i];
  const int32_t* pm = pssm2 + (rs.rc[i] ? PSSM_WORDS : 0);
  int32_t* base = scratch + scratch_off[t];
  int32_t* T = base;                               // [len2][n1]
  int32_t* S0 = base + (int64_t)len2 * n1;         // three rotating score rows
  int32_t* colkey = S0 + 3 * (int64_t)n1;
  int32_t* colrow = colkey + n1;
  auto rcode = [&](int r) { return (int)((rp[r >> 1] >> ((r & 1) * 4)) & 15); };
  int32_t *prev2 = S0, *prev = S0 + n1, *cur = S0 + 2 * (int64_t)n1;
  {
    const int c2 = rcode(0);
    for (int c = 0; c < n1; c++) {
      cur[c] = pm[(0 * 5 + c1[c]) * 5 + c2];
      T[c] = 0;
      colkey[c] = cur[c];
      colrow[c] = 0;
    }
  }
  for (int r = 1; r < len2; r++) {
    int32_t* tmp = prev2; prev2 = prev; prev = cur; cur = tmp;
    const int d = sm_depth(r, len2), c2 = rcode(r);
    const int fresh = -(GOP + GEP * (r + 1));
    int32_t* tr = T + (int64_t)r * n1;
    cur[0] = pm[(d * 5 + c1[0]) * 5 + c2] + fresh;
    tr[0] = 0;
    int rowkey = prev[0], rowcol = 0;
    for (int c = 1; c < n1; c++) {
      const int sub = pm[(d * 5 + c1[c]) * 5 + c2];
      int gapc = -(1 << 30), gapr = -(1 << 30);
      if (c >= 2) {
        int k = prev[c - 2] + GEP * (c - 2);
        if (k > rowkey) { rowkey = k; rowcol = c - 2; }
        gapc = rowkey - GOP - GEP * (c - 1);
      }
      if (r >= 2) {
        int k = prev2[c - 1] + GEP * (r - 2);
        if (k > colkey[c - 1]) { colkey[c - 1] = k; colrow[c - 1] = r - 2; }
        gapr = colkey[c - 1] - GOP - GEP * (r - 1);
      }
      const int diag = prev[c - 1];
      if (fresh > diag && fresh > gapc && fresh > gapr) { cur[c] = fresh; tr[c] = c; }
      else if (diag >= gapc && diag >= gapr) { cur[c] = sub + diag; tr[c] = 0; }
      else if (gapc >= gapr) { cur[c] = sub + gapc; tr[c] = rowcol; }
      else { cur[c] = sub + gapr; tr[c] = -colrow[c - 1]; }
    }
  }
  int best = INT32_MIN, aec = 0;
  for (int c = 0; c < n1; c++) if (cur[c] > best) { best = cur[c]; aec = c; }
  // traceback (src/mia.c:612-637,1440-1497): first walk finds the alignment start so that the
  // script can be stored relative to it (whole-reference windows exceed int16 otherwise)
  int16_t* cols = rs.cols + (int64_t)i * rs.stride;
  int r = len2 - 1, c = aec;
  for (;;) {
    const int t2 = T[(int64_t)r * n1 + c];
    if (t2 == c || t2 == -r) break;
    if (t2 == 0) { r--; c--; }
    else if (t2 < 0) { r = -t2; c--; }
    else { c = t2; r--; }
  }
  const int abr = r, abc = c;
  int ncol = 0;
  r = len2 - 1; c = aec;
  for (;;) {
    const int t2 = T[(int64_t)r * n1 + c];
    const int rel = c - abc;
    cols[r] = (int16_t)(rel > 32767 ? 32767 : rel);
    ncol++;
    if (t2 == c || t2 == -r) break;
    if (t2 == 0) { r--; c--; }
    else if (t2 < 0) { int nr = -t2; r--; c--; while (r > nr) { cols[r] = COL_INSERT; r--; ncol++; } }
    else { ncol += c - 1 - t2; r--; c = t2; }
  }
  for (int q = 0; q < abr; q++) cols[q] = COL_CLIP;
  uint32_t st = ST_OK;
  if (ncol > 2 * MAX_READ) st |= ST_TOO_LONG;
  rs.score[i] = best;
  rs.refstart[i] = s + abc;   // script columns are relative to the alignment start here
  rs.abr[i] = (int16_t)abr;
  rs.as[i] = abc + s;         // src/mia_main.c:254-255
  rs.ae[i] = aec + s;
  rs.status[i] = st;
}

}  // namespace mia
