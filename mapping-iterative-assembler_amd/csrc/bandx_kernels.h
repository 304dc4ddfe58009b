// bandx_kernels.h -- gfx950 kernels around bandx_body.h: the re-alignment of every stored read in its window
// (reiterate_assembly, /root/reference/src/mia_main.c:24-280: dyn_prog, max_sg_score, find_align_begin,
// populate_pwaln_to_begin per read) for any substitution matrix.
//
//   k_ref_nibbles  the reference as 4-bit codes (the DP looks the substitution score of a cell up by this code)
//   k_bx_umax      U of every read (once per read set and matrix)
//   k_bx_plan      one read per thread: band from the 10-mer anchors; one-diagonal bands are finished here; the rest is
//                  appended to one of eight lists (values / trace x band class)
//   k_bx_values    persistent wavefronts take chunks of 64 reads of one class: values-only DP in registers, substitution
//                  table in LDS; a read whose best score is the plan's diagonal's is finished, the others stay open for the
//                  full-window kernels.  Runs BESIDE k_bx_trace (two streams): neither depends on the other.
//   k_bx_trace     the same with a byte trace in a private slab and the reference's traceback
// Nothing here waits for the host: list lengths and chunk cursors stay on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bandx_body.h"
#include "bandx_lanes.h"
#include "mia_kernels.h"

namespace mia {

// counters of the pipeline; each on a cache line of its own (BXC_STRIDE words apart): thousands of wavefronts add to them,
// and atomics on one line are served one after the other
enum { BXC_LIST0 = 0, BXC_CUR_VALUES = 2 * BX_NCLS, BXC_CUR_TRACE, BXC_DONE_PLAN, BXC_DONE_VALUES, BXC_DONE_TRACE, BXC_SEEN, BXC_FAIL0 = 16, BXC_LATE0 = 24, BXC_OPEN = 29, BXC_CAND = 30, BXC_CAND2 = 31, BXC_LATE2_0 = 32, BXC_COUNTERS = 40 };     // LATE2: what the full plan lists beside the DPs (BxDev::to_late); the first 32 are what mia_hip_bx_counters reports     // CAND: reads handed from the plan's first launch to its second; CAND2: on to the third (fine blocks)
static_assert(BXC_SEEN < BXC_FAIL0 && BXC_FAIL0 + BXF_KINDS <= BXC_LATE0 && BXC_LATE0 + BX_NCLS <= BXC_OPEN && BXC_OPEN < BXC_CAND && BXC_CAND2 < BXC_LATE2_0 && BXC_LATE2_0 + BX_NCLS <= BXC_COUNTERS, "counter layout");     // OPEN: reads the plan leaves to the full-window kernels (BxDev::open)
constexpr int BXC_STRIDE = 64;
constexpr int BXC_WORDS = BXC_COUNTERS * BXC_STRIDE;
__device__ __forceinline__ uint32_t* bxc(const uint32_t* ctr, int k) { return const_cast<uint32_t*>(ctr) + (size_t)k * BXC_STRIDE; }

struct BxCandRec;
struct BxDev {
  BxTab tab;               // tables in global memory
  const int32_t* sub256;   // tab.sub times 256 (the trace DP's packed words)
  const uint32_t* refnib;
  const int32_t* umax;     // [n] U of every read, -1 for a read with N; nullptr: a borrowed read set, U is summed where it is needed
  const uint64_t* rplanes; // [n][2 * rplane_words] bit planes of the reads (k_read_planes), or nullptr
  int32_t rplane_words;
  uint32_t* plan;          // [n] packed BxPlan of a read on a list
  int32_t* expect;         // [n] U - b0: the score of the plan's diagonal
  int32_t* lists;          // [4 * BX_NCLS][list_stride] read indices: values lists, trace lists, late trace lists (what the values DP could not finish), second late lists (what the full plan lists beside the DPs: to_late)
  int32_t* retry;          // reads no band kernel could finish (the reference's index-0 quirk ...): the one-read-per-wavefront window kernel takes them
  int32_t* retry_n;        // ... their number
  int32_t listed_mark;     // bin_of of a read on a list: -5 (the planner leaves it alone: it runs BESIDE the band kernels), or 0 (open: round 2's order)
  int64_t list_stride;
  uint32_t* ctr;           // BXC_*
  int32_t lazy_scripts;    // the scripts of reads finished as pure diagonals are not written (k_diag_scripts makes them when asked for)
  uint8_t* early;          // [n] or nullptr: 1 for the reads the plan finishes (the early tally, mia_consensus_kernels.h: k_rec_early), 0 for all others
  uint32_t dbg;            // MIA_HIP_BX_DEBUG (profiling only, results are wrong): 1 no traceback, 2 one DP row only
  // k_bx_plan in two launches (phase 1 / phase 2): the reads whose anchors lie on two diagonals (or that want the end-indel
  // rescue) are handed from the first to the second through this list; nullptr: one launch, the block's first threads finish them
  struct BxCandRec* cand;
  uint32_t* cand_n;
  // ... and the reads whose loss exceeds what the 10-mers vouch for go on to a third launch (phase 3: bx_fine_anchors); nullptr: no third launch
  struct BxCandRec* cand2;
  uint32_t* cand2_n;
  // the band DPs in two rounds (align_all: split_dp): `snap` holds the lengths of the values / trace lists as the plan's first launch left
  // them (k_bx_snap); round 1 takes the entries below, round 2 -- behind the plan's last launch -- the ones from there on
  const uint32_t* snap;
  // round 6: the reads the plan leaves OPEN (given up on, or strand unknown) listed by the plan itself -- at steady state a few hundred per
  // million, which k_align_open then takes one per wavefront, instead of the planner's count / scan / fill over ALL reads and a quad pass
  // (align_all: direct_open); nullptr: the planner bins them
  int32_t* open;
  uint32_t* open_n;
  // round 6: the QUICK plan (bx_quick, phase 4 of k_bx_plan) looks at every read on the diagonal it was aligned on before and finishes or
  // lists nine in ten; what it cannot decide goes on qlist, which the plan's first launch proper then takes as its in_list.  mark_all: the
  // reads of in_list get their marks (bin_of, the open list) from this launch -- they have none yet (in_list of the diagonal filter: 0)
  int32_t* qlist;
  uint32_t* qlist_n;
  int32_t mark_all;
  KmerBits kb;             // the two bitmaps the quick plan asks (w == nullptr: no quick plan)
  int32_t plane_words;     // phase 5 (the quick plan with the reference's planes in LDS): words per plane
  // mia_hip_iterate forks behind the QUICK plan: both band DPs start there (the trace DP of its lists is the longest kernel of the step),
  // and the full plan's launch for the hundredth read the quick plan left runs beside them on the context's stream.  What that launch
  // lists goes on the SECOND LATE lists (the DPs are reading their own by then), values and trace plans alike; a trace launch of their
  // own takes them (k_bxl_trace_late2), beside the one behind the values DP
  int32_t to_late;
  // stretches of 256 reads per workgroup of the quick plan (<= BX_QCH).  Every workgroup ends with one round of reservations on the lists'
  // counters, and a counter serves them one at a time (20 ns each): 1 M flat reads 122 us with four stretches, 148 with two (twice the
  // workgroups), 147 with eight (two wavefronts per SIMD); 10 M solexa reads 5.44 ms per step with four, 5.33 with eight
  int32_t qch;
};
// which entries of the plan's lists a launch of the band DPs takes
enum { BX_PART_ALL = 0, BX_PART_HEAD = 1, BX_PART_TAIL = 2 };
__global__ void k_bx_snap(const uint32_t* ctr, uint32_t* snap);
struct BxCandRec { int32_t i; BxAnchors an; };
constexpr int BX_FINE_LANES = 1;         // lanes that share the fine blocks of one read in the plan's third launch (measured: what a read costs there is
                                         // bx_finish, which one lane walks alone -- eight lanes per read made the launch no shorter and the bulk case four times longer)

__device__ __forceinline__ uint32_t bx_pack(const BxPlan& p) {
  return (uint32_t)(p.d0 + 512) | ((uint32_t)(p.dstar - p.d0) << 11) | ((uint32_t)p.edge << 17) | ((uint32_t)p.w << 18);
}

__device__ __forceinline__ uint32_t ref_nibble_word(const uint8_t* codes, int64_t n_codes, int64_t w) {
  uint32_t v = 0;
  for (int k = 0; k < 8; k++) {
    const int64_t p = w * 8 + k - BX_NIB_LEAD;
    const uint32_t c = (p >= 0 && p < n_codes) ? codes[p] : 4u;
    v |= (c > 4u ? 4u : c) << (4 * k);
  }
  return v;
}
__global__ __launch_bounds__(256) void k_ref_nibbles(const uint8_t* codes, int64_t n_codes, int64_t words, uint32_t* out) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= words) return;
  out[w] = ref_nibble_word(codes, n_codes, w);
}

__global__ __launch_bounds__(256) void k_bx_umax(ReadSet rs, const int32_t* mrow, int32_t* umax) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < rs.n) umax[i] = bx_umax(mrow, rs.packed + rs.roff[i], rs.len[i], rs.rc[i] ? 1 : 0);
}

// append `i` to list `which` for the lanes with want == true: one atomic per wavefront and list
__device__ __forceinline__ void bx_append(const BxDev& bx, int which, bool want, int32_t i) {
  const unsigned long long m = __ballot(want);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == __builtin_ctzll(m)) base = atomicAdd(bxc(bx.ctr, BXC_LIST0 + which), (uint32_t)__popcll(m));
  base = __shfl(base, __builtin_ctzll(m));
  if (want) bx.lists[(int64_t)which * bx.list_stride + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}

// append `i` to a plain list for the lanes with want == true: one atomic per wavefront
__device__ __forceinline__ void bxl_append(int32_t* list, uint32_t* counter, bool want, int32_t i) {
  const unsigned long long m = __ballot(want);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == __builtin_ctzll(m)) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = __shfl(base, __builtin_ctzll(m));
  if (want) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}

// the script of a pure diagonal (consecutive columns from c0 on) for the lanes in `mask`: the wavefront writes one read
// per step, four columns (8 bytes) per lane
__device__ __forceinline__ void bx_diag_scripts(const ReadSet& rs, unsigned long long mask, int32_t i, int c0, int len2) {
  const int lane = threadIdx.x & 63;
  while (mask) {
    const int l = __builtin_ctzll(mask);
    mask &= mask - 1;
    const int64_t j = __shfl(i, l);
    const int d = __shfl(c0, l), n = __shfl(len2, l);
    int16_t* cols = rs.cols + j * rs.stride;        // stride is a multiple of 4 and >= len2
    const int base = lane * 4;
    if (base < n) {
      const uint32_t c = (uint32_t)(d + base);
      uint2 v;
      v.x = (c & 0xFFFFu) | ((c + 1u) << 16);
      v.y = ((c + 2u) & 0xFFFFu) | ((c + 3u) << 16);
      *reinterpret_cast<uint2*>(cols + base) = v;
    }
  }
}

// The scripts the band pipeline did not write (BxDev::lazy_scripts): a read finished as a pure diagonal (ST_DIAG, abr = 0)
// sits on consecutive columns from as - refstart on.  Nothing on the device reads the script of such a read (k_rec_geom,
// the tallies and the insert events all go by the flag); the host does, through mia_hip_get_scripts.
__global__ __launch_bounds__(256) void k_diag_scripts(ReadSet rs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool mine = i < rs.n && rs.sk[i] && (rs.status[i] & ST_DIAG) && rs.abr[i] == 0;     // (the band pipeline's are all of this kind)
  bx_diag_scripts(rs, __ballot(mine), (int32_t)(mine ? i : 0), mine ? rs.as[i] - rs.refstart[i] : 0, mine ? rs.len[i] : 0);
}

// the reference's 10-mers into the hash table (bandx_body.h: KmerHash); slots and ovf are all ones / anything before.
// A 10-mer with up to `wild` N columns goes in under each of its 4^k spellings (the host sized the table for them).
// one spelling (10-mer index idx) of the 10-mer at reference position p
__device__ __forceinline__ void kmer_hash_insert_key(uint32_t idx, int64_t p, uint32_t* slot, int32_t* ovf, uint32_t mask, int32_t shift) {
  uint32_t h = (idx * 2654435761u) >> shift;
  for (;;) {
    uint32_t* e = slot + 4 * (size_t)h;
    const uint32_t prev = atomicCAS(&e[0], KH_EMPTY, idx);
    if (prev == KH_EMPTY || prev == idx) {
      const uint32_t c = atomicAdd(&e[2], 1u) + 1u;           // (starts at all ones)
      if (c == 0) e[1] = (uint32_t)p; else if (c == 1) e[3] = (uint32_t)p; else if (c < 4) ovf[2 * (size_t)h + c - 2] = (int32_t)p;
      break;
    }
    h = (h + 1) & mask;
  }
}
__device__ __forceinline__ void kmer_hash_insert(const uint8_t* codes, int64_t n_codes, int64_t p, uint32_t* slot, int32_t* ovf, uint32_t mask, int32_t shift, int32_t wild) {
  uint32_t base;
  uint64_t npos;
  const int k = kmer_wild_at(codes, n_codes, p, &base, &npos);
  if (k < 0 || k > wild) return;
  for (uint32_t x = 0; x < (1u << (2 * k)); x++) kmer_hash_insert_key(kmer_wild_key(base, npos, k, x), p, slot, ovf, mask, shift);
}
// the same for 64 consecutive positions by a whole workgroup: a 10-mer with two or three N columns has 16 or 64 spellings, and
// a thread that enters them one after the other is what a reference full of ambiguity codes (mt311: one 10-mer in six) made
// k_ref_prep wait for -- 0.16 ms.  Here the positions' spellings are numbered through (a prefix sum over the 64 counts) and
// dealt to the 256 threads one by one, whatever position they belong to.
__device__ __forceinline__ void kmer_hash_insert_block(const uint8_t* codes, int64_t n_codes, int64_t p0, uint32_t* slot, int32_t* ovf, uint32_t mask, int32_t shift,
                                                       int32_t wild) {
  __shared__ uint32_t s_base[64];
  __shared__ uint64_t s_npos[64];
  __shared__ int32_t s_k[64], s_off[65];
  if (threadIdx.x < 64) {
    uint32_t base;
    uint64_t npos;
    const int k = kmer_wild_at(codes, n_codes, p0 + threadIdx.x, &base, &npos);
    const int32_t cnt = (k >= 0 && k <= wild) ? (1 << (2 * k)) : 0;
    s_base[threadIdx.x] = base; s_npos[threadIdx.x] = npos; s_k[threadIdx.x] = k;
    int32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int32_t u = __shfl_up(inc, o); if ((int)threadIdx.x >= o) inc += u; }
    s_off[threadIdx.x + 1] = inc;
    if (threadIdx.x == 0) s_off[0] = 0;
  }
  __syncthreads();
  const int32_t total = s_off[64];
  for (int32_t item = threadIdx.x; item < total; item += 256) {
    int j = 0;                                     // the position whose spellings hold `item`: largest j with s_off[j] <= item
#pragma unroll
    for (int step = 32; step; step >>= 1) if (s_off[j + step] <= item) j += step;
    kmer_hash_insert_key(kmer_wild_key(s_base[j], s_npos[j], s_k[j], (uint32_t)(item - s_off[j])), p0 + j, slot, ovf, mask, shift);
  }
  __syncthreads();
}
__global__ __launch_bounds__(256) void k_kmer_hash(const uint8_t* codes, int64_t n_codes, uint32_t* slot, int32_t* ovf, uint32_t mask, int32_t shift, int32_t wild) {
  kmer_hash_insert(codes, n_codes, (int64_t)blockIdx.x * 256 + threadIdx.x, slot, ovf, mask, shift, wild);
}

// the quick plan's two bitmaps (bandx_body.h: KmerBits) from the start positions 0 .. L - 1 of the wrapped reference; both cleared before
__global__ __launch_bounds__(256) void k_kmer_bits(const uint8_t* codes, int64_t n_codes, int64_t L, KbPair* w) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < L) kmer_bits_insert(codes, n_codes, p, w);
}

// bit planes of every read (diag_filter.h: DiagScan::load_read), once per read set: words lo words, then words hi words
__global__ __launch_bounds__(256) void k_read_planes(ReadSet rs, int32_t words, uint64_t* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rs.n) return;
  DiagScan<4> sc;
  sc.load_read(rs.packed + rs.roff[i], rs.len[i]);
  uint64_t* o = out + i * 2 * words;
  for (int j = 0; j < words; j++) { o[j] = sc.rlo[j]; o[words + j] = sc.rhi[j]; }
}

// in_list / n_in_p: the reads to look at (what the diagonal filter left over), or nullptr: all n_all reads, and then
// every read gets its mark in bin_of (0 = open, -4 = finished here or later by the band kernels).
// NW: 64-row words of the longest read of the set.  Two phases: every thread takes its read through the anchors and, if
// they lie on one diagonal (nine reads in ten), through the rest; the others are collected and finished by the block's
// first threads, so that their longer way (two diagonals, the switch row between them) is not walked by whole wavefronts
// for the sake of a few lanes.
// PH (a template parameter: each launch carries only the registers and LDS of its own part): 0 = everything in one launch (the
// reads with anchors on two diagonals by the block's first threads); 1 = the first of three launches (those reads go on
// bx.cand, the reads the 10-mers cannot vouch for on bx.cand2); 2 = bx.cand with every lane at work; 3 = bx.cand2, the
// fine blocks (bx_fine_anchors), the diagonals of one read dealt to BX_FINE_LANES neighbouring lanes.
#ifdef MIA_HIP_ALT_PATHS
// (MIA_HIP_BX_DEBUG & 512, alt build: shader-clock cycles of a wavefront's stretches of k_bx_plan's first launch, summed over the wavefronts
// -- 0 set-up, 1 fetch + window, 2 planes, 3 anchors, 4 finish, 5 emit, 6 hand-over; slot 11 counts the wavefronts; tools/plan_clk_probe.py)
// (the stamps stay in registers; a block adds its four wavefronts' sums to one of 64 stripes at its end: an atomic per stamp and
// wavefront on seven words made the launch six times as long)
__device__ unsigned long long g_plan_clk[64 * 16];
// (with the quick plan in front -- phase 4 -- it is that launch the stamps are taken in: the first launch proper then has an in_list and stays out)
#define PLAN_CLK(k) do { if (pclk_on_) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pacc_[k] += now_ - pclk_; pclk_ = now_; } } while (0)
#define PLAN_CLK_DECL const bool pclk_on_ = (bx.dbg & 512u) != 0u && (PH == 4 || PH == 5 || in_list == nullptr); unsigned long long pclk_ = pclk_on_ ? __builtin_amdgcn_s_memtime() : 0ull, pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; __shared__ unsigned long long pclk_lds_[8]; if (threadIdx.x < 8) pclk_lds_[threadIdx.x] = 0
#define PLAN_CLK_FLUSH do { if (pclk_on_) { if ((threadIdx.x & 63) == 0) for (int q_ = 0; q_ < 7; q_++) atomicAdd(&pclk_lds_[q_], pacc_[q_]); __syncthreads(); \
    if (threadIdx.x < 7) atomicAdd(&g_plan_clk[(blockIdx.x & 63) * 16 + threadIdx.x], pclk_lds_[threadIdx.x]); if (threadIdx.x == 7) atomicAdd(&g_plan_clk[(blockIdx.x & 63) * 16 + 11], 4ull); } } while (0)
#else
#define PLAN_CLK(k) do { } while (0)
#define PLAN_CLK_DECL do { } while (0)
#define PLAN_CLK_FLUSH do { } while (0)
#endif
// 64-bit words of dynamic LDS the quick plan's per-read arrays take (two 16-bit words and a byte per read), in front of the planes
__host__ __device__ constexpr int bx_quick_lds_words(int qch) { return (256 * qch * 5 + 7) / 8; }
constexpr int BX_QCH = 8;          // stretches of 256 reads a workgroup of the quick plan (phase 4) takes AT MOST (BxDev::qch: four up to four million reads, eight beyond)
template <int NW, int PH>
__global__ __launch_bounds__(256) void k_bx_plan(ReadSet rs, RefInfo ref, RefPlanes rp, KmerHash ko, int64_t n_ref, BxDev bx, const int32_t* in_list,
                                                  const uint32_t* n_in_p, int64_t n_all, int32_t* bin_of) {
  // PH: 0 everything in one launch; 1 / 2 / 3 the three launches of the full plan; 4 the quick plan, 5 the same with the reference's planes in
  // LDS; 6 = phase 0 over a list that may be LONGER than the grid (the quick plan's undecided reads, taken beside the band DPs by a grid
  // the chip holds at once: four thousand workgroups, nearly all with nothing to do, queue for slots the persistent DP grids hold, and the
  // launch is not over until the last of them has had one): the same code with a loop around it -- an instance of its own, the loop costs
  // registers
  constexpr int P = PH == 6 ? 0 : PH;
  constexpr int SLOT_FINE = 2 * BX_NCLS + 2 + BXF_KINDS;     // blk_cnt: reads handed on to the third launch
  __shared__ int16_t loss_lds[BX_LOSS_WORDS];
  __shared__ BxAnchors cand_an[P < 2 ? 256 : 1];
  __shared__ uint8_t cand_tid[P < 2 ? 256 : 1];
  __shared__ int32_t n_cand;
  __shared__ uint32_t blk_cnt[SLOT_FINE + 1], blk_base[2 * BX_NCLS + 1];   // per block: list appends, finished, seen, reasons, hand-overs
  PLAN_CLK_DECL;
  if (P < 2 && in_list && (int64_t)blockIdx.x * 256 >= (int64_t)*n_in_p) return;      // (a short list leaves most of the grid nothing to do)
  for (int k = threadIdx.x; k < BX_LOSS_WORDS; k += 256) loss_lds[k] = bx.tab.loss[k];
  if (threadIdx.x == 0) n_cand = 0;
  if (threadIdx.x <= SLOT_FINE) blk_cnt[threadIdx.x] = 0;
  __syncthreads();
  BxTab T = bx.tab;
  T.loss = loss_lds;
  const int64_t total = in_list ? (int64_t)*n_in_p : n_all;
  int64_t t0 = (int64_t)blockIdx.x * 256;      // (PH 6: a workgroup takes every gridDim-th stretch of the list)
  // (a launch beside the band DPs -- to_late -- is a short chain the step waits for: ahead of the DPs' wavefronts wherever a SIMD holds both)
  if (bx.to_late) __builtin_amdgcn_s_setprio(3);
  // a read the 10-mers cannot vouch for goes on to the third launch instead of being given up
  const bool fine_on = (P == 1 || P == 2) && bx.cand2 != nullptr;

  struct Rd { int32_t i; int len2, s, l1, st; bool ok; };
  auto fetch = [&](int tid, DiagScan<NW>& sc) -> Rd {
    Rd r{0, 0, 0, 0, 0, false};
    const int64_t t = t0 + tid;
    if (t >= total) return r;
    r.i = in_list ? in_list[t] : (int32_t)t;
    if (!rs.sk[r.i]) return r;
    r.len2 = rs.len[r.i];
    r.st = rs.rc[r.i] ? 1 : 0;
    read_window(ref, rs.as[r.i], rs.ae[r.i], r.len2, &r.s, &r.l1);
    r.ok = true;
    return r;
  };
  auto load_planes = [&](const Rd& r, DiagScan<NW>& sc) -> bool {
    if (bx.rplanes) {
      if (bx.umax[r.i] < 0) return false;                       // a read with N
      const uint64_t* pl = bx.rplanes + (int64_t)r.i * 2 * bx.rplane_words;
      sc.set_read(pl, pl + bx.rplane_words, r.len2);
      return true;
    }
    return sc.load_read(rs.packed + rs.roff[r.i], r.len2);
  };
  // results of a planned read: finished (scripts follow), on to a list -- or, with to_fine, on to the third launch with its anchors
  auto emit = [&](const Rd& r, const BxPlan& bp, bool mark_open, bool to_fine, const BxAnchors& an) {
    if (r.ok || mark_open) {
      if (bp.mode == BX_DONE || bp.mode == BX_VALUES) {
        const int u = bx.umax ? bx.umax[r.i] : bx_umax(bx.tab.mrow, rs.packed + rs.roff[r.i], r.len2, r.st);
        const int expect = u - bp.b0;
        if (bp.mode == BX_DONE) {
          rs.score[r.i] = expect;
          rs.refstart[r.i] = r.s;
          rs.abr[r.i] = 0;
          rs.as[r.i] = r.s + bp.dstar;                   // src/mia_main.c:254-255
          rs.ae[r.i] = r.s + bp.dstar + r.len2 - 1;
          rs.status[r.i] = ST_DIAG;
          bin_of[r.i] = -4;
        } else bx.expect[r.i] = expect;
      }
      if (bp.mode == BX_VALUES || bp.mode == BX_TRACE) bx.plan[r.i] = bx_pack(bp);
      if (mark_open && bp.mode != BX_DONE) bin_of[r.i] = (bp.mode == BX_VALUES || bp.mode == BX_TRACE) ? bx.listed_mark : 0;
    }
    // (every read is marked by exactly one launch of the plan -- the one that finishes it: so it is listed exactly once)
    if (bx.open) bxl_append(bx.open, bx.open_n, mark_open && bp.mode != BX_DONE && bp.mode != BX_VALUES && bp.mode != BX_TRACE, r.i);
    // list appends and counters go through the block: one global atomic per block and list instead of one per wavefront
    const bool late = bx.to_late != 0;
    int which = bp.mode == BX_VALUES ? bx_class_of(bp.w) : (bp.mode == BX_TRACE ? (late ? 0 : BX_NCLS) + bx_class_of(bp.w) : -1);
    uint32_t rank = 0;
    if (which >= 0) rank = atomicAdd(&blk_cnt[which], 1u);
    if (to_fine) rank = atomicAdd(&blk_cnt[SLOT_FINE], 1u);
    if (bp.mode == BX_DONE) atomicAdd(&blk_cnt[2 * BX_NCLS], 1u);
    if (r.ok && bp.mode == BX_NONE && bp.b0 > 0 && bp.b0 < BXF_KINDS) atomicAdd(&blk_cnt[2 * BX_NCLS + 2 + bp.b0], 1u);
    __syncthreads();
    if (threadIdx.x < 2 * BX_NCLS && blk_cnt[threadIdx.x]) blk_base[threadIdx.x] = atomicAdd(bxc(bx.ctr, (late ? BXC_LATE2_0 : BXC_LIST0) + threadIdx.x), blk_cnt[threadIdx.x]);
    if (threadIdx.x == 2 * BX_NCLS && blk_cnt[2 * BX_NCLS]) atomicAdd(bxc(bx.ctr, BXC_DONE_PLAN), blk_cnt[2 * BX_NCLS]);
    if (threadIdx.x == 2 * BX_NCLS + 1 && blk_cnt[2 * BX_NCLS + 1]) atomicAdd(bxc(bx.ctr, BXC_SEEN), blk_cnt[2 * BX_NCLS + 1]);
    if (threadIdx.x > 2 * BX_NCLS + 2 && threadIdx.x < 2 * BX_NCLS + 2 + BXF_KINDS && blk_cnt[threadIdx.x])
      atomicAdd(bxc(bx.ctr, BXC_FAIL0 + (int)threadIdx.x - 2 * BX_NCLS - 2), blk_cnt[threadIdx.x]);
    if (threadIdx.x == SLOT_FINE && blk_cnt[SLOT_FINE]) blk_base[2 * BX_NCLS] = atomicAdd(bx.cand2_n, blk_cnt[SLOT_FINE]);
    __syncthreads();
    if (which >= 0) bx.lists[(int64_t)((late ? 3 * BX_NCLS : 0) + which) * bx.list_stride + blk_base[which] + rank] = r.i;
    if (to_fine) { BxCandRec rec; rec.i = r.i; rec.an = an; bx.cand2[blk_base[2 * BX_NCLS] + rank] = rec; }
    if (!bx.lazy_scripts) bx_diag_scripts(rs, __ballot(bp.mode == BX_DONE), r.i, bp.dstar, r.len2);
    __syncthreads();
    if (threadIdx.x <= SLOT_FINE) blk_cnt[threadIdx.x] = 0;
    __syncthreads();
  };
  const bool marks = !in_list || bx.mark_all != 0;
  if ((PH == 4 || PH == 5)) {
    // THE QUICK PLAN (bandx_body.h: bx_quick, bx_quick2).  A workgroup takes BX_QCH stretches of 256 reads, one read per thread and stretch,
    // through the one-diagonal form (the diagonal the read was aligned on before, asked first); the reads that leaves undecided -- one in
    // ten: the reads with an indel -- are collected and taken through the one-indel form by the block's first threads; then ONE round of
    // list reservations for all of them.  What the two forms leave -- one read in a hundred -- goes on qlist, the in_list of the full plan's
    // launch(es) behind this one; a read that is not plannable at all (strand unknown, an N in it, a window the plan does not take) is left
    // open here and now.
    // (The first version emitted per stretch of 256 as the other phases do -- four barriers and two returning atomics on the lists'
    // counters per block: 41 % of a wavefront's cycles were that wait, bx_quick itself 3 % (tools/plan_clk_probe.py) -- a counter that
    // four thousand blocks add to serves them one at a time, 20 ns each.  Tried and dropped: the full plan for the hundredth read inside
    // this launch, by the block's first threads (+ 59 us for the launch, the fork behind it that much later); the full plan's launches
    // beside the band DPs, their lists the late ones (the step waits as long for them there as in front of the fork).)
    constexpr int NL = 2 * BX_NCLS + 2, LQ = 2 * BX_NCLS, LO = 2 * BX_NCLS + 1;      // the lists a read can go on: the plan's ten, qlist, the open list
    // (per read of the block, in the launch's dynamic LDS -- its size goes with bx.qch: q_which = the list of read li; 255: none (finished
    // here, open without a list, or no such read); q_rank = its place among the block's entries of that list; q_cand = the undecided)
    extern __shared__ uint64_t lds_dyn[];
    uint16_t* const q_rank = reinterpret_cast<uint16_t*>(lds_dyn);
    uint16_t* const q_cand = q_rank + 256 * bx.qch;
    uint8_t* const q_which = reinterpret_cast<uint8_t*>(q_cand + 256 * bx.qch);
    __shared__ uint32_t q_cnt[NL], q_base[NL], q_ncand, q_fin, q_seen, q_fail[BXF_KINDS];
    if (threadIdx.x < NL) q_cnt[threadIdx.x] = 0;
    if (threadIdx.x < BXF_KINDS) q_fail[threadIdx.x] = 0;
    if (threadIdx.x == 0) { q_ncand = 0; q_fin = 0; q_seen = 0; }
    // PHASE 5: the reference's three planes in LDS (a mitochondrion's are 6.5 KB).  A seek is a dozen 8-byte loads at an address of the
    // lane's own -- every one of them sixty-four trips through the compute unit's one address unit --, the quick plan seeks twice per read,
    // and with the table walks gone those loads and the bitmaps' were what a wavefront waited for (tools/plan_clk_probe.py: half its cycles)
    uint64_t* const lds_planes = lds_dyn + bx_quick_lds_words(bx.qch);
    if ((PH == 5)) {
      const int pw = bx.plane_words;
      for (int k = threadIdx.x; k < pw; k += 256) { lds_planes[k] = rp.lo[k]; lds_planes[pw + k] = rp.hi[k]; lds_planes[2 * pw + k] = rp.ok[k]; }
      rp.lo = lds_planes; rp.hi = lds_planes + pw; rp.ok = lds_planes + 2 * pw;
    }
    __syncthreads();
    PLAN_CLK(0);
    const int64_t tq = (int64_t)blockIdx.x * (256 * bx.qch);
    auto fetch_at = [&](int64_t t) -> Rd {
      Rd r{0, 0, 0, 0, 0, false};
      if (t >= total) return r;
      r.i = (int32_t)t;
      if (!rs.sk[r.i]) return r;
      r.len2 = rs.len[r.i];
      r.st = rs.rc[r.i] ? 1 : 0;
      read_window(ref, rs.as[r.i], rs.ae[r.i], r.len2, &r.s, &r.l1);
      r.ok = true;
      return r;
    };
    // what a planned read leaves behind at once: its result (finished) or its plan (listed), its mark; returns its list, 255 for none
    auto write_plan = [&](const Rd& r, const BxPlan& bp) -> int {
      if (bp.mode == BX_DONE || bp.mode == BX_VALUES) {
        const int expect = bx.umax[r.i] - bp.b0;                         // (the quick plan runs on the context's own reads: umax is there)
        if (bp.mode == BX_DONE) {
          rs.score[r.i] = expect;
          rs.refstart[r.i] = r.s;
          rs.abr[r.i] = 0;
          rs.as[r.i] = r.s + bp.dstar;                   // src/mia_main.c:254-255
          rs.ae[r.i] = r.s + bp.dstar + r.len2 - 1;
          rs.status[r.i] = ST_DIAG;
          bin_of[r.i] = -4;
          return 255;
        }
        bx.expect[r.i] = expect;
      }
      bx.plan[r.i] = bx_pack(bp);
      bin_of[r.i] = bx.listed_mark;
      return bp.mode == BX_VALUES ? bx_class_of(bp.w) : BX_NCLS + bx_class_of(bp.w);
    };
    // a read the plan leaves open (given up on, not plannable, strand unknown): its mark, its reason, the open list if there is one
    auto leave_open = [&](int32_t i, bool counted, int why) -> int {
      bin_of[i] = 0;
      if (counted && why > 0 && why < BXF_KINDS) atomicAdd(&q_fail[why], 1u);
      return bx.open ? LO : 255;
    };
    auto place = [&](int li, int which) {
      q_which[li] = (uint8_t)which;
      if (which < NL) q_rank[li] = (uint16_t)atomicAdd(&q_cnt[which], 1u);
    };
    const int lane = threadIdx.x & 63;
    for (int c = 0; c < bx.qch; c++) {
      const int li = c * 256 + (int)threadIdx.x;
      const int64_t t = tq + li;
      DiagScan<NW> sc;
      Rd r = fetch_at(t);
      BxPlan bp;
      bp.mode = BX_NONE; bp.d0 = 0; bp.w = 1; bp.dstar = 0; bp.b0 = 0; bp.edge = 0;
      bool planned = false;
      const bool plannable = r.ok && bx_plannable(rp, ko, n_ref, r.s, r.l1, r.len2);
      const bool can = plannable && load_planes(r, sc);
      PLAN_CLK(1);
      if (can) planned = bx_quick<NW>(sc, rp, ko, bx.kb, r.s, r.l1, r.len2, r.st, rs.as[r.i] - r.s, T, &bp);
      PLAN_CLK(2);
      int which = 255;
      if (planned) which = write_plan(r, bp);
      else if (can) { q_cand[atomicAdd(&q_ncand, 1u)] = (uint16_t)li; which = 254; }       // (254: the one-indel form decides)
      else if (t < total) which = leave_open((int32_t)t, r.ok, plannable ? BXF_READ : BXF_WINDOW);
      place(li, which);
      const unsigned long long fm = __ballot(planned && bp.mode == BX_DONE), sm = __ballot(r.ok);
      if (lane == 0 && fm) atomicAdd(&q_fin, (uint32_t)__popcll(fm));
      if (lane == 0 && sm) atomicAdd(&q_seen, (uint32_t)__popcll(sm));
      if (!bx.lazy_scripts) bx_diag_scripts(rs, fm, r.i, bp.dstar, r.len2);
      PLAN_CLK(3);
    }
    __syncthreads();
    // ... the one-indel form, by as many threads as there are undecided reads
    for (uint32_t k = threadIdx.x; k < ((q_ncand + 63u) & ~63u); k += 256) {            // (whole wavefronts: the ballots below)
      const bool mine = k < q_ncand;
      const int li = mine ? (int)q_cand[k] : 0;
      DiagScan<NW> sc;
      Rd r{0, 0, 0, 0, 0, false};
      BxPlan bp;
      bp.mode = BX_NONE; bp.d0 = 0; bp.w = 1; bp.dstar = 0; bp.b0 = 0; bp.edge = 0;
      bool planned = false;
      if (mine) {
        r = fetch_at(tq + li);
        if (r.ok && load_planes(r, sc)) planned = bx_quick2<NW>(sc, rp, ko, bx.kb, r.s, r.l1, r.len2, r.st, rs.as[r.i] - r.s, T, &bp);
        place(li, planned ? write_plan(r, bp) : LQ);
      }
      const unsigned long long fm = __ballot(planned && bp.mode == BX_DONE);
      if (lane == 0 && fm) atomicAdd(&q_fin, (uint32_t)__popcll(fm));
      if (!bx.lazy_scripts) bx_diag_scripts(rs, fm, r.i, bp.dstar, r.len2);
    }
    PLAN_CLK(4);
    __syncthreads();
    // one reservation per list and block, then every read to its place
    if (threadIdx.x < NL && q_cnt[threadIdx.x])
      q_base[threadIdx.x] = atomicAdd(threadIdx.x == LQ ? bx.qlist_n : (threadIdx.x == LO ? bx.open_n : bxc(bx.ctr, BXC_LIST0 + threadIdx.x)), q_cnt[threadIdx.x]);
    if (threadIdx.x == NL && q_fin) atomicAdd(bxc(bx.ctr, BXC_DONE_PLAN), q_fin);
    if (threadIdx.x == NL + 1 && q_seen) atomicAdd(bxc(bx.ctr, BXC_SEEN), q_seen);
    if (threadIdx.x == NL + 2 && q_ncand) atomicAdd(bxc(bx.ctr, BXC_FAIL0), q_ncand);      // (statistics: reads the one-diagonal form left to the one-indel form)
    if (threadIdx.x >= 32 && threadIdx.x < 32 + BXF_KINDS && threadIdx.x > 32 && q_fail[threadIdx.x - 32]) atomicAdd(bxc(bx.ctr, BXC_FAIL0 + (int)threadIdx.x - 32), q_fail[threadIdx.x - 32]);
    __syncthreads();
    PLAN_CLK(5);
    for (int c = 0; c < bx.qch; c++) {
      const int li = c * 256 + (int)threadIdx.x, which = q_which[li];
      if (which >= NL) continue;
      const uint32_t at = q_base[which] + q_rank[li];
      if (which == LQ) bx.qlist[at] = (int32_t)(tq + li);
      else if (which == LO) bx.open[at] = (int32_t)(tq + li);
      else bx.lists[(int64_t)which * bx.list_stride + at] = (int32_t)(tq + li);
    }
    PLAN_CLK(6);
    PLAN_CLK_FLUSH;
    return;
  }
  for (;;) {        // (one pass -- except PH 6: a stretch of the list per pass)
  if (P < 2) {
    PLAN_CLK(0);
    DiagScan<NW> sc;
    Rd r = fetch((int)threadIdx.x, sc);
    BxPlan bp;
    bp.mode = BX_NONE; bp.d0 = 0; bp.w = 1; bp.dstar = 0; bp.b0 = 0; bp.edge = 0;
    BxAnchors an{};
    bool waits = false, to_fine = false;
#ifdef MIA_HIP_ALT_PATHS
    if (P == 1 && pclk_on_) {
      // the same stretches one after the other over the whole wavefront (a clock between them): what the kernel does, in four steps
      const bool pl = r.ok && bx_plannable(rp, ko, n_ref, r.s, r.l1, r.len2);
      PLAN_CLK(1);
      const bool lp = pl && load_planes(r, sc);
      PLAN_CLK(2);
      if (lp) bx_anchors<NW>(sc, ko, reinterpret_cast<const uint32_t*>(rs.packed + rs.roff[r.i]), r.s, r.l1, r.len2, r.st, T, &an);
      PLAN_CLK(3);
      if (r.ok) {
        bp.b0 = !pl ? BXF_WINDOW : (!lp ? BXF_READ : an.fail);
        if (lp && !an.fail) {
          bool later = an.d_first != an.d_last;
          if (!later) {
            bx_finish<NW, 1>(sc, rp, an, r.s, r.l1, r.len2, r.st, T, &bp);
            if (bp.mode == BX_NONE && (bp.b0 == BXF_BUDGET || bp.b0 == BXF_WIDTH) && an.a_lo == an.a_hi) { later = true; an.rescue = bp.b0; }
            else if (fine_on && bx_wants_fine(bp) && a_hi_ok(an, T)) { an.rescue = bp.b0; to_fine = true; waits = true; bp.mode = BX_NONE; bp.b0 = 0; }
          }
          if (later) {
            const int slot = atomicAdd(&n_cand, 1);
            cand_an[slot] = an;
            cand_tid[slot] = (uint8_t)threadIdx.x;
            waits = true;
            bp.mode = BX_NONE; bp.b0 = 0;
          }
        }
      }
      PLAN_CLK(4);
    } else
#endif
    if (r.ok) {
      bp.b0 = BXF_WINDOW;
      if (bx_plannable(rp, ko, n_ref, r.s, r.l1, r.len2)) {
        bp.b0 = BXF_READ;
        if (load_planes(r, sc)) {
          bx_anchors<NW>(sc, ko, reinterpret_cast<const uint32_t*>(rs.packed + rs.roff[r.i]), r.s, r.l1, r.len2, r.st, T, &an);
          bp.b0 = an.fail;
          if (!an.fail) {
            bool later = an.d_first != an.d_last;
            if (!later) {
              bx_finish<NW, 1>(sc, rp, an, r.s, r.l1, r.len2, r.st, T, &bp);
              // an end indel, perhaps (bx_rescue): with the reads of two diagonals, for the block's first threads
              if (bp.mode == BX_NONE && (bp.b0 == BXF_BUDGET || bp.b0 == BXF_WIDTH) && an.a_lo == an.a_hi) { later = true; an.rescue = bp.b0; }
              else if (fine_on && bx_wants_fine(bp) && a_hi_ok(an, T)) { an.rescue = bp.b0; to_fine = true; waits = true; bp.mode = BX_NONE; bp.b0 = 0; }
            }
            if (later) {
              const int slot = atomicAdd(&n_cand, 1);
              cand_an[slot] = an;
              cand_tid[slot] = (uint8_t)threadIdx.x;
              waits = true;
              bp.mode = BX_NONE; bp.b0 = 0;
            }
          }
        }
      }
    }
    const unsigned long long sm = __ballot(r.ok && !bx.mark_all);      // (mark_all: the quick plan in front has seen -- and counted -- every read of this list)
    if ((threadIdx.x & 63) == 0 && sm) atomicAdd(&blk_cnt[2 * BX_NCLS + 1], (uint32_t)__popcll(sm));
    // (a waiting read's mark is written by the launch that finishes it)
    Rd rr = r;
    if (waits) rr.ok = false;
    emit(rr, bp, marks && t0 + threadIdx.x < total && !waits, to_fine, an);
    PLAN_CLK(5);
    if (bx.early && !in_list && t0 + threadIdx.x < total) bx.early[t0 + threadIdx.x] = bp.mode == BX_DONE ? 1 : 0;      // (every read passes here once)
    if (P == 1) {                                  // hand the waiting reads over: one reservation per block
      __shared__ uint32_t cand_base;
      __syncthreads();
      if (threadIdx.x == 0 && n_cand) cand_base = atomicAdd(bx.cand_n, (uint32_t)n_cand);
      __syncthreads();
      if ((int)threadIdx.x < n_cand) {
        BxCandRec rec;
        rec.i = (int32_t)(in_list ? in_list[t0 + cand_tid[threadIdx.x]] : t0 + cand_tid[threadIdx.x]);
        rec.an = cand_an[threadIdx.x];
        bx.cand[cand_base + threadIdx.x] = rec;
      }
      PLAN_CLK(6);
      PLAN_CLK_FLUSH;
      return;
    }
  }
  __syncthreads();
  if (!((P == 0 && n_cand == 0) || (bx.dbg & 32u)) && !(P == 3 && !bx.cand2)) {      // (MIA_HIP_BX_DEBUG=32, profiling only: no second phase)
  // the candidates: this block's own (P 0), or a stretch of the list per step (P 2, 3; every thread loops alike: emit has barriers)
  constexpr bool listed = P >= 2;
  const BxCandRec* const list = P == 3 ? bx.cand2 : bx.cand;
  const int64_t n_list = P == 3 ? (int64_t)*bx.cand2_n : (P == 2 ? (int64_t)*bx.cand_n : 0);
  constexpr int per = P == 3 ? 256 / BX_FINE_LANES : 256;            // reads per block and step
  for (int64_t c0 = listed ? (int64_t)blockIdx.x * per : 0; listed ? c0 < n_list : c0 == 0; c0 += listed ? (int64_t)gridDim.x * per : 1) {
    DiagScan<NW> sc;
    Rd r{0, 0, 0, 0, 0, false};
    BxPlan bp;
    bp.mode = BX_NONE; bp.d0 = 0; bp.w = 1; bp.dstar = 0; bp.b0 = 0; bp.edge = 0;
    const int slot = P == 3 ? (int)threadIdx.x / BX_FINE_LANES : (int)threadIdx.x, u = P == 3 ? (int)threadIdx.x % BX_FINE_LANES : 0;
    const bool mine = listed ? c0 + slot < n_list : (int)threadIdx.x < n_cand;
    bool to_fine = false;
    BxAnchors an{};
    if (mine) {
      if (listed) {
        const BxCandRec rec = list[c0 + slot];
        an = rec.an;
        r.i = rec.i;
        r.len2 = rs.len[r.i];
        r.st = rs.rc[r.i] ? 1 : 0;
        read_window(ref, rs.as[r.i], rs.ae[r.i], r.len2, &r.s, &r.l1);
        r.ok = true;                                // (only reads that passed the first launch's tests are on the list)
      } else {
        r = fetch(cand_tid[threadIdx.x], sc);
        an = cand_an[threadIdx.x];
      }
      load_planes(r, sc);
      if (P == 3) {
        // the fine blocks (bx_fine_anchors): more blocks, more budget; the reason the 10-mers gave stands if they do not help either
        const int why = an.rescue;
        an.rescue = 0;
        const bool usable = bx_fine_usable<NW>(r.l1, r.len2, r.st, T);
        uint64_t in = 0, out = 0;
        int a_lo = an.a_lo, a_hi = an.a_hi;
        uint64_t ge[3 * BX_FDW];
#pragma unroll
        for (int q = 0; q < 3 * BX_FDW; q++) ge[q] = 0;
        const bool counted = BX_FINE_LANES == 1 && r.l1 + r.len2 <= 64 * BX_FDW;      // (per-diagonal counts: one lane sees all blocks)
        if (usable) {
          // this lane's share of the blocks (u, u + BX_FINE_LANES, ...) against the window's column planes
          BxWinPlanes wp;
          bx_win_planes(rp, r.s, r.l1, &wp);
          bx_fine_scan<NW>(sc, wp, r.l1, r.len2, an.a_lo - BX_FINE_RADIUS, an.a_hi + BX_FINE_RADIUS, u, BX_FINE_LANES, &in, &out, &a_lo, &a_hi, counted ? ge : nullptr);
        }
        // (all lanes of a read are here together: `mine`, the record and `usable` are the same for them)
#pragma unroll
        for (int o = 1; o < BX_FINE_LANES; o <<= 1) {
          in |= __shfl_xor(in, o); out |= __shfl_xor(out, o);
          const int lo2 = __shfl_xor(a_lo, o), hi2 = __shfl_xor(a_hi, o);
          a_lo = lo2 < a_lo ? lo2 : a_lo;
          a_hi = hi2 > a_hi ? hi2 : a_hi;
        }
        if (u == 0) {
          if (usable) {
            bx_fine_sums<NW>(in, out, a_lo, a_hi, r.len2, r.st, T, &an, counted ? ge : nullptr);
            bx_finish<NW>(sc, rp, an, r.s, r.l1, r.len2, r.st, T, &bp);
          }
          if (bp.mode == BX_NONE && bp.b0 != BXF_WIDTH) bp.b0 = why;
        } else r.ok = false;                        // (the read's other lanes have nothing to report)
      } else {
        if (an.rescue && !bx_rescue<NW>(sc, rp, an, r.s, r.l1, r.len2)) bp.b0 = an.rescue;      // (the reason it was not planned stands)
        else bx_finish<NW, 2>(sc, rp, an, r.s, r.l1, r.len2, r.st, T, &bp);
        if (fine_on && bx_wants_fine(bp) && a_hi_ok(an, T)) { an.rescue = bp.b0; to_fine = true; bp.mode = BX_NONE; bp.b0 = 0; }
      }
    }
    Rd rr = r;
    if (to_fine) rr.ok = false;
    emit(rr, bp, marks && rr.ok, to_fine, an);
    if (bx.early && rr.ok && bp.mode == BX_DONE) bx.early[r.i] = 1;
  }
  }
  if (PH != 6) return;
  t0 += (int64_t)gridDim.x * 256;
  if (t0 >= total) return;
  __syncthreads();
  if (threadIdx.x == 0) n_cand = 0;
  __syncthreads();
  }
}

// the chunk a wavefront takes next: classes in descending width (the long chunks first), 64 reads each.  The entries of
// list `list0 + c` from *lo (counter index lo_ctr + c, or 0 if lo_ctr < 0) up to *hi (counter index hi_ctr + c).
struct BxChunk { int cls; uint32_t first, count; };
__device__ __forceinline__ bool bx_next_chunk(const BxDev& bx, int cursor, int lo_ctr, int hi_ctr, BxChunk* out) {
  uint32_t chunk = 0;
  if ((threadIdx.x & 63) == 0) chunk = atomicAdd(bxc(bx.ctr, cursor), 1u);
  chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
  for (int c = BX_NCLS - 1; c >= 0; c--) {
    const uint32_t lo = lo_ctr < 0 ? 0u : *bxc(bx.ctr, lo_ctr + c), hi = *bxc(bx.ctr, hi_ctr + c);
    const uint32_t cnt = hi > lo ? hi - lo : 0u, nch = (cnt + 63u) >> 6;
    if (chunk < nch) { out->cls = c; out->first = lo + chunk * 64u; out->count = hi; return true; }
    chunk -= nch;
  }
  return false;
}
struct BxRead { int32_t i; int len2, s, l1, d0, jstar, st; bool edge; const uint32_t* rw; };
__device__ __forceinline__ BxRead bx_load(const ReadSet& rs, const RefInfo& ref, const BxDev& bx, int32_t i) {
  BxRead r;
  r.i = i;
  r.len2 = rs.len[i];
  read_window(ref, rs.as[i], rs.ae[i], r.len2, &r.s, &r.l1);
  const uint32_t w = bx.plan[i];
  r.d0 = (int)(w & 0x7FFu) - 512;
  r.jstar = (int)((w >> 11) & 63u);
  r.edge = ((w >> 17) & 1u) != 0;
  r.st = rs.rc[i] ? 1 : 0;
  r.rw = reinterpret_cast<const uint32_t*>(rs.packed + rs.roff[i]);
  return r;
}

constexpr int BX_SLAB_ROW_WORDS = 64 * (32 / 4);      // trace slab row of the one-lane kernel: 64 lanes x 32 bytes
#ifdef MIA_HIP_ALT_PATHS      // the one-lane statement of the band DPs: MIA_HIP_NO_LANES=1 of the alt build (and, as plain functions, the CPU emulation tests)
__global__ __launch_bounds__(256) void k_bx_values(ReadSet rs, RefInfo ref, BxDev bx, int32_t* bin_of) {
  __shared__ int32_t sub_lds[BX_SUB_WORDS];
  for (int k = threadIdx.x; k < BX_SUB_WORDS; k += 256) sub_lds[k] = bx.tab.sub[k];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t done = 0;
  BxChunk ch;
  while (bx_next_chunk(bx, BXC_CUR_VALUES, -1, BXC_LIST0, &ch)) {
    const uint32_t t = ch.first + lane;
    const bool live = t < ch.count;
    int best = BX_NEG, bj = -1;
    BxRead r{};
    if (live) r = bx_load(rs, ref, bx, bx.lists[(int64_t)ch.cls * bx.list_stride + t]);
    const bool edge = __ballot(live && r.edge) != 0ull;       // one form of the recurrence per wavefront
    if (live) {
      const int32_t* sub = sub_lds + r.st * (31 * 4 * BX_SUB_ROW);
#define BXV(W)                                                                                                \
  if (edge) bx_values<W, true>(bx.refnib, r.s, r.l1, r.rw, r.len2, r.d0, sub, &best, &bj);                  \
  else bx_values<W, false>(bx.refnib, r.s, r.l1, r.rw, r.len2, r.d0, sub, &best, &bj)
      switch (ch.cls) {
        case 0: BXV(8); break;
        case 1: BXV(16); break;
        case 2: BXV(24); break;
        default: BXV(32); break;
      }
#undef BXV
    }
    const bool ok = live && bj >= 0 && bj == r.jstar && best == bx.expect[r.i];
    if (ok) {
      const int dstar = r.d0 + r.jstar;
      rs.score[r.i] = best;
      rs.refstart[r.i] = r.s;
      rs.abr[r.i] = 0;
      rs.as[r.i] = r.s + dstar;                   // src/mia_main.c:254-255
      rs.ae[r.i] = r.s + dstar + r.len2 - 1;
      rs.status[r.i] = ST_DIAG;
      bin_of[r.i] = -4;
      done++;
    }
    // (a read whose best score is not the plan's diagonal's has a gap or a soft clip: it stays open -- bin_of = 0 -- and the
    // planner hands it to the full-window kernels; a second trace launch for these few would cost a whole chunk's latency)
    if (!bx.lazy_scripts) bx_diag_scripts(rs, __ballot(ok), r.i, r.d0 + r.jstar, r.len2);
  }
  for (int o = 32; o; o >>= 1) done += __shfl_xor(done, o);
  if (lane == 0 && done) atomicAdd(bxc(bx.ctr, BXC_DONE_VALUES), done);
}

// trace slab of a wavefront: [row][lane][BX_MAXW bytes], so that the 64 stores of a row are one stretch
// trace slab of a wavefront: [row][lane][W bytes], so that the 64 stores of a row are one stretch
__global__ __launch_bounds__(256) void k_bx_trace(ReadSet rs, RefInfo ref, BxDev bx, uint32_t* slabs, int64_t slab_words, int32_t* bin_of) {
  __shared__ int32_t sub_lds[BX_SUB_WORDS];
  for (int k = threadIdx.x; k < BX_SUB_WORDS; k += 256) sub_lds[k] = bx.sub256[k];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t* slab = slabs + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * slab_words;
  uint32_t done = 0;
  BxChunk ch;
  while (bx_next_chunk(bx, BXC_CUR_TRACE, -1, BXC_LIST0 + BX_NCLS, &ch)) {
    const uint32_t t = ch.first + lane;
    const bool live = t < ch.count;
    BxRead r{};
    if (live) r = bx_load(rs, ref, bx, bx.lists[(int64_t)(BX_NCLS + ch.cls) * bx.list_stride + t]);
    const bool edge = __ballot(live && r.edge) != 0ull;
    bool got = false;
    BxResult res{};
    if (live) {
      const int32_t* sub = sub_lds + r.st * (31 * 4 * BX_SUB_ROW);
      int16_t* cols = rs.cols + (int64_t)r.i * rs.stride;
#define BXT(W)                                                                                                                              \
  got = edge ? bx_trace<W, true>(bx.refnib, r.s, r.l1, r.rw, r.len2, r.d0, sub, slab + lane * (W / 4), 64 * (W / 4), cols, &res)          \
             : bx_trace<W, false>(bx.refnib, r.s, r.l1, r.rw, r.len2, r.d0, sub, slab + lane * (W / 4), 64 * (W / 4), cols, &res)
      switch (ch.cls) {
        case 0: BXT(8); break;
        case 1: BXT(16); break;
        case 2: BXT(24); break;
        default: BXT(32); break;
      }
#undef BXT
    }
    if (got) {                                   // (not got: the reference's index-0 quirk -- the full-window kernels take the read)
      rs.score[r.i] = res.score;
      rs.refstart[r.i] = r.s;
      rs.abr[r.i] = (int16_t)res.abr;
      rs.as[r.i] = res.abc + r.s;                // src/mia_main.c:254-255
      rs.ae[r.i] = res.aec + r.s;
      rs.status[r.i] = res.gaps == 0 ? ST_DIAG : (res.gaps == 1 ? (ST_ONEGAP | (res.gap_desc << 8)) : ST_OK);
      bin_of[r.i] = -4;
      done++;
    }
  }
  for (int o = 32; o; o >>= 1) done += __shfl_xor(done, o);
  if (lane == 0 && done) atomicAdd(bxc(bx.ctr, BXC_DONE_TRACE), done);
}
#endif

// ---- the same two kernels with a read spread over W / 8 lanes (bandx_lanes.h) ---------------------------------------------
// A chunk is bxl_chunk_reads(class) reads -- 64, 32, 20, 16 -- so that every wavefront walks eight cells per row and lane.
// Chunks are dealt out statically: wavefront w of the grid takes chunks w, w + waves, w + 2 waves ... (classes in descending
// width, so every wavefront gets its share of the long ones).  A shared cursor looks more flexible, but thousands of
// wavefronts adding to ONE word are served one after the other by the L2 (~12 ns each): with a grid as large as the chunk
// list that was 60-75 us of every launch, more than the DP itself took (tools/bxl_probe.py).
// part (BX_PART_*; HEAD / TAIL only for the plan's own lists, whose counters are the first 2 BX_NCLS: hi_ctr = BXC_LIST0 or BXC_LIST0 + BX_NCLS)
__device__ __forceinline__ bool bxl_chunk_at(const BxDev& bx, uint32_t chunk, int hi_ctr, BxChunk* out, int part = BX_PART_ALL) {
  for (int c = BX_NCLS - 1; c >= 0; c--) {
    const uint32_t live = *bxc(bx.ctr, hi_ctr + c);
    const uint32_t mark = part == BX_PART_ALL ? 0u : bx.snap[hi_ctr - BXC_LIST0 + c];
    const uint32_t lo = part == BX_PART_TAIL ? mark : 0u, hi = part == BX_PART_HEAD ? mark : live;
    const uint32_t cnt = hi > lo ? hi - lo : 0u, per = (uint32_t)bxl_chunk_reads(c), nch = (cnt + per - 1u) / per;
    if (chunk < nch) { out->cls = c; out->first = lo + chunk * per; out->count = hi; return true; }
    chunk -= nch;
  }
  return false;
}
__global__ void k_bx_snap(const uint32_t* ctr, uint32_t* snap) {
  if (blockIdx.x == 0 && threadIdx.x < 2 * BX_NCLS) snap[threadIdx.x] = *bxc(ctr, BXC_LIST0 + (int)threadIdx.x);
}
// what the wavefronts of a workgroup finished, added up in LDS: one atomic per workgroup on the statistics word
__device__ __forceinline__ void bxl_count_done(uint32_t done, uint32_t* lds_word, uint32_t* global_word) {
  for (int o = 32; o; o >>= 1) done += __shfl_xor(done, o);
  if ((threadIdx.x & 63) == 0 && done) atomicAdd(lds_word, done);
  __syncthreads();
  if (threadIdx.x == 0 && *lds_word) atomicAdd(global_word, *lds_word);
}
// which read of the chunk this lane works on, and its place among that read's lanes; false: an idle lane
template <int LPR>
__device__ __forceinline__ bool bxl_place(const BxChunk& ch, uint32_t* t, int* u) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, slot = l16 / LPR;
  *u = l16 - slot * LPR;
  *t = ch.first + (uint32_t)((lane >> 4) * bxl_reads_per_row<LPR>() + slot);
  return slot < bxl_reads_per_row<LPR>() && *t < ch.count;
}

template <int LPR>
__device__ __forceinline__ uint32_t bxl_values_chunk(const ReadSet& rs, const RefInfo& ref, const BxDev& bx, int32_t* bin_of, const BxChunk& ch, const int32_t* sub_lds) {
  uint32_t t;
  int u;
  const bool live = bxl_place<LPR>(ch, &t, &u);
  int best = BX_NEG, bj = -1;
  BxRead r{};
  if (live) r = bx_load(rs, ref, bx, bx.lists[(int64_t)ch.cls * bx.list_stride + t]);
  const bool edge = __ballot(live && r.edge) != 0ull;       // one form of the recurrence per wavefront
  if (live) {
    const int32_t* sub = sub_lds + r.st * (31 * 4 * BX_SUB_ROW);
    const int rows = (bx.dbg & 2u) ? 1 : r.len2;
    if (edge) bxl_values<LPR, true>(bx.refnib, r.s, r.l1, r.rw, rows, r.d0, sub, u, &best, &bj);
    else if (bx.dbg & 64u) bxl_values<LPR, false>(bx.refnib, r.s, r.l1, r.rw, rows, r.d0, sub, u, &best, &bj);      // (MIA_HIP_BX_DEBUG=64: the aged form)
    else bxl_values_star<LPR>(bx.refnib, r.s, r.rw, rows, r.d0, sub, u, bx.umax ? bx.umax[r.i] - bx.expect[r.i] : -1, &best, &bj);
  }
  const bool ok = live && u == 0 && bj >= 0 && bj == r.jstar && best == bx.expect[r.i];
  if (ok) {
    const int dstar = r.d0 + r.jstar;
    rs.score[r.i] = best;
    rs.refstart[r.i] = r.s;
    rs.abr[r.i] = 0;
    rs.as[r.i] = r.s + dstar;                   // src/mia_main.c:254-255
    rs.ae[r.i] = r.s + dstar + r.len2 - 1;
    rs.status[r.i] = ST_DIAG;
    bin_of[r.i] = -4;
  }
  if (!bx.lazy_scripts) bx_diag_scripts(rs, __ballot(ok), r.i, r.d0 + r.jstar, r.len2);
  // a read whose best score is not the plan's diagonal's has a gap or a soft clip: on to the late trace list of its class
  // (k_bxl_trace runs once more, behind this kernel); with listed_mark == 0 it simply stays open for the planner
  if (bx.listed_mark != 0)
    bxl_append(bx.lists + (int64_t)(2 * BX_NCLS + ch.cls) * bx.list_stride, bxc(bx.ctr, BXC_LATE0 + ch.cls), live && u == 0 && !ok, r.i);
  return ok ? 1u : 0u;
}

__global__ __launch_bounds__(256, 4) void k_bxl_values(ReadSet rs, RefInfo ref, BxDev bx, int32_t* bin_of, int32_t part) {
  // the step's chain runs through this kernel and the late trace behind it, while k_bxl_trace beside them has slack: their
  // wavefronts go first wherever a SIMD has both to choose from (MIA_HIP_BX_DEBUG=128: all at priority 0)
  if (!(bx.dbg & 128u)) __builtin_amdgcn_s_setprio(2);
  __shared__ int32_t sub_lds[BX_SUB_WORDS];
  for (int k = threadIdx.x; k < BX_SUB_WORDS; k += 256) sub_lds[k] = bx.tab.sub[k];
  __syncthreads();
  __shared__ uint32_t done_wg;
  if (threadIdx.x == 0) done_wg = 0;
  __syncthreads();
  uint32_t done = 0;
  BxChunk ch;
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), waves = gridDim.x * 4u;
  for (uint32_t chunk = wave; bxl_chunk_at(bx, chunk, BXC_LIST0, &ch, part); chunk += waves) {
    switch (ch.cls) {
      case 0: done += bxl_values_chunk<1>(rs, ref, bx, bin_of, ch, sub_lds); break;
      case 1: done += bxl_values_chunk<2>(rs, ref, bx, bin_of, ch, sub_lds); break;
      case 2: done += bxl_values_chunk<3>(rs, ref, bx, bin_of, ch, sub_lds); break;
      case 3: done += bxl_values_chunk<4>(rs, ref, bx, bin_of, ch, sub_lds); break;
      default: done += bxl_values_chunk<8>(rs, ref, bx, bin_of, ch, sub_lds); break;
    }
  }
  bxl_count_done(done, &done_wg, bxc(bx.ctr, BXC_DONE_VALUES));
}

template <int LPR>
__device__ __forceinline__ uint32_t bxl_trace_chunk(const ReadSet& rs, const RefInfo& ref, const BxDev& bx, int32_t* bin_of, const BxChunk& ch, const int32_t* sub_lds,
                                                    uint32_t* slab, int list0) {
  uint32_t t;
  int u;
  const bool live = bxl_place<LPR>(ch, &t, &u);
  BxRead r{};
  if (live) r = bx_load(rs, ref, bx, bx.lists[(int64_t)(list0 + ch.cls) * bx.list_stride + t]);
  const bool edge = __ballot(live && r.edge) != 0ull;
  bool got = false;
  BxResult res{};
  if (live) {
    const int32_t* sub = sub_lds + r.st * (31 * 4 * BX_SUB_ROW);
    int16_t* cols = rs.cols + (int64_t)r.i * rs.stride;
    const int rows = (bx.dbg & 2u) ? 1 : r.len2;
    got = edge ? bxl_trace<LPR, true>(bx.refnib, r.s, r.l1, r.rw, rows, r.d0, sub, u, (int)(threadIdx.x & 63), slab, cols, &res, bx.lazy_scripts != 0, (bx.dbg & 1u) != 0)
               : bxl_trace<LPR, false>(bx.refnib, r.s, r.l1, r.rw, rows, r.d0, sub, u, (int)(threadIdx.x & 63), slab, cols, &res, bx.lazy_scripts != 0, (bx.dbg & 1u) != 0);
  }
  if (got) {                                   // (not got on a read's first lane: the reference's index-0 quirk -- the full-window kernels take the read)
    rs.score[r.i] = res.score;
    rs.refstart[r.i] = r.s;
    rs.abr[r.i] = (int16_t)res.abr;
    rs.as[r.i] = res.abc + r.s;                // src/mia_main.c:254-255
    rs.ae[r.i] = res.aec + r.s;
    rs.status[r.i] = res.gaps == 0 ? ST_DIAG : (res.gaps == 1 ? (ST_ONEGAP | (res.gap_desc << 8)) : ST_OK);
    bin_of[r.i] = -4;
  }
  // not finished on a read's first lane: the reference's index-0 quirk (bandx_lanes.h).  With the planner running beside
  // this kernel the read goes on the retry list (a window kernel behind everything takes it); otherwise it is open again.
  if (bx.listed_mark != 0) bxl_append(bx.retry, reinterpret_cast<uint32_t*>(bx.retry_n), live && u == 0 && !got && !(bx.dbg & 3u), r.i);
  return got ? 1u : 0u;
}

// trace slab of a wavefront: [row][lane][2 words]
constexpr int BXL_SLAB_ROW_WORDS = 128;
// list0 / ctr0: BX_NCLS / BXC_LIST0 + BX_NCLS for the plan's trace lists, 2 BX_NCLS / BXC_LATE0 for the values DP's left-overs
__device__ __forceinline__ void bxl_trace_grid(const ReadSet& rs, const RefInfo& ref, const BxDev& bx, uint32_t* slabs, int64_t slab_words, int32_t* bin_of, int list0, int ctr0,
                                               int part = BX_PART_ALL) {
  __shared__ int32_t sub_lds[BX_SUB_WORDS];
  for (int k = threadIdx.x; k < BX_SUB_WORDS; k += 256) sub_lds[k] = bx.sub256[k];
  __syncthreads();
  uint32_t* slab = slabs + ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * slab_words;       // (uniform: scalar base + 32-bit lane offsets)
  __shared__ uint32_t done_wg;
  if (threadIdx.x == 0) done_wg = 0;
  __syncthreads();
  uint32_t done = 0;
  BxChunk ch;
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), waves = gridDim.x * 4u;
  for (uint32_t chunk = wave; bxl_chunk_at(bx, chunk, ctr0, &ch, part); chunk += waves) {
    switch (ch.cls) {
      case 0: done += bxl_trace_chunk<1>(rs, ref, bx, bin_of, ch, sub_lds, slab, list0); break;
      case 1: done += bxl_trace_chunk<2>(rs, ref, bx, bin_of, ch, sub_lds, slab, list0); break;
      case 2: done += bxl_trace_chunk<3>(rs, ref, bx, bin_of, ch, sub_lds, slab, list0); break;
      case 3: done += bxl_trace_chunk<4>(rs, ref, bx, bin_of, ch, sub_lds, slab, list0); break;
      default: done += bxl_trace_chunk<8>(rs, ref, bx, bin_of, ch, sub_lds, slab, list0); break;
    }
  }
  bxl_count_done(done, &done_wg, bxc(bx.ctr, BXC_DONE_TRACE));
}
// (two names for one body: a profile tells the plan's lists and the values DP's left-overs apart)
__global__ __launch_bounds__(256, 4) void k_bxl_trace(ReadSet rs, RefInfo ref, BxDev bx, uint32_t* slabs, int64_t slab_words, int32_t* bin_of, int32_t part) {
  bxl_trace_grid(rs, ref, bx, slabs, slab_words, bin_of, BX_NCLS, BXC_LIST0 + BX_NCLS, part);
}
__global__ __launch_bounds__(256, 4) void k_bxl_trace_late(ReadSet rs, RefInfo ref, BxDev bx, uint32_t* slabs, int64_t slab_words, int32_t* bin_of) {
  if (!(bx.dbg & 128u)) __builtin_amdgcn_s_setprio(3);
  bxl_trace_grid(rs, ref, bx, slabs, slab_words, bin_of, 2 * BX_NCLS, BXC_LATE0);
}
__global__ __launch_bounds__(256, 4) void k_bxl_trace_late2(ReadSet rs, RefInfo ref, BxDev bx, uint32_t* slabs, int64_t slab_words, int32_t* bin_of) {
  if (!(bx.dbg & 128u)) __builtin_amdgcn_s_setprio(3);
  bxl_trace_grid(rs, ref, bx, slabs, slab_words, bin_of, 3 * BX_NCLS, BXC_LATE2_0);
}

}  // namespace mia
