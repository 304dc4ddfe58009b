// mia_hip.hip -- libmia_hip.so: C ABI (include/mia_hip.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mia_hip.h"
#include "mia_comm.h"
#include "mia_consensus_kernels.h"
#include "mia_kernels.h"
#include "bandx_kernels.h"
#include "mia_pass1_kernels.h"
#include "mia_myers_kernels.h"
#include "mia_trim_kernels.h"
#include "mia_peak_kernels.h"
#include "mia_iter_kernels.h"

using namespace mia;

// timed stages (HIP events on the context's stream around the kernel launches of that kind)
enum Stage { STG_TRACE = 0, STG_PLAIN, STG_FILTER, STG_BAND, STG_BX_PLAN, STG_BX_VALUES, STG_BX_TRACE, STG_TALLY, STG_PASS1, STG_COUNT };
static const char* const STAGE_NAMES[STG_COUNT] = {"k_align_quad", "k_align_quad_plain", "k_diag_filter", "k_band_align", "k_bx_plan",
                                                   "k_bx_values", "k_bx_trace", "k_tally_binned", "k_pass1"};

// One device block for every small counter of an iteration (planner bins and header, filter / band-pipeline counters, link
// count, cull and tally flags, insert-event count): one memset at the start of the alignment clears them all, and one copy
// brings back what the host wants to see.  Offsets in 32-bit words.
constexpr int CTRL_BINS = 0;                                   // [count N_BINS][off N_BINS][cursor N_BINS][wide_count][retry_count]
constexpr int CTRL_HDR = (3 * N_BINS + 2 + 1) & ~1;            // PH_* (8-byte aligned: the DP kernels fetch their range as a pair)
constexpr int CTRL_FILTER = CTRL_HDR + PH_WORDS;               // 4 words (k_diag_filter / k_band_align)
constexpr int CTRL_LKN = CTRL_FILTER + 4, CTRL_CULLF = CTRL_LKN + 1, CTRL_NEV = CTRL_CULLF + 1, CTRL_TFLAGS = CTRL_NEV + 1;
constexpr int CTRL_BXC = (CTRL_TFLAGS + 1 + 63) & ~63;         // BXC_* counters, a cache line each
constexpr int CTRL_WORDS = CTRL_BXC + BXC_WORDS;

struct mia_hip_ctx {
  int device = 0;
  int32_t* d_ctrl = nullptr;
  uint32_t stage_mask = ~0u;               // timed stages (mia_hip_set_stage_mask): an event pair costs the stream a few microseconds
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // the trace DP of the plan's own lists runs beside the values DP
  hipStream_t stream3 = nullptr; hipEvent_t ev_join3 = nullptr;                      // ... and both beside the planner and the full-window kernels of the reads the plan gave up on
  int32_t* d_retry2 = nullptr; int64_t retry2_cap = 0;                              // reads no band kernel could finish
  uint32_t* d_bx_slabs_late = nullptr; int64_t bx_slab_late_cap = 0; int bx_late_wgs = 0;
  // mia_hip_iterate without a host wait behind the alignment: cull, tally and consensus are queued at once and every one of
  // their kernels returns at its first instruction if *abort_if != 0 (reads are waiting for the exact one-read-per-thread
  // kernel: the host sees that with the consensus, runs it, and queues the chain again)
  const int32_t* abort_if = nullptr;
  bool spec_ok = false, spec_pending = false, spec_filtered = false, spec_bx = false, spec_plain = false;
  int64_t spec_redone = 0;                  // iterations whose cull / tally / consensus were queued twice (reads for the exact kernel)
  bool pend_encode = false; int32_t pend_L = 0, pend_wl = 0, pend_total = 0;      // mia_hip_iterate: d_ascii holds the new reference, d_ref not yet
  uint32_t* d_prep_bar = nullptr; uint32_t prep_bar_count = 0; bool no_prep_fuse = false;   // k_ref_prep's grid barrier (MIA_HIP_NO_PREP_FUSE=1: six launches)
  bool spec_force = false; int32_t* d_one = nullptr;      // MIA_HIP_SPEC_TEST=1 (tests): a word that holds 1
  bool no_spec = false;                     // MIA_HIP_NO_SPEC=1: wait for the alignment's counters before the cull is queued
  bool no_side_buckets = false;             // MIA_HIP_NO_SIDE_BUCKETS=1
  int buckets_queued = 0;                   // the tally's counting sort is already queued: 1 on the context's stream, 2 on stream2 (ev_join behind it)
  unsigned char* d_slabs_retry[3] = {nullptr, nullptr, nullptr};      // trace slabs of the band kernels' retry launch (see launch_window)
  bool bx_planner_aside = false;            // this call: the planner and the full-window kernels run on stream2, the band DPs on the context's stream
  bool bx_pending_join = false;                                                     // band kernels are still running on stream2 / stream3
  std::string err;
  struct PoolBlock { void* p; size_t cap; bool lent; };
  std::vector<PoolBlock> pool;             // device temporaries of the one-off calls (pool_alloc)
  // PSSMs (fwd, rc)
  int32_t* d_pssm = nullptr;
  int max_abs = 0;
  int tally_pk_bias = -1;                   // the tally's packed end-base sums (k_tally_binned): the bias that makes every score positive, or -1: entries too large
  int max_pos = 0;   // largest positive PSSM entry: bounds any score by rows * max_pos
  bool have_pssm = false;
  PackSet packs;
  // reads
  ReadSet rs{};
  uint8_t* d_packed = nullptr; uint32_t* d_roff = nullptr; uint16_t* d_len = nullptr;
  uint8_t* d_rc = nullptr; uint8_t* d_sk = nullptr;
  int32_t *d_as = nullptr, *d_ae = nullptr, *d_score = nullptr, *d_refstart = nullptr;
  int16_t* d_abr = nullptr; uint32_t* d_status = nullptr; int16_t* d_cols = nullptr;
  int max_len = 0;
  // plan
  int32_t *d_bin_of = nullptr, *d_list = nullptr, *d_wide_list = nullptr, *d_retry_list = nullptr;
  int use_band = 1;   // MIA_HIP_NO_BAND=1: the quad kernel stores the full trace
  int32_t* d_bins = nullptr;  // [count N_BINS][off N_BINS][cursor N_BINS][wide_count 1]
  // reference
  uint8_t* d_ref = nullptr; int ref_cap = 0; int L = 0, wrap = 0; int explicit_win = 0; bool have_ref = false; bool aligned = false;
  // cull
  int64_t* d_slot = nullptr; int64_t* d_partial = nullptr; int64_t* d_total = nullptr;
  uint8_t *d_slot_dropped = nullptr, *d_drop_f = nullptr, *d_drop_b = nullptr; int64_t n_slots = 0;
  // stale back_asp emulation (mia_consensus_kernels.h, k_cull_mark)
  int64_t* d_back_slot = nullptr;          // per read, persistent across iterations
  int64_t* d_front_slot0 = nullptr;        // per read: pass-1 front slot (followed for strand-unknown reads only)
  int32_t* d_link_act = nullptr;
  RecInfo ri{};                            // per read, per iteration
  SlotInfo si{};                           // per local slot, per iteration
  int64_t slot_cap = 0;
  Links lk{};                              // links produced by this context in the last cull
  int64_t* d_links_all = nullptr; int32_t n_links_all = 0; int64_t links_all_cap = 0;   // links to apply (own, or gathered from all ranks)
  int32_t* d_n_links_all = nullptr; int64_t links_cap_all = 0; int32_t* d_n_links_gathered = nullptr;
  int32_t* d_link_len = nullptr; int64_t link_len_cap = 0;
  uint32_t* d_cull_flags = nullptr;
  unsigned long long* d_sums = nullptr;
  int64_t read_base = 0;                   // global index of this context's first read (sharded runs)
  int64_t slot_base = 0;
  bool culled = false, links_applied = false;
  std::vector<int64_t*> owned_links;       // gathered link buffers (mia_hip_set_links)
  // tally
  TallyBuf tb{}; int tally_cap = 0; int32_t* d_ins_off = nullptr; int32_t* d_ins_total = nullptr;
  int32_t* d_ins_tally = nullptr; int64_t ins_tally_cap = 0; char* d_calls = nullptr; char* d_ins_calls = nullptr;
  int64_t ins_calls_cap = 0; int32_t n_events_host = 0; bool tallied = false;
  // trace slabs of the persistent DP grid (one per workgroup, per CPL class)
  unsigned char* d_slabs[N_CPL] = {nullptr, nullptr, nullptr};
  unsigned char* d_quad_slabs = nullptr;
  int quad_wgs = 0;
  int use_quad = 1;   // MIA_HIP_NO_QUAD=1 routes everything through the one-read-per-wave kernels
  int plain_behind_band = 0;
  int use_plain = 1;  // MIA_HIP_NO_PLAIN=1: no values-only first pass, every quad goes straight to the trace kernel
  int64_t plain_retried = 0, plain_total = 0;
  // the diagonal filter (diag_filter.h): flat matrix only
  bool tally_linear = false;               // MIA_HIP_NO_LINEAR_TALLY=1: the tally adds the four scores of every base
  bool ref_mostly_bases = true;            // fewer than 2 % of the reference columns are N
  int64_t kh_entries = 0;                  // > 0: the reference has N columns and its 10-mer table lists them (bandx_body.h, N COLUMNS)
  bool diag_scripts_missing = false;       // the last alignment left the scripts of its ST_DIAG reads unwritten (k_diag_scripts)
  int lazy_scripts = 1;                    // MIA_HIP_EAGER_SCRIPTS=1: the band pipeline writes them as it goes
  bool wide_to_caller = false;             // run_wide marks its reads ST_ESCAPE instead of aligning them (the anchored pass 1)
  int use_wild = 1;                        // MIA_HIP_NO_WILD=1: reads whose window holds an N go to the full-window kernels
  bool flat = false; int use_filter = 1;   // MIA_HIP_NO_DIAG_FILTER=1 sends every read to the DP kernels
  int64_t pre_cull_records = 0, pre_cull_links = 0; bool pre_cull_valid = false;   // mia_hip_score_sums' by-products
  int32_t *d_kocc_cnt = nullptr, *d_kocc_pos = nullptr;   // 10-mer table of the reference (diag_filter.h: KmerOcc)
  uint64_t* d_planes = nullptr; int64_t plane_cap = 0;   // lo | hi | ok, plane_cap words each
  uint32_t* d_filter_n = nullptr; int64_t filter_proven = 0, filter_seen = 0;   // device: {finished by the filter, left over, finished by the banded DP}
  int use_banddp = 1;                       // MIA_HIP_NO_BAND_DP=1: the filter's left-overs go straight to the full-window kernels
  int32_t* d_left_list = nullptr; int64_t left_cap = 0;
  uint32_t* d_band_slabs = nullptr; int64_t band_slab_cap = 0;
  int64_t band_done = 0;
  // the matrix-agnostic band pipeline (bandx_kernels.h): plan -> values-only DP -> trace DP, for any PSSM
  bool bx_ok = false;                       // the matrices allow it (bx_make_tables)
  uint32_t* d_cull_sync = nullptr;          // k_slot_count's arrival counter (zero between launches)
  double myers_kernel_ms = 0; bool myers_no_lanes = false;   // the kernels of the last mia_hip_myers call (HIP events); MIA_HIP_MYERS_NO_LANES=1: every pair through k_myers
  int bucket_clean_nb = -1;                 // the tally's bucket counts are zero for this bucket count (k_bucket_scan leaves them so)
  bool no_auto_plain = false;               // MIA_HIP_NO_AUTO_PLAIN=1: the values-only pass behind the band only when MIA_HIP_PLAIN_BEHIND_BAND asks for it
  bool bx_serial = false;                   // MIA_HIP_BX_SERIAL=1
  uint32_t bx_dbg = 0;                      // MIA_HIP_BX_DEBUG (profiling): 1 no traceback, 2 one DP row, 4 no values launch, 8 no trace launch
  int use_lanes = 1;                        // MIA_HIP_NO_LANES=1: the band DPs one read per lane (bx_values / bx_trace) instead of W/8 lanes per read (bandx_lanes.h)
  int use_bx = 1;                           // MIA_HIP_NO_BANDX=1: the round-1 path (flat: filter + k_band_align; PSSM: full-window kernels)
  int bx_filter_first = 0;                  // MIA_HIP_BX_FILTER=1 (flat matrix): k_diag_filter runs ahead of the plan
  int32_t *d_bx_sub = nullptr, *d_bx_mrow = nullptr; int16_t *d_bx_loss = nullptr, *d_bx_dl = nullptr;   // sub | sub * 256; M; losses; block costs
  int32_t bx_min_m = 0, bx_max_m = 0;
  uint64_t* d_rplanes = nullptr; int rplane_words = 0;   // bit planes of the stored reads (k_read_planes)
  uint32_t* d_refnib = nullptr; int64_t refnib_cap = 0;
  uint32_t* d_khash = nullptr; int32_t* d_khash_ovf = nullptr; uint32_t khash_cap = 0;   // 10-mers of the reference (bandx_body.h: KmerHash)
  int32_t* d_umax = nullptr; bool umax_valid = false;
  uint32_t* d_bx_plan = nullptr; int32_t* d_bx_expect = nullptr; int32_t* d_bx_lists = nullptr; int64_t bx_cap = 0;
  uint32_t* d_bx_ctr = nullptr;
  uint32_t* d_bx_slabs = nullptr; int64_t bx_slab_cap = 0;
  int bx_values_wgs = 0, bx_trace_wgs = 0;
  int64_t bx_seen = 0, bx_done[3] = {0, 0, 0};   // reads planned on; finished by the plan / the values DP / the trace DP
  int64_t bx_launches = 0;
  uint32_t bx_last[BXC_COUNTERS] = {0};     // counters of the last call (list lengths, reasons a read was not planned)
  int grid_wgs = 0;
  int window_wgs[N_CPL] = {0, 0, 0};
  int cus = 1;
  uint32_t dbg = 0;   // MIA_HIP_DEBUG_SKIP: timing experiments only, results are wrong when set
  // read bucketing for the LDS-privatised tally
  int32_t* d_bucket = nullptr; int bucket_cap = 0; int32_t* d_order = nullptr;
  int use_binned_tally = 1;   // MIA_HIP_NO_BINNED_TALLY=1: plain global-atomic tally
  int32_t* d_tally_slabs = nullptr; int64_t tally_slab_cap = 0;   // one LDS window per tally workgroup, summed by k_tally_reduce
  // wide scratch
  int32_t* d_scratch = nullptr; int64_t scratch_cap = 0; int64_t* d_scratch_off = nullptr; int64_t scratch_off_cap = 0;
  // timing
  // HIP-event timers of the kernels bench.py reports, one per stage (mia_hip_stage_stats); pairs are recycled through ev_free
  struct StageTimer { std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; double ms = 0; int64_t launches = 0; };
  StageTimer stg[STG_COUNT];
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
  // pinned staging: small copies to and from pageable memory wait for the stream, pinned ones do not
  unsigned char* h_pin = nullptr; static constexpr size_t PIN_BYTES = 1 << 20, PIN_MISC = 64 << 10;
  double pass1_ms = 0; int64_t pass1_filtered = 0, pass1_anchored = 0;   // reads of the last pass-1 call that the diagonal filter decided
  bool consensus_done = false;
  // mia_hip_iterate: one iteration with the planner's answers, the cut line and the insert-event count left on the device
  bool deferred = false;                    // align_all: no host round trip before its end
  bool in_iterate = false;                  // the control block was cleared as a whole by the alignment's one memset
  int32_t* d_plan_hdr = nullptr;            // PH_* (k_plan_scan)
  const double* dev_cut = nullptr; double* d_cut_buf = nullptr;
  int min_len = 0;                          // shortest stored read
  std::vector<int32_t> h_len;               // read lengths on the host (the score-cut regression of reads of different lengths)
  char* d_ascii = nullptr; int64_t ascii_cap = 0;
  char* d_cons = nullptr; int64_t cons_cap = 0;              // result of an iteration: [CH_WORDS header][consensus string]
  int32_t* d_cons_pos = nullptr; int64_t cons_pos_cap = 0;
  unsigned char* h_pin2 = nullptr; size_t pin2_bytes = 0;   // results of an iteration (header + consensus string)
  int64_t iter_fallbacks = 0;
  // sharded runs: one context per GPU; the exchanges go through a table of collectives (RCCL over xGMI from
  // mia_hip_comm_init, or whatever mia_hip_comm_attach was given), on the context's own stream
  mia_hip_collectives coll{}; bool comm = false; int comm_ranks = 1, comm_rank = 0;
  unsigned long long* d_gather = nullptr;   // [PRE_WORDS * ranks] score sums, record, link and exact-kernel counts of every rank
  std::vector<int64_t> h_gather;            // ... on the host, once the alignment's one wait is over
  int64_t ev_pad = 0;                       // insert events per rank the event all-gather is sized for (0: not known yet)
  int64_t* d_lstage = nullptr; int64_t lstage_cap = 0;       // links / insert events of all ranks, padded to the longest
  int64_t* d_lmine = nullptr; int64_t lmine_cap = 0;
  int64_t* d_lall = nullptr; int64_t lall_cap = 0;
  int32_t* d_scores_all = nullptr; int64_t scores_all_cap = 0;
  int64_t trim_escapes = 0;   // reads of the last mia_hip_trim call that took the exact scalar path
  int64_t ins_total_host = 0;
};

#define HIPCHK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                                    \
      return MIA_HIP_ERR_DEVICE;                                                                       \
    }                                                                                                  \
  } while (0)

// frees the watched temporaries when the scope is left, whichever way (every early return of a long entry point)
struct ScopeFree {
  std::vector<void**> w;
  void watch(void** pp) { w.push_back(pp); }
  ~ScopeFree() { for (void** pp : w) if (*pp) { (void)hipFree(*pp); *pp = nullptr; } }
};

template <class T>
static int dev_alloc(mia_hip_ctx* ctx, T** p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (n == 0) n = 1;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); *p = nullptr; return MIA_HIP_ERR_NOMEM; }
  return MIA_HIP_OK;
}

// Temporaries of the one-off calls (pass 1, adapter trimming) come from blocks the context keeps: hipMalloc/hipFree of
// twenty-odd buffers per call cost more than the kernels between them.  A block is lent for the length of a scope.
template <class T>
static int pool_alloc(mia_hip_ctx* ctx, T** p, size_t n) {
  size_t bytes = (n ? n : 1) * sizeof(T);
  int best = -1;
  for (size_t k = 0; k < ctx->pool.size(); k++) {
    auto& b = ctx->pool[k];
    if (!b.lent && b.cap >= bytes && (best < 0 || b.cap < ctx->pool[best].cap)) best = (int)k;
  }
  if (best < 0) {
    // grow the largest idle block rather than keep one that nothing fits any more
    int idle = -1;
    for (size_t k = 0; k < ctx->pool.size(); k++)
      if (!ctx->pool[k].lent && (idle < 0 || ctx->pool[k].cap > ctx->pool[idle].cap)) idle = (int)k;
    if (idle >= 0) { (void)hipFree(ctx->pool[idle].p); ctx->pool.erase(ctx->pool.begin() + idle); }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); *p = nullptr; return MIA_HIP_ERR_NOMEM; }
    ctx->pool.push_back({q, bytes, false});
    best = (int)ctx->pool.size() - 1;
  }
  ctx->pool[best].lent = true;
  *p = (T*)ctx->pool[best].p;
  return MIA_HIP_OK;
}
struct PoolScope {
  mia_hip_ctx* c;
  std::vector<void**> w;
  explicit PoolScope(mia_hip_ctx* ctx) : c(ctx) {}
  void watch(void** pp) { w.push_back(pp); }
  ~PoolScope() {
    for (void** pp : w) {
      if (!*pp) continue;
      for (auto& b : c->pool) if (b.p == *pp) b.lent = false;
      *pp = nullptr;
    }
  }
};
static void pool_drop(mia_hip_ctx* ctx) {
  for (auto& b : ctx->pool) (void)hipFree(b.p);
  ctx->pool.clear();
}

extern "C" const char* mia_hip_last_error(const mia_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int mia_hip_create(mia_hip_ctx** out, int device_index) {
  if (!out) return MIA_HIP_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MIA_HIP_ERR_DEVICE;   // no CPU fallback, by design
  if (device_index < 0 || device_index >= ndev) return MIA_HIP_ERR_ARG;
  mia_hip_ctx* ctx = new mia_hip_ctx();
  ctx->device = device_index;
  if (hipSetDevice(device_index) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join3, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
    delete ctx;
    return MIA_HIP_ERR_DEVICE;
  }
  {
    // persistent DP grid: one 64-lane workgroup per wave slot of the chip (8 waves/SIMD x 4 SIMDs x CUs)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_index) != hipSuccess) { delete ctx; return MIA_HIP_ERR_DEVICE; }
    ctx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
    ctx->grid_wgs = prop.multiProcessorCount * 32;
    ctx->quad_wgs = prop.multiProcessorCount * 16;   // 128 VGPRs -> 4 waves per SIMD
    const char* nbt = getenv("MIA_HIP_NO_BINNED_TALLY");
    if (nbt && atoi(nbt)) ctx->use_binned_tally = 0;
    const char* nband = getenv("MIA_HIP_NO_BAND");
    if (nband && atoi(nband)) ctx->use_band = 0;
    const char* npl = getenv("MIA_HIP_NO_PLAIN");
    if (npl && atoi(npl)) ctx->use_plain = 0;
    const char* pbb = getenv("MIA_HIP_PLAIN_BEHIND_BAND");
    if (pbb && atoi(pbb)) ctx->plain_behind_band = 1;
    const char* nf = getenv("MIA_HIP_NO_DIAG_FILTER");
    if (nf && atoi(nf)) { ctx->use_filter = 0; ctx->use_bx = 0; }      // every shortcut off: the full-window DP kernels only
    const char* nbd = getenv("MIA_HIP_NO_BAND_DP");
    if (nbd && atoi(nbd)) { ctx->use_banddp = 0; ctx->use_bx = 0; }
    const char* nbx = getenv("MIA_HIP_NO_BANDX");
    if (nbx && atoi(nbx)) ctx->use_bx = 0;
    if (const char* nl2 = getenv("MIA_HIP_NO_LANES")) if (atoi(nl2)) ctx->use_lanes = 0;
    if (const char* bd2 = getenv("MIA_HIP_BX_DEBUG")) ctx->bx_dbg = (uint32_t)atoi(bd2);
    if (const char* bs2 = getenv("MIA_HIP_BX_SERIAL")) ctx->bx_serial = atoi(bs2) != 0;
    if (const char* sb2 = getenv("MIA_HIP_NO_SIDE_BUCKETS")) ctx->no_side_buckets = atoi(sb2) != 0;
    if (const char* ns2 = getenv("MIA_HIP_NO_SPEC")) ctx->no_spec = atoi(ns2) != 0;
    if (const char* pf2 = getenv("MIA_HIP_NO_PREP_FUSE")) ctx->no_prep_fuse = atoi(pf2) != 0;
    if (const char* st2 = getenv("MIA_HIP_SPEC_TEST")) ctx->spec_force = atoi(st2) != 0;
    if (const char* ml = getenv("MIA_HIP_MYERS_NO_LANES")) ctx->myers_no_lanes = atoi(ml) != 0;
    if (const char* na = getenv("MIA_HIP_NO_AUTO_PLAIN")) ctx->no_auto_plain = atoi(na) != 0;
    const char* egs = getenv("MIA_HIP_EAGER_SCRIPTS");
    if (egs && atoi(egs)) ctx->lazy_scripts = 0;
    const char* nwl = getenv("MIA_HIP_NO_WILD");
    if (nwl && atoi(nwl)) ctx->use_wild = 0;
    const char* bxf = getenv("MIA_HIP_BX_FILTER");
    if (bxf && atoi(bxf)) ctx->bx_filter_first = 1;
    const char* nq = getenv("MIA_HIP_NO_QUAD");
    if (nq && atoi(nq)) ctx->use_quad = 0;
    const char* qw = getenv("MIA_HIP_QUAD_WAVES_PER_CU");
    if (qw && atoi(qw) > 0) ctx->quad_wgs = prop.multiProcessorCount * atoi(qw);
    const char* dbg = getenv("MIA_HIP_DEBUG_SKIP");
    if (dbg) ctx->dbg = (uint32_t)atoi(dbg);
    const char* g = getenv("MIA_HIP_GRID_WAVES_PER_CU");
    if (g && atoi(g) > 0) ctx->grid_wgs = prop.multiProcessorCount * atoi(g);
  }
  if (dev_alloc(ctx, &ctx->d_pssm, 2 * PSSM_WORDS) || dev_alloc(ctx, &ctx->d_ctrl, CTRL_WORDS) ||
      dev_alloc(ctx, &ctx->d_total, 1) || dev_alloc(ctx, &ctx->d_ins_total, 1) ||
      dev_alloc(ctx, &ctx->d_bx_sub, 2 * BX_SUB_WORDS) || dev_alloc(ctx, &ctx->d_bx_mrow, 2 * 31 * 4) || dev_alloc(ctx, &ctx->d_bx_loss, BX_LOSS_WORDS) ||
      dev_alloc(ctx, &ctx->d_bx_dl, BX_DL_WORDS) ||
      dev_alloc(ctx, &ctx->d_cut_buf, 2)) {
    delete ctx;
    return MIA_HIP_ERR_NOMEM;
  }
  ctx->d_bins = ctx->d_ctrl + CTRL_BINS;
  ctx->d_plan_hdr = ctx->d_ctrl + CTRL_HDR;
  ctx->d_filter_n = reinterpret_cast<uint32_t*>(ctx->d_ctrl + CTRL_FILTER);
  ctx->lk.n = ctx->d_ctrl + CTRL_LKN;
  ctx->d_cull_flags = reinterpret_cast<uint32_t*>(ctx->d_ctrl + CTRL_CULLF);
  ctx->tb.n_events = ctx->d_ctrl + CTRL_NEV;
  ctx->tb.flags = reinterpret_cast<uint32_t*>(ctx->d_ctrl + CTRL_TFLAGS);
  ctx->d_bx_ctr = reinterpret_cast<uint32_t*>(ctx->d_ctrl + CTRL_BXC);
  if (hipMemset(ctx->d_ctrl, 0, CTRL_WORDS * 4) != hipSuccess) { delete ctx; return MIA_HIP_ERR_DEVICE; }
  {
    const int32_t one = 1;
    if (hipMalloc((void**)&ctx->d_one, 4) != hipSuccess || hipMemcpy(ctx->d_one, &one, 4, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return MIA_HIP_ERR_DEVICE; }
    if (hipMalloc((void**)&ctx->d_prep_bar, 4) != hipSuccess || hipMemset(ctx->d_prep_bar, 0, 4) != hipSuccess) { delete ctx; return MIA_HIP_ERR_DEVICE; }
  }
  if (hipHostMalloc((void**)&ctx->h_pin, mia_hip_ctx::PIN_BYTES, hipHostMallocDefault) != hipSuccess) ctx->h_pin = nullptr;   // optional
  *out = ctx;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_comm_destroy(mia_hip_ctx* ctx);
extern "C" void mia_hip_destroy(mia_hip_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)mia_hip_comm_destroy(ctx);
  void* ptrs[] = {ctx->d_ctrl, ctx->d_pssm, ctx->d_packed, ctx->d_roff, ctx->d_len, ctx->d_rc, ctx->d_sk, ctx->d_as, ctx->d_ae, ctx->d_score,
                  ctx->d_refstart, ctx->d_abr, ctx->d_status, ctx->d_cols, ctx->d_bin_of, ctx->d_list, ctx->d_wide_list, ctx->d_retry_list,
                  ctx->d_ref, ctx->d_slot, ctx->d_partial, ctx->d_total, ctx->d_slot_dropped, ctx->d_drop_f,
                  ctx->d_drop_b, ctx->tb.tally, ctx->tb.events, ctx->d_ins_off,
                  ctx->d_ins_total, ctx->d_ins_tally, ctx->d_calls, ctx->d_ins_calls, ctx->d_scratch, ctx->d_scratch_off,
                  ctx->d_slabs[0], ctx->d_slabs[1], ctx->d_slabs[2], ctx->d_quad_slabs, ctx->d_bucket, ctx->d_order,
                  ctx->d_back_slot, ctx->ri.flen, ctx->ri.blen, ctx->ri.actf, ctx->ri.trec, ctx->si.reclen, ctx->si.writer, ctx->si.mult,
                  ctx->lk.rec, ctx->d_link_len, ctx->d_link_act, ctx->d_front_slot0, ctx->si.recact, ctx->d_sums, ctx->d_n_links_gathered, ctx->d_tally_slabs, ctx->d_planes, ctx->d_kocc_cnt, ctx->d_kocc_pos, ctx->d_left_list, ctx->d_band_slabs,
                  ctx->d_bx_sub, ctx->d_bx_mrow, ctx->d_bx_loss, ctx->d_bx_dl, ctx->d_rplanes, ctx->d_khash, ctx->d_khash_ovf, ctx->d_refnib, ctx->d_umax, ctx->d_bx_plan, ctx->d_bx_expect, ctx->d_bx_lists, ctx->d_bx_slabs, ctx->d_cut_buf, ctx->d_ascii, ctx->d_cons, ctx->d_cons_pos, ctx->d_gather, ctx->d_lstage, ctx->d_lmine, ctx->d_lall, ctx->d_scores_all};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  pool_drop(ctx);
  for (int64_t* p : ctx->owned_links) if (p) (void)hipFree(p);
  if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
  if (ctx->h_pin2) (void)hipHostFree(ctx->h_pin2);
  for (auto& e : ctx->ev_free) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& t : ctx->stg) for (auto& e : t.pending) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
  if (ctx->ev_join3) (void)hipEventDestroy(ctx->ev_join3);
  if (ctx->d_retry2) (void)hipFree(ctx->d_retry2);
  if (ctx->d_cull_sync) (void)hipFree(ctx->d_cull_sync);
  if (ctx->d_bx_slabs_late) (void)hipFree(ctx->d_bx_slabs_late);
  if (ctx->d_one) (void)hipFree(ctx->d_one);
  for (int k = 0; k < 3; k++) if (ctx->d_slabs_retry[k]) (void)hipFree(ctx->d_slabs_retry[k]);
  if (ctx->d_prep_bar) (void)hipFree(ctx->d_prep_bar);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  delete ctx;
}

extern "C" int mia_hip_sync(mia_hip_ctx* ctx) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

extern "C" int mia_hip_set_pssm(mia_hip_ctx* ctx, const int32_t* fwd, const int32_t* rc) {
  if (!ctx || !fwd || !rc) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  int m = 0;
  for (int i = 0; i < PSSM_WORDS; i++) {
    int a = fwd[i] < 0 ? -fwd[i] : fwd[i], b = rc[i] < 0 ? -rc[i] : rc[i];
    if (a > m) m = a;
    if (b > m) m = b;
  }
  if (m > 32000) { ctx->err = "PSSM entries beyond +-32000 do not fit the int16 substitution table"; return MIA_HIP_ERR_RANGE; }
  ctx->max_abs = m;
  {
    int lo = 0, hi = 0;
    for (int k = 0; k < PSSM_WORDS; k++) { lo = std::min(lo, std::min((int)fwd[k], (int)rc[k])); hi = std::max(hi, std::max((int)fwd[k], (int)rc[k])); }
    ctx->tally_pk_bias = (-lo + hi <= 2047 && !getenv("MIA_HIP_NO_PACKED_TALLY")) ? -lo : -1;
  }
  ctx->max_pos = 0;
  for (int i = 0; i < PSSM_WORDS; i++) { if (fwd[i] > ctx->max_pos) ctx->max_pos = fwd[i]; if (rc[i] > ctx->max_pos) ctx->max_pos = rc[i]; }
  const int cpls[N_CPL] = {4, 8, 12};
  for (int c = 0; c < N_CPL; c++) ctx->packs.ok[c] = make_pack_params(64 * cpls[c], m, &ctx->packs.p[c]) ? 1 : 0;
  HIPCHK(hipMemcpyAsync(ctx->d_pssm, fwd, PSSM_WORDS * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_pssm + PSSM_WORDS, rc, PSSM_WORDS * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->have_pssm = true;
  ctx->flat = pssm_is_flat(fwd, rc);
  // the score words of a column are linear in its base counts iff sm[d][X][b] (X = A,C,G,T; src/map_align.c:258-261)
  // depends neither on the depth nor on the strand: k_tally_binned then counts only and derives the scores
  ctx->tally_linear = true;
  for (int d = 0; d < 2 * PSSM_DEPTH + 1; d++)
    for (int x = 0; x < 4; x++)
      for (int b = 0; b < 5; b++)
        if (fwd[(d * 5 + x) * 5 + b] != fwd[x * 5 + b] || rc[(d * 5 + x) * 5 + b] != fwd[x * 5 + b]) ctx->tally_linear = false;
  if (const char* nl = getenv("MIA_HIP_NO_LINEAR_TALLY")) if (atoi(nl)) ctx->tally_linear = false;
  // tables of the band pipeline (bandx_body.h): substitution scores by (strand, depth, read base, reference code), the best
  // score of every row kind, and what a non-identical base costs at least
  {
    std::vector<int32_t> sub(2 * BX_SUB_WORDS, 0), mrow(2 * 31 * 4, 0);
    std::vector<int16_t> loss(BX_LOSS_WORDS, 0), dl(BX_DL_WORDS, 0);
    ctx->bx_ok = bx_make_tables(fwd, rc, sub.data(), mrow.data(), loss.data(), dl.data(), &ctx->bx_min_m, &ctx->bx_max_m);
    if (ctx->bx_ok) {
      for (int k = 0; k < BX_SUB_WORDS; k++) sub[(size_t)BX_SUB_WORDS + k] = sub[(size_t)k] * 256;
      HIPCHK(hipMemcpyAsync(ctx->d_bx_sub, sub.data(), sub.size() * 4, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(ctx->d_bx_mrow, mrow.data(), mrow.size() * 4, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(ctx->d_bx_loss, loss.data(), loss.size() * 2, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(ctx->d_bx_dl, dl.data(), dl.size() * 2, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    ctx->umax_valid = false;
    if (ctx->bx_ok && ctx->d_umax && ctx->rs.n > 0) {
      hipLaunchKernelGGL(k_bx_umax, dim3((unsigned)((ctx->rs.n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->rs, ctx->d_bx_mrow, ctx->d_umax);
      HIPCHK(hipGetLastError());
      ctx->umax_valid = true;
    }
  }
  return MIA_HIP_OK;
}

static inline uint8_t base_code(char b) {   // src/map_align.c:16-29: only upper-case ACGT are bases
  switch (b) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

// 4-bit base codes, two per byte, each read starting on a 4-byte boundary of `packed` (zero-filled by the caller)
// (on the host's threads: a million reads are 100 MB of characters, 0.1 s on one core -- as long as thirty iterations)
static void pack_reads(int64_t n, const char* bases, const int64_t* offsets, const uint32_t* roff, const uint16_t* len, uint8_t* packed) {
  uint8_t lut[256];
  for (int c = 0; c < 256; c++) lut[c] = base_code((char)c);
  auto range = [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; i++) {
      const unsigned char* s = reinterpret_cast<const unsigned char*>(bases) + offsets[i];
      uint8_t* d = packed + roff[i];
      const int l = len[i];
      int k = 0;
      for (; k + 1 < l; k += 2) d[k >> 1] = (uint8_t)(lut[s[k]] | (lut[s[k + 1]] << 4));
      if (k < l) d[k >> 1] = lut[s[k]];
    }
  };
  int T = (int)std::thread::hardware_concurrency();
  if (const char* e = getenv("MIA_HIP_THREADS")) T = atoi(e);
  T = std::max(1, std::min(std::min(T, 32), (int)(n / 16384) + 1));
  if (T == 1) { range(0, n); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) th.emplace_back(range, n * t / T, n * (t + 1) / T);
  range(0, n / T);
  for (auto& x : th) x.join();
}

extern "C" int mia_hip_upload_reads(mia_hip_ctx* ctx, int64_t n, const char* bases, const int64_t* offsets, const uint8_t* rc,
                                    const uint8_t* strand_known, const int32_t* as, const int32_t* ae) {
  if (!ctx || n < 0 || (n > 0 && (!bases || !offsets || !rc || !strand_known || !as || !ae))) return MIA_HIP_ERR_ARG;
  if (n >= (int64_t)1 << 31) { ctx->err = "more than 2^31 reads per context"; return MIA_HIP_ERR_ARG; }
  HIPCHK(hipSetDevice(ctx->device));
  std::vector<uint32_t> roff((size_t)n);
  std::vector<uint16_t> len((size_t)n);
  uint64_t total = 0;
  int max_len = 1;
  for (int64_t i = 0; i < n; i++) {
    int64_t l = offsets[i + 1] - offsets[i];
    if (l < 1 || l > MIA_HIP_MAX_READ) { ctx->err = "read length outside 1..256 (INIT_ALN_SEQ_LEN)"; return MIA_HIP_ERR_ARG; }
    roff[i] = (uint32_t)total;
    len[i] = (uint16_t)l;
    if (l > max_len) max_len = (int)l;
    total += (uint64_t)(((l + 1) / 2 + 3) & ~3);
    if (total >= ((uint64_t)1 << 32)) { ctx->err = "packed read store exceeds 4 GiB per context"; return MIA_HIP_ERR_ARG; }
  }
  std::vector<uint8_t> packed((size_t)total + 8, 0);
  pack_reads(n, bases, offsets, roff.data(), len.data(), packed.data());
  ctx->max_len = max_len;
  ctx->min_len = max_len;
  ctx->h_len.assign(len.begin(), len.end());
  for (int64_t i = 0; i < n; i++) if ((int)len[(size_t)i] < ctx->min_len) ctx->min_len = len[(size_t)i];
  const int stride = (max_len + 3) & ~3;
  int rcx = 0;
  rcx |= dev_alloc(ctx, &ctx->d_packed, packed.size());
  rcx |= dev_alloc(ctx, &ctx->d_roff, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_len, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_rc, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_sk, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_as, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_ae, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_score, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_refstart, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_abr, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_status, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_cols, (size_t)n * stride);
  rcx |= dev_alloc(ctx, &ctx->d_bin_of, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_umax, (size_t)n);
  ctx->rplane_words = (max_len + 63) >> 6;
  rcx |= dev_alloc(ctx, &ctx->d_rplanes, (size_t)n * 2 * ctx->rplane_words);
  rcx |= dev_alloc(ctx, &ctx->d_list, (size_t)n + 4 * N_BINS);   // quad bins are padded to multiples of four
  rcx |= dev_alloc(ctx, &ctx->d_wide_list, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_retry_list, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_slot, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_partial, (size_t)(n / 256 + 2));
  rcx |= dev_alloc(ctx, &ctx->d_drop_f, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_drop_b, (size_t)n);
  ctx->n_slots = 2 * n + 16;
  rcx |= dev_alloc(ctx, &ctx->d_slot_dropped, (size_t)ctx->n_slots);
  rcx |= dev_alloc(ctx, &ctx->d_back_slot, (size_t)n) | dev_alloc(ctx, &ctx->d_front_slot0, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->ri.flen, (size_t)n) | dev_alloc(ctx, &ctx->ri.blen, (size_t)n) | dev_alloc(ctx, &ctx->ri.actf, (size_t)n) |
         dev_alloc(ctx, &ctx->ri.trec, (size_t)n * 16);
  ctx->slot_cap = 2 * n + 16;
  rcx |= dev_alloc(ctx, &ctx->si.reclen, (size_t)ctx->slot_cap) | dev_alloc(ctx, &ctx->si.recact, (size_t)ctx->slot_cap) | dev_alloc(ctx, &ctx->si.writer, (size_t)ctx->slot_cap) |
         dev_alloc(ctx, &ctx->si.mult, (size_t)ctx->slot_cap);
  ctx->lk.cap = (int32_t)std::min<int64_t>(n + 16, (int64_t)1 << 20);     // link index field of SlotInfo::writer: 20 bits
  if (ctx->lk.cap > (int32_t)LINK_NONE - 1) ctx->lk.cap = (int32_t)LINK_NONE - 1;
  rcx |= dev_alloc(ctx, &ctx->lk.rec, (size_t)ctx->lk.cap * 4);
  if (rcx) return MIA_HIP_ERR_NOMEM;
  HIPCHK(hipMemcpyAsync(ctx->d_packed, packed.data(), packed.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_roff, roff.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_len, len.data(), (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_rc, rc, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_sk, strand_known, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_as, as, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_ae, ae, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_score, 0, (size_t)n * 4, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_status, 0, (size_t)n * 4, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_abr, 0, (size_t)n * 2, ctx->stream));   // no soft clip known after pass 1
  HIPCHK(hipMemsetAsync(ctx->d_slot_dropped, 0, (size_t)ctx->n_slots, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_back_slot, 0xFF, (size_t)n * 8, ctx->stream));   // -1: never split
  HIPCHK(hipMemsetAsync(ctx->d_front_slot0, 0xFF, (size_t)n * 8, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_drop_f, 0, (size_t)n, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_drop_b, 0, (size_t)n, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ReadSet& r = ctx->rs;
  r.n = n; r.packed = ctx->d_packed; r.roff = ctx->d_roff; r.len = ctx->d_len; r.rc = ctx->d_rc; r.sk = ctx->d_sk;
  r.as = ctx->d_as; r.ae = ctx->d_ae; r.score = ctx->d_score; r.refstart = ctx->d_refstart; r.abr = ctx->d_abr;
  r.status = ctx->d_status; r.cols = ctx->d_cols; r.stride = stride;
  if (ctx->d_order) { (void)hipFree(ctx->d_order); ctx->d_order = nullptr; }
  ctx->umax_valid = false;
  if (n > 0) hipLaunchKernelGGL(k_read_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->rs, ctx->rplane_words, ctx->d_rplanes);
  if (ctx->have_pssm && ctx->bx_ok && n > 0) {
    hipLaunchKernelGGL(k_bx_umax, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->rs, ctx->d_bx_mrow, ctx->d_umax);
    HIPCHK(hipGetLastError());
    ctx->umax_valid = true;
  }
  ctx->aligned = false;
  ctx->tallied = false;
  ctx->culled = false;
  return MIA_HIP_OK;
}

// an event pair for one launch of stage `st`; the start event is recorded here, the end event by stage_end
static int stage_begin(mia_hip_ctx* ctx, Stage st, hipStream_t on = nullptr) {
  if (!((ctx->stage_mask >> st) & 1u)) return 0;
  if (ctx->ev_free.empty()) {
    hipEvent_t x, y;
    if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return -1;
    ctx->ev_free.push_back({x, y});
  }
  auto p = ctx->ev_free.back();
  ctx->ev_free.pop_back();
  ctx->stg[st].pending.push_back(p);
  (void)hipEventRecord(p.first, on ? on : ctx->stream);
  return 0;
}
static void stage_end(mia_hip_ctx* ctx, Stage st, hipStream_t on = nullptr) {
  if (((ctx->stage_mask >> st) & 1u) && !ctx->stg[st].pending.empty()) (void)hipEventRecord(ctx->stg[st].pending.back().second, on ? on : ctx->stream);
}

static void drain_events(mia_hip_ctx* ctx) {
  for (auto& t : ctx->stg) {
    for (auto& e : t.pending) {
      float ms = 0;
      if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) {
        t.ms += ms;
        t.launches++;
      }
      ctx->ev_free.push_back(e);
    }
    t.pending.clear();
  }
}

extern "C" int mia_hip_set_stage_mask(mia_hip_ctx* ctx, uint32_t mask) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  drain_events(ctx);
  ctx->stage_mask = mask;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_stage_stats(mia_hip_ctx* ctx, int reset, int32_t cap, const char** names, double* ms, int64_t* launches, int32_t* n_stages) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  drain_events(ctx);
  if (n_stages) *n_stages = STG_COUNT;
  for (int k = 0; k < STG_COUNT && k < cap; k++) {
    if (names) names[k] = STAGE_NAMES[k];
    if (ms) ms[k] = ctx->stg[k].ms;
    if (launches) launches[k] = ctx->stg[k].launches;
  }
  if (reset) for (auto& t : ctx->stg) { t.ms = 0; t.launches = 0; }
  return MIA_HIP_OK;
}

extern "C" int mia_hip_kernel_time(mia_hip_ctx* ctx, int reset, double* align_ms, int64_t* launches) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  drain_events(ctx);
  if (align_ms) *align_ms = ctx->stg[STG_TRACE].ms;
  if (launches) *launches = ctx->stg[STG_TRACE].launches;
  if (reset) { ctx->stg[STG_TRACE].ms = 0; ctx->stg[STG_TRACE].launches = 0; }
  return MIA_HIP_OK;
}

extern "C" int mia_hip_filter_stats(mia_hip_ctx* ctx, int reset, int64_t* reads_seen, int64_t* reads_finished, double* kernel_ms, int64_t* launches) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  drain_events(ctx);
  if (reads_seen) *reads_seen = ctx->filter_seen;
  if (reads_finished) *reads_finished = ctx->filter_proven;
  if (kernel_ms) *kernel_ms = ctx->stg[STG_FILTER].ms;
  if (launches) *launches = ctx->stg[STG_FILTER].launches;
  if (reset) { ctx->filter_seen = 0; ctx->filter_proven = 0; ctx->stg[STG_FILTER].ms = 0; ctx->stg[STG_FILTER].launches = 0; }
  return MIA_HIP_OK;
}

extern "C" int mia_hip_band_stats(mia_hip_ctx* ctx, int reset, int64_t* reads_finished, double* kernel_ms, int64_t* launches) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  drain_events(ctx);
  if (reads_finished) *reads_finished = ctx->band_done;
  if (kernel_ms) *kernel_ms = ctx->stg[STG_BAND].ms;
  if (launches) *launches = ctx->stg[STG_BAND].launches;
  if (reset) { ctx->band_done = 0; ctx->stg[STG_BAND].ms = 0; ctx->stg[STG_BAND].launches = 0; }
  return MIA_HIP_OK;
}

extern "C" int mia_hip_bx_stats(mia_hip_ctx* ctx, int reset, int64_t* reads4, double* kernel_ms3, int64_t* launches) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  drain_events(ctx);
  if (reads4) { reads4[0] = ctx->bx_seen; reads4[1] = ctx->bx_done[0]; reads4[2] = ctx->bx_done[1]; reads4[3] = ctx->bx_done[2]; }
  if (kernel_ms3) for (int k = 0; k < 3; k++) kernel_ms3[k] = ctx->stg[STG_BX_PLAN + k].ms;
  if (launches) *launches = ctx->bx_launches;
  if (reset) { ctx->bx_seen = 0; for (int k = 0; k < 3; k++) { ctx->bx_done[k] = 0; ctx->stg[STG_BX_PLAN + k].ms = 0; ctx->stg[STG_BX_PLAN + k].launches = 0; } ctx->bx_launches = 0; }
  return MIA_HIP_OK;
}

extern "C" int mia_hip_bx_counters(mia_hip_ctx* ctx, uint32_t* out32) {
  if (!ctx || !out32) return MIA_HIP_ERR_ARG;
  for (int k = 0; k < BXC_COUNTERS; k++) out32[k] = ctx->bx_last[k];
  return MIA_HIP_OK;
}

extern "C" int mia_hip_plain_stats(mia_hip_ctx* ctx, int reset, double* plain_ms, int64_t* plain_launches, int64_t* reads_in, int64_t* reads_retried) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  drain_events(ctx);
  if (plain_ms) *plain_ms = ctx->stg[STG_PLAIN].ms;
  if (plain_launches) *plain_launches = ctx->stg[STG_PLAIN].launches;
  if (reads_in) *reads_in = ctx->plain_total;
  if (reads_retried) *reads_retried = ctx->plain_retried;
  if (reset) { ctx->stg[STG_PLAIN].ms = 0; ctx->stg[STG_PLAIN].launches = 0; ctx->plain_total = 0; ctx->plain_retried = 0; }
  return MIA_HIP_OK;
}

template <int CPL>
// own_slabs: the launch gets trace slabs of its own (ctx->d_slabs_retry) and a grid of at most 1 024 workgroups -- it may run
// beside another launch of the same class on another stream (the band kernels' retry list beside the planner's kernels)
static hipError_t launch_window(mia_hip_ctx* ctx, int ci, const int32_t* list, int count, const int32_t* dev_range = nullptr, hipStream_t on = nullptr,
                                bool own_slabs = false) {
  // slab = the largest trace of this class: 256 rows x 64*CPL columns, one byte per cell
  const int64_t slab = (int64_t)MAX_READ * 64 * CPL;
  // persistent grid: never more workgroups than are resident at once (a late starter would work through its whole
  // share of the list on a drained GPU)
  if (!ctx->window_wgs[ci]) {
    int occ = 0;
    const int cus = ctx->cus;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_align_window<CPL>, 64, 0) != hipSuccess || occ < 1) occ = 1;
    const int per_cu = ctx->grid_wgs / cus;      // the configured ceiling (MIA_HIP_GRID_WAVES_PER_CU)
    ctx->window_wgs[ci] = cus * (occ < per_cu ? occ : per_cu);
  }
  int grid = (dev_range || count >= ctx->window_wgs[ci]) ? ctx->window_wgs[ci] : count;   // (a count on the device: the whole persistent grid)
  unsigned char* slabs = nullptr;
  if (own_slabs) {
    grid = std::min(grid, 1024);
    if (!ctx->d_slabs_retry[ci] && hipMalloc((void**)&ctx->d_slabs_retry[ci], (size_t)slab * 1024) != hipSuccess) return hipErrorOutOfMemory;
    slabs = ctx->d_slabs_retry[ci];
  } else {
    if (!ctx->d_slabs[ci]) {
      if (hipMalloc((void**)&ctx->d_slabs[ci], (size_t)slab * ctx->grid_wgs) != hipSuccess) return hipErrorOutOfMemory;
    }
    slabs = ctx->d_slabs[ci];
  }
  RefInfo ref{ctx->d_ref, ctx->L, ctx->wrap, ctx->explicit_win};
  if (stage_begin(ctx, STG_TRACE, on)) return hipErrorOutOfMemory;
  hipLaunchKernelGGL((k_align_window<CPL>), dim3(grid), dim3(64), 0, on ? on : ctx->stream, ctx->rs, ref, ctx->d_pssm, ctx->packs.p[ci], list,
                     count, slabs, slab, ctx->d_wide_list, ctx->d_bins + 3 * N_BINS, ctx->dbg, dev_range);
  stage_end(ctx, STG_TRACE, on);
  return hipGetLastError();
}

// The band kernels of the three-stream order (align_all) are still running beside the context's stream: wait for them
// there, then let a window kernel take what they put on the retry list (its length stays on the device).
static int bx_join_and_retry(mia_hip_ctx* ctx) {
  if (!ctx->bx_pending_join) return MIA_HIP_OK;
  ctx->bx_pending_join = false;
  const int32_t* range = ctx->d_plan_hdr + PH_RETRY2;
  const int cols = ctx->max_len + 2 * REALIGN_BUFFER + 2;
  const bool aside = ctx->bx_planner_aside;
  // mia_hip_iterate: the retry list is the band kernels' alone, so its window kernel only waits for THEM (the trace DP on
  // stream3; values and late trace are ahead of it on this stream) and runs beside the tail of the planner's chain on
  // stream2, with trace slabs of its own; the planner is waited for behind it
  if (!aside) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join3, 0));
  hipError_t e = cols <= 64 * 4 ? launch_window<4>(ctx, 0, ctx->d_retry2, 0, range, nullptr, aside)
               : cols <= 64 * 8 ? launch_window<8>(ctx, 1, ctx->d_retry2, 0, range, nullptr, aside)
                                : launch_window<12>(ctx, 2, ctx->d_retry2, 0, range, nullptr, aside);
  if (e != hipSuccess) { ctx->err = std::string("k_align_window (band retry list) launch: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
  if (aside) {
    HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));      // (the planner's chain ends here)
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  }
  return MIA_HIP_OK;
}

static int align_all(mia_hip_ctx* ctx);
static int comm_pre_cull_enqueue(mia_hip_ctx* ctx, const int32_t* d_wide_count);
static bool comm_pre_cull_collect(mia_hip_ctx* ctx);

extern "C" int mia_hip_realign(mia_hip_ctx* ctx, const char* new_ref, int32_t ref_len, int circular) {
  if (!ctx || !new_ref || ref_len <= 0) return MIA_HIP_ERR_ARG;
  if (!ctx->have_pssm || !ctx->d_packed) { ctx->err = "set_pssm and upload_reads must precede realign"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  // add_ref_wrap (src/mia.c:657-689): first min(L,256) bases appended when circular
  const int L = ref_len, wl = circular ? (L < MAX_READ ? L : MAX_READ) : 0, wrap = L + wl;
  std::vector<uint8_t> codes((size_t)wrap + 64, 4);
  int64_t n_other = 0;
  for (int i = 0; i < L; i++) { codes[i] = base_code(new_ref[i]); n_other += codes[i] > 3; }
  for (int i = 0; i < wl; i++) codes[L + i] = codes[i];
  // a reference full of ambiguity codes (mt311 itself: every other column) leaves the diagonal filter nothing to decide
  ctx->ref_mostly_bases = n_other * 50 <= L;
  ctx->kh_entries = 0;
  if (n_other && ctx->use_wild) {
    ctx->kh_entries = kh_wild_entries(codes.data(), wrap, BX_WILD);
    if (ctx->kh_entries > ((int64_t)1 << 24)) ctx->kh_entries = 0;
  }
  if ((int)codes.size() > ctx->ref_cap) {
    if (dev_alloc(ctx, &ctx->d_ref, codes.size() * 2)) return MIA_HIP_ERR_NOMEM;
    ctx->ref_cap = (int)codes.size() * 2;
  }
  if (ctx->h_pin && codes.size() <= mia_hip_ctx::PIN_BYTES - mia_hip_ctx::PIN_MISC) {
    HIPCHK(hipStreamSynchronize(ctx->stream));             // the staging area may still feed an earlier copy
    memcpy(ctx->h_pin + mia_hip_ctx::PIN_MISC, codes.data(), codes.size());
    HIPCHK(hipMemcpyAsync(ctx->d_ref, ctx->h_pin + mia_hip_ctx::PIN_MISC, codes.size(), hipMemcpyHostToDevice, ctx->stream));
  } else {
    HIPCHK(hipMemcpyAsync(ctx->d_ref, codes.data(), codes.size(), hipMemcpyHostToDevice, ctx->stream));
  }
  ctx->L = L; ctx->wrap = wrap; ctx->have_ref = true; ctx->explicit_win = 0;
  return align_all(ctx);
}

// exact kernel for whole-reference windows and escaped reads (rare): one read per thread, int32 scores and trace in scratch
static int run_wide(mia_hip_ctx* ctx, const RefInfo& ref, int32_t n_wide) {
    if (ctx->wide_to_caller) {
      // pass 1's windows: a read the trace kernels could not finish (a gap of 63 or more on the path -- a window around a
      // stray cluster, where the read does not belong) goes to the whole-strand DP with everything else that is left over;
      // one read per thread here would take longer than that whole kernel
      hipLaunchKernelGGL(k_mark_status, dim3((n_wide + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_wide_list, n_wide, ctx->rs.status, ST_ESCAPE);
      HIPCHK(hipGetLastError());
      return MIA_HIP_OK;
    }
    std::vector<int32_t> wl((size_t)n_wide), as((size_t)n_wide), ae((size_t)n_wide);
    std::vector<uint16_t> ln((size_t)n_wide);
    HIPCHK(hipMemcpy(wl.data(), ctx->d_wide_list, (size_t)n_wide * 4, hipMemcpyDeviceToHost));
    std::vector<int64_t> soff((size_t)n_wide);
    int64_t total = 0;
    for (int t = 0; t < n_wide; t++) {
      HIPCHK(hipMemcpy(&as[t], ctx->rs.as + wl[t], 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(&ae[t], ctx->rs.ae + wl[t], 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(&ln[t], ctx->rs.len + wl[t], 2, hipMemcpyDeviceToHost));
      int s, n1;
      read_window(ref, as[t], ae[t], ln[t], &s, &n1);
      soff[t] = total;
      total += (int64_t)ln[t] * n1 + 5 * (int64_t)n1 + 16;
    }
    if (total > ctx->scratch_cap) {
      if (dev_alloc(ctx, &ctx->d_scratch, (size_t)total)) return MIA_HIP_ERR_NOMEM;
      ctx->scratch_cap = total;
    }
    if (n_wide > ctx->scratch_off_cap) {
      if (dev_alloc(ctx, &ctx->d_scratch_off, (size_t)n_wide)) return MIA_HIP_ERR_NOMEM;
      ctx->scratch_off_cap = n_wide;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_scratch_off, soff.data(), (size_t)n_wide * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_align_wide, dim3((n_wide + 63) / 64), dim3(64), 0, ctx->stream, ctx->rs, ref, ctx->d_pssm, ctx->d_wide_list,
                       n_wide, ctx->d_scratch_off, ctx->d_scratch);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

// mia_hip_iterate left the new reference as ASCII in d_ascii: its codes, if no k_ref_prep makes them
static void encode_now(mia_hip_ctx* ctx) {
  if (!ctx->pend_encode) return;
  ctx->pend_encode = false;
  hipLaunchKernelGGL(k_ref_encode, dim3((unsigned)((ctx->pend_total + 255) / 256)), dim3(256), 0, ctx->stream, (const char*)ctx->d_ascii, ctx->pend_L, ctx->pend_wl,
                     ctx->d_ref, ctx->pend_total);
}

// the counters of a deferred alignment, once its control block has reached the host (hb: the words from CTRL_BINS + 3 N_BINS on)
static void align_counters_collect(mia_hip_ctx* ctx, const int32_t* hb, bool filtered, bool bx, bool plain) {
  constexpr int C0 = CTRL_BINS + 3 * N_BINS;
  const int32_t* h_hdr = hb + (CTRL_HDR - C0);
  const uint32_t* h_filter = reinterpret_cast<const uint32_t*>(hb + (CTRL_FILTER - C0));
  const uint32_t* h_bxc = reinterpret_cast<const uint32_t*>(hb + (CTRL_BXC - C0));
  ctx->filter_seen += ctx->rs.n;
  if (filtered) {
    ctx->filter_proven += h_filter[0];
    ctx->band_done += h_filter[2];
    if (bx) {
      ctx->filter_proven += h_bxc[BXC_DONE_PLAN * BXC_STRIDE];
      for (int k = 0; k < 3; k++) ctx->bx_done[k] += h_bxc[(BXC_DONE_PLAN + k) * BXC_STRIDE];
      ctx->bx_seen += h_bxc[BXC_SEEN * BXC_STRIDE];
    }
  }
  for (int k = 0; k < BXC_COUNTERS; k++) ctx->bx_last[k] = bx ? h_bxc[k * BXC_STRIDE] : 0;
  if (plain) ctx->plain_retried += h_hdr[PH_RETRIED_PLAIN];
}

// every strand_known read against its window of ctx->d_ref: plan, values-only pass, trace kernels, exact kernel
static int align_all(mia_hip_ctx* ctx) {
  const int64_t n = ctx->rs.n;
  const int wrap = ctx->wrap;
  ctx->bx_pending_join = false;
  if (!ctx->spec_ok) { ctx->spec_pending = false; ctx->abort_if = nullptr; }
  ctx->bx_planner_aside = false;
  ctx->buckets_queued = 0;                  // (a counting sort queued for an earlier alignment is void)
  if (n == 0) { ctx->aligned = true; return MIA_HIP_OK; }
  RefInfo ref{ctx->d_ref, ctx->L, wrap, ctx->explicit_win};
  int32_t* d_count = ctx->d_bins;
  int32_t* d_off = ctx->d_bins + N_BINS;
  int32_t* d_cursor = ctx->d_bins + 2 * N_BINS;
  int32_t* d_wide_count = ctx->d_bins + 3 * N_BINS;
  int32_t* d_retry_count = ctx->d_bins + 3 * N_BINS + 1;
  const int tb = 256, gb = (int)((n + (int64_t)tb * PLAN_PER - 1) / ((int64_t)tb * PLAN_PER));
  const int filter_ok = ctx->flat && ctx->use_filter && ctx->ref_mostly_bases;
  // the band pipeline for any matrix (bandx_kernels.h); it needs the 10-mer table and windows free of N
  const bool bx = ctx->bx_ok && ctx->use_bx && (ctx->ref_mostly_bases || ctx->kh_entries > 0) && wrap <= (1 << 22) && !(ctx->dbg & 128u);
  const bool run_filter = filter_ok && (!bx || ctx->bx_filter_first);
  // mia_hip_iterate with the band pipeline alone: codes, control block, planes, nibbles and 10-mer table in one launch (k_ref_prep)
  const bool fused_prep = ctx->pend_encode && bx && !run_filter && !ctx->no_prep_fuse;
  if (!fused_prep) {
    encode_now(ctx);
    HIPCHK(hipMemsetAsync(ctx->d_ctrl, 0, (size_t)CTRL_WORDS * 4, ctx->stream));     // every counter of the iteration at once
  }
  const int filtered = run_filter || bx;            // bin_of carries marks for the planner
  uint32_t h_filter_n = 0;
  bool banded = false;
  if (filtered) {
    // reads whose alignment is provably one gap-free diagonal never reach the DP kernels (diag_filter.h)
    const int64_t words = plane_words((int64_t)wrap + 64);
    if (words > ctx->plane_cap) {
      if (dev_alloc(ctx, &ctx->d_planes, (size_t)words * 3)) return MIA_HIP_ERR_NOMEM;
      ctx->plane_cap = words;
    }
    RefPlanes rp{ctx->d_planes, ctx->d_planes + ctx->plane_cap, ctx->d_planes + 2 * ctx->plane_cap};
    if (!fused_prep)
    hipLaunchKernelGGL(k_ref_planes, dim3((unsigned)((words + 3) / 4 < 4096 ? (words + 3) / 4 : 4096)), dim3(256), 0, ctx->stream, ctx->d_ref, (int64_t)wrap + 64, words,
                       ctx->d_planes, ctx->d_planes + ctx->plane_cap, ctx->d_planes + 2 * ctx->plane_cap);
    // the 10-mer table of this reference (rule (c) looks long clean stretches up instead of sliding over every diagonal;
    // the band plans are made of its anchors); not for the very long concatenated strings mia_hip_align_windows may be given
    KmerOcc ko{nullptr, nullptr};
    if (wrap <= (1 << 22) && (run_filter || !bx)) {
      if (!ctx->d_kocc_cnt && (dev_alloc(ctx, &ctx->d_kocc_cnt, (size_t)DF_KTAB) || dev_alloc(ctx, &ctx->d_kocc_pos, (size_t)DF_KTAB * DF_KCAP)))
        return MIA_HIP_ERR_NOMEM;
      HIPCHK(hipMemsetAsync(ctx->d_kocc_cnt, 0, (size_t)DF_KTAB * 4, ctx->stream));
      hipLaunchKernelGGL(k_kmer_occ, dim3((unsigned)((wrap + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ref, (int64_t)wrap, ctx->d_kocc_cnt, ctx->d_kocc_pos, 0);
      ko.cnt = ctx->d_kocc_cnt; ko.pos = ctx->d_kocc_pos;
    }
    // what the filter leaves over goes through a banded DP first (bandx_kernels.h, or round 1's band_body.h); both need the table
    banded = bx || (ctx->use_banddp && ko.cnt && !(ctx->dbg & 128u));
    if (run_filter) {
      if (banded && n > ctx->left_cap) {
        if (dev_alloc(ctx, &ctx->d_left_list, (size_t)n)) return MIA_HIP_ERR_NOMEM;
        ctx->left_cap = n;
      }
      if (stage_begin(ctx, STG_FILTER)) return MIA_HIP_ERR_NOMEM;
      hipLaunchKernelGGL(k_diag_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->rs, ref, rp, ko, (int64_t)wrap, ctx->d_bin_of, ctx->dbg,
                         banded ? ctx->d_left_list : nullptr, ctx->d_filter_n + 1);
      stage_end(ctx, STG_FILTER);
      HIPCHK(hipGetLastError());
    }
    if (bx) {
      // the reference as 4-bit codes, the lists, one slab per wavefront of the trace kernel's persistent grid
      const int64_t nw = bx_nib_words((int64_t)wrap + 64);
      if (nw > ctx->refnib_cap) {
        if (dev_alloc(ctx, &ctx->d_refnib, (size_t)nw * 2)) return MIA_HIP_ERR_NOMEM;
        ctx->refnib_cap = nw * 2;
      }
      if (!fused_prep)
      hipLaunchKernelGGL(k_ref_nibbles, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ref, (int64_t)wrap, nw, ctx->d_refnib);
      const uint32_t kslots = kh_slots_for_entries(wrap, ctx->kh_entries);
      if (kslots > ctx->khash_cap) {
        if (dev_alloc(ctx, &ctx->d_khash, (size_t)kslots * 4) || dev_alloc(ctx, &ctx->d_khash_ovf, (size_t)kslots * 2)) return MIA_HIP_ERR_NOMEM;
        ctx->khash_cap = kslots;
      }
      const KmerHash kh{ctx->d_khash, ctx->d_khash_ovf, kslots - 1, kh_shift_for(kslots), ctx->kh_entries > 0 ? BX_WILD : 0};
      if (fused_prep) {
        RefPrep rp2;
        rp2.ascii = (const char*)ctx->d_ascii; rp2.L = ctx->pend_L; rp2.wl = ctx->pend_wl; rp2.total = ctx->pend_total; rp2.codes = ctx->d_ref;
        rp2.ctrl = ctx->d_ctrl; rp2.ctrl_words = CTRL_WORDS;
        rp2.kslot = ctx->d_khash; rp2.kovf = ctx->d_khash_ovf; rp2.kslots = kslots; rp2.kmask = kh.mask; rp2.kshift = kh.shift; rp2.kwild = kh.wild;
        rp2.plane_words = words; rp2.plo = ctx->d_planes; rp2.phi = ctx->d_planes + ctx->plane_cap; rp2.pok = ctx->d_planes + 2 * ctx->plane_cap;
        rp2.nib_words = nw; rp2.nib = ctx->d_refnib;
        const int64_t want = ((int64_t)wrap + 255) / 256;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, ctx->cus));      // one workgroup per compute unit at most: all resident
        ctx->prep_bar_count += grid;
        rp2.bar = ctx->d_prep_bar; rp2.bar_target = ctx->prep_bar_count;
        ctx->pend_encode = false;
        hipLaunchKernelGGL(k_ref_prep, dim3(grid), dim3(256), 0, ctx->stream, rp2);
      } else {
      HIPCHK(hipMemsetAsync(ctx->d_khash, 0xFF, (size_t)kslots * 16, ctx->stream));
      hipLaunchKernelGGL(k_kmer_hash, dim3((unsigned)((wrap + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ref, (int64_t)wrap, ctx->d_khash, ctx->d_khash_ovf,
                         kh.mask, kh.shift, kh.wild);
      }
      if (n > ctx->bx_cap) {
        if (dev_alloc(ctx, &ctx->d_bx_plan, (size_t)n) || dev_alloc(ctx, &ctx->d_bx_expect, (size_t)n) || dev_alloc(ctx, &ctx->d_bx_lists, (size_t)n * 3 * BX_NCLS))
          return MIA_HIP_ERR_NOMEM;
        ctx->bx_cap = n;
      }
      if (!ctx->bx_values_wgs) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ctx->use_lanes ? (const void*)k_bxl_values : (const void*)k_bx_values, 256, 0) != hipSuccess || occ < 1) occ = 1;
        ctx->bx_values_wgs = ctx->cus * std::min(occ, 4);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ctx->use_lanes ? (const void*)k_bxl_trace : (const void*)k_bx_trace, 256, 0) != hipSuccess || occ < 1) occ = 1;
        ctx->bx_trace_wgs = ctx->cus * std::min(occ, 4);      // (every wavefront of the trace grid owns a slab)
      }
      // (the lanes kernels store four rows per word: whole blocks of four rows)
      const int64_t slab_words = ctx->use_lanes ? (int64_t)((ctx->max_len + 3) & ~3) * BXL_SLAB_ROW_WORDS : (int64_t)ctx->max_len * BX_SLAB_ROW_WORDS;
      if (slab_words * ctx->bx_trace_wgs * 4 > ctx->bx_slab_cap) {
        if (dev_alloc(ctx, &ctx->d_bx_slabs, (size_t)(slab_words * ctx->bx_trace_wgs * 4))) return MIA_HIP_ERR_NOMEM;
        ctx->bx_slab_cap = slab_words * ctx->bx_trace_wgs * 4;
      }
      BxDev bd;
      bd.tab.sub = ctx->d_bx_sub; bd.tab.mrow = ctx->d_bx_mrow; bd.tab.loss = ctx->d_bx_loss; bd.tab.dl = ctx->d_bx_dl;
      bd.lazy_scripts = ctx->lazy_scripts;
      bd.dbg = ctx->bx_dbg & (3u | 32u);
      // MIA_HIP_BX_SERIAL=1: round 2's order (band kernels, then the planner over everything they left open)
      // (caller-supplied windows -- mia_hip_align_windows -- can be of any length: the retry list's window kernel is picked by read length)
      const bool new_flow = ctx->use_lanes && !ctx->bx_serial && !(ctx->dbg & 256u) && !ctx->explicit_win;
      if (new_flow) {
        if (n > ctx->retry2_cap) { if (dev_alloc(ctx, &ctx->d_retry2, (size_t)n)) return MIA_HIP_ERR_NOMEM; ctx->retry2_cap = n; }
        if (!ctx->bx_late_wgs) ctx->bx_late_wgs = ctx->cus;        // the values DP's left-overs are few: one workgroup per CU
        const int64_t late_words = (int64_t)((ctx->max_len + 3) & ~3) * BXL_SLAB_ROW_WORDS * ctx->bx_late_wgs * 4;
        if (late_words > ctx->bx_slab_late_cap) { if (dev_alloc(ctx, &ctx->d_bx_slabs_late, (size_t)late_words)) return MIA_HIP_ERR_NOMEM; ctx->bx_slab_late_cap = late_words; }
      }
      bd.retry = ctx->d_retry2; bd.retry_n = ctx->d_plan_hdr + PH_RETRY2 + 1;
      bd.listed_mark = new_flow ? -5 : 0;
      if (ctx->lazy_scripts) ctx->diag_scripts_missing = true;
      bd.tab.min_m = ctx->bx_min_m; bd.tab.max_m = ctx->bx_max_m;
      bd.sub256 = ctx->d_bx_sub + BX_SUB_WORDS;
      bd.refnib = ctx->d_refnib;
      bd.umax = (ctx->umax_valid && ctx->rs.roff == ctx->d_roff) ? ctx->d_umax : nullptr;    // (a borrowed read set has none)
      bd.rplanes = bd.umax ? ctx->d_rplanes : nullptr;
      bd.rplane_words = ctx->rplane_words;
      bd.plan = ctx->d_bx_plan; bd.expect = ctx->d_bx_expect; bd.lists = ctx->d_bx_lists; bd.list_stride = ctx->bx_cap; bd.ctr = ctx->d_bx_ctr;
      if (stage_begin(ctx, STG_BX_PLAN)) return MIA_HIP_ERR_NOMEM;
      {
        const int32_t* in_list = run_filter ? ctx->d_left_list : nullptr;
        const dim3 pg((unsigned)((n + 255) / 256)), pb(256);
        switch ((ctx->max_len + 63) >> 6) {       // 64-row words of the longest read
          case 1: hipLaunchKernelGGL(k_bx_plan<1>, pg, pb, 0, ctx->stream, ctx->rs, ref, rp, kh, (int64_t)wrap, bd, in_list, ctx->d_filter_n + 1, n, ctx->d_bin_of); break;
          case 2: hipLaunchKernelGGL(k_bx_plan<2>, pg, pb, 0, ctx->stream, ctx->rs, ref, rp, kh, (int64_t)wrap, bd, in_list, ctx->d_filter_n + 1, n, ctx->d_bin_of); break;
          case 3: hipLaunchKernelGGL(k_bx_plan<3>, pg, pb, 0, ctx->stream, ctx->rs, ref, rp, kh, (int64_t)wrap, bd, in_list, ctx->d_filter_n + 1, n, ctx->d_bin_of); break;
          default: hipLaunchKernelGGL(k_bx_plan<4>, pg, pb, 0, ctx->stream, ctx->rs, ref, rp, kh, (int64_t)wrap, bd, in_list, ctx->d_filter_n + 1, n, ctx->d_bin_of); break;
        }
      }
      stage_end(ctx, STG_BX_PLAN);
      HIPCHK(hipGetLastError());
      if (!(ctx->dbg & 256u)) {
        if (new_flow) {
          // Three things that do not depend on each other run side by side: the trace DP of the plan's own lists (stream3),
          // the values DP with one more trace launch behind it for what it could not finish (stream2), and -- on the
          // context's stream, below -- the planner and the full-window kernels for the reads the plan gave up on.  The
          // step pays the longest of the three instead of their sum; what no band kernel can finish (the reference's
          // index-0 quirk: next to nothing) goes on a retry list that a window kernel reads behind the join.
          // mia_hip_iterate (deferred): the values DP and the trace launch behind it ARE the step's critical path, so they
          // stay on the context's stream, right behind the plan -- a cross-stream wait costs 20-30 us each way, and it is the
          // planner with the full-window kernels (short, done long before) that moves to stream2.
          hipStream_t vs = ctx->deferred ? ctx->stream : ctx->stream2;
          HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
          HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
          HIPCHK(hipStreamWaitEvent(ctx->stream3, ctx->ev_fork, 0));
          if (stage_begin(ctx, STG_BX_TRACE, ctx->stream3)) return MIA_HIP_ERR_NOMEM;
          if (!(ctx->bx_dbg & 8u))
            hipLaunchKernelGGL(k_bxl_trace, dim3((unsigned)ctx->bx_trace_wgs), dim3(256), 0, ctx->stream3, ctx->rs, ref, bd, ctx->d_bx_slabs, slab_words, ctx->d_bin_of);
          stage_end(ctx, STG_BX_TRACE, ctx->stream3);
          HIPCHK(hipEventRecord(ctx->ev_join3, ctx->stream3));
          if (stage_begin(ctx, STG_BX_VALUES, vs)) return MIA_HIP_ERR_NOMEM;
          if (!(ctx->bx_dbg & 4u)) {
            hipLaunchKernelGGL(k_bxl_values, dim3((unsigned)ctx->bx_values_wgs), dim3(256), 0, vs, ctx->rs, ref, bd, ctx->d_bin_of);
            hipLaunchKernelGGL(k_bxl_trace_late, dim3((unsigned)ctx->bx_late_wgs), dim3(256), 0, vs, ctx->rs, ref, bd, ctx->d_bx_slabs_late, slab_words, ctx->d_bin_of);
          }
          stage_end(ctx, STG_BX_VALUES, vs);
          if (!ctx->deferred) HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
          HIPCHK(hipGetLastError());
          ctx->bx_planner_aside = ctx->deferred;
          ctx->bx_pending_join = true;
        } else {
        // The two band DPs do not depend on each other (a read the values DP cannot finish stays open for the full-window
        // kernels): they run side by side on two streams -- both are persistent grids whose wavefronts leave as soon as the
        // chunks run out, so each fills what the other leaves idle, and the step pays the longer of the two tails, not both.
        HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        if (stage_begin(ctx, STG_BX_TRACE, ctx->stream2)) return MIA_HIP_ERR_NOMEM;
        if (ctx->bx_dbg & 8u) {}
        else if (ctx->use_lanes) hipLaunchKernelGGL(k_bxl_trace, dim3((unsigned)ctx->bx_trace_wgs), dim3(256), 0, ctx->stream2, ctx->rs, ref, bd, ctx->d_bx_slabs, slab_words, ctx->d_bin_of);
        else hipLaunchKernelGGL(k_bx_trace, dim3((unsigned)ctx->bx_trace_wgs), dim3(256), 0, ctx->stream2, ctx->rs, ref, bd, ctx->d_bx_slabs, slab_words, ctx->d_bin_of);
        stage_end(ctx, STG_BX_TRACE, ctx->stream2);
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
        if (stage_begin(ctx, STG_BX_VALUES)) return MIA_HIP_ERR_NOMEM;
        if (ctx->bx_dbg & 4u) {}
        else if (ctx->use_lanes) hipLaunchKernelGGL(k_bxl_values, dim3((unsigned)ctx->bx_values_wgs), dim3(256), 0, ctx->stream, ctx->rs, ref, bd, ctx->d_bin_of);
        else hipLaunchKernelGGL(k_bx_values, dim3((unsigned)ctx->bx_values_wgs), dim3(256), 0, ctx->stream, ctx->rs, ref, bd, ctx->d_bin_of);
        stage_end(ctx, STG_BX_VALUES);
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        HIPCHK(hipGetLastError());
        }
      }
      ctx->bx_launches++;
    } else if (banded) {
      // a persistent grid of wavefronts, each with its own trace slab (the number of left-over reads stays on the device)
      const int64_t chunks = (n + 63) / 64;
      const int grid = (int)(chunks < 3072 ? chunks : 3072);          // three wavefronts per SIMD
      const int64_t slab_words = (int64_t)ctx->max_len * BAND_ROW_WORDS;
      if (slab_words * grid > ctx->band_slab_cap) {
        if (dev_alloc(ctx, &ctx->d_band_slabs, (size_t)(slab_words * grid))) return MIA_HIP_ERR_NOMEM;
        ctx->band_slab_cap = slab_words * grid;
      }
      if (stage_begin(ctx, STG_BAND)) return MIA_HIP_ERR_NOMEM;
      hipLaunchKernelGGL(k_band_align, dim3(grid), dim3(64), 0, ctx->stream, ctx->rs, ref, rp, ko, (int64_t)wrap, ctx->d_left_list, ctx->d_filter_n + 1,
                         ctx->d_band_slabs, slab_words, ctx->d_bin_of, ctx->d_filter_n + 2, ctx->d_filter_n + 3, ctx->dbg);
      stage_end(ctx, STG_BAND);
      HIPCHK(hipGetLastError());
    }
  }
  // behind the banded DP the values-only pass has nothing left to prove: what the band could not take nearly always needs a trace
  // (MIA_HIP_PLAIN_BEHIND_BAND=1: values-only pass over the band pipeline's left-overs all the same -- with a position-specific
  // matrix most of them are gap-free reads with many substitutions, which it finishes at half the trace kernel's price)
  // ... unless the plan gives up on many reads: against a reference full of ambiguity codes (every run's first iteration
  // against mt311: the N columns alone exhaust the loss budget of one read in twenty, one in five with the ancient matrix),
  // or when it did so in the iteration before.  Most of those reads are gap-free; the values-only pass finishes them at
  // half the trace kernel's price (first iteration 2.84 -> 2.59 ms flat, 5.35 -> 4.08 ms ancient, per 1 M reads).
  // Either way every read gets the reference's alignment: the choice only moves work between exact kernels.
  int64_t last_rejects = 0;
  for (int k = 1; k < BXF_KINDS; k++) last_rejects += ctx->bx_last[BXC_FAIL0 + k];
  const bool many_rejects = bx && !ctx->no_auto_plain && (!ctx->ref_mostly_bases || last_rejects * 20 > n);
  const bool use_plain = ctx->use_plain && (!banded || ctx->plain_behind_band || many_rejects);
  hipStream_t ps = ctx->bx_planner_aside ? ctx->stream2 : ctx->stream;      // the planner's stream (see the band launches above)
  hipLaunchKernelGGL(k_plan_count, dim3(gb), dim3(tb), 0, ps, ctx->rs, ref, ctx->packs, ctx->use_quad, filtered, filtered && use_plain, ctx->d_bin_of, d_count, ctx->d_filter_n);
  if (ctx->deferred) {
    // ---- mia_hip_iterate: the same plan, but its numbers stay on the device (k_plan_scan) and every DP kernel reads its own
    // range; the host looks at the counters once, when everything has been queued --------------------------------------
    int32_t* hdr = ctx->d_plan_hdr;
    const bool dbg_steps = getenv("MIA_HIP_ITER_DEBUG") != nullptr;
    auto ck = [&](const char* what) { if (dbg_steps) { hipError_t e = hipStreamSynchronize(ps); fprintf(stderr, "[align_all deferred] %s: %s\n", what, hipGetErrorString(e)); fflush(stderr); } };
    ck("plan_count");
    int32_t* d_retry_cnt = hdr + PH_RETRY + 1;
    HIPCHK(hipMemsetAsync(ctx->d_list, 0xFF, ((size_t)n + 4 * N_BINS) * 4, ps));           // -1 = empty slot (quad padding)
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(512), 0, ps, d_count, d_off, hdr, 0);
    hipLaunchKernelGGL(k_plan_fill, dim3(gb), dim3(tb), 0, ps, n, ctx->d_bin_of, d_off, d_cursor, ctx->d_list);
    ck("memsets");
    hipLaunchKernelGGL(k_wide_seed, dim3(1), dim3(256), 0, ps, ctx->d_list, hdr, ctx->d_wide_list, d_wide_count);
    ck("scan fill seed");
    // every window class reads its own range from the header (the list is rewritten by the re-plan below, so they all go
    // first; a class without reads costs an empty launch -- a few microseconds; putting them on the second stream beside
    // the quad kernels was tried and gained nothing)
    for (int ci = 0; ci < N_CPL; ci++) {
      hipError_t e = ci == 0 ? launch_window<4>(ctx, ci, ctx->d_list, 0, hdr + PH_WIN + 2 * ci, ps)
                   : ci == 1 ? launch_window<8>(ctx, ci, ctx->d_list, 0, hdr + PH_WIN + 2 * ci, ps)
                             : launch_window<12>(ctx, ci, ctx->d_list, 0, hdr + PH_WIN + 2 * ci, ps);
      if (e != hipSuccess) { ctx->err = std::string("k_align_window launch: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
    }
    ck("window classes");
    const size_t quad_lds = (size_t)Q_G * q_sub_bytes(ctx->max_len) + 16;
    if (ctx->use_quad) {
      if (use_plain) {
        if (stage_begin(ctx, STG_PLAIN, ps)) return MIA_HIP_ERR_NOMEM;
        hipLaunchKernelGGL(k_align_quad_plain, dim3(ctx->quad_wgs), dim3(64), quad_lds, ps, ctx->rs, ref, ctx->d_pssm, ctx->d_list, 0, ctx->d_bin_of,
                           (const int32_t*)(hdr + PH_QUAD));
        stage_end(ctx, STG_PLAIN, ps);
        ck("quad plain");
        // what it could not finish (and what the filter's gap hint kept out of it), re-planned into quads
        HIPCHK(hipMemsetAsync(d_count, 0, (size_t)N_BINS * 4, ps));
        HIPCHK(hipMemsetAsync(d_cursor, 0, (size_t)N_BINS * 4, ps));
        hipLaunchKernelGGL(k_plan_recount, dim3(gb), dim3(tb), 0, ps, n, ctx->d_bin_of, d_count);
        HIPCHK(hipMemsetAsync(ctx->d_list, 0xFF, ((size_t)n + 4 * N_BINS) * 4, ps));
        hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(512), 0, ps, d_count, d_off, hdr, 1);
        hipLaunchKernelGGL(k_plan_fill, dim3(gb), dim3(tb), 0, ps, n, ctx->d_bin_of, d_off, d_cursor, ctx->d_list);
        ck("replan");
      }
      const int64_t slab = (int64_t)Q_G * MAX_READ * Q_TRACE_STRIDE;
      if (!ctx->d_quad_slabs && hipMalloc((void**)&ctx->d_quad_slabs, (size_t)slab * ctx->quad_wgs) != hipSuccess) return MIA_HIP_ERR_NOMEM;
      const int qgrid = ctx->quad_wgs;
      if (dbg_steps) {
        int32_t hh[PH_WORDS];
        (void)hipMemcpy(hh, hdr, sizeof hh, hipMemcpyDeviceToHost);
        fprintf(stderr, "[align_all deferred] hdr:");
        for (int k = 0; k < PH_WORDS; k++) fprintf(stderr, " %d", hh[k]);
        fprintf(stderr, "  n=%lld max_len=%d quad_lds=%zu\n", (long long)n, ctx->max_len, quad_lds);
      }
      if (stage_begin(ctx, STG_TRACE, ps)) return MIA_HIP_ERR_NOMEM;
      hipLaunchKernelGGL(k_align_quad, dim3(qgrid), dim3(64), quad_lds, ps, ctx->rs, ref, ctx->d_pssm, ctx->packs.p[0], ctx->d_list, 0,
                         ctx->d_quad_slabs, slab, ctx->d_wide_list, d_wide_count, ctx->d_retry_list, d_retry_cnt, ctx->use_band, ctx->dbg,
                         (const int32_t*)(hdr + PH_QUAD));
      stage_end(ctx, STG_TRACE, ps);
      HIPCHK(hipGetLastError());
      ck("quad trace");
      if (ctx->use_band) {
        hipError_t e = launch_window<4>(ctx, 0, ctx->d_retry_list, 0, hdr + PH_RETRY, ps);
        if (e != hipSuccess) { ctx->err = std::string("k_align_window retry launch: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
      }
    }
    ck("retry");
    if (int rcj = bx_join_and_retry(ctx)) return rcj;
    ck("band join");
    // the one look at the counters: wide / retry counts, the planner's header, filter and band-pipeline counters lie side by
    // side in the control block
    constexpr int C0 = CTRL_BINS + 3 * N_BINS, CN = CTRL_WORDS - C0;
    std::vector<int32_t> pageable;
    int32_t* hb;
    if (ctx->h_pin) hb = reinterpret_cast<int32_t*>(ctx->h_pin);
    else { pageable.resize(CN); hb = pageable.data(); }
    const bool pre_cull = ctx->in_iterate && ctx->comm;
    if (pre_cull) { if (int rcp = comm_pre_cull_enqueue(ctx, d_wide_count)) return rcp; }
    HIPCHK(hipMemcpyAsync(hb, ctx->d_ctrl + C0, (size_t)CN * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->spec_ok && ctx->h_pin && !pre_cull && !dbg_steps) {
      // mia_hip_iterate, one context, a cut line that needs no scores on the host: no wait here.  The caller queues cull,
      // tally and consensus behind this copy; their kernels look at the exact-kernel count themselves (abort_if) and
      // iterate_body reads these counters when it waits for the consensus (align_counters_collect).
      ctx->spec_pending = true; ctx->spec_filtered = filtered != 0; ctx->spec_bx = bx; ctx->spec_plain = use_plain && ctx->use_quad;
      ctx->abort_if = ctx->spec_force ? ctx->d_one : d_wide_count;      // (MIA_HIP_SPEC_TEST=1: every iteration takes the second round)
      ctx->aligned = true; ctx->culled = false; ctx->tallied = false; ctx->pre_cull_valid = false;
      return MIA_HIP_OK;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    align_counters_collect(ctx, hb, filtered != 0, bx, use_plain && ctx->use_quad);
    const int32_t n_wide = hb[0];
    if (n_wide > 0) { if (int rcw = run_wide(ctx, ref, n_wide)) return rcw; }
    if (pre_cull && comm_pre_cull_collect(ctx)) {
      // some rank had reads for the exact kernel: its sums were taken before their scores were final.  Every rank saw the
      // same eight words per rank, so every rank comes here and the all-gather is repeated with the final numbers.
      if (int rcp = comm_pre_cull_enqueue(ctx, nullptr)) return rcp;
      HIPCHK(hipStreamSynchronize(ctx->stream));
      (void)comm_pre_cull_collect(ctx);
    }
    ctx->aligned = true; ctx->culled = false; ctx->tallied = false; ctx->pre_cull_valid = false;
    return MIA_HIP_OK;
  }
  // host copies of the counters live in pinned memory when there is some: a copy to or from pageable memory makes the
  // host wait for the stream even when it is called "async"
  int32_t local_buf[4 * N_BINS + 16];
  int32_t* hb = ctx->h_pin ? reinterpret_cast<int32_t*>(ctx->h_pin) : local_buf;
  int32_t *h_count = hb, *h_off = hb + N_BINS, *h_count2 = hb + 2 * N_BINS, *h_off2 = hb + 3 * N_BINS, *h_misc = hb + 4 * N_BINS;
  h_misc[0] = 0;
  h_misc[6] = 0;
  std::vector<uint32_t> bxc_pageable;
  uint32_t* h_bxc;                                  // all counters of the band pipeline (8 KB), read once the stream is waited for anyway
  if (ctx->h_pin) h_bxc = reinterpret_cast<uint32_t*>(ctx->h_pin + (24 << 10));
  else { bxc_pageable.resize(BXC_WORDS); h_bxc = bxc_pageable.data(); }
  for (int k = 0; k < BXC_COUNTERS; k++) h_bxc[k * BXC_STRIDE] = 0;
  if (filtered) {
    HIPCHK(hipMemcpyAsync(&h_misc[0], ctx->d_filter_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&h_misc[6], ctx->d_filter_n + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (bx) HIPCHK(hipMemcpyAsync(h_bxc, ctx->d_bx_ctr, BXC_WORDS * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipMemcpyAsync(h_count, d_count, (size_t)N_BINS * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  h_filter_n = (uint32_t)h_misc[0];
  ctx->filter_proven += h_filter_n;
  ctx->band_done += (uint32_t)h_misc[6];
  ctx->filter_seen += n;
  // (with the band kernels still running beside this stream their counters are read behind the join, below)
  const bool bx_counters_later = ctx->bx_pending_join;
  auto take_bx_counters = [&]() {
    ctx->filter_proven += h_bxc[BXC_DONE_PLAN * BXC_STRIDE];          // finished without any DP: by the diagonal filter or by the band plan
    for (int k = 0; k < 3; k++) ctx->bx_done[k] += h_bxc[(BXC_DONE_PLAN + k) * BXC_STRIDE];
    ctx->bx_seen += h_bxc[BXC_SEEN * BXC_STRIDE];
    for (int k = 0; k < BXC_COUNTERS; k++) ctx->bx_last[k] = h_bxc[k * BXC_STRIDE];
  };
  if (!bx_counters_later) take_bx_counters();
  int run = 0;
  for (int b = 0; b < N_BINS; b++) {
    h_off[b] = run;
    run += b >= BIN_QUAD0 ? ((h_count[b] + 3) & ~3) : h_count[b];   // quad bins: every read length padded to whole quads
  }
  const int quad_begin = h_off[BIN_QUAD0], n_quads = (run - quad_begin) / 4;
  if (n_quads > 0) HIPCHK(hipMemsetAsync(ctx->d_list + quad_begin, 0xFF, (size_t)(run - quad_begin) * 4, ctx->stream));   // -1 = empty slot
  HIPCHK(hipMemcpyAsync(d_off, h_off, (size_t)N_BINS * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_plan_fill, dim3(gb), dim3(tb), 0, ctx->stream, n, ctx->d_bin_of, d_off, d_cursor, ctx->d_list);
  // reads that need the exact kernel from the start: copy their list to the head of wide_list
  const int n_wide0 = h_count[BIN_WIDE];
  if (n_wide0 > 0) {
    HIPCHK(hipMemcpyAsync(ctx->d_wide_list, ctx->d_list + h_off[BIN_WIDE], (size_t)n_wide0 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_wide_count, &h_count[BIN_WIDE], 4, hipMemcpyHostToDevice, ctx->stream));
  }
  for (int ci = 0; ci < N_CPL; ci++) {
    if (h_count[ci] == 0) continue;
    hipError_t e = ci == 0 ? launch_window<4>(ctx, ci, ctx->d_list + h_off[ci], h_count[ci])
                 : ci == 1 ? launch_window<8>(ctx, ci, ctx->d_list + h_off[ci], h_count[ci])
                           : launch_window<12>(ctx, ci, ctx->d_list + h_off[ci], h_count[ci]);
    if (e != hipSuccess) { ctx->err = std::string("k_align_window launch: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
  }
  int n_quads_trace = n_quads, quad_begin_trace = quad_begin;
  bool wide_known = false;
  if ((n_quads > 0 || filtered) && use_plain) {
    // first pass: values only; reads whose alignment is provably the pure diagonal are finished there
    if (n_quads > 0) {
    const int grid = n_quads < ctx->quad_wgs ? n_quads : ctx->quad_wgs;
    const size_t quad_lds = (size_t)Q_G * q_sub_bytes(ctx->max_len) + 16;
    if (stage_begin(ctx, STG_PLAIN)) return MIA_HIP_ERR_NOMEM;
    hipLaunchKernelGGL(k_align_quad_plain, dim3(grid), dim3(64), quad_lds, ctx->stream, ctx->rs, ref, ctx->d_pssm, ctx->d_list + quad_begin,
                       n_quads, ctx->d_bin_of, (const int32_t*)nullptr);
    stage_end(ctx, STG_PLAIN);
    HIPCHK(hipGetLastError());
    }
    // re-plan what is left (and what the filter's gap hint kept out of the first pass) into quads of equal read length
    HIPCHK(hipMemsetAsync(d_count, 0, (size_t)N_BINS * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(d_cursor, 0, (size_t)N_BINS * 4, ctx->stream));
    hipLaunchKernelGGL(k_plan_recount, dim3(gb), dim3(tb), 0, ctx->stream, n, ctx->d_bin_of, d_count);
    HIPCHK(hipMemcpyAsync(h_count2, d_count, (size_t)N_BINS * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int run2 = 0;
    for (int b = 0; b < N_BINS; b++) { h_off2[b] = run2; run2 += b >= BIN_QUAD0 ? ((h_count2[b] + 3) & ~3) : 0; }
    n_quads_trace = run2 / 4;
    quad_begin_trace = 0;
    ctx->plain_total += (int64_t)n_quads * 4;
    for (int b = BIN_QUAD0; b < N_BINS; b++) ctx->plain_retried += h_count2[b];
    if (n_quads_trace > 0) {
      HIPCHK(hipMemsetAsync(ctx->d_list, 0xFF, (size_t)run2 * 4, ctx->stream));
      HIPCHK(hipMemcpyAsync(d_off, h_off2, (size_t)N_BINS * 4, hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(k_plan_fill, dim3(gb), dim3(tb), 0, ctx->stream, n, ctx->d_bin_of, d_off, d_cursor, ctx->d_list);
    }
  }
  if (n_quads_trace > 0) {
    const int n_quads = n_quads_trace, quad_begin = quad_begin_trace;
    const int64_t slab = (int64_t)Q_G * MAX_READ * Q_TRACE_STRIDE;
    const int grid = n_quads < ctx->quad_wgs ? n_quads : ctx->quad_wgs;
    if (!ctx->d_quad_slabs && hipMalloc((void**)&ctx->d_quad_slabs, (size_t)slab * ctx->quad_wgs) != hipSuccess) return MIA_HIP_ERR_NOMEM;
    const size_t quad_lds = (size_t)Q_G * q_sub_bytes(ctx->max_len) + 16;
    if (stage_begin(ctx, STG_TRACE)) return MIA_HIP_ERR_NOMEM;
    hipLaunchKernelGGL(k_align_quad, dim3(grid), dim3(64), quad_lds, ctx->stream, ctx->rs, ref, ctx->d_pssm, ctx->packs.p[0], ctx->d_list + quad_begin,
                       n_quads, ctx->d_quad_slabs, slab, ctx->d_wide_list, d_wide_count, ctx->d_retry_list, d_retry_count, ctx->use_band,
                       ctx->dbg, (const int32_t*)nullptr);
    stage_end(ctx, STG_TRACE);
    HIPCHK(hipGetLastError());
    if (ctx->use_band) {
      // reads whose path left the stored trace band: one-read kernel with the full trace (windows <= 208 fit class 0)
      // (the two counters sit side by side: one copy, one wait; the wide count is final unless the retry kernel runs)
      HIPCHK(hipMemcpyAsync(&h_misc[2], d_wide_count, 8, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      const int32_t n_retry = h_misc[3];
      if (n_retry > 0) {
        hipError_t e = launch_window<4>(ctx, 0, ctx->d_retry_list, n_retry);
        if (e != hipSuccess) { ctx->err = std::string("k_align_window retry launch: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
      } else {
        wide_known = true;
      }
    }
  }
  if (ctx->bx_pending_join) { if (int rcj = bx_join_and_retry(ctx)) return rcj; wide_known = false; }
  // exact kernel for whole-reference windows and escaped reads
  if (!wide_known) {
    HIPCHK(hipMemcpyAsync(&h_misc[2], d_wide_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (bx_counters_later) HIPCHK(hipMemcpyAsync(h_bxc, ctx->d_bx_ctr, BXC_WORDS * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (bx_counters_later) take_bx_counters();
  }
  const int32_t n_wide = h_misc[2];
  if (n_wide > 0) { if (int rcw = run_wide(ctx, ref, n_wide)) return rcw; }
  ctx->aligned = true;
  ctx->culled = false;
  ctx->tallied = false;
  ctx->pre_cull_valid = false;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_align_windows(mia_hip_ctx* ctx, const char* ref, int64_t ref_len, const int64_t* win_start, const int32_t* win_len) {
  if (!ctx || !ref || !win_start || !win_len || ref_len <= 0) return MIA_HIP_ERR_ARG;
  if (!ctx->have_pssm || !ctx->d_packed) { ctx->err = "set_pssm and upload_reads must precede align_windows"; return MIA_HIP_ERR_STATE; }
  if (ref_len > INT32_MAX - 128) { ctx->err = "align_windows: the reference string must stay below 2^31 characters"; return MIA_HIP_ERR_ARG; }
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = ctx->rs.n;
  std::vector<int32_t> as((size_t)n), ae((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    if (win_len[i] <= 0 || win_start[i] < 0 || win_start[i] + win_len[i] > ref_len) {
      ctx->err = "align_windows: window " + std::to_string(i) + " is empty or leaves the reference string";
      return MIA_HIP_ERR_ARG;
    }
    as[(size_t)i] = (int32_t)win_start[i];
    ae[(size_t)i] = (int32_t)(win_start[i] + win_len[i] - 1);
  }
  std::vector<uint8_t> codes((size_t)ref_len + 64, 4);
  for (int64_t i = 0; i < ref_len; i++) codes[(size_t)i] = base_code(ref[i]);
  if ((int64_t)codes.size() > ctx->ref_cap) {
    if (dev_alloc(ctx, &ctx->d_ref, codes.size())) return MIA_HIP_ERR_NOMEM;
    ctx->ref_cap = (int)codes.size();
  }
  HIPCHK(hipMemcpyAsync(ctx->d_ref, codes.data(), codes.size(), hipMemcpyHostToDevice, ctx->stream));
  if (n) {
    HIPCHK(hipMemcpyAsync(ctx->d_as, as.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_ae, ae.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  // not a reference the consensus path can use: cull / tally need a realign first
  ctx->L = (int)ref_len; ctx->wrap = (int)ref_len; ctx->have_ref = false; ctx->explicit_win = 1; ctx->ref_mostly_bases = true; ctx->kh_entries = 0;
  const int rcode = align_all(ctx);
  HIPCHK(hipStreamSynchronize(ctx->stream));   // as / ae / codes are host buffers of this call
  return rcode;
}

extern "C" int mia_hip_get_alignments(mia_hip_ctx* ctx, int32_t* score, int32_t* as, int32_t* ae) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  if (!ctx->aligned) { ctx->err = "realign first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->rs.n;
  if (score) HIPCHK(hipMemcpyAsync(score, ctx->d_score, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (as) HIPCHK(hipMemcpyAsync(as, ctx->d_as, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (ae) HIPCHK(hipMemcpyAsync(ae, ctx->d_ae, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

extern "C" int mia_hip_get_scripts(mia_hip_ctx* ctx, int16_t* cols, int32_t stride, int32_t* ref_start) {
  if (!ctx || (cols && stride < ctx->max_len)) return MIA_HIP_ERR_ARG;
  if (!ctx->aligned) { ctx->err = "realign first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->rs.n;
  if (cols) {
    if (ctx->diag_scripts_missing && n > 0) {
      hipLaunchKernelGGL(k_diag_scripts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->rs);
      HIPCHK(hipGetLastError());
      ctx->diag_scripts_missing = false;
    }
    HIPCHK(hipMemcpy2DAsync(cols, (size_t)stride * 2, ctx->d_cols, (size_t)ctx->rs.stride * 2, (size_t)ctx->max_len * 2, n,
                            hipMemcpyDeviceToHost, ctx->stream));
  }
  if (ref_start) HIPCHK(hipMemcpyAsync(ref_start, ctx->d_refstart, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

// ---- cull ----------------------------------------------------------------------
static int finish_params(mia_hip_ctx* ctx);
// second half of the cull: every link (own or gathered) acts on the slot it points at, then each read's dropped bits,
// depth-code parameters and multiplicities are final.  Runs at the end of mia_hip_cull and again after mia_hip_set_links.
static int finish_cull(mia_hip_ctx* ctx) {
  if (ctx->links_cap_all > ctx->link_len_cap) {
    if (dev_alloc(ctx, &ctx->d_link_len, (size_t)ctx->links_cap_all + 64) || dev_alloc(ctx, &ctx->d_link_act, (size_t)ctx->links_cap_all + 64))
      return MIA_HIP_ERR_NOMEM;
    ctx->link_len_cap = ctx->links_cap_all + 64;
  }
  hipLaunchKernelGGL(k_links_apply, dim3(32), dim3(256), 0, ctx->stream, ctx->d_links_all, ctx->d_n_links_all, (int32_t)ctx->links_cap_all, ctx->si,
                     ctx->d_slot_dropped, ctx->n_slots, ctx->d_link_len, ctx->d_link_act, ctx->d_cull_flags, ctx->abort_if);
  ctx->links_applied = true;
  return finish_params(ctx);
}

static int finish_params(mia_hip_ctx* ctx) {
  const int64_t n = ctx->rs.n;
  hipLaunchKernelGGL(k_rec_params, dim3((int)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->rs, ctx->L, ctx->d_slot, ctx->d_slot_dropped,
                     ctx->n_slots, ctx->d_back_slot, ctx->ri, ctx->si, ctx->d_links_all, ctx->d_link_len, ctx->d_link_act, ctx->d_n_links_all,
                     (int32_t)ctx->links_cap_all, ctx->read_base, ctx->d_drop_f, ctx->d_drop_b, ctx->d_cull_flags, ctx->abort_if);
  HIPCHK(hipGetLastError());
  return MIA_HIP_OK;
}

// the error flags of the cull kernels, checked where the stream is synchronised anyway
static int check_cull_flags(mia_hip_ctx* ctx, uint32_t fl) {
  if (fl & 8u) { ctx->err = "more formerly split reads than the link list holds"; return MIA_HIP_ERR_RANGE; }
  if (fl & 4u) {
    ctx->err = "a formerly split read points at an AlnSeq slot that holds no record of this iteration (the reference would show "
               "the slot's content from an earlier iteration); not reproduced";
    return MIA_HIP_ERR_RANGE;
  }
  return MIA_HIP_OK;
}

extern "C" int mia_hip_cull(mia_hip_ctx* ctx, int32_t hard_cut, double slope, double intercept, int64_t slot_base) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  if (!ctx->aligned) { ctx->err = "realign first"; return MIA_HIP_ERR_STATE; }
  if (ctx->explicit_win) { ctx->err = "the last alignment ran on caller-supplied windows (align_windows): realign first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = ctx->rs.n;
  if (n == 0) return MIA_HIP_OK;
  // (two launches: record counts per block of 256 reads, scanned by the block that finishes last; then slots, record
  // geometry, dropped marks and links of every read in one kernel -- k_slot_count / k_cull_records)
  const int nb = (int)((n + 255) / 256);
  if (!ctx->d_cull_sync) {
    if (dev_alloc(ctx, &ctx->d_cull_sync, 4)) return MIA_HIP_ERR_NOMEM;
    HIPCHK(hipMemsetAsync(ctx->d_cull_sync, 0, 16, ctx->stream));
  }
  hipLaunchKernelGGL(k_slot_count, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, ctx->stream, ctx->rs, ctx->L, ctx->d_partial, nb, slot_base, ctx->d_total, (uint32_t*)nullptr, ctx->abort_if);
  hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, ctx->stream, ctx->d_partial, nb, slot_base, ctx->d_total, ctx->abort_if);
  if (slot_base + 2 * n + 16 > ctx->n_slots) {   // sharded runs: slots are global indices
    uint8_t* nd = nullptr;
    const int64_t ns = slot_base + 2 * n + 16;
    if (hipMalloc((void**)&nd, (size_t)ns) != hipSuccess) return MIA_HIP_ERR_NOMEM;
    HIPCHK(hipMemsetAsync(nd, 0, (size_t)ns, ctx->stream));
    HIPCHK(hipMemcpyAsync(nd, ctx->d_slot_dropped, (size_t)ctx->n_slots, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->d_slot_dropped);
    ctx->d_slot_dropped = nd;
    ctx->n_slots = ns;
  }
  // records per read and per slot, the reads' own dropped marks, and the links of formerly split reads -- all without a
  // host round trip: the record total and the link count stay on the device, the error flags are read by mia_hip_tally
  ctx->slot_base = slot_base;
  ctx->si.base = slot_base;
  ctx->si.n_local_p = ctx->d_total;
  if (!ctx->in_iterate) HIPCHK(hipMemsetAsync(ctx->lk.n, 0, 8, ctx->stream));                     // link count, cull flags (neighbours in the control block)
  hipLaunchKernelGGL(k_cull_records, dim3((unsigned)nb), dim3(256), 0, ctx->stream, ctx->rs, ctx->L, (const int64_t*)ctx->d_partial, ctx->d_slot, ctx->ri, ctx->si,
                     ctx->read_base, ctx->d_cull_flags, ctx->d_slot_dropped, ctx->n_slots, hard_cut, slope, intercept, ctx->d_back_slot,
                     (const int64_t*)ctx->d_front_slot0, ctx->lk, ctx->dev_cut, ctx->abort_if);
  HIPCHK(hipGetLastError());
  // by default the links to apply are this context's own; a sharded run replaces them with the gathered list (mia_hip_set_links)
  ctx->d_links_all = ctx->lk.rec;
  ctx->d_n_links_all = ctx->lk.n;
  ctx->links_cap_all = ctx->lk.cap;
  ctx->culled = true;
  ctx->links_applied = false;
  return finish_cull(ctx);
}

extern "C" int mia_hip_set_back_slots(mia_hip_ctx* ctx, const int64_t* back_slot) {
  if (ctx) ctx->pre_cull_valid = false;
  if (!ctx || !back_slot) return MIA_HIP_ERR_ARG;
  if (!ctx->d_back_slot) { ctx->err = "upload_reads first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(ctx->d_back_slot, back_slot, (size_t)ctx->rs.n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

extern "C" int mia_hip_set_pass1_state(mia_hip_ctx* ctx, const int64_t* front_slot, const int64_t* back_slot, const int32_t* score) {
  if (ctx) ctx->pre_cull_valid = false;
  if (!ctx) return MIA_HIP_ERR_ARG;
  if (!ctx->d_back_slot) { ctx->err = "upload_reads first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->rs.n;
  if (front_slot) HIPCHK(hipMemcpyAsync(ctx->d_front_slot0, front_slot, n * 8, hipMemcpyHostToDevice, ctx->stream));
  if (back_slot) HIPCHK(hipMemcpyAsync(ctx->d_back_slot, back_slot, n * 8, hipMemcpyHostToDevice, ctx->stream));
  if (score) HIPCHK(hipMemcpyAsync(ctx->d_score, score, n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

extern "C" int mia_hip_set_read_base(mia_hip_ctx* ctx, int64_t read_base) {
  if (!ctx || read_base < 0) return MIA_HIP_ERR_ARG;
  ctx->read_base = read_base;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_links(mia_hip_ctx* ctx, int64_t** d_links, int64_t* n_links) {
  if (!ctx || !ctx->culled) return MIA_HIP_ERR_STATE;
  int32_t nl = 0;
  HIPCHK(hipSetDevice(ctx->device));
  // on the context's own (non-blocking) stream, behind the k_cull_mark that counts the links; once it has been waited for,
  // the list itself is complete too and the caller's collective may read it from any stream
  HIPCHK(hipMemcpyAsync(&nl, ctx->lk.n, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (d_links) *d_links = ctx->lk.rec;
  if (n_links) *n_links = nl;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_set_links(mia_hip_ctx* ctx, const int64_t* d_links_all, int64_t n_all) {
  if (!ctx || n_all < 0 || (n_all > 0 && !d_links_all)) return MIA_HIP_ERR_ARG;
  if (!ctx->culled) { ctx->err = "cull first"; return MIA_HIP_ERR_STATE; }
  if (n_all >= (int64_t)LINK_NONE) { ctx->err = "too many links"; return MIA_HIP_ERR_RANGE; }
  HIPCHK(hipSetDevice(ctx->device));
  // own copy of the gathered list (the caller's buffer may be a temporary of the collective)
  if (n_all > ctx->links_all_cap) {
    int64_t* nb = nullptr;
    if (hipMalloc((void**)&nb, (size_t)(n_all + 64) * 32) != hipSuccess) return MIA_HIP_ERR_NOMEM;
    ctx->links_all_cap = n_all + 64;
    ctx->d_links_all = nb;          // (the previous gathered buffer, if any, is released with the context)
    ctx->owned_links.push_back(nb);
  } else if (!ctx->owned_links.empty()) ctx->d_links_all = ctx->owned_links.back();
  if (n_all > 0) HIPCHK(hipMemcpyAsync(ctx->d_links_all, d_links_all, (size_t)n_all * 32, hipMemcpyDeviceToDevice, ctx->stream));
  ctx->n_links_all = (int32_t)n_all;
  if (!ctx->d_n_links_gathered && dev_alloc(ctx, &ctx->d_n_links_gathered, 1)) return MIA_HIP_ERR_NOMEM;
  const int32_t n32 = (int32_t)n_all;
  HIPCHK(hipMemcpyAsync(ctx->d_n_links_gathered, &n32, 4, hipMemcpyHostToDevice, ctx->stream));
  ctx->d_n_links_all = ctx->d_n_links_gathered;
  ctx->links_cap_all = n_all;
  // slot state back to "owners only", then every link once
  const int64_t n = ctx->rs.n;
  HIPCHK(hipMemsetAsync(ctx->d_cull_flags, 0, 4, ctx->stream));
  hipLaunchKernelGGL(k_rec_geom, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->rs, ctx->L, ctx->d_slot, ctx->ri, ctx->si,
                     ctx->read_base, ctx->d_cull_flags);
  HIPCHK(hipGetLastError());
  const int rc_fc = finish_cull(ctx);
  HIPCHK(hipStreamSynchronize(ctx->stream));   // n32 above is a stack local; the caller's buffer may be a temporary
  return rc_fc;
}

extern "C" int mia_hip_link_lengths(mia_hip_ctx* ctx, int32_t** d_len, int32_t** d_act, int64_t* n) {
  if (!ctx || !ctx->links_applied) return MIA_HIP_ERR_STATE;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));   // k_links_apply fills both buffers; the caller reduces them on a stream of its own
  if (d_len) *d_len = ctx->d_link_len;
  if (d_act) *d_act = ctx->d_link_act;
  if (n) *n = ctx->n_links_all;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_finish_links(mia_hip_ctx* ctx) {
  if (!ctx || !ctx->links_applied) return MIA_HIP_ERR_STATE;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemsetAsync(ctx->d_cull_flags, 0, 4, ctx->stream));
  return finish_params(ctx);
}

extern "C" int mia_hip_get_record_params(mia_hip_ctx* ctx, int32_t* params, int64_t* back_slot) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  if (!ctx->culled) { ctx->err = "cull first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->rs.n;
  if (params) HIPCHK(hipMemcpy2DAsync(params, 32, ctx->ri.trec + TREC_PARAMS, 64, 32, n, hipMemcpyDeviceToHost, ctx->stream));   // words 8..15 of each record
  if (back_slot) HIPCHK(hipMemcpyAsync(back_slot, ctx->d_back_slot, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

extern "C" int mia_hip_get_dropped(mia_hip_ctx* ctx, uint8_t* front, uint8_t* back) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n = (size_t)ctx->rs.n;
  if (front) HIPCHK(hipMemcpyAsync(front, ctx->d_drop_f, n, hipMemcpyDeviceToHost, ctx->stream));
  if (back) HIPCHK(hipMemcpyAsync(back, ctx->d_drop_b, n, hipMemcpyDeviceToHost, ctx->stream));
  uint32_t cflags = 0;
  if (ctx->culled) HIPCHK(hipMemcpyAsync(&cflags, ctx->d_cull_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return check_cull_flags(ctx, cflags);
}

extern "C" int mia_hip_set_slot_dropped(mia_hip_ctx* ctx, const uint8_t* flags, int64_t n_flags) {
  if (!ctx || !flags || n_flags < 0) return MIA_HIP_ERR_ARG;
  if (!ctx->d_slot_dropped) { ctx->err = "upload_reads first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  if (n_flags > ctx->n_slots) {            // a shard is handed the marks of ALL slots: its own lie behind the other ranks'
    if (dev_alloc(ctx, &ctx->d_slot_dropped, (size_t)n_flags + 16)) return MIA_HIP_ERR_NOMEM;
    ctx->n_slots = n_flags + 16;
  }
  HIPCHK(hipMemsetAsync(ctx->d_slot_dropped, 0, (size_t)ctx->n_slots, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_slot_dropped, flags, (size_t)n_flags, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

// find_fsdb_score_cut (src/fsdb.c:269-383): host-side IEEE double (built with -ffp-contract=off).
// Bit-exactness argument per pass:
//   1. xbar/ybar: every partial sum is an integer below 2^53, so each double addition is exact and
//      the result does not depend on the order -> accumulate in int64.
//   2. ssxy/ssxx: sums of inexact products, order dependent -> sequential, in fsdb order, exactly as
//      the reference.  If all reads have the same length every (len - xbar) is exactly 0, both sums
//      are exactly 0 and slope_bf = 0/0 = NaN: that case is answered without the pass.
//   3. max slope delta: a maximum is order independent (NaN deltas never win, as in the reference).
extern "C" void mia_hip_score_cut(const int32_t* score, const int32_t* seq_len, const uint8_t* unique_best, int64_t n,
                                  double* slope, double* intercept) {
  int64_t sx = 0, sy = 0, j = 0;
  int32_t lmin = INT32_MAX, lmax = INT32_MIN;
  for (int64_t i = 0; i < n; i++) {
    const bool u = (!unique_best || unique_best[i]) && score[i] >= 2000;
    sx += u ? seq_len[i] : 0;
    sy += u ? score[i] : 0;
    j += u;
    if (u) { lmin = seq_len[i] < lmin ? seq_len[i] : lmin; lmax = seq_len[i] > lmax ? seq_len[i] : lmax; }
  }
  double xbar = (double)sx, ybar = (double)sy;
  xbar /= (double)(size_t)j;
  ybar /= (double)(size_t)j;
  double slope_bf, intercept_bf, max_delta = 0;
  if (j > 0 && lmin == lmax) {
    const double zero = 0.0;
    slope_bf = zero / zero;                      // ssxy / ssxx with both sums exactly 0
    intercept_bf = ybar - slope_bf * xbar;
  } else {
    double ssxy = 0, ssxx = 0;
    auto used = [&](int64_t i) { return (!unique_best || unique_best[i]) && score[i] >= 2000; };
    for (int64_t i = 0; i < n; i++) if (used(i)) {
      ssxy += (seq_len[i] - xbar) * (score[i] - ybar);
      ssxx += (seq_len[i] - xbar) * (seq_len[i] - xbar);
    }
    slope_bf = ssxy / ssxx;
    intercept_bf = ybar - slope_bf * xbar;
    double m4[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; i++) if (used(i)) {
      double d = (score[i] - ((slope_bf * seq_len[i]) + intercept_bf)) / seq_len[i];
      if (d > m4[i & 3]) m4[i & 3] = d;
    }
    for (int k = 0; k < 4; k++) if (m4[k] > max_delta) max_delta = m4[k];
  }
  *intercept = intercept_bf;
  if ((slope_bf - max_delta) > 0) *slope = slope_bf - (max_delta * 2.0);
  else *slope = (double)(slope_bf * (80 / 100.0));
}

extern "C" int mia_hip_score_sums(mia_hip_ctx* ctx, int64_t* sums5) {
  if (!ctx || !sums5) return MIA_HIP_ERR_ARG;
  if (!ctx->aligned) { ctx->err = "realign first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->d_sums && dev_alloc(ctx, &ctx->d_sums, 8)) return MIA_HIP_ERR_NOMEM;
  ctx->pre_cull_valid = false;
  hipLaunchKernelGGL(k_score_sums_init, dim3(1), dim3(64), 0, ctx->stream, ctx->d_sums, (const int32_t*)nullptr);
  const int64_t n = ctx->rs.n;
  if (n > 0) {
    int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cus);   // few blocks: five same-address atomics per block
    hipLaunchKernelGGL(k_score_sums, dim3(grid), dim3(256), 0, ctx->stream, ctx->rs, ctx->d_sums, ctx->L, ctx->d_back_slot, ctx->d_front_slot0);
    HIPCHK(hipGetLastError());
  }
  int64_t local7[7];
  int64_t* stage = ctx->h_pin ? reinterpret_cast<int64_t*>(ctx->h_pin + (16 << 10)) : local7;   // pinned: the copy does not block by itself
  HIPCHK(hipMemcpyAsync(stage, ctx->d_sums, 56, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  memcpy(sums5, stage, 40);
  ctx->pre_cull_records = stage[5];
  ctx->pre_cull_links = stage[6];
  ctx->pre_cull_valid = true;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_pre_cull_counts(mia_hip_ctx* ctx, int64_t* n_records, int64_t* n_links) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  if (!ctx->pre_cull_valid) { ctx->err = "score_sums (after the last realign) first"; return MIA_HIP_ERR_STATE; }
  if (n_records) *n_records = ctx->pre_cull_records;
  if (n_links) *n_links = ctx->pre_cull_links;
  return MIA_HIP_OK;
}

// HOST helper: what find_fsdb_score_cut returns when every used read has the same length -- both regression sums
// are exactly 0, slope_bf = 0/0 -- from the (possibly all-reduced) sums alone.  Returns 1 if the lengths differ: the
// order-dependent double sums of passes 2 and 3 then need the scores on the host (mia_hip_score_cut).
extern "C" int mia_hip_score_cut_from_sums(const int64_t* sums5, double* slope, double* intercept) {
  const int64_t sx = sums5[0], sy = sums5[1], j = sums5[2];
  if (!(j > 0 && sums5[3] == sums5[4])) return 1;
  double xbar = (double)sx, ybar = (double)sy;
  xbar /= (double)(size_t)j;
  ybar /= (double)(size_t)j;
  const double zero = 0.0;
  const double slope_bf = zero / zero;
  const double intercept_bf = ybar - slope_bf * xbar;
  const double max_delta = 0;
  *intercept = intercept_bf;
  if ((slope_bf - max_delta) > 0) *slope = slope_bf - (max_delta * 2.0);
  else *slope = (double)(slope_bf * (80 / 100.0));
  return 0;
}

extern "C" int mia_hip_num_records(mia_hip_ctx* ctx, int64_t* n_records) {
  if (!ctx || !n_records) return MIA_HIP_ERR_ARG;
  if (!ctx->aligned) { ctx->err = "realign first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = ctx->rs.n;
  *n_records = 0;
  if (n == 0) return MIA_HIP_OK;
  const int nb = (int)((n + 4095) / 4096);
  hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(256), 0, ctx->stream, ctx->rs, ctx->L, ctx->d_partial);
  hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, ctx->stream, ctx->d_partial, nb, (int64_t)0, ctx->d_total);
  HIPCHK(hipMemcpyAsync(n_records, ctx->d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

// ---- tally + consensus -------------------------------------------------------------
static int ensure_tally(mia_hip_ctx* ctx) {
  const int Lp = ctx->L + 1;
  if (Lp > ctx->tally_cap) {
    int rc = 0;
    rc |= dev_alloc(ctx, &ctx->tb.tally, (size_t)(TALLY_WORDS + 1) * Lp + 256);      // tally[12][Lp], then gaps[Lp] (+ one slot per rank of a sharded run): one memset clears all
    rc |= dev_alloc(ctx, &ctx->d_ins_off, (size_t)Lp);
    rc |= dev_alloc(ctx, &ctx->d_calls, (size_t)Lp);
    if (rc) return MIA_HIP_ERR_NOMEM;
    ctx->tally_cap = Lp;
  }
  if (!ctx->tb.events) {
    // every inserted read base is one event; 1/8 of all bases is far beyond any real data,
    // overflow is detected and reported
    int64_t cap = ctx->rs.n * 32 + 4096;
    if (cap > ((int64_t)1 << 30)) cap = (int64_t)1 << 30;
    int rc = 0;
    rc |= dev_alloc(ctx, &ctx->tb.events, (size_t)cap);
    if (rc) return MIA_HIP_ERR_NOMEM;
    ctx->tb.cap_events = (int32_t)cap;
  }
  ctx->tb.Lp = Lp;
  ctx->tb.gaps = ctx->tb.tally + (size_t)TALLY_WORDS * Lp;      // right behind the tally of THIS reference length
  return MIA_HIP_OK;
}

// everything of mia_hip_tally that is queued on the stream; the event count and the error flags are read afterwards
// the binned tally's layout for this reference: buckets of TALLY_BUCKET columns, one workgroup per TALLY_CHUNK reads of a bucket
static bool tally_is_binned(const mia_hip_ctx* ctx) {
  const int nb = ctx->wrap / TALLY_BUCKET + 1;
  return ctx->rs.n > 0 && ctx->use_binned_tally && nb <= 4096 && ctx->max_abs <= 32767;   // (the LDS copy of the matrices is int16)
}
// Counting sort of the reads by alignment start (k_bucket_count / _scan / _fill) on stream `on`.  It reads nothing but the
// alignment starts, so mia_hip_iterate queues it on stream2 BESIDE the cull kernels (it also clears the tally buffers:
// nothing adds to them before the tally kernel).  tally_launch waits for ev_join if `on` is not the context's stream.
static int bucket_launch(mia_hip_ctx* ctx, hipStream_t on) {
  int rc = ensure_tally(ctx);
  if (rc) return rc;
  const int Lp = ctx->tb.Lp;
  const int64_t tally_words = (int64_t)(TALLY_WORDS + 1) * Lp + 256;                               // tally, gaps, the ranks' event counts
  const int64_t n = ctx->rs.n;
  const int nb = ctx->wrap / TALLY_BUCKET + 1;
  const int grid = (int)(n / TALLY_CHUNK) + nb + 1;
  if (4 * (nb + 1) + grid > ctx->bucket_cap) {
    if (dev_alloc(ctx, &ctx->d_bucket, (size_t)(4 * (nb + 1) + grid) * 2)) return MIA_HIP_ERR_NOMEM;
    ctx->bucket_cap = (4 * (nb + 1) + grid) * 2;
    ctx->bucket_clean_nb = -1;
  }
  if (!ctx->d_order && dev_alloc(ctx, &ctx->d_order, (size_t)n)) return MIA_HIP_ERR_NOMEM;
  int32_t *d_cnt = ctx->d_bucket, *d_off = d_cnt + (nb + 1), *d_wgoff = d_off + (nb + 1), *d_cur = d_wgoff + (nb + 1), *d_wgb = d_cur + (nb + 1);
  // the bucket counts are left at zero by k_bucket_scan; only a fresh (or differently laid out) buffer is cleared here
  if (ctx->bucket_clean_nb != nb) { HIPCHK(hipMemsetAsync(d_cnt, 0, (size_t)(nb + 1) * 4, on)); ctx->bucket_clean_nb = nb; }
  const int gb = (int)((n + 256 * BUCKET_PER - 1) / (256 * BUCKET_PER));
  hipLaunchKernelGGL(k_bucket_count, dim3(gb), dim3(256), (size_t)nb * 4, on, ctx->rs, nb, d_cnt, ctx->tb.tally, tally_words, ctx->abort_if);
  hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(256), 0, on, d_cnt, nb, d_off, d_wgoff, d_cur, d_wgb, ctx->abort_if);
  hipLaunchKernelGGL(k_bucket_fill, dim3(gb), dim3(256), (size_t)nb * 8, on, ctx->rs, nb, d_off, d_cur, ctx->d_order, ctx->abort_if);
  HIPCHK(hipGetLastError());
  if (on != ctx->stream) HIPCHK(hipEventRecord(ctx->ev_join, on));
  ctx->buckets_queued = on != ctx->stream ? 2 : 1;
  return MIA_HIP_OK;
}

static int tally_launch(mia_hip_ctx* ctx) {
  if (!ctx->aligned) { ctx->err = "realign first"; return MIA_HIP_ERR_STATE; }
  if (!ctx->culled) { ctx->err = "cull first (the dropped bits and record parameters are its output)"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  int rc = ensure_tally(ctx);
  if (rc) return rc;
  const int Lp = ctx->tb.Lp;
  const int64_t tally_words = (int64_t)(TALLY_WORDS + 1) * Lp + 256;                               // tally, gaps, the ranks' event counts
  if (!ctx->in_iterate) HIPCHK(hipMemsetAsync(ctx->tb.n_events, 0, 8, ctx->stream));              // event count, flags (neighbours in the control block)
  const int64_t n = ctx->rs.n;
  const bool binned = tally_is_binned(ctx);
  if (!binned) HIPCHK(hipMemsetAsync(ctx->tb.tally, 0, (size_t)tally_words * 4, ctx->stream));     // (the binned path clears it inside k_bucket_count)
  if (n > 0) {
    RefInfo ref{ctx->d_ref, ctx->L, ctx->wrap};
    const int nb = ctx->wrap / TALLY_BUCKET + 1;
    if (binned) {
      // counting sort of the reads by alignment start, then one LDS tally window per workgroup
      if (!ctx->buckets_queued) { if (int rcb = bucket_launch(ctx, ctx->stream)) return rcb; }
      if (ctx->buckets_queued == 2) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
      ctx->buckets_queued = 0;
      const int grid = (int)(n / TALLY_CHUNK) + nb + 1;
      int32_t *d_cnt = ctx->d_bucket, *d_off = d_cnt + (nb + 1), *d_wgoff = d_off + (nb + 1), *d_cur = d_wgoff + (nb + 1), *d_wgb = d_cur + (nb + 1);
      const int64_t slab_words = (int64_t)grid * (TALLY_WORDS - 1) * TALLY_WIN;
      if (slab_words > ctx->tally_slab_cap) {
        if (dev_alloc(ctx, &ctx->d_tally_slabs, (size_t)slab_words)) return MIA_HIP_ERR_NOMEM;
        ctx->tally_slab_cap = slab_words;
      }
      // (the bit planes and the N marks of the context's own reads: k_read_planes / k_bx_umax at upload)
      const bool planes_ok = ctx->umax_valid && ctx->rs.roff == ctx->d_roff && ctx->d_rplanes && ctx->d_umax;
      if (stage_begin(ctx, STG_TALLY)) return MIA_HIP_ERR_NOMEM;
      if (ctx->tally_linear)
        hipLaunchKernelGGL(k_tally_binned<true>, dim3(grid), dim3(256), 0, ctx->stream, ctx->rs, ref, ctx->d_pssm, ctx->d_drop_f, ctx->d_drop_b,
                           ctx->tb, nb, d_off, d_wgoff, ctx->d_order, ctx->ri.trec, ctx->ri.actf, ctx->d_tally_slabs, ctx->dbg,
                           planes_ok ? ctx->d_rplanes : nullptr, ctx->rplane_words, planes_ok ? ctx->d_umax : nullptr, d_wgb, -1, ctx->abort_if);
      else
        hipLaunchKernelGGL(k_tally_binned<false>, dim3(grid), dim3(256), 0, ctx->stream, ctx->rs, ref, ctx->d_pssm, ctx->d_drop_f, ctx->d_drop_b,
                           ctx->tb, nb, d_off, d_wgoff, ctx->d_order, ctx->ri.trec, ctx->ri.actf, ctx->d_tally_slabs, ctx->dbg,
                           (const uint64_t*)nullptr, 0, (const int32_t*)nullptr, d_wgb, ctx->tally_pk_bias, ctx->abort_if);
      stage_end(ctx, STG_TALLY);
      hipLaunchKernelGGL(k_tally_reduce, dim3((Lp + 255) / 256, TALLY_WORDS - 1), dim3(256), 0, ctx->stream, ctx->tb, nb, d_wgoff, ctx->d_tally_slabs, ctx->abort_if);
    } else {
      hipLaunchKernelGGL(k_tally, dim3((int)((n + 3) / 4)), dim3(256), 0, ctx->stream, ctx->rs, ref, ctx->d_pssm, ctx->d_drop_f,
                         ctx->d_drop_b, ctx->tb, ctx->ri.trec, ctx->ri.actf);
    }
    HIPCHK(hipGetLastError());
  }
  return MIA_HIP_OK;
}

// what the tally kernels left in their counters (event count, overflow and geometry flags, the cull's flags)
static int tally_finish(mia_hip_ctx* ctx, uint32_t n_events, uint32_t flags, uint32_t cflags) {
  ctx->n_events_host = (int32_t)n_events;
  if (int rcf = check_cull_flags(ctx, cflags)) return rcf;
  if (flags & 1u) { ctx->err = "insert event list overflow"; return MIA_HIP_ERR_NOMEM; }
  if (ctx->n_events_host > ctx->tb.cap_events) ctx->n_events_host = ctx->tb.cap_events;
  ctx->tallied = true;
  ctx->consensus_done = false;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_tally(mia_hip_ctx* ctx) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  if (int rc = tally_launch(ctx)) return rc;
  uint32_t local3[3] = {0, 0, 0};
  uint32_t* h3 = ctx->h_pin ? reinterpret_cast<uint32_t*>(ctx->h_pin + (17 << 10)) : local3;   // pinned: three copies, one wait
  HIPCHK(hipMemcpyAsync(&h3[0], ctx->tb.n_events, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&h3[1], ctx->tb.flags, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&h3[2], ctx->d_cull_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return tally_finish(ctx, h3[0], h3[1], h3[2]);
}

extern "C" int mia_hip_tally_buffers(mia_hip_ctx* ctx, int32_t** d_tally, int64_t* n_tally_words, int32_t** d_gaps,
                                     int64_t* n_gaps_words) {
  if (!ctx || !ctx->tallied) return MIA_HIP_ERR_STATE;
  if (d_tally) *d_tally = ctx->tb.tally;
  if (n_tally_words) *n_tally_words = (int64_t)TALLY_WORDS * ctx->tb.Lp;
  if (d_gaps) *d_gaps = ctx->tb.gaps;
  if (n_gaps_words) *n_gaps_words = ctx->tb.Lp;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_ins_events(mia_hip_ctx* ctx, uint64_t** d_events, int64_t* n_events) {
  if (!ctx || !ctx->tallied) return MIA_HIP_ERR_STATE;
  if (d_events) *d_events = ctx->tb.events;
  if (n_events) *n_events = ctx->n_events_host;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_set_ins_events(mia_hip_ctx* ctx, const uint64_t* d_events, int64_t n_events) {
  if (!ctx || !ctx->tallied || n_events < 0) return MIA_HIP_ERR_STATE;
  HIPCHK(hipSetDevice(ctx->device));
  if (n_events > ctx->tb.cap_events) {
    if (dev_alloc(ctx, &ctx->tb.events, (size_t)n_events)) return MIA_HIP_ERR_NOMEM;
    ctx->tb.cap_events = (int32_t)n_events;
  }
  if (n_events > 0 && d_events != ctx->tb.events)
    HIPCHK(hipMemcpyAsync(ctx->tb.events, d_events, (size_t)n_events * 8, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->n_events_host = (int32_t)n_events;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_get_tally(mia_hip_ctx* ctx, int32_t* tally, int32_t* gaps) {
  if (!ctx || !ctx->tallied) return MIA_HIP_ERR_STATE;
  HIPCHK(hipSetDevice(ctx->device));
  const int Lp = ctx->tb.Lp;
  if (tally) HIPCHK(hipMemcpyAsync(tally, ctx->tb.tally, (size_t)TALLY_WORDS * Lp * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (gaps) HIPCHK(hipMemcpyAsync(gaps, ctx->tb.gaps, (size_t)Lp * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

extern "C" int mia_hip_set_tally(mia_hip_ctx* ctx, int32_t ref_len, const int32_t* tally, const int32_t* gaps) {
  if (!ctx || ref_len <= 0 || !tally) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  if (ref_len != ctx->L) {
    // tallies of another reference than the one the reads were aligned to: the alignment state (windows, wrap, cull marks)
    // no longer belongs to them -- whatever needs it must realign first
    ctx->aligned = false; ctx->culled = false; ctx->have_ref = false; ctx->links_applied = false;
  }
  ctx->L = ref_len;
  int rc = ensure_tally(ctx);
  if (rc) return rc;
  const int Lp = ctx->tb.Lp;
  HIPCHK(hipMemcpyAsync(ctx->tb.tally, tally, (size_t)TALLY_WORDS * Lp * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->tb.gaps, 0, ((size_t)Lp + 256) * 4, ctx->stream));      // gaps and the ranks' event-count slots behind them
  if (gaps) HIPCHK(hipMemcpyAsync(ctx->tb.gaps, gaps, (size_t)Lp * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->tb.n_events, 0, 4, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->n_events_host = 0;
  ctx->tallied = true;
  ctx->consensus_done = false;
  return MIA_HIP_OK;
}

// the kernels of mia_hip_consensus: column calls, then the insert columns tallied and called into buffers of `cap` slots.
// events_on_device: the insert-event count is read by k_ins_tally itself (mia_hip_iterate has not seen it yet)
static int consensus_launch(mia_hip_ctx* ctx, int cons_code, int64_t cap, bool columns, bool events_on_device) {
  const int L = ctx->L, Lp = ctx->tb.Lp;
  if (columns) {
    hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, ctx->stream, (const int32_t*)ctx->tb.gaps, Lp, 1, L, ctx->d_ins_off, ctx->d_ins_total);   // ins_off[p] = gaps[1] + .. + gaps[p-1]
    hipLaunchKernelGGL(k_call_columns, dim3((L + 255) / 256), dim3(256), 0, ctx->stream, ctx->tb.tally, Lp, L, cons_code, ctx->d_calls);
  }
  if (cap <= 0) return MIA_HIP_OK;
  HIPCHK(hipMemsetAsync(ctx->d_ins_tally, 0, (size_t)cap * 9 * 4, ctx->stream));
  const int ne = ctx->n_events_host;
  if (events_on_device)
    hipLaunchKernelGGL(k_ins_tally, dim3(256), dim3(256), 0, ctx->stream, ctx->tb.events, 0, ctx->d_pssm, ctx->d_ins_off, ctx->tb.gaps, L, ctx->d_ins_tally,
                       (int32_t)cap, (const int32_t*)ctx->tb.n_events, ctx->tb.cap_events);
  else if (ne > 0)
    hipLaunchKernelGGL(k_ins_tally, dim3((ne + 255) / 256), dim3(256), 0, ctx->stream, ctx->tb.events, ne, ctx->d_pssm, ctx->d_ins_off,
                       ctx->tb.gaps, L, ctx->d_ins_tally, (int32_t)cap, (const int32_t*)nullptr, 0);
  hipLaunchKernelGGL(k_call_inserts, dim3((L + 255) / 256), dim3(256), 0, ctx->stream, ctx->tb.tally, Lp, L, ctx->tb.gaps,
                     ctx->d_ins_off, ctx->d_ins_tally, cons_code, ctx->d_ins_calls, (int32_t)cap);
  return MIA_HIP_OK;
}

extern "C" int mia_hip_consensus(mia_hip_ctx* ctx, int cons_code, char* out, int64_t out_cap, int64_t* out_len) {
  if (!ctx || !out) return MIA_HIP_ERR_ARG;
  if (!ctx->tallied) { ctx->err = "tally first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  const int L = ctx->L, Lp = ctx->tb.Lp;
  // The insert columns are tallied and called into buffers of the capacity the last call left behind, and their total
  // comes back with the results: one wait for the stream instead of two.  Only when the total has outgrown the buffers
  // are they enlarged and the insert part run again.
  auto insert_part = [&](int64_t cap) -> int { return consensus_launch(ctx, cons_code, cap, false, false); };
  if (int rcc = consensus_launch(ctx, cons_code, 0, true, false)) return rcc;
  int rc0 = insert_part(ctx->ins_tally_cap);
  if (rc0) return rc0;
  HIPCHK(hipGetLastError());
  // results through the pinned staging area when they fit (copies into pageable memory make the host wait one by one)
  const size_t need = 16 + (size_t)L + 2 * (size_t)Lp * 4 + (size_t)ctx->ins_tally_cap + 64;
  std::vector<unsigned char> pageable;
  unsigned char* base;
  if (ctx->h_pin && need <= mia_hip_ctx::PIN_BYTES - mia_hip_ctx::PIN_MISC) base = ctx->h_pin + mia_hip_ctx::PIN_MISC;
  else { pageable.resize(need); base = pageable.data(); }
  int32_t* h_total = reinterpret_cast<int32_t*>(base);
  int32_t* gaps = reinterpret_cast<int32_t*>(base + 16);
  int32_t* off = gaps + Lp;
  char* calls = reinterpret_cast<char*>(off + Lp);
  char* ins = calls + L;
  HIPCHK(hipMemcpyAsync(h_total, ctx->d_ins_total, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(calls, ctx->d_calls, (size_t)L, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(gaps, ctx->tb.gaps, (size_t)Lp * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(off, ctx->d_ins_off, (size_t)Lp * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (ctx->ins_tally_cap > 0) HIPCHK(hipMemcpyAsync(ins, ctx->d_ins_calls, (size_t)ctx->ins_tally_cap, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const int32_t total = *h_total;
  std::vector<char> ins_big;
  if (total > ctx->ins_tally_cap) {
    const int64_t cap = (int64_t)total + total / 4 + 1024;
    if (dev_alloc(ctx, &ctx->d_ins_tally, (size_t)cap * 9) || dev_alloc(ctx, &ctx->d_ins_calls, (size_t)cap)) return MIA_HIP_ERR_NOMEM;
    ctx->ins_tally_cap = cap;
    rc0 = insert_part(cap);
    if (rc0) return rc0;
    HIPCHK(hipGetLastError());
    ins_big.resize((size_t)total + 1);
    HIPCHK(hipMemcpyAsync(ins_big.data(), ctx->d_ins_calls, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ins = ins_big.data();
  }
  // string assembly of consensus_assembly_string (src/mia.c:551-600): insert calls, then the column call; '-' is skipped
  int64_t o = 0;
  for (int p = 0; p < L; p++) {
    if (p > 0)
      for (int j = 0; j < gaps[p]; j++) {
        char c = ins[off[p] + j];
        if (c != '-' && c != ' ') { if (o + 1 >= out_cap) { ctx->err = "consensus buffer too small"; return MIA_HIP_ERR_ARG; } out[o++] = c; }
      }
    char c = calls[p];
    if (c != '-' && c != ' ') { if (o + 1 >= out_cap) { ctx->err = "consensus buffer too small"; return MIA_HIP_ERR_ARG; } out[o++] = c; }
  }
  out[o] = 0;
  if (out_len) *out_len = o;
  ctx->consensus_done = true;
  ctx->ins_total_host = total;
  return MIA_HIP_OK;
}

// ---- several GPUs (reads shard, tallies all-reduce) ------------------------------------------
// The exchanges of a sharded iteration go through ctx->coll, a table of two collectives (include/mia_hip.h); the
// transports themselves -- RCCL over xGMI, the in-process loopback -- live in mia_comm.hip.
constexpr int PRE_WORDS = 8;         // per rank before the cull: five score sums, AlnSeq records, links, reads waiting for the exact kernel
#define COLLCHK(call)                                                                                  \
  do {                                                                                                 \
    const int r_ = (call);                                                                             \
    if (r_ != MIA_HIP_OK) {                                                                            \
      const char* t_ = ctx->coll.error ? ctx->coll.error(ctx->coll.user) : nullptr;                    \
      ctx->err = std::string(#call) + ": " + (t_ && *t_ ? t_ : "collective failed");                   \
      return r_;                                                                                       \
    }                                                                                                  \
  } while (0)

static int comm_attach_table(mia_hip_ctx* ctx, const mia_hip_collectives& t) {
  if (dev_alloc(ctx, &ctx->d_gather, (size_t)PRE_WORDS * t.n_ranks)) return MIA_HIP_ERR_NOMEM;
  ctx->coll = t;
  ctx->comm = true; ctx->comm_ranks = t.n_ranks; ctx->comm_rank = t.rank;
  ctx->h_gather.assign((size_t)PRE_WORDS * t.n_ranks, 0);
  ctx->ev_pad = 0;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_comm_init(mia_hip_ctx* ctx, const void* id128, int32_t n_ranks, int32_t rank) {
  if (!ctx || !id128 || n_ranks < 1 || n_ranks > 256 || rank < 0 || rank >= n_ranks) return MIA_HIP_ERR_ARG;
  if (ctx->comm) { ctx->err = "this context already has a communicator"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  mia_hip_collectives t;
  if (int rc = mia_comm_rccl_table(id128, n_ranks, rank, &t, &ctx->err)) return rc;
  if (int rc = comm_attach_table(ctx, t)) { if (t.destroy) t.destroy(t.user); return rc; }     // (no half-made communicator stays behind)
  return MIA_HIP_OK;
}

extern "C" int mia_hip_comm_attach(mia_hip_ctx* ctx, const mia_hip_collectives* table) {
  if (!ctx || !table || !table->all_gather || !table->all_reduce_i32 || table->n_ranks < 1 || table->n_ranks > 256 || table->rank < 0 ||
      table->rank >= table->n_ranks)
    return MIA_HIP_ERR_ARG;
  if (ctx->comm) { ctx->err = "this context already has a communicator"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  return comm_attach_table(ctx, *table);
}

extern "C" int mia_hip_comm_info(mia_hip_ctx* ctx, int32_t* n_ranks, int32_t* rank, const char** transport) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  int32_t nr = 1, rk = 0;
  if (ctx->comm) {
    nr = ctx->comm_ranks; rk = ctx->comm_rank;
    if (ctx->coll.query) { if (int rc = ctx->coll.query(ctx->coll.user, &nr, &rk)) { ctx->err = "the transport could not say how many ranks it has"; return rc; } }
  }
  if (n_ranks) *n_ranks = nr;
  if (rank) *rank = rk;
  if (transport) *transport = ctx->comm ? (ctx->coll.name ? ctx->coll.name : "caller") : "none";
  return MIA_HIP_OK;
}

extern "C" int mia_hip_comm_destroy(mia_hip_ctx* ctx) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  if (!ctx->comm) return MIA_HIP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->coll.destroy) ctx->coll.destroy(ctx->coll.user);
  ctx->coll = mia_hip_collectives{};
  ctx->comm = false; ctx->comm_ranks = 1; ctx->comm_rank = 0;
  return MIA_HIP_OK;
}

// all-gather of ragged 8-byte records (links: 4 words each; insert events: 1): every rank contributes counts[rank] * words
// words from `mine`; the concatenation in rank order lands in ctx->d_lall.  counts are known on every rank.
static int comm_gather_ragged(mia_hip_ctx* ctx, const int64_t* mine, const std::vector<int64_t>& counts, int words, int64_t* total_out) {
  const int W = ctx->comm_ranks;
  int64_t mx = 0, total = 0;
  for (int r = 0; r < W; r++) { mx = std::max(mx, counts[(size_t)r]); total += counts[(size_t)r]; }
  *total_out = total;
  if (total == 0) return MIA_HIP_OK;
  const int64_t pad = mx * words;
  if (pad > ctx->lmine_cap) { if (dev_alloc(ctx, &ctx->d_lmine, (size_t)pad * 2)) return MIA_HIP_ERR_NOMEM; ctx->lmine_cap = pad * 2; }
  if (pad * W > ctx->lstage_cap) { if (dev_alloc(ctx, &ctx->d_lstage, (size_t)pad * W * 2)) return MIA_HIP_ERR_NOMEM; ctx->lstage_cap = pad * W * 2; }
  if (total * words > ctx->lall_cap) { if (dev_alloc(ctx, &ctx->d_lall, (size_t)total * words * 2)) return MIA_HIP_ERR_NOMEM; ctx->lall_cap = total * words * 2; }
  HIPCHK(hipMemsetAsync(ctx->d_lmine, 0, (size_t)pad * 8, ctx->stream));
  const int64_t nm = counts[(size_t)ctx->comm_rank] * words;
  if (nm > 0) HIPCHK(hipMemcpyAsync(ctx->d_lmine, mine, (size_t)nm * 8, hipMemcpyDeviceToDevice, ctx->stream));
  COLLCHK(ctx->coll.all_gather(ctx->coll.user, ctx->d_lmine, ctx->d_lstage, (size_t)pad * 8, ctx->stream));
  int64_t o = 0;
  for (int r = 0; r < W; r++) {
    const int64_t c = counts[(size_t)r] * words;
    if (c > 0) HIPCHK(hipMemcpyAsync(ctx->d_lall + o, ctx->d_lstage + (int64_t)r * pad, (size_t)c * 8, hipMemcpyDeviceToDevice, ctx->stream));
    o += c;
  }
  return MIA_HIP_OK;
}

// What the cull needs from the other ranks -- their score sums (find_fsdb_score_cut's first pass), how many AlnSeq records
// precede this rank's, how many links each cull will emit, and whether some rank still has reads for the exact kernel --
// in ONE small all-gather, queued BEHIND the alignment kernels and in front of the alignment's one host wait: a sharded
// iteration waits for the host as often as a single context does.
static int comm_pre_cull_enqueue(mia_hip_ctx* ctx, const int32_t* d_wide_count) {
  const int64_t n = ctx->rs.n;
  const int W = ctx->comm_ranks;
  if (!ctx->d_sums && dev_alloc(ctx, &ctx->d_sums, 8)) return MIA_HIP_ERR_NOMEM;
  hipLaunchKernelGGL(k_score_sums_init, dim3(1), dim3(64), 0, ctx->stream, ctx->d_sums, d_wide_count);
  const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cus);
  hipLaunchKernelGGL(k_score_sums, dim3(grid), dim3(256), 0, ctx->stream, ctx->rs, ctx->d_sums, ctx->L, ctx->d_back_slot, ctx->d_front_slot0);
  HIPCHK(hipGetLastError());
  COLLCHK(ctx->coll.all_gather(ctx->coll.user, ctx->d_sums, ctx->d_gather, (size_t)PRE_WORDS * 8, ctx->stream));
  int64_t* stage = ctx->h_pin ? reinterpret_cast<int64_t*>(ctx->h_pin + (40 << 10)) : ctx->h_gather.data();
  HIPCHK(hipMemcpyAsync(stage, ctx->d_gather, (size_t)PRE_WORDS * W * 8, hipMemcpyDeviceToHost, ctx->stream));
  return MIA_HIP_OK;
}
// ... and once the stream has been waited for
static bool comm_pre_cull_collect(mia_hip_ctx* ctx) {
  const int W = ctx->comm_ranks;
  if (ctx->h_pin) memcpy(ctx->h_gather.data(), ctx->h_pin + (40 << 10), (size_t)PRE_WORDS * W * 8);
  bool any_wide = false;
  for (int r = 0; r < W; r++) any_wide = any_wide || ctx->h_gather[(size_t)PRE_WORDS * r + 7] != 0;
  return any_wide;
}

__global__ void k_put_i32(int32_t* dst, const int32_t* src, int32_t clamp) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = min(*src, clamp); }

// The insert events of all ranks, gathered in blocks of `pad` (stage[r * pad + i], i < counts[r]), packed into one list in
// rank order; *n_out = their number.  counts[] are the W slots behind ref->gaps that rode on its max-reduce.  If some rank
// had more than `pad` events nothing is touched and bit 8 of the tally flags says so (the host repeats the exchange with
// the counts in hand).  One workgroup per rank in y.
__global__ __launch_bounds__(256) void k_events_compact(const uint64_t* __restrict__ stage, int64_t pad, const int32_t* __restrict__ counts, int W,
                                                          uint64_t* __restrict__ out, int32_t cap, int32_t* n_out, uint32_t* flags) {
  const int r = blockIdx.y;
  int64_t before = 0, total = 0;
  bool over = false;
  for (int k = 0; k < W; k++) {
    const int64_t c = counts[k];
    over = over || c > pad;
    if (k < r) before += c;
    total += c;
  }
  over = over || total > cap;
  if (over) { if (r == 0 && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(flags, 8u); return; }
  const int64_t mine = counts[r];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < mine; i += (int64_t)gridDim.x * 256) out[before + i] = stage[(int64_t)r * pad + i];
  if (r == 0 && blockIdx.x == 0 && threadIdx.x == 0) *n_out = (int32_t)total;
}

// ---- one whole iteration -------------------------------------------------------------------------
// reiterate_assembly + pop_smp_from_FSDB + cull_maln_from_fsdb + consensus_assembly_string (src/mia_main.c:931-963) as
// one call: the same kernels as mia_hip_realign / _cull / _tally / _consensus, but what those entry points hand back to
// the host between the kernels -- the planner's bin sizes, the cut line, the insert-event count, five result arrays --
// stays on the device.  The host waits twice: once behind the alignment (reads that need the exact scalar kernel must be
// known before anything is culled) and once for the consensus string -- with or without a communicator.
static int iterate_body(mia_hip_ctx* ctx, const char* new_ref, int32_t ref_len, int circular, int32_t hard_cut, const double* slope_intercept,
                        int cons_code, char* out, int64_t out_cap, int64_t* out_len);

extern "C" int mia_hip_iterate(mia_hip_ctx* ctx, const char* new_ref, int32_t ref_len, int circular, int32_t hard_cut, const double* slope_intercept,
                               int cons_code, char* out, int64_t out_cap, int64_t* out_len) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  int rc;
  if (!new_ref || ref_len <= 0 || !out || out_cap < 1) { ctx->err = "iterate: bad argument"; rc = MIA_HIP_ERR_ARG; }
  else if (!ctx->have_pssm || !ctx->d_packed) { ctx->err = "set_pssm and upload_reads must precede iterate"; rc = MIA_HIP_ERR_STATE; }
  else rc = iterate_body(ctx, new_ref, ref_len, circular, hard_cut, slope_intercept, cons_code, out, out_cap, out_len);
  ctx->in_iterate = false; ctx->deferred = false;
  // a rank that fails here will not come to the collectives the others are about to enter: tell the transport, so that
  // they return an error as well instead of waiting for ever
  if (rc != MIA_HIP_OK && ctx->comm && ctx->coll.abort) ctx->coll.abort(ctx->coll.user);
  return rc;
}

static int iterate_body(mia_hip_ctx* ctx, const char* new_ref, int32_t ref_len, int circular, int32_t hard_cut, const double* slope_intercept,
                        int cons_code, char* out, int64_t out_cap, int64_t* out_len) {
  HIPCHK(hipSetDevice(ctx->device));
  // -- the new reference: ASCII up, codes and wrap made on the device (make_ref_upper / add_ref_wrap, src/mia.c:642-689)
  const int L = ref_len, wl = circular ? (L < MAX_READ ? L : MAX_READ) : 0, wrap = L + wl, total = wrap + 64;
  int64_t n_other = 0;
  {
    static const struct Other { uint8_t t[256]; Other() { for (int c = 0; c < 256; c++) t[c] = base_code((char)c) > 3; } } other_of;   // (a load and an add per character)
    for (int i = 0; i < L; i++) n_other += other_of.t[(uint8_t)new_ref[i]];
  }
  ctx->ref_mostly_bases = n_other * 50 <= L;
  ctx->kh_entries = 0;
  if (n_other && ctx->use_wild) {            // (kh_wild_entries over the wrapped string, without making the codes here)
    int64_t e = 0;
    int k = 0;
    auto other = [&](int j) { return base_code(new_ref[j < L ? j : j - L]) > 3 ? 1 : 0; };
    for (int pq = 0; pq < wrap; pq++) {
      k += other(pq);
      if (pq >= DF_K) k -= other(pq - DF_K);
      if (pq >= DF_K - 1 && k <= BX_WILD) e += (int64_t)1 << (2 * k);
    }
    ctx->kh_entries = e > ((int64_t)1 << 24) ? 0 : e;
  }
  if (total > ctx->ref_cap) {
    if (dev_alloc(ctx, &ctx->d_ref, (size_t)total * 2)) return MIA_HIP_ERR_NOMEM;
    ctx->ref_cap = total * 2;
  }
  if (L > ctx->ascii_cap) {
    if (dev_alloc(ctx, &ctx->d_ascii, (size_t)L * 2)) return MIA_HIP_ERR_NOMEM;
    ctx->ascii_cap = (int64_t)L * 2;
  }
  if (ctx->h_pin && (size_t)L <= mia_hip_ctx::PIN_BYTES - mia_hip_ctx::PIN_MISC) {
    HIPCHK(hipStreamSynchronize(ctx->stream));             // the staging area may still feed an earlier copy
    memcpy(ctx->h_pin + mia_hip_ctx::PIN_MISC, new_ref, (size_t)L);
    HIPCHK(hipMemcpyAsync(ctx->d_ascii, ctx->h_pin + mia_hip_ctx::PIN_MISC, (size_t)L, hipMemcpyHostToDevice, ctx->stream));
  } else {
    HIPCHK(hipMemcpyAsync(ctx->d_ascii, new_ref, (size_t)L, hipMemcpyHostToDevice, ctx->stream));
  }
  // (the codes are made by align_all's first launch: k_ref_prep with everything else that is derived from the reference, or k_ref_encode)
  ctx->pend_encode = true; ctx->pend_L = L; ctx->pend_wl = wl; ctx->pend_total = total;
  if (ctx->rs.n == 0) { encode_now(ctx); HIPCHK(hipGetLastError()); }
  const bool dbg_steps = getenv("MIA_HIP_ITER_DEBUG") != nullptr;      // diagnostic: wait and report after every stage
  auto checkpoint = [&](const char* what) { if (dbg_steps) { hipError_t e = hipStreamSynchronize(ctx->stream); fprintf(stderr, "[mia_hip_iterate] %s: %s\n", what, hipGetErrorString(e)); fflush(stderr); } };
  checkpoint("reference");
  ctx->L = L; ctx->wrap = wrap; ctx->have_ref = true; ctx->explicit_win = 0;
  const int64_t n = ctx->rs.n;
  const int W = ctx->comm_ranks;
  if (ctx->comm && n == 0) { ctx->err = "iterate: every rank of a sharded run needs reads"; return MIA_HIP_ERR_ARG; }
  if (ctx->comm && W > 256) { ctx->err = "iterate: more than 256 ranks"; return MIA_HIP_ERR_ARG; }
  // -- re-alignment (one wait, at its end; a sharded run's pre-cull all-gather rides in front of that wait)
  ctx->in_iterate = true;
  ctx->deferred = true;
  // no host wait behind the alignment when nothing on the host depends on it: one context, and a cut line that is given,
  // hard, or NaN by construction (reads of one length) -- see align_all / abort_if
  ctx->spec_pending = false; ctx->abort_if = nullptr;
  ctx->spec_ok = !ctx->comm && n > 0 && !ctx->no_spec && (hard_cut > 0 || slope_intercept || ctx->min_len == ctx->max_len);
  const int rca = align_all(ctx);
  ctx->deferred = false;
  ctx->spec_ok = false;
  ctx->pend_encode = false;                 // (never outlives the call)
  if (rca) { ctx->spec_pending = false; ctx->abort_if = nullptr; return rca; }
  checkpoint("realign");
  int64_t slot_base = 0, g_sums[5] = {0, 0, 0, 0, 0};
  std::vector<int64_t> link_counts((size_t)W, 0), n_of((size_t)W, n);
  if (ctx->comm) {
    g_sums[3] = INT32_MAX; g_sums[4] = INT32_MIN;
    for (int r = 0; r < W; r++) {
      const int64_t* q = ctx->h_gather.data() + (size_t)PRE_WORDS * r;
      g_sums[0] += q[0]; g_sums[1] += q[1]; g_sums[2] += q[2];
      g_sums[3] = std::min(g_sums[3], q[3]); g_sums[4] = std::max(g_sums[4], q[4]);
      if (r < ctx->comm_rank) slot_base += q[5];
      link_counts[(size_t)r] = q[6];
    }
  }
  // -- the cut line of find_fsdb_score_cut (src/fsdb.c:269-383)
  double slope = 0, intercept = 0;
  const bool one_length = ctx->comm ? (g_sums[2] == 0 || g_sums[3] == g_sums[4]) : (ctx->min_len == ctx->max_len || n == 0);
  if (hard_cut > 0) {
  } else if (slope_intercept) {
    slope = slope_intercept[0]; intercept = slope_intercept[1];
  } else if (one_length) {
    // reads of one length: both regression sums are exactly 0, slope_bf = 0/0, and every derived quantity is that NaN
    // whatever the scores (mia_hip_score_cut_from_sums spells the arithmetic out) -- nothing to compute
    const double zero = 0.0;
    slope = intercept = zero / zero;
  } else if (!ctx->comm) {
    // sums of products in IEEE double are order dependent: the reference's sequential order over the scores, on the host
    std::vector<int32_t> score((size_t)n);
    HIPCHK(hipMemcpyAsync(score.data(), ctx->d_score, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    mia_hip_score_cut(score.data(), ctx->h_len.data(), nullptr, n, &slope, &intercept);
  } else {
    // the same over the reads of ALL ranks in fsdb order (= rank order): scores and lengths gathered, the regression on
    // every rank's host (identical inputs, identical arithmetic)
    int32_t* d_nl = nullptr;                               // {score..., len...} of this rank, padded to the longest
    ScopeFree sf; sf.watch((void**)&d_nl);
    std::vector<int64_t> cnt1((size_t)W, 1);
    int64_t tot = 0;
    int64_t n64 = n;
    int64_t* d_n = nullptr; sf.watch((void**)&d_n);
    if (hipMalloc((void**)&d_n, 8) != hipSuccess) return MIA_HIP_ERR_NOMEM;
    HIPCHK(hipMemcpyAsync(d_n, &n64, 8, hipMemcpyHostToDevice, ctx->stream));
    if (int rcg = comm_gather_ragged(ctx, d_n, cnt1, 1, &tot)) return rcg;
    HIPCHK(hipMemcpyAsync(n_of.data(), ctx->d_lall, (size_t)W * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int64_t nmax = 0, ntot = 0;
    for (int r = 0; r < W; r++) { nmax = std::max(nmax, n_of[(size_t)r]); ntot += n_of[(size_t)r]; }
    if (hipMalloc((void**)&d_nl, (size_t)nmax * 2 * 4) != hipSuccess) return MIA_HIP_ERR_NOMEM;
    if ((int64_t)W * nmax * 2 > ctx->scores_all_cap) {
      if (dev_alloc(ctx, &ctx->d_scores_all, (size_t)W * nmax * 2)) return MIA_HIP_ERR_NOMEM;
      ctx->scores_all_cap = (int64_t)W * nmax * 2;
    }
    HIPCHK(hipMemsetAsync(d_nl, 0, (size_t)nmax * 2 * 4, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_nl, ctx->d_score, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_nl + nmax, ctx->h_len.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    COLLCHK(ctx->coll.all_gather(ctx->coll.user, d_nl, ctx->d_scores_all, (size_t)nmax * 2 * 4, ctx->stream));
    std::vector<int32_t> all((size_t)W * nmax * 2), score((size_t)ntot), lens((size_t)ntot);
    HIPCHK(hipMemcpyAsync(all.data(), ctx->d_scores_all, all.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int64_t o = 0;
    for (int r = 0; r < W; r++) {
      memcpy(score.data() + o, all.data() + (size_t)r * nmax * 2, (size_t)n_of[(size_t)r] * 4);
      memcpy(lens.data() + o, all.data() + (size_t)r * nmax * 2 + nmax, (size_t)n_of[(size_t)r] * 4);
      o += n_of[(size_t)r];
    }
    mia_hip_score_cut(score.data(), lens.data(), nullptr, ntot, &slope, &intercept);
  }
  if (!(hard_cut > 0) && slope <= 0) slope = 100.0;         // src/mia.c:440-442
  auto queue_cull = [&]() -> int {
    ctx->buckets_queued = 0;
    const bool side = tally_is_binned(ctx) && !ctx->no_side_buckets;
    // (nothing has been waited for: stream2 must stay behind the alignment -- but not behind the cull)
    if (side && ctx->spec_pending) HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
    if (int rcc = mia_hip_cull(ctx, hard_cut, slope, intercept, slot_base)) return rcc;
    // the tally's counting sort reads nothing the cull writes: on stream2, beside the cull kernels (queued behind them on the
    // host side -- the GPU is waiting for the first cull kernel at this point, not for these)
    if (side) {
      if (ctx->spec_pending) HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
      if (int rcb = bucket_launch(ctx, ctx->stream2)) return rcb;
    }
    return MIA_HIP_OK;
  };
  if (int rcq = queue_cull()) return rcq;
  if (ctx->comm) {
    // links of formerly split reads (stale fs->back_asp, include/mia_hip.h) may point at slots of another rank: every rank
    // gets all links, applies those that hit its own slots, the record lengths the readers need come back by a max-reduce
    int64_t any = 0;
    for (int r = 0; r < W; r++) any += link_counts[(size_t)r];
    if (any > 0) {
      int64_t* dl = nullptr; int64_t nl = 0, total_l = 0;
      if (int rcl = mia_hip_links(ctx, &dl, &nl)) return rcl;
      if (nl != link_counts[(size_t)ctx->comm_rank]) { ctx->err = "iterate: the cull emitted another number of links than the score sweep announced"; return MIA_HIP_ERR_STATE; }
      if (int rcg = comm_gather_ragged(ctx, dl, link_counts, 4, &total_l)) return rcg;
      if (int rcs = mia_hip_set_links(ctx, ctx->d_lall, total_l)) return rcs;
      int32_t *dlen = nullptr, *dact = nullptr; int64_t nn = 0;
      if (int rcl = mia_hip_link_lengths(ctx, &dlen, &dact, &nn)) return rcl;
      COLLCHK(ctx->coll.all_reduce_i32(ctx->coll.user, dlen, (size_t)nn, MIA_HIP_OP_MAX, ctx->stream));
      COLLCHK(ctx->coll.all_reduce_i32(ctx->coll.user, dact, (size_t)nn, MIA_HIP_OP_MAX, ctx->stream));
      if (int rcf = mia_hip_finish_links(ctx)) return rcf;
    }
  }
  checkpoint("cull");
  if (n == 0) { out[0] = 0; if (out_len) *out_len = 0; return MIA_HIP_OK; }
  // -- tally and consensus, queued back to back
  if (ctx->comm && ctx->ev_pad > 0) {
    if (const char* e = getenv("MIA_HIP_EV_PAD")) { const long long v = atoll(e); if (v > 0) ctx->ev_pad = v; }   // (tests: force the overflow path; every rank reads the same value)
    // the gathered events land in this context's list: room for every rank's block before anything is tallied into it
    if (int rct = ensure_tally(ctx)) return rct;
    if (ctx->ev_pad * W > ctx->tb.cap_events) {
      const int64_t cap = std::min<int64_t>(ctx->ev_pad * W + 4096, (int64_t)1 << 30);
      if (dev_alloc(ctx, &ctx->tb.events, (size_t)cap)) return MIA_HIP_ERR_NOMEM;
      ctx->tb.cap_events = (int32_t)cap;
    }
  }
  if (int rct = tally_launch(ctx)) return rct;
  checkpoint("tally");
  const int Lp = ctx->tb.Lp;
  ctx->n_events_host = 0;
  int32_t* h_evc = ctx->h_pin ? reinterpret_cast<int32_t*>(ctx->h_pin + (56 << 10)) : nullptr;      // the ranks' event counts, read with the results
  std::vector<int32_t> evc_pageable;
  if (ctx->comm && !h_evc) { evc_pageable.resize((size_t)W); h_evc = evc_pageable.data(); }
  // the events of all ranks with the counts in hand (one more wait for the host): the first sharded iteration, and whenever
  // a rank had more events than the blocks of the other way were sized for
  auto exchange_events_counted = [&]() -> int {
    HIPCHK(hipMemcpyAsync(h_evc, ctx->tb.gaps + Lp, (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> ev_counts((size_t)W);
    for (int r = 0; r < W; r++) ev_counts[(size_t)r] = h_evc[r];
    int64_t total_ev = 0;
    if (int rcg = comm_gather_ragged(ctx, reinterpret_cast<const int64_t*>(ctx->tb.events), ev_counts, 1, &total_ev)) return rcg;
    if (total_ev > ctx->tb.cap_events) {
      if (dev_alloc(ctx, &ctx->tb.events, (size_t)total_ev + 4096)) return MIA_HIP_ERR_NOMEM;
      ctx->tb.cap_events = (int32_t)std::min<int64_t>(total_ev + 4096, INT32_MAX);
    }
    if (total_ev > 0) HIPCHK(hipMemcpyAsync(ctx->tb.events, ctx->d_lall, (size_t)total_ev * 8, hipMemcpyDeviceToDevice, ctx->stream));
    int32_t* te = ctx->h_pin ? reinterpret_cast<int32_t*>(ctx->h_pin + (60 << 10)) : nullptr;
    int32_t te_local = (int32_t)total_ev;
    if (te) { *te = te_local; HIPCHK(hipMemcpyAsync(ctx->tb.n_events, te, 4, hipMemcpyHostToDevice, ctx->stream)); }
    else { HIPCHK(hipMemcpyAsync(ctx->tb.n_events, &te_local, 4, hipMemcpyHostToDevice, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream)); }
    return MIA_HIP_OK;
  };
  // ... and without: every rank contributes a block of ev_pad events (sized from the counts of the iteration before, the
  // same on every rank), the counts that rode on the gaps reduce tell k_events_compact how much of each block is real
  auto exchange_events_padded = [&]() -> int {
    const int64_t pad = ctx->ev_pad;
    if (pad * W > ctx->lstage_cap) { if (dev_alloc(ctx, &ctx->d_lstage, (size_t)pad * W * 2)) return MIA_HIP_ERR_NOMEM; ctx->lstage_cap = pad * W * 2; }
    COLLCHK(ctx->coll.all_gather(ctx->coll.user, ctx->tb.events, ctx->d_lstage, (size_t)pad * 8, ctx->stream));
    const unsigned gx = (unsigned)std::min<int64_t>((pad + 255) / 256, 64);
    hipLaunchKernelGGL(k_events_compact, dim3(gx, (unsigned)W), dim3(256), 0, ctx->stream, (const uint64_t*)ctx->d_lstage, pad, (const int32_t*)(ctx->tb.gaps + Lp), W,
                       ctx->tb.events, ctx->tb.cap_events, ctx->tb.n_events, ctx->tb.flags);
    HIPCHK(hipGetLastError());
    return MIA_HIP_OK;
  };
  bool counted = false;
  if (ctx->comm) {
    // integer column tallies add up, ref->gaps is a maximum (src/mia.c:486-504); every rank's insert-event count rides on
    // the max-reduce in W extra slots behind gaps; then the events themselves are gathered
    hipLaunchKernelGGL(k_put_i32, dim3(1), dim3(1), 0, ctx->stream, ctx->tb.gaps + Lp + ctx->comm_rank, (const int32_t*)ctx->tb.n_events, ctx->tb.cap_events);
    COLLCHK(ctx->coll.all_reduce_i32(ctx->coll.user, ctx->tb.tally, (size_t)TALLY_WORDS * Lp, MIA_HIP_OP_SUM, ctx->stream));
    COLLCHK(ctx->coll.all_reduce_i32(ctx->coll.user, ctx->tb.gaps, (size_t)Lp + W, MIA_HIP_OP_MAX, ctx->stream));
    counted = ctx->ev_pad <= 0;
    if (int rce = counted ? exchange_events_counted() : exchange_events_padded()) return rce;
  }
  const int64_t cons_cap = (int64_t)L + ctx->ins_tally_cap + 64;          // string bytes
  const int64_t res_bytes = (int64_t)CH_WORDS * 4 + cons_cap;
  if (res_bytes > ctx->cons_cap) {
    if (dev_alloc(ctx, &ctx->d_cons, (size_t)res_bytes * 2)) return MIA_HIP_ERR_NOMEM;
    ctx->cons_cap = res_bytes * 2;
  }
  if (Lp > ctx->cons_pos_cap) {
    if (dev_alloc(ctx, &ctx->d_cons_pos, (size_t)Lp * 2)) return MIA_HIP_ERR_NOMEM;
    ctx->cons_pos_cap = (int64_t)Lp * 2;
  }
  const size_t need = (size_t)res_bytes;
  if (need > ctx->pin2_bytes) {
    if (ctx->h_pin2) (void)hipHostFree(ctx->h_pin2);
    ctx->h_pin2 = nullptr; ctx->pin2_bytes = 0;
    if (hipHostMalloc((void**)&ctx->h_pin2, need * 2, hipHostMallocDefault) != hipSuccess) { ctx->err = "hipHostMalloc"; return MIA_HIP_ERR_NOMEM; }
    ctx->pin2_bytes = need * 2;
  }
  int32_t* h_hdr = reinterpret_cast<int32_t*>(ctx->h_pin2);
  char* h_str = reinterpret_cast<char*>(ctx->h_pin2) + CH_WORDS * 4;
  // consensus calls, the string consensus_assembly_string returns put together on the device (characters per column, their
  // prefix sums, a scatter), header and string back in one copy, the second and last wait
  auto consensus_tail = [&]() -> int {
    // (a workgroup alone on sixteen thousand columns is slower than a launch costs -- 44 + 111 us for two single-workgroup
    // kernels against 5 us each for these: the launches that remain are the ones with a chip-wide dependence between them)
    const int64_t cap = ctx->ins_tally_cap;
    int32_t* d_res = reinterpret_cast<int32_t*>(ctx->d_cons);
    const unsigned gl = (unsigned)((L + 255) / 256);
    hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, ctx->stream, (const int32_t*)ctx->tb.gaps, Lp, 1, L, ctx->d_ins_off, ctx->d_ins_total, ctx->abort_if);   // ins_off[p] = gaps[1] + .. + gaps[p-1]
    hipLaunchKernelGGL(k_call_columns_z, dim3(gl), dim3(256), 0, ctx->stream, (const int32_t*)ctx->tb.tally, Lp, L, cons_code, ctx->d_calls, ctx->d_ins_tally, cap * 9, ctx->abort_if);
    if (cap > 0)
      hipLaunchKernelGGL(k_ins_tally, dim3(256), dim3(256), 0, ctx->stream, ctx->tb.events, 0, ctx->d_pssm, ctx->d_ins_off, ctx->tb.gaps, L, ctx->d_ins_tally,
                         (int32_t)cap, (const int32_t*)ctx->tb.n_events, ctx->tb.cap_events, ctx->abort_if);
    checkpoint("consensus kernels");
    hipLaunchKernelGGL(k_call_inserts_count, dim3(gl), dim3(256), 0, ctx->stream, (const int32_t*)ctx->tb.tally, Lp, L, (const int32_t*)ctx->tb.gaps,
                       (const int32_t*)ctx->d_ins_off, (const int32_t*)ctx->d_ins_tally, cons_code, (const char*)ctx->d_calls, ctx->d_ins_calls, (int32_t)cap,
                       (const int32_t*)ctx->d_ins_total, ctx->d_cons_pos, ctx->abort_if);
    hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, ctx->stream, (const int32_t*)ctx->d_cons_pos, L, 0, L, ctx->d_cons_pos, d_res + CH_LEN, ctx->abort_if);
    hipLaunchKernelGGL(k_cons_scatter, dim3(gl), dim3(256), 0, ctx->stream, (const char*)ctx->d_calls, (const char*)ctx->d_ins_calls,
                       (const int32_t*)ctx->tb.gaps, (const int32_t*)ctx->d_ins_off, L, (int32_t)cap, (const int32_t*)ctx->d_ins_total,
                       (const int32_t*)ctx->d_cons_pos, d_res, (int32_t)cons_cap, (const int32_t*)ctx->tb.n_events, (const uint32_t*)ctx->tb.flags,
                       (const uint32_t*)ctx->d_cull_flags, ctx->abort_if);
    HIPCHK(hipGetLastError());
    checkpoint("assemble");
    HIPCHK(hipMemcpyAsync(ctx->h_pin2, ctx->d_cons, need, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->comm && !counted) HIPCHK(hipMemcpyAsync(h_evc, ctx->tb.gaps + Lp, (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MIA_HIP_OK;
  };
  if (int rct = consensus_tail()) return rct;
  if (ctx->spec_pending) {
    // the alignment's counters arrived with the consensus.  Reads waiting for the exact kernel (next to never): every kernel
    // queued behind the alignment has returned without touching anything; the exact kernel runs, and cull, tally and consensus
    // are queued again -- this time unconditionally
    ctx->spec_pending = false;
    ctx->abort_if = nullptr;
    const int32_t* hb = reinterpret_cast<const int32_t*>(ctx->h_pin);
    align_counters_collect(ctx, hb, ctx->spec_filtered, ctx->spec_bx, ctx->spec_plain);
    const int32_t n_wide = hb[0];
    if (n_wide > 0 || ctx->spec_force) {
      ctx->spec_redone++;
      RefInfo rinfo{ctx->d_ref, ctx->L, ctx->wrap, ctx->explicit_win};
      if (n_wide > 0) { if (int rcw = run_wide(ctx, rinfo, n_wide)) return rcw; }
      if (int rcq = queue_cull()) return rcq;
      if (int rct = tally_launch(ctx)) return rct;
      if (int rct = consensus_tail()) return rct;
    }
  }
  uint32_t tflags = (uint32_t)h_hdr[CH_TALLY_FLAGS];
  if (ctx->comm && !counted && (tflags & 8u)) {
    // some rank had more insert events than the blocks held (every rank sees the same counts, so every rank is here):
    // the lists and counts are untouched -- exchange them with the counts in hand and call the consensus again
    counted = true;
    if (int rce = exchange_events_counted()) return rce;
    if (int rct = consensus_tail()) return rct;
    tflags = (uint32_t)h_hdr[CH_TALLY_FLAGS];
  }
  tflags &= ~8u;
  if (ctx->comm) {
    // the block size of the next iteration's event exchange, from counts every rank has seen: half as much again as the
    // largest, re-sized only when that leaves the band [pad/4, 0.8 pad] (identical arithmetic on identical numbers)
    int64_t mx = 0;
    for (int r = 0; r < W; r++) mx = std::max<int64_t>(mx, h_evc[r]);
    if (ctx->ev_pad <= 0 || mx * 5 > ctx->ev_pad * 4 || mx * 4 < ctx->ev_pad) ctx->ev_pad = mx + mx / 2 + 1024;
  }
  ctx->in_iterate = false;
  if (int rcf = tally_finish(ctx, (uint32_t)h_hdr[CH_N_EVENTS], tflags, (uint32_t)h_hdr[CH_CULL_FLAGS])) return rcf;
  if (h_hdr[CH_OVERFLOW]) {
    // more insert columns than the buffers of the last call hold: the step-wise entry point enlarges them and calls again
    ctx->iter_fallbacks++;
    return mia_hip_consensus(ctx, cons_code, out, out_cap, out_len);
  }
  const int64_t len = h_hdr[CH_LEN];
  if (len + 1 > out_cap) { ctx->err = "consensus buffer too small"; return MIA_HIP_ERR_ARG; }
  memcpy(out, h_str, (size_t)len);
  out[len] = 0;
  if (out_len) *out_len = len;
  ctx->consensus_done = true;
  ctx->ins_total_host = h_hdr[CH_INS_TOTAL];
  return MIA_HIP_OK;
}

// ---- adapter trimming ---------------------------------------------------------------------------
extern "C" int mia_hip_trim(mia_hip_ctx* ctx, const char* adapter, int64_t n, const char* bases, const int64_t* offsets,
                            uint8_t* trimmed, int32_t* trim_point) {
  if (!ctx || !adapter || n < 0 || (n > 0 && (!bases || !offsets || !trimmed || !trim_point))) return MIA_HIP_ERR_ARG;
  const int len2 = (int)strlen(adapter);
  if (len2 < 1 || len2 > MAX_ADAPTER) { ctx->err = "adapter length outside 1..127"; return MIA_HIP_ERR_ARG; }
  HIPCHK(hipSetDevice(ctx->device));
  if (n == 0) return MIA_HIP_OK;
  for (int64_t i = 0; i < n; i++) {
    const int64_t l = offsets[i + 1] - offsets[i];
    if (l < 1 || l > MIA_HIP_MAX_READ) { ctx->err = "read length outside 1..256"; return MIA_HIP_ERR_ARG; }
  }
  // init_flatsubmat (src/pssm.c:96-126)
  std::vector<int32_t> flat((size_t)PSSM_WORDS);
  for (int d = 0; d < 31; d++)
    for (int i = 0; i < 5; i++)
      for (int j = 0; j < 5; j++)
        flat[(size_t)(d * 5 + i) * 5 + j] = i == 4 ? -10 : (j == 4 ? -100 : (i == j ? 200 : -600));
  PackParams pk;
  if (!make_pack_params(256, 600, &pk)) return MIA_HIP_ERR_RANGE;
  const int64_t chars = offsets[n] - offsets[0];
  std::vector<uint8_t> codes((size_t)chars), acodes((size_t)len2), apacked((size_t)(len2 + 1) / 2 + 4, 0);
  for (int64_t k = 0; k < chars; k++) codes[(size_t)k] = base_code(bases[offsets[0] + k]);
  for (int k = 0; k < len2; k++) { acodes[(size_t)k] = base_code(adapter[k]); apacked[(size_t)k >> 1] |= (uint8_t)(acodes[(size_t)k] << ((k & 1) * 4)); }
  std::vector<int64_t> off0((size_t)n + 1);
  for (int64_t i = 0; i <= n; i++) off0[(size_t)i] = offsets[i] - offsets[0];
  int occ = 0;
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_trim, 64, 0));
  if (occ < 1) occ = 1;
  if (occ > 32) occ = 32;
  if (occ > 4) occ &= ~3;
  int64_t grid = (int64_t)ctx->cus * occ;
  if (grid > n) grid = n;
  const int64_t slab = (int64_t)(MAX_ADAPTER + 1) * MAX_READ;
  uint8_t *d_codes = nullptr, *d_ap = nullptr, *d_ac = nullptr, *d_trimmed = nullptr;
  int64_t* d_off = nullptr;
  int32_t *d_flat = nullptr, *d_tp = nullptr, *d_list = nullptr, *d_scratch = nullptr;
  uint32_t* d_status = nullptr;
  unsigned char* d_slabs = nullptr;
  int16_t* d_cols = nullptr;
  ScopeFree guard;   // every temporary is released on any return
  for (void** pp : {(void**)&d_codes, (void**)&d_ap, (void**)&d_ac, (void**)&d_trimmed, (void**)&d_off, (void**)&d_flat, (void**)&d_tp,
                    (void**)&d_list, (void**)&d_scratch, (void**)&d_status, (void**)&d_slabs, (void**)&d_cols})
    guard.watch(pp);
  int rcx = dev_alloc(ctx, &d_codes, (size_t)chars + 8) | dev_alloc(ctx, &d_ap, apacked.size()) | dev_alloc(ctx, &d_ac, (size_t)len2) |
            dev_alloc(ctx, &d_trimmed, (size_t)n) | dev_alloc(ctx, &d_off, (size_t)n + 1) | dev_alloc(ctx, &d_flat, (size_t)PSSM_WORDS) |
            dev_alloc(ctx, &d_tp, (size_t)n) | dev_alloc(ctx, &d_status, (size_t)n) | dev_alloc(ctx, &d_slabs, (size_t)(slab * grid)) |
            dev_alloc(ctx, &d_cols, (size_t)(grid * MAX_READ));
  if (rcx) return MIA_HIP_ERR_NOMEM;
  hipError_t e = hipSuccess;
  auto up = [&](void* d, const void* h, size_t b) { if (e == hipSuccess && b) e = hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, ctx->stream); };
  up(d_codes, codes.data(), (size_t)chars); up(d_ap, apacked.data(), apacked.size()); up(d_ac, acodes.data(), (size_t)len2);
  up(d_off, off0.data(), (size_t)(n + 1) * 8); up(d_flat, flat.data(), (size_t)PSSM_WORDS * 4);
  TrimReads tr{n, d_codes, d_off, d_trimmed, d_tp, d_status};
  std::vector<uint32_t> status((size_t)n);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_trim, dim3((unsigned)grid), dim3(64), 0, ctx->stream, tr, d_ap, len2, d_flat, pk, d_slabs, slab, d_cols);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(status.data(), d_status, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  // gaps of 63 or more on the optimal path: exact scalar re-run of those reads
  std::vector<int32_t> esc;
  if (e == hipSuccess)
    for (int64_t i = 0; i < n; i++) if (status[(size_t)i] != ST_OK) esc.push_back((int32_t)i);
  if (e == hipSuccess && !esc.empty()) {
    const int64_t words = (int64_t)len2 * MAX_READ + 5 * (int64_t)MAX_READ;
    if (dev_alloc(ctx, &d_list, esc.size()) || dev_alloc(ctx, &d_scratch, (size_t)(words * (int64_t)esc.size()))) return MIA_HIP_ERR_NOMEM;
    up(d_list, esc.data(), esc.size() * 4);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_trim_wide, dim3((unsigned)((esc.size() + 63) / 64)), dim3(64), 0, ctx->stream, tr, d_ac, len2, d_flat, d_list,
                         (int32_t)esc.size(), d_scratch, words);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess) e = hipMemcpyAsync(trimmed, d_trimmed, (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(trim_point, d_tp, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { ctx->err = std::string("trim: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
  ctx->trim_escapes = (int64_t)esc.size();
  return MIA_HIP_OK;
}

extern "C" int mia_hip_trim_stats(mia_hip_ctx* ctx, int64_t* exact_reruns) {
  if (!ctx || !exact_reruns) return MIA_HIP_ERR_ARG;
  *exact_reruns = ctx->trim_escapes;
  return MIA_HIP_OK;
}

// ---- ma -------------------------------------------------------------------------------------
extern "C" int mia_hip_ma_tally(mia_hip_ctx* ctx, int32_t ref_len, const int32_t* gaps, int64_t n, const int32_t* start,
                                const uint8_t* revcom, const int64_t* col_off, const char* seq, const char* smp, int64_t n_ins,
                                const int32_t* ins_record, const int32_t* ins_pos, const int64_t* ins_off, const char* ins_bases) {
  if (!ctx || ref_len <= 0 || !gaps || n < 0 || n_ins < 0 || (n > 0 && (!start || !revcom || !col_off || !seq || !smp)) ||
      (n_ins > 0 && (!ins_record || !ins_pos || !ins_off || !ins_bases)))
    return MIA_HIP_ERR_ARG;
  if (!ctx->have_pssm) { ctx->err = "set_pssm must precede ma_tally"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  for (int64_t r = 0; r < n; r++)
    if (col_off[r + 1] < col_off[r] || start[r] < 0) { ctx->err = "malformed record geometry"; return MIA_HIP_ERR_ARG; }
  for (int64_t e = 0; e < n_ins; e++)
    if (ins_record[e] < 0 || ins_record[e] >= n || ins_off[e + 1] < ins_off[e]) { ctx->err = "malformed insert list"; return MIA_HIP_ERR_ARG; }
  ctx->L = ref_len;
  ctx->wrap = ref_len;
  const int64_t ins_chars = n_ins ? ins_off[n_ins] : 0;
  if (ctx->tb.events && ctx->tb.cap_events < ins_chars + 16) {   // the list was sized for another job
    (void)hipFree(ctx->tb.events); ctx->tb.events = nullptr;
    if (dev_alloc(ctx, &ctx->tb.events, (size_t)ins_chars + 4096)) return MIA_HIP_ERR_NOMEM;
    ctx->tb.cap_events = (int32_t)std::min<int64_t>(ins_chars + 4096, INT32_MAX);
  }
  if (!ctx->tb.events) {
    int rc0 = dev_alloc(ctx, &ctx->tb.events, (size_t)ins_chars + 4096);
    if (rc0) return MIA_HIP_ERR_NOMEM;
    ctx->tb.cap_events = (int32_t)std::min<int64_t>(ins_chars + 4096, INT32_MAX);
  }
  int rc = ensure_tally(ctx);
  if (rc) return rc;
  const int Lp = ctx->tb.Lp;
  const int64_t chars = n ? col_off[n] : 0;
  int32_t *d_start = nullptr, *d_irec = nullptr, *d_ipos = nullptr;
  uint8_t* d_rev = nullptr;
  int64_t *d_coff = nullptr, *d_ioff = nullptr;
  char *d_seq = nullptr, *d_smp = nullptr, *d_ib = nullptr;
  ScopeFree guard;   // every temporary is released on any return
  for (void** pp : {(void**)&d_start, (void**)&d_rev, (void**)&d_coff, (void**)&d_seq, (void**)&d_smp, (void**)&d_irec, (void**)&d_ipos,
                    (void**)&d_ioff, (void**)&d_ib})
    guard.watch(pp);
  int rcx = dev_alloc(ctx, &d_start, (size_t)n + 1) | dev_alloc(ctx, &d_rev, (size_t)n + 1) | dev_alloc(ctx, &d_coff, (size_t)n + 1) |
            dev_alloc(ctx, &d_seq, (size_t)chars + 1) | dev_alloc(ctx, &d_smp, (size_t)chars + 1) | dev_alloc(ctx, &d_irec, (size_t)n_ins + 1) |
            dev_alloc(ctx, &d_ipos, (size_t)n_ins + 1) | dev_alloc(ctx, &d_ioff, (size_t)n_ins + 1) | dev_alloc(ctx, &d_ib, (size_t)ins_chars + 1);
  if (rcx) return MIA_HIP_ERR_NOMEM;
  hipError_t e = hipSuccess;
  auto up = [&](void* d, const void* h, size_t b) { if (e == hipSuccess && b) e = hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, ctx->stream); };
  auto zero = [&](void* d, size_t b) { if (e == hipSuccess) e = hipMemsetAsync(d, 0, b, ctx->stream); };
  zero(ctx->tb.tally, (size_t)TALLY_WORDS * Lp * 4);
  zero(ctx->tb.gaps, (size_t)Lp * 4);
  zero(ctx->tb.n_events, 4);
  zero(ctx->tb.flags, 4);
  up(ctx->tb.gaps, gaps, (size_t)ref_len * 4);
  up(d_start, start, (size_t)n * 4); up(d_rev, revcom, (size_t)n); up(d_coff, col_off, (size_t)(n + 1) * 8);
  up(d_seq, seq, (size_t)chars); up(d_smp, smp, (size_t)chars);
  if (n_ins) { up(d_irec, ins_record, (size_t)n_ins * 4); up(d_ipos, ins_pos, (size_t)n_ins * 4); up(d_ioff, ins_off, (size_t)(n_ins + 1) * 8); up(d_ib, ins_bases, (size_t)ins_chars); }
  if (e == hipSuccess && n > 0) {
    MaRecords mr{n, d_start, d_rev, d_coff, d_seq, d_smp};
    hipLaunchKernelGGL(k_ma_tally, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, mr, ctx->d_pssm, ctx->tb);
    if (n_ins > 0)
      hipLaunchKernelGGL(k_ma_ins_events, dim3((unsigned)((n_ins + 255) / 256)), dim3(256), 0, ctx->stream, mr, n_ins, d_irec, d_ipos, d_ioff,
                         d_ib, ctx->tb);
    e = hipGetLastError();
  }
  uint32_t flags = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&ctx->n_events_host, ctx->tb.n_events, 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&flags, ctx->tb.flags, 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { ctx->err = std::string("ma_tally: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
  if (flags & 1u) { ctx->err = "insert event list overflow"; return MIA_HIP_ERR_NOMEM; }
  if (flags & 2u) { ctx->err = "a record reaches past the reference or carries a depth code outside A.._"; return MIA_HIP_ERR_ARG; }
  if (ctx->n_events_host > ctx->tb.cap_events) ctx->n_events_host = ctx->tb.cap_events;
  ctx->tallied = true;
  ctx->consensus_done = false;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_get_ins_tally(mia_hip_ctx* ctx, int32_t* ins_off, int32_t* ins_tally, int64_t cap_slots, int64_t* n_slots) {
  if (!ctx) return MIA_HIP_ERR_ARG;
  if (!ctx->tallied || !ctx->consensus_done) { ctx->err = "consensus first"; return MIA_HIP_ERR_STATE; }
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t total = ctx->ins_total_host;
  if (n_slots) *n_slots = total;
  if (ins_off) HIPCHK(hipMemcpyAsync(ins_off, ctx->d_ins_off, (size_t)ctx->tb.Lp * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (ins_tally && total > 0) {
    if (cap_slots < total) { ctx->err = "ins_tally buffer too small"; return MIA_HIP_ERR_ARG; }
    HIPCHK(hipMemcpyAsync(ins_tally, ctx->d_ins_tally, (size_t)total * 9 * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MIA_HIP_OK;
}

// ---- pass 1 ----------------------------------------------------------------------------
static char revcom_char_host(char b) {   // src/map_align.c:418-432
  static const char tbl[] = "TVGH\0\0CD\0\0M\0KN\0\0\0YSAABWXR\0";
  char r = 0;
  if (b == '-') return '-';
  if (b >= 'A' && b <= 'Z') r = tbl[b - 'A'];
  else if (b >= 'a' && b <= 'z') r = (char)(tbl[b - 'a'] + 32);
  return r ? r : 'N';
}

// populate_kpa + add_kmer (src/kmer.c:65-107,153-168): the k-mers that occur in `seq`, each with its ascending
// positions (at most 128; further ones are silently dropped, :75-77).  Only the occurring k-mers are listed -- the
// dense 4^k table is zeroed and filled on the device.
static void build_kmer_lists(const std::string& seq, int k, int soft_mask, std::vector<uint32_t>& kmer, std::vector<uint32_t>& entry,
                             std::vector<int32_t>& pos) {
  const int n = (int)seq.size();
  std::vector<uint64_t> kp;   // (k-mer << 32) | position: sorting keeps positions ascending inside a k-mer
  kp.reserve((size_t)(n > 0 ? n : 1));
  for (int i = 0; i + k <= n; i++) {
    bool ok = true;
    uint64_t v = 0;
    for (int t = 0; t < k && ok; t++) {
      char c = seq[i + t];
      if (soft_mask && (c >= 'a' && c <= 'z')) ok = false;
      switch (c & ~32) { case 'A': v = v << 2; break; case 'C': v = (v << 2) | 1; break; case 'G': v = (v << 2) | 2; break;
                         case 'T': v = (v << 2) | 3; break; default: ok = false; }
    }
    if (ok) kp.push_back((v << 32) | (uint32_t)i);
  }
  std::sort(kp.begin(), kp.end());
  kmer.clear(); entry.clear(); pos.clear();
  for (size_t i = 0; i < kp.size();) {
    size_t j = i;
    while (j < kp.size() && (kp[j] >> 32) == (kp[i] >> 32)) j++;
    const size_t cnt = (j - i) < (size_t)MAX_KMER_POS ? (j - i) : (size_t)MAX_KMER_POS;
    kmer.push_back((uint32_t)(kp[i] >> 32));
    entry.push_back((uint32_t)(pos.size() << 8) | (uint32_t)cnt);
    for (size_t t = 0; t < cnt; t++) pos.push_back((int32_t)(uint32_t)kp[i + t]);
    i = j;
  }
  pos.push_back(0);
}

// Lets align_all run on a borrowed read set and reference (the anchored part of pass 1) and puts the context back as it was.
struct AlignBorrow {
  mia_hip_ctx* c;
  ReadSet rs; int32_t *bin_of, *list, *wide, *retry; int max_len; uint8_t* d_ref; int L, wrap, explicit_win, use_filter;
  bool aligned, culled, tallied, pre_cull_valid, ref_mostly_bases, diag_scripts_missing;
  int64_t kh_entries;
  int64_t plain_total, plain_retried, filter_seen, filter_proven, bx_seen, bx_done0, bx_done1, bx_done2;
  double stg_ms[STG_COUNT]; int64_t stg_launches[STG_COUNT];
  explicit AlignBorrow(mia_hip_ctx* ctx)
      : c(ctx), rs(ctx->rs), bin_of(ctx->d_bin_of), list(ctx->d_list), wide(ctx->d_wide_list), retry(ctx->d_retry_list), max_len(ctx->max_len),
        d_ref(ctx->d_ref), L(ctx->L), wrap(ctx->wrap), explicit_win(ctx->explicit_win), use_filter(ctx->use_filter), aligned(ctx->aligned),
        culled(ctx->culled), tallied(ctx->tallied), pre_cull_valid(ctx->pre_cull_valid), ref_mostly_bases(ctx->ref_mostly_bases),
        diag_scripts_missing(ctx->diag_scripts_missing), kh_entries(ctx->kh_entries), plain_total(ctx->plain_total), plain_retried(ctx->plain_retried), filter_seen(ctx->filter_seen), filter_proven(ctx->filter_proven),
        bx_seen(ctx->bx_seen), bx_done0(ctx->bx_done[0]), bx_done1(ctx->bx_done[1]), bx_done2(ctx->bx_done[2]) {
    // the stage timers of the iteration path must not see what the borrowed runs add (bench.py's roofline reads them)
    for (int k = 0; k < STG_COUNT; k++) {
      if (k == STG_PASS1) continue;
      float ms = 0;
      for (auto& e : ctx->stg[k].pending) {
        if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { ctx->stg[k].ms += ms; ctx->stg[k].launches++; }
        ctx->ev_free.push_back(e);
      }
      ctx->stg[k].pending.clear();
      stg_ms[k] = ctx->stg[k].ms; stg_launches[k] = ctx->stg[k].launches;
    }
  }
  ~AlignBorrow() {
    for (int k = 0; k < STG_COUNT; k++) {
      if (k == STG_PASS1) continue;
      for (auto& e : c->stg[k].pending) { (void)hipEventSynchronize(e.second); c->ev_free.push_back(e); }
      c->stg[k].pending.clear();
      c->stg[k].ms = stg_ms[k]; c->stg[k].launches = stg_launches[k];
    }
    c->rs = rs; c->d_bin_of = bin_of; c->d_list = list; c->d_wide_list = wide; c->d_retry_list = retry; c->max_len = max_len; c->d_ref = d_ref;
    c->L = L; c->wrap = wrap; c->explicit_win = explicit_win; c->use_filter = use_filter; c->aligned = aligned; c->culled = culled;
    c->tallied = tallied; c->pre_cull_valid = pre_cull_valid; c->ref_mostly_bases = ref_mostly_bases; c->kh_entries = kh_entries; c->diag_scripts_missing = diag_scripts_missing; c->plain_total = plain_total;
    c->plain_retried = plain_retried; c->filter_seen = filter_seen; c->filter_proven = filter_proven;
    c->bx_seen = bx_seen; c->bx_done[0] = bx_done0; c->bx_done[1] = bx_done1; c->bx_done[2] = bx_done2;
  }
};

extern "C" int mia_hip_pass1(mia_hip_ctx* ctx, const char* ref, int32_t ref_len, int circular, int kmer_len, int soft_mask,
                             int64_t n, const char* bases, const int64_t* offsets, int32_t* score, uint8_t* rc, int32_t* as,
                             int32_t* ae, uint8_t* flags) {
  if (!ctx || !ref || ref_len <= 0 || n < 0 || (n > 0 && (!bases || !offsets || !score || !rc || !as || !ae || !flags))) return MIA_HIP_ERR_ARG;
  if (!ctx->have_pssm) { ctx->err = "set_pssm must precede pass1"; return MIA_HIP_ERR_STATE; }
  const bool timing = getenv("MIA_HIP_P1_TIMING") != nullptr;
  auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[mia_hip_pass1] %-12s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_start).count());
    t_start = t;
  };
  if (kmer_len > 14) { ctx->err = "Cannot use kmer length greater than 14"; return MIA_HIP_ERR_ARG; }   // MAX_KMER_LEN
  HIPCHK(hipSetDevice(ctx->device));
  if (n == 0) return MIA_HIP_OK;
  PackParams pk;
  if (!make_pack_params(1024, ctx->max_abs, &pk)) { ctx->err = "PSSM too large for the packed pass-1 kernel"; return MIA_HIP_ERR_RANGE; }
  // reference strands: make_reverse_complement, add_ref_wrap, (k-mer tables), make_ref_upper -- src/mia_main.c:637-676
  const int L = ref_len, wl = circular ? (L < MAX_READ ? L : MAX_READ) : 0, wrap = L + wl, len1 = circular ? wrap : L;
  std::string fw(ref, ref + L), rcs((size_t)L, 'N');
  for (int i = 0; i < L; i++) rcs[i] = revcom_char_host(ref[L - 1 - i]);
  fw += fw.substr(0, wl);
  rcs += rcs.substr(0, wl);
  std::vector<uint8_t> cf((size_t)wrap + 64, 4), cr((size_t)wrap + 64, 4);
  for (int i = 0; i < wrap; i++) { cf[i] = base_code((char)toupper((unsigned char)fw[i])); cr[i] = base_code((char)toupper((unsigned char)rcs[i])); }
  // reads (as sequenced); validated before anything is allocated on the device
  std::vector<uint32_t> roff((size_t)n);
  std::vector<uint16_t> len((size_t)n);
  uint64_t total = 0;
  int max_len = 1;
  for (int64_t i = 0; i < n; i++) {
    int64_t l = offsets[i + 1] - offsets[i];
    if (l < 1 || l > MIA_HIP_MAX_READ) { ctx->err = "read length outside 1..256"; return MIA_HIP_ERR_ARG; }
    roff[i] = (uint32_t)total; len[i] = (uint16_t)l;
    if (l > max_len) max_len = (int)l;
    total += (uint64_t)(((l + 1) / 2 + 3) & ~3);
    if (total >= ((uint64_t)1 << 32)) { ctx->err = "packed read store exceeds 4 GiB per call"; return MIA_HIP_ERR_ARG; }
  }
  // Chunk width.  Dropping candidates further left than the horizon `rel` is exact only if they can never beat a
  // new start: a score is at most rows * max positive entry, a new start costs P(rows+1), a gap of >= rel-1 columns
  // costs P(rel-1).  Wide chunks (768 columns, horizon 256) for the unmasked sweep when that holds, else narrow
  // ones (256 columns, horizon 768), which also skip masked stretches at a finer grain.
  auto horizon_ok = [&](int rel) {
    return (int64_t)max_len * ctx->max_pos + (GOP + GEP * (max_len + 1)) < (int64_t)(GOP + GEP * (rel - 1));
  };
  int cpl = (kmer_len <= 0 && horizon_ok(p1_rel(P1_CPL_WIDE))) ? P1_CPL_WIDE : P1_CPL_NARROW;
  if (const char* ev = getenv("MIA_HIP_P1_CPL")) {
    const int want = atoi(ev);
    if ((want == P1_CPL_WIDE && horizon_ok(p1_rel(P1_CPL_WIDE))) || want == P1_CPL_NARROW) cpl = want;
  }
  if (!horizon_ok(p1_rel(cpl))) { ctx->err = "PSSM too large for the pass-1 candidate horizon"; return MIA_HIP_ERR_RANGE; }
  const int P1_CH = p1_ch(cpl);
  // the wide unmasked sweep runs on plain keys (pass1_body.h, run_plain); MIA_HIP_P1_PLAIN=0 keeps the packed sweep
  int plain = 1;
  if (const char* ev = getenv("MIA_HIP_P1_PLAIN")) plain = atoi(ev) != 0;
  if (kmer_len > 0 && (size_t)wrap >= ((size_t)1 << 24)) { ctx->err = "reference too long for the k-mer table"; return MIA_HIP_ERR_RANGE; }
  uint8_t *d_cf = nullptr, *d_cr = nullptr;
  uint32_t *d_tab[2] = {nullptr, nullptr}, *d_kl[2] = {nullptr, nullptr}, *d_el[2] = {nullptr, nullptr};
  int32_t* d_pos[2] = {nullptr, nullptr};
  uint8_t *d_packed = nullptr, *d_rc = nullptr, *d_flags = nullptr;
  uint32_t *d_roff = nullptr, *d_status = nullptr;
  uint16_t* d_len = nullptr;
  int32_t *d_score = nullptr, *d_as = nullptr, *d_ae = nullptr;
  unsigned char* d_trace = nullptr;
  uint32_t* d_ckpt = nullptr;
  uint64_t* d_p1planes = nullptr;
  int32_t *d_p1kcnt = nullptr, *d_p1kpos = nullptr;
  int32_t* d_todo = nullptr;
  uint32_t* d_ntodo = nullptr;
  PoolScope guard(ctx);   // every temporary below goes back to the context's pool on any return
  guard.watch((void**)&d_p1planes); guard.watch((void**)&d_todo); guard.watch((void**)&d_ntodo); guard.watch((void**)&d_p1kcnt); guard.watch((void**)&d_p1kpos);
  for (void** pp : {(void**)&d_cf, (void**)&d_cr, (void**)&d_tab[0], (void**)&d_tab[1], (void**)&d_kl[0], (void**)&d_kl[1], (void**)&d_el[0],
                    (void**)&d_el[1], (void**)&d_pos[0], (void**)&d_pos[1], (void**)&d_packed, (void**)&d_rc, (void**)&d_flags, (void**)&d_roff,
                    (void**)&d_status, (void**)&d_len, (void**)&d_score, (void**)&d_as, (void**)&d_ae, (void**)&d_trace, (void**)&d_ckpt})
    guard.watch(pp);
  int rcx = pool_alloc(ctx, &d_cf, cf.size()) | pool_alloc(ctx, &d_cr, cr.size());
  KmerIndex kx{};
  kx.k = kmer_len > 0 ? kmer_len : -1;
  if (kx.k > 0 && !rcx) {
    const size_t nk = (size_t)1 << (2 * kx.k);
    for (int s = 0; s < 2 && !rcx; s++) {
      std::vector<uint32_t> kmer, entry;
      std::vector<int32_t> pos;
      build_kmer_lists(s ? rcs : fw, kx.k, soft_mask, kmer, entry, pos);
      const size_t ne = kmer.size();
      rcx |= pool_alloc(ctx, &d_tab[s], nk) | pool_alloc(ctx, &d_pos[s], pos.size()) | pool_alloc(ctx, &d_kl[s], ne + 1) |
             pool_alloc(ctx, &d_el[s], ne + 1);
      if (rcx) break;
      hipError_t ke = hipMemsetAsync(d_tab[s], 0, nk * 4, ctx->stream);
      if (ke == hipSuccess) ke = hipMemcpy(d_pos[s], pos.data(), pos.size() * 4, hipMemcpyHostToDevice);
      if (ke == hipSuccess && ne) ke = hipMemcpy(d_kl[s], kmer.data(), ne * 4, hipMemcpyHostToDevice);
      if (ke == hipSuccess && ne) ke = hipMemcpy(d_el[s], entry.data(), ne * 4, hipMemcpyHostToDevice);
      if (ke == hipSuccess && ne) {
        hipLaunchKernelGGL(k_kmer_fill, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream, (int32_t)ne, d_kl[s], d_el[s], d_tab[s]);
        ke = hipGetLastError();
      }
      if (ke != hipSuccess) { ctx->err = std::string("pass1 k-mer table: ") + hipGetErrorString(ke); rcx = 2; }
      kx.tab[s] = d_tab[s];
      kx.pos[s] = d_pos[s];
    }
  }
  lap("ref+kmer");
  std::vector<uint8_t> packed((size_t)total + 8, 0);
  pack_reads(n, bases, offsets, roff.data(), len.data(), packed.data());
  rcx |= pool_alloc(ctx, &d_packed, packed.size()) | pool_alloc(ctx, &d_roff, (size_t)n) | pool_alloc(ctx, &d_len, (size_t)n) |
         pool_alloc(ctx, &d_rc, (size_t)n) | pool_alloc(ctx, &d_flags, (size_t)n) | pool_alloc(ctx, &d_status, (size_t)n) |
         pool_alloc(ctx, &d_score, (size_t)n) | pool_alloc(ctx, &d_as, (size_t)n) | pool_alloc(ctx, &d_ae, (size_t)n);
  // persistent grid; LDS: sub table + 5 carry arrays + 2 column masks
  // (the column masks are only read when the k-mer filter is on)
  const int nch = (len1 + P1_CH - 1) / P1_CH, mask_words = kmer_len > 0 ? nch * (P1_CH / 32) + 4 : 0;
  const int lds = MAX_READ * 10 + 5 * MAX_READ * 4 + 2 * mask_words * 4;
  const int rows_p = (max_len + 3) & ~3;
  const int64_t trace_bytes = (int64_t)rows_p * P1_CH * 2, ckpt_words = (int64_t)2 * nch * 5 * rows_p;
  if (lds > 160 * 1024) { ctx->err = "reference too long for the pass-1 LDS masks"; return MIA_HIP_ERR_RANGE; }
  // persistent grid = exactly the waves that are resident at once (registers AND LDS): a wave that only starts
  // when another has drained would run its whole share of the reads on a nearly idle GPU
  auto kfn = (cpl == P1_CPL_WIDE) ? k_pass1<P1_CPL_WIDE> : k_pass1<P1_CPL_NARROW>;
  HIPCHK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int waves_cu = 0;
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&waves_cu, (const void*)kfn, 64, (size_t)lds));
  if (waves_cu < 1) { ctx->err = "pass-1 kernel does not fit a compute unit"; return MIA_HIP_ERR_RANGE; }
  // whole waves per SIMD only: an uneven remainder (13 = 4+3+3+3) measured slower than 12, the extra workgroups start late
  if (waves_cu > 4) waves_cu &= ~3;
  if (const char* ev = getenv("MIA_HIP_P1_WAVES_PER_CU")) { const int wv = atoi(ev); if (wv > 0 && wv < waves_cu) waves_cu = wv; }
  if (timing) fprintf(stderr, "[mia_hip_pass1] cpl %d plain %d waves/CU %d lds %d\n", cpl, plain, waves_cu, lds);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
  int64_t grid = (int64_t)prop.multiProcessorCount * waves_cu;
  if (grid > n) grid = n;
  rcx |= pool_alloc(ctx, &d_trace, (size_t)(trace_bytes * grid)) | pool_alloc(ctx, &d_ckpt, (size_t)(ckpt_words * grid));
  if (rcx) return rcx == 2 ? MIA_HIP_ERR_DEVICE : MIA_HIP_ERR_NOMEM;
  lap("pack+alloc");
  hipError_t e = hipSuccess;
  bool p1_timed = false;
  auto cp = [&](void* d, const void* h, size_t b) { if (e == hipSuccess) e = hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, ctx->stream); };
  cp(d_cf, cf.data(), cf.size()); cp(d_cr, cr.data(), cr.size());
  cp(d_packed, packed.data(), packed.size()); cp(d_roff, roff.data(), (size_t)n * 4); cp(d_len, len.data(), (size_t)n * 2);
  // The diagonal filter (diag_filter.h) first, when its premises hold: flat matrix and no k-mer mask (a masked DP is a
  // different recurrence).  It decides most reads by bit-parallel comparison against every diagonal of both strands;
  // the whole-reference DP then runs on what is left.
  int64_t p1_other = 0;
  for (int i = 0; i < L; i++) p1_other += cf[(size_t)i] > 3;
  const bool fast_ok = ctx->flat && ctx->use_filter && kmer_len <= 0 && len1 >= max_len;
  // any other matrix the band pipeline has tables for: the anchored windows in losses (mia_pass1_kernels.h, GEN); the
  // diagonal filter stays the flat matrix's
  const bool gen_ok = !ctx->flat && ctx->bx_ok && ctx->use_bx && ctx->use_filter && kmer_len <= 0 && len1 >= max_len && !getenv("MIA_HIP_NO_ANCHOR_GEN");
  const bool filtered = fast_ok && p1_other * 50 <= L;
  // the anchored stage behind it (or in its place: a reference full of ambiguity codes, mt311 itself, leaves the filter
  // nothing to decide): the windows' 10-mer tables list the N columns under every spelling (bandx_body.h, N COLUMNS)
  int64_t wild_entries = 0;
  if ((fast_ok || gen_ok) && p1_other && ctx->use_wild && len1 <= (1 << 22)) {
    std::vector<uint8_t> both(cf.begin(), cf.begin() + len1);
    both.insert(both.end(), cr.begin(), cr.begin() + len1);
    wild_entries = kh_wild_entries(both.data(), (int64_t)both.size(), BX_WILD);
    if (wild_entries > ((int64_t)1 << 24)) wild_entries = 0;
  }
  const bool anchored_ok = (fast_ok || gen_ok) && (p1_other == 0 || wild_entries > 0) && len1 <= (1 << 22) && !getenv("MIA_HIP_NO_ANCHOR");
  int64_t n_dp = n;
  ctx->pass1_filtered = 0;
  ctx->pass1_anchored = 0;
  if (e == hipSuccess) {
    Pass1Reads pr{n, d_packed, d_roff, d_len, d_score, d_as, d_ae, d_rc, d_flags, d_status};
    drain_events(ctx);                       // (nothing of an earlier call may sit in the pass-1 timer)
    p1_timed = ((ctx->stage_mask >> STG_PASS1) & 1u) && stage_begin(ctx, STG_PASS1) == 0;
    if (filtered || anchored_ok) {
      if (pool_alloc(ctx, &d_todo, (size_t)n) || pool_alloc(ctx, &d_ntodo, 1)) return MIA_HIP_ERR_NOMEM;
      if (len1 <= (1 << 22) && (pool_alloc(ctx, &d_p1kcnt, (size_t)DF_KTAB * 2) || pool_alloc(ctx, &d_p1kpos, (size_t)DF_KTAB * DF_KCAP * 2))) return MIA_HIP_ERR_NOMEM;
    }
    if (!filtered && anchored_ok) hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, d_todo);
    if (filtered) {
      const int64_t words = plane_words(len1);
      if (pool_alloc(ctx, &d_p1planes, (size_t)words * 6)) return MIA_HIP_ERR_NOMEM;
      uint64_t* pl = d_p1planes;
      hipLaunchKernelGGL(k_ref_planes, dim3((unsigned)((words + 3) / 4 < 4096 ? (words + 3) / 4 : 4096)), dim3(256), 0, ctx->stream, d_cf, (int64_t)len1, words, pl, pl + words, pl + 2 * words);
      hipLaunchKernelGGL(k_ref_planes, dim3((unsigned)((words + 3) / 4 < 4096 ? (words + 3) / 4 : 4096)), dim3(256), 0, ctx->stream, d_cr, (int64_t)len1, words, pl + 3 * words, pl + 4 * words, pl + 5 * words);
      e = hipMemsetAsync(d_ntodo, 0, 4, ctx->stream);
      RefPlanes pf{pl, pl + words, pl + 2 * words}, prc{pl + 3 * words, pl + 4 * words, pl + 5 * words};
      // 10-mer tables of both strands for rule (c) (diag_filter.h: KmerOcc)
      KmerOcc kf{nullptr, nullptr}, kr{nullptr, nullptr};
      if (d_p1kcnt) {
        if (e == hipSuccess) e = hipMemsetAsync(d_p1kcnt, 0, (size_t)DF_KTAB * 2 * 4, ctx->stream);
        hipLaunchKernelGGL(k_kmer_occ, dim3((unsigned)((len1 + 255) / 256)), dim3(256), 0, ctx->stream, d_cf, (int64_t)len1, d_p1kcnt, d_p1kpos, 0);
        hipLaunchKernelGGL(k_kmer_occ, dim3((unsigned)((len1 + 255) / 256)), dim3(256), 0, ctx->stream, d_cr, (int64_t)len1, d_p1kcnt + DF_KTAB,
                           d_p1kpos + DF_KTAB * DF_KCAP, 0);
        kf.cnt = d_p1kcnt; kf.pos = d_p1kpos; kr.cnt = d_p1kcnt + DF_KTAB; kr.pos = d_p1kpos + DF_KTAB * DF_KCAP;
      }
      hipLaunchKernelGGL(k_pass1_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, pr, pf, prc, kf, kr, len1, L, d_todo, d_ntodo);
      if (e == hipSuccess) e = hipGetLastError();
      uint32_t h_ntodo = 0;
      if (e == hipSuccess) e = hipMemcpyAsync(&h_ntodo, d_ntodo, 4, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      n_dp = h_ntodo;
      ctx->pass1_filtered = n - n_dp;
      if (timing) fprintf(stderr, "[mia_hip_pass1] diagonal filter: %lld of %lld reads decided\n", (long long)(n - n_dp), (long long)n);
    }
    // Anchored stage (mia_pass1_kernels.h): what the filter left over is aligned in +-50 windows around the places where
    // six 10-mers of the read occur -- provably the same result as the whole-strand DP when the budget check of
    // k_pass1_select holds; everything else goes on to k_pass1.  Needs both strands free of N (a 10-mer with an N is not
    // in the table and an N column costs less than a mismatch).
    if (e == hipSuccess && anchored_ok && n_dp > 0 && d_p1kcnt) {
      if (p1_other || !filtered) {            // the anchors' tables: with the N columns' spellings (the filter's rule (c) wants them without)
        HIPCHK(hipMemsetAsync(d_p1kcnt, 0, (size_t)DF_KTAB * 2 * 4, ctx->stream));
        const int wild = p1_other ? BX_WILD : 0;
        hipLaunchKernelGGL(k_kmer_occ, dim3((unsigned)((len1 + 255) / 256)), dim3(256), 0, ctx->stream, d_cf, (int64_t)len1, d_p1kcnt, d_p1kpos, wild);
        hipLaunchKernelGGL(k_kmer_occ, dim3((unsigned)((len1 + 255) / 256)), dim3(256), 0, ctx->stream, d_cr, (int64_t)len1, d_p1kcnt + DF_KTAB,
                           d_p1kpos + DF_KTAB * DF_KCAP, wild);
      }
      const int64_t m = n_dp * P1A_SLOTS;
      const int stride = (max_len + 3) & ~3;
      // one allocation for all the temporaries of this stage (twenty separate ones cost more than the stage)
      unsigned char* arena = nullptr;
      PoolScope g2(ctx);
      g2.watch((void**)&arena);
      size_t top = 0;
      auto carve = [&](size_t bytes) { const size_t at = top; top += (bytes + 255) & ~(size_t)255; return at; };
      const size_t o_roff = carve((size_t)m * 4), o_status = carve((size_t)m * 4), o_nrest = carve(64), o_len = carve((size_t)m * 2),
                   o_rc = carve((size_t)m), o_sk = carve((size_t)m), o_ref2 = carve((size_t)2 * len1 + 64), o_as = carve((size_t)m * 4),
                   o_ae = carve((size_t)m * 4), o_score = carve((size_t)m * 4), o_refstart = carve((size_t)m * 4), o_bin = carve((size_t)m * 4),
                   o_list = carve(((size_t)m + 4 * N_BINS) * 4), o_wide = carve((size_t)m * 4), o_retry = carve((size_t)m * 4),
                   o_rest = carve((size_t)n_dp * 4), o_abr = carve((size_t)m * 2), o_cols = carve((size_t)m * stride * 2),
                   o_bound = carve((size_t)n_dp * 4), o_budget = carve((size_t)n_dp * 4), o_u = carve((size_t)n_dp * 4);
      if (pool_alloc(ctx, &arena, top)) return MIA_HIP_ERR_NOMEM;
      uint32_t *w_roff = (uint32_t*)(arena + o_roff), *w_status = (uint32_t*)(arena + o_status), *d_nrest = (uint32_t*)(arena + o_nrest);
      uint16_t* w_len = (uint16_t*)(arena + o_len);
      uint8_t *w_rc = arena + o_rc, *w_sk = arena + o_sk, *d_ref2 = arena + o_ref2;
      int32_t *w_as = (int32_t*)(arena + o_as), *w_ae = (int32_t*)(arena + o_ae), *w_score = (int32_t*)(arena + o_score),
              *w_refstart = (int32_t*)(arena + o_refstart), *w_bin = (int32_t*)(arena + o_bin), *w_list = (int32_t*)(arena + o_list),
              *w_wide = (int32_t*)(arena + o_wide), *w_retry = (int32_t*)(arena + o_retry), *d_rest = (int32_t*)(arena + o_rest),
              *w_bound = (int32_t*)(arena + o_bound), *w_budget = (int32_t*)(arena + o_budget), *w_u = (int32_t*)(arena + o_u);
      int16_t *w_abr = (int16_t*)(arena + o_abr), *w_cols = (int16_t*)(arena + o_cols);
      HIPCHK(hipMemsetAsync(w_rc, 0, (size_t)m, ctx->stream));
      HIPCHK(hipMemsetAsync(w_abr, 0, (size_t)m * 2, ctx->stream));
      HIPCHK(hipMemsetAsync(w_status, 0, (size_t)m * 4, ctx->stream));
      HIPCHK(hipMemsetAsync(w_score, 0, (size_t)m * 4, ctx->stream));
      HIPCHK(hipMemsetAsync(d_nrest, 0, 64, ctx->stream));
      HIPCHK(hipMemsetAsync(d_ref2, 4, (size_t)2 * len1 + 64, ctx->stream));
      HIPCHK(hipMemcpyAsync(d_ref2, d_cf, (size_t)len1, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(d_ref2 + len1, d_cr, (size_t)len1, hipMemcpyDeviceToDevice, ctx->stream));
      KmerOcc kf{d_p1kcnt, d_p1kpos}, kr{d_p1kcnt + DF_KTAB, d_p1kpos + DF_KTAB * DF_KCAP};
      BxTab p1tab;
      p1tab.sub = ctx->d_bx_sub; p1tab.mrow = ctx->d_bx_mrow; p1tab.loss = ctx->d_bx_loss; p1tab.dl = ctx->d_bx_dl; p1tab.min_m = ctx->bx_min_m; p1tab.max_m = ctx->bx_max_m;
      if (ctx->flat)
        hipLaunchKernelGGL(k_pass1_anchor<false>, dim3((unsigned)((n_dp + 255) / 256)), dim3(256), 0, ctx->stream, pr, d_todo, n_dp, kf, kr, len1, w_roff, w_len,
                           w_sk, w_as, w_ae, w_bound, w_budget, w_u, p1tab);
      else
        hipLaunchKernelGGL(k_pass1_anchor<true>, dim3((unsigned)((n_dp + 255) / 256)), dim3(256), 0, ctx->stream, pr, d_todo, n_dp, kf, kr, len1, w_roff, w_len,
                           w_sk, w_as, w_ae, w_bound, w_budget, w_u, p1tab);
      HIPCHK(hipGetLastError());
      int rc_inner;
      {
        AlignBorrow borrow(ctx);
        ReadSet& r = ctx->rs;
        r.n = m; r.packed = d_packed; r.roff = w_roff; r.len = w_len; r.rc = w_rc; r.sk = w_sk; r.as = w_as; r.ae = w_ae; r.score = w_score;
        r.refstart = w_refstart; r.abr = w_abr; r.status = w_status; r.cols = w_cols; r.stride = stride;
        ctx->d_bin_of = w_bin; ctx->d_list = w_list; ctx->d_wide_list = w_wide; ctx->d_retry_list = w_retry; ctx->max_len = max_len;
        ctx->d_ref = d_ref2; ctx->L = 2 * len1; ctx->wrap = 2 * len1; ctx->explicit_win = 1; ctx->use_filter = 0;
        ctx->ref_mostly_bases = p1_other * 50 <= L; ctx->kh_entries = wild_entries;
        ctx->wide_to_caller = true;
        rc_inner = align_all(ctx);
        ctx->wide_to_caller = false;
      }
      if (rc_inner != MIA_HIP_OK) return rc_inner;
      hipLaunchKernelGGL(k_pass1_select, dim3((unsigned)((n_dp + 255) / 256)), dim3(256), 0, ctx->stream, pr, d_todo, n_dp, len1, L, w_sk, w_score, w_as,
                         w_ae, w_abr, w_status, w_bound, w_budget, w_u, d_rest, d_nrest);
      HIPCHK(hipGetLastError());
      uint32_t h_rest[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      HIPCHK(hipMemcpyAsync(h_rest, d_nrest, 64, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      const uint32_t h_nrest = h_rest[0];
      if (timing) fprintf(stderr, "[mia_hip_pass1] left to the whole-strand DP: %u no cluster, %u unfinished windows, %u clipped, %u over budget, %u weak clusters\n",
                          h_rest[1], h_rest[2], h_rest[3], h_rest[4], h_rest[5]);
      if (timing) fprintf(stderr, "[mia_hip_pass1] no cluster: %u reads with N, %u too few blocks, %u too many clusters, %u too wide, %u too many strong, %u none strong, %u no room\n",
                          h_rest[9], h_rest[10], h_rest[11], h_rest[12], h_rest[13], h_rest[14], h_rest[15]);
      // the survivors' list replaces the filter's (d_todo is at least as long)
      HIPCHK(hipMemcpyAsync(d_todo, d_rest, (size_t)h_nrest * 4, hipMemcpyDeviceToDevice, ctx->stream));
      ctx->pass1_anchored = n_dp - h_nrest;
      if (timing) fprintf(stderr, "[mia_hip_pass1] anchored windows: %lld of %lld left-over reads decided\n", (long long)(n_dp - h_nrest), (long long)n_dp);
      n_dp = h_nrest;
    }
    if (e == hipSuccess && n_dp > 0) {
      const int64_t g = grid < n_dp ? grid : n_dp;
      hipLaunchKernelGGL(kfn, dim3((unsigned)g), dim3(64), lds, ctx->stream, pr, d_cf, d_cr, len1, L, ctx->d_pssm, pk, kx, d_trace,
                         trace_bytes, d_ckpt, ckpt_words, rows_p, mask_words, plain, (filtered || anchored_ok) ? d_todo : nullptr, n_dp);
      e = hipGetLastError();
    }
    if (p1_timed) stage_end(ctx, STG_PASS1);
  }
  auto back = [&](void* h, const void* d, size_t b) { if (e == hipSuccess) e = hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, ctx->stream); };
  back(score, d_score, (size_t)n * 4); back(as, d_as, (size_t)n * 4); back(ae, d_ae, (size_t)n * 4);
  back(rc, d_rc, (size_t)n); back(flags, d_flags, (size_t)n);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  lap("h2d+kernel+d2h");
  if (p1_timed && !ctx->stg[STG_PASS1].pending.empty()) {
    // the whole call's device time (filter, anchored windows and DP), measured on the pair taken above
    const auto pr = ctx->stg[STG_PASS1].pending.back();
    float ms = 0;
    if (e == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) ctx->pass1_ms = ms;
  }
  lap("done");
  if (e != hipSuccess) { ctx->err = std::string("pass1: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
  return MIA_HIP_OK;
}

// ---- Myers ------------------------------------------------------------------------------
extern "C" int mia_hip_myers(mia_hip_ctx* ctx, int64_t n, const char* const* seq_a, const char* const* seq_b, const int32_t* mode,
                             const int32_t* maxd, uint32_t* dist) {
  if (!ctx || n < 0 || (n > 0 && (!seq_a || !seq_b || !mode || !maxd || !dist))) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  if (n == 0) return MIA_HIP_OK;
  // two kinds of pair: seq_a of up to 320 characters -- one pair per lane (k_myers_lanes), sequences packed as 4-bit IUPAC
  // bitmaps -- and longer ones, one pair per wavefront (k_myers), as ASCII
  static const struct Bits { uint8_t t[256]; Bits() { for (int c = 0; c < 256; c++) t[c] = (uint8_t)iupac_bits((char)c); } } bits;
  std::vector<MyersPair> pairs;
  std::vector<MyersLanePair> lp;
  std::vector<int32_t> long_index, lane_index;
  std::vector<uint32_t> codes;
  std::string blob;
  std::vector<size_t> oa, ob;
  int max_blk = 1;
  // lengths first (on all host threads: a hundred thousand strlen calls and forty million characters to pack are most of
  // this call's time next to a 0.3 ms kernel), then the offsets, then the packing into the places they name
  std::vector<uint32_t> la_of((size_t)n), lb_of((size_t)n);
  int T = (int)std::thread::hardware_concurrency();
  T = std::max(1, std::min(std::min(T, 32), (int)(n / 2048) + 1));
  auto parallel = [&](auto&& fn) {
    if (T == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back([&fn, t] { fn(t); });
    fn(0);
    for (auto& x : th) x.join();
  };
  bool too_long = false;
  parallel([&](int t) {
    for (int64_t i = n * t / T, hi = n * (t + 1) / T; i < hi; i++) {
      const size_t la = strlen(seq_a[i]), lb = strlen(seq_b[i]);
      if (la > 64u * 64u * MYERS_MAX_K || lb >= ((size_t)1 << 31)) { too_long = true; la_of[(size_t)i] = 0; lb_of[(size_t)i] = 0; continue; }
      la_of[(size_t)i] = (uint32_t)la; lb_of[(size_t)i] = (uint32_t)lb;
    }
  });
  if (too_long) { ctx->err = "seq_a longer than 32768 characters"; return MIA_HIP_ERR_ARG; }
  std::vector<uint32_t> word_off((size_t)n, 0);
  {
    uint64_t words = 0;
    for (int64_t i = 0; i < n; i++) {
      const size_t la = la_of[(size_t)i], lb = lb_of[(size_t)i];
      const uint64_t need = (la + 7) / 8 + 1 + (lb + 7) / 8 + 1;       // (one spare word behind each sequence: the kernel may fetch it)
      if (la <= 64u * MYERS_LANE_K && words + need < ((uint64_t)1 << 31) && !ctx->myers_no_lanes) {
        MyersLanePair q;
        q.a_off = (uint32_t)words; q.b_off = (uint32_t)(words + (la + 7) / 8 + 1);
        q.la = (int32_t)la; q.lb = (int32_t)lb; q.mode = mode[i]; q.maxd = maxd[i];
        word_off[(size_t)i] = (uint32_t)words;
        words += need;
        lp.push_back(q);
        lane_index.push_back((int32_t)i);
      } else {
        MyersPair q;
        oa.push_back(blob.size()); blob.append(seq_a[i], la);
        ob.push_back(blob.size()); blob.append(seq_b[i], lb);
        q.a = nullptr; q.b = nullptr;
        q.la = (int32_t)la; q.lb = (int32_t)lb; q.mode = mode[i]; q.maxd = maxd[i];
        pairs.push_back(q);
        long_index.push_back((int32_t)i);
        const int nb = (int)((la + 63) / 64);
        if (nb > max_blk) max_blk = nb;
      }
    }
    codes.assign((size_t)words, 0u);
  }
  {
    const size_t nlp = lp.size();
    parallel([&](int t) {
      for (size_t k = nlp * (size_t)t / (size_t)T, hi = nlp * (size_t)(t + 1) / (size_t)T; k < hi; k++) {
        const MyersLanePair& q = lp[k];
        const int64_t i = lane_index[k];
        for (int side = 0; side < 2; side++) {
          const unsigned char* s2 = reinterpret_cast<const unsigned char*>(side ? seq_b[i] : seq_a[i]);
          const size_t len = side ? (size_t)q.lb : (size_t)q.la;
          uint32_t* dst = codes.data() + (side ? q.b_off : q.a_off);
          for (size_t w = 0; w < (len + 7) / 8; w++) {
            uint32_t v = 0;
            const size_t lim = std::min<size_t>(8, len - w * 8);
            for (size_t c = 0; c < lim; c++) v |= (uint32_t)bits.t[s2[w * 8 + c]] << (4 * c);
            dst[w] = v;
          }
        }
      }
    });
  }
  char* d_blob = nullptr;
  MyersPair* d_pairs = nullptr;
  MyersLanePair* d_lp = nullptr;
  uint32_t *d_out = nullptr, *d_out_long = nullptr, *d_codes = nullptr;
  int32_t* d_index = nullptr;
  PoolScope guard(ctx);
  guard.watch((void**)&d_blob); guard.watch((void**)&d_pairs); guard.watch((void**)&d_out); guard.watch((void**)&d_lp); guard.watch((void**)&d_codes);
  guard.watch((void**)&d_index); guard.watch((void**)&d_out_long);
  const size_t nl = lp.size(), ng = pairs.size();
  if (pool_alloc(ctx, &d_out, (size_t)n)) return MIA_HIP_ERR_NOMEM;
  hipError_t e = hipSuccess;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  if (nl) {
    if (pool_alloc(ctx, &d_lp, nl) || pool_alloc(ctx, &d_codes, codes.size() + 8) || pool_alloc(ctx, &d_index, nl)) return MIA_HIP_ERR_NOMEM;
    e = hipMemcpyAsync(d_lp, lp.data(), nl * sizeof(MyersLanePair), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_codes, codes.data(), codes.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_index, lane_index.data(), nl * 4, hipMemcpyHostToDevice, ctx->stream);
  }
  if (ng && e == hipSuccess) {
    if (pool_alloc(ctx, &d_blob, blob.size() + 1) || pool_alloc(ctx, &d_pairs, ng) || pool_alloc(ctx, &d_out_long, ng)) return MIA_HIP_ERR_NOMEM;
    for (size_t i = 0; i < ng; i++) { pairs[i].a = d_blob + oa[i]; pairs[i].b = d_blob + ob[i]; }
    e = hipMemcpyAsync(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pairs, pairs.data(), ng * sizeof(MyersPair), hipMemcpyHostToDevice, ctx->stream);
  }
  if (e0) (void)hipEventRecord(e0, ctx->stream);
  if (nl && e == hipSuccess) {
    hipLaunchKernelGGL(k_myers_lanes, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, ctx->stream, (const MyersLanePair*)d_lp, (const uint32_t*)d_codes, (int32_t)nl,
                       (const int32_t*)d_index, d_out);
    e = hipGetLastError();
  }
  if (ng && e == hipSuccess) {
    const int lds = 16 * max_blk * 8;
    e = hipFuncSetAttribute((const void*)k_myers, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) {
      const int grid = (int)(ng < 8192 ? ng : 8192);
      hipLaunchKernelGGL(k_myers, dim3(grid), dim3(64), lds, ctx->stream, d_pairs, (int32_t)ng, d_out_long);
      e = hipGetLastError();
    }
  }
  if (e1) (void)hipEventRecord(e1, ctx->stream);
  std::vector<uint32_t> out_long(ng);
  if (e == hipSuccess && nl) e = hipMemcpyAsync(dist, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);      // (the long pairs' entries are filled in below)
  if (e == hipSuccess && ng) e = hipMemcpyAsync(out_long.data(), d_out_long, ng * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess && e0 && e1) { float ms = 0; if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ctx->myers_kernel_ms = ms; }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (e != hipSuccess) { ctx->err = std::string("myers: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
  for (size_t i = 0; i < ng; i++) dist[long_index[i]] = out_long[i];
  return MIA_HIP_OK;
}

extern "C" int mia_hip_myers_time(mia_hip_ctx* ctx, double* kernel_ms) {
  if (!ctx || !kernel_ms) return MIA_HIP_ERR_ARG;
  *kernel_ms = ctx->myers_kernel_ms;
  return MIA_HIP_OK;
}

// The alignment behind a distance the device has already established: Myers' furthest-reaching D-paths (1986) for
// D = 0 .. dist, then the walk back with the reference's preferences (mismatch, then a seq_b-only column, then a
// seq_a-only column, else one step down the snake; src/myers_align.c:47-83).  Host code: the table has (dist+1)^2
// cells and is walked once; the O(len * dist / 64) search for `dist` itself ran on the GPU.
static bool ond_backtrace(const char* a, int la, const char* b, int lb, int mode, int dist, std::string* row_a, std::string* row_b) {
  const int NONE = INT32_MIN / 2;
  auto bits = [](char c) { return (int)iupac_bits(c); };
  std::vector<int> v((size_t)(dist + 1) * (size_t)(dist + 1), NONE);   // row d starts at d*d, diagonal k at index k+d
  auto at = [&](int d, int k) -> int { return (k < -d || k > d) ? NONE : v[(size_t)d * d + (size_t)(k + d)]; };
  int end_k = 0;
  bool found = false;
  for (int d = 0; d <= dist && !found; d++) {
    const int klo = -d > -la ? -d : -la, khi = d < lb ? d : lb;
    for (int k = klo; k <= khi; k++) {
      int x = 0;
      if (d > 0) {
        const int keep = at(d - 1, k), from_left = at(d - 1, k - 1), from_right = at(d - 1, k + 1);
        x = keep == NONE ? NONE : keep + 1;
        if (from_left != NONE && from_left + 1 > x) x = from_left + 1;
        if (from_right != NONE && from_right > x) x = from_right;
        if (x == NONE) continue;
      }
      int y = x - k;
      while (x < lb && y < la && x >= 0 && y >= 0 && (bits(b[x]) & bits(a[y]))) { x++; y++; }
      v[(size_t)d * d + (size_t)(k + d)] = x;
      if ((mode == 1 || y == la) && (mode == 2 || x == lb)) {
        if (d != dist) return false;
        end_k = k; found = true;
        break;
      }
    }
  }
  if (!found) return false;
  std::string ra, rb;   // built back to front
  // In the prefix modes a D-path may run past the end of the sequence that need not be consumed (the recurrence has no
  // bound there, src/myers_align.c:26-32): the reference then copies that sequence's terminator into the row, which
  // ends the C string early.  Same here, without reading past the terminator.
  auto ca = [&](int y) { return y < la ? a[y] : '\0'; };
  auto cb = [&](int x) { return x < lb ? b[x] : '\0'; };
  int k = end_k, x = at(dist, k), y = x - k;
  for (int d = dist; d != 0;) {
    if (k != -d && k != d && x == at(d - 1, k) + 1) { d--; x--; y--; rb.push_back(cb(x)); ra.push_back(ca(y)); }
    else if (k > -d + 1 && x == at(d - 1, k - 1) + 1) { x--; k--; d--; rb.push_back(cb(x)); ra.push_back('-'); }
    else if (k < d - 1 && x == at(d - 1, k + 1)) { k++; y--; d--; rb.push_back('-'); ra.push_back(ca(y)); }
    else { x--; y--; rb.push_back(cb(x)); ra.push_back(ca(y)); }
    if (x < 0 || y < 0) return false;
  }
  while (x > 0) { x--; rb.push_back(cb(x)); ra.push_back(ca(x)); }
  row_a->assign(ra.rbegin(), ra.rend());
  row_b->assign(rb.rbegin(), rb.rend());
  return true;
}

extern "C" int mia_hip_myers_align(mia_hip_ctx* ctx, const char* seq_a, int32_t mode, const char* seq_b, int32_t maxd, uint32_t* dist,
                                   char* bt_a, char* bt_b) {
  if (!ctx || !seq_a || !seq_b || !dist || mode < 0 || mode > 2) return MIA_HIP_ERR_ARG;
  const int rc = mia_hip_myers(ctx, 1, &seq_a, &seq_b, &mode, &maxd, dist);
  if (rc != MIA_HIP_OK || *dist == 0xFFFFFFFFu || (!bt_a && !bt_b)) return rc;
  std::string ra, rb;
  if (!ond_backtrace(seq_a, (int)strlen(seq_a), seq_b, (int)strlen(seq_b), mode, (int)*dist, &ra, &rb)) {
    ctx->err = "myers_align: the D-path table does not end at the distance the device computed";
    return MIA_HIP_ERR_DEVICE;
  }
  if (bt_a) memcpy(bt_a, ra.c_str(), ra.size() + 1);
  if (bt_b) memcpy(bt_b, rb.c_str(), rb.size() + 1);
  return MIA_HIP_OK;
}

// ---- measured ceilings (bench.py's roofline) -----------------------------------------------
extern "C" int mia_hip_measure_peaks(mia_hip_ctx* ctx, int64_t copy_bytes, double* hbm_copy_gbs, double* valu_ginst_s) {
  if (!ctx || copy_bytes < (1 << 20)) return MIA_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  void *a = nullptr, *b = nullptr;
  ScopeFree sf; sf.watch(&a); sf.watch(&b);
  int rc = MIA_HIP_OK;
  if (hbm_copy_gbs) {
    const int64_t n16 = copy_bytes / 16;
    if (hipMalloc(&a, (size_t)n16 * 16) != hipSuccess || hipMalloc(&b, (size_t)n16 * 16) != hipSuccess) { ctx->err = "measure_peaks: hipMalloc"; rc = MIA_HIP_ERR_NOMEM; }
    else {
      HIPCHK(hipMemsetAsync(a, 1, (size_t)n16 * 16, ctx->stream));
      float best = 1e30f;
      for (int variant = 0; variant < 8; variant++) {
        const int grid = ctx->cus * (variant & 1 ? 32 : 8);
        for (int rep = 0; rep < 4; rep++) {                   // the first repetition warms the TLBs
          (void)hipEventRecord(e0, ctx->stream);
          switch (variant >> 1) {
            case 0: hipLaunchKernelGGL((k_peak_copy<1, false>), dim3(grid * 2), dim3(256), 0, ctx->stream, (const uint4*)a, (uint4*)b, n16); break;
            case 1: hipLaunchKernelGGL((k_peak_copy<4, false>), dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)a, (uint4*)b, n16); break;
            case 2: hipLaunchKernelGGL((k_peak_copy<4, true>), dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)a, (uint4*)b, n16); break;
            default: hipLaunchKernelGGL((k_peak_copy<8, true>), dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)a, (uint4*)b, n16); break;
          }
          (void)hipEventRecord(e1, ctx->stream);
          HIPCHK(hipEventSynchronize(e1));
          float ms = 0;
          HIPCHK(hipEventElapsedTime(&ms, e0, e1));
          if (getenv("MIA_HIP_PEAK_DEBUG")) fprintf(stderr, "[peak copy] variant %d rep %d: %.1f GB/s\n", variant, rep, 2.0 * (double)n16 * 16 / (ms * 1e-3) / 1e9);
          if (rep > 0 && ms < best) best = ms;
        }
      }
      *hbm_copy_gbs = 2.0 * (double)n16 * 16 / (best * 1e-3) / 1e9;      // bytes read + bytes written
    }
  }
  if (valu_ginst_s && rc == MIA_HIP_OK) {
    int32_t* out = nullptr;
    const int wgs = ctx->cus * 8, iters = 4096;               // 8 waves per SIMD: enough to cover the issue latency
    if (hipMalloc((void**)&out, (size_t)wgs * 256 * 4) != hipSuccess) { ctx->err = "measure_peaks: hipMalloc"; rc = MIA_HIP_ERR_NOMEM; }
    else {
      float best = 1e30f;
      for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, ctx->stream);
        hipLaunchKernelGGL(k_peak_valu, dim3(wgs), dim3(256), 0, ctx->stream, out, iters, 12345 + rep);
        (void)hipEventRecord(e1, ctx->stream);
        if (hipEventSynchronize(e1) != hipSuccess) break;
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && rep > 0 && ms < best) best = ms;
      }
      (void)hipFree(out);
      *valu_ginst_s = (double)wgs * 4 * iters * PEAK_VALU_OPS_PER_ITER / (best * 1e-3) / 1e9;   // wave instructions per second
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

extern "C" int mia_hip_pass1_time(mia_hip_ctx* ctx, double* kernel_ms) {
  if (!ctx || !kernel_ms) return MIA_HIP_ERR_ARG;
  *kernel_ms = ctx->pass1_ms;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_pass1_filtered(mia_hip_ctx* ctx, int64_t* reads) {
  if (!ctx || !reads) return MIA_HIP_ERR_ARG;
  *reads = ctx->pass1_filtered;
  return MIA_HIP_OK;
}

extern "C" int mia_hip_pass1_anchored(mia_hip_ctx* ctx, int64_t* reads) {
  if (!ctx || !reads) return MIA_HIP_ERR_ARG;
  *reads = ctx->pass1_anchored;
  return MIA_HIP_OK;
}
