// mia_hip.hip -- libmia_hip.so: C ABI (include/mia_hip.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <atomic>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
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
constexpr int OPEN_WGS = 1024;                                   // k_align_open's persistent grid (one read per wavefront; a steady-state list holds a few hundred)
constexpr int64_t OPEN_SLAB_BYTES = (int64_t)MAX_READ * 64 * 12;   // its trace slab: the widest window class's
constexpr int CTRL_BINS = 0;                                   // [count N_BINS][off N_BINS][cursor N_BINS][wide_count][retry_count]
constexpr int CTRL_HDR = (3 * N_BINS + 2 + 1) & ~1;            // PH_* (8-byte aligned: the DP kernels fetch their range as a pair)
constexpr int CTRL_FILTER = CTRL_HDR + PH_WORDS;               // 4 words (k_diag_filter / k_band_align)
constexpr int CTRL_LKN = CTRL_FILTER + 4, CTRL_CULLF = CTRL_LKN + 1, CTRL_NEV = CTRL_CULLF + 1, CTRL_TFLAGS = CTRL_NEV + 1, CTRL_FIXN = CTRL_TFLAGS + 1;   // FIXN: reads on the early tally's fix list
constexpr int CTRL_GENN = CTRL_FIXN + 1;                           // reads k_tally_binned left to k_tally_general
constexpr int CTRL_BXC = (CTRL_GENN + 1 + 63) & ~63;           // BXC_* counters, a cache line each
constexpr int CTRL_WORDS = CTRL_BXC + BXC_WORDS;
constexpr int CTRL_C0 = CTRL_BINS + 3 * N_BINS, CTRL_CN = CTRL_WORDS - CTRL_C0;    // what the host looks at behind an alignment: wide / retry counts, planner header, filter and band counters

// Environment switches.  The release library reads FIVE (documented in DESIGN.md 1): MIA_HIP_SPIN_WAIT and MIA_HIP_LOOPBACK_TIMEOUT
// here, MIA_HIP_THREADS, MIA_HIP_TIMING and MIA_DATA_PATH in the host programs.  Everything else -- the "off" sides of the
// differential tests, the round-1 and one-lane routes, the profiling switches that make results wrong on purpose
// (MIA_HIP_DEBUG_SKIP, MIA_HIP_BX_DEBUG) -- exists only in libmia_hip_alt.so, built with -DMIA_HIP_ALT_PATHS; the Python binding
// loads that build for a context made while such a variable is set (tests, tools), and nothing else does.
#ifdef MIA_HIP_ALT_PATHS
static inline const char* alt_env(const char* name) { return getenv(name); }
#define ALT_KERNEL(k) ((const void*)(k))
#else
static inline const char* alt_env(const char*) { return nullptr; }
#define ALT_KERNEL(k) ((const void*)nullptr)        // (the one-lane band kernels are not part of the release build's code object)
#endif

struct mia_hip_ctx {
  int device = 0;
  int32_t* d_ctrl = nullptr;
  uint32_t stage_mask = ~0u;               // timed stages (mia_hip_set_stage_mask): an event pair costs the stream a few microseconds
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // the trace DP of the plan's own lists runs beside the values DP
  hipStream_t stream3 = nullptr; hipEvent_t ev_join3 = nullptr;                      // ... and both beside the planner and the full-window kernels of the reads the plan gave up on
  int32_t* d_retry2 = nullptr; int64_t retry2_cap = 0;                              // reads no band kernel could finish
  uint32_t* d_bx_slabs_late = nullptr; int64_t bx_slab_late_cap = 0; int bx_late_wgs = 0;
  uint32_t* d_bx_slabs_late2 = nullptr; bool dp_aside = false;      // the second late trace's slabs; this alignment's values DP and late trace are on stream2 (align_all: fork_at_quick)
  // mia_hip_iterate without a host wait behind the alignment: cull, tally and consensus are queued at once and every one of
  // their kernels returns at its first instruction if *abort_if != 0 (reads are waiting for the exact one-read-per-thread
  // kernel: the host sees that with the consensus, runs it, and queues the chain again)
  const int32_t* abort_if = nullptr;
  bool spec_ok = false, spec_pending = false, spec_filtered = false, spec_bx = false, spec_plain = false;
  int64_t spec_redone = 0;                  // iterations whose cull / tally / consensus were queued twice (reads for the exact kernel)
  bool pend_encode = false; int32_t pend_L = 0, pend_wl = 0, pend_total = 0;      // mia_hip_iterate: d_ascii holds the new reference, d_ref not yet
  uint32_t* d_prep_bar = nullptr; uint32_t prep_bar_count = 0; bool no_prep_fuse = false;   // k_ref_prep's grid barrier (MIA_HIP_NO_PREP_FUSE=1: six launches)
  // round 6: the band plan lists the reads it leaves open itself and k_align_open takes them one per wavefront (align_all: direct_open;
  // MIA_HIP_NO_DIRECT_OPEN=1, alt build: the planner's count / scan / fill and the quad kernels, as in every iteration with many open reads)
  KbPair* d_kbits = nullptr;            // the quick plan's bitmaps over all 4^10 10-mers (bandx_body.h: KmerBits), remade with the table
  int use_quick = 1; int64_t quick_steps = 0;      // the quick plan in front of k_bx_plan's launches (MIA_HIP_NO_QUICK_PLAN=1, alt build: the full plan for every read)
  bool direct_open_now = false;        // this alignment's open reads are on d_open_list (set by align_all, read by bx_join_and_retry)
  int32_t* d_open_list = nullptr; int64_t open_cap = 0; unsigned char* d_slabs_open = nullptr; bool use_direct_open = true; int64_t direct_open_steps = 0;   // the plan's own open list (n entries), k_align_open's trace slabs (one per workgroup of its grid)
  bool spec_force = false; int32_t* d_one = nullptr;      // MIA_HIP_SPEC_TEST=1 (tests): a word that holds 1
  bool zero_copy = true;        // mia_hip_iterate: the last kernel writes consensus and counters into pinned host memory itself (MIA_HIP_NO_ZERO_COPY=1: two copies)
  // the launches other streams wait for signal their events themselves (launch_k) instead of a marker behind them; MIA_HIP_NO_EXT_EVENTS=1: markers
  bool cull_scan = false, tail_scans = false, stage_markers = false;
  bool spin_wait = true;        // mia_hip_iterate's one wait asks (hipStreamQuery) instead of sleeping on an interrupt; MIA_HIP_SPIN_WAIT=0: hipStreamSynchronize
  uint32_t ext_events = 31u; bool planner_end_signalled = false, align_end_signalled = false;
  BxCandRec* d_bx_cand2 = nullptr; int use_fine = 1;      // the third launch's list (reads for the fine blocks); MIA_HIP_NO_FINE=1: none; MIA_HIP_FINE=2: in every iteration
  BxCandRec* d_bx_cand = nullptr; int64_t cand_cap = 0, cand2_cap = 0; bool plan_split = true;      // k_bx_plan's hand-over list between its two launches (MIA_HIP_NO_PLAN_SPLIT=1: one launch)
  bool no_spec = false;                     // MIA_HIP_NO_SPEC=1: wait for the alignment's counters before the cull is queued
  bool no_side_buckets = false;             // MIA_HIP_NO_SIDE_BUCKETS=1
  int buckets_queued = 0;                   // the tally's counting sort is already queued: 1 on the context's stream, 2 on stream2 (ev_join behind it)
  // the early tally (mia_consensus_kernels.h, k_rec_early): the reads the plan finishes are tallied on stream4 beside the band DPs
  hipStream_t stream4 = nullptr; hipEvent_t ev_early = nullptr;
  // the band DPs in two rounds (align_all: split_dp): the first beside the plan's second and third launch, on the lists its first launch made
  hipEvent_t ev_snap = nullptr, ev_v1 = nullptr; uint32_t* d_bx_snap = nullptr; int split_dp_mode = 0;      // mode: 0 auto, 1 always (MIA_HIP_SPLIT_DP=1), -1 never (MIA_HIP_SPLIT_DP=0)
  bool cull_with_records = true;      // k_cull_records writes the tally records of an iteration without links itself (MIA_HIP_NO_CULL_RECORDS=1: k_rec_params always)
  int64_t split_dp_steps = 0;
  // MIA_HIP_EARLY_TALLY=1 (alt build only; an experiment that measured SLOWER): the plan's reads tallied beside the band DPs
  bool use_early = false, early_queued = false;
  int early_wgs_per_cu = 2;
  uint8_t* d_early = nullptr; int32_t* d_trec_early = nullptr; int32_t* d_order_e = nullptr; int32_t* d_fix_list = nullptr; int64_t early_cap = 0;
  int32_t* d_bucket_e = nullptr; int bucket_e_cap = 0, bucket_e_clean_nb = -1;
  int32_t* d_tally_slabs_e = nullptr; int64_t tally_slab_e_cap = 0;
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
  bool ref_few_n = true;                   // ... fewer than 0.2 %: four windows in five hold none (the quick plan asks those even where the table spells out N columns)
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
  double myers_kernel_ms = 0; bool myers_no_lanes = false, myers_no_ond = false, fake_event_overflow = false;   // the kernels of the last mia_hip_myers call (HIP events); MIA_HIP_MYERS_NO_LANES=1: every pair through k_myers
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
  bool planner_beside = false;            // MIA_HIP_PLANNER_BESIDE=1 (alt build): the planner's head kernels beside the band DPs in every iteration
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
  int32_t* d_gen_list = nullptr; int64_t gen_cap = 0;             // the reads k_tally_binned leaves to k_tally_reduce's extra workgroups (a thousand in a million; room for all)
  // MIA_HIP_STRAND_SPLIT=1 (alt build only): position-specific matrices, the tally's buckets by column AND strand, the rows of depth code 15
  // through the vertical counters.  Measured: configs[2] 1.313 -> 1.306 ms, configs[4] at 5 M reads 9.91 -> 10.2 ms (twice the part-filled
  // workgroups and slabs; the end rows' packed atomics are what the kernel waits for either way) -- off
  bool tally_strand_split = true;          // position-specific matrices: a workgroup's reads all of one strand, sorted by start (round 5; MIA_HIP_STRAND_SPLIT=0: round 4's tally)
  bool tally_rall = true;                  // reads of 129 .. 256 bases: every row through the runs (MIA_HIP_NO_TALLY_RALL=1: round 4's tally for them)
  bool tally_runs = true;                  // ... and the rows at either end of a read reduced over runs of equal starts (MIA_HIP_NO_TALLY_RUNS=1: per read)
  int32_t* d_order2 = nullptr; uint8_t* d_okey = nullptr; int32_t* d_sort2 = nullptr; int64_t sort2_cap = 0;      // the reads of every bucket by start; histogram + cursors of that sort
  int tally_chunk_linear = TALLY_CHUNK_LINEAR;                    // MIA_HIP_TALLY_CHUNK=256|512|768 (alt build)
  bool tally_defer = true;                                        // MIA_HIP_TALLY_INLINE=1 (alt build): they are taken inside k_tally_binned, one per wavefront
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
  char* d_ascii = nullptr; int64_t ascii_cap = 0; const char* ascii_src = nullptr;     // ascii_src: where the pending reference is read from (d_ascii, or the pinned staging area)
  char* d_cons = nullptr; int64_t cons_cap = 0;              // result of an iteration: [CH_WORDS header][consensus string]
  int32_t* d_cons_pos = nullptr; int64_t cons_pos_cap = 0;
  unsigned char* h_pin2 = nullptr; size_t pin2_bytes = 0;   // results of an iteration (header + consensus string)
  int64_t iter_fallbacks = 0;
  // sharded runs: one context per GPU; the exchanges go through a table of collectives (RCCL over xGMI from
  // mia_hip_comm_init, or whatever mia_hip_comm_attach was given), on the context's own stream
  mia_hip_collectives coll{}; bool comm = false; int comm_ranks = 1, comm_rank = 0; std::string coll_name;
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
  // the events order kernels of this context's streams on this device and nothing else (MIA_HIP_EVENT_DEVICE_SCOPE=1: say so)
  unsigned evf = hipEventDisableTiming;
  if (const char* es = alt_env("MIA_HIP_EVENT_DEVICE_SCOPE")) if (atoi(es)) evf |= hipEventReleaseToDevice;
  if (hipSetDevice(device_index) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join3, evf) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->stream4, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_early, evf) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_snap, evf) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_v1, evf) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, evf) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join, evf) != hipSuccess) {
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
    const char* nbt = alt_env("MIA_HIP_NO_BINNED_TALLY");
    if (nbt && atoi(nbt)) ctx->use_binned_tally = 0;
    const char* nband = alt_env("MIA_HIP_NO_BAND");
    if (nband && atoi(nband)) ctx->use_band = 0;
    const char* npl = alt_env("MIA_HIP_NO_PLAIN");
    if (npl && atoi(npl)) ctx->use_plain = 0;
    const char* pbb = alt_env("MIA_HIP_PLAIN_BEHIND_BAND");
    if (pbb && atoi(pbb)) ctx->plain_behind_band = 1;
    const char* nf = alt_env("MIA_HIP_NO_DIAG_FILTER");
    if (nf && atoi(nf)) { ctx->use_filter = 0; ctx->use_bx = 0; }      // every shortcut off: the full-window DP kernels only
    const char* nbd = alt_env("MIA_HIP_NO_BAND_DP");
    if (nbd && atoi(nbd)) { ctx->use_banddp = 0; ctx->use_bx = 0; }
    const char* nbx = alt_env("MIA_HIP_NO_BANDX");
    if (nbx && atoi(nbx)) ctx->use_bx = 0;
    if (const char* nl2 = alt_env("MIA_HIP_NO_LANES")) if (atoi(nl2)) ctx->use_lanes = 0;
    if (const char* bd2 = alt_env("MIA_HIP_BX_DEBUG")) ctx->bx_dbg = (uint32_t)atoi(bd2);
    if (const char* bs2 = alt_env("MIA_HIP_BX_SERIAL")) ctx->bx_serial = atoi(bs2) != 0;
    if (const char* sb2 = alt_env("MIA_HIP_NO_SIDE_BUCKETS")) ctx->no_side_buckets = atoi(sb2) != 0;
    if (const char* ns2 = alt_env("MIA_HIP_NO_SPEC")) ctx->no_spec = atoi(ns2) != 0;
    if (const char* pf2 = alt_env("MIA_HIP_NO_PREP_FUSE")) ctx->no_prep_fuse = atoi(pf2) != 0;
    if (const char* ps2 = alt_env("MIA_HIP_NO_PLAN_SPLIT")) ctx->plan_split = atoi(ps2) == 0;
    if (const char* zc = alt_env("MIA_HIP_NO_ZERO_COPY")) ctx->zero_copy = atoi(zc) == 0;
    if (const char* ef = alt_env("MIA_HIP_NO_EXT_EVENTS")) ctx->ext_events = atoi(ef) ? 0u : 31u;
    if (const char* cs = alt_env("MIA_HIP_CULL_SCAN")) ctx->cull_scan = atoi(cs) != 0;
    if (const char* sm = alt_env("MIA_HIP_STAGE_MARKERS")) ctx->stage_markers = atoi(sm) != 0;
    if (const char* ts = alt_env("MIA_HIP_TAIL_SCANS")) ctx->tail_scans = atoi(ts) != 0;
    if (const char* dop = alt_env("MIA_HIP_NO_DIRECT_OPEN")) ctx->use_direct_open = atoi(dop) == 0;
    if (const char* qp = alt_env("MIA_HIP_NO_QUICK_PLAN")) ctx->use_quick = atoi(qp) == 0 ? 1 : 0;
    if (const char* qp = alt_env("MIA_HIP_QUICK_PLAN")) ctx->use_quick = atoi(qp);      // (2: also where the size rule below says no)
    if (const char* sw = getenv("MIA_HIP_SPIN_WAIT")) ctx->spin_wait = atoi(sw) != 0;
    if (const char* em = alt_env("MIA_HIP_EXT_EVENTS_MASK")) ctx->ext_events = (uint32_t)atoi(em);
    if (const char* st2 = alt_env("MIA_HIP_SPEC_TEST")) ctx->spec_force = atoi(st2) != 0;
    if (const char* ml = alt_env("MIA_HIP_MYERS_NO_LANES")) ctx->myers_no_lanes = atoi(ml) != 0;
    if (const char* mo = alt_env("MIA_HIP_MYERS_NO_OND")) ctx->myers_no_ond = atoi(mo) != 0;
    if (const char* fo = alt_env("MIA_HIP_FAKE_EVENT_OVERFLOW")) ctx->fake_event_overflow = atoi(fo) != 0;
    if (const char* na = alt_env("MIA_HIP_NO_AUTO_PLAIN")) ctx->no_auto_plain = atoi(na) != 0;
    const char* egs = alt_env("MIA_HIP_EAGER_SCRIPTS");
    if (egs && atoi(egs)) ctx->lazy_scripts = 0;
    const char* nwl = alt_env("MIA_HIP_NO_WILD");
    if (nwl && atoi(nwl)) ctx->use_wild = 0;
    if (const char* ne = alt_env("MIA_HIP_EARLY_TALLY")) ctx->use_early = atoi(ne) != 0;
    if (const char* ti = alt_env("MIA_HIP_TALLY_INLINE")) ctx->tally_defer = atoi(ti) == 0;
    if (const char* ss = alt_env("MIA_HIP_STRAND_SPLIT")) ctx->tally_strand_split = atoi(ss) != 0;
    if (const char* nr = alt_env("MIA_HIP_NO_TALLY_RUNS")) ctx->tally_runs = atoi(nr) == 0;
    if (const char* sd = alt_env("MIA_HIP_SPLIT_DP")) ctx->split_dp_mode = atoi(sd) != 0 ? 1 : -1;
    if (const char* ra = alt_env("MIA_HIP_NO_TALLY_RALL")) ctx->tally_rall = atoi(ra) == 0;
    if (const char* cr = alt_env("MIA_HIP_NO_CULL_RECORDS")) ctx->cull_with_records = atoi(cr) == 0;
    if (const char* tc = alt_env("MIA_HIP_TALLY_CHUNK")) { const int c = atoi(tc); if (c == 256 || c == 512 || c == 768) ctx->tally_chunk_linear = c; }
    if (const char* ew = alt_env("MIA_HIP_EARLY_WGS")) ctx->early_wgs_per_cu = atoi(ew);
    if (const char* nf = alt_env("MIA_HIP_NO_FINE")) ctx->use_fine = atoi(nf) == 0 ? 1 : 0;
    if (const char* nf = alt_env("MIA_HIP_FINE")) ctx->use_fine = atoi(nf);
    if (const char* pb = alt_env("MIA_HIP_PLANNER_BESIDE")) ctx->planner_beside = atoi(pb) != 0;
    const char* bxf = alt_env("MIA_HIP_BX_FILTER");
    if (bxf && atoi(bxf)) ctx->bx_filter_first = 1;
    const char* nq = alt_env("MIA_HIP_NO_QUAD");
    if (nq && atoi(nq)) ctx->use_quad = 0;
    const char* qw = alt_env("MIA_HIP_QUAD_WAVES_PER_CU");
    if (qw && atoi(qw) > 0) ctx->quad_wgs = prop.multiProcessorCount * atoi(qw);
    const char* dbg = alt_env("MIA_HIP_DEBUG_SKIP");
    if (dbg) ctx->dbg = (uint32_t)atoi(dbg);
    const char* g = alt_env("MIA_HIP_GRID_WAVES_PER_CU");
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
  if (ctx->h_pin) memset(ctx->h_pin, 0, mia_hip_ctx::PIN_MISC);
  else ctx->no_prep_fuse = true;            // (k_ref_prep reports a barrier it gave up on through a pinned word: without one, six plain launches)
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
                  ctx->d_slabs[0], ctx->d_slabs[1], ctx->d_slabs[2], ctx->d_quad_slabs, ctx->d_bucket, ctx->d_order, ctx->d_order2, ctx->d_okey, ctx->d_sort2,
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
  if (ctx->stream4) (void)hipStreamDestroy(ctx->stream4);
  if (ctx->ev_early) (void)hipEventDestroy(ctx->ev_early);
  if (ctx->ev_snap) (void)hipEventDestroy(ctx->ev_snap);
  if (ctx->ev_v1) (void)hipEventDestroy(ctx->ev_v1);
  if (ctx->d_bx_snap) (void)hipFree(ctx->d_bx_snap);
  if (ctx->d_gen_list) (void)hipFree(ctx->d_gen_list);
  for (void* p : {(void*)ctx->d_early, (void*)ctx->d_trec_early, (void*)ctx->d_order_e, (void*)ctx->d_fix_list, (void*)ctx->d_bucket_e, (void*)ctx->d_tally_slabs_e}) if (p) (void)hipFree(p);
  if (ctx->ev_join3) (void)hipEventDestroy(ctx->ev_join3);
  if (ctx->d_retry2) (void)hipFree(ctx->d_retry2);
  if (ctx->d_cull_sync) (void)hipFree(ctx->d_cull_sync);
  if (ctx->d_bx_slabs_late) (void)hipFree(ctx->d_bx_slabs_late);
  if (ctx->d_bx_slabs_late2) (void)hipFree(ctx->d_bx_slabs_late2);
  if (ctx->d_one) (void)hipFree(ctx->d_one);
  if (ctx->d_bx_cand) (void)hipFree(ctx->d_bx_cand);
  if (ctx->d_bx_cand2) (void)hipFree(ctx->d_bx_cand2);
  for (int k = 0; k < 3; k++) if (ctx->d_slabs_retry[k]) (void)hipFree(ctx->d_slabs_retry[k]);
  if (ctx->d_prep_bar) (void)hipFree(ctx->d_prep_bar);
  if (ctx->d_slabs_open) (void)hipFree(ctx->d_slabs_open);
  if (ctx->d_open_list) (void)hipFree(ctx->d_open_list);
  if (ctx->d_kbits) (void)hipFree(ctx->d_kbits);
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
    ctx->tally_pk_bias = (-lo + hi <= 2047 && !alt_env("MIA_HIP_NO_PACKED_TALLY")) ? -lo : -1;
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
  if (const char* nl = alt_env("MIA_HIP_NO_LINEAR_TALLY")) if (atoi(nl)) ctx->tally_linear = false;
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
  rcx |= dev_alloc(ctx, &ctx->d_rplanes, (size_t)n * 2 * ctx->rplane_words + 2);      // (+ 2: k_tally_binned asks for a second word of every plane)
  rcx |= dev_alloc(ctx, &ctx->d_list, (size_t)n + 4 * N_BINS);   // quad bins are padded to multiples of four
  rcx |= dev_alloc(ctx, &ctx->d_wide_list, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_retry_list, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_slot, (size_t)n);
  rcx |= dev_alloc(ctx, &ctx->d_partial, (size_t)(n / 256 + n / 4096 + 8));      // per 256 reads, and behind them per 4 096 (k_slot_count)
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
  if (ctx->d_order2) { (void)hipFree(ctx->d_order2); ctx->d_order2 = nullptr; }
  if (ctx->d_okey) { (void)hipFree(ctx->d_okey); ctx->d_okey = nullptr; }
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

// A launch that signals `stop` with its own completion (hipExtLaunchKernelGGL), or a plain launch if stop is null.  An event
// recorded BEHIND a launch is a marker packet of its own: the next kernel of that stream starts 15-20 us later (measured
// between the plan and the values DP), and a stream waiting for the event sees it as late.
template <typename... KArgs, typename... Args>
static void launch_k(void (*kernel)(KArgs...), dim3 g, dim3 b, size_t shm, hipStream_t s, hipEvent_t stop, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "every kernel argument, defaults included");
  if (stop) hipExtLaunchKernelGGL(kernel, g, b, (uint32_t)shm, s, nullptr, stop, 0, static_cast<KArgs>(args)...);
  else hipLaunchKernelGGL(kernel, g, b, shm, s, static_cast<KArgs>(args)...);
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

// One launch of a timed stage whose kernel carries no event for another stream: the event pair rides on the launch itself
// (start and end of that dispatch, no marker packets in front of and behind the kernel -- on the critical path those cost
// the step 10-20 us each, and the timed region of bench.py times the dominant stage in every step); MIA_HIP_STAGE_MARKERS=1
// or an untimed stage: the launch as it is, between stage_begin / stage_end.
template <typename... KArgs, typename... Args>
static int stage_launch(mia_hip_ctx* ctx, Stage st, void (*kernel)(KArgs...), dim3 g, dim3 b, size_t shm, hipStream_t s, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "every kernel argument, defaults included");
  if (!((ctx->stage_mask >> st) & 1u) || ctx->stage_markers) {
    if (stage_begin(ctx, st, s)) return -1;
    hipLaunchKernelGGL(kernel, g, b, shm, s, static_cast<KArgs>(args)...);
    stage_end(ctx, st, s);
    return 0;
  }
  if (ctx->ev_free.empty()) {
    hipEvent_t x, y;
    if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return -1;
    ctx->ev_free.push_back({x, y});
  }
  auto p = ctx->ev_free.back();
  ctx->ev_free.pop_back();
  ctx->stg[st].pending.push_back(p);
  hipExtLaunchKernelGGL(kernel, g, b, (uint32_t)shm, s, p.first, p.second, 0, static_cast<KArgs>(args)...);
  return 0;
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

#ifdef MIA_HIP_ALT_PATHS
// (alt build only, not in include/mia_hip.h: the counts MIA_HIP_DEBUG_SKIP & 65536 collects in k_tally_binned, and their reset)
extern "C" int mia_hip_debug_tally_kinds(mia_hip_ctx* ctx, uint64_t* out8) {
  if (!ctx || !out8) return MIA_HIP_ERR_ARG;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tally_kinds), 8 * sizeof(uint64_t)));
  uint64_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tally_kinds), z, sizeof z));
  HIPCHK(hipMemcpyFromSymbol(out8 + 8, HIP_SYMBOL(g_tally_clk), 8 * sizeof(uint64_t)));      // (out8: 16 words)
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tally_clk), z, sizeof z));
  return MIA_HIP_OK;
}
// (alt build only: the clocks MIA_HIP_BX_DEBUG & 512 collects in k_bx_plan's first launch, and their reset)
extern "C" int mia_hip_debug_plan_clk(mia_hip_ctx* ctx, uint64_t* out12) {
  if (!ctx || !out12) return MIA_HIP_ERR_ARG;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  static uint64_t all[64 * 16], z[64 * 16];
  HIPCHK(hipMemcpyFromSymbol(all, HIP_SYMBOL(g_plan_clk), sizeof all));
  for (int k = 0; k < 12; k++) { out12[k] = 0; for (int st = 0; st < 64; st++) out12[k] += all[st * 16 + k]; }
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_plan_clk), z, sizeof z));
  return MIA_HIP_OK;
}
#endif
extern "C" int mia_hip_bx_counters(mia_hip_ctx* ctx, uint32_t* out32) {
  if (!ctx || !out32) return MIA_HIP_ERR_ARG;
  for (int k = 0; k < 32; k++) out32[k] = ctx->bx_last[k];
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
                                bool own_slabs = false, hipEvent_t stop = nullptr) {
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
  launch_k(k_align_window<CPL>, dim3(grid), dim3(64), 0, on ? on : ctx->stream, stop, ctx->rs, ref, ctx->d_pssm, ctx->packs.p[ci], list,
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
  if (!aside || ctx->dp_aside) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));      // (dp_aside: values DP and late trace on stream2)
  HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join3, 0));
  // (this launch is the last of the alignment on the context's stream: with ext_events it signals ev_fork, which
  // mia_hip_iterate's counting sort on stream2 waits for -- see queue_cull)
  hipEvent_t stop = (aside && (ctx->ext_events & 8u)) ? ctx->ev_fork : nullptr;
  ctx->align_end_signalled = stop != nullptr;
  hipError_t e;
  if (ctx->direct_open_now) {
    // the plan's own open list and the band DPs' retry list in ONE launch, a read of either per wavefront (k_align_open): the planner's
    // stream has had nothing to do in this alignment -- no kernel beside the DPs for the open reads, no wait for that stream here
    RefInfo ref{ctx->d_ref, ctx->L, ctx->wrap, ctx->explicit_win};
    if (stage_begin(ctx, STG_TRACE)) return MIA_HIP_ERR_NOMEM;
    launch_k(k_align_open, dim3(OPEN_WGS), dim3(64), 0, ctx->stream, stop, ctx->rs, ref, (const int32_t*)ctx->d_pssm, ctx->packs, (const int32_t*)ctx->d_open_list,
             (const uint32_t*)(ctx->d_bx_ctr + (size_t)BXC_OPEN * BXC_STRIDE), ctx->d_slabs_open, (int64_t)OPEN_SLAB_BYTES, ctx->d_wide_list, ctx->d_bins + 3 * N_BINS, ctx->dbg,
             (const int32_t*)ctx->d_retry2, (const uint32_t*)(range + 1));
    stage_end(ctx, STG_TRACE);
    e = hipGetLastError();
    if (e != hipSuccess) { ctx->err = std::string("k_align_open launch: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
    return MIA_HIP_OK;
  }
  e = cols <= 64 * 4 ? launch_window<4>(ctx, 0, ctx->d_retry2, 0, range, nullptr, aside, stop)
    : cols <= 64 * 8 ? launch_window<8>(ctx, 1, ctx->d_retry2, 0, range, nullptr, aside, stop)
                     : launch_window<12>(ctx, 2, ctx->d_retry2, 0, range, nullptr, aside, stop);
  if (e != hipSuccess) { ctx->err = std::string("k_align_window (band retry list) launch: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
  if (aside) {
    if (!ctx->planner_end_signalled) HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));      // (the planner's chain ends here)
    ctx->planner_end_signalled = false;
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  }
  return MIA_HIP_OK;
}

static int align_all(mia_hip_ctx* ctx);
static int early_tally_launch(mia_hip_ctx* ctx);
static bool tally_is_binned(const mia_hip_ctx* ctx);
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
  ctx->ref_few_n = n_other * 500 <= L;
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
  hipLaunchKernelGGL(k_ref_encode, dim3((unsigned)((ctx->pend_total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->ascii_src, ctx->pend_L, ctx->pend_wl,
                     ctx->d_ref, ctx->pend_total);
}

// the counters of a deferred alignment, once its control block has reached the host (hb: the words from CTRL_BINS + 3 N_BINS on)
static void align_counters_collect(mia_hip_ctx* ctx, const int32_t* hb, bool filtered, bool bx, bool plain) {
  constexpr int C0 = CTRL_C0;
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
  if (bx) ctx->bx_last[BXC_CUR_VALUES] = h_filter[1];      // (statistics: reads the quick plan left to the full plan -- the lanes kernels have no use for this slot)
  if (plain) ctx->plain_retried += h_hdr[PH_RETRIED_PLAIN];
}

// every strand_known read against its window of ctx->d_ref: plan, values-only pass, trace kernels, exact kernel
static int align_all(mia_hip_ctx* ctx) {
  const int64_t n = ctx->rs.n;
  const int wrap = ctx->wrap;
  ctx->bx_pending_join = false;
  ctx->direct_open_now = false;
  ctx->dp_aside = false;
  if (!ctx->spec_ok) { ctx->spec_pending = false; ctx->abort_if = nullptr; }
  ctx->bx_planner_aside = false;
  ctx->planner_end_signalled = false; ctx->align_end_signalled = false;
  ctx->buckets_queued = 0;                  // (a counting sort queued for an earlier alignment is void)
  ctx->early_queued = false;
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
  bool planner_head_first = false;      // count / scan / fill of the planner queued in front of the band DPs (see the band launches)
  bool direct_open = false;             // the plan lists its open reads itself, k_align_open takes them (no planner, no quad kernels: see below)
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
      // (the quick plan's bitmaps: made wherever the table is made, unless the reference is N all over -- hardly a window without one then)
      const bool want_bits = ctx->use_quick && (kh.wild == 0 || ctx->ref_few_n);
      if (want_bits && !ctx->d_kbits && dev_alloc(ctx, &ctx->d_kbits, (size_t)KB_WORDS)) return MIA_HIP_ERR_NOMEM;
      if (fused_prep) {
        RefPrep rp2;
        rp2.ascii = ctx->ascii_src; rp2.L = ctx->pend_L; rp2.wl = ctx->pend_wl; rp2.total = ctx->pend_total; rp2.codes = ctx->d_ref;
        rp2.ctrl = ctx->d_ctrl; rp2.ctrl_words = CTRL_WORDS;
        rp2.kslot = ctx->d_khash; rp2.kovf = ctx->d_khash_ovf; rp2.kslots = kslots; rp2.kmask = kh.mask; rp2.kshift = kh.shift; rp2.kwild = kh.wild;
        rp2.plane_words = words; rp2.plo = ctx->d_planes; rp2.phi = ctx->d_planes + ctx->plane_cap; rp2.pok = ctx->d_planes + 2 * ctx->plane_cap;
        rp2.nib_words = nw; rp2.nib = ctx->d_refnib;
        const int64_t want = ((int64_t)wrap + (kh.wild > 0 ? 63 : 255)) / (kh.wild > 0 ? 64 : 256);       // (N columns spelled out: 64 positions per workgroup, see kmer_hash_insert_block)
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, ctx->cus));      // one workgroup per compute unit at most: all resident
        // the grid barrier's target: what the counter reaches when this launch's workgroups have all arrived.  The host's count only
        // moves once the launch is known to be queued (a refused launch adds nothing on the device either), the wait inside the
        // kernel is bounded, and a barrier given up on is reported through a pinned word the step's one wait looks at.
        rp2.bar = ctx->d_prep_bar; rp2.bar_target = ctx->prep_bar_count + grid;
        rp2.stuck = ctx->h_pin ? reinterpret_cast<uint32_t*>(ctx->h_pin + (62 << 10)) : nullptr;
        rp2.kbits = want_bits ? ctx->d_kbits : nullptr;
        hipLaunchKernelGGL(k_ref_prep, dim3(grid), dim3(256), 0, ctx->stream, rp2);
        HIPCHK(hipGetLastError());
        ctx->prep_bar_count += grid;
        ctx->pend_encode = false;
      } else {
      HIPCHK(hipMemsetAsync(ctx->d_khash, 0xFF, (size_t)kslots * 16, ctx->stream));
      hipLaunchKernelGGL(k_kmer_hash, dim3((unsigned)((wrap + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ref, (int64_t)wrap, ctx->d_khash, ctx->d_khash_ovf,
                         kh.mask, kh.shift, kh.wild);
      if (want_bits) {
        HIPCHK(hipMemsetAsync(ctx->d_kbits, 0, (size_t)KB_WORDS * sizeof(KbPair), ctx->stream));
        hipLaunchKernelGGL(k_kmer_bits, dim3((unsigned)((ctx->L + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ref, (int64_t)wrap, (int64_t)ctx->L, ctx->d_kbits);
      }
      }
      if (n > ctx->bx_cap) {
        if (dev_alloc(ctx, &ctx->d_bx_plan, (size_t)n) || dev_alloc(ctx, &ctx->d_bx_expect, (size_t)n) || dev_alloc(ctx, &ctx->d_bx_lists, (size_t)n * 4 * BX_NCLS))
          return MIA_HIP_ERR_NOMEM;
        ctx->bx_cap = n;
      }
      if (!ctx->bx_values_wgs) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ctx->use_lanes ? (const void*)k_bxl_values : ALT_KERNEL(k_bx_values), 256, 0) != hipSuccess || occ < 1) occ = 1;
        ctx->bx_values_wgs = ctx->cus * std::min(occ, 4);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ctx->use_lanes ? (const void*)k_bxl_trace : ALT_KERNEL(k_bx_trace), 256, 0) != hipSuccess || occ < 1) occ = 1;
        ctx->bx_trace_wgs = ctx->cus * std::min(occ, 4);      // (every wavefront of the trace grid owns a slab)
        // (alt build, percent of the above: how much room the persistent grids leave the planner's chain)
        if (const char* pv = alt_env("MIA_HIP_BX_VALUES_PCT")) if (atoi(pv) > 0) ctx->bx_values_wgs = std::max(1, ctx->bx_values_wgs * atoi(pv) / 100);
        if (const char* pt = alt_env("MIA_HIP_BX_TRACE_PCT")) if (atoi(pt) > 0) ctx->bx_trace_wgs = std::max(1, ctx->bx_trace_wgs * atoi(pt) / 100);
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
      bd.dbg = ctx->bx_dbg & (3u | 32u | 64u | 128u | 512u);
      // MIA_HIP_BX_SERIAL=1: round 2's order (band kernels, then the planner over everything they left open)
      // (caller-supplied windows -- mia_hip_align_windows -- can be of any length: the retry list's window kernel is picked by read length)
      const bool new_flow = ctx->use_lanes && !ctx->bx_serial && !(ctx->dbg & 256u) && !ctx->explicit_win;
      if (new_flow) {
        if (n > ctx->retry2_cap) { if (dev_alloc(ctx, &ctx->d_retry2, (size_t)n)) return MIA_HIP_ERR_NOMEM; ctx->retry2_cap = n; }
        if (!ctx->bx_late_wgs) ctx->bx_late_wgs = ctx->cus;        // the values DP's left-overs are few: one workgroup per CU
        const int64_t late_words = (int64_t)((ctx->max_len + 3) & ~3) * BXL_SLAB_ROW_WORDS * ctx->bx_late_wgs * 4;
        if (late_words > ctx->bx_slab_late_cap) {
          if (dev_alloc(ctx, &ctx->d_bx_slabs_late, (size_t)late_words) || dev_alloc(ctx, &ctx->d_bx_slabs_late2, (size_t)late_words)) return MIA_HIP_ERR_NOMEM;
          ctx->bx_slab_late_cap = late_words;
        }
      }
      bd.retry = ctx->d_retry2; bd.retry_n = ctx->d_plan_hdr + PH_RETRY2 + 1;
      bd.listed_mark = new_flow ? -5 : 0;
      if (ctx->lazy_scripts) ctx->diag_scripts_missing = true;
      bd.tab.min_m = ctx->bx_min_m; bd.tab.max_m = ctx->bx_max_m;
      bd.tab.maxw = ctx->use_lanes ? BX_MAXW : 32;           // (the widest class needs a read spread over eight lanes)
      bd.sub256 = ctx->d_bx_sub + BX_SUB_WORDS;
      bd.refnib = ctx->d_refnib;
      bd.umax = (ctx->umax_valid && ctx->rs.roff == ctx->d_roff) ? ctx->d_umax : nullptr;    // (a borrowed read set has none)
      bd.rplanes = bd.umax ? ctx->d_rplanes : nullptr;
      bd.rplane_words = ctx->rplane_words;
      bd.plan = ctx->d_bx_plan; bd.expect = ctx->d_bx_expect; bd.lists = ctx->d_bx_lists; bd.list_stride = ctx->bx_cap; bd.ctr = ctx->d_bx_ctr;
      // the plan in two launches (bandx_kernels.h, phase): the reads with anchors on two diagonals are finished by a second launch
      // with every lane at work (MIA_HIP_NO_PLAN_SPLIT=1: by the first threads of their blocks, one launch)
      const bool split = ctx->plan_split;
      const bool fork_by_launch = (ctx->ext_events & 1u) && new_flow && !(ctx->dbg & 256u);
      bd.cand = nullptr; bd.cand_n = ctx->d_bx_ctr + (size_t)BXC_CAND * BXC_STRIDE;
      bd.cand2 = nullptr; bd.cand2_n = ctx->d_bx_ctr + (size_t)BXC_CAND2 * BXC_STRIDE;
      // a third launch for the reads whose loss exceeds what the 10-mers vouch for (bx_fine_anchors; MIA_HIP_NO_FINE=1: given up as before)
      // ... with a position-specific matrix (two reads in a hundred exceed the 10-mers' budget in every iteration, and the full-window
      // kernels they went to were the largest consumer of vector instructions at 10 M reads: configs[3] 8.47 -> 7.44 ms, configs[2]
      // 1.41 -> 1.35 ms), with many reads (the launch costs next to nothing where the step is bound by throughput), and whenever
      // the plan gives up on many reads: against a reference full of ambiguity codes (every run's first iteration), or when it did so
      // in the iteration before.  With the flat matrix and a million reads the launch would sit on the step's critical path (~55 us)
      // for the sake of a few thousand reads whose full-window kernels run beside the band DPs anyway.
      int64_t rejects_before = 0;
      for (int k = 1; k < BXF_KINDS; k++) rejects_before += ctx->bx_last[BXC_FAIL0 + k];
      const bool fine = split && ctx->use_fine && (ctx->use_fine > 1 || !ctx->flat || n >= 4000000 || !ctx->ref_mostly_bases || rejects_before * 20 > n);
      if (split) {
        if (n > ctx->cand_cap) {
          if (dev_alloc(ctx, &ctx->d_bx_cand, (size_t)n)) return MIA_HIP_ERR_NOMEM;
          ctx->cand_cap = n;
        }
        // (the third launch's list only where that launch runs: a record is 72 bytes -- 0.7 GB at 10 M reads)
        if (fine && n > ctx->cand2_cap) {
          if (dev_alloc(ctx, &ctx->d_bx_cand2, (size_t)n)) return MIA_HIP_ERR_NOMEM;
          ctx->cand2_cap = n;
        }
        bd.cand = ctx->d_bx_cand;
        if (fine) bd.cand2 = ctx->d_bx_cand2;
      }
      const int last_phase = split ? (fine ? 3 : 2) : 0;
      // PLANNER FIRST.  Where the plan gives up on many reads (against a reference full of ambiguity codes -- every run's first
      // iteration --, or when it did so in the iteration before) the planner's chain on stream2 -- count, scan, fill, the values-only
      // quad kernel, the re-plan, the trace quad kernel -- is the longest of the three, and its three small head kernels, launched
      // beside the persistent band grids, wait for wave slots: k_plan_scan's single workgroup 70 us, the other two 65-80 us each
      // instead of 10 (first iteration of 1 M flat reads; at 10 M reads 1 ms and 0.8 ms).  They go in front of the fork then: ~35 us
      // later for the band DPs, ~190 us earlier for the chain that the step waits for.  (Not at steady state, where the
      // planner's chain has slack and the band DPs' start is the step's critical path.)
      {
        int64_t lr = 0;
        for (int k = 1; k < BXF_KINDS; k++) lr += ctx->bx_last[BXC_FAIL0 + k];
        const bool many_early = !ctx->no_auto_plain && (!ctx->ref_mostly_bases || lr * 20 > n);
        planner_head_first = new_flow && ctx->deferred && many_early && !ctx->planner_beside && !(ctx->dbg & 256u);
        // DIRECT OPEN LIST (round 6).  At steady state the plan leaves a few hundred reads per million open.  For their sake the planner
        // counted, scanned and filled over ALL reads (three launches beside the persistent band grids: 13 + 7 + 71 us), a quad kernel took
        // them four to a wavefront (110 us: one quad's latency), three window-class launches and a retry launch followed -- nine launches,
        // 270 us, the longest of the step's three DP chains.  Now the plan appends such a read to a list as it gives up on it and
        // k_align_open takes the list, one read per wavefront (90-115 us beside the band DPs, 35 on an idle chip: off the step's chain either way).  Where the plan gives up on MANY reads (the values-only quad pass is on:
        // every run's first iteration, N-rich references) the planner and the quad kernels stay: four reads per wavefront is what pays there.
        const bool plain_wanted = ctx->use_plain && (ctx->plain_behind_band || many_early);      // (= use_plain below, the band pipeline being on)
        direct_open = new_flow && ctx->deferred && ctx->use_direct_open && !plain_wanted && !planner_head_first && !run_filter && !(ctx->dbg & 256u) &&
                      !(ctx->bx_dbg & (4u | 8u));
      }
      bd.open = nullptr; bd.open_n = ctx->d_bx_ctr + (size_t)BXC_OPEN * BXC_STRIDE;
      ctx->direct_open_now = direct_open;
      if (direct_open) {
        if (n > ctx->open_cap) { if (dev_alloc(ctx, &ctx->d_open_list, (size_t)n)) return MIA_HIP_ERR_NOMEM; ctx->open_cap = n; }
        if (!ctx->d_slabs_open && hipMalloc((void**)&ctx->d_slabs_open, (size_t)OPEN_SLAB_BYTES * OPEN_WGS) != hipSuccess) return MIA_HIP_ERR_NOMEM;
        bd.open = ctx->d_open_list;
        ctx->direct_open_steps++;
      }
      // the early tally (k_rec_early): the plan marks the reads it finishes, their tally runs on stream4 beside the band DPs.  Only in
      // mia_hip_iterate (the whole step is queued at once), for read sets small enough that the step is a chain of latencies.
      const bool early = ctx->deferred && new_flow && ctx->use_early && !run_filter && tally_is_binned(ctx) && n <= 4000000;
      bd.early = nullptr;
      if (early) {
        if (n > ctx->early_cap) {
          if (dev_alloc(ctx, &ctx->d_early, (size_t)n) || dev_alloc(ctx, &ctx->d_trec_early, (size_t)n * 16) || dev_alloc(ctx, &ctx->d_order_e, (size_t)n) ||
              dev_alloc(ctx, &ctx->d_fix_list, (size_t)n)) return MIA_HIP_ERR_NOMEM;
          ctx->early_cap = n;
        }
        bd.early = ctx->d_early;
      }
      // ... and where the plan's third launch is off (flat matrix, a million reads, hardly any rejects: the step is a chain of
      // latencies) the widest class is not used at all: those few reads keep going to the full-window kernels on the planner's stream,
      // whose chain is as long with them as without (measured: 0.960 against 0.944 ms per step with the class in use)
      if (!fine && ctx->use_fine < 2) bd.tab.maxw = 32;
      // THE BAND DPs IN TWO ROUNDS (MIA_HIP_SPLIT_DP=1, alt build).  Where the plan gives up on many reads (every run's
      // first iteration: a reference full of ambiguity codes) its second and third launch are long -- 0.5 + 1.7 ms of 10 M solexa reads,
      // behind 3.1 ms of the first -- and the band DPs wait for all three.  Nine lists in ten are complete after the first launch:
      // k_bx_snap notes their lengths, the DPs of those entries start at once (values DP on stream4, trace DP on stream3) beside the other
      // two launches, and a second round behind the last launch takes what they appended.  MEASURED, NO GAIN: first iteration of 10 M
      // solexa reads 17.13 ms with it, 16.90 without; 1 M flat reads 1.90 / 1.83 -- the plan's launches and the DPs are all bound by
      // vector issue, and what runs side by side only shares the chip.  Kept behind the switch (tests/test_gpu_switches.py runs it).
      const bool split_dp = new_flow && ctx->deferred && split && last_phase >= 2 && !early && !(ctx->bx_dbg & (4u | 8u)) && ctx->split_dp_mode > 0;
      bd.snap = nullptr;
      if (split_dp) {
        if (!ctx->d_bx_snap && dev_alloc(ctx, &ctx->d_bx_snap, (size_t)(2 * BX_NCLS))) return MIA_HIP_ERR_NOMEM;
        bd.snap = ctx->d_bx_snap;
        ctx->split_dp_steps++;
      }
      // THE QUICK PLAN FIRST (round 6; bandx_body.h: bx_quick, k_bx_plan<NW, 4>): every read on the diagonal it was aligned on before -- nine
      // in ten of a steady-state iteration are finished or listed there for a tenth of the full plan's instructions; the rest goes on a
      // list (the diagonal filter's: d_left_list, d_filter_n[1]) that the launches below take as their in_list.  Not against a reference
      // that is N all over (every run's first iteration; one with a few N columns: the windows that hold none, want_bits above), not with the
      // diagonal filter in front, not with the early tally's marks.
      // (... and with the fine blocks on only from two million reads: the three launches of the full plan stay behind it then, each with a
      // launch's floor of 30-90 us for the few reads it has left -- 1 M ancient reads 1.12 ms without it, 1.15 with; 10 M solexa 6.0 / 5.4)
      const bool quick = split && new_flow && !run_filter && !early && !split_dp && want_bits && !(ctx->bx_dbg & 32u) && bd.umax != nullptr &&      // (the context's own reads: their U is at hand)
                         (!fine || n >= 2000000 || ctx->use_quick > 1);
      bd.qlist = nullptr; bd.qlist_n = ctx->d_filter_n + 1; bd.mark_all = 0; bd.to_late = 0; bd.qch = 4;
      // mia_hip_iterate: the fork is behind the QUICK plan (BxDev::to_late) -- values DP and late trace are on the context's stream there, behind
      // the full plan's launch, so no wait between streams is added; mia_hip_realign keeps every launch of the plan in front of the fork
      // (... and only where the plan lists its open reads itself: the planner's kernels on stream2 would want every read's mark at the fork)
      bool plan_launched = false;
      // (... and not with the fine blocks: the three launches of the full plan are then a chain of their own, longer beside the DPs than
      // both DPs together -- configs[2]: 45 + 174 + 114 us and 211 for their lists' trace against 280 for the DPs; in front of the fork)
      const bool fork_at_quick = quick && ctx->deferred && direct_open && !fine && !planner_head_first && !(ctx->bx_dbg & (4u | 8u));
      bd.kb = KmerBits{quick ? ctx->d_kbits : nullptr, ctx->L};
      // (the reference's planes in LDS for the quick plan while they are small: 38 KB holds a hundred thousand columns)
      const bool quick_lds = quick && words * 24 <= 38 * 1024;
      bd.plane_words = (int32_t)(quick_lds ? words : 0);
      if (quick) {
        if (n > ctx->left_cap) { if (dev_alloc(ctx, &ctx->d_left_list, (size_t)n)) return MIA_HIP_ERR_NOMEM; ctx->left_cap = n; }
        bd.qlist = ctx->d_left_list;
        ctx->quick_steps++;
      }
      if (stage_begin(ctx, STG_BX_PLAN)) return MIA_HIP_ERR_NOMEM;
      {
        const int32_t* in_list = (run_filter || quick) ? ctx->d_left_list : nullptr;
        const dim3 pb(256);
        const int nwords = (ctx->max_len + 63) >> 6;       // 64-row words of the longest read
        // behind the quick plan the full plan has one read in a hundred left: without fine blocks ONE launch (phase 0: the reads with anchors on
        // two diagonals by the block's first threads -- a second launch would be a second chain through the table, 40 us, for fifty workgroups)
        const bool one_launch = quick && !fine;
        if (quick) {
          // (eight stretches per workgroup where the lists' counters are the wait -- many reads -- and the planes in LDS are small: with a
          // 100 kb reference's 37 KB of planes the larger per-read arrays cost the third workgroup per compute unit, configs[4] 7.85 -> 8.05 ms)
          bd.qch = (n >= 4000000 && words * 24 <= 16 * 1024) ? BX_QCH : 4;
          const dim3 pg((unsigned)((n + 256 * bd.qch - 1) / (256 * bd.qch)));
          hipEvent_t done = (fork_at_quick && fork_by_launch) ? ctx->ev_fork : nullptr;
          const int32_t* in_list = nullptr;               // (this launch walks all n reads)
#define MIA_PLAN(NWV, PHV) launch_k(k_bx_plan<NWV, PHV>, pg, pb, (size_t)(bx_quick_lds_words(bd.qch) * 8 + (PHV == 5 ? words * 24 : 0)), ctx->stream, done, ctx->rs, ref, rp, kh, (int64_t)wrap, bd, in_list, (const uint32_t*)(ctx->d_filter_n + 1), n, ctx->d_bin_of)
          if (quick_lds) { switch (nwords) { case 1: MIA_PLAN(1, 5); break; case 2: MIA_PLAN(2, 5); break; case 3: MIA_PLAN(3, 5); break; default: MIA_PLAN(4, 5); break; } }
          else { switch (nwords) { case 1: MIA_PLAN(1, 4); break; case 2: MIA_PLAN(2, 4); break; case 3: MIA_PLAN(3, 4); break; default: MIA_PLAN(4, 4); break; } }
#undef MIA_PLAN
          bd.mark_all = 1;
          if (fork_at_quick) {
            if (!fork_by_launch) HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
            bd.to_late = 1;
          }
        }
        const int phase_first = (split && !one_launch) ? 1 : 0, phase_last = one_launch ? 0 : last_phase;
        plan_launched = true;
        for (int phase = phase_first; phase <= phase_last; phase++) {
          // (one_launch: a grid the chip holds at once, looping over the list -- see PH 6 in bandx_kernels.h)
          const dim3 pg(phase >= 2 ? (unsigned)std::min<int64_t>((n + 255) / 256, 1024) : (unsigned)(one_launch ? std::min<int64_t>((n + 255) / 256, 256) : (n + 255) / 256));
          // (the fork event rides on the last launch's own completion signal: no marker between the plan and the values DP)
          hipEvent_t done = (fork_by_launch && phase == phase_last && !planner_head_first && !fork_at_quick) ? ctx->ev_fork : nullptr;
#define MIA_PLAN(NWV, PHV) launch_k(k_bx_plan<NWV, PHV>, pg, pb, 0, ctx->stream, done, ctx->rs, ref, rp, kh, (int64_t)wrap, bd, in_list, (const uint32_t*)(ctx->d_filter_n + 1), n, ctx->d_bin_of)
#define MIA_PLAN_NW(PHV) switch (nwords) { case 1: MIA_PLAN(1, PHV); break; case 2: MIA_PLAN(2, PHV); break; case 3: MIA_PLAN(3, PHV); break; default: MIA_PLAN(4, PHV); break; }
          switch (phase) {
            case 0: if (one_launch) { MIA_PLAN_NW(6) } else { MIA_PLAN_NW(0) } break;
            case 1: MIA_PLAN_NW(1) break;
            case 2: MIA_PLAN_NW(2) break;
            default: MIA_PLAN_NW(3) break;
          }
#undef MIA_PLAN_NW
#undef MIA_PLAN
          if (split_dp && phase == 1) {
            launch_k(k_bx_snap, dim3(1), dim3(64), 0, ctx->stream, ctx->ev_snap, (const uint32_t*)ctx->d_bx_ctr, ctx->d_bx_snap);
            HIPCHK(hipStreamWaitEvent(ctx->stream3, ctx->ev_snap, 0));
            HIPCHK(hipStreamWaitEvent(ctx->stream4, ctx->ev_snap, 0));
            if (stage_begin(ctx, STG_BX_TRACE, ctx->stream3)) return MIA_HIP_ERR_NOMEM;
            hipLaunchKernelGGL(k_bxl_trace, dim3((unsigned)ctx->bx_trace_wgs), dim3(256), 0, ctx->stream3, ctx->rs, ref, bd, ctx->d_bx_slabs, slab_words, ctx->d_bin_of, (int32_t)BX_PART_HEAD);
            stage_end(ctx, STG_BX_TRACE, ctx->stream3);
            if (stage_launch(ctx, STG_BX_VALUES, k_bxl_values, dim3((unsigned)ctx->bx_values_wgs), dim3(256), 0, ctx->stream4, ctx->rs, ref, bd, ctx->d_bin_of, (int32_t)BX_PART_HEAD))
              return MIA_HIP_ERR_NOMEM;
            HIPCHK(hipEventRecord(ctx->ev_v1, ctx->stream4));
          }
        }
        bd.to_late = 0;
      }
      stage_end(ctx, STG_BX_PLAN);
      HIPCHK(hipGetLastError());
      if (planner_head_first) {
        const bool use_plain_h = ctx->use_plain;      // (banded, many rejects: the values-only quad pass is on)
        hipLaunchKernelGGL(k_plan_count, dim3(gb), dim3(tb), 0, ctx->stream, ctx->rs, ref, ctx->packs, ctx->use_quad, filtered, filtered && use_plain_h, ctx->d_bin_of, d_count, ctx->d_filter_n);
        hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(512), 0, ctx->stream, d_count, d_off, ctx->d_plan_hdr, 0, ctx->d_list);
        launch_k(k_plan_fill, dim3(gb), dim3(tb), 0, ctx->stream, fork_by_launch ? ctx->ev_fork : nullptr, n, (const int32_t*)ctx->d_bin_of, (const int32_t*)d_off, d_cursor, ctx->d_list);
        HIPCHK(hipGetLastError());
      }
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
          // fork_at_quick (round 6): the context's stream holds the full plan's launch for the reads the quick plan left and the trace launch
          // of what THAT lists (the second late lists); values DP and late trace run beside them on stream2, the trace DP on stream3 -- three
          // chains of about the same length instead of plan -> values DP -> late trace one after the other, and the join waits for all
          hipStream_t vs = (ctx->deferred && !fork_at_quick) ? ctx->stream : ctx->stream2;
          ctx->dp_aside = fork_at_quick;
          if (fork_at_quick && plan_launched) {
            if (stage_launch(ctx, STG_BX_VALUES, k_bxl_trace_late2, dim3((unsigned)ctx->bx_late_wgs), dim3(256), 0, ctx->stream, ctx->rs, ref, bd, ctx->d_bx_slabs_late2, slab_words, ctx->d_bin_of))
              return MIA_HIP_ERR_NOMEM;
          }
          if (!fork_by_launch && !fork_at_quick) HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
          HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
          HIPCHK(hipStreamWaitEvent(ctx->stream3, ctx->ev_fork, 0));
          if (early) HIPCHK(hipStreamWaitEvent(ctx->stream4, ctx->ev_fork, 0));
          if (stage_begin(ctx, STG_BX_TRACE, ctx->stream3)) return MIA_HIP_ERR_NOMEM;
          const bool sig3 = (ctx->ext_events & 2u) && new_flow && !(ctx->bx_dbg & 8u);
          const int32_t part2 = split_dp ? BX_PART_TAIL : BX_PART_ALL;
          if (!(ctx->bx_dbg & 8u))
            launch_k(k_bxl_trace, dim3((unsigned)ctx->bx_trace_wgs), dim3(256), 0, ctx->stream3, sig3 ? ctx->ev_join3 : nullptr, ctx->rs, ref, bd, ctx->d_bx_slabs, slab_words, ctx->d_bin_of, part2);
          stage_end(ctx, STG_BX_TRACE, ctx->stream3);
          if (!sig3) HIPCHK(hipEventRecord(ctx->ev_join3, ctx->stream3));
          if (!(ctx->bx_dbg & 4u)) {
            if (stage_launch(ctx, STG_BX_VALUES, k_bxl_values, dim3((unsigned)ctx->bx_values_wgs), dim3(256), 0, vs, ctx->rs, ref, bd, ctx->d_bin_of, part2)) return MIA_HIP_ERR_NOMEM;
            if (split_dp) HIPCHK(hipStreamWaitEvent(vs, ctx->ev_v1, 0));          // (the late lists hold both rounds' left-overs)
            if (stage_launch(ctx, STG_BX_VALUES, k_bxl_trace_late, dim3((unsigned)ctx->bx_late_wgs), dim3(256), 0, vs, ctx->rs, ref, bd, ctx->d_bx_slabs_late, slab_words, ctx->d_bin_of))
              return MIA_HIP_ERR_NOMEM;
          }
          if (!ctx->deferred || fork_at_quick) HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
          HIPCHK(hipGetLastError());
          ctx->bx_planner_aside = ctx->deferred;
          ctx->bx_pending_join = true;
          // (queued behind the band DPs on the host side: the GPU starts it as soon as the plan's last launch is done)
          if (early) { if (int rce = early_tally_launch(ctx)) return rce; }
        } else {
        // The two band DPs do not depend on each other (a read the values DP cannot finish stays open for the full-window
        // kernels): they run side by side on two streams -- both are persistent grids whose wavefronts leave as soon as the
        // chunks run out, so each fills what the other leaves idle, and the step pays the longer of the two tails, not both.
        HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        if (stage_begin(ctx, STG_BX_TRACE, ctx->stream2)) return MIA_HIP_ERR_NOMEM;
        if (ctx->bx_dbg & 8u) {}
        else if (ctx->use_lanes) hipLaunchKernelGGL(k_bxl_trace, dim3((unsigned)ctx->bx_trace_wgs), dim3(256), 0, ctx->stream2, ctx->rs, ref, bd, ctx->d_bx_slabs, slab_words, ctx->d_bin_of, (int32_t)BX_PART_ALL);
#ifdef MIA_HIP_ALT_PATHS
        else hipLaunchKernelGGL(k_bx_trace, dim3((unsigned)ctx->bx_trace_wgs), dim3(256), 0, ctx->stream2, ctx->rs, ref, bd, ctx->d_bx_slabs, slab_words, ctx->d_bin_of);
#endif
        stage_end(ctx, STG_BX_TRACE, ctx->stream2);
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
        if (stage_begin(ctx, STG_BX_VALUES)) return MIA_HIP_ERR_NOMEM;
        if (ctx->bx_dbg & 4u) {}
        else if (ctx->use_lanes) hipLaunchKernelGGL(k_bxl_values, dim3((unsigned)ctx->bx_values_wgs), dim3(256), 0, ctx->stream, ctx->rs, ref, bd, ctx->d_bin_of, (int32_t)BX_PART_ALL);
#ifdef MIA_HIP_ALT_PATHS
        else hipLaunchKernelGGL(k_bx_values, dim3((unsigned)ctx->bx_values_wgs), dim3(256), 0, ctx->stream, ctx->rs, ref, bd, ctx->d_bin_of);
#endif
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
  if (!planner_head_first && !direct_open)
  hipLaunchKernelGGL(k_plan_count, dim3(gb), dim3(tb), 0, ps, ctx->rs, ref, ctx->packs, ctx->use_quad, filtered, filtered && use_plain, ctx->d_bin_of, d_count, ctx->d_filter_n);
  if (ctx->deferred) {
    // ---- mia_hip_iterate: the same plan, but its numbers stay on the device (k_plan_scan) and every DP kernel reads its own
    // range; the host looks at the counters once, when everything has been queued --------------------------------------
    int32_t* hdr = ctx->d_plan_hdr;
    const bool dbg_steps = alt_env("MIA_HIP_ITER_DEBUG") != nullptr;
    auto ck = [&](const char* what) { if (dbg_steps) { hipError_t e = hipStreamSynchronize(ps); fprintf(stderr, "[align_all deferred] %s: %s\n", what, hipGetErrorString(e)); fflush(stderr); } };
    ck("plan_count");
    int32_t* d_retry_cnt = hdr + PH_RETRY + 1;
    if (direct_open) {
      // (the plan's open list is taken by the alignment's LAST launch, together with the band DPs' retry list: bx_join_and_retry)
      ctx->planner_end_signalled = false;
    } else {
    if (!planner_head_first) {
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(512), 0, ps, d_count, d_off, hdr, 0, ctx->d_list);      // (writes the quad bins' padding itself: -1 = empty slot)
    hipLaunchKernelGGL(k_plan_fill, dim3(gb), dim3(tb), 0, ps, n, ctx->d_bin_of, d_off, d_cursor, ctx->d_list);
    }
    ck("memsets");
    hipLaunchKernelGGL(k_wide_seed, dim3(1), dim3(256), 0, ps, ctx->d_list, hdr, ctx->d_wide_list, d_wide_count,
                       (ctx->use_quad && use_plain) ? d_count : (int32_t*)nullptr, d_cursor);      // (cleared for the re-plan below)
    ck("scan fill seed");
    // every window class reads its own range from the header (the list is rewritten by the re-plan below, so they all go
    // before it; a class without reads costs an empty launch -- a few microseconds -- and three of them in front of the quad
    // kernel delayed it by ~50 us of launch gaps on a chain that ends about when the band DPs' does: they go behind its first launch)
    auto window_classes = [&]() -> int {
      for (int ci = 0; ci < N_CPL; ci++) {
        hipError_t e = ci == 0 ? launch_window<4>(ctx, ci, ctx->d_list, 0, hdr + PH_WIN + 2 * ci, ps)
                     : ci == 1 ? launch_window<8>(ctx, ci, ctx->d_list, 0, hdr + PH_WIN + 2 * ci, ps)
                               : launch_window<12>(ctx, ci, ctx->d_list, 0, hdr + PH_WIN + 2 * ci, ps);
        if (e != hipSuccess) { ctx->err = std::string("k_align_window launch: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
      }
      ck("window classes");
      return MIA_HIP_OK;
    };
    bool windows_done = false;
    const size_t quad_lds = (size_t)Q_G * q_sub_bytes(ctx->max_len) + 16;
    if (ctx->use_quad) {
      if (use_plain) {
        if (stage_begin(ctx, STG_PLAIN, ps)) return MIA_HIP_ERR_NOMEM;
        hipLaunchKernelGGL(k_align_quad_plain, dim3(ctx->quad_wgs), dim3(64), quad_lds, ps, ctx->rs, ref, ctx->d_pssm, ctx->d_list, 0, ctx->d_bin_of,
                           (const int32_t*)(hdr + PH_QUAD));
        stage_end(ctx, STG_PLAIN, ps);
        ck("quad plain");
        if (int rcw = window_classes()) return rcw;
        windows_done = true;
        // what it could not finish (and what the filter's gap hint kept out of it), re-planned into quads
        // (d_count / d_cursor: cleared by k_wide_seed)
        hipLaunchKernelGGL(k_plan_recount, dim3(gb), dim3(tb), 0, ps, n, ctx->d_bin_of, d_count);
        hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(512), 0, ps, d_count, d_off, hdr, 1, ctx->d_list);
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
      if (!windows_done) { if (int rcw = window_classes()) return rcw; windows_done = true; }
      if (ctx->use_band) {
        // (the last launch of the planner's chain: on stream2 it signals ev_join itself, see bx_join_and_retry)
        const bool sig = ctx->bx_planner_aside && ctx->bx_pending_join && (ctx->ext_events & 4u);
        hipError_t e = launch_window<4>(ctx, 0, ctx->d_retry_list, 0, hdr + PH_RETRY, ps, false, sig ? ctx->ev_join : nullptr);
        ctx->planner_end_signalled = sig && e == hipSuccess;
        if (e != hipSuccess) { ctx->err = std::string("k_align_window retry launch: ") + hipGetErrorString(e); return MIA_HIP_ERR_DEVICE; }
      }
    }
    if (!windows_done) { if (int rcw = window_classes()) return rcw; }
    }
    ck("retry");
    if (int rcj = bx_join_and_retry(ctx)) return rcj;
    ck("band join");
    // the one look at the counters: wide / retry counts, the planner's header, filter and band-pipeline counters lie side by
    // side in the control block
    constexpr int C0 = CTRL_C0, CN = CTRL_CN;
    std::vector<int32_t> pageable;
    int32_t* hb;
    if (ctx->h_pin) hb = reinterpret_cast<int32_t*>(ctx->h_pin);
    else { pageable.resize(CN); hb = pageable.data(); }
    const bool pre_cull = ctx->in_iterate && ctx->comm;
    if (pre_cull) { if (int rcp = comm_pre_cull_enqueue(ctx, d_wide_count)) return rcp; }
    const bool spec = ctx->spec_ok && ctx->h_pin && !pre_cull && !dbg_steps;
    // (a copy in the middle of the step holds the stream up for 20 us: with zero_copy the step's last kernel, k_cons_scatter,
    // writes these words into h_pin itself, whether or not the speculation held)
    if (!(spec && ctx->zero_copy)) HIPCHK(hipMemcpyAsync(hb, ctx->d_ctrl + C0, (size_t)CN * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (spec) {
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
                     (int32_t)ctx->links_cap_all, ctx->read_base, ctx->d_drop_f, ctx->d_drop_b, ctx->d_cull_flags, ctx->abort_if,
                     ctx->early_queued ? ctx->d_early : (const uint8_t*)nullptr, ctx->d_fix_list, ctx->d_ctrl + CTRL_FIXN,
                     (ctx->umax_valid && ctx->rs.roff == ctx->d_roff && ctx->d_rplanes && ctx->d_umax) ? ctx->d_umax : (const int32_t*)nullptr,
                     (ctx->cull_with_records && ctx->d_n_links_all == ctx->lk.n && !ctx->early_queued) ? 1 : 0);
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
  // (MIA_HIP_CULL_SCAN=1: the counts scanned by a single-workgroup launch in between, as before)
  if (ctx->cull_scan) hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, ctx->stream, ctx->d_partial, nb, slot_base, ctx->d_total, ctx->abort_if);
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
                     (const int64_t*)ctx->d_front_slot0, ctx->lk, ctx->dev_cut, ctx->abort_if, ctx->cull_scan ? 0 : nb, slot_base, ctx->d_total,
                     ctx->cull_with_records ? 1 : 0, ctx->d_drop_f, ctx->d_drop_b,
                     (ctx->umax_valid && ctx->rs.roff == ctx->d_roff && ctx->d_rplanes && ctx->d_umax) ? ctx->d_umax : (const int32_t*)nullptr);
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
// the position-specific matrices' tally sorts by strand as well (bucket_of, mia_consensus_kernels.h): twice the buckets
// (reads of more than 128 bases -- configs[4]'s 150 -- take the RALL instance of k_tally_binned: every row through the runs; the vertical
// counters and the end-row runs work on two 64-row plane words.  With the split but neither -- a first attempt -- such reads paid for the
// part-filled workgroups and got nothing for it: 2.18 against round 4's 1.51 ms per 5 M reads of 150 bases.)
static bool tally_rall(const mia_hip_ctx* ctx) { return ctx->max_len > 128 && ctx->max_len <= 256 && ctx->tally_runs && ctx->tally_rall; }
static int tally_split(const mia_hip_ctx* ctx) {
  return (!ctx->tally_linear && ctx->tally_strand_split && !ctx->early_queued && !ctx->use_early && (ctx->max_len <= 128 || tally_rall(ctx))) ? 1 : 0;
}
static int tally_nb(const mia_hip_ctx* ctx) { return (ctx->wrap / TALLY_BUCKET + 1) << tally_split(ctx); }
static bool tally_is_binned(const mia_hip_ctx* ctx) {
  const int nb = ctx->wrap / TALLY_BUCKET + 1;
  return ctx->rs.n > 0 && ctx->use_binned_tally && nb <= 4096 && ctx->max_abs <= 32767;   // (the LDS copy of the matrices is int16)
}
// Counting sort of the reads by alignment start (k_bucket_count / _scan / _fill) on stream `on`.  It reads nothing but the
// alignment starts, so mia_hip_iterate queues it on stream2 BESIDE the cull kernels (it also clears the tally buffers:
// nothing adds to them before the tally kernel).  tally_launch waits for ev_join if `on` is not the context's stream.
// reads per workgroup of the binned tally (the linear matrix: one read per lane -- measured, mia_consensus_kernels.h)
static int tally_chunk(const mia_hip_ctx* ctx) { return ctx->early_queued ? TALLY_CHUNK_LATE : (ctx->tally_linear ? ctx->tally_chunk_linear : TALLY_CHUNK); }
static int bucket_launch(mia_hip_ctx* ctx, hipStream_t on) {
  int rc = ensure_tally(ctx);
  if (rc) return rc;
  const int Lp = ctx->tb.Lp;
  const int64_t tally_words = (int64_t)(TALLY_WORDS + 1) * Lp + 256;                               // tally, gaps, the ranks' event counts
  const int64_t n = ctx->rs.n;
  const int nb = tally_nb(ctx), split = tally_split(ctx);
  // (behind an early tally what is left are the reads with substitutions and gaps -- a fifth of them, and the slow ones: smaller shares
  // per workgroup, or a few hundred workgroups with 512 slow reads each take longer than the whole tally did)
  const int chunk = tally_chunk(ctx);
  const int grid = (int)(n / chunk) + nb + 1;
  if (4 * (nb + 1) + 4 * grid > ctx->bucket_cap) {
    if (dev_alloc(ctx, &ctx->d_bucket, (size_t)(4 * (nb + 1) + 4 * grid) * 2)) return MIA_HIP_ERR_NOMEM;
    ctx->bucket_cap = (4 * (nb + 1) + 4 * grid) * 2;
    ctx->bucket_clean_nb = -1;
  }
  if (!ctx->d_order && dev_alloc(ctx, &ctx->d_order, (size_t)n)) return MIA_HIP_ERR_NOMEM;
  int32_t *d_cnt = ctx->d_bucket, *d_off = d_cnt + (nb + 1), *d_wgoff = d_off + (nb + 1), *d_cur = d_wgoff + (nb + 1), *d_wgb = d_cur + (nb + 1);
  // the bucket counts are left at zero by k_bucket_scan; only a fresh (or differently laid out) buffer is cleared here
  if (ctx->bucket_clean_nb != nb) { HIPCHK(hipMemsetAsync(d_cnt, 0, (size_t)(nb + 1) * 4, on)); ctx->bucket_clean_nb = nb; }
  const int gb = (int)((n + 256 * BUCKET_PER - 1) / (256 * BUCKET_PER));
  // (with an early tally queued: only the reads it did not take, and the tally buffers are left alone -- its own sort cleared them)
  const uint8_t* part = ctx->early_queued ? ctx->d_early : nullptr;
  hipLaunchKernelGGL(k_bucket_count, dim3(gb), dim3(256), (size_t)nb * 4, on, ctx->rs, nb, d_cnt, part ? (int32_t*)nullptr : ctx->tb.tally, part ? (int64_t)0 : tally_words,
                     ctx->abort_if, part, 0, split);
  hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(256), 0, on, d_cnt, nb, d_off, d_wgoff, d_cur, d_wgb, ctx->abort_if, chunk);
  const bool sigb = on != ctx->stream && (ctx->ext_events & 16u);
  if (!split) launch_k(k_bucket_fill, dim3(gb), dim3(256), (size_t)nb * 8, on, sigb ? ctx->ev_join : nullptr, ctx->rs, nb, d_off, d_cur, ctx->d_order, ctx->abort_if, part, 0, split, (uint8_t*)nullptr, 0);
  else {
    // ... and every bucket by alignment start (k_sort2_count / k_sort2_fill: the tally's runs of equal starts)
    const int64_t words2 = (int64_t)nb * SORT2_KEYS * 2;
    if (words2 > ctx->sort2_cap) { if (dev_alloc(ctx, &ctx->d_sort2, (size_t)words2)) return MIA_HIP_ERR_NOMEM; ctx->sort2_cap = words2; }
    // (each on its own: a failure of the second must not leave the first behind as "both are there" -- ADVICE r05)
    if (!ctx->d_order2 && dev_alloc(ctx, &ctx->d_order2, (size_t)n)) return MIA_HIP_ERR_NOMEM;
    if (!ctx->d_okey && dev_alloc(ctx, &ctx->d_okey, (size_t)n + 64)) return MIA_HIP_ERR_NOMEM;
    HIPCHK(hipMemsetAsync(ctx->d_sort2, 0, (size_t)words2 * 4, on));
    const int32_t packed = (n < (1 << 24) && !alt_env("MIA_HIP_SORT2_UNPACKED")) ? 1 : 0;      // (the key in the entry's top byte, or -- 2^24 reads and more -- in a byte array beside it)
    hipLaunchKernelGGL(k_bucket_fill, dim3(gb), dim3(256), (size_t)nb * 8, on, ctx->rs, nb, (const int32_t*)d_off, d_cur, ctx->d_order, ctx->abort_if, part, 0, split, ctx->d_okey, packed);
    hipLaunchKernelGGL(k_sort2_count, dim3(grid), dim3(256), 0, on, nb, (const int32_t*)d_wgoff, (const int32_t*)d_wgb, (const int32_t*)ctx->d_order, (const uint8_t*)ctx->d_okey, packed,
                       ctx->d_sort2, ctx->abort_if);
    launch_k(k_sort2_fill, dim3(grid), dim3(256), 0, on, sigb ? ctx->ev_join : nullptr, nb, (const int32_t*)d_off, (const int32_t*)d_wgoff, (const int32_t*)d_wgb,
             (const int32_t*)ctx->d_order, (const uint8_t*)ctx->d_okey, packed, (const int32_t*)ctx->d_sort2, ctx->d_sort2 + (int64_t)nb * SORT2_KEYS, ctx->d_order2, ctx->abort_if);
  }
  HIPCHK(hipGetLastError());
  if (on != ctx->stream && !sigb) HIPCHK(hipEventRecord(ctx->ev_join, on));
  ctx->buckets_queued = on != ctx->stream ? 2 : 1;
  return MIA_HIP_OK;
}

// The early tally: record, counting sort and LDS-window tally of the reads the plan has finished, on stream4 (behind ev_fork);
// ev_early tells tally_launch when its windows may be summed.  None of its kernels looks at abort_if: the reads it takes do not
// change if the step's second half is queued again (reads for the exact kernel), and its sort clears the tally buffers once.
static int early_tally_launch(mia_hip_ctx* ctx) {
  if (int rc = ensure_tally(ctx)) return rc;
  hipStream_t on = ctx->stream4;
  const int Lp = ctx->tb.Lp;
  const int64_t tally_words = (int64_t)(TALLY_WORDS + 1) * Lp + 256;
  const int64_t n = ctx->rs.n;
  const int nb = ctx->wrap / TALLY_BUCKET + 1;
  const int grid = (int)(n / TALLY_CHUNK) + nb + 1;
  if (4 * (nb + 1) + 4 * grid > ctx->bucket_e_cap) {
    if (dev_alloc(ctx, &ctx->d_bucket_e, (size_t)(4 * (nb + 1) + 4 * grid) * 2)) return MIA_HIP_ERR_NOMEM;
    ctx->bucket_e_cap = (4 * (nb + 1) + 4 * grid) * 2;
    ctx->bucket_e_clean_nb = -1;
  }
  int32_t *d_cnt = ctx->d_bucket_e, *d_off = d_cnt + (nb + 1), *d_wgoff = d_off + (nb + 1), *d_cur = d_wgoff + (nb + 1), *d_wgb = d_cur + (nb + 1);
  if (ctx->bucket_e_clean_nb != nb) { HIPCHK(hipMemsetAsync(d_cnt, 0, (size_t)(nb + 1) * 4, on)); ctx->bucket_e_clean_nb = nb; }
  const int64_t slab_words = (int64_t)grid * (TALLY_WORDS - 1) * TALLY_WIN;
  if (slab_words > ctx->tally_slab_e_cap) {
    if (dev_alloc(ctx, &ctx->d_tally_slabs_e, (size_t)slab_words)) return MIA_HIP_ERR_NOMEM;
    ctx->tally_slab_e_cap = slab_words;
  }
  RefInfo ref{ctx->d_ref, ctx->L, ctx->wrap};
  hipLaunchKernelGGL(k_rec_early, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, on, ctx->rs, ctx->L, (const uint8_t*)ctx->d_early, ctx->d_trec_early,
                     (ctx->umax_valid && ctx->rs.roff == ctx->d_roff && ctx->d_rplanes && ctx->d_umax) ? ctx->d_umax : (const int32_t*)nullptr);
  const int gb = (int)((n + 256 * BUCKET_PER - 1) / (256 * BUCKET_PER));
  hipLaunchKernelGGL(k_bucket_count, dim3(gb), dim3(256), (size_t)nb * 4, on, ctx->rs, nb, d_cnt, ctx->tb.tally, tally_words, (const int32_t*)nullptr,
                     (const uint8_t*)ctx->d_early, 1);
  hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(256), 0, on, d_cnt, nb, d_off, d_wgoff, d_cur, d_wgb, (const int32_t*)nullptr, TALLY_CHUNK);
  hipLaunchKernelGGL(k_bucket_fill, dim3(gb), dim3(256), (size_t)nb * 8, on, ctx->rs, nb, (const int32_t*)d_off, d_cur, ctx->d_order_e, (const int32_t*)nullptr,
                     (const uint8_t*)ctx->d_early, 1);
  const bool planes_ok = ctx->umax_valid && ctx->rs.roff == ctx->d_roff && ctx->d_rplanes && ctx->d_umax;
  // This launch has the whole band-DP phase to finish in and must not take the DPs' issue slots: extra LDS per workgroup (unused)
  // keeps it to `early_wgs_per_cu` workgroups per compute unit (MIA_HIP_EARLY_WGS, alt build; 0: no limit)
  size_t throttle = 0;
  if (ctx->early_wgs_per_cu > 0) {
    const size_t lds_static = ctx->tally_linear ? 27 * 1024 : 41 * 1024, want = (size_t)(160 * 1024) / (size_t)ctx->early_wgs_per_cu;
    if (want > lds_static + 1024) throttle = std::min<size_t>(want - lds_static - 512, (size_t)(64 * 1024) - lds_static - 512);
  }
  // (the plan's reads all take the one-read-per-lane route: nothing of this launch goes to k_tally_general)
  if (ctx->tally_linear
        ? stage_launch(ctx, STG_TALLY, k_tally_binned<true, false>, dim3(grid), dim3(256), throttle, on, ctx->rs, ref, ctx->d_pssm, ctx->d_drop_f, ctx->d_drop_b,
                       ctx->tb, nb, d_off, d_wgoff, ctx->d_order_e, ctx->d_trec_early, ctx->ri.actf, ctx->d_tally_slabs_e, ctx->dbg,
                       planes_ok ? ctx->d_rplanes : nullptr, ctx->rplane_words, planes_ok ? ctx->d_umax : nullptr, d_wgb, -1, (const int32_t*)nullptr, (int32_t)TALLY_CHUNK,
                       (int32_t*)nullptr, (int32_t*)nullptr, 0)
        : stage_launch(ctx, STG_TALLY, k_tally_binned<false, false>, dim3(grid), dim3(256), throttle, on, ctx->rs, ref, ctx->d_pssm, ctx->d_drop_f, ctx->d_drop_b,
                       ctx->tb, nb, d_off, d_wgoff, ctx->d_order_e, ctx->d_trec_early, ctx->ri.actf, ctx->d_tally_slabs_e, ctx->dbg,
                       (const uint64_t*)nullptr, 0, (const int32_t*)nullptr, d_wgb, ctx->tally_pk_bias, (const int32_t*)nullptr, (int32_t)TALLY_CHUNK,
                       (int32_t*)nullptr, (int32_t*)nullptr, 0))
    return MIA_HIP_ERR_NOMEM;
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev_early, on));
  ctx->early_queued = true;
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
    const int nb = tally_nb(ctx), split = tally_split(ctx);
    if (binned) {
      // counting sort of the reads by alignment start, then one LDS tally window per workgroup
      if (!ctx->buckets_queued) { if (int rcb = bucket_launch(ctx, ctx->stream)) return rcb; }
      if (ctx->buckets_queued == 2) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
      ctx->buckets_queued = 0;
      const int chunk = tally_chunk(ctx);
      const int grid = (int)(n / chunk) + nb + 1;
      int32_t *d_cnt = ctx->d_bucket, *d_off = d_cnt + (nb + 1), *d_wgoff = d_off + (nb + 1), *d_cur = d_wgoff + (nb + 1), *d_wgb = d_cur + (nb + 1);
      const int64_t slab_words = (int64_t)grid * (TALLY_WORDS - 1) * TALLY_WIN;
      if (slab_words > ctx->tally_slab_cap) {
        if (dev_alloc(ctx, &ctx->d_tally_slabs, (size_t)slab_words)) return MIA_HIP_ERR_NOMEM;
        ctx->tally_slab_cap = slab_words;
      }
      // (the bit planes and the N marks of the context's own reads: k_read_planes / k_bx_umax at upload)
      const bool planes_ok = ctx->umax_valid && ctx->rs.roff == ctx->d_roff && ctx->d_rplanes && ctx->d_umax;
      const bool defer = ctx->tally_defer;
      if (defer && n > ctx->gen_cap) {
        if (dev_alloc(ctx, &ctx->d_gen_list, (size_t)n)) return MIA_HIP_ERR_NOMEM;
        ctx->gen_cap = n;
      }
      int32_t* n_gen = ctx->d_ctrl + CTRL_GENN;
      if (defer && !ctx->in_iterate) HIPCHK(hipMemsetAsync(n_gen, 0, 4, ctx->stream));     // (mia_hip_iterate clears the whole control block)
#define MIA_TALLY_ARGS(PL, RW, UM, BIAS)                                                                                                              \
  dim3(grid), dim3(256), 0, ctx->stream, ctx->rs, ref, ctx->d_pssm, ctx->d_drop_f, ctx->d_drop_b, ctx->tb, nb, d_off, d_wgoff, split ? ctx->d_order2 : ctx->d_order, ctx->ri.trec, \
      ctx->ri.actf, ctx->d_tally_slabs, ctx->dbg, PL, RW, UM, d_wgb, BIAS, ctx->abort_if, chunk, defer ? ctx->d_gen_list : (int32_t*)nullptr, n_gen, split | ((split && ctx->tally_runs) ? 2 : 0)
      const uint64_t* pl_arg = planes_ok ? ctx->d_rplanes : nullptr;
      const int32_t* um_arg = planes_ok ? ctx->d_umax : nullptr;
      int rct;
      if (ctx->tally_linear)
        rct = defer ? stage_launch(ctx, STG_TALLY, k_tally_binned<true, true>, MIA_TALLY_ARGS(pl_arg, ctx->rplane_words, um_arg, -1))
                    : stage_launch(ctx, STG_TALLY, k_tally_binned<true, false>, MIA_TALLY_ARGS(pl_arg, ctx->rplane_words, um_arg, -1));
      else if (split && tally_rall(ctx))
        rct = defer ? stage_launch(ctx, STG_TALLY, k_tally_binned<false, true, true>, MIA_TALLY_ARGS(pl_arg, ctx->rplane_words, um_arg, ctx->tally_pk_bias))
                    : stage_launch(ctx, STG_TALLY, k_tally_binned<false, false, true>, MIA_TALLY_ARGS(pl_arg, ctx->rplane_words, um_arg, ctx->tally_pk_bias));
      else
        rct = defer ? stage_launch(ctx, STG_TALLY, k_tally_binned<false, true>, MIA_TALLY_ARGS(split ? pl_arg : nullptr, ctx->rplane_words, split ? um_arg : nullptr, ctx->tally_pk_bias))
                    : stage_launch(ctx, STG_TALLY, k_tally_binned<false, false>, MIA_TALLY_ARGS(split ? pl_arg : nullptr, ctx->rplane_words, split ? um_arg : nullptr, ctx->tally_pk_bias));
#undef MIA_TALLY_ARGS
      if (rct) return MIA_HIP_ERR_NOMEM;
      if (ctx->early_queued) {
        // the early tally's corrections (reads whose true record is not the ordinary one it assumed), then its windows join the sum
        hipLaunchKernelGGL(k_tally_fix, dim3(64), dim3(256), 0, ctx->stream, ctx->rs, ref, ctx->d_pssm, ctx->d_drop_f, ctx->d_drop_b, ctx->tb, ctx->ri.trec,
                           ctx->d_trec_early, ctx->ri.actf, ctx->d_fix_list, ctx->d_ctrl + CTRL_FIXN, ctx->abort_if);
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_early, 0));
        int32_t* e_wgoff = ctx->d_bucket_e + 2 * (nb + 1);
        GenReads none{ctx->rs, ref, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(k_tally_reduce, dim3((Lp + 255) / 256, TALLY_WORDS - 1), dim3(256), 0, ctx->stream, ctx->tb, nb, e_wgoff, ctx->d_tally_slabs_e, ctx->abort_if, none);
      }
      GenReads gen{ctx->rs, ref, ctx->d_pssm, ctx->d_drop_f, ctx->d_drop_b, ctx->ri.trec, ctx->ri.actf, defer ? ctx->d_gen_list : nullptr, n_gen};
      hipLaunchKernelGGL(k_tally_reduce, dim3((Lp + 255) / 256 + (defer ? TALLY_GEN_BLOCKS : 0), TALLY_WORDS - 1, TALLY_REDUCE_SHARES), dim3(256), 0, ctx->stream, ctx->tb, nb, d_wgoff,
                         ctx->d_tally_slabs, ctx->abort_if, gen, split);
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
  HIPCHK(hipMemsetAsync(ctx->d_ins_tally, 0, (size_t)ins_words(cap) * 4, ctx->stream));
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
    if (dev_alloc(ctx, &ctx->d_ins_tally, (size_t)ins_words(cap)) || dev_alloc(ctx, &ctx->d_ins_calls, (size_t)cap)) return MIA_HIP_ERR_NOMEM;
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

// (the communicator entry points and mia_hip_iterate: mia_hip_iterate.inc; trimming, ma, pass 1, Myers, ceilings: mia_hip_tools.inc)
#include "mia_hip_iterate.inc"
#include "mia_hip_tools.inc"
