// mia_consensus_kernels.h -- cull, per-column tallies and consensus calls.
//
// The reference builds the consensus with an O(L * N) double loop over
// (reference position, aligned record) (src/mia.c:551-599).  Here every read
// scatter-adds its own columns into a [12][L+1] int32 tally (O(sum of read
// lengths)); integer adds commute, so the result is bit-identical and can be
// all-reduced across GPUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mia_kernels.h"
#include "mia_layout.h"

namespace mia {

struct TallyBuf {
  int32_t* tally;      // [TALLY_WORDS][Lp]
  int32_t Lp;          // L + 1 (a record may end on column L, src/mia_main.c:259-263 uses '>')
  int32_t* gaps;       // [Lp]  longest insert before each column (ref->gaps, src/mia.c:486-504)
  uint64_t* events;    // insert bases: col | j<<32 | code<<42 | depth<<45 | rc<<50
  int32_t* n_events;
  int32_t cap_events;
  uint32_t* flags;     // bit 0: event list overflow, bit 1: geometry the reference leaves undefined
};

// how a read's alignment maps onto AlnSeq records (src/mia_main.c:259-276, src/mia.c:1376-1438)
struct RecGeom {
  int start_w, end, split, ncols_f, ncols_b;
};
MIA_HD inline RecGeom rec_geom(int as, int ae, int L) {
  RecGeom g;
  g.start_w = as;
  g.end = ae > L ? ae - L : ae;
  g.split = as > g.end;
  g.ncols_f = g.split ? (L - as) : (g.end - as + 1);
  g.ncols_b = g.split ? g.end + 1 : 0;
  return g;
}

// ---- records per read + exclusive scan -> AlnSeq slot of every read ---------------
__global__ void k_scan_blocks(ReadSet rs, int32_t L, int64_t* partial) {
  __shared__ int32_t sh[256];
  const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  int32_t s = 0;
  for (int k = 0; k < 16; k++) {
    int64_t i = base + k;
    if (i < rs.n && rs.sk[i]) s += rec_geom(rs.as[i], rs.ae[i], L).split ? 2 : 1;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void k_scan_partials(int64_t* partial, int nb, int64_t base, int64_t* total, const int32_t* abort_if = nullptr) {
  if (abort_if && *abort_if != 0) return;     // (see k_slot_count)
  // one workgroup of 256 threads: every thread a stretch of the partial sums, a scan over the 256 stretch sums in LDS
  __shared__ int64_t s_run[256];
  const int t = threadIdx.x, per = (nb + 255) / 256, b0 = t * per, b1 = b0 + per < nb ? b0 + per : nb;
  int64_t mine = 0;
  for (int b = b0; b < b1; b++) mine += partial[b];
  // inclusive scan of the 256 run sums: a shuffle scan per wavefront, the four wavefronts joined through LDS (two barriers
  // instead of sixteen)
  const int lane = t & 63, w = t >> 6;
  int64_t inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int64_t u = (int64_t)__shfl_up((long long)inc, o); if (lane >= o) inc += u; }
  if (lane == 63) s_run[w] = inc;
  __syncthreads();
  int64_t off = 0;
  for (int k = 0; k < w; k++) off += s_run[k];
  const int64_t all = s_run[0] + s_run[1] + s_run[2] + s_run[3];
  int64_t run = base + off + inc - mine;
  for (int b = b0; b < b1; b++) { const int64_t v = partial[b]; partial[b] = run; run += v; }
  if (t == 255) *total = all;
}
__global__ void k_scan_apply(ReadSet rs, int32_t L, const int64_t* partial, int64_t* slot) {
  __shared__ int32_t sh[256];
  const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  int32_t cnt[16], s = 0;
  for (int k = 0; k < 16; k++) {
    int64_t i = base + k;
    cnt[k] = (i < rs.n && rs.sk[i]) ? (rec_geom(rs.as[i], rs.ae[i], L).split ? 2 : 1) : 0;
    s += cnt[k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  // exclusive scan of the 256 thread sums (Hillis-Steele; tiny)
  for (int o = 1; o < 256; o <<= 1) {
    int32_t v = ((int)threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += v;
    __syncthreads();
  }
  int64_t run = partial[blockIdx.x] + sh[threadIdx.x] - s;
  for (int k = 0; k < 16; k++) {
    int64_t i = base + k;
    if (i < rs.n) slot[i] = run;
    run += cnt[k];
  }
}

// the neutral element of k_score_sums' accumulators; word 7 = reads still waiting for the exact kernel (a sharded
// iteration gathers these words before the host has looked at that counter), 0 without one
__global__ void k_score_sums_init(unsigned long long* sums, const int32_t* wide_count) {
  if (threadIdx.x == 0) {
    sums[0] = 0; sums[1] = 0; sums[2] = 0; sums[3] = (unsigned long long)(long long)INT32_MAX; sums[4] = (unsigned long long)(long long)INT32_MIN; sums[5] = 0; sums[6] = 0;
    sums[7] = wide_count ? (unsigned long long)*wide_count : 0;
  }
}

// ---- first pass of find_fsdb_score_cut (src/fsdb.c:269-383) on the device: the sums of length and score over the reads
// the regression uses (unique_best, score >= 2000) are sums of integers -- exact in any order -- plus the length range.
// out: {sum len, sum score, count, min len, max len}
// Two more counts a sharded run needs before the cull, taken in the same sweep: out[5] = AlnSeq records of this context
// (one per strand-known read, two if it is split at the origin: what k_scan_* will number), out[6] = links k_cull_mark
// will emit (same conditions as there: a strand-unknown read with pass-1 slots, or a read that was split once and is not
// now) -- so that the ranks can agree in ONE small all-gather that there is nothing to exchange.
__global__ __launch_bounds__(256) void k_score_sums(ReadSet rs, unsigned long long* out, int32_t L, const int64_t* back_slot,
                                                     const int64_t* front_slot0) {
  __shared__ unsigned long long sh[5][256];
  __shared__ int shl[2][256];
  unsigned long long sx = 0, sy = 0, j = 0, nrec = 0, nlink = 0;
  int lmin = INT32_MAX, lmax = INT32_MIN;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rs.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int sc = rs.score[i], ln = rs.len[i];
    if (sc >= 2000) { sx += (unsigned long long)ln; sy += (unsigned long long)sc; j++; lmin = ln < lmin ? ln : lmin; lmax = ln > lmax ? ln : lmax; }
    if (rs.sk[i]) {
      const bool split = rec_geom(rs.as[i], rs.ae[i], L).split;
      nrec += split ? 2 : 1;
      if (!split && back_slot[i] >= 0) nlink++;
    } else {
      nlink += (front_slot0[i] >= 0) + (back_slot[i] >= 0);
    }
  }
  sh[0][threadIdx.x] = sx; sh[1][threadIdx.x] = sy; sh[2][threadIdx.x] = j; sh[3][threadIdx.x] = nrec; sh[4][threadIdx.x] = nlink;
  shl[0][threadIdx.x] = lmin; shl[1][threadIdx.x] = lmax;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      for (int k = 0; k < 5; k++) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
      shl[0][threadIdx.x] = min(shl[0][threadIdx.x], shl[0][threadIdx.x + o]);
      shl[1][threadIdx.x] = max(shl[1][threadIdx.x], shl[1][threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&out[0], sh[0][0]); atomicAdd(&out[1], sh[1][0]); atomicAdd(&out[2], sh[2][0]);
    atomicAdd(&out[5], sh[3][0]); atomicAdd(&out[6], sh[4][0]);
    atomicMin(reinterpret_cast<long long*>(&out[3]), (long long)shl[0][0]);
    atomicMax(reinterpret_cast<long long*>(&out[4]), (long long)shl[1][0]);
  }
}

// ---- cull_maln_from_fsdb (src/mia.c:451-481): `dropped` is sticky per AlnSeq slot -----
//
// The reference addresses AlnSeq records through fs->front_asp / fs->back_asp.  reiterate_assembly sets back_asp only
// when the read is split at the origin and NEVER clears it (src/mia_main.c:259-276): a read that was split in an
// earlier iteration and is not any more keeps pointing at its old slot, which by now holds the record of whichever
// read was merged into that position this time.  Everything that walks the fsdb then sees that record through the
// stale pointer as well (a "link" below):
//   * cull_maln_from_fsdb lists it a second time (it is counted twice by every consensus loop and printed twice) and
//     marks it dropped when the READER's score is low (src/mia.c:469-480);
//   * pop_smp_from_FSDB treats it as the reader's back segment: the reader's own depth codes are computed with its
//     length added, and its depth codes are overwritten with the reader's geometry -- the overwrite survives iff the
//     reader comes later in the fsdb than the record's owner (src/fsdb.c:542-619 runs in fsdb order).
// All of this is reproduced: per read a persistent back slot, per iteration the list of links, per slot the last
// writer of its depth codes and the number of times it is listed.
struct RecInfo {              // per read, rebuilt every iteration by k_rec_geom
  int32_t* flen;              // asp_len of the front record (columns + inserted bases, src/fsdb.c:518-530)
  int32_t* blen;              // same for the read's own back record (0 if not split)
  int32_t* actf;              // read bases in the front record
  int32_t* trec;              // [n][16]: everything the tally needs about a read in ONE 64-byte line (see TREC_*)
};
// The tally visits the reads in bucket order, i.e. at random with respect to the per-read arrays: gathering a dozen
// fields costs a dozen cache lines per read (1.8 GB per 1 M reads).  k_rec_params writes them side by side instead.
enum { TREC_AS = 0, TREC_AE, TREC_LEN_ABR, TREC_FLAGS, TREC_REFSTART, TREC_ROFF, TREC_ACTF, TREC_SPARE, TREC_PARAMS = 8 };
constexpr int TRF_RC = 1, TRF_DF = 2, TRF_DB = 4, TRF_DIAG = 8, TRF_TOO_LONG = 16, TRF_SK = 32, TRF_ONEGAP = 64;   // ONEGAP: TREC_SPARE describes the gap
// NO_N: the read holds no N (k_bx_umax's mark, umax[i] >= 0: its bit planes say everything) -- in the record so that the tally does
// not fetch a 128-byte line per read for that one bit (the gather of record, planes and mark is 60 % of k_tally_binned's time)
constexpr int TRF_NO_N = 128;
// the 64-byte record of read i and its dropped bits, as the tally reads them: written by k_rec_params -- and, for an iteration without
// links, by k_cull_records already (rec_write is the one place that says what the words are)
__device__ __forceinline__ void rec_write(const ReadSet& rs, int64_t i, int32_t* trec, uint8_t* drop_front, uint8_t* drop_back, uint8_t df, uint8_t db, const int32_t* p,
                                          uint32_t st, int32_t actf, const int32_t* umax) {
  drop_front[i] = df;
  drop_back[i] = db;
  // the 64-byte record in four 16-byte stores (sixteen 4-byte ones, 64 bytes apart across the lanes, took twice as long)
  int4* t4 = reinterpret_cast<int4*>(trec + i * 16);
  const int fl = (rs.rc[i] ? TRF_RC : 0) | (df ? TRF_DF : 0) | (db ? TRF_DB : 0) | ((st & ST_DIAG) ? TRF_DIAG : 0) |
                 ((st & ST_TOO_LONG) ? TRF_TOO_LONG : 0) | (rs.sk[i] ? TRF_SK : 0) | ((st & ST_ONEGAP) ? TRF_ONEGAP : 0) |
                 ((umax && umax[i] >= 0) ? TRF_NO_N : 0);
  t4[0] = make_int4(rs.as[i], rs.ae[i], (int32_t)((uint32_t)rs.len[i] | ((uint32_t)(uint16_t)rs.abr[i] << 16)), fl);   // TREC_AS, _AE, _LEN_ABR, _FLAGS
  t4[1] = make_int4(rs.refstart[i], (int32_t)rs.roff[i], actf, (int32_t)(st >> 8));                                      // TREC_REFSTART, _ROFF, _ACTF, _SPARE
  t4[2] = make_int4(p[0], p[1], p[2], p[3]);                                                                              // TREC_PARAMS ..
  t4[3] = make_int4(p[4], p[5], p[6], p[7]);
}

// ... the same record as four 16-byte words in registers (k_cull_records hands a wavefront's records over through LDS: rec_store_wave)
__device__ __forceinline__ void rec_make(const ReadSet& rs, int64_t i, uint8_t* drop_front, uint8_t* drop_back, uint8_t df, uint8_t db, const int32_t* p,
                                         uint32_t st, int32_t actf, const int32_t* umax, int4* r4) {
  drop_front[i] = df;
  drop_back[i] = db;
  const int fl = (rs.rc[i] ? TRF_RC : 0) | (df ? TRF_DF : 0) | (db ? TRF_DB : 0) | ((st & ST_DIAG) ? TRF_DIAG : 0) |
                 ((st & ST_TOO_LONG) ? TRF_TOO_LONG : 0) | (rs.sk[i] ? TRF_SK : 0) | ((st & ST_ONEGAP) ? TRF_ONEGAP : 0) |
                 ((umax && umax[i] >= 0) ? TRF_NO_N : 0);
  r4[0] = make_int4(rs.as[i], rs.ae[i], (int32_t)((uint32_t)rs.len[i] | ((uint32_t)(uint16_t)rs.abr[i] << 16)), fl);
  r4[1] = make_int4(rs.refstart[i], (int32_t)rs.roff[i], actf, (int32_t)(st >> 8));
  r4[2] = make_int4(p[0], p[1], p[2], p[3]);
  r4[3] = make_int4(p[4], p[5], p[6], p[7]);
}
// The records of the 64 reads i0 .. i0 + 63 of a wavefront, 4 KB in a row: a lane storing its own record writes four 16-byte pieces 64
// bytes apart from its neighbours' -- every store instruction touches 64 lines and fills a quarter of each.  Through LDS (buf: this
// wavefront's 256 words of 16 bytes) every store instruction writes 1 KB without a hole.  Every lane of the wavefront calls this.
__device__ __forceinline__ void rec_store_wave(int32_t* trec, int64_t i0, int64_t n, const int4* r4, int lane, int4* buf) {
#pragma unroll
  for (int q = 0; q < 4; q++) buf[lane * 4 + q] = r4[q];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int4* out = reinterpret_cast<int4*>(trec) + i0 * 4;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int w = k * 64 + lane;
    if (i0 + (w >> 2) < n) out[w] = buf[w];
  }
}

struct SlotInfo {             // per local AlnSeq slot (global slot - slot_base)
  int64_t base;               // first global slot of this context
  const int64_t* n_local_p;   // slots owned by this context in this iteration (device: the scan's total)
  int32_t* reclen;            // asp_len of the record in the slot
  int32_t* recact;            // read bases in that record
  unsigned long long* writer; // (global read index << 20 | link index) of the last pop_smp writer; owner: link index 0xFFFFF
  int32_t* mult;              // times the slot is listed in the culled maln
};
constexpr unsigned long long LINK_NONE = 0xFFFFFull;
// kinds of link: 0 = back_asp of a read that is no longer split; 1 / 2 = front_asp / back_asp of a read whose strand is
// unknown (pass-1 score exactly 2000): reiterate_assembly skips it (src/mia_main.c:178), so BOTH its pointers stay on the
// pass-1 slots for good
struct Links {                // [cap][4] int64: reader (global read index), slot (global), flen << 32 | actf, low score flag | kind << 8
  int64_t* rec;
  int32_t* n;
  int32_t cap;
};

// inserted bases per record and bases in the front: one read per wavefront (the script walk of the tally's pass A)
// A block takes 256 reads.  Proven-diagonal reads (ST_DIAG, the large majority) need no script: one thread each.  The
// others are walked by the block's four wavefronts, 64 script rows at a time (ballot counts).
__device__ __forceinline__ void rec_store_at(int64_t i, const RecGeom& g, int nf, int nb, int af, int bases, int64_t slot_i, RecInfo& ri,
                                             SlotInfo& si, int64_t read_base, uint32_t* flags) {
  const int flen = g.ncols_f + nf, blen = g.split ? g.ncols_b + nb : 0;
  ri.flen[i] = flen; ri.blen[i] = blen; ri.actf[i] = af + nf;
  const int64_t ls = slot_i - si.base;
  if (ls >= 0 && ls + (g.split ? 1 : 0) < (*si.n_local_p)) {
    const unsigned long long me = ((unsigned long long)(read_base + i) << 20) | LINK_NONE;
    si.reclen[ls] = flen; si.recact[ls] = af + nf; si.mult[ls] = 1; si.writer[ls] = me;
    if (g.split) { si.reclen[ls + 1] = blen; si.recact[ls + 1] = bases - (af + nf); si.mult[ls + 1] = 1; si.writer[ls + 1] = me; }
  } else atomicOr(flags, 4u);
}
__device__ __forceinline__ void rec_store(int64_t i, const RecGeom& g, int nf, int nb, int af, int bases, const int64_t* slot, RecInfo& ri,
                                          SlotInfo& si, int64_t read_base, uint32_t* flags) {
  rec_store_at(i, g, nf, nb, af, bases, slot[i], ri, si, read_base, flags);
}

__global__ __launch_bounds__(256) void k_rec_geom(ReadSet rs, int32_t L, const int64_t* slot, RecInfo ri, SlotInfo si, int64_t read_base,
                                                    uint32_t* flags) {
  const int lane = threadIdx.x & 63;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63);     // first read of this wavefront
  const int64_t me = i0 + lane;
  bool walk = false;
  if (me < rs.n) {
    if (!rs.sk[me]) { ri.flen[me] = 0; ri.blen[me] = 0; ri.actf[me] = 0; }
    else if (rs.status[me] & ST_DIAG) {
      // one column per read base from the alignment start on: nothing to count in the script
      const int len2 = rs.len[me], abr = rs.abr[me];
      const RecGeom g = rec_geom(rs.as[me], rs.ae[me], L);
      const int af = (len2 - abr) < g.ncols_f ? (len2 - abr) : g.ncols_f;
      rec_store(me, g, 0, 0, af, len2 - abr, slot, ri, si, read_base, flags);
    } else if ((rs.status[me] & ST_ONEGAP) && !rec_geom(rs.as[me], rs.ae[me], L).split) {
      // k_band_align's reads with one gap, one record: inserted bases and aligned bases follow from the gap's description
      const uint32_t desc = rs.status[me] >> 8;
      const int len2 = rs.len[me], abr = rs.abr[me], gn = (int)((desc >> 10) & 63u), ins = (int)(desc & 1u);
      const RecGeom g = rec_geom(rs.as[me], rs.ae[me], L);
      rec_store(me, g, ins ? gn : 0, 0, len2 - abr - (ins ? gn : 0), len2 - abr, slot, ri, si, read_base, flags);
    } else walk = true;
  }
  unsigned long long todo = __ballot(walk);
  while (todo) {
    const int64_t i = i0 + __builtin_ctzll(todo);
    todo &= todo - 1;
    const int len2 = rs.len[i], abr = rs.abr[i];
    const RecGeom g = rec_geom(rs.as[i], rs.ae[i], L);
    const int16_t* cols = rs.cols + (int64_t)i * rs.stride;
    const int cbase = rs.refstart[i] - g.start_w;
    int nf = 0, nb = 0, af = 0;
    // on the way: does the script hold exactly one gap?  (One run of inserted rows, or one jump over reference columns
    // between two aligned rows.)  Then the tally can take the read without walking the script again (ST_ONEGAP).
    int n_ev = 0, ins_rows = 0, ev_row = 0, ev_len = 0, ev_kind = 0;
    for (int r0 = abr; r0 < len2; r0 += 64) {
      const int r = r0 + lane;
      bool isF = false, isB = false, alF = false, ins_row = false, ins_start = false, jump = false;
      int jump_len = 0;
      if (r < len2) {
        const int cp = r > abr ? (int)cols[r - 1] : -3;
        if (cols[r] == COL_INSERT) {
          int rn = r + 1;
          while (rn < len2 && cols[rn] < 0) rn++;
          const int o = cbase + cols[rn];
          if (o < g.ncols_f) isF = true; else if (g.split && o - g.ncols_f < g.ncols_b) isB = true;
          ins_row = true;
          ins_start = cp != COL_INSERT;
        } else if (cols[r] >= 0) {
          alF = (cbase + cols[r]) < g.ncols_f;
          if (cp >= 0 && cols[r] - cp > 1) { jump = true; jump_len = cols[r] - cp - 1; }
        }
      }
      nf += __popcll(__ballot(isF));
      nb += __popcll(__ballot(isB));
      af += __popcll(__ballot(alF));
      const unsigned long long m_ins = __ballot(ins_start), m_del = __ballot(jump);
      ins_rows += __popcll(__ballot(ins_row));
      if (n_ev == 0 && (m_ins | m_del)) {
        const int l = __builtin_ctzll(m_ins | m_del);
        ev_row = r0 + l;
        ev_kind = (int)((m_ins >> l) & 1ull);
        ev_len = __shfl(jump_len, l);
      }
      n_ev += __popcll(m_ins) + __popcll(m_del);
    }
    if (lane == 0) {
      const uint32_t st = rs.status[i];
      if (!(st & (ST_ESCAPE | ST_TOO_LONG | ST_SKIPPED | ST_BAND | ST_DIAG))) {
        const int gn = ev_kind ? ins_rows : ev_len, n_al = len2 - abr;
        // (the columns must be what one gap explains: an insert followed by a jump is left to the script walk)
        const bool one = n_ev == 1 && !g.split && gn > 0 && gn < 64 && ev_row < 512 && g.ncols_f == (ev_kind ? n_al - gn : n_al + gn);
        const uint32_t want = one ? (ST_ONEGAP | (((uint32_t)ev_kind | ((uint32_t)ev_row << 1) | ((uint32_t)gn << 10)) << 8)) : ST_OK;
        if (st != want) rs.status[i] = want;        // (k_band_align's own description of its one gap is the same word)
      }
    }
    if (lane == 0) rec_store(i, g, nf, nb, af, len2 - abr, slot, ri, si, read_base, flags);
  }
}

// ---- the cull's first half as two launches (mia_hip_cull; k_scan_blocks / _partials / _apply, k_rec_geom and k_cull_mark
// above remain for the entry points that need one of them alone) -----------------------------------------------------------
// k_slot_count: AlnSeq records per stretch of 256 reads (a workgroup takes sixteen stretches one after the other: one
// arrival per 4 096 reads -- a counter every block of 256 adds to is served one add at a time, 80 us for a million reads);
// the workgroup that arrives last turns the counts into exclusive offsets (+ base) and leaves the total in *total.
// nb: stretches.  *blocks_done must be 0 on entry and is 0 again on exit.
__global__ __launch_bounds__(256) void k_slot_count(ReadSet rs, int32_t L, int64_t* partial, int nb, int64_t base, int64_t* total, uint32_t* blocks_done, const int32_t* abort_if = nullptr) {
  if (abort_if && *abort_if != 0) return;     // (mia_hip_iterate queued this launch before the alignment's exact-kernel count was known: see iterate_body)
  __shared__ bool last;
  __shared__ int64_t s_run[256];
  __shared__ int32_t wsum[16][4];
  // stretch k of this workgroup = reads base + 256 k .. + 255, one per thread (neighbouring threads, neighbouring reads)
  const int64_t base_read = (int64_t)blockIdx.x * 4096;
  // (all sixteen stretches' loads first: one after the other, each waited for, they were 25 of this kernel's 40 us)
  uint8_t sk16[16];
  int32_t as16[16], ae16[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int64_t i = base_read + k * 256 + threadIdx.x;
    const bool in = i < rs.n;
    sk16[k] = in ? rs.sk[i] : (uint8_t)0;
    as16[k] = in ? rs.as[i] : 0;
    ae16[k] = in ? rs.ae[i] : 0;
  }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int cnt = sk16[k] ? (rec_geom(as16[k], ae16[k], L).split ? 2 : 1) : 0;
    const int w = __popcll(__ballot(cnt >= 1)) + __popcll(__ballot(cnt == 2));
    if ((threadIdx.x & 63) == 0) wsum[k][threadIdx.x >> 6] = w;
  }
  __syncthreads();
  if (threadIdx.x < 16 && blockIdx.x * 16 + (int)threadIdx.x < nb)
    partial[blockIdx.x * 16 + threadIdx.x] = (int64_t)wsum[threadIdx.x][0] + wsum[threadIdx.x][1] + wsum[threadIdx.x][2] + wsum[threadIdx.x][3];
  if (!blocks_done) {
    // the counts alone, and behind them (partial[nb + workgroup]) this workgroup's sum: k_cull_records adds up what lies before
    // each of its blocks itself (raw_nb) -- the arrival below, with its device-scope fence per workgroup, was 25 of this
    // kernel's 40 us, and a single-workgroup scan between the two kernels (k_scan_partials) another 15
    if (threadIdx.x == 0) {
      int64_t g = 0;
      for (int k = 0; k < 16; k++) g += (int64_t)wsum[k][0] + wsum[k][1] + wsum[k][2] + wsum[k][3];
      partial[nb + blockIdx.x] = g;
    }
    return;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(blocks_done, 1u) == (uint32_t)gridDim.x - 1u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  // (k_scan_partials: every thread a run of the partial sums, a scan over the 256 run sums in LDS)
  const int t = threadIdx.x, per = (nb + 255) / 256, b0 = t * per, b1 = b0 + per < nb ? b0 + per : nb;
  int64_t mine = 0;
  for (int b2 = b0; b2 < b1; b2++) mine += __builtin_nontemporal_load(partial + b2);
  s_run[t] = mine;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int64_t a2 = t >= o ? s_run[t - o] : 0;
    __syncthreads();
    s_run[t] += a2;
    __syncthreads();
  }
  int64_t run = base + s_run[t] - mine;
  for (int b2 = b0; b2 < b1; b2++) { const int64_t v = __builtin_nontemporal_load(partial + b2); partial[b2] = run; run += v; }
  if (t == 255) *total = s_run[255];
  if (t == 0) *blocks_done = 0;
}

// k_cull_records: slot numbers (k_scan_apply), record geometry (k_rec_geom) and the reads' own dropped marks / back slots /
// links (k_cull_mark) of a block of 256 reads in one go: what the three kernels hand each other per read stays in registers.
__global__ __launch_bounds__(256) void k_cull_records(ReadSet rs, int32_t L, const int64_t* partial, int64_t* slot, RecInfo ri, SlotInfo si, int64_t read_base,
                                                        uint32_t* flags, uint8_t* slot_dropped, int64_t n_slots, int32_t hard_cut, double slope, double intercept,
                                                        int64_t* back_slot, const int64_t* front_slot0, Links lk, const double* dev_cut, const int32_t* abort_if = nullptr,
                                                        int32_t raw_nb = 0, int64_t slot_base = 0, int64_t* total_out = nullptr,
                                                        int32_t with_records = 0, uint8_t* drop_front = nullptr, uint8_t* drop_back = nullptr, const int32_t* umax = nullptr) {
  if (abort_if && *abort_if != 0) return;     // (mia_hip_iterate queued this launch before the alignment's exact-kernel count was known: see iterate_body)
  __shared__ int32_t wsum[4];
  __shared__ int64_t s_red[2][4], s_first, s_total;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (raw_nb) {
    // partial[] holds k_slot_count's counts as they are -- per block of 256 reads, and from [raw_nb] on per sixteen blocks:
    // the first slot of this block = slot_base + the sixteens before its own + the blocks before it in its sixteen
    const int ng = (raw_nb + 15) >> 4, myg = (int)(blockIdx.x >> 4);
    int64_t pre = 0, tot = 0;
    for (int g = threadIdx.x; g < ng; g += 256) { const int64_t v = partial[raw_nb + g]; tot += v; if (g < myg) pre += v; }
    if (threadIdx.x < 16) { const int b = myg * 16 + (int)threadIdx.x; if (b < (int)blockIdx.x) pre += partial[b]; }
#pragma unroll
    for (int o = 32; o; o >>= 1) { pre += __shfl_xor(pre, o); tot += __shfl_xor(tot, o); }
    if (lane == 0) { s_red[0][wv] = pre; s_red[1][wv] = tot; }
    __syncthreads();
    if (threadIdx.x == 0) {
      s_first = slot_base + s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
      s_total = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
      if (blockIdx.x == 0 && total_out) *total_out = s_total;
    }
    si.n_local_p = &s_total;                  // (rec_store_at reads the total through the pointer: this block's own sum, not a word another block is writing)
  }
  const int64_t i0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63);     // first read of this wavefront
  const int64_t me = i0 + lane;
  const bool in = me < rs.n;
  const bool sk = in && rs.sk[me];
  int as_me = 0, ae_me = 0;
  if (sk) { as_me = rs.as[me]; ae_me = rs.ae[me]; }
  const RecGeom g_me = rec_geom(as_me, ae_me, L);
  // ---- slot of every read: exclusive scan of the record counts (1, or 2 when split at the origin) over the block
  const int cnt = sk ? (g_me.split ? 2 : 1) : 0;
  const unsigned long long m1 = __ballot(cnt >= 1), m2 = __ballot(cnt == 2), below = (1ull << lane) - 1ull;
  if (lane == 0) wsum[wv] = __popcll(m1) + __popcll(m2);
  __syncthreads();
  int64_t my_slot = (raw_nb ? s_first : partial[blockIdx.x]) + __popcll(m1 & below) + __popcll(m2 & below);
  for (int k = 0; k < wv; k++) my_slot += wsum[k];
  if (in) slot[me] = my_slot;
  // ---- record geometry (k_rec_geom)
  int my_flen = 0, my_actf = 0, my_blen = 0;
  uint32_t my_st = in ? rs.status[me] : 0u;          // (the walk below may rewrite a read's status: its own lane keeps the new one)
  bool walk = false;
  if (in) {
    if (!sk) { ri.flen[me] = 0; ri.blen[me] = 0; ri.actf[me] = 0; }
    else if (rs.status[me] & ST_DIAG) {
      const int len2 = rs.len[me], abr = rs.abr[me];
      const int af = (len2 - abr) < g_me.ncols_f ? (len2 - abr) : g_me.ncols_f;
      rec_store_at(me, g_me, 0, 0, af, len2 - abr, my_slot, ri, si, read_base, flags);
      my_flen = g_me.ncols_f; my_actf = af; my_blen = g_me.split ? g_me.ncols_b : 0;
    } else if ((rs.status[me] & ST_ONEGAP) && !g_me.split) {
      const uint32_t desc = rs.status[me] >> 8;
      const int len2 = rs.len[me], abr = rs.abr[me], gn = (int)((desc >> 10) & 63u), ins = (int)(desc & 1u);
      rec_store_at(me, g_me, ins ? gn : 0, 0, len2 - abr - (ins ? gn : 0), len2 - abr, my_slot, ri, si, read_base, flags);
      my_flen = g_me.ncols_f + (ins ? gn : 0); my_actf = len2 - abr;
    } else walk = true;
  }
  unsigned long long todo = __ballot(walk);
  while (todo) {
    const int li = __builtin_ctzll(todo);
    const int64_t i = i0 + li;
    todo &= todo - 1;
    const int len2 = rs.len[i], abr = rs.abr[i];
    const RecGeom g = rec_geom(rs.as[i], rs.ae[i], L);
    const int16_t* cols = rs.cols + (int64_t)i * rs.stride;
    const int cbase = rs.refstart[i] - g.start_w;
    int nf = 0, nb = 0, af = 0;
    int n_ev = 0, ins_rows = 0, ev_row = 0, ev_len = 0, ev_kind = 0;
    for (int r0 = abr; r0 < len2; r0 += 64) {
      const int r = r0 + lane;
      bool isF = false, isB = false, alF = false, ins_row = false, ins_start = false, jump = false;
      int jump_len = 0;
      if (r < len2) {
        const int cp = r > abr ? (int)cols[r - 1] : -3;
        if (cols[r] == COL_INSERT) {
          int rn = r + 1;
          while (rn < len2 && cols[rn] < 0) rn++;
          const int o = cbase + cols[rn];
          if (o < g.ncols_f) isF = true; else if (g.split && o - g.ncols_f < g.ncols_b) isB = true;
          ins_row = true;
          ins_start = cp != COL_INSERT;
        } else if (cols[r] >= 0) {
          alF = (cbase + cols[r]) < g.ncols_f;
          if (cp >= 0 && cols[r] - cp > 1) { jump = true; jump_len = cols[r] - cp - 1; }
        }
      }
      nf += __popcll(__ballot(isF));
      nb += __popcll(__ballot(isB));
      af += __popcll(__ballot(alF));
      const unsigned long long m_ins = __ballot(ins_start), m_del = __ballot(jump);
      ins_rows += __popcll(__ballot(ins_row));
      if (n_ev == 0 && (m_ins | m_del)) {
        const int l = __builtin_ctzll(m_ins | m_del);
        ev_row = r0 + l;
        ev_kind = (int)((m_ins >> l) & 1ull);
        ev_len = __shfl(jump_len, l);
      }
      n_ev += __popcll(m_ins) + __popcll(m_del);
    }
    const int64_t slot_i = __shfl(my_slot, li);
    uint32_t st_new = rs.status[i];                        // (wave-uniform, like everything the walk counted)
    if (!(st_new & (ST_ESCAPE | ST_TOO_LONG | ST_SKIPPED | ST_BAND | ST_DIAG))) {
      const int gn = ev_kind ? ins_rows : ev_len, n_al = len2 - abr;
      const bool one = n_ev == 1 && !g.split && gn > 0 && gn < 64 && ev_row < 512 && g.ncols_f == (ev_kind ? n_al - gn : n_al + gn);
      const uint32_t want = one ? (ST_ONEGAP | (((uint32_t)ev_kind | ((uint32_t)ev_row << 1) | ((uint32_t)gn << 10)) << 8)) : ST_OK;
      if (st_new != want) { if (lane == 0) rs.status[i] = want; st_new = want; }
    }
    if (lane == 0) rec_store_at(i, g, nf, nb, af, len2 - abr, slot_i, ri, si, read_base, flags);
    if (lane == li) { my_flen = g.ncols_f + nf; my_actf = af + nf; my_blen = g.split ? g.ncols_b + nb : 0; my_st = st_new; }        // (the counts are wave-uniform: the read's own lane keeps them for its link and its record)
  }
  // ---- own dropped marks, the persistent back slot, the links of formerly split reads (k_cull_mark)
  __shared__ int4 s_rec[4][256];              // (rec_store_wave: a wavefront's 64 records)
  int4 r4[4] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
  if (in) {
  if (dev_cut) { slope = dev_cut[0]; intercept = dev_cut[1]; }
  const double min_score = hard_cut > 0 ? (double)hard_cut : (double)(intercept + (slope * (double)rs.len[me]));
  const bool low = (double)rs.score[me] < min_score;
  auto add_link = [&](int64_t target, int kind, int fl, int ac) {
    const int e = atomicAdd(lk.n, 1);
    if (e < lk.cap) {
      int64_t* r = lk.rec + (int64_t)e * 4;
      r[0] = read_base + me; r[1] = target;
      r[2] = ((int64_t)fl << 32) | (uint32_t)ac;
      r[3] = (low ? 1 : 0) | (kind << 8);
    } else atomicOr(flags, 8u);
  };
  int32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (!sk) {
    if (front_slot0[me] >= 0) add_link(front_slot0[me], 1, 0, 0);
    if (back_slot[me] >= 0) add_link(back_slot[me], 2, 0, 0);
    if (with_records) rec_make(rs, me, drop_front, drop_back, 0, 0, p, my_st, 0, umax, r4);
  } else {
  // WITH_RECORDS (round 5): the read's tally record and dropped bits as k_rec_params would write them if no link existed -- its own slot(s),
  // its own depth-code parameters, listed once.  k_rec_params then returns at once unless the iteration has a link (a formerly split read, a
  // strand-unknown read with pass-1 slots): it read every per-read array again to write 64 bytes per read that are known here.
  uint8_t df = 0, db = 0;
  if (with_records && my_slot < n_slots) df = low ? (uint8_t)1 : slot_dropped[my_slot];
  if (with_records && g_me.split && my_slot + 1 < n_slots) db = low ? (uint8_t)1 : slot_dropped[my_slot + 1];
  if (low && my_slot < n_slots) slot_dropped[my_slot] = 1;
  if (g_me.split) {
    if (low && my_slot + 1 < n_slots) slot_dropped[my_slot + 1] = 1;
    back_slot[me] = my_slot + 1;
  } else if (back_slot[me] >= 0) add_link(back_slot[me], 0, my_flen, my_actf);
  if (with_records) {
    const int64_t ls = my_slot - si.base, n_local = *si.n_local_p;
    if (ls >= 0 && ls < n_local) { p[2] = my_flen + my_blen; p[3] = 1; }
    if (g_me.split && ls + 1 < n_local) { p[4] = my_flen; p[5] = my_actf; p[6] = my_flen + my_blen; p[7] = 1; }
    rec_make(rs, me, drop_front, drop_back, df, db, p, my_st, my_actf, umax, r4);
  }
  }
  }
  if (with_records) rec_store_wave(ri.trec, i0, rs.n, r4, lane, s_rec[wv]);
}

// own dropped marks, the persistent back slot, and the links of formerly split reads
__global__ void k_cull_mark(ReadSet rs, int32_t L, const int64_t* slot, uint8_t* slot_dropped, int64_t n_slots, int32_t hard_cut, double slope,
                            double intercept, int64_t* back_slot, const int64_t* front_slot0, RecInfo ri, Links lk, int64_t read_base,
                            uint32_t* flags, const double* dev_cut) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rs.n) return;
  if (dev_cut) { slope = dev_cut[0]; intercept = dev_cut[1]; }   // the line of find_fsdb_score_cut, left on the device
  const double min_score = hard_cut > 0 ? (double)hard_cut : (double)(intercept + (slope * (double)rs.len[i]));
  const bool low = (double)rs.score[i] < min_score;
  auto add_link = [&](int64_t target, int kind, int fl, int ac) {
    const int e = atomicAdd(lk.n, 1);
    if (e < lk.cap) {
      int64_t* r = lk.rec + (int64_t)e * 4;
      r[0] = read_base + i; r[1] = target;
      r[2] = ((int64_t)fl << 32) | (uint32_t)ac;
      r[3] = (low ? 1 : 0) | (kind << 8);
    } else atomicOr(flags, 8u);
  };
  if (!rs.sk[i]) {                                             // strand unknown: both pointers are the pass-1 ones
    if (front_slot0[i] >= 0) add_link(front_slot0[i], 1, 0, 0);
    if (back_slot[i] >= 0) add_link(back_slot[i], 2, 0, 0);
    return;
  }
  const bool split = rec_geom(rs.as[i], rs.ae[i], L).split;
  const int64_t s = slot[i];
  if (low && s < n_slots) slot_dropped[s] = 1;
  if (split) {
    if (low && s + 1 < n_slots) slot_dropped[s + 1] = 1;
    back_slot[i] = s + 1;                                      // fs->back_asp (src/mia_main.c:266-267)
  } else if (back_slot[i] >= 0) add_link(back_slot[i], 0, ri.flen[i], ri.actf[i]);   // stale back_asp
}

// every link (of this context or gathered from the others): effects on the slot it points at, if that slot is ours
__global__ void k_links_apply(const int64_t* links, const int32_t* n_links_p, int32_t cap, SlotInfo si, uint8_t* slot_dropped, int64_t n_slots,
                              int32_t* link_len, int32_t* link_act, uint32_t* flags, const int32_t* abort_if = nullptr) {
  if (abort_if && *abort_if != 0) return;     // (mia_hip_iterate queued this launch before the alignment's exact-kernel count was known: see iterate_body)
  const int n_links = min(*n_links_p, cap);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_links; e += gridDim.x * blockDim.x) {
    const int64_t* r = links + (int64_t)e * 4;
    const int64_t ls = r[1] - si.base;
    link_len[e] = -1;                                            // -1: the slot is not ours
    link_act[e] = -1;
    if (ls < 0 || ls >= (*si.n_local_p)) continue;                // another context's slot (or none: flagged by k_rec_params)
    atomicMax(&si.writer[ls], ((unsigned long long)r[0] << 20) | (unsigned long long)e);
    atomicAdd(&si.mult[ls], 1);
    if ((r[3] & 1) && r[1] < n_slots) slot_dropped[r[1]] = 1;
    link_len[e] = si.reclen[ls];
    link_act[e] = si.recact[ls];
  }
}

// depth-code parameters and multiplicities of every read's records + its dropped bits, after all links are in
// ---- the early tally (round 4) ------------------------------------------------------------------------------------------
// Four reads in five are finished by the PLAN (k_bx_plan: a proven gap-free diagonal) and do not change while the band DPs work
// on the rest -- a quarter of a millisecond per million reads during which the tally's LDS pipes have nothing to do.  Their tally
// runs then, beside the DPs, on what is known of them at that point: end points, strand, bases.  What is NOT known yet is what the
// cull decides (it needs every read's records: AlnSeq slots are numbered across all reads, `dropped` lives in the slot) -- the
// read's dropped marks, and the parameters of formerly split or doubly listed records.  The early tally ASSUMES the ordinary
// case (not dropped, listed once, its own depth codes: the record k_rec_early writes); k_rec_params, which knows, puts every
// early read whose true record differs on a list, and k_tally_fix takes the assumed contribution off again and adds the true
// one (integer sums: exact).  k_bx_plan marks the reads it finishes in early[].
__device__ __forceinline__ void rec_default_params(const RecGeom& g, int n_al, int32_t* p, int* actf) {
  const int flen = g.ncols_f, blen = g.split ? g.ncols_b : 0;          // a gap-free read: a record's asp_len is its columns
  *actf = g.split ? g.ncols_f : n_al;
  p[0] = 0; p[1] = 0; p[2] = flen + blen; p[3] = 1;
  p[4] = g.split ? flen : 0; p[5] = g.split ? *actf : 0; p[6] = g.split ? flen + blen : 0; p[7] = g.split ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_rec_early(ReadSet rs, int32_t L, const uint8_t* early, int32_t* trec, const int32_t* umax = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rs.n || !early[i]) return;
  const RecGeom g = rec_geom(rs.as[i], rs.ae[i], L);
  int32_t p[8];
  int actf;
  rec_default_params(g, (int)rs.len[i], p, &actf);
  int4* t4 = reinterpret_cast<int4*>(trec + i * 16);
  t4[0] = make_int4(rs.as[i], rs.ae[i], (int32_t)(uint32_t)rs.len[i], (rs.rc[i] ? TRF_RC : 0) | TRF_DIAG | TRF_SK | ((umax && umax[i] >= 0) ? TRF_NO_N : 0));      // (abr = 0: the plan's reads start in row 0)
  t4[1] = make_int4(rs.refstart[i], (int32_t)rs.roff[i], actf, 0);
  t4[2] = make_int4(p[0], p[1], p[2], p[3]);
  t4[3] = make_int4(p[4], p[5], p[6], p[7]);
}

// skip_if_no_links: k_cull_records has written every record already (its with_records argument) -- they stand unless a link exists:
// a link changes the dropped bit, the multiplicity or the depth-code parameters of the slot it points at, and the record of the
// read it comes from (its back length)
__global__ void k_rec_params(ReadSet rs, int32_t L, const int64_t* slot, const uint8_t* slot_dropped, int64_t n_slots, const int64_t* back_slot,
                             RecInfo ri, SlotInfo si, const int64_t* links, const int32_t* link_len, const int32_t* link_act, const int32_t* n_links_p,
                             int32_t cap, int64_t read_base,
                             uint8_t* drop_front, uint8_t* drop_back, uint32_t* flags, const int32_t* abort_if = nullptr,
                             const uint8_t* early = nullptr, int32_t* fix_list = nullptr, int32_t* n_fix = nullptr, const int32_t* umax = nullptr,
                             int32_t skip_if_no_links = 0) {
  if (abort_if && *abort_if != 0) return;     // (mia_hip_iterate queued this launch before the alignment's exact-kernel count was known: see iterate_body)
  if (skip_if_no_links && *n_links_p == 0) return;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rs.n) return;
  const int n_links = min(*n_links_p, cap);
  uint8_t df = 0, db = 0;
  int32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // front {dffBase, actOff, B, mult}, back {...}: words 8..15 of the read's record below
  if (rs.sk[i]) {
    const bool split = rec_geom(rs.as[i], rs.ae[i], L).split;
    const int64_t s = slot[i], ls = s - si.base;
    if (s < n_slots) df = slot_dropped[s];
    if (split && s + 1 < n_slots) db = slot_dropped[s + 1];
    const int flen = ri.flen[i], blen = ri.blen[i], actf = ri.actf[i];
    int back_len = blen;                                       // what pop_smp adds to the front as "the back segment"
    if (!split && back_slot[i] >= 0) {
      back_len = -1;
      for (int e = 0; e < n_links; e++)                        // (a handful of links; only formerly split reads get here)
        if (links[(int64_t)e * 4] == read_base + i && (links[(int64_t)e * 4 + 3] >> 8) == 0) { back_len = link_len[e]; break; }
      if (back_len < 0) { atomicOr(flags, 4u); back_len = 0; }
    }
    auto fill = [&](int32_t* q, int64_t lsl, int own_base, int own_off, int own_B, int own_len) {
      const unsigned long long w = si.writer[lsl];
      if ((w & LINK_NONE) == LINK_NONE) { q[0] = own_base; q[1] = own_off; q[2] = own_B; }
      else {                                                   // overwritten by a later reader through its stale pointer
        const int we = (int)(w & LINK_NONE);
        const int64_t* r = links + (int64_t)we * 4;
        const int kind = (int)(r[3] >> 8);
        if (kind == 0) {                                       // back segment of a formerly split read
          const int rf = (int)(r[2] >> 32), ra = (int)(uint32_t)r[2];
          q[0] = rf; q[1] = ra; q[2] = rf + own_len;
        } else {                                               // front / back segment of a strand-unknown read: its other
          int sib_len = 0, sib_act = 0;                        // segment is whatever its other pointer addresses
          for (int e = 0; e < n_links; e++)
            if (links[(int64_t)e * 4] == r[0] && (links[(int64_t)e * 4 + 3] >> 8) == 3 - kind) {
              sib_len = link_len[e]; sib_act = link_act[e];
              if (sib_len < 0) { atomicOr(flags, 4u); sib_len = 0; sib_act = 0; }
              break;
            }
          if (kind == 1) { q[0] = 0; q[1] = 0; q[2] = own_len + sib_len; }
          else { q[0] = sib_len; q[1] = sib_act; q[2] = sib_len + own_len; }
        }
      }
      q[3] = si.mult[lsl];
    };
    if (ls >= 0 && ls < (*si.n_local_p)) fill(p, ls, 0, 0, flen + back_len, flen);
    if (split && ls + 1 < (*si.n_local_p)) fill(p + 4, ls + 1, flen, actf, flen + blen, blen);
  }
  const uint32_t st = rs.status[i];
  rec_write(rs, i, ri.trec, drop_front, drop_back, df, db, p, st, ri.actf[i], umax);
  if (early && early[i]) {
    // tallied already, on the ordinary record (k_rec_early): does the true one say the same?
    int32_t q[8];
    int actf0;
    rec_default_params(rec_geom(rs.as[i], rs.ae[i], L), (int)rs.len[i], q, &actf0);
    bool same = df == 0 && db == 0 && rs.sk[i] && ri.actf[i] == actf0 && (st & ST_DIAG) && !(st & ST_TOO_LONG) && rs.abr[i] == 0;
    for (int k = 0; k < 8; k++) same = same && p[k] == q[k];
    if (!same) fix_list[atomicAdd(n_fix, 1)] = (int32_t)i;
  }
}

// ---- tally: one read per wavefront, one read row per lane (4 passes for 256-base reads) ----
__device__ __forceinline__ int depth_code(int dff, int dfb) {   // src/fsdb.c:572-582
  return dff <= PSSM_DEPTH ? dff : (dfb < PSSM_DEPTH ? 2 * PSSM_DEPTH - dfb : PSSM_DEPTH);
}

// BINNED: the reads of one workgroup start inside one 256-column bucket of the reference
// (k_bucket_* below), so their columns fall into a TALLY_WIN-column window that is tallied in
// LDS and flushed once; anything outside the window (the back part of a read that wraps
// around the origin) takes the global atomic.  Same integer sums either way.
constexpr int TALLY_EV_CAP = 256;    // insert events a workgroup buffers in LDS
constexpr int TALLY_CHUNK_LATE = 256;  // reads per workgroup of the tally's second half when the plan's reads went through the early tally
constexpr int TALLY_CHUNK_LINEAR = 256;  // ... with the linear matrix: measured best of 256 / 512 / 768 (0.899 / 0.906 / 0.934 ms per million-read step)
constexpr int TALLY_BUCKET = 128, TALLY_WIN = 384, TALLY_CHUNK = 512;   // 20 KB of LDS per workgroup: 8 workgroups (32 waves) per CU

// The LDS window of a workgroup is circular: slot k holds column win_base + k, and past the end of the reference the
// columns 0, 1, ... (the back records of reads over the origin, which all start in the last bucket).
__device__ __forceinline__ int tally_slot(int gc, int win_base, int Lp) {
  const int wc = gc - win_base;
  return wc < 0 ? wc + Lp : wc;
}

template <bool BINNED>
__device__ __forceinline__ void tally_one_read(int64_t i, int lane, const ReadSet& rs, const RefInfo& ref, const int32_t* pssm2,
                                               const uint8_t* drop_front, const uint8_t* drop_back, const TallyBuf& tb,
                                               int32_t* lds, int win_base, const int32_t* rec_params, const int32_t* rec_actf,
                                               const int16_t* pssm_lds, unsigned long long* ev_buf, int* ev_cnt, int32_t* n_cnt = nullptr,
                                               const int32_t* rec_lds = nullptr) {
  // n_cnt != nullptr (binned tally, matrix independent of depth and strand): the four score words of a column are a
  // linear function of its base counts; the LDS window then only counts (N in n_cnt) and the flush derives the scores.
  // rec_lds != nullptr: the read's 64-byte record, staged in LDS by the lane that already holds it (a scalar load
  // from global memory here costs a microsecond per read, and these reads are taken one after the other).
  const int32_t* tr = rec_lds ? rec_lds : rec_params + i * 16;
  const int flags = tr[TREC_FLAGS];
  if (!(flags & TRF_SK)) return;
  if (flags & TRF_TOO_LONG) { if (lane == 0) atomicOr(tb.flags, 2u); return; }
  const int L = ref.L, Lp = tb.Lp;
  const int len2 = tr[TREC_LEN_ABR] & 0xFFFF, abr = (int)(int16_t)((uint32_t)tr[TREC_LEN_ABR] >> 16);
  const RecGeom g = rec_geom(tr[TREC_AS], tr[TREC_AE], L);
  const int16_t* cols = rs.cols + (int64_t)i * rs.stride;
  const int cbase = tr[TREC_REFSTART] - g.start_w;     // path offset of window column 0
  const uint8_t* rp = rs.packed + (uint32_t)tr[TREC_ROFF];
  const bool is_rc = (flags & TRF_RC) != 0;
  const int pmo = is_rc ? PSSM_WORDS : 0;                   // src/mia.c:584-589
  const int32_t* pm = pssm2 + pmo;
  const bool dF = (flags & TRF_DF) != 0, dB = (flags & TRF_DB) != 0;
  const bool diag = (flags & TRF_DIAG) != 0;                 // proven pure diagonal: the script is not needed
  if (g.split && g.start_w >= L) { if (lane == 0) atomicOr(tb.flags, 2u); return; }  // split_pwaln mis-places such a record
  // depth-code parameters and multiplicity of the two records (k_rec_params): normally {0, 0, flen+blen, 1} for the
  // front and {flen, bases in the front, flen+blen, 1} for the back (src/fsdb.c:568-581,597-610)
  const int32_t* prm = tr + TREC_PARAMS;
  const int fBase = prm[0], fOff = prm[1], fB = prm[2], fMult = prm[3], bBase = prm[4], bOff = prm[5], bB = prm[6], bMult = prm[7];
  const int actF = tr[TREC_ACTF];                          // read bases in the front record
  // code of a column reached after `act` read bases, front or back record
  auto dcode = [&](bool back, int act) {
    const int a = back ? bOff + (act - actF) : fOff + act;
    return depth_code((back ? bBase : fBase) + a, (back ? bB : fB) - a - 1);
  };
  auto emit = [&](int o, int act, int code /* 0..4, or 5 = '-' */) {
    int p, gc, mult;
    bool dropped, back;
    if (o < g.ncols_f) { p = o; gc = g.start_w + o; back = false; dropped = dF; mult = fMult; }
    else if (g.split && o - g.ncols_f < g.ncols_b) { p = o - g.ncols_f; gc = p; back = true; dropped = dB; mult = bMult; }  // dff = flen + act, sic: src/fsdb.c:597
    else return;
    if (gc < 0 || gc >= Lp) { atomicOr(tb.flags, 2u); return; }
    const int d = dcode(back, act);
    if (d < 0 || d > 2 * PSSM_DEPTH) { atomicOr(tb.flags, 2u); return; }
    const int wc = tally_slot(gc, win_base, Lp);
    const bool in_lds = BINNED && wc >= 0 && wc < TALLY_WIN;
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;                    // sm[d][X][code], X = A,C,G,T (src/map_align.c:258-261)
    if (!dropped && code != 5) {
      if (BINNED) { const int16_t* row = pssm_lds + pmo + d * 25 + code; s0 = row[0]; s1 = row[5]; s2 = row[10]; s3 = row[15]; }
      else { const int32_t* row = pm + d * 25 + code; s0 = row[0]; s1 = row[5]; s2 = row[10]; s3 = row[15]; }
    }
    // mult > 1: the record is listed again through the stale back_asp of formerly split reads (see k_cull_mark).
    // Two copies of the same adds, one per address space: a pointer chosen at run time would turn every one of them
    // into a FLAT atomic, which is several times slower on LDS than ds_add.
    auto aadd = [](auto* q, int v) { (void)__hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto add_all = [&](auto* t, const int ws, auto* ncount) {
      if (!dropped) {                                      // src/mia.c:580-582
        aadd(&t[T_COV * ws], mult);
        if (code == 5) aadd(&t[T_GAP * ws], mult);
        else {
          if (code < 4) aadd(&t[(T_A + code) * ws], mult);
          else if (ncount) aadd(ncount, mult);
          if (!ncount) {
            aadd(&t[T_SA * ws], mult * s0);
            aadd(&t[T_SC * ws], mult * s1);
            aadd(&t[T_SG * ws], mult * s2);
            aadd(&t[T_ST * ws], mult * s3);
          }
        }
      }
      if (p > 0) aadd(&t[T_SPAN * ws], mult);         // start < pos <= end (src/map_align.c:466-469), dropped or not
    };
    typedef __attribute__((address_space(3))) int32_t lds_word;
    if (in_lds) add_all((lds_word*)lds + wc, TALLY_WIN, n_cnt ? (lds_word*)n_cnt + wc : (lds_word*)nullptr);
    else add_all(tb.tally + gc, Lp, (int32_t*)nullptr);
  };

  for (int r0 = abr; r0 < len2; r0 += 64) {
    const int r = r0 + lane;
    if (r >= len2) continue;
    const int code = (rp[r >> 1] >> ((r & 1) * 4)) & 15;
    if (diag) { emit(r - abr, r - abr, code); continue; }   // column offset == bases before it
    const int cv = cols[r];
    if (cv >= 0) {
      const int o = cbase + cv, act = r - abr;
      if (r > abr && cols[r - 1] >= 0)
        for (int o2 = cbase + cols[r - 1] + 1; o2 < o; o2++) emit(o2, act, 5);   // deleted reference columns: '-'
      emit(o, act, code);
    } else if (cv == COL_INSERT) {
      int r1 = r;                       // first row of this insert run
      while (r1 - 1 >= abr && cols[r1 - 1] == COL_INSERT) r1--;
      int rn = r + 1;                   // aligned row that follows the run
      while (rn < len2 && cols[rn] < 0) rn++;
      const int o = cbase + cols[rn], act = rn - abr, j = r - r1, glen = rn - r1;
      int p, gc, mult;
      bool ok = true, back = false;
      if (o < g.ncols_f) { p = o; gc = g.start_w + o; mult = fMult; }
      else if (g.split && o - g.ncols_f < g.ncols_b) { p = o - g.ncols_f; gc = p; back = true; mult = bMult; }
      else ok = false;
      if (ok && p > 0 && gc < Lp) {     // an insert in front of a record's first column is never counted (src/mia.c:492)
        const int d = dcode(back, act);
        if (j == 0) atomicMax(&tb.gaps[gc], glen);
        const uint64_t ev = (uint64_t)(uint32_t)gc | ((uint64_t)j << 32) | ((uint64_t)code << 42) | ((uint64_t)(d & 31) << 45) |
                            ((uint64_t)(is_rc ? 1 : 0) << 50);
        for (int m = 0; m < mult; m++) {
          // the event list has ONE global counter: a hundred thousand same-address atomics cost a millisecond.  A
          // binned workgroup collects its events in LDS and reserves their slots with one atomic at the end.
          int slot = BINNED ? atomicAdd(ev_cnt, 1) : TALLY_EV_CAP;
          if (slot < TALLY_EV_CAP) ev_buf[slot] = ev;
          else {
            const int e = atomicAdd(tb.n_events, 1);
            if (e < tb.cap_events) tb.events[e] = ev; else atomicOr(tb.flags, 1u);
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_tally(ReadSet rs, RefInfo ref, const int32_t* pssm2, const uint8_t* drop_front,
                                                const uint8_t* drop_back, TallyBuf tb, const int32_t* rec_params, const int32_t* rec_actf) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= rs.n) return;
  tally_one_read<false>(__builtin_amdgcn_readfirstlane((int)i), lane, rs, ref, pssm2, drop_front, drop_back, tb, nullptr, 0, rec_params, rec_actf, nullptr, nullptr, nullptr);
}

// ---- bucketing of the reads by alignment start (counting sort, one pass per iteration) ----
// the early tally's corrections: one read per wavefront, its assumed contribution (the record k_rec_early wrote) off again --
// the same adds with multiplicity -1 -- and its true one (k_rec_params' record) on; global atomics (a handful of reads)
__global__ __launch_bounds__(256) void k_tally_fix(ReadSet rs, RefInfo ref, const int32_t* pssm2, const uint8_t* drop_front, const uint8_t* drop_back, TallyBuf tb,
                                                    const int32_t* rec_true, const int32_t* rec_early, const int32_t* rec_actf, const int32_t* fix_list,
                                                    const int32_t* n_fix, const int32_t* abort_if = nullptr) {
  if (abort_if && *abort_if != 0) return;
  __shared__ int32_t stage[4][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = *n_fix;
  for (int k = (int)blockIdx.x * 4 + wv; k < n; k += (int)gridDim.x * 4) {
    const int i = fix_list[k];
    if (lane < 16) {
      int v = rec_early[(int64_t)i * 16 + lane];
      if (lane == TREC_PARAMS + 3 || lane == TREC_PARAMS + 7) v = -v;          // the two records' multiplicities
      stage[wv][lane] = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    tally_one_read<false>(i, lane, rs, ref, pssm2, drop_front, drop_back, tb, nullptr, 0, rec_early, rec_actf, nullptr, nullptr, nullptr, nullptr, stage[wv]);
    __builtin_amdgcn_wave_barrier();
    tally_one_read<false>(i, lane, rs, ref, pssm2, drop_front, drop_back, tb, nullptr, 0, rec_true, rec_actf, nullptr, nullptr, nullptr);
  }
}

constexpr int BUCKET_PER = 8;   // reads per thread of the bucketing kernels
// zero / zero_words: a buffer this launch clears on the side (the tally, the gaps and the ranks' event-count slots behind them:
// nothing adds to them before the tally kernel, which runs behind this one), or nullptr
// part / want (the early tally, k_rec_early): only the reads with (part[i] != 0) == (want != 0) are sorted; part == nullptr: all
// a read's bucket: its alignment start's stretch of TALLY_BUCKET columns -- and, split != 0 (the position-specific matrices' tally:
// a workgroup's reads all of one strand, so that its vertical counters need no strand), the strand in the lowest bit.  nb counts
// all buckets (twice the column stretches when split).
__device__ __forceinline__ int bucket_of(const ReadSet& rs, int64_t i, int nb, int split) {
  return (min(rs.as[i] / TALLY_BUCKET, (nb >> split) - 1) << split) | (split ? (rs.rc[i] ? 1 : 0) : 0);
}
__device__ __forceinline__ bool bucket_takes(const ReadSet& rs, int64_t i, const uint8_t* part, int want) {
  return i < rs.n && rs.sk[i] && (!part || (part[i] != 0) == (want != 0));
}
// the key of the second sort (k_sort2_*: every bucket by alignment start), worked out where the reads are walked in their own order
// (k_bucket_fill) and carried beside `order` as a byte: looked up through `order` it was three gathers per read and launch, and the two
// launches -- 0.47 + 0.51 ms at 10 M reads, beside the cull on the other stream -- took the cull's memory bandwidth (k_cull_records 0.48 -> 1.29 ms)
constexpr int SORT2_KEYS = 256;           // starts 0 .. 254 of a bucket's first column (the last bucket's reads may start beyond its 128); 255: see below
// (a read with a gap or a soft end goes behind all the gap-free ones of its bucket: the tally adds such a read's rows one by
// one, and a few of them in every wavefront made every wavefront walk that loop -- 1.0 of the tally's 1.5 ms at 10 M reads;
// collected at the end of the bucket they fill a few wavefronts of their own)
__device__ __forceinline__ int sort2_key(const ReadSet& rs, int64_t i, int b, int split) {
  if (!(rs.status[i] & ST_DIAG) || rs.abr[i] != 0) return SORT2_KEYS - 1;
  const int k = rs.as[i] - (b >> split) * TALLY_BUCKET;
  return k < 0 ? 0 : (k >= SORT2_KEYS - 1 ? SORT2_KEYS - 2 : k);
}
__global__ __launch_bounds__(256) void k_bucket_count(ReadSet rs, int32_t nb, int32_t* count, int32_t* zero, int64_t zero_words, const int32_t* abort_if = nullptr,
                                                       const uint8_t* part = nullptr, int32_t want = 0, int32_t split = 0) {
  if (abort_if && *abort_if != 0) return;     // (mia_hip_iterate queued this launch before the alignment's exact-kernel count was known: see iterate_body)
  extern __shared__ int32_t hist[];
  for (int b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = 0;
  if (zero) for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < zero_words; k += (int64_t)gridDim.x * 256) zero[k] = 0;
  __syncthreads();
  for (int k = 0; k < BUCKET_PER; k++) {
    const int64_t i = ((int64_t)blockIdx.x * BUCKET_PER + k) * 256 + threadIdx.x;
    if (bucket_takes(rs, i, part, want)) atomicAdd(&hist[bucket_of(rs, i, nb, split)], 1);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += blockDim.x) if (hist[b]) atomicAdd(&count[b], hist[b]);
}
// off[b] = first read of bucket b in `order`, wgoff[b] = first workgroup of bucket b (TALLY_CHUNK reads each)
// chunk: reads per tally workgroup (TALLY_CHUNK at most; the late half of a split tally takes fewer -- its reads are the slow ones)
__global__ __launch_bounds__(256) void k_bucket_scan(int32_t* count, int32_t nb, int32_t* off, int32_t* wgoff, int32_t* cursor, int32_t* wg_bucket, const int32_t* abort_if = nullptr,
                                                      int32_t chunk = TALLY_CHUNK) {
  if (abort_if && *abort_if != 0) return;     // (mia_hip_iterate queued this launch before the alignment's exact-kernel count was known: see iterate_body)
  // one workgroup of 256 threads: every thread a stretch of buckets, a scan over the 256 partial sums in LDS
  __shared__ int32_t s_run[256], s_wg[256];
  const int t = threadIdx.x, per = (nb + 255) / 256, b0 = t * per, b1 = b0 + per < nb ? b0 + per : nb;
  int run = 0, wg = 0;
  for (int b = b0; b < b1; b++) { run += count[b]; wg += (count[b] + chunk - 1) / chunk; }
  s_run[t] = run; s_wg[t] = wg;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int a = t >= o ? s_run[t - o] : 0, c = t >= o ? s_wg[t - o] : 0;
    __syncthreads();
    s_run[t] += a; s_wg[t] += c;
    __syncthreads();
  }
  int r = s_run[t] - run, w = s_wg[t] - wg;
  for (int b = b0; b < b1; b++) {
    off[b] = r; wgoff[b] = w; cursor[b] = 0;
    const int cb = count[b];
    const int nw = (cb + chunk - 1) / chunk;
    // (the tally's workgroups look bucket and reads up -- one load -- instead of searching wgoff and reading off[] behind it)
    for (int q = 0; q < nw; q++) reinterpret_cast<int4*>(wg_bucket)[w + q] = make_int4(b, r + q * chunk, min(r + (q + 1) * chunk, r + cb), 0);
    r += cb; w += nw;
    count[b] = 0;                                               // read for the last time: clean for the next call's k_bucket_count
  }
  if (t == 255) { off[nb] = s_run[255]; wgoff[nb] = s_wg[255]; }
}
__global__ __launch_bounds__(256) void k_bucket_fill(ReadSet rs, int32_t nb, const int32_t* off, int32_t* cursor, int32_t* order, const int32_t* abort_if = nullptr,
                                                      const uint8_t* part = nullptr, int32_t want = 0, int32_t split = 0, uint8_t* okey = nullptr, int32_t packed = 0) {
  if (abort_if && *abort_if != 0) return;     // (mia_hip_iterate queued this launch before the alignment's exact-kernel count was known: see iterate_body)
  extern __shared__ int32_t sh[];
  int32_t* hist = sh;
  int32_t* base = sh + nb;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = 0;
  __syncthreads();
  int bb[BUCKET_PER], rank[BUCKET_PER];
  for (int k = 0; k < BUCKET_PER; k++) {
    const int64_t i = ((int64_t)blockIdx.x * BUCKET_PER + k) * 256 + threadIdx.x;
    bb[k] = -1; rank[k] = 0;
    if (bucket_takes(rs, i, part, want)) { bb[k] = bucket_of(rs, i, nb, split); rank[k] = atomicAdd(&hist[bb[k]], 1); }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nb; k += blockDim.x) if (hist[k]) base[k] = atomicAdd(&cursor[k], hist[k]);
  __syncthreads();
  for (int k = 0; k < BUCKET_PER; k++) {
    const int64_t i = ((int64_t)blockIdx.x * BUCKET_PER + k) * 256 + threadIdx.x;
    if (bb[k] >= 0) {
      const int32_t at = off[bb[k]] + base[bb[k]] + rank[k];
      // (fewer than 2^24 reads: the key rides in the entry's top byte -- a scattered byte store beside the scattered word was 0.16 ms at 10 M reads)
      if (okey && packed) order[at] = (int32_t)i | (sort2_key(rs, i, bb[k], split) << 24);
      else {
        order[at] = (int32_t)i;
        if (okey) okey[at] = (uint8_t)sort2_key(rs, i, bb[k], split);
      }
    }
  }
}

// ---- the reads of a bucket in the order of their alignment starts (round 5) -------------------------------------------------
// The position-specific tally reduces the fifteen rows at either end of a read over RUNS of reads that start at the same
// column (k_tally_binned: tally_runs): for that, reads of equal start have to sit in neighbouring lanes.  k_bucket_fill
// leaves a bucket's reads in no particular order; these two launches sort every bucket by start -- a counting sort with
// SORT2_KEYS keys per bucket whose unit of work is the tally's own workgroup share (k_bucket_scan's table: bucket, first,
// last), so a bucket that holds all the reads is no slower than any other.  Order is a matter of speed only: the tally checks
// the keys of a run itself.
// packed: the keys are the top bytes of `order` (see k_bucket_fill), okey is not used
__global__ __launch_bounds__(256) void k_sort2_count(int32_t nb, const int32_t* wgoff, const int32_t* wg_bucket, const int32_t* order, const uint8_t* okey, int32_t packed,
                                                      int32_t* hist2, const int32_t* abort_if = nullptr) {
  if (abort_if && *abort_if != 0) return;
  if ((int)blockIdx.x >= wgoff[nb]) return;
  __shared__ int32_t h[SORT2_KEYS];
  const int4 wgi = reinterpret_cast<const int4*>(wg_bucket)[blockIdx.x];
  const int b = wgi.x, first = wgi.y, last = wgi.z;
  h[threadIdx.x] = 0;
  __syncthreads();
  for (int t = first + (int)threadIdx.x; t < last; t += 256) atomicAdd(&h[packed ? (int)((uint32_t)order[t] >> 24) : (int)okey[t]], 1);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist2[(int64_t)b * SORT2_KEYS + threadIdx.x], h[threadIdx.x]);
}
// cursor2: as hist2, zero before the launch; order2[off[b] + (reads of smaller keys in b) + (place among the key's reads)] = read
__global__ __launch_bounds__(256) void k_sort2_fill(int32_t nb, const int32_t* off, const int32_t* wgoff, const int32_t* wg_bucket, const int32_t* order, const uint8_t* okey,
                                                     int32_t packed, const int32_t* hist2, int32_t* cursor2, int32_t* order2, const int32_t* abort_if = nullptr) {
  if (abort_if && *abort_if != 0) return;
  if ((int)blockIdx.x >= wgoff[nb]) return;
  static_assert(SORT2_KEYS == 256, "one key per thread");
  __shared__ int32_t h[SORT2_KEYS], pre[SORT2_KEYS], base[SORT2_KEYS];
  const int4 wgi = reinterpret_cast<const int4*>(wg_bucket)[blockIdx.x];
  const int b = wgi.x, first = wgi.y, last = wgi.z;
  const int t = threadIdx.x;
  h[t] = 0;
  pre[t] = hist2[(int64_t)b * SORT2_KEYS + t];
  __syncthreads();
  constexpr int PER = (TALLY_CHUNK + 255) / 256;
  int32_t rd[PER], key[PER], rank[PER];
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const int e = first + q * 256 + t;
    rd[q] = -1; key[q] = 0; rank[q] = 0;
    if (e < last) {
      const int32_t v = order[e];
      rd[q] = packed ? (v & 0xFFFFFF) : v;
      key[q] = packed ? (int)((uint32_t)v >> 24) : (int)okey[e];
      rank[q] = atomicAdd(&h[key[q]], 1);
    }
  }
  // exclusive prefix of the bucket's histogram over the 256 keys (Hillis-Steele in LDS)
  const int mine = pre[t];
  for (int o = 1; o < SORT2_KEYS; o <<= 1) {
    __syncthreads();
    const int v = t >= o ? pre[t - o] : 0;
    __syncthreads();
    pre[t] += v;
  }
  __syncthreads();
  pre[t] -= mine;
  base[t] = h[t] ? atomicAdd(&cursor2[(int64_t)b * SORT2_KEYS + t], h[t]) : 0;
  __syncthreads();
  const int ob = off[b];
#pragma unroll
  for (int q = 0; q < PER; q++)
    if (rd[q] >= 0) order2[ob + pre[key[q]] + base[key[q]] + rank[q]] = rd[q];
}

#ifdef MIA_HIP_ALT_PATHS
// (MIA_HIP_DEBUG_SKIP & 65536, alt build: how many reads take which route of k_tally_binned -- tools/tally_kinds_probe.py)
__device__ unsigned long long g_tally_kinds[8];
#define TALLY_KIND(k) do { if (dbg & 65536u) atomicAdd(&g_tally_kinds[k], 1ull); } while (0)
// (MIA_HIP_DEBUG_SKIP & 131072: shader-clock cycles of a workgroup's phases, summed over the workgroups; slot 7 counts them)
__device__ unsigned long long g_tally_clk[8];
#define TALLY_CLK(k) do { if ((dbg & 131072u) && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_tally_clk[k], now_ - clk_); clk_ = now_; } } while (0)
#define TALLY_CLK_DECL unsigned long long clk_ = (dbg & 131072u) ? __builtin_amdgcn_s_memtime() : 0ull
// (MIA_HIP_DEBUG_SKIP bits 1 << 18 .. 1 << 21, timing only: no adds to the vertical counters / counters not folded into the window / no slab
// store / records fetched but no read taken -- tools/tally_prof.py)
#define TALLY_ABL(bit) ((dbg & (bit)) != 0u)
#else
#define TALLY_ABL(bit) false
#define TALLY_KIND(k) do { } while (0)
#define TALLY_CLK(k) do { } while (0)
#define TALLY_CLK_DECL do { } while (0)
#endif
// Bit-matrix transposition inside each half of a wavefront (five butterfly steps): lane l holds 32 bits going in; coming out,
// lane r of half h holds in bit c what bit r of lane 32 h + c was.
__device__ __forceinline__ uint32_t wave_transpose32(uint32_t x, int lane) {
#pragma unroll
  for (int j = 16; j >= 1; j >>= 1) {
    const uint32_t m = j == 16 ? 0x0000FFFFu : (j == 8 ? 0x00FF00FFu : (j == 4 ? 0x0F0F0F0Fu : (j == 2 ? 0x33333333u : 0x55555555u)));
    const uint32_t y = (uint32_t)__shfl_xor((int)x, j);
    x = (lane & j) ? ((x & ~m) | ((y >> j) & m)) : ((x & m) | ((y << j) & ~m));
  }
  return x;
}

// DEFER: the reads that fit none of the one-read-per-lane routes (two gaps, soft ends, odd records: a thousand in a million) are
// not tallied here, one per wavefront while the other lanes wait, but put on gen_list for k_tally_reduce's extra workgroups.  Seven hundred such reads cost 54 of this kernel's 215 us per million reads: every one of them is a stretch of code
// nobody else runs (instruction fetches from memory) and a chain of loads in front of a workgroup's barrier.
// RALL (round 5, position-specific matrices, reads of 129 .. 256 bases -- configs[4]'s 150): EVERY row of a gap-free read goes through the
// runs (tally_runs below), 64 rows of the reads' planes per round; no vertical counters in this instance (they hold 128 rows and 256
// columns), four plane words per read instead of two.
template <bool LINEAR, bool DEFER, bool RALL = false>
__global__ __launch_bounds__(256, 3) void k_tally_binned(ReadSet rs, RefInfo ref, const int32_t* pssm2, const uint8_t* drop_front,
                                                       const uint8_t* drop_back, TallyBuf tb, int32_t nb, const int32_t* off,
                                                       const int32_t* wgoff, const int32_t* order, const int32_t* rec_params,
                                                       const int32_t* rec_actf, int32_t* slabs, uint32_t dbg, const uint64_t* rplanes,
                                                       int32_t rplane_words, const int32_t* umax, const int32_t* wg_bucket, int32_t pk_bias, const int32_t* abort_if = nullptr,
                                                       int32_t chunk_reads = TALLY_CHUNK, int32_t* gen_list = nullptr, int32_t* n_gen = nullptr, int32_t split_flags = 0) {
  if (abort_if && *abort_if != 0) return;     // (mia_hip_iterate queued this launch before the alignment's exact-kernel count was known: see iterate_body)
  constexpr bool linear = LINEAR;
  // split_flags: bit 0 -- the buckets are split by strand (bucket_of) and sorted by start (k_sort2_*); bit 1 -- tally_runs below
  const int split = split_flags & 1;
  const bool runs_on = !LINEAR && (split_flags & 2) != 0;
  __shared__ int32_t lds[(TALLY_WORDS - 1) * TALLY_WIN];     // the pad word is never written
  // !LINEAR (a position-specific matrix): the scores of a base depend on its depth code and its strand -- but every base
  // further than 15 from both ends of its read has depth code 15 (src/pssm.c:6-46), seven in ten of a 100 bp read.  Those
  // are only COUNTED, per strand (forward strand in the low half of the word, reverse in the high half; a workgroup holds
  // at most TALLY_CHUNK reads), and their base counts and score sums are filled in from sm[strand][15] at the flush: one
  // LDS atomic per base instead of five.  The 15 bases at either end keep their explicit adds.
  __shared__ int32_t mid_cnt[LINEAR ? 1 : 5 * TALLY_WIN];
  // !LINEAR, the 15 bases at either end of a read (depth codes other than 15): their base count and four scores used to be
  // five LDS atomics each -- 150 of a 100 bp read's 220, and LDS atomic instructions are what this kernel waits for.  With
  // every score biased to be positive (pk_bias >= the most negative entry, bias + largest entry <= 2047: the host checks)
  // three 20-bit score sums share one 64-bit word and the fourth shares another with four 10-bit base counts: TWO 64-bit
  // atomics per base.  A workgroup adds at most TALLY_CHUNK = 512 bases to a column, so no field overflows into the next
  // (512 x 2047 < 2^20); the flush takes bias x count off again.  Bases that are not A/C/G/T keep the explicit adds.
  __shared__ unsigned long long pk1[LINEAR ? 1 : TALLY_WIN], pk2[LINEAR ? 1 : TALLY_WIN];
  static_assert(TALLY_CHUNK <= 512, "packed end-base sums: 512 adds of at most 2047 stay below 2^20");
  // linear: the matrix does not depend on depth or strand (the flat matrix), so scoreX(column) = sum_b count_b * sm[X][b].
  // The window then takes ONE LDS atomic per base (its count; N in n_cnt) instead of five, and the four score rows are
  // filled in from the counts when the window is flushed.  Same integer sums.
  __shared__ int32_t n_cnt[TALLY_WIN];
  __shared__ int32_t rec_stage[4][16];                       // one 64-byte read record per wavefront (see tally_one_read)
  __shared__ int16_t pssm_lds[2 * PSSM_WORDS];               // both matrices: every aligned base looks four entries up
  // LDS atomics are the limit of this kernel (~1.4 lane-atomics per cycle and CU, whatever the addresses).  Coverage
  // and span of a gap-free record are range counts: +1 / -1 at the two ends of a difference array instead of one
  // atomic per column; the prefix sums are taken when the window is flushed.
  __shared__ int32_t cov_diff[TALLY_WIN], span_diff[TALLY_WIN];
  __shared__ unsigned long long ev_buf[TALLY_EV_CAP];
  __shared__ int ev_cnt, ev_base;
  // TALLY RUNS (round 5; !LINEAR, the buckets split by strand and sorted by start).  The fifteen rows at either end of a read
  // cost two packed LDS atomics each, sixty per read, and those atomics were what this kernel waited for (r04: 130 M
  // bank-conflict cycles on 168 M active LDS cycles at 10 M reads).  But reads that start at the same column put the same
  // depth code on the same column: for a RUN of such reads in neighbouring lanes the adds of a row differ only in the base,
  // so the run's leader-row adds  sum_b count_b * (packed scores of (depth, b))  ONCE.  The counts come from the reads' bit
  // planes: rows 0..14 and the last fifteen of every lane, transposed over the wavefront (wave_transpose32: lane = row,
  // bits = lanes), popcounted under the run's lane mask.  run_tab[d][b] = the two packed words add_base adds for a base b at
  // depth code d on this workgroup's strand.
  __shared__ unsigned long long run_tab[LINEAR ? 1 : (2 * PSSM_DEPTH + 1) * 4 * 2];
  if ((int)blockIdx.x >= wgoff[nb]) return;   // the grid is an upper bound (no host round trip for the exact count)
  TALLY_CLK_DECL;
  const int4 wgi = reinterpret_cast<const int4*>(wg_bucket)[blockIdx.x];       // (k_bucket_scan's table: bucket, first read, last read)
  const int b = wgi.x >> split, first = wgi.y, last = wgi.z;                    // (split: the bucket's lowest bit is its reads' strand, bucket_of)
  const int wg_rc = split ? (wgi.x & 1) : 0;
  (void)wg_rc;
  const int win_base = b * TALLY_BUCKET;
  (void)off; (void)chunk_reads;
  // the first two passes' reads, asked for before the window is set up (every pass is a chain order -> record; the loop below keeps
  // the next pass's records in flight while it works on this one's)
  int i_nx = first + (int)threadIdx.x < last ? order[first + (int)threadIdx.x] : -1;
  int i_nx2 = first + 256 + (int)threadIdx.x < last ? order[first + 256 + (int)threadIdx.x] : -1;
  for (int k = threadIdx.x; k < (TALLY_WORDS - 1) * TALLY_WIN; k += blockDim.x) lds[k] = 0;
  // (LINEAR with the one-per-wavefront reads deferred: only sm[0][X][b] of the forward table is looked up, by the flush -- the whole
  // copy was six dependent rounds of global loads per thread, a sixth of a workgroup's time)
  for (int k = threadIdx.x; k < ((LINEAR && DEFER) ? 25 : 2 * PSSM_WORDS); k += blockDim.x) pssm_lds[k] = (int16_t)pssm2[k];
  for (int k = threadIdx.x; k < TALLY_WIN; k += blockDim.x) { cov_diff[k] = 0; span_diff[k] = 0; n_cnt[k] = 0; }
  if (!LINEAR) for (int k = threadIdx.x; k < 5 * TALLY_WIN; k += blockDim.x) mid_cnt[k] = 0;
  if (!LINEAR) for (int k = threadIdx.x; k < TALLY_WIN; k += blockDim.x) { pk1[k] = 0; pk2[k] = 0; }
  if (threadIdx.x == 0) ev_cnt = 0;
  __syncthreads();
  if (!LINEAR && runs_on && pk_bias >= 0) {
    for (int k = threadIdx.x; k < (2 * PSSM_DEPTH + 1) * 4; k += blockDim.x) {
      const int16_t* row = pssm_lds + (wg_rc ? PSSM_WORDS : 0) + (k >> 2) * 25 + (k & 3);
      run_tab[2 * k] = (unsigned long long)(uint32_t)((int)row[0] + pk_bias) | ((unsigned long long)(uint32_t)((int)row[5] + pk_bias) << 20) |
                       ((unsigned long long)(uint32_t)((int)row[10] + pk_bias) << 40);
      run_tab[2 * k + 1] = (unsigned long long)(uint32_t)((int)row[15] + pk_bias) | (1ull << (20 + 10 * (k & 3)));
    }
    __syncthreads();
  }
  TALLY_CLK(0);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  (void)wv;
  // The common case by far -- proven diagonal, one record, listed once, its own depth codes, every column inside the LDS
  // window -- goes ONE READ PER LANE: 64 independent chains of (order -> record -> bases) loads per wavefront instead of
  // one, which is what the kernel was waiting on (80 % of the wave cycles in SQ_WAIT_ANY).  Everything else takes the
  // general one-read-per-wavefront path.  Same integer sums either way.
  typedef __attribute__((address_space(3))) int32_t lds_i32;
  auto aadd = [](lds_i32* q, int v) { (void)__hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // one aligned base of a record that is listed once and not dropped, at window slot wc: code 0..4, depth code d, strand
  auto add_base = [&](int wc, int code, int d, bool is_rc, bool pack_ok = true) {      // pack_ok: this record is listed once (see pk1)
    lds_i32* t = (lds_i32*)lds + wc;
    if (LINEAR) { if (code < 4) aadd(&t[(T_A + code) * TALLY_WIN], 1); else aadd((lds_i32*)n_cnt + wc, 1); return; }
    if (d == PSSM_DEPTH) { aadd((lds_i32*)mid_cnt + (code < 4 ? code : 4) * TALLY_WIN + wc, is_rc ? 65536 : 1); return; }
    const int16_t* row = pssm_lds + (is_rc ? PSSM_WORDS : 0) + d * 25 + code;
    if (pk_bias >= 0 && code < 4 && pack_ok) {
      typedef __attribute__((address_space(3))) unsigned long long lds_u64;
      const unsigned long long v1 = (unsigned long long)(uint32_t)((int)row[0] + pk_bias) | ((unsigned long long)(uint32_t)((int)row[5] + pk_bias) << 20) |
                                    ((unsigned long long)(uint32_t)((int)row[10] + pk_bias) << 40);
      const unsigned long long v2 = (unsigned long long)(uint32_t)((int)row[15] + pk_bias) | (1ull << (20 + 10 * code));
      (void)__hip_atomic_fetch_add((lds_u64*)pk1 + wc, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      (void)__hip_atomic_fetch_add((lds_u64*)pk2 + wc, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (code < 4) aadd(&t[(T_A + code) * TALLY_WIN], 1);
    aadd(&t[T_SA * TALLY_WIN], (int)row[0]);
    aadd(&t[T_SC * TALLY_WIN], (int)row[5]);
    aadd(&t[T_SG * TALLY_WIN], (int)row[10]);
    aadd(&t[T_ST * TALLY_WIN], (int)row[15]);
  };
  const int L = ref.L, Lp = tb.Lp;
  int warm = 0;
  // BIT-SLICED COUNTS (linear matrix, reads of up to 128 bases without N, the bit planes of k_read_planes at hand).  The
  // window takes its base counts through LDS atomics, and those run at ~1.4 lane-atomics per cycle and CU whatever the
  // addresses: a hundred per read was two thirds of this kernel.  Instead every lane keeps, for each base X and each 64-column
  // word w of the window's first 256 columns, a 2-bit vertical counter (two 64-bit planes): a read's planes, shifted to its
  // place, give four one-hot masks per word, and adding a mask to a vertical counter is three logic operations for 64
  // columns at once.  A lane sees at most TALLY_CHUNK / 256 = 2 reads.  At the end the 64 lanes' counters are added up in
  // a halving exchange (the lane keeps the half of the items its lane-id bit names, sends the other half, adds what it
  // receives bit plane by bit plane: 16 items on 64 lanes become one 8-bit item on each group of four lanes) and each lane
  // adds 16 columns of its item to the window -- 1 024 atomics per wavefront instead of 12 800.
  constexpr int BS_W = 4;                                   // window words covered: columns 0 .. 255
  unsigned long long bs0[4 * BS_W], bs1[4 * BS_W];
  // (position-specific matrices: the counters take the bases of depth code 15 -- seven in ten -- of a workgroup whose reads are all of
  // one strand, split != 0; the fifteen rows at either end of a read keep their packed adds)
  static_assert(!(LINEAR && RALL), "RALL is a mode of the position-specific tally");
  const bool bs_on = !RALL && (LINEAR || split != 0) && rplanes && umax && !(dbg & 4096u);
  const bool planes_all = RALL && split != 0 && rplanes && umax;
  // a stretch of a read's planes (rows from r_lo on, n_rows of them) counted at window columns c .. c + n_rows - 1 < 256
  auto bs_count = [&](unsigned long long l0, unsigned long long l1, unsigned long long h0, unsigned long long h1, int r_lo, int n_rows, int c) {
    // rows r_lo.. down to bit 0 (a 128-bit shift right), then n_rows of them kept
    const int rs_ = r_lo & 63;
    const bool rw = r_lo >= 64;
    auto down = [&](unsigned long long a0, unsigned long long a1, unsigned long long* o0, unsigned long long* o1) {
      const unsigned long long b0 = rw ? a1 : a0, b1 = rw ? 0ull : a1;
      *o0 = (b0 >> rs_) | ((b1 << 1) << (63 - rs_));
      *o1 = b1 >> rs_;
    };
    unsigned long long dl0, dl1, dh0, dh1;
    down(l0, l1, &dl0, &dl1); down(h0, h1, &dh0, &dh1);
    const unsigned long long v0 = n_rows >= 64 ? ~0ull : ((1ull << n_rows) - 1ull);
    const unsigned long long v1 = n_rows <= 64 ? 0ull : (n_rows >= 128 ? ~0ull : ((1ull << (n_rows - 64)) - 1ull));
    const int bsh = c & 63, ws = c >> 6;
    auto place = [&](unsigned long long a0, unsigned long long a1, unsigned long long* o) {
      const unsigned long long t0 = a0 << bsh, t1 = (a1 << bsh) | ((a0 >> 1) >> (63 - bsh)), t2 = (a1 >> 1) >> (63 - bsh);
#pragma unroll
      for (int w = 0; w < BS_W; w++) o[w] = w == ws ? t0 : (w == ws + 1 ? t1 : (w == ws + 2 ? t2 : 0ull));
    };
    unsigned long long pl_lo[BS_W], pl_hi[BS_W], pl_v[BS_W];
    place(dl0 & v0, dl1 & v1, pl_lo); place(dh0 & v0, dh1 & v1, pl_hi); place(v0, v1, pl_v);
#pragma unroll
    for (int w = 0; w < BS_W; w++) {
      const unsigned long long lo_w = pl_lo[w], hi_w = pl_hi[w];
      const unsigned long long m[4] = {pl_v[w] & ~(lo_w | hi_w), lo_w & ~hi_w, hi_w & ~lo_w, lo_w & hi_w};
#pragma unroll
      for (int x = 0; x < 4; x++) {
        const unsigned long long cy = bs0[x * BS_W + w] & m[x];
        bs0[x * BS_W + w] ^= m[x];
        bs1[x * BS_W + w] |= cy;
      }
    }
  };
  // the same for a whole read whose planes are already laid out by COLUMN (bit q of lo / hi / valid: the base at window column c + q;
  // a read with one gap is laid out by bs_gap below), c < 128: the three 128-bit planes land in words c / 64 .. c / 64 + 2
  auto bs_add = [&](unsigned long long lo0, unsigned long long lo1, unsigned long long hi0, unsigned long long hi1, unsigned long long v0, unsigned long long v1, int c) {
    const int bsh = c & 63;
    const bool up = c >= 64;
    auto place = [&](unsigned long long a0, unsigned long long a1, unsigned long long* o) {
      const unsigned long long t0 = a0 << bsh, t1 = (a1 << bsh) | ((a0 >> 1) >> (63 - bsh)), t2 = (a1 >> 1) >> (63 - bsh);
      o[0] = up ? 0ull : t0; o[1] = up ? t0 : t1; o[2] = up ? t1 : t2; o[3] = up ? t2 : 0ull;
    };
    static_assert(BS_W == 4, "bs_add places three words at word 0 or 1");
    unsigned long long pl_lo[BS_W], pl_hi[BS_W], pl_v[BS_W];
    place(lo0, lo1, pl_lo); place(hi0, hi1, pl_hi); place(v0, v1, pl_v);
#pragma unroll
    for (int w = 0; w < BS_W; w++) {
      const unsigned long long lo_w = pl_lo[w], hi_w = pl_hi[w];
      const unsigned long long m[4] = {pl_v[w] & ~(lo_w | hi_w), lo_w & ~hi_w, hi_w & ~lo_w, lo_w & hi_w};
#pragma unroll
      for (int x = 0; x < 4; x++) {
        const unsigned long long cy = bs0[x * BS_W + w] & m[x];
        bs0[x * BS_W + w] ^= m[x];
        bs1[x * BS_W + w] |= cy;
      }
    }
  };
  // rows -> columns for a read with one gap in front of row `grow`: gn inserted rows drop out and the rows behind them move down
  // (ins), or gn deleted reference columns open up and the rows from grow on move up; gn = 0: as it is.  The same for every plane.
  auto bs_gap = [&](unsigned long long* x0, unsigned long long* x1, bool ins, int grow, int gn) {
    const unsigned long long m0 = grow >= 64 ? ~0ull : ((1ull << grow) - 1ull), m1 = grow <= 64 ? 0ull : ((1ull << (grow - 64)) - 1ull);
    const unsigned long long a0 = *x0, a1 = *x1;
    const unsigned long long r0 = (a0 >> gn) | ((a1 << 1) << (63 - gn)), r1 = a1 >> gn;                                  // down by gn (< 64)
    const unsigned long long k0 = a0 & ~m0, k1 = a1 & ~m1;
    const unsigned long long u0 = k0 << gn, u1 = (k1 << gn) | ((k0 >> 1) >> (63 - gn));                                  // the rows from grow on, up by gn
    *x0 = (a0 & m0) | (ins ? r0 & ~m0 : u0);
    *x1 = (a1 & m1) | (ins ? r1 & ~m1 : u1);
  };
#pragma unroll
  for (int t = 0; t < 4 * BS_W; t++) { bs0[t] = 0; bs1[t] = 0; }
  static_assert(TALLY_CHUNK <= 3 * 256 && TALLY_CHUNK_LINEAR <= 3 * 256, "a lane's vertical counters hold two bits");
  // a read's record, and -- fetched before the record says whether they are needed, one round trip less -- its planes and N mark
  // Straight-line loads (a lane without a read asks for read 0's, the second words of one-word planes are the first again; what
  // was loaded is only looked at a pass later): with the loads inside `if (ii >= 0)` the number of loads in flight at the first
  // use of the PREVIOUS pass's record depended on the path, and the wait the compiler put there was for all of them -- the
  // records "a pass ahead" were waited for on the spot, the gather's latency (60 % of this kernel) hidden by nothing.
  struct ReadIn { int4 a, b4, c4; unsigned long long l0, l1, h0, h1, l2, l3, h2, h3; };
  const int pw1 = rplane_words > 1 ? 1 : 0;
  auto fetch = [&](int ii) -> ReadIn {
    ReadIn q;
    const int64_t jj = ii >= 0 ? ii : 0;
    const int4* tr4 = reinterpret_cast<const int4*>(rec_params + jj * 16);
    q.a = tr4[0]; q.b4 = tr4[1]; q.c4 = tr4[2];
    q.l0 = q.l1 = q.h0 = q.h1 = 0ull;
    // (four different addresses whatever the number of words -- the planes are padded by two words --: given `pl[words > 1 ? 1 : 0]` the
    // compiler loads one word, copies it and branches around the second load, and the copy waits for the load on the spot)
    if (LINEAR) {
      const uint64_t* pl = bs_on ? rplanes + jj * 2 * rplane_words : reinterpret_cast<const uint64_t*>(rec_params);
      q.l0 = pl[0]; q.l1 = pl[1]; q.h0 = pl[rplane_words]; q.h1 = pl[rplane_words + 1];
    } else if (bs_on) {
      const uint64_t* pl = rplanes + jj * 2 * rplane_words;
      q.l0 = pl[0]; q.l1 = pl[1]; q.h0 = pl[rplane_words]; q.h1 = pl[rplane_words + 1];
    }
    q.l2 = q.l3 = q.h2 = q.h3 = 0ull;
    if (RALL && planes_all) {
      // (words beyond a read's own are the next plane's or the next read's -- the buffer is padded --: rows beyond the read are never looked at)
      const uint64_t* pl = rplanes + jj * 2 * rplane_words;
      const int w1 = rplane_words > 1 ? 1 : 0, w2 = rplane_words > 2 ? 2 : w1, w3 = rplane_words > 3 ? 3 : w2;
      q.l0 = pl[0]; q.l1 = pl[w1]; q.l2 = pl[w2]; q.l3 = pl[w3];
      q.h0 = pl[rplane_words]; q.h1 = pl[rplane_words + w1]; q.h2 = pl[rplane_words + w2]; q.h3 = pl[rplane_words + w3];
    }
    return q;
  };
  ReadIn in_nx = fetch(i_nx);
  bool nx2_ok = true;                 // (i_nx2 holds -1 already where there is no read)
  const int last_k = last > 0 ? last - 1 : 0;
  for (int k0 = first; k0 < last; k0 += 256) {
    const int i = i_nx;
    const ReadIn in = in_nx;
    const bool have = i >= 0 && !(TALLY_ABL(1u << 21) && in.a.x != 0x7FFFFFFF);
    i_nx = nx2_ok ? i_nx2 : -1;
    {
      const int kk = k0 + 512 + (int)threadIdx.x;
      nx2_ok = kk < last;
      i_nx2 = order[kk < last_k ? kk : last_k];
    }
    in_nx = fetch(i_nx);
    const int4 a = in.a, b4 = in.b4, c4 = in.c4;
    const unsigned long long pv_l0 = in.l0, pv_l1 = pw1 ? in.l1 : 0ull, pv_h0 = in.h0, pv_h1 = pw1 ? in.h1 : 0ull;
    const bool um_ok = bs_on && (a.w & TRF_NO_N);
    bool fast = false;
    bool coop = false;                 // this lane's read takes part in the wavefront's runs (its end rows are not added one by one)
    int c_w0 = 0, c_nal = 0;
    if (have) {
      const int flags = a.w;
      const int len2 = a.z & 0xFFFF, abr = (int)(int16_t)((uint32_t)a.z >> 16);
      const RecGeom g = rec_geom(a.x, a.y, L);
      const int n_al = len2 - abr, w0 = g.start_w - win_base;
      const int fBase = c4.x, fOff = c4.y, fB = c4.z, fMult = c4.w;
      fast = (flags & TRF_SK) && !(flags & TRF_TOO_LONG) && (flags & TRF_DIAG) && !g.split && fMult == 1 && fBase == 0 && fOff == 0 && w0 >= 0 &&
             w0 + n_al <= TALLY_WIN && g.start_w + n_al <= Lp && n_al <= g.ncols_f && n_al > 0;
      // !LINEAR: the BASES of a one-gap read go through this loop as well, beside the gap-free reads' (the same per-base
      // adds at columns shifted by the gap) -- a loop of their own further down was a second pass of a hundred LDS-atomic
      // round trips for a tenth of the lanes, 0.1 of configs[2]'s 0.40 ms.  The gap itself stays with the one-gap block.
      bool one_here = false;
      int og_row = 1 << 20, og_n = 0, og_ins = 0;
      if (!LINEAR && !fast && (flags & TRF_ONEGAP)) {
        const uint32_t desc = (uint32_t)b4.w;
        const int ins = (int)(desc & 1u), grow = (int)((desc >> 1) & 511u), gn = (int)((desc >> 10) & 63u);
        const int ncol = ins ? n_al - gn : n_al + gn;
        one_here = (flags & TRF_SK) && !(flags & TRF_TOO_LONG) && !(flags & TRF_DIAG) && !g.split && fMult == 1 && fBase == 0 && fOff == 0 && w0 >= 0 &&
                   ncol > 0 && w0 + ncol <= TALLY_WIN && g.start_w + ncol <= Lp && ncol == g.ncols_f && gn > 0 && grow > abr && grow + (ins ? gn : 0) < len2;
        if (one_here) { og_row = grow; og_n = gn; og_ins = ins; }
      }
      // LINEAR: a one-gap read whose two stretches fit the vertical counters is taken HERE as well, its loads (record, planes, the
      // inserted bases' word) side by side with the gap-free reads' -- in a block of its own further down the tenth of the lanes
      // that have one waited for the same chain of loads a second time, 39 of the tally's 215 us per million reads.
      bool one_sl = false;
      int sl_ins = 0, sl_row = 0, sl_n = 0, sl_ncol = 0;
      if (LINEAR && !fast && (flags & TRF_ONEGAP) && bs_on && abr == 0 && len2 <= 128 && !(dbg & (8u | 2048u))) {
        const uint32_t desc = (uint32_t)b4.w;
        const int ins = (int)(desc & 1u), grow = (int)((desc >> 1) & 511u), gn = (int)((desc >> 10) & 63u);
        const int ncol = ins ? n_al - gn : n_al + gn;
        one_sl = (flags & TRF_SK) && !(flags & TRF_TOO_LONG) && !(flags & TRF_DIAG) && !g.split && fMult == 1 && fBase == 0 && fOff == 0 && w0 >= 0 &&
                 ncol > 0 && ncol <= 128 && w0 < 128 && g.start_w + ncol <= Lp && ncol == g.ncols_f && gn > 0 && grow > abr && grow + (ins ? gn : 0) < len2 &&
                 um_ok;
        if (one_sl) { sl_ins = ins; sl_row = grow; sl_n = gn; sl_ncol = ncol; }
      }
      if (fast) TALLY_KIND(0);
      if ((fast || one_here || one_sl) && !(dbg & 8u)) {
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(rs.packed + (uint32_t)b4.y);    // reads start on 4-byte boundaries
        const bool dF = (flags & TRF_DF) != 0;
        lds_i32* t = (lds_i32*)lds + w0;
        uint32_t word = 0;
        int bad = 0;
        if (linear) {
          // depth codes are not needed for the sums; they only have to be valid: act <= 15 always is, beyond that
          // dfb = fB - act - 1 must not be negative (depth_code above)
          bad = n_al > PSSM_DEPTH + 1 && fB < n_al;
          const bool fast_sl = !one_sl && !dF && bs_on && abr == 0 && len2 <= 128 && w0 < 128 && um_ok;
          uint32_t iw0 = 0, iw1 = 0;
          if (one_sl && sl_ins) { iw0 = rp[sl_row >> 3]; iw1 = rp[(sl_row + sl_n - 1) >> 3]; }
          if (fast_sl || (one_sl && !dF)) {
            // gap-free reads and reads with one gap in ONE instruction stream: the planes go from rows to columns (bs_gap; nothing to do
            // without a gap) and into the vertical counters once.  (Two calls of bs_count either side of the gap, in a branch only the
            // one-gap lanes take, was as much work again as all the gap-free lanes' -- and this loop is bound by its 64-bit logic.)
            unsigned long long l0 = pv_l0, l1 = pv_l1, h0 = pv_h0, h1 = pv_h1;
            unsigned long long v0 = len2 >= 64 ? ~0ull : ((1ull << len2) - 1ull), v1 = len2 <= 64 ? 0ull : (len2 >= 128 ? ~0ull : ((1ull << (len2 - 64)) - 1ull));
            l0 &= v0; l1 &= v1; h0 &= v0; h1 &= v1;
            const int gn = one_sl ? sl_n : 0, grow = one_sl ? sl_row : 0;
            const bool gi = one_sl && sl_ins;
            if (!TALLY_ABL(1u << 18)) {
            bs_gap(&l0, &l1, gi, grow, gn); bs_gap(&h0, &h1, gi, grow, gn); bs_gap(&v0, &v1, gi, grow, gn);
            bs_add(l0, l1, h0, h1, v0, v1, w0);
            }
          }
          if (one_sl) {
            TALLY_KIND(3); TALLY_KIND(4);
            // the gap itself: deleted reference columns count as '-', inserted read rows become insert events at the column that
            // follows (src/map_align.c:444-510)
            const int grow = sl_row, gn = sl_n, ins = sl_ins;
            if (!dF && !ins) for (int q = 0; q < gn; q++) aadd(&t[T_GAP * TALLY_WIN + grow + q], 1);
            if (ins)
              for (int j = 0; j < gn; j++) {
                const int r = grow + j;
                const uint32_t wj = (r >> 3) == (grow >> 3) ? iw0 : ((r >> 3) == ((grow + gn - 1) >> 3) ? iw1 : rp[r >> 3]);
                const int code = (int)((wj >> ((r & 7) * 4)) & 15u);
                const int gc = g.start_w + grow, act = grow + gn;
                if (j == 0) atomicMax(&tb.gaps[gc], gn);
                const int d = depth_code(act, fB - act - 1);
                const uint64_t ev = (uint64_t)(uint32_t)gc | ((uint64_t)j << 32) | ((uint64_t)code << 42) | ((uint64_t)(d & 31) << 45) |
                                    ((uint64_t)((flags & TRF_RC) ? 1 : 0) << 50);
                const int slot = atomicAdd(&ev_cnt, 1);
                if (slot < TALLY_EV_CAP) ev_buf[slot] = ev;
                else {
                  const int e = atomicAdd(tb.n_events, 1);
                  if (e < tb.cap_events) tb.events[e] = ev; else atomicOr(tb.flags, 1u);
                }
              }
          } else if (fast_sl) {
            TALLY_KIND(1);
          } else if (!dF) {
            lds_i32* nc = (lds_i32*)n_cnt + w0;
            for (int act = 0; act < n_al; act++) {
              const int r = abr + act;
              if (act == 0 || (r & 7) == 0) word = rp[r >> 3];
              const int code = (int)((word >> ((r & 7) * 4)) & 15u);
              if (code < 4) aadd(&t[(T_A + code) * TALLY_WIN], 1); else aadd(nc, 1);
              t++; nc++;
            }
          }
        } else {
        // (the read's words three ahead of their use: a load per eight rows that is waited for on the spot was a
        // microsecond of every eight iterations; packed reads are padded, so the words beyond the last may be fetched)
        // The rows of depth code 15 (row 15 .. B - 16: src/fsdb.c:572-582 through depth_code) of a read that fits the vertical
        // counters -- up to 128 bases, no N, its strand the workgroup's -- are counted there, gap-free and one-gap reads alike
        // (bs_gap), and scored at the flush like mid_cnt's; the loop below then only walks the rows at either end.
        const int og_cols = n_al + (one_here ? (og_ins ? -og_n : og_n) : 0);
        const bool mid_sl = bs_on && split != 0 && abr == 0 && len2 <= 128 && w0 < 128 && og_cols <= 128 && um_ok && ((flags & TRF_RC) != 0) == (wg_rc != 0);
        const int mid_lo = PSSM_DEPTH, mid_hi = min(fB - PSSM_DEPTH - 1, n_al - 1);
        const bool have_mid = mid_sl && mid_hi >= mid_lo;
        if (have_mid && !dF) {
          auto below = [](int q, unsigned long long* o0, unsigned long long* o1) {      // bits 0 .. q - 1 of 128
            *o0 = q >= 64 ? ~0ull : ((1ull << q) - 1ull);
            *o1 = q <= 64 ? 0ull : (q >= 128 ? ~0ull : ((1ull << (q - 64)) - 1ull));
          };
          unsigned long long a0, a1, b0, b1;
          below(mid_hi + 1, &a0, &a1); below(mid_lo, &b0, &b1);
          unsigned long long v0 = a0 & ~b0, v1 = a1 & ~b1;
          unsigned long long l0 = pv_l0 & v0, l1 = pv_l1 & v1, h0 = pv_h0 & v0, h1 = pv_h1 & v1;
          const int gn = one_here ? og_n : 0, grow = one_here ? og_row : 0;
          const bool gi = one_here && og_ins;
          bs_gap(&l0, &l1, gi, grow, gn); bs_gap(&h0, &h1, gi, grow, gn); bs_gap(&v0, &v1, gi, grow, gn);
          bs_add(l0, l1, h0, h1, v0, v1, w0);
        }
        const int wlast = (len2 - 1) >> 3;
        auto rows = [&](int a_from, int a_to) {
        if (a_from >= a_to) return;
        int wi = (abr + a_from) >> 3;
        uint32_t wq1 = rp[wi + 1 <= wlast ? wi + 1 : wlast], wq2 = rp[wi + 2 <= wlast ? wi + 2 : wlast], wq3 = rp[wi + 3 <= wlast ? wi + 3 : wlast];
        word = rp[wi];
        for (int act = a_from; act < a_to; act++) {
          const int r = abr + act;
          if (act != a_from && (r & 7) == 0) { wi++; word = wq1; wq1 = wq2; wq2 = wq3; wq3 = rp[wi + 3 <= wlast ? wi + 3 : wlast]; }
          const int code = (int)((word >> ((r & 7) * 4)) & 15u);
          const int d = depth_code(act, fB - act - 1);
          if (one_here) {
            // (a one-gap read: its depth codes are checked by its own block below; an inserted row is an event there, the rows
            // behind the gap sit og_n columns further left (insert) or right (deleted reference columns))
            if (og_ins && r >= og_row && r < og_row + og_n) continue;
            const int dd = d < 0 ? 0 : (d > 2 * PSSM_DEPTH ? 2 * PSSM_DEPTH : d);
            if (!dF) add_base(w0 + act + (r >= og_row ? (og_ins ? -og_n : og_n) : 0), code, dd, (flags & TRF_RC) != 0);
            continue;
          }
          bad |= (d < 0) | (d > 2 * PSSM_DEPTH);
          const int dd = d < 0 ? 0 : (d > 2 * PSSM_DEPTH ? 2 * PSSM_DEPTH : d);
          if (!dF && !(dbg & 16u)) add_base(w0 + act, code, dd, (flags & TRF_RC) != 0);
        }
        };
        // (a gap-free read of at least 31 bases whose depth codes are its own: rows 0..14 carry codes 0..14, the last fifteen 16..30)
        if (RALL)      // every row through the runs: a gap-free read of this workgroup's strand without N, its planes at hand
          coop = runs_on && planes_all && fast && !one_here && !dF && pk_bias >= 0 && fB == n_al && abr == 0 && len2 <= 64 * rplane_words && len2 <= 256 &&
                 (a.w & TRF_NO_N) && ((flags & TRF_RC) != 0) == (wg_rc != 0) && !(dbg & 16u);
        else
        coop = runs_on && have_mid && fast && !one_here && !dF && pk_bias >= 0 && n_al >= 2 * PSSM_DEPTH + 1 && fB == n_al && !(dbg & 16u);
        if (coop) { c_w0 = w0; c_nal = n_al; }
        else {
          rows(0, have_mid ? mid_lo : n_al);
          if (have_mid) rows(mid_hi + 1, n_al);
        }
        }
        if (fast || one_sl) {
        // coverage (not dropped): columns w0 .. w0+n-1; span (start < pos <= end, dropped or not): w0+1 .. w0+n-1 (n: the read's columns)
        const int nc = one_sl ? sl_ncol : n_al;
        if (!dF) { aadd((lds_i32*)cov_diff + w0, 1); if (w0 + nc < TALLY_WIN) aadd((lds_i32*)cov_diff + w0 + nc, -1); }
        if (nc > 1) { aadd((lds_i32*)span_diff + w0 + 1, 1); if (w0 + nc < TALLY_WIN) aadd((lds_i32*)span_diff + w0 + nc, -1); }
        if (bad | (int)(word & (dbg & 16u ? 0x40000000u : 0u))) atomicOr(tb.flags, 2u);
        }
      }
      if (one_sl) fast = true;
    }
    if constexpr (!LINEAR) {
      const unsigned long long cm = runs_on ? __ballot(coop) : 0ull;
      if (RALL && cm) {
        // runs: neighbouring lanes with the same start and length (as below)
        const int key = (c_w0 << 9) | c_nal;
        const int pkey = __shfl_up(key, 1);
        const bool pcoop = __shfl_up((int)coop, 1) != 0;
        const unsigned long long lm0 = __ballot(coop && (lane == 0 || !pcoop || pkey != key));
        const unsigned long long brk = lm0 | ~cm;
        typedef __attribute__((address_space(3))) unsigned long long lds_u64a;
        const unsigned long long pl_lo[4] = {in.l0, in.l1, in.l2, in.l3}, pl_hi[4] = {in.h0, in.h1, in.h2, in.h3};
        // a round: 64 rows -- the lower half of the wavefront takes rows 64 t .. 64 t + 31, the upper half the 32 behind them
        for (int t = 0; t < 4; t++) {
          if (__ballot(coop && c_nal > 64 * t) == 0ull) break;           // (wave-uniform)
          const uint32_t lo_a = coop ? (uint32_t)pl_lo[t] : 0u, lo_b = coop ? (uint32_t)(pl_lo[t] >> 32) : 0u;
          const uint32_t hi_a = coop ? (uint32_t)pl_hi[t] : 0u, hi_b = coop ? (uint32_t)(pl_hi[t] >> 32) : 0u;
          const uint32_t tla = wave_transpose32(lo_a, lane), tlb = wave_transpose32(lo_b, lane), tha = wave_transpose32(hi_a, lane), thb = wave_transpose32(hi_b, lane);
          const uint32_t gl = (uint32_t)__shfl_xor((int)(lane < 32 ? tlb : tla), 32), gh = (uint32_t)__shfl_xor((int)(lane < 32 ? thb : tha), 32);
          const unsigned long long Ml = lane < 32 ? ((unsigned long long)tla | ((unsigned long long)gl << 32)) : ((unsigned long long)gl | ((unsigned long long)tlb << 32));
          const unsigned long long Mh2 = lane < 32 ? ((unsigned long long)tha | ((unsigned long long)gh << 32)) : ((unsigned long long)gh | ((unsigned long long)thb << 32));
          const int row = 64 * t + lane;                       // (lane < 32: rows 64 t + lane of word a; lane >= 32: rows 64 t + 32 + (lane - 32) of word b)
          unsigned long long lm = lm0;
          while (lm) {
            const int a0 = __builtin_ctzll(lm);
            lm &= lm - 1;
            const unsigned long long above = a0 == 63 ? 0ull : (~0ull << (a0 + 1));
            const unsigned long long nxt = brk & above;
            const int e0 = nxt ? __builtin_ctzll(nxt) : 64;
            const unsigned long long seg = (e0 == 64 ? ~0ull : ((1ull << e0) - 1ull)) & (~0ull << a0);
            const int s_w0 = __builtin_amdgcn_readlane(c_w0, a0), s_nal = __builtin_amdgcn_readlane(c_nal, a0);
            if (row < s_nal) {
              const unsigned long long lo = Ml & seg, hi = Mh2 & seg;
              const int cT = __popcll(lo & hi), cC = __popcll(lo & ~hi), cG = __popcll(hi & ~lo), cA = __popcll(seg) - cT - cC - cG;
              const unsigned long long* tab = run_tab + depth_code(row, s_nal - row - 1) * 8;
              const unsigned long long v1 = (unsigned long long)cA * tab[0] + (unsigned long long)cC * tab[2] + (unsigned long long)cG * tab[4] + (unsigned long long)cT * tab[6];
              const unsigned long long v2 = (unsigned long long)cA * tab[1] + (unsigned long long)cC * tab[3] + (unsigned long long)cG * tab[5] + (unsigned long long)cT * tab[7];
              (void)__hip_atomic_fetch_add((lds_u64a*)pk1 + s_w0 + row, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              (void)__hip_atomic_fetch_add((lds_u64a*)pk2 + s_w0 + row, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      } else if (cm) {
        // rows 0 .. 14 and the last fifteen of this lane's read: low plane in bits 0 .. 14, high plane in bits 16 .. 30
        uint32_t fw = 0, bw = 0;
        if (coop) {
          const int sh = c_nal - PSSM_DEPTH;                                  // (>= 16)
          const unsigned long long bl = sh >= 64 ? (pv_l1 >> (sh - 64)) : ((pv_l0 >> sh) | ((pv_l1 << 1) << (63 - sh)));
          const unsigned long long bh = sh >= 64 ? (pv_h1 >> (sh - 64)) : ((pv_h0 >> sh) | ((pv_h1 << 1) << (63 - sh)));
          fw = ((uint32_t)pv_l0 & 0x7FFFu) | (((uint32_t)pv_h0 & 0x7FFFu) << 16);
          bw = ((uint32_t)bl & 0x7FFFu) | (((uint32_t)bh & 0x7FFFu) << 16);
        }
        // lane = row: the lower half of the wavefront takes the front rows, the upper half the back rows; each half's transposition
        // knows its own 32 lanes, the partner lane (xor 32) has the other 32
        const uint32_t tf = wave_transpose32(fw, lane), tk = wave_transpose32(bw, lane);
        const uint32_t got = (uint32_t)__shfl_xor((int)(lane < 32 ? tk : tf), 32);
        const unsigned long long M = lane < 32 ? ((unsigned long long)tf | ((unsigned long long)got << 32)) : ((unsigned long long)got | ((unsigned long long)tk << 32));
        const unsigned long long Mh = (unsigned long long)__shfl_xor((long long)M, 16);       // row q's high plane sits sixteen lanes up
        const bool row_lane = (lane & 31) < PSSM_DEPTH;
        const int dcode = lane < 32 ? lane : lane - 32 + PSSM_DEPTH + 1;
        // runs: neighbouring lanes with the same start and length
        const int key = (c_w0 << 8) | c_nal;
        const int pkey = __shfl_up(key, 1);
        const bool pcoop = __shfl_up((int)coop, 1) != 0;
        unsigned long long lm = __ballot(coop && (lane == 0 || !pcoop || pkey != key));
        const unsigned long long brk = lm | ~cm;
        typedef __attribute__((address_space(3))) unsigned long long lds_u64r;
        while (lm) {
          const int a0 = __builtin_ctzll(lm);
          lm &= lm - 1;
          const unsigned long long above = a0 == 63 ? 0ull : (~0ull << (a0 + 1));
          const unsigned long long nxt = brk & above;
          const int e0 = nxt ? __builtin_ctzll(nxt) : 64;
          const unsigned long long seg = (e0 == 64 ? ~0ull : ((1ull << e0) - 1ull)) & (~0ull << a0);
          const int s_w0 = __builtin_amdgcn_readlane(c_w0, a0), s_nal = __builtin_amdgcn_readlane(c_nal, a0);
          if (row_lane) {
            const unsigned long long lo = M & seg, hi = Mh & seg;
            const int cT = __popcll(lo & hi), cC = __popcll(lo & ~hi), cG = __popcll(hi & ~lo), cA = __popcll(seg) - cT - cC - cG;
            const unsigned long long* tab = run_tab + dcode * 8;
            const unsigned long long v1 = (unsigned long long)cA * tab[0] + (unsigned long long)cC * tab[2] + (unsigned long long)cG * tab[4] + (unsigned long long)cT * tab[6];
            const unsigned long long v2 = (unsigned long long)cA * tab[1] + (unsigned long long)cC * tab[3] + (unsigned long long)cG * tab[5] + (unsigned long long)cT * tab[7];
            const int wc = s_w0 + (lane < 32 ? lane : s_nal - PSSM_DEPTH + (lane - 32));
            (void)__hip_atomic_fetch_add((lds_u64r*)pk1 + wc, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add((lds_u64r*)pk2 + wc, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    // Proven-diagonal reads that run over the origin (two records: front in this window, back at the start of the
    // reference) all start in the last bucket, where nine reads in ten are of this kind: one per lane as well, every base
    // through the general path's rules (tally_one_read's emit: record, depth-code validity, dropped flags, multiplicity;
    // outside the LDS window the scores are added explicitly).
    if (have && !fast && !(dbg & 8u)) {
      const int4* tr4 = reinterpret_cast<const int4*>(rec_params + (int64_t)i * 16);
      const int4 a = tr4[0], b4 = tr4[1], c4 = tr4[2], d4 = tr4[3];
      const int flags = a.w;
      const int len2 = a.z & 0xFFFF, abr = (int)(int16_t)((uint32_t)a.z >> 16);
      const RecGeom g = rec_geom(a.x, a.y, L);
      const int n_al = len2 - abr, w0 = g.start_w - win_base, actF = b4.z;
      const bool wrap2 = (flags & TRF_SK) && !(flags & TRF_TOO_LONG) && (flags & TRF_DIAG) && g.split && g.start_w < L && w0 >= 0 &&
                         w0 + g.ncols_f <= TALLY_WIN && n_al == g.ncols_f + g.ncols_b && n_al > 0 && actF == g.ncols_f;
      if (wrap2) {
        fast = true;
        TALLY_KIND(2);
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(rs.packed + (uint32_t)b4.y);
        const bool dF = (flags & TRF_DF) != 0, dB = (flags & TRF_DB) != 0;
        const int fBase = c4.x, fOff = c4.y, fB = c4.z, fMult = c4.w, bBase = d4.x, bOff = d4.y, bB = d4.z, bMult = d4.w;
        uint32_t word = 0;
        int bad = 0;
        // coverage and span of the two records as range counts (see cov_diff) when both are listed once and the back
        // record fits the window's circular part; per column otherwise
        const int sb = tally_slot(0, win_base, Lp);
        const bool ranges = fMult == 1 && bMult == 1 && sb >= 0 && sb + g.ncols_b <= TALLY_WIN;
        for (int act = 0; act < n_al; act++) {
          const int r = abr + act;
          if (act == 0 || (r & 7) == 0) word = rp[r >> 3];
          const int code = (int)((word >> ((r & 7) * 4)) & 15u);
          const bool back = act >= g.ncols_f;
          const int p = back ? act - g.ncols_f : act, gc = back ? p : g.start_w + act, mult = back ? bMult : fMult;
          const bool dropped = back ? dB : dF;
          const int aa = back ? bOff + (act - actF) : fOff + act;
          const int d = depth_code((back ? bBase : fBase) + aa, (back ? bB : fB) - aa - 1);
          if (gc < 0 || gc >= Lp || d < 0 || d > 2 * PSSM_DEPTH) { bad = 1; continue; }
          const int wc = tally_slot(gc, win_base, Lp);
          if (wc >= 0 && wc < TALLY_WIN) {
            lds_i32* t = (lds_i32*)lds + wc;
            if (!dropped) {
              if (!ranges) aadd(&t[T_COV * TALLY_WIN], mult);
              for (int m = 0; m < mult; m++) add_base(wc, code, d, (flags & TRF_RC) != 0, mult == 1);
            }
            if (!ranges && p > 0) aadd(&t[T_SPAN * TALLY_WIN], mult);
          } else {
            int32_t* t = tb.tally + gc;
            if (!dropped) {
              const int16_t* row = pssm_lds + (LINEAR ? 0 : ((flags & TRF_RC) ? PSSM_WORDS : 0) + d * 25) + code;
              atomicAdd(&t[T_COV * Lp], mult);
              if (code < 4) atomicAdd(&t[(T_A + code) * Lp], mult);
              atomicAdd(&t[T_SA * Lp], mult * (int)row[0]);
              atomicAdd(&t[T_SC * Lp], mult * (int)row[5]);
              atomicAdd(&t[T_SG * Lp], mult * (int)row[10]);
              atomicAdd(&t[T_ST * Lp], mult * (int)row[15]);
            }
            if (p > 0) atomicAdd(&t[T_SPAN * Lp], mult);
          }
        }
        if (ranges) {
          auto range = [&](lds_i32* diff, int from, int to) {          // +1 on slots from .. to-1
            if (to <= from) return;
            aadd(diff + from, 1);
            if (to < TALLY_WIN) aadd(diff + to, -1);
          };
          if (!dF) range((lds_i32*)cov_diff, w0, w0 + g.ncols_f);
          if (!dB) range((lds_i32*)cov_diff, sb, sb + g.ncols_b);
          range((lds_i32*)span_diff, w0 + 1, w0 + g.ncols_f);            // start < pos <= end of each record, dropped or not
          range((lds_i32*)span_diff, sb + 1, sb + g.ncols_b);
        }
        if (bad) atomicOr(tb.flags, 2u);
      }
    }
    // Second most common: one gap (k_band_align says where), otherwise as above, counts only (linear).  Also one read per
    // lane: the bases before the gap, the gap -- deleted reference columns count as '-', inserted read rows become insert
    // events (src/map_align.c:444-510 through the general path's event format) --, the bases after it.
    if (have && !fast && !(dbg & 8u)) {
      const int4* tr4 = reinterpret_cast<const int4*>(rec_params + (int64_t)i * 16);
      const int4 a = tr4[0], b4 = tr4[1], c4 = tr4[2];
      const int flags = a.w;
      const int len2 = a.z & 0xFFFF, abr = (int)(int16_t)((uint32_t)a.z >> 16);
      const RecGeom g = rec_geom(a.x, a.y, L);
      const uint32_t desc = (uint32_t)b4.w;
      const int ins = (int)(desc & 1u), grow = (int)((desc >> 1) & 511u), gn = (int)((desc >> 10) & 63u);
      const int n_al = len2 - abr, ncol = ins ? n_al - gn : n_al + gn, w0 = g.start_w - win_base;
      const int fBase = c4.x, fOff = c4.y, fB = c4.z, fMult = c4.w;
      const bool one = (flags & TRF_SK) && !(flags & TRF_TOO_LONG) && (flags & TRF_ONEGAP) && !(flags & TRF_DIAG) && !g.split && fMult == 1 &&
                       fBase == 0 && fOff == 0 && w0 >= 0 && ncol > 0 && w0 + ncol <= TALLY_WIN && g.start_w + ncol <= Lp && ncol == g.ncols_f &&
                       gn > 0 && grow > abr && grow + (ins ? gn : 0) < len2;
      if (one) {
        fast = true;
        TALLY_KIND(3);
        if (!(dbg & 2048u)) {
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(rs.packed + (uint32_t)b4.y);
        const bool dF = (flags & TRF_DF) != 0;
        lds_i32* t = (lds_i32*)lds + w0;
        const int bad = n_al > PSSM_DEPTH + 1 && fB < n_al;            // (depth codes have to be valid: act <= 15 always is, beyond that fB - act - 1 >= 0)
        uint32_t word = 0;
        const bool sliced = LINEAR && bs_on && abr == 0 && len2 <= 128 && w0 + ncol <= 64 * BS_W && (flags & TRF_NO_N);
        if (sliced) {
          TALLY_KIND(4);
          // the two stretches of the read either side of its gap through the vertical counters; the gap itself as before
          if (!dF) {
            const uint64_t* pl = rplanes + (int64_t)i * 2 * rplane_words;
            const unsigned long long l0 = pl[0], l1 = rplane_words > 1 ? pl[1] : 0ull, h0 = pl[rplane_words], h1 = rplane_words > 1 ? pl[rplane_words + 1] : 0ull;
            const int r2 = ins ? grow + gn : grow;                       // first row behind the gap, at column w0 + grow (+ gn behind a deletion)
            bs_count(l0, l1, h0, h1, 0, grow, w0);
            bs_count(l0, l1, h0, h1, r2, len2 - r2, w0 + grow + (ins ? 0 : gn));
            if (!ins) for (int q = 0; q < gn; q++) aadd(&t[T_GAP * TALLY_WIN + grow + q], 1);
          }
          if (ins)
            for (int j = 0; j < gn; j++) {
              const int r = grow + j;
              const int code = (int)((rp[r >> 3] >> ((r & 7) * 4)) & 15u);
              const int gc = g.start_w + grow, act = grow + gn;
              if (j == 0) atomicMax(&tb.gaps[gc], gn);
              const int d = depth_code(act, fB - act - 1);
              const uint64_t ev = (uint64_t)(uint32_t)gc | ((uint64_t)j << 32) | ((uint64_t)code << 42) | ((uint64_t)(d & 31) << 45) |
                                  ((uint64_t)((flags & TRF_RC) ? 1 : 0) << 50);
              const int slot = atomicAdd(&ev_cnt, 1);
              if (slot < TALLY_EV_CAP) ev_buf[slot] = ev;
              else {
                const int e = atomicAdd(tb.n_events, 1);
                if (e < tb.cap_events) tb.events[e] = ev; else atomicOr(tb.flags, 1u);
              }
            }
        } else if (!LINEAR) {
          // the bases went through the gap-free reads' loop above; here only the gap: deleted reference columns count as '-',
          // inserted read rows become insert events at the column that follows
          const int o = grow - abr;
          if (!ins) { if (!dF) for (int q = 0; q < gn; q++) aadd(&t[T_GAP * TALLY_WIN + o + q], 1); }
          else
            for (int j = 0; j < gn; j++) {
              const int r = grow + j;
              const int code = (int)((rp[r >> 3] >> ((r & 7) * 4)) & 15u);
              const int gc = g.start_w + o, act = grow + gn - abr;
              if (j == 0) atomicMax(&tb.gaps[gc], gn);
              const int d = depth_code(act, fB - act - 1);
              const uint64_t ev = (uint64_t)(uint32_t)gc | ((uint64_t)j << 32) | ((uint64_t)code << 42) | ((uint64_t)(d & 31) << 45) |
                                  ((uint64_t)((flags & TRF_RC) ? 1 : 0) << 50);
              const int slot = atomicAdd(&ev_cnt, 1);
              if (slot < TALLY_EV_CAP) ev_buf[slot] = ev;
              else {
                const int e = atomicAdd(tb.n_events, 1);
                if (e < tb.cap_events) tb.events[e] = ev; else atomicOr(tb.flags, 1u);
              }
            }
        } else
        for (int r = abr; r < len2; r++) {
          if (r == abr || (r & 7) == 0) word = rp[r >> 3];
          const int code = (int)((word >> ((r & 7) * 4)) & 15u);
          if (r == grow && !ins) {                                     // deleted reference columns: '-' (covered, not a base)
            for (int q = 0; q < gn; q++) { if (!dF) aadd(&t[T_GAP * TALLY_WIN], 1); t++; }
          }
          if (ins && r >= grow && r < grow + gn) {                     // an inserted base: an event at the column that follows
            const int o = grow - abr, gc = g.start_w + o, act = grow + gn - abr, j = r - grow;
            if (j == 0) atomicMax(&tb.gaps[gc], gn);
            const int d = depth_code(act, fB - act - 1);
            const uint64_t ev = (uint64_t)(uint32_t)gc | ((uint64_t)j << 32) | ((uint64_t)code << 42) | ((uint64_t)(d & 31) << 45) |
                                ((uint64_t)((flags & TRF_RC) ? 1 : 0) << 50);
            const int slot = atomicAdd(&ev_cnt, 1);
            if (slot < TALLY_EV_CAP) ev_buf[slot] = ev;
            else {
              const int e = atomicAdd(tb.n_events, 1);
              if (e < tb.cap_events) tb.events[e] = ev; else atomicOr(tb.flags, 1u);
            }
            continue;
          }
          if (!dF && LINEAR) {                                                   // (!LINEAR: the bases went through the gap-free reads' loop above)
            const int act = r - abr, d = depth_code(act, fB - act - 1);          // inserted rows count as read bases (src/fsdb.c:568-581)
            add_base((int)(t - ((lds_i32*)lds)), code, d < 0 ? 0 : (d > 2 * PSSM_DEPTH ? 2 * PSSM_DEPTH : d), (flags & TRF_RC) != 0);
          }
          t++;
        }
        if (!dF) { aadd((lds_i32*)cov_diff + w0, 1); if (w0 + ncol < TALLY_WIN) aadd((lds_i32*)cov_diff + w0 + ncol, -1); }
        if (ncol > 1) { aadd((lds_i32*)span_diff + w0 + 1, 1); if (w0 + ncol < TALLY_WIN) aadd((lds_i32*)span_diff + w0 + ncol, -1); }
        if (bad) atomicOr(tb.flags, 2u);
        }
      }
    }
    // The reads left over are taken one per wavefront and one after the other; each begins with loads of its record,
    // script and bases.  The record is already in the registers of the lane that owns the read and is handed over
    // through LDS; script and bases are touched by all owning lanes first, side by side, so that the serial part finds
    // them in L2.
    if (DEFER) {
      const bool mine = have && !fast && !(dbg & 32u);
      const unsigned long long m = __ballot(mine);
      if (m) {                                              // one reservation per wavefront
        int base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(n_gen, __builtin_popcountll(m));
        base = __shfl(base, __builtin_ctzll(m));
        if (mine) { TALLY_KIND(5); gen_list[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i; }
      }
      continue;
    }
    int4 r0 = make_int4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
    if (have && !fast) {
      TALLY_KIND(5);
      if (reinterpret_cast<const int4*>(rec_params + (int64_t)i * 16)[0].w & TRF_ONEGAP) TALLY_KIND(6);
      if (reinterpret_cast<const int4*>(rec_params + (int64_t)i * 16)[0].w & TRF_DIAG) TALLY_KIND(7);
      const int4* tr4 = reinterpret_cast<const int4*>(rec_params + (int64_t)i * 16);
      r0 = tr4[0]; r1 = tr4[1]; r2 = tr4[2]; r3 = tr4[3];
      const int16_t* cl = rs.cols + (int64_t)i * rs.stride;
      const uint32_t* rp0 = reinterpret_cast<const uint32_t*>(rs.packed + (uint32_t)r1.y);
      const int n2 = r0.z & 0xFFFF;
      warm += (int)cl[0] + (int)cl[n2 > 64 ? 64 : 0] + (int)cl[n2 > 128 ? 128 : 0] + (int)cl[n2 > 192 ? 192 : 0] + (int)rp0[0] +
              (int)rp0[n2 > 128 ? 16 : 0];
    }
    unsigned long long todo = __ballot(have && !fast && !(dbg & 32u));
    while (todo) {
      const int l = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int ii = __shfl(i, l);
      if (lane == l) {
        int4* st = reinterpret_cast<int4*>(rec_stage[wv]);
        st[0] = r0; st[1] = r1; st[2] = r2; st[3] = r3;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the record is in LDS before anyone reads it (one wavefront)
      __builtin_amdgcn_wave_barrier();
      tally_one_read<true>(__builtin_amdgcn_readfirstlane(ii), lane, rs, ref, pssm2, drop_front, drop_back, tb, lds, win_base, rec_params, rec_actf, pssm_lds, ev_buf, &ev_cnt,
                           linear ? n_cnt : nullptr, rec_stage[wv]);
      __builtin_amdgcn_wave_barrier();
    }
  }
  TALLY_CLK(1);
  if (bs_on && !TALLY_ABL(1u << 19)) {
    // the wavefront's 64 x 16 vertical counters -> one 8-bit item per group of four lanes -> the window
    auto xch = [&](unsigned long long v, int m) -> unsigned long long { return (unsigned long long)__shfl_xor((long long)v, m); };
    // stage A (lane bit 5): items t and t + 8, 2-bit + 2-bit -> 3 bits; lane bit set keeps the upper half
    unsigned long long a3[8][3];
    {
      const bool hi = (lane & 32) != 0;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const unsigned long long k0 = hi ? bs0[t + 8] : bs0[t], k1 = hi ? bs1[t + 8] : bs1[t];
        const unsigned long long s0 = hi ? bs0[t] : bs0[t + 8], s1 = hi ? bs1[t] : bs1[t + 8];
        const unsigned long long r0 = xch(s0, 32), r1 = xch(s1, 32);
        const unsigned long long c0 = k0 & r0, x1 = k1 ^ r1;
        a3[t][0] = k0 ^ r0; a3[t][1] = x1 ^ c0; a3[t][2] = (k1 & r1) | (c0 & x1);
      }
    }
    // stage B (lane bit 4): 3 + 3 -> 4 bits
    unsigned long long a4[4][4];
    {
      const bool hi = (lane & 16) != 0;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        unsigned long long carry = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const unsigned long long k = hi ? a3[t + 4][q] : a3[t][q], sd = hi ? a3[t][q] : a3[t + 4][q], r = xch(sd, 16), x = k ^ r;
          a4[t][q] = x ^ carry; carry = (k & r) | (carry & x);
        }
        a4[t][3] = carry;
      }
    }
    // stage C (lane bit 3): 4 + 4 -> 5 bits
    unsigned long long a5[2][5];
    {
      const bool hi = (lane & 8) != 0;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        unsigned long long carry = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const unsigned long long k = hi ? a4[t + 2][q] : a4[t][q], sd = hi ? a4[t][q] : a4[t + 2][q], r = xch(sd, 8), x = k ^ r;
          a5[t][q] = x ^ carry; carry = (k & r) | (carry & x);
        }
        a5[t][4] = carry;
      }
    }
    // stage D (lane bit 2): 5 + 5 -> 6 bits; then two plain exchanges inside the group of four: 7, 8 bits
    unsigned long long a8[8];
    {
      const bool hi = (lane & 4) != 0;
      unsigned long long carry = 0;
#pragma unroll
      for (int q = 0; q < 5; q++) {
        const unsigned long long k = hi ? a5[1][q] : a5[0][q], sd = hi ? a5[0][q] : a5[1][q], r = xch(sd, 4), x = k ^ r;
        a8[q] = x ^ carry; carry = (k & r) | (carry & x);
      }
      a8[5] = carry; a8[6] = 0; a8[7] = 0;
#pragma unroll
      for (int st = 0; st < 2; st++) {
        const int bits = 6 + st;
        carry = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          if (q < bits) {
            const unsigned long long k = a8[q], r = xch(k, st == 0 ? 2 : 1), x = k ^ r;
            a8[q] = x ^ carry; carry = (k & r) | (carry & x);
          }
        }
        a8[bits] = carry;
      }
    }
    // this lane's item: t = (lane >> 2) & 15 = base * BS_W + word; its 16 columns: (lane & 3) * 16 ...
    const int item = (lane >> 2) & 15, bx_ = item / BS_W, wd = item % BS_W, c0 = (lane & 3) * 16;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      int v = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) v |= (int)((a8[q] >> (c0 + j)) & 1ull) << q;
      if (v) {
        if (LINEAR) aadd((lds_i32*)lds + (T_A + bx_) * TALLY_WIN + wd * 64 + c0 + j, v);
        else aadd((lds_i32*)mid_cnt + bx_ * TALLY_WIN + wd * 64 + c0 + j, wg_rc ? v << 16 : v);      // (depth code 15, this workgroup's strand: see mid_cnt)
      }
    }
  }
  if (warm == 0x7FFFFFF1) atomicOr(tb.flags, 4u);   // keeps the warming loads alive; never true for real data
  TALLY_CLK(2);
  __syncthreads();
  TALLY_CLK(3);
  // The window goes to this workgroup's slab with plain stores; k_tally_reduce adds the slabs up.  (Flushing with
  // global atomics cost more than the tally itself: ~9 M device-scope atomics per 1 M reads, and the eight XCDs
  // share no L2, so every one of them is resolved at the memory side.)
  // the buffered insert events: one reservation for all of them -- asked for here, used at the very end (a returning atomic on ONE
  // address from two thousand workgroups: a few microseconds each that the rest of the flush does not have to wait for)
  const int ne = ev_cnt < TALLY_EV_CAP ? ev_cnt : TALLY_EV_CAP;
  int ev_base_reg = 0;
  if (threadIdx.x == 255 && ne > 0) ev_base_reg = atomicAdd(tb.n_events, ne);
  // prefix sums of the two difference arrays, folded into the cov / span rows: one wavefront each, six consecutive entries
  // per lane and a shuffle scan over the lanes' sums (a Hillis-Steele scan by the whole workgroup was eighteen barriers)
  static_assert(TALLY_WIN == 64 * 6, "the difference arrays are scanned six entries per lane");
  if (wv < 2) {
    int32_t* d = wv == 0 ? cov_diff : span_diff;
    int32_t e[6], sum = 0;
#pragma unroll
    for (int q = 0; q < 6; q++) { e[q] = d[lane * 6 + q]; sum += e[q]; }
    int32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    int32_t run = inc - sum;
#pragma unroll
    for (int q = 0; q < 6; q++) { run += e[q]; d[lane * 6 + q] = run; }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < TALLY_WIN; k += blockDim.x) { lds[T_COV * TALLY_WIN + k] += cov_diff[k]; lds[T_SPAN * TALLY_WIN + k] += span_diff[k]; }
  if (!LINEAR) {
    // the bases of depth code 15: their counts into the base rows, their scores from sm[strand][15][X][b]
    for (int k = threadIdx.x; k < TALLY_WIN; k += blockDim.x) {
      int sc[4] = {0, 0, 0, 0};
      for (int b5 = 0; b5 < 5; b5++) {
        const uint32_t w = (uint32_t)mid_cnt[b5 * TALLY_WIN + k];
        const int cf = (int)(w & 0xFFFFu), cr = (int)(w >> 16);
        if (b5 < 4) lds[(T_A + b5) * TALLY_WIN + k] += cf + cr;
        for (int x = 0; x < 4; x++)
          sc[x] += cf * (int)pssm_lds[PSSM_DEPTH * 25 + x * 5 + b5] + cr * (int)pssm_lds[PSSM_WORDS + PSSM_DEPTH * 25 + x * 5 + b5];
      }
      // ... and the end bases' packed sums: counts, and the scores less the bias every add carried
      const unsigned long long p1 = pk1[k], p2 = pk2[k];
      int cnt = 0;
      for (int b5 = 0; b5 < 4; b5++) { const int c = (int)((p2 >> (20 + 10 * b5)) & 1023ull); lds[(T_A + b5) * TALLY_WIN + k] += c; cnt += c; }
      const int off = pk_bias > 0 ? pk_bias * cnt : 0;
      sc[0] += (int)(p1 & 0xFFFFFull) - off; sc[1] += (int)((p1 >> 20) & 0xFFFFFull) - off; sc[2] += (int)((p1 >> 40) & 0xFFFFFull) - off;
      sc[3] += (int)(p2 & 0xFFFFFull) - off;
      lds[T_SA * TALLY_WIN + k] += sc[0]; lds[T_SC * TALLY_WIN + k] += sc[1]; lds[T_SG * TALLY_WIN + k] += sc[2]; lds[T_ST * TALLY_WIN + k] += sc[3];
    }
  }
  if (linear) {
    // the score rows from the counts: sm[X][b] of depth 0, forward table (src/map_align.c:258-261 adds sm[d][X][b] per base)
    for (int k = threadIdx.x; k < TALLY_WIN; k += blockDim.x) {
      int c[5];
      for (int b5 = 0; b5 < 4; b5++) c[b5] = lds[(T_A + b5) * TALLY_WIN + k];
      c[4] = n_cnt[k];
      const int rows[4] = {T_SA, T_SC, T_SG, T_ST};
      for (int x = 0; x < 4; x++) {
        int acc = 0;
        for (int b5 = 0; b5 < 5; b5++) acc += c[b5] * (int)pssm_lds[x * 5 + b5];
        lds[rows[x] * TALLY_WIN + k] += acc;
      }
    }
  }
  if (threadIdx.x == 255) ev_base = ev_base_reg;             // (the first use of the atomic's result: only here is it waited for)
  __syncthreads();
  TALLY_CLK(4);
  int32_t* slab = slabs + (int64_t)blockIdx.x * ((TALLY_WORDS - 1) * TALLY_WIN);
  if (!TALLY_ABL(1u << 20))
  for (int k = threadIdx.x; k < (TALLY_WORDS - 1) * TALLY_WIN; k += blockDim.x) slab[k] = lds[k];
  for (int k = threadIdx.x; k < ne; k += blockDim.x) {
    if (ev_base + k < tb.cap_events) tb.events[ev_base + k] = ev_buf[k]; else atomicOr(tb.flags, 1u);
  }
  TALLY_CLK(5);
#ifdef MIA_HIP_ALT_PATHS
  if ((dbg & 131072u) && threadIdx.x == 0) atomicAdd(&g_tally_clk[7], 1ull);
#endif
}

// tally[word][gc] += sum of the windows that cover column gc: buckets floor(gc/128)-2 .. floor(gc/128), all their chunks,
// and the last buckets' windows where they wrap around to the start of the reference.
// Runs after k_tally_binned; nothing else writes the tally then, so plain read-modify-write.
// gen (gen.list != nullptr): the reads k_tally_binned<.., true> put aside are tallied by TALLY_GEN_BLOCKS further workgroups of this
// launch, one read per wavefront, straight into the tally -- every add of this kernel is an atomic, so the two kinds of workgroup
// do not have to wait for each other (a launch of their own behind this one was 22 us for seven hundred reads).
struct GenReads {
  ReadSet rs; RefInfo ref; const int32_t* pssm2; const uint8_t* drop_front; const uint8_t* drop_back; const int32_t* rec_params; const int32_t* rec_actf;
  const int32_t* list; const int32_t* n;
};
constexpr int TALLY_GEN_BLOCKS = 192;
constexpr int TALLY_REDUCE_SHARES = 4;     // gridDim.z of k_tally_reduce: the workgroups of a bucket are summed in this many interleaved shares
__global__ __launch_bounds__(256) void k_tally_reduce(TallyBuf tb, int32_t nb, const int32_t* wgoff, const int32_t* slabs, const int32_t* abort_if, GenReads gen, int32_t split = 0) {
  if (abort_if && *abort_if != 0) return;     // (mia_hip_iterate queued this launch before the alignment's exact-kernel count was known: see iterate_body)
  const int Lp = tb.Lp;
  // (the workgroups of the put-aside reads come FIRST in the grid: theirs is the longest chain of this launch -- list, record,
  // script, bases, atomics -- and workgroups are started in order)
  const int gen_blocks = gen.list ? TALLY_GEN_BLOCKS : 0;
  if ((int)blockIdx.x < gen_blocks) {
    if (blockIdx.y != 0 || blockIdx.z != 0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = *gen.n;
    for (int k = (int)blockIdx.x * 4 + wv; k < n; k += TALLY_GEN_BLOCKS * 4)
      tally_one_read<false>(__builtin_amdgcn_readfirstlane(gen.list[k]), lane, gen.rs, gen.ref, gen.pssm2, gen.drop_front, gen.drop_back, tb, nullptr, 0, gen.rec_params,
                            gen.rec_actf, nullptr, nullptr, nullptr);
    return;
  }
  // one thread per (column, tally word, share of the bucket's workgroups): blockIdx.y is the word, blockIdx.z the share (a
  // thread per column alone is 66 workgroups for a mitochondrion, each thread a chain of five hundred loads; with 256 reads per
  // tally workgroup a thread per column and word still had a hundred)
  const int gc = ((int)blockIdx.x - gen_blocks) * blockDim.x + threadIdx.x, w = blockIdx.y;
  const int zs = blockIdx.z, zn = gridDim.z;
  if (gc >= Lp) return;
  int acc = 0;
  nb >>= split;                                                // stretches of columns; the workgroups of stretch b: wgoff[b << split] .. wgoff[(b + 1) << split]
  const int bhi = min(gc / TALLY_BUCKET, nb - 1);
  for (int b = max(0, gc / TALLY_BUCKET - (TALLY_WIN / TALLY_BUCKET - 1)); b <= bhi; b++) {
    const int wc = gc - b * TALLY_BUCKET;
    if (wc < 0 || wc >= TALLY_WIN) continue;
    for (int wg = wgoff[b << split] + zs; wg < wgoff[(b + 1) << split]; wg += zn) acc += slabs[(int64_t)wg * ((TALLY_WORDS - 1) * TALLY_WIN) + w * TALLY_WIN + wc];
  }
  // ... and the circular part of the last buckets' windows (tally_slot): slot gc + Lp - b * TALLY_BUCKET
  for (int b = max(0, (Lp + gc - TALLY_WIN) / TALLY_BUCKET); b < nb; b++) {
    const int wc = gc + Lp - b * TALLY_BUCKET;
    if (wc < 0 || wc >= TALLY_WIN || gc >= b * TALLY_BUCKET) continue;     // (columns from win_base on sit in their direct slot)
    for (int wg = wgoff[b << split] + zs; wg < wgoff[(b + 1) << split]; wg += zn) acc += slabs[(int64_t)wg * ((TALLY_WORDS - 1) * TALLY_WIN) + w * TALLY_WIN + wc];
  }
  if (acc) (void)__hip_atomic_fetch_add(&tb.tally[w * Lp + gc], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- ma: tally of stored AlnSeq records (show_consensus, src/map_alignment.c:139-170) -------------
// One record per wavefront, one column per lane.  No `dropped` test: ma counts every record.
struct MaRecords {
  int64_t n;
  const int32_t* start;
  const uint8_t* revcom;
  const int64_t* col_off;   // [n+1]
  const char* seq;
  const char* smp;
};
__device__ __forceinline__ int ma_code(char b) {   // add_base / base2inx (src/map_align.c:16-29,229-263)
  return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : b == '-' ? 5 : 4;
}
__global__ __launch_bounds__(256) void k_ma_tally(MaRecords mr, const int32_t* pssm2, TallyBuf tb) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= mr.n) return;
  const int64_t o0 = mr.col_off[r];
  const int ncols = (int)(mr.col_off[r + 1] - o0), start = mr.start[r], Lp = tb.Lp;
  const int32_t* pm = pssm2 + (mr.revcom[r] ? PSSM_WORDS : 0);
  for (int p = lane; p < ncols; p += 64) {
    const int gc = start + p;
    if (gc < 0 || gc >= Lp) { atomicOr(tb.flags, 2u); continue; }
    const int code = ma_code(mr.seq[o0 + p]), d = (int)mr.smp[o0 + p] - 'A';
    int32_t* t = tb.tally + gc;
    atomicAdd(&t[T_COV * Lp], 1);
    if (code == 5) atomicAdd(&t[T_GAP * Lp], 1);
    else {
      if (d < 0 || d > 2 * PSSM_DEPTH) { atomicOr(tb.flags, 2u); continue; }
      if (code < 4) atomicAdd(&t[(T_A + code) * Lp], 1);
      const int32_t* row = pm + d * 25 + code;
      atomicAdd(&t[T_SA * Lp], row[0]);
      atomicAdd(&t[T_SC * Lp], row[5]);
      atomicAdd(&t[T_SG * Lp], row[10]);
      atomicAdd(&t[T_ST * Lp], row[15]);
    }
    if (p > 0) atomicAdd(&t[T_SPAN * Lp], 1);              // start < pos <= end (src/map_align.c:466-469)
  }
}
// INS_POS pairs -> the event list k_ins_tally consumes; one pair per thread (inserts are short)
__global__ void k_ma_ins_events(MaRecords mr, int64_t n_ins, const int32_t* ins_record, const int32_t* ins_pos, const int64_t* ins_off,
                                const char* ins_bases, TallyBuf tb) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ins) return;
  const int r = ins_record[e], p = ins_pos[e];
  const int64_t o0 = mr.col_off[r];
  const int ncols = (int)(mr.col_off[r + 1] - o0);
  if (p <= 0 || p >= ncols) return;                        // not "start < pos <= end": never looked at
  const int gc = mr.start[r] + p, d = (int)mr.smp[o0 + p] - 'A';
  if (gc >= tb.Lp || d < 0 || d > 2 * PSSM_DEPTH) { atomicOr(tb.flags, 2u); return; }
  const int len = (int)(ins_off[e + 1] - ins_off[e]);
  for (int j = 0; j < len && j < 1024; j++) {
    const int code = ma_code(ins_bases[ins_off[e] + j]);
    const int slot = atomicAdd(tb.n_events, 1);
    if (slot < tb.cap_events)
      tb.events[slot] = (uint64_t)(uint32_t)gc | ((uint64_t)j << 32) | ((uint64_t)(code == 5 ? 4 : code) << 42) | ((uint64_t)(d & 31) << 45) |
                        ((uint64_t)(mr.revcom[r] ? 1 : 0) << 50);
    else atomicOr(tb.flags, 1u);
  }
}

// ---- insert columns (find_ins_cons, src/map_align.c:444-510) ----------------------
// The buffer of the insert tallies for `cap` slots: nine 32-bit words per slot as find_ins_cons reads them, then -- 8-byte aligned -- three
// 64-bit words per slot that k_ins_tally adds to instead of five of the nine (ins_sc_of; call_inserts_at folds them back before it calls)
MIA_HD inline int64_t ins_words(int64_t cap) { return ((cap * 9 + 1) & ~(int64_t)1) + cap * 6; }
MIA_HD inline unsigned long long* ins_sc_of(int32_t* ins_tally, int64_t cap) { return reinterpret_cast<unsigned long long*>(ins_tally + ((cap * 9 + 1) & ~(int64_t)1)); }
// ins_off[pos] = sum of gaps[0..pos-1]; slot (pos, j) -> ins_off[pos] + j; 9 words per slot:
// A,C,G,T counts, number of reads with a base there, scoreA..scoreT
// cap: slots the insert buffers hold.  The host launches with the capacity left from the last call and reads the real total
// back with the results; only if the total outgrew the capacity does it enlarge the buffers and run the insert part again.
__global__ void k_ins_tally(const uint64_t* events, int32_t n_events, const int32_t* pssm2, const int32_t* ins_off,
                            const int32_t* gaps, int32_t L, int32_t* ins_tally, int32_t cap, const int32_t* n_events_dev, int32_t cap_events, const int32_t* abort_if = nullptr) {
  if (abort_if && *abort_if != 0) return;     // (mia_hip_iterate queued this launch before the alignment's exact-kernel count was known: see iterate_body)
  if (n_events_dev) n_events = min(*n_events_dev, cap_events);      // the count stayed on the device: a grid-stride sweep
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_events; e += gridDim.x * blockDim.x) {
  const uint64_t ev = events[e];
  const int gc = (int)(uint32_t)ev, j = (int)((ev >> 32) & 1023), code = (int)((ev >> 42) & 7), d = (int)((ev >> 45) & 31);
  const int rc = (int)((ev >> 50) & 1);
  if (gc <= 0 || gc >= L || j >= gaps[gc] || ins_off[gc] + j >= cap) continue;
  // A device-scope atomic is a trip to memory of its own on this part (eight L2s: 64 bytes written per add, 12 G adds/s whatever the
  // address), and an event used to cost six.  Three: the base's count (or "another character"); scores A and C in one 64-bit add, G and
  // T in another -- (c << 32) + a with a sign-extended: the low half of the sum is the sum of the a's as long as that fits 32 bits (it did
  // before: the words were 32 bits), the rest the sum of the c's, exactly.  The number of reads with a base here is the counts' sum.
  const int64_t sl = (int64_t)(ins_off[gc] + j);
  int32_t* t = ins_tally + sl * 9;
  unsigned long long* sc = ins_sc_of(ins_tally, cap) + sl * 3;
  const int32_t* row = pssm2 + (rc ? PSSM_WORDS : 0) + d * 25 + code;
  if (code < 4) atomicAdd(&t[code], 1);
  else atomicAdd(&sc[2], 1ull);
  atomicAdd(&sc[0], (unsigned long long)(((long long)row[5] << 32) + (long long)row[0]));
  atomicAdd(&sc[1], (unsigned long long)(((long long)row[15] << 32) + (long long)row[10]));
  }
}

// find_consensus (src/map_align.c:294-391)
__device__ __forceinline__ char call_base(int As, int Cs, int Gs, int Ts, int gapsn, int cov, int sA, int sC, int sG, int sT,
                                          int cons_code) {
  (void)As; (void)Cs; (void)Gs; (void)Ts;
  if (cov == 0) return 'N';
  if (((double)gapsn / (double)cov) >= (double)(50 / 100.0)) return '-';
  int top = sA, second = INT32_MIN;
  char base = 'A';
  if (sC >= top) { second = top; top = sC; base = 'C'; } else second = sC;
  if (sG >= top) { second = top; top = sG; base = 'G'; } else if (sG >= second) second = sG;
  if (sT >= top) { second = top; top = sT; base = 'T'; } else if (sT >= second) second = sT;
  if (cons_code == 2) return (top >= 0 || (top - 2400) > second) ? base : 'N';
  return (top >= -399) ? base : 'N';
}

__global__ void k_call_columns(const int32_t* tally, int32_t Lp, int32_t L, int cons_code, char* calls) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= L) return;
  calls[p] = call_base(tally[T_A * Lp + p], tally[T_C * Lp + p], tally[T_G * Lp + p], tally[T_T * Lp + p],
                       tally[T_GAP * Lp + p], tally[T_COV * Lp + p], tally[T_SA * Lp + p], tally[T_SC * Lp + p],
                       tally[T_SG * Lp + p], tally[T_ST * Lp + p], cons_code);
}
__device__ __forceinline__ void call_inserts_at(int p, const int32_t* tally, int32_t Lp, const int32_t* gaps, const int32_t* ins_off,
                                                int32_t* ins_tally, int cons_code, char* ins_calls, int32_t cap) {
  const int span = tally[T_SPAN * Lp + p];
  for (int j = 0; j < gaps[p] && ins_off[p] + j < cap; j++) {
    const int64_t sl = (int64_t)(ins_off[p] + j);
    int32_t* t = ins_tally + sl * 9;
    // (k_ins_tally's three 64-bit words folded back into the slot's own: whoever reads the nine words after the calls -- mia_hip_get_ins_tally
    // -- finds them as add_base's counterpart leaves them.  A slot belongs to one column, a column to one thread; folding twice changes nothing.)
    const unsigned long long* sc = ins_sc_of(ins_tally, cap) + sl * 3;
    const long long x0 = (long long)sc[0], x1 = (long long)sc[1];
    const int32_t sA = (int32_t)(uint32_t)x0, sC = (int32_t)((x0 - (long long)sA) >> 32), sG = (int32_t)(uint32_t)x1, sT = (int32_t)((x1 - (long long)sG) >> 32);
    t[4] = t[0] + t[1] + t[2] + t[3] + (int32_t)sc[2];
    t[5] = sA; t[6] = sC; t[7] = sG; t[8] = sT;
    ins_calls[ins_off[p] + j] = call_base(t[0], t[1], t[2], t[3], span - t[4], span, t[5], t[6], t[7], t[8], cons_code);
  }
}
__global__ void k_call_inserts(const int32_t* tally, int32_t Lp, int32_t L, const int32_t* gaps, const int32_t* ins_off,
                               int32_t* ins_tally, int cons_code, char* ins_calls, int32_t cap) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p <= 0 || p >= L) return;
  call_inserts_at(p, tally, Lp, gaps, ins_off, ins_tally, cons_code, ins_calls, cap);
}

}  // namespace mia
