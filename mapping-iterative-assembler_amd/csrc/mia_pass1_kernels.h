// mia_pass1_kernels.h -- pass 1 on gfx950: the body of the read loop of main()
// (/root/reference/src/mia_main.c:759-805): new_kmer_filter (src/kmer.c:239-331)
// followed by sg_align (src/mia.c:1500-1665), one read per wavefront, persistent grid.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
