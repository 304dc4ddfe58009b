// mia_iter_kernels.h -- the small kernels that let one whole iteration of MIA's main loop (reference
// /root/reference/src/mia_main.c:915-964) run without the host in between (mia_hip_iterate):
//   k_ref_encode      the new reference as the aligner sees it: base codes of the ASCII string, wrap appended
//                     (make_ref_upper / add_ref_wrap / base2inx: src/mia.c:642-689, src/map_align.c:16-29)
//   k_cons_assemble   the string consensus_assembly_string returns (src/mia.c:551-600): insert-column calls, then the
//                     column's own call, '-' left out -- from the per-column calls, by one workgroup
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mia {

__device__ __forceinline__ uint8_t dev_base_code(char b) {   // only upper-case ACGT are bases
  return b == 'A' ? 0 : (b == 'C' ? 1 : (b == 'G' ? 2 : (b == 'T' ? 3 : 4)));
}

// codes[p] for p < total: the L bases, then the first wl of them again, then padding (code 4)
__global__ __launch_bounds__(256) void k_ref_encode(const char* ascii, int32_t L, int32_t wl, uint8_t* codes, int32_t total) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= total) return;
  codes[p] = p < L ? dev_base_code(ascii[p]) : (p < L + wl ? dev_base_code(ascii[p - L]) : (uint8_t)4);
}

enum { CH_LEN = 0, CH_INS_TOTAL, CH_OVERFLOW, CH_N_EVENTS, CH_TALLY_FLAGS, CH_CULL_FLAGS, CH_WORDS = 8 };

// One 1024-thread workgroup; each of its 16 wavefronts owns a contiguous stretch of columns and walks it 64 at a time.
// pos[Lp]: scratch (where each column's characters start).  hdr[CH_LEN] = strlen, hdr[CH_INS_TOTAL] = insert columns in
// all, hdr[CH_OVERFLOW] = 1 if the insert buffers (ins_cap slots) or `out` (out_cap bytes incl. the terminator) were too
// small: the host then repeats the consensus with larger buffers.
__global__ __launch_bounds__(1024) void k_cons_assemble(const char* calls, const char* ins_calls, const int32_t* gaps, const int32_t* ins_off, int32_t L,
                                                         int32_t ins_cap, const int32_t* ins_total, int32_t* pos, char* out, int32_t out_cap, int32_t* hdr,
                                                         const int32_t* n_events, const uint32_t* tally_flags, const uint32_t* cull_flags) {
  __shared__ int32_t wsum[16];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int per = ((L + 15) / 16 + 63) & ~63;
  const int lo = w * per, hi = min(lo + per, L);
  const int total = *ins_total;
  const bool ins_ok = total <= ins_cap;
  auto emits = [](char c) { return c != '-' && c != ' '; };
  int32_t carry = 0;
  for (int base = lo; base < hi; base += 64) {
    const int p = base + lane;
    int32_t v = 0;
    if (p < hi) {
      if (p > 0 && ins_ok) for (int j = 0; j < gaps[p]; j++) v += emits(ins_calls[ins_off[p] + j]) ? 1 : 0;
      v += emits(calls[p]) ? 1 : 0;
    }
    int32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (p < hi) pos[p] = carry + inc - v;
    carry += __shfl(inc, 63);
  }
  if (lane == 0) wsum[w] = carry;
  __syncthreads();
  int32_t off = 0, len = 0;
  for (int k = 0; k < 16; k++) { if (k < w) off += wsum[k]; len += wsum[k]; }
  const bool fits = len + 1 <= out_cap;
  if (fits)
    for (int p = lo + lane; p < hi; p += 64) {
      int32_t o = pos[p] + off;
      if (p > 0 && ins_ok) for (int j = 0; j < gaps[p]; j++) { const char c = ins_calls[ins_off[p] + j]; if (emits(c)) out[o++] = c; }
      const char c = calls[p];
      if (emits(c)) out[o++] = c;
    }
  if (t == 0) {
    if (fits) out[len] = 0;
    hdr[CH_LEN] = len; hdr[CH_INS_TOTAL] = total; hdr[CH_OVERFLOW] = (ins_ok && fits) ? 0 : 1;
    hdr[CH_N_EVENTS] = n_events ? *n_events : 0;
    hdr[CH_TALLY_FLAGS] = tally_flags ? (int32_t)*tally_flags : 0;
    hdr[CH_CULL_FLAGS] = cull_flags ? (int32_t)*cull_flags : 0;
  }
}

}  // namespace mia
