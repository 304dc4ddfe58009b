// mia_peak_kernels.h -- the two ceilings bench.py prices the hot path against, MEASURED on the device it runs on
// ("confirm with a device copy benchmark on the box", "measure with a micro-kernel"):
//   k_peak_copy   streaming copy of a large buffer, 16 bytes per lane per step: HBM read + write bandwidth
//   k_peak_valu   eight independent chains of v_max3_i32 / v_add_u32 per lane, the DP kernels' own instruction mix:
//                 wave-instructions per second the 1 024 SIMDs issue when nothing else is in the way
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mia {

// U independent 16-byte loads in flight per lane before the first store (one load per step leaves the memory pipe of a CU
// half empty: 4.7 TB/s); NT: non-temporal loads and stores -- a stream that is read once should not push anything out of
// the caches.  mia_hip_measure_peaks runs the variants and keeps the fastest: it is a ceiling, not a kernel of the product.
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_peak_copy(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (NT) {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(in + i + u * stride);
        const unsigned long long lo = __builtin_nontemporal_load(q), hi = __builtin_nontemporal_load(q + 1);
        v[u] = make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
      } else v[u] = in[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (NT) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(out + i + u * stride);
        __builtin_nontemporal_store(((unsigned long long)v[u].y << 32) | v[u].x, q);
        __builtin_nontemporal_store(((unsigned long long)v[u].w << 32) | v[u].z, q + 1);
      } else out[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride) out[i] = in[i];
}

constexpr int PEAK_VALU_OPS_PER_ITER = 16 * 4;   // 8 chains x (max3 + add), unrolled four times
__global__ __launch_bounds__(256) void k_peak_valu(int32_t* out, int iters, int32_t seed) {
  int32_t a0 = seed + (int)threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const int32_t b = seed ^ 0x55, c = seed - 77, k = 3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#define MIA_PEAK_STEP(A)                                                        \
  asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(A) : "v"(b), "v"(c));         \
  asm volatile("v_add_u32 %0, %0, %1" : "+v"(A) : "v"(k));
      MIA_PEAK_STEP(a0) MIA_PEAK_STEP(a1) MIA_PEAK_STEP(a2) MIA_PEAK_STEP(a3)
      MIA_PEAK_STEP(a4) MIA_PEAK_STEP(a5) MIA_PEAK_STEP(a6) MIA_PEAK_STEP(a7)
#undef MIA_PEAK_STEP
    }
  }
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// The pure issue rate (k_peak_valu's max3 / add chains are the DP kernels' mix and reach 60 % of the guide's
// 2-cycle bound -- whether that is the instruction's cost or the clock cannot be told from outside): sixteen INDEPENDENT v_add_u32 per
// round, nothing waits for anything, and the shader clock the kernel itself ran at -- s_memtime (shader clock) over s_memrealtime
// (constant 100 MHz), taken by one lane per workgroup around the loop.
constexpr int PEAK_ISSUE_OPS_PER_ITER = 16 * 4;
__global__ __launch_bounds__(256) void k_peak_issue(int32_t* out, int iters, int32_t seed, unsigned long long* clocks) {
  int32_t a[16];
#pragma unroll
  for (int q = 0; q < 16; q++) a[q] = seed + (int)threadIdx.x + q;
  const int32_t k = 3;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int q = 0; q < 16; q++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[q]) : "v"(k));
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int32_t x = 0;
#pragma unroll
  for (int q = 0; q < 16; q++) x ^= a[q];
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = x;
  if (threadIdx.x == 0 && clocks) { clocks[2 * (int64_t)blockIdx.x] = c1 - c0; clocks[2 * (int64_t)blockIdx.x + 1] = r1 - r0; }
}

}  // namespace mia
