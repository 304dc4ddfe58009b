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

}  // namespace mia
