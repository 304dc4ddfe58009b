// wave_dev.h -- the gfx950 "wave policy" for align_body.h: one value per lane,
// cross-lane traffic through DPP (no LDS round trips), wave64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mia_layout.h"

namespace mia {

// DPP controls (GFX9 encoding)
constexpr int DPP_ROW_SHR = 0x110;     // row_shr:n  = 0x110 + n
constexpr int DPP_WAVE_SHR1 = 0x138;   // wave_shr:1
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;

struct DevWave {
  typedef uint32_t U;
  typedef bool M;
  unsigned char* lds;  // this wave's LDS window

  __device__ __forceinline__ explicit DevWave(unsigned char* l) : lds(l) {}

  __device__ __forceinline__ U lane() const { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

  // value of lane-1; lane 0 receives `fill`
  __device__ __forceinline__ U shr1(U x, U fill) const {
    return (U)__builtin_amdgcn_update_dpp((int)fill, (int)x, DPP_WAVE_SHR1, 0xF, 0xF, false);
  }
  __device__ __forceinline__ static U umax(U a, U b) { return a > b ? a : b; }
  __device__ __forceinline__ static U umin(U a, U b) { return a < b ? a : b; }
  __device__ __forceinline__ static U umax3(U a, U b, U c) { return umax(umax(a, b), c); }
  __device__ __forceinline__ static U sel(M c, U a, U b) { return c ? a : b; }

  // inclusive prefix max (unsigned) across the 64 lanes: 4 row_shr steps + 2 row broadcasts
  __device__ __forceinline__ U scan_max(U v) const {
    v = umax(v, (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 1, 0xF, 0xF, false));
    v = umax(v, (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 2, 0xF, 0xF, false));
    v = umax(v, (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 4, 0xF, 0xF, false));
    v = umax(v, (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 8, 0xF, 0xF, false));
    v = umax(v, (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST15, 0xA, 0xF, false));
    v = umax(v, (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST31, 0xC, 0xF, false));
    return v;
  }
  __device__ __forceinline__ uint32_t reduce_max(U v) const { return (uint32_t)__builtin_amdgcn_readlane((int)scan_max(v), 63); }
  __device__ __forceinline__ uint32_t reduce_min(U v) const { return ~reduce_max(~v); }
  __device__ __forceinline__ uint64_t ballot(M m) const { return __builtin_amdgcn_ballot_w64(m); }
  __device__ __forceinline__ uint32_t lane_val(U v, int l) const { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
  __device__ __forceinline__ bool lane_bit(M m, int l) const { return (ballot(m) >> l) & 1ull; }

  __device__ __forceinline__ static U udiv5(U e) { return e / 5u; }
  __device__ __forceinline__ static U depth(U row, uint32_t len) { return (U)sm_depth((int)row, (int)len); }

  // global memory
  __device__ __forceinline__ static U gload_u8(const uint8_t* p, U i, M ok) { return ok ? (U)p[i] : 0u; }
  __device__ __forceinline__ static U gload_i32(const int32_t* p, U i, M ok) { return ok ? (U)p[i] : 0u; }
  __device__ __forceinline__ static void gstore_i16(int16_t* p, U i, U v, M ok) { if (ok) p[i] = (int16_t)(uint16_t)v; }

  // LDS (byte offsets inside this wave's window)
  __device__ __forceinline__ void lds_w16(U off, U v, M ok) const { if (ok) *reinterpret_cast<uint16_t*>(lds + off) = (uint16_t)v; }
  __device__ __forceinline__ void lds_w32(U off, U v, M ok) const { if (ok) *reinterpret_cast<uint32_t*>(lds + off) = v; }
  __device__ __forceinline__ U lds_ri16(U off) const { return (U)(int32_t)*reinterpret_cast<const int16_t*>(lds + off); }
  __device__ __forceinline__ U lds_r8(U off, M ok) const { return ok ? (U)lds[off] : 0u; }
  // order this wave's LDS writes before its later LDS reads (other lanes' data)
  __device__ __forceinline__ void lds_fence() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
};

}  // namespace mia
