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
  unsigned char* lds;    // this wave's LDS window (substitution table)
  unsigned char* trace;  // this workgroup's trace slab in global memory

  __device__ __forceinline__ DevWave(unsigned char* l, unsigned char* t) : lds(l), trace(t) {}

  __device__ __forceinline__ U lane() const { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

  // value of lane-1; lane 0 receives `fill`
  __device__ __forceinline__ U shr1(U x, U fill) const {
    return (U)__builtin_amdgcn_update_dpp((int)fill, (int)x, DPP_WAVE_SHR1, 0xF, 0xF, false);
  }
  // max(value of lane-1, unav); lane 0 receives unav (one v_max_u32_dpp with zero fill)
  __device__ __forceinline__ U shr1_max(U x, U unav) const {
    U s = (U)__builtin_amdgcn_update_dpp(0, (int)x, DPP_WAVE_SHR1, 0xF, 0xF, true);
    return s > unav ? s : unav;
  }
  __device__ __forceinline__ static U sext_lo(U x) { return (U)(((int32_t)(x << 16)) >> 16); }
  __device__ __forceinline__ static U sext_hi(U x) { return (U)(((int32_t)x) >> 16); }
  // 4 packed score words -> 4 trace bytes [type:2][len:6, saturating at 63]
  template <int IB>
  __device__ __forceinline__ static U trace_pack4(U b0, U b1, U b2, U b3) {
    if constexpr (IB == 8) {
      // low 10 bits of each word = [type:2][len:8]; two words per dword, packed 16-bit min saturates len
      typedef unsigned short us2 __attribute__((ext_vector_type(2)));
      const U d01 = __builtin_amdgcn_perm(b1, b0, 0x05040100u);   // {b1.lo16, b0.lo16}
      const U d23 = __builtin_amdgcn_perm(b3, b2, 0x05040100u);
      const us2 sat = {63, 63};
      U x01 = d01 & 0x00FF00FFu, x23 = d23 & 0x00FF00FFu;
      us2 y01 = __builtin_elementwise_min(__builtin_bit_cast(us2, x01), sat);
      us2 y23 = __builtin_elementwise_min(__builtin_bit_cast(us2, x23), sat);
      const U e01 = ((d01 >> 2) & 0x00C000C0u) | __builtin_bit_cast(U, y01);
      const U e23 = ((d23 >> 2) & 0x00C000C0u) | __builtin_bit_cast(U, y23);
      return __builtin_amdgcn_perm(e23, e01, 0x06040200u);        // byte 0 of each 16-bit half
    } else {
      auto one = [](U x) { U len = x & ((1u << IB) - 1u); return (((x >> IB) & 3u) << 6) | (len > 63u ? 63u : len); };
      return one(b0) | (one(b1) << 8) | (one(b2) << 16) | (one(b3) << 24);
    }
  }
  // 16-lane row forms (one read per DPP row in the quad kernel): row_shr never leaves its row
  __device__ __forceinline__ U rshr1(U x, U fill) const {
    return (U)__builtin_amdgcn_update_dpp((int)fill, (int)x, DPP_ROW_SHR + 1, 0xF, 0xF, false);
  }
  __device__ __forceinline__ U rshr1_max(U x, U unav) const {
    U s = (U)__builtin_amdgcn_update_dpp(0, (int)x, DPP_ROW_SHR + 1, 0xF, 0xF, true);
    return s > unav ? s : unav;
  }
  __device__ __forceinline__ U rscan_max(U v) const {
    v = umax(v, (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 1, 0xF, 0xF, false));
    v = umax(v, (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 2, 0xF, 0xF, false));
    v = umax(v, (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 4, 0xF, 0xF, false));
    v = umax(v, (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 8, 0xF, 0xF, false));
    return v;
  }
  __device__ __forceinline__ U row_last(U v) const { return (U)__shfl((int)v, (int)((lane() & ~15u) | 15u)); }
  __device__ __forceinline__ static U udiv13(U e) { return e / 13u; }
  // a + b + c with c wave-uniform, as ONE v_add3_u32 that the optimiser cannot hoist into a per-column register
  __device__ __forceinline__ static U add3(U a, U b, uint32_t c) {
    U r;
    asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
  }
  __device__ __forceinline__ void tr_w128m(U off, U v0, U v1, U v2, U v3, M ok) const {
    if (ok) *reinterpret_cast<uint4*>(trace + off) = make_uint4(v0, v1, v2, v3);
  }
  // (x << sh) + y as ONE v_lshl_add_u32 (the optimiser otherwise splits it)
  __device__ __forceinline__ static U shl_add(U x, int sh, U y) {
    U r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(10), "v"(y));
    (void)sh;
    return r;
  }
  template <int S>
  __device__ __forceinline__ static U shl_addc(U x, U y) {
    U r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(S), "v"(y));
    return r;
  }
  // a wave-uniform constant the optimiser must keep in a scalar register (it otherwise re-derives per-row constants
  // from the previous row's with an extra VECTOR add per cell)
  __device__ __forceinline__ static uint32_t sconst(uint32_t c) {
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    asm("" : "+s"(c));
    return c;
  }
  // (m & x) | (~m & y), m wave-uniform: one v_bfi_b32
  __device__ __forceinline__ static U bfi(uint32_t m, U x, U y) {
    U r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(m), "v"(x), "v"(y));
    return r;
  }
  __device__ __forceinline__ U lds_ri16o(U off, uint32_t imm) const { return (U)(int32_t)*reinterpret_cast<const int16_t*>(lds + off + imm); }
  // absolute LDS address of a byte offset, laundered so that the optimiser keeps it in a register instead of
  // re-adding the (zero) base of the dynamic LDS block before every read
  __device__ __forceinline__ U lds_abs(U off) const {
    uint32_t p = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(lds + off);
    asm("" : "+v"(p));
    return p;
  }
  // keep a value in its register as it is (stops the optimiser from re-deriving loop-carried addresses per row)
  __device__ __forceinline__ static void keep(U& x) { asm volatile("" : "+v"(x)); }
  __device__ __forceinline__ static U lds_ri16a(U addr, uint32_t imm) {
    return (U)(int32_t)*reinterpret_cast<const __attribute__((address_space(3))) int16_t*>((uintptr_t)(addr + imm));
  }
  __device__ __forceinline__ static U pack16(U lo, U hi) { return __builtin_amdgcn_perm(hi, lo, 0x05040100u); }   // {hi.lo16, lo.lo16}
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
  // inclusive prefix sum across the 64 lanes (same DPP ladder as scan_max; invalid sources contribute 0)
  __device__ __forceinline__ U scan_add(U v) const {
    v += (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 1, 0xF, 0xF, false);
    v += (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 2, 0xF, 0xF, false);
    v += (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 4, 0xF, 0xF, false);
    v += (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR + 8, 0xF, 0xF, false);
    v += (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST15, 0xA, 0xF, false);
    v += (U)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST31, 0xC, 0xF, false);
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
  __device__ __forceinline__ U lds_r32(U off) const { return *reinterpret_cast<const uint32_t*>(lds + off); }
  __device__ __forceinline__ U lds_r32m(U off, M ok) const { return ok ? *reinterpret_cast<const uint32_t*>(lds + off) : 0u; }
  __device__ __forceinline__ static U gload_u32(const uint32_t* p, U i, M ok) { return ok ? p[i] : 0u; }
  __device__ __forceinline__ static void gstore_u32(uint32_t* p, U i, U v, M ok) { if (ok) p[i] = v; }
  __device__ __forceinline__ U tr_r16(U off, M ok) const {
    return ok ? (U)__hip_atomic_load(reinterpret_cast<const uint16_t*>(trace + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  }
  // trace slab: plain global stores / byte loads; the slab is private to the wave, re-used
  // for every read it processes and small enough (<= 32 waves/CU x 256 CUs) to live in
  // L2 / Infinity Cache, so the bytes rarely reach HBM
  __device__ __forceinline__ void tr_w32(U off, U v, M ok) const { if (ok) *reinterpret_cast<uint32_t*>(trace + off) = v; }
  // loads for the traceback are agent-scope relaxed (global_load_ubyte sc1): served by L2, never by a stale L1 line
  __device__ __forceinline__ U tr_r8(U off, M ok) const {
    return ok ? (U)__hip_atomic_load(trace + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  }
  // this wave's own stores must be visible to its own loads: same CU, same L1 -> ordering only
  __device__ __forceinline__ void tr_fence() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  // nothing is scheduled across (a batch of independent loads stays in front of the first wait for one of them)
  __device__ __forceinline__ void sched_fence() const { __builtin_amdgcn_sched_barrier(0); }
  // order this wave's LDS writes before its later LDS reads (other lanes' data)
  __device__ __forceinline__ void lds_fence() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
};

}  // namespace mia
