// wave_emu.h -- TEST INFRASTRUCTURE: a CPU "wave policy" for csrc/align_body.h.
// Every per-lane value is a 64-element array and every operation is applied to
// all lanes before the next one starts (lock-step), so cross-lane operations
// (shift, prefix max, ballot) see exactly what the GPU's DPP/LDS versions see.
// Used only by tests/emul/emu_align.cpp to unit-test the kernel logic without
// a GPU.  The product never links this.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

#include "mia_layout.h"

namespace mia {

struct EV {  // 64 x uint32
  uint32_t a[64];
  EV() { memset(a, 0, sizeof a); }
  EV(uint32_t x) { for (int i = 0; i < 64; i++) a[i] = x; }
};
struct EM {  // 64 x bool
  bool a[64];
  EM() { memset(a, 0, sizeof a); }
};

#define EV_BIN(op)                                                                                      \
  inline EV operator op(const EV& x, const EV& y) { EV r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] op y.a[i]; return r; } \
  inline EV operator op(const EV& x, uint32_t y) { EV r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] op y; return r; }       \
  inline EV operator op(uint32_t x, const EV& y) { EV r; for (int i = 0; i < 64; i++) r.a[i] = x op y.a[i]; return r; }
EV_BIN(+) EV_BIN(-) EV_BIN(*) EV_BIN(&) EV_BIN(|) EV_BIN(^)
#undef EV_BIN
inline EV operator<<(const EV& x, int s) { EV r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] << s; return r; }
inline EV operator>>(const EV& x, int s) { EV r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] >> s; return r; }
inline EV operator<<(const EV& x, const EV& s) { EV r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] << s.a[i]; return r; }
inline EV operator>>(const EV& x, const EV& s) { EV r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] >> s.a[i]; return r; }
inline EV operator~(const EV& x) { EV r; for (int i = 0; i < 64; i++) r.a[i] = ~x.a[i]; return r; }
#define EV_CMP(op)                                                                                      \
  inline EM operator op(const EV& x, const EV& y) { EM r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] op y.a[i]; return r; } \
  inline EM operator op(const EV& x, uint32_t y) { EM r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] op y; return r; }
EV_CMP(==) EV_CMP(!=) EV_CMP(<) EV_CMP(<=) EV_CMP(>) EV_CMP(>=)
#undef EV_CMP
inline EM operator&(const EM& x, const EM& y) { EM r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] && y.a[i]; return r; }
inline EM operator|(const EM& x, const EM& y) { EM r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] || y.a[i]; return r; }
inline EM operator!(const EM& x) { EM r; for (int i = 0; i < 64; i++) r.a[i] = !x.a[i]; return r; }

struct EmuWave {
  typedef EV U;
  typedef EM M;
  std::vector<unsigned char> lds, trace;
  EmuWave(size_t lds_bytes, size_t trace_bytes) : lds(lds_bytes + 64, 0), trace(trace_bytes + 64, 0) {}

  U lane() const { EV r; for (int i = 0; i < 64; i++) r.a[i] = (uint32_t)i; return r; }
  U shr1(const U& x, const U& fill) const { EV r; r.a[0] = fill.a[0]; for (int i = 1; i < 64; i++) r.a[i] = x.a[i - 1]; return r; }
  U shr1_max(const U& x, const U& unav) const { EV r; r.a[0] = unav.a[0]; for (int i = 1; i < 64; i++) r.a[i] = x.a[i - 1] > unav.a[i] ? x.a[i - 1] : unav.a[i]; return r; }
  static U sext_lo(const U& x) { EV r; for (int i = 0; i < 64; i++) r.a[i] = (uint32_t)(int32_t)(int16_t)(x.a[i] & 0xFFFFu); return r; }
  static U sext_hi(const U& x) { EV r; for (int i = 0; i < 64; i++) r.a[i] = (uint32_t)(int32_t)(int16_t)(x.a[i] >> 16); return r; }
  // 4 packed score words -> 4 trace bytes [type:2][len:6, saturating at 63]
  template <int IB>
  static U trace_pack4(const U& b0, const U& b1, const U& b2, const U& b3) {
    EV r;
    const U* b[4] = {&b0, &b1, &b2, &b3};
    for (int i = 0; i < 64; i++) {
      uint32_t w = 0;
      for (int k = 0; k < 4; k++) {
        uint32_t x = b[k]->a[i], len = x & ((1u << IB) - 1u), ty = (x >> IB) & 3u;
        w |= ((ty << 6) | (len > 63u ? 63u : len)) << (8 * k);
      }
      r.a[i] = w;
    }
    return r;
  }
  // 16-lane row forms (one read per DPP row in the quad kernel)
  U rshr1(const U& x, const U& fill) const { EV r; for (int i = 0; i < 64; i++) r.a[i] = (i & 15) ? x.a[i - 1] : fill.a[i]; return r; }
  U rshr1_max(const U& x, const U& unav) const { EV r; for (int i = 0; i < 64; i++) { uint32_t s = (i & 15) ? x.a[i - 1] : 0u; r.a[i] = s > unav.a[i] ? s : unav.a[i]; } return r; }
  U rscan_max(const U& v) const { EV r; uint32_t m = 0; for (int i = 0; i < 64; i++) { if ((i & 15) == 0) m = 0; m = v.a[i] > m ? v.a[i] : m; r.a[i] = m; } return r; }
  U row_last(const U& v) const { EV r; for (int i = 0; i < 64; i++) r.a[i] = v.a[(i & ~15) | 15]; return r; }
  static U udiv13(const U& e) { EV r; for (int i = 0; i < 64; i++) r.a[i] = e.a[i] / 13u; return r; }
  static U add3(const U& x, const U& y, uint32_t c) { EV r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] + y.a[i] + c; return r; }
  void tr_w128m(const U& off, const U& v0, const U& v1, const U& v2, const U& v3, const M& ok) {
    for (int i = 0; i < 64; i++) if (ok.a[i]) { trace.at(off.a[i] + 15); memcpy(&trace[off.a[i]], &v0.a[i], 4); memcpy(&trace[off.a[i] + 4], &v1.a[i], 4);
                                               memcpy(&trace[off.a[i] + 8], &v2.a[i], 4); memcpy(&trace[off.a[i] + 12], &v3.a[i], 4); }
  }
  static uint32_t sconst(uint32_t c) { return c; }
  static void keep(U&) {}
  U lds_abs(const U& off) const { return off; }
  U lds_ri16a(const U& addr, uint32_t imm) const { return lds_ri16o(addr, imm); }
  template <int S>
  static U shl_addc(const U& x, const U& y) { EV r; for (int i = 0; i < 64; i++) r.a[i] = (x.a[i] << S) + y.a[i]; return r; }
  static U shl_add(const U& x, int sh, const U& y) { EV r; for (int i = 0; i < 64; i++) r.a[i] = (x.a[i] << sh) + y.a[i]; return r; }
  static U pack16(const U& lo, const U& hi) { EV r; for (int i = 0; i < 64; i++) r.a[i] = (lo.a[i] & 0xFFFFu) | (hi.a[i] << 16); return r; }
  static U bfi(uint32_t m, const U& x, const U& y) { EV r; for (int i = 0; i < 64; i++) r.a[i] = (x.a[i] & m) | (y.a[i] & ~m); return r; }
  U lds_ri16o(const U& off, uint32_t imm) const { EV r; for (int i = 0; i < 64; i++) { int16_t x; lds.at(off.a[i] + imm + 1); memcpy(&x, &lds[off.a[i] + imm], 2); r.a[i] = (uint32_t)(int32_t)x; } return r; }
  static U umax(const U& x, const U& y) { EV r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] > y.a[i] ? x.a[i] : y.a[i]; return r; }
  static U umin(const U& x, const U& y) { EV r; for (int i = 0; i < 64; i++) r.a[i] = x.a[i] < y.a[i] ? x.a[i] : y.a[i]; return r; }
  static U umax3(const U& x, const U& y, const U& z) { return umax(umax(x, y), z); }
  static U sel(const M& c, const U& x, const U& y) { EV r; for (int i = 0; i < 64; i++) r.a[i] = c.a[i] ? x.a[i] : y.a[i]; return r; }
  U scan_max(const U& v) const { EV r; uint32_t m = 0; for (int i = 0; i < 64; i++) { m = v.a[i] > m ? v.a[i] : m; r.a[i] = m; } return r; }
  U scan_add(const U& v) const { EV r; uint32_t m = 0; for (int i = 0; i < 64; i++) { m += v.a[i]; r.a[i] = m; } return r; }
  uint32_t reduce_max(const U& v) const { return scan_max(v).a[63]; }
  uint32_t reduce_min(const U& v) const { return ~reduce_max(~v); }
  uint64_t ballot(const M& m) const { uint64_t b = 0; for (int i = 0; i < 64; i++) if (m.a[i]) b |= 1ull << i; return b; }
  uint32_t lane_val(const U& v, int l) const { return v.a[l]; }
  bool lane_bit(const M& m, int l) const { return m.a[l]; }
  static U udiv5(const U& e) { EV r; for (int i = 0; i < 64; i++) r.a[i] = e.a[i] / 5u; return r; }
  static U depth(const U& row, uint32_t len) { EV r; for (int i = 0; i < 64; i++) r.a[i] = (uint32_t)sm_depth((int)row.a[i], (int)len); return r; }
  static U gload_u8(const uint8_t* p, const U& idx, const M& ok) { EV r; for (int i = 0; i < 64; i++) r.a[i] = ok.a[i] ? p[idx.a[i]] : 0u; return r; }
  static U gload_i32(const int32_t* p, const U& idx, const M& ok) { EV r; for (int i = 0; i < 64; i++) r.a[i] = ok.a[i] ? (uint32_t)p[idx.a[i]] : 0u; return r; }
  static void gstore_i16(int16_t* p, const U& idx, const U& v, const M& ok) { for (int i = 0; i < 64; i++) if (ok.a[i]) p[idx.a[i]] = (int16_t)(uint16_t)v.a[i]; }
  void lds_w16(const U& off, const U& v, const M& ok) { for (int i = 0; i < 64; i++) if (ok.a[i]) { uint16_t x = (uint16_t)v.a[i]; memcpy(&lds.at(off.a[i]), &x, 2); } }
  void lds_w32(const U& off, const U& v, const M& ok) { for (int i = 0; i < 64; i++) if (ok.a[i]) { lds.at(off.a[i] + 3); memcpy(&lds[off.a[i]], &v.a[i], 4); } }
  U lds_ri16(const U& off) const { EV r; for (int i = 0; i < 64; i++) { int16_t x; lds.at(off.a[i] + 1); memcpy(&x, &lds[off.a[i]], 2); r.a[i] = (uint32_t)(int32_t)x; } return r; }
  U lds_r32(const U& off) const { EV r; for (int i = 0; i < 64; i++) { lds.at(off.a[i] + 3); memcpy(&r.a[i], &lds[off.a[i]], 4); } return r; }
  U lds_r32m(const U& off, const M& ok) const { EV r; for (int i = 0; i < 64; i++) if (ok.a[i]) { lds.at(off.a[i] + 3); memcpy(&r.a[i], &lds[off.a[i]], 4); } return r; }
  static U gload_u32(const uint32_t* p, const U& idx, const M& ok) { EV r; for (int i = 0; i < 64; i++) r.a[i] = ok.a[i] ? p[idx.a[i]] : 0u; return r; }
  static void gstore_u32(uint32_t* p, const U& idx, const U& v, const M& ok) { for (int i = 0; i < 64; i++) if (ok.a[i]) p[idx.a[i]] = v.a[i]; }
  U tr_r16(const U& off, const M& ok) const { EV r; for (int i = 0; i < 64; i++) if (ok.a[i]) { uint16_t x; trace.at(off.a[i] + 1); memcpy(&x, &trace[off.a[i]], 2); r.a[i] = x; } return r; }
  void lds_fence() const {}
  void sched_fence() const {}
  void tr_w32(const U& off, const U& v, const M& ok) { for (int i = 0; i < 64; i++) if (ok.a[i]) { trace.at(off.a[i] + 3); memcpy(&trace[off.a[i]], &v.a[i], 4); } }
  U tr_r8(const U& off, const M& ok) const { EV r; for (int i = 0; i < 64; i++) r.a[i] = ok.a[i] ? trace.at(off.a[i]) : 0u; return r; }
  void tr_fence() const {}
};

}  // namespace mia
