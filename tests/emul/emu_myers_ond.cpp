// emu_myers_ond.cpp -- TEST INFRASTRUCTURE: csrc/myers_ond_body.h (the functions k_myers_ond runs per lane) driven row by
// row on the CPU in the kernel's order -- every diagonal of a row from the finished row before, the smallest arrived diagonal
// ends the search -- so the packed snakes, the cell recurrence and the walk back can be checked against the reference's
// myers_diff answers (tests/golden/myers_vectors.txt) without a GPU.  Built by tests/test_emul_myers_ond.py.
#include <limits.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "myers_ond_body.h"

using namespace mia;

static uint32_t bits_of(char x) {   // src/myers_align.h:40-67 (restated in csrc/mia_myers_kernels.h: iupac_bits)
  switch (x & ~32) {
    case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8;
    case 'S': return 6; case 'W': return 9; case 'R': return 5; case 'Y': return 10; case 'K': return 12; case 'M': return 3;
    case 'B': return 14; case 'D': return 13; case 'H': return 11; case 'V': return 7; case 'N': return 15;
    default: return 0;
  }
}

static std::vector<uint32_t> pack(const char* s, int len) {
  std::vector<uint32_t> w((size_t)(len + 7) / 8 + OND_PAD_WORDS, 0u);
  for (int i = 0; i < len; i++) w[(size_t)i >> 3] |= bits_of(s[i]) << (4 * (i & 7));
  return w;
}

// returns the distance, 0xFFFFFFFF (not below maxd) or 0xFFFFFFFE (not within cap < maxd); rows into bt_a / bt_b
extern "C" uint32_t emu_myers_ond(const char* a, int mode, const char* b, int maxd, int cap_in, char* bt_a, char* bt_b) {
  const int la = (int)strlen(a), lb = (int)strlen(b);
  if (maxd > la + lb) maxd = la + lb;
  const int cap = cap_in < maxd ? cap_in : maxd;
  const std::vector<uint32_t> A = pack(a, la), B = pack(b, lb);
  std::vector<int32_t> table((size_t)(cap > 0 ? cap : 0) * (size_t)(cap > 0 ? cap : 0) + 1, OND_NONE);
  std::vector<int32_t> rows[2];
  rows[0].assign((size_t)2 * (cap > 0 ? cap : 0) + 3, 0); rows[1] = rows[0];
  int dist = -1, found_k = INT_MAX;
  for (int d = 0; d < cap; d++) {
    std::vector<int32_t>& cur = rows[d & 1];
    const std::vector<int32_t>& prow = rows[(d & 1) ^ 1];
    const int klo = -d > -la ? -d : -la, khi = d < lb ? d : lb;
    auto prev = [&](int kk) -> int32_t { return (kk < -(d - 1) || kk > d - 1) ? OND_NONE : prow[(size_t)(kk + cap + 1)]; };
    for (int k = d; k >= -d; k--) {        // (descending on purpose: the order within a row must not matter)
      int32_t x = OND_NONE;
      if (k >= klo && k <= khi) {
        x = ond_cell(d, k, prev);
        if (x != OND_NONE) {
          const int alone = ond_snake(A, B, x - k, x, la, lb);
          // the kernel's way: eight characters by the diagonal's own lane, the rest in rounds of 64 lanes x 8 characters
          int shared = 0;
          if (x >= 0 && x - k >= 0) {
            shared = ond_shared_chunk(A, B, x - k, x, la, lb);
            if (shared == 8) {
              const int xs = x + 8;
              bool stop = false;
              for (int off = 0; !stop; off += OND_SHARED_SPAN)
                for (int lane = 0; lane < 64 && !stop; lane++) {
                  const int xx = xs + off + 8 * lane, c = ond_shared_chunk(A, B, xx - k, xx, la, lb);
                  if (c < 8) { shared = 8 + off + 8 * lane + c; stop = true; }
                }
            }
          }
          if (shared != alone) return 0xFFFFFFFCu;
          x += shared;
          if (ond_arrived(mode, x, k, la, lb) && k < found_k) found_k = k;
        }
      }
      cur[(size_t)(k + cap + 1)] = x;
      table[ond_at(d, k)] = x;
    }
    if (found_k != INT_MAX) { dist = d; break; }
  }
  if (dist < 0) return cap < maxd ? 0xFFFFFFFEu : 0xFFFFFFFFu;
  std::string ra, rb;
  if (!ond_walk_back(table.data(), a, la, b, lb, dist, found_k, &ra, &rb)) return 0xFFFFFFFDu;
  if (bt_a) memcpy(bt_a, ra.c_str(), ra.size() + 1);
  if (bt_b) memcpy(bt_b, rb.c_str(), rb.size() + 1);
  return (uint32_t)dist;
}
