// mia_trim_kernels.h -- adapter trimming, trim_frag of the reference
// (/root/reference/src/mia.c:1318-1368, set up in src/mia_main.c:694-717): a semi-global alignment of the adapter
// (rows, sg5 = 1: unaligned adapter prefix pays; sg3 = 0) against the read (columns) on the flat matrix, best
// score over the LAST COLUMN, walk back to the alignment start.  One read per wavefront with the window
// aligner in LASTCOL mode; the rare path whose trace would need a gap length of 63 or more is re-run exactly,
// one read per thread, with an int32 trace in global scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "align_body.h"
#include "mia_layout.h"
#include "wave_dev.h"

namespace mia {

constexpr int TRIM_SCORE_CUT = 1000;   // src/params.h:32
constexpr int MAX_ADAPTER = 127;       // src/mia_main.c:559

struct TrimReads {
  int64_t n;
  const uint8_t* codes;     // one code (0..4) per read base
  const int64_t* off;       // [n+1]
  uint8_t* trimmed;
  int32_t* trim_point;
  uint32_t* status;
};

__device__ __forceinline__ void trim_decide(const TrimReads& tr, int64_t i, int score, int aer, int abr, int abc) {
  const bool t = (score >= TRIM_SCORE_CUT) || (score >= (aer - abr + 1) * FLAT_MATCH);   // src/mia.c:1359-1366
  tr.trimmed[i] = t ? 1 : 0;
  tr.trim_point[i] = t ? abc - 1 : 0;
}

__global__ __launch_bounds__(64) void k_trim(TrimReads tr, const uint8_t* adapter_packed, int32_t len2, const int32_t* flat_pssm,
                                              PackParams pk, unsigned char* trace_slabs, int64_t slab_bytes, int16_t* cols_scratch) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[MAX_READ * 10];
  DevWave wave(lds_raw, trace_slabs + (int64_t)blockIdx.x * slab_bytes);
  for (int64_t i = blockIdx.x; i < tr.n; i += gridDim.x) {
    const int len1 = (int)(tr.off[i + 1] - tr.off[i]);
    AlignArgs a;
    a.ref_codes = tr.codes + tr.off[i];
    a.ref_start = 0;
    a.len1 = len1;
    a.read_packed = adapter_packed;
    a.len2 = len2;
    a.pssm = flat_pssm;
    a.sg5 = 1;
    a.pk = pk;
    a.lds_sub = 0;
    a.trace_stride = (uint32_t)((len1 + 3) & ~3);
    a.cols_out = cols_scratch + (int64_t)blockIdx.x * MAX_READ;
    a.dbg = 0;
    AlignResult r = WindowAligner<DevWave, 4, true>::run(wave, a);
    if (wave.lane() == 0) {
      tr.status[i] = r.status;
      trim_decide(tr, i, r.score, r.aer, r.abr, r.abc);
    }
    wave.lds_fence();
  }
}

// exact re-run of the reads in `list`: scalar dyn_prog with the reference's own cascade (src/mia.c:740-981)
__global__ void k_trim_wide(TrimReads tr, const uint8_t* adapter_codes, int32_t len2, const int32_t* flat_pssm, const int32_t* list,
                            int32_t count, int32_t* scratch, int64_t words_per_read) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int64_t i = list[t];
  const uint8_t* c1 = tr.codes + tr.off[i];
  const int n1 = (int)(tr.off[i + 1] - tr.off[i]);
  int32_t* base = scratch + (int64_t)t * words_per_read;
  int32_t* T = base;                                // [len2][n1]
  int32_t* S0 = base + (int64_t)len2 * n1;          // three rotating score rows
  int32_t* colkey = S0 + 3 * (int64_t)n1;
  int32_t* colrow = colkey + n1;
  int32_t *prev2 = S0, *prev = S0 + n1, *cur = S0 + 2 * (int64_t)n1;
  const int last = n1 - 1;
  int best = INT32_MIN, aer = 0;
  for (int c = 0; c < n1; c++) {
    cur[c] = flat_pssm[(0 * 5 + c1[c]) * 5 + adapter_codes[0]];
    T[c] = 0;
    colkey[c] = cur[c];
    colrow[c] = 0;
  }
  if (cur[last] > best) { best = cur[last]; aer = 0; }
  for (int r = 1; r < len2; r++) {
    int32_t* tmp = prev2; prev2 = prev; prev = cur; cur = tmp;
    const int d = sm_depth(r, len2), c2 = adapter_codes[r];
    const int fresh = -(GOP + GEP * (r + 1));
    int32_t* trw = T + (int64_t)r * n1;
    cur[0] = flat_pssm[(d * 5 + c1[0]) * 5 + c2] + fresh;
    trw[0] = 0;
    int rowkey = prev[0], rowcol = 0;
    for (int c = 1; c < n1; c++) {
      const int sub = flat_pssm[(d * 5 + c1[c]) * 5 + c2];
      int gapc = -(1 << 30), gapr = -(1 << 30);
      if (c >= 2) {
        int k = prev[c - 2] + GEP * (c - 2);
        if (k > rowkey) { rowkey = k; rowcol = c - 2; }
        gapc = rowkey - GOP - GEP * (c - 1);
      }
      if (r >= 2) {
        int k = prev2[c - 1] + GEP * (r - 2);
        if (k > colkey[c - 1]) { colkey[c - 1] = k; colrow[c - 1] = r - 2; }
        gapr = colkey[c - 1] - GOP - GEP * (r - 1);
      }
      const int diag = prev[c - 1];
      if (fresh > diag && fresh > gapc && fresh > gapr) { cur[c] = fresh; trw[c] = c; }
      else if (diag >= gapc && diag >= gapr) { cur[c] = sub + diag; trw[c] = 0; }
      else if (gapc >= gapr) { cur[c] = sub + gapc; trw[c] = rowcol; }
      else { cur[c] = sub + gapr; trw[c] = -colrow[c - 1]; }
    }
    if (cur[last] > best) { best = cur[last]; aer = r; }
  }
  int r = aer, c = last;
  for (;;) {                                         // find_align_begin, src/mia.c:612-637
    const int t2 = T[(int64_t)r * n1 + c];
    if (t2 == c || t2 == -r) break;
    if (t2 == 0) { r--; c--; }
    else if (t2 < 0) { r = -t2; c--; }
    else { c = t2; r--; }
  }
  tr.status[i] = ST_OK;
  trim_decide(tr, i, best, aer, r, c);
}

}  // namespace mia
