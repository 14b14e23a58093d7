// Shared GEMM epilogue (fp32-MFMA and split-fp16 kernels have the same 32x32 accumulator layout:
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).
#pragma once
#include "common.h"

// wm0 / wn0: global row / column of this WAVE's (32*WM) x (32*WN) output tile.
// FULL = the whole wave tile is inside the matrix (wave-uniform): no per-element predicates, and every address is one
// 64-bit base per 32x32 tile plus a 32-bit offset (the predicated form spent ~2300 VALU instructions per wave).
template <int WM, int WN, bool FULL>
__device__ __forceinline__ void gemm_epilogue_impl(f32x16 (&acc)[WM][WN], const GemmParams& p, int z, int wm0, int wn0,
                                                   int M_store, float acc_scale) {
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, l31 = lane & 31;
  float* __restrict__ Cg = p.C + (long)z * p.sC;
  const float* __restrict__ addg = p.add ? p.add + (long)z * p.sAdd : nullptr;
  const float* __restrict__ maskg = p.keep_mask ? p.keep_mask + (long)z * p.M_store * p.N : nullptr;
  const float bscale = p.batch_scale ? p.batch_scale[z] : 1.0f;
  const bool do_drop = p.drop_thr != 0u;
  // stacked calls (sdy_conv_args.rows_per_call): which call, which trajectory (0 = launchers without a dropout stream)
  const int rpc = p.rows_per_call > 0 ? p.rows_per_call : 0x7FFFFFFF;
  const int zq = z / rpc;
  const uint32_t call_z = p.call + (uint32_t)zq;
  const uint32_t c1_base = (uint32_t)(((uint64_t)((z - zq * rpc) + p.batch_offset) * (uint64_t)(p.M_store >> 2)) & 0xFFFFFFFFu);

  // Two-phase per 32x32 tile: first every bias / addend load of the tile (16 independent loads in flight), then the
  // arithmetic and the stores.  (Interleaving them lets the possible add == C aliasing serialise each element behind a
  // full memory round trip: 128 dependent L2 latencies per lane.)
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int row0 = wm0 + i * 32 + 4 * h;  // row of reg r: row0 + (r & 3) + 8 * (r >> 2)
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gm = row0 + (r & 3) + 8 * (r >> 2);
      bias_r[r] = (p.bias && (FULL || gm < M_store)) ? p.bias[gm] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int gn = wn0 + j * 32 + l31;
      const bool col_ok = FULL || gn < p.N;
      float add_r[16], keep_r[16];
      // one 64-bit base per tile, 32-bit element offsets (row strides of the matrices involved are < 2^26 floats)
      const float* add_t = addg ? addg + (long)row0 * p.ldadd + gn : nullptr;
      const float* mask_t = maskg ? maskg + (long)row0 * p.N + gn : nullptr;
      float* c_t = Cg + (long)row0 * p.ldc + gn;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        const bool ok = FULL || (col_ok && row0 + dr < M_store);
        add_r[r] = (p.add_mode != 0 && ok) ? add_t[dr * p.ldadd] : 0.0f;
        keep_r[r] = (maskg && ok) ? mask_t[dr * p.N] : 1.0f;
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int row_base = row0 + 8 * rg;
        uint32_t words[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (do_drop && !maskg && (FULL || (col_ok && row_base < M_store))) {
          const philox4 w = philox4x32((uint32_t)gn & ~32u, c1_base + (uint32_t)(row_base >> 2), p.stream_id, call_z,
                                          p.seed_lo, p.seed_hi);
          words[0] = w.x; words[1] = w.y; words[2] = w.z; words[3] = w.w;
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int r = rg * 4 + r4;
          const int gm = row_base + r4;
          float v = acc[i][j][r] * acc_scale + bias_r[r];
          if (p.add_mode == 1) v += add_r[r];
          if (p.act == 1) v = gelu_erf(v);
          if (do_drop) {
            const bool keep = maskg ? (keep_r[r] != 0.0f) : sdy_keep16(words[r4], (gn >> 5) & 1, p.drop_thr);
            v = keep ? v * p.drop_scale : 0.0f;
          }
          v *= bscale;
          if (p.add_mode == 2) v += add_r[r];
          if (FULL || (col_ok && gm < M_store)) c_t[(r4 + 8 * rg) * p.ldc] = v;
        }
      }
    }
  }
}

template <int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_at(f32x16 (&acc)[WM][WN], const GemmParams& p, int z, int wm0, int wn0,
                                                 int M_store, float acc_scale) {
  const bool full = (wm0 + 32 * WM <= M_store) && (wn0 + 32 * WN <= p.N);   // wave-uniform
  if (full)
    gemm_epilogue_impl<WM, WN, true>(acc, p, z, wm0, wn0, M_store, acc_scale);
  else
    gemm_epilogue_impl<WM, WN, false>(acc, p, z, wm0, wn0, M_store, acc_scale);
}

// workgroup of 4 waves arranged 2 x 2 (gemm.hip, gemm_h3.hip 128 x 128 kernel)
template <int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[WM][WN], const GemmParams& p, int z, int m0, int n0,
                                              int M_store, float acc_scale) {
  const int wave = threadIdx.x >> 6;
  gemm_epilogue_at<WM, WN>(acc, p, z, m0 + (wave >> 1) * (32 * WM), n0 + (wave & 1) * (32 * WN), M_store, acc_scale);
}
