// Persistent, warp-specialised variant of the split-precision conv GEMM (gemm_h3.hip, MODE_COLS), gfx950.
//
// Why: with three f16 MFMA passes the arithmetic of a 128 x 128 x 32 tile step takes ~770 cycles, a loaded HBM round
// trip ~4000.  In gemm_h3.hip every wave loads, converts AND multiplies, so each k-tile exposes that latency (PMC: waves
// parked 40 % of their life) and prologue / epilogue of a workgroup overlap with nothing.  Here one workgroup per CU
// lives for the whole launch and walks a list of output tiles:
//   * waves 4-7 (producers) keep THREE k-tile steps of global loads in flight in registers, convert / split / transpose
//     the oldest one into one of two LDS slots and immediately issue the next loads -- across tile boundaries, so the
//     stream never drains at a tile's prologue or epilogue;
//   * waves 0-3 (consumers) only read fragments and issue MFMAs (2 x 2 tiles of 32 x 32 each), then run the fused epilogue.
// One s_barrier per step hands slot (s & 1) from the producers to the consumers; the loads are branch-free (clamped
// address + select) so hipcc emits counted vmcnt waits that retire only the oldest stage.
//
// STATUS (round 1): correct (passes the conv parity tests with SDY_H3_WS=1) but NOT the default: measured 0.68 ms vs
// 0.43 ms for the 256 -> 256 conv at B = 8.  A step costs ~4 k cycles instead of the ~0.8 k the MFMAs need because (a) the
// single producer wave per SIMD is VALU-bound on the split / address arithmetic (~350 instructions per step) and (b) with
// only two LDS slots the producers stall for the whole consumer epilogue (~10 k cycles per tile).  Next: more, smaller
// LDS slots with flag hand-off so producers run a full tile ahead, and a second consumer group for the epilogue.
#include "common.h"
#include "gemm_epilogue.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SBK = 32;       // k per step
constexpr int SLD = 40;       // LDS row stride in halfs
constexpr int SBM = 128, SBN = 128;
constexpr int SLOT_HALFS = (2 * SBM + 2 * SBN) * SLD;   // As_hi | As_lo | Bs_hi | Bs_lo

struct WsPacked {
  const _Float16* hi;
  const _Float16* lo;
  int Kpad;
};

struct Stage {
  f32x4 a_hi[2], a_lo[2];   // 128 rows x 4 chunks x 2 planes over 256 producer threads
  f32x4 b[4];               // 4 k x 4 pixels
  float ca[4], cd[4];
};

template <int TAG>
__global__ __launch_bounds__(512) void gemm_h3_ws_kernel(const GemmParams p, const WsPacked pk, float sx, float out_scale,
                                                          int MT, int NT, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) _Float16 smem_ws[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int nk = (p.K + SBK - 1) / SBK;
  // tiles of this workgroup: t = blockIdx.x + i * gridDim.x
  const int my_tiles = (ntiles > (int)blockIdx.x) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int S = my_tiles * nk;

  auto decode = [&](int ti, int& z, int& m0, int& n0) {
    const int t = blockIdx.x + ti * gridDim.x;
    const int mt = t % MT;
    const int r = t / MT;
    const int nt = r % NT;
    z = r / NT;
    m0 = mt * SBM;
    n0 = nt * SBN;
  };

  if (producer) {
    const int pt = tid - 256;
    const int kq = pt & 7, pq = pt >> 3;          // 8 k-quads x 32 pixel quads
    const bool has_aff = p.pa != nullptr;
    Stage st0, st1, st2;

    // incremental item cursor for LOADS
    int l_ti = 0, l_kt = 0, l_z = 0, l_m0 = 0, l_n0 = 0;
    if (my_tiles > 0) decode(0, l_z, l_m0, l_n0);

    auto load_item = [&](Stage& T) {
      const int k0 = l_kt * SBK;
      const _Float16* Ph = pk.hi;
      const _Float16* Pl = pk.lo;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int chunk = pt + i * 256;
        const int row = chunk >> 2, c = chunk & 3;
        const long off = (long)(l_m0 + row) * pk.Kpad + k0 + c * 8;
        T.a_hi[i] = *reinterpret_cast<const f32x4*>(Ph + off);
        T.a_lo[i] = *reinterpret_cast<const f32x4*>(Pl + off);
      }
      const float* Fg = p.B + (long)l_z * p.sB;
      const float* pa_s = has_aff ? p.pa + (long)l_z * p.p_bstride : Fg;
      const float* pd_s = has_aff ? p.pd + (long)l_z * p.p_bstride : Fg;
      const int gn = l_n0 + pq * 4;
      const bool n_ok = gn < p.N;
      const int gn_c = n_ok ? gn : 0;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int gk = k0 + kq * 4 + kk;
        const bool ok = n_ok && gk < p.K;
        const int gk_c = gk < p.K ? gk : p.K - 1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(Fg + (long)gk_c * p.ldb + gn_c);
        const float a = pa_s[gk_c], d = pd_s[gk_c];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        T.b[kk] = ok ? v : zero;
        T.ca[kk] = ok ? (has_aff ? a : 1.0f) : 0.0f;
        T.cd[kk] = (ok && has_aff) ? d : 0.0f;
      }
      // advance the load cursor
      if (++l_kt == nk) {
        l_kt = 0;
        ++l_ti;
        if (l_ti < my_tiles) decode(l_ti, l_z, l_m0, l_n0);
      }
    };

    auto store_item = [&](const Stage& T, int slot) {
      _Float16* As_hi = smem_ws + slot * SLOT_HALFS;
      _Float16* As_lo = As_hi + SBM * SLD;
      _Float16* Bs_hi = As_lo + SBM * SLD;
      _Float16* Bs_lo = Bs_hi + SBN * SLD;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int chunk = pt + i * 256;
        const int row = chunk >> 2, c = chunk & 3;
        *reinterpret_cast<f32x4*>(As_hi + row * SLD + c * 8) = T.a_hi[i];
        *reinterpret_cast<f32x4*>(As_lo + row * SLD + c * 8) = T.a_lo[i];
      }
      _Float16 hi[4][4], lo[4][4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float a = T.ca[kk] * sx, d = T.cd[kk] * sx;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const float v = fmaf(T.b[kk][pp], a, d);
          const _Float16 hv = (_Float16)v;
          hi[kk][pp] = hv;
          lo[kk][pp] = (_Float16)(v - (float)hv);
        }
      }
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int prow = pq * 4 + pp;
        f16x4 vh = {hi[0][pp], hi[1][pp], hi[2][pp], hi[3][pp]};
        f16x4 vl = {lo[0][pp], lo[1][pp], lo[2][pp], lo[3][pp]};
        *reinterpret_cast<f16x4*>(Bs_hi + prow * SLD + kq * 4) = vh;
        *reinterpret_cast<f16x4*>(Bs_lo + prow * SLD + kq * 4) = vl;
      }
    };

    int issued = 0;
    if (issued < S) { load_item(st0); ++issued; }
    if (issued < S) { load_item(st1); ++issued; }
    if (issued < S) { load_item(st2); ++issued; }
    for (int s = 0; s < S; s += 3) {
      store_item(st0, s & 1);
      if (issued < S) { load_item(st0); ++issued; }
      __syncthreads();
      if (s + 1 < S) {
        store_item(st1, (s + 1) & 1);
        if (issued < S) { load_item(st1); ++issued; }
        __syncthreads();
      }
      if (s + 2 < S) {
        store_item(st2, (s + 2) & 1);
        if (issued < S) { load_item(st2); ++issued; }
        __syncthreads();
      }
    }
    __syncthreads();   // pairs with the consumers' last step
  } else {
    const int wr = wave >> 1, wc = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    int c_ti = 0, c_kt = 0;
    for (int s = 0; s <= S; ++s) {
      if (s >= 1) {
        const int slot = (s - 1) & 1;
        const _Float16* As_hi = smem_ws + slot * SLOT_HALFS;
        const _Float16* As_lo = As_hi + SBM * SLD;
        const _Float16* Bs_hi = As_lo + SBM * SLD;
        const _Float16* Bs_lo = Bs_hi + SBN * SLD;
#pragma unroll
        for (int ks = 0; ks < SBK / 16; ++ks) {
          f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int off = (wr * 64 + i * 32 + l31) * SLD + ks * 16 + h * 8;
            ah[i] = *reinterpret_cast<const f16x8*>(As_hi + off);
            al[i] = *reinterpret_cast<const f16x8*>(As_lo + off);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int off = (wc * 64 + j * 32 + l31) * SLD + ks * 16 + h * 8;
            bh[j] = *reinterpret_cast<const f16x8*>(Bs_hi + off);
            bl[j] = *reinterpret_cast<const f16x8*>(Bs_lo + off);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
        }
        if (++c_kt == nk) {   // last k-step of this tile: fused epilogue, then start the next tile
          int z, m0, n0;
          decode(c_ti, z, m0, n0);
          gemm_epilogue_at<2, 2>(acc, p, z, m0 + wr * 64, n0 + wc * 64, p.M_store, out_scale);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
          c_kt = 0;
          ++c_ti;
        }
      }
      __syncthreads();
    }
  }
}

template <int TAG>
int launch_ws(const GemmParams& p, const WsPacked& pk, float sx, float out_scale, hipStream_t stream) {
  constexpr size_t smem = (size_t)2 * SLOT_HALFS * sizeof(_Float16);
  static bool attr_done = false;
  static int ncu = 256;
  if (!attr_done) {
    SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3_ws_kernel<TAG>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
    attr_done = true;
  }
  const int MT = (p.M_store + SBM - 1) / SBM, NT = (p.N + SBN - 1) / SBN;
  const long ntiles_l = (long)MT * NT * p.nbatch;
  if (ntiles_l > 0x7FFFFFFF) return SDY_ERR_UNSUPPORTED;
  const int ntiles = (int)ntiles_l;
  const int grid = ntiles < ncu ? ntiles : ncu;
  hipLaunchKernelGGL((gemm_h3_ws_kernel<TAG>), dim3(grid), dim3(512), smem, stream, p, pk, sx, out_scale, MT, NT, ntiles);
  return sdy_launch_status();
}

}  // namespace

// conv mode only (shared packed weight, no triangular structure); same argument meaning as sdy_gemm_h3_launch
int sdy_gemm_h3_ws_launch(const GemmParams& p, const void* packed, int rows_pad, int Kpad, long plane_halfs, float w_scale,
                          hipStream_t stream) {
  if (!packed || !p.B || !p.C) return SDY_ERR_ARG;
  if (p.M_store <= 0 || p.N <= 0 || p.K <= 0 || p.nbatch <= 0 || p.tri_mode != SDY_TRI_NONE) return SDY_ERR_ARG;
  if ((p.N & 3) || (p.ldb & 3) || (p.sB & 3)) return SDY_ERR_ALIGN;
  if (Kpad % 64 || Kpad < p.K || rows_pad % 128 || rows_pad < ((p.M_store + 127) / 128) * 128) return SDY_ERR_ARG;
  if (p.drop_thr != 0u && !p.keep_mask && (p.M_store & 3)) return SDY_ERR_ALIGN;
  WsPacked pk;
  pk.hi = reinterpret_cast<const _Float16*>(packed);
  pk.lo = pk.hi + plane_halfs;
  pk.Kpad = Kpad;
  const float sx = 16.0f;
  const float out_scale = 1.0f / (w_scale * sx);
  switch (p.tag) {
    case 1: return launch_ws<1>(p, pk, sx, out_scale, stream);
    case 2: return launch_ws<2>(p, pk, sx, out_scale, stream);
    case 3: return launch_ws<3>(p, pk, sx, out_scale, stream);
    default: return launch_ws<0>(p, pk, sx, out_scale, stream);
  }
}
