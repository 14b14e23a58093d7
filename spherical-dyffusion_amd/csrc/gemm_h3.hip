// Split-precision ("3 x fp16") MFMA GEMM for the 1x1 convolutions, gfx950.
//
// fp32-class accuracy at the f16 matrix-core rate: every fp32 operand is split as v = hi + lo with hi = fp16(v),
// lo = fp16(v - hi) (22 significant bits), and the product is accumulated in fp32 as
//     A.B  ~=  Ah.Bh + Ah.Bl + Al.Bh          (the dropped Al.Bl term is ~2^-22 relative)
// with v_mfma_f32_32x32x16_f16 (fp16 products are exact in fp32; accumulation is fp32).  Three MFMA passes at 16x the
// fp32-MFMA rate = ~5x the throughput ceiling of gemm.hip at the same parity (tests hold it to the same tolerances).
//
//   out[b][o][p] = epilogue( (1/(sw*sx)) * sum_i  split(sw*W[o][i]) . split(sx*(pa[b][i]*x[b][i][p] + pd[b][i])) )
//
// * weights are split once at load time on the host (`sdy_h3_pack_weight`): [Mpad][Kpad] fp16 hi | lo, k contiguous,
//   scaled by a power of two `sw` so that the lo parts stay out of the fp16 subnormal range
// * activations are split on the fly while they are staged: each thread loads a 4(channel) x 4(pixel) block as four
//   16-byte loads, converts, transposes in registers and writes 8-byte (4 channel) pieces, so the LDS image is
//   [pixel][channel] -- the k-contiguous fragment the MFMA wants -- without a transposing kernel
// * LDS rows are 32 halfs + 8 pad (80 B): ds_read_b128 fragment reads and ds_write_b64 staging writes are both
//   bank-conflict free
// * accumulator layout == gemm.hip, so the fused epilogue (bias / add / GELU / Philox dropout / residual) is shared
#include "common.h"
#include "gemm_epilogue.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HBK = 64;      // k-tile (channels): two 128-byte fp16 lines per weight row, 8 KB of activations per 32 px
constexpr int HLD = 72;      // LDS row stride in halfs (64 + 8 pad = 144 B: ds_read_b128 / ds_write_b64 conflict free)
constexpr int HCH = HBK / 8; // 16-byte chunks per weight row per k-tile
constexpr int HKQ = HBK / 4; // 4-channel groups per k-tile

template <int WM, int WN, int TAG>
__global__ __launch_bounds__(256) void gemm_h3_kernel(const GemmParams p, const _Float16* __restrict__ Ah_g,
                                                       const _Float16* __restrict__ Al_g, int Kpad, float sx,
                                                       float out_scale) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int NA = BM * HCH / 256;      // 16-byte chunks of A (per hi / lo) per thread per k-tile
  constexpr int NBLK = (HBK * BN) / (16 * 256);  // 4x4 activation blocks per thread per k-tile
  constexpr int PQN = 256 / HKQ;          // pixel quads covered by one pass of the workgroup
  static_assert((HBK * BN) % (16 * 256) == 0, "tile must split into 4x4 blocks");
  extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
  _Float16* As_hi = smem_h;
  _Float16* As_lo = As_hi + BM * HLD;
  _Float16* Bs_hi = As_lo + BM * HLD;
  _Float16* Bs_lo = Bs_hi + BN * HLD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int z = blockIdx.z;
  // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (id % 8), each with a private L2.  The
  // M-tiles that share one activation tile are given ids L, L+8, L+16, ... so they run back to back on ONE XCD: the
  // tile is fetched from HBM once and re-read from that XCD's L2 by its siblings (speed only, never correctness).
  int mt, nt;
  {
    const int MT = gridDim.x, NT = gridDim.y;
    const int L = blockIdx.x + MT * blockIdx.y;
    const int full = (NT / 8) * 8 * MT;     // ids covered by complete groups of 8 pixel tiles
    if (L < full) {
      const int xcd = L & 7, slot = L >> 3;
      nt = (slot / MT) * 8 + xcd;
      mt = slot % MT;
    } else {                                 // ragged tail: plain m-fastest order
      const int r = L - full;
      nt = (NT / 8) * 8 + r / MT;
      mt = r % MT;
    }
  }
  const int m0 = mt * BM, n0 = nt * BN;

  const float* __restrict__ Bg = p.B + (long)z * p.sB;
  const float* __restrict__ pa = p.pa ? p.pa + (long)z * p.p_bstride : nullptr;
  const float* __restrict__ pd = p.pd ? p.pd + (long)z * p.p_bstride : nullptr;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // staged registers
  f32x4 ra_hi[NA], ra_lo[NA];             // raw 16-byte chunks of the pre-split weights
  f32x4 rb[NBLK][4];                      // 4 channels x 4 pixels (fp32)
  float ca[NBLK][4], cd[NBLK][4];         // affine of the 4 channels

  const int kq = tid % HKQ, pq = tid / HKQ;  // activation block: channels 4*kq.., pixels 4*pq.. (+4*PQN per extra block)

  auto load_tile = [&](int kt) {
    const int k0 = kt * HBK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int chunk = tid + i * 256;
      const int row = chunk / HCH, c = chunk % HCH;
      const long off = (long)(m0 + row) * Kpad + k0 + c * 8;   // rows are padded to a multiple of BM: no predicate
      ra_hi[i] = *reinterpret_cast<const f32x4*>(Ah_g + off);
      ra_lo[i] = *reinterpret_cast<const f32x4*>(Al_g + off);
    }
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi) {
      const int gn = n0 + (bi * PQN + pq) * 4;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int gk = k0 + kq * 4 + kk;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        float a = 0.f, d = 0.f;
        if (gk < p.K && gn < p.N) {
          v = *reinterpret_cast<const f32x4*>(Bg + (long)gk * p.ldb + gn);
          a = 1.0f;
          if (pa) {
            a = pa[gk];
            d = pd[gk];
          }
        }
        rb[bi][kk] = v;
        ca[bi][kk] = a;
        cd[bi][kk] = d;
      }
    }
  };

  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int chunk = tid + i * 256;
      const int row = chunk / HCH, c = chunk % HCH;
      *reinterpret_cast<f32x4*>(As_hi + row * HLD + c * 8) = ra_hi[i];
      *reinterpret_cast<f32x4*>(As_lo + row * HLD + c * 8) = ra_lo[i];
    }
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi) {
      _Float16 hi[4][4], lo[4][4];   // [channel kk][pixel pp]
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float a = ca[bi][kk] * sx, d = cd[bi][kk] * sx;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const float v = fmaf(rb[bi][kk][pp], a, d);
          const _Float16 hv = (_Float16)v;
          hi[kk][pp] = hv;
          lo[kk][pp] = (_Float16)(v - (float)hv);
        }
      }
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int prow = (bi * PQN + pq) * 4 + pp;
        f16x4 vh = {hi[0][pp], hi[1][pp], hi[2][pp], hi[3][pp]};
        f16x4 vl = {lo[0][pp], lo[1][pp], lo[2][pp], lo[3][pp]};
        *reinterpret_cast<f16x4*>(Bs_hi + prow * HLD + kq * 4) = vh;
        *reinterpret_cast<f16x4*>(Bs_lo + prow * HLD + kq * 4) = vl;
      }
    }
  };

  const int nk = (p.K + HBK - 1) / HBK;
  if (nk > 0) load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);   // global loads in flight under the MFMAs

#pragma unroll
    for (int s = 0; s < HBK / 16; ++s) {
      f16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int off = (wr * (32 * WM) + i * 32 + l31) * HLD + s * 16 + h * 8;
        ah[i] = *reinterpret_cast<const f16x8*>(As_hi + off);
        al[i] = *reinterpret_cast<const f16x8*>(As_lo + off);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int off = (wc * (32 * WN) + j * 32 + l31) * HLD + s * 16 + h * 8;
        bh[j] = *reinterpret_cast<const f16x8*>(Bs_hi + off);
        bl[j] = *reinterpret_cast<const f16x8*>(Bs_lo + off);
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  gemm_epilogue<WM, WN>(acc, p, z, m0, n0, p.M_store, out_scale);
}

template <int WM, int WN, int TAG>
int launch_h3(const GemmParams& p, const _Float16* Ah, const _Float16* Al, int Kpad, float sx, float out_scale,
              hipStream_t stream) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr size_t smem = (size_t)(2 * BM + 2 * BN) * HLD * sizeof(_Float16);
  static bool attr_done = false;
  if (!attr_done) {
    SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3_kernel<WM, WN, TAG>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_done = true;
  }
  dim3 grid((p.M_store + BM - 1) / BM, (p.N + BN - 1) / BN, p.nbatch);
  hipLaunchKernelGGL((gemm_h3_kernel<WM, WN, TAG>), grid, dim3(256), smem, stream, p, Ah, Al, Kpad, sx, out_scale);
  return sdy_launch_status();
}


}  // namespace

// `p` carries B / C / epilogue exactly as for sdy_gemm_launch (A fields unused); packed = [Mpad][Kpad] hi then lo.
int sdy_gemm_h3_launch(const GemmParams& p, const void* packed, int Mpad, int Kpad, float w_scale, hipStream_t stream) {
  if (!packed || !p.B || !p.C) return SDY_ERR_ARG;
  if (p.M_store <= 0 || p.N <= 0 || p.K <= 0 || p.nbatch <= 0) return SDY_ERR_ARG;
  if ((p.N & 3) || (p.ldb & 3) || (p.sB & 3)) return SDY_ERR_ALIGN;
  if (Kpad % HBK || Kpad < p.K || Mpad % 128 || Mpad < p.M_store) return SDY_ERR_ARG;
  if (p.drop_thr != 0u && !p.keep_mask && (p.M_store & 3)) return SDY_ERR_ALIGN;
  const _Float16* Ah = reinterpret_cast<const _Float16*>(packed);
  const _Float16* Al = Ah + (size_t)Mpad * Kpad;
  const float sx = 16.0f;                       // keeps the lo parts of O(1) activations out of the fp16 subnormals
  const float out_scale = 1.0f / (w_scale * sx);
  switch (p.tag) {
    case 1: return launch_h3<2, 2, 1>(p, Ah, Al, Kpad, sx, out_scale, stream);
    case 2: return launch_h3<2, 2, 2>(p, Ah, Al, Kpad, sx, out_scale, stream);
    case 3: return launch_h3<2, 2, 3>(p, Ah, Al, Kpad, sx, out_scale, stream);
    default: return launch_h3<2, 2, 0>(p, Ah, Al, Kpad, sx, out_scale, stream);
  }
}
