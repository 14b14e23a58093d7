// Split-precision ("3 x fp16") MFMA GEMM, gfx950: the 1x1 convolutions, the Legendre transforms and the dhconv.
//
// fp32-class accuracy at the f16 matrix-core rate: every fp32 operand is split as v = hi + lo with hi = fp16(v),
// lo = fp16(v - hi) (22 significant bits), and the product is accumulated in fp32 as
//     A.B  ~=  Ah.Bh + Ah.Bl + Al.Bh          (the dropped Al.Bl term is ~2^-22 relative)
// with v_mfma_f32_32x32x16_f16 (fp16 products are exact in fp32; accumulation is fp32).  Three MFMA passes at 16x the
// fp32-MFMA rate; the tests hold this path to the same tolerances as the fp32-MFMA kernels of gemm.hip.
//
// One operand is always a constant of the model (conv weight, Legendre table, expanded dhconv weight): it is split once
// on the host (`sdy_h3_pack_*`): fp16 hi | lo planes, [batch][rows_pad][Kpad], k contiguous, times a power of two so
// that the lo parts stay out of the fp16 subnormal range.  The other operand is an fp32 activation tensor split on the
// fly while it is staged into LDS:
//   MODE_COLS (conv, Legendre): packed = A [M][K];  activations B[K][N], n contiguous.  Each thread loads a 4(k) x 4(n)
//        block as four 16-byte loads, converts, transposes in registers and writes 8-byte (4 k) pieces, so the LDS image is
//        [n][k] -- the k-contiguous fragment the MFMA wants -- without a transposing kernel.
//   MODE_ROWS (dhconv): activations A[M][K], k contiguous (rows (m,b) of the coefficient layout): split, no transpose;
//        packed = B [N][K].
// LDS rows are 32 halfs + 8 pad (80 B): ds_read_b128 fragment reads and ds_write_b64 staging writes are conflict free.
// Global loads run TWO k-tiles ahead of the MFMAs (two register sets, loop unrolled by 2; hipcc's counted vmcnt waits
// retire only the set that is about to be converted), because at f16 MFMA rates one tile of compute (~0.3 us) is far
// shorter than a loaded memory round trip (~2 us): with one tile in flight the waves sat parked ~40 % of their life.
// The accumulator layout equals gemm.hip's, so the fused epilogue (bias / add / GELU / Philox dropout / residual) is shared.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HBK = 32;      // k-tile (two tiles are kept in flight, see the main loop)
constexpr int HLD = 40;      // LDS row stride in halfs (32 + 8 pad = 80 B)
constexpr int HCH = HBK / 8; // 16-byte chunks per packed row per k-tile
constexpr int HKQ = HBK / 4; // 4-wide k groups per k-tile

enum { MODE_COLS = 0, MODE_ROWS = 1 };

struct H3Packed {
  const _Float16* hi;   // [batch][rows_pad][Kpad]
  const _Float16* lo;
  int rows_pad, Kpad;
  long bstride;         // halfs between batches (0 = shared)
  unsigned* flags;      // sticky status word (sdy_status_flags): fp16 range guard of the activation split
};

__device__ __forceinline__ void split4(const f32x4 v, f16x4& vh, f16x4& vl) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 hv = (_Float16)v[e];
    vh[e] = hv;
    vl[e] = (_Float16)(v[e] - (float)hv);
  }
}

template <int WM, int WN, int MODE, int TAG>
__global__ __launch_bounds__(256, 2) void gemm_h3_kernel(const GemmParams p, const H3Packed pk, float sx, float out_scale) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int PROWS = (MODE == MODE_COLS) ? BM : BN;   // rows of the packed operand per tile
  constexpr int FROWS = (MODE == MODE_COLS) ? BN : BM;   // rows (LDS rows) of the fp32 operand per tile
  constexpr int NP = PROWS * HCH / 256;                  // 16-byte chunks (per plane) per thread per k-tile
  constexpr int NBLK = (HBK * FROWS) / (16 * 256);       // MODE_COLS: 4x4 blocks per thread
  constexpr int PQN = 256 / HKQ;                         // MODE_COLS: pixel quads covered per pass
  constexpr int NR = FROWS * HKQ / 256;                  // MODE_ROWS: float4 loads per thread
  static_assert((HBK * FROWS) % (16 * 256) == 0, "tile must split evenly");
  extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
  _Float16* As_hi = smem_h;
  _Float16* As_lo = As_hi + BM * HLD;
  _Float16* Bs_hi = As_lo + BM * HLD;
  _Float16* Bs_lo = Bs_hi + BN * HLD;
  _Float16* Ps_hi = (MODE == MODE_COLS) ? As_hi : Bs_hi;
  _Float16* Ps_lo = (MODE == MODE_COLS) ? As_lo : Bs_lo;
  _Float16* Fs_hi = (MODE == MODE_COLS) ? Bs_hi : As_hi;
  _Float16* Fs_lo = (MODE == MODE_COLS) ? Bs_lo : As_lo;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int z = blockIdx.z;

  int mt, nt;
  if (p.tri_mode == SDY_TRI_NONE) {
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (id % 8), each with a private L2.  The
    // M-tiles that share one activation tile get ids L, L+8, L+16, ... so they run back to back on ONE XCD and re-read
    // the tile from its L2 (speed only, never correctness).
    const int MT = gridDim.x, NT = gridDim.y;
    const int L = blockIdx.x + MT * blockIdx.y;
    const int full = (NT / 8) * 8 * MT;
    if (L < full) {
      const int xcd = L & 7, slot = L >> 3;
      nt = (slot / MT) * 8 + xcd;
      mt = slot % MT;
    } else {
      const int r = L - full;
      nt = (NT / 8) * 8 + r / MT;
      mt = r % MT;
    }
  } else {
    mt = blockIdx.x;
    nt = blockIdx.y;
  }
  const int m0 = mt * BM, n0 = nt * BN;

  // triangular structure of the spectral transforms (same rules as gemm.hip)
  int M_valid = p.M_store, k_lo = 0;
  if (p.tri_mode == SDY_TRI_LEG_FWD) {
    if (m0 + BM <= z) return;
  } else if (p.tri_mode == SDY_TRI_LEG_INV) {
    k_lo = z;
  } else if (p.tri_mode == SDY_TRI_DHCONV) {
    M_valid = min(p.M_store, (z + 1) * p.tri_B);
    if (m0 >= M_valid) return;
  }

  const _Float16* __restrict__ Ph = pk.hi + (long)z * pk.bstride;
  const _Float16* __restrict__ Pl = pk.lo + (long)z * pk.bstride;
  const float* __restrict__ Fg = (MODE == MODE_COLS ? p.B + (long)z * p.sB : p.A + (long)z * p.sA);
  const float* __restrict__ pa = p.pa ? p.pa + (long)z * p.p_bstride : nullptr;
  const float* __restrict__ pd = p.pd ? p.pd + (long)z * p.p_bstride : nullptr;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float amax = 0.0f;   // largest |scaled activation| this thread split into fp16 (range guard)
  struct Stage {
    f32x4 rp_hi[NP], rp_lo[NP];                  // raw 16-byte chunks of the pre-split operand
    f32x4 rb[MODE == MODE_COLS ? NBLK : 1][4];   // MODE_COLS: 4 k x 4 n
    float ca[MODE == MODE_COLS ? NBLK : 1][4], cd[MODE == MODE_COLS ? NBLK : 1][4];
    f32x4 rr[MODE == MODE_ROWS ? NR : 1];        // MODE_ROWS: 4 consecutive k of one row
  };
  Stage st0, st1;

  const int kq = tid % HKQ, pq = tid / HKQ;
  const int prow0 = (MODE == MODE_COLS) ? m0 : n0;   // first packed row of this tile

  // FAST (interior tile, K a multiple of the k-tile): no predicates and therefore no control flow between the loads, so
  // hipcc keeps exact vmcnt counts and the wait in store_tile retires only the stage being converted -- the younger
  // stage stays in flight.  (With `if`-guarded loads it falls back to vmcnt(0) and the two-stage prefetch is moot.)
  auto load_tile = [&](Stage& S, int kt, auto fast_tag, auto aff_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    constexpr bool AFF = decltype(aff_tag)::value;
    const int k0 = kt * HBK;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int chunk = tid + i * 256;
      const int row = chunk / HCH, c = chunk % HCH;
      const long off = (long)(prow0 + row) * pk.Kpad + k0 + c * 8;   // rows / k are padded: no predicate
      S.rp_hi[i] = *reinterpret_cast<const f32x4*>(Ph + off);
      S.rp_lo[i] = *reinterpret_cast<const f32x4*>(Pl + off);
    }
    if constexpr (MODE == MODE_COLS) {
#pragma unroll
      for (int bi = 0; bi < NBLK; ++bi) {
        const int gn = n0 + (bi * PQN + pq) * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int gk = k0 + kq * 4 + kk;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          float a = 0.f, d = 0.f;
          if constexpr (FAST) {
            v = *reinterpret_cast<const f32x4*>(Fg + (long)gk * p.ldb + gn);
            a = 1.0f;
            if constexpr (AFF) {
              a = pa[gk];
              d = pd[gk];
            }
          } else if (gk >= k_lo && gk < p.K && gn < p.N) {
            v = *reinterpret_cast<const f32x4*>(Fg + (long)gk * p.ldb + gn);
            a = 1.0f;
            if (pa) {
              a = pa[gk];
              d = pd[gk];
            }
          }
          S.rb[bi][kk] = v;
          S.ca[bi][kk] = a;
          S.cd[bi][kk] = d;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / HKQ, q = idx % HKQ;
        const int gm = m0 + row, gk = k0 + q * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FAST)
          v = *reinterpret_cast<const f32x4*>(Fg + (long)gm * p.lda + gk);
        else if (gm < M_valid && gk < p.K)
          v = *reinterpret_cast<const f32x4*>(Fg + (long)gm * p.lda + gk);
        S.rr[i] = v;
      }
    }
  };

  auto store_tile = [&](const Stage& S) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int chunk = tid + i * 256;
      const int row = chunk / HCH, c = chunk % HCH;
      *reinterpret_cast<f32x4*>(Ps_hi + row * HLD + c * 8) = S.rp_hi[i];
      *reinterpret_cast<f32x4*>(Ps_lo + row * HLD + c * 8) = S.rp_lo[i];
    }
    if constexpr (MODE == MODE_COLS) {
#pragma unroll
      for (int bi = 0; bi < NBLK; ++bi) {
        _Float16 hi[4][4], lo[4][4];   // [k kk][n pp]
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const float a = S.ca[bi][kk] * sx, d = S.cd[bi][kk] * sx;
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) {
            const float v = fmaf(S.rb[bi][kk][pp], a, d);
            amax = __builtin_fmaxf(amax, __builtin_fabsf(v));
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
          *reinterpret_cast<f16x4*>(Fs_hi + prow * HLD + kq * 4) = vh;
          *reinterpret_cast<f16x4*>(Fs_lo + prow * HLD + kq * 4) = vl;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / HKQ, q = idx % HKQ;
        f16x4 vh, vl;
        const f32x4 sv = S.rr[i] * sx;
        amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(sv.x), __builtin_fabsf(sv.y)),
                                                     __builtin_fmaxf(__builtin_fabsf(sv.z), __builtin_fabsf(sv.w))));
        split4(sv, vh, vl);
        *reinterpret_cast<f16x4*>(Fs_hi + row * HLD + q * 4) = vh;
        *reinterpret_cast<f16x4*>(Fs_lo + row * HLD + q * 4) = vl;
      }
    }
  };

  auto compute_tile = [&]() {
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
          SDY_CROSS_TERM(acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0));
          SDY_CROSS_TERM(acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0));
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  const int kt_begin = k_lo / HBK;
  const int kt_end = (p.K + HBK - 1) / HBK;
  auto run = [&](auto fast_tag, auto aff_tag) {
    if (kt_begin < kt_end) load_tile(st0, kt_begin, fast_tag, aff_tag);
    if (kt_begin + 1 < kt_end) load_tile(st1, kt_begin + 1, fast_tag, aff_tag);
    for (int kt = kt_begin; kt < kt_end; kt += 2) {
      store_tile(st0);                                  // waits for set 0 only: set 1 stays in flight
      __syncthreads();
      if (kt + 2 < kt_end) load_tile(st0, kt + 2, fast_tag, aff_tag);
      compute_tile();
      __syncthreads();
      if (kt + 1 < kt_end) {
        store_tile(st1);
        __syncthreads();
        if (kt + 3 < kt_end) load_tile(st1, kt + 3, fast_tag, aff_tag);
        compute_tile();
        __syncthreads();
      }
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  const bool interior = (p.K % HBK == 0) && (k_lo == 0) &&
                        (MODE == MODE_COLS ? (n0 + BN <= p.N) : (m0 + BM <= M_valid));   // workgroup-uniform
  if (interior) {
    if (pa) run(T_{}, T_{}); else run(T_{}, F_{});
  } else {
    run(F_{}, F_{});
  }
  sdy_flag_range(pk.flags, amax);

  gemm_epilogue<WM, WN>(acc, p, z, m0, n0, M_valid, out_scale);
}

template <int WM, int WN, int MODE, int TAG>
int launch_h3(const GemmParams& p, const H3Packed& pk, float sx, float out_scale, hipStream_t stream) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr size_t smem = (size_t)(2 * BM + 2 * BN) * HLD * sizeof(_Float16);
  static SdyOncePerDevice once;
  std::atomic<bool>* attr_done = nullptr;
  SDY_TRY(once.slot(&attr_done));
  if (!*attr_done) {
    SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3_kernel<WM, WN, MODE, TAG>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    *attr_done = true;
  }
  dim3 grid((p.M_store + BM - 1) / BM, (p.N + BN - 1) / BN, p.nbatch);
  hipLaunchKernelGGL((gemm_h3_kernel<WM, WN, MODE, TAG>), grid, dim3(256), smem, stream, p, pk, sx, out_scale);
  return sdy_launch_status();
}


// ---- wide variant (MODE_COLS only): 512 threads, 256 x 128 x 64 tile, 8 waves as 4 (M) x 2 (N) ------------------------
// One workgroup covers ALL rows of a <=256-row packed operand (256-wide conv layers, the 180-row Legendre tables), so the
// activation tile is fetched, converted and split ONCE instead of once per 128-row M-tile.
constexpr int WBK = 64, WLD = 72, WCH = WBK / 8, WKQ = WBK / 4;

template <int TAG>
__global__ __launch_bounds__(512) void gemm_h3_wide_kernel(const GemmParams p, const H3Packed pk, float sx,
                                                            float out_scale) {
  constexpr int BM = 256, BN = 128, WM = 2, WN = 2;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
  _Float16* As_hi = smem_h;
  _Float16* As_lo = As_hi + BM * WLD;
  _Float16* Bs_hi = As_lo + BM * WLD;
  _Float16* Bs_lo = Bs_hi + BN * WLD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;   // 4 x 2
  const int h = lane >> 5, l31 = lane & 31;
  const int z = blockIdx.z;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  int k_lo = 0;
  if (p.tri_mode == SDY_TRI_LEG_FWD) {
    if (m0 + BM <= z) return;
  } else if (p.tri_mode == SDY_TRI_LEG_INV) {
    k_lo = z;
  }
  // rows of this wave that lie entirely below the diagonal of a Legendre analysis produce zeros that nobody reads
  const bool wave_dead = (p.tri_mode == SDY_TRI_LEG_FWD) && (m0 + wr * 64 + 64 <= z);

  const _Float16* __restrict__ Ph = pk.hi + (long)z * pk.bstride;
  const _Float16* __restrict__ Pl = pk.lo + (long)z * pk.bstride;
  const float* __restrict__ Fg = p.B + (long)z * p.sB;
  const float* __restrict__ pa = p.pa ? p.pa + (long)z * p.p_bstride : nullptr;
  const float* __restrict__ pd = p.pd ? p.pd + (long)z * p.p_bstride : nullptr;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float amax = 0.0f;   // largest |scaled activation| this thread split into fp16 (range guard)

  f32x4 rp_hi[4], rp_lo[4];
  f32x4 rb[4];
  float ca[4], cd[4];
  const int kq = tid % WKQ, pq = tid / WKQ;   // 16 k-quads x 32 pixel quads = 64 k x 128 pixels

  auto load_tile = [&](int kt) {
    const int k0 = kt * WBK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int chunk = tid + i * 512;
      const int row = chunk / WCH, c = chunk % WCH;
      const long off = (long)(m0 + row) * pk.Kpad + k0 + c * 8;
      rp_hi[i] = *reinterpret_cast<const f32x4*>(Ph + off);
      rp_lo[i] = *reinterpret_cast<const f32x4*>(Pl + off);
    }
    const int gn = n0 + pq * 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int gk = k0 + kq * 4 + kk;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      float a = 0.f, d = 0.f;
      if (gk >= k_lo && gk < p.K && gn < p.N) {
        v = *reinterpret_cast<const f32x4*>(Fg + (long)gk * p.ldb + gn);
        a = 1.0f;
        if (pa) {
          a = pa[gk];
          d = pd[gk];
        }
      }
      rb[kk] = v;
      ca[kk] = a;
      cd[kk] = d;
    }
  };

  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int chunk = tid + i * 512;
      const int row = chunk / WCH, c = chunk % WCH;
      *reinterpret_cast<f32x4*>(As_hi + row * WLD + c * 8) = rp_hi[i];
      *reinterpret_cast<f32x4*>(As_lo + row * WLD + c * 8) = rp_lo[i];
    }
    _Float16 hi[4][4], lo[4][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float a = ca[kk] * sx, d = cd[kk] * sx;
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const float v = fmaf(rb[kk][pp], a, d);
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v));
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
      *reinterpret_cast<f16x4*>(Bs_hi + prow * WLD + kq * 4) = vh;
      *reinterpret_cast<f16x4*>(Bs_lo + prow * WLD + kq * 4) = vl;
    }
  };

  const int kt_begin = k_lo / WBK;
  const int kt_end = (p.K + WBK - 1) / WBK;
  if (kt_begin < kt_end) load_tile(kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < kt_end) load_tile(kt + 1);
    if (!wave_dead) {
#pragma unroll
      for (int s = 0; s < WBK / 16; ++s) {
        f16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const int off = (wr * 64 + i * 32 + l31) * WLD + s * 16 + h * 8;
          ah[i] = *reinterpret_cast<const f16x8*>(As_hi + off);
          al[i] = *reinterpret_cast<const f16x8*>(As_lo + off);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int off = (wc * 64 + j * 32 + l31) * WLD + s * 16 + h * 8;
          bh[j] = *reinterpret_cast<const f16x8*>(Bs_hi + off);
          bl[j] = *reinterpret_cast<const f16x8*>(Bs_lo + off);
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            SDY_CROSS_TERM(acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0));
            SDY_CROSS_TERM(acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0));
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }
  sdy_flag_range(pk.flags, amax);
  if (!wave_dead) gemm_epilogue_at<WM, WN>(acc, p, z, m0 + wr * 64, n0 + wc * 64, p.M_store, out_scale);
}

template <int TAG>
int launch_h3_wide(const GemmParams& p, const H3Packed& pk, float sx, float out_scale, hipStream_t stream) {
  constexpr size_t smem = (size_t)(2 * 256 + 2 * 128) * WLD * sizeof(_Float16);
  static SdyOncePerDevice once;
  std::atomic<bool>* attr_done = nullptr;
  SDY_TRY(once.slot(&attr_done));
  if (!*attr_done) {
    SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3_wide_kernel<TAG>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    *attr_done = true;
  }
  dim3 grid((p.M_store + 255) / 256, (p.N + 127) / 128, p.nbatch);
  hipLaunchKernelGGL((gemm_h3_wide_kernel<TAG>), grid, dim3(512), smem, stream, p, pk, sx, out_scale);
  return sdy_launch_status();
}

}  // namespace

// `p` carries the fp32 operand (B for rows_mode 0, A for rows_mode 1), C and the epilogue exactly as for
// sdy_gemm_launch.  packed: fp16 hi plane then lo plane (plane_halfs apart), each [nbatch][rows_pad][Kpad] with batch
// stride `bstride` halfs (0 = one matrix shared by every batch).
int sdy_gemm_h3_launch(const GemmParams& p, const void* packed, int rows_pad, int Kpad, long bstride, long plane_halfs,
                       float w_scale, int rows_mode, hipStream_t stream) {
  if (!packed || !p.C) return SDY_ERR_ARG;
  if (p.M_store <= 0 || p.N <= 0 || p.K <= 0 || p.nbatch <= 0) return SDY_ERR_ARG;
  if (Kpad % 64 || Kpad < p.K || rows_pad % 128) return SDY_ERR_ARG;
  if (p.drop_thr != 0u && !p.keep_mask && (p.M_store & 3)) return SDY_ERR_ALIGN;
  H3Packed pk;
  pk.hi = reinterpret_cast<const _Float16*>(packed);
  pk.lo = pk.hi + plane_halfs;
  pk.rows_pad = rows_pad;
  pk.Kpad = Kpad;
  pk.bstride = bstride;
  SDY_TRY(sdy_flags_ptr(&pk.flags));
  const float sx = SDY_ACT_SX;                       // keeps the lo parts of O(1) activations out of the fp16 subnormals
  const float out_scale = 1.0f / (w_scale * sx);
  if (rows_mode) {
    if (!p.A) return SDY_ERR_ARG;
    if ((p.K & 3) || (p.lda & 3) || (p.sA & 3)) return SDY_ERR_ALIGN;
    if (rows_pad < ((p.N + 127) / 128) * 128) return SDY_ERR_ARG;
    return launch_h3<2, 2, MODE_ROWS, 0>(p, pk, sx, out_scale, stream);
  }
  if (!p.B) return SDY_ERR_ARG;
  if ((p.N & 3) || (p.ldb & 3) || (p.sB & 3)) return SDY_ERR_ALIGN;
  if (rows_pad < ((p.M_store + 127) / 128) * 128) return SDY_ERR_ARG;
  // wide tile for 256-row conv layers (measured: 256 -> 256 and 512 -> 256 convs +8-10 %; the short-K, triangular
  // Legendre GEMMs and the 512-row fc1 are faster on the 128 x 128 kernel)
  if (p.tri_mode == SDY_TRI_NONE && p.M_store == 256 && rows_pad % 256 == 0) {
    switch (p.tag) {
      case 1: return launch_h3_wide<1>(p, pk, sx, out_scale, stream);
      case 2: return launch_h3_wide<2>(p, pk, sx, out_scale, stream);
      case 3: return launch_h3_wide<3>(p, pk, sx, out_scale, stream);
      default: return launch_h3_wide<0>(p, pk, sx, out_scale, stream);
    }
  }
  switch (p.tag) {
    case 1: return launch_h3<2, 2, MODE_COLS, 1>(p, pk, sx, out_scale, stream);
    case 2: return launch_h3<2, 2, MODE_COLS, 2>(p, pk, sx, out_scale, stream);
    case 3: return launch_h3<2, 2, MODE_COLS, 3>(p, pk, sx, out_scale, stream);
    default: return launch_h3<2, 2, MODE_COLS, 0>(p, pk, sx, out_scale, stream);
  }
}
