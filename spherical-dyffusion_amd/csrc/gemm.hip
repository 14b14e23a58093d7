// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), LDS-staged, register-prefetch double buffered.
//
// One kernel serves every GEMM-shaped op of the SFNO block:
//   * 1x1 convolutions (encoder / inner skip / MLP / decoder): A = transposed weight [Cin][Cout], B = activations
//     [Cin][HW] of one sample, fused prologue (InstanceNorm affine on B rows) and epilogue (bias, residual, GELU,
//     dropout, drop-path, residual)                         -- src/models/sfno/sfnonet.py:303-335, layers.py:73-80
//   * Legendre analysis / synthesis, batched over m, triangular (l >= m)   -- torch_harmonics RealSHT/InverseRealSHT
//   * dhconv complex contraction, batched over l, rows (m <= l)            -- src/models/sfno/contractions.py:159-169
//
// Tile: BM x BN x 32 with 4 waves (2 x 2), each wave WM x WN MFMA tiles of 32 x 32.
//   MFMA operand map (cdna_hip_programming.md section 3): lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
//   C/D: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
// LDS images are k-major (As[k][m], Bs[k][n]) so one ds_read_b32 per operand is bank-conflict free.
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BK = 32;

// TAG only changes the symbol name: identical code, separate rows in rocprofv3 --stats (fc1 / fc2 / inner skip).
template <int WM, int WN, bool A_KCONTIG, bool B_CPLX, int TAG>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int LDA_S = A_KCONTIG ? BM + 1 : BM;
  constexpr int LDB_S = BN;
  constexpr int A_TILE = BK * LDA_S, B_TILE = BK * LDB_S;
  constexpr int NA = BM / 32, NB = BN / 32;  // float4 loads per thread per tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][A_TILE]
  float* Bs = smem + 2 * A_TILE;    // [2][B_TILE]   (2*A_TILE is a multiple of 4 floats for every instantiation)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int z = blockIdx.z;
  // conv mode (no triangular structure): XCD-aware tile order -- the M-tiles sharing one activation tile get ids L,
  // L+8, ... so they run back to back on one XCD and re-read the tile from its L2 (see gemm_h3.hip); batched
  // Legendre / dhconv keep the plain (n fastest) order
  int m0, n0;
  if (p.tri_mode == SDY_TRI_NONE) {
    const int MT = gridDim.x, NT = gridDim.y;
    const int L = blockIdx.x + MT * blockIdx.y;
    const int full = (NT / 8) * 8 * MT;
    int mt, nt;
    if (L < full) {
      const int xcd = L & 7, slot = L >> 3;
      nt = (slot / MT) * 8 + xcd;
      mt = slot % MT;
    } else {
      const int r = L - full;
      nt = (NT / 8) * 8 + r / MT;
      mt = r % MT;
    }
    m0 = mt * BM;
    n0 = nt * BN;
  } else {
    m0 = blockIdx.y * BM;
    n0 = blockIdx.x * BN;
  }

  int M_valid = p.M, k_lo = 0;
  const int k_hi = p.K;
  if (p.tri_mode == SDY_TRI_LEG_FWD) {
    if (m0 + BM <= z) return;  // every row l of this tile has l < m: structurally zero, never read downstream
  } else if (p.tri_mode == SDY_TRI_LEG_INV) {
    k_lo = z;                  // P[m][l][k] = 0 for l < m
  } else if (p.tri_mode == SDY_TRI_DHCONV) {
    M_valid = min(p.M, (z + 1) * p.tri_B);  // rows (m, b) with m <= l
    if (m0 >= M_valid) return;
  }
  const int M_store = min(p.M_store, M_valid);

  const float* __restrict__ Ag = p.A + (long)z * p.sA;
  const float* __restrict__ Bg = p.B + (long)z * p.sB;
  const float* __restrict__ pa = p.pa ? p.pa + (long)z * p.p_bstride : nullptr;
  const float* __restrict__ pd = p.pd ? p.pd + (long)z * p.p_bstride : nullptr;

  f32x4 ra[NA], rb[NB];
  float pa_r[NB], pd_r[NB];  // prologue affine of each staged B row, applied at LDS-store time so that the
                             // global loads stay in flight under the MFMAs (no dependent math in load_tile)
  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + i * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if constexpr (!A_KCONTIG) {
        const int row = idx / (BM / 4), c4 = idx % (BM / 4);
        const int gk = k0 + row, gm = m0 + c4 * 4;
        if (gk >= k_lo && gk < k_hi && gm < p.M) v = *reinterpret_cast<const f32x4*>(Ag + (long)gk * p.lda + gm);
      } else {
        const int r = idx >> 3, c4 = idx & 7;
        const int gm = m0 + r, gk = k0 + c4 * 4;
        if (gm < M_valid && gk < k_hi) v = *reinterpret_cast<const f32x4*>(Ag + (long)gm * p.lda + gk);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (BN / 4), c4 = idx % (BN / 4);
      const int gk = k0 + row, gn = n0 + c4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      pa_r[i] = 0.f;
      pd_r[i] = 0.f;
      if (gk >= k_lo && gk < k_hi && gn < p.N) {
        if constexpr (!B_CPLX) {
          v = *reinterpret_cast<const f32x4*>(Bg + (long)gk * p.ldb + gn);
          if (pa) {
            pa_r[i] = pa[gk];
            pd_r[i] = pd[gk];
          }
        } else {
          // expanded real form of the complex weight: rows (ri_in, i), cols (ri_out, o):
          //   [[ wr, wi ], [ -wi, wr ]]   so that  [xr xi] * W' = [xr wr - xi wi, xr wi + xi wr]
          const int Ei = p.cplx_Ei, Eo = p.cplx_Eo;
          const int qi = gk >= Ei, ki = gk - qi * Ei;
          const int qo = gn >= Eo, no = gn - qo * Eo;
          const float* src = Bg + (long)((qi != qo) ? 1 : 0) * Ei * Eo + (long)ki * Eo + no;
          v = *reinterpret_cast<const f32x4*>(src);
          if (qi && !qo) v = -v;
        }
      }
      rb[i] = v;
    }
  };

  auto store_tile = [&](int buf) {
    float* as = As + buf * A_TILE;
    float* bs = Bs + buf * B_TILE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + i * 256;
      if constexpr (!A_KCONTIG) {
        const int row = idx / (BM / 4), c4 = idx % (BM / 4);
        *reinterpret_cast<f32x4*>(as + row * LDA_S + c4 * 4) = ra[i];
      } else {
        const int r = idx >> 3, c4 = idx & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) as[(c4 * 4 + j) * LDA_S + r] = ra[i][j];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (BN / 4), c4 = idx % (BN / 4);
      f32x4 v = rb[i];
      if constexpr (!B_CPLX) {
        if (pa) v = v * pa_r[i] + pd_r[i];
      }
      *reinterpret_cast<f32x4*>(bs + row * LDB_S + c4 * 4) = v;
    }
  };

  const int kt_begin = k_lo / BK;
  const int kt_end = (k_hi + BK - 1) / BK;

  if (kt_begin < kt_end) {
    load_tile(kt_begin);
    store_tile(0);
  }
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    const bool more = (kt + 1) < kt_end;
    if (more) load_tile(kt + 1);  // global loads stay in flight under the MFMAs below

    const float* as = As + buf * A_TILE + wr * (32 * WM) + l31;
    const float* bs = Bs + buf * B_TILE + wc * (32 * WN) + l31;
    // software-pipelined fragment reads: the ds_reads of k-pair kp+1 are issued before the MFMAs of k-pair kp
    float a_cur[WM], b_cur[WN], a_nxt[WM], b_nxt[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a_cur[i] = as[h * LDA_S + i * 32];
#pragma unroll
    for (int j = 0; j < WN; ++j) b_cur[j] = bs[h * LDB_S + j * 32];
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      if (kp + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a_nxt[i] = as[(2 * kp + 2 + h) * LDA_S + i * 32];
#pragma unroll
        for (int j = 0; j < WN; ++j) b_nxt[j] = bs[(2 * kp + 2 + h) * LDB_S + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (hipcc otherwise sinks it behind them)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kp + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
        for (int j = 0; j < WN; ++j) b_cur[j] = b_nxt[j];
      }
    }

    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue<WM, WN>(acc, p, z, m0, n0, M_store, 1.0f);
}

template <int WM, int WN, bool A_KCONTIG, bool B_CPLX, int TAG = 0>
int launch_inst(const GemmParams& p, hipStream_t stream) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int LDA_S = A_KCONTIG ? BM + 1 : BM;
  constexpr size_t smem = (size_t)2 * BK * (LDA_S + BN) * sizeof(float);
  static SdyOncePerDevice once;
  std::atomic<bool>* attr_done = nullptr;
  SDY_TRY(once.slot(&attr_done));
  if (!*attr_done) {
    if (smem > 48 * 1024) {
      SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<WM, WN, A_KCONTIG, B_CPLX, TAG>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    *attr_done = true;
  }
  const unsigned gn = (p.N + BN - 1) / BN, gm = (p.M + BM - 1) / BM;
  dim3 grid(p.tri_mode == SDY_TRI_NONE ? gm : gn, p.tri_mode == SDY_TRI_NONE ? gn : gm, p.nbatch);
  hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, A_KCONTIG, B_CPLX, TAG>), grid, dim3(256), smem, stream, p);
  return sdy_launch_status();
}

}  // namespace

int sdy_gemm_launch(const GemmParams& p, hipStream_t stream) {
  if (!p.A || !p.B || !p.C) return SDY_ERR_ARG;
  if (p.M <= 0 || p.N <= 0 || p.K < 0 || p.nbatch <= 0) return SDY_ERR_ARG;
  // float4 granularity along every contiguous dimension
  if ((p.N & 3) || (p.ldb & 3)) return SDY_ERR_ALIGN;
  if (!p.b_cplx && (p.sB & 3)) return SDY_ERR_ALIGN;
  if (p.a_kcontig) {
    if ((p.K & 3) || (p.lda & 3) || (p.sA & 3)) return SDY_ERR_ALIGN;
  } else {
    if ((p.M & 3) || (p.lda & 3) || (p.sA & 3)) return SDY_ERR_ALIGN;
  }
  if (p.b_cplx && ((p.cplx_Ei & 3) || (p.cplx_Eo & 3))) return SDY_ERR_ALIGN;
  if (p.drop_thr != 0u && !p.keep_mask && (p.M_store & 3)) return SDY_ERR_ALIGN;

  if (p.a_kcontig && p.b_cplx) return launch_inst<1, 2, true, true>(p, stream);
  if (p.a_kcontig || p.b_cplx) return SDY_ERR_UNSUPPORTED;
  if (p.tile == SDY_TILE_64x128) return launch_inst<1, 2, false, false>(p, stream);
  switch (p.tag) {
    case 1: return launch_inst<2, 2, false, false, 1>(p, stream);
    case 2: return launch_inst<2, 2, false, false, 2>(p, stream);
    case 3: return launch_inst<2, 2, false, false, 3>(p, stream);
    default: return launch_inst<2, 2, false, false, 0>(p, stream);
  }
}
