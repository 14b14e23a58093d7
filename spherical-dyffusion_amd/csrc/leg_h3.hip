// Legendre transforms (both directions) for grids with nlat, lmax <= 192 -- the 180 x 360 model grid -- as one skinny
// split-fp16 GEMM per zonal order m:   C_m[M][N] = T_m[M][K] . X_m[K][N],   M, K <= 192,  N = 2 * B * C columns.
//     analysis  (sdy_legendre_fwd): T = quadrature-weighted P_l^m (rows l, K = latitude), X = rfft output of order m
//     synthesis (sdy_legendre_inv): T = P_l^m (rows = latitude, K = l),                   X = spectral coefficients
// The generic tile GEMM (gemm_h3.hip) re-stages the SAME 192 x 192 table through LDS for every 128-column tile and pads
// M to 256: as much L2->LDS table traffic as activation traffic, at ~2 TB/s effective.  Here (the fused-MLP recipe):
//   * one workgroup (3 waves) owns 64 columns and ALL rows: wave w computes rows 64w .. 64w+63 (2 x 2 MFMA tiles);
//   * the activation tile [192 k][64 n] is fetched once, split hi/lo and parked in LDS ([n][k], XOR-swizzled);
//   * the table never touches LDS: it is pre-packed per (m, wave) as a linear stream of MFMA A-fragment pairs (hi, lo)
//     in consumption order and flows L2 -> registers through an 8-group ring, 4 k-steps ahead of the MFMAs;
//   * triangular structure: analysis skips waves whose rows all have l < m and never stores rows l < m; synthesis skips
//     the k-steps below m (in the table stream too);
//   * the result goes through LDS once more so that global stores are 16-byte row segments.
#include <cmath>
#include <vector>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int LM = 192;          // rows per workgroup (>= lmax, nlat)
constexpr int LKP = 192;         // LDS row length in halfs (384 B = 24 chunks of 8 k)
constexpr int LTN = 64;          // columns per workgroup
constexpr int LKS = 12;          // k-steps of 16
constexpr int LRING = 8;         // groups in flight = one block of 4 k-steps x 2 m-tiles
constexpr int LGPW = 2 * LKS;    // groups per (m, wave)
constexpr int LGROUP = 2 * 64;   // f16x8 elements per group
constexpr float LSX = SDY_ACT_SX;

struct LegParams {
  const f16x8* table;            // [nz][3 waves][LGPW groups][hi | lo][64 lanes] (+ LRING groups of padding)
  const float* X; long ldx, sX;  // activations: row k at X + z * sX + k * ldx, columns contiguous
  float* C; long ldc, sC;        // result: row r at C + z * sC + r * ldc
  int M_store, K, N;
  int tri;                       // SDY_TRI_LEG_FWD / SDY_TRI_LEG_INV
  float out_scale;
  unsigned* flags;               // sticky status word (sdy_status_flags)
};

// 384-byte rows: the 16-byte slot of (px, chunk c) in the 256-byte bank row is (8 * (px & 1) + c) mod 16.  XOR-ing the low
// 3 bits of c with (px >> 1) & 7 makes the 16 pixels of every ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,
// 28-31}) land on 16 different slots: conflict-free fragment reads, 48 KB per workgroup, three workgroups per CU.
__device__ __forceinline__ int lg_off(int px, int c) { return px * LKP + (((c & ~7) | ((c ^ (px >> 1)) & 7)) << 3); }

__global__ __launch_bounds__(192, 3) void leg_h3_kernel(const LegParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * LTN * LKP * 2];   // 48 KB
  _Float16* Xs_hi = reinterpret_cast<_Float16*>(smem);
  _Float16* Xs_lo = Xs_hi + LTN * LKP;
  float* Os = reinterpret_cast<float*>(smem);   // epilogue: [192 rows][64 cols] fp32 = 48 KB (aliases the x tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int z = blockIdx.y;
  const int n0 = blockIdx.x * LTN;
  const bool full = n0 + LTN <= p.N;

  const bool fwd = p.tri == SDY_TRI_LEG_FWD;
  const int row_lo = fwd ? z : 0;                        // analysis: rows l < m are never read downstream
  const bool wave_dead = fwd && (64 * wave + 63 < z);    // all rows of this wave have l < m
  const int ks0 = fwd ? 0 : z >> 4;                      // synthesis: k-steps entirely below l = m contribute nothing
  const int kb0 = ks0 >> 2;                              // the loop runs in blocks of 4 k-steps
  const int k_lo = fwd ? 0 : z;                          // synthesis: rows l < m of the coefficients were never written

  // ---- table ring (slot = 2 * (k-step % 4) + m-tile)
  f16x8 r_hi[LRING], r_lo[LRING];
  const f16x8* __restrict__ wp = p.table + ((size_t)(z * 3 + wave) * LGPW + 8 * kb0) * LGROUP + lane;
#pragma unroll
  for (int s = 0; s < LRING; ++s) {
    r_hi[s] = wp[s * LGROUP];
    r_lo[s] = wp[s * LGROUP + 64];
  }
  wp += LRING * LGROUP;

  // ---- phase 0: activation tile -> LDS (fp16 hi / lo, [n][k]); thread = (column quad q, octets o and o + 12)
  {
    float amax = 0.0f;   // range guard of the fp16 split
    const int q = tid & 15, o = tid >> 4;
    const bool ok = full || (n0 + 4 * q < p.N);
    const float* __restrict__ xg = p.X + (long)z * p.sX + (ok ? n0 + 4 * q : 0);
    f32x4 xr[2][8];
#pragma unroll
    for (int oc = 0; oc < 2; ++oc)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 8 * (o + 12 * oc) + e;
        const int kc = (k < p.K && k >= k_lo) ? k : k_lo;    // clamped: branch-free loads, zeroed below
        xr[oc][e] = *reinterpret_cast<const f32x4*>(xg + (long)kc * p.ldx);
      }
#pragma unroll
    for (int oc = 0; oc < 2; ++oc) {
      const int c = o + 12 * oc;
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        f16x8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 8 * c + e;
          const float v = (ok && k < p.K && k >= k_lo) ? xr[oc][e][pp] * LSX : 0.0f;
          amax = __builtin_fmaxf(amax, __builtin_fabsf(v));
          const _Float16 hv = (_Float16)v;
          vh[e] = hv;
          vl[e] = (_Float16)(v - (float)hv);
        }
        const int off = lg_off(4 * q + pp, c);
        *reinterpret_cast<f16x8*>(Xs_hi + off) = vh;
        *reinterpret_cast<f16x8*>(Xs_lo + off) = vl;
      }
    }
    sdy_flag_range(p.flags, amax);
  }
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.0f;

  for (int kb = kb0; kb < LKS / 4; ++kb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ks = 4 * kb + i;
      if (!wave_dead && ks >= ks0) {
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int off = lg_off(32 * j + l31, 2 * ks + h);
          bh[j] = *reinterpret_cast<const f16x8*>(Xs_hi + off);
          bl[j] = *reinterpret_cast<const f16x8*>(Xs_lo + off);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int s = 2 * i + mi;
#pragma unroll
          for (int j = 0; j < 2; ++j) SDY_CROSS_TERM(acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_lo[s], bh[j], acc[mi][j], 0, 0, 0));
#pragma unroll
          for (int j = 0; j < 2; ++j) SDY_CROSS_TERM(acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[s], bl[j], acc[mi][j], 0, 0, 0));
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[s], bh[j], acc[mi][j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int s = 2 * i + mi;
        r_hi[s] = wp[s * LGROUP];
        r_lo[s] = wp[s * LGROUP + 64];
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the refill here (the scheduler otherwise sinks it next to its use)
    }
    wp += LRING * LGROUP;
  }

  // ---- epilogue: accumulators -> LDS [row][col] -> 16-byte row stores of the live rows
  __syncthreads();   // every wave is done reading the x tile
  if (!wave_dead) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 64 * wave + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * h;
          Os[row * LTN + 32 * j + l31] = acc[mi][j][r] * p.out_scale;
        }
  }
  __syncthreads();
  {
    const int q = tid & 15, r0 = tid >> 4;
    const int col = n0 + 4 * q;
    if (full || col < p.N) {
      float* cg = p.C + (long)z * p.sC + col;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = r0 + 12 * i;
        if (row >= row_lo && row < p.M_store)
          *reinterpret_cast<f32x4*>(cg + (long)row * p.ldc) = *reinterpret_cast<const f32x4*>(Os + row * LTN + 4 * q);
      }
    }
  }
}

}  // namespace

size_t sdy_leg_h3_table_bytes(int nz) { return ((size_t)nz * 3 * LGPW + LRING) * LGROUP * sizeof(f16x8); }

int sdy_leg_h3_supported(int rows, int K) { return (rows <= LM && K <= LKS * 16) ? 1 : 0; }

// value(ctx, z, row, k) for row < rows, k < K (zero elsewhere); returns the power-of-two scale applied to the table
int sdy_leg_h3_pack(int nz, int rows, int K, sdy_leg_value_fn value, void* ctx, void* dev, float* scale) {
  if (!sdy_leg_h3_supported(rows, K) || !value || !dev || !scale) return SDY_ERR_ARG;
  float mx = 0.f;
  for (int z = 0; z < nz; ++z)
    for (int r = 0; r < rows; ++r)
      for (int k = 0; k < K; ++k) mx = std::fmax(mx, std::fabs(value(ctx, z, r, k)));
  float s = 1.0f;
  if (mx > 0.f && std::isfinite(mx)) {
    int e;
    std::frexp(mx, &e);
    s = std::ldexp(1.0f, 13 - e);
  }
  const size_t gh = (size_t)LGROUP * 8;
  std::vector<_Float16> buf(((size_t)nz * 3 * LGPW + LRING) * gh, (_Float16)0.0f);
  for (int z = 0; z < nz; ++z)
    for (int w = 0; w < 3; ++w) {
      _Float16* d = buf.data() + (size_t)(z * 3 + w) * LGPW * gh;
      for (int ks = 0; ks < LKS; ++ks)
        for (int mi = 0; mi < 2; ++mi, d += gh)
          for (int ln = 0; ln < 64; ++ln)
            for (int e = 0; e < 8; ++e) {
              const int row = 64 * w + 32 * mi + (ln & 31), k = 16 * ks + 8 * (ln >> 5) + e;
              const float v = (row < rows && k < K) ? value(ctx, z, row, k) * s : 0.0f;
              const _Float16 hv = (_Float16)v;
              d[ln * 8 + e] = hv;
              d[64 * 8 + ln * 8 + e] = (_Float16)(v - (float)hv);
            }
    }
  SDY_HIP_TRY(hipMemcpy(dev, buf.data(), buf.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  *scale = s;
  return SDY_OK;
}

int sdy_leg_h3_launch(const void* table, float scale, int nz, const float* X, long ldx, long sX, float* C, long ldc, long sC,
                      int M_store, int K, int N, int tri, hipStream_t stream) {
  if (!table || !X || !C || nz <= 0 || N <= 0) return SDY_ERR_ARG;
  if (!sdy_leg_h3_supported(M_store, K) || nz > LM) return SDY_ERR_UNSUPPORTED;
  if ((N & 3) || (ldx & 3) || (sX & 3) || (ldc & 3) || (sC & 3)) return SDY_ERR_ALIGN;
  if (tri != SDY_TRI_LEG_FWD && tri != SDY_TRI_LEG_INV) return SDY_ERR_ARG;
  LegParams p;
  p.table = reinterpret_cast<const f16x8*>(table);
  p.X = X; p.ldx = ldx; p.sX = sX;
  p.C = C; p.ldc = ldc; p.sC = sC;
  p.M_store = M_store; p.K = K; p.N = N; p.tri = tri;
  p.out_scale = 1.0f / (scale * LSX);
  SDY_TRY(sdy_flags_ptr(&p.flags));
  dim3 grid((N + LTN - 1) / LTN, nz);
  hipLaunchKernelGGL(leg_h3_kernel, grid, dim3(192), 0, stream, p);
  return sdy_launch_status();
}
