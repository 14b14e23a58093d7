// Legendre transforms with the equatorial symmetry folded in (even nlat <= 192, lmax <= 192), gfx950.
//
// Same contract as leg_h3.hip (torch_harmonics' Legendre contractions behind RealSHT / InverseRealSHT; reference call
// sites src/models/sfno/s2convolutions.py:165,186).  Both grids are symmetric about the equator and
//     P_l^m(-x) = (-1)^(l+m) P_l^m(x),
// so with k' < nlat/2 and its mirror nlat-1-k':
//   analysis :  c[l] = sum_k' T[l][k'] (x[k'] + s x[mirror]),      s = +1 if l + m is even ("E" rows), -1 if odd ("O" rows)
//   synthesis:  x[k'] = E[k'] + O[k'],  x[mirror] = E[k'] - O[k'],  E / O = sum over the l of one parity of P[k'][l] c[l]
// i.e. every output needs HALF the products of the plain GEMM, and the table is half as long.  (The reference multiplies
// the full matrices; the results agree to rounding.)
//
// Layout of the work (otherwise the leg_h3 recipe: activation tile in LDS, table streamed L2 -> registers as MFMA
// fragments, result through LDS for 16-byte stores): the LDS tile is [64 columns][E half: 96 k | O half: 96 k];
//   analysis : the halves hold x[k'] + x[mirror] and x[k'] - x[mirror]; wave w owns the 64 degrees l = 64w .. 64w+63 as one
//              tile of 32 "E" degrees and one of 32 "O" degrees;
//   synthesis: the halves hold the coefficients of the two parities (position l >> 1); wave w owns latitudes
//              k' = 32w .. 32w+31 and accumulates E and O side by side, so x[k'] and x[mirror] are formed in registers.
// Each of the 6 k-steps issues 6 MFMAs on the E half and 6 on the O half (12 k-steps x 12 MFMAs before).
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int PM = 192;          // output rows a workgroup covers
constexpr int PKP = 192;         // LDS row length in halfs: E half | O half, 96 positions each
constexpr int PTN = 64;          // columns per workgroup
constexpr int PKS = 6;           // k-steps of 16 per half
constexpr int PRING = 8;         // groups in flight = 4 k-steps x (E, O)
constexpr int PGPW = 2 * PKS;    // groups per (m, wave)
constexpr int PGROUP = 2 * 64;   // f16x8 elements per group
constexpr float PSX = 16.0f;

struct ParParams {
  const f16x8* table;            // [nz][3 waves][PGPW groups: (k-step, E | O)][hi | lo][64 lanes]
  const float* X; long ldx, sX;  // input rows (latitudes / degrees) at X + z * sX + row * ldx, columns contiguous
  float* C; long ldc, sC;        // output rows at C + z * sC + row * ldc
  int rows_out, K, N;            // output rows stored, input rows, columns
  int fwd;                       // 1 analysis, 0 synthesis
  const int* kdead;              // polar cut-off per order (rows k' < kdead[m] and their mirrors are skipped) or nullptr
  float out_scale;
  unsigned* flags;               // sticky status word (sdy_status_flags)
};

// same swizzle as leg_h3.hip: 16-byte chunk c (0..23) of column px
__device__ __forceinline__ int pr_off(int px, int c) { return px * PKP + (((c & ~7) | ((c ^ (px >> 1)) & 7)) << 3); }

__global__ __launch_bounds__(192, 3) void leg_par_kernel(const ParParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PTN * PKP * 2];   // 48 KB
  _Float16* Xs_hi = reinterpret_cast<_Float16*>(smem);
  _Float16* Xs_lo = Xs_hi + PTN * PKP;
  float* Os = reinterpret_cast<float*>(smem);   // epilogue: [192 rows][64 cols] fp32 (aliases the tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int z = blockIdx.y;                     // zonal order m
  const int n0 = blockIdx.x * PTN;
  const bool full = n0 + PTN <= p.N;
  const int Kh = p.K >> 1;                      // analysis: latitudes per hemisphere
  const int cE = z & 1, cO = cE ^ 1;            // degree l = 2 r + cE is an "E" degree (l + m even), 2 r + cO an "O" degree

  const int kd = p.kdead ? p.kdead[z] : 0;                // polar rows of this order (table < 1e-12 of its maximum there)
  const bool wave_dead = p.fwd ? (64 * wave + 63 < z)     // analysis: all degrees of this wave are below m
                               : (32 * wave + 31 < kd);   // synthesis: all latitudes of this wave are polar
  const int ks0 = p.fwd ? kd >> 4 : z >> 5;               // k-steps that contribute nothing: polar latitudes / degrees below m
  const int row_lo = p.fwd ? z : 0;                       // analysis: degrees l < m are never read downstream

  // ---- table ring (slot = 2 * (k-step % 4) + half)
  f16x8 r_hi[PRING], r_lo[PRING];
  const f16x8* __restrict__ wp = p.table + (size_t)(z * 3 + wave) * PGPW * PGROUP + lane;
#pragma unroll
  for (int s = 0; s < PRING; ++s) {
    r_hi[s] = wp[s * PGROUP];
    r_lo[s] = wp[s * PGROUP + 64];
  }

  // ---- phase 0: input rows -> LDS halves; thread = (column quad q, octet o of the half: positions 8 o .. 8 o + 7)
  // (No fp16 range guard here: at 168 registers for three workgroups per CU the tracker spills.  The inputs are the FFT of a
  //  normalised field / the output of the guarded dhconv; an overflow would still surface as non-finite statistics.)
  {
    const int q = tid & 15, o = tid >> 4;
    const bool ok = full || (n0 + 4 * q < p.N);
    const float* __restrict__ xg = p.X + (long)z * p.sX + (ok ? n0 + 4 * q : 0);
    f32x4 xr[2][8];
    bool v_ok[2][8];
    // synthesis: an octet whose 16 degrees are all below m is zero -- no loads at all (half of the octets on average)
    const bool oct_live = p.fwd ? (8 * o + 7 >= kd) : (16 * o + 15 >= z);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xr[0][e] = f32x4{0.f, 0.f, 0.f, 0.f};
      xr[1][e] = f32x4{0.f, 0.f, 0.f, 0.f};
      v_ok[0][e] = v_ok[1][e] = false;
    }
    if (oct_live)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * o + e;
      int rowA, rowB;
      if (p.fwd) {   // latitude k' and its mirror
        rowA = r; rowB = p.K - 1 - r;
        v_ok[0][e] = v_ok[1][e] = r < Kh && r >= kd;
      } else {       // the "E" and the "O" degree of position r
        rowA = 2 * r + cE; rowB = 2 * r + cO;
        v_ok[0][e] = rowA < p.K && rowA >= z;      // degrees below m were never written
        v_ok[1][e] = rowB < p.K && rowB >= z;
      }
      const int safe = p.fwd ? 0 : z;              // clamped: branch-free loads, zeroed below
      xr[0][e] = *reinterpret_cast<const f32x4*>(xg + (long)(v_ok[0][e] ? rowA : safe) * p.ldx);
      xr[1][e] = *reinterpret_cast<const f32x4*>(xg + (long)(v_ok[1][e] ? rowB : safe) * p.ldx);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        f16x8 vh, vl;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (p.fwd) {
            const float a = xr[0][e][pp], b = xr[1][e][pp];
            v[e] = (ok && v_ok[0][e]) ? (hf == 0 ? a + b : a - b) * PSX : 0.0f;
          } else {
            v[e] = (ok && v_ok[hf][e]) ? xr[hf][e][pp] * PSX : 0.0f;
          }
        }
        sdy_split8(v, vh, vl);
        const int off = pr_off(4 * q + pp, o + 12 * hf);
        *reinterpret_cast<f16x8*>(Xs_hi + off) = vh;
        *reinterpret_cast<f16x8*>(Xs_lo + off) = vl;
      }
    }
  }
  __syncthreads();

  f32x16 acc[2][2];   // [half: E, O][column tile]
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[hf][j][r] = 0.0f;

#pragma unroll
  for (int ks = 0; ks < PKS; ++ks) {
    if (!wave_dead && ks >= ks0) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int off = pr_off(32 * j + l31, 12 * hf + 2 * ks + h);
          bh[j] = *reinterpret_cast<const f16x8*>(Xs_hi + off);
          bl[j] = *reinterpret_cast<const f16x8*>(Xs_lo + off);
        }
        const int s = 2 * (ks & 3) + hf;
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[hf][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_lo[s], bh[j], acc[hf][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[hf][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[s], bl[j], acc[hf][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[hf][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[s], bh[j], acc[hf][j], 0, 0, 0);
      }
    }
    if (ks + 4 < PKS) {   // the ring holds 4 k-steps: refill the two slots just used with k-step ks + 4
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int s = 2 * (ks & 3) + hf;
        r_hi[s] = wp[(PRING + s) * PGROUP];
        r_lo[s] = wp[(PRING + s) * PGROUP + 64];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: accumulators -> LDS [output row][col] -> 16-byte row stores of the live rows
  __syncthreads();   // every wave is done reading the tile
  if (!wave_dead) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rt = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;   // row inside the wave's tile pair
        const int col = 32 * j + l31;
        if (p.fwd) {          // degrees 2 rt + cE ("E" tile) and 2 rt + cO ("O" tile)
          Os[(2 * rt + cE) * PTN + col] = acc[0][j][r] * p.out_scale;
          Os[(2 * rt + cO) * PTN + col] = acc[1][j][r] * p.out_scale;
        } else if (rt < (p.rows_out >> 1)) {   // latitude rt and its mirror
          Os[rt * PTN + col] = (acc[0][j][r] + acc[1][j][r]) * p.out_scale;
          Os[(p.rows_out - 1 - rt) * PTN + col] = (acc[0][j][r] - acc[1][j][r]) * p.out_scale;
        }
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
        if (row >= row_lo && row < p.rows_out && (p.fwd || (row >= kd && row < p.rows_out - kd)))
          SDY_STREAM_STORE(cg + (long)row * p.ldc, *reinterpret_cast<const f32x4*>(Os + row * PTN + 4 * q));
      }
    }
  }
}

}  // namespace

size_t sdy_leg_par_table_bytes(int nz) { return ((size_t)nz * 3 * PGPW + PRING) * PGROUP * sizeof(f16x8); }

// rows_out x K problem of leg_h3 (analysis: lmax x nlat, synthesis: nlat x lmax); the folded axis (nlat) must be even
int sdy_leg_par_supported(int nlat, int lmax) { return (nlat <= PM && lmax <= PM && (nlat & 1) == 0 && nlat >= 2) ? 1 : 0; }

// value(ctx, z, row, k): the UNFOLDED table of leg_h3 (analysis: row = l, k = latitude; synthesis: row = latitude, k = l).
// Only latitudes of the first hemisphere are read.
int sdy_leg_par_pack(int nz, int nlat, int lmax, int fwd, sdy_leg_value_fn value, void* ctx, void* dev, float* scale) {
  if (!sdy_leg_par_supported(nlat, lmax) || !value || !dev || !scale) return SDY_ERR_ARG;
  const int Kh = nlat / 2;
  float mx = 0.f;
  for (int z = 0; z < nz; ++z)
    for (int l = 0; l < lmax; ++l)
      for (int k = 0; k < Kh; ++k) mx = std::fmax(mx, std::fabs(fwd ? value(ctx, z, l, k) : value(ctx, z, k, l)));
  float s = 1.0f;
  if (mx > 0.f && std::isfinite(mx)) {
    int e;
    std::frexp(mx, &e);
    s = std::ldexp(1.0f, 13 - e);
  }
  const size_t gh = (size_t)PGROUP * 8;
  std::vector<_Float16> buf(((size_t)nz * 3 * PGPW + PRING) * gh, (_Float16)0.0f);
  for (int z = 0; z < nz; ++z) {
    const int cE = z & 1;
    for (int w = 0; w < 3; ++w) {
      _Float16* d = buf.data() + (size_t)(z * 3 + w) * PGPW * gh;
      for (int ks = 0; ks < PKS; ++ks)
        for (int hf = 0; hf < 2; ++hf, d += gh) {
          const int cls = hf == 0 ? cE : cE ^ 1;
          for (int ln = 0; ln < 64; ++ln)
            for (int e = 0; e < 8; ++e) {
              const int rt = 32 * w + (ln & 31), kk = 16 * ks + 8 * (ln >> 5) + e;
              float v = 0.0f;
              if (fwd) {   // row: degree 2 rt + cls, contraction: latitude kk of the first hemisphere
                const int l = 2 * rt + cls;
                if (l < lmax && kk < Kh) v = value(ctx, z, l, kk) * s;
              } else {     // row: latitude rt, contraction: degree 2 kk + cls
                const int l = 2 * kk + cls;
                if (rt < Kh && l < lmax) v = value(ctx, z, rt, l) * s;
              }
              const _Float16 hv = (_Float16)v;
              d[ln * 8 + e] = hv;
              d[64 * 8 + ln * 8 + e] = (_Float16)(v - (float)hv);
            }
        }
    }
  }
  SDY_HIP_TRY(hipMemcpy(dev, buf.data(), buf.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  *scale = s;
  return SDY_OK;
}

int sdy_leg_par_launch(const void* table, float scale, int nz, const float* X, long ldx, long sX, float* C, long ldc,
                       long sC, int rows_out, int K, int N, int fwd, const int* kdead, hipStream_t stream) {
  if (!table || !X || !C || nz <= 0 || N <= 0) return SDY_ERR_ARG;
  if (rows_out > PM || K > PM || nz > PM) return SDY_ERR_UNSUPPORTED;
  if ((fwd ? K : rows_out) & 1) return SDY_ERR_UNSUPPORTED;
  if ((N & 3) || (ldx & 3) || (sX & 3) || (ldc & 3) || (sC & 3)) return SDY_ERR_ALIGN;
  ParParams p;
  p.table = reinterpret_cast<const f16x8*>(table);
  p.X = X; p.ldx = ldx; p.sX = sX;
  p.C = C; p.ldc = ldc; p.sC = sC;
  p.rows_out = rows_out; p.K = K; p.N = N; p.fwd = fwd; p.kdead = kdead;
  p.out_scale = 1.0f / (scale * PSX);
  SDY_TRY(sdy_flags_ptr(&p.flags));
  dim3 grid((N + PTN - 1) / PTN, nz);
  hipLaunchKernelGGL(leg_par_kernel, grid, dim3(192), 0, stream, p);
  return sdy_launch_status();
}
