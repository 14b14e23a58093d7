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
constexpr float PSX = SDY_ACT_SX;

struct ParParams {
  const f16x8* table;            // [nz][3 waves][PGPW groups: (k-step, E | O)][hi | lo][64 lanes]
  const float* X; long ldx, sX;  // input rows (latitudes / degrees) at X + z * sX + row * ldx, columns contiguous
  float* C; long ldc, sC;        // output rows at C + z * sC + row * ldc
  long tsx, tsc;                 // != 0: that side is TILE-MAJOR (fft.h, ilv == 2): the workgroup's 64 columns start at
                                 // blockIdx.x * ts instead of at column n0, and its rows are ld = 64 floats apart
  int rows_out, K, N;            // output rows stored, input rows, columns
  const int* kdead;              // polar cut-off per order (rows k' < kdead[m] and their mirrors are skipped) or nullptr
  float out_scale;
  unsigned long long* stamps;    // timing experiments only (SDY_LEG_STAMPS): [4 sampled workgroups][3 waves][8]
};

// same swizzle as leg_h3.hip: 16-byte chunk c (0..23) of column px
__device__ __forceinline__ int pr_off(int px, int c) { return px * PKP + (((c & ~7) | ((c ^ (px >> 1)) & 7)) << 3); }

// 4-byte global store as (wave-uniform base) + (32-bit lane byte offset), see sdy_st16s in common.h
__device__ __forceinline__ void st4s(float* ubase, unsigned off_b, float v) {
  sdy_gptr_t b = (sdy_gptr_t)ubase;
  asm volatile("" : "+s"(b), "+v"(off_b));
  *reinterpret_cast<float __attribute__((address_space(1)))*>(b + off_b) = v;
}

// The kernel is bound by instruction ISSUE, not by HBM latency (tools/leg_stamps.py: with three workgroups per CU every
// phase of a workgroup takes 2-3x its instruction count; two workgroups per CU: 1.3x slower, one: 2.3x).  So: the direction
// is a template parameter, every global access is (uniform row base) + (one lane offset) -- no 64-bit lane arithmetic, no
// per-row clamps --, dead rows are masked by selects on lane masks computed once per thread, and the accumulators go
// straight to global memory as 128-byte row segments (no LDS pass, no barrier after the MFMA loop).
template <bool FWD, bool STAMPS>
__global__ __launch_bounds__(192, 3) void leg_par_kernel(const ParParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PTN * PKP * 2];   // 48 KB
  _Float16* Xs_hi = reinterpret_cast<_Float16*>(smem);
  _Float16* Xs_lo = Xs_hi + PTN * PKP;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // SGPR: row bases of the epilogue are scalar arithmetic
  const int h = lane >> 5, l31 = lane & 31;
  const int z = blockIdx.y;                     // zonal order m
  const int n0 = blockIdx.x * PTN;
  const bool full = n0 + PTN <= p.N;
  const int Kh = (FWD ? p.K : p.rows_out) >> 1;   // latitudes per hemisphere
  const int cE = z & 1, cO = cE ^ 1;            // degree l = 2 r + cE is an "E" degree (l + m even), 2 r + cO an "O" degree

  const int kd = p.kdead ? p.kdead[z] : 0;                // polar rows of this order (table < 1e-12 of its maximum there)
  const bool wave_dead = FWD ? (64 * wave + 63 < z)       // analysis: all degrees of this wave are below m
                             : (32 * wave + 31 < kd);     // synthesis: all latitudes of this wave are polar
  const int ks0 = FWD ? kd >> 4 : z >> 5;                 // k-steps that contribute nothing: polar latitudes / degrees below m
  auto stamp = [&](int i) {
    if (STAMPS && lane == 0 && (blockIdx.x == 37 || blockIdx.x == 150) && (z == 20 || z == 90))
      p.stamps[(((blockIdx.x == 150) * 2 + (z == 90)) * 3 + wave) * 8 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);

  // ---- table ring (slot = 2 * (k-step % 4) + half)
  f16x8 r_hi[PRING], r_lo[PRING];
  // (wave-uniform stream base in SGPRs) + (the lane's running offset): sdy_ring_ld in common.h
  constexpr int PGROUP_BYTES = PGROUP * (int)sizeof(f16x8);
  const char* const wbase = reinterpret_cast<const char*>(p.table) +
                            (size_t)__builtin_amdgcn_readfirstlane(z * 3 + wave) * PGPW * PGROUP_BYTES;
  unsigned woff = (unsigned)lane * 16u;
  if (!wave_dead) {
#pragma unroll
    for (int s = 0; s < PRING; ++s) {
      r_hi[s] = sdy_ring_ld(wbase, woff, 0);
      r_lo[s] = sdy_ring_ld(wbase, woff, PGROUP_BYTES / 2);
      woff += PGROUP_BYTES;
    }
  }

  // ---- phase 0: input rows -> LDS halves; thread = (column quad q, octet o of the half: positions 8 o .. 8 o + 7)
  // (No fp16 range guard here: at 168 registers for three workgroups per CU the tracker spills.  The inputs are the FFT of a
  //  normalised field / the output of the guarded dhconv; an overflow would still surface as non-finite statistics.)
  {
    const int q = tid & 15, o = tid >> 4;
    const bool ok = full || (n0 + 4 * q < p.N);
    const float* xz = p.X + (long)z * p.sX + (p.tsx ? (long)blockIdx.x * p.tsx : (long)n0);   // uniform; column 0 of the tile
    f32x4 xa[8], xb[8];
    bool va[8], vb[8];   // lane masks (one SGPR pair each), computed once and used for the four pixels of the quad
    // analysis : xa = latitude 8 o + e, xb = its mirror K - 1 - (8 o + e) = (K - 8 - 8 o) + (7 - e)
    // synthesis: xa / xb = the "E" / "O" degree of position 8 o + e: 16 o + 2 e + cE / cO
    // Octets that contribute nothing are not loaded at all: polar latitudes (rows the producer never wrote), degrees below
    // m, positions past the end.  Inside a live octet every row is in bounds (K >= 16), so the loads are unconditional and
    // the dead ones (uninitialised memory) are masked by SELECTS, never by a multiplication.
    const bool oct_live = FWD ? (8 * o + 7 >= kd && 8 * o < Kh) : (16 * o + 15 >= z && 16 * o < p.K);
    const bool oct_full = FWD ? true : (16 * o + 15 < p.K);   // synthesis: the last octet may reach past lmax
    const unsigned colc = ok ? 4 * q : 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xa[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      xb[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int r = 8 * o + e;
      if (FWD) {
        va[e] = vb[e] = ok && r < Kh && r >= kd;
      } else {
        va[e] = ok && 2 * r + cE < p.K && 2 * r + cE >= z;   // degrees below m were never written
        vb[e] = ok && 2 * r + cO < p.K && 2 * r + cO >= z;
      }
    }
    if (oct_live) {
      if (FWD) {
        const unsigned offA = (unsigned)((long)(8 * o) * p.ldx + colc) * 4u;
        const unsigned offB = (unsigned)((long)(p.K - 8 - 8 * o) * p.ldx + colc) * 4u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xa[e] = sdy_ld16s(xz + (long)e * p.ldx, offA);
          xb[e] = sdy_ld16s(xz + (long)(7 - e) * p.ldx, offB);
        }
      } else {
        const unsigned off = (unsigned)((long)(16 * o) * p.ldx + colc) * 4u;
        if (oct_full) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xa[e] = sdy_ld16s(xz + (long)(2 * e + cE) * p.ldx, off);
            xb[e] = sdy_ld16s(xz + (long)(2 * e + cO) * p.ldx, off);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (16 * o + 2 * e + cE < p.K) xa[e] = sdy_ld16s(xz + (long)(2 * e + cE) * p.ldx, off);
            if (16 * o + 2 * e + cO < p.K) xb[e] = sdy_ld16s(xz + (long)(2 * e + cO) * p.ldx, off);
          }
        }
      }
    }
    stamp(1);   // loads issued
    if (STAMPS) { asm volatile("s_waitcnt vmcnt(0)"); stamp(2); }   // loads arrived
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        f16x8 vh, vl;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (FWD) {
            const float a = xa[e][pp], b = xb[e][pp];
            v[e] = va[e] ? (hf == 0 ? a + b : a - b) * PSX : 0.0f;
          } else {
            v[e] = (hf == 0 ? va[e] : vb[e]) ? (hf == 0 ? xa[e][pp] : xb[e][pp]) * PSX : 0.0f;
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
  stamp(3);   // tile in LDS
  if (wave_dead) return;   // (after the only barrier)

  f32x16 acc[2][2];   // [half: E, O][column tile]
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[hf][j][r] = 0.0f;

#pragma unroll
  for (int ks = 0; ks < PKS; ++ks) {
    if (ks >= ks0) {
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
        for (int j = 0; j < 2; ++j) SDY_CROSS_TERM(acc[hf][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_lo[s], bh[j], acc[hf][j], 0, 0, 0));
#pragma unroll
        for (int j = 0; j < 2; ++j) SDY_CROSS_TERM(acc[hf][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[s], bl[j], acc[hf][j], 0, 0, 0));
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[hf][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[s], bh[j], acc[hf][j], 0, 0, 0);
      }
    }
    if (ks + 4 < PKS) {   // the ring holds 4 k-steps: refill the two slots just used with k-step ks + 4
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int s = 2 * (ks & 3) + hf;
        r_hi[s] = sdy_ring_ld(wbase, woff, 0);     // group PRING + s: the stream front to back
        r_lo[s] = sdy_ring_ld(wbase, woff, PGROUP_BYTES / 2);
        woff += PGROUP_BYTES;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  stamp(4);   // MFMA loop done

  // ---- epilogue: accumulators -> global.  Register r of a tile is row 8 (r >> 2) + 4 h + (r & 3) of the wave's 32, lanes
  // l31 are 32 consecutive columns: one store instruction writes two 128-byte row segments.  Row = uniform part (SGPR
  // arithmetic on the wave index, running pointers) + a lane part that depends on h only.  Groups of four registers
  // (r >> 2) that lie inside the stored range take plain stores; a group that touches the edge (rows below m, the last
  // degrees / latitudes, a ragged column tile) is predicated with lane masks built from scalar compares.
  float* cz = p.C + (long)z * p.sC + (p.tsc ? (long)blockIdx.x * p.tsc : (long)n0);   // uniform
  constexpr unsigned long long MH0 = 0x00000000FFFFFFFFull, MH1 = 0xFFFFFFFF00000000ull;   // lanes with h = 0 / h = 1
  unsigned long long colm[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) colm[j] = __builtin_amdgcn_ballot_w64(n0 + 32 * j + l31 < p.N);
  if (FWD) {
    // degree l = 2 rt + c = [64 wave + 16 r4 + 2 r2 + c] + 8 h, c = cE for the E tile, cO for the O tile
    const unsigned offl = (unsigned)((long)(8 * h) * p.ldc + l31) * 4u;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int base = 64 * wave + 16 * r4 + (hf == 0 ? cE : cO);   // uniform; the group holds degrees base .. base + 14
        if (base + 14 < z || base >= p.rows_out) continue;            // uniform: all below m / past lmax
        float* rowp = cz + (long)base * p.ldc;
        if (full && base >= z && base + 14 < p.rows_out) {
#pragma unroll
          for (int r2 = 0; r2 < 4; ++r2, rowp += 2 * p.ldc)
#pragma unroll
            for (int j = 0; j < 2; ++j) st4s(rowp + 32 * j, offl, acc[hf][j][4 * r4 + r2] * p.out_scale);
        } else {
#pragma unroll
          for (int r2 = 0; r2 < 4; ++r2, rowp += 2 * p.ldc) {
            const int lu = base + 2 * r2;
            const unsigned long long m = ((lu >= z && lu < p.rows_out) ? MH0 : 0ull) | ((lu + 8 >= z && lu + 8 < p.rows_out) ? MH1 : 0ull);
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (__builtin_amdgcn_inverse_ballot_w64(m & colm[j])) st4s(rowp + 32 * j, offl, acc[hf][j][4 * r4 + r2] * p.out_scale);
          }
        }
      }
  } else {
    // latitude rt = [32 wave + 8 r4 + r2] + 4 h and its mirror rows_out - 1 - rt = [rows_out - 5 - (32 wave + 8 r4 + r2)] + 4 (1 - h)
    const unsigned off1 = (unsigned)((long)(4 * h) * p.ldc + l31) * 4u;
    const unsigned off2 = (unsigned)((long)(4 * (1 - h)) * p.ldc + l31) * 4u;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int base = 32 * wave + 8 * r4;                 // uniform; the group holds latitudes base .. base + 7
      if (base + 7 < kd || base >= Kh) continue;           // uniform: polar rows / past the equator
      float* row1 = cz + (long)base * p.ldc;
      float* row2 = cz + (long)(p.rows_out - 5 - base) * p.ldc;
      if (full && base >= kd && base + 7 < Kh) {
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2, row1 += p.ldc, row2 -= p.ldc)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float e = acc[0][j][4 * r4 + r2], od = acc[1][j][4 * r4 + r2];
            st4s(row1 + 32 * j, off1, (e + od) * p.out_scale);
            st4s(row2 + 32 * j, off2, (e - od) * p.out_scale);
          }
      } else {
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2, row1 += p.ldc, row2 -= p.ldc) {
          const int ru = base + r2;
          const unsigned long long m = ((ru >= kd && ru < Kh) ? MH0 : 0ull) | ((ru + 4 >= kd && ru + 4 < Kh) ? MH1 : 0ull);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (__builtin_amdgcn_inverse_ballot_w64(m & colm[j])) {
              const float e = acc[0][j][4 * r4 + r2], od = acc[1][j][4 * r4 + r2];
              st4s(row1 + 32 * j, off1, (e + od) * p.out_scale);
              st4s(row2 + 32 * j, off2, (e - od) * p.out_scale);
            }
        }
      }
    }
  }
  stamp(5);
}

}  // namespace

#if SDY_STAMPS_ON
static unsigned long long* g_lstamps = nullptr;
SDY_DEBUG_EXPORT int sdy_leg_par_debug_stamps(unsigned long long* host96) {
  if (!g_lstamps || !host96) return SDY_ERR_STATE;
  SDY_HIP_TRY(hipDeviceSynchronize());
  SDY_HIP_TRY(hipMemcpy(host96, g_lstamps, 96 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SDY_OK;
}
#endif

size_t sdy_leg_par_table_bytes(int nz) { return ((size_t)nz * 3 * PGPW + PRING) * PGROUP * sizeof(f16x8); }

// rows_out x K problem of leg_h3 (analysis: lmax x nlat, synthesis: nlat x lmax); the folded axis (nlat) must be even
// (nlat, lmax >= 16: inside a live octet of the load phase every row is then in bounds, see the kernel)
int sdy_leg_par_supported(int nlat, int lmax) {
  return (nlat <= PM && lmax <= PM && (nlat & 1) == 0 && nlat >= 16 && lmax >= 16) ? 1 : 0;
}

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
                       long sC, int rows_out, int K, int N, int fwd, const int* kdead, hipStream_t stream, long tsx, long tsc) {
  if (!table || !X || !C || nz <= 0 || N <= 0) return SDY_ERR_ARG;
  if ((tsx || tsc) && N % PTN) return SDY_ERR_UNSUPPORTED;   // tile-major sides hold whole 64-column tiles
  if (rows_out > PM || K > PM || nz > PM) return SDY_ERR_UNSUPPORTED;
  if ((fwd ? K : rows_out) & 1) return SDY_ERR_UNSUPPORTED;
  if ((N & 3) || (ldx & 3) || (sX & 3) || (ldc & 3) || (sC & 3)) return SDY_ERR_ALIGN;
  ParParams p;
  p.table = reinterpret_cast<const f16x8*>(table);
  p.X = X; p.ldx = ldx; p.sX = sX;
  p.C = C; p.ldc = ldc; p.sC = sC;
  p.tsx = tsx; p.tsc = tsc;
  // 32-bit lane offsets: 96 input rows / 8 output rows of lane-dependent distance
  if ((long)PM * ldx * 4 + (long)N * 4 >= (1L << 32) || (long)8 * ldc * 4 + (long)N * 4 >= (1L << 32)) return SDY_ERR_UNSUPPORTED;
  p.rows_out = rows_out; p.K = K; p.N = N; p.kdead = kdead;
  p.out_scale = 1.0f / (scale * PSX);
  p.stamps = nullptr;
  dim3 grid((N + PTN - 1) / PTN, nz);
#if SDY_STAMPS_ON
  if (std::getenv("SDY_LEG_STAMPS")) {
    if (!g_lstamps) SDY_HIP_TRY(hipMalloc(&g_lstamps, 96 * sizeof(unsigned long long)));
    p.stamps = g_lstamps;
  }
  if (p.stamps) {
    if (fwd) hipLaunchKernelGGL((leg_par_kernel<true, true>), grid, dim3(192), 0, stream, p);
    else hipLaunchKernelGGL((leg_par_kernel<false, true>), grid, dim3(192), 0, stream, p);
    return sdy_launch_status();
  }
#endif
  if (fwd) hipLaunchKernelGGL((leg_par_kernel<true, false>), grid, dim3(192), 0, stream, p);
  else hipLaunchKernelGGL((leg_par_kernel<false, false>), grid, dim3(192), 0, stream, p);
  return sdy_launch_status();
}
