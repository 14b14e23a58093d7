// Longitude real FFT / inverse real FFT specialised for nlon = 360 (the production grid), gfx950.
//
// Same contract as the generic kernels in fft.hip (torch_harmonics' rfft / irfft calls, reference call sites
// src/models/sfno/s2convolutions.py:165,168,186, plus the NCHW <-> m-major layout change):
//   forward : Xf[m][k][b][ri][c] = 2*pi/N * rfft(a*x+d)[m]        (m < mtr)
//   inverse : y[b][c][k][:]      = irfft(Yf[..], n=N) (+ bias[c])  (imag of the m=0 / Nyquist bins ignored)
//
// The generic Stockham kernel is instruction bound (PMC: ~900 VALU + 256 LDS instructions per wave and ring, nearly all
// of it index arithmetic and 4-byte LDS accesses, 7 workgroup barriers per ring).  This kernel does the length-180
// complex transform as TWO register-resident passes, 180 = 12 x 15:
//   pass A: lane (row, a), a < 15: 12-point DFT over x[a + 15 b] (Good-Thomas 3 x 4, no inner twiddles), times
//           w180^(a k1), written back in place;
//   pass B: lane (row, k1), k1 < 12: 15-point DFT over the 15 consecutive values (Good-Thomas 3 x 5), written in natural
//           order X[k1 + 12 k2].
// Every LDS access is an 8-byte (re, im) word at "lane base + compile-time offset", complex arithmetic is written on
// float2 so it maps to v_pk_*_f32, and the 11 pass-A twiddles live in registers for the life of the workgroup.  A wave owns
// 4 of the workgroup's 16 rows for both passes, so no workgroup barrier sits between them: the only barriers are the two
// around the transposing pass (the m-major side is written / read as 64-byte channel runs).  The real-FFT split (forward)
// is folded into the m-major store pass and the merge (inverse) into the loads of pass A.
//
// LDS: 16 row slots x P = 200 complex words.  Channel 4*c4 + j of the workgroup sits in slot c4 + 4*j; wave w owns slots
// 4w..4w+3 and pairs slots {0,2} / {1,3} in its two half-waves.  With P % 32 == 8 (in 8-byte words) every access pattern
// below is bank-conflict free: pass A/B (two rows 16 words apart, lanes contiguous or at stride 15), the m-major pass
// (8 consecutive m x 4 slots 8 words apart) and the 16-byte row I/O.
#include "common.h"
#include "fft.h"

namespace {

typedef float c2 __attribute__((ext_vector_type(2)));   // (re, im)

constexpr int NH = 180;      // complex length
constexpr int NLON = 360;
constexpr int ROWS = 16;     // channels per workgroup
constexpr int NT = 256;
constexpr int P = 200;       // row pitch in complex words (>= 181, % 32 == 8)
constexpr int Q4 = NLON / 4; // 16-byte pieces per row
constexpr int MIT = 3;                    // m-major iterations: m = mg + 64*it covers 0..191
#ifndef SDY_FFT_MINB_F
#define SDY_FFT_MINB_F 4
#endif
#ifndef SDY_FFT_MINB_I
#define SDY_FFT_MINB_I 4
#endif
// latitude rings per workgroup (the ring loop prefetches ring k + 1 under ring k's passes when > 1).  Measured end to end:
// 4 rings 130.9, 2 rings 131.6 / 139.2, 1 ring 139.9 member-forecast-steps/s -- more, smaller workgroups balance better than
// the prefetch helps (4 workgroups per CU already overlap each other's loads).
#ifndef SDY_FFT_KPW
#define SDY_FFT_KPW 1
#endif
constexpr int MINB_F = SDY_FFT_MINB_F, MINB_I = SDY_FFT_MINB_I;   // workgroups per CU the register budgets are set for

template <int SG>
__device__ __forceinline__ c2 mul_i(c2 a) {   // (SG * i) * a
  return SG > 0 ? c2{-a.y, a.x} : c2{a.y, -a.x};
}
__device__ __forceinline__ c2 cmul(c2 a, c2 w) { return a.xx * w + a.yy * c2{-w.y, w.x}; }
__device__ __forceinline__ c2 cmul_conj(c2 a, c2 w) { return a.xx * c2{w.x, -w.y} + a.yy * c2{w.y, w.x}; }   // a * conj(w)

// SG = -1: forward (exp(-2 pi i ..)), +1: inverse
template <int SG>
__device__ __forceinline__ void dft3(c2& v0, c2& v1, c2& v2) {
  const float sn = 0.86602540378443864676f;
  const c2 a = v1 + v2, b = v1 - v2;
  const c2 m = v0 - 0.5f * a;
  const c2 t = mul_i<SG>(sn * b);
  v0 = v0 + a;
  v1 = m + t;
  v2 = m - t;
}
template <int SG>
__device__ __forceinline__ void dft4(c2& v0, c2& v1, c2& v2, c2& v3) {
  const c2 t0 = v0 + v2, t1 = v0 - v2, t2 = v1 + v3, t3 = mul_i<SG>(v1 - v3);
  v0 = t0 + t2;
  v1 = t1 + t3;
  v2 = t0 - t2;
  v3 = t1 - t3;
}
template <int SG>
__device__ __forceinline__ void dft5(c2& v0, c2& v1, c2& v2, c2& v3, c2& v4) {
  const float c1 = 0.30901699437494742410f, c2_ = -0.80901699437494742410f;
  const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  const c2 a1 = v1 + v4, a2 = v2 + v3, b1 = v1 - v4, b2 = v2 - v3;
  const c2 r1 = v0 + c1 * a1 + c2_ * a2;
  const c2 r2 = v0 + c2_ * a1 + c1 * a2;
  const c2 i1 = mul_i<SG>(s1 * b1 + s2 * b2);
  const c2 i2 = mul_i<SG>(s2 * b1 - s1 * b2);
  v0 = v0 + a1 + a2;
  v1 = r1 + i1;
  v4 = r1 - i1;
  v2 = r2 + i2;
  v3 = r2 - i2;
}
template <int R, int SG>
__device__ __forceinline__ void dft_small(c2* t) {
  if constexpr (R == 3) dft3<SG>(t[0], t[1], t[2]);
  if constexpr (R == 4) dft4<SG>(t[0], t[1], t[2], t[3]);
  if constexpr (R == 5) dft5<SG>(t[0], t[1], t[2], t[3], t[4]);
}

// Good-Thomas DFT of length N1*N2 (coprime) on registers: input n lives in v[n]; afterwards output k lives in
// v[pfa_slot<N1,N2>(k)].  All indices are compile-time after unrolling, so the permutations cost nothing.
template <int N1, int N2>
__host__ __device__ constexpr int pfa_slot(int k) { return (N2 * (k % N1) + N1 * (k % N2)) % (N1 * N2); }
template <int N1, int N2, int SG>
__device__ __forceinline__ void pfa(c2* v) {
  constexpr int N = N1 * N2;
#pragma unroll
  for (int n2 = 0; n2 < N2; ++n2) {
    c2 t[N1];
#pragma unroll
    for (int n1 = 0; n1 < N1; ++n1) t[n1] = v[(N2 * n1 + N1 * n2) % N];
    dft_small<N1, SG>(t);
#pragma unroll
    for (int n1 = 0; n1 < N1; ++n1) v[(N2 * n1 + N1 * n2) % N] = t[n1];
  }
#pragma unroll
  for (int k1 = 0; k1 < N1; ++k1) {
    c2 u[N2];
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) u[n2] = v[(N2 * k1 + N1 * n2) % N];
    dft_small<N2, SG>(u);
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) v[(N2 * k1 + N1 * n2) % N] = u[n2];
  }
}

// lane roles shared by both kernels
struct Lane {
  int wave, lane, t;       // wave is wave-uniform (SGPR); t = lane & 15: the a / k1 index of the two passes
  c2* row;                 // LDS row of the two passes
};
__device__ __forceinline__ Lane lane_roles(c2* Z) {
  Lane L;
  L.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  L.lane = threadIdx.x & 63;
  L.t = L.lane & 15;
  const int rl = L.lane >> 4;
  L.row = Z + (4 * L.wave + (rl & 1) * 2 + (rl >> 1)) * P;
  return L;
}

// pass-A twiddles w180^(a k1) as an LDS table [k1][16]: one base register per lane, compile-time offsets per k1
constexpr int TWN = 12 * 16;
__device__ __forceinline__ void fill_twiddles(const SdyFftDesc& f, c2* TW) {
  if (threadIdx.x < TWN) {
    const int k1 = threadIdx.x >> 4, a = threadIdx.x & 15;
    TW[threadIdx.x] = reinterpret_cast<const c2*>(f.tw)[a < 15 ? a * k1 : 0];   // a*k1 <= 154 < 180
  }
}

// passes A and B on the wave's own rows (in-order LDS of one wave: no workgroup barrier between them)
template <int SG, bool NYQ_COPY>
__device__ __forceinline__ void pass_b(const Lane& L) {
  __builtin_amdgcn_wave_barrier();
  if (L.t < 12) {
    c2 u[15];
    const c2* src = L.row + 15 * L.t;
#pragma unroll
    for (int a = 0; a < 15; ++a) u[a] = src[a];
    pfa<3, 5, SG>(u);
    __builtin_amdgcn_wave_barrier();   // every lane's reads above precede the natural-order writes below
    c2* dst = L.row + L.t;
#pragma unroll
    for (int k2 = 0; k2 < 15; ++k2) dst[12 * k2] = u[pfa_slot<3, 5>(k2)];
    if (NYQ_COPY && L.t == 0) dst[NH] = u[pfa_slot<3, 5>(0)];   // Z[180] := Z[0] (periodic), read by the split step
  }
  __builtin_amdgcn_wave_barrier();
}
template <int SG>
__device__ __forceinline__ void pass_a_tail(const Lane& L, c2* v, const c2* TW) {   // v[b] = x[a + 15 b] already loaded
  pfa<3, 4, SG>(v);
  c2* dst = L.row + L.t;
  const c2* tw = TW + L.t;
  dst[0] = v[pfa_slot<3, 4>(0)];
#pragma unroll
  for (int k1 = 1; k1 < 12; ++k1) {
    const c2 y = v[pfa_slot<3, 4>(k1)];
    dst[15 * k1] = SG < 0 ? cmul(y, tw[16 * k1]) : cmul_conj(y, tw[16 * k1]);
  }
}

// Row I/O of a wave's own 4 rows (slot 4w + r holds channel 4r + w): 90 16-byte pieces per row = all 64 lanes + lanes 0..25,
// so every address is "row base (uniform) + 16 * lane (+ 1024)" and no per-piece index arithmetic is left in the ring loop.
constexpr int TAIL = Q4 - 64;   // 26

// Workgroup -> (channel block, ring group, image).  The hardware deals consecutive workgroup ids round-robin over the 8
// XCDs, each with its own L2.  The 16 channel blocks of one (ring group, image) write / read the two 64-byte halves of the
// same 128-byte lines on the m-major side, so they are mapped to consecutive slots of ONE XCD: its L2 merges the halves
// instead of every half-line going to HBM on its own.
// Order of the workgroups an XCD runs at one time (160 of them): SDY_FFT_GROUP channel blocks fastest, then the ring, then the
// remaining channel blocks, then the image.  Measured on one device (profiles/r4c/e2e_ab_fft_*.txt, inverse / forward ms per
// launch): all 16 blocks of a ring first (10 rings in flight, rounds 1-3) 0.613 / 0.567; 8: the same; 4 (40 rings of two
// tiles): 0.605 / 0.564; 2 (80 rings of one tile): 0.641 / 0.589; the images of one ring first: 0.72 / 0.63.
#ifndef SDY_FFT_GROUP
#define SDY_FFT_GROUP 4
#endif
struct WgId {
  int x, y, z;
};
__device__ __forceinline__ WgId wg_id(int nx, int ny) {
  const unsigned total = gridDim.x, L = blockIdx.x;
  const unsigned xcd = L % 8u, slot = L / 8u;
  const unsigned id = xcd * (total / 8u) + min(xcd, total % 8u) + slot;
  WgId w;
#if SDY_FFT_GROUP > 0
  if (nx % SDY_FFT_GROUP == 0) {   // SDY_FFT_GROUP channel blocks fastest, then the ring, then the rest of the channel blocks
    const unsigned G = SDY_FFT_GROUP;
    w.x = (int)(G * ((id / (G * ny)) % ((unsigned)nx / G)) + id % G);
    w.y = (int)((id / G) % (unsigned)ny);
    w.z = (int)(id / ((unsigned)nx * ny));
    return w;
  }
#endif
  w.x = id % nx;
  w.y = (id / nx) % ny;
  w.z = id / (nx * ny);
  return w;
}

// ----------------------------------------------------------------------------------------------------------- forward
template <int KPW>
__global__ __launch_bounds__(NT, MINB_F) void rfft360_kernel(const SdyFftDesc f, const float* __restrict__ x,
                                                           const float* __restrict__ pa, const float* __restrict__ pd,
                                                           float* __restrict__ xn_out, float* __restrict__ Xf, int B,
                                                           int C, int K, int mtr, int ilv,
                                                           const int* __restrict__ mcut, const SdyImgMap xmap,
                                                           unsigned* flags, unsigned* head, int x_mod) {
  __shared__ __attribute__((aligned(16))) c2 Z[ROWS * P];
  __shared__ c2 TW[TWN];
  __shared__ c2 AD[ROWS];   // per-slot affine (a, d)
  const Lane L = lane_roles(Z);
  const WgId id = wg_id(C / ROWS, (K + KPW - 1) / KPW);
  const int c0 = id.x * ROWS, b = id.z;
  const int bx = sdy_img(xmap, b);   // drop-path skip: spectral row b of this launch is batch row bx of x / pa / pd / xn_out
  const int k_begin = id.y * KPW, k_end = min(K, k_begin + KPW);

  fill_twiddles(f, TW);
  if (threadIdx.x < ROWS) {
    const int s = threadIdx.x, ch = c0 + 4 * (s & 3) + (s >> 2);
    AD[s] = pa ? c2{pa[bx * C + ch], pd[bx * C + ch]} : c2{1.0f, 0.0f};
  }

  // x_mod > 0: the batch stacks calls that share their input tensor, which holds x_mod rows (the affine and xn_out stay per row)
  const int bsrc = x_mod > 0 ? bx % x_mod : bx;
  const float* xw = x + (((long)bsrc * C + c0 + L.wave) * K) * NLON + 4 * L.lane;    // row r of the wave: + 4 r K N
  float* xnw = xn_out ? xn_out + (((long)bx * C + c0 + L.wave) * K) * NLON + 4 * L.lane : nullptr;
  const long rstride = 4L * K * NLON;
  c2* zw = Z + 4 * L.wave * P + 2 * L.lane;
  const bool tail = L.lane < TAIL;

  // m-major pass: thread (c4, mg) stores channels 4*c4..4*c4+3 of m = mg + 64*it
  const int c4 = threadIdx.x & 3, mg = threadIdx.x >> 2;
  const float scale = 6.28318530717958647692f / (float)NLON;
  c2 W[MIT];        // -i * exp(-2 pi i m / N) * scale / 2
#pragma unroll
  for (int it = 0; it < MIT; ++it) {
    const c2 w = reinterpret_cast<const c2*>(f.pw)[min(mg + 64 * it, NH)];
    W[it] = c2{w.y, -w.x} * (0.5f * scale);
  }
  const c2* za = Z + c4 * P + mg;                  // Z[m]:     za[64 it]           (Z[180] is the copy of Z[0])
  const c2* zn01 = Z + c4 * P + (NH - 64 - mg);    // Z[n - m]: zn01[64 (1 - it)]   it = 0, 1
  const c2* zn2 = Z + c4 * P + max(NH - 128 - mg, 0);
  const long mstride = (long)K * B * 2 * C;
  const int re_off = (ilv ? 2 * c0 : c0) + 4 * c4, im_off = ilv ? 16 : C;   // see fft.h
  // ilv == 2 (tile-major, fft.h): this workgroup's 128-byte line of ring k, order m sits at
  //   m * mstride + (column tile j = (b * C/16 + c0/16) / 2) * K * 64 + k * 64 + (odd channel block ? 32 : 0)
  const long tile_base = (long)((b * (C / ROWS) + id.x) >> 1) * K * 64 + (id.x & 1) * 32 + 4 * c4;

  // fp16 range guard on behalf of the Legendre analysis (leg_par.hip stages (x[k] +- x[mirror]) * SDY_ACT_SX as fp16 and has
  // no register to spare): the largest magnitude this workgroup stores, checked once at the end (the kernel is HBM bound at
  // 52 registers: one v_max3 per two stored values is free)
  float amax = 0.0f;
  f32x4 regs[8];
  auto gload = [&](int k) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* src = xw + r * rstride + (long)k * NLON;
      regs[2 * r] = *reinterpret_cast<const f32x4*>(src);
      if (tail) regs[2 * r + 1] = *reinterpret_cast<const f32x4*>(src + 256);
    }
  };
  gload(k_begin);
  __syncthreads();   // AD, TW

  for (int k = k_begin; k < k_end; ++k) {
    // ---- stage the wave's rows: affine, optional store of the normalised field, LDS image z[j] = x[2j] + i x[2j+1]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const c2 ad = AD[4 * L.wave + r];
      const f32x4 v0 = regs[2 * r] * ad.x + ad.y, v1 = regs[2 * r + 1] * ad.x + ad.y;
      if (xnw) {
        float* dst = xnw + r * rstride + (long)k * NLON;
        *reinterpret_cast<f32x4*>(dst) = v0;
        if (tail) *reinterpret_cast<f32x4*>(dst + 256) = v1;
      }
      *reinterpret_cast<f32x4*>(zw + r * P) = v0;
      if (tail) *reinterpret_cast<f32x4*>(zw + r * P + 128) = v1;
    }
    if (k + 1 < k_end) gload(k + 1);   // in flight under the two passes
    __builtin_amdgcn_wave_barrier();

    // ---- pass A
    if (L.t < 15) {
      c2 v[12];
      const c2* src = L.row + L.t;
#pragma unroll
      for (int bb = 0; bb < 12; ++bb) v[bb] = src[15 * bb];
      pass_a_tail<-1>(L, v, TW);
    }
    pass_b<-1, true>(L);
    __syncthreads();

    // ---- split step + m-major stores: X[m] = h (A + conj B) + W (A - conj B), A = Z[m], B = Z[n - m]
    float* Xk = ilv == 2 ? Xf + tile_base + (long)k * 64 : Xf + ((long)k * B + b) * 2 * C + re_off;
    const int mlive = mcut ? min(mtr, mcut[k]) : mtr;   // polar cut-off: orders beyond it are never read downstream
#pragma unroll
    for (int it = 0; it < MIT; ++it) {
      const int m = mg + 64 * it;
      if (m < mlive) {
        f32x4 vr, vi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const c2 A = za[64 * it + 4 * j * P];
          const c2 Bv = it < 2 ? zn01[64 * (1 - it) + 4 * j * P] : zn2[4 * j * P];
          const c2 S = c2{A.x + Bv.x, A.y - Bv.y}, D = c2{A.x - Bv.x, A.y + Bv.y};
          const c2 X = (0.5f * scale) * S + cmul(D, W[it]);
          vr[j] = X.x;
          vi[j] = X.y;
          amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(X.x), __builtin_fabsf(X.y)));
        }
        float* o = Xk + (long)m * mstride;
        SDY_STREAM_STORE(o, vr);
        SDY_STREAM_STORE(o + im_off, vi);
      }
    }
    __syncthreads();   // the rows are rewritten by the next ring
  }
  // the fold adds the entries of the two hemispheres: 2 x SDY_ACT_SX x |Xf| bounds what the analysis stages
  sdy_flag_range(flags, amax * (2.0f * SDY_ACT_SX), head);
}

// ----------------------------------------------------------------------------------------------------------- inverse
template <int KPW>
__global__ __launch_bounds__(NT, MINB_I) void irfft360_kernel(const SdyFftDesc f, const float* __restrict__ Yf,
                                                            const float* __restrict__ bias, float* __restrict__ y, int B,
                                                            int C, int K, int mtr, int ilv,
                                                            const int* __restrict__ mcut, float* __restrict__ zt,
                                                            long zt_bs, double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) c2 Z[ROWS * P];
  __shared__ c2 TW[TWN];
  __shared__ c2 PW[NH];       // exp(+2 pi i j / N)
  __shared__ float BS[ROWS];  // per-slot bias
  const Lane L = lane_roles(Z);
  const WgId id = wg_id(C / ROWS, (K + KPW - 1) / KPW);
  const int c0 = id.x * ROWS, b = id.z;
  const int k_begin = id.y * KPW, k_end = min(K, k_begin + KPW);

  fill_twiddles(f, TW);
  if (threadIdx.x < NH) {
    const c2 w = reinterpret_cast<const c2*>(f.pw)[threadIdx.x];
    PW[threadIdx.x] = c2{w.x, -w.y};
  }
  if (threadIdx.x < ROWS) {
    const int s = threadIdx.x;
    BS[s] = bias ? bias[c0 + 4 * (s & 3) + (s >> 2)] : 0.0f;
  }

  // m-major loads: thread (c4, mg), m = mg + 64*it <= 180; rows of channels 4*c4 + j are slots c4 + 4*j
  const int c4 = threadIdx.x & 3, mg = threadIdx.x >> 2;
  const long mstride = (long)K * B * 2 * C;
  const int re_off = (ilv ? 2 * c0 : c0) + 4 * c4, im_off = ilv ? 16 : C;   // see fft.h
  const long tile_base = (long)((b * (C / ROWS) + id.x) >> 1) * K * 64 + (id.x & 1) * 32 + 4 * c4;   // ilv == 2, as in the forward
  f32x4 rr[MIT], ri[MIT];
  auto gload = [&](int k) {
    const float* Yk = ilv == 2 ? Yf + tile_base + (long)k * 64 : Yf + ((long)k * B + b) * 2 * C + re_off;
    const int mlive = mcut ? min(mtr, mcut[k]) : mtr;   // polar cut-off: orders beyond it were never written (zero)
#pragma unroll
    for (int it = 0; it < MIT; ++it) {
      const int m = mg + 64 * it;
      f32x4 vr = {0.f, 0.f, 0.f, 0.f}, vi = {0.f, 0.f, 0.f, 0.f};
      if (m < mlive) {
        const float* o = Yk + (long)m * mstride;
        vr = *reinterpret_cast<const f32x4*>(o);
        vi = *reinterpret_cast<const f32x4*>(o + im_off);
      }
      if (m == 0 || m == NH) vi = f32x4{0.f, 0.f, 0.f, 0.f};   // irfft ignores the imaginary parts of DC / Nyquist
      rr[it] = vr;
      ri[it] = vi;
    }
  };

  float* yw = y + (((long)b * C + c0 + L.wave) * K) * NLON + 4 * L.lane;
  const long rstride = 4L * K * NLON;
  const c2* zw = Z + 4 * L.wave * P + 2 * L.lane;
  const bool tail = L.lane < TAIL;
  c2* zs = Z + c4 * P + mg;

  gload(k_begin);
  for (int k = k_begin; k < k_end; ++k) {
    // ---- spectrum rows X[m], m = 0..180
#pragma unroll
    for (int it = 0; it < MIT; ++it) {
      if (mg + 64 * it <= NH) {
#pragma unroll
        for (int j = 0; j < 4; ++j) zs[4 * j * P + 64 * it] = c2{rr[it][j], ri[it][j]};
      }
    }
    __syncthreads();
    if (k + 1 < k_end) gload(k + 1);

    // ---- merge step folded into the loads of pass A: z[j] = S + i D, S = X[j] + conj X[n-j], D = e^{+2 pi i j/N} (X[j] - conj X[n-j])
    if (L.t < 15) {
      c2 v[12];
      const c2* lo = L.row + L.t;          // X[a + 15 b]
      const c2* hi = L.row + (15 - L.t);   // X[180 - a - 15 b] = hi[15 * (11 - b)]
      const c2* pw = PW + L.t;
#pragma unroll
      for (int bb = 0; bb < 12; ++bb) {
        const c2 A = lo[15 * bb], Bv = hi[15 * (11 - bb)], w = pw[15 * bb];
        const c2 S = c2{A.x + Bv.x, A.y - Bv.y}, Dm = c2{A.x - Bv.x, A.y + Bv.y};
        const c2 D = cmul(Dm, w);
        v[bb] = c2{S.x - D.y, S.y + D.x};
      }
      __builtin_amdgcn_wave_barrier();   // all merge reads of the wave precede the in-place writes
      pass_a_tail<+1>(L, v, TW);
    }
    pass_b<+1, false>(L);

    // ---- rows out: y[2j] = Re z[j], y[2j+1] = Im z[j]
    if (zt) {
      // Block with the inner skip folded into the dhconv (capi.hip, skip_foldable): the ring IS filter + skip + biases, so the
      // block's `x = act(...)` (sfnonet.py:310-311) happens here -- GELU, the store in the MLP's tile-major layout
      // [64-pixel tile][C][64] (a lane's four pixels never straddle a tile: 360 k and 4 lane are multiples of 4), and the
      // ring's share of the norm1 statistics: fp32 per 4-pixel quad, fp64 across the wave, ONE writer per (image, channel,
      // ring) slot -- no atomics, summed over the rings in a fixed order by instnorm_from_partials.
      float* ztb = zt + (long)b * zt_bs;
      const int px0 = k * NLON + 4 * L.lane, px1 = px0 + 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float bs = BS[4 * L.wave + r];
        const int ch = c0 + L.wave + 4 * r;
        f32x4 v = *reinterpret_cast<const f32x4*>(zw + r * P) + bs;
        v = f32x4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
        SDY_STREAM_STORE(ztb + (long)(px0 >> 6) * (C * 64) + ch * 64 + (px0 & 63), v);
        float qs = sdy_quad_sum(v), qs2 = sdy_quad_sumsq(v);
        if (tail) {
          f32x4 u = *reinterpret_cast<const f32x4*>(zw + r * P + 128) + bs;
          u = f32x4{gelu_erf(u.x), gelu_erf(u.y), gelu_erf(u.z), gelu_erf(u.w)};
          SDY_STREAM_STORE(ztb + (long)(px1 >> 6) * (C * 64) + ch * 64 + (px1 & 63), u);
          qs += sdy_quad_sum(u);
          qs2 += sdy_quad_sumsq(u);
        }
        // fp32 over the 16 lanes of a DPP row (<= 128 values, as conv_h3 does), fp64 across the wave's four rows
        double ds = (double)row16_sum(qs), ds2 = (double)row16_sum(qs2);
        ds += __shfl_xor(ds, 16, 64);
        ds2 += __shfl_xor(ds2, 16, 64);
        ds += __shfl_xor(ds, 32, 64);
        ds2 += __shfl_xor(ds2, 32, 64);
        if (L.lane == 0) {
          double* slot = part + (((long)b * K + k) * C + ch) * 2;   // [b][k][c]: the reader's threads (b, c) read coalesced
          slot[0] = ds;
          slot[1] = ds2;
        }
      }
    } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float bs = BS[4 * L.wave + r];
      float* dst = yw + r * rstride + (long)k * NLON;
      SDY_STREAM_STORE(dst, *reinterpret_cast<const f32x4*>(zw + r * P) + bs);
      if (tail) SDY_STREAM_STORE(dst + 256, *reinterpret_cast<const f32x4*>(zw + r * P + 128) + bs);
    }
    }
    __syncthreads();   // the rows are rewritten by the next ring
  }
}

}  // namespace

// nlon = 360 fast path; returns SDY_ERR_UNSUPPORTED when the shape does not fit (the caller falls back to fft.hip)
int sdy_fft360_launch_fwd(const SdyFftDesc& f, const float* x, const float* a, const float* d, float* xn_out, float* Xf,
                          int B, int C, int K, int mtr, int ilv, const int* mcut, hipStream_t stream,
                          const unsigned char* x_rows, int x_mod) {
  if (f.n != NH || C % ROWS != 0 || mtr > NH + 1 || (long)ROWS * K * NLON >= (1L << 31)) return SDY_ERR_UNSUPPORTED;
  if (ilv == 2 && (C % (2 * ROWS) != 0 || ROWS != 16)) return SDY_ERR_UNSUPPORTED;   // a 64-column tile = two channel blocks
  constexpr int KPW = SDY_FFT_KPW;
  SdyImgMap xmap;
  SDY_TRY(sdy_img_map_fill(xmap, x_rows, B));
  dim3 grid((C / ROWS) * ((K + KPW - 1) / KPW) * B);
  unsigned *flags = nullptr, *head = nullptr;
  if (f.guard_f16) {
    SDY_TRY(sdy_flags_ptr(&flags));
    SDY_TRY(sdy_headroom_ptr(SDY_RANGE_LEG_ANALYSIS, &head));
  }
  hipLaunchKernelGGL((rfft360_kernel<KPW>), grid, dim3(NT), 0, stream, f, x, a, d, xn_out, Xf, B, C, K, mtr, ilv, mcut, xmap,
                     flags, head, x_mod);
  return sdy_launch_status();
}

int sdy_fft360_launch_inv(const SdyFftDesc& f, const float* Yf, const float* bias, float* y, int B, int C, int K,
                          int mtr, int ilv, const int* mcut, hipStream_t stream, float* zt, long zt_bs, double* part) {
  if (f.n != NH || C % ROWS != 0 || mtr > NH + 1 || (long)ROWS * K * NLON >= (1L << 31)) return SDY_ERR_UNSUPPORTED;
  if (ilv == 2 && (C % (2 * ROWS) != 0 || ROWS != 16)) return SDY_ERR_UNSUPPORTED;   // a 64-column tile = two channel blocks
  if ((zt == nullptr) != (part == nullptr) || (!zt && !y)) return SDY_ERR_ARG;
  constexpr int KPW = SDY_FFT_KPW;
  dim3 grid((C / ROWS) * ((K + KPW - 1) / KPW) * B);
  hipLaunchKernelGGL((irfft360_kernel<KPW>), grid, dim3(NT), 0, stream, f, Yf, bias, y, B, C, K, mtr, ilv, mcut, zt, zt_bs, part);
  return sdy_launch_status();
}
