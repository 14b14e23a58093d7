// Longitude real FFT / inverse real FFT for the spherical-harmonic transform, gfx950.
//
// Restates the two torch.fft calls inside torch_harmonics (third-party; reference call sites
// src/models/sfno/s2convolutions.py:165,168,186):
//   forward : X = 2*pi * rfft(x, dim=-1, norm="forward")           (keeps m < mtr)
//   inverse : x = irfft(Y, n=nlon, dim=-1, norm="forward")          (imag of the m=0 / Nyquist bins ignored)
// and does the layout change between NCHW activations and the m-major spectral layout the Legendre GEMMs want:
//   x[b][c][k][w]  <->  Xf[m][k][b][ri][c]
// A workgroup owns CB channels of one (b, k) latitude ring: rows are read as coalesced 16-byte loads, transformed in
// LDS (real FFT of length N as a complex Stockham FFT of length n = N/2 with radices {4,2,3,5} plus a split step),
// and written as CB-float runs for each m.  HBM-bound; see DESIGN.md for the byte model.
// Fused on the way in: the per-(b,c) affine a*x+d (InstanceNorm + time scale/shift, sfnonet.py:292,298-299) and the
// optional store of the normalised field (the block's residual); on the way out: the filter bias
// (s2convolutions.py:188-189).
#include <cstdlib>
#include "common.h"
#include "fft.h"

namespace {

constexpr int CB = 16;        // channels (rows) per workgroup
constexpr int NT = 256;       // threads

struct cpx {
  float r, i;
};
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return cpx{a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return cpx{a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return cpx{a.r - b.r, a.i - b.i}; }
// multiply by (s * i), s = +-1
__device__ __forceinline__ cpx cmul_si(cpx a, float s) { return cpx{-s * a.i, s * a.r}; }

template <int R>
__device__ __forceinline__ void dft(cpx* v, float s);
template <>
__device__ __forceinline__ void dft<2>(cpx* v, float) {
  cpx a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}
template <>
__device__ __forceinline__ void dft<4>(cpx* v, float s) {
  cpx t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = cmul_si(csub(v[1], v[3]), s);
  v[0] = cadd(t0, t2);
  v[1] = cadd(t1, t3);
  v[2] = csub(t0, t2);
  v[3] = csub(t1, t3);
}
template <>
__device__ __forceinline__ void dft<3>(cpx* v, float s) {
  const float c = -0.5f, sn = 0.86602540378443864676f;
  cpx a = cadd(v[1], v[2]), b = csub(v[1], v[2]);
  cpx m = cpx{v[0].r + c * a.r, v[0].i + c * a.i};
  cpx t = cmul_si(cpx{sn * b.r, sn * b.i}, s);
  v[0] = cadd(v[0], a);
  v[1] = cadd(m, t);
  v[2] = csub(m, t);
}
template <>
__device__ __forceinline__ void dft<5>(cpx* v, float s) {
  const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
  const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  cpx a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
  cpx r1 = cpx{v[0].r + c1 * a1.r + c2 * a2.r, v[0].i + c1 * a1.i + c2 * a2.i};
  cpx r2 = cpx{v[0].r + c2 * a1.r + c1 * a2.r, v[0].i + c2 * a1.i + c1 * a2.i};
  cpx i1 = cmul_si(cpx{s1 * b1.r + s2 * b2.r, s1 * b1.i + s2 * b2.i}, s);
  cpx i2 = cmul_si(cpx{s2 * b1.r - s1 * b2.r, s2 * b1.i - s1 * b2.i}, s);
  v[0] = cadd(v[0], cadd(a1, a2));
  v[1] = cadd(r1, i1);
  v[4] = csub(r1, i1);
  v[2] = cadd(r2, i2);
  v[3] = csub(r2, i2);
}

// One Stockham butterfly: inputs src[j + q*nb], outputs dst[j0 + t*Ns];  s = -1 forward, +1 inverse.
template <int R>
__device__ __forceinline__ void butterfly(const float* sre, const float* sim, float* dre, float* dim_, int j, int nb,
                                          int Ns, int tstride, const float* twr, const float* twi, float s) {
  const int k = j % Ns;
  cpx v[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    cpx x = cpx{sre[j + q * nb], sim[j + q * nb]};
    if (q > 0 && k > 0) {
      const int ti = q * k * tstride;
      x = cmul(x, cpx{twr[ti], -s * twi[ti]});  // table holds exp(-2*pi*i*j/n)
    }
    v[q] = x;
  }
  dft<R>(v, s);
  const int j0 = (j / Ns) * Ns * R + k;
#pragma unroll
  for (int t = 0; t < R; ++t) {
    dre[j0 + t * Ns] = v[t].r;
    dim_[j0 + t * Ns] = v[t].i;
  }
}

struct LdsView {
  float *a_re, *a_im, *b_re, *b_im, *tw_re, *tw_im, *pw_re, *pw_im;
};
__device__ __forceinline__ LdsView carve(float* sm, int S, int n) {
  LdsView v;
  v.a_re = sm;
  v.a_im = v.a_re + CB * S;
  v.b_re = v.a_im + CB * S;
  v.b_im = v.b_re + CB * S;
  v.tw_re = v.b_im + CB * S;
  v.tw_im = v.tw_re + n;
  v.pw_re = v.tw_im + n;
  v.pw_im = v.pw_re + (n + 1);
  return v;
}

// complex FFT of length n on CB rows, ping-pong between (a) and (b); returns with result in *res_re / *res_im
__device__ __forceinline__ void fft_rows(const SdyFftDesc& f, LdsView& L, int S, float s, float** res_re,
                                         float** res_im) {
  float *sre = L.a_re, *sim = L.a_im, *dre = L.b_re, *dim_ = L.b_im;
  const int n = f.n;
  int Ns = 1;
  for (int st = 0; st < f.nstages; ++st) {
    const int R = f.radices[st];
    const int nb = n / R;
    const int tstride = n / (Ns * R);
    for (int id = threadIdx.x; id < CB * nb; id += NT) {
      const int r = id / nb, j = id - r * nb;
      const float* pr = sre + r * S;
      const float* pi = sim + r * S;
      float* qr = dre + r * S;
      float* qi = dim_ + r * S;
      switch (R) {
        case 2: butterfly<2>(pr, pi, qr, qi, j, nb, Ns, tstride, L.tw_re, L.tw_im, s); break;
        case 3: butterfly<3>(pr, pi, qr, qi, j, nb, Ns, tstride, L.tw_re, L.tw_im, s); break;
        case 4: butterfly<4>(pr, pi, qr, qi, j, nb, Ns, tstride, L.tw_re, L.tw_im, s); break;
        default: butterfly<5>(pr, pi, qr, qi, j, nb, Ns, tstride, L.tw_re, L.tw_im, s); break;
      }
    }
    __syncthreads();
    float* t;
    t = sre; sre = dre; dre = t;
    t = sim; sim = dim_; dim_ = t;
    Ns *= R;
  }
  *res_re = sre;
  *res_im = sim;
}

// Compile-time variant: every stride / divisor is a constant, so hipcc folds the integer divisions of the generic path
// (several per butterfly, ~25 VALU instructions each on CDNA) into multiply-shifts and fully unrolls the stage loop.
template <int N_HALF>
struct FftShape;
template <>
struct FftShape<180> {  // nlon = 360
  static constexpr int nst = 4;
  static constexpr int r[4] = {3, 5, 4, 3};   // order chosen for the fewest LDS bank conflicts of the Stockham writes (simulated: 476 extra cycles vs 742 for 4,3,3,5)
};
template <>
struct FftShape<32> {   // nlon = 64 (test grids)
  static constexpr int nst = 3;
  static constexpr int r[4] = {4, 4, 2, 1};
};

template <int N_HALF, int ST, int NS>
__device__ __forceinline__ void fft_stage_ct(float*& sre, float*& sim, float*& dre, float*& dim_, const float* twr,
                                             const float* twi, int S, float s) {
  if constexpr (ST < FftShape<N_HALF>::nst) {
    constexpr int R = FftShape<N_HALF>::r[ST];
    constexpr int nb = N_HALF / R;
    constexpr int tstride = N_HALF / (NS * R);
    for (int id = threadIdx.x; id < CB * nb; id += NT) {
      const int r = id / nb, j = id - r * nb;
      butterfly<R>(sre + r * S, sim + r * S, dre + r * S, dim_ + r * S, j, nb, NS, tstride, twr, twi, s);
    }
    __syncthreads();
    float* t;
    t = sre; sre = dre; dre = t;
    t = sim; sim = dim_; dim_ = t;
    fft_stage_ct<N_HALF, ST + 1, NS * R>(sre, sim, dre, dim_, twr, twi, S, s);
  }
}

template <int N_HALF>
__device__ __forceinline__ void fft_rows_ct(LdsView& L, int S, float s, float** res_re, float** res_im) {
  float *sre = L.a_re, *sim = L.a_im, *dre = L.b_re, *dim_ = L.b_im;
  fft_stage_ct<N_HALF, 0, 1>(sre, sim, dre, dim_, L.tw_re, L.tw_im, S, s);
  *res_re = sre;
  *res_im = sim;
}

__device__ __forceinline__ void load_tables(const SdyFftDesc& f, LdsView& L) {
  const int n = f.n;
  for (int i = threadIdx.x; i < n; i += NT) {
    L.tw_re[i] = f.tw[2 * i];
    L.tw_im[i] = f.tw[2 * i + 1];
  }
  for (int i = threadIdx.x; i <= n; i += NT) {
    L.pw_re[i] = f.pw[2 * i];
    L.pw_im[i] = f.pw[2 * i + 1];
  }
}

// x (B,C,K,N) -> Xf[m][k][b][ri][c]      (NH = nlon/2 known at compile time, or 0 for the generic path)
// A workgroup walks KPW consecutive latitude rings of its (b, channel block): the global loads of ring k+1 are issued
// into registers before the FFT passes of ring k, so the load latency hides under the LDS work instead of opening every
// ring (the kernel is otherwise a strict load -> transform -> store sequence with 3 workgroups per CU).
template <int NH, int KPW>
__global__ __launch_bounds__(NT) void rfft_fwd_kernel(const SdyFftDesc f, const float* __restrict__ x,
                                                       const float* __restrict__ pa, const float* __restrict__ pd,
                                                       float* __restrict__ xn_out, float* __restrict__ Xf, int B,
                                                       int C, int K, int mtr, int ilv, unsigned* flags) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int n = NH ? NH : f.n, N = 2 * n, S = NH ? ((NH + 1) | 1) : f.S;
  LdsView L = carve(sm, S, n);
  float amax = 0.0f;   // fp16 range guard on behalf of the folded Legendre analysis (fft360.hip has the reasoning)
  const int c0 = blockIdx.x * CB, b = blockIdx.z;
  const int k_begin = blockIdx.y * KPW, k_end = min(K, k_begin + KPW);
  load_tables(f, L);

  const int q4 = N / 4;
  constexpr int ITER = NH ? (CB * (NH / 2) + NT - 1) / NT : 1;
  f32x4 regs[ITER];
  auto gload = [&](int k) {          // NH != 0 only
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = threadIdx.x + it * NT;
      const int r = idx / q4, q = idx - r * q4;
      const int c = c0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < CB * q4 && c < C) v = *reinterpret_cast<const f32x4*>(x + (((long)b * C + c) * K + k) * N + 4 * q);
      regs[it] = v;
    }
  };
  auto stage_row = [&](int idx, f32x4 v, int k) {   // affine, optional residual store, LDS image
    const int r = idx / q4, q = idx - r * q4;
    const int c = c0 + r;
    if (c < C) {
      if (pa) v = v * pa[b * C + c] + pd[b * C + c];
      if (xn_out) *reinterpret_cast<f32x4*>(xn_out + (((long)b * C + c) * K + k) * N + 4 * q) = v;
    }
    L.a_re[r * S + 2 * q] = v.x;
    L.a_im[r * S + 2 * q] = v.y;
    L.a_re[r * S + 2 * q + 1] = v.z;
    L.a_im[r * S + 2 * q + 1] = v.w;
  };

  if constexpr (NH != 0) gload(k_begin);
  for (int k = k_begin; k < k_end; ++k) {
    if constexpr (NH != 0) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int idx = threadIdx.x + it * NT;
        if (idx < CB * q4) stage_row(idx, regs[it], k);
      }
    } else {
      for (int idx = threadIdx.x; idx < CB * q4; idx += NT) {
        const int r = idx / q4, q = idx - r * q4;
        const int c = c0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < C) v = *reinterpret_cast<const f32x4*>(x + (((long)b * C + c) * K + k) * N + 4 * q);
        stage_row(idx, v, k);
      }
    }
    __syncthreads();
    if constexpr (NH != 0) {
      if (k + 1 < k_end) gload(k + 1);   // in flight under the FFT passes below
    }

    float *zr, *zi;
    if constexpr (NH != 0)
      fft_rows_ct<NH>(L, S, -1.0f, &zr, &zi);
    else
      fft_rows(f, L, S, -1.0f, &zr, &zi);

    // split step: X[m] = E + w^m O, X[n-m] = conj(E - w^m O); scaled by 2*pi/N
    const float scale = 6.28318530717958647692f / (float)N;
    const int np = n / 2 + 1;
    for (int id = threadIdx.x; id < CB * np; id += NT) {
      const int r = id / np, m = id - r * np;
      float* re = zr + r * S;
      float* im = zi + r * S;
      const int m2 = (m == 0) ? 0 : n - m;
      const float a = re[m], bq = im[m], c = re[m2], d = im[m2];
      const cpx E = cpx{0.5f * (a + c), 0.5f * (bq - d)};
      const cpx O = cpx{0.5f * (bq + d), -0.5f * (a - c)};
      const cpx T = cmul(cpx{L.pw_re[m], L.pw_im[m]}, O);
      re[m] = scale * (E.r + T.r);
      im[m] = scale * (E.i + T.i);
      re[n - m] = scale * (E.r - T.r);
      im[n - m] = -scale * (E.i - T.i);
    }
    __syncthreads();

    // 16-byte stores: a thread owns 4 consecutive channels (C % 4 == 0) of one m; 4 lanes cover the workgroup's 64-byte
    // run, 16 m's per wave instruction
    const int c4 = threadIdx.x & 3, mg = threadIdx.x >> 2;
    const int c = c0 + 4 * c4;
    if (c < C) {
      for (int m = mg; m < mtr; m += NT / 4) {
        const long o = (((long)m * K + k) * B + b) * (2L * C) + (ilv ? c0 + c : c);   // ilv: fft.h
        f32x4 vr, vi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vr[j] = zr[(4 * c4 + j) * S + m];
          vi[j] = zi[(4 * c4 + j) * S + m];
          amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(vr[j]), __builtin_fabsf(vi[j])));
        }
        *reinterpret_cast<f32x4*>(Xf + o) = vr;
        *reinterpret_cast<f32x4*>(Xf + o + (ilv ? CB : C)) = vi;
      }
    }
    __syncthreads();   // the LDS image is rewritten by the next ring
  }
  sdy_flag_range(flags, amax * (2.0f * SDY_ACT_SX));
}

// Yf[m][k][b][ri][c] -> y (B,C,K,N) (+ bias[c]); same ring pipeline as the forward kernel
template <int NH, int KPW>
__global__ __launch_bounds__(NT) void irfft_kernel(const SdyFftDesc f, const float* __restrict__ Yf,
                                                    const float* __restrict__ bias, float* __restrict__ y, int B, int C,
                                                    int K, int mtr, int ilv) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int n = NH ? NH : f.n, N = 2 * n, S = NH ? ((NH + 1) | 1) : f.S;
  LdsView L = carve(sm, S, n);
  const int c0 = blockIdx.x * CB, b = blockIdx.z;
  const int k_begin = blockIdx.y * KPW, k_end = min(K, k_begin + KPW);
  load_tables(f, L);

  const int c4 = threadIdx.x & 3, mg = threadIdx.x >> 2;
  const int cq = c0 + 4 * c4;
  constexpr int MIT = NH ? (NH + 1 + NT / 4 - 1) / (NT / 4) : 1;
  f32x4 rr[MIT], ri[MIT];
  auto gload = [&](int k) {          // NH != 0 only
#pragma unroll
    for (int it = 0; it < MIT; ++it) {
      const int m = mg + it * (NT / 4);
      f32x4 vr = {0.f, 0.f, 0.f, 0.f}, vi = {0.f, 0.f, 0.f, 0.f};
      if (cq < C && m < mtr) {
        const long o = (((long)m * K + k) * B + b) * (2L * C) + (ilv ? c0 + cq : cq);
        vr = *reinterpret_cast<const f32x4*>(Yf + o);
        vi = *reinterpret_cast<const f32x4*>(Yf + o + (ilv ? CB : C));
      }
      rr[it] = vr;
      ri[it] = vi;
    }
  };
  auto stage_m = [&](int m, const f32x4& vr, const f32x4& vi) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      L.a_re[(4 * c4 + j) * S + m] = vr[j];
      L.a_im[(4 * c4 + j) * S + m] = vi[j];
    }
  };

  if constexpr (NH != 0) gload(k_begin);
  for (int k = k_begin; k < k_end; ++k) {
    if constexpr (NH != 0) {
#pragma unroll
      for (int it = 0; it < MIT; ++it) {
        const int m = mg + it * (NT / 4);
        if (m <= n) stage_m(m, rr[it], ri[it]);
      }
    } else {
      for (int m = mg; m <= n; m += NT / 4) {
        f32x4 vr = {0.f, 0.f, 0.f, 0.f}, vi = {0.f, 0.f, 0.f, 0.f};
        if (cq < C && m < mtr) {
          const long o = (((long)m * K + k) * B + b) * (2L * C) + (ilv ? c0 + cq : cq);
          vr = *reinterpret_cast<const f32x4*>(Yf + o);
          vi = *reinterpret_cast<const f32x4*>(Yf + o + (ilv ? CB : C));
        }
        stage_m(m, vr, vi);
      }
    }
    __syncthreads();
    if constexpr (NH != 0) {
      if (k + 1 < k_end) gload(k + 1);
    }

    // merge step: Z[m] = S + i D, Z[n-m] = conj(S - i D),  S = X[m] + conj(X[n-m]),  D = e^{+2 pi i m/N} (X[m] - conj(X[n-m]))
    const int np = n / 2 + 1;
    for (int id = threadIdx.x; id < CB * np; id += NT) {
      const int r = id / np, m = id - r * np;
      float* re = L.a_re + r * S;
      float* im = L.a_im + r * S;
      cpx A = cpx{re[m], im[m]};
      cpx Bc = cpx{re[n - m], -im[n - m]};
      if (m == 0) {  // irfft ignores the imaginary parts of the DC and Nyquist bins
        A.i = 0.f;
        Bc.i = 0.f;
      }
      const cpx Sm = cadd(A, Bc);
      const cpx D = cmul(cpx{L.pw_re[m], -L.pw_im[m]}, csub(A, Bc));
      re[m] = Sm.r - D.i;   // Z[m] = S + iD
      im[m] = Sm.i + D.r;
      if (m > 0) {          // Z[n-m] = conj(S - iD)
        re[n - m] = Sm.r + D.i;
        im[n - m] = -(Sm.i - D.r);
      }
    }
    __syncthreads();

    float *zr, *zi;
    if constexpr (NH != 0)
      fft_rows_ct<NH>(L, S, +1.0f, &zr, &zi);
    else
      fft_rows(f, L, S, +1.0f, &zr, &zi);

    const int q4 = N / 4;
    for (int idx = threadIdx.x; idx < CB * q4; idx += NT) {
      const int r = idx / q4, q = idx - r * q4;
      const int c = c0 + r;
      if (c < C) {
        const float bv = bias ? bias[c] : 0.0f;
        f32x4 v;
        v.x = zr[r * S + 2 * q] + bv;
        v.y = zi[r * S + 2 * q] + bv;
        v.z = zr[r * S + 2 * q + 1] + bv;
        v.w = zi[r * S + 2 * q + 1] + bv;
        *reinterpret_cast<f32x4*>(y + (((long)b * C + c) * K + k) * N + 4 * q) = v;
      }
    }
    __syncthreads();   // the LDS image is rewritten by the next ring
  }
}

size_t fft_smem_bytes(const SdyFftDesc& f) { return ((size_t)4 * CB * f.S + 2 * f.n + 2 * (f.n + 1)) * sizeof(float); }

// SDY_NO_FFT360=1 keeps the generic Stockham kernels for nlon = 360 too (A/B measurements)
bool use_generic_only() {
  static const bool v = getenv("SDY_NO_FFT360") != nullptr;   // (set = on, like every other switch: capi.hip, SdySwitches)
  return v;
}

}  // namespace

int sdy_fft_launch_fwd(const SdyFftDesc& f, const float* x, const float* a, const float* d, float* xn_out, float* Xf,
                       int B, int C, int K, int mtr, int ilv, const int* mcut, hipStream_t stream,
                       const unsigned char* x_rows, int x_mod) {
  if (ilv && C % CB != 0) return SDY_ERR_UNSUPPORTED;
  if (f.n == 180 && !use_generic_only()) {
    const int rc = sdy_fft360_launch_fwd(f, x, a, d, xn_out, Xf, B, C, K, mtr, ilv, mcut, stream, x_rows, x_mod);
    if (rc != SDY_ERR_UNSUPPORTED) return rc;
  }
  // polar cut-off / tile-major layout / image map / shared input rows: contracts of fft360 only
  if (mcut || ilv == 2 || x_rows || x_mod) return SDY_ERR_UNSUPPORTED;
  const size_t smem = fft_smem_bytes(f);
  if (smem > 64 * 1024) return SDY_ERR_UNSUPPORTED;
  constexpr int KPW = 4;   // latitude rings per workgroup on the pipelined (compile-time size) paths
  dim3 grid((C + CB - 1) / CB, (K + KPW - 1) / KPW, B);
  unsigned* flags = nullptr;
  if (f.guard_f16) SDY_TRY(sdy_flags_ptr(&flags));
  if (f.n == 180) {
    hipLaunchKernelGGL((rfft_fwd_kernel<180, KPW>), grid, dim3(NT), smem, stream, f, x, a, d, xn_out, Xf, B, C, K, mtr, ilv, flags);
  } else if (f.n == 32) {
    hipLaunchKernelGGL((rfft_fwd_kernel<32, KPW>), grid, dim3(NT), smem, stream, f, x, a, d, xn_out, Xf, B, C, K, mtr, ilv, flags);
  } else {
    grid.y = K;
    hipLaunchKernelGGL((rfft_fwd_kernel<0, 1>), grid, dim3(NT), smem, stream, f, x, a, d, xn_out, Xf, B, C, K, mtr, ilv, flags);
  }
  return sdy_launch_status();
}

int sdy_fft_launch_inv(const SdyFftDesc& f, const float* Yf, const float* bias, float* y, int B, int C, int K, int mtr,
                       int ilv, const int* mcut, hipStream_t stream) {
  if (ilv && C % CB != 0) return SDY_ERR_UNSUPPORTED;
  if (f.n == 180 && !use_generic_only()) {
    const int rc = sdy_fft360_launch_inv(f, Yf, bias, y, B, C, K, mtr, ilv, mcut, stream);
    if (rc != SDY_ERR_UNSUPPORTED) return rc;
  }
  if (mcut || ilv == 2) return SDY_ERR_UNSUPPORTED;
  const size_t smem = fft_smem_bytes(f);
  if (smem > 64 * 1024) return SDY_ERR_UNSUPPORTED;
  constexpr int KPW = 4;
  dim3 grid((C + CB - 1) / CB, (K + KPW - 1) / KPW, B);
  if (f.n == 180) {
    hipLaunchKernelGGL((irfft_kernel<180, KPW>), grid, dim3(NT), smem, stream, f, Yf, bias, y, B, C, K, mtr, ilv);
  } else if (f.n == 32) {
    hipLaunchKernelGGL((irfft_kernel<32, KPW>), grid, dim3(NT), smem, stream, f, Yf, bias, y, B, C, K, mtr, ilv);
  } else {
    grid.y = K;
    hipLaunchKernelGGL((irfft_kernel<0, 1>), grid, dim3(NT), smem, stream, f, Yf, bias, y, B, C, K, mtr, ilv);
  }
  return sdy_launch_status();
}
