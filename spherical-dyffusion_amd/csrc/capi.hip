// extern "C" boundary (include/sdy_amd.h): SHT plan, stage launchers, 1x1-conv GEMM, the SFNO network object and
// the sampler arithmetic.  Host orchestration only -- every kernel lives in gemm.hip / fft.hip / pointwise.hip.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "fft.h"
#include "pointwise.h"

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline size_t round_up_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }

// Generic host-side splitter: value(b, r, k) -> fp16 hi | lo planes [nb][rows_pad][Kpad], scaled by a power of two so
// that max|v|*scale is in [2^12, 2^13) (hi stays far below the fp16 maximum, lo parts stay out of the subnormals).
template <class F>
static float h3_pack_host(std::vector<_Float16>& buf, int nb, int rows, int K, int rows_pad, int Kpad, F value) {
  float mx = 0.f;
  for (int b = 0; b < nb; ++b)
    for (int r = 0; r < rows; ++r)
      for (int k = 0; k < K; ++k) mx = std::fmax(mx, std::fabs(value(b, r, k)));
  float s = 1.0f;
  if (mx > 0.f && std::isfinite(mx)) {
    int e;
    std::frexp(mx, &e);            // mx = f * 2^e, f in [0.5, 1)
    s = std::ldexp(1.0f, 13 - e);
  }
  const size_t plane = (size_t)nb * rows_pad * Kpad;
  buf.assign(2 * plane, (_Float16)0.0f);
  _Float16* hi = buf.data();
  _Float16* lo = hi + plane;
  for (int b = 0; b < nb; ++b)
    for (int r = 0; r < rows; ++r) {
      const size_t base = ((size_t)b * rows_pad + r) * Kpad;
      for (int k = 0; k < K; ++k) {
        const float v = value(b, r, k) * s;
        const _Float16 hv = (_Float16)v;
        hi[base + k] = hv;
        lo[base + k] = (_Float16)(v - (float)hv);
      }
    }
  return s;
}


extern "C" int sdy_version(void) { return 101; }

extern "C" int sdy_abi_check(const size_t* sizes, int n) {
  const size_t mine[SDY_ABI_STRUCTS] = {sizeof(sdy_conv_args), sizeof(sdy_mlp_args), sizeof(sdy_pair_args), sizeof(sdy_sfno_config),
                                        sizeof(sdy_sfno_fwd_args), sizeof(sdy_var_table), sizeof(sdy_step_finish_args)};
  if (!sizes || n != SDY_ABI_STRUCTS) return SDY_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (sizes[i] != mine[i]) return SDY_ERR_ARG;
  return SDY_OK;
}

extern "C" const char* sdy_error_string(int code) {
  switch (code) {
    case SDY_OK: return "ok";
    case SDY_ERR_ARG: return "bad argument (null pointer or non-positive extent)";
    case SDY_ERR_UNSUPPORTED: return "unsupported size/configuration";
    case SDY_ERR_ALIGN: return "extent along a contiguous dimension is not a multiple of 4";
    case SDY_ERR_WORKSPACE: return "workspace too small";
    case SDY_ERR_NAME: return "unknown parameter name";
    case SDY_ERR_SHAPE: return "parameter has the wrong number of elements";
    case SDY_ERR_STATE: return "object not fully initialised";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

// =========================================================================================================
// SHT plan
// =========================================================================================================
struct sdy_sht_plan {
  int nlat, nlon, lmax, mmax, mtr, grid;
  int Lpad4, Kpad4;
  float* d_wqT = nullptr;  // [mtr][nlat][Lpad4]   forward table, quadrature-weighted, l contiguous
  float* d_pct = nullptr;  // [mtr][lmax][Kpad4]   inverse table, k contiguous
  float* d_tw = nullptr;
  float* d_pw = nullptr;
  SdyFftDesc fft;
  // gemm_mode 1: split-fp16 copies of the two tables, [mtr][rows_pad][Kpad] hi | lo (gemm_h3.hip)
  int gemm_mode = 0;
  void* d_wq_h3 = nullptr;   // analysis: rows l, k = latitude
  void* d_pct_h3 = nullptr;  // synthesis: rows = latitude, k = l
  float s_wq = 1.f, s_pct = 1.f;
  int h3_rows_fwd = 0, h3_k_fwd = 0, h3_rows_inv = 0, h3_k_inv = 0;
  // grids with nlat, lmax <= 192: fragment-stream tables of the skinny Legendre kernel (leg_h3.hip)
  void* d_wq_frag = nullptr;
  void* d_pct_frag = nullptr;
  float s_wq_frag = 1.f, s_pct_frag = 1.f;
  // the same with the equatorial symmetry folded in (leg_par.hip; even nlat)
  void* d_wq_par = nullptr;
  void* d_pct_par = nullptr;
  float s_wq_par = 1.f, s_pct_par = 1.f;
  // Polar cut-off shared by fft360 and leg_par inside the fused forward: rows k' < kdead[m] (and their mirrors) of order m
  // carry table entries below 1e-12 of the order's maximum in BOTH tables, so they are neither written, read nor multiplied;
  // mcut[k] = number of orders that are live on ring k (kdead is non-decreasing in m).
  int* d_kdead = nullptr;   // [mtr]
  int* d_mcut = nullptr;    // [nlat]
};

// Every environment switch of the library, read once per process.  Each routes a stage to its fallback kernel (the only
// implementation for other shapes; tests/test_gpu_variants.py holds all of them to the default path's output) -- INTEGRATION.md
// section 5 lists what they do.  (Retired in round 6 with their questions settled: SDY_NO_FUSED_STATS, SDY_NO_POLAR_SKIP,
// SDY_NO_PAIR, SDY_NO_CONV_FRAG -- the paths they selected are still what other shapes take, and are tested there.)
namespace {
struct SdySwitches {
  bool gemm_f32, no_fft360, no_leg_par, no_leg_frag, no_dh_frag, no_fused_mlp, no_drop_skip, no_skip_fold;
};
const SdySwitches& sw() {
  static const SdySwitches v = [] {
    auto on = [](const char* n) { return std::getenv(n) != nullptr; };
    const char* g = std::getenv("SDY_GEMM_MODE");
    return SdySwitches{g && std::string(g) == "f32", on("SDY_NO_FFT360"), on("SDY_NO_LEG_PAR"), on("SDY_NO_LEG_FRAG"),
                       on("SDY_NO_DH_FRAG"), on("SDY_NO_FUSED_MLP"), on("SDY_NO_DROP_SKIP"), on("SDY_NO_SKIP_FOLD")};
  }();
  return v;
}
}  // namespace
static int env_gemm_mode() { return sw().gemm_f32 ? 0 : 1; }

extern "C" int sdy_sht_plan_create(int nlat, int nlon, int lmax, int mmax, int grid, sdy_sht_plan** out) {
  return sdy_sht_plan_create_ex(nlat, nlon, lmax, mmax, grid, env_gemm_mode(), out);
}

extern "C" int sdy_sht_plan_create_ex(int nlat, int nlon, int lmax, int mmax, int grid, int gemm_mode,
                                      sdy_sht_plan** out) {
  if (gemm_mode != 0 && gemm_mode != 1) return SDY_ERR_ARG;
  if (!out || nlat < 2 || nlon < 4 || lmax < 1 || mmax < 1) return SDY_ERR_ARG;
  if (nlon % 4) return SDY_ERR_ALIGN;
  if (mmax > nlon / 2 + 1) return SDY_ERR_ARG;
  if (grid != SDY_GRID_EQUIANGULAR && grid != SDY_GRID_LEGENDRE_GAUSS) return SDY_ERR_ARG;
  sdy_sht_plan* p = new sdy_sht_plan();
  p->nlat = nlat; p->nlon = nlon; p->lmax = lmax; p->mmax = mmax; p->grid = grid;
  p->mtr = mmax < lmax ? mmax : lmax;  // columns m >= lmax are identically zero
  p->Lpad4 = round_up(lmax, 4);
  p->Kpad4 = round_up(nlat, 4);
  const int n = nlon / 2;
  p->fft.n = n;
  p->fft.S = (n + 1) | 1;
  p->fft.guard_f16 = gemm_mode == 1 ? 1 : 0;   // split-precision Legendre kernels stage the transform's output as fp16
  int r = sdy_factor_radices(n, p->fft.radices, &p->fft.nstages);
  if (r != SDY_OK) { delete p; return r; }
  if ((size_t)(4 * 16 * p->fft.S + 4 * n + 2) * sizeof(float) > 64 * 1024) { delete p; return SDY_ERR_UNSUPPORTED; }

  // fp64 tables on the host, cast to fp32 like `.float()` (src/models/sfno/sfnonet.py:551-554)
  std::vector<double> pct((size_t)mmax * lmax * nlat), w(nlat);
  r = sdy_sht_tables_host(nlat, nlon, lmax, mmax, grid, pct.data(), w.data(), nullptr);
  if (r != SDY_OK) { delete p; return r; }
  const int mtr = p->mtr;
  std::vector<float> wqT((size_t)mtr * nlat * p->Lpad4, 0.0f), pf((size_t)mtr * lmax * p->Kpad4, 0.0f);
  for (int m = 0; m < mtr; ++m)
    for (int l = 0; l < lmax; ++l)
      for (int k = 0; k < nlat; ++k) {
        const double v = pct[((size_t)m * lmax + l) * nlat + k];
        // reference casts the fp64 product pct*w to fp32 (weights buffer), and pct to fp32 (pct buffer)
        wqT[((size_t)m * nlat + k) * p->Lpad4 + l] = (float)(v * w[k]);
        pf[((size_t)m * lmax + l) * p->Kpad4 + k] = (float)v;
      }
  std::vector<float> tw(2 * (size_t)n), pw(2 * (size_t)(n + 1));
  for (int j = 0; j < n; ++j) {
    const double a = -2.0 * M_PI * j / n;
    tw[2 * j] = (float)std::cos(a);
    tw[2 * j + 1] = (float)std::sin(a);
  }
  for (int m = 0; m <= n; ++m) {
    const double a = -2.0 * M_PI * m / nlon;
    pw[2 * m] = (float)std::cos(a);
    pw[2 * m + 1] = (float)std::sin(a);
  }
  hipError_t e;
#define PLAN_UP(dst, vec)                                                                         \
  e = hipMalloc((void**)&(dst), (vec).size() * sizeof(float));                                    \
  if (e == hipSuccess) e = hipMemcpy((dst), (vec).data(), (vec).size() * sizeof(float), hipMemcpyHostToDevice); \
  if (e != hipSuccess) { sdy_sht_plan_destroy(p); return (int)e; }
  PLAN_UP(p->d_wqT, wqT)
  PLAN_UP(p->d_pct, pf)
  PLAN_UP(p->d_tw, tw)
  PLAN_UP(p->d_pw, pw)
#undef PLAN_UP
  p->gemm_mode = gemm_mode;
  if (gemm_mode == 1) {
    // the fp32 tables (exactly what the reference's `.float()` buffers hold) split into fp16 hi + lo
    std::vector<_Float16> buf;
    p->h3_rows_fwd = lmax > 128 ? round_up(lmax, 256) : 128; p->h3_k_fwd = round_up(nlat, 64);
    p->s_wq = h3_pack_host(buf, mtr, lmax, nlat, p->h3_rows_fwd, p->h3_k_fwd, [&](int m, int l, int k) {
      return wqT[((size_t)m * nlat + k) * p->Lpad4 + l];
    });
    e = hipMalloc(&p->d_wq_h3, buf.size() * sizeof(_Float16));
    if (e == hipSuccess) e = hipMemcpy(p->d_wq_h3, buf.data(), buf.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    if (e != hipSuccess) { sdy_sht_plan_destroy(p); return (int)e; }
    p->h3_rows_inv = nlat > 128 ? round_up(nlat, 256) : 128; p->h3_k_inv = round_up(lmax, 64);
    p->s_pct = h3_pack_host(buf, mtr, nlat, lmax, p->h3_rows_inv, p->h3_k_inv, [&](int m, int k, int l) {
      return pf[((size_t)m * lmax + l) * p->Kpad4 + k];
    });
    e = hipMalloc(&p->d_pct_h3, buf.size() * sizeof(_Float16));
    if (e == hipSuccess) e = hipMemcpy(p->d_pct_h3, buf.data(), buf.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    if (e != hipSuccess) { sdy_sht_plan_destroy(p); return (int)e; }
  }
  if (gemm_mode == 1 && sdy_leg_h3_supported(lmax, nlat) && sdy_leg_h3_supported(nlat, lmax)) {
    struct Ctx { const float* t; int nlat, lmax, Lpad4, Kpad4; } cx{nullptr, nlat, lmax, p->Lpad4, p->Kpad4};
    e = hipMalloc(&p->d_wq_frag, sdy_leg_h3_table_bytes(mtr));
    if (e == hipSuccess) e = hipMalloc(&p->d_pct_frag, sdy_leg_h3_table_bytes(mtr));
    if (e != hipSuccess) { sdy_sht_plan_destroy(p); return (int)e; }
    cx.t = wqT.data();
    r = sdy_leg_h3_pack(mtr, lmax, nlat, [](void* c, int m, int l, int k) {
      const Ctx* x = static_cast<const Ctx*>(c);
      return x->t[((size_t)m * x->nlat + k) * x->Lpad4 + l];
    }, &cx, p->d_wq_frag, &p->s_wq_frag);
    cx.t = pf.data();
    if (r == SDY_OK) r = sdy_leg_h3_pack(mtr, nlat, lmax, [](void* c, int m, int k, int l) {
      const Ctx* x = static_cast<const Ctx*>(c);
      return x->t[((size_t)m * x->lmax + l) * x->Kpad4 + k];
    }, &cx, p->d_pct_frag, &p->s_pct_frag);
    if (r != SDY_OK) { sdy_sht_plan_destroy(p); return r; }
    if (sdy_leg_par_supported(nlat, lmax)) {
      e = hipMalloc(&p->d_wq_par, sdy_leg_par_table_bytes(mtr));
      if (e == hipSuccess) e = hipMalloc(&p->d_pct_par, sdy_leg_par_table_bytes(mtr));
      if (e != hipSuccess) { sdy_sht_plan_destroy(p); return (int)e; }
      cx.t = wqT.data();
      r = sdy_leg_par_pack(mtr, nlat, lmax, 1, [](void* c, int m, int l, int k) {
        const Ctx* x = static_cast<const Ctx*>(c);
        return x->t[((size_t)m * x->nlat + k) * x->Lpad4 + l];
      }, &cx, p->d_wq_par, &p->s_wq_par);
      cx.t = pf.data();
      if (r == SDY_OK) r = sdy_leg_par_pack(mtr, nlat, lmax, 0, [](void* c, int m, int k, int l) {
        const Ctx* x = static_cast<const Ctx*>(c);
        return x->t[((size_t)m * x->lmax + l) * x->Kpad4 + k];
      }, &cx, p->d_pct_par, &p->s_pct_par);
      if (r != SDY_OK) { sdy_sht_plan_destroy(p); return r; }
      // polar cut-off tables
      std::vector<int> kdead(mtr, 0), mcut(nlat, mtr);
      const int Kh = nlat / 2;
      for (int m = 0; m < mtr; ++m) {
        float mxa = 0.f, mxs = 0.f;
        for (int k = 0; k < nlat; ++k)
          for (int l = m; l < lmax; ++l) {
            mxa = std::fmax(mxa, std::fabs(wqT[((size_t)m * nlat + k) * p->Lpad4 + l]));
            mxs = std::fmax(mxs, std::fabs(pf[((size_t)m * lmax + l) * p->Kpad4 + k]));
          }
        int kd = 0;
        for (; kd < Kh; ++kd) {
          float ra = 0.f, rs = 0.f;
          for (int kk : {kd, nlat - 1 - kd})
            for (int l = m; l < lmax; ++l) {
              ra = std::fmax(ra, std::fabs(wqT[((size_t)m * nlat + kk) * p->Lpad4 + l]));
              rs = std::fmax(rs, std::fabs(pf[((size_t)m * lmax + l) * p->Kpad4 + kk]));
            }
          if (!(ra < 1e-12f * mxa && rs < 1e-12f * mxs)) break;
        }
        kdead[m] = kd;
      }
      for (int m = mtr - 2; m >= 0; --m) kdead[m] = std::min(kdead[m], kdead[m + 1]);   // non-decreasing in m (conservative)
      for (int k = 0; k < nlat; ++k) {
        const int kp = std::min(k, nlat - 1 - k);
        int c = 0;
        while (c < mtr && kdead[c] <= kp) ++c;
        mcut[k] = c;
      }
      e = hipMalloc((void**)&p->d_kdead, mtr * sizeof(int));
      if (e == hipSuccess) e = hipMalloc((void**)&p->d_mcut, nlat * sizeof(int));
      if (e == hipSuccess) e = hipMemcpy(p->d_kdead, kdead.data(), mtr * sizeof(int), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(p->d_mcut, mcut.data(), nlat * sizeof(int), hipMemcpyHostToDevice);
      if (e != hipSuccess) { sdy_sht_plan_destroy(p); return (int)e; }
    }
  }
  p->fft.tw = p->d_tw;
  p->fft.pw = p->d_pw;
  *out = p;
  return SDY_OK;
}

extern "C" void sdy_sht_plan_destroy(sdy_sht_plan* p) {
  if (!p) return;
  if (p->d_wqT) (void)hipFree(p->d_wqT);
  if (p->d_pct) (void)hipFree(p->d_pct);
  if (p->d_tw) (void)hipFree(p->d_tw);
  if (p->d_pw) (void)hipFree(p->d_pw);
  if (p->d_wq_h3) (void)hipFree(p->d_wq_h3);
  if (p->d_pct_h3) (void)hipFree(p->d_pct_h3);
  if (p->d_kdead) (void)hipFree(p->d_kdead);
  if (p->d_mcut) (void)hipFree(p->d_mcut);
  if (p->d_wq_par) (void)hipFree(p->d_wq_par);
  if (p->d_pct_par) (void)hipFree(p->d_pct_par);
  if (p->d_wq_frag) (void)hipFree(p->d_wq_frag);
  if (p->d_pct_frag) (void)hipFree(p->d_pct_frag);
  delete p;
}

extern "C" int sdy_sht_plan_dims(const sdy_sht_plan* p, int dims[6]) {
  if (!p || !dims) return SDY_ERR_ARG;
  dims[0] = p->nlat; dims[1] = p->nlon; dims[2] = p->lmax; dims[3] = p->mmax; dims[4] = p->mtr; dims[5] = p->grid;
  return SDY_OK;
}

static inline size_t xf_floats(const sdy_sht_plan* p, int B, int C) { return (size_t)p->mtr * p->nlat * B * 2 * C; }
static inline size_t cs_floats(const sdy_sht_plan* p, int B, int C) { return (size_t)p->lmax * p->mtr * B * 2 * C; }

extern "C" size_t sdy_sht_workspace_floats(const sdy_sht_plan* p, int B, int C) {
  if (!p || B <= 0 || C <= 0) return 0;
  return round_up_sz(xf_floats(p, B, C), 64) + round_up_sz(cs_floats(p, B, C), 64);
}

extern "C" int sdy_rfft_lon(const sdy_sht_plan* p, const float* x, const float* a, const float* d, float* xn_out,
                            float* Xf, int B, int C, void* stream) {
  if (!p || !x || !Xf || B <= 0 || C <= 0) return SDY_ERR_ARG;
  if ((a == nullptr) != (d == nullptr)) return SDY_ERR_ARG;
  return sdy_fft_launch_fwd(p->fft, x, a, d, xn_out, Xf, B, C, p->nlat, p->mtr, 0, nullptr, (hipStream_t)stream);
}

extern "C" int sdy_irfft_lon(const sdy_sht_plan* p, const float* Yf, const float* bias, float* y, int B, int C,
                             void* stream) {
  if (!p || !Yf || !y || B <= 0 || C <= 0) return SDY_ERR_ARG;
  return sdy_fft_launch_inv(p->fft, Yf, bias, y, B, C, p->nlat, p->mtr, 0, nullptr, (hipStream_t)stream);
}

static int legendre_fwd_impl(const sdy_sht_plan* p, const float* Xf, float* Cs, int B, int C, bool polar, void* stream,
                             bool tiled = false, bool cs_tiled = false);
extern "C" int sdy_legendre_fwd(const sdy_sht_plan* p, const float* Xf, float* Cs, int B, int C, void* stream) {
  return legendre_fwd_impl(p, Xf, Cs, B, C, false, stream);
}
// polar: skip the rows / orders of the polar cut-off (only valid when the producer / consumer of Xf is fft360 with the same
// cut-off: plan_polar_ok)
static bool plan_polar_ok(const sdy_sht_plan* p, int C) {
  const bool off = sw().no_fft360 || sw().no_leg_par || sw().no_leg_frag;
  return !off && p->d_kdead && p->d_wq_par && p->fft.n == 180 && C % 16 == 0;
}
// Tile-major grid-frequency tensor (fft.h, ilv == 2): like the polar cut-off a contract between fft360 and leg_par only --
// the plane of one order is stored as whole 64-column tiles, so a Legendre workgroup's activation tile is one contiguous
// block.
static bool plan_tiled_ok(const sdy_sht_plan* p, int C, int ilv) {
  const bool off = sw().no_fft360 || sw().no_leg_par || sw().no_leg_frag;
  return !off && ilv == 1 && p->d_wq_par && p->d_pct_par && p->fft.n == 180 && C % 32 == 0;
}
static int legendre_fwd_impl(const sdy_sht_plan* p, const float* Xf, float* Cs, int B, int C, bool polar, void* stream,
                             bool tiled, bool cs_tiled) {
  if (!p || !Xf || !Cs || B <= 0 || C <= 0) return SDY_ERR_ARG;
  if (C & 1) return SDY_ERR_ALIGN;
  const int N = 2 * B * C;
  GemmParams g{};
  g.A = p->d_wqT; g.lda = p->Lpad4; g.sA = (long)p->nlat * p->Lpad4;
  g.B = Xf; g.ldb = N; g.sB = (long)p->nlat * N;
  g.C = Cs; g.ldc = p->mtr * N; g.sC = N;
  g.M = p->Lpad4; g.M_store = p->lmax; g.N = N; g.K = p->nlat; g.nbatch = p->mtr;
  g.tri_mode = SDY_TRI_LEG_FWD; g.tile = SDY_TILE_64x128;
  const bool no_frag = sw().no_leg_frag, no_par = sw().no_leg_par;
  if (p->d_wq_par && !no_frag && !no_par)
    // cs_tiled: the coefficients TILE-MAJOR by order, [m][column tile][l][64] (dh_h3.hip, DhParams::tiled)
    return sdy_leg_par_launch(p->d_wq_par, p->s_wq_par, p->mtr, Xf, tiled ? 64 : N, (long)p->nlat * N, Cs,
                              cs_tiled ? 64L : (long)p->mtr * N, cs_tiled ? (long)(N / 64) * p->lmax * 64 : (long)N,
                              p->lmax, p->nlat, N, 1, polar ? p->d_kdead : nullptr, (hipStream_t)stream,
                              tiled ? 64L * p->nlat : 0L, cs_tiled ? 64L * p->lmax : 0L);
  if (tiled || cs_tiled) return SDY_ERR_UNSUPPORTED;
  if (p->d_wq_frag && !no_frag)
    return sdy_leg_h3_launch(p->d_wq_frag, p->s_wq_frag, p->mtr, Xf, N, (long)p->nlat * N, Cs, (long)p->mtr * N, N, p->lmax,
                             p->nlat, N, SDY_TRI_LEG_FWD, (hipStream_t)stream);
  if (p->gemm_mode == 1)
    return sdy_gemm_h3_launch(g, p->d_wq_h3, p->h3_rows_fwd, p->h3_k_fwd, (long)p->h3_rows_fwd * p->h3_k_fwd,
                              (long)p->mtr * p->h3_rows_fwd * p->h3_k_fwd, p->s_wq, 0, (hipStream_t)stream);
  return sdy_gemm_launch(g, (hipStream_t)stream);
}

static int legendre_inv_impl(const sdy_sht_plan* p, const float* Cs, float* Yf, int B, int C, bool polar, void* stream,
                             bool tiled = false, bool cs_tiled = false);
extern "C" int sdy_legendre_inv(const sdy_sht_plan* p, const float* Cs, float* Yf, int B, int C, void* stream) {
  return legendre_inv_impl(p, Cs, Yf, B, C, false, stream);
}
static int legendre_inv_impl(const sdy_sht_plan* p, const float* Cs, float* Yf, int B, int C, bool polar, void* stream,
                             bool tiled, bool cs_tiled) {
  if (!p || !Cs || !Yf || B <= 0 || C <= 0) return SDY_ERR_ARG;
  if (C & 1) return SDY_ERR_ALIGN;
  const int N = 2 * B * C;
  GemmParams g{};
  g.A = p->d_pct; g.lda = p->Kpad4; g.sA = (long)p->lmax * p->Kpad4;
  g.B = Cs; g.ldb = p->mtr * N; g.sB = N;
  g.C = Yf; g.ldc = N; g.sC = (long)p->nlat * N;
  g.M = p->Kpad4; g.M_store = p->nlat; g.N = N; g.K = p->lmax; g.nbatch = p->mtr;
  g.tri_mode = SDY_TRI_LEG_INV; g.tile = SDY_TILE_64x128;
  const bool no_frag = sw().no_leg_frag, no_par = sw().no_leg_par;
  if (p->d_pct_par && !no_frag && !no_par)
    return sdy_leg_par_launch(p->d_pct_par, p->s_pct_par, p->mtr, Cs, cs_tiled ? 64L : (long)p->mtr * N,
                              cs_tiled ? (long)(N / 64) * p->lmax * 64 : (long)N, Yf, tiled ? 64 : N,
                              (long)p->nlat * N, p->nlat, p->lmax, N, 0, polar ? p->d_kdead : nullptr, (hipStream_t)stream,
                              cs_tiled ? 64L * p->lmax : 0L, tiled ? 64L * p->nlat : 0L);
  if (tiled || cs_tiled) return SDY_ERR_UNSUPPORTED;
  if (p->d_pct_frag && !no_frag)
    return sdy_leg_h3_launch(p->d_pct_frag, p->s_pct_frag, p->mtr, Cs, (long)p->mtr * N, N, Yf, N, (long)p->nlat * N, p->nlat,
                             p->lmax, N, SDY_TRI_LEG_INV, (hipStream_t)stream);
  if (p->gemm_mode == 1)
    return sdy_gemm_h3_launch(g, p->d_pct_h3, p->h3_rows_inv, p->h3_k_inv, (long)p->h3_rows_inv * p->h3_k_inv,
                              (long)p->mtr * p->h3_rows_inv * p->h3_k_inv, p->s_pct, 0, (hipStream_t)stream);
  return sdy_gemm_launch(g, (hipStream_t)stream);
}

extern "C" int sdy_sht_forward(const sdy_sht_plan* p, const float* x, float* out_c64, int B, int C, float* ws,
                               size_t ws_floats, void* stream) {
  if (!p || !x || !out_c64 || !ws) return SDY_ERR_ARG;
  if (ws_floats < sdy_sht_workspace_floats(p, B, C)) return SDY_ERR_WORKSPACE;
  float* Xf = ws;
  float* Cs = ws + round_up_sz(xf_floats(p, B, C), 64);
  SDY_TRY(sdy_rfft_lon(p, x, nullptr, nullptr, nullptr, Xf, B, C, stream));
  SDY_TRY(sdy_legendre_fwd(p, Xf, Cs, B, C, stream));
  return sdy_spec_to_torch_launch(Cs, out_c64, B, C, p->lmax, p->mtr, p->mmax, (hipStream_t)stream);
}

extern "C" int sdy_sht_inverse(const sdy_sht_plan* p, const float* in_c64, float* y, int B, int C, float* ws,
                               size_t ws_floats, void* stream) {
  if (!p || !in_c64 || !y || !ws) return SDY_ERR_ARG;
  if (ws_floats < sdy_sht_workspace_floats(p, B, C)) return SDY_ERR_WORKSPACE;
  float* Yf = ws;
  float* Cs = ws + round_up_sz(xf_floats(p, B, C), 64);
  SDY_TRY(sdy_torch_to_spec_launch(in_c64, Cs, B, C, p->lmax, p->mtr, p->mmax, (hipStream_t)stream));
  SDY_TRY(sdy_legendre_inv(p, Cs, Yf, B, C, stream));
  return sdy_irfft_lon(p, Yf, nullptr, y, B, C, stream);
}

// =========================================================================================================
// dhconv
// =========================================================================================================
extern "C" int sdy_dhconv_pack_weight(const float* w_host, int Ci, int Co, int L, float* w_packed_dev, void* stream) {
  if (!w_host || !w_packed_dev || Ci <= 0 || Co <= 0 || L <= 0) return SDY_ERR_ARG;
  // (Ci, Co, L, 2) -> [l][2][Ci][Co]
  std::vector<float> packed((size_t)L * 2 * Ci * Co);
  for (int i = 0; i < Ci; ++i)
    for (int o = 0; o < Co; ++o)
      for (int l = 0; l < L; ++l) {
        const float* s = w_host + (((size_t)i * Co + o) * L + l) * 2;
        packed[(((size_t)l * 2 + 0) * Ci + i) * Co + o] = s[0];
        packed[(((size_t)l * 2 + 1) * Ci + i) * Co + o] = s[1];
      }
  (void)stream;
  SDY_HIP_TRY(hipMemcpy(w_packed_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
  return SDY_OK;
}

extern "C" int sdy_dhconv(const float* Cs_in, const float* w_packed, float* Cs_out, int L, int mtr, int B, int Ci,
                          int Co, void* stream) {
  if (!Cs_in || !w_packed || !Cs_out || L <= 0 || mtr <= 0 || B <= 0 || Ci <= 0 || Co <= 0) return SDY_ERR_ARG;
  if ((Ci & 3) || (Co & 3)) return SDY_ERR_ALIGN;
  GemmParams g{};
  g.A = Cs_in; g.a_kcontig = 1; g.lda = 2 * Ci; g.sA = (long)mtr * B * 2 * Ci;
  g.B = w_packed; g.b_cplx = 1; g.cplx_Ei = Ci; g.cplx_Eo = Co; g.ldb = 2 * Co; g.sB = (long)2 * Ci * Co;
  g.C = Cs_out; g.ldc = 2 * Co; g.sC = (long)mtr * B * 2 * Co;
  g.M = mtr * B; g.M_store = mtr * B; g.N = 2 * Co; g.K = 2 * Ci; g.nbatch = L;
  g.tri_mode = SDY_TRI_DHCONV; g.tri_B = B; g.tile = SDY_TILE_64x128;
  return sdy_gemm_launch(g, (hipStream_t)stream);
}

static inline int dh_npad(int Co) { return round_up(2 * Co, 128); }
static inline int dh_kpad(int Ci) { return round_up(2 * Ci, 64); }

extern "C" size_t sdy_dhconv_h3_pack_bytes(int Ci, int Co, int L) {
  if (Ci <= 0 || Co <= 0 || L <= 0) return 0;
  return (size_t)2 * L * dh_npad(Co) * dh_kpad(Ci) * sizeof(_Float16);
}

// (Ci, Co, L, 2) -> per l the TRANSPOSED expanded real matrix W'^T[o'][i'], W' = [[wr, wi], [-wi, wr]] with rows
// i' = (ri_in, i) and columns o' = (ri_out, o), split into fp16 hi | lo
// ilv = 1: rows / columns in the [c / 16][ri][16] order of the fused forward's spectral buffers (fft.h)
static int dhconv_h3_pack(const float* w, int Ci, int Co, int L, void* packed_dev, float* scale, int ilv) {
  if (!w || !packed_dev || !scale || Ci <= 0 || Co <= 0 || L <= 0) return SDY_ERR_ARG;
  if (ilv && ((Ci & 15) || (Co & 15))) return SDY_ERR_UNSUPPORTED;
  std::vector<_Float16> buf;
  *scale = h3_pack_host(buf, L, 2 * Co, 2 * Ci, dh_npad(Co), dh_kpad(Ci), [&](int l, int op, int ip) {
    const int ro = ilv ? (op >> 4) & 1 : op >= Co, o = ilv ? (op >> 5) * 16 + (op & 15) : op - ro * Co;
    const int ri = ilv ? (ip >> 4) & 1 : ip >= Ci, i = ilv ? (ip >> 5) * 16 + (ip & 15) : ip - ri * Ci;
    const float* e = w + (((size_t)i * Co + o) * L + l) * 2;
    if (ri == ro) return e[0];
    return ri ? -e[1] : e[1];
  });
  SDY_HIP_TRY(hipMemcpy(packed_dev, buf.data(), buf.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  return SDY_OK;
}
extern "C" int sdy_dhconv_h3_pack_weight(const float* w, int Ci, int Co, int L, void* packed_dev, float* scale) {
  return dhconv_h3_pack(w, Ci, Co, L, packed_dev, scale, 0);
}

extern "C" int sdy_dhconv_h3(const float* Cs_in, const void* packed, float scale, float* Cs_out, int L, int mtr, int B,
                             int Ci, int Co, void* stream) {
  if (!Cs_in || !packed || !Cs_out || L <= 0 || mtr <= 0 || B <= 0 || Ci <= 0 || Co <= 0) return SDY_ERR_ARG;
  if ((Ci & 3) || (Co & 3)) return SDY_ERR_ALIGN;
  GemmParams g{};
  g.A = Cs_in; g.lda = 2 * Ci; g.sA = (long)mtr * B * 2 * Ci;
  g.C = Cs_out; g.ldc = 2 * Co; g.sC = (long)mtr * B * 2 * Co;
  g.M = mtr * B; g.M_store = mtr * B; g.N = 2 * Co; g.K = 2 * Ci; g.nbatch = L;
  g.tri_mode = SDY_TRI_DHCONV; g.tri_B = B;
  const int Np = dh_npad(Co), Kp = dh_kpad(Ci);
  return sdy_gemm_h3_launch(g, packed, Np, Kp, (long)Np * Kp, (long)L * Np * Kp, scale, 1, (hipStream_t)stream);
}

// =========================================================================================================
// InstanceNorm coefficients, 1x1 conv, sampler arithmetic
// =========================================================================================================
extern "C" int sdy_instnorm_coeffs(const float* x, int B, int C, int HW, const float* gamma, const float* beta,
                                   const float* scale_shift, long ss_stride, float eps, float* a, float* d,
                                   void* stream) {
  return sdy_instnorm_coeffs_launch(x, B, C, HW, gamma, beta, scale_shift, ss_stride, eps, a, d, (hipStream_t)stream);
}

static inline int h3_mpad(int Cout) { return Cout > 128 ? round_up(Cout, 256) : 128; }
static inline int h3_kpad(int Cin) { return round_up(Cin, 64); }

extern "C" size_t sdy_h3_pack_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  return (size_t)2 * h3_mpad(Cout) * h3_kpad(Cin) * sizeof(_Float16);
}

extern "C" int sdy_h3_pack_weight(const float* w, int Cout, int Cin, void* packed_dev, float* scale) {
  if (!w || !packed_dev || !scale || Cout <= 0 || Cin <= 0) return SDY_ERR_ARG;
  std::vector<_Float16> buf;
  *scale = h3_pack_host(buf, 1, Cout, Cin, h3_mpad(Cout), h3_kpad(Cin),
                        [&](int, int o, int i) { return w[(size_t)o * Cin + i]; });
  SDY_HIP_TRY(hipMemcpy(packed_dev, buf.data(), buf.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  return SDY_OK;
}

extern "C" int sdy_conv1x1(const sdy_conv_args* a, void* stream) {
  if (!a || !a->x || (!a->wt && !a->w_h3 && !a->w_frag) || !a->out) return SDY_ERR_ARG;
  if (a->B <= 0 || a->Cin <= 0 || a->Cout <= 0 || a->HW <= 0) return SDY_ERR_ARG;
  if (!a->w_h3 && a->ldw < a->Cout) return SDY_ERR_ARG;
  if ((a->pa == nullptr) != (a->pd == nullptr)) return SDY_ERR_ARG;
  if (a->add_mode != 0 && !a->add) return SDY_ERR_ARG;
  if ((!a->w_h3 && (a->ldw & 3)) || (a->HW & 3) || (a->x_bstride & 3) || (a->out_bstride & 3)) return SDY_ERR_ALIGN;
  if (a->drop_p < 0.0f || a->drop_p >= 1.0f) return SDY_ERR_ARG;
  if (a->w_frag && sdy_conv256_h3_supported(a->Cin, a->Cout) && a->drop_p == 0.0f && !a->keep_mask && !a->batch_scale)
    return sdy_conv256_h3_launch(a, (hipStream_t)stream);
  // statistics, the tile-major output and the image map exist in the persistent kernel only
  if (a->stats || a->out_tiled || a->x_rows) return SDY_ERR_UNSUPPORTED;
  GemmParams g{};
  g.A = a->wt; g.lda = a->ldw; g.sA = 0;
  g.B = a->x; g.ldb = a->HW; g.sB = a->x_bstride;
  g.C = a->out; g.ldc = a->HW; g.sC = a->out_bstride;
  g.M = round_up(a->Cout, 4); g.M_store = a->Cout; g.N = a->HW; g.K = a->Cin; g.nbatch = a->B;
  g.tile = a->Cout > 64 ? SDY_TILE_128x128 : SDY_TILE_64x128;
  g.tag = (a->kernel_tag >= 1 && a->kernel_tag <= 3) ? a->kernel_tag : 0;
  g.pa = a->pa; g.pd = a->pd; g.p_bstride = a->Cin;
  g.bias = a->bias;
  g.add = a->add_mode ? a->add : nullptr; g.sAdd = a->add_bstride; g.ldadd = a->HW; g.add_mode = a->add_mode;
  g.act = a->act;
  if (a->drop_p > 0.0f) {
    g.drop_thr = sdy_drop_threshold16(a->drop_p);
    if (g.drop_thr == 0u) g.drop_thr = 1u;
    g.drop_scale = 1.0f / (1.0f - a->drop_p);
    g.keep_mask = a->keep_mask;
  }
  g.seed_lo = (uint32_t)(a->seed & 0xFFFFFFFFu); g.seed_hi = (uint32_t)(a->seed >> 32);
  g.stream_id = a->stream_id; g.call = a->call; g.batch_offset = a->batch_offset;
  if (a->rows_per_call < 0 || (a->rows_per_call > 0 && a->B % a->rows_per_call)) return SDY_ERR_ARG;
  g.rows_per_call = a->rows_per_call > 0 ? a->rows_per_call : a->B;
  g.batch_scale = a->batch_scale;
  if (a->w_h3) {
    const int Mp = h3_mpad(a->Cout), Kp = h3_kpad(a->Cin);
    return sdy_gemm_h3_launch(g, a->w_h3, Mp, Kp, 0, (long)Mp * Kp, a->w_h3_scale, 0, (hipStream_t)stream);
  }
  return sdy_gemm_launch(g, (hipStream_t)stream);
}

extern "C" int sdy_cold_update(const float* x_s, const float* x_ip_next, const float* x_ip_s, float* out, size_t n,
                               void* stream) {
  return sdy_cold_update_launch(x_s, x_ip_next, x_ip_s, out, n, (hipStream_t)stream);
}

extern "C" int sdy_concat_channels(const float* const* src, const int* chans, int nsrc, float* out, int B, int HW,
                                   void* stream) {
  if (!src || !chans || nsrc < 1 || nsrc > 4) return SDY_ERR_ARG;
  long total = 0;
  for (int i = 0; i < nsrc; ++i) total += chans[i];
  return sdy_concat_launch(src, chans, nsrc, out, total * HW, B, HW, (hipStream_t)stream);
}

// =========================================================================================================
// SFNO network object
// =========================================================================================================
namespace {

struct DevBuf {
  float* p = nullptr;
  size_t n = 0;
  bool set = false;
  // split-fp16 copy of a conv weight (gemm_mode 1)
  void* h3 = nullptr;
  float h3_scale = 1.0f;
  // fragment stream of the persistent 256 -> 256 kernel (conv_h3.hip)
  void* frag = nullptr;
  float frag_scale = 1.0f;
};

struct BlockW {
  DevBuf n0w, n0b, n1w, n1b;  // [E]
  DevBuf tw, tb;              // time_mlp: staged into the shared wbt/bb arrays (flags only)
  DevBuf fw;                  // dhconv packed [l][2][E][E]
  DevBuf fb;                  // [E]
  DevBuf skw, skb;            // inner skip: [E][E] transposed, [E]
  DevBuf w1, b1, w2, b2;      // MLP: [E][hid], [hid], [hid][E], [E]
  // fused MLP kernel (mlp_h3.hip): host copies until both weights are known, then one packed device stream
  std::vector<float> w1_host, w2_host;
  void* mlp = nullptr;
  float mlp_s1 = 1.0f, mlp_s2 = 1.0f;
  // Inner skip FOLDED into the dhconv (grid-changing blocks only, see skip_foldable): host copies of the filter weight, the
  // skip weight and the two biases until all are known; then `fw.frag` holds W_dh[l] + W_skip and `fbs` = filter bias + skip bias
  std::vector<float> fw_host, skw_host, fb_host, skb_host;
  bool skip_folded = false;
  DevBuf fbs;
};

}  // namespace

// Channel order of the forward's spectral buffers (fft.h): the 128-byte-line order whenever the dhconv weights can be
// packed for it (split-fp16 GEMM path, E % 16 == 0).
static int spec_ilv(const sdy_sfno_config& c) { return (c.gemm_mode == 1 && c.embed_dim % 16 == 0) ? 1 : 0; }

// Folding the inner skip into the dhconv.  In a grid-changing block (the first / last block of an equiangular data grid) the
// residual the inner skip reads is ITSELF an inverse transform of the block's coefficients, residual = ISHT(Cs)
// (s2convolutions.py:79-83,165-168), and a 1 x 1 convolution mixes channels at every pixel, so it commutes with the transform:
//     ISHT(W_dh[l] Cs) + b_f + W_s ISHT(Cs) + b_s  =  ISHT((W_dh[l] + W_s) Cs) + (b_f + b_s)        (sfnonet.py:303-311)
// -- the 256 -> 256 convolution of those blocks (8.5 GF and three tensors of traffic each) costs one matrix addition at load
// time, and `x = act(x + inner_skip(residual))` becomes a GELU pass over the inverse FFT's output.  Exact in exact arithmetic
// (the two transforms are the same linear map, polar cut-off included); in fp32 the rounding differs at the 1e-7 level.  Only
// on the dh_h3 path (split-fp16 mode, 256 channels); every other configuration runs the convolution as the reference does.
static bool skip_foldable(const sdy_sfno_config& c, int i) {
  const bool first = i == 0, last = i == c.num_layers - 1;
  return c.gemm_mode == 1 && c.data_grid != SDY_GRID_LEGENDRE_GAUSS && first != last && !sw().no_skip_fold;
}

struct sdy_sfno {
  sdy_sfno_config cfg;
  int HW, catC, decC, ldo;
  sdy_sht_plan* plan_data = nullptr;  // data grid (first forward / last inverse)
  sdy_sht_plan* plan_lg = nullptr;    // legendre-gauss (inner transforms); may alias plan_data
  DevBuf pos, e0w, e0b, e2w, d0w, d0b, d2w;
  // encoder / decoder as ONE launch each (pair_h3.hip): host copies until both weights of a pair are known
  struct Pair { std::vector<float> w1_host, w2_host; void* w = nullptr; float s1 = 1.0f, s2 = 1.0f; } enc, dec;
  DevBuf t1w, t1b, t3w, t3b, freq, wbt, bb;
  std::vector<BlockW> blk;
  SdyTimeMlp tm;
  std::string missing;
  // sdy_sfno_fwd_args.reuse_encoder: where (and for which batch) the last forward left its encoder output
  const float* enc_ws = nullptr;
  int enc_B = 0;
  bool enc_has_stats = false;
};

static int dev_alloc(DevBuf& b, size_t n) {
  if (b.p && b.n == n) return SDY_OK;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  SDY_HIP_TRY(hipMalloc((void**)&b.p, n * sizeof(float)));
  b.n = n;
  return SDY_OK;
}
static int dev_upload(DevBuf& b, const float* host, size_t n) {
  SDY_TRY(dev_alloc(b, n));
  SDY_HIP_TRY(hipMemcpy(b.p, host, n * sizeof(float), hipMemcpyHostToDevice));
  b.set = true;
  return SDY_OK;
}
// host (rows=out, cols=in) row-major -> dev [in][ld] (ld >= out, zero padded)
static int dev_upload_T(DevBuf& b, const float* host, int out, int in, int ld) {
  std::vector<float> t((size_t)in * ld, 0.0f);
  for (int o = 0; o < out; ++o)
    for (int i = 0; i < in; ++i) t[(size_t)i * ld + o] = host[(size_t)o * in + i];
  return dev_upload(b, t.data(), t.size());
}
static void dev_free(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  if (b.h3) (void)hipFree(b.h3);
  if (b.frag) (void)hipFree(b.frag);
  b.p = nullptr;
  b.h3 = nullptr;
  b.frag = nullptr;
  b.set = false;
}
// conv weight: fp32 transposed copy (gemm_mode 0 path, also kept for reference) + split-fp16 pack (gemm_mode 1)
static int dev_upload_conv(DevBuf& b, const float* host, int out, int in, int ld, bool want_h3) {
  SDY_TRY(dev_upload_T(b, host, out, in, ld));
  if (want_h3) {
    if (b.h3) (void)hipFree(b.h3);
    b.h3 = nullptr;
    SDY_HIP_TRY(hipMalloc(&b.h3, sdy_h3_pack_bytes(out, in)));
    SDY_TRY(sdy_h3_pack_weight(host, out, in, b.h3, &b.h3_scale));
    if (sdy_conv256_h3_supported(in, out)) {
      if (b.frag) (void)hipFree(b.frag);
      b.frag = nullptr;
      SDY_HIP_TRY(hipMalloc(&b.frag, sdy_conv256_h3_pack_bytes_cin(in)));
      SDY_TRY(sdy_conv256_h3_pack_cin(host, in, b.frag, &b.frag_scale));
    }
  }
  return SDY_OK;
}

extern "C" int sdy_sfno_create(const sdy_sfno_config* c, sdy_sfno** out) {
  if (!c || !out) return SDY_ERR_ARG;
  if (c->nlat < 2 || c->nlon < 4 || c->in_chans < 1 || c->out_chans < 1 || c->embed_dim < 4 || c->num_layers < 1 ||
      c->num_layers > 32 || c->mlp_hidden < 4)
    return SDY_ERR_ARG;
  if ((c->embed_dim & 3) || (c->mlp_hidden & 3) || ((c->nlat * c->nlon) & 3)) return SDY_ERR_ALIGN;
  if (c->with_time_emb && (c->time_dim < 4 || (c->embed_dim & 1))) return SDY_ERR_ARG;
  if (c->dropout_mlp < 0.f || c->dropout_mlp >= 1.f || c->drop_path_rate < 0.f || c->drop_path_rate >= 1.f)
    return SDY_ERR_ARG;
  sdy_sfno* n = new sdy_sfno();
  n->cfg = *c;
  n->HW = c->nlat * c->nlon;
  n->catC = c->embed_dim + c->in_chans;                      // [ last block output | concatenated inputs ]
  n->decC = c->embed_dim + (c->big_skip ? c->in_chans : 0);  // what the decoder reads (sfnonet.py:736)
  n->ldo = round_up(c->out_chans, 4);
  n->blk.resize(c->num_layers);
  if (c->gemm_mode != 0 && c->gemm_mode != 1) { delete n; return SDY_ERR_ARG; }
  int r = sdy_sht_plan_create_ex(c->nlat, c->nlon, c->lmax, c->mmax, c->data_grid, c->gemm_mode, &n->plan_data);
  if (r != SDY_OK) { delete n; return r; }
  if (c->data_grid == SDY_GRID_LEGENDRE_GAUSS) {
    n->plan_lg = n->plan_data;
  } else {
    r = sdy_sht_plan_create_ex(c->nlat, c->nlon, c->lmax, c->mmax, SDY_GRID_LEGENDRE_GAUSS, c->gemm_mode, &n->plan_lg);
    if (r != SDY_OK) { sdy_sfno_destroy(n); return r; }
  }
  // defaults that a caller may override with "@time_freq" / "@drop_path_rates"
  std::memset(&n->tm, 0, sizeof(n->tm));
  n->tm.num_layers = c->num_layers;
  for (int i = 0; i < c->num_layers; ++i) {
    const float rate = c->num_layers > 1 ? c->drop_path_rate * (float)i / (float)(c->num_layers - 1) : 0.0f;
    n->tm.dp_rate[i] = rate;
    n->tm.dp_thr[i] = sdy_drop_threshold(rate);
  }
  if (c->with_time_emb) {
    const int E = c->embed_dim, T = c->time_dim, L = c->num_layers, half = E / 2;
    std::vector<float> f(half);
    const float e = (float)(-std::log(10000.0) / (half - 1));
    for (int i = 0; i < half; ++i) f[i] = expf((float)i * e);
    r = dev_upload(n->freq, f.data(), half);
    if (r == SDY_OK) r = dev_alloc(n->wbt, (size_t)L * T * 2 * E);
    if (r == SDY_OK) r = dev_alloc(n->bb, (size_t)L * 2 * E);
    if (r != SDY_OK) { sdy_sfno_destroy(n); return r; }
  }
  *out = n;
  return SDY_OK;
}

extern "C" void sdy_sfno_destroy(sdy_sfno* n) {
  if (!n) return;
  if (n->plan_lg && n->plan_lg != n->plan_data) sdy_sht_plan_destroy(n->plan_lg);
  if (n->plan_data) sdy_sht_plan_destroy(n->plan_data);
  DevBuf* top[] = {&n->pos, &n->e0w, &n->e0b, &n->e2w, &n->d0w, &n->d0b, &n->d2w, &n->t1w, &n->t1b,
                   &n->t3w, &n->t3b, &n->freq, &n->wbt, &n->bb};
  for (DevBuf* b : top) dev_free(*b);
  for (BlockW& w : n->blk) {
    DevBuf* bs[] = {&w.n0w, &w.n0b, &w.n1w, &w.n1b, &w.fw, &w.fb, &w.skw, &w.skb, &w.w1, &w.b1, &w.w2, &w.b2, &w.fbs};
    for (DevBuf* b : bs) dev_free(*b);
    if (w.mlp) (void)hipFree(w.mlp);
  }
  if (n->enc.w) (void)hipFree(n->enc.w);
  if (n->dec.w) (void)hipFree(n->dec.w);
  delete n;
}

#define EXPECT_NUMEL(n_expected)                      \
  do {                                                \
    if (numel != (size_t)(n_expected)) return SDY_ERR_SHAPE; \
  } while (0)

extern "C" int sdy_sfno_set_param(sdy_sfno* n, const char* name_c, const float* host, size_t numel) {
  if (!n || !name_c || !host) return SDY_ERR_ARG;
  const sdy_sfno_config& c = n->cfg;
  const int E = c.embed_dim, T = c.time_dim, L = c.num_layers, H = c.mlp_hidden, Cin = c.in_chans;
  const std::string name(name_c);
  const bool h3 = c.gemm_mode == 1;
  n->enc_ws = nullptr;   // new weights: a stored encoder output (sdy_sfno_fwd_args.reuse_encoder) is no longer this network's
  // non-persistent SHT buffers of older torch-harmonics releases: accepted, ignored
  if (name.find("trans") != std::string::npos && (name.find(".weights") != std::string::npos ||
                                                   name.find(".pct") != std::string::npos))
    return SDY_OK;
  if (name == "@time_freq") { EXPECT_NUMEL(E / 2); return dev_upload(n->freq, host, numel); }
  if (name == "@drop_path_rates") {
    EXPECT_NUMEL(L);
    for (int i = 0; i < L; ++i) {
      if (host[i] < 0.f || host[i] >= 1.f) return SDY_ERR_ARG;
      n->tm.dp_rate[i] = host[i];
      n->tm.dp_thr[i] = sdy_drop_threshold(host[i]);
    }
    return SDY_OK;
  }
  if (name == "pos_embed") { EXPECT_NUMEL((size_t)E * n->HW); return dev_upload(n->pos, host, numel); }
  // (the pair kernel's weight stream needs both matrices of a pair: packed when the second one arrives)
  auto pack_pair = [&](sdy_sfno::Pair& pr, int cin, int cout) -> int {
    if (!h3 || !sdy_pair_h3_supported(cin, E, cout) || pr.w1_host.empty() || pr.w2_host.empty()) return SDY_OK;
    if (!pr.w) SDY_HIP_TRY(hipMalloc(&pr.w, sdy_pair_h3_pack_bytes(cin, E, cout)));
    return sdy_pair_h3_pack(pr.w1_host.data(), pr.w2_host.data(), cin, E, cout, pr.w, &pr.s1, &pr.s2);
  };
  if (name == "encoder.0.weight") {
    EXPECT_NUMEL((size_t)E * Cin);
    n->enc.w1_host.assign(host, host + numel);
    SDY_TRY(pack_pair(n->enc, Cin, E));
    return dev_upload_conv(n->e0w, host, E, Cin, E, h3);
  }
  if (name == "encoder.0.bias") { EXPECT_NUMEL(E); return dev_upload(n->e0b, host, numel); }
  if (name == "encoder.2.weight") {
    EXPECT_NUMEL((size_t)E * E);
    n->enc.w2_host.assign(host, host + numel);
    SDY_TRY(pack_pair(n->enc, Cin, E));
    return dev_upload_conv(n->e2w, host, E, E, E, h3);
  }
  if (name == "decoder.0.weight") {
    EXPECT_NUMEL((size_t)E * n->decC);
    n->dec.w1_host.assign(host, host + numel);
    SDY_TRY(pack_pair(n->dec, n->decC, c.out_chans));
    return dev_upload_conv(n->d0w, host, E, n->decC, E, h3);
  }
  if (name == "decoder.0.bias") { EXPECT_NUMEL(E); return dev_upload(n->d0b, host, numel); }
  if (name == "decoder.2.weight") {
    EXPECT_NUMEL((size_t)c.out_chans * E);
    n->dec.w2_host.assign(host, host + numel);
    SDY_TRY(pack_pair(n->dec, n->decC, c.out_chans));
    return dev_upload_conv(n->d2w, host, c.out_chans, E, n->ldo, h3);
  }
  if (c.with_time_emb) {
    if (name == "time_emb_mlp.1.weight") { EXPECT_NUMEL((size_t)T * E); return dev_upload_T(n->t1w, host, T, E, T); }
    if (name == "time_emb_mlp.1.bias") { EXPECT_NUMEL(T); return dev_upload(n->t1b, host, numel); }
    if (name == "time_emb_mlp.3.weight") { EXPECT_NUMEL((size_t)T * T); return dev_upload_T(n->t3w, host, T, T, T); }
    if (name == "time_emb_mlp.3.bias") { EXPECT_NUMEL(T); return dev_upload(n->t3b, host, numel); }
  }
  if (name.rfind("blocks.", 0) == 0) {
    const size_t dot = name.find('.', 7);
    if (dot == std::string::npos) return SDY_ERR_NAME;
    const int i = std::atoi(name.substr(7, dot - 7).c_str());
    if (i < 0 || i >= L) return SDY_ERR_NAME;
    const std::string rest = name.substr(dot + 1);
    BlockW& w = n->blk[i];
    if (rest == "norm0.weight") { EXPECT_NUMEL(E); return dev_upload(w.n0w, host, numel); }
    if (rest == "norm0.bias") { EXPECT_NUMEL(E); return dev_upload(w.n0b, host, numel); }
    if (rest == "norm1.weight") { EXPECT_NUMEL(E); return dev_upload(w.n1w, host, numel); }
    if (rest == "norm1.bias") { EXPECT_NUMEL(E); return dev_upload(w.n1b, host, numel); }
    if (c.with_time_emb && rest == "time_mlp.1.weight") {
      EXPECT_NUMEL((size_t)2 * E * T);
      std::vector<float> t((size_t)T * 2 * E);
      for (int o = 0; o < 2 * E; ++o)
        for (int k = 0; k < T; ++k) t[(size_t)k * 2 * E + o] = host[(size_t)o * T + k];
      SDY_HIP_TRY(hipMemcpy(n->wbt.p + (size_t)i * T * 2 * E, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
      w.tw.set = true;
      return SDY_OK;
    }
    if (c.with_time_emb && rest == "time_mlp.1.bias") {
      EXPECT_NUMEL((size_t)2 * E);
      SDY_HIP_TRY(hipMemcpy(n->bb.p + (size_t)i * 2 * E, host, numel * sizeof(float), hipMemcpyHostToDevice));
      w.tb.set = true;
      return SDY_OK;
    }
    // (re)packs the dh_h3 stream of a foldable block once the filter weight AND the skip weight are known: W_dh[l] + W_skip
    auto pack_folded = [&]() -> int {
      if (w.fw_host.empty() || w.skw_host.empty()) return SDY_OK;
      std::vector<float> sum(w.fw_host);
      for (int ii = 0; ii < E; ++ii)
        for (int oo = 0; oo < E; ++oo) {
          const float ws = w.skw_host[(size_t)oo * E + ii];            // conv weight (out, in); dhconv weight (in, out, l, re | im)
          float* d = sum.data() + ((size_t)ii * E + oo) * c.lmax * 2;
          for (int l = 0; l < c.lmax; ++l) d[2 * l] += ws;
        }
      if (!w.fw.frag) SDY_HIP_TRY(hipMalloc(&w.fw.frag, sdy_dhconv_frag_pack_bytes(c.lmax)));
      SDY_TRY(sdy_dh_h3_pack(sum.data(), c.lmax, w.fw.frag, &w.fw.frag_scale, spec_ilv(c)));
      w.skip_folded = true;
      return SDY_OK;
    };
    const bool fold = skip_foldable(c, i) && sdy_dhconv_frag_supported(E, E) && !sw().no_dh_frag;
    if (rest == "filter.filter.weight") {
      EXPECT_NUMEL((size_t)E * E * c.lmax * 2);
      const bool no_dh_frag = sw().no_dh_frag;
      if (fold) {   // packed when the skip weight is here too
        w.fw_host.assign(host, host + numel);
        SDY_TRY(pack_folded());
      } else if (h3 && sdy_dhconv_frag_supported(E, E) && !no_dh_frag) {   // persistent fragment-stream kernel (dh_h3.hip)
        if (!w.fw.frag) SDY_HIP_TRY(hipMalloc(&w.fw.frag, sdy_dhconv_frag_pack_bytes(c.lmax)));
        SDY_TRY(sdy_dh_h3_pack(host, c.lmax, w.fw.frag, &w.fw.frag_scale, spec_ilv(c)));
      } else if (h3) {
        if (w.fw.h3) (void)hipFree(w.fw.h3);
        w.fw.h3 = nullptr;
        SDY_HIP_TRY(hipMalloc(&w.fw.h3, sdy_dhconv_h3_pack_bytes(E, E, c.lmax)));
        SDY_TRY(dhconv_h3_pack(host, E, E, c.lmax, w.fw.h3, &w.fw.h3_scale, spec_ilv(c)));
      } else {
        SDY_TRY(dev_alloc(w.fw, numel));
        SDY_TRY(sdy_dhconv_pack_weight(host, E, E, c.lmax, w.fw.p, nullptr));
      }
      w.fw.set = true;
      return SDY_OK;
    }
    auto fold_bias = [&]() -> int {
      if (!fold || w.fb_host.empty() || w.skb_host.empty()) return SDY_OK;
      std::vector<float> sum(E);
      for (int k = 0; k < E; ++k) sum[k] = w.fb_host[k] + w.skb_host[k];
      return dev_upload(w.fbs, sum.data(), E);
    };
    if (rest == "filter.filter.bias") {
      EXPECT_NUMEL(E);
      w.fb_host.assign(host, host + numel);
      SDY_TRY(fold_bias());
      return dev_upload(w.fb, host, numel);
    }
    if (rest == "inner_skip.weight") {
      EXPECT_NUMEL((size_t)E * E);
      if (fold) {
        w.skw_host.assign(host, host + numel);
        SDY_TRY(pack_folded());
      }
      return dev_upload_conv(w.skw, host, E, E, E, h3);
    }
    if (rest == "inner_skip.bias") {
      EXPECT_NUMEL(E);
      w.skb_host.assign(host, host + numel);
      SDY_TRY(fold_bias());
      return dev_upload(w.skb, host, numel);
    }
    // fused MLP stream: (re)packed whenever both fc weights are known
    auto pack_mlp = [&]() -> int {
      if (!h3 || !sdy_mlp_h3_supported(E, H) || w.w1_host.empty() || w.w2_host.empty()) return SDY_OK;
      if (!w.mlp) SDY_HIP_TRY(hipMalloc(&w.mlp, sdy_mlp_h3_pack_bytes(E, H)));
      return sdy_mlp_h3_pack(w.w1_host.data(), w.w2_host.data(), E, H, w.mlp, &w.mlp_s1, &w.mlp_s2);
    };
    if (rest == "mlp.fwd.0.weight") {
      EXPECT_NUMEL((size_t)H * E);
      SDY_TRY(dev_upload_conv(w.w1, host, H, E, H, h3));
      if (h3 && sdy_mlp_h3_supported(E, H)) { w.w1_host.assign(host, host + numel); SDY_TRY(pack_mlp()); }
      return SDY_OK;
    }
    if (rest == "mlp.fwd.0.bias") { EXPECT_NUMEL(H); return dev_upload(w.b1, host, numel); }
    // layers.py:76-80: fc2 is index 3 of the Sequential when dropout > 0, else index 2
    if (rest == "mlp.fwd.2.weight" || rest == "mlp.fwd.3.weight") {
      EXPECT_NUMEL((size_t)E * H);
      SDY_TRY(dev_upload_conv(w.w2, host, E, H, E, h3));
      if (h3 && sdy_mlp_h3_supported(E, H)) { w.w2_host.assign(host, host + numel); SDY_TRY(pack_mlp()); }
      return SDY_OK;
    }
    if (rest == "mlp.fwd.2.bias" || rest == "mlp.fwd.3.bias") { EXPECT_NUMEL(E); return dev_upload(w.b2, host, numel); }
  }
  return SDY_ERR_NAME;
}

extern "C" int sdy_sfno_ready(const sdy_sfno* n_c) {
  sdy_sfno* n = const_cast<sdy_sfno*>(n_c);
  if (!n) return SDY_ERR_ARG;
  const sdy_sfno_config& c = n->cfg;
  n->missing.clear();
  auto need = [&](const DevBuf& b, const std::string& nm) {
    if (!b.set && n->missing.empty()) n->missing = nm;
  };
  if (c.pos_embed) need(n->pos, "pos_embed");
  need(n->e0w, "encoder.0.weight"); need(n->e0b, "encoder.0.bias"); need(n->e2w, "encoder.2.weight");
  need(n->d0w, "decoder.0.weight"); need(n->d0b, "decoder.0.bias"); need(n->d2w, "decoder.2.weight");
  if (c.with_time_emb) {
    need(n->t1w, "time_emb_mlp.1.weight"); need(n->t1b, "time_emb_mlp.1.bias");
    need(n->t3w, "time_emb_mlp.3.weight"); need(n->t3b, "time_emb_mlp.3.bias");
  }
  for (int i = 0; i < c.num_layers; ++i) {
    const BlockW& w = n->blk[i];
    const std::string p = "blocks." + std::to_string(i) + ".";
    need(w.n0w, p + "norm0.weight"); need(w.n0b, p + "norm0.bias");
    need(w.n1w, p + "norm1.weight"); need(w.n1b, p + "norm1.bias");
    if (c.with_time_emb) { need(w.tw, p + "time_mlp.1.weight"); need(w.tb, p + "time_mlp.1.bias"); }
    need(w.fw, p + "filter.filter.weight"); need(w.fb, p + "filter.filter.bias");
    need(w.skw, p + "inner_skip.weight"); need(w.skb, p + "inner_skip.bias");
    need(w.w1, p + "mlp.fwd.0.weight"); need(w.b1, p + "mlp.fwd.0.bias");
    need(w.w2, p + "mlp.fwd.{2|3}.weight"); need(w.b2, p + "mlp.fwd.{2|3}.bias");
  }
  if (!n->missing.empty()) return SDY_ERR_STATE;
  if (c.with_time_emb) {
    n->tm.E = c.embed_dim; n->tm.T = c.time_dim;
    n->tm.freq = n->freq.p; n->tm.w1t = n->t1w.p; n->tm.b1 = n->t1b.p; n->tm.w2t = n->t3w.p; n->tm.b2 = n->t3b.p;
    n->tm.wbt = n->wbt.p; n->tm.bb = n->bb.p;
  } else {
    n->tm.E = 0;
  }
  return SDY_OK;
}

extern "C" const char* sdy_sfno_missing(const sdy_sfno* n) { return n ? n->missing.c_str() : ""; }

namespace {
struct WsLayout {
  size_t cat, xa, xb, xe, xn, y, zt, hid, xf, cs, cs2, ca, cd, ca1, cd1, st0, st1, ste, ss, dp, trep, sp, total;
};
WsLayout ws_layout(const sdy_sfno* n, int B) {
  const sdy_sfno_config& c = n->cfg;
  const size_t HW = n->HW, E = c.embed_dim;
  WsLayout w;
  size_t off = 0;
  auto take = [&](size_t floats) { size_t o = off; off += round_up_sz(floats, 64); return o; };
  w.cat = take((size_t)B * n->catC * HW);
  w.xa = take((size_t)B * E * HW);
  w.xb = take((size_t)B * E * HW);
  w.xe = take((size_t)B * E * HW);    // encoder output: block 0 reads it, nothing overwrites it (sdy_sfno_fwd_args.reuse_encoder)
  w.xn = take((size_t)B * E * HW);
  w.y = take((size_t)B * E * HW);
  w.zt = take((size_t)B * ((HW + 63) / 64) * E * 64);   // inner-skip output, tile-major (conv_h3 -> mlp_h3), tiles padded
  w.hid = take((size_t)B * c.mlp_hidden * HW);
  w.xf = take(xf_floats(n->plan_data, B, (int)E));
  w.cs = take(cs_floats(n->plan_data, B, (int)E));
  w.cs2 = take(cs_floats(n->plan_data, B, (int)E));
  w.ca = take((size_t)B * E);
  w.cd = take((size_t)B * E);
  w.ca1 = take((size_t)B * E);
  w.cd1 = take((size_t)B * E);
  w.st1 = take((size_t)B * E * 4);   // same for norm1, filled by the inner-skip convolution's epilogue
  w.st0 = take((size_t)B * E * 4);   // (sum, sumsq) doubles of the next block's norm0, filled by the fused MLP epilogue
  w.ste = take((size_t)B * E * 4);   // the encoder output's statistics (block 0's norm0), kept for reuse_encoder
  w.ss = take((size_t)B * c.num_layers * 2 * E);
  w.dp = take((size_t)B * c.num_layers);
  w.trep = take((size_t)2 * B * (c.with_time_emb ? c.time_dim : 1));   // t_repr, then the time MLP's hidden layer
  w.sp = take((size_t)B * E * c.nlat * 4);   // per-ring (sum, sumsq) doubles of a folded block's act output (fft360.hip)
  w.total = off;
  return w;
}
}  // namespace

extern "C" size_t sdy_sfno_workspace_floats(const sdy_sfno* n, int B) {
  if (!n || B <= 0) return 0;
  return ws_layout(n, B).total;
}

// Largest batch one sdy_sfno_forward call covers.  The kernels address rows as (wave-uniform 64-bit base) + (32-bit lane
// offset).  On the default path of the production shape (tile-major spectral tensors: every block cs_tiled, below) the widest
// 32-bit quantity is dh_h3's element offset inside a coefficient tensor, mtr * B * 8 * L * 64 < 2^32: 258 rows at 180 x 360
// (sdy_dh_h3_launch refuses more; the offset is unsigned arithmetic).  With row-major coefficients (other shapes, the fallback switches) it is the Legendre
// synthesis reading up to 192 degree rows of Cs[l][m][b][2E], whose row stride is mtr * 2 B E floats: 60 rows.
extern "C" int sdy_sfno_max_batch(const sdy_sfno* n) {
  if (!n) return 0;
  const sdy_sfno_config& c = n->cfg;
  const int mtr = c.mmax < c.lmax ? c.mmax : c.lmax;
  bool all_tiled = c.gemm_mode == 1 && n->plan_data && n->plan_lg;
  if (all_tiled) {
    const int ilv = spec_ilv(c);
    all_tiled = plan_tiled_ok(n->plan_data, c.embed_dim, ilv) && plan_tiled_ok(n->plan_lg, c.embed_dim, ilv) &&
                n->plan_data->lmax == n->plan_lg->lmax && n->plan_data->mtr == n->plan_lg->mtr;
    for (const BlockW& w : n->blk) all_tiled = all_tiled && w.fw.frag != nullptr;
  }
  long b;
  if (all_tiled) {
    b = ((1L << 32) - 1) / ((long)mtr * 8 * c.lmax * 64);
  } else {
    const long per_row = 192L * mtr * 2 * c.embed_dim * 4 + 2L * c.embed_dim * 4;   // bytes of lane offset per batch row
    b = ((1L << 32) - 1) / per_row;
  }
  // No more rows than the drop-path row maps cover (SDY_MAP_MAX = 128): beyond it a forward would run unskipped, its workspace
  // (0.8 GB per row at 180 x 360) outgrows what a window driver should hold at once, and no test runs a single native call
  // there -- the Python network splits larger batches into near-equal calls (same call number, batch_offset + first row).
  if (b > SDY_MAP_MAX) b = SDY_MAP_MAX;
  return b < 1 ? 1 : (int)b;
}

extern "C" int sdy_sfno_time_embed(sdy_sfno* n, const float* time, int B, float* t_repr, float* ss, void* stream) {
  if (!n || !time || B <= 0) return SDY_ERR_ARG;
  SDY_TRY(sdy_sfno_ready(n));
  if (!n->cfg.with_time_emb) return SDY_ERR_UNSUPPORTED;
  if (!ss) return SDY_ERR_ARG;
  // (not on the forward's path, which carries the scratch in its workspace: a stream-ordered allocation will do)
  float* scratch = nullptr;
  SDY_HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&scratch), (size_t)2 * B * n->cfg.time_dim * sizeof(float), (hipStream_t)stream));
  const int rc = sdy_time_mlp_launch(n->tm, time, B, t_repr, ss, nullptr, nullptr, 0, 0, 0, 0, 0, (hipStream_t)stream, scratch);
  SDY_HIP_TRY(hipFreeAsync(scratch, (hipStream_t)stream));
  return rc;
}

// ---- stage timing of the forward (measurement only; off by default) ------------------------------------------------
namespace {
enum SdyStage {
  ST_CONCAT, ST_TIME_MLP, ST_ENC0, ST_ENC2, ST_NORM_COEFFS, ST_FFT_FWD, ST_LEG_FWD, ST_LEG_INV, ST_FFT_INV, ST_DHCONV,
  ST_SKIP_CONV, ST_MLP_FUSED, ST_MLP_FUSED_DROP, ST_FC1, ST_FC2, ST_DEC0, ST_DEC2, ST_ENC_PAIR, ST_DEC_PAIR, ST_DROP_COPY,
  ST_SKIP_GELU, ST_COUNT
};
const char* const kStageNames[ST_COUNT] = {
  "concat", "time_mlp", "encoder.0 conv", "encoder.2 conv", "instnorm coefficients", "rfft (lon)", "legendre analysis",
  "legendre synthesis", "irfft (lon)", "dhconv", "inner-skip conv", "mlp fused", "mlp fused (dropout)", "mlp fc1", "mlp fc2",
  "decoder.0 conv", "decoder.2 conv", "encoder (fused pair)", "decoder (fused pair)", "drop-path copy",
  "inner skip folded (gelu)"};
// Single-threaded by contract (one host thread issues forwards while the switch is on); the mutex only keeps the record list
// consistent against a concurrent sdy_profile_read.  Events come from a pool and are reused across reads.
struct SdyProfiler {
  bool on = false;
  struct Rec { int stage; int rows; hipEvent_t e0, e1; bool closed; };   // rows: batch rows the launch worked on
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  std::mutex mu;
  int get_event(hipEvent_t* e) {   // (under mu)
    if (!pool.empty()) { *e = pool.back(); pool.pop_back(); return SDY_OK; }
    SDY_HIP_TRY(hipEventCreate(e));
    return SDY_OK;
  }
  // returns the record's index (>= 0) in *idx; nothing is leaked when an event cannot be created or recorded
  int begin(int stage, int rows, hipStream_t s, int* idx) {
    std::lock_guard<std::mutex> g(mu);
    Rec r{stage, rows, nullptr, nullptr, false};
    SDY_TRY(get_event(&r.e0));
    int rc = get_event(&r.e1);
    if (rc == SDY_OK) {
      hipError_t e = hipEventRecord(r.e0, s);
      if (e != hipSuccess) rc = (int)e;
    }
    if (rc != SDY_OK) {
      pool.push_back(r.e0);
      if (r.e1) pool.push_back(r.e1);
      return rc;
    }
    recs.push_back(r);
    *idx = (int)recs.size() - 1;
    return SDY_OK;
  }
  int end(int idx, hipStream_t s) {
    std::lock_guard<std::mutex> g(mu);
    if (idx < 0 || idx >= (int)recs.size()) return SDY_ERR_STATE;   // the list was read in between: drop the sample
    SDY_HIP_TRY(hipEventRecord(recs[idx].e1, s));
    recs[idx].closed = true;
    return SDY_OK;
  }
} g_prof;
}  // namespace

// (a launch that fails returns before end(): its record stays open and sdy_profile_read skips it)
#define SDY_STAGE_N(stage, rows, call)                                          \
  do {                                                                          \
    int prof_idx_ = -1;                                                         \
    if (g_prof.on) SDY_TRY(g_prof.begin((stage), (rows), stream, &prof_idx_));  \
    SDY_TRY(call);                                                              \
    if (prof_idx_ >= 0) (void)g_prof.end(prof_idx_, stream);                    \
  } while (0)
#define SDY_STAGE(stage, call) SDY_STAGE_N(stage, B, call)

extern "C" int sdy_profile_enable(int on) {
  g_prof.on = on != 0;
  return SDY_OK;
}
extern "C" int sdy_profile_stage_count(void) { return ST_COUNT; }
extern "C" const char* sdy_profile_stage_name(int stage) {
  return (stage >= 0 && stage < ST_COUNT) ? kStageNames[stage] : "";
}
extern "C" int sdy_profile_read(double* total_ms, long* launches, int n) {
  return sdy_profile_read_rows(total_ms, launches, nullptr, n);
}
extern "C" int sdy_profile_read_rows(double* total_ms, long* launches, long* rows, int n) {
  if (!total_ms || !launches || n < ST_COUNT) return SDY_ERR_ARG;
  for (int i = 0; i < n; ++i) { total_ms[i] = 0.0; launches[i] = 0; if (rows) rows[i] = 0; }
  std::lock_guard<std::mutex> g(g_prof.mu);
  int rc = SDY_OK;
  for (auto& r : g_prof.recs) {
    if (r.closed) {   // records whose launch failed were never closed: no elapsed time exists for them
      float ms = 0.f;
      hipError_t e = hipEventSynchronize(r.e1);
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.e0, r.e1);
      if (e != hipSuccess) rc = (int)e;
      else { total_ms[r.stage] += ms; launches[r.stage] += 1; if (rows) rows[r.stage] += r.rows; }
    }
    g_prof.pool.push_back(r.e0);
    g_prof.pool.push_back(r.e1);
  }
  g_prof.recs.clear();
  return rc;
}

extern "C" int sdy_sfno_forward(sdy_sfno* n, const sdy_sfno_fwd_args* a, void* stream_v) {
  if (!n || !a || !a->out || !a->ws || a->B <= 0) return SDY_ERR_ARG;
  SDY_TRY(sdy_sfno_ready(n));
  const sdy_sfno_config& c = n->cfg;
  hipStream_t stream = (hipStream_t)stream_v;
  const int B = a->B, E = c.embed_dim, L = c.num_layers, HW = n->HW, Cin = c.in_chans, Hd = c.mlp_hidden;
  if (c.with_time_emb && !a->time) return SDY_ERR_ARG;
  const WsLayout w = ws_layout(n, B);
  if (a->ws_floats < w.total) return SDY_ERR_WORKSPACE;
  float* ws = a->ws;
  float *cat = ws + w.cat, *xa = ws + w.xa, *xb = ws + w.xb, *xn = ws + w.xn, *y = ws + w.y, *hid = ws + w.hid;
  float *Xf = ws + w.xf, *Cs = ws + w.cs, *Cs2 = ws + w.cs2, *ca = ws + w.ca, *cd = ws + w.cd;
  float *ca1 = ws + w.ca1, *cd1 = ws + w.cd1;
  double* st0 = reinterpret_cast<double*>(ws + w.st0);   // take() rounds offsets to 64 floats: 8-byte aligned
  double* st1 = reinterpret_cast<double*>(ws + w.st1);
  SDY_HIP_TRY(hipMemsetAsync(st1, 0, (size_t)B * E * 2 * sizeof(double), stream));
  bool have_st0 = false;                                  // statistics of `cur` are waiting in st0
  SDY_HIP_TRY(hipMemsetAsync(st0, 0, (size_t)B * E * 2 * sizeof(double), stream));
  float *ss = ws + w.ss, *dp = ws + w.dp, *trep = ws + w.trep;
  float* xe = ws + w.xe;
  double* ste = reinterpret_cast<double*>(ws + w.ste);
  const bool reuse = a->reuse_encoder != 0;
  if (reuse && (n->enc_ws != ws || n->enc_B != B)) return SDY_ERR_STATE;   // no previous forward on this workspace / batch
  // (a forward that overwrites the inputs invalidates the stored encoder output until its own encoder launch has been
  //  enqueued: an early error return must not leave a handle that claims a valid one)
  if (!reuse) n->enc_ws = nullptr;

  // ---- input concat (BaseModel.concat_condition_if_needed, _base_model.py:166-192) into the tail of the big-skip
  //      buffer: cat = [ block output (E) | inputs (Cin) ]  (sfnonet.py:804-805,831-832)
  const float* srcs[3];
  int chans[3];
  int ns = 0, ctot = 0;
  for (int i = 0; i < 3; ++i)
    if (a->in[i] && a->in_chans[i] > 0) { srcs[ns] = a->in[i]; chans[ns] = a->in_chans[i]; ctot += a->in_chans[i]; ++ns; }
  if (ns == 0 || ctot != Cin) return SDY_ERR_SHAPE;
  const long cat_bs = (long)n->catC * HW;
  float* cat_in = cat + (size_t)(n->catC - Cin) * HW;
  const bool drop = a->enable_dropout != 0;
  if (a->rows_per_call < 0 || (a->rows_per_call > 0 && B % a->rows_per_call)) return SDY_ERR_ARG;
  const int rpc = a->rows_per_call > 0 ? a->rows_per_call : B;   // stacked calls (sdy_sfno_fwd_args.rows_per_call)
  // shared_inputs: the stacked calls read the same rpc input rows (row b: input row b % rpc).  The concat still writes all B
  // rows (the decoder reads them beside every row's own block output); the ENCODER runs on the rpc distinct rows when block 0
  // can take its input through a row rule -- its only reader is then the forward FFT (x_mod): the grid-changing first block of
  // an equiangular data grid on the fft360 + leg_par path, with the encoder's own statistics (the production shape).
  const bool shared = a->shared_inputs != 0;
  if (shared && (reuse || rpc >= B)) return SDY_ERR_ARG;
  // (reuse_encoder: the inputs already sit in the tail of `cat` -- no block writes there -- and the encoder output in xe)
  if (!reuse) SDY_STAGE(ST_CONCAT, sdy_concat_launch(srcs, chans, ns, cat_in, cat_bs, B, HW, stream, shared ? rpc : 0));
  const bool enc_once = shared && n->enc.w && n->plan_data != n->plan_lg && plan_tiled_ok(n->plan_data, E, spec_ilv(c));
  const int Be = enc_once ? rpc : B;                               // rows the encoder computes

  // ---- time embedding + per-layer (scale|shift) + drop-path scales
  SDY_STAGE(ST_TIME_MLP, sdy_time_mlp_launch(n->tm, a->time, B, trep, ss, dp, a->drop_path_keep, drop ? 1 : 0, a->seed,
                                             a->call, a->batch_offset, rpc, stream,
                                             trep + (size_t)B * (n->cfg.with_time_emb ? n->cfg.time_dim : 1)));

  // ---- drop-path skip.  DropPath (src/models/modules/drop_path.py:15-22, applied at sfnonet.py:330) multiplies the whole
  //      branch of a trajectory -- transforms, dhconv, inner skip, MLP -- by 0 with the layer's rate; the keep decision is a
  //      Philox function of (seed, call, layer, trajectory) that the host evaluates here exactly as time_dense_kernel does on
  //      the device.  A block with dropped trajectories runs its kernels on the kept ones only (per-block intermediates are
  //      indexed compactly, SdyImgMap) and writes the dropped ones' output a x + d with sdy_affine_copy_stats_launch.
  //      Only on the default kernel path (fft360 + leg_par + dh_h3 + conv_h3 + mlp_h3); injected decisions (tests) are
  //      device data, so they run unskipped.  SDY_NO_DROP_SKIP=1 computes everything (A/B: bit-identical results).
  const bool no_drop_skip = sw().no_drop_skip;
  const bool skip_allowed = drop && !a->drop_path_keep && !no_drop_skip && B <= SDY_MAP_MAX;
  auto drop_path_keeps = [&](int layer, int b) {
    const int bq = b / rpc;
    const philox4 wd = philox4x32((uint32_t)(b - bq * rpc) + a->batch_offset, 0xFFFFFFFFu, 0x1000u + (uint32_t)layer,
                                     a->call + (uint32_t)bq, (uint32_t)(a->seed & 0xFFFFFFFFu), (uint32_t)(a->seed >> 32));
    return wd.x >= n->tm.dp_thr[layer];
  };

  sdy_conv_args cv;
  auto use_w = [&](const DevBuf& b) {
    cv.wt = b.p;
    if (c.gemm_mode == 1) {
      cv.w_h3 = b.h3; cv.w_h3_scale = b.h3_scale;
      cv.w_frag = b.frag; cv.w_frag_scale = b.frag_scale;
    }
  };
  auto conv_reset = [&]() {
    std::memset(&cv, 0, sizeof(cv));
    cv.B = B; cv.HW = HW; cv.seed = a->seed; cv.call = a->call; cv.batch_offset = a->batch_offset; cv.rows_per_call = rpc;
  };

  // ---- encoder (sfnonet.py:609-618,810,824): conv+bias -> GELU -> conv (no bias) -> + pos_embed
  // The encoder writes to a buffer of its own (xe) and its statistics to `ste`; block 0 works on a copy of the statistics
  // (sdy_instnorm_from_stats clears what it reads), so that a following forward on the same inputs can restart here.
  bool have_ste = false;
  if (reuse) {
    have_ste = n->enc_has_stats;
  } else if (n->enc.w) {   // one launch (pair_h3.hip), block 0's norm0 statistics from its epilogue
    sdy_pair_args pa{};
    pa.x = cat_in; pa.x_bstride = cat_bs; pa.w = n->enc.w; pa.w1_scale = n->enc.s1; pa.w2_scale = n->enc.s2;
    pa.b1 = n->e0b.p; pa.out = xe; pa.out_bstride = (long)E * HW;
    if (c.pos_embed) { pa.add = n->pos.p; pa.add_bstride = 0; }
    pa.B = Be; pa.Cin = Cin; pa.hidden = E; pa.Cout = E; pa.HW = HW;
    SDY_HIP_TRY(hipMemsetAsync(ste, 0, (size_t)Be * E * 2 * sizeof(double), stream));
    pa.stats = ste; have_ste = true;
    SDY_STAGE_N(ST_ENC_PAIR, Be, sdy_pair_h3(&pa, stream));
  } else {
  conv_reset();
  cv.x = cat_in; cv.x_bstride = cat_bs; use_w(n->e0w); cv.ldw = E; cv.out = xa; cv.out_bstride = (long)E * HW;
  cv.Cin = Cin; cv.Cout = E; cv.bias = n->e0b.p; cv.act = 1;
  SDY_STAGE(ST_ENC0, sdy_conv1x1(&cv, stream));
  conv_reset();
  cv.x = xa; cv.x_bstride = (long)E * HW; use_w(n->e2w); cv.ldw = E; cv.out = xe; cv.out_bstride = (long)E * HW;
  cv.Cin = E; cv.Cout = E;
  if (c.pos_embed) { cv.add = n->pos.p; cv.add_bstride = 0; cv.add_mode = 2; }
  // block 0's norm0 statistics from this convolution's epilogue (persistent kernel only): no pass over its output
  if (cv.w_frag && sdy_conv256_h3_supported(E, E)) {
    SDY_HIP_TRY(hipMemsetAsync(ste, 0, (size_t)B * E * 2 * sizeof(double), stream));
    cv.stats = ste; have_ste = true;
  }
  SDY_STAGE(ST_ENC2, sdy_conv1x1(&cv, stream));
  }
  // (a forward that ran the encoder on the shared rows only leaves nothing a later reuse_encoder forward of B rows could take)
  n->enc_ws = enc_once ? nullptr : ws; n->enc_B = B; n->enc_has_stats = have_ste;
  if (have_ste) {   // every stacked call starts from the statistics of the rows it shares
    for (int b0 = 0; b0 < B; b0 += Be)
      SDY_HIP_TRY(hipMemcpyAsync(st0 + (size_t)b0 * E * 2, ste, (size_t)Be * E * 2 * sizeof(double), hipMemcpyDeviceToDevice, stream));
    have_st0 = true;
  }

  float* cur = xe;
  float* nxt = xa;
  const int ilv = spec_ilv(c);   // channel order of Xf / Cs / Cs2 inside this forward
  for (int i = 0; i < L; ++i) {
    const BlockW& bw = n->blk[i];
    const sdy_sht_plan* pin = (i == 0) ? n->plan_data : n->plan_lg;
    const sdy_sht_plan* pout = (i == L - 1) ? n->plan_data : n->plan_lg;
    const bool scale_residual = pin != pout;  // s2convolutions.py:79-83
    const float pm = (drop && c.dropout_mlp > 0.f) ? c.dropout_mlp : 0.f;
    const bool no_fused = sw().no_fused_mlp;
    const bool fused_mlp = bw.mlp && !no_fused;   // (injected masks too: sdy_mlp_args.keep_hidden / keep_out)
    // The block's residual is norm0(x) (or its SHT round trip when the grids differ).  With the fused MLP kernel the
    // normalised tensor is never materialised: its two consumers (inner skip, final residual add) apply a*x + d to `cur`.
    const bool lazy_norm = fused_mlp && !scale_residual;
    // norm0 + time scale/shift folded into a*x+d (sfnonet.py:292,298-299)
    if (have_st0)   // statistics were accumulated by the previous block's MLP epilogue: no pass over `cur`
      SDY_STAGE(ST_NORM_COEFFS, sdy_instnorm_from_stats(st0, B, E, HW, bw.n0w.p, bw.n0b.p,
                                                        c.with_time_emb ? ss + (size_t)i * 2 * E : nullptr, (long)L * 2 * E,
                                                        1e-6f, ca, cd, stream));
    else
      SDY_STAGE(ST_NORM_COEFFS, sdy_instnorm_coeffs_launch(cur, B, E, HW, bw.n0w.p, bw.n0b.p,
                                                           c.with_time_emb ? ss + (size_t)i * 2 * E : nullptr,
                                                           (long)L * 2 * E, 1e-6f, ca, cd, stream));
    have_st0 = false;
    // SpectralConvS2.forward (s2convolutions.py:158-193)
    const bool polar_in = plan_polar_ok(pin, E), polar_out = plan_polar_ok(pout, E);
    const bool tiled_in = plan_tiled_ok(pin, E, ilv), tiled_out = plan_tiled_ok(pout, E, ilv);   // Xf / Yf tile-major (fft.h)
    // Cs / Cs2 tile-major by order too (analysis stores and synthesis loads become contiguous tiles; dh_h3 reads and writes
    // 256-byte pieces instead of 2 KB rows, which it does not notice: it is matrix / issue bound).
    const bool cs_tiled = tiled_in && tiled_out && bw.fw.frag && pin->lmax == pout->lmax && pin->mtr == pout->mtr;
    const bool frag_conv = c.gemm_mode == 1 && bw.skw.frag && sdy_conv256_h3_supported(E, E);
    const bool stats1 = frag_conv;   // norm1 statistics from the inner-skip convolution's epilogue
    // The tensor between the inner skip and the fused MLP has exactly one producer and one consumer, both walking 64-pixel
    // tiles: it is stored TILE-MAJOR (a tile = one contiguous 64 KB block for the stores of one and the loads of the other).
    const bool z_tiled = stats1 && fused_mlp;
    const long zt_bs = (long)((HW + 63) / 64) * E * 64;
    float* dst = (i == L - 1) ? cat : nxt;
    const long dst_bs = (i == L - 1) ? cat_bs : (long)E * HW;
    const long cur_bs = (cur == cat) ? cat_bs : (long)E * HW;

    // drop-path skip: the rows this block's kernels run on (Bp of them: perm[0 .. Bp) are their batch rows) and the dropped
    // ones (perm[Bp .. B)).  Two forms:
    //  * the residual is a x + d of the block input (lazy_norm: every block between the first and the last, or all of them when
    //    the data grid is the Gauss grid): NOTHING of the block runs for a dropped row, its output is the affine copy;
    //  * the residual is the SHT round trip of a x + d (scale_residual, s2convolutions.py:79-83: the first / last block on an
    //    equiangular data grid): forward transform and the residual's inverse transform run on all rows, in the ORDER perm
    //    (kept rows first -- a row's transform does not depend on its place in the batch), so that the dhconv, the second
    //    inverse transform, the inner skip and the MLP work on the leading Bp rows of tensors in that order; a dropped row's
    //    output is the copy of its residual.
    unsigned char perm[SDY_MAP_MAX];
    int Bp = B, nD = 0;
    if (skip_allowed && n->tm.dp_rate[i] > 0.f && fused_mlp && z_tiled && cs_tiled) {
      unsigned char drp[SDY_MAP_MAX];
      Bp = 0;
      for (int b = 0; b < B; ++b) {
        if (drop_path_keeps(i, b)) perm[Bp++] = (unsigned char)b;
        else drp[nD++] = (unsigned char)b;
      }
      for (int k = 0; k < nD; ++k) perm[Bp + k] = drp[k];
    }
    const unsigned char* rows = nD > 0 ? perm : nullptr;   // nullptr: every row is active, identity map
    const int Bf = scale_residual ? B : Bp;                // rows of the forward transform

    if (Bf > 0) {
      SDY_STAGE_N(ST_FFT_FWD, Bf, sdy_fft_launch_fwd(pin->fft, cur, ca, cd, (scale_residual || lazy_norm) ? nullptr : xn, Xf, Bf, E,
                                                     pin->nlat, pin->mtr, tiled_in ? 2 : ilv, polar_in ? pin->d_mcut : nullptr, stream, rows,
                                                     (i == 0 && enc_once) ? Be : 0));
      SDY_STAGE_N(ST_LEG_FWD, Bf, legendre_fwd_impl(pin, Xf, Cs, Bf, E, polar_in, stream, tiled_in, cs_tiled));
      if (scale_residual) {  // residual = inverse_transform(forward_transform(x)); in the order `perm` when rows were dropped
        SDY_STAGE(ST_LEG_INV, legendre_inv_impl(pout, Cs, Xf, B, E, polar_out, stream, tiled_out, cs_tiled));
        SDY_STAGE(ST_FFT_INV, sdy_fft_launch_inv(pout->fft, Xf, nullptr, xn, B, E, pout->nlat, pout->mtr, tiled_out ? 2 : ilv,
                                                 polar_out ? pout->d_mcut : nullptr, stream));
      }
    }
    if (Bp > 0) {
      if (bw.fw.frag)
        SDY_STAGE_N(ST_DHCONV, Bp, sdy_dh_h3_launch(Cs, bw.fw.frag, bw.fw.frag_scale, Cs2, c.lmax, pin->mtr, Bp, ilv, (hipStream_t)stream,
                                                    cs_tiled ? 1 : 0, Bf));
      else if (c.gemm_mode == 1)
        SDY_STAGE(ST_DHCONV, sdy_dhconv_h3(Cs, bw.fw.h3, bw.fw.h3_scale, Cs2, c.lmax, pin->mtr, B, E, E, stream));
      else
        SDY_STAGE(ST_DHCONV, sdy_dhconv(Cs, bw.fw.p, Cs2, c.lmax, pin->mtr, B, E, E, stream));
      // grid-changing block with the inner skip folded into the dhconv weights (skip_foldable): y = filter + skip + both biases
      // (the packed dhconv stream of such a block HOLDS the skip: running the convolution as well would count it twice)
      if (bw.skip_folded && !(scale_residual && bw.fw.frag && bw.fbs.set)) return SDY_ERR_STATE;
      const bool folded = bw.skip_folded;
      SDY_STAGE_N(ST_LEG_INV, Bp, legendre_inv_impl(pout, Cs2, Xf, Bp, E, polar_out, stream, tiled_out, cs_tiled));
      // ... and on the fft360 path with the tile-major conv -> MLP tensor the act itself rides on the inverse FFT's stores
      // (GELU, tile-major layout, per-ring norm1 statistics): no pass over y at all
      const bool act_in_fft = folded && tiled_out && z_tiled;
      if (act_in_fft) {
        double* part = reinterpret_cast<double*>(ws + w.sp);
        SDY_STAGE_N(ST_FFT_INV, Bp, sdy_fft360_launch_inv(pout->fft, Xf, bw.fbs.p, nullptr, Bp, E, pout->nlat, pout->mtr, 2,
                                                          polar_out ? pout->d_mcut : nullptr, stream, ws + w.zt, zt_bs, part));
        SDY_STAGE_N(ST_NORM_COEFFS, Bp, sdy_instnorm_from_partials_launch(part, pout->nlat, Bp, E, HW, bw.n1w.p, bw.n1b.p, 1e-6f, ca1,
                                                                          cd1, stream));
      } else
      SDY_STAGE_N(ST_FFT_INV, Bp, sdy_fft_launch_inv(pout->fft, Xf, folded ? bw.fbs.p : bw.fb.p, y, Bp, E, pout->nlat, pout->mtr,
                                                     tiled_out ? 2 : ilv, polar_out ? pout->d_mcut : nullptr, stream));
      if (act_in_fft) {
      } else if (folded) {
        // x = GELU(y): the 256 -> 256 convolution has become a matrix addition at load time; norm1 statistics from this pass
        float* zo = z_tiled ? ws + w.zt : y;
        SDY_STAGE_N(ST_SKIP_GELU, Bp, sdy_gelu_stats_launch(y, (long)E * HW, zo, z_tiled ? zt_bs : (long)E * HW, z_tiled ? 1 : 0, st1,
                                                           Bp, E, HW, stream));
        SDY_STAGE_N(ST_NORM_COEFFS, Bp, sdy_instnorm_from_stats(st1, Bp, E, HW, bw.n1w.p, bw.n1b.p, nullptr, 0, 1e-6f, ca1, cd1, stream));
      } else {
      // x = GELU(y + inner_skip(residual))  (sfnonet.py:303-311), in place over y
      conv_reset();
      cv.B = Bp; cv.x_rows = lazy_norm ? rows : nullptr;   // (xn is in the launch's own row order)
      cv.x = lazy_norm ? cur : xn; cv.x_bstride = lazy_norm ? cur_bs : (long)E * HW;
      if (lazy_norm) { cv.pa = ca; cv.pd = cd; }
      use_w(bw.skw); cv.ldw = E; cv.out = y; cv.out_bstride = (long)E * HW;
      cv.Cin = E; cv.Cout = E; cv.bias = bw.skb.p; cv.add = y; cv.add_bstride = (long)E * HW; cv.add_mode = 1; cv.act = 1; cv.kernel_tag = 3;
      if (stats1) cv.stats = st1;
      if (z_tiled) { cv.out = ws + w.zt; cv.out_bstride = zt_bs; cv.out_tiled = 1; }
      SDY_STAGE_N(ST_SKIP_CONV, Bp, sdy_conv1x1(&cv, stream));
      // norm1 (sfnonet.py:313-320) folded into the fc1 prologue; its statistics come from the convolution's epilogue
      if (stats1)
        SDY_STAGE_N(ST_NORM_COEFFS, Bp, sdy_instnorm_from_stats(st1, Bp, E, HW, bw.n1w.p, bw.n1b.p, nullptr, 0, 1e-6f, ca1, cd1, stream));
      else
        SDY_STAGE(ST_NORM_COEFFS, sdy_instnorm_coeffs_launch(y, B, E, HW, bw.n1w.p, bw.n1b.p, nullptr, 0, 1e-6f, ca1, cd1, stream));
      }
    }
    // MLP (layers.py:73-80): fc1 + GELU + dropout
    const bool stats_next = fused_mlp && i < L - 1;   // the next block's norm0 statistics from this block's epilogue
    if (Bp == 0) {
      // (every trajectory dropped: nothing of the branch runs)
    } else if (fused_mlp) {
      // fc1 + GELU + dropout + fc2 + dropout + DropPath + residual in one launch: the hidden activation stays on the CU
      sdy_mlp_args ma;
      std::memset(&ma, 0, sizeof(ma));
      ma.x = y; ma.x_bstride = (long)E * HW; ma.pa = ca1; ma.pd = cd1;
      if (z_tiled) { ma.x = ws + w.zt; ma.x_bstride = zt_bs; ma.x_tiled = 1; }
      ma.w = bw.mlp; ma.w1_scale = bw.mlp_s1; ma.w2_scale = bw.mlp_s2; ma.b1 = bw.b1.p; ma.b2 = bw.b2.p;
      ma.out = dst; ma.out_bstride = dst_bs;
      if (lazy_norm) { ma.add = cur; ma.add_bstride = cur_bs; ma.add_a = ca; ma.add_d = cd; }
      else { ma.add = xn; ma.add_bstride = (long)E * HW; ma.add_by_launch_row = rows ? 1 : 0; }
      ma.B = Bp; ma.out_rows = rows; ma.E = E; ma.hidden = Hd; ma.HW = HW;
      ma.drop_p = pm; ma.seed = a->seed; ma.call = a->call; ma.stream_fc1 = 2u * i; ma.stream_fc2 = 2u * i + 1u;
      ma.batch_offset = a->batch_offset; ma.rows_per_call = rpc;
      if (drop && n->tm.dp_rate[i] > 0.f) ma.batch_scale = dp + (size_t)i * B;  // dp is laid out [layer][b]
      if (pm > 0.f && a->keep_masks) {   // tests: the reference's recorded masks drive the fused kernel's INJECT instantiation
        ma.keep_hidden = static_cast<const float*>(a->keep_masks[2 * i]);
        ma.keep_out = static_cast<const float*>(a->keep_masks[2 * i + 1]);
      }
      if (stats_next) ma.stats = st0;
      SDY_STAGE_N(pm > 0.f ? ST_MLP_FUSED_DROP : ST_MLP_FUSED, Bp, sdy_mlp_h3(&ma, stream));
    } else {
      conv_reset();
      cv.x = y; cv.x_bstride = (long)E * HW; use_w(bw.w1); cv.ldw = Hd; cv.out = hid; cv.out_bstride = (long)Hd * HW;
      cv.Cin = E; cv.Cout = Hd; cv.pa = ca1; cv.pd = cd1; cv.bias = bw.b1.p; cv.act = 1; cv.kernel_tag = 1;
      cv.drop_p = pm; cv.stream_id = 2u * i; cv.keep_mask = (pm > 0.f && a->keep_masks) ? a->keep_masks[2 * i] : nullptr;
      SDY_STAGE(ST_FC1, sdy_conv1x1(&cv, stream));
      // fc2 + dropout, DropPath, + residual (sfnonet.py:325-335)
      conv_reset();
      cv.x = hid; cv.x_bstride = (long)Hd * HW; use_w(bw.w2); cv.ldw = E; cv.out = dst; cv.out_bstride = dst_bs;
      cv.Cin = Hd; cv.Cout = E; cv.bias = bw.b2.p; cv.kernel_tag = 2;
      cv.drop_p = pm; cv.stream_id = 2u * i + 1u; cv.keep_mask = (pm > 0.f && a->keep_masks) ? a->keep_masks[2 * i + 1] : nullptr;
      if (drop && n->tm.dp_rate[i] > 0.f) cv.batch_scale = dp + (size_t)i * B;  // dp is laid out [layer][b]
      cv.add = xn; cv.add_bstride = (long)E * HW; cv.add_mode = 2;
      SDY_STAGE(ST_FC2, sdy_conv1x1(&cv, stream));
    }
    if (nD > 0) {   // dropped trajectories: block output = the residual; their share of the next block's statistics
      if (lazy_norm)
        SDY_STAGE_N(ST_DROP_COPY, nD, sdy_affine_copy_stats_launch(cur, cur_bs, ca, cd, dst, dst_bs, stats_next ? st0 : nullptr, E, HW,
                                                                   perm + Bp, nD, stream));
      else
        SDY_STAGE_N(ST_DROP_COPY, nD, sdy_affine_copy_stats_launch(xn, (long)E * HW, nullptr, nullptr, dst, dst_bs,
                                                                   stats_next ? st0 : nullptr, E, HW, perm + Bp, nD, stream, Bp));
    }
    if (stats_next) have_st0 = true;
    cur = dst;
    nxt = (dst == xa) ? xb : xa;
  }

  // ---- decoder (sfnonet.py:734-744,831-837)
  if (n->dec.w) {
    sdy_pair_args pa{};
    pa.x = cat; pa.x_bstride = cat_bs; pa.w = n->dec.w; pa.w1_scale = n->dec.s1; pa.w2_scale = n->dec.s2;
    pa.b1 = n->d0b.p; pa.out = a->out; pa.out_bstride = (long)c.out_chans * HW;
    pa.B = B; pa.Cin = n->decC; pa.hidden = E; pa.Cout = c.out_chans; pa.HW = HW;
    SDY_STAGE(ST_DEC_PAIR, sdy_pair_h3(&pa, stream));
    return SDY_OK;
  }
  float* dh = xa;
  conv_reset();
  cv.x = cat; cv.x_bstride = cat_bs; use_w(n->d0w); cv.ldw = E;
  cv.out = dh; cv.out_bstride = (long)E * HW; cv.Cin = n->decC; cv.Cout = E; cv.bias = n->d0b.p; cv.act = 1;
  SDY_STAGE(ST_DEC0, sdy_conv1x1(&cv, stream));
  conv_reset();
  cv.x = dh; cv.x_bstride = (long)E * HW; use_w(n->d2w); cv.ldw = n->ldo; cv.out = a->out;
  cv.out_bstride = (long)c.out_chans * HW; cv.Cin = E; cv.Cout = c.out_chans;
  SDY_STAGE(ST_DEC2, sdy_conv1x1(&cv, stream));
  return SDY_OK;
}
