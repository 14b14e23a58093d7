// Host-side fp64 tables of the spherical-harmonic transform (no GPU involved).
//
// Restates the table construction of the third-party `torch-harmonics` package (un-vendored, un-pinned in the
// reference: setup.py:98; used at src/models/sfno/sfnonet.py:551-554), following its published algorithm:
//   quadrature.py  : legendre_gauss_weights (numpy leggauss), clenshaw_curtiss_weights (Waldvogel)
//   legendre.py    : legpoly / _precompute_legpoly (3-term recursion, ortho norm, Condon-Shortley phase)
//   sht.py         : colatitudes = flip(arccos(nodes)); forward weights = pct * w_k
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/sdy_amd.h"

namespace {

const long double PI_L = 3.141592653589793238462643383279502884L;

// Gauss-Legendre nodes (ascending) and weights on [-1, 1] by Newton iteration in extended precision.
void leggauss(int n, std::vector<double>& x, std::vector<double>& w) {
  x.assign(n, 0.0);
  w.assign(n, 0.0);
  for (int i = 0; i < (n + 1) / 2; ++i) {
    long double z = cosl(PI_L * (i + 0.75L) / (n + 0.5L));  // descending roots
    long double pp = 0.0L;
    for (int it = 0; it < 100; ++it) {
      long double p1 = 1.0L, p2 = 0.0L;
      for (int j = 1; j <= n; ++j) {
        long double p3 = p2;
        p2 = p1;
        p1 = ((2.0L * j - 1.0L) * z * p2 - (j - 1.0L) * p3) / j;
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0L);
      long double z1 = z;
      z = z1 - p1 / pp;
      if (fabsl(z - z1) < 1e-19L) break;
    }
    long double wi = 2.0L / ((1.0L - z * z) * pp * pp);
    x[i] = (double)(-z);
    x[n - 1 - i] = (double)z;
    w[i] = (double)wi;
    w[n - 1 - i] = (double)wi;
  }
  if (n & 1) x[n / 2] = 0.0;
}

// Clenshaw-Curtis nodes cos(linspace(pi, 0, n)) (both poles) and weights (Waldvogel's DFT construction).
void clenshaw_curtiss(int n, std::vector<double>& x, std::vector<double>& w) {
  x.assign(n, 0.0);
  w.assign(n, 0.0);
  const double step = (0.0 - M_PI) / (n - 1);
  for (int i = 0; i < n; ++i) {
    double t = (i == n - 1) ? 0.0 : M_PI + i * step;
    x[i] = std::cos(t);
  }
  if (n == 2) {
    w[0] = w[1] = 1.0;
    return;
  }
  const int n1 = n - 1;
  std::vector<long double> Nn;
  for (int k = 1; k < n1; k += 2) Nn.push_back((long double)k);
  const int l = (int)Nn.size();
  const int m = n1 - l;
  std::vector<long double> v0(l + 1 + m, 0.0L);
  for (int i = 0; i < l; ++i) v0[i] = 2.0L / Nn[i] / (Nn[i] - 2.0L);
  v0[l] = 1.0L / Nn[l - 1];
  // v = 0 - v0[:-1] - v0[-1:0:-1]
  std::vector<long double> v(n1);
  const int len = l + 1 + m;  // = n1 + 1
  for (int i = 0; i < n1; ++i) v[i] = 0.0L - v0[i] - v0[len - 1 - i];
  std::vector<long double> g(n1, -1.0L);
  g[l] += n1;
  g[m] += n1;
  const long double den = (long double)n1 * n1 - 1.0L + (n1 % 2);
  for (int i = 0; i < n1; ++i) g[i] /= den;
  // w_j = Re ifft(v + g)[j]
  for (int j = 0; j < n1; ++j) {
    long double acc = 0.0L;
    for (int k = 0; k < n1; ++k) {
      long long jk = ((long long)j * k) % n1;
      acc += (v[k] + g[k]) * cosl(2.0L * PI_L * jk / n1);
    }
    w[j] = (double)(acc / n1);
  }
  w[n1] = w[0];
}

}  // namespace

extern "C" int sdy_sht_tables_host(int nlat, int nlon, int lmax, int mmax, int grid, double* pct, double* wq,
                                   double* theta_out) {
  if (nlat < 2 || nlon < 2 || lmax < 1 || mmax < 1) return SDY_ERR_ARG;
  std::vector<double> cost, w;
  if (grid == SDY_GRID_LEGENDRE_GAUSS)
    leggauss(nlat, cost, w);
  else if (grid == SDY_GRID_EQUIANGULAR)
    clenshaw_curtiss(nlat, cost, w);
  else
    return SDY_ERR_ARG;
  // tq = flip(arccos(cost)); table evaluated at cos(tq)
  std::vector<double> theta(nlat), x(nlat);
  for (int k = 0; k < nlat; ++k) theta[k] = std::acos(cost[nlat - 1 - k]);
  for (int k = 0; k < nlat; ++k) x[k] = std::cos(theta[k]);
  if (theta_out) std::memcpy(theta_out, theta.data(), sizeof(double) * nlat);
  if (wq) std::memcpy(wq, w.data(), sizeof(double) * nlat);
  if (!pct) return SDY_OK;

  const int nmax = mmax > lmax ? mmax : lmax;
  const size_t K = (size_t)nlat;
  std::vector<double> vdm((size_t)nmax * nmax * K, 0.0);
  auto V = [&](int m, int l) { return vdm.data() + ((size_t)m * nmax + l) * K; };
  const double v00 = 1.0 / std::sqrt(4.0 * M_PI);
  for (size_t k = 0; k < K; ++k) V(0, 0)[k] = v00;
  for (int l = 1; l < nmax; ++l) {
    const double a = std::sqrt(2.0 * l + 1.0);
    for (size_t k = 0; k < K; ++k) {
      V(l - 1, l)[k] = a * x[k] * V(l - 1, l - 1)[k];
      V(l, l)[k] = std::sqrt((2.0 * l + 1.0) * (1.0 + x[k]) * (1.0 - x[k]) / 2.0 / l) * V(l - 1, l - 1)[k];
    }
  }
  for (int l = 2; l < nmax; ++l) {
    for (int m = 0; m < l - 1; ++m) {
      const double a = std::sqrt((2.0 * l - 1.0) / (l - m) * (2.0 * l + 1.0) / (l + m));
      const double b = std::sqrt((double)(l + m - 1) / (l - m) * (2.0 * l + 1.0) / (2.0 * l - 3.0) * (l - m - 1) / (l + m));
      for (size_t k = 0; k < K; ++k) V(m, l)[k] = x[k] * a * V(m, l - 1)[k] - b * V(m, l - 2)[k];
    }
  }
  for (int m = 0; m < mmax; ++m) {
    const double sgn = (m & 1) ? -1.0 : 1.0;  // Condon-Shortley phase
    for (int l = 0; l < lmax; ++l) {
      const double* src = V(m, l);
      double* dst = pct + ((size_t)m * lmax + l) * K;
      for (size_t k = 0; k < K; ++k) dst[k] = sgn * src[k];
    }
  }
  return SDY_OK;
}

// n = product of radices, radices from {4, 2, 3, 5}; returns SDY_ERR_UNSUPPORTED for other prime factors.
int sdy_factor_radices(int n, int* radices, int* nstages) {
  int cnt = 0;
  const int cand[4] = {4, 2, 3, 5};
  for (int ci = 0; ci < 4; ++ci) {
    while (n % cand[ci] == 0 && n > 1) {
      if (cnt >= 10) return SDY_ERR_UNSUPPORTED;
      radices[cnt++] = cand[ci];
      n /= cand[ci];
    }
  }
  if (n != 1) return SDY_ERR_UNSUPPORTED;
  *nstages = cnt;
  return SDY_OK;
}
