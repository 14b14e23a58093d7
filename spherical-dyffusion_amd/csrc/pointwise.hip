// HBM-bound pointwise / reduction kernels of the SFNO block and the DYffusion sampler, gfx950.
#include <cmath>
#include <atomic>
#include <mutex>
#include <vector>
#include "common.h"
#include "pointwise.h"

namespace {

// ---- InstanceNorm statistics -> per-(b,c) affine coefficients ------------------------------------------
// nn.InstanceNorm2d(eps=1e-6, affine=True, track_running_stats=False): biased variance over H*W per (b,c)
// (src/models/sfno/sfnonet.py:641-648), folded with the block's time scale/shift (sfnonet.py:280-287).
// One workgroup per (c, b) plane; 16-byte loads; fp64 accumulation (HBM-bound, the DP adds are free).
__global__ __launch_bounds__(256) void instnorm_coeffs_kernel(const float* __restrict__ x, int C, int HW,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ ss, long ss_stride, float eps,
                                                               float* __restrict__ a_out, float* __restrict__ d_out,
                                                               unsigned* __restrict__ flags) {
  const int c = blockIdx.x, b = blockIdx.y;
  const float* px = x + ((long)b * C + c) * HW;
  double s = 0.0, s2 = 0.0;
  const int n4 = HW >> 2;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(px);
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 v = p4[i];
    s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  for (int i = (n4 << 2) + threadIdx.x; i < HW; i += 256) {
    const double v = px[i];
    s += v;
    s2 += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    s2 += __shfl_down(s2, off, 64);
  }
  __shared__ double sh[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[wave] = s;
    sh[4 + wave] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double S = sh[0] + sh[1] + sh[2] + sh[3];
    const double S2 = sh[4] + sh[5] + sh[6] + sh[7];
    if (flags && !(fabs(S) <= 1.7e308 && S2 <= 1.7e308)) atomicOr(flags, (unsigned)SDY_FLAG_NONFINITE);   // inf or NaN
    const double mean = S / HW;
    double var = S2 / HW - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    float a = gamma[c] * rstd;
    float d = beta[c] - (float)mean * a;
    if (ss) {
      const float sc = ss[(long)b * ss_stride + c] + 1.0f;
      const float sh_ = ss[(long)b * ss_stride + C + c];
      a = a * sc;
      d = d * sc + sh_;
    }
    a_out[b * C + c] = a;
    d_out[b * C + c] = d;
  }
}

// Same coefficients from statistics accumulated by a producer kernel's epilogue (mlp_h3.hip): stats[b][c] = (sum, sumsq).
__global__ __launch_bounds__(256) void instnorm_from_stats_kernel(double* __restrict__ stats, int BC, int C, int HW,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   const float* __restrict__ ss, long ss_stride, float eps,
                                                                   float* __restrict__ a_out, float* __restrict__ d_out,
                                                                   unsigned* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= BC) return;
  const int b = i / C, c = i - b * C;
  const double S = stats[2 * (long)i], S2 = stats[2 * (long)i + 1];
  stats[2 * (long)i] = 0.0;
  stats[2 * (long)i + 1] = 0.0;
  if (flags && !(fabs(S) <= 1.7e308 && S2 <= 1.7e308)) atomicOr(flags, (unsigned)SDY_FLAG_NONFINITE);   // inf or NaN
  const double mean = S / HW;
  double var = S2 / HW - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  float a = gamma[c] * rstd;
  float d = beta[c] - (float)mean * a;
  if (ss) {
    const float sc = ss[(long)b * ss_stride + c] + 1.0f;
    const float sh_ = ss[(long)b * ss_stride + C + c];
    a = a * sc;
    d = d * sc + sh_;
  }
  a_out[i] = a;
  d_out[i] = d;
}

// The same from PARTIAL statistics with one writer per slot: part[b][k][c] = (sum, sumsq) of ring k (the inverse FFT's fused
// act epilogue, fft360.hip), summed here in the order of k -- no atomics anywhere, bit-reproducible coefficients.
__global__ __launch_bounds__(256) void instnorm_from_partials_kernel(const double* __restrict__ part, int K, int BC, int C, int HW,
                                                                      const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, float eps,
                                                                      float* __restrict__ a_out, float* __restrict__ d_out,
                                                                      unsigned* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= BC) return;
  const int b = i / C, c = i - b * C;
  const double* q = part + ((long)b * K * C + c) * 2;
  // (one thread walks a slot's K partials: the loads of a group are independent and issued together, the additions keep the
  //  order of k -- 180 dependent load-add round trips took 76 us per launch)
  constexpr int G = 20;
  double S = 0.0, S2 = 0.0;
  int k = 0;
  for (; k + G <= K; k += G) {
    double v[G], w[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      v[j] = q[(long)(k + j) * C * 2];
      w[j] = q[(long)(k + j) * C * 2 + 1];
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      S += v[j];
      S2 += w[j];
    }
  }
  for (; k < K; ++k) {
    S += q[(long)k * C * 2];
    S2 += q[(long)k * C * 2 + 1];
  }
  if (flags && !(fabs(S) <= 1.7e308 && S2 <= 1.7e308)) atomicOr(flags, (unsigned)SDY_FLAG_NONFINITE);   // inf or NaN
  const double mean = S / HW;
  double var = S2 / HW - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float a = gamma[c] * rstd;
  a_out[i] = a;
  d_out[i] = beta[c] - (float)mean * a;
}

// ---- drop-path skip: the block output of a trajectory whose branch DropPath zeroed ------------------------------------
// The block computes  out = drop_path(mlp(...)) + residual  with residual = norm0(x) (+ time scale / shift) = a x + d
// (src/models/sfno/sfnonet.py:289-337, src/models/modules/drop_path.py:15-22); for a dropped trajectory that is
// out = 0 * (...) + (a x + d).  capi.hip runs the block's kernels on the kept trajectories only; this kernel writes the dropped
// ones' output -- the same fma the fused MLP's epilogue applies to its residual rows -- and, like that epilogue, the
// (sum, sum of squares) of what it stores for the next block's InstanceNorm: fp32 per 4-pixel quad, fp64 across quads.
// One workgroup per (channel, dropped image).
__global__ __launch_bounds__(256) void affine_copy_stats_kernel(const float* __restrict__ x, long x_bs,
                                                                 const float* __restrict__ a, const float* __restrict__ d,
                                                                 float* __restrict__ out, long out_bs,
                                                                 double* __restrict__ stats, int C, int HW,
                                                                 const SdyImgMap rows, int src_row0) {
  // batch row b of out / a / d / stats; x row: the same, or src_row0 + (index in the list) when the source is in launch order
  const int c = blockIdx.x, b = sdy_img(rows, blockIdx.y), bs = src_row0 >= 0 ? src_row0 + (int)blockIdx.y : b;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(x + (long)bs * x_bs + (long)c * HW);
  f32x4* o4 = reinterpret_cast<f32x4*>(out + (long)b * out_bs + (long)c * HW);
  const float av = a ? a[b * C + c] : 1.0f, dv = a ? d[b * C + c] : 0.0f;   // (fma(r, 1, 0) == r: the plain copy is exact)
  double s = 0.0, s2 = 0.0;
  const int n4 = HW >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 r = p4[i];
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(r[e], av, dv);
    o4[i] = v;
    s += (double)sdy_quad_sum(v);
    s2 += (double)sdy_quad_sumsq(v);
  }
  if (!stats) return;   // (uniform)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    s2 += __shfl_down(s2, off, 64);
  }
  __shared__ double sh[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[wave] = s;
    sh[4 + wave] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // the one writer of this (image, channel): the caller zeroed the slot
    stats[((long)b * C + c) * 2] += (sh[0] + sh[1]) + (sh[2] + sh[3]);
    stats[((long)b * C + c) * 2 + 1] += (sh[4] + sh[5]) + (sh[6] + sh[7]);
  }
}

// ---- GELU + InstanceNorm statistics (+ tile-major store) of one tensor ---------------------------------------------------
// The block's `x = act(x + inner_skip(residual))` (sfnonet.py:303-311) where the inner skip has been FOLDED into the dhconv
// weights (capi.hip: the grid-changing blocks, whose residual is itself an inverse transform of the same coefficients): what
// the inverse FFT left in `y` already holds filter(x) + inner_skip(residual) + both biases, and the 256 -> 256 convolution
// launch shrinks to this pass -- read y, exact-erf GELU, the norm1 statistics conv_h3's epilogue would have produced (fp32 per
// 4-pixel quad, fp64 across quads), store in the layout the MLP reads (tile-major [64-pixel tile][C][64], or NCHW).
// One workgroup per (channel, image): it is the one writer of its statistics slot.
__global__ __launch_bounds__(256) void gelu_stats_kernel(const float* __restrict__ y, long y_bs, float* __restrict__ out,
                                                          long out_bs, int out_tiled, double* __restrict__ stats, int C,
                                                          int HW) {
  const int c = blockIdx.x, b = blockIdx.y;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(y + (long)b * y_bs + (long)c * HW);
  float* ob = out + (long)b * out_bs + (out_tiled ? (long)c * 64 : (long)c * HW);
  const long tile_stride = (long)C * 64;
  double s = 0.0, s2 = 0.0;
  const int n4 = HW >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 r = p4[i];
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf_exact(r[e]);
    const int px = 4 * i;
    float* o = out_tiled ? ob + (long)(px >> 6) * tile_stride + (px & 63) : ob + px;
    *reinterpret_cast<f32x4*>(o) = v;
    s += (double)sdy_quad_sum(v);
    s2 += (double)sdy_quad_sumsq(v);
  }
  if (!stats) return;   // (uniform)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    s2 += __shfl_down(s2, off, 64);
  }
  __shared__ double sh[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[wave] = s;
    sh[4 + wave] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // the one writer of this (image, channel): the caller zeroed the slot
    stats[((long)b * C + c) * 2] += (sh[0] + sh[1]) + (sh[2] + sh[3]);
    stats[((long)b * C + c) * 2 + 1] += (sh[4] + sh[5]) + (sh[6] + sh[7]);
  }
}

// ---- channel concat (torch.cat(dim=1)) ------------------------------------------------------------------
struct ConcatArgs {
  const float* src[4];
  int chans[4];
  int nsrc;
};
__global__ __launch_bounds__(256) void concat_kernel(const ConcatArgs a, float* __restrict__ out, long out_bstride,
                                                      int HW4, int src_rows) {
  // grid: (ceil(HW4/256), total_chans, B); one float4 per thread
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW4) return;
  int c = blockIdx.y;
  const int b = blockIdx.z;
  const int bs = src_rows > 0 ? b % src_rows : b;   // stacked calls that share their inputs (sdy_sfno_fwd_args.shared_inputs)
  int s = 0, coff = 0;
  while (s < a.nsrc - 1 && c >= a.chans[s]) {
    c -= a.chans[s];
    coff += a.chans[s];
    ++s;
  }
  const f32x4* src = reinterpret_cast<const f32x4*>(a.src[s] + ((long)bs * a.chans[s] + c) * HW4 * 4);
  f32x4* dst = reinterpret_cast<f32x4*>(out + (long)b * out_bstride + (long)(coff + c) * HW4 * 4);
  dst[i] = src[i];
}

// ---- cold-sampling update: out = x_s + (x_ip_next - x_ip_s)   (src/diffusion/dyffusion.py:517-519) -------
__global__ __launch_bounds__(256) void cold_update_kernel(const float* __restrict__ xs, const float* __restrict__ xn,
                                                           const float* __restrict__ xi, float* __restrict__ out,
                                                           size_t n) {
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 a = reinterpret_cast<const f32x4*>(xs)[i];
    const f32x4 b = reinterpret_cast<const f32x4*>(xn)[i];
    const f32x4 c = xi ? reinterpret_cast<const f32x4*>(xi)[i] : a;
    reinterpret_cast<f32x4*>(out)[i] = a + (b - c);
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float a = xs[i], b = xn[i], c = xi ? xi[i] : a;
    out[i] = a + (b - c);
  }
}

// ---- time embedding ---------------------------------------------------------------------------------------
// SinusoidalPosEmb -> Linear -> GELU -> Linear (src/models/modules/misc.py:21-33,132-148), then per block
// SiLU -> Linear (src/models/sfno/sfnonet.py:210-213,280-284); also the per-(sample, layer) drop-path scale
// (src/models/modules/drop_path.py:15-22).
// Three launches of one dense-layer kernel (h1 = GELU(W1 emb + b1), t_repr = W2 h1 + b2, (scale|shift)_l = Wb_l SiLU(t_repr) +
// bb_l): a workgroup owns 64 output columns and 8 batch rows, its four waves split the contraction, so every weight is read
// once per 8 rows and the three layers' 7 MB are spread over 64 ... 256 workgroups.  (Until round 4 one launch did it all with
// every workgroup recomputing t_repr from all 5 MB of W1 / W2 through its own 64 B/clk of L2 bandwidth: 77-103 us per forward
// whatever the batch -- 2.3 % of a single-member pass.)  All weights are stored transposed ([in][out]): a wave reads 256
// consecutive bytes per k.  Sums are fp32 in (wave slice, then four slices) order: rounding-level differences to a serial sum.
constexpr int TM_ROWS = 8, TM_KC = 1024, TM_UNROLL = 8;
enum { TM_EMB_GELU = 0, TM_PLAIN = 1, TM_SILU_LAYERS = 2 };
template <int MODE>
__global__ __launch_bounds__(256) void time_dense_kernel(const SdyTimeMlp t, const float* __restrict__ in,
                                                         float* __restrict__ out, int B, float* __restrict__ dp_out,
                                                         const float* __restrict__ dp_keep_in, int enable_dropout,
                                                         uint32_t seed_lo, uint32_t seed_hi, uint32_t call,
                                                         uint32_t batch_offset, int rows_per_call) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = MODE == TM_EMB_GELU ? t.E : t.T;                       // contraction length
  const int O = MODE == TM_SILU_LAYERS ? 2 * t.E : t.T;                // outputs per weight matrix
  const int cbs = (O + 63) / 64;                                       // column blocks per matrix
  const int layer = MODE == TM_SILU_LAYERS ? (int)blockIdx.x / cbs : 0;
  const int cb = MODE == TM_SILU_LAYERS ? (int)blockIdx.x - layer * cbs : (int)blockIdx.x;
  const float* wt = MODE == TM_EMB_GELU ? t.w1t : MODE == TM_PLAIN ? t.w2t : t.wbt + (long)layer * t.T * 2 * t.E;
  const float* bias = MODE == TM_EMB_GELU ? t.b1 : MODE == TM_PLAIN ? t.b2 : t.bb + (long)layer * 2 * t.E;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = (int)blockIdx.y * TM_ROWS;
  const int nrows = B - b0 < TM_ROWS ? B - b0 : TM_ROWS;
  const int o = cb * 64 + lane, oc = o < O ? o : O - 1;
  const int kc_max = K < TM_KC ? (K + 3) & ~3 : TM_KC;
  float* in_s = sm;                        // [TM_ROWS][kc_max]
  float* red = sm + TM_ROWS * kc_max;      // [4 waves][TM_ROWS][64]
  float acc[TM_ROWS];
#pragma unroll
  for (int r = 0; r < TM_ROWS; ++r) acc[r] = 0.0f;

  for (int k0 = 0; k0 < K; k0 += TM_KC) {
    const int kc = K - k0 < TM_KC ? K - k0 : TM_KC, kc4 = (kc + 3) & ~3;
    // stage the rows' inputs of this k range (zeros past K and past the last row)
    for (int i = tid; i < TM_ROWS * kc4; i += 256) {
      const int r = i / kc4, k = i - r * kc4;
      float v = 0.0f;
      if (r < nrows && k < kc) {
        if (MODE == TM_EMB_GELU) {
          const int half = t.E / 2, e = k0 + k;
          const float arg = in[b0 + r] * t.freq[e < half ? e : e - half];   // in = time[B]
          v = e < half ? sinf(arg) : cosf(arg);
        } else {
          v = in[(long)(b0 + r) * K + k0 + k];
          if (MODE == TM_SILU_LAYERS) v = silu(v);
        }
      }
      in_s[r * kc_max + k] = v;
    }
    __syncthreads();
    // wave w takes the k quads g = w, w + 4, ...: four weight rows (256 bytes each per wave) against eight broadcast quads
    const int ngroups = kc4 >> 2;
    // (eight quads per trip: 32 weight loads in flight per thread -- the layer's weights come from HBM on every forward, and a
    // workgroup's 256 KB at 4 loads per thread were 64 dependent round trips)
    for (int g0 = wave; g0 < ngroups; g0 += 4 * TM_UNROLL) {
      float w4[TM_UNROLL][4];
#pragma unroll
      for (int u = 0; u < TM_UNROLL; ++u) {
        const int k = 4 * (g0 + 4 * u);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = k0 + k + e < K ? k0 + k + e : K - 1;   // (past K: the staged inputs are zero, or the quad is skipped)
          w4[u][e] = wt[(long)kk * O + oc];
        }
      }
#pragma unroll
      for (int u = 0; u < TM_UNROLL; ++u) {
        const int g = g0 + 4 * u;
        if (g < ngroups) {
#pragma unroll
          for (int r = 0; r < TM_ROWS; ++r) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(in_s + r * kc_max + 4 * g);
            // explicit FMAs, one chain per row: left to the compiler's contraction the unrolled rows came out with DIFFERENT
            // roundings (row 7 of a block differed from rows 0 .. 6 in the last bit) -- a batch row must not depend on its place
            acc[r] = __builtin_fmaf(w4[u][3], v.w, __builtin_fmaf(w4[u][2], v.z, __builtin_fmaf(w4[u][1], v.y, __builtin_fmaf(w4[u][0], v.x, acc[r]))));
          }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < TM_ROWS; ++r) red[(wave * TM_ROWS + r) * 64 + lane] = acc[r];
  __syncthreads();
  for (int r = wave; r < nrows; r += 4) {
    if (o < O) {
      float v = bias[o] + ((red[(0 * TM_ROWS + r) * 64 + lane] + red[(1 * TM_ROWS + r) * 64 + lane]) +
                           (red[(2 * TM_ROWS + r) * 64 + lane] + red[(3 * TM_ROWS + r) * 64 + lane]));
      if (MODE == TM_EMB_GELU) v = gelu_erf_exact(v);
      if (MODE == TM_SILU_LAYERS) out[((long)(b0 + r) * t.num_layers + layer) * O + o] = v;
      else out[(long)(b0 + r) * O + o] = v;
    }
  }
  if (MODE == TM_SILU_LAYERS && dp_out && blockIdx.x == 0) {   // drop-path scales of this block's rows, [layer][b]
    for (int i = tid; i < t.num_layers * nrows; i += 256) {
      const int ly = i / nrows, b = b0 + (i - ly * nrows);
      float scale = 1.0f;
      const float p = t.dp_rate[ly];
      if (enable_dropout && p > 0.0f) {
        bool keep;
        if (dp_keep_in) {
          keep = dp_keep_in[ly * B + b] != 0.0f;
        } else {
          const int bq = b / rows_per_call;   // stacked calls: trajectory b % rows_per_call of call + b / rows_per_call
          const philox4 w = philox4x32((uint32_t)((b - bq * rows_per_call) + batch_offset), 0xFFFFFFFFu,
                                          0x1000u + (uint32_t)ly, call + (uint32_t)bq, seed_lo, seed_hi);
          keep = w.x >= t.dp_thr[ly];
        }
        scale = keep ? 1.0f / (1.0f - p) : 0.0f;
      }
      dp_out[ly * B + b] = scale;
    }
  }
}

// drop-path scales only (network without time embedding)
__global__ void droppath_kernel(const SdyTimeMlp t, float* __restrict__ dp_out, const float* __restrict__ dp_keep_in,
                                int B, int enable_dropout, uint32_t seed_lo, uint32_t seed_hi, uint32_t call,
                                uint32_t batch_offset, int rows_per_call) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int L = t.num_layers;
  if (i >= B * L) return;
  const int b = i / L, layer = i - b * L;
  float scale = 1.0f;
  const float p = t.dp_rate[layer];
  if (enable_dropout && p > 0.0f) {
    bool keep;
    if (dp_keep_in) {
      keep = dp_keep_in[layer * B + b] != 0.0f;
    } else {
      const int bq = b / rows_per_call;
      const philox4 w = philox4x32((uint32_t)((b - bq * rows_per_call) + batch_offset), 0xFFFFFFFFu,
                                      0x1000u + (uint32_t)layer, call + (uint32_t)bq, seed_lo, seed_hi);
      keep = w.x >= t.dp_thr[layer];
    }
    scale = keep ? 1.0f / (1.0f - p) : 0.0f;
  }
  dp_out[layer * B + b] = scale;  // [layer][b]
}

// ---- layout converters for the stand-alone SHT entry points (parity tests; not on the fused path) ---------
// Cs[l][m][b][ri][c]  ->  out (B,C,L,Mfull) complex64 interleaved; zero where m > l or m >= mtr
__global__ __launch_bounds__(256) void spec_to_torch_kernel(const float* __restrict__ Cs, float* __restrict__ out,
                                                             int B, int C, int L, int mtr, int Mfull) {
  const long total = (long)B * C * L * Mfull;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i % Mfull);
    long r = i / Mfull;
    const int l = (int)(r % L);
    r /= L;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    float re = 0.f, im = 0.f;
    if (m < mtr && m <= l) {
      const long o = (((long)l * mtr + m) * B + b) * (2L * C) + c;
      re = Cs[o];
      im = Cs[o + C];
    }
    out[2 * i] = re;
    out[2 * i + 1] = im;
  }
}
// in (B,C,L,Mfull) complex64 -> Cs[l][m][b][ri][c] (all m < mtr written, also m > l: the synthesis table is zero there)
__global__ __launch_bounds__(256) void torch_to_spec_kernel(const float* __restrict__ in, float* __restrict__ Cs,
                                                             int B, int C, int L, int mtr, int Mfull) {
  const long total = (long)L * mtr * B * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long r = i / C;
    const int b = (int)(r % B);
    r /= B;
    const int m = (int)(r % mtr);
    const int l = (int)(r / mtr);
    const long src = ((((long)b * C + c) * L + l) * Mfull + m) * 2;
    const long o = (((long)l * mtr + m) * B + b) * (2L * C) + c;
    Cs[o] = in[src];
    Cs[o + C] = in[src + 1];
  }
}

// ---- stepper glue (src/ace_inference/core/stepper_multistep.py:298-466) -----------------------------------------
__global__ __launch_bounds__(256) void norm_pack_kernel(const sdy_var_table v, int t, int T1, int HW4,
                                                         float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW4) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const f32x4 x = reinterpret_cast<const f32x4*>(v.data[c] + ((long)b * T1 + t) * HW4 * 4)[i];
  const float m = v.mean[c], s = v.std[c];
  f32x4 y;
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = (x[e] - m) / s;   // same two operations as the reference, no reciprocal
  reinterpret_cast<f32x4*>(out + ((long)b * v.nvars + c) * HW4 * 4)[i] = y;
}

struct PtrTable {
  float* p[SDY_MAX_VARS];
};
__global__ __launch_bounds__(256) void init_timeline_kernel(const sdy_var_table v, int T1, int HW4, const PtrTable tln_t,
                                                             const PtrTable tld_t) {
  float* const* tln = tln_t.p;
  float* const* tld = tld_t.p;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW4) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const long off = (long)b * T1 * HW4 * 4;
  const f32x4 x = reinterpret_cast<const f32x4*>(v.data[c] + off)[i];
  const float m = v.mean[c], s = v.std[c];
  f32x4 y, d;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    y[e] = (x[e] - m) / s;
    d[e] = y[e] * s + m;
  }
  reinterpret_cast<f32x4*>(tln[c] + off)[i] = y;
  reinterpret_cast<f32x4*>(tld[c] + off)[i] = d;
}

__global__ __launch_bounds__(256) void step_finish_kernel(const sdy_step_finish_args a) {
  const int HW4 = a.HW >> 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW4) return;
  const int e = blockIdx.y, b = blockIdx.z;
  const int oi = a.out_idx[e], ii = a.in_idx[e];
  f32x4 g;
  if (oi >= 0) {
    g = reinterpret_cast<const f32x4*>(a.gen + ((long)b * a.n_out + oi) * a.HW)[i];
    const long toff = ((long)b * a.T1 + a.t) * a.HW;
    f32x4 fb = g;   // what is fed back into the next step's input
    if (a.ar_init) fb = reinterpret_cast<const f32x4*>(a.ar_init + ((long)b * a.n_out + oi) * a.HW)[i];
    if (e == a.presc_entry) {   // Prescriber.__call__ (prescriber.py:68-92), on the prediction and on the fed-back state
      const f32x4 tv = reinterpret_cast<const f32x4*>(a.presc_target + toff)[i];
      const f32x4 mk = reinterpret_cast<const f32x4*>(a.presc_mask + toff)[i];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float tn = (tv[q] - a.mean[e]) / a.std[e];
        if (a.interpolate) {
          g[q] = mk[q] * tn + (1.0f - mk[q]) * g[q];
          fb[q] = mk[q] * tn + (1.0f - mk[q]) * fb[q];
        } else {
          const bool on = (int)rintf(mk[q]) == a.mask_value;   // torch.round = round half to even
          g[q] = on ? tn : g[q];
          fb[q] = on ? tn : fb[q];
        }
      }
    }
    reinterpret_cast<f32x4*>(a.gen_norm_tl[e] + toff)[i] = g;
    reinterpret_cast<f32x4*>(a.gen_tl[e] + toff)[i] = g * a.std[e] + a.mean[e];
    g = fb;
  } else {
    g = reinterpret_cast<const f32x4*>(a.prev_in + ((long)b * a.n_in + ii) * a.HW)[i];   // input-only: carried over
  }
  if (ii >= 0) reinterpret_cast<f32x4*>(a.next_in + ((long)b * a.n_in + ii) * a.HW)[i] = g;
}

__global__ __launch_bounds__(256) void lp_terms_kernel(const float* __restrict__ gen, const sdy_var_table v, int t, int T1,
                                                        int HW4, double* __restrict__ terms) {
  const int c = blockIdx.y, b = blockIdx.z;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gen + ((long)b * v.nvars + c) * HW4 * 4);
  const f32x4* x4 = reinterpret_cast<const f32x4*>(v.data[c] + ((long)b * T1 + t) * HW4 * 4);
  const float m = v.mean[c], s = v.std[c];
  double d2 = 0.0, y2 = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW4; i += gridDim.x * 256) {
    const f32x4 g = g4[i], x = x4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float y = (x[e] - m) / s;
      const float d = g[e] - y;
      d2 += (double)d * d;
      y2 += (double)y * y;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    d2 += __shfl_down(d2, off, 64);
    y2 += __shfl_down(y2, off, 64);
  }
  __shared__ double sh[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[wave] = d2;
    sh[4 + wave] = y2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&terms[2 * b], sh[0] + sh[1] + sh[2] + sh[3]);
    atomicAdd(&terms[2 * b + 1], sh[4] + sh[5] + sh[6] + sh[7]);
  }
}

// ---- on-device ensemble diagnostics (src/ace_inference/core/metrics.py:32-54,107-208) -------------------------------
// One pass over the ensemble: per (sample, time) plane p and pixel, the M member values stay in registers; the plane's
// area-weighted sums of  (mean_m x - truth)^2,  var_m(x) (unbiased),  fair CRPS = mean_m|x - truth| - sum_{i,j}|x_i - x_j|
// / (2 M (M - 1))  and  (mean_m x - truth)  are accumulated in fp64 (block reduce + one atomic per block and quantity).
constexpr int ENS_MAX = 64;
__global__ __launch_bounds__(256) void ens_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                           const float* __restrict__ w, int M, long member_stride, int HW,
                                                           double* __restrict__ out) {
  const int pl = blockIdx.y;
  const float* pp = pred + (long)pl * HW;
  const float* tp = truth + (long)pl * HW;
  double a_se = 0.0, a_var = 0.0, a_crps = 0.0, a_bias = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    float x[ENS_MAX];
    float mean = 0.f;
#pragma unroll 8
    for (int m = 0; m < M; ++m) {
      x[m] = pp[(long)m * member_stride + i];
      mean += x[m];
    }
    mean /= (float)M;
    const float t = tp[i], wi = w[i];
    float var = 0.f, skill = 0.f, pair = 0.f;
    for (int m = 0; m < M; ++m) {
      const float d = x[m] - mean;
      var += d * d;
      skill += fabsf(x[m] - t);
      for (int n = m + 1; n < M; ++n) pair += fabsf(x[m] - x[n]);
    }
    var = M > 1 ? var / (float)(M - 1) : 0.f;
    const float crps = M > 1 ? skill / (float)M - pair / (float)(M * (M - 1)) : skill;   // 2 * pair / (2 M (M-1))
    const float e = mean - t;
    a_se += (double)wi * e * e;
    a_var += (double)wi * var;
    a_crps += (double)wi * crps;
    a_bias += (double)wi * e;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a_se += __shfl_down(a_se, off, 64);
    a_var += __shfl_down(a_var, off, 64);
    a_crps += __shfl_down(a_crps, off, 64);
    a_bias += __shfl_down(a_bias, off, 64);
  }
  __shared__ double sh[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[wave] = a_se; sh[4 + wave] = a_var; sh[8 + wave] = a_crps; sh[12 + wave] = a_bias;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    atomicAdd(&out[4 * pl + q], sh[4 * q] + sh[4 * q + 1] + sh[4 * q + 2] + sh[4 * q + 3]);
  }
}

// ---- per-timestep series of the reduced inference aggregator (src/ace_inference/core/aggregator/inference/reduced.py:144-266)
// The same pass with the planes addressed as (sample, time) through strides -- the window driver's member-stacked view
// (members, samples, time, H, W) is a transposed view of the device batch, no copy -- and four more sums per plane for the
// area-weighted mean / standard deviation of the ensemble mean and of the target (metrics.py:32-82):
//   out[p][0..7] += sum_w (mean - t)^2 | sum_w var_m | sum_w fair CRPS | sum_w (mean - t) | sum_w mean | sum_w mean^2 |
//                   sum_w t | sum_w t^2
__global__ __launch_bounds__(256) void ens_series_kernel(const float* __restrict__ pred, int M, long member_stride,
                                                          long sample_stride, const float* __restrict__ truth,
                                                          long truth_sample_stride, const float* __restrict__ w, int T, int HW,
                                                          double* __restrict__ out) {
  const int pl = blockIdx.y;
  const int smp = pl / T, t_i = pl - smp * T;
  const float* pp = pred + (long)smp * sample_stride + (long)t_i * HW;
  const float* tp = truth + (long)smp * truth_sample_stride + (long)t_i * HW;
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    float x[ENS_MAX];
    float mean = 0.f;
#pragma unroll 8
    for (int m = 0; m < M; ++m) {
      x[m] = pp[(long)m * member_stride + i];
      mean += x[m];
    }
    mean /= (float)M;
    const float t = tp[i], wi = w[i];
    float var = 0.f, skill = 0.f, pair = 0.f;
    for (int m = 0; m < M; ++m) {
      const float d = x[m] - mean;
      var += d * d;
      skill += fabsf(x[m] - t);
      for (int n = m + 1; n < M; ++n) pair += fabsf(x[m] - x[n]);
    }
    var = M > 1 ? var / (float)(M - 1) : 0.f;
    const float crps = M > 1 ? skill / (float)M - pair / (float)(M * (M - 1)) : skill;
    const float e = mean - t;
    const double wd = (double)wi;
    a[0] += wd * e * e;
    a[1] += wd * var;
    a[2] += wd * crps;
    a[3] += wd * e;
    a[4] += wd * mean;
    a[5] += wd * ((double)mean * mean);
    a[6] += wd * t;
    a[7] += wd * ((double)t * t);
  }
  __shared__ double sh[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    double v = a[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) sh[4 * q + wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int q = threadIdx.x;
    atomicAdd(&out[8 * pl + q], sh[4 * q] + sh[4 * q + 1] + sh[4 * q + 2] + sh[4 * q + 3]);
  }
}

}  // namespace

// ---- time-mean accumulation (src/ace_inference/core/aggregator/inference/time_mean.py:97-117) ---------------------------
// acc[p] += scale * sum_{r0 < n0} sum_{r1 < n1} sum_{t0 <= t < T} x[r0 * s0 + r1 * s1 + t * HW + p]: the mean over members,
// samples and the window's time steps of one variable, added to the running (H, W) map.  One pass over the data, 16-byte
// loads, fp32 sums per pixel (at most a few hundred terms), HBM-bound.
__global__ __launch_bounds__(256) void time_mean_kernel(const float* __restrict__ x, int n0, long s0, int n1, long s1, int t0,
                                                         int T, int HW4, float scale, float* __restrict__ acc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW4) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int a = 0; a < n0; ++a)
    for (int b = 0; b < n1; ++b) {
      const f32x4* px = reinterpret_cast<const f32x4*>(x + (long)a * s0 + (long)b * s1) + i;
      for (int t = t0; t < T; ++t) s += px[(long)t * HW4];
    }
  f32x4* pa = reinterpret_cast<f32x4*>(acc) + i;
  *pa = *pa + s * scale;
}
extern "C" int sdy_time_mean_accumulate(const float* x, int n0, long stride0, int n1, long stride1, int t0, int T, int HW,
                                        float scale, float* acc, void* stream) {
  if (!x || !acc || n0 < 1 || n1 < 1 || T < 1 || t0 < 0 || t0 >= T || HW < 1) return SDY_ERR_ARG;
  if ((HW & 3) || (stride0 & 3) || (stride1 & 3)) return SDY_ERR_ALIGN;
  hipLaunchKernelGGL(time_mean_kernel, dim3((HW / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, n0, stride0, n1,
                     stride1, t0, T, HW / 4, scale, acc);
  return sdy_launch_status();
}

extern "C" int sdy_ensemble_metrics(const float* pred, const float* truth, const float* weights, int M, long member_stride,
                                    int n_planes, int HW, double* out, void* stream) {
  if (!pred || !truth || !weights || !out || M < 1 || n_planes < 1 || HW < 1) return SDY_ERR_ARG;
  if (M > ENS_MAX) return SDY_ERR_UNSUPPORTED;
  int gx = (HW + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(ens_metrics_kernel, dim3(gx, n_planes), dim3(256), 0, (hipStream_t)stream, pred, truth, weights, M,
                     member_stride, HW, out);
  return sdy_launch_status();
}

extern "C" int sdy_ensemble_series(const float* pred, int M, long member_stride, long sample_stride, const float* truth,
                                   long truth_sample_stride, const float* weights, int n_sample, int T, int HW, double* out,
                                   void* stream) {
  if (!pred || !truth || !weights || !out || M < 1 || n_sample < 1 || T < 1 || HW < 1) return SDY_ERR_ARG;
  if (M > ENS_MAX) return SDY_ERR_UNSUPPORTED;
  if ((long)n_sample * T > 65535) return SDY_ERR_UNSUPPORTED;
  int gx = (HW + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(ens_series_kernel, dim3(gx, n_sample * T), dim3(256), 0, (hipStream_t)stream, pred, M, member_stride,
                     sample_stride, truth, truth_sample_stride, weights, T, HW, out);
  return sdy_launch_status();
}

extern "C" int sdy_norm_pack(const sdy_var_table* vars, int t, int T1, int B, int HW, float* out, void* stream) {
  if (!vars || !out || vars->nvars < 1 || vars->nvars > SDY_MAX_VARS || t < 0 || t >= T1 || B <= 0 || HW <= 0) return SDY_ERR_ARG;
  if (HW & 3) return SDY_ERR_ALIGN;
  for (int i = 0; i < vars->nvars; ++i)
    if (!vars->data[i] || vars->std[i] == 0.0f) return SDY_ERR_ARG;
  hipLaunchKernelGGL(norm_pack_kernel, dim3((HW / 4 + 255) / 256, vars->nvars, B), dim3(256), 0, (hipStream_t)stream, *vars,
                     t, T1, HW / 4, out);
  return sdy_launch_status();
}

extern "C" int sdy_init_timeline(const sdy_var_table* vars, int T1, int B, int HW, float* const* tl_norm,
                                 float* const* tl_denorm, void* stream) {
  if (!vars || !tl_norm || !tl_denorm || vars->nvars < 1 || vars->nvars > SDY_MAX_VARS || B <= 0 || HW <= 0) return SDY_ERR_ARG;
  if (HW & 3) return SDY_ERR_ALIGN;
  PtrTable tn, td;   // the host pointer arrays travel by value in the kernel arguments
  for (int i = 0; i < vars->nvars; ++i) {
    if (!tl_norm[i] || !tl_denorm[i] || !vars->data[i] || vars->std[i] == 0.0f) return SDY_ERR_ARG;
    tn.p[i] = tl_norm[i];
    td.p[i] = tl_denorm[i];
  }
  hipLaunchKernelGGL(init_timeline_kernel, dim3((HW / 4 + 255) / 256, vars->nvars, B), dim3(256), 0, (hipStream_t)stream,
                     *vars, T1, HW / 4, tn, td);
  return sdy_launch_status();
}

extern "C" int sdy_step_finish(const sdy_step_finish_args* a, void* stream) {
  if (!a || !a->gen || !a->next_in || a->B <= 0 || a->HW <= 0 || a->n_entries < 1 || a->n_entries > SDY_MAX_VARS) return SDY_ERR_ARG;
  if (a->HW & 3) return SDY_ERR_ALIGN;
  if (a->t < 1 || a->t >= a->T1) return SDY_ERR_ARG;
  for (int e = 0; e < a->n_entries; ++e) {
    if (a->out_idx[e] < 0 && a->in_idx[e] < 0) return SDY_ERR_ARG;
    if (a->out_idx[e] >= a->n_out || a->in_idx[e] >= a->n_in) return SDY_ERR_ARG;
    if (a->out_idx[e] >= 0 && (!a->gen_norm_tl[e] || !a->gen_tl[e])) return SDY_ERR_ARG;
    if (a->out_idx[e] < 0 && !a->prev_in) return SDY_ERR_ARG;
  }
  if (a->presc_entry >= 0 && (a->presc_entry >= a->n_entries || !a->presc_target || !a->presc_mask ||
                              a->out_idx[a->presc_entry] < 0))
    return SDY_ERR_ARG;
  hipLaunchKernelGGL(step_finish_kernel, dim3((a->HW / 4 + 255) / 256, a->n_entries, a->B), dim3(256), 0,
                     (hipStream_t)stream, *a);
  return sdy_launch_status();
}

extern "C" int sdy_lp_rel_terms(const float* gen, const sdy_var_table* targets, int t, int T1, int B, int HW,
                                double* terms, void* stream) {
  if (!gen || !targets || !terms || targets->nvars < 1 || targets->nvars > SDY_MAX_VARS || t < 0 || t >= T1 || B <= 0)
    return SDY_ERR_ARG;
  if (HW & 3) return SDY_ERR_ALIGN;
  int gx = (HW / 4 + 255) / 256;
  if (gx > 16) gx = 16;
  hipLaunchKernelGGL(lp_terms_kernel, dim3(gx, targets->nvars, B), dim3(256), 0, (hipStream_t)stream, gen, *targets, t, T1,
                     HW / 4, terms);
  return sdy_launch_status();
}

// ---- sticky status word (include/sdy_amd.h) ------------------------------------------------------------------------
int sdy_flags_ptr(unsigned** flags) {
  static unsigned* words[SDY_MAX_DEVICES] = {};
  static std::mutex mu;
  int dev = 0;
  SDY_TRY(sdy_current_device(&dev));
  std::lock_guard<std::mutex> g(mu);
  if (!words[dev]) {
    SDY_HIP_TRY(hipMalloc(&words[dev], 64));
    SDY_HIP_TRY(hipMemset(words[dev], 0, 64));
  }
  *flags = words[dev];
  return SDY_OK;
}
// ---- GELU table (common.h): cubic Taylor pieces of 16 Phi(x) around 385 nodes on [-6, 6], fp64 on the host --------------
int sdy_gelu_table_ptr(const float** table_dev) {
  static float* tabs[SDY_MAX_DEVICES] = {};
  static std::mutex mu;
  int dev = 0;
  SDY_TRY(sdy_current_device(&dev));
  std::lock_guard<std::mutex> g(mu);
  if (!tabs[dev]) {
    std::vector<float> h((size_t)SDY_GELU_NODES * 4);
    const double hstep = 1.0 / 32.0, sx = (double)SDY_GELU_SX / (double)SDY_GELU_WS, inv_sqrt_2pi = 0.39894228040143267794;
    for (int n = 0; n < SDY_GELU_NODES; ++n) {
      const double x = -6.0 + n * hstep;
      const double Phi = 0.5 * std::erfc(-x * 0.70710678118654752440), phi = inv_sqrt_2pi * std::exp(-0.5 * x * x);
      double c0 = sx * Phi, c1 = sx * phi * hstep, c2 = -sx * x * phi * hstep * hstep / 2.0,
             c3 = sx * (x * x - 1.0) * phi * hstep * hstep * hstep / 6.0;
      if (n == 0) { c0 = 0.0; c1 = c2 = c3 = 0.0; }                       // clamped inputs: exactly 0 ...
      if (n == SDY_GELU_NODES - 1) { c0 = sx; c1 = c2 = c3 = 0.0; }       // ... and exactly 16 v = 2 w
      h[4 * n] = (float)c0; h[4 * n + 1] = (float)c1; h[4 * n + 2] = (float)c2; h[4 * n + 3] = (float)c3;
    }
    float* d = nullptr;
    SDY_HIP_TRY(hipMalloc(&d, h.size() * sizeof(float)));
    SDY_HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    tabs[dev] = d;
  }
  *table_dev = tabs[dev];
  return SDY_OK;
}

// ---- range headroom (debug read-back): words 1 .. 5 of the status allocation hold the float bits of the largest magnitude
// each consumer class has staged since the last reset
static std::atomic<bool> g_headroom_on{false};
int sdy_headroom_ptr(int slot, unsigned** word) {
  *word = nullptr;
  if (!g_headroom_on.load(std::memory_order_relaxed) || slot < 0 || slot >= SDY_RANGE_SLOTS) return SDY_OK;
  unsigned* d = nullptr;
  SDY_TRY(sdy_flags_ptr(&d));
  *word = d + 1 + slot;
  return SDY_OK;
}
extern "C" int sdy_range_headroom_enable(int on) {
  g_headroom_on.store(on != 0);
  return SDY_OK;
}
extern "C" int sdy_range_headroom(float* max_staged, int reset, void* stream) {
  if (!max_staged) return SDY_ERR_ARG;
  unsigned* d = nullptr;
  SDY_TRY(sdy_flags_ptr(&d));
  SDY_HIP_TRY(hipMemcpyAsync(max_staged, d + 1, SDY_RANGE_SLOTS * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (reset) SDY_HIP_TRY(hipMemsetAsync(d + 1, 0, SDY_RANGE_SLOTS * sizeof(unsigned), (hipStream_t)stream));
  SDY_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return SDY_OK;
}
// Host evaluation of the dropout stream's generator (the same __host__ __device__ function the kernels call)
extern "C" int sdy_dropout_stream_rounds(void) { return SDY_PHILOX_ROUNDS; }
extern "C" int sdy_dropout_stream_words(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t key_lo, uint32_t key_hi,
                                        uint32_t* out4) {
  if (!out4) return SDY_ERR_ARG;
  const philox4 w = philox4x32(c0, c1, c2, c3, key_lo, key_hi);
  out4[0] = w.x; out4[1] = w.y; out4[2] = w.z; out4[3] = w.w;
  return SDY_OK;
}

extern "C" int sdy_status_flags(unsigned* flags, int reset, void* stream) {
  if (!flags) return SDY_ERR_ARG;
  unsigned* d = nullptr;
  SDY_TRY(sdy_flags_ptr(&d));
  SDY_HIP_TRY(hipMemcpyAsync(flags, d, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (reset) SDY_HIP_TRY(hipMemsetAsync(d, 0, sizeof(unsigned), (hipStream_t)stream));
  SDY_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return SDY_OK;
}
extern "C" int sdy_status_flags_async(unsigned* flags_host, int reset, void* stream) {
  if (!flags_host) return SDY_ERR_ARG;
  unsigned* d = nullptr;
  SDY_TRY(sdy_flags_ptr(&d));
  SDY_HIP_TRY(hipMemcpyAsync(flags_host, d, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (reset) SDY_HIP_TRY(hipMemsetAsync(d, 0, sizeof(unsigned), (hipStream_t)stream));
  return SDY_OK;
}

int sdy_instnorm_coeffs_launch(const float* x, int B, int C, int HW, const float* gamma, const float* beta,
                               const float* ss, long ss_stride, float eps, float* a, float* d, hipStream_t stream) {
  if (!x || !gamma || !beta || !a || !d || B <= 0 || C <= 0 || HW <= 0) return SDY_ERR_ARG;
  if (HW & 3) return SDY_ERR_ALIGN;
  unsigned* flags = nullptr;
  SDY_TRY(sdy_flags_ptr(&flags));
  hipLaunchKernelGGL(instnorm_coeffs_kernel, dim3(C, B), dim3(256), 0, stream, x, C, HW, gamma, beta, ss, ss_stride,
                     eps, a, d, flags);
  return sdy_launch_status();
}

extern "C" int sdy_instnorm_from_stats(double* stats, int B, int C, int HW, const float* gamma, const float* beta,
                                       const float* ss, long ss_stride, float eps, float* a, float* d, void* stream) {
  if (!stats || !gamma || !beta || !a || !d || B <= 0 || C <= 0 || HW <= 0) return SDY_ERR_ARG;
  const int BC = B * C;
  unsigned* flags = nullptr;
  SDY_TRY(sdy_flags_ptr(&flags));
  hipLaunchKernelGGL(instnorm_from_stats_kernel, dim3((BC + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, BC, C, HW,
                     gamma, beta, ss, ss_stride, eps, a, d, flags);
  return sdy_launch_status();
}

int sdy_instnorm_from_partials_launch(const double* part, int K, int B, int C, int HW, const float* gamma, const float* beta,
                                      float eps, float* a, float* d, hipStream_t stream) {
  if (!part || !gamma || !beta || !a || !d || K <= 0 || B <= 0 || C <= 0 || HW <= 0) return SDY_ERR_ARG;
  const int BC = B * C;
  unsigned* flags = nullptr;
  SDY_TRY(sdy_flags_ptr(&flags));
  hipLaunchKernelGGL(instnorm_from_partials_kernel, dim3((BC + 255) / 256), dim3(256), 0, stream, part, K, BC, C, HW, gamma, beta, eps,
                     a, d, flags);
  return sdy_launch_status();
}

int sdy_gelu_stats_launch(const float* y, long y_bs, float* out, long out_bs, int out_tiled, double* stats, int B, int C,
                          int HW, hipStream_t stream) {
  if (!y || !out || B <= 0 || C <= 0 || HW <= 0) return SDY_ERR_ARG;
  if ((HW & 3) || (y_bs & 3) || (out_bs & 3)) return SDY_ERR_ALIGN;
  hipLaunchKernelGGL(gelu_stats_kernel, dim3(C, B), dim3(256), 0, stream, y, y_bs, out, out_bs, out_tiled, stats, C, HW);
  return sdy_launch_status();
}

int sdy_affine_copy_stats_launch(const float* x, long x_bs, const float* a, const float* d, float* out, long out_bs,
                                 double* stats, int C, int HW, const unsigned char* rows, int n_rows, hipStream_t stream,
                                 int src_row0) {
  if (!x || (a == nullptr) != (d == nullptr) || !out || !rows || C <= 0 || HW <= 0 || n_rows <= 0) return SDY_ERR_ARG;
  if ((HW & 3) || (x_bs & 3) || (out_bs & 3)) return SDY_ERR_ALIGN;
  SdyImgMap m;
  SDY_TRY(sdy_img_map_fill(m, rows, n_rows));
  hipLaunchKernelGGL(affine_copy_stats_kernel, dim3(C, n_rows), dim3(256), 0, stream, x, x_bs, a, d, out, out_bs, stats, C, HW, m,
                     src_row0);
  return sdy_launch_status();
}

int sdy_concat_launch(const float* const* src, const int* chans, int nsrc, float* out, long out_bstride, int B, int HW,
                      hipStream_t stream, int src_rows) {
  if (!src || !chans || !out || nsrc < 1 || nsrc > 4 || B <= 0 || HW <= 0) return SDY_ERR_ARG;
  if (HW & 3) return SDY_ERR_ALIGN;
  ConcatArgs a;
  int total = 0;
  for (int i = 0; i < 4; ++i) {
    a.src[i] = i < nsrc ? src[i] : nullptr;
    a.chans[i] = i < nsrc ? chans[i] : 0;
    if (i < nsrc) {
      if (!src[i] || chans[i] <= 0) return SDY_ERR_ARG;
      total += chans[i];
    }
  }
  a.nsrc = nsrc;
  const int HW4 = HW / 4;
  hipLaunchKernelGGL(concat_kernel, dim3((HW4 + 255) / 256, total, B), dim3(256), 0, stream, a, out, out_bstride, HW4, src_rows);
  return sdy_launch_status();
}

int sdy_cold_update_launch(const float* xs, const float* xn, const float* xi, float* out, size_t n,
                           hipStream_t stream) {
  if (!xs || !xn || !out || n == 0) return SDY_ERR_ARG;
  size_t blocks = ((n >> 2) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cold_update_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, xs, xn, xi, out, n);
  return sdy_launch_status();
}

int sdy_time_mlp_launch(const SdyTimeMlp& t, const float* time, int B, float* trep, float* ss, float* dp,
                        const float* dp_keep_in, int enable_dropout, uint64_t seed, uint32_t call,
                        uint32_t batch_offset, int rows_per_call, hipStream_t stream, float* scratch) {
  if (rows_per_call < 1) rows_per_call = B > 0 ? B : 1;
  const uint32_t slo = (uint32_t)(seed & 0xFFFFFFFFu), shi = (uint32_t)(seed >> 32);
  if (t.E > 0) {
    // scratch: B * T floats for the hidden layer, B * T more when the caller does not want t_repr
    if (!time || !scratch || !ss) return SDY_ERR_ARG;
    float* h1 = scratch;
    float* tr = trep ? trep : scratch + (size_t)B * t.T;
    const int rb = (B + TM_ROWS - 1) / TM_ROWS;
    auto smem = [&](int K) { return (size_t)(TM_ROWS * (K < TM_KC ? (K + 3) & ~3 : TM_KC) + 4 * TM_ROWS * 64) * sizeof(float); };
    hipLaunchKernelGGL(time_dense_kernel<TM_EMB_GELU>, dim3((t.T + 63) / 64, rb), dim3(256), smem(t.E), stream, t, time, h1, B,
                       nullptr, nullptr, 0, 0u, 0u, 0u, 0u, 1);
    hipLaunchKernelGGL(time_dense_kernel<TM_PLAIN>, dim3((t.T + 63) / 64, rb), dim3(256), smem(t.T), stream, t, h1, tr, B,
                       nullptr, nullptr, 0, 0u, 0u, 0u, 0u, 1);
    hipLaunchKernelGGL(time_dense_kernel<TM_SILU_LAYERS>, dim3(t.num_layers * ((2 * t.E + 63) / 64), rb), dim3(256), smem(t.T),
                       stream, t, tr, ss, B, dp, dp_keep_in, enable_dropout, slo, shi, call, batch_offset, rows_per_call);
  } else if (dp) {
    const int n = B * t.num_layers;
    hipLaunchKernelGGL(droppath_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, t, dp, dp_keep_in, B, enable_dropout,
                       slo, shi, call, batch_offset, rows_per_call);
  }
  return sdy_launch_status();
}

int sdy_spec_to_torch_launch(const float* Cs, float* out, int B, int C, int L, int mtr, int Mfull, hipStream_t stream) {
  hipLaunchKernelGGL(spec_to_torch_kernel, dim3(1024), dim3(256), 0, stream, Cs, out, B, C, L, mtr, Mfull);
  return sdy_launch_status();
}
int sdy_torch_to_spec_launch(const float* in, float* Cs, int B, int C, int L, int mtr, int Mfull, hipStream_t stream) {
  hipLaunchKernelGGL(torch_to_spec_kernel, dim3(1024), dim3(256), 0, stream, in, Cs, B, C, L, mtr, Mfull);
  return sdy_launch_status();
}
