// Descriptor shared between the SHT plan (plan.hip) and the FFT kernels (fft.hip).
#pragma once
#include <hip/hip_runtime.h>

struct SdyFftDesc {
  int n;           // complex length = nlon / 2
  int S;           // LDS row stride (odd, >= n + 1)
  int nstages;
  int radices[10]; // product = n, each in {2,3,4,5}
  const float* tw; // dev [n][2]    exp(-2*pi*i*j/n)
  const float* pw; // dev [n+1][2]  exp(-2*pi*i*m/nlon)
  int guard_f16;   // the consumer of the forward transform stages it as fp16 (split-precision Legendre kernels): fft360 raises
                   // SDY_FLAG_F16_RANGE for the folded analysis kernel, which has no register left for a tracker of its own
};

// Polar cut-off (fused forward only): for latitude ring k only the orders m < mcut[k] are written (forward) / read (inverse);
// the Legendre tables are negligible (< 1e-12 of their maximum) beyond it, see sdy_sht_plan::d_kdead.  nullptr = all orders.

// x_rows (host, [B], or nullptr = identity; fft360 only, B <= 128): row b of Xf is the transform of batch row x_rows[b] of
// x / a / d / xn_out -- the drop-path skip of the fused forward runs a block on its active trajectories only (common.h, SdyImgMap)
// x_mod > 0 (fft360 only): x holds x_mod rows and batch row r reads x row r % x_mod (stacked calls that share their inputs)
int sdy_fft_launch_fwd(const SdyFftDesc& f, const float* x, const float* a, const float* d, float* xn_out, float* Xf,
                       int B, int C, int K, int mtr, int ilv, const int* mcut, hipStream_t stream,
                       const unsigned char* x_rows = nullptr, int x_mod = 0);
int sdy_fft_launch_inv(const SdyFftDesc& f, const float* Yf, const float* bias, float* y, int B, int C, int K, int mtr,
                       int ilv, const int* mcut, hipStream_t stream);

// ilv selects the channel order of the m-major side (Xf / Yf, 2C floats per (m, k, b)):
//   0: [ri][c]            -- the layout of the C ABI (include/sdy_amd.h)
//   1: [c / 16][ri][16]   -- a workgroup's 16 channels x (re, im) form ONE 128-byte line instead of two 64-byte half
//                            lines 4*C bytes apart (C % 16 == 0).  The Legendre stages are agnostic (flat columns); the
//                            dhconv weights are packed for the same order (capi.hip).  Used inside the fused forward.
//   2: TILE-MAJOR (fft360 + leg_par only): the channel order of 1, and the plane of one order m cut into column tiles of 64
//      (two channel blocks x (re, im) x 16) that are stored whole, [m][tile j][k][64] instead of [m][k][2 B C] -- the
//      Legendre kernel's activation tile (all latitudes x 64 columns of one order) is then ONE contiguous 46 KB block instead
//      of 180 runs of 256 bytes 2 B C floats apart.  C % 32 == 0.  The generic kernels refuse it.
// nlon = 360 specialisation (fft360.hip); SDY_ERR_UNSUPPORTED when the shape does not fit
int sdy_fft360_launch_fwd(const SdyFftDesc& f, const float* x, const float* a, const float* d, float* xn_out, float* Xf,
                          int B, int C, int K, int mtr, int ilv, const int* mcut, hipStream_t stream,
                          const unsigned char* x_rows = nullptr, int x_mod = 0);
// zt / part (both or neither; fft360 only): instead of y (B, C, K, nlon) the kernel stores GELU(ring + bias) TILE-MAJOR
// ([b][64-pixel tile][C][64], zt_bs floats per image) and the ring's (sum, sum of squares) per channel into
// part[b][k][c][2] (doubles, one writer per slot) -- the act + norm1 statistics of a block whose inner skip is folded into
// its dhconv weights (capi.hip, skip_foldable)
int sdy_fft360_launch_inv(const SdyFftDesc& f, const float* Yf, const float* bias, float* y, int B, int C, int K,
                          int mtr, int ilv, const int* mcut, hipStream_t stream, float* zt = nullptr, long zt_bs = 0,
                          double* part = nullptr);
