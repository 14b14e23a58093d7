// Launchers of pointwise.hip (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct SdyTimeMlp {
  int E;            // embed_dim (0 = network has no time embedding)
  int T;            // time_dim
  int num_layers;
  const float* freq;  // dev [E/2]   exp(arange(E/2) * -log(1e4)/(E/2-1))
  const float* w1t;   // dev [E][T]
  const float* b1;    // dev [T]
  const float* w2t;   // dev [T][T]
  const float* b2;    // dev [T]
  const float* wbt;   // dev [L][T][2E]
  const float* bb;    // dev [L][2E]
  float dp_rate[32];
  uint32_t dp_thr[32];
};

int sdy_instnorm_coeffs_launch(const float* x, int B, int C, int HW, const float* gamma, const float* beta,
                               const float* ss, long ss_stride, float eps, float* a, float* d, hipStream_t stream);
int sdy_concat_launch(const float* const* src, const int* chans, int nsrc, float* out, long out_bstride, int B, int HW,
                      hipStream_t stream, int src_rows = 0);   // src_rows > 0: output row b reads source row b % src_rows
int sdy_cold_update_launch(const float* xs, const float* xn, const float* xi, float* out, size_t n, hipStream_t stream);
// scratch: dev, B * T floats (2 B T when trep is NULL) -- the hidden layer between the launches
int sdy_time_mlp_launch(const SdyTimeMlp& t, const float* time, int B, float* trep, float* ss, float* dp,
                        const float* dp_keep_in, int enable_dropout, uint64_t seed, uint32_t call,
                        uint32_t batch_offset, int rows_per_call, hipStream_t stream, float* scratch);
int sdy_spec_to_torch_launch(const float* Cs, float* out, int B, int C, int L, int mtr, int Mfull, hipStream_t stream);
int sdy_torch_to_spec_launch(const float* in, float* Cs, int B, int C, int L, int mtr, int Mfull, hipStream_t stream);
// InstanceNorm coefficients (no time scale / shift) from per-ring partial statistics part[b][k][c][2] (fft360.hip's act epilogue)
int sdy_instnorm_from_partials_launch(const double* part, int K, int B, int C, int HW, const float* gamma, const float* beta,
                                      float eps, float* a, float* d, hipStream_t stream);
// out = GELU(y) (exact erf), B images of C channels; `stats` (dev double [B][C][2], zeroed by the caller, or null) receives the
// (sum, sum of squares) of what is stored; out_tiled: [b][64-pixel tile][C][64] with out_bs floats per image, else NCHW
int sdy_gelu_stats_launch(const float* y, long y_bs, float* out, long out_bs, int out_tiled, double* stats, int B, int C,
                          int HW, hipStream_t stream);
// drop-path skip (capi.hip): out[b] = a[b] * x[b] + d[b] for the n_rows batch rows b = rows[i] (host array, <= 128 entries),
// with the (sum, sum of squares) of every stored plane ADDED to stats[b] (dev double [B][C][2]) unless stats is null.
// a, d null: a plain copy.  src_row0 >= 0: the source of rows[i] is row src_row0 + i of x (a tensor in the launch's own order).
int sdy_affine_copy_stats_launch(const float* x, long x_bs, const float* a, const float* d, float* out, long out_bs,
                                 double* stats, int C, int HW, const unsigned char* rows, int n_rows, hipStream_t stream,
                                 int src_row0 = -1);
