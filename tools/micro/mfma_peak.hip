// Calibration: sustained v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_f16 rates with operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_f32(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_f16(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(a0 + threadIdx.x * 1e-3f + j); b[j] = (_Float16)(b0 - threadIdx.x * 1e-3f - j); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 256 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc = 1; wpc <= 2; ++wpc) {
    int grid = 256 * wpc;  // blocks of 4 waves: wpc waves per SIMD
    for (int pass = 0; pass < 2; ++pass) {
      int iters = 20000;
      hipLaunchKernelGGL(k_f32, dim3(grid), dim3(256), 0, 0, d, 100, 0.5f, 0.25f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_f32, dim3(grid), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = (double)grid * 4 * iters * 16 * (32.0 * 32 * 2 * 2);
      printf("f32 32x32x2  waves/SIMD=%d: %.3f ms  %.1f TF/s\n", wpc, ms, fl / ms / 1e9);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_f16, dim3(grid), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      fl = (double)grid * 4 * iters * 16 * (32.0 * 32 * 16 * 2);
      printf("f16 32x32x16 waves/SIMD=%d: %.3f ms  %.1f TF/s\n", wpc, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
