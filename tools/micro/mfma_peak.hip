// Dense-MFMA ceiling of the chip as it actually runs (clock management included): back-to-back
// v_mfma_f32_32x32x16_f16 on register operands, no memory traffic.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float* out, unsigned long long* ticks, int iters) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.01f); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&ticks, 8);
  const int iters = 20000;
  for (int threads : {256, 512, 1024}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, ticks, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      const double flop = 256.0 * (threads / 64) * iters * 4 * 32768.0;
      const double mf_cycles = (double)(threads / 256) * iters * 4 * 32;   // per SIMD
      printf("threads/CU %4d: %.3f ms  %.1f TFLOP/s  s_memtime ticks %llu (%.1f MHz)  implied MFMA clock if 100%% busy %.2f GHz\n",
             threads, ms, flop / ms * 1e-9, t, t / ms * 1e-3, mf_cycles / ms * 1e-6);
    }
  }
  return 0;
}
