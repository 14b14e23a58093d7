// Which f16 MFMA shape for a VALU-heavy loop?  v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 at EQUAL flops and an
// EQUAL number of independent VALU instructions beside them, on random operands, one wave per SIMD on every CU:
//     A: { 1 x 32x32x16 ; N x v_fma_f32 }          32768 flop per group
//     B: { 2 x 16x16x32 ; N x v_fma_f32 }          2 x 16384 flop per group (N split N/2 + N - N/2 behind the two)
// Printed per N: shader cycles per group (s_memtime), the clock the chip holds (s_memtime / s_memrealtime x 100 MHz, median
// over workgroups) and chip TFLOP/s by wall time.  MI355X_MICROARCH.md ('DVFS give-back' item 7) measures 1.12-1.15x the
// FLOP/s on the 16x16x32 shape in BARE MFMA loops at equal cycles per flop, because the chip holds a higher clock on it; an
// MFMA also holds the SIMD's issue port for 8 of its 32 (32x32x16) or 8 of its 16 (16x16x32) cycles, so per 32 cycles of
// matrix pipe the big shape leaves 24 cycles of issue for other instructions and the small one 16 -- in a loop whose bound is
// VALU ISSUE (mlp_h3: ~5 VALU instructions per MFMA) that is what decides.
//   make -C tools/micro mfma_shape && tools/micro/mfma_shape
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VALU1(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i) & 7]) : "v"(w));
#define V0
#define V1 VALU1(0)
#define V2 V1 VALU1(1)
#define V3 V2 VALU1(2)
#define V4 V3 VALU1(3)
#define V5 V4 VALU1(4)
#define V6 V5 VALU1(5)

#define BIG(k) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(big[(k) & 1]) : "v"(a), "v"(b));
#define SMALL(k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(sm[(k) & 7]) : "v"(a), "v"(b));

template <int MODE, int N>   // MODE 0: big shape, 1: small shape; N VALU per group (0, 4, 6, 8, 10, 12)
__global__ __launch_bounds__(256) void kern(float* out, unsigned long long* ticks, int iters, const f16x8* rnd) {
  f16x8 a = rnd[threadIdx.x], b = rnd[256 + threadIdx.x];
  f32x16 big[2];
  f32x4 sm[8];
  for (int k = 0; k < 2; ++k) for (int r = 0; r < 16; ++r) big[k][r] = 0.f;
  for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) sm[k][r] = 0.f;
  float v[8], w = 1.0001f;
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#define GROUP(k, VA, VB)                                            \
    if (MODE == 0) { BIG(k) VA VB } else { SMALL(2 * (k)) VA SMALL(2 * (k) + 1) VB }
#define GROUPN(k)                                                   \
    if (N == 0) { GROUP(k, V0, V0) } else if (N == 4) { GROUP(k, V2, V2) } else if (N == 6) { GROUP(k, V3, V3) }   \
    else if (N == 8) { GROUP(k, V4, V4) } else if (N == 10) { GROUP(k, V5, V5) } else { GROUP(k, V6, V6) }
    GROUPN(0) GROUPN(1) GROUPN(2) GROUPN(3) GROUPN(0) GROUPN(1) GROUPN(2) GROUPN(3)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int k = 0; k < 2; ++k) for (int r = 0; r < 16; ++r) s += big[k][r];
  for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) s += sm[k][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE, int N>
static void run(const char* name, float* out, unsigned long long* ticks, const f16x8* rnd, int n_cu) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {   // the second launch is timed (clocks settled)
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<MODE, N>), dim3(n_cu), dim3(256), 0, 0, out, ticks, iters, rnd);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * n_cu);
  hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> cyc, clk;
  for (int i = 0; i < n_cu; ++i) { cyc.push_back((double)h[2 * i] / (iters * 8.0)); clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const double flops = 32768.0 * 8.0 * iters * 4.0 * n_cu;   // 4 waves per CU
  printf("%-22s N=%2d  cycles/group %6.1f  clock %.2f GHz  %7.1f TFLOP/s (wall %.2f ms)\n", name, N, cyc[n_cu / 2], clk[n_cu / 2],
         flops / (ms * 1e-3) / 1e12, ms);
}

int main() {
  int n_cu = 0;
  hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; unsigned long long* ticks; f16x8* rnd;
  hipMalloc(&out, (size_t)n_cu * 256 * 4); hipMalloc(&ticks, (size_t)n_cu * 16); hipMalloc(&rnd, 512 * sizeof(f16x8));
  std::vector<_Float16> h(512 * 8);
  unsigned s = 12345u;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 0.001f); }   // uniform [-1, 1]
  hipMemcpy(rnd, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
#define BOTH(N) run<0, N>("32x32x16", out, ticks, rnd, n_cu); run<1, N>("2 x 16x16x32", out, ticks, rnd, n_cu);
  BOTH(0) BOTH(4) BOTH(6) BOTH(8) BOTH(10) BOTH(12)
  return 0;
}
