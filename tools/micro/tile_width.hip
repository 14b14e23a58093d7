// How much HBM bandwidth does an NCHW tensor give when it is streamed as [256 channels][TN pixels] tiles (one tile per
// workgroup, persistent workgroups, 16-byte loads, 16-byte stores of a same-shaped output)?  TN * 4 bytes is the length
// of the contiguous run per channel row.  Build: hipcc --offload-arch=gfx950 -O3 tile_width.hip -o tile_width
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TN, int WGS_PER_CU>
__global__ __launch_bounds__(256) void stream_tiles(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int B) {
  constexpr int QN = TN / 4;            // 16-byte quads per row of the tile
  constexpr int RPP = 256 / QN;         // rows covered per pass
  const int tid = threadIdx.x, q = tid % QN, r0 = tid / QN;
  const int tpi = HW / TN, ntiles = tpi * B;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int z = t / tpi, n0 = (t - z * tpi) * TN;
    const float* xb = x + (long)z * C * HW + n0 + 4 * q;
    float* yb = y + (long)z * C * HW + n0 + 4 * q;
    constexpr int NI = (256 / RPP) < 16 ? (256 / RPP) : 16;   // C = 256
    f32x4 v[NI];
    for (int p0 = 0; p0 < C / RPP; p0 += NI) {
#pragma unroll
      for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const f32x4*>(xb + (long)(r0 + (p0 + i) * RPP) * HW);
#pragma unroll
      for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(yb + (long)(r0 + (p0 + i) * RPP) * HW) = v[i] * 2.0f;
    }
  }
}

template <int TN, int W>
void run(const float* x, float* y, int C, int HW, int B) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * W;
  float best = 1e9;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_tiles<TN, W>), dim3(grid), dim3(256), 0, 0, x, y, C, HW, B);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes = 2.0 * B * C * (double)(HW / TN * TN) * 4;
  printf("TN=%4d (%4d B runs) WGs/CU=%d: %.3f ms  %.0f GB/s (read+write)\n", TN, TN * 4, W, best, bytes / best / 1e6);
}

int main() {
  const int B = 8, C = 256, HW = 64800;
  float *x, *y;
  hipMalloc(&x, (size_t)B * C * HW * 4); hipMalloc(&y, (size_t)B * C * HW * 4);
  hipMemset(x, 0, (size_t)B * C * HW * 4);
  run<32, 4>(x, y, C, HW, B); run<64, 2>(x, y, C, HW, B); run<64, 4>(x, y, C, HW, B); run<64, 8>(x, y, C, HW, B);
  run<128, 2>(x, y, C, HW, B); run<128, 4>(x, y, C, HW, B); run<128, 8>(x, y, C, HW, B);
  run<256, 4>(x, y, C, HW, B); run<256, 8>(x, y, C, HW, B);
  return 0;
}
