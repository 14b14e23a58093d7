// How many bytes per clock can one CU pull from L2 with 16-byte-per-lane loads?  (The fragment-stream kernels move 0.5-1 MB
// of L2-resident weights per tile through this path; dh_h3's ablations put its cost at ~11-16k cycles per MB.)
// Every workgroup re-reads the same 1 MB buffer (L2 resident) with independent 1 KB wave loads, 8 loads in flight per wave.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/l2_stream.hip -o /tmp/l2_stream && /tmp/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const f32x4* __restrict__ buf, float* out, unsigned long long* cyc, int iters, int n16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    // each wave walks its own slice of the buffer, 8 independent loads per step
    for (int base = wave * 512; base + 512 <= n16; base += nw * 512) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = buf[base + u * 64 + lane];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  const int n16 = 65536;   // 1 MB
  f32x4* buf; float* out; unsigned long long* cyc;
  hipMalloc(&buf, n16 * 16); hipMemset(buf, 0, n16 * 16);
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 50;
  for (int threads : {256, 512, 1024}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, buf, out, cyc, 2, n16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, buf, out, cyc, iters, n16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double bytes_per_cu = (double)iters * n16 * 16;
    printf("threads/CU %4d: %.3f ms, %.1f TB/s aggregate, %.1f B/clk/CU (wave-0 cycles %llu)\n", threads, ms,
           bytes_per_cu * 256 / ms * 1e-9, bytes_per_cu / (double)c, c);
  }
  return 0;
}
