// Access-pattern probe: stream an NCHW tensor through CUs with the same tile shapes as the conv GEMM (K x 128-pixel
// tiles read with 16-byte loads, 128 x 128 output tiles written 4 bytes per lane) but no arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// pattern A: per k-tile of 64 channels: thread (kq = tid%16, pq = tid/16) loads rows 4kq..4kq+3, pixels 4pq.. (2 blocks)
template <int WRITE_MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ x, float* __restrict__ out, int K, int Cout, int HW) {
  const int tid = threadIdx.x, z = blockIdx.z;
  const int n0 = blockIdx.y * 128, m0 = blockIdx.x * 128;
  const float* xb = x + (long)z * K * HW;
  f32x4 acc = {0, 0, 0, 0};
  const int kq = tid % 16, pq = tid / 16;
  for (int k0 = 0; k0 < K; k0 += 64) {
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const int gn = n0 + (bi * 16 + pq) * 4;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int gk = k0 + kq * 4 + kk;
        if (gn < HW) acc += *reinterpret_cast<const f32x4*>(xb + (long)gk * HW + gn);
      }
    }
  }
  float* ob = out + (long)z * Cout * HW;
  const int lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, h = lane >> 5, l31 = lane & 31;
  if (WRITE_MODE == 0) {   // MFMA accumulator pattern: 4 B per lane, 2 rows x 128 B per instruction
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int r = 0; r < 16; ++r) {
          const int gm = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int gn = n0 + wc * 64 + j * 32 + l31;
          if (gn < HW) ob[(long)gm * HW + gn] = acc[r & 3];
        }
  } else {                 // 16 B per lane: row = tid/32 (+8 per step), 32 lanes x 16 B = 512 B per row
    for (int rr = 0; rr < 16; ++rr) {
      const int gm = m0 + rr * 8 + (tid >> 5);
      const int gn = n0 + (tid & 31) * 4;
      if (gn < HW) *reinterpret_cast<f32x4*>(ob + (long)gm * HW + gn) = acc;
    }
  }
}

int main() {
  const int B = 8, K = 256, Cout = 256, HW = 64800;
  float *x, *o;
  hipMalloc(&x, (size_t)B * K * HW * 4);
  hipMalloc(&o, (size_t)B * Cout * HW * 4);
  hipMemset(x, 0, (size_t)B * K * HW * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int mt = 1; mt <= 2; ++mt) {
      dim3 grid(mt, (HW + 127) / 128, B);
      for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, grid, dim3(256), 0, 0, x, o, K, Cout, HW);
        else hipLaunchKernelGGL(probe<1>, grid, dim3(256), 0, 0, x, o, K, Cout, HW);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double rd = (double)B * K * HW * 4 * mt, wr = (double)B * 128 * mt * HW * 4;
      printf("write_mode=%d m_tiles=%d: %.3f ms  read %.2f GB write %.2f GB  -> %.2f TB/s\n", mode, mt, ms, rd / 1e9, wr / 1e9, (rd + wr) / ms / 1e9);
    }
  return 0;
}
