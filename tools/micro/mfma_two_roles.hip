// Warp specialisation on one SIMD: does a wave that only issues MFMAs (+ its LDS fragment reads and L2 weight loads, the fc2
// loop of mlp_h3) keep the matrix pipe at 32 cycles per MFMA while a SECOND wave on the same SIMD runs the GELU chain (plain
// VALU + v_rcp / v_exp + LDS writes)?  One workgroup of 8 waves per CU: waves 0-3 = role C (MFMA), waves 4-7 = role P (VALU).
// Printed: cycles per 6-MFMA group of C alone, of P per chain piece alone, and of both when they run together.
//   make -C tools/micro mfma_two_roles && tools/micro/mfma_two_roles
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const f16x8 __attribute__((address_space(1)))* wptr_t;

constexpr int RING = 8;

template <int MODE>   // bit 0: role C runs, bit 1: role P runs, bit 2: role P raises its priority (s_setprio 3),
                      // bit 3: C without its LDS fragment reads, bit 4: C without its weight refills, bit 5: the refills land in registers no MFMA reads, bit 6: refills pinned one behind an MFMA (k = 4, 5),
                      // bit 7: pinned refills addressed as (SGPR base) + (constant lane offset): no VALU per load
__global__ __launch_bounds__(512, 1) void kern(const f16x8* w, float* out, unsigned long long* ticks, int iters, int iters_p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* Hs = reinterpret_cast<_Float16*>(smem);            // 64 KB: fragments read by C
  _Float16* Ps = Hs + 32768;                                   // 32 KB: written by P
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 49152; i += 512) Hs[i] = (_Float16)((i * 7 % 13) * 0.01f);
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  float res = 0.f;
  if (wave < 4) {
    if (MODE & 1) {
      f16x8 r_hi[RING], r_lo[RING], d_hi[RING], d_lo[RING];
      wptr_t wp = (wptr_t)(w + (size_t)wave * 4096 + lane);     // 64 KB of weights per wave, L2 resident, re-read in a cycle
#pragma unroll
      for (int s = 0; s < RING; ++s) { r_hi[s] = wp[s * 128]; r_lo[s] = wp[s * 128 + 64]; d_hi[s] = r_hi[s]; d_lo[s] = r_lo[s]; }
      f32x16 acc[2][2];
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      f16x8 bh[2][2], bl[2][2];
      const int l31 = lane & 31, h = lane >> 5;
      const unsigned lane_off = (unsigned)lane * 16u;
      auto ldb1 = [&](int set, int t, int what) {
        const int j = what & 1;
        const int off = (32 * j + l31) * 128 + (((2 * t + h) ^ (l31 & 15)) & 15) * 8;
        if (what < 2) bh[set][j] = *reinterpret_cast<const f16x8*>(Hs + off);
        else bl[set][j] = *reinterpret_cast<const f16x8*>(Hs + 8192 + off);
      };
      for (int what = 0; what < 4; ++what) ldb1(0, 0, what);
      t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters; ++it) {
        const int base = (it & 3) * RING * 128;                 // cycle through 32 groups (64 KB)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int cur = t & 1;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const int s = 2 * t + mi;
            const f16x8 a_lo = r_lo[s], a_hi = r_hi[s];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              const int j = k & 1;
              acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k < 2 ? a_lo : a_hi, (k >= 2 && k < 4) ? bl[cur][j] : bh[cur][j],
                                                                 acc[mi][j], 0, 0, 0);
              if (!(MODE & 8) && mi == 0 && k < 4) ldb1(cur ^ 1, (t + 1) & 3, k);
              if ((MODE & 64) && k == 4) r_lo[s] = wp[base + s * 128 + 64];
              if ((MODE & 64) && k == 5) r_hi[s] = wp[base + s * 128];
              if ((MODE & 128) && k >= 4) {
                typedef const char __attribute__((address_space(1)))* gcp;
                gcp b = (gcp)(w + (size_t)wave * 4096) + (size_t)(base + s * 128 + (k == 4 ? 64 : 0)) * 16;
                unsigned lo = lane_off;
                asm volatile("" : "+s"(b), "+v"(lo));
                const f16x8 val = *reinterpret_cast<const f16x8 __attribute__((address_space(1)))*>(b + lo);
                if (k == 4) r_lo[s] = val; else r_hi[s] = val;
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            if (!(MODE & 16) && !(MODE & 64) && !(MODE & 128)) {
              if (MODE & 32) { d_hi[s] = wp[base + s * 128]; d_lo[s] = wp[base + s * 128 + 64]; }
              else { r_hi[s] = wp[base + s * 128]; r_lo[s] = wp[base + s * 128 + 64]; }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      t1 = __builtin_amdgcn_s_memtime();
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) res += acc[a][b][r];
      for (int s = 0; s < RING; ++s) res += (float)d_hi[s][0] + (float)d_lo[s][1];
    }
  } else {
    if (MODE & 2) {
      if (MODE & 4) __builtin_amdgcn_s_setprio(3);
      float v[4], t[4], e[4], q[4];
      for (int r = 0; r < 4; ++r) v[r] = 0.001f * (tid + r);
      t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters_p; ++it) {                     // one chain piece per iteration: the arithmetic of mlp_h3's chain_slot
        constexpr float CT = 0.2316f, KAP = -0.7213f, HS = 8.0f, A5 = 8.49f, A4 = -11.6f, A3 = 11.4f, A2 = -2.28f, A1 = 2.04f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[r] = fmaf(__builtin_fabsf(v[r]), CT, 1.0f); e[r] = v[r] * KAP * v[r]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[r] = __builtin_amdgcn_rcpf(t[r]); e[r] = __builtin_amdgcn_exp2f(e[r]); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          q[r] = fmaf(t[r], A5, A4); q[r] = fmaf(q[r], t[r], A3); q[r] = fmaf(q[r], t[r], A2); q[r] = fmaf(q[r], t[r], A1);
          q[r] = fmaf(-q[r] * t[r], e[r], HS) * __builtin_fabsf(v[r]);
          v[r] = fmaf(v[r], HS, q[r]);
        }
        f16x4 vh, vl;
#pragma unroll
        for (int r = 0; r < 4; ++r) { vh[r] = (_Float16)v[r]; vl[r] = (_Float16)(v[r] - (float)vh[r]); }
        const int off = ((tid - 256) * 4 + (it & 7) * 1024) & 8191;
        *reinterpret_cast<f16x4*>(Ps + off) = vh;
        *reinterpret_cast<f16x4*>(Ps + 8192 + off) = vl;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * 1e-3f + 0.01f * r;
      }
      t1 = __builtin_amdgcn_s_memtime();
      for (int r = 0; r < 4; ++r) res += v[r];
    }
  }
  out[blockIdx.x * 512 + tid] = res;
  if (lane == 0) { ticks[(blockIdx.x * 8 + wave) * 2] = t0; ticks[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
}

template <int MODE>
static void run(const char* name, const f16x8* w, float* out, unsigned long long* ticks, int n_cu, int iters_c, int iters_p) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&kern<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kern<MODE>), dim3(n_cu), dim3(512), 98304, 0, w, out, ticks, iters_c, iters_p);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(n_cu * 16);
  hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> c, p;
  for (int b = 0; b < n_cu; ++b) {
    c.push_back((double)(h[(b * 8 + 0) * 2 + 1] - h[(b * 8 + 0) * 2]) / (iters_c ? iters_c : 1) / 8.0);     // per 6-MFMA group
    p.push_back((double)(h[(b * 8 + 4) * 2 + 1] - h[(b * 8 + 4) * 2]) / (iters_p ? iters_p : 1));
  }
  std::sort(c.begin(), c.end()); std::sort(p.begin(), p.end());
  printf("%-28s C: %7.1f cycles per 6-MFMA group (%.1f per MFMA)   P: %7.1f cycles per chain piece\n", name,
         (MODE & 1) ? c[n_cu / 2] : 0.0, (MODE & 1) ? c[n_cu / 2] / 6 : 0.0, (MODE & 2) ? p[n_cu / 2] : 0.0);
}

// both roles with the SAME iteration count run different amounts of work; the host picks counts so that they end together
int main() {
  int n_cu = 0;
  hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
  f16x8* w; float* out; unsigned long long* ticks;
  hipMalloc(&w, 4 * 4096 * sizeof(f16x8) + 65536); hipMalloc(&out, (size_t)n_cu * 512 * 4); hipMalloc(&ticks, (size_t)n_cu * 16 * 8);
  std::vector<_Float16> hw(4 * 4096 * 8 + 32768);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)((i % 17) * 0.01f);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  run<1>("C alone", w, out, ticks, n_cu, 2000, 0);
  run<2>("P alone", w, out, ticks, n_cu, 0, 16000);
  run<3>("C + P (P saturating)", w, out, ticks, n_cu, 2000, 12000);  // P busy for the whole of C's run
  run<3>("C + P (P at 1/3 duty)", w, out, ticks, n_cu, 2000, 3000);   // mlp_h3's ratio: ~1 chain piece per 15-20 MFMAs
  run<7>("C + P, P at priority 3", w, out, ticks, n_cu, 2000, 12000);
  run<7>("C + P prio 3, 1/3 duty", w, out, ticks, n_cu, 2000, 3000);
  run<1 + 8>("C alone, no LDS reads", w, out, ticks, n_cu, 2000, 0);
  run<1 + 16>("C alone, no refills", w, out, ticks, n_cu, 2000, 0);
  run<1 + 8 + 16>("C alone, MFMAs only", w, out, ticks, n_cu, 2000, 0);
  run<1 + 32>("C alone, refills to spare regs", w, out, ticks, n_cu, 2000, 0);
  run<1 + 64>("C alone, refills pinned", w, out, ticks, n_cu, 2000, 0);
  run<1 + 128>("C alone, pinned, SGPR base", w, out, ticks, n_cu, 2000, 0);
  return 0;
}
