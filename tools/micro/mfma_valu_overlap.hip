// Does ONE wave overlap its own MFMA with independent VALU work?  Loop of { v_mfma_f32_32x32x16_f16 ; N x VALU } with N = 0..8,
// for plain VGPR VALU, for VALU that reads accumulator registers of a DIFFERENT accumulator (v_accvgpr_read), and with the
// MFMA accumulators in VGPRs or AGPRs.  One wave per SIMD (256 threads per CU) and two (512).
//   make -C tools/micro mfma_valu_overlap && tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VALU1(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i) & 7]) : "v"(w));
#define REPV_0
#define REPV_1 VALU1(0)
#define REPV_2 REPV_1 VALU1(1)
#define REPV_3 REPV_2 VALU1(2)
#define REPV_4 REPV_3 VALU1(3)
#define REPV_5 REPV_4 VALU1(4)
#define REPV_6 REPV_5 VALU1(5)
#define REPV_8 REPV_6 VALU1(6) VALU1(7)
// four instructions of another kind in the shadow of each MFMA
#define PK1(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[(i) & 3]) : "v"(d[((i) + 1) & 3]));
#define REP_PK4 PK1(0) PK1(1) PK1(2) PK1(3)
#define REP_PK2 PK1(0) PK1(1)
#define RCP1(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[(i) & 7]));
#define REP_RCP2 RCP1(0) RCP1(1)
#define REP_RCP4 RCP1(0) RCP1(1) RCP1(2) RCP1(3)
#define CVT1(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[(i) & 7]) : "v"(w));
#define REP_CVT4 CVT1(0) CVT1(1) CVT1(2) CVT1(3)
#define ACCRD1(i) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[(i) & 7]) : "a"(spare));
#define REP_ACC4 ACCRD1(0) ACCRD1(1) ACCRD1(2) ACCRD1(3)
#define SNOP1(i) asm volatile("s_nop 0");
#define REP_NOP4 SNOP1(0) SNOP1(1) SNOP1(2) SNOP1(3)
#define SALU1(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) : : "scc");
#define REP_SALU4 SALU1(0) SALU1(1) SALU1(2) SALU1(3)
#define REP_SALU8 REP_SALU4 REP_SALU4
#define WAIT1(i) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
#define REP_WAIT4 WAIT1(0) WAIT1(1) WAIT1(2) WAIT1(3)
#define LDS1(i) asm volatile("ds_read_b128 %0, %1" : "=v"(q4[(i) & 1]) : "v"(ldsaddr));
#define REP_LDS2 LDS1(0) LDS1(1)
#define REP_LDS4 LDS1(0) LDS1(1) LDS1(0) LDS1(1)
#define REP_MIX REPV_2 PK1(0) CVT1(3) SALU1(0) WAIT1(0)
// integer multiplies of a Philox round
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d[(i) & 3]) : "v"(v[(i) & 7]), "v"(w) : "vcc");
#define REP_MAD2 MAD64(0) MAD64(1)
#define REP_MAD4 MAD64(0) MAD64(1) MAD64(2) MAD64(3)
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[(i) & 7]) : "v"(w));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[(i) & 7]) : "v"(w));
#define REP_MULHL4 MULHI(0) MULLO(1) MULHI(2) MULLO(3)
#define XOR3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(v[(i) & 7]) : "v"(w));
#define REP_XOR4 XOR3(0) XOR3(1) XOR3(2) XOR3(3)
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[(i) & 7]) : "v"(w), "s"(msk));
#define REP_CND4 CND(0) CND(1) CND(2) CND(3)
#define CMP(i) asm volatile("v_cmp_ge_u32 %0, %1, %2" : "=s"(msk) : "v"(v[(i) & 7]), "v"(w));
#define REP_CMPCND CMP(0) CND(1) CMP(2) CND(3)
// six DEPENDENT fp32 FMAs (one register), and the same as two chains of three / three chains of two
#define DEP1(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(w));
#define REP_DEP6 DEP1(0) DEP1(0) DEP1(0) DEP1(0) DEP1(0) DEP1(0)
#define REP_DEP4 DEP1(0) DEP1(0) DEP1(0) DEP1(0)
#define DEP2(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i) & 1]) : "v"(w));
#define REP_DEP2x3 DEP2(0) DEP2(1) DEP2(0) DEP2(1) DEP2(0) DEP2(1)
#define EXP1(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i) & 7]));
// a slot like those of the MLP chain: 2 transcendental + 4 FMAs that use their results
#define REP_SLOT RCP1(0) RCP1(1) VALU1(2) VALU1(3) VALU1(4) VALU1(5)
#define REP_SLOTDEP RCP1(0) RCP1(1) DEP2(0) DEP2(1) DEP2(0) DEP2(1)

// accumulators pinned to AGPRs ("a") or VGPRs ("v") through the asm constraint of the MFMA itself
#define MFMA_A(k) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a), "v"(b));
#define MFMA_V(k) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));

#define KERNEL(NAME, MF, REPV)                                                              \
  __global__ void NAME(float* out, unsigned long long* ticks, int iters) {                  \
    f16x8 a, b;                                                                             \
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.01f); } \
    f32x16 acc[2];                                                                          \
    for (int k = 0; k < 2; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;               \
    float v[8], w = 1.0001f;                                                                \
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;                              \
    double d[4]; for (int i = 0; i < 4; ++i) d[i] = v[i];                                   \
    float spare = v[3]; asm volatile("" : "+a"(spare));                                     \
    unsigned long long msk = 0x5555555555555555ull;                                         \
    unsigned sc = 0; typedef float f4 __attribute__((ext_vector_type(4))); f4 q4[2] = {{0,0,0,0},{0,0,0,0}}; \
    __shared__ float lds[1024]; lds[threadIdx.x & 1023] = v[0]; __syncthreads();            \
    unsigned ldsaddr = (threadIdx.x & 63) * 16;                                             \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
    for (int it = 0; it < iters; ++it) {                                                    \
      MF(0) REPV MF(1) REPV MF(0) REPV MF(1) REPV MF(0) REPV MF(1) REPV MF(0) REPV MF(1) REPV \
    }                                                                                       \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
    float s = 0.f;                                                                          \
    for (int k = 0; k < 2; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];                \
    for (int i = 0; i < 8; ++i) s += v[i];                                                  \
    for (int i = 0; i < 4; ++i) s += (float)d[i];                                           \
    s += spare + (float)sc + q4[0][0] + q4[1][1] + (float)(msk & 1);                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;                            \
  }

KERNEL(ka0, MFMA_A, REPV_0) KERNEL(ka1, MFMA_A, REPV_1) KERNEL(ka2, MFMA_A, REPV_2) KERNEL(ka3, MFMA_A, REPV_3)
KERNEL(ka4, MFMA_A, REPV_4) KERNEL(ka5, MFMA_A, REPV_5) KERNEL(ka6, MFMA_A, REPV_6) KERNEL(ka8, MFMA_A, REPV_8)
KERNEL(kpk2, MFMA_A, REP_PK2) KERNEL(kpk4, MFMA_A, REP_PK4) KERNEL(krcp2, MFMA_A, REP_RCP2) KERNEL(krcp4, MFMA_A, REP_RCP4)
KERNEL(kcvt4, MFMA_A, REP_CVT4) KERNEL(kacc4, MFMA_A, REP_ACC4) KERNEL(knop4, MFMA_A, REP_NOP4) KERNEL(ksalu4, MFMA_A, REP_SALU4)
KERNEL(ksalu8, MFMA_A, REP_SALU8) KERNEL(kwait4, MFMA_A, REP_WAIT4) KERNEL(klds2, MFMA_A, REP_LDS2) KERNEL(klds4, MFMA_A, REP_LDS4)
KERNEL(kmix, MFMA_A, REP_MIX)
KERNEL(kmad2, MFMA_A, REP_MAD2) KERNEL(kmad4, MFMA_A, REP_MAD4) KERNEL(kmulhl4, MFMA_A, REP_MULHL4) KERNEL(kxor4, MFMA_A, REP_XOR4)
KERNEL(kcnd4, MFMA_A, REP_CND4) KERNEL(kcmpcnd, MFMA_A, REP_CMPCND)
KERNEL(kdep6, MFMA_A, REP_DEP6) KERNEL(kdep4, MFMA_A, REP_DEP4) KERNEL(kdep2x3, MFMA_A, REP_DEP2x3) KERNEL(kslot, MFMA_A, REP_SLOT) KERNEL(kslotdep, MFMA_A, REP_SLOTDEP)
KERNEL(kv0, MFMA_V, REPV_0) KERNEL(kv2, MFMA_V, REPV_2) KERNEL(kv4, MFMA_V, REPV_4) KERNEL(kv6, MFMA_V, REPV_6) KERNEL(kv8, MFMA_V, REPV_8)


// ---- { 8 x MFMA ; L x global_load_dwordx4 (1 KB per wave, L2 hits) } : what a weight-ring refill costs the issuing wave
typedef float f4_t __attribute__((ext_vector_type(4)));
#define GLD(i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[(i) & 3]) : "v"(gp + 64 * ((i) & 7)) : "memory");
template <int L>
__global__ void kload(float* out, unsigned long long* ticks, int iters, const f4_t* src) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.01f); }
  f32x16 acc[2];
  for (int k = 0; k < 2; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  f4_t ld[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const f4_t* gp = src + (blockIdx.x & 7) * 4096 + (threadIdx.x & 63);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    MFMA_A(0) if (L >= 1) { GLD(0) } MFMA_A(1) if (L >= 5) { GLD(4) } MFMA_A(0) if (L >= 3) { GLD(2) } MFMA_A(1) if (L >= 7) { GLD(6) }
    MFMA_A(0) if (L >= 2) { GLD(1) } MFMA_A(1) if (L >= 6) { GLD(5) } MFMA_A(0) if (L >= 4) { GLD(3) } MFMA_A(1) if (L >= 8) { GLD(7) }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int k = 0; k < 2; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  for (int i = 0; i < 4; ++i) s += ld[i][0] + ld[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// ---- the fc1 loop of mlp_h3 in isolation: per k-step 6 MFMAs whose B operands were read from LDS during the previous
// k-step (two register sets, s_waitcnt before use) and whose A operands come from a ring refilled by two 1-KB global loads
template <int LDSR, int GLD_>
__global__ void kfc1(float* out, unsigned long long* ticks, int iters, const f4_t* src) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 16; i += blockDim.x) reinterpret_cast<f4_t*>(lds)[i] = f4_t{0.01f, 0.02f, 0.03f, 0.04f};
  __syncthreads();
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 ring_hi[4], ring_lo[4], bh[2][2], bl[2][2];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { ring_hi[i][e] = (_Float16)(0.01f * e + i); ring_lo[i][e] = (_Float16)(0.001f * e); }
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 8; ++e) { bh[a][b][e] = (_Float16)0.5f; bl[a][b][e] = (_Float16)0.25f; }
  f32x16 acc[2];
  for (int k = 0; k < 2; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  const unsigned la = (threadIdx.x & 63) * 16;
  const h8* gp = reinterpret_cast<const h8*>(src) + (blockIdx.x & 7) * 4096 + (threadIdx.x & 63);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks & 1;
      if (LDSR == 1 || LDSR == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bh[c ^ 1][j] = *reinterpret_cast<const h8*>(lds + ((la + 2048 * j + 4096 * ks) & 65535));
          bl[c ^ 1][j] = *reinterpret_cast<const h8*>(lds + ((la + 2048 * j + 4096 * ks + 32768) & 65535));
        }
        if (LDSR == 2) __builtin_amdgcn_sched_barrier(0);   // keep them a k-step ahead of their use (two register sets)
      }
      if (LDSR == 3) {   // ONE memory instruction pinned behind each MFMA: 4 LDS reads of the next k-step, 2 ring loads
        const h8 a_lo = ring_lo[ks], a_hi = ring_hi[ks];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int j = k & 1;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k < 2 ? a_lo : a_hi, (k >= 2 && k < 4) ? bl[c][j] : bh[c][j], acc[j], 0, 0, 0);
          if (k == 0) bh[c ^ 1][0] = *reinterpret_cast<const h8*>(lds + ((la + 4096 * ks) & 65535));
          if (k == 1) bh[c ^ 1][1] = *reinterpret_cast<const h8*>(lds + ((la + 2048 + 4096 * ks) & 65535));
          if (k == 2) bl[c ^ 1][0] = *reinterpret_cast<const h8*>(lds + ((la + 4096 * ks + 32768) & 65535));
          if (k == 3) bl[c ^ 1][1] = *reinterpret_cast<const h8*>(lds + ((la + 2048 + 4096 * ks + 32768) & 65535));
          if (GLD_ && k == 4) ring_lo[ks] = gp[ks * 128 + 64];
          if (GLD_ && k == 5) ring_hi[ks] = gp[ks * 128];
          __builtin_amdgcn_sched_barrier(0);
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring_lo[ks], bh[c][j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring_hi[ks], bl[c][j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring_hi[ks], bh[c][j], acc[j], 0, 0, 0);
      if (GLD_) {
        ring_hi[ks] = gp[ks * 128];
        ring_lo[ks] = gp[ks * 128 + 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int k = 0; k < 2; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

typedef void (*kern_t)(float*, unsigned long long*, int);
int main() {
  float* out; unsigned long long* ticks;
  (void)hipMalloc(&out, 256 * 1024 * sizeof(float)); (void)hipMalloc(&ticks, 8);
  const int iters = 4000;
  struct { const char* name; kern_t k; int n; } list[] = {
      {"acc in AGPRs", ka0, 0}, {"acc in AGPRs", ka1, 1}, {"acc in AGPRs", ka2, 2}, {"acc in AGPRs", ka3, 3}, {"acc in AGPRs", ka4, 4},
      {"acc in AGPRs", ka5, 5}, {"acc in AGPRs", ka6, 6}, {"acc in AGPRs", ka8, 8},
      {"2 x v_pk_fma_f32", kpk2, 2}, {"4 x v_pk_fma_f32", kpk4, 4}, {"2 x v_rcp_f32", krcp2, 2}, {"4 x v_rcp_f32", krcp4, 4},
      {"4 x v_cvt_pk_f16_f32", kcvt4, 4}, {"4 x v_accvgpr_read", kacc4, 4}, {"4 x s_nop", knop4, 4}, {"4 x s_add_u32", ksalu4, 4},
      {"8 x s_add_u32", ksalu8, 8}, {"4 x s_waitcnt (nothing pending)", kwait4, 4}, {"2 x ds_read_b128", klds2, 2}, {"4 x ds_read_b128", klds4, 4},
      {"2 fma + pk_fma + cvt_pk + s_add + s_waitcnt", kmix, 6},
      {"2 x v_mad_u64_u32", kmad2, 2}, {"4 x v_mad_u64_u32", kmad4, 4}, {"2 x (v_mul_hi_u32 + v_mul_lo_u32)", kmulhl4, 4},
      {"4 x v_bitop3_b32 (xor3)", kxor4, 4}, {"4 x v_cndmask_b32 (SGPR mask)", kcnd4, 4}, {"2 x (v_cmp_ge_u32 -> SGPR + v_cndmask)", kcmpcnd, 4},
      {"6 dependent v_fma_f32 (one chain)", kdep6, 6}, {"4 dependent v_fma_f32", kdep4, 4}, {"2 chains x 3 dependent v_fma_f32", kdep2x3, 6},
      {"2 v_rcp + 4 independent fma", kslot, 6}, {"2 v_rcp + 4 fma that read the rcp results", kslotdep, 6},
      {"acc in VGPRs", kv0, 0}, {"acc in VGPRs", kv2, 2}, {"acc in VGPRs", kv4, 4}, {"acc in VGPRs", kv6, 6}, {"acc in VGPRs", kv8, 8}};
  for (int threads : {256, 512}) {
    printf("== %d threads per CU (%d wave(s) per SIMD): core cycles per (MFMA + N x v_fma_f32), per wave\n", threads, threads / 256);
    for (auto& e : list) {
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, ticks, 10);
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, ticks, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      printf("  %s, N = %d: %6.1f ticks per group  (%.3f ms)\n", e.name, e.n, (double)t / (iters * 8.0), ms);
    }
  }
  f4_t* src; (void)hipMalloc(&src, 8 * 4096 * sizeof(f4_t) + 65536); (void)hipMemset(src, 0, 8 * 4096 * sizeof(f4_t) + 65536);
  for (int threads : {256, 512}) {
    printf("== %d threads per CU: core cycles per MFMA with L 1-KB global loads (L2 hits) per 8 MFMAs\n", threads);
    auto run = [&](auto kern, int L) {
      hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, ticks, 10, src);
      hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, ticks, iters, src);
      (void)hipDeviceSynchronize();
      unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      printf("  L = %d: %6.1f ticks per MFMA\n", L, (double)t / (iters * 8.0));
    };
    auto run2 = [&](auto kern, const char* what) {
      hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, ticks, 10, src);
      hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, ticks, iters, src);
      (void)hipDeviceSynchronize();
      unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      printf("  fc1-like loop, %s: %6.1f ticks per MFMA\n", what, (double)t / (iters * 24.0));
    };
    run2(kfc1<0, 0>, "registers only"); run2(kfc1<1, 0>, "+ 4 ds_read_b128 per 6 MFMAs feeding the next k-step");
    run2(kfc1<0, 1>, "+ 2 global loads per 6 MFMAs refilling the ring (4 k-steps ahead)"); run2(kfc1<1, 1>, "both");
    run2(kfc1<2, 0>, "+ 4 ds_read_b128 FENCED a k-step ahead"); run2(kfc1<2, 1>, "fenced LDS reads + ring loads");
    run2(kfc1<3, 0>, "one LDS read pinned behind each of 4 MFMAs"); run2(kfc1<3, 1>, "one LDS read / ring load pinned behind each MFMA");
    run(kload<0>, 0); run(kload<1>, 1); run(kload<2>, 2); run(kload<3>, 3); run(kload<4>, 4); run(kload<6>, 6); run(kload<8>, 8);
  }
  return 0;
}
