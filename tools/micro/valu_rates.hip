// Issue cost of single VALU instructions on gfx950 as the fused kernels see it: one wave per SIMD (256 threads, one
// workgroup per CU), 8 independent chains of one instruction, 64 instructions per loop trip, timed with s_memtime.
// Prints core cycles per instruction (per wave).  Build + run on the GPU box:
//   make -C tools/micro valu_rates && tools/micro/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// each kernel: v[8] floats (or uint pairs) updated in place by the instruction under test
#define KERNEL(NAME, ASM, CONSTR)                                                              \
  __global__ void NAME(float* out, unsigned long long* ticks, int iters) {                     \
    float v[8], w[8];                                                                          \
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i; w[i] = 1.0f + i * 0.125f; } \
    double d[8];                                                                               \
    for (int i = 0; i < 8; ++i) d[i] = v[i];                                                   \
    (void)d;                                                                                   \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                      \
    for (int it = 0; it < iters; ++it) {                                                       \
      REP64(ASM)                                                                               \
    }                                                                                          \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                      \
    float s = 0.f;                                                                             \
    for (int i = 0; i < 8; ++i) s += v[i] + w[i] + (float)d[i];                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;                               \
  }

#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define A_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define A_MIX(i) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(w[i]));
#define A_CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_CVTF(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i]));
#define A_CVTSDWA(i) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(v[i]));
#define A_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
#define A_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define A_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(d[i]) : "v"(v[i]), "v"(w[i]) : "vcc");
#define A_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_XOR3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(v[i]) : "v"(w[i]));
#define A_ADDF64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define A_CVTF64(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(v[i]));
#define A_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_DPP(i) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(w[i]));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(w[i]));
#define A_CMP(i) asm volatile("v_cmp_ge_u32 vcc, %0, %1" : : "v"(v[i]), "v"(w[i]) : "vcc");
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
#define A_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(v[i]) : "v"(w[i]));

KERNEL(k_fma, A_FMA, 0)
KERNEL(k_pkfma, A_PKFMA, 0)
KERNEL(k_pkmul, A_PKMUL, 0)
KERNEL(k_mix, A_MIX, 0)
KERNEL(k_cvtpk, A_CVTPK, 0)
KERNEL(k_cvtf, A_CVTF, 0)
KERNEL(k_cvtsdwa, A_CVTSDWA, 0)
KERNEL(k_rcp, A_RCP, 0)
KERNEL(k_exp, A_EXP, 0)
KERNEL(k_mulhi, A_MULHI, 0)
KERNEL(k_mullo, A_MULLO, 0)
KERNEL(k_mad64, A_MAD64, 0)
KERNEL(k_mul24, A_MUL24, 0)
KERNEL(k_mulhi24, A_MULHI24, 0)
KERNEL(k_mad24, A_MAD24, 0)
KERNEL(k_xor, A_XOR, 0)
KERNEL(k_xor3, A_XOR3, 0)
KERNEL(k_addf64, A_ADDF64, 0)
KERNEL(k_cvtf64, A_CVTF64, 0)
KERNEL(k_max3, A_MAX3, 0)
KERNEL(k_dpp, A_DPP, 0)
KERNEL(k_perm, A_PERM, 0)
KERNEL(k_cndmask, A_CNDMASK, 0)
KERNEL(k_cmp, A_CMP, 0)
KERNEL(k_lshladd, A_LSHLADD, 0)
KERNEL(k_add3, A_ADD3, 0)
KERNEL(k_alignbit, A_ALIGNBIT, 0)

typedef void (*kern_t)(float*, unsigned long long*, int);

int main() {
  float* out;
  unsigned long long* ticks;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMalloc(&ticks, 8);
  const int iters = 4000;
  struct { const char* name; kern_t k; } list[] = {
      {"v_fma_f32", k_fma}, {"v_pk_fma_f32", k_pkfma}, {"v_pk_mul_f32", k_pkmul}, {"v_fma_mix_f32", k_mix},
      {"v_cvt_pk_f16_f32", k_cvtpk}, {"v_cvt_f32_f16", k_cvtf}, {"v_cvt_f32_f16_sdwa", k_cvtsdwa},
      {"v_rcp_f32", k_rcp}, {"v_exp_f32", k_exp}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_lo_u32", k_mullo},
      {"v_mad_u64_u32", k_mad64}, {"v_mul_u32_u24", k_mul24}, {"v_mul_hi_u32_u24", k_mulhi24}, {"v_mad_u32_u24", k_mad24},
      {"v_xor_b32", k_xor}, {"v_bitop3_b32", k_xor3}, {"v_add_f64", k_addf64}, {"v_cvt_f64_f32", k_cvtf64},
      {"v_max3_f32", k_max3}, {"v_add_f32_dpp", k_dpp}, {"v_perm_b32", k_perm}, {"v_cndmask_b32", k_cndmask},
      {"v_cmp_ge_u32", k_cmp}, {"v_lshl_add_u32", k_lshladd}, {"v_add3_u32", k_add3}, {"v_alignbit_b32", k_alignbit},
  };
  for (int threads : {256, 512}) {
    printf("== %d threads per CU (%d wave(s) per SIMD): cycles per instruction and wave\n", threads, threads / 256);
    for (auto& e : list) {
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, ticks, 10);   // warm
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, ticks, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long t;
      hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      printf("  %-22s %6.2f ticks/instr   (%.3f ms, %.0f MHz tick rate)\n", e.name, (double)t / (iters * 64.0), ms, t / ms * 1e-3);
    }
  }
  return 0;
}
