// Shared host/device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>

#include "../../include/sdy_amd.h"

#define SDY_HIP_TRY(expr)                       \
  do {                                          \
    hipError_t _e = (expr);                     \
    if (_e != hipSuccess) return (int)_e;       \
  } while (0)

#define SDY_TRY(expr)             \
  do {                            \
    int _r = (expr);              \
    if (_r != SDY_OK) return _r;  \
  } while (0)

// launch check that does not synchronise
static inline int sdy_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SDY_OK : (int)e;
}

// Per-DEVICE launch caches: one process may drive several devices (sfno.py keeps one native handle per device index), so
// nothing about "the" device is cached per process.  hipGetDevice is a thread-local read.
constexpr int SDY_MAX_DEVICES = 64;
static inline int sdy_current_device(int* dev) {
  SDY_HIP_TRY(hipGetDevice(dev));
  return (*dev >= 0 && *dev < SDY_MAX_DEVICES) ? SDY_OK : SDY_ERR_ARG;
}
inline int sdy_cu_count(int* n_cu) {                 // compute units of the CURRENT device
  static std::atomic<int> cache[SDY_MAX_DEVICES] = {};   // (atomics: two host threads may issue a device's first launch)
  int dev = 0;
  SDY_TRY(sdy_current_device(&dev));
  int v = cache[dev].load(std::memory_order_relaxed);
  if (!v) {
    SDY_HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    cache[dev].store(v, std::memory_order_relaxed);
  }
  *n_cu = v;
  return SDY_OK;
}
// Device address of the current device's sticky status word (allocated and zeroed on first use; pointwise.hip).
int sdy_flags_ptr(unsigned** flags);
// Range headroom (include/sdy_amd.h, sdy_range_headroom): device word of consumer class `slot` while the debug read-back is
// enabled, nullptr otherwise (kernels skip the bookkeeping on a null pointer).
enum { SDY_RANGE_CONV = 0, SDY_RANGE_MLP = 1, SDY_RANGE_DHCONV = 2, SDY_RANGE_LEG_ANALYSIS = 3, SDY_RANGE_LEG_SYNTHESIS = 4 };
int sdy_headroom_ptr(int slot, unsigned** word);
struct SdyOncePerDevice {                            // `static SdyOncePerDevice once;` next to a kernel's attribute setup
  std::atomic<bool> done[SDY_MAX_DEVICES] = {};      // (setting a function attribute twice is harmless; the flag is not a lock)
  int slot(std::atomic<bool>** flag) {
    int dev = 0;
    SDY_TRY(sdy_current_device(&dev));
    *flag = &done[dev];
    return SDY_OK;
  }
};

// Image map of the drop-path skip (capi.hip, sdy_sfno_forward): a block whose DropPath draw zeroes the branch of some
// trajectories (src/models/modules/drop_path.py:15-22) runs on the ACTIVE trajectories only.  Its per-block intermediates are
// indexed compactly (j = 0 .. n_active - 1); the tensors that live across blocks keep their batch row idx[j].  Passed by value
// in the kernel arguments (a uniform index into the kernarg segment: scalar loads); `on == 0` is the identity.
constexpr int SDY_MAP_MAX = 128;
struct SdyImgMap {
  int on;
  unsigned char idx[SDY_MAP_MAX];
};
__host__ __device__ __forceinline__ int sdy_img(const SdyImgMap& m, int j) { return m.on ? (int)m.idx[j] : j; }
static inline int sdy_img_map_fill(SdyImgMap& m, const unsigned char* rows, int n) {   // rows == nullptr: identity
  m.on = 0;
  for (int i = 0; i < SDY_MAP_MAX; ++i) m.idx[i] = 0;
  if (!rows) return SDY_OK;
  if (n > SDY_MAP_MAX) return SDY_ERR_UNSUPPORTED;
  m.on = 1;
  for (int i = 0; i < n; ++i) m.idx[i] = rows[i];
  return SDY_OK;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Compile-time loop: f(std::integral_constant<int, I>) for I = B .. E - 1.  `#pragma unroll` gives up on long bodies
// ("unrolled size too large"), and register arrays indexed by the loop variable then live in scratch.
template <int B, int E, class F>
__device__ __forceinline__ void sdy_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    sdy_static_for<B + 1, E>(f);
  }
}
// fp16 hi / lo split of 8 fp32 values (already scaled) for the split-precision MFMA kernels, written on 2-vectors so that it
// compiles to packed conversions (v_cvt_pk_f16_f32, v_pk_add_f32 / fma_mix) instead of ~6 scalar VALU ops per element.
typedef float sdy_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sdy_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sdy_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void sdy_split8(const float* v, sdy_f16x8& hi, sdy_f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const sdy_f32x2 x = {v[e], v[e + 1]};
    const sdy_f16x2 h = __builtin_convertvector(x, sdy_f16x2);
    const sdy_f32x2 r = x - __builtin_convertvector(h, sdy_f32x2);
    const sdy_f16x2 l = __builtin_convertvector(r, sdy_f16x2);
    hi[e] = h[0]; hi[e + 1] = h[1];
    lo[e] = l[0]; lo[e + 1] = l[1];
  }
}
// The same, also tracking max |v| of everything a thread splits: fp16 overflows at 65504, and an overflowed `hi` turns into
// inf / NaN products silently.  Kernels end with sdy_flag_range(flags, amax) (one v_max3_f32 per two values here; NaNs are
// ignored by the max and are caught by the InstanceNorm statistics instead).
__device__ __forceinline__ void sdy_split8(const float* v, sdy_f16x8& hi, sdy_f16x8& lo, float& amax) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v[e]), __builtin_fabsf(v[e + 1])));
  sdy_split8(v, hi, lo);
}
// Activation pre-scale of every split-fp16 kernel (x -> SDY_ACT_SX * x before the hi / lo split, undone in the accumulator
// scale).  fp16 overflows at 65504, so a staged |value| >= 65504 / SDY_ACT_SX sets SDY_FLAG_F16_RANGE; below
// 2^-3 / SDY_ACT_SX the `lo` part is an fp16 subnormal and the split keeps an ABSOLUTE precision of 2^-25 / SDY_ACT_SX
// instead of 22 relative bits.  One constant for all kernels (EXTRA=-DSDY_ACT_SX=... builds a variant for A/B runs).
#ifndef SDY_ACT_SX
#define SDY_ACT_SX 16.0f
#endif
// Sticky status word of a device (include/sdy_amd.h, sdy_status_flags): bits are only ever set by kernels.
#define SDY_F16_LIMIT 65504.0f
__device__ __forceinline__ void sdy_flag_range(unsigned* flags, float amax) {
  if (flags && amax >= SDY_F16_LIMIT) atomicOr(flags, (unsigned)SDY_FLAG_F16_RANGE);
}
// The same plus the debug read-back of sdy_range_headroom: `head` (null unless enabled) keeps the largest staged magnitude
// its consumer class has seen, as float bits (non-negative floats order like unsigned integers).
__device__ __forceinline__ void sdy_flag_range(unsigned* flags, float amax, unsigned* head) {
  sdy_flag_range(flags, amax);
  if (head) {
    const unsigned bits = __float_as_uint(amax);
    if (bits > *reinterpret_cast<volatile unsigned*>(head)) atomicMax(head, bits);
  }
}
// 16-byte store of a streaming output (written once, read by a later kernel after 1.6 GB of other traffic)
#ifdef SDY_NT_STORE
#define SDY_STREAM_STORE(ptr, v) __builtin_nontemporal_store((v), reinterpret_cast<f32x4*>(ptr))
#else
#define SDY_STREAM_STORE(ptr, v) (*reinterpret_cast<f32x4*>(ptr) = (v))
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
// InstanceNorm statistics of a stored 4-pixel quad as fp32 partials (summed in fp64 by the callers).  Spelled out with explicit
// FMAs: the fused MLP's epilogue and the drop-path copy kernel (pointwise.hip) must round alike, or a block output written by one
// instead of the other would change the next block's coefficients in the last bit (left to -ffp-contract, one kernel came out
// as fma(x, x, y*y), the other as fma(y, y, x*x)).
__device__ __forceinline__ float sdy_quad_sum(f32x4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float sdy_quad_sumsq(f32x4 v) {
  return __builtin_fmaf(v.x, v.x, v.y * v.y) + __builtin_fmaf(v.z, v.z, v.w * v.w);
}
// sum over the 16 lanes of a DPP row (all of them end up with the total)
__device__ __forceinline__ float row16_sum(float x) {
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  x += dpp(x, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  x += dpp(x, std::integral_constant<int, 0x141>{});   // row_half_mirror
  x += dpp(x, std::integral_constant<int, 0x140>{});   // row_mirror
  return x;
}

// MEASUREMENT BUILDS ONLY (csrc/Makefile EXTRA=-DSDY_H3_PASSES=1): drop the two cross terms Ah.Bl + Al.Bh of every
// split-precision product, i.e. single-pass f16 MFMA arithmetic (fp16-class accuracy: ~1e-3) with everything else -- the
// hi / lo splits, the loads of the lo fragments, the schedules -- unchanged.  Bounds what the three passes cost and gives the
// "bf16-class" number BASELINE.json's configs[1] label asks about; never the shipped arithmetic.
// In-kernel phase stamps (s_memtime per phase of one sampled workgroup) and their read-back entry points
// sdy_*_debug_stamps exist in MEASUREMENT BUILDS ONLY (csrc/Makefile EXTRA=-DSDY_STAMPS, tools/*_stamps.py): the product
// library carries neither the stamp branches in its tile loops nor the entry points.
#ifdef SDY_STAMPS
#define SDY_STAMPS_ON 1
#else
#define SDY_STAMPS_ON 0
#endif
#define SDY_DEBUG_EXPORT extern "C" __attribute__((visibility("default")))

#ifndef SDY_H3_PASSES
#define SDY_H3_PASSES 3
#endif
#define SDY_CROSS_TERM(stmt) do { if (SDY_H3_PASSES == 3) { stmt; } } while (0)

// 16-byte global access as (wave-uniform 64-bit base in SGPRs) + (32-bit byte offset per lane): the SADDR form of
// global_load / global_store.  The persistent kernels address rows as image base + row * HW + lane part; written as plain
// pointer arithmetic hipcc re-associates that into one 64-bit lane address PER ROW, hoists all of them out of the tile loop
// and spills them.  The empty asm pins the base to SGPRs (so the row step is scalar arithmetic) and hides it from the
// re-association; the explicit global address space keeps the access from degrading to FLAT.  `ubase` must be wave-uniform.
typedef const char __attribute__((address_space(1)))* sdy_gcptr_t;
typedef char __attribute__((address_space(1)))* sdy_gptr_t;
typedef f32x4 __attribute__((address_space(1))) sdy_gf32x4;
__device__ __forceinline__ f32x4 sdy_ld16s(const float* ubase, unsigned off_b) {
  sdy_gcptr_t b = (sdy_gcptr_t)ubase;
  asm volatile("" : "+s"(b), "+v"(off_b));   // (the offset too: its zero-extension must sit in the block of the access)
  return *reinterpret_cast<const sdy_gf32x4*>(b + off_b);
}
__device__ __forceinline__ void sdy_st16s(float* ubase, unsigned off_b, f32x4 v) {
  sdy_gptr_t b = (sdy_gptr_t)ubase;
  asm volatile("" : "+s"(b), "+v"(off_b));
#ifdef SDY_NT_STORE
  __builtin_nontemporal_store(v, reinterpret_cast<sdy_gf32x4*>(b + off_b));
#else
  *reinterpret_cast<sdy_gf32x4*>(b + off_b) = v;
#endif
}

// One fragment (16 bytes per lane) of a per-wave weight stream as (wave-uniform stream base in SGPRs) + (the lane's running
// 32-bit byte offset) + (an immediate < 4 KB): the SADDR form of global_load, no VALU per load.  The fragment-stream kernels
// used a per-lane 64-bit pointer; every refill then carried a v_add_co / v_addc pair, and a refill behind an MFMA delayed the
// next one: tools/micro/mfma_two_roles.hip measures 35.7 cycles per MFMA for the pointer form against 32.9 for this one (38.7
// when a group's two refills are issued back to back).  `off` is laundered in place (a laundered copy is one more live
// register); the caller steps it by the group size (one v_add per group).
__device__ __forceinline__ sdy_f16x8 sdy_ring_ld(const void* ubase, unsigned& off, int imm) {
  sdy_gcptr_t b = (sdy_gcptr_t)ubase;
  asm volatile("" : "+s"(b), "+v"(off));
  return *reinterpret_cast<const sdy_f16x8 __attribute__((address_space(1)))*>(b + off + imm);
}

// ---- exact-erf GELU (nn.GELU default, src/models/sfno/sfnonet.py:602-603) ------------------------------
// erfc(z), z >= 0, by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7): one v_rcp_f32, one v_exp_f32 and a
// 5-term Horner chain instead of ocml's branchy erff (~4x fewer VALU cycles in the GEMM epilogues).  Using the
// erfc form on the negative side avoids the 1 - erf cancellation, so GELU keeps ~2e-7 relative accuracy there too.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float q = fmaf(1.061405429f, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  q = q * t * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);  // erfc(z)
  const float hq = 0.5f * q;
  return x >= 0.0f ? x * (1.0f - hq) : x * hq;
}
// Two values at once: the Horner chain, the products and the final blend as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 run at
// the scalar rate for two lanes of data); rcp and exp2 stay scalar.  Same arithmetic as gelu_erf, term by term.
typedef float sdy_gf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sdy_gf2 gelu_erf2(sdy_gf2 x) {
  const sdy_gf2 ax = __builtin_elementwise_abs(x);
  const sdy_gf2 z = ax * 0.70710678118654752440f;
  const sdy_gf2 d = z * 0.3275911f + 1.0f;
  const sdy_gf2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  sdy_gf2 q = t * 1.061405429f + -1.453152027f;
  q = q * t + 1.421413741f;
  q = q * t + -0.284496736f;
  q = q * t + 0.254829592f;
  const sdy_gf2 e2 = z * z * -1.44269504088896340736f;
  const sdy_gf2 e = {__builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y)};
  const sdy_gf2 hq = q * t * e * 0.5f;                 // erfc(z) / 2
  // x >= 0: x (1 - hq);  x < 0: x hq   ==   0.5 (x + |x|) - |x| hq ... written as x*hq' with hq' = (x >= 0 ? 1 - hq : hq)
  const sdy_gf2 pos = x * (1.0f - hq), neg = x * hq;
  return sdy_gf2{x.x >= 0.0f ? pos.x : neg.x, x.y >= 0.0f ? pos.y : neg.y};
}
// ---- exact-erf GELU from a table: 16 gelu(v) with NO transcendental and 9 plain VALU instructions ------------------
// The persistent kernels are VALU-issue bound: gelu_erf above costs ~14 plain instructions plus v_rcp and
// v_exp (8 issue cycles each).  Here Phi(v) = (1 + erf(v / sqrt 2)) / 2 is a cubic Taylor piece around the nearest of 385 nodes
// spaced 1/32 apart on [-6, 6] -- |error| <= 0.55 * (1/64)^4 / 24 = 1.4e-9, far below the 7.5e-8 of the A&S formula -- with
// the four coefficients of a node in ONE 16-byte LDS word.  The kernels work on w = 8 v (their accumulator scale and bias
// absorb the 8), which makes every multiplier an inline constant and leaves one scalar operand per instruction, the most a
// gfx9 VOP3 encoding takes (with v itself the constants 32.0 and +-6.0 need VGPRs):
//     wc = med3(w, -48, 48)                    clamp (the end nodes hold exactly 0 and 1 with zero slope)
//     t  = fma(wc, 4, 2^23 + 192)              the node index lands in the low mantissa bits (round to nearest)
//     a  = (bits(t) << 4) + const              its byte address: one v_lshl_add_u32, the constant cancels the exponent bits
//     f  = fma(wc, 4, -(t - MAGIC))            offset from the node in node spacings, |f| <= 1/2 (exact)
//     16 gelu(v) = w * (c0 + f (c1 + f (c2 + f c3)))        coefficients pre-multiplied by 16 / 8
// The table is built once per device in fp64 (pointwise.hip) and copied into LDS by the kernels that use it.
#define SDY_GELU_NODES 385
#define SDY_GELU_TAB_BYTES (SDY_GELU_NODES * 16)
#define SDY_GELU_SX SDY_ACT_SX       // activation pre-scale of the split-fp16 kernels: the table yields SDY_GELU_SX * gelu
#define SDY_GELU_WS 8.0f        // w = SDY_GELU_WS * v
#define SDY_GELU_MAGIC (8388608.0f + 192.0f)
int sdy_gelu_table_ptr(const float** table_dev);   // device pointer of the current device's table ([385][4] floats)
typedef const f32x4 __attribute__((address_space(3)))* sdy_lds_f32x4_cptr;
__device__ __forceinline__ f32x4 gelu_tab_load(unsigned addr) { return *reinterpret_cast<sdy_lds_f32x4_cptr>((uintptr_t)addr); }
// w = SDY_GELU_WS * v; tab_lds_base = LDS byte address of the table; returns SDY_GELU_SX * gelu(v)
__device__ __forceinline__ float gelu_tab16(float w, unsigned tab_lds_base) {
  const float wc = __builtin_amdgcn_fmed3f(w, -6.0f * SDY_GELU_WS, 6.0f * SDY_GELU_WS);
  const float t = fmaf(wc, 4.0f, SDY_GELU_MAGIC);
  const f32x4 c = gelu_tab_load((__builtin_bit_cast(unsigned, t) << 4) + (tab_lds_base - 0xB0000000u));   // bits(2^23) << 4
  const float f = fmaf(wc, 4.0f, -(t - SDY_GELU_MAGIC));
  return w * fmaf(fmaf(fmaf(c.w, f, c.z), f, c.y), f, c.x);
}
// cooperative copy of the table into LDS (dst 16-byte aligned; follow with a barrier).  `scale` multiplies the coefficients:
// 1 gives SDY_GELU_SX * gelu from gelu_tab16, 1 / SDY_GELU_SX plain gelu.
__device__ __forceinline__ void gelu_tab_to_lds(float* dst, const float* tab_global, int tid, int nthreads, float scale = 1.0f) {
  for (int i = tid; i < SDY_GELU_NODES; i += nthreads)
    reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(tab_global)[i] * scale;
}

__device__ __forceinline__ float gelu_erf_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

// ---- Philox4x32-7 (dropout stream, see include/sdy_amd.h) ----------------------------------------------
// Seven rounds: the smallest member of the family that passes BigCrush (Salmon, Moraes, Dror, Shaw: "Parallel random numbers:
// as easy as 1, 2, 3", SC'11, table 2; Random123's philox4x32_R<7>); ten is that paper's safety margin.  A round is two
// quarter-rate v_mad_u64_u32 and four XORs (gfx950 has no three-input XOR), and the fused MLP with dropout is VALU-issue bound:
// measured in the network, 10 -> 7 rounds takes 3.02 -> 2.91 ms off its launch (profiles/r5a/e2e_ab_philox7.txt).
#ifndef SDY_PHILOX_ROUNDS
#define SDY_PHILOX_ROUNDS 7
#endif
struct philox4 {
  uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ philox4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < SDY_PHILOX_ROUNDS; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return philox4{c0, c1, c2, c3};
}
// Element dropout takes 16 bits per decision: one Philox call (4 words) covers 4 channels x the pixel pair (n, n + 32),
// low half-word for the pixel with bit 5 clear, high half-word for its partner (include/sdy_amd.h).  The pair is what one
// lane of a 64-pixel MFMA tile holds, so every call is computed once and used 8 times; the 64-bit multiplies of Philox
// run at quarter rate and were 27 % of the fused MLP with dropout.
static inline uint32_t sdy_drop_threshold16(float p) {
  double t = (double)p * 65536.0;
  if (t >= 65535.0) return 0xFFFFu;
  if (t <= 0.0) return 0u;
  return (uint32_t)t;
}
// (written so that neither half needs an extraction: the high half-word compares as the whole word against thr16 << 16 -- the
//  low bits cannot change the outcome --, the low one as a 16-bit compare (v_cmp_*_u16 reads the low halves of its operands);
//  two instructions per decision with the select instead of three, in kernels that are VALU-issue bound with dropout on)
__host__ __device__ __forceinline__ bool sdy_keep16(uint32_t word, int half, uint32_t thr16) {
  return half ? (word >= (thr16 << 16)) : ((uint16_t)word >= (uint16_t)thr16);
}
static inline uint32_t sdy_drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t >= 4294967295.0) return 0xFFFFFFFFu;
  if (t <= 0.0) return 0u;
  return (uint32_t)t;
}

// ---- GEMM launcher (gemm.hip) --------------------------------------------------------------------------
enum { SDY_TRI_NONE = 0, SDY_TRI_LEG_FWD = 1, SDY_TRI_LEG_INV = 2, SDY_TRI_DHCONV = 3 };
enum { SDY_TILE_128x128 = 0, SDY_TILE_64x128 = 1 };

// C[z][m][n] = epilogue( sum_k A[z][k][m] * prologue(B[z][k][n]) )
struct GemmParams {
  const float* A; const float* B; float* C;
  int M, N, K;         // loader extents; M is the padded row count of A (multiple of 4 when A is k-major)
  int M_store;         // rows actually stored (<= M)
  int lda, ldb, ldc;
  long sA, sB, sC;     // batch strides (floats)
  int nbatch;
  int a_kcontig;       // 0: A stored [K][M] (m contiguous); 1: A stored [M][K] (k contiguous)
  int b_cplx;          // 1: B is packed complex weight [z][2][E][E], expanded to [[wr, wi], [-wi, wr]]
  int cplx_Ei, cplx_Eo;
  int tri_mode, tri_B;
  int tile;
  int tag;             // profiler label only (see gemm.hip)
  // prologue affine on B rows
  const float* pa; const float* pd; long p_bstride;
  // epilogue
  const float* bias;
  const float* add; long sAdd; int ldadd; int add_mode;
  int act;
  uint32_t drop_thr; float drop_scale; const float* keep_mask;
  uint32_t seed_lo, seed_hi, stream_id, call, batch_offset;
  int rows_per_call;   // >= 1: batch row z draws the stream of (call + z / rows_per_call, trajectory batch_offset + z % rows_per_call)
  const float* batch_scale;
};
int sdy_gemm_launch(const GemmParams& p, hipStream_t stream);
// split-fp16 (3-pass) conv GEMM, gemm_h3.hip; packed = fp16 [Mpad][Kpad] hi then lo, scaled by w_scale
//   rows_mode 0: packed = A [M][K], p.B = fp32 [K][N];  rows_mode 1: p.A = fp32 [M][K] (k contiguous), packed = B [N][K]
int sdy_gemm_h3_launch(const GemmParams& p, const void* packed, int rows_pad, int Kpad, long bstride, long plane_halfs,
                       float w_scale, int rows_mode, hipStream_t stream);

// ---- persistent 256 -> 256 convolution (conv_h3.hip)
int sdy_conv256_h3_launch(const sdy_conv_args* a, hipStream_t stream);
extern "C" int sdy_conv256_h3_pack_cin(const float* w_host, int Cin, void* dev, float* scale);   // (256, Cin) weight, Cin <= 384
extern "C" size_t sdy_conv256_h3_pack_bytes_cin(int Cin);
// dh_h3.hip: fragment-stream pack of the dhconv weight; ilv = channel order of the 2C axis (fft.h)
int sdy_dh_h3_pack(const float* w_host, int L, void* packed_dev, float* scale, int ilv);
// B_in (tiled only; 0 = B): the input tensor holds B_in >= B images per order and the first B of them are contracted
int sdy_dh_h3_launch(const float* Cs_in, const void* packed, float scale, float* Cs_out, int L, int mtr, int B, int ilv,
                      hipStream_t stream, int tiled = 0, int B_in = 0);

// ---- skinny Legendre GEMM with the table streamed as MFMA fragments (leg_h3.hip), rows and K <= 192
typedef float (*sdy_leg_value_fn)(void* ctx, int z, int row, int k);
int sdy_leg_h3_supported(int rows, int K);
size_t sdy_leg_h3_table_bytes(int nz);
int sdy_leg_h3_pack(int nz, int rows, int K, sdy_leg_value_fn value, void* ctx, void* dev, float* scale);
// leg_par.hip: the same transforms with the equatorial symmetry folded in (even nlat)
int sdy_leg_par_supported(int nlat, int lmax);
size_t sdy_leg_par_table_bytes(int nz);
int sdy_leg_par_pack(int nz, int nlat, int lmax, int fwd, sdy_leg_value_fn value, void* ctx, void* dev, float* scale);
int sdy_leg_par_launch(const void* table, float scale, int nz, const float* X, long ldx, long sX, float* C, long ldc,
                       long sC, int rows_out, int K, int N, int fwd, const int* kdead, hipStream_t stream, long tsx = 0, long tsc = 0);
int sdy_leg_h3_launch(const void* table, float scale, int nz, const float* X, long ldx, long sX, float* C, long ldc, long sC,
                      int M_store, int K, int N, int tri, hipStream_t stream);

// ---- host tables (tables.cpp) --------------------------------------------------------------------------
int sdy_factor_radices(int n, int* radices, int* nstages);  // n = prod(radices), radices in {4,2,3,5}
