// Cin (<= 256) -> 256 channel 1x1 convolution (the block's inner skip, both encoder layers) as a persistent split-fp16
// kernel in the style of mlp_h3.hip:
//   out[b] = act( W . (pa[b]*x[b] + pd[b]) + bias + add_pre[b] ) + add_post[b]        (+ InstanceNorm statistics of out)
//   * one workgroup owns 64 pixels and all 256 output rows; the activation tile is fetched once, split hi/lo and
//     parked in LDS ([px][k], XOR-swizzled); wave w computes rows 64w .. 64w+63 (2 x 2 MFMA tiles, K = 256);
//   * the weight never touches LDS: packed per wave as a linear stream of MFMA A-fragment pairs (hi, lo) in consumption
//     order, L2 -> registers through an 8-group ring;
//   * the epilogue goes through LDS so that the addend loads and the stores are 16-byte row segments, applies
//     bias / addend / GELU there, and keeps per-thread (sum, sum of squares) of what it stores: the workgroup is
//     persistent (a contiguous range of tiles), so the statistics cost one fp64 atomic pair per row and IMAGE
//     instead of one per tile -- the next InstanceNorm (norm1) needs no pass of its own over the tensor.
// One workgroup per CU, software-pipelined over its tiles (see the tile loop).
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// explicit global address space: a pointer laundered through an empty asm loses its provenance, and hipcc then emits
// FLAT loads, which also count in lgkmcnt -- every LDS-fragment wait became lgkmcnt(0), i.e. a wait for the weight refills
typedef const f16x8 __attribute__((address_space(1)))* wptr_t;

namespace {

constexpr int CE = 256;          // channels in = out
constexpr int CTN = 64;          // pixels per tile
constexpr int CKS = CE / 16;     // k-steps
// Waves per workgroup.  8 = two per SIMD: the arbiter then fills one wave's LDS / memory waits and its VALU-only phases
// (fp16 split of the tile, GELU + statistics of the store loop) with the other wave's instructions; with 4 waves every
// phase of the tile ran alone on its SIMD (dh_h3.hip has the measurements that led here).
#ifndef SDY_CONV_NW
#define SDY_CONV_NW 8
#endif
constexpr int CNW = SDY_CONV_NW;
static_assert(CNW == 4 || CNW == 8, "4 or 8 waves");
constexpr int CNT = 64 * CNW;    // threads
constexpr int CMT = 8 / CNW;     // 32-row output tiles per wave
constexpr int COC = 8 / CNW;     // channel octets per thread in the staging role (octets o0 + 4 CNW * oc)
constexpr int CRPT = 64 / CNW;   // output rows per thread in the store loop (rows o0 + 4 CNW * i)
constexpr int CRS = 4 * CNW;     // = range of o0 = tid / 16
constexpr int CRING = 4 * CMT;   // groups in flight (4 k-steps x CMT m-tiles)
constexpr int CGPW = CMT * CKS;  // groups per wave
constexpr int CGROUP = 2 * 64;   // f16x8 elements per group
constexpr float CSX = 16.0f;

struct ConvParams {
  const float* x; long x_bs;
  const float* pa; const float* pd;
  const f16x8* w;                  // [CNW waves][CGPW groups][hi | lo][64 lanes]
  const float* bias;
  const float* add; long add_bs; int add_mode;   // 1: before the activation, 2: after it
  int act;
  float* out; long out_bs;
  double* stats;
  int HW, B;
  int Cin;                         // input channels (<= 256; the weight stream is zero-padded to KBLK * 64)
  float out_scale;
  unsigned* flags;                 // sticky status word (sdy_status_flags)
  unsigned long long* stamps;    // timing experiments only (SDY_CONV_STAMPS)
};

__device__ __forceinline__ int cv_swz(int px) { return (px & 15) ^ (((px >> 4) & 1) * 3); }
__device__ __forceinline__ int cv_off(int px, int c) { return px * CE + (((c & 16) | ((c ^ cv_swz(px)) & 15)) << 3); }

// KBLK = ceil(Cin / 64): k-blocks of 4 k-steps actually streamed (4 for the block's 256 -> 256 convs, 2 / 3 for the encoders)
template <int KBLK>
__global__ __launch_bounds__(CNT, 1) void conv_h3_kernel(const ConvParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * CTN * CE * 2];   // 64 KB
  _Float16* Xs_hi = reinterpret_cast<_Float16*>(smem);
  _Float16* Xs_lo = Xs_hi + CTN * CE;
  float* Os = reinterpret_cast<float*>(smem);   // epilogue: [256 rows][64 px] fp32 (aliases the x tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int h = lane >> 5, l31 = lane & 31;
  int q0 = tid & 15, o0 = tid >> 4;
  const int tpi = (p.HW + CTN - 1) / CTN;
  const int ntiles = tpi * p.B;
  // contiguous tile range per workgroup: a workgroup then crosses an image boundary at most ~once, so the per-image
  // flush of the statistics (and the reload of per-image coefficients) is rare instead of every few tiles
  const int t_per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * t_per;
  const int t_end = (t_begin + t_per < ntiles) ? t_begin + t_per : ntiles;

  f16x8 r_hi[CRING], r_lo[CRING];
  const wptr_t wbase = (wptr_t)(p.w + (size_t)wave * CGPW * CGROUP + lane);
  wptr_t wp = wbase;
#pragma unroll
  for (int s = 0; s < CRING; ++s) {
    r_hi[s] = wp[s * CGROUP];
    r_lo[s] = wp[s * CGROUP + 64];
  }
  wp += CRING * CGROUP;

  float brow[CRPT];          // bias of the rows this thread stores (rows tid / 16 + CRS i): the same for every tile
#pragma unroll
  for (int i = 0; i < CRPT; ++i) brow[i] = p.bias ? p.bias[o0 + CRS * i] : 0.0f;
  double psum[CRPT], psq[CRPT];   // statistics partials of those rows; fp64: sums of fp32 per-tile partials are then exact
#pragma unroll
  for (int i = 0; i < CRPT; ++i) { psum[i] = 0.0; psq[i] = 0.0; }

  // Software pipeline over tiles (one workgroup per CU, full register budget).  Loads are issued ONE per step, never as
  // a burst: a wave that issues more than the CU can keep in flight stalls at issue until HBM has delivered (a burst of
  // 32 KB per wave measured 9-20k idle cycles).  Tile t+1's pixels (xr) and addend rows (addv, reloaded right after their
  // use) both ride on tile t's store loop; the MFMA phase only streams the L2-resident weight ring.
  f32x4 xr[COC][8], addv[CRPT];
  auto x_ptr = [&](int t) {
    const int zz = t / tpi, nn = (t - zz * tpi) * CTN;
    return p.x + (long)zz * p.x_bs + ((nn + 4 * q0 < p.HW) ? nn + 4 * q0 : 0);   // ragged slice: clamped, zeroed later
  };
  auto add_ptr = [&](int t) {
    const int zz = t / tpi, nn = (t - zz * tpi) * CTN;
    const int cc = nn + 4 * q0;
    return (p.add ? p.add + (long)zz * p.add_bs : p.out + (long)zz * p.out_bs) + (long)o0 * p.HW + (cc < p.HW ? cc : 0);
  };
  if (t_begin < t_end) {
    const float* xg = x_ptr(t_begin);
    const float* ag = add_ptr(t_begin);
#pragma unroll
    for (int oc = 0; oc < COC; ++oc)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = 8 * (o0 + CRS * oc) + e;
        xr[oc][e] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ch < p.Cin) xr[oc][e] = *reinterpret_cast<const f32x4*>(xg + (long)ch * p.HW);   // channels past Cin: no load
      }
#pragma unroll
    for (int i = 0; i < CRPT; ++i) {
      addv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.add) addv[i] = *reinterpret_cast<const f32x4*>(ag + (long)(CRS * i) * p.HW);
    }
  }

  for (int tile = t_begin; tile < t_end; ++tile) {
    // laundered per tile: keeps the unrolled loops' LDS addresses from being hoisted into (spilled) loop invariants
    asm volatile("" : "+v"(l31), "+v"(h), "+v"(q0), "+v"(o0));
    const int tile_it = tile - t_begin;
    auto stamp = [&](int i) {
      if (p.stamps && blockIdx.x == 3 && tid == 0 && tile_it >= 2 && tile_it < 6)
        p.stamps[(tile_it - 2) * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const int z = tile / tpi;
    const int n0 = (tile - z * tpi) * CTN;
    const bool full = n0 + CTN <= p.HW;

    // ---- phase 0: x tile (already in registers) -> LDS (fp16 hi / lo, [px][k]); thread = (pixel quad q0, octets o0 + CRS oc)
    {
      float amax = 0.0f;   // range guard of the fp16 split, flagged per tile (no register lives across the MFMA loop)
      const bool ok = full || (n0 + 4 * q0 < p.HW);
#pragma unroll
      for (int oc = 0; oc < COC; ++oc) {
        const int c0 = 8 * (o0 + CRS * oc);
        float av[8], dv[8];
        if (p.pa) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(p.pa + (long)z * CE + c0);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(p.pa + (long)z * CE + c0 + 4);
          const f32x4 d0 = *reinterpret_cast<const f32x4*>(p.pd + (long)z * CE + c0);
          const f32x4 d1 = *reinterpret_cast<const f32x4*>(p.pd + (long)z * CE + c0 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            av[e] = a0[e] * CSX; av[e + 4] = a1[e] * CSX;
            dv[e] = d0[e] * CSX; dv[e + 4] = d1[e] * CSX;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) { av[e] = CSX; dv[e] = 0.f; }
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          f16x8 vh, vl;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (ok && c0 + e < p.Cin) ? fmaf(xr[oc][e][pp], av[e], dv[e]) : 0.0f;
          sdy_split8(v, vh, vl, amax);
          const int off = cv_off(4 * q0 + pp, o0 + CRS * oc);
          *reinterpret_cast<f16x8*>(Xs_hi + off) = vh;
          *reinterpret_cast<f16x8*>(Xs_lo + off) = vl;
        }
      }
      sdy_flag_range(p.flags, amax);
    }
    __syncthreads();
    stamp(1);
    const int col = n0 + 4 * q0;
    const bool c_ok = full || col < p.HW;
    const long roff = (long)o0 * p.HW + (c_ok ? col : 0);
    const int tnext = (tile + 1 < t_end) ? tile + 1 : tile;   // past the end: a harmless re-read
    const float* xnext = x_ptr(tnext);
    const float* anext = add_ptr(tnext);
    stamp(2);
    // ---- MFMA phase: rows 32 CMT wave .. + 32 CMT, all 64 px, K = 64 KBLK
    f32x16 acc[CMT][2];
#pragma unroll
    for (int mi = 0; mi < CMT; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < KBLK; ++kb) {
      if (kb == KBLK - 1) {   // the refills of the last block fetch block 0 again: ring ready for the next tile
        wp = wbase;
        asm volatile("" : "+v"(wp));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ks = 4 * kb + i;
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int off = cv_off(32 * j + l31, 2 * ks + h);
          bh[j] = *reinterpret_cast<const f16x8*>(Xs_hi + off);
          bl[j] = *reinterpret_cast<const f16x8*>(Xs_lo + off);
        }
#pragma unroll
        for (int mi = 0; mi < CMT; ++mi) {
          const int s = CMT * i + mi;
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_lo[s], bh[j], acc[mi][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[s], bl[j], acc[mi][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[s], bh[j], acc[mi][j], 0, 0, 0);
          r_hi[s] = wp[s * CGROUP];
          r_lo[s] = wp[s * CGROUP + 64];
          __builtin_amdgcn_sched_barrier(0);   // keep the loads here (the scheduler otherwise sinks them next to their use)
        }
      }
      wp += CRING * CGROUP;
    }

    stamp(3);
    // ---- epilogue: accumulators -> LDS [row][px]; bias / addend / GELU / statistics on 16-byte row segments
    __syncthreads();
    stamp(4);   // every wave is done reading the x tile
#pragma unroll
    for (int mi = 0; mi < CMT; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * (CMT * wave + mi) + (r & 3) + 8 * (r >> 2) + 4 * h;
          Os[row * CTN + 32 * j + l31] = acc[mi][j][r] * p.out_scale;
        }
    __syncthreads();
    stamp(5);
    {
      float* og = p.out + (long)z * p.out_bs + roff;
#pragma unroll
      for (int i = 0; i < CRPT; ++i) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Os + (o0 + CRS * i) * CTN + 4 * q0);
        v += brow[i];
        if (p.add_mode == 1) v += addv[i];
        if (p.act == 1) {
          const sdy_gf2 g0 = gelu_erf2(sdy_gf2{v.x, v.y}), g1 = gelu_erf2(sdy_gf2{v.z, v.w});
          v = f32x4{g0.x, g0.y, g1.x, g1.y};
        }
        if (p.add_mode == 2) v += addv[i];
        if (c_ok) SDY_STREAM_STORE(og + (long)(CRS * i) * p.HW, v);
        // next tile's addend row, one per step (every lane: a lane beyond a ragged tile's edge still owns pixels of the next)
        if (p.add) addv[i] = *reinterpret_cast<const f32x4*>(anext + (long)(CRS * i) * p.HW);   // (workgroup-uniform)
        // next tile's pixels, one piece per step as well: with them out of the MFMA phase the weight ring no longer queues
        // behind HBM loads there (vmcnt retires in order): MFMA phase 12.0k -> 7.6k cycles, kernel -3 %
        {
          const int ch = 8 * (o0 + CRS * (i >> 3)) + (i & 7);
          if (ch < p.Cin) xr[i >> 3][i & 7] = *reinterpret_cast<const f32x4*>(xnext + (long)ch * p.HW);
        }
        if (c_ok) {
          psum[i] += (double)((v.x + v.y) + (v.z + v.w));
          psq[i] += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
        }
      }
    }
    stamp(6);
    // statistics flush at this workgroup's last tile of the image (see mlp_h3.hip)
    if (p.stats) {
      const int nt = tile + 1;
      if (nt >= t_end || nt / tpi != z) {   // workgroup-uniform
#pragma unroll
        for (int i = 0; i < CRPT; ++i) {
          double s1 = psum[i], s2 = psq[i];
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) {
            s1 += __shfl_xor(s1, m, 64);
            s2 += __shfl_xor(s2, m, 64);
          }
          if (q0 == 0) {
            double* st = p.stats + ((long)z * CE + o0 + CRS * i) * 2;
            __hip_atomic_fetch_add(st, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(st + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          psum[i] = 0.0; psq[i] = 0.0;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < CRPT; ++i) { psum[i] = 0.0; psq[i] = 0.0; }
    }
    stamp(7);
    __syncthreads();   // the store phase is done with the LDS tile
  }
}

}  // namespace

static unsigned long long* g_cstamps = nullptr;
extern "C" int sdy_conv256_h3_debug_stamps(unsigned long long* host64) {
  if (!g_cstamps || !host64) return SDY_ERR_STATE;
  SDY_HIP_TRY(hipDeviceSynchronize());
  SDY_HIP_TRY(hipMemcpy(host64, g_cstamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SDY_OK;
}

extern "C" int sdy_conv256_h3_supported(int Cin, int Cout) { return (Cin >= 1 && Cin <= CE && Cout == CE) ? 1 : 0; }

extern "C" size_t sdy_conv256_h3_pack_bytes(void) { return (size_t)CNW * CGPW * CGROUP * sizeof(f16x8); }

// w_host: (256, Cin) row-major (Cout, Cin), Cin <= 256 (zero-padded to 256 in the stream)
extern "C" int sdy_conv256_h3_pack_cin(const float* w_host, int Cin, void* dev, float* scale) {
  if (!w_host || !dev || !scale || Cin < 1 || Cin > CE) return SDY_ERR_ARG;
  float mx = 0.f;
  for (int i = 0; i < CE * Cin; ++i) mx = std::fmax(mx, std::fabs(w_host[i]));
  float s = 1.0f;
  if (mx > 0.f && std::isfinite(mx)) {
    int e;
    std::frexp(mx, &e);
    s = std::ldexp(1.0f, 13 - e);
  }
  const size_t gh = (size_t)CGROUP * 8;
  std::vector<_Float16> buf((size_t)CNW * CGPW * gh);
  _Float16* d = buf.data();
  for (int w = 0; w < CNW; ++w)
    for (int ks = 0; ks < CKS; ++ks)
      for (int mi = 0; mi < CMT; ++mi, d += gh)
        for (int ln = 0; ln < 64; ++ln)
          for (int e = 0; e < 8; ++e) {
            const int kk = 16 * ks + 8 * (ln >> 5) + e;
            const float v = kk < Cin ? w_host[(size_t)(32 * (CMT * w + mi) + (ln & 31)) * Cin + kk] * s : 0.0f;
            const _Float16 hv = (_Float16)v;
            d[ln * 8 + e] = hv;
            d[64 * 8 + ln * 8 + e] = (_Float16)(v - (float)hv);
          }
  SDY_HIP_TRY(hipMemcpy(dev, buf.data(), buf.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  *scale = s;
  return SDY_OK;
}
extern "C" int sdy_conv256_h3_pack(const float* w_host, void* dev, float* scale) {
  return sdy_conv256_h3_pack_cin(w_host, CE, dev, scale);
}

int sdy_conv256_h3_launch(const sdy_conv_args* a, hipStream_t stream) {
  if (!a || !a->x || !a->w_frag || !a->out || a->B <= 0 || a->HW <= 0) return SDY_ERR_ARG;
  if (!sdy_conv256_h3_supported(a->Cin, a->Cout)) return SDY_ERR_UNSUPPORTED;
  if (a->drop_p > 0.f || a->keep_mask || a->batch_scale) return SDY_ERR_UNSUPPORTED;
  if (a->pa && a->Cin != CE) return SDY_ERR_UNSUPPORTED;   // the affine prologue reads 8-channel groups
  if ((a->HW & 3) || (a->x_bstride & 3) || (a->out_bstride & 3) || (a->add_mode && (a->add_bstride & 3))) return SDY_ERR_ALIGN;
  ConvParams p;
  p.x = a->x; p.x_bs = a->x_bstride; p.pa = a->pa; p.pd = a->pd;
  p.w = reinterpret_cast<const f16x8*>(a->w_frag);
  p.bias = a->bias;
  p.add = a->add_mode ? a->add : nullptr; p.add_bs = a->add_bstride; p.add_mode = a->add_mode;
  p.act = a->act;
  p.out = a->out; p.out_bs = a->out_bstride;
  p.stats = a->stats;
  SDY_TRY(sdy_flags_ptr(&p.flags));
  p.HW = a->HW; p.B = a->B; p.Cin = a->Cin;
  p.out_scale = 1.0f / (a->w_frag_scale * CSX);
  p.stamps = nullptr;
  if (std::getenv("SDY_CONV_STAMPS")) {
    if (!g_cstamps) SDY_HIP_TRY(hipMalloc(&g_cstamps, 64 * sizeof(unsigned long long)));
    p.stamps = g_cstamps;
  }
  int n_cu = 0;
  SDY_TRY(sdy_cu_count(&n_cu));
  const long ntiles = (long)((a->HW + CTN - 1) / CTN) * a->B;
  const long want = n_cu;   // persistent: one workgroup per CU
  dim3 grid((unsigned)(ntiles < want ? ntiles : want));
  switch ((a->Cin + 63) / 64) {
    case 1: hipLaunchKernelGGL(conv_h3_kernel<1>, grid, dim3(CNT), 0, stream, p); break;
    case 2: hipLaunchKernelGGL(conv_h3_kernel<2>, grid, dim3(CNT), 0, stream, p); break;
    case 3: hipLaunchKernelGGL(conv_h3_kernel<3>, grid, dim3(CNT), 0, stream, p); break;
    default: hipLaunchKernelGGL(conv_h3_kernel<4>, grid, dim3(CNT), 0, stream, p); break;
  }
  return sdy_launch_status();
}
