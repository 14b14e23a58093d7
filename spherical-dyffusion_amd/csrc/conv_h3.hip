// Cin (<= 384) -> 256 channel 1x1 convolution (the block's inner skip, both encoder layers, the decoder's first layer) as a
// persistent split-fp16 kernel in the style of mlp_h3.hip:
//   out[b] = act( W . (pa[b]*x[b] + pd[b]) + bias + add_pre[b] ) + add_post[b]        (+ InstanceNorm statistics of out)
//   * one workgroup owns 64 pixels and all 256 output rows; the activation tile is fetched once, split hi/lo and
//     parked in LDS ([px][k], XOR-swizzled); wave w computes 256 / NW rows (1 or 2 x 2 MFMA tiles);
//   * the weight never touches LDS: packed per wave as a linear stream of MFMA A-fragment pairs (hi, lo) in consumption
//     order, L2 -> registers through a 4-group ring;
//   * the accumulators start from the bias; the epilogue goes through LDS so that the addend loads and the stores are
//     16-byte row segments, applies addend / GELU there, and sums what it stores per row: DPP over the 16 lanes that share
//     a row, then one fp64 pair per row in LDS, flushed with one atomic pair per row and IMAGE (the workgroup is persistent
//     over a contiguous range of tiles) -- the next InstanceNorm (norm1) needs no pass of its own over the tensor.
// Cin <= 256: 4 waves, <= 256 registers, 69 KB of LDS -> TWO workgroups per CU.  They are independent (no barrier couples
// them), drift apart, and one's MFMA phase / barriers / LDS passes run under the other's HBM traffic (+2..7 % over one
// 8-wave workgroup per CU).  256 < Cin <= 384 (decoder: [block output | inputs]): 8 waves, one workgroup per CU, 96 KB tile.
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int CE = 256;          // output channels
constexpr int CTN = 64;          // pixels per tile
constexpr int CKMAX = 384;       // input channels supported
constexpr int CRING = 4;         // weight groups in flight: one "window" of the stream
constexpr int CGROUP = 2 * 64;   // f16x8 elements per group (hi | lo)
constexpr float CSX = SDY_ACT_SX;
constexpr int CSTAT_BYTES = CE * 2 * 8;
#ifndef SDY_CONV_STAMP_T0
#define SDY_CONV_STAMP_T0 2      // first of the four tiles of workgroup 3 that SDY_CONV_STAMPS samples
#endif

// Tiling for NW waves per workgroup and KBLK k-blocks of 64 input channels.
template <int KBLK, int NW>
struct ConvCfg {
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static constexpr int NT = 64 * NW;                    // threads
  static constexpr int MT = 8 / NW;                     // 32-row output tiles per wave
  static constexpr int RS = 4 * NW;                     // range of o0 = tid / 16: staging octets o0 + RS oc, rows o0 + RS i
  static constexpr int OC = (8 * KBLK + RS - 1) / RS;   // channel octets per thread in the staging role
  static constexpr int RPT = 64 / NW;                   // output rows per thread in the store loop
  static constexpr int PPI = (8 * OC + RPT - 1) / RPT;  // x pieces prefetched per store-loop step
  static constexpr int KROW = KBLK <= 4 ? 256 : 384;   // LDS row length in halfs (the swizzle permutes whole groups of 16 chunks)
  static constexpr int KSW = KROW / 16;                 // k-steps per wave in the packed stream (zero-padded)
  static constexpr int GPK = 4 * MT;                    // groups per k-block of 4 k-steps
  static constexpr int GPW = MT * KSW;                  // groups per wave
  static constexpr int WGS = NW == 4 ? 2 : 1;           // workgroups per CU
  static constexpr int TILE_BYTES = 2 * CTN * KROW * 2; // hi + lo
  static constexpr int LDS_BYTES = TILE_BYTES + CSTAT_BYTES + CE * 4 + SDY_GELU_TAB_BYTES;   // (4 waves: 2 x 76.8 KB per CU)
};
#ifndef SDY_CONV_GELU_TAB
#define SDY_CONV_GELU_TAB 1      // GELU of the store loop from the LDS table (common.h, gelu_tab16); 0: gelu_erf2 (A&S 7.1.26)
#endif

struct ConvParams {
  const float* x; long x_bs;
  const float* pa; const float* pd;
  const f16x8* w;                  // [NW waves][GPW groups][hi | lo][64 lanes]
  const float* bias;
  const float* add; long add_bs; int add_mode;   // 1: before the activation, 2: after it
  int act;
  float* out; long out_bs;
  int out_tiled;                   // output TILE-MAJOR: [b][64-pixel tile][256 rows][64] (sdy_conv_args.out_tiled), out_bs per image
  double* stats;
  int HW, B;
  int Cin;                         // input channels (the weight stream is zero-padded to KBLK * 64)
  float out_scale;
  unsigned* flags;                 // sticky status word (sdy_status_flags)
  unsigned* head;                  // range headroom word of this consumer class (sdy_range_headroom; null unless enabled)
  const float* gelu_tab;           // GELU table in global memory (sdy_gelu_table_ptr)
  SdyImgMap xmap;                  // drop-path skip: image z of this launch reads x / pa / pd of batch row xmap.idx[z] (common.h)
  unsigned long long* stamps;    // timing experiments only (SDY_CONV_STAMPS)
};

__device__ __forceinline__ int cv_swz(int px) { return (px & 15) ^ (((px >> 4) & 1) * 3); }
// 16-byte chunk c of pixel px in a row of KROW halfs
template <int KROW>
__device__ __forceinline__ int cv_off(int px, int c) { return px * KROW + (((c & ~15) | ((c ^ cv_swz(px)) & 15)) << 3); }

// KBLK = ceil(Cin / 64): k-blocks of 4 k-steps actually streamed (4 for the block's 256 -> 256 convs, 2 / 3 for the
// encoders, 6 for the decoder)
template <int KBLK, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_h3_kernel(const ConvParams p) {
  using G = ConvCfg<KBLK, NW>;
  constexpr int CMT = G::MT, CRS = G::RS, COC = G::OC, CRPT = G::RPT, KROW = G::KROW, CGPK = G::GPK, CGPW = G::GPW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];   // x tile (64 / 96 KB) + 4 KB + 1 KB
  _Float16* Xs_hi = reinterpret_cast<_Float16*>(smem);
  _Float16* Xs_lo = Xs_hi + CTN * KROW;
  float* Os = reinterpret_cast<float*>(smem);   // epilogue: [256 rows][64 px] fp32 (aliases the x tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int h = lane >> 5, l31 = lane & 31;
  int q0 = tid & 15, o0 = tid >> 4;
  const int tpi = (p.HW + CTN - 1) / CTN;
  const int ntiles = tpi * p.B;
  // contiguous tile range per workgroup: a workgroup then crosses an image boundary at most ~once, so the per-image
  // flush of the statistics is rare instead of every few tiles
  const int t_per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * t_per;
  const int t_end = (t_begin + t_per < ntiles) ? t_begin + t_per : ntiles;

  f16x8 r_hi[CRING], r_lo[CRING];
  // ring loads: (wave-uniform stream base in SGPRs) + (the lane's running offset), sdy_ring_ld in common.h; the stream of a
  // tile (NWIN windows of CRING groups) is fetched front to back, one window ahead of its use, and wraps
  constexpr int CGROUP_BYTES = CGROUP * (int)sizeof(f16x8);
  const char* const wbase = reinterpret_cast<const char*>(p.w) + (size_t)__builtin_amdgcn_readfirstlane(wave) * CGPW * CGROUP_BYTES;
  unsigned woff = (unsigned)lane * 16u;
#pragma unroll
  for (int s = 0; s < CRING; ++s) {
    r_hi[s] = sdy_ring_ld(wbase, woff, 0);
    r_lo[s] = sdy_ring_ld(wbase, woff, CGROUP_BYTES / 2);
    woff += CGROUP_BYTES;
  }
  constexpr int NWIN = KBLK * CMT;                     // windows of CRING groups per tile
  if (NWIN == 1) woff -= CRING * CGROUP_BYTES;         // (the ring holds the whole stream: its refills fetch it again)

  // bias: pre-divided by out_scale and parked in LDS once; the accumulators START from it (no registers across tiles, no
  // add in the store loop).  Statistics: [256 rows][sum, sum of squares] fp64.
  double* Ss = reinterpret_cast<double*>(smem + G::TILE_BYTES);
  float* Bs = reinterpret_cast<float*>(smem + G::TILE_BYTES + CSTAT_BYTES);
  if (tid < CE) {
    Bs[tid] = p.bias ? p.bias[tid] / p.out_scale : 0.0f;
    Ss[2 * tid] = 0.0; Ss[2 * tid + 1] = 0.0;
  }
  // GELU table (plain gelu: coefficients / 16); the store loop then works on w = 8 v, the accumulator pass scales by 8
  float* Gt = reinterpret_cast<float*>(smem + G::TILE_BYTES + CSTAT_BYTES + CE * 4);
  const unsigned gt_base = (unsigned)(uintptr_t)Gt;
  const bool tab_gelu = SDY_CONV_GELU_TAB && p.act == 1;
  if (tab_gelu) gelu_tab_to_lds(Gt, p.gelu_tab, tid, G::NT, 1.0f / SDY_GELU_SX);
  const float os_scale = tab_gelu ? p.out_scale * SDY_GELU_WS : p.out_scale;
  // (ordered before their first use by the barrier that ends the first tile's phase 0)

  // Software pipeline over tiles.  Loads are issued ONE per step where they ride on other work, never as a burst inside
  // the MFMA phase: tile t+1's pixels (xr) ride on tile t's store loop; tile t's addend rows are requested when its MFMA
  // loop has ended (their registers are the accumulators' until then) and travel during the accumulator pass through LDS;
  // the MFMA phase only streams the L2-resident weight ring (vmcnt retires in order: an HBM load issued there makes the
  // ring wait, measured 7.6k -> 12k cycles).
  f32x4 xr[COC][8], addv[CRPT];
  // Addressing: every global access is (wave-uniform base of the image and row: SGPR arithmetic) + (ONE 32-bit byte offset per
  // lane and tile), the SADDR form of global_load / global_store -- per-row 64-bit lane addresses (16 rows x 3 streams) would
  // be hoisted out of the tile loop and spilled.  Lane offsets: x rows 8 o0 + e, addend / output rows o0 + CRS i.
  auto img = [&](const float* base, long bs, int t) { return base + (long)(t / tpi) * bs; };   // uniform
  auto ximg = [&](int t) { return p.x + (long)sdy_img(p.xmap, t / tpi) * p.x_bs; };             // uniform
  auto lane_col = [&](int t) {   // first pixel of this lane's quad in tile t; a lane beyond the ragged edge re-reads column 0
    const int nn = (t % tpi) * CTN + 4 * q0;
    return nn < p.HW ? nn : 0;
  };
  const float* addsrc = p.add ? p.add : p.out;          // (p.out: never dereferenced without an addend, keeps it branch-free)
  const long addsrc_bs = p.add ? p.add_bs : p.out_bs;
  if (t_begin < t_end) {
    const float* xz = ximg(t_begin);
    const unsigned xo = (unsigned)(8 * o0 * p.HW + lane_col(t_begin)) * 4u;
#pragma unroll
    for (int oc = 0; oc < COC; ++oc)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = 8 * (o0 + CRS * oc) + e;
        xr[oc][e] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ch < p.Cin) xr[oc][e] = sdy_ld16s(xz + (long)(8 * CRS * oc + e) * p.HW, xo);   // channels past Cin: no load
      }
  }

  for (int tile = t_begin; tile < t_end; ++tile) {
    // laundered per tile: keeps the unrolled loops' LDS addresses from being hoisted into (spilled) loop invariants
    asm volatile("" : "+v"(l31), "+v"(h), "+v"(q0), "+v"(o0));
    const int tile_it = tile - t_begin;
    auto stamp = [&](int i) {
      if (SDY_STAMPS_ON && p.stamps && blockIdx.x == 3 && tid == 0 && tile_it >= SDY_CONV_STAMP_T0 && tile_it < SDY_CONV_STAMP_T0 + 4)
        p.stamps[(tile_it - SDY_CONV_STAMP_T0) * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const int z = tile / tpi;
    const int zx = sdy_img(p.xmap, z);
    const int n0 = (tile - z * tpi) * CTN;
    const bool full = n0 + CTN <= p.HW;

    // ---- phase 0: x tile (already in registers) -> LDS (fp16 hi / lo, [px][k]); thread = (pixel quad q0, octets o0 + CRS oc)
    {
      float amax = 0.0f;   // range guard of the fp16 split, flagged per tile (no register lives across the MFMA loop)
      const bool ok = full || (n0 + 4 * q0 < p.HW);
#pragma unroll
      for (int oc = 0; oc < COC; ++oc) {
        const int c0 = 8 * (o0 + CRS * oc);
        if (8 * CRS * oc + 8 * CRS > KROW && c0 >= KROW) continue;   // octets beyond the LDS row
        float av[8], dv[8];
        if (p.pa) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(p.pa + (long)zx * CE + c0);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(p.pa + (long)zx * CE + c0 + 4);
          const f32x4 d0 = *reinterpret_cast<const f32x4*>(p.pd + (long)zx * CE + c0);
          const f32x4 d1 = *reinterpret_cast<const f32x4*>(p.pd + (long)zx * CE + c0 + 4);
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
          const int off = cv_off<KROW>(4 * q0 + pp, o0 + CRS * oc);
          *reinterpret_cast<f16x8*>(Xs_hi + off) = vh;
          *reinterpret_cast<f16x8*>(Xs_lo + off) = vl;
        }
      }
      sdy_flag_range(p.flags, amax, p.head);
    }
    __syncthreads();
    stamp(1);
    const int col = n0 + 4 * q0;
    const bool c_ok = full || col < p.HW;
    const unsigned ro = (unsigned)(o0 * p.HW + (c_ok ? col : 0)) * 4u;   // this tile: addend / output rows o0 + CRS i
    // tile-major output: the tile is one contiguous [256 rows][64 px] block, rows 64 floats apart
    const unsigned ro_out = p.out_tiled ? (unsigned)(o0 * CTN + 4 * q0) * 4u : ro;
    const int out_rs = p.out_tiled ? CTN : p.HW;
    const int tnext = (tile + 1 < t_end) ? tile + 1 : tile;   // past the end: a harmless re-read
    const float* xz_next = ximg(tnext);
    const unsigned xo_next = (unsigned)(8 * o0 * p.HW + lane_col(tnext)) * 4u;
    stamp(2);
    // ---- MFMA phase: rows 32 CMT wave .. + 32 CMT, all 64 px, K = 64 KBLK
    f32x16 acc[CMT][2];
#pragma unroll
    for (int mi = 0; mi < CMT; ++mi)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {   // rows 32 (CMT wave + mi) + 8 r4 + 4 h .. + 3 of both pixel tiles
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(Bs + 32 * (CMT * wave + mi) + 8 * r4 + 4 * h);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mi][j][4 * r4 + r] = b4[r];
      }
    // The stream is consumed in windows of CRING groups; while window c is consumed its slots are refilled with window
    // c + 1, and the last window's refills fetch window 0 again, so the ring is ready for the next tile.
#pragma unroll
    for (int kb = 0; kb < KBLK; ++kb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ks = 4 * kb + i;
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int off = cv_off<KROW>(32 * j + l31, 2 * ks + h);
          bh[j] = *reinterpret_cast<const f16x8*>(Xs_hi + off);
          bl[j] = *reinterpret_cast<const f16x8*>(Xs_lo + off);
        }
#pragma unroll
        for (int mi = 0; mi < CMT; ++mi) {
          const int g = kb * CGPK + CMT * i + mi;   // group of the tile's stream
          const int s = g % CRING;
#pragma unroll
          for (int j = 0; j < 2; ++j) SDY_CROSS_TERM(acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_lo[s], bh[j], acc[mi][j], 0, 0, 0));
#pragma unroll
          for (int j = 0; j < 2; ++j) SDY_CROSS_TERM(acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[s], bl[j], acc[mi][j], 0, 0, 0));
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[s], bh[j], acc[mi][j], 0, 0, 0);
          r_hi[s] = sdy_ring_ld(wbase, woff, 0);     // group (g + CRING) mod (NWIN CRING) of the stream
          r_lo[s] = sdy_ring_ld(wbase, woff, CGROUP_BYTES / 2);
          woff += CGROUP_BYTES;
          if ((g + CRING + 1) % (NWIN * CRING) == 0) woff -= NWIN * CRING * CGROUP_BYTES;   // wrapped: next fetch is group 0
          __builtin_amdgcn_sched_barrier(0);   // keep the loads here (the scheduler otherwise sinks them next to their use)
        }
      }
    }

    stamp(3);
    {   // this tile's addend rows (defined on both paths: not live across the MFMA loop)
      const float* az = img(addsrc, addsrc_bs, tile);
#pragma unroll
      for (int i = 0; i < CRPT; ++i) addv[i] = p.add ? sdy_ld16s(az + (long)(CRS * i) * p.HW, ro) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- epilogue: accumulators -> LDS [row][px]; addend / GELU / statistics on 16-byte row segments
    __syncthreads();
    stamp(4);   // every wave is done reading the x tile
#pragma unroll
    for (int mi = 0; mi < CMT; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * (CMT * wave + mi) + (r & 3) + 8 * (r >> 2) + 4 * h;
          Os[row * CTN + 32 * j + l31] = acc[mi][j][r] * os_scale;
        }
    __syncthreads();
    stamp(5);
    {
      float* oz = p.out + (long)z * p.out_bs + (p.out_tiled ? (long)(tile - z * tpi) * (CE * CTN) : 0L);   // uniform
#pragma unroll
      for (int i = 0; i < CRPT; ++i) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Os + (o0 + CRS * i) * CTN + 4 * q0);
        if (tab_gelu) {   // (workgroup-uniform) v holds 8 x the pre-activation: 10 plain VALU instructions per value, no v_rcp / v_exp
          if (p.add_mode == 1) v = addv[i] * SDY_GELU_WS + v;
          v = f32x4{gelu_tab16(v.x, gt_base), gelu_tab16(v.y, gt_base), gelu_tab16(v.z, gt_base), gelu_tab16(v.w, gt_base)};
        } else {
        if (p.add_mode == 1) v += addv[i];
        if (p.act == 1) {
          const sdy_gf2 g0 = gelu_erf2(sdy_gf2{v.x, v.y}), g1 = gelu_erf2(sdy_gf2{v.z, v.w});
          v = f32x4{g0.x, g0.y, g1.x, g1.y};
        }
        }
        if (p.add_mode == 2) v += addv[i];
        if (c_ok) sdy_st16s(oz + (long)(CRS * i) * out_rs, ro_out, v);
        // next tile's pixels, a piece (or two) per step (every lane: a lane beyond a ragged tile's edge still owns pixels of
        // the next tile)
#pragma unroll
        for (int k = 0; k < G::PPI; ++k) {
          const int piece = G::PPI * i + k, oc = piece >> 3, e = piece & 7;
          if (oc < COC) {
            const int ch = 8 * (o0 + CRS * oc) + e;
            if (ch < p.Cin) xr[oc][e] = sdy_ld16s(xz_next + (long)(8 * CRS * oc + e) * p.HW, xo_next);
          }
        }
        if (p.stats) {   // (workgroup-uniform)
          const float s1 = row16_sum(c_ok ? (v.x + v.y) + (v.z + v.w) : 0.0f);
          const float s2 = row16_sum(c_ok ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.0f);
          if (q0 == 0) {   // the one owner of this row in the workgroup
            double* st = Ss + 2 * (o0 + CRS * i);
            __hip_atomic_fetch_add(st, (double)s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(st + 1, (double)s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    stamp(6);
    // statistics flush at this workgroup's last tile of the image
    {
      const int nt = tile + 1;
      if (p.stats && (nt >= t_end || nt / tpi != z)) {   // workgroup-uniform
        __syncthreads();
        if (tid < CE) {
          double* st = p.stats + ((long)z * CE + tid) * 2;
          __hip_atomic_fetch_add(st, Ss[2 * tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(st + 1, Ss[2 * tid + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          Ss[2 * tid] = 0.0; Ss[2 * tid + 1] = 0.0;
        }
      }
    }
    stamp(7);
    __syncthreads();   // the store phase is done with the LDS tile
  }
}

}  // namespace

#if SDY_STAMPS_ON
static unsigned long long* g_cstamps = nullptr;
SDY_DEBUG_EXPORT int sdy_conv256_h3_debug_stamps(unsigned long long* host64) {
  if (!g_cstamps || !host64) return SDY_ERR_STATE;
  SDY_HIP_TRY(hipDeviceSynchronize());
  SDY_HIP_TRY(hipMemcpy(host64, g_cstamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SDY_OK;
}
#endif

extern "C" int sdy_conv256_h3_supported(int Cin, int Cout) { return (Cin >= 1 && Cin <= CKMAX && Cout == CE) ? 1 : 0; }

// layout of the packed stream for Cin input channels: waves, k-steps per wave (zero-padded), 32-row tiles per wave
static void conv_layout(int Cin, int* nw, int* ksw, int* mt) {
  const int kblk = (Cin + 63) / 64;
  *nw = kblk <= 4 ? 4 : 8;
  *ksw = kblk <= 4 ? 16 : 24;
  *mt = 8 / *nw;
}

extern "C" size_t sdy_conv256_h3_pack_bytes_cin(int Cin) {
  int nw, ksw, mt;
  conv_layout(Cin < 1 ? 1 : Cin, &nw, &ksw, &mt);
  return (size_t)nw * mt * ksw * CGROUP * sizeof(f16x8);
}
extern "C" size_t sdy_conv256_h3_pack_bytes(void) { return sdy_conv256_h3_pack_bytes_cin(CE); }

// w_host: (256, Cin) row-major (Cout, Cin), Cin <= 384 (zero-padded in the stream)
extern "C" int sdy_conv256_h3_pack_cin(const float* w_host, int Cin, void* dev, float* scale) {
  if (!w_host || !dev || !scale || Cin < 1 || Cin > CKMAX) return SDY_ERR_ARG;
  float mx = 0.f;
  for (int i = 0; i < CE * Cin; ++i) mx = std::fmax(mx, std::fabs(w_host[i]));
  float s = 1.0f;
  if (mx > 0.f && std::isfinite(mx)) {
    int e;
    std::frexp(mx, &e);
    s = std::ldexp(1.0f, 13 - e);
  }
  int nw, ksw, mt;
  conv_layout(Cin, &nw, &ksw, &mt);
  const size_t gh = (size_t)CGROUP * 8;
  std::vector<_Float16> buf((size_t)nw * mt * ksw * gh);
  _Float16* d = buf.data();
  for (int w = 0; w < nw; ++w)
    for (int ks = 0; ks < ksw; ++ks)
      for (int mi = 0; mi < mt; ++mi, d += gh)
        for (int ln = 0; ln < 64; ++ln)
          for (int e = 0; e < 8; ++e) {
            const int kk = 16 * ks + 8 * (ln >> 5) + e;
            const float v = kk < Cin ? w_host[(size_t)(32 * (mt * w + mi) + (ln & 31)) * Cin + kk] * s : 0.0f;
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

template <int KBLK, int NW>
static void conv_launch(const ConvParams& p, long ntiles, int n_cu, hipStream_t stream) {
  using G = ConvCfg<KBLK, NW>;
  const long want = (long)n_cu * G::WGS;   // persistent: WGS workgroups per CU
  dim3 grid((unsigned)(ntiles < want ? ntiles : want));
  hipLaunchKernelGGL((conv_h3_kernel<KBLK, NW>), grid, dim3(G::NT), 0, stream, p);
}

int sdy_conv256_h3_launch(const sdy_conv_args* a, hipStream_t stream) {
  if (!a || !a->x || !a->w_frag || !a->out || a->B <= 0 || a->HW <= 0) return SDY_ERR_ARG;
  if (!sdy_conv256_h3_supported(a->Cin, a->Cout)) return SDY_ERR_UNSUPPORTED;
  if (a->drop_p > 0.f || a->keep_mask || a->batch_scale) return SDY_ERR_UNSUPPORTED;
  if (a->pa && a->Cin != CE) return SDY_ERR_UNSUPPORTED;   // the affine prologue reads 8-channel groups of 256 channels
  if ((a->HW & 3) || (a->x_bstride & 3) || (a->out_bstride & 3) || (a->add_mode && (a->add_bstride & 3))) return SDY_ERR_ALIGN;
  if ((long)a->HW * CKMAX * 4 >= (1L << 32)) return SDY_ERR_UNSUPPORTED;   // 32-bit lane offsets inside an image
  ConvParams p;
  p.x = a->x; p.x_bs = a->x_bstride; p.pa = a->pa; p.pd = a->pd;
  p.w = reinterpret_cast<const f16x8*>(a->w_frag);
  p.bias = a->bias;
  p.add = a->add_mode ? a->add : nullptr; p.add_bs = a->add_bstride; p.add_mode = a->add_mode;
  p.act = a->act;
  p.out = a->out; p.out_bs = a->out_bstride; p.out_tiled = a->out_tiled;
  if (a->out_tiled && a->add_mode && a->add == a->out) return SDY_ERR_ARG;   // a tile-major output cannot alias the NCHW addend
  p.stats = a->stats;
  SDY_TRY(sdy_flags_ptr(&p.flags));
  SDY_TRY(sdy_headroom_ptr(SDY_RANGE_CONV, &p.head));
  SDY_TRY(sdy_gelu_table_ptr(&p.gelu_tab));
  p.HW = a->HW; p.B = a->B; p.Cin = a->Cin;
  SDY_TRY(sdy_img_map_fill(p.xmap, a->x_rows, a->B));
  p.out_scale = 1.0f / (a->w_frag_scale * CSX);
  p.stamps = nullptr;
#if SDY_STAMPS_ON
  if (std::getenv("SDY_CONV_STAMPS")) {
    if (!g_cstamps) SDY_HIP_TRY(hipMalloc(&g_cstamps, 64 * sizeof(unsigned long long)));
    p.stamps = g_cstamps;
  }
#endif
  int n_cu = 0;
  SDY_TRY(sdy_cu_count(&n_cu));
  const long ntiles = (long)((a->HW + CTN - 1) / CTN) * a->B;
  switch ((a->Cin + 63) / 64) {
    case 1: conv_launch<1, 4>(p, ntiles, n_cu, stream); break;
    case 2: conv_launch<2, 4>(p, ntiles, n_cu, stream); break;
    case 3: conv_launch<3, 4>(p, ntiles, n_cu, stream); break;
    case 4: conv_launch<4, 4>(p, ntiles, n_cu, stream); break;
    case 5: conv_launch<5, 8>(p, ntiles, n_cu, stream); break;
    default: conv_launch<6, 8>(p, ntiles, n_cu, stream); break;
  }
  return sdy_launch_status();
}
