// Two 1x1 convolutions with a GELU between them in ONE launch for gfx950: the encoder and the decoder of the SFNO,
//     out = W2 . GELU(W1 . x + b1) + add
// (src/models/sfno/sfnonet.py:609-618 + the position embedding of :824 -- Cin -> 256 -> 256 -- and :734-744 applied to
// [block output | inputs] at :831-837 -- 256 + Cin -> 256 -> Cout).  Run as two convolutions the pair writes its 256-channel
// hidden activation to HBM and reads it back (3.3 of the encoder's 5.7 GB and of the decoder's 6.1 GB at B = 25); here one
// workgroup owns 64 pixels end to end as in mlp_h3.hip and the hidden tile stays in LDS.
//
//   x tile    fp32 -> x16, fp16 hi / lo -> LDS [px][k], rows padded to an ODD number of 16-byte chunks (fragment reads and
//             staging writes are then bank-conflict free for any row length).  More than 13 k-steps of input (the decoder)
//             go through the tile in NPART = 2 parts, fc1 accumulating over both.
//   fc1       wave w: hidden rows 32 w .. +32 of BOTH 128-row chunks, all 64 px (2 x 2 accumulator tiles): a k-step's four
//             LDS fragment reads feed 12 MFMAs.
//   chain     bias + exact-erf GELU, x16, split -> LDS H[chunk][px][128 k] (mlp_h3's layout).  chain(0) stands alone;
//             chain(1) runs beside the MFMAs of fc2(0) when there are enough of them (256 output rows), else alone.
//   fc2       256 output rows: wave w rows 64 w .. +64 (2 x 2 tiles) as in mlp_h3.  <= 64 output rows (decoder): wave w
//             owns ONE tile, rows 32 (w / 2) .. +32 of pixels 32 (w % 2) .. +32.
//   epilogue  through LDS -> 16-byte row stores with the addend (position embedding: batch stride 0) and, for the encoder,
//             the InstanceNorm statistics of what is stored (block 0's norm0), fp64 per row and image as in mlp_h3.
// The weights stream from L2 as MFMA A-fragment pairs in consumption order through a ring of 16 groups; a tile's N groups
// are numbered 0 .. NPAD - 1 (NPAD = N rounded up to 16; the holes are never loaded), so ring slot = position mod 16 holds
// for every tile, and position g's slot is refilled with group g + 16 right behind g's last MFMA.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PH = 256;        // hidden channels
constexpr int PTN = 64;        // pixels per workgroup
constexpr int PHC = 128;       // hidden channels per chunk
constexpr int PKSC = PHC / 16; // k-steps of fc2 per chunk
constexpr int PRING = 16;
constexpr int PGROUP = 2 * 64; // f16x8 elements per group (hi | lo)
constexpr float PSX = SDY_ACT_SX;   // activation pre-scale

template <int KSP, int NPART, int MO>
struct PairCfg {
  static_assert(MO == 8 || MO == 2, "256 output rows (8 row tiles) or <= 64 (2)");
  static constexpr int NF1 = 2 * NPART * KSP;             // fc1 groups per wave and tile
  static constexpr int NF2 = MO == 8 ? 16 : 8;            // fc2 groups per chunk
  static constexpr int N = NF1 + 2 * NF2;
  static constexpr int NPAD = (N + PRING - 1) / PRING * PRING;
  static constexpr int XROW = 16 * KSP + 8;               // halfs per pixel row of the x tile (2 KSP + 1 chunks)
  static constexpr int NI = (2 * KSP + 15) / 16;          // channel octets per thread in the staging role
  static constexpr int KMAX = 16 * KSP * NPART;           // input channels
  static constexpr size_t XS_BYTES = (size_t)2 * PTN * XROW * 2;
  static constexpr size_t HS_BYTES = (size_t)4 * PTN * PHC * 2;
  static constexpr size_t LDS_BYTES = XS_BYTES + HS_BYTES + PH * 4 + 64;   // + 8 wave maxima of the tile's dynamic scale
  static constexpr size_t STREAM_BYTES = (size_t)4 * (NPAD + PRING) * PGROUP * 16;   // weight stream; 64 bytes of tail follow
  static_assert(N >= PRING, "the first ring fill takes 16 real groups");
  static_assert(NPART <= 2, "two slots of part maxima in LDS");
  static_assert(MO == 8 || XS_BYTES >= (size_t)64 * PTN * 4, "the 64-row output tile is staged in the x tile's storage");
};

// chain(0) beside the MFMAs of fc1's last part (chunk-major there) or on its own behind a k-step-major fc1
#ifndef SDY_PAIR_FC1_CHAIN
#define SDY_PAIR_FC1_CHAIN 1
#endif
constexpr bool kChainInFc1 = SDY_PAIR_FC1_CHAIN != 0;
#ifndef SDY_PAIR_EARLY_X
#define SDY_PAIR_EARLY_X 1
#endif

struct PairParams {
  const float* x; long x_bs; int Cin;
  const f16x8* w;                          // [4 waves][NPAD + 16 groups][hi | lo][64 lanes]
  const float* b1;
  float* out; long out_bs; int Cout;
  const float* add; long add_bs;
  int HW, B;
  float s1, s2;                            // accumulator scales: 1 / w1_scale (the x tile's own scale joins per tile) and
                                           // 1 / (w2_scale * PSX)
  double* stats;                           // optional [B][256][2] (MO == 8 only)
  unsigned* flags;
  unsigned long long* stamps;              // timing experiments only (SDY_PAIR_STAMPS)
};

__device__ __forceinline__ int hs_swz(int px) { return (px & 15) ^ (((px >> 4) & 1) * 3); }   // (mlp_h3.hip, px_swz)
__device__ __forceinline__ int hs_off(int px, int c) { return px * PHC + (((c ^ hs_swz(px)) & 15) << 3); }

// (a workgroup-uniform float the compiler would otherwise carry in a VGPR)
__device__ __forceinline__ float sgpr_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// Maximum of a wave's NON-NEGATIVE values by DPP (row shifts, then the two row broadcasts of gfx9): valid in lane 63.
// (__shfl_xor is six dependent ds_bpermute round trips -- with one wave per SIMD every one of them is exposed.)
#define SDY_DPP_MAX(v, ctrl, rmask)                                                                                         \
  v = __builtin_fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, true)))
__device__ __forceinline__ float wave_max_nonneg_l63(float v) {
  SDY_DPP_MAX(v, 0x111, 0xF);   // row_shr:1
  SDY_DPP_MAX(v, 0x112, 0xF);   // row_shr:2
  SDY_DPP_MAX(v, 0x114, 0xF);   // row_shr:4
  SDY_DPP_MAX(v, 0x118, 0xF);   // row_shr:8   -> lane 15 of every row holds its row's maximum
  SDY_DPP_MAX(v, 0x142, 0xA);   // row_bcast:15 into rows 1 and 3
  SDY_DPP_MAX(v, 0x143, 0xC);   // row_bcast:31 into rows 2 and 3 -> lane 63
  return v;
}
#undef SDY_DPP_MAX

struct Piece { float v[4], vr[4], t[4], e[4], q[4]; };   // vr = v * (this tile's hidden scale r <= 1, see the kernel)
constexpr float G_CT = 0.3275911f * 0.70710678118654752440f, G_KAP = -0.5f * 1.44269504088896340736f, G_HS = 0.5f * PSX;
constexpr float G_A5 = 1.061405429f * G_HS, G_A4 = -1.453152027f * G_HS, G_A3 = 1.421413741f * G_HS,
                G_A2 = -0.284496736f * G_HS, G_A1 = 0.254829592f * G_HS;
// PSX * gelu(v) = HS v + |v| (HS - Q),  Q = t poly(t) exp(-v^2 / 2),  t = 1 / (1 + CT |v|)  (A&S 7.1.26: the arithmetic of
// gelu_erf in common.h and of mlp_h3's chain), in 10 slots of <= 6 plain VALU instructions: one slot hides behind one MFMA.
__device__ __forceinline__ void gelu_slot(Piece& s, int st) {
  switch (st) {
    case 1:
#pragma unroll
      for (int r = 0; r < 4; ++r) s.t[r] = fmaf(__builtin_fabsf(s.v[r]), G_CT, 1.0f);
      s.e[0] = s.v[0] * G_KAP; s.e[1] = s.v[1] * G_KAP;
      break;
    case 2:
      s.t[0] = __builtin_amdgcn_rcpf(s.t[0]); s.t[1] = __builtin_amdgcn_rcpf(s.t[1]);
      s.e[2] = s.v[2] * G_KAP; s.e[3] = s.v[3] * G_KAP;
      s.e[0] *= s.v[0]; s.e[1] *= s.v[1];
      break;
    case 3:
      s.t[2] = __builtin_amdgcn_rcpf(s.t[2]); s.t[3] = __builtin_amdgcn_rcpf(s.t[3]);
      s.e[2] *= s.v[2]; s.e[3] *= s.v[3];
      s.q[0] = fmaf(s.t[0], G_A5, G_A4); s.q[1] = fmaf(s.t[1], G_A5, G_A4);
      break;
    case 4:
      s.e[0] = __builtin_amdgcn_exp2f(s.e[0]); s.e[1] = __builtin_amdgcn_exp2f(s.e[1]);
      s.q[2] = fmaf(s.t[2], G_A5, G_A4); s.q[3] = fmaf(s.t[3], G_A5, G_A4);
      s.q[0] = fmaf(s.q[0], s.t[0], G_A3); s.q[1] = fmaf(s.q[1], s.t[1], G_A3);
      break;
    case 5:
      s.e[2] = __builtin_amdgcn_exp2f(s.e[2]); s.e[3] = __builtin_amdgcn_exp2f(s.e[3]);
      s.q[2] = fmaf(s.q[2], s.t[2], G_A3); s.q[3] = fmaf(s.q[3], s.t[3], G_A3);
      s.q[0] = fmaf(s.q[0], s.t[0], G_A2); s.q[1] = fmaf(s.q[1], s.t[1], G_A2);
      break;
    case 6:
      s.q[2] = fmaf(s.q[2], s.t[2], G_A2); s.q[3] = fmaf(s.q[3], s.t[3], G_A2);
#pragma unroll
      for (int r = 0; r < 4; ++r) s.q[r] = fmaf(s.q[r], s.t[r], G_A1);
      break;
    case 7:
#pragma unroll
      for (int r = 0; r < 4; ++r) s.q[r] *= s.t[r];
      s.q[0] = fmaf(-s.q[0], s.e[0], G_HS); s.q[1] = fmaf(-s.q[1], s.e[1], G_HS);   // HS - Q
      break;
    case 8:
      s.q[2] = fmaf(-s.q[2], s.e[2], G_HS); s.q[3] = fmaf(-s.q[3], s.e[3], G_HS);
#pragma unroll
      for (int r = 0; r < 4; ++r) s.q[r] *= __builtin_fabsf(s.vr[r]);
      break;
    case 9:   // r * PSX * gelu(v) = HS (r v) + |r v| (HS - Q(v))
#pragma unroll
      for (int r = 0; r < 4; ++r) s.v[r] = fmaf(s.vr[r], G_HS, s.q[r]);
      break;
    case 10:   // fp16 hi and the fp32 remainder (kept in e / t for the last slot)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const sdy_f16x2 h2 = __builtin_convertvector(sdy_f32x2{s.v[r], s.v[r + 1]}, sdy_f16x2);
        s.e[r >> 1] = __builtin_bit_cast(float, h2);
        s.t[r] = s.v[r] - (float)h2[0]; s.t[r + 1] = s.v[r + 1] - (float)h2[1];
      }
      break;
    default: break;
  }
}

template <int KSP, int NPART, int MO>
__global__ __launch_bounds__(256, 1) void pair_h3_kernel(const PairParams p) {
  using Cfg = PairCfg<KSP, NPART, MO>;
  constexpr int XROW = Cfg::XROW, NI = Cfg::NI, NF1 = Cfg::NF1, NF2 = Cfg::NF2, N = Cfg::N, NPAD = Cfg::NPAD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* Xs_hi = reinterpret_cast<_Float16*>(smem);
  _Float16* Xs_lo = Xs_hi + PTN * XROW;
  _Float16* Hs = reinterpret_cast<_Float16*>(smem + Cfg::XS_BYTES);   // [chunk][hi | lo][px][PHC]
  float* Cb1 = reinterpret_cast<float*>(smem + Cfg::XS_BYTES + Cfg::HS_BYTES);
  float* Am = Cb1 + PH;   // [12]: the four waves' maxima of part 0 / part 1 of the x tile (0..3 / 4..7), of |b1| (8..11)
  float* Os = MO == 8 ? reinterpret_cast<float*>(Hs) : reinterpret_cast<float*>(smem);   // output tile [rows][64 px]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int h = lane >> 5, l31 = lane & 31;
  const int tpi = (p.HW + PTN - 1) / PTN;
  const int ntiles = tpi * p.B;
  const int t_per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;   // contiguous tile range per workgroup
  const int t_begin = (int)blockIdx.x * t_per;
  const int t_end = (t_begin + t_per < ntiles) ? t_begin + t_per : ntiles;

  // ---- weight ring
  f16x8 r_hi[PRING], r_lo[PRING];
  constexpr int GROUP_BYTES = PGROUP * (int)sizeof(f16x8);
  const char* const wbase = reinterpret_cast<const char*>(p.w) + (size_t)wave * (NPAD + PRING) * GROUP_BYTES;
  unsigned woff = (unsigned)lane * 16u;   // lane's byte offset of the NEXT position to fetch (sdy_ring_ld, common.h)
#pragma unroll
  for (int s = 0; s < PRING; ++s) {
    r_hi[s] = sdy_ring_ld(wbase, woff, 0);
    r_lo[s] = sdy_ring_ld(wbase, woff, GROUP_BYTES / 2);
    woff += GROUP_BYTES;
  }
  // behind the last MFMA of position g: its slot takes the group 16 positions on (a hole of the numbering is not loaded);
  // positions are visited in order, each exactly once with which != 0, so the offset just steps on
  auto refill = [&](int g, int which) {   // which: 0 = the lo fragment, 1 = the hi fragment (after lo), 2 = both
    if ((g + PRING) % NPAD < N) {
      if (which != 1) r_lo[g & 15] = sdy_ring_ld(wbase, woff, GROUP_BYTES / 2);
      if (which != 0) r_hi[g & 15] = sdy_ring_ld(wbase, woff, 0);
    }
    if (which != 0) woff += GROUP_BYTES;
  };

  // ---- x staging: thread = (pixel quad q0, channel octets o0 + 16 i)
  int q0 = tid & 15, o0 = tid >> 4;
  f32x4 xr[NI][8];
  // One 16-byte load of the staging role (idx = 8 i + e: octet o0 + 16 i, channel e of it), branch-free so that the
  // compiler's vmcnt bookkeeping of the weight ring stays exact (a predicated load makes every later ring wait ~16 loads too
  // strict): channels past Cin (zero weights; zeroed at the conversion) and the idle lanes of the last octet round read the
  // last real channel.  Issued as ONE burst where nothing that is needed soon queues up behind it (vmcnt retires in order):
  // spread one behind an MFMA each, every ring refill issued after an x load waits for HBM instead of L2 (measured: the
  // decoder's fc1 4.6k -> 8.9k cycles).
  bool x_ok = true;
  bool bad = false;   // a non-finite value met in some x tile: raised as SDY_FLAG_NONFINITE when the kernel leaves
  const float* lx_base = nullptr;   // (load_x in pieces: the decoder issues the next tile's first part between its chain pieces)
  unsigned lx_pxo = 0u;
  auto load_x_setup = [&](int t) {
    const int tu = __builtin_amdgcn_readfirstlane(t);   // (workgroup-uniform; the base below must sit in SGPRs)
    const int zz = tu / tpi, nn = (tu - zz * tpi) * PTN;
    const bool ok = nn + 4 * q0 < p.HW;
    x_ok = ok;
    lx_base = p.x + (long)zz * p.x_bs;
    lx_pxo = (unsigned)(ok ? nn + 4 * q0 : 0);
  };
  auto load_x_piece = [&](int part, int idx) {   // idx = 8 i + e
    const int i = idx >> 3, e = idx & 7;
    int ch = 16 * KSP * part + 8 * (o0 + 16 * i) + e;
    ch = ch < p.Cin ? ch : p.Cin - 1;
    xr[i][e] = sdy_ld16s(lx_base, ((unsigned)ch * (unsigned)p.HW + lx_pxo) * 4u);
  };
  auto load_x = [&](int t, int part) {
    load_x_setup(t);
#pragma unroll
    for (int idx = 0; idx < 8 * NI; ++idx) load_x_piece(part, idx);
  };
  // DYNAMIC SCALE of the x tile (round 4).  The encoder and the decoder are where un-normalised tensors enter the network -- user
  // data, the last block's output -- so their x tile is not staged with the fixed pre-scale of the other kernels (overflow at
  // |x| >= 4094, subnormal `lo` parts below 0.01) but with a power of two chosen per tile and part from its own maximum:
  // max |x| * sx in [2^11, 2^12).  Any finite input magnitude keeps 22 significant bits relative to its tile's maximum; the
  // factor leaves again through the accumulator scale.  `part_amax`: this thread's values -> the workgroup's maximum
  // (wave reduction, one LDS word per wave and part).  It runs where the part's loads have landed anyway and a barrier the
  // kernel has already stands between it and the reader: behind the previous part's MFMA loop (part > 0), behind fc2 of the
  // previous tile (part 0) -- computed in front of the split, with a barrier of its own, the decoder was 13 % slower.
  auto part_amax = [&](int part) {
    // Branch-free (a select per value made the compiler wrap every load's wait in its own exec-masked block): the maximum over
    // everything this thread loaded, then one AND per octet round drops the lanes whose loads were stand-ins -- pixels past the
    // row's end (x_ok, of the tile the registers were loaded for) and the idle lanes of the last octet round.  Channels past Cin
    // repeat the last real channel of this tile: they do not change the maximum.
    float am = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float a = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const f32x4 v = xr[i][e];
        a = __builtin_fmaxf(a, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)),
                                               __builtin_fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
      }
      const int keep = (x_ok && o0 + 16 * i < 2 * KSP) ? -1 : 0;
      am = __builtin_fmaxf(am, __builtin_bit_cast(float, __builtin_bit_cast(int, a) & keep));
    }
    am = wave_max_nonneg_l63(am);
    if (lane == 63) Am[4 * part + wave] = am;
  };
  auto split_x_impl = [&](int part, int n0, bool full, float sx, auto plain_t) {
    constexpr bool plain = decltype(plain_t)::value;   // a full tile of real channels: no select per value
    // The decoder is the one consumer of the last block's output, which meets no InstanceNorm (whose statistics flag non-finite
    // tensors everywhere else), and the max-based range guard ignores NaNs: here a sum of magnitudes goes NaN / inf with them.
    float nansum = 0.0f;
    const bool ok = full || (n0 + 4 * q0 < p.HW);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int o = o0 + 16 * i;
      if (o < 2 * KSP) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const int px = 4 * q0 + pp;
          sdy_f16x8 vh, vl;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] = (plain || (ok && 16 * KSP * part + 8 * o + e < p.Cin)) ? xr[i][e][pp] * sx : 0.0f;
          sdy_split8(v, vh, vl);
          if (MO != 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) nansum += __builtin_fabsf(v[e]);
          }
          *reinterpret_cast<sdy_f16x8*>(Xs_hi + px * XROW + 8 * o) = vh;
          *reinterpret_cast<sdy_f16x8*>(Xs_lo + px * XROW + 8 * o) = vl;
        }
      }
    }
    if (MO != 8) bad |= !(nansum <= 3.0e38f);
  };
  auto split_x = [&](int part, int n0, bool full, float sx) {
    if (full && 16 * KSP * (part + 1) <= p.Cin) split_x_impl(part, n0, full, sx, std::true_type{});   // (workgroup-uniform)
    else split_x_impl(part, n0, full, sx, std::false_type{});
  };

  Cb1[tid] = p.b1 ? p.b1[tid] : 0.0f;
  {   // max |b1| (with the L1 bound of W1 below: a bound of the hidden activation from the x tile's maximum)
    const float bm = wave_max_nonneg_l63(__builtin_fabsf(p.b1 ? p.b1[tid] : 0.0f));
    if (lane == 63) Am[8 + wave] = bm;
  }
  // max over the hidden rows of sum_k |W1[row][k]|, stored behind the weight stream by sdy_pair_h3_pack
  const float w1_l1 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.w) + Cfg::STREAM_BYTES);
  if (t_begin < t_end) { load_x(t_begin, 0); part_amax(0); }
  double psum[MO == 8 ? 16 : 1], psq[MO == 8 ? 16 : 1];
#pragma unroll
  for (int i = 0; i < (MO == 8 ? 16 : 1); ++i) { psum[i] = 0.0; psq[i] = 0.0; }
  __syncthreads();

  for (int tile = t_begin; tile < t_end; ++tile) {
    asm volatile("" : "+v"(l31), "+v"(h), "+v"(q0), "+v"(o0));   // (lane indices laundered per tile: see mlp_h3.hip)
    const int z = tile / tpi;
    const int n0 = (tile - z * tpi) * PTN;
    const bool full = n0 + PTN <= p.HW;
    const int tile_it = tile - t_begin;
    auto stamp = [&](int i) {
      if (SDY_STAMPS_ON && p.stamps && blockIdx.x == 3 && tid == 0 && tile_it >= 2 && tile_it < 6)
        p.stamps[(tile_it - 2) * 16 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    f32x16 acc[2][2];     // [chunk][pixel tile]: hidden rows 128 c + 32 wave .. +32
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.0f;
    // Per-tile scales (set when the last part is staged, before the first chain slot runs):
    //   s1_t = 1 / (w1_scale * sx): the fc1 accumulators hold w1_scale * sx * (W1 . x);
    //   r_t (a power of two): the hidden activation is staged as r_t * PSX * GELU(.), with r_t the largest power of two
    //         that keeps the BOUND |hidden| <= max_row ||W1 row||_1 * max |x| + max |b1| below 2^15 after the PSX --
    //         below 1 for |x| of a few hundred and more (no fp16 overflow), above 1 for small tiles (their `lo` parts stay
    //         normal fp16 numbers); fc2's accumulator scale takes the 1 / r_t.
    int se = 127;
    float am_all = 0.0f, s1_t = p.s1, r_t = 1.0f, s2_t = p.s2;

    // ---- chain: bias + GELU + split of the accumulators -> hidden chunk in LDS, 8 pieces (pixel tile j, row group g4) of
    //      12 slots each
    auto chain_store = [&](const Piece& s, int c, int j, int g4) {
      _Float16* Hh = Hs + c * (2 * PTN * PHC);
      _Float16* Hl = Hh + PTN * PHC;
      f16x4 vh, vl;
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const sdy_f16x2 h2 = __builtin_bit_cast(sdy_f16x2, s.e[r >> 1]);
        const sdy_f16x2 l2 = __builtin_convertvector(sdy_f32x2{s.t[r], s.t[r + 1]}, sdy_f16x2);
        vh[r] = h2[0]; vh[r + 1] = h2[1];
        vl[r] = l2[0]; vl[r + 1] = l2[1];
      }
      const int off = hs_off(32 * j + l31, 4 * wave + g4) + 4 * h;   // local k = 32 wave + 8 g4 + 4 h .. +3
      *reinterpret_cast<f16x4*>(Hh + off) = vh;
      *reinterpret_cast<f16x4*>(Hl + off) = vl;
    };
    auto chain_slot = [&](Piece& s, int st, int c, int j, int g4) {
      if (st == 0) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(Cb1 + PHC * c + 32 * wave + 4 * h + 8 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) s.v[r] = fmaf(acc[c][j][4 * g4 + r], s1_t, b4[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) s.vr[r] = s.v[r] * r_t;
      } else if (st < 11) {
        gelu_slot(s, st);
      } else {
        chain_store(s, c, j, g4);
      }
    };
    // `lx0`, `lx_per`: the next tile's x loads lx0 + lx_per * (piece - first_piece) ... are issued behind this chain's pieces
    auto chain_alone = [&](int c, int first_piece, int lx0 = 0, int lx_per = 0) {
#pragma unroll
      for (int pc = first_piece; pc < 8; ++pc) {
        Piece s;
#pragma unroll
        for (int st = 0; st < 12; ++st) chain_slot(s, st, c, pc & 1, pc >> 1);
#pragma unroll
        for (int k = 0; k < lx_per; ++k) {
          const int idx = lx0 + lx_per * (pc - first_piece) + k;
          if (idx < 8 * NI) load_x_piece(0, idx);
        }
      }
    };

    // ---- fc1 over the parts of the input.  Every part but the last: k-step major, both chunks behind one set of fragment
    //      reads.  The last part: chunk major, and the first NPI pieces of chain(0) ride in the slots beside chunk 1's MFMAs.
    constexpr int NPI = !kChainInFc1 ? 0 : (KSP / 2 < 8 ? KSP / 2 : 8);
#pragma unroll
    for (int part = 0; part < NPART; ++part) {
      if (part > 0) __syncthreads();          // every wave is done reading the previous part (and has left its maximum)
      // All of the scale arithmetic is on exponents (powers of two, workgroup-uniform -> scalar unit): no division.
      const float am = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int,
          __builtin_fmaxf(__builtin_fmaxf(Am[4 * part], Am[4 * part + 1]), __builtin_fmaxf(Am[4 * part + 2], Am[4 * part + 3])))));
      const int e = (__builtin_bit_cast(int, am) >> 23) & 0xFF;            // (am >= 0)
      bad |= e == 255;   // inf (NaNs: split_x / the statistics)
      // biased exponent of the power of two that puts am into [2^11, 2^12); 2^0 for a zero / subnormal / non-finite maximum
      int se_new = 265 - e;
      se_new = (e == 0 || e == 255) ? 127 : (se_new < 1 ? 1 : (se_new > 253 ? 253 : se_new));
      if (part > 0) {
        // A later part keeps the scale of the one before it where its own maximum then lies in [2^8, 2^15) (22 bits relative
        // to the maximum all the same) -- nearly always; otherwise the accumulators, which hold sx * (W1 . x) of the earlier
        // parts, move to the new scale (exact).
        const int te = e + se - 127;
        if (e == 0 || (te >= 127 + 8 && te < 127 + 15)) se_new = se;
        if (se_new != se) {
          int d = se_new - se;
          d = d < -126 ? -126 : (d > 127 ? 127 : d);
          const float ratio = __builtin_bit_cast(float, (127 + d) << 23);
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[c][j][r] *= ratio;
        }
      }
      se = se_new;
      const float sx = __builtin_bit_cast(float, se << 23);
      am_all = sgpr_f(__builtin_fmaxf(am_all, am));
      if (part == NPART - 1) {
        s1_t = sgpr_f(p.s1 * __builtin_bit_cast(float, (254 - se) << 23));      // p.s1 / sx
        const float bmax = __builtin_fmaxf(__builtin_fmaxf(Am[8], Am[9]), __builtin_fmaxf(Am[10], Am[11]));
        const float hb = fmaf(w1_l1, am_all, bmax);                      // bound of |hidden| on this tile
        const int eh = (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hb)) >> 23) & 0xFF;
        // hb < 2^(eh - 126): r = 2^(137 - eh) keeps PSX * r * |hidden| < 2^15; at most 2^40 (an all-zero tile and bias), 1 for
        // a non-finite bound (flagged above or by split_x)
        int re = 264 - eh;
        re = eh == 255 ? 127 : (re > 167 ? 167 : (re < 1 ? 1 : re));
        r_t = __builtin_bit_cast(float, re << 23);
        s2_t = sgpr_f(p.s2 * __builtin_bit_cast(float, (254 - re) << 23));      // p.s2 / r_t
      }
      split_x(part, n0, full, sx);
      if (part + 1 < NPART) load_x(tile, part + 1);   // the x registers are free: they take the next part
      __syncthreads();
      stamp(1 + 2 * part);
      f16x8 bh[2][2], bl[2][2];
      auto ldb1 = [&](int set, int ks, int what) {   // (hi j0, hi j1, lo j0, lo j1)
        const int j = what & 1;
        const int off = (32 * j + l31) * XROW + 8 * (2 * ks + h);
        if (what < 2) bh[set][j] = *reinterpret_cast<const f16x8*>(Xs_hi + off);
        else bl[set][j] = *reinterpret_cast<const f16x8*>(Xs_lo + off);
      };
#pragma unroll
      for (int what = 0; what < 4; ++what) ldb1(0, 0, what);
      if (part + 1 < NPART || !kChainInFc1) {
#pragma unroll
        for (int ks = 0; ks < KSP; ++ks) {
          const int cur = ks & 1;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int g = 2 * (part * KSP + ks) + c;
            const f16x8 a_lo = r_lo[g & 15], a_hi = r_hi[g & 15];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              const int j = k & 1;
              if (SDY_H3_PASSES == 3 || k >= 4)
                acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k < 2 ? a_lo : a_hi, (k >= 2 && k < 4) ? bl[cur][j] : bh[cur][j],
                                                                   acc[c][j], 0, 0, 0);
              // one memory instruction behind each MFMA (mlp_h3.hip, SDY_MLP_PINNED)
              if (c == 0 && k < 4) { if (ks + 1 < KSP) ldb1(cur ^ 1, ks + 1, k); }
              else if (k == 4) refill(g, 0);
              else if (k == 5) refill(g, 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      } else {
        Piece ps;
        sdy_static_for<0, 2 * KSP>([&](auto i_c) {
          constexpr int i = decltype(i_c)::value;
          constexpr int c = i / KSP, ks = i % KSP, cur = i & 1;
          const int g = 2 * part * KSP + i;
          const f16x8 a_lo = r_lo[g & 15], a_hi = r_hi[g & 15];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            const int j = k & 1;
            if (SDY_H3_PASSES == 3 || k >= 4)
              acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k < 2 ? a_lo : a_hi, (k >= 2 && k < 4) ? bl[cur][j] : bh[cur][j],
                                                                 acc[c][j], 0, 0, 0);
            if (k < 4) { if (i + 1 < 2 * KSP) ldb1(cur ^ 1, (ks + 1) % KSP, k); }
            else if (k == 4) refill(g, 0);
            else refill(g, 1);
            if (c == 1 && 6 * ks + k < 12 * NPI) {
              const int pc = (6 * ks + k) / 12;
              chain_slot(ps, (6 * ks + k) % 12, 0, pc & 1, pc >> 1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      }
      if (part + 1 < NPART) part_amax(part + 1);   // (its loads were issued in front of this part's MFMA loop)
    }
    stamp(4);
    // The decoder's fc2 is 48 MFMAs: issued in front of its second chunk, the next tile's pixels had 2.5k cycles to arrive and
    // part_amax below waited 2.2k more for them (stamps).  Its x registers are free once the last part is split, and the two
    // chains that follow touch neither the ring nor the vector-memory queue: the burst goes in front of them.
    constexpr bool kEarlyX = SDY_PAIR_EARLY_X && MO != 8;
    if (kEarlyX) {   // ... spread over the chain pieces: a burst stalls at issue for 2-3k cycles wherever it stands
      const int nt = tile + 1;
      load_x_setup(nt < t_end ? nt : tile);
      constexpr int P0 = 8 - NPI, PER = (8 * NI + P0 + 8 - 1) / (P0 + 8);
      chain_alone(0, NPI, 0, PER);
      chain_alone(1, 0, PER * P0, PER);
#pragma unroll
      for (int idx = PER * (P0 + 8); idx < 8 * NI; ++idx) load_x_piece(0, idx);
      // (the encoder keeps its burst in front of fc2's second chunk: its chain(1) rides beside fc2(0), and with the x registers
      //  live across that phase the 128-channel instantiation spills -- measured: 1.442 -> 1.436 ms, nothing)
    } else {
      chain_alone(0, NPI);
      if (MO != 8) chain_alone(1, 0);
    }
    stamp(5);
    __syncthreads();
    stamp(6);

    // ---- fc2
    const int e_col = n0 + 4 * (tid & 15);
    const bool e_ok = full || e_col < p.HW;
    const unsigned e_ro = (unsigned)((tid >> 4) * p.HW + (e_ok ? e_col : 0)) * 4u;
    f32x16 oacc[MO == 8 ? 2 : 1][MO == 8 ? 2 : 1];
#pragma unroll
    for (int mi = 0; mi < (MO == 8 ? 2 : 1); ++mi)
#pragma unroll
      for (int j = 0; j < (MO == 8 ? 2 : 1); ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[mi][j][r] = 0.0f;
    f32x4 rres[MO == 8 ? 16 : 1];

#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const _Float16* Hh = Hs + c * (2 * PTN * PHC);
      const _Float16* Hl = Hh + PTN * PHC;
      if (c == 1) {
        stamp(7);
        if (MO == 8) __syncthreads();          // chain(1) wrote chunk 1 beside fc2(0)
        const int nt = tile + 1;
        if (!(SDY_PAIR_EARLY_X && MO != 8))
          load_x(nt < t_end ? nt : tile, 0);   // next tile's pixels; past the end a harmless re-read keeps it branch-free
        stamp(11);
      }
      if constexpr (MO == 8) {
        f16x8 bh[2][2], bl[2][2];
        auto ldb1 = [&](int set, int t, int what) {
          const int j = what & 1;
          const int off = hs_off(32 * j + l31, 2 * t + h);
          if (what < 2) bh[set][j] = *reinterpret_cast<const f16x8*>(Hh + off);
          else bl[set][j] = *reinterpret_cast<const f16x8*>(Hl + off);
        };
#pragma unroll
        for (int what = 0; what < 4; ++what) ldb1(0, 0, what);
#pragma unroll
        for (int t = 0; t < PKSC; ++t) {
          const int cur = t & 1;
          __builtin_amdgcn_sched_barrier(0);
          Piece ps;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const int g = NF1 + NF2 * c + 2 * t + mi;
            const f16x8 a_lo = r_lo[g & 15], a_hi = r_hi[g & 15];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              const int j = k & 1;
              if (SDY_H3_PASSES == 3 || k >= 4)
                oacc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k < 2 ? a_lo : a_hi, (k >= 2 && k < 4) ? bl[cur][j] : bh[cur][j],
                                                                    oacc[mi][j], 0, 0, 0);
              if (mi == 0 && k < 4 && t + 1 < PKSC) ldb1(cur ^ 1, t + 1, k);
              if (c == 0) chain_slot(ps, 6 * mi + k, 1, t & 1, t >> 1);   // chain(1): piece t, one slot per MFMA
              __builtin_amdgcn_sched_barrier(0);
            }
            refill(g, 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        const int mi = wave >> 1, j = wave & 1;   // this wave's output tile
        (void)mi;
        f16x8 bh[2], bl[2];
        auto ldb1 = [&](int set, int t, int what) {
          const int off = hs_off(32 * j + l31, 2 * t + h);
          if (what == 0) bh[set] = *reinterpret_cast<const f16x8*>(Hh + off);
          else bl[set] = *reinterpret_cast<const f16x8*>(Hl + off);
        };
        ldb1(0, 0, 0); ldb1(0, 0, 1);
#pragma unroll
        for (int t = 0; t < PKSC; ++t) {
          const int cur = t & 1;
          const int g = NF1 + NF2 * c + t;
          const f16x8 a_lo = r_lo[g & 15], a_hi = r_hi[g & 15];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            if (SDY_H3_PASSES == 3 || k == 2)
              oacc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k == 0 ? a_lo : a_hi, k == 1 ? bl[cur] : bh[cur], oacc[0][0], 0, 0, 0);
            if (k < 2) { if (t + 1 < PKSC) ldb1(cur ^ 1, t + 1, k); }
            else refill(g, 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    stamp(12);
    part_amax(0);   // the next tile's first part (loaded beside fc2's second chunk); the barriers below publish it
    // the holes of the numbering: their slots take the next tile's first groups
#pragma unroll
    for (int g = N; g < NPAD; ++g) refill(g, 2);
    woff -= NPAD * GROUP_BYTES;   // the refills above fetched the stream's tail = the next tile's first 16 groups
    stamp(8);

    // ---- epilogue: accumulators -> LDS [rows][64 px] -> addend + 16-byte row stores (+ statistics)
    if constexpr (MO == 8) {
      if (p.add) {
        const float* az = p.add + (long)z * p.add_bs;
#pragma unroll
        for (int i = 0; i < 16; ++i) rres[i] = sdy_ld16s(az + (long)(16 * i) * p.HW, e_ro);
      }
      __syncthreads();   // every wave is done reading the hidden chunks: their storage takes the output tile
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int row0 = 64 * wave + 32 * mi + 4 * h;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            Os[(row0 + (r & 3) + 8 * (r >> 2)) * PTN + 32 * j + l31] = oacc[mi][j][r] * s2_t;
      }
      __syncthreads();
      stamp(9);
      if (e_ok) {
        float* oz = p.out + (long)z * p.out_bs;
        const float* os = Os + (tid >> 4) * PTN + 4 * (tid & 15);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          f32x4 v = *reinterpret_cast<const f32x4*>(os + 16 * i * PTN);
          if (p.add) v += rres[i];
          sdy_st16s(oz + (long)(16 * i) * p.HW, e_ro, v);
          if (p.stats) {
            psum[i] += (double)((v.x + v.y) + (v.z + v.w));
            psq[i] += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
          }
        }
      }
      if (p.stats) {
        const int nt = tile + 1;
        if (nt >= t_end || nt / tpi != z) {   // workgroup-uniform: last tile here of image z
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            double s1 = psum[i], s2 = psq[i];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
              s1 += __shfl_xor(s1, m, 64);
              s2 += __shfl_xor(s2, m, 64);
            }
            if ((tid & 15) == 0) {
              double* st = p.stats + ((long)z * PH + (tid >> 4) + 16 * i) * 2;
              __hip_atomic_fetch_add(st, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_fetch_add(st + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            psum[i] = 0.0; psq[i] = 0.0;
          }
        }
      }
    } else {
      {
        const int row0 = 32 * (wave >> 1) + 4 * h, px = 32 * (wave & 1) + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) Os[(row0 + (r & 3) + 8 * (r >> 2)) * PTN + px] = oacc[0][0][r] * s2_t;
      }
      __syncthreads();
      stamp(9);
      if (e_ok) {
        float* oz = p.out + (long)z * p.out_bs;
        const float* az = p.add ? p.add + (long)z * p.add_bs : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (tid >> 4) + 16 * i;
          if (row < p.Cout) {
            f32x4 v = *reinterpret_cast<const f32x4*>(Os + row * PTN + 4 * (tid & 15));
            if (az) v += sdy_ld16s(az + (long)(16 * i) * p.HW, e_ro);
            sdy_st16s(oz + (long)(16 * i) * p.HW, e_ro, v);
          }
        }
      }
    }
    stamp(10);
    __syncthreads();   // the output tile's storage (hidden chunks / x tile) is free for the next tile
  }
  if (bad && p.flags) atomicOr(p.flags, (unsigned)SDY_FLAG_NONFINITE);   // (once, outside the tile loop: no live state across it)
}

float pick_scale(const float* w, size_t n) {   // power of two that puts max|w| in [2^12, 2^13)
  float mx = 0.f;
  for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(w[i]));
  if (!(mx > 0.f) || !std::isfinite(mx)) return 1.0f;
  int e;
  std::frexp(mx, &e);
  return std::ldexp(1.0f, 13 - e);
}

// (hi, lo) A-fragment pair of rows 32 mt .. +32, columns 16 ks .. +16 of the row-major [rows][K] matrix w, zero outside it
void put_group(_Float16* dst, const float* w, int rows, int K, int mt, int ks, float s) {
  for (int ln = 0; ln < 64; ++ln)
    for (int e = 0; e < 8; ++e) {
      const int r = 32 * mt + (ln & 31), k = 16 * ks + 8 * (ln >> 5) + e;
      const float v = (r < rows && k < K) ? w[(size_t)r * K + k] * s : 0.0f;
      const _Float16 hv = (_Float16)v;
      dst[ln * 8 + e] = hv;
      dst[64 * 8 + ln * 8 + e] = (_Float16)(v - (float)hv);
    }
}

struct Shape { int ksp, npart, mo; };
bool pick_shape(int Cin, int hidden, int Cout, Shape* s) {
  if (hidden != PH || Cin < 1 || Cout < 1) return false;
  if (Cout == 256) {
    if (Cin <= 80) { *s = {5, 1, 8}; return true; }
    if (Cin <= 144) { *s = {9, 1, 8}; return true; }
    return false;
  }
  if (Cout <= 64) {
    if (Cin <= 352) { *s = {11, 2, 2}; return true; }
    if (Cin <= 416) { *s = {13, 2, 2}; return true; }
  }
  return false;
}

// (the weight stream + 64 bytes of tail: [0] = max over the hidden rows of sum_k |W1[row][k]|, float)
template <int KSP, int NPART, int MO>
size_t pack_bytes() { return PairCfg<KSP, NPART, MO>::STREAM_BYTES + 64; }

template <int KSP, int NPART, int MO>
void pack(const float* w1, const float* w2, int Cin, int Cout, float s1, float s2, std::vector<_Float16>& buf) {
  using Cfg = PairCfg<KSP, NPART, MO>;
  const size_t gh = (size_t)PGROUP * 8;   // halfs per group
  static_assert(Cfg::STREAM_BYTES == (size_t)4 * (Cfg::NPAD + PRING) * PGROUP * 8 * sizeof(_Float16), "stream size");
  buf.assign((size_t)4 * (Cfg::NPAD + PRING) * gh + 32, (_Float16)0.0f);   // + 64 bytes of tail
  {
    float l1 = 0.0f;
    for (int r = 0; r < PH; ++r) {
      double acc = 0.0;
      for (int k = 0; k < Cin; ++k) acc += std::fabs((double)w1[(size_t)r * Cin + k]);
      l1 = std::fmax(l1, (float)acc);
    }
    std::memcpy(buf.data() + (size_t)4 * (Cfg::NPAD + PRING) * gh, &l1, sizeof(float));
  }
  for (int w = 0; w < 4; ++w) {
    _Float16* base = buf.data() + (size_t)w * (Cfg::NPAD + PRING) * gh;
    _Float16* d = base;
    const int n_major = kChainInFc1 ? (NPART - 1) * KSP : NPART * KSP;
    for (int ks = 0; ks < n_major; ++ks)             // every part but the last: k-step major
      for (int c = 0; c < 2; ++c, d += gh) put_group(d, w1, PH, Cin, 4 * c + w, ks, s1);   // hidden rows 128 c + 32 w
    for (int c = 0; c < 2; ++c)                      // the last part: chunk major
      for (int ks = n_major; ks < NPART * KSP; ++ks, d += gh) put_group(d, w1, PH, Cin, 4 * c + w, ks, s1);
    for (int c = 0; c < 2; ++c)
      for (int t = 0; t < PKSC; ++t) {
        if (MO == 8) {
          for (int mi = 0; mi < 2; ++mi, d += gh) put_group(d, w2, Cout, PH, 2 * w + mi, PKSC * c + t, s2);
        } else {
          put_group(d, w2, Cout, PH, w >> 1, PKSC * c + t, s2);
          d += gh;
        }
      }
    // (holes stay zero) + the first 16 groups again: the ring's refills run one 16-block ahead across the tile boundary
    std::copy(base, base + PRING * gh, base + (size_t)Cfg::NPAD * gh);
  }
}

template <int KSP, int NPART, int MO>
int launch(const PairParams& p, hipStream_t stream) {
  using Cfg = PairCfg<KSP, NPART, MO>;
  int n_cu = 0;
  SDY_TRY(sdy_cu_count(&n_cu));
  const long ntiles = (long)((p.HW + PTN - 1) / PTN) * p.B;
  dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu));
  static SdyOncePerDevice once;
  std::atomic<bool>* attr_done = nullptr;
  SDY_TRY(once.slot(&attr_done));
  if (!*attr_done) {
    SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_h3_kernel<KSP, NPART, MO>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
    *attr_done = true;
  }
  hipLaunchKernelGGL((pair_h3_kernel<KSP, NPART, MO>), grid, dim3(256), Cfg::LDS_BYTES, stream, p);
  return sdy_launch_status();
}

#define PAIR_DISPATCH(s, CALL)                                      \
  do {                                                              \
    if ((s).ksp == 5) { CALL(5, 1, 8); }                            \
    else if ((s).ksp == 9) { CALL(9, 1, 8); }                       \
    else if ((s).ksp == 11) { CALL(11, 2, 2); }                     \
    else { CALL(13, 2, 2); }                                        \
  } while (0)

}  // namespace

#if SDY_STAMPS_ON
static unsigned long long* g_pair_stamps = nullptr;
// timing experiments: 4 tiles x 16 phase stamps of wave 0 of workgroup 3 (valid after a launch with SDY_PAIR_STAMPS set)
SDY_DEBUG_EXPORT int sdy_pair_h3_debug_stamps(unsigned long long* host64) {
  if (!g_pair_stamps || !host64) return SDY_ERR_STATE;
  SDY_HIP_TRY(hipDeviceSynchronize());
  SDY_HIP_TRY(hipMemcpy(host64, g_pair_stamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SDY_OK;
}
#endif

extern "C" int sdy_pair_h3_supported(int Cin, int hidden, int Cout) {
  Shape s;
  return pick_shape(Cin, hidden, Cout, &s) ? 1 : 0;
}

extern "C" size_t sdy_pair_h3_pack_bytes(int Cin, int hidden, int Cout) {
  Shape s;
  if (!pick_shape(Cin, hidden, Cout, &s)) return 0;
#define PB(K, P, M) return pack_bytes<K, P, M>()
  PAIR_DISPATCH(s, PB);
#undef PB
  return 0;
}

// w1_host: (hidden, Cin) row-major;  w2_host: (Cout, hidden) row-major
extern "C" int sdy_pair_h3_pack(const float* w1_host, const float* w2_host, int Cin, int hidden, int Cout, void* packed_dev,
                                float* scale1, float* scale2) {
  if (!w1_host || !w2_host || !packed_dev || !scale1 || !scale2) return SDY_ERR_ARG;
  Shape s;
  if (!pick_shape(Cin, hidden, Cout, &s)) return SDY_ERR_UNSUPPORTED;
  const float s1 = pick_scale(w1_host, (size_t)hidden * Cin), s2 = pick_scale(w2_host, (size_t)Cout * hidden);
  std::vector<_Float16> buf;
#define PK(K, P, M) pack<K, P, M>(w1_host, w2_host, Cin, Cout, s1, s2, buf)
  PAIR_DISPATCH(s, PK);
#undef PK
  SDY_HIP_TRY(hipMemcpy(packed_dev, buf.data(), buf.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  *scale1 = s1;
  *scale2 = s2;
  return SDY_OK;
}

extern "C" int sdy_pair_h3(const sdy_pair_args* a, void* stream) {
  if (!a || !a->x || !a->w || !a->out) return SDY_ERR_ARG;
  if (a->B <= 0 || a->HW <= 0) return SDY_ERR_ARG;
  Shape s;
  if (!pick_shape(a->Cin, a->hidden, a->Cout, &s)) return SDY_ERR_UNSUPPORTED;
  if ((a->HW & 3) || (a->x_bstride & 3) || (a->out_bstride & 3) || (a->add && (a->add_bstride & 3))) return SDY_ERR_ALIGN;
  if ((long)a->HW * (a->Cin > 256 ? a->Cin : 256) * 4 >= (1L << 32)) return SDY_ERR_UNSUPPORTED;   // 32-bit lane offsets
  if (a->stats && a->Cout != 256) return SDY_ERR_UNSUPPORTED;
  if (a->B > 65535) return SDY_ERR_UNSUPPORTED;
  PairParams p{};
  p.x = a->x; p.x_bs = a->x_bstride; p.Cin = a->Cin;
  p.w = reinterpret_cast<const f16x8*>(a->w);
  p.b1 = a->b1;
  p.out = a->out; p.out_bs = a->out_bstride; p.Cout = a->Cout;
  p.add = a->add; p.add_bs = a->add_bstride;
  p.HW = a->HW; p.B = a->B;
  p.s1 = 1.0f / a->w1_scale;   // (the x tile's dynamic scale joins per tile)
  p.s2 = 1.0f / (a->w2_scale * PSX);
  p.stats = a->stats;
  SDY_TRY(sdy_flags_ptr(&p.flags));
#if SDY_STAMPS_ON
  if (std::getenv("SDY_PAIR_STAMPS")) {
    if (!g_pair_stamps) SDY_HIP_TRY(hipMalloc(&g_pair_stamps, 64 * sizeof(unsigned long long)));
    p.stamps = g_pair_stamps;
  }
#endif
#define LN(K, P, M) return launch<K, P, M>(p, (hipStream_t)stream)
  PAIR_DISPATCH(s, LN);
#undef LN
  return SDY_ERR_UNSUPPORTED;
}
