// Fused SFNO MLP for gfx950:  out = post( W2 . dropout(GELU(W1 . (a*x + d) + b1)) + b2 ) + residual
// (src/models/sfno/layers.py:73-80 inside the block of src/models/sfno/sfnonet.py:313-335), E = 256 -> 512 -> 256.
//
// Unfused, the hidden activation (B x 512 x HW fp32, 3.3 GB at B = 25) is written by fc1 and read back by fc2: two thirds
// of the HBM traffic of the pair.  Here one workgroup (4 waves, one per SIMD, full register budget) owns 64 pixels end
// to end and the hidden tile never leaves the CU:
//
//   phase 0   x[256 ch][64 px] fp32 -> norm affine, x16, split hi/lo fp16 -> LDS, [px][k] rows, 16-byte chunks
//             XOR-swizzled so that the MFMA fragment reads (ds_read_b128) are bank-conflict free.
//   chunk hc  (4 chunks of 128 hidden channels, a runtime loop):
//     fc1     wave w: hidden rows 128 hc + 32 w .. +32, all 64 px (1 x 2 tiles of 32 x 32), K = 256.
//     chain   bias + GELU + dropout select on the accumulators, x16, split hi/lo -> LDS H[px][128 k] (same swizzle),
//             two alternating buffers: ONE barrier per chunk.  chain(hc) runs in 12-instruction-slot pieces beside the
//             MFMAs of fc2(hc - 1); the chunk's Philox masks are generated beside the MFMAs of its fc1.
//     fc2     wave w: output rows 64 w .. +64 (2 x 2 tiles, accumulators live across all chunks), K = this chunk's
//             128 hidden channels read from LDS.  No cross-wave reduction.
//   epilogue  bias, Philox dropout, drop-path scale, residual add, store.
//
// Weights never touch LDS: they are packed on the host into ONE linear stream per wave, in exactly the order the wave
// consumes them, as MFMA A-fragments (32 rows x 16 k, 1 KB, lane-major): a "group" = (hi, lo) fragment pair = two fully
// coalesced 1 KB loads feeding 6 MFMAs (3 split-precision passes x 2 pixel tiles).  128 groups per wave per tile.  At
// f16 MFMA rates a group is ~190 cycles of matrix work while an L2 round trip under load is > 2000 cycles, so a ring of
// RING = 16 groups (128 VGPRs) is kept in flight -- that, not LDS, is what the register budget is spent on.
//
// Precision is the split-fp16 scheme of gemm_h3.hip (Ah.Bh + Ah.Bl + Al.Bh, fp32 accumulation); the dropout stream and
// epilogue arithmetic equal sdy_conv1x1's for the same (seed, call, stream, batch_offset): the unfused and fused paths
// agree to fp32 round-off and draw identical masks.
#include <cstdlib>
#include <vector>
#include <cmath>
#include <type_traits>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// explicit global address space: a pointer laundered through an empty asm loses its provenance, and hipcc then emits
// FLAT loads, which also count in lgkmcnt -- every LDS-fragment wait became lgkmcnt(0), i.e. a wait for the weight refills
typedef const f16x8 __attribute__((address_space(1)))* wptr_t;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ME = 256;        // embedding channels
constexpr int MH = 512;        // hidden channels
constexpr int TN = 64;         // pixels per workgroup
constexpr int HC = 128;        // hidden channels per chunk
constexpr int NCH = MH / HC;   // chunks
constexpr int KS1 = ME / 16;   // k-steps of fc1
constexpr int KSC = HC / 16;   // k-steps of fc2 per chunk
constexpr int RING = 16;                // groups in flight = groups per block (16 fc1 k-steps, 8 x 2 fc2 groups)
constexpr int NGROUPS = NCH * 2 * RING;      // 128
static_assert(KS1 == RING && 2 * KSC == RING, "ring slot = index inside a block");
constexpr float SX = SDY_ACT_SX;    // activation pre-scale (keeps lo parts out of the fp16 subnormals)
constexpr int GROUP_F8 = 2 * 64;        // f16x8 elements per group (hi fragment, lo fragment)

struct MlpParams {
  const float* x; long x_bs;
  int x_tiled;                             // x TILE-MAJOR: [b][64-pixel tile][256 rows][64] (sdy_mlp_args.x_tiled), x_bs per image
  const float* pa; const float* pd;
  const f16x8* w;                          // [4 waves][NGROUPS + RING groups][hi | lo][64 lanes]
  const float* b1; const float* b2;
  float* out; long out_bs;
  const float* add; long add_bs;
  const float* add_a; const float* add_d;  // optional per-(b, row) affine of the residual: add_a * add + add_d
  int HW, B;
  float s1, s2;                            // accumulator scales: 1 / (w_scale * SX)
  uint32_t drop_thr; float drop_scale;
  uint32_t seed_lo, seed_hi, stream1, stream2, call, batch_offset;
  int rows_per_call;                       // >= 1 (sdy_mlp_args.rows_per_call; 0 there = B)
  const float* batch_scale;
  double* stats;                           // optional [B][ME][2]: sum and sum of squares of the stored output rows
  const float* keep_h; const float* keep_o; // INJECT instantiation only (tests): 0/1 masks (B, MH, HW) / (B, ME, HW)
  unsigned* flags;                         // sticky status word (sdy_status_flags)
  unsigned* head;                          // range headroom word of the x tile (sdy_range_headroom; null unless enabled)
  unsigned long long* stamps;              // timing experiments only (SDY_MLP_STAMPS): per-phase s_memtime of one wave
  SdyImgMap omap;                          // drop-path skip (common.h): image z of this launch (x, pa, pd) is batch row omap.idx[z]
                                           // of add / add_a / add_d / out / stats / batch_scale / keep_* and of the dropout stream
  int add_local;                           // 1: `add` is indexed by the launch's image z like x (sdy_mlp_args.add_by_launch_row)
};

// 4-bit slot swizzle of pixel row px: injective on each ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}
// of the 32 pixels one fragment read touches), so B-fragment reads are conflict free for 256-B and 512-B rows alike.
__device__ __forceinline__ int px_swz(int px) { return (px & 15) ^ (((px >> 4) & 1) * 3); }
// x tile: rows of 256 k = 512 B (two bank rows); position in halfs of the 16-byte chunk c (0..31)
__device__ __forceinline__ int xs_off(int px, int c) { return px * ME + (((c & 16) | ((c ^ px_swz(px)) & 15)) << 3); }
// hidden chunk: rows of 128 k = 256 B (one bank row); chunk c (0..15)
__device__ __forceinline__ int hs_off(int px, int c) { return px * HC + (((c ^ px_swz(px)) & 15) << 3); }

// DROP: dropout on (compile time: a runtime test would split every chain piece into basic blocks and the scheduler
// interleaves VALU with MFMAs only inside one block)
// INJECT (tests only; instantiated in its own translation unit, mlp_h3_inject.hip, so that the two product instantiations
// keep their register allocation): the keep decisions come from mask tensors -- e.g. the ones the reference's nn.Dropout
// layers drew, tests/golden -- instead of the Philox stream; everything else is the same code.
template <bool DROP, bool INJECT = false>
__global__ __launch_bounds__(256, 1) void mlp_h3_kernel(const MlpParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 64 KB x tile + 2 x 32 KB hidden chunk
  _Float16* Xs_hi = reinterpret_cast<_Float16*>(smem);
  _Float16* Xs_lo = Xs_hi + TN * ME;
  _Float16* Hs = Xs_lo + TN * ME;          // [buf][hi | lo][px][HC]
  float* Cf = reinterpret_cast<float*>(Hs + 4 * TN * HC);   // per-image coefficients: pa | pd | add_a | add_d, [4][ME]
  float* Cb2 = Cf + 4 * ME;                                 // fc2 bias [ME] (constant): the epilogue reads it at LDS latency

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // SGPR: row / weight-stream bases become scalar arithmetic
  int h = lane >> 5, l31 = lane & 31;
  // Persistent workgroup (one per CU): a contiguous range of tiles; tile t = (image z, 64-pixel slice n0).
  // The NEXT tile's pixels (and its norm coefficients) are requested into registers while this tile computes, so no
  // tile after the first waits for HBM, and there is no workgroup launch gap between tiles.
  const int tpi = (p.HW + TN - 1) / TN;
  const int ntiles = tpi * p.B;
  // contiguous tile range per workgroup: a workgroup then crosses an image boundary at most ~once, so the per-image
  // flush of the statistics (and the reload of per-image coefficients) is rare instead of every few tiles
  const int t_per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * t_per;
  const int t_end = (t_begin + t_per < ntiles) ? t_begin + t_per : ntiles;
  int z = 0, zo = 0, n0 = 0;   // zo: the image's batch row outside this launch (omap)
  bool full = true;

  // ---- weight ring: slot s holds group s of the block being consumed; refilled for the next block right after its use
  f16x8 r_hi[RING], r_lo[RING];
  // (recomputed where it is needed, from a laundered lane index: kept in a register pair across the tile it is spilled)
  auto wbase_of = [&](int ln) { return (wptr_t)(p.w + (size_t)wave * (NGROUPS + RING) * GROUP_F8 + ln); };
  wptr_t wp = wbase_of(lane);
#pragma unroll
  for (int s = 0; s < RING; ++s) {
    r_hi[s] = wp[s * GROUP_F8];
    r_lo[s] = wp[s * GROUP_F8 + 64];
  }
  wp += RING * GROUP_F8;   // from here on wp[i * GROUP_F8] is group i of the block AFTER the one being consumed

  // ---- x tile prefetch: thread = (pixel quad q, channel octets o and o + 16)
  int q0 = tid & 15, o0 = tid >> 4;
  f32x4 xr[2][8];
  // The next tile's 16 pixel loads and this tile's 16 residual rows are NOT issued as bursts (a burst in front of the last fc2
  // made that phase 6.5-9.5k cycles for 3.7k of MFMA work: the wave stalls at issue while the vector-memory queue drains, and the
  // ring refills behind it wait for HBM): two of each ride behind every one of the epilogue's eight (row tile, row group)
  // rounds, a phase of plain VALU work that touches neither the ring nor vector memory (round 5: 3.006 -> 2.904 ms per launch
  // with dropout, 2.980 -> 2.914 without, profiles/r5c/e2e_ab_mlp_prefetch_late2.txt; spread over the last fc2 instead they
  // cost 1.3 %, e2e_ab_mlp_prefetch_spread.txt).
  const float* pf_base = nullptr;
  const float* res_base = nullptr;
  unsigned pf_off = 0u;
  int pf_rs = 0;
  auto prefetch_setup = [&](int t) {
    const int zz = t / tpi, nn = (t - zz * tpi) * TN;
    const bool ok = nn + 4 * q0 < p.HW;
    pf_base = p.x + (long)zz * p.x_bs + (p.x_tiled ? (long)(t - zz * tpi) * (ME * TN) : 0L);
    pf_rs = p.x_tiled ? TN : p.HW;
    pf_off = p.x_tiled ? (unsigned)(8 * o0 * TN + 4 * q0) * 4u : (unsigned)(8 * o0 * p.HW + (ok ? nn + 4 * q0 : 0)) * 4u;
  };
  auto prefetch_piece = [&](int piece) {   // piece = 8 oc + e
    const int oc = piece >> 3, e = piece & 7;
    xr[oc][e] = sdy_ld16s(pf_base + (long)(8 * 16 * oc + e) * pf_rs, pf_off);
  };
  auto prefetch_x = [&](int t) {   // (the workgroup's first tile: all at once)
    const int zz = t / tpi, nn = (t - zz * tpi) * TN;
    // ragged last slice of an image: clamp the address (branch-free loads keep exact vmcnt counts), zero at conversion
    const bool ok = nn + 4 * q0 < p.HW;
    // (wave-uniform row base in SGPRs) + (one 32-bit lane offset): see sdy_ld16s in common.h
    // (tile-major x: the tile is one contiguous [256 rows][64 px] block -- the producer pads an image's ragged last tile)
    const float* xz = p.x + (long)zz * p.x_bs + (p.x_tiled ? (long)(t - zz * tpi) * (ME * TN) : 0L);
    const int rs = p.x_tiled ? TN : p.HW;
    const unsigned xo = p.x_tiled ? (unsigned)(8 * o0 * TN + 4 * q0) * 4u : (unsigned)(8 * o0 * p.HW + (ok ? nn + 4 * q0 : 0)) * 4u;
#pragma unroll
    for (int oc = 0; oc < 2; ++oc)
#pragma unroll
      for (int e = 0; e < 8; ++e) xr[oc][e] = sdy_ld16s(xz + (long)(8 * 16 * oc + e) * rs, xo);
  };
  // per-image coefficient table in LDS (norm affine of x, affine of the residual): rewritten only when the image changes
  auto load_coeffs = [&](int zz) {
    // the thread index is rebuilt from the SGPR wave index and the lane count of the exec mask: kept in a register across the
    // tile loop for this rare use (image changes) it is the one value the dropout variant spills
    const int t = 64 * wave + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const long c = (long)zz * ME + t, co = (long)sdy_img(p.omap, zz) * ME + t;
    Cf[t] = p.pa ? p.pa[c] * SX : SX;
    Cf[ME + t] = p.pa ? p.pd[c] * SX : 0.0f;
    Cf[2 * ME + t] = p.add_a ? p.add_a[co] : 1.0f;
    Cf[3 * ME + t] = p.add_a ? p.add_d[co] : 0.0f;
  };
  Cb2[tid] = DROP ? p.b2[tid] * p.drop_scale : p.b2[tid];   // (see s2e in the epilogue)
  if (t_begin < t_end) load_coeffs(t_begin / tpi);
  __syncthreads();
  if (t_begin < t_end) prefetch_x(t_begin);
  // per-thread partial statistics of the output rows this thread stores (rows tid / 16 + 16 i), flushed per image
  double psum[16], psq[16];   // fp64: sums of fp32 per-tile partials are then exact, i.e. independent of the tiling
#pragma unroll
  for (int i = 0; i < 16; ++i) { psum[i] = 0.0; psq[i] = 0.0; }

  for (int tile = t_begin; tile < t_end; ++tile) {
  // lane-derived indices are laundered once per tile: otherwise every LDS address of the unrolled loops below is hoisted
  // out of the tile loop (~160 registers of loop invariants, spilled and reloaded from scratch inside the MFMA loops)
  asm volatile("" : "+v"(l31), "+v"(h), "+v"(q0), "+v"(o0));
  const int tile_it = tile - t_begin;
  // Phase marks.  Measurement builds (-DSDY_STAMPS) record s_memtime of one sampled wave here.  The product build keeps a
  // BASIC-BLOCK BOUNDARY at every mark (a branch on a kernel argument that is always zero: two scalar instructions): hipcc's
  // register allocator splits live ranges at block boundaries, and without them the kernel goes over its 512 registers and
  // spills (176 / 272 bytes of scratch with the marks removed, 272 / 272 with sched_barrier(0) in their place;
  // tests/test_code_objects.py).
  auto stamp = [&](int i) {
#if SDY_STAMPS_ON
    if (p.stamps && blockIdx.x == 3 && tid == 0 && tile_it >= 2 && tile_it < 6)
      p.stamps[(tile_it - 2) * 16 + i] = __builtin_amdgcn_s_memtime();
#else
    (void)i; (void)tile_it;
    if (p.stamps && tid == 0) asm volatile("s_nop 0");
#endif
  };
  stamp(0);
  z = tile / tpi;
  zo = sdy_img(p.omap, z);
  n0 = (tile - z * tpi) * TN;
  full = n0 + TN <= p.HW;   // workgroup-uniform

  // ---- phase 0: x tile (already in registers) -> LDS (fp16 hi / lo, [px][k]); a full tile (all but an image's last) takes
  //      no select per value
  auto stage_x = [&](auto full_t) {
    constexpr bool kFull = decltype(full_t)::value;
    float amax = 0.0f;   // range guard of the fp16 split (flagged per tile: no register lives across the tile loop)
    const bool ok = kFull || (n0 + 4 * q0 < p.HW);
#pragma unroll
    for (int oc = 0; oc < 2; ++oc) {
      float av[8], dv[8];
      {
        const int c0 = 8 * (o0 + 16 * oc);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(Cf + c0), a1 = *reinterpret_cast<const f32x4*>(Cf + c0 + 4);
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(Cf + ME + c0), d1 = *reinterpret_cast<const f32x4*>(Cf + ME + c0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { av[e] = a0[e]; av[e + 4] = a1[e]; dv[e] = d0[e]; dv[e + 4] = d1[e]; }
      }
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int px = 4 * q0 + pp;
        f16x8 vh, vl;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (kFull || ok) ? fmaf(xr[oc][e][pp], av[e], dv[e]) : 0.0f;
        sdy_split8(v, vh, vl, amax);
        const int off = xs_off(px, o0 + 16 * oc);
        *reinterpret_cast<f16x8*>(Xs_hi + off) = vh;
        *reinterpret_cast<f16x8*>(Xs_lo + off) = vl;
      }
    }
    sdy_flag_range(p.flags, amax, p.head);
  };
  if (full) stage_x(std::true_type{});   // (workgroup-uniform)
  else stage_x(std::false_type{});
  __syncthreads();

  f32x16 oacc[2][2];   // this wave's 64 output rows x 64 px (zeroed right before the first fc2: not live earlier)
  f32x4 rres[16];
  const int e_col = n0 + 4 * (tid & 15);                    // store phase: first pixel of this thread's quad
  const bool e_ok = full || e_col < p.HW;
  const unsigned e_ro = (unsigned)((tid >> 4) * p.HW + (e_ok ? e_col : 0)) * 4u;   // lane byte offset of rows tid / 16 + 16 i

  constexpr bool do_drop = DROP;
  // stacked calls: image z is trajectory z % rows_per_call of call + z / rows_per_call
  const int zq = zo / p.rows_per_call, zt = zo - zq * p.rows_per_call;
  const uint32_t call_z = p.call + (uint32_t)zq;
  const uint32_t c1_base = (uint32_t)(((uint64_t)(zt + p.batch_offset) * (uint64_t)(MH >> 2)) & 0xFFFFFFFFu);

  // The chunk pipeline (MFMA work of one chunk overlaps the VALU work of the next):
  //     fc1(0) | chain(0) | B | fc1(1) | { chain(1) || fc2(0) } | B | fc1(2) | { chain(2) || fc2(1) } | B | fc1(3) |
  //     { chain(3) || fc2(2) } | B | fc2(3)            (B = barrier; chain(c) writes hidden buffer c & 1, fc2(c) reads it)
  // The weight stream is packed in exactly this block order, 16 groups per block, so ring slot = index in the block.
  f32x16 acc[2];
  f32x4 bv[4];
  auto load_bias = [&](int hc) {   // requested before the block's ring refills (see fc1); an LDS copy read inside the chain
    const int row0 = HC * hc + 32 * wave + 4 * h;   // was slower: the wait for it lands in front of the stage that uses it
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) bv[g4] = *reinterpret_cast<const f32x4*>(p.b1 + row0 + 8 * g4);
  };
  // fc1 of chunk hc: hidden rows 128 hc + 32 wave .. +32 (1 x 2 tiles), K = 256 from the x tile.  The B fragments of
  // k-step ks + 1 are read from LDS while the MFMAs of k-step ks run (two register sets).
// (With the Philox rounds inside the chain the dropout variant LOST 4 % with the pinned loops; with the rounds moved under
//  fc1's MFMAs -- SDY_MLP_PHILOX_AHEAD below -- both variants take them.)
#ifndef SDY_MLP_PINNED
#define SDY_MLP_PINNED 1
#endif
  // Memory instructions of the MFMA loops are pinned ONE behind each MFMA (tools/micro/mfma_valu_overlap.hip, fc1-like
  // loop): a k-step's four LDS fragment reads issued back to back hold up the next MFMA by ~34 cycles (37.8 cycles per MFMA),
  // left to the scheduler they end up in front of the MFMA they feed behind an s_waitcnt lgkmcnt(0) (37-41 in this kernel),
  // one per MFMA a whole k-step ahead of its use they are free (32.3).
  // Dropout masks of a hidden chunk generated AHEAD, under the MFMAs of its fc1 (which leave issue slots free: a Philox
  // round -- two v_mad_u64_u32 and two three-input XORs -- hides behind an MFMA), instead of inside the chain, whose slots
  // beside fc2's MFMAs are full: 4 calls (row groups g4) x 10 rounds, one round behind every other MFMA; words kept in
  // mw[g4][].  In the network: 3.39 -> 3.26 ms per launch (together with the scalar chain and the pinned loops, which the
  // dropout variant could not use while the rounds sat in the chain).
  static_assert(SDY_MLP_PINNED, "the ahead-of-time Philox rounds ride on the pinned fc1 loop");
  uint32_t mw[4][4];
  uint32_t ac0 = 0, ac1 = 0, ac2 = 0, ac3 = 0, ak0 = 0, ak1 = 0;
  constexpr int PR = SDY_PHILOX_ROUNDS;
  constexpr int PSTEP = 96 / (4 * PR);   // the chunk's 4 PR rounds spread over the 96 MFMAs of its fc1: one behind every PSTEP-th
  static_assert(PSTEP >= 2, "at most one round behind every other MFMA of fc1");
  auto philox_ahead = [&](int hc, int idx) {   // idx = PR g4 + round
    const int g4 = idx / PR, rnd = idx % PR;
    if (rnd == 0) {
      const int row0 = HC * hc + 32 * wave + 4 * h;
      ac0 = (uint32_t)(n0 + l31); ac1 = c1_base + (uint32_t)((row0 + 8 * g4) >> 2); ac2 = p.stream1; ac3 = call_z;
      ak0 = p.seed_lo; ak1 = p.seed_hi;
    }
    const uint64_t p0 = (uint64_t)0xD2511F53u * ac0, p1 = (uint64_t)0xCD9E8D57u * ac2;
    const uint32_t m0 = (uint32_t)(p1 >> 32) ^ ac1 ^ ak0, m2 = (uint32_t)(p0 >> 32) ^ ac3 ^ ak1;
    ac0 = m0; ac1 = (uint32_t)p1; ac2 = m2; ac3 = (uint32_t)p0;
    ak0 += 0x9E3779B9u; ak1 += 0xBB67AE85u;
    if (rnd == PR - 1) { mw[g4][0] = ac0; mw[g4][1] = ac1; mw[g4][2] = ac2; mw[g4][3] = ac3; }
  };
  auto fc1 = [&](int hc_fc1) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    f16x8 bh[2][2], bl[2][2];
    auto ldb1 = [&](int set, int ks, int part) {   // part: (hi j0, hi j1, lo j0, lo j1)
      const int j = part & 1;
      const int off = xs_off(32 * j + l31, 2 * ks + h);
      if (part < 2) bh[set][j] = *reinterpret_cast<const f16x8*>(Xs_hi + off);
      else bl[set][j] = *reinterpret_cast<const f16x8*>(Xs_lo + off);
    };
#pragma unroll
    for (int part = 0; part < 4; ++part) ldb1(0, 0, part);
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const int c = ks & 1;
      if constexpr (SDY_MLP_PINNED) {
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 a_lo = r_lo[ks], a_hi = r_hi[ks];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int j = k & 1;
        if (SDY_H3_PASSES == 3 || k >= 4)   // (k < 4: the cross terms, dropped by single-pass measurement builds only)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k < 2 ? a_lo : a_hi, (k >= 2 && k < 4) ? bl[c][j] : bh[c][j], acc[j], 0, 0, 0);
        if (k < 4) { if (ks + 1 < KS1) ldb1(c ^ 1, ks + 1, k); }
        else if (k == 4) r_lo[ks] = wp[ks * GROUP_F8 + 64];
        else r_hi[ks] = wp[ks * GROUP_F8];
        if (DROP && !INJECT && (6 * ks + k) % PSTEP == 0 && (6 * ks + k) / PSTEP < 4 * PR) philox_ahead(hc_fc1, (6 * ks + k) / PSTEP);
        __builtin_amdgcn_sched_barrier(0);
      }
      } else {
      if (ks + 1 < KS1) {
#pragma unroll
        for (int part = 0; part < 4; ++part) ldb1(c ^ 1, ks + 1, part);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) SDY_CROSS_TERM(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_lo[ks], bh[c][j], acc[j], 0, 0, 0));
#pragma unroll
      for (int j = 0; j < 2; ++j) SDY_CROSS_TERM(acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[ks], bl[c][j], acc[j], 0, 0, 0));
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r_hi[ks], bh[c][j], acc[j], 0, 0, 0);
      r_hi[ks] = wp[ks * GROUP_F8];
      r_lo[ks] = wp[ks * GROUP_F8 + 64];
      // one non-MFMA instruction per MFMA: each MFMA shadows ~24 issue cycles, bunched LDS reads / loads do not hide
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // next k-step's LDS read
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // ring refill
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the refill here: the scheduler otherwise sinks it next to its use
      }
    }
    wp += RING * GROUP_F8;
  };
  // One eighth of chain(hc): 4 accumulator values (tile j, row group g4): bias + exact-erf GELU (same arithmetic as
  // gelu_erf in common.h) + Philox dropout, x16, fp16 hi/lo split -> LDS.  Cut into 12 stages of a few VALU instructions
  // each, so that fc2 can issue ONE stage after EACH of the 12 MFMAs of a k-step: an MFMA holds the matrix pipe for 32
  // cycles but the issue port for 4, and the stage runs in that shadow.  (Left to itself the scheduler puts the whole
  // piece after the MFMAs; sched_group_barrier hints were honoured for the first k-step only.)
  struct Piece {
    float v[4], t[4], e[4], q[4];
  };
  // INJECT: keep flag of hidden row `row`, tile pixel `px` of image z from the mask tensor (a global load per value: tests only)
  auto keep_h_at = [&](int row, int px) {
    const int pix = (n0 + px < p.HW) ? n0 + px : p.HW - 1;
    return p.keep_h[((long)zo * MH + row) * p.HW + pix] != 0.0f;
  };
  auto chain_stage = [&](Piece& s, int st, int hc, int j, int g4) {
    const int row0 = HC * hc + 32 * wave + 4 * h;   // hidden row of reg r: row0 + (r & 3) + 8 * (r >> 2)
    const int px = 32 * j + l31;
    switch (st) {
      // (the arithmetic stages work on value PAIRS: v_pk_fma_f32 / v_pk_mul_f32 do two values per instruction at the scalar
      // rate -- the chain is what bounds the interleaved fc2, see the timeline in DESIGN.md; rcp / exp2 stay scalar)
      case 0: {
        const f32x4 b4 = bv[g4];
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const sdy_gf2 v2 = sdy_gf2{acc[j][4 * g4 + r], acc[j][4 * g4 + r + 1]} * p.s1 + sdy_gf2{b4[r], b4[r + 1]};
          s.v[r] = v2.x; s.v[r + 1] = v2.y;
        }
        break;
      }
      // exact-erf GELU times SX: with z = |v| / sqrt(2), t = 1 / (1 + p z) and
      // Q = SX * erfc(z) / 2 = t * poly(t) * exp(-z^2) (A&S 7.1.26, coefficients pre-multiplied by SX / 2),
      //   SX * gelu(v) = (SX/2) v + |v| (SX/2 - Q)          (both signs of v, no compare / select)
      case 1:
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const sdy_gf2 v2 = {s.v[r], s.v[r + 1]};
          const sdy_gf2 t2 = __builtin_elementwise_abs(v2) * (0.3275911f * 0.70710678118654752440f) + 1.0f;
          const sdy_gf2 e2 = v2 * v2;
          s.t[r] = t2.x; s.t[r + 1] = t2.y;
          s.e[r] = e2.x; s.e[r + 1] = e2.y;
        }
        break;
      case 2: {
        s.t[0] = __builtin_amdgcn_rcpf(s.t[0]); s.t[1] = __builtin_amdgcn_rcpf(s.t[1]);
        const sdy_gf2 e2 = sdy_gf2{s.e[0], s.e[1]} * (-0.5f * 1.44269504088896340736f);
        s.e[0] = e2.x; s.e[1] = e2.y;
        break;
      }
      case 3: {
        s.t[2] = __builtin_amdgcn_rcpf(s.t[2]); s.t[3] = __builtin_amdgcn_rcpf(s.t[3]);
        const sdy_gf2 e2 = sdy_gf2{s.e[2], s.e[3]} * (-0.5f * 1.44269504088896340736f);
        s.e[2] = e2.x; s.e[3] = e2.y;
        break;
      }
      case 4: s.e[0] = __builtin_amdgcn_exp2f(s.e[0]); s.e[1] = __builtin_amdgcn_exp2f(s.e[1]); break;
      case 5: s.e[2] = __builtin_amdgcn_exp2f(s.e[2]); s.e[3] = __builtin_amdgcn_exp2f(s.e[3]); break;
      case 6:
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const sdy_gf2 t2 = {s.t[r], s.t[r + 1]};
          sdy_gf2 q2 = t2 * (1.061405429f * (0.5f * SX)) + (-1.453152027f * (0.5f * SX));
          q2 = q2 * t2 + (1.421413741f * (0.5f * SX));
          s.q[r] = q2.x; s.q[r + 1] = q2.y;
        }
        break;
      case 7:
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const sdy_gf2 t2 = {s.t[r], s.t[r + 1]};
          sdy_gf2 q2 = sdy_gf2{s.q[r], s.q[r + 1]} * t2 + (-0.284496736f * (0.5f * SX));
          q2 = q2 * t2 + (0.254829592f * (0.5f * SX));
          s.q[r] = q2.x; s.q[r + 1] = q2.y;
        }
        break;
      case 8:
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const sdy_gf2 q2 = sdy_gf2{s.q[r], s.q[r + 1]} * sdy_gf2{s.t[r], s.t[r + 1]} * sdy_gf2{s.e[r], s.e[r + 1]};   // Q
          s.q[r] = q2.x; s.q[r + 1] = q2.y;
        }
        break;
      case 9:
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const sdy_gf2 v2 = {s.v[r], s.v[r + 1]};
          const sdy_gf2 o2 = v2 * (0.5f * SX) + __builtin_elementwise_abs(v2) * ((0.5f * SX) - sdy_gf2{s.q[r], s.q[r + 1]});
          s.v[r] = o2.x; s.v[r + 1] = o2.y;
        }
        break;
      case 10: {
        if (do_drop) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bool keep;
            if constexpr (INJECT) keep = keep_h_at(row0 + 8 * g4 + r, px);
            else keep = sdy_keep16(mw[g4][r], j, p.drop_thr);
            s.v[r] = keep ? s.v[r] : 0.0f;   // (1 / (1 - p): see s2e)
          }
        }
        break;
      }
      default: {
        _Float16* Hh = Hs + (hc & 1) * (2 * TN * HC);
        _Float16* Hl = Hh + TN * HC;
        f16x4 vh, vl;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {   // fp16 hi / lo split on pairs (packed conversions)
          const sdy_f32x2 x2 = {s.v[r], s.v[r + 1]};
          const sdy_f16x2 h2 = __builtin_convertvector(x2, sdy_f16x2);
          const sdy_f16x2 l2 = __builtin_convertvector(x2 - __builtin_convertvector(h2, sdy_f32x2), sdy_f16x2);
          vh[r] = h2[0]; vh[r + 1] = h2[1];
          vl[r] = l2[0]; vl[r + 1] = l2[1];
        }
        // local k of these 4 values: 32 wave + 8 g4 + 4 h .. +3  ->  16-byte chunk 4 wave + g4, half h
        const int off = hs_off(px, 4 * wave + g4) + 4 * h;
        *reinterpret_cast<f16x4*>(Hh + off) = vh;
        *reinterpret_cast<f16x4*>(Hl + off) = vl;
      }
    }
  };
  // The same piece for the slots BESIDE fc2's MFMAs.  tools/micro/mfma_valu_overlap.hip: one wave hides up to ~5 plain VALU
  // instructions (fp32 FMA / MUL, conversions, v_fma_mix) under each of its own MFMAs (MFMA + 5 x v_fma_f32 = 34 cycles, MFMA
  // alone 32; every further one +4.7), v_rcp / v_exp mostly (+2.6 each), but NOT packed fp32: v_pk_fma_f32 waits for the
  // matrix pipe (MFMA + 2 x v_pk_fma_f32 = 56 cycles).  So this version is scalar throughout (the file is built with
  // -fno-slp-vectorize) and cut into 12 slots of <= 6 instructions; the stand-alone chain(0) above keeps the packed form,
  // which halves the instruction count where there is no MFMA to hide under.
  auto chain_slot = [&](Piece& s, int st, int hc, int j, int g4) {
    const int row0 = HC * hc + 32 * wave + 4 * h;   // hidden row of reg r: row0 + (r & 3) + 8 * (r >> 2)
    const int px = 32 * j + l31;
    constexpr float CT = 0.3275911f * 0.70710678118654752440f, KAP = -0.5f * 1.44269504088896340736f, HS = 0.5f * SX;
    constexpr float A5 = 1.061405429f * HS, A4 = -1.453152027f * HS, A3 = 1.421413741f * HS, A2 = -0.284496736f * HS,
                    A1 = 0.254829592f * HS;
    // SX * gelu(v) = HS v + |v| (HS - Q),  Q = t poly(t) exp(-v^2 / 2),  t = 1 / (1 + CT |v|)   (as chain_stage)
    switch (st) {
      case 0: {
        const f32x4 b4 = bv[g4];
#pragma unroll
        for (int r = 0; r < 4; ++r) s.v[r] = fmaf(acc[j][4 * g4 + r], p.s1, b4[r]);
        break;
      }
      case 1:
#pragma unroll
        for (int r = 0; r < 4; ++r) s.t[r] = fmaf(__builtin_fabsf(s.v[r]), CT, 1.0f);
        s.e[0] = s.v[0] * KAP; s.e[1] = s.v[1] * KAP;
        break;
      case 2:
        s.t[0] = __builtin_amdgcn_rcpf(s.t[0]); s.t[1] = __builtin_amdgcn_rcpf(s.t[1]);
        s.e[2] = s.v[2] * KAP; s.e[3] = s.v[3] * KAP;
        s.e[0] *= s.v[0]; s.e[1] *= s.v[1];
        break;
      case 3:
        s.t[2] = __builtin_amdgcn_rcpf(s.t[2]); s.t[3] = __builtin_amdgcn_rcpf(s.t[3]);
        s.e[2] *= s.v[2]; s.e[3] *= s.v[3];
        s.q[0] = fmaf(s.t[0], A5, A4); s.q[1] = fmaf(s.t[1], A5, A4);
        break;
      case 4:
        s.e[0] = __builtin_amdgcn_exp2f(s.e[0]); s.e[1] = __builtin_amdgcn_exp2f(s.e[1]);
        s.q[2] = fmaf(s.t[2], A5, A4); s.q[3] = fmaf(s.t[3], A5, A4);
        s.q[0] = fmaf(s.q[0], s.t[0], A3); s.q[1] = fmaf(s.q[1], s.t[1], A3);
        break;
      case 5:
        s.e[2] = __builtin_amdgcn_exp2f(s.e[2]); s.e[3] = __builtin_amdgcn_exp2f(s.e[3]);
        s.q[2] = fmaf(s.q[2], s.t[2], A3); s.q[3] = fmaf(s.q[3], s.t[3], A3);
        s.q[0] = fmaf(s.q[0], s.t[0], A2); s.q[1] = fmaf(s.q[1], s.t[1], A2);
        break;
      case 6:
        s.q[2] = fmaf(s.q[2], s.t[2], A2); s.q[3] = fmaf(s.q[3], s.t[3], A2);
#pragma unroll
        for (int r = 0; r < 4; ++r) s.q[r] = fmaf(s.q[r], s.t[r], A1);
        break;
      case 7:
#pragma unroll
        for (int r = 0; r < 4; ++r) s.q[r] *= s.t[r];
        s.q[0] = fmaf(-s.q[0], s.e[0], HS); s.q[1] = fmaf(-s.q[1], s.e[1], HS);   // HS - Q
        break;
      case 8:
        s.q[2] = fmaf(-s.q[2], s.e[2], HS); s.q[3] = fmaf(-s.q[3], s.e[3], HS);
#pragma unroll
        for (int r = 0; r < 4; ++r) s.q[r] *= __builtin_fabsf(s.v[r]);
        break;
      case 9:
#pragma unroll
        for (int r = 0; r < 4; ++r) s.v[r] = fmaf(s.v[r], HS, s.q[r]);
        if (do_drop) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bool keep;
            if constexpr (INJECT) keep = keep_h_at(row0 + 8 * g4 + r, px);
            else keep = sdy_keep16(mw[g4][r], j, p.drop_thr);
            s.v[r] = keep ? s.v[r] : 0.0f;   // (1 / (1 - p): see s2e)
          }
        }
        break;
      case 10: {   // fp16 hi and the fp32 residual (kept in e / t for the last slot; an inline-asm v_fma_mix_f32 here costs
                   // the register allocator more than the instruction it saves: 124 bytes of scratch)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const sdy_f16x2 h2 = __builtin_convertvector(sdy_f32x2{s.v[r], s.v[r + 1]}, sdy_f16x2);
          s.e[r >> 1] = __builtin_bit_cast(float, h2);
          s.t[r] = s.v[r] - (float)h2[0]; s.t[r + 1] = s.v[r + 1] - (float)h2[1];
        }
        break;
      }
      default: {
        _Float16* Hh = Hs + (hc & 1) * (2 * TN * HC);
        _Float16* Hl = Hh + TN * HC;
        f16x4 vh, vl;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const sdy_f16x2 h2 = __builtin_bit_cast(sdy_f16x2, s.e[r >> 1]);
          const sdy_f16x2 l2 = __builtin_convertvector(sdy_f32x2{s.t[r], s.t[r + 1]}, sdy_f16x2);
          vh[r] = h2[0]; vh[r + 1] = h2[1];
          vl[r] = l2[0]; vl[r + 1] = l2[1];
        }
        const int off = hs_off(px, 4 * wave + g4) + 4 * h;
        *reinterpret_cast<f16x4*>(Hh + off) = vh;
        *reinterpret_cast<f16x4*>(Hl + off) = vl;
      }
    }
  };
  // Measured (B = 25): without dropout fc2 || chain 5.44k -> 4.89k cycles per phase, kernel 3.11 -> 3.08 ms; with dropout it
  // only pays once the Philox rounds are out of the chain (SDY_MLP_PHILOX_AHEAD).
#ifndef SDY_MLP_SCALAR_CHAIN
#define SDY_MLP_SCALAR_CHAIN 1
#endif
  auto chain_piece = [&](int hc, int j, int g4) {
    Piece s;
#pragma unroll
    for (int st = 0; st < 12; ++st) chain_stage(s, st, hc, j, g4);
  };
  // fc2 of chunk hc2 (output rows 64 wave .. +64 += W2[:, chunk] . hidden chunk from LDS), optionally with the chain
  // of chunk hc2 + 1 interleaved: k-step t carries piece t = (tile j = t & 1, row group t >> 1), one stage per MFMA, each (MFMA, stage) pair fenced
  auto fc2 = [&](int hc2, auto with_chain) {
    constexpr bool CHAIN = decltype(with_chain)::value;
    const _Float16* Hh = Hs + (hc2 & 1) * (2 * TN * HC);
    const _Float16* Hl = Hh + TN * HC;
    f16x8 bh[2][2], bl[2][2];
    auto ldb = [&](int set, int t) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int off = hs_off(32 * j + l31, 2 * t + h);
        bh[set][j] = *reinterpret_cast<const f16x8*>(Hh + off);
        bl[set][j] = *reinterpret_cast<const f16x8*>(Hl + off);
      }
    };
    auto ldb1 = [&](int set, int t, int part) {   // one of a k-step's four fragment reads: (hi j0, hi j1, lo j0, lo j1)
      const int j = part & 1;
      const int off = hs_off(32 * j + l31, 2 * t + h);
      if (part < 2) bh[set][j] = *reinterpret_cast<const f16x8*>(Hh + off);
      else bl[set][j] = *reinterpret_cast<const f16x8*>(Hl + off);
    };
    ldb(0, 0);
#pragma unroll
    for (int t = 0; t < KSC; ++t) {
      const int c = t & 1;
      if (!SDY_MLP_PINNED && t + 1 < KSC) ldb(c ^ 1, t + 1);
      __builtin_amdgcn_sched_barrier(0);
      Piece ps;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int s = 2 * t + mi;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int j = k & 1;
          const f16x8 a = (k < 2) ? r_lo[s] : r_hi[s];
          const f16x8 b = (k >= 2 && k < 4) ? bl[c][j] : bh[c][j];
          if (SDY_H3_PASSES == 3 || k >= 4) oacc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, oacc[mi][j], 0, 0, 0);
          if (SDY_MLP_PINNED && mi == 0 && k < 4 && t + 1 < KSC) ldb1(c ^ 1, t + 1, k);   // one LDS read behind each MFMA
          if (SDY_MLP_PINNED && !CHAIN) __builtin_amdgcn_sched_barrier(0);
          if constexpr (CHAIN) {
            if (SDY_MLP_SCALAR_CHAIN) chain_slot(ps, 6 * mi + k, hc2 + 1, t & 1, t >> 1);
            else chain_stage(ps, 6 * mi + k, hc2 + 1, t & 1, t >> 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        r_hi[s] = wp[s * GROUP_F8];
        r_lo[s] = wp[s * GROUP_F8 + 64];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    wp += RING * GROUP_F8;
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  stamp(1);   // x tile in LDS
  load_bias(0);
  fc1(0);
  stamp(2);
#pragma unroll
  for (int pc = 0; pc < 8; ++pc) chain_piece(0, pc & 1, pc >> 1);
  stamp(3);
  __syncthreads();
  stamp(4);
  load_bias(1);
  fc1(1);
  stamp(5);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[mi][j][r] = 0.0f;
#pragma unroll 1
  for (int hc = 1; hc < NCH; ++hc) {
    fc2(hc - 1, T_{});          // || chain(hc)
    stamp(4 + 2 * hc);
    __syncthreads();
    if (hc < NCH - 1) {
      load_bias(hc + 1);
      fc1(hc + 1);
    }
    stamp(5 + 2 * hc);
  }
  {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    wp = wbase_of(ln);   // the refills of the last block fetch block 0 again: the ring is ready for the next tile
  }
  asm volatile("" : "+v"(wp));   // (laundered: otherwise the 16 refill addresses become loop invariants in VGPRs)
  {
    const int nt = tile + 1;
    prefetch_setup(nt < t_end ? nt : tile);   // next tile's pixels (issued in the epilogue); past the end a harmless re-read
  }
  fc2(NCH - 1, F_{});
  stamp(12);
  {
    // residual rows for the final store phase (thread = pixel quad q, rows tid / 16 + 16 i): requested two per round of the
    // epilogue arithmetic below, in front of the next tile's pixels
    res_base = p.add ? p.add + (long)(p.add_local ? z : zo) * p.add_bs : p.x + (long)z * p.x_bs;   // uniform
  }

  // ---- epilogue: bias, dropout, drop-path scale in accumulator layout -> LDS [256 rows][64 px] (the x tile's storage:
  //      every wave passed the last chunk's barrier after its final x read), then residual add + 16-byte row stores
  {
    const float bscale = p.batch_scale ? p.batch_scale[zo] : 1.0f;
    // Both dropouts scale what they keep by 1 / (1 - p): the hidden one is folded into the fc2 accumulator scale, the output
    // one into that scale and the bias (Cb2 holds b2 / (1 - p)) -- no multiply per value, and the fp16 split of the hidden
    // activation sees the unscaled values.
    const float s2e = do_drop ? p.s2 * p.drop_scale * p.drop_scale : p.s2;
    const uint32_t c1_base2 = (uint32_t)(((uint64_t)(zt + p.batch_offset) * (uint64_t)(ME >> 2)) & 0xFFFFFFFFu);
    float* Os = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int row0 = 64 * wave + 32 * mi + 4 * h;
      f32x4 bv[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) bv[g4] = *reinterpret_cast<const f32x4*>(Cb2 + row0 + 8 * g4);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint32_t words[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (do_drop && !INJECT) {   // one call per 4 rows x the pixel pair (l31, l31 + 32)
          const philox4 w = philox4x32((uint32_t)(n0 + l31), c1_base2 + (uint32_t)((row0 + 8 * g4) >> 2), p.stream2,
                                          call_z, p.seed_lo, p.seed_hi);
          words[0] = w.x; words[1] = w.y; words[2] = w.z; words[3] = w.w;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int px = 32 * j + l31;
#pragma unroll
          for (int r4 = 0; r4 < 4; r4 += 2) {   // value pairs: packed fp32 FMA / MUL
            sdy_gf2 o = sdy_gf2{oacc[mi][j][4 * g4 + r4], oacc[mi][j][4 * g4 + r4 + 1]} * s2e +
                        sdy_gf2{bv[g4][r4], bv[g4][r4 + 1]};
            if (do_drop) {
              if constexpr (INJECT) {
                const int pix = (n0 + px < p.HW) ? n0 + px : p.HW - 1;
                const float* km = p.keep_o + ((long)zo * ME + row0 + 8 * g4 + r4) * p.HW + pix;
                o = sdy_gf2{km[0] != 0.0f ? o.x : 0.0f, km[p.HW] != 0.0f ? o.y : 0.0f};
              } else {
                o = sdy_gf2{sdy_keep16(words[r4], j, p.drop_thr) ? o.x : 0.0f,
                            sdy_keep16(words[r4 + 1], j, p.drop_thr) ? o.y : 0.0f};
              }
            }
            o = o * bscale;
            Os[(row0 + 8 * g4 + r4) * TN + px] = o.x;
            Os[(row0 + 8 * g4 + r4 + 1) * TN + px] = o.y;
          }
        }
        {   // two residual rows and two pixel pieces of the next tile behind this round
          const int i0 = 2 * (4 * mi + g4);
          rres[i0] = sdy_ld16s(res_base + (long)(16 * i0) * p.HW, e_ro);
          rres[i0 + 1] = sdy_ld16s(res_base + (long)(16 * (i0 + 1)) * p.HW, e_ro);
          prefetch_piece(i0);
          prefetch_piece(i0 + 1);
        }
      }
    }
    stamp(13);
    __syncthreads();
    stamp(14);
    if (e_ok) {
      float* oz = p.out + (long)zo * p.out_bs;   // uniform
      const float* os = Os + (tid >> 4) * TN + 4 * (tid & 15);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (tid >> 4) + 16 * i;
        f32x4 v = *reinterpret_cast<const f32x4*>(os + 16 * i * TN);
        if (p.add) v += rres[i] * Cf[2 * ME + row] + Cf[3 * ME + row];
        sdy_st16s(oz + (long)(16 * i) * p.HW, e_ro, v);
        if (p.stats) {
          psum[i] += (double)sdy_quad_sum(v);
          psq[i] += (double)sdy_quad_sumsq(v);
        }
      }
    }
    // InstanceNorm statistics of what was just stored, for the NEXT block's norm0: the per-thread partials cover the
    // few tiles this workgroup owns of one image; at the image's last tile here they are summed over the 16 lanes that
    // share a row (fp64) and added to the global (sum, sum of squares) with one atomic pair per row
    if (p.stats) {
      const int nt = tile + 1;
      if (nt >= t_end || nt / tpi != z) {   // workgroup-uniform
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          double s1 = psum[i], s2 = psq[i];
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) {
            s1 += __shfl_xor(s1, m, 64);
            s2 += __shfl_xor(s2, m, 64);
          }
          if ((tid & 15) == 0) {
            double* st = p.stats + ((long)zo * ME + (tid >> 4) + 16 * i) * 2;
            __hip_atomic_fetch_add(st, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(st + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          psum[i] = 0.0; psq[i] = 0.0;
        }
      }
    }
  }
  stamp(15);
  __syncthreads();   // the store phase is done with the LDS tile: the next x tile goes to the same storage
  {
    const int nt = tile + 1;
    if (nt < t_end && nt / tpi != z) {   // workgroup-uniform
      load_coeffs(nt / tpi);
      __syncthreads();
    }
  }
  }   // tile loop
}

// power-of-two scale that puts max|w| in [2^12, 2^13) (same rule as the other split-fp16 packers)
float pick_scale(const float* w, size_t n) {
  float mx = 0.f;
  for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(w[i]));
  if (!(mx > 0.f) || !std::isfinite(mx)) return 1.0f;
  int e;
  std::frexp(mx, &e);
  return std::ldexp(1.0f, 13 - e);
}

// one MFMA A-fragment pair (hi, lo) of rows 32 mt .. +32, columns 16 ks .. +16 of the row-major [.][K] matrix w
void put_group(_Float16* dst, const float* w, int K, int mt, int ks, float s) {
  for (int ln = 0; ln < 64; ++ln)
    for (int e = 0; e < 8; ++e) {
      const float v = w[(size_t)(32 * mt + (ln & 31)) * K + 16 * ks + 8 * (ln >> 5) + e] * s;
      const _Float16 hv = (_Float16)v;
      dst[ln * 8 + e] = hv;
      dst[64 * 8 + ln * 8 + e] = (_Float16)(v - (float)hv);
    }
}

}  // namespace

#ifndef SDY_MLP_INJECT_TU
#if SDY_STAMPS_ON
static unsigned long long* g_stamps = nullptr;
// timing experiments: 4 tiles x 16 phase stamps of wave 0 of workgroup 3 (valid after a launch with SDY_MLP_STAMPS set)
SDY_DEBUG_EXPORT int sdy_mlp_h3_debug_stamps(unsigned long long* host64) {
  if (!g_stamps || !host64) return SDY_ERR_STATE;
  SDY_HIP_TRY(hipDeviceSynchronize());
  SDY_HIP_TRY(hipMemcpy(host64, g_stamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SDY_OK;
}
#endif

extern "C" int sdy_mlp_h3_supported(int E, int hidden) { return (E == ME && hidden == MH) ? 1 : 0; }

extern "C" size_t sdy_mlp_h3_pack_bytes(int E, int hidden) {
  if (!sdy_mlp_h3_supported(E, hidden)) return 0;
  return (size_t)4 * (NGROUPS + RING) * GROUP_F8 * sizeof(f16x8);
}

// w1_host: (hidden, E) row-major = mlp.fwd.0.weight;  w2_host: (E, hidden) row-major = mlp.fwd.{2|3}.weight
extern "C" int sdy_mlp_h3_pack(const float* w1_host, const float* w2_host, int E, int hidden, void* packed_dev,
                               float* scale1, float* scale2) {
  if (!w1_host || !w2_host || !packed_dev || !scale1 || !scale2) return SDY_ERR_ARG;
  if (!sdy_mlp_h3_supported(E, hidden)) return SDY_ERR_UNSUPPORTED;
  const float s1 = pick_scale(w1_host, (size_t)E * hidden), s2 = pick_scale(w2_host, (size_t)E * hidden);
  const size_t gh = (size_t)GROUP_F8 * 8;   // halfs per group
  std::vector<_Float16> buf((size_t)4 * (NGROUPS + RING) * gh, (_Float16)0.0f);
  for (int w = 0; w < 4; ++w) {
    _Float16* d = buf.data() + (size_t)w * (NGROUPS + RING) * gh;
    auto put_fc1 = [&](int hc) {   // hidden rows 128 hc + 32 w .. +32, all of K = E
      for (int ks = 0; ks < KS1; ++ks, d += gh) put_group(d, w1_host, ME, 4 * hc + w, ks, s1);
    };
    auto put_fc2 = [&](int hc) {   // output rows 64 w .. +64, K = hidden chunk hc
      for (int t = 0; t < KSC; ++t)
        for (int mi = 0; mi < 2; ++mi, d += gh) put_group(d, w2_host, MH, 2 * w + mi, KSC * hc + t, s2);
    };
    // block order of the kernel's chunk pipeline
    put_fc1(0);
    put_fc1(1);
    for (int hc = 1; hc < NCH; ++hc) {
      put_fc2(hc - 1);
      if (hc < NCH - 1) put_fc1(hc + 1);
    }
    put_fc2(NCH - 1);
  }
  SDY_HIP_TRY(hipMemcpy(packed_dev, buf.data(), buf.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  *scale1 = s1;
  *scale2 = s2;
  return SDY_OK;
}

#endif   // !SDY_MLP_INJECT_TU

// (defined by the translation unit mlp_h3_inject.hip: the same launch code around the INJECT instantiation)
int sdy_mlp_h3_inject(const sdy_mlp_args* a, void* stream);

#ifdef SDY_MLP_INJECT_TU
int sdy_mlp_h3_inject(const sdy_mlp_args* a, void* stream) {
#else
extern "C" int sdy_mlp_h3(const sdy_mlp_args* a, void* stream) {
  if (a && (a->keep_hidden || a->keep_out)) {   // injected masks (tests): both or neither, and only with dropout on
    if (!a->keep_hidden || !a->keep_out || !(a->drop_p > 0.0f)) return SDY_ERR_ARG;
    return sdy_mlp_h3_inject(a, stream);
  }
#endif
  if (!a || !a->x || !a->w || !a->b1 || !a->b2 || !a->out) return SDY_ERR_ARG;
  if (a->B <= 0 || a->HW <= 0) return SDY_ERR_ARG;
  if (!sdy_mlp_h3_supported(a->E, a->hidden)) return SDY_ERR_UNSUPPORTED;
  if ((a->pa == nullptr) != (a->pd == nullptr) || (a->add_a == nullptr) != (a->add_d == nullptr)) return SDY_ERR_ARG;
  if ((a->HW & 3) || (a->x_bstride & 3) || (a->out_bstride & 3) || (a->add && (a->add_bstride & 3))) return SDY_ERR_ALIGN;
  if ((long)a->HW * ME * 4 >= (1L << 32)) return SDY_ERR_UNSUPPORTED;   // 32-bit lane offsets inside an image
  if (a->drop_p < 0.0f || a->drop_p >= 1.0f) return SDY_ERR_ARG;
  if (a->B > 65535) return SDY_ERR_UNSUPPORTED;
  MlpParams p{};
  p.x = a->x; p.x_bs = a->x_bstride; p.x_tiled = a->x_tiled; p.pa = a->pa; p.pd = a->pd;
  if (a->x_tiled && !a->add) return SDY_ERR_ARG;   // (without `add` the residual is x itself, read in NCHW order)
  p.w = reinterpret_cast<const f16x8*>(a->w);
  p.b1 = a->b1; p.b2 = a->b2;
  p.out = a->out; p.out_bs = a->out_bstride; p.add = a->add; p.add_bs = a->add_bstride;
  p.add_a = a->add ? a->add_a : nullptr; p.add_d = a->add ? a->add_d : nullptr;
  p.HW = a->HW; p.B = a->B;
  p.s1 = 1.0f / (a->w1_scale * SX);
  p.s2 = 1.0f / (a->w2_scale * SX);
  if (a->drop_p > 0.0f) {
    p.drop_thr = sdy_drop_threshold16(a->drop_p);
    if (p.drop_thr == 0u) p.drop_thr = 1u;
    p.drop_scale = 1.0f / (1.0f - a->drop_p);
  }
  p.seed_lo = (uint32_t)(a->seed & 0xFFFFFFFFu); p.seed_hi = (uint32_t)(a->seed >> 32);
  p.stream1 = a->stream_fc1; p.stream2 = a->stream_fc2; p.call = a->call; p.batch_offset = a->batch_offset;
  // (with out_rows the launch's B counts the mapped rows only: the stacking is a property of the batch rows they map to)
  if (a->rows_per_call < 0 || (!a->out_rows && a->rows_per_call > 0 && a->B % a->rows_per_call)) return SDY_ERR_ARG;
  if (a->out_rows && a->rows_per_call == 0) return SDY_ERR_ARG;
  p.rows_per_call = a->rows_per_call > 0 ? a->rows_per_call : a->B;
  p.batch_scale = a->batch_scale;
  p.stats = a->stats;
  p.keep_h = a->keep_hidden; p.keep_o = a->keep_out;
  if (a->out_rows && !a->add) return SDY_ERR_ARG;   // (without `add` the residual is x itself: one index for both)
  if (a->add_by_launch_row && (!a->out_rows || a->add_a)) return SDY_ERR_ARG;
  p.add_local = a->add_by_launch_row ? 1 : 0;
  SDY_TRY(sdy_img_map_fill(p.omap, a->out_rows, a->B));
  SDY_TRY(sdy_flags_ptr(&p.flags));
  SDY_TRY(sdy_headroom_ptr(SDY_RANGE_MLP, &p.head));
  p.stamps = nullptr;
#if SDY_STAMPS_ON && !defined(SDY_MLP_INJECT_TU)
  if (std::getenv("SDY_MLP_STAMPS")) {
    if (!g_stamps) SDY_HIP_TRY(hipMalloc(&g_stamps, 64 * sizeof(unsigned long long)));
    p.stamps = g_stamps;
  }
#endif
  int n_cu = 0;
  SDY_TRY(sdy_cu_count(&n_cu));
  const long ntiles = (long)((a->HW + TN - 1) / TN) * a->B;
  dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu));   // persistent: one workgroup per CU (128 KB of LDS each)
  constexpr size_t smem = (size_t)(2 * TN * ME + 4 * TN * HC) * sizeof(_Float16) + 5 * ME * sizeof(float);
  static SdyOncePerDevice once;
  std::atomic<bool>* attr_done = nullptr;
  SDY_TRY(once.slot(&attr_done));
#ifdef SDY_MLP_INJECT_TU
  if (!*attr_done) {
    SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_h3_kernel<true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    *attr_done = true;
  }
  hipLaunchKernelGGL((mlp_h3_kernel<true, true>), grid, dim3(256), smem, (hipStream_t)stream, p);
#else
  if (!*attr_done) {
    SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_h3_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_h3_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    *attr_done = true;
  }
  if (p.drop_thr != 0u)
    hipLaunchKernelGGL(mlp_h3_kernel<true>, grid, dim3(256), smem, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(mlp_h3_kernel<false>, grid, dim3(256), smem, (hipStream_t)stream, p);
#endif
  return sdy_launch_status();
}
