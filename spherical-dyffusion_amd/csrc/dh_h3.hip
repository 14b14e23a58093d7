// dhconv (the SFNO filter's complex channel contraction, per degree l) for 256 -> 256 channels as a persistent split-fp16
// kernel in the style of conv_h3.hip / mlp_h3.hip.
//
// Reference: _contract_dhconv, src/models/sfno/contractions.py:159-169 (via factorizations.py:165-186, call site
// s2convolutions.py:182):  out[b,o,l,m] = sum_i x[b,i,l,m] * w[i,o,l]  (complex), only m <= l.
// In the coefficient layout Cs[l][m][b][2C] this is, per degree l, a real GEMM
//     out_l[(m,b)][o'] = sum_i' X_l[(m,b)][i'] * W'_l[i'][o'],   W'_l = [[wr, wi], [-wi, wr]]  (512 x 512),
// over the (l + 1) * B rows with m <= l.
//
// The generic tile GEMM (gemm_h3.hip) cuts N = 512 into four 128-column tiles, so every activation row is fetched and
// split four times and both operands go through LDS.  Here a workgroup (8 waves) owns 64 rows and ALL 512 columns:
//   * the 64 x 512 activation tile is fetched once, split hi/lo and parked in LDS ([row][k], XOR-swizzled, 128 KB);
//   * wave w computes output channels 32w .. 32w+31, real AND imaginary part (out_re = x_re wr - x_im wi, out_im = x_re wi +
//     x_im wr): the weight never touches LDS, it is packed per (l, wave) as a linear stream of (wr, wi) MFMA B-fragment
//     pairs (hi, lo) in consumption order -- 512 KB per degree, half of the real-expanded matrix -- and flows L2 -> registers
//     through an 8-group ring;
//   * the accumulators (rows x columns, column = lane) are transposed per wave through a private 2 KB LDS area and stored
//     as 16 bytes per lane, 256-byte row segments.
// Work distribution: each degree l belongs to ONE XCD (boustrophedon over l, so the (l+1)-proportional work balances), and
// the 32 workgroups of an XCD walk its tile list interleaved -- at any time they sit on the same one or two degrees, whose
// 1 MB weight streams stay in that XCD's 4 MB L2 and are fetched from HBM once.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int DE = 256;            // channels in = out
constexpr int DK = 2 * DE;         // contraction length (ri, c)
constexpr int DN = 2 * DE;         // output columns
constexpr int DTN = 64;            // rows per tile
constexpr int DCB = DE / 16;       // blocks of 16 input channels (16): one re k-step + one im k-step each
constexpr int DWAVES = 8;
constexpr int DRING = 8;           // groups in flight (4 channel blocks x (wr, wi))
constexpr int DGPW = 2 * DCB;      // groups per wave and degree (32)
constexpr int DGROUP = 2 * 64;     // f16x8 elements per group (hi | lo)
constexpr long DLSTRIDE = (long)DWAVES * DGPW * DGROUP;   // f16x8 elements per degree (512 KB)
constexpr float DSX = SDY_ACT_SX;

struct DhParams {
  const float* X; long sX;         // Cs_in,  per-degree stride (floats)
  float* out; long sC;             // Cs_out, per-degree stride
  const f16x8* w;                  // [l][8 waves][16 channel blocks][wr | wi][hi | lo][64 lanes]
  int L, mtr, B;
  int ilv;                         // order of the 2C axis: 0 = [ri][c], 1 = [c / 16][ri][16] (fft.h)
  int tiled;                       // coefficient tensors TILE-MAJOR by order (ilv == 1 only): [m][column tile j][l][64] with
                                   // j = b * 8 + (column / 64) -- row (m, b) of a degree is row index r = m B + b, its eight
                                   // 64-column pieces sit at ((8 r + piece) L + l) * 64: no division, no per-degree stride
  float out_scale;
  unsigned* flags;                 // sticky status word (sdy_status_flags)
  unsigned* head_in;               // range headroom words (sdy_range_headroom; null unless enabled): the rows this kernel stages,
  unsigned* head_out;              // and what it stores -- the Legendre synthesis stages that with the same pre-scale
  unsigned long long* stamps;      // timing experiments only (SDY_DH_STAMPS)
  int B_in;                        // images per order of the INPUT tensor (>= B; tiled only): row (m, b) of this launch is input
                                   // row m B_in + b -- the drop-path skip contracts only the first B of B_in images (capi.hip)
  unsigned b_magic;                // floor(2^32 / B) + 1: m = umulhi(row, b_magic) for row < 2^16 (B >= 2; B == 1: m = row)
  int use_table;                   // degree -> XCD from xcd_of (L <= 256), else the boustrophedon formula
  unsigned char xcd_of[256];       // balanced by the host for this (L, mtr, B): longest-processing-time over the degrees' tiles
};

__device__ __forceinline__ int dh_swz(int r) { return (r & 15) ^ (((r >> 4) & 1) * 3); }
// half offset of 16-byte chunk c (0..63) of row r
__device__ __forceinline__ int dh_off(int r, int c) { return r * DK + (((c & ~15) | ((c ^ dh_swz(r)) & 15)) << 3); }

// degree -> XCD, rows and tiles of a degree.  The boustrophedon rule balances (l + 1) over groups of 16 degrees, but L = 180 / 181
// leaves 4 / 5 of the heaviest degrees over (71 tiles each at B = 25): XCDs 0 .. 4 got 840 tiles, 5 .. 7 got 770, and the launch
// took 27 rounds of 32 workgroups for 25.5.  The host's table (largest degree first to the least loaded XCD) is within one
// tile of even: 26 rounds.
__host__ __device__ __forceinline__ int dh_xcd_formula(int l) { return (l & 8) ? 7 - (l & 7) : (l & 7); }
__device__ __forceinline__ int dh_xcd(const DhParams& p, int l) { return p.use_table ? (int)p.xcd_of[l] : dh_xcd_formula(l); }
__device__ __forceinline__ int dh_rows(const DhParams& p, int l) { return (l + 1 < p.mtr ? l + 1 : p.mtr) * p.B; }

struct TileIt {
  int l, t;   // degree (descending; -1 = done) and tile index inside it
};
__device__ __forceinline__ void dh_advance(const DhParams& p, TileIt& it, int xcd, int step) {
  it.t += step;
  while (it.l >= 0) {
    const int nt = (dh_xcd(p, it.l) == xcd) ? (dh_rows(p, it.l) + DTN - 1) / DTN : 0;
    if (it.t < nt) return;
    it.t -= nt;
    --it.l;
  }
}

__global__ __launch_bounds__(512) void dh_h3_kernel(const DhParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * 64 * 512 * 2 = 128 KB
  _Float16* Xs_hi = reinterpret_cast<_Float16*>(smem);
  _Float16* Xs_lo = Xs_hi + DTN * DK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int h = lane >> 5, l31 = lane & 31;
  int oc = tid & 63, r0 = tid >> 6;   // staging role: octet oc (floats 8 oc .. 8 oc + 7) of rows r0 + 8 i

  const int nslots = (int)gridDim.x >> 3;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  if (slot >= nslots) return;
  TileIt cur{p.L - 1, slot - nslots};
  dh_advance(p, cur, xcd, nslots);
  if (cur.l < 0) return;

  f16x8 r_hi[DRING], r_lo[DRING];
  // ring loads: (wave-uniform stream base of the degree in SGPRs) + (the lane's running offset), sdy_ring_ld in common.h
  constexpr int DGROUP_BYTES = DGROUP * (int)sizeof(f16x8);
  auto w_base = [&](int l) {
    return reinterpret_cast<const char*>(p.w) + ((size_t)l * DLSTRIDE + (size_t)wave * DGPW * DGROUP) * sizeof(f16x8);
  };
  const char* wbase = w_base(cur.l);
  unsigned woff = (unsigned)lane * 16u;
#pragma unroll
  for (int s = 0; s < DRING; ++s) {
    r_hi[s] = sdy_ring_ld(wbase, woff, 0);
    r_lo[s] = sdy_ring_ld(wbase, woff, DGROUP_BYTES / 2);
    woff += DGROUP_BYTES;
  }

  // activation rows of a tile: clamped to row 0 of the degree beyond the ragged edge (zeroed when staged)
  f32x4 xr[8][2];
  auto x_ptr = [&](const TileIt& it, int i) {   // uniform degree base + a 32-bit lane offset
    int rr = r0;
    asm volatile("" : "+v"(rr));   // computed where it is used: 16 hoisted addresses would not fit the register budget
    const int row = it.t * DTN + rr + 8 * i;
    int rc = row < dh_rows(p, it.l) ? row : 0;
    if (p.B_in != p.B) rc += (p.B == 1 ? rc : (int)__umulhi((unsigned)rc, p.b_magic)) * (p.B_in - p.B);   // (uniform branches)
    // (unsigned arithmetic: the element offset passes 2^31 beyond 128 images at 180 x 360; the launcher bounds it by 2^32)
    if (p.tiled) return p.X + ((((unsigned)rc * 8u + (unsigned)(oc >> 3)) * (unsigned)p.L + (unsigned)it.l) * 64u + 8u * (unsigned)(oc & 7));
    return p.X + (long)it.l * p.sX + (unsigned)(rc * DK + 8 * oc);
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float* g = x_ptr(cur, i);
    xr[i][0] = *reinterpret_cast<const f32x4*>(g);
    xr[i][1] = *reinterpret_cast<const f32x4*>(g + 4);
  }

  int tile_it = -1;
  while (true) {
    ++tile_it;
    // laundered per tile: keeps the unrolled loops' LDS addresses from being hoisted into (spilled) loop invariants
    asm volatile("" : "+v"(l31), "+v"(h), "+v"(oc), "+v"(r0));
    const int M = dh_rows(p, cur.l);
    auto stamp = [&](int i) {
      if (SDY_STAMPS_ON && p.stamps && blockIdx.x == 11 && lane == 0 && tile_it >= 2 && tile_it < 6)
        p.stamps[((tile_it - 2) * 8 + wave) * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    if (SDY_STAMPS_ON && p.stamps && blockIdx.x == 11 && tile_it < 64) {   // every tile of the sampled workgroup
      const unsigned long long tm = __builtin_amdgcn_s_memtime();
      const unsigned long long id = (unsigned long long)(cur.l * 1000 + cur.t);
      if (lane == 0 && wave == 0) { p.stamps[256 + tile_it] = tm; p.stamps[320 + tile_it] = id; }
    }
    const int row0 = cur.t * DTN;
    TileIt nxt = cur;
    dh_advance(p, nxt, xcd, nslots);
    const bool more = nxt.l >= 0;
    const TileIt pre = more ? nxt : cur;   // past the end: a harmless re-read

    // ---- phase 0: the tile (already in registers) -> LDS, fp16 hi / lo, [row][k]
    float amax = 0.0f;   // range guard of the fp16 split, flagged per tile (no register lives across the MFMA loop)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + 8 * i;
      const float sc = (row0 + r < M) ? DSX : 0.0f;   // rows past the ragged edge (clamped re-reads of a valid row): zero
      f16x8 vh, vl;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = xr[i][e >> 2][e & 3] * sc;
      sdy_split8(v, vh, vl, amax);
      const int off = dh_off(r, oc);
      *reinterpret_cast<f16x8*>(Xs_hi + off) = vh;
      *reinterpret_cast<f16x8*>(Xs_lo + off) = vl;
    }
    sdy_flag_range(p.flags, amax, p.head_in);
    stamp(1);
    __syncthreads();
    stamp(2);

    // ---- MFMA phase.  Wave w owns the 32 output channels o = 32 w .. 32 w + 31 and computes BOTH parts of them:
    //        out_re += x_re . wr - x_im . wi        out_im += x_re . wi + x_im . wr
    // so one (wr, wi) fragment pair per block of 16 input channels feeds 24 MFMAs -- the real-expanded 512 x 512 matrix
    // [[wr, wi], [-wi, wr]] holds every number twice and streaming it cost 1 MB per tile through the 64 B/clk vector-memory
    // path (ablations in DESIGN.md).  The minus sign is applied to the x_im fragments (sign bits of hi and lo).
    const char* const wnext = w_base(pre.l);
    f32x16 acc[2][2];   // [part: re, im][row tile j]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.0f;
    auto mma3 = [&](f32x16& c, const f16x8& ah, const f16x8& al, const f16x8& bhi, const f16x8& blo) {
      SDY_CROSS_TERM(c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhi, c, 0, 0, 0));
      SDY_CROSS_TERM(c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blo, c, 0, 0, 0));
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhi, c, 0, 0, 0);
    };
    // Two register sets for the LDS fragments (221 -> 255 registers, no scratch; 0.621 -> 0.606 ms per launch,
    // profiles/r4c/e2e_ab_dh_frag_ahead.txt): x_im's of a channel block are requested before the MFMAs on x_re run, the next
    // block's x_re before those on x_im -- a wave that has the SIMD to itself (the younger four at the end of the phase, all of
    // them whenever their partner waits) no longer sits out an LDS round trip per half block.
    f16x8 fa_h[2], fa_l[2], fb_h[2], fb_l[2];
    auto ld_frags = [&](f16x8* fh, f16x8* fl, int kstep) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int off = dh_off(32 * j + l31, 2 * kstep + h);
        fh[j] = *reinterpret_cast<const f16x8*>(Xs_hi + off);
        fl[j] = *reinterpret_cast<const f16x8*>(Xs_lo + off);
      }
    };
    ld_frags(fa_h, fa_l, p.ilv ? 0 : 0);
#pragma unroll
    for (int kb = 0; kb < DCB / 4; ++kb) {
      if (kb == DCB / 4 - 1) {   // the refills of the last block fetch block 0 of the next tile's stream
        wbase = wnext;
        woff -= DGPW * DGROUP_BYTES;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cb = 4 * kb + i, s0 = 2 * i, s1 = 2 * i + 1;   // ring slots of (wr, wi)
        const int kim = p.ilv ? 2 * cb + 1 : DCB + cb;            // k-step of x_im (x_re's: 2 cb / cb, requested a half block ago)
        ld_frags(fb_h, fb_l, kim);
#pragma unroll
        for (int j = 0; j < 2; ++j) mma3(acc[0][j], fa_h[j], fa_l[j], r_hi[s0], r_lo[s0]);   // re += x_re . wr
#pragma unroll
        for (int j = 0; j < 2; ++j) mma3(acc[1][j], fa_h[j], fa_l[j], r_hi[s1], r_lo[s1]);   // im += x_re . wi
        __builtin_amdgcn_sched_barrier(0);
        if (cb + 1 < DCB) ld_frags(fa_h, fa_l, p.ilv ? 2 * (cb + 1) : cb + 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) mma3(acc[1][j], fb_h[j], fb_l[j], r_hi[s0], r_lo[s0]);   // im += x_im . wr
        r_hi[s0] = sdy_ring_ld(wbase, woff, 0);
        r_lo[s0] = sdy_ring_ld(wbase, woff, DGROUP_BYTES / 2);
        woff += DGROUP_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // -x_im
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          u32x4 uh = __builtin_bit_cast(u32x4, fb_h[j]) ^ 0x80008000u, ul = __builtin_bit_cast(u32x4, fb_l[j]) ^ 0x80008000u;
          fb_h[j] = __builtin_bit_cast(f16x8, uh);
          fb_l[j] = __builtin_bit_cast(f16x8, ul);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) mma3(acc[0][j], fb_h[j], fb_l[j], r_hi[s1], r_lo[s1]);   // re -= x_im . wi
        r_hi[s1] = sdy_ring_ld(wbase, woff, 0);
        r_lo[s1] = sdy_ring_ld(wbase, woff, DGROUP_BYTES / 2);
        woff += DGROUP_BYTES;
#ifndef DH_NOPREFETCH
        {   // one 16-byte piece of the next tile per channel block, never a burst
          xr[cb >> 1][cb & 1] = *reinterpret_cast<const f32x4*>(x_ptr(pre, cb >> 1) + 4 * (cb & 1));
        }
#endif
        __builtin_amdgcn_sched_barrier(0);   // keep the loads here (the scheduler otherwise sinks them next to their use)
      }
    }
    stamp(3);
    // ---- epilogue: the accumulators hold channel = lane, i.e. a direct store is 4 bytes per lane (measured: the 64 dword
    // stores per lane took 40 % of the tile).  Each wave transposes 8-row chunks of its 64 x 64 block through a private
    // 2 KB LDS staging area (in-order LDS of one wave: no barrier) and stores 16 bytes per lane, 256-byte row segments.
    {
      float* stg = reinterpret_cast<float*>(smem + 2 * DTN * DK * sizeof(_Float16)) + wave * (8 * 64);
      const int srow = lane >> 4, sc4 = lane & 15;
      // staging column of (part t, channel 32 w + l31) and the global column of staging columns 4 sc4 .. + 3:
      //   [c/16][ri][16]: the wave's 64 outputs are contiguous, [16-block l31 >> 4][t][l31 & 15]
      //   [ri][c]       : two 32-float runs, [t][l31] -> column 256 t + 32 w + ...
      const int scol = p.ilv ? 32 * (l31 >> 4) + (l31 & 15) : l31;   // + (ilv ? 16 : 32) * t
      const int tstep = p.ilv ? 16 : 32;
      const int gcol = p.ilv ? 64 * wave + 4 * sc4 : DE * (sc4 >> 3) + 32 * wave + 4 * (sc4 & 7);
      // tile-major: the wave's 64 outputs of a row are piece `wave` of that row; rows are 8 L 64 floats apart
      const long orow = p.tiled ? 8L * p.L * 64 : (long)DN;
      float* og = p.tiled ? p.out + ((long)row0 * 8 + wave) * p.L * 64 + (long)cur.l * 64
                          : p.out + (long)cur.l * p.sC + (long)row0 * DN;   // uniform: rows are added in SGPR arithmetic
      const unsigned olane = p.tiled ? (unsigned)(srow * orow + 4 * sc4) * 4u
                                     : (unsigned)(srow * DN + gcol) * 4u;   // the lane part of every store address
      // The stored coefficients are what the Legendre synthesis stages as fp16 (x SDY_ACT_SX, leg_par.hip), and that kernel has
      // no register left for a range guard of its own: its input is guarded HERE, where it is produced.
      float omax = 0.0f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const float v0 = acc[t][j][4 * g + q] * p.out_scale, v1 = acc[t][j][4 * g + q + 1] * p.out_scale;
              omax = __builtin_fmaxf(omax, __builtin_fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1)));
              stg[(q + 4 * h) * 64 + scol + tstep * t] = v0;
              stg[(q + 1 + 4 * h) * 64 + scol + tstep * t] = v1;
            }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int row = 32 * j + 8 * g + srow + 4 * hh;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (srow + 4 * hh) * 64 + 4 * sc4);
            if (row0 + row < M) sdy_st16s(og + (long)(32 * j + 8 * g + 4 * hh) * orow, olane, v);
          }
          __builtin_amdgcn_wave_barrier();
        }
      sdy_flag_range(p.flags, omax * DSX, p.head_out);   // (rows past the ragged edge are zero rows: they cannot raise it)
    }
    stamp(4);
    if (SDY_STAMPS_ON && p.stamps && blockIdx.x == 11 && tile_it < 63 && !more) {
      const unsigned long long tm = __builtin_amdgcn_s_memtime();
      if (lane == 0 && wave == 0) p.stamps[256 + tile_it + 1] = tm;
    }
    if (!more) break;
    cur = nxt;
    __syncthreads();   // every wave is done reading the LDS tile
    stamp(5);
  }
}

}  // namespace

#if SDY_STAMPS_ON
static unsigned long long* g_dstamps = nullptr;
SDY_DEBUG_EXPORT int sdy_dhconv_frag_debug_stamps(unsigned long long* host384) {   // 256 phase stamps + 64 tile starts + 64 (l, t)
  unsigned long long* host64 = host384;
  if (!g_dstamps || !host64) return SDY_ERR_STATE;
  SDY_HIP_TRY(hipDeviceSynchronize());
  SDY_HIP_TRY(hipMemcpy(host64, g_dstamps, 384 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SDY_OK;
}
#endif

extern "C" int sdy_dhconv_frag_supported(int Ci, int Co) { return (Ci == DE && Co == DE) ? 1 : 0; }

extern "C" size_t sdy_dhconv_frag_pack_bytes(int L) { return L > 0 ? (size_t)L * DLSTRIDE * sizeof(f16x8) : 0; }

// w_host: (256, 256, L, 2) reference layout (i, o, l, re | im).  The stream does not depend on the order of the 2C axis
// (the kernel maps its k-steps and output columns); `ilv` is accepted for the caller's convenience and ignored.
int sdy_dh_h3_pack(const float* w, int L, void* dev, float* scale, int ilv) {
  (void)ilv;
  if (!w || !dev || !scale || L <= 0) return SDY_ERR_ARG;
  float mx = 0.f;
  for (size_t i = 0; i < (size_t)DE * DE * L * 2; ++i) mx = std::fmax(mx, std::fabs(w[i]));
  float s = 1.0f;
  if (mx > 0.f && std::isfinite(mx)) {
    int e;
    std::frexp(mx, &e);
    s = std::ldexp(1.0f, 13 - e);
  }
  const size_t gh = (size_t)DGROUP * 8;   // halves per group
  std::vector<_Float16> buf((size_t)DWAVES * DGPW * gh);
  for (int l = 0; l < L; ++l) {
    _Float16* d = buf.data();
    for (int wv = 0; wv < DWAVES; ++wv)
      for (int cb = 0; cb < DCB; ++cb)
        for (int comp = 0; comp < 2; ++comp, d += gh)
          for (int ln = 0; ln < 64; ++ln)
            for (int e = 0; e < 8; ++e) {
              const int i = 16 * cb + 8 * (ln >> 5) + e, o = 32 * wv + (ln & 31);
              const float v = w[(((size_t)i * DE + o) * L + l) * 2 + comp] * s;
              const _Float16 hv = (_Float16)v;
              d[ln * 8 + e] = hv;
              d[64 * 8 + ln * 8 + e] = (_Float16)(v - (float)hv);
            }
    SDY_HIP_TRY(hipMemcpy(reinterpret_cast<char*>(dev) + (size_t)l * DLSTRIDE * sizeof(f16x8), buf.data(),
                          buf.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  }
  *scale = s;
  return SDY_OK;
}
extern "C" int sdy_dhconv_frag_pack(const float* w_host, int L, void* packed_dev, float* scale) {
  return sdy_dh_h3_pack(w_host, L, packed_dev, scale, 0);
}

int sdy_dh_h3_launch(const float* Cs_in, const void* packed, float scale, float* Cs_out, int L, int mtr, int B, int ilv,
                      hipStream_t stream, int tiled, int B_in) {
  if (!Cs_in || !packed || !Cs_out || L <= 0 || mtr <= 0 || B <= 0 || !(scale > 0.f)) return SDY_ERR_ARG;
  if (B_in <= 0) B_in = B;
  if (B_in < B || (B_in != B && (!tiled || (long)mtr * B >= 65536))) return SDY_ERR_UNSUPPORTED;
  if (tiled && (!ilv || (long)mtr * B_in * 8 * L * 64 >= (1L << 32))) return SDY_ERR_UNSUPPORTED;   // 32-bit element offsets
  DhParams p;
  p.X = Cs_in; p.sX = (long)mtr * B * DK;
  p.out = Cs_out; p.sC = (long)mtr * B * DN;
  p.w = reinterpret_cast<const f16x8*>(packed);
  p.L = L; p.mtr = mtr; p.B = B; p.ilv = ilv ? 1 : 0; p.tiled = tiled ? 1 : 0;
  p.B_in = B_in; p.b_magic = (unsigned)((1ull << 32) / (unsigned)B) + 1u;
  p.out_scale = 1.0f / (scale * DSX);
  SDY_TRY(sdy_flags_ptr(&p.flags));
  SDY_TRY(sdy_headroom_ptr(SDY_RANGE_DHCONV, &p.head_in));
  SDY_TRY(sdy_headroom_ptr(SDY_RANGE_LEG_SYNTHESIS, &p.head_out));
  p.stamps = nullptr;
  {   // degree -> XCD: degrees by falling tile count (ties: higher degree first), each to the XCD with the fewest tiles so far
    p.use_table = L <= 256 ? 1 : 0;
    if (p.use_table) {
      int order[256];
      long tiles[256], load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int l = 0; l < L; ++l) {
        order[l] = L - 1 - l;
        tiles[l] = ((long)(l + 1 < mtr ? l + 1 : mtr) * B + DTN - 1) / DTN;
      }
      std::stable_sort(order, order + L, [&](int a, int b) { return tiles[a] > tiles[b]; });
      for (int i = 0; i < L; ++i) {
        int best = 0;
        for (int x = 1; x < 8; ++x)
          if (load[x] < load[best]) best = x;
        p.xcd_of[order[i]] = (unsigned char)best;
        load[best] += tiles[order[i]];
      }
      for (int l = L; l < 256; ++l) p.xcd_of[l] = 0;
    }
  }
#if SDY_STAMPS_ON
  if (std::getenv("SDY_DH_STAMPS")) {
    if (!g_dstamps) {
      SDY_HIP_TRY(hipMalloc(&g_dstamps, 384 * sizeof(unsigned long long)));
      SDY_HIP_TRY(hipMemset(g_dstamps, 0, 384 * sizeof(unsigned long long)));
    }
    p.stamps = g_dstamps;
  }
#endif
  int n_cu = 0;
  SDY_TRY(sdy_cu_count(&n_cu));
  const int smem = 2 * DTN * DK * (int)sizeof(_Float16) + DWAVES * 8 * 64 * (int)sizeof(float);   // x tile + staging
  static SdyOncePerDevice once;
  std::atomic<bool>* attr_done = nullptr;
  SDY_TRY(once.slot(&attr_done));
  if (!*attr_done) {
    SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(dh_h3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    *attr_done = true;
  }
  const int grid = n_cu >= 8 ? (n_cu / 8) * 8 : 8;   // whole slots of 8 XCDs: one workgroup per CU
  hipLaunchKernelGGL(dh_h3_kernel, dim3(grid), dim3(512), smem, stream, p);
  return sdy_launch_status();
}
extern "C" int sdy_dhconv_frag(const float* Cs_in, const void* packed, float scale, float* Cs_out, int L, int mtr, int B,
                               void* stream) {
  return sdy_dh_h3_launch(Cs_in, packed, scale, Cs_out, L, mtr, B, 0, (hipStream_t)stream, 0, 0);
}
