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
//   * wave w computes columns 64w .. 64w+63: the weight never touches LDS, it is packed per (l, wave) as a linear stream of
//     MFMA B-fragment pairs (hi, lo) in consumption order and flows L2 -> registers through an 8-group ring;
//   * the accumulators (rows x columns, column = lane) are transposed per wave through a private 2 KB LDS area and stored
//     as 16 bytes per lane, 256-byte row segments.
// Work distribution: each degree l belongs to ONE XCD (boustrophedon over l, so the (l+1)-proportional work balances), and
// the 32 workgroups of an XCD walk its tile list interleaved -- at any time they sit on the same one or two degrees, whose
// 1 MB weight streams stay in that XCD's 4 MB L2 and are fetched from HBM once.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef const f16x8 __attribute__((address_space(1)))* wptr_t;   // see conv_h3.hip

namespace {

constexpr int DE = 256;            // channels in = out
constexpr int DK = 2 * DE;         // contraction length (ri, c)
constexpr int DN = 2 * DE;         // output columns
constexpr int DTN = 64;            // rows per tile
constexpr int DKS = DK / 16;       // k-steps (32)
constexpr int DWAVES = 8;
constexpr int DRING = 8;           // groups in flight (4 k-steps x 2 n-tiles)
constexpr int DGPW = 2 * DKS;      // groups per wave and degree (64)
constexpr int DGROUP = 2 * 64;     // f16x8 elements per group (hi | lo)
constexpr long DLSTRIDE = (long)DWAVES * DGPW * DGROUP;   // f16x8 elements per degree (1 MB)
constexpr float DSX = 16.0f;

struct DhParams {
  const float* X; long sX;         // Cs_in,  per-degree stride (floats)
  float* out; long sC;             // Cs_out, per-degree stride
  const f16x8* w;                  // [l][8 waves][DGPW groups][hi | lo][64 lanes]
  int L, mtr, B;
  float out_scale;
  unsigned long long* stamps;      // timing experiments only (SDY_DH_STAMPS)
};

__device__ __forceinline__ int dh_swz(int r) { return (r & 15) ^ (((r >> 4) & 1) * 3); }
// half offset of 16-byte chunk c (0..63) of row r
__device__ __forceinline__ int dh_off(int r, int c) { return r * DK + (((c & ~15) | ((c ^ dh_swz(r)) & 15)) << 3); }

// degree -> XCD (boustrophedon), rows and tiles of a degree
__device__ __forceinline__ int dh_xcd(int l) { return (l & 8) ? 7 - (l & 7) : (l & 7); }
__device__ __forceinline__ int dh_rows(const DhParams& p, int l) { return (l + 1 < p.mtr ? l + 1 : p.mtr) * p.B; }

struct TileIt {
  int l, t;   // degree (descending; -1 = done) and tile index inside it
};
__device__ __forceinline__ void dh_advance(const DhParams& p, TileIt& it, int xcd, int step) {
  it.t += step;
  while (it.l >= 0) {
    const int nt = (dh_xcd(it.l) == xcd) ? (dh_rows(p, it.l) + DTN - 1) / DTN : 0;
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
  auto w_base = [&](int l) { return (wptr_t)(p.w + (size_t)l * DLSTRIDE + (size_t)wave * DGPW * DGROUP + lane); };
  wptr_t wp = w_base(cur.l);
#pragma unroll
  for (int s = 0; s < DRING; ++s) {
    r_hi[s] = wp[s * DGROUP];
    r_lo[s] = wp[s * DGROUP + 64];
  }
  wp += DRING * DGROUP;

  // activation rows of a tile: clamped to row 0 of the degree beyond the ragged edge (zeroed when staged)
  f32x4 xr[8][2];
  auto x_ptr = [&](const TileIt& it, int i) {   // uniform degree base + a 32-bit lane offset
    int rr = r0;
    asm volatile("" : "+v"(rr));   // computed where it is used: 16 hoisted addresses would not fit the register budget
    const int row = it.t * DTN + rr + 8 * i;
    return p.X + (long)it.l * p.sX + (unsigned)((row < dh_rows(p, it.l) ? row : 0) * DK + 8 * oc);
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
      if (p.stamps && blockIdx.x == 11 && lane == 0 && tile_it >= 2 && tile_it < 6)
        p.stamps[((tile_it - 2) * 8 + wave) * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const int row0 = cur.t * DTN;
    TileIt nxt = cur;
    dh_advance(p, nxt, xcd, nslots);
    const bool more = nxt.l >= 0;
    const TileIt pre = more ? nxt : cur;   // past the end: a harmless re-read

    // ---- phase 0: the tile (already in registers) -> LDS, fp16 hi / lo, [row][k]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + 8 * i;
      const bool ok = row0 + r < M;
      f16x8 vh, vl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = ok ? xr[i][e >> 2][e & 3] * DSX : 0.0f;
        const _Float16 hv = (_Float16)v;
        vh[e] = hv;
        vl[e] = (_Float16)(v - (float)hv);
      }
      const int off = dh_off(r, oc);
      *reinterpret_cast<f16x8*>(Xs_hi + off) = vh;
      *reinterpret_cast<f16x8*>(Xs_lo + off) = vl;
    }
    stamp(1);
    __syncthreads();
    stamp(2);

    // ---- MFMA phase: all 64 rows x columns 64 wave .. +64, K = 512
    const wptr_t wnext = w_base(pre.l);
    f32x16 acc[2][2];   // [row tile j][column tile ni]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][ni][r] = 0.0f;
    // Order inside a k-step: the four (x lo) x (w hi) products first -- then the x-lo fragments of the NEXT k-step are
    // requested into the same registers -- then the eight products that read the x-hi fragments, whose successors are
    // requested last and arrive under the next k-step's first four MFMAs.  (A wave that loads its fragments at the top of
    // every k-step and waits keeps the matrix pipe about 60 % busy; double-buffered fragment registers do not fit.)
    f16x8 ah[2], al[2];
    auto load_lo = [&](int ks) {
#pragma unroll
      for (int j = 0; j < 2; ++j) al[j] = *reinterpret_cast<const f16x8*>(Xs_lo + dh_off(32 * j + l31, 2 * ks + h));
    };
    auto load_hi = [&](int ks) {
#pragma unroll
      for (int j = 0; j < 2; ++j) ah[j] = *reinterpret_cast<const f16x8*>(Xs_hi + dh_off(32 * j + l31, 2 * ks + h));
    };
    load_lo(0);
    load_hi(0);
#pragma unroll
    for (int kb = 0; kb < DKS / 4; ++kb) {
      if (kb == DKS / 4 - 1) {   // the refills of the last block fetch block 0 of the next tile's stream
        wp = wnext;
        asm volatile("" : "+v"(wp));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ks = 4 * kb + i, s0 = 2 * i, s1 = 2 * i + 1;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[j][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], r_hi[s0 + ni], acc[j][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifndef DH_NOLDS
        if (ks + 1 < DKS) load_lo(ks + 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[j][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], r_lo[s0 + ni], acc[j][ni], 0, 0, 0);
#ifndef DH_NOW
        r_lo[s0] = wp[s0 * DGROUP + 64];
        r_lo[s1] = wp[s1 * DGROUP + 64];
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[j][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], r_hi[s0 + ni], acc[j][ni], 0, 0, 0);
#ifndef DH_NOW
        r_hi[s0] = wp[s0 * DGROUP];
        r_hi[s1] = wp[s1 * DGROUP];
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifndef DH_NOLDS
        if (ks + 1 < DKS) load_hi(ks + 1);
#endif
#ifndef DH_NOPREFETCH
        if ((i & 1) == 0) {   // one 16-byte piece of the next tile per two k-steps, never a burst
          const int g = 2 * kb + (i >> 1);   // 0..15
          xr[g >> 1][g & 1] = *reinterpret_cast<const f32x4*>(x_ptr(pre, g >> 1) + 4 * (g & 1));
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#ifndef DH_WSAME
      wp += DRING * DGROUP;
#endif
    }

    stamp(3);
    // ---- epilogue: the accumulators hold column = lane, i.e. a direct store is 4 bytes per lane (measured: the 64 dword
    // stores per lane took 40 % of the tile).  Each wave transposes 8-row chunks of its 64 x 64 block through a private
    // 2 KB LDS staging area (in-order LDS of one wave: no barrier) and stores 16 bytes per lane, 256-byte row segments.
    {
      float* stg = reinterpret_cast<float*>(smem + 2 * DTN * DK * sizeof(_Float16)) + wave * (8 * 64);
      const int srow = lane >> 4, sc4 = lane & 15;
      float* og = p.out + (long)cur.l * p.sC + (long)row0 * DN + 64 * wave + 4 * sc4;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) stg[(q + 4 * h) * 64 + 32 * ni + l31] = acc[j][ni][4 * g + q] * p.out_scale;
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int row = 32 * j + 8 * g + srow + 4 * hh;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (srow + 4 * hh) * 64 + 4 * sc4);
            if (row0 + row < M) SDY_STREAM_STORE(og + (long)row * DN, v);
          }
          __builtin_amdgcn_wave_barrier();
        }
    }
    stamp(4);
    if (!more) break;
    cur = nxt;
    __syncthreads();   // every wave is done reading the LDS tile
    stamp(5);
  }
}

}  // namespace

static unsigned long long* g_dstamps = nullptr;
extern "C" int sdy_dhconv_frag_debug_stamps(unsigned long long* host256) {
  unsigned long long* host64 = host256;
  if (!g_dstamps || !host64) return SDY_ERR_STATE;
  SDY_HIP_TRY(hipDeviceSynchronize());
  SDY_HIP_TRY(hipMemcpy(host64, g_dstamps, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SDY_OK;
}

extern "C" int sdy_dhconv_frag_supported(int Ci, int Co) { return (Ci == DE && Co == DE) ? 1 : 0; }

extern "C" size_t sdy_dhconv_frag_pack_bytes(int L) { return L > 0 ? (size_t)L * DLSTRIDE * sizeof(f16x8) : 0; }

// w_host: (256, 256, L, 2) reference layout.  ilv: column / row order of the 2C axis, 0 = [ri][c], 1 = [c/16][ri][16]
int sdy_dh_h3_pack(const float* w, int L, void* dev, float* scale, int ilv) {
  if (!w || !dev || !scale || L <= 0) return SDY_ERR_ARG;
  auto value = [&](int l, int op, int ip) {   // W'_l[ip][op]
    const int ro = ilv ? (op >> 4) & 1 : op >= DE, o = ilv ? (op >> 5) * 16 + (op & 15) : op - ro * DE;
    const int ri = ilv ? (ip >> 4) & 1 : ip >= DE, i = ilv ? (ip >> 5) * 16 + (ip & 15) : ip - ri * DE;
    const float* e = w + (((size_t)i * DE + o) * L + l) * 2;
    if (ri == ro) return e[0];
    return ri ? -e[1] : e[1];
  };
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
      for (int ks = 0; ks < DKS; ++ks)
        for (int ni = 0; ni < 2; ++ni, d += gh)
          for (int ln = 0; ln < 64; ++ln)
            for (int e = 0; e < 8; ++e) {
              const float v = value(l, 64 * wv + 32 * ni + (ln & 31), 16 * ks + 8 * (ln >> 5) + e) * s;
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

extern "C" int sdy_dhconv_frag(const float* Cs_in, const void* packed, float scale, float* Cs_out, int L, int mtr, int B,
                               void* stream) {
  if (!Cs_in || !packed || !Cs_out || L <= 0 || mtr <= 0 || B <= 0 || !(scale > 0.f)) return SDY_ERR_ARG;
  DhParams p;
  p.X = Cs_in; p.sX = (long)mtr * B * DK;
  p.out = Cs_out; p.sC = (long)mtr * B * DN;
  p.w = reinterpret_cast<const f16x8*>(packed);
  p.L = L; p.mtr = mtr; p.B = B;
  p.out_scale = 1.0f / (scale * DSX);
  p.stamps = nullptr;
  if (std::getenv("SDY_DH_STAMPS")) {
    if (!g_dstamps) SDY_HIP_TRY(hipMalloc(&g_dstamps, 256 * sizeof(unsigned long long)));
    p.stamps = g_dstamps;
  }
  static int n_cu = 0;
  const int smem = 2 * DTN * DK * (int)sizeof(_Float16) + DWAVES * 8 * 64 * (int)sizeof(float);   // x tile + staging
  if (!n_cu) {
    int dev = 0;
    SDY_HIP_TRY(hipGetDevice(&dev));
    SDY_HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    SDY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(dh_h3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  }
  const int grid = n_cu >= 8 ? (n_cu / 8) * 8 : 8;   // whole slots of 8 XCDs: one workgroup per CU
  hipLaunchKernelGGL(dh_h3_kernel, dim3(grid), dim3(512), smem, (hipStream_t)stream, p);
  return sdy_launch_status();
}
