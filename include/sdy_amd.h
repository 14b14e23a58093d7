/*
 * sdy_amd.h -- C ABI of the MI355X-native Spherical-DYffusion sampling path.
 *
 * The reference (Rose-STL-Lab/spherical-dyffusion) is 100 % Python and has no FFI boundary for this path;
 * the objects this library replaces are Python callables.  Each entry point below cites the reference
 * interface it stands in for (file:line under the reference root).  INTEGRATION.md shows the ctypes stub a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer marked "dev" is a device (HBM) pointer; "host" pointers are ordinary host memory
 *   - all tensors are contiguous fp32; grid-space activations are NCHW = (B, C, nlat, nlon)
 *   - every function returns SDY_OK (0), a negative SDY_ERR_* for a bad argument, or a positive hipError_t
 *   - nothing here synchronises the device, allocates in a launch path, or throws; work is enqueued on the
 *     given stream (a hipStream_t passed as void*), so calls are graph-capturable
 *   - workspaces are supplied by the caller (PyTorch caching allocator on the Python side)
 *
 * Internal spectral layouts (fp32):
 *   grid-frequency  Xf[m][k][b][ri][c]   m < mtr = min(mmax, lmax), k < nlat, ri in {re, im}
 *   coefficients    Cs[l][m][b][ri][c]   l < lmax, m < mtr; entries with m > l are never read or written
 * (These are the layouts of the stage-level entry points below.  sdy_sfno_forward keeps its own spectral workspace in a
 *  private variant: the 2C axis ordered [c/16][ri][16], and (order, latitude) pairs whose Legendre table entries are below
 *  1e-12 of the order's maximum omitted altogether -- DESIGN.md section 3.)
 *
 * Dropout stream (the reference uses torch's global generator: src/models/sfno/layers.py:76-78,
 * src/models/modules/drop_path.py:19 -- not reproducible across devices, so the product defines its own):
 *   Philox4x32-7 (the Random123 generator with seven rounds, the smallest count that passes BigCrush; round constants as in
 *   Random123), key = (seed_lo, seed_hi), counter = (c0, c1, stream, call)
 *     element dropout : n = pixel (h*nlon + w); c0 = n with bit 5 cleared, c1 = b_global*(C/4) + (ch>>2), word = ch & 3,
 *                       half-word = bit 5 of n (0: low 16 bits, 1: high 16 bits),
 *                       stream = 2*layer + kind (kind 0 = MLP hidden, 1 = MLP output);
 *                       keep <=> half-word >= floor(p * 2^16)   (one call serves 4 channels x the pixel pair n, n + 32)
 *     drop path       : c0 = b_global, c1 = 0xFFFFFFFF, stream = 0x1000 + layer, word 0;
 *                       keep <=> word >= floor(p * 2^32)
 *   kept values are scaled by 1/(1-p).
 */
#ifndef SDY_AMD_H
#define SDY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported. */
#pragma GCC visibility push(default)

#define SDY_OK 0
#define SDY_ERR_ARG (-1)         /* null pointer / non-positive extent */
#define SDY_ERR_UNSUPPORTED (-2) /* size the kernels do not cover (e.g. nlon with a prime factor > 5) */
#define SDY_ERR_ALIGN (-3)       /* extent along a contiguous dimension not a multiple of 4 */
#define SDY_ERR_WORKSPACE (-4)   /* workspace too small */
#define SDY_ERR_NAME (-5)        /* unknown parameter name */
#define SDY_ERR_SHAPE (-6)       /* parameter has the wrong number of elements */
#define SDY_ERR_STATE (-7)       /* object not fully initialised (missing parameters) */

#define SDY_GRID_EQUIANGULAR 0
#define SDY_GRID_LEGENDRE_GAUSS 1

int sdy_version(void);
const char* sdy_error_string(int code);
/* The argument structures below carry no size field: a caller built against another revision of this header would hand the
 * launchers uninitialised tail bytes (fields are only ever appended).  Contract: zero-initialise every structure with the
 * sizeof of THIS header, and call sdy_abi_check once after loading the library with your own sizeofs, in the order
 * {sdy_conv_args, sdy_mlp_args, sdy_pair_args, sdy_sfno_config, sdy_sfno_fwd_args, sdy_var_table, sdy_step_finish_args}:
 * SDY_OK when all seven equal the library's, SDY_ERR_ARG otherwise (the Python bindings do this at import). */
#define SDY_ABI_STRUCTS 7
int sdy_abi_check(const size_t* sizes, int n);

/* ---------------------------------------------------------------------------------------------------------
 * Spherical-harmonic transform plan.
 * Replaces torch_harmonics.RealSHT / InverseRealSHT construction (third-party, un-vendored; reference call
 * sites src/models/sfno/sfnonet.py:551-554; attributes read at src/models/sfno/s2convolutions.py:73-83).
 * One plan holds the forward (quadrature-weighted) and inverse Legendre tables of one grid, in fp32 on the
 * device (computed in fp64 on the host, cast like `.float()` at sfnonet.py:551-554), plus FFT twiddles. */
typedef struct sdy_sht_plan sdy_sht_plan;

/* Host-only: fp64 tables exactly as torch_harmonics builds them.  pct: [mmax][lmax][nlat], w: [nlat]
 * quadrature weights, theta: [nlat] colatitudes (any may be NULL).  No GPU needed. */
int sdy_sht_tables_host(int nlat, int nlon, int lmax, int mmax, int grid, double* pct, double* w, double* theta);

int sdy_sht_plan_create(int nlat, int nlon, int lmax, int mmax, int grid, sdy_sht_plan** out);
/* gemm_mode: 0 = Legendre GEMMs on fp32 MFMA; 1 = split-fp16 3-pass MFMA (fp32-class accuracy, see sdy_sfno_config).
 * sdy_sht_plan_create uses $SDY_GEMM_MODE ("f32" -> 0, otherwise 1). */
int sdy_sht_plan_create_ex(int nlat, int nlon, int lmax, int mmax, int grid, int gemm_mode, sdy_sht_plan** out);
void sdy_sht_plan_destroy(sdy_sht_plan* plan);
/* dims[0..5] = nlat, nlon, lmax, mmax, mtr, grid */
int sdy_sht_plan_dims(const sdy_sht_plan* plan, int dims[6]);
/* floats needed by sdy_sht_forward / sdy_sht_inverse for B*C fields */
size_t sdy_sht_workspace_floats(const sdy_sht_plan* plan, int B, int C);

/* RealSHT.forward (torch_harmonics; called at src/models/sfno/s2convolutions.py:165):
 * x dev (B,C,nlat,nlon) f32 -> out dev (B,C,lmax,mmax) complex64 (re,im interleaved). C % 4 == 0. */
int sdy_sht_forward(const sdy_sht_plan* plan, const float* x, float* out_c64, int B, int C, float* ws,
                    size_t ws_floats, void* stream);
/* InverseRealSHT.forward (called at src/models/sfno/s2convolutions.py:168,186):
 * in dev (B,C,lmax,mmax) complex64 -> y dev (B,C,nlat,nlon) f32. */
int sdy_sht_inverse(const sdy_sht_plan* plan, const float* in_c64, float* y, int B, int C, float* ws,
                    size_t ws_floats, void* stream);

/* Stage-level entry points on the internal layouts (what the fused network path launches). */
/* longitude real FFT x (2*pi/nlon) fused with the per-(b,c) affine a*x+d of InstanceNorm + time scale/shift
 * (src/models/sfno/sfnonet.py:292,298-299).  a, d: dev [B*C] or NULL.  xn_out: dev (B,C,nlat,nlon) or NULL. */
int sdy_rfft_lon(const sdy_sht_plan* plan, const float* x, const float* a, const float* d, float* xn_out,
                 float* Xf, int B, int C, void* stream);
/* Legendre analysis: Cs[l][m][n] = sum_k Wq[m][l][k] Xf[m][k][n]  (einsum '...km,mlk->...lm') */
int sdy_legendre_fwd(const sdy_sht_plan* plan, const float* Xf, float* Cs, int B, int C, void* stream);
/* Legendre synthesis: Yf[m][k][n] = sum_l P[m][l][k] Cs[l][m][n]  (einsum '...lm,mlk->...km') */
int sdy_legendre_inv(const sdy_sht_plan* plan, const float* Cs, float* Yf, int B, int C, void* stream);
/* inverse longitude FFT (irfft n=nlon, norm="forward") + optional per-channel bias (s2convolutions.py:188-189) */
int sdy_irfft_lon(const sdy_sht_plan* plan, const float* Yf, const float* bias, float* y, int B, int C,
                  void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * _contract_dhconv (src/models/sfno/contractions.py:159-169 via factorizations.py:165-186; called at
 * src/models/sfno/s2convolutions.py:173-178):  out[b,o,l,m] = sum_i x[b,i,l,m] * w[i,o,l]  (complex).
 * sdy_dhconv_pack_weight: host (Ci,Co,L,2) reference layout -> dev packed [l][2][Ci][Co].
 * sdy_dhconv: Cs_in / Cs_out in the coefficient layout above (channel counts Ci / Co, both % 4 == 0). */
int sdy_dhconv_pack_weight(const float* w_host, int Ci, int Co, int L, float* w_packed_dev, void* stream);
int sdy_dhconv(const float* Cs_in, const float* w_packed, float* Cs_out, int L, int mtr, int B, int Ci, int Co,
               void* stream);
/* Same contraction on the f16 matrix cores in split precision (3 passes, fp32-class accuracy): the weight is expanded to
 * its real 2Ci x 2Co form, transposed and split into fp16 hi | lo on the host. */
size_t sdy_dhconv_h3_pack_bytes(int Ci, int Co, int L);
int sdy_dhconv_h3_pack_weight(const float* w_host, int Ci, int Co, int L, void* packed_dev, float* scale);
int sdy_dhconv_h3(const float* Cs_in, const void* packed, float scale, float* Cs_out, int L, int mtr, int B, int Ci,
                  int Co, void* stream);

/* Same contraction for Ci = Co = 256 as a persistent fragment-stream kernel (dh_h3.hip): a workgroup owns 64 rows and
 * all 512 output columns, so every coefficient row is read once; the packed weight (1 MB per degree, MFMA fragment
 * order) streams L2 -> registers and each degree is served by one XCD.  Same reference lines as sdy_dhconv
 * (src/models/sfno/contractions.py:159-169).  `scale` is what the pack returned. */
int sdy_dhconv_frag_supported(int Ci, int Co);
size_t sdy_dhconv_frag_pack_bytes(int L);
int sdy_dhconv_frag_pack(const float* w_host, int L, void* packed_dev, float* scale);
int sdy_dhconv_frag(const float* Cs_in, const void* packed, float scale, float* Cs_out, int L, int mtr, int B,
                    void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * nn.InstanceNorm2d(C, eps, affine=True, track_running_stats=False) statistics folded with the block's time
 * scale/shift (src/models/sfno/sfnonet.py:280-299,641-648) into per-(b,c) coefficients:
 *   xn = a*x + d,  a = gamma*rstd*(1+scale),  d = (beta - mean*gamma*rstd)*(1+scale) + shift
 * scale_shift: dev, element (b, j) at scale_shift[b*ss_stride + j], j < 2C (scale | shift), or NULL. */
int sdy_instnorm_coeffs(const float* x, int B, int C, int HW, const float* gamma, const float* beta,
                        const float* scale_shift, long ss_stride, float eps, float* a, float* d, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * 1x1 convolution = per-pixel GEMM with fused prologue/epilogue.  Replaces nn.Conv2d(k=1) + bias + GELU +
 * Dropout + DropPath + residual adds of src/models/sfno/sfnonet.py:303-335,609-618,734-744 and
 * src/models/sfno/layers.py:73-80.
 *   out[b,o,p] = post( act( sum_i wt[i,o] * (pa[b,i]*x[b,i,p] + pd[b,i]) + bias[o] + add_pre[b,o,p] ) )
 *   post(v)    = batch_scale[b] * dropout(v) + add_post[b,o,p]
 * wt is the TRANSPOSED weight, dev [Cin][ldw] (ldw >= Cout, ldw % 4 == 0, columns >= Cout zero). */
typedef struct sdy_conv_args {
  const float* x;  long x_bstride;   /* dev (B, >=Cin, HW) ; batch stride in floats */
  const float* wt; int ldw;          /* dev [Cin][ldw] */
  float* out;      long out_bstride; /* dev (B, >=Cout, HW) */
  int B, Cin, Cout, HW;
  const float* pa; const float* pd;  /* dev [B*Cin] each or NULL (prologue affine) */
  const float* bias;                 /* dev [Cout] or NULL */
  const float* add; long add_bstride;/* dev (B or 1, Cout, HW) or NULL; add_bstride 0 broadcasts over b */
  int add_mode;                      /* 0 none, 1 before activation, 2 after dropout/batch_scale */
  int act;                           /* 0 none, 1 exact-erf GELU */
  float drop_p;                      /* 0 = no dropout */
  const float* keep_mask;            /* dev (B,Cout,HW) 0/1 injected mask (tests) or NULL = Philox stream */
  uint64_t seed; uint32_t call; uint32_t stream_id; uint32_t batch_offset;
  int rows_per_call;                   /* 0: every batch row belongs to `call`.  n > 0: the batch stacks several calls of n
                                          rows each: row b draws the stream of (call + b / n, trajectory batch_offset + b % n)
                                          -- bit-identical to issuing those calls one after the other (the two interpolator
                                          calls of a DYffusion sampling step share their inputs: dyffusion.py:497,515) */
  const float* batch_scale;          /* dev [B] or NULL (drop-path scale) */
  int kernel_tag;                    /* 0 generic; 1 = MLP fc1, 2 = MLP fc2, 3 = inner skip: identical code under a
                                        distinct symbol name so profilers attribute time per use */
  const void* w_h3;                  /* dev, optional: weight packed by sdy_h3_pack_weight -> the GEMM runs on the f16
                                        matrix cores in split precision (3 passes, fp32-class accuracy); wt may be NULL */
  float w_h3_scale;                  /* scale returned by sdy_h3_pack_weight */
  const void* w_frag;                /* dev, optional, Cout == 256, Cin <= 384 (a pre-affine pa/pd needs Cin == 256): weight packed by sdy_conv256_h3_pack[_cin] -> the
                                        persistent fragment-stream kernel (no dropout / batch_scale / keep_mask) */
  float w_frag_scale;
  double* stats;                     /* dev [B*Cout*2] or NULL, w_frag path only: (sum, sum of squares) over HW of every
                                        output plane are ADDED here (zero it first; sdy_instnorm_from_stats turns them into
                                        the next InstanceNorm's coefficients) */
  int out_tiled;                     /* w_frag path only: write `out` TILE-MAJOR, [b][tile of 64 pixels][Cout][64] (an image's last
                                        tile padded to 64; out_bstride >= ceil(HW / 64) * Cout * 64), the layout sdy_mlp_args.x_tiled
                                        reads: a tile of the intermediate tensor between the two persistent kernels is then one
                                        contiguous 64 KB block for its producer and its consumer.  `out` must not alias `add`. */
  const unsigned char* x_rows;       /* host [B] or NULL, w_frag path only, B <= 128: image z of this launch reads batch row x_rows[z] of
                                        x / pa / pd (add, out and stats stay indexed by z).  sdy_sfno_forward runs a block whose
                                        DropPath draw (src/models/modules/drop_path.py:15-22) zeroes some trajectories' branch on
                                        the active trajectories only: its per-block tensors are compact, the block input is not. */
} sdy_conv_args;
int sdy_conv1x1(const sdy_conv_args* args, void* stream);

/* Split-precision weight packing for sdy_conv1x1 (w_h3): host (Cout, Cin) row-major fp32 -> dev fp16 hi|lo planes,
 * [Mpad][Kpad] each (Mpad = 128 for Cout <= 128, else Cout rounded up to 256; Kpad = Cin rounded up to 64), multiplied by a power of two
 * (*scale) chosen so that max|w|*scale is in [2^12, 2^13). */
/* 256 -> 256 weight (Cout, Cin) row-major -> per-wave MFMA fragment stream for sdy_conv_args.w_frag */
int sdy_conv256_h3_supported(int Cin, int Cout);
size_t sdy_conv256_h3_pack_bytes(void);                                                    /* for Cin <= 256 */
size_t sdy_conv256_h3_pack_bytes_cin(int Cin);                                             /* for any supported Cin */
int sdy_conv256_h3_pack(const float* w_host, void* packed_dev, float* scale);             /* (256, 256) weight */
int sdy_conv256_h3_pack_cin(const float* w_host, int Cin, void* packed_dev, float* scale); /* (256, Cin), Cin <= 384 */
size_t sdy_h3_pack_bytes(int Cout, int Cin);
int sdy_h3_pack_weight(const float* w_host, int Cout, int Cin, void* packed_dev, float* scale);

/* Fused MLP of one SFNO block (src/models/sfno/layers.py:73-80 as called from src/models/sfno/sfnonet.py:313-335):
 *   out[b] = batch_scale[b] * dropout2( W2 . dropout1( GELU( W1 . (pa[b]*x[b] + pd[b]) + b1 ) ) + b2 )
 *            + (add_a[b]*add[b] + add_d[b])
 * in ONE launch; the hidden activation stays on the compute unit (never written to HBM).  Split-fp16 arithmetic and
 * Philox stream identical to two sdy_conv1x1 calls with (stream_fc1, stream_fc2).  Supported shape: E = 256,
 * hidden = 512 (sdy_mlp_h3_supported); anything else returns SDY_ERR_UNSUPPORTED and the caller uses sdy_conv1x1. */
typedef struct sdy_mlp_args {
  const float* x;  long x_bstride;     /* dev (B, E, HW) */
  const float* pa; const float* pd;    /* dev [B*E] each or NULL (norm affine folded into the load) */
  const void* w; float w1_scale; float w2_scale;     /* sdy_mlp_h3_pack output and the two scales it returned */
  const float* b1; const float* b2;                  /* dev [hidden], dev [E] */
  float* out;      long out_bstride;   /* dev (B, E, HW) */
  const float* add; long add_bstride;  /* dev (B, E, HW) residual or NULL */
  const float* add_a; const float* add_d; /* dev [B*E] each or NULL: the residual is add_a*add + add_d (a norm folded
                                           into its consumer instead of being materialised) */
  int B, E, hidden, HW;
  float drop_p;                        /* 0 = no dropout */
  uint64_t seed; uint32_t call; uint32_t stream_fc1; uint32_t stream_fc2; uint32_t batch_offset;
  int rows_per_call;                   /* as in sdy_conv_args */
  const float* batch_scale;            /* dev [B] or NULL (drop-path scale) */
  double* stats;                       /* dev [B*E*2] or NULL: (sum, sum of squares) over HW of every output plane are
                                          ADDED here (InstanceNorm statistics of the next block, sfnonet.py:292): zero
                                          it before the launch, turn it into coefficients with sdy_instnorm_from_stats */
  int x_tiled;                         /* x is TILE-MAJOR (sdy_conv_args.out_tiled; x_bstride = floats per image); needs `add` */
  const float* keep_hidden;            /* tests only, dev (B, hidden, HW) and (B, E, HW) 0/1 masks or NULL: with drop_p > 0 the */
  const float* keep_out;               /* keep decisions come from these (e.g. masks the reference's nn.Dropout drew) instead of
                                          the Philox stream -- same kernel code, a separate (untimed) instantiation */
  const unsigned char* out_rows;       /* host [B] or NULL, B <= 128: image z of this launch (x, pa, pd indexed by z) IS batch row
                                          out_rows[z] -- add, add_a, add_d, out, stats, batch_scale, keep_* and the dropout stream
                                          (call, trajectory) are taken at that row (see sdy_conv_args.x_rows) */
  int add_by_launch_row;               /* with out_rows and without add_a / add_d: `add` is indexed by the launch's image z, like x
                                          (a residual that was materialised in the launch's own row order) */
} sdy_mlp_args;
/* (sum, sumsq) statistics -> the same per-(b,c) affine coefficients as sdy_instnorm_coeffs; clears `stats` for reuse. */
int sdy_instnorm_from_stats(double* stats, int B, int C, int HW, const float* gamma, const float* beta,
                            const float* scale_shift, long ss_bstride, float eps, float* a_out, float* d_out, void* stream);
int sdy_mlp_h3_supported(int E, int hidden);
size_t sdy_mlp_h3_pack_bytes(int E, int hidden);
/* w1_host: (hidden, E) row-major = mlp.fwd.0.weight;  w2_host: (E, hidden) row-major = mlp.fwd.{2|3}.weight.
 * Both are split to fp16 hi|lo (times a power of two each, returned) and interleaved into one per-wave stream in the
 * order the kernel consumes them. */
int sdy_mlp_h3_pack(const float* w1_host, const float* w2_host, int E, int hidden, void* packed_dev, float* scale1,
                    float* scale2);
int sdy_mlp_h3(const sdy_mlp_args* args, void* stream);

/* Two 1x1 convolutions with a GELU between them in ONE launch -- the encoder (src/models/sfno/sfnonet.py:609-618, with the
 * position embedding of :824 as the addend) and the decoder (:734-744 on [block output | inputs], :831-837):
 *   out[b] = W2 . GELU( W1 . x[b] + b1 ) + add[b or broadcast]
 * The 256-channel hidden activation stays on the compute unit.  Split-fp16 arithmetic of sdy_conv1x1 (w_h3), except that the
 * activations are not staged with the fixed pre-scale: x by a power of two per 64-pixel tile (and channel part) from the tile's
 * own maximum, the hidden activation by one from the bound max_row ||W1 row||_1 * max|x| + max|b1| (the L1 norm is computed by
 * sdy_pair_h3_pack and travels in the last 64 bytes of the packed buffer).  Inputs of any finite magnitude keep 22 bits
 * relative to their tile's maximum; SDY_FLAG_F16_RANGE is never raised here, SDY_FLAG_NONFINITE is for inf / NaN inputs.  Supported
 * shapes (sdy_pair_h3_supported): hidden == 256 and either Cout == 256 with Cin <= 144, or Cout <= 64 with Cin <= 416;
 * anything else returns SDY_ERR_UNSUPPORTED and the caller issues two sdy_conv1x1 calls. */
typedef struct sdy_pair_args {
  const float* x;  long x_bstride;     /* dev (B, >=Cin, HW) */
  const void* w; float w1_scale; float w2_scale;     /* sdy_pair_h3_pack output and the two scales it returned */
  const float* b1;                     /* dev [hidden] or NULL */
  float* out;      long out_bstride;   /* dev (B, >=Cout, HW) */
  const float* add; long add_bstride;  /* dev (B or 1, Cout, HW) or NULL; add_bstride 0 broadcasts over b */
  int B, Cin, hidden, Cout, HW;
  double* stats;                       /* dev [B*Cout*2] or NULL (Cout == 256 only): (sum, sum of squares) over HW of every
                                          output plane are ADDED here, as in sdy_mlp_args */
} sdy_pair_args;
int sdy_pair_h3_supported(int Cin, int hidden, int Cout);
size_t sdy_pair_h3_pack_bytes(int Cin, int hidden, int Cout);
/* w1_host: (hidden, Cin) row-major;  w2_host: (Cout, hidden) row-major */
int sdy_pair_h3_pack(const float* w1_host, const float* w2_host, int Cin, int hidden, int Cout, void* packed_dev,
                     float* scale1, float* scale2);
int sdy_pair_h3(const sdy_pair_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Whole network.  Replaces SphericalFourierNeuralOperatorNet.__init__/forward
 * (src/models/sfno/sfnonet.py:426-841) + BaseModel.concat_condition_if_needed (src/models/_base_model.py:166-192)
 * for the configuration the shipped YAML selects (filter_type=linear, operator_type=dhconv, factorization=None,
 * instance_norm, use_mlp, pos_embed, big_skip, scale_factor=1, encoder_layers=1). */
typedef struct sdy_sfno_config {
  int nlat, nlon;
  int in_chans;        /* inputs + conditional channels, as the encoder sees them */
  int out_chans;
  int embed_dim;
  int num_layers;
  int mlp_hidden;      /* int(embed_dim * mlp_ratio) */
  int lmax, mmax;      /* modes_lat, modes_lon (sfnonet.py:526-527) */
  int data_grid;       /* SDY_GRID_* of the first forward / last inverse transform */
  int with_time_emb;
  int time_dim;        /* embed_dim * time_dim_mult */
  float dropout_mlp;   /* MLP dropout rate (active only when a forward call enables dropout) */
  float drop_path_rate;
  int big_skip, pos_embed;
  int gemm_mode;       /* 0: fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere; 1: the 1x1 convolutions run as split-fp16
                          3-pass MFMA GEMMs (fp32-class accuracy at the f16 matrix rate) */
} sdy_sfno_config;

typedef struct sdy_sfno sdy_sfno;
int sdy_sfno_create(const sdy_sfno_config* cfg, sdy_sfno** out);
void sdy_sfno_destroy(sdy_sfno* net);
/* Load one tensor by its reference state_dict name, host fp32, `numel` elements.
 * The library re-lays it out for the kernels.  Names of non-persistent SHT buffers are accepted and ignored. */
int sdy_sfno_set_param(sdy_sfno* net, const char* name, const float* host, size_t numel);
/* 0 when every required parameter has been set; otherwise SDY_ERR_STATE (missing name via sdy_sfno_missing). */
int sdy_sfno_ready(const sdy_sfno* net);
const char* sdy_sfno_missing(const sdy_sfno* net);
size_t sdy_sfno_workspace_floats(const sdy_sfno* net, int B);
/* Largest B one sdy_sfno_forward call takes: 128 on the default kernel path (the drop-path row maps; the 32-bit offsets inside
 * the spectral workspace would allow 258 rows at 180 x 360, embed 256), 60 there with row-major coefficient tensors; larger
 * batches run as consecutive calls on row ranges, each with its own batch_offset (the Python module does). */
int sdy_sfno_max_batch(const sdy_sfno* net);

typedef struct sdy_sfno_fwd_args {
  /* up to three channel groups concatenated on dim 1 (inputs | condition | static_condition) */
  const float* in[3]; int in_chans[3];   /* dev (B, in_chans[i], nlat, nlon); unused slots NULL / 0 */
  const float* time;                     /* dev [B] or NULL when !with_time_emb */
  float* out;                            /* dev (B, out_chans, nlat, nlon) */
  int B;
  int enable_dropout;                    /* inference_dropout_scope (src/models/_base_model.py:273-286) */
  uint64_t seed; uint32_t call; uint32_t batch_offset;
  int rows_per_call;       /* 0, or n: the B rows are B / n stacked calls (call, call + 1, ...) of n trajectories each, see
                              sdy_conv_args; B must be a multiple of n */
  const float* const* keep_masks;        /* optional injected masks (tests): [num_layers*2] dev pointers
                                            (hidden, out) per layer, or NULL */
  const float* drop_path_keep;           /* optional injected drop-path keep flags, dev [num_layers][B], or NULL */
  float* ws; size_t ws_floats;
  int reuse_encoder;       /* 1: every `in` tensor holds the same values as in the PREVIOUS forward of this network on this
                              workspace with the same B (the two interpolations of a cold-sampling step share their inputs,
                              src/diffusion/dyffusion.py:497,515): the input concat and the encoder are skipped and the forward
                              restarts from the stored encoder output -- bit-identical results (time, dropout call number and
                              masks may differ: they enter after the encoder).  SDY_ERR_STATE if there is no such forward. */
  int shared_inputs;       /* 1 (needs rows_per_call = n < B): the stacked calls share their inputs -- every `in` tensor holds n
                              rows, and row b of the forward reads input row b % n (the two interpolations of a cold-sampling
                              step as ONE forward of 2n rows: same (x_0, forecast) and static condition, other time and dropout
                              call).  The encoder then runs on n rows; results are bit-identical to stacking copies. */
} sdy_sfno_fwd_args;
int sdy_sfno_forward(sdy_sfno* net, const sdy_sfno_fwd_args* args, void* stream);

/* Debug/parity taps: time embedding + per-block (scale|shift) as the network computes them.
 * t_repr dev [B*time_dim] (or NULL), ss dev [B*num_layers*2*embed_dim] (or NULL). */
int sdy_sfno_time_embed(sdy_sfno* net, const float* time, int B, float* t_repr, float* ss, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Sampler arithmetic of BaseDYffusion.sample_loop (src/diffusion/dyffusion.py:517-519, :655-661). */
/* out = x_s + (x_ip_next - x_ip_s); any of the three may alias out.  x_ip_s NULL means "x_ip_s == x_s" (s = 0). */
int sdy_cold_update(const float* x_s, const float* x_ip_next, const float* x_ip_s, float* out, size_t n,
                    void* stream);
/* channel concat of up to 4 NCHW tensors (torch.cat(dim=1)) */
int sdy_concat_channels(const float* const* src, const int* chans, int nsrc, float* out, int B, int HW,
                        void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Stepper glue either side of the sampler ("next" row of the scope table): the device arithmetic of
 * run_on_batch_multistep (src/ace_inference/core/stepper_multistep.py:298-466): StandardNormalizer
 * (src/ace_inference/core/normalizer.py:96-110), Packer (src/utilities/packer.py:70-77), Prescriber
 * (src/ace_inference/core/prescriber.py:68-92), relative LpLoss (src/ace_inference/training/utils/darcy_loss.py:214-228).
 * Variables are separate dev tensors (B, T1, nlat*nlon) as in the reference's data dict. */
#define SDY_MAX_VARS 96
typedef struct sdy_var_table {
  int nvars;
  const float* data[SDY_MAX_VARS];  /* dev (B, T1, HW), denormalised */
  float mean[SDY_MAX_VARS];         /* 0 / 1 for variables without statistics (passed through) */
  float std[SDY_MAX_VARS];
} sdy_var_table;
/* normalise + pack: out[b][v][p] = (data_v[b][t][p] - mean_v) / std_v,  out dev (B, nvars, HW) */
int sdy_norm_pack(const sdy_var_table* vars, int t, int T1, int B, int HW, float* out, void* stream);

typedef struct sdy_step_finish_args {
  int B, HW, T1, t;                   /* t = time index being written (1..n_forward_steps) */
  const float* gen; int n_out;        /* dev (B, n_out, HW): the module's normalised prediction for step t */
  const float* prev_in; float* next_in; int n_in;  /* dev (B, n_in, HW) packed model state (in_packer order) */
  int n_entries;                      /* one entry per distinct variable in (in_packer names) U (out names) */
  int out_idx[SDY_MAX_VARS];          /* index in the out packer or -1 (input-only: carried over, e.g. HGTsfc) */
  int in_idx[SDY_MAX_VARS];           /* index in the in packer or -1 (diagnostic-only output) */
  float* gen_norm_tl[SDY_MAX_VARS];   /* per entry with out_idx >= 0: dev (B, T1, HW) normalised timeline */
  float* gen_tl[SDY_MAX_VARS];        /* ... and denormalised timeline (value*std + mean) */
  float mean[SDY_MAX_VARS], std[SDY_MAX_VARS];
  int presc_entry;                    /* entry overwritten by the prescriber, or -1 */
  const float* presc_target;          /* dev (B, T1, HW) denormalised data of the prescribed variable */
  const float* presc_mask;            /* dev (B, T1, HW) mask variable */
  int mask_value, interpolate;
  const float* ar_init;               /* NULL, or dev (B, n_out, HW): a separate state to feed back into next_in instead of
                                         `gen` (use_cold_sampling_for_last_step = False hands the cold-sampled state over as
                                         "preds_autoregressive_init", stepper_multistep.py:412-418); prescribed like `gen`,
                                         the timelines still receive `gen` */
} sdy_step_finish_args;
/* prescriber + unpack into the timelines + denormalise + autoregressive feedback into next_in */
int sdy_step_finish(const sdy_step_finish_args* args, void* stream);
/* timeline slot 0: gen_norm_tl[e][b][0] = (data - mean)/std, gen_tl = that * std + mean, for the table's variables
 * (tl_norm / tl_denorm: arrays of nvars dev pointers) */
int sdy_init_timeline(const sdy_var_table* vars, int T1, int B, int HW, float* const* tl_norm, float* const* tl_denorm,
                      void* stream);
/* LpLoss.rel terms: terms[b][0] += sum (gen - target_norm)^2, terms[b][1] += sum target_norm^2 over the packed
 * (n_out, HW) fields of sample b, target_norm from `targets` at time t.  terms: dev double [B*2], zeroed by the caller. */
int sdy_lp_rel_terms(const float* gen, const sdy_var_table* targets, int t, int T1, int B, int HW, double* terms,
                     void* stream);

/* On-device ensemble diagnostics (src/ace_inference/core/metrics.py:32-54 weighted_mean, :107-132 RMSE, :135-144
 * ensemble_spread, :158-208 weighted_crps, :84-104 bias), one pass over the ensemble.
 *   pred: dev, member m of plane p at pred + m*member_stride + p*HW (M <= 64 members); truth: dev (n_planes, HW);
 *   weights: dev (HW) area weights (any normalisation: the caller divides by their sum)
 *   out[p][0..3] += sum_w (mean_m x - truth)^2 | sum_w var_m(x) (unbiased) | sum_w fair CRPS | sum_w (mean_m x - truth)
 * out: dev double [n_planes*4], zeroed by the caller. */
int sdy_ensemble_metrics(const float* pred, const float* truth, const float* weights, int M, long member_stride,
                         int n_planes, int HW, double* out, void* stream);

/* Per-timestep series terms of the reduced inference aggregator (MeanAggregator / AreaWeightedReducedMetric,
 * src/ace_inference/core/aggregator/inference/reduced.py:105-266; metrics src/ace_inference/core/metrics.py:32-208), one
 * pass over the ensemble.  Plane p = (sample s, time t), p = s*T + t:
 *   member m of plane p at pred + m*member_stride + s*sample_stride + t*HW (M <= 64; the strides let the window driver's
 *   member-stacked (members, samples, time, HW) VIEW of its device batch be read without a copy);
 *   truth plane p at truth + s*truth_sample_stride + t*HW; weights: dev (HW), any normalisation.
 *   out[p][0..7] += sum_w (mean_m x - truth)^2 | sum_w var_m(x) (unbiased) | sum_w fair CRPS | sum_w (mean_m x - truth) |
 *                   sum_w mean_m x | sum_w (mean_m x)^2 | sum_w truth | sum_w truth^2
 * out: dev double [n_sample*T*8], zeroed by the caller. */
int sdy_ensemble_series(const float* pred, int M, long member_stride, long sample_stride, const float* truth,
                        long truth_sample_stride, const float* weights, int n_sample, int T, int HW, double* out,
                        void* stream);

/* Time-mean accumulation of the inference aggregator (src/ace_inference/core/aggregator/inference/time_mean.py:97-117,
 * _add_or_initialize_time_mean): acc[p] += scale * sum over rows (r0, r1) and times t0 <= t < T of
 * x[r0*stride0 + r1*stride1 + t*HW + p].  x: dev, one variable of a window, (n0, n1, T, HW) with float strides for the
 * two leading axes (members, samples; a transposed view needs no copy); acc: dev (HW) running map.  scale = 1 / (n0 * n1 *
 * (T - t0)) gives the reference's mean over members, samples and time (t0 = 1 skips a window's initial condition). */
int sdy_time_mean_accumulate(const float* x, int n0, long stride0, int n1, long stride1, int t0, int T, int HW, float scale,
                             float* acc, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Sticky status word of the CURRENT device.  Kernels only ever set bits; the host reads (and optionally clears) it once per
 * window, not per launch (MultiStepStepper.run_on_batch does, after the window's single loss read-back).
 *   SDY_FLAG_NONFINITE  an InstanceNorm statistic (sum or sum of squares over H x W) came out inf / NaN: the tensor feeding
 *                       that norm holds non-finite values.
 *   SDY_FLAG_F16_RANGE  a split-precision ("h3") kernel was handed an activation with |value * scale| >= 65504, the fp16
 *                       range of its hi part (scale is 16 for the convolutions and the MLP): the products turn into inf / NaN.
 *                       The fp32-MFMA mode (gemm_mode 0, SDY_GEMM_MODE=f32) has no such limit.
 * sdy_status_flags synchronises `stream` before reading. */
#define SDY_FLAG_NONFINITE 1u
#define SDY_FLAG_F16_RANGE 2u
int sdy_status_flags(unsigned* flags, int reset, void* stream);
/* The same without the synchronisation: the word is copied to `flags_host` (pinned host memory) when `stream` reaches the
 * call, and cleared behind the copy if `reset`; read it after an event recorded behind the call has completed (the window
 * driver reads a window's word, together with its loss terms, while the next window computes). */
int sdy_status_flags_async(unsigned* flags_host, int reset, void* stream);
/* Range headroom (debug read-back; first contact with a trained checkpoint should report "x N below the cliff", not pass /
 * fail).  While enabled (process-wide switch; off by default: the bookkeeping is one atomic per tile), the split-precision
 * kernels of the default path record the largest magnitude they stage as fp16 -- pre-scale included, i.e. the number that
 * must stay below 65504 -- per consumer class, on the current device:
 *   [0] conv_h3 (inner skip and the other Cin -> 256 convolutions)   [1] mlp_h3's x tile   [2] dh_h3's coefficient rows
 *   [3] Legendre analysis input, recorded where it is PRODUCED (rfft360's stores; the folded kernel adds the two hemispheres'
 *       entries, so the recorded value is 2 x 16 x |Xf|)   [4] Legendre synthesis input, recorded at dh_h3's stores (16 x |Cs|).
 * The folded Legendre kernel has no register to spare for a tracker of its own: the two producers raise SDY_FLAG_F16_RANGE
 * for it (rfft360 at 2 x 16 x |Xf| >= 65504 -- conservative by at most the factor 2 of the fold).
 * sdy_range_headroom copies the five values (0 where a class has not run), clears them if `reset`, and synchronises `stream`.
 * headroom factor = 65504 / value. */
#define SDY_RANGE_SLOTS 5
int sdy_range_headroom_enable(int on);
int sdy_range_headroom(float* max_staged5, int reset, void* stream);
/* The dropout stream's generator evaluated on the host by the library itself (the same function the kernels inline):
 * the number of Philox rounds it was built with (7; tests hold it to oracle/philox.py and to Random123's seven-round
 * known-answer vectors) and the four words of one counter / key pair. */
int sdy_dropout_stream_rounds(void);
int sdy_dropout_stream_words(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t key_lo, uint32_t key_hi,
                             uint32_t* out4);

/* ---------------------------------------------------------------------------------------------------------
 * Measurement.  While enabled, every kernel launch of sdy_sfno_forward is bracketed by a pair of
 * hipEvents recorded ON THE LAUNCH STREAM, tagged with its stage (fused MLP with / without dropout, inner-skip conv,
 * Legendre analysis / synthesis, rfft, irfft, dhconv, encoder / decoder convs, ...).  sdy_profile_read synchronises
 * those events, returns per stage the summed elapsed milliseconds and the launch count since the last read, and resets.
 * Not for timed regions: the extra event records add a few microseconds between kernels.  Process-wide switch. */
int sdy_profile_enable(int on);
int sdy_profile_stage_count(void);
const char* sdy_profile_stage_name(int stage);
int sdy_profile_read(double* total_ms, long* launches, int n);   /* arrays of n >= sdy_profile_stage_count() */
/* The same, plus per stage the summed batch rows its launches worked on (rows / launches = average batch of a launch: with
 * the drop-path skip a block's kernels run on the trajectories its DropPath draw keeps). */
int sdy_profile_read_rows(double* total_ms, long* launches, long* rows, int n);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SDY_AMD_H */
