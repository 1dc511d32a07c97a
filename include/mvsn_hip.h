/*
 * mvsn_hip.h -- C ABI of libmvsn_hip.so, the MI355X (gfx950) plane-sweep hot path of
 * MultiViewStereoNet.
 *
 * The reference (robustrobotics/multi_view_stereonet) has no native layer: its "kernels" are
 * stock ATen ops called from multi_view_stereonet/multi_view_stereonet.py.  Each entry point
 * below replaces one group of those call sites (cited as file:line into the reference).  A
 * maintainer binds them with ctypes (INTEGRATION.md shows the stub) from inside
 * MultiViewStereoNet.forward().
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into HBM; tensors are dense fp32, row-major in the
 *     index order written next to them (NCHW / NCDHW as in the reference); masks are uint8
 *     (0/1), bit-compatible with torch.bool storage;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); every call only
 *     enqueues work on that stream and never synchronises the device;
 *   - return value: 0 on success, a positive hipError_t, or a negative MVSN_E_* code;
 *     mvsn_last_error() returns a thread-local message for the last non-zero return;
 *   - a "chain" is one (reference image, source view) pair; chains are indexed
 *     n = source * B + image, so n % B is the reference-image index;
 *   - P = rows*cols of the coarsest (1/16) pyramid level, D = number of idepth hypotheses.
 */
#ifndef MVSN_HIP_H
#define MVSN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVSN_ABI_VERSION 5

#define MVSN_E_BADARG (-1)      /* null pointer, non-positive size, unsupported channel count */
#define MVSN_E_TOOLARGE (-2)    /* shape exceeds what the kernel's LDS/global plan supports */
#define MVSN_E_WORKSPACE (-3)   /* workspace pointer null or smaller than the *_workspace_bytes() answer */

typedef void *mvsn_stream_t;

int mvsn_abi_version(void);
const char *mvsn_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Plane-sweep set-up: per chain, the baseline-normalised pose, the D idepth samples, the
 * fronto-parallel homographies at the coarsest level, the incremental homographies
 * H_inc[d] = H[d-1]^-1 H[d] (H_inc[0] = I) and the full-resolution homography of plane 0.
 * Replaces create_idepth_samples (multi_view_stereonet.py:131-165 -> stereo/image_predictor.py:120-209),
 * create_plane_sweep_homographies (:167-194 -> image_predictor.py:400-461), the per-source
 * baseline renormalisation (:566-571) and the per-step torch.inverse/matmul (:281-282).
 *   T_right_in_left (N,4,4)  K_lvl0 (N,4,4)  K_lvl4 (N,4,4)
 *   idepth_samples (N,D)  H_lvl4 (N,D,3,3)  H_inc (N,D,3,3)  H_lvl0_plane0 (N,3,3)  baseline (N)
 * The idepth samples and the three homography outputs carry the bits of the reference's own fp32 evaluation (its
 * per-pixel tensor program and torch.sum's order; torch's CPU inverse of the pose, of the intrinsics and of H[d-1],
 * its 3x3 products: csrc/mvsn_setup.hip, namespace ref32; pinned by tests/golden/g11_incremental_homographies.npz)
 * when the intrinsics are [[fx,0,cx],[0,fy,cy],[0,0,1]]; for any other intrinsics they are a double-precision
 * evaluation rounded once.
 * ------------------------------------------------------------------------------------------- */
int mvsn_plane_sweep_setup(const float *T_right_in_left, const float *K_lvl0, const float *K_lvl4,
                           int n_chains, int rows4, int cols4, int num_idepth_samples,
                           float *idepth_samples, float *H_lvl4, float *H_inc, float *H_lvl0_plane0,
                           float *baseline, mvsn_stream_t stream);
/* The same with the poses and intrinsics as the forward holds them: one (batch, 4, 4) pose tensor per source view
 * (host array of n_sources <= 8 device pointers) and the batch's intrinsics shared by its sources; chain n = s * batch + b.
 * Saves the torch.cat / repeat of :553,:587-592 in front of the launch. */
int mvsn_plane_sweep_setup_sources(const float *const *T_right_in_lefts, int n_sources, const float *K_lvl0,
                                   const float *K_lvl4, int batch, int rows4, int cols4, int num_idepth_samples,
                                   float *idepth_samples, float *H_lvl4, float *H_inc, float *H_lvl0_plane0,
                                   float *baseline, mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Homography warp with bilinear, clamp-to-edge sampling and out-of-image zeroing.
 * Replaces PlaneSweepWarper.forward (multi_view_stereonet.py:205-235) ->
 * HomographyImagePredictor.forward (stereo/image_predictor.py:470-523, grid_sample bilinear /
 * border / align_corners=False).  mask = |nx|>1 or |ny|>1 on the normalised coordinate.
 *   image (B,C,rows,cols)  H (B,n,3,3)  ->  volume (B,C,n,rows,cols)  mask (B,n,rows,cols) u8
 * ------------------------------------------------------------------------------------------- */
int mvsn_homography_warp(const float *image, const float *H, int batch, int channels, int n_planes,
                         int rows, int cols, float *volume, uint8_t *mask, mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The fused incremental chain: for every chain, one persistent workgroup keeps the source-view
 * feature map resident on-chip and walks d = 1..D-1: warp the previous plane's features by
 * H_inc[d], warp the coarse source image by H[d], run the FeatureRefiner (conv 35->32, GN,
 * LReLU, residual block, conv 32->32, +features) on fp32 MFMA, and emit the cost-volume slice
 * (not mask) * |left - right| and the mask slice.  Replaces
 * IncrementalFastGeometryAwareFeatureNetwork.forward :270-300 (everything after the plane-0
 * extractor), FeatureRefiner.forward :424-440 and the cost-volume build :553,:587-592.
 *   src_image_lvl4 (N,3,rows,cols)   H_lvl4, H_inc (N,D,3,3)   plane0_features (N,32,rows,cols)
 *   left_features (B,32,rows,cols)   refiner_packed: mvsn_feature_refiner_packed_floats() floats
 *   cost_volume (N,32,D,rows,cols)   mask_volume (N,D,rows,cols) u8
 *   feature_volume: optional (N,32,D,rows,cols) masked source features, or NULL
 *   workspace: mvsn_incremental_cost_volume_workspace_bytes_for() bytes, 16-byte aligned (may be 0 -> NULL allowed)
 *   form: MVSN_CHAIN_AUTO picks per coarse grid and number of chains (mvsn_incremental_cost_volume_form_for tells which);
 *         MVSN_CHAIN_DIRECT = the three 3x3 convolutions as direct implicit GEMMs (any grid up to 2048 px);
 *         MVSN_CHAIN_WINOGRAD = as Winograd F(2x2,3x3) products (fp32 throughout, 2.25x fewer multiplies; even
 *         rows/cols whose planes + one layer of transformed weights fit LDS, e.g. 16x32), MVSN_E_TOOLARGE otherwise.
 * ------------------------------------------------------------------------------------------- */
#define MVSN_CHAIN_AUTO 0
#define MVSN_CHAIN_DIRECT 1
#define MVSN_CHAIN_WINOGRAD 2
#define MVSN_CHAIN_STEPWISE 3 /* one plane per round of full-chip launches (warp, three Winograd convolutions with the
                                 GroupNorm statistics of the producing launch, the residual pass, the cost slice):
                                 for coarse grids whose planes do not fit one CU (30x40, 32x64), any number of chains
                                 (cols % 4 == 0); a selectable form -- since round 5 AUTO stays on the banded form there */
#define MVSN_CHAIN_BANDED 4   /* one chain on SEVERAL workgroups: the coarse plane cut into bands of pixel rows (16x32: 8
                                 bands of 2 rows up to CUs / 8 chains -- every layer split by transform-row half across a
                                 band's waves -- and 4 bands of 4 rows beyond, bit-identical; 30x40: 15 of 2; 32x64: 16 of
                                 2), Winograd arithmetic of
                                 MVSN_CHAIN_WINOGRAD; per step the bands hand each other the new feature rows their
                                 gathers reach into, their GroupNorm sums and one halo row per layer as tagged 8-byte
                                 write-through granules -- no fence, no placement assumption.  For few chains in flight
                                 (batch 1: the reference's loop, test.py:38,197-200): the workgroups of a launch must be
                                 co-resident, so more chains than CUs / bands run as consecutive passes inside the call
                                 (AUTO: one pass on 16x32), and only ONE such call may be in flight per device.
                                 30x40 / 32x64 beyond one pass of the thin bands (17 / 16 chains on 256 CUs): the SLAB
                                 plan -- 3 bands of 10 rows / 4 bands of 8 rows per chain, each a 512-thread workgroup
                                 that keeps its band's activation planes resident in LDS like the plane-resident
                                 Winograd kernel (85 / 64 chains per pass; passes of equal size, or full passes + ONE
                                 thin-band pass for a remainder of at most 17 / 16 chains; four hand-offs per
                                 step; mvsn_incremental_cost_volume_banded_groups tells which plan a call runs).
                                 Needs workspace (..._workspace_bytes_for); the word at
                                 mvsn_incremental_cost_volume_status_offset() inside it is 0 after a clean run. */
size_t mvsn_feature_refiner_packed_floats(void);
/* Pack the ten FeatureRefiner tensors (state_dict order: conv0.{weight,bias}, bn0.{weight,bias},
 * res0.conv1.{weight,bias}, res0.bn1.{weight,bias}, conv_final.{weight,bias}) into the MFMA
 * fragment order the chain kernel streams. */
int mvsn_pack_feature_refiner(const float *conv0_w, const float *conv0_b, const float *bn0_w,
                              const float *bn0_b, const float *res0_w, const float *res0_b,
                              const float *res0_bn_w, const float *res0_bn_b, const float *final_w,
                              const float *final_b, float *packed, mvsn_stream_t stream);
size_t mvsn_incremental_cost_volume_workspace_bytes(int n_chains, int rows, int cols);   /* MVSN_CHAIN_DIRECT: the step's moved
                                                                                             features (always, since ABI 3) + the
                                                                                             activation planes that do not fit LDS */
int mvsn_incremental_cost_volume_form(int rows, int cols);   /* the fused form of this grid: WINOGRAD or DIRECT */
/* what MVSN_CHAIN_AUTO resolves to for this many chains on this grid (BANDED, WINOGRAD, STEPWISE or DIRECT) in the
 * entry points that put a repair launch behind a banded call (mvsn_incremental_cost_volume_guarded / _bf16), and the
 * workspace `form` needs for num_idepth_samples planes (AUTO allowed: enough for either entry point).
 * The PLAIN mvsn_incremental_cost_volume has no repair launch, so its AUTO is more conservative: on 30x40 / 32x64 it
 * takes the banded form only while the chains fit ONE thin-band pass (17 / 16 chains on 256 CUs) and the
 * co-residency-free STEPWISE form (DIRECT for cols % 4 != 0) beyond -- the multi-pass slab plan holds the whole device
 * for milliseconds and a time-out without repair would leave NaN in the cost slice.  Ask for MVSN_CHAIN_BANDED (or use
 * the guarded entry) to get the slab plan. */
int mvsn_incremental_cost_volume_form_for(int n_chains, int rows, int cols);
size_t mvsn_incremental_cost_volume_workspace_bytes_for(int n_chains, int num_idepth_samples, int rows, int cols,
                                                        int form);
/* MVSN_CHAIN_BANDED only: byte offset, inside the workspace, of the 32-bit status word the launch leaves behind
 * (0 = every inter-workgroup hand-off completed; non-zero = a bounded wait timed out and the outputs are invalid) */
size_t mvsn_incremental_cost_volume_status_offset(int n_chains, int rows, int cols);
/* MVSN_CHAIN_BANDED only: workgroups per chain of the plan a call with this many chains runs (16x32: 8 or 4 thin bands;
 * 30x40 / 32x64: 15 / 16 thin bands, or 3 / 4 slabs with many chains in flight; 0: the grid has no banded plan) */
int mvsn_incremental_cost_volume_banded_groups(int n_chains, int rows, int cols);   /* (ABI 4) */
int mvsn_incremental_cost_volume(const float *src_image_lvl4, const float *H_lvl4, const float *H_inc,
                                 const float *plane0_features, const float *left_features,
                                 const float *refiner_packed, int n_chains, int batch,
                                 int num_idepth_samples, int rows, int cols, float *cost_volume,
                                 uint8_t *mask_volume, float *feature_volume, void *workspace,
                                 size_t workspace_bytes, int form, mvsn_stream_t stream);
/* The same call with the small-batch default made safe on a SHARED device (ABI 3).  When `form` resolves to
 * MVSN_CHAIN_BANDED, a second launch follows the banded one(s) on the stream: the single-launch form of the grid (plane-
 * resident Winograd on 16x32, the direct kernel elsewhere) GATED on the banded status word -- its workgroups read the word
 * and return at once when it is 0 (a few microseconds), and recompute cost_volume / mask_volume / feature_volume
 * completely when a hand-off timed out (the banded workgroups were not co-resident).  No host round trip, graph-capturable;
 * the outputs are valid either way.  `repair_workspace`: mvsn_incremental_cost_volume_repair_workspace_bytes() bytes (0 on
 * 16x32).  `sticky_status`: optional two 32-bit words in device-VISIBLE memory (device or pinned host memory; plain
 * system-scope loads / stores by one thread of the repair launch): [0] |= the status of every repaired call, [1] += 1
 * per repair -- never cleared by the library, so a host that looks at it later (or never synchronises per call) still
 * learns that the banded form does not fit this device's load.
 * Other forms: identical to mvsn_incremental_cost_volume.
 * Replaces: the reference's chain loop as above (multi_view_stereonet.py:279-290) -- which has no failure mode to repair. */
size_t mvsn_incremental_cost_volume_repair_workspace_bytes(int n_chains, int rows, int cols);
int mvsn_incremental_cost_volume_guarded(const float *src_image_lvl4, const float *H_lvl4, const float *H_inc,
                                         const float *plane0_features, const float *left_features,
                                         const float *refiner_packed, int n_chains, int batch,
                                         int num_idepth_samples, int rows, int cols, float *cost_volume,
                                         uint8_t *mask_volume, float *feature_volume, void *workspace,
                                         size_t workspace_bytes, int form, void *repair_workspace,
                                         size_t repair_workspace_bytes, unsigned *sticky_status, mvsn_stream_t stream);
/* bf16 FEATURE tier (BASELINE config 5's "bf16 features"; reported, never the parity path; ABI 5): the guarded call with
 * the cost volume STORED as bf16 -- `cost_volume_bf16` is (n_chains, 32, D, rows, cols) 2-byte elements, the values the
 * fp32 call writes rounded to nearest-even (v_cvt_pk_bf16_f32): Kernel A's dominant HBM stream halves (SURVEY 8d:
 * config 5 137.5 MB -> ~69 MB per depth map).  mask_volume / feature_volume / workspaces / status as in the guarded
 * call; every form but MVSN_CHAIN_STEPWISE has the variant (MVSN_E_BADARG there: run the fp32 call and convert).
 * Consumer: mvsn_conv_forward_bf16_storage(in_is_bf16 = 1), whose bf16 operand conversion of an fp32 volume yields the
 * same bits -- the stored volume costs the regulariser no accuracy beyond the bf16-operand tier's.
 * Replaces: the cost volume of multi_view_stereonet.py:553,587-592 at config 5's storage precision. */
int mvsn_incremental_cost_volume_bf16(const float *src_image_lvl4, const float *H_lvl4, const float *H_inc,
                                         const float *plane0_features, const float *left_features,
                                         const float *refiner_packed, int n_chains, int batch,
                                         int num_idepth_samples, int rows, int cols, void *cost_volume_bf16,
                                         uint8_t *mask_volume, float *feature_volume, void *workspace,
                                         size_t workspace_bytes, int form, void *repair_workspace,
                                         size_t repair_workspace_bytes, unsigned *sticky_status, mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Direct convolution on fp32 MFMA (implicit GEMM, weights = A, activations = B), 2-D or 3-D,
 * C_out in {1..32}, 'same' padding = dilation*(k/2), stride 1 or 2 (2-D only), with
 *   - an optional input transform fused into the tile load:
 *         x = LeakyReLU_0.2(GroupNorm_4(in))                        (in_stats != NULL)
 *         x = in_residual + LeakyReLU_0.2(GroupNorm_4(in))          (in_residual != NULL: a residual
 *             block x + LReLU(GN(conv(x))), utils/resnet.py:93-109, folded into the NEXT layer's load)
 *     and, with out_staged != NULL, x itself written out once (N,C_in,H,W) as a by-product
 *     (2-D 3x3 stride-1 layers only) -- the normalise/activate/add pass never runs on its own;
 *   - bias add, and per-workgroup GroupNorm partials of the OUTPUT (out_partials != NULL) that
 *     mvsn_groupnorm_finalize turns into (mean, rstd) per (sample, group).
 * Replaces every conv2d/conv3d + GroupNorm + LeakyReLU call site of FeatureNetwork (:109-129),
 * CostVolumeFilter (:341-353) and IDepthmapRefiner (:468-484).
 *   in (N,C_in,[D,]H,W)  weight_packed: mvsn_conv_packed_floats() floats  bias (C_out) or NULL
 *   in_stats (N,4,2) mean,rstd   in_gamma,in_beta (C_in)   in_residual, out_staged (N,C_in,H,W)
 *   out (N,C_out,[D,]Ho,Wo)
 *   out_partials (N, R, 4, 3) {count, mean, M2}, R = mvsn_conv_num_tiles() partial records per sample
 *   (one per workgroup tile and wave; written without any workgroup-level synchronisation)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int n;         /* samples */
  int c_in;      /* input channels (any >= 1) */
  int c_out;     /* 1..32 */
  int depth;     /* 1 for 2-D */
  int rows, cols;/* input spatial size */
  int kd, kh, kw;/* kernel extent; kd = 1 for 2-D */
  int stride;    /* 1 or 2 (rows/cols only) */
  int dilation;  /* rows/cols only */
  int precision; /* MVSN_CONV_FP32, MVSN_CONV_FP32_WINO or MVSN_CONV_BF16X3: arithmetic / algorithm, see below */
} mvsn_conv_desc;

/* Arithmetic of mvsn_conv_forward.
 *   MVSN_CONV_FP32    v_mfma_f32_16x16x4_f32: bit-for-bit an fp32 fmaf chain.
 *   MVSN_CONV_BF16X3  fp32 operands split into bf16 hi + lo, a*b ~= ah*bh + ah*bl + al*bh on
 *                     v_mfma_f32_16x16x32_bf16 with fp32 accumulation (~2^-16 relative per product;
 *                     "3 x bf16 split", the fp32-equivalent tier BASELINE.md section 2 allows).  Only the
 *                     32 -> 32 channel 3x3 / 3x3x3 stride-1 layers (mvsn_conv_bf16x3_supported); weights are
 *                     packed per precision, in_residual / out_staged are not available.
 *   MVSN_CONV_FP32_WINO  the same fp32 MFMA arithmetic on the Winograd F(2x2,3x3) form of the layer: 16
 *                     products per 2x2 outputs and (cin, cout) pair instead of 36; every operand and
 *                     accumulation is fp32, the result differs from MVSN_CONV_FP32 by rounding only (~1e-6
 *                     relative).  2-D 3x3 stride-1 layers with 32 output channels, dilation 1/2/4/8, and the
 *                     3x3x3 32 -> 32 layers (volume form: 2-D Winograd products summed over the depth tap, 12
 *                     multiplies per output instead of 27), cols % 4 == 0 (mvsn_conv_winograd_supported);
 *                     weights are packed per form.  Also the 5x5 stride-2 32 -> 32 layers of the feature extractor
 *                     (multi_view_stereonet.py:78-129), cols % 8 == 0: F(2x2,3x3) on the input's four stride-2 phases,
 *                     392 products per 2x2 outputs and (cin, cout) pair instead of 800; no fused input transform, no
 *                     GroupNorm partials on that form (MVSN_E_BADARG).
 *   MVSN_CONV_BF16    plain bf16 operands (the hi halves only), fp32 accumulation, on the same kernels and packed
 *                     weights as MVSN_CONV_BF16X3: BASELINE config 5's speed tier.  ~2^-9 relative per operand:
 *                     the final depth lands OUTSIDE the 1e-3 parity contract (measured ~2e-3 mean-rel). */
#define MVSN_CONV_FP32 0
#define MVSN_CONV_BF16X3 1
#define MVSN_CONV_FP32_WINO 2
#define MVSN_CONV_BF16 3
int mvsn_conv_bf16x3_supported(const mvsn_conv_desc *desc);
/* bf16 STORAGE on top of MVSN_CONV_BF16 (ABI 4; BASELINE config 5's "bf16 features" for the 3x3x3 regulariser layers,
 * multi_view_stereonet.py:341-353): `in` and / or `out` hold bf16 (same NCDHW element order, two bytes per element) instead
 * of fp32 -- `in_is_bf16` / `out_is_bf16`, at least one set.  Products on the bf16 matrix cores with fp32 accumulation,
 * the biased accumulators rounded to bf16 (RNE) by the storing epilogue; `out_partials` (GroupNorm records) and
 * `in_stats` stay fp32 and are formed from the unrounded accumulators.  desc: kd = 3, precision MVSN_CONV_BF16;
 * weight_packed as for MVSN_CONV_BF16.  A speed tier outside the 1e-3 parity contract (its own asserted budget:
 * tests/test_hip_parity.py); never part of the default path. */
int mvsn_conv_forward_bf16_storage(const mvsn_conv_desc *desc, const void *in, int in_is_bf16,
                                   const float *weight_packed, const float *bias, const float *in_stats,
                                   const float *in_gamma, const float *in_beta, void *out, int out_is_bf16,
                                   float *out_partials, mvsn_stream_t stream);
int mvsn_conv_winograd_supported(const mvsn_conv_desc *desc);

size_t mvsn_conv_packed_floats(const mvsn_conv_desc *desc);
int mvsn_conv_pack_weights(const mvsn_conv_desc *desc, const float *weight, float *packed,
                           mvsn_stream_t stream);
int mvsn_conv_num_tiles(const mvsn_conv_desc *desc);
int mvsn_conv_forward(const mvsn_conv_desc *desc, const float *in, const float *weight_packed,
                      const float *bias, const float *in_stats, const float *in_gamma,
                      const float *in_beta, const float *in_residual, float *out_staged, float *out,
                      float *out_partials, mvsn_stream_t stream);
/* mvsn_conv_forward on an input handed over as up to three channel blocks instead of one tensor: block b is
 * a contiguous (N, block_channels[b], rows, cols) tensor and the layer convolves their channel-wise
 * concatenation (sum of block_channels = desc->c_in) without it ever being assembled.  Replaces the
 * torch.cat([image, features, idepth], 1) in front of every IDepthmapRefiner (multi_view_stereonet.py:466,
 * :602-605).  Winograd form only (desc->precision = MVSN_CONV_FP32_WINO, mvsn_conv_winograd_supported);
 * 1 <= num_blocks <= 3, every block 16-byte aligned. */
int mvsn_conv_forward_blocks(const mvsn_conv_desc *desc, const float *const *in_blocks, const int *block_channels,
                             int num_blocks, const float *weight_packed, const float *bias, float *out,
                             float *out_partials, mvsn_stream_t stream);
/* A normalise / activate / add pass handed over as a job: the arguments of mvsn_groupnorm_lrelu_apply (r_stats NULL,
 * residual optional) or of mvsn_groupnorm_lrelu_add2 (r_stats / r_gamma / r_beta set: the residual is a raw conv output). */
typedef struct mvsn_apply_job {
  const float *x, *stats, *gamma, *beta;
  const float *residual, *r_stats, *r_gamma, *r_beta;
  float *out;      /* may alias x */
  int n;           /* samples: (n, 32, spatial) */
  int reverse;     /* 1: the carrying launch walks its tiles, and with them the job, from the end to the beginning
                      (consecutive launches of a pipelined tower alternate, so that each starts on what the one before
                      it touched last: a few per cent of the bytes then come from the memory-side cache); results
                      identical either way */
  long spatial;
} mvsn_apply_job;
/* mvsn_conv_forward (no in_residual / out_staged) AND an independent apply job in one call.  The residual blocks
 * x + LeakyReLU(GroupNorm(conv(x))) (multi_view_stereonet.py:38-48, used :448-474) alternate a convolution bound by the
 * matrix pipe with a pass bound by HBM; with the batch cut in two slices, slice B's convolution can carry slice A's
 * pass in its own launch (the job's 16-byte loads / stores are issued by the convolution's waves between their
 * multiplies).  The job must not depend on this convolution's output, nor the convolution on the job's.  Where the
 * layer's kernel cannot carry it (not a Winograd 32 -> 32 2-D layer, spatial % 256 != 0, job larger than the layer's
 * own output) the job runs as its own launch first: results are bit-identical either way; *carried (may be NULL)
 * says which it was. */
int mvsn_conv_forward_carry(const mvsn_conv_desc *desc, const float *in, const float *weight_packed, const float *bias,
                            const float *in_stats, const float *in_gamma, const float *in_beta, float *out,
                            float *out_partials, const mvsn_apply_job *job, int *carried, mvsn_stream_t stream);
/* partials (N,R,4,3) {count, mean, M2} per record and group (R = mvsn_conv_num_tiles records per sample, passed as
 * `tiles`) -> stats (N,4,2) = {mean, rstd}, eps 1e-5, biased variance; records are 48 bytes, 16-byte aligned */
int mvsn_groupnorm_finalize(const float *partials, int n, int tiles, float *stats, mvsn_stream_t stream);
/* The same statistics with a sample's records cut into up to 16 slices reduced by separate workgroups and added in slice
 * order (ABI 4): for layers that leave MANY records per sample (more than 2048: a level-0 layer of a 1024x512 frame
 * leaves 32768 = 1.5 MB) on FEW samples -- one workgroup per sample then reads megabytes alone.  The number of slices
 * depends on `tiles` only (a sample's statistics do not depend on the batch it travels in; up to 2048 records it is 1 and
 * the call IS mvsn_groupnorm_finalize).  Deterministic; equal to the one-workgroup result up to the rounding of the
 * double-precision sums.  `workspace`: mvsn_groupnorm_finalize_split_workspace_bytes() bytes, 8-byte aligned. */
size_t mvsn_groupnorm_finalize_split_workspace_bytes(int n, int tiles);
int mvsn_groupnorm_finalize_split(const float *partials, int n, int tiles, float *stats, void *workspace,
                                  size_t workspace_bytes, mvsn_stream_t stream);
/* out = [residual +] LeakyReLU_0.2(GroupNorm(x)) on (N,32,spatial); residual may be NULL; out may alias x */
int mvsn_groupnorm_lrelu_apply(const float *x, const float *stats, const float *gamma, const float *beta,
                               const float *residual, int n, long spatial, float *out, mvsn_stream_t stream);
/* out = LeakyReLU(GroupNorm(x)) + LeakyReLU(GroupNorm_r(r)): the first residual block of a refiner
 * (:472-474) when the head's activation x0 = LReLU(GN(conv0)) is never materialised -- conv0's raw output r
 * feeds block 1's convolution through mvsn_conv_forward's in_stats transform and this call as the residual. */
int mvsn_groupnorm_lrelu_add2(const float *x, const float *stats, const float *gamma, const float *beta,
                              const float *r, const float *r_stats, const float *r_gamma, const float *r_beta, int n,
                              long spatial, float *out, mvsn_stream_t stream);
/* mvsn_groupnorm_lrelu_apply / _add2 handed the producing convolution's RECORDS (N, tiles, 4, 3) instead of finalised
 * statistics: every workgroup of the pass forms x's (mean, rstd) itself with mvsn_groupnorm_finalize's own code (same
 * bits), so the dependent finalize launch in front of the pass disappears -- small batches, where that launch (7 us)
 * costs more than re-reading a few hundred records per workgroup.  r_stats NULL: plain pass (residual optional);
 * r_stats set (finalised, (N,4,2)): the _add2 form with the raw residual `residual`. */
int mvsn_groupnorm_lrelu_apply_records(const float *x, const float *records, int tiles, const float *gamma,
                                       const float *beta, const float *residual, const float *r_stats,
                                       const float *r_gamma, const float *r_beta, int n, long spatial, float *out,
                                       mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 32 -> 1 channel 3x3 (kd = 1) or 3x3x3 (kd = 3) convolution, 'same' padding, dilation 1: the last
 * layer of CostVolumeFilter (conv4, :337,:351) and of every IDepthmapRefiner (conv_final, :464,:480).
 * HBM-bound (128 input bytes per output): a tap GEMM P[taps][position] = W[taps x 32] * in[32 x position]
 * on the fp32 matrix cores (every input element is loaded once, 9 or 27 taps on the MFMA row dimension),
 * then a shift-and-add of the tap planes out of LDS.
 * With prior != NULL it also applies the refiner's epilogue (:482 and the gain trick :607-611):
 *     out = relu(prior * fx[n] + conv + bias) / fx[n]
 *   in (N,32,[D,]H,W)  weight (1,32,[3,]3,3) UNPACKED  bias (1) or NULL  prior (N,1,H,W)  fx (N)
 *   out (N,[D,]H,W).  Requires cols % 4 == 0 (mvsn_conv_to1_supported); otherwise use mvsn_conv_forward.
 * ------------------------------------------------------------------------------------------- */
int mvsn_conv_to1_supported(int rows, int cols);
int mvsn_conv_to1(const float *in, const float *weight, const float *bias, const float *prior, const float *fx,
                  int n, int depth, int rows, int cols, int kd, float *out, mvsn_stream_t stream);
/* The 3-D layer on a RAW convolution output: LeakyReLU(GroupNorm(in_raw)) (CostVolumeFilter's bn3 + relu,
 * multi_view_stereonet.py:349-350) is applied while the planes are loaded, so the regulariser's last normalise /
 * activate pass never goes through HBM.   in_raw (N,32,D,H,W)  in_stats (N,4,2)  in_gamma, in_beta (32)  out (N,D,H,W) */
int mvsn_conv_to1_volume_norm(const float *in_raw, const float *in_stats, const float *in_gamma, const float *in_beta,
                              const float *weight, const float *bias, int n, int depth, int rows, int cols, float *out,
                              mvsn_stream_t stream);
/* The same 2-D layer with the tower's LAST residual block folded into its load: the input
 * in_residual + LeakyReLU(GroupNorm(in_raw)) (SimpleBasicBlock, multi_view_stereonet.py:21-38) is formed in
 * registers from the raw output of the block's convolution and the block's input, never written.
 *   in_raw, in_residual (N,32,H,W)  in_stats (N,4,2)  in_gamma, in_beta (32)  in_residual may be NULL */
int mvsn_conv_to1_block(const float *in_raw, const float *in_stats, const float *in_gamma, const float *in_beta,
                        const float *in_residual, const float *weight, const float *bias, const float *prior,
                        const float *fx, int n, int rows, int cols, float *out, mvsn_stream_t stream);
/* The same with in_raw's statistics formed inside the launch from its records (N, tiles, 4, 3), see
 * mvsn_groupnorm_lrelu_apply_records. */
int mvsn_conv_to1_block_records(const float *in_raw, const float *in_records, int tiles, const float *in_gamma,
                                const float *in_beta, const float *in_residual, const float *weight, const float *bias,
                                const float *prior, const float *fx, int n, int rows, int cols, float *out,
                                mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Soft-argmin over the hypothesis axis: out = sum_d softmax(-cost)_d * idepth_d.
 * Replaces extract_idepthmap (multi_view_stereonet.py:486-492).
 *   cost (N,D,P)  idepth_samples (N,D)  ->  idepth (N,P)
 * ------------------------------------------------------------------------------------------- */
int mvsn_soft_argmin(const float *cost, const float *idepth_samples, int n, int num_idepth_samples,
                     int pixels, float *idepth, mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The three elementwise steps the flag branches of forward() leave outside the fused kernels:
 *   mvsn_channel_l2_norm      do_cost_volume_filter = False: cost = ||cost_volume||_2 over the channel axis
 *                             (torch.norm(.., dim=1), multi_view_stereonet.py:598).  x (N,C,P) -> out (N,P)
 *   mvsn_idepth_scale         out = prior * fx[n]: the refiner's input channel in the gain trick (:607-611 etc.)
 *   mvsn_refiner_epilogue     out = relu(prior * fx[n] + delta) / fx[n] (:482 + the division that follows), for the
 *                             shapes whose 32 -> 1 layer does not take the fused form of mvsn_conv_to1
 *   prior, delta, out (N,P)   fx (N)
 * ------------------------------------------------------------------------------------------- */
int mvsn_channel_l2_norm(const float *x, int n, int channels, long pixels, float *out, mvsn_stream_t stream);
int mvsn_idepth_scale(const float *prior, const float *fx, int n, long pixels, float *out, mvsn_stream_t stream);
int mvsn_refiner_epilogue(const float *prior, const float *fx, const float *delta, int n, long pixels, float *out,
                          mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Bilinear resize, align_corners=False, to an arbitrary target size (Upsampler :372-380), and
 * the boolean-mask variant float -> bilinear -> (> 0.5) (MaskUpsampler :389-396).
 *   in (N,C,h,w) -> out (N,C,H,W); mask bytes are 0 or 1 (torch.bool)
 * ------------------------------------------------------------------------------------------- */
int mvsn_upsample_bilinear(const float *in, int n, int channels, int rows_in, int cols_in, int rows_out,
                           int cols_out, float *out, mvsn_stream_t stream);
/* upsample_bilinear of a one-channel idepth map plus its per-sample scaled copy out * fx[n] (the refiner's input
 * channel idepth * fx, multi_view_stereonet.py:607-611) in one pass */
int mvsn_upsample_prior(const float *in, const float *fx, int n, int rows_in, int cols_in, int rows_out, int cols_out,
                        float *out, float *out_scaled, mvsn_stream_t stream);
int mvsn_upsample_mask(const uint8_t *in, int n, int channels, int rows_in, int cols_in, int rows_out,
                       int cols_out, uint8_t *out, mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * One level of the ceil-halving area pyramid the forward's inputs are built from
 * (build_image_pyramid, utils/image_utils.py:111-128 = interpolate(mode="area")).
 *   in (N,C,h,w) -> out (N,C,(h+1)/2,(w+1)/2)
 * ------------------------------------------------------------------------------------------- */
int mvsn_area_downsample(const float *in, int n, int channels, int rows_in, int cols_in, float *out,
                         mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Input preparation in two launches (multi_view_unpack_batch, multi_view_stereonet_utils.py:541-641):
 * mvsn_image_pyramid: every level of the area pyramid in one pass over the frames, for sizes divisible by
 * 2^(levels-1) (mvsn_image_pyramid_supported; otherwise mvsn_area_downsample level by level).
 *   in (N,C,rows,cols)  out_levels[l-1] -> (N,C,rows>>l,cols>>l), l = 1..levels-1 (HOST array of device pointers)
 * mvsn_prepare_cameras: K pyramid (:575-581: fx*=sx, fy*=sy, cx = sx(cx+0.5)-0.5, cy likewise, sx = w_l/w_0),
 * the source poses and their inverses with the translations divided by the baseline to the FIRST source (:597-604),
 * and that baseline.
 *   K (B,4,4)  T_right_in_left (S,B,4,4) un-normalised  level_sizes_dev: DEVICE int[2*levels] = rows_0, cols_0, rows_1, ...
 *   -> K_pyr (levels,B,4,4)  T_normalised, T_inverse_normalised (S,B,4,4)  baseline (B)
 * ------------------------------------------------------------------------------------------- */
int mvsn_image_pyramid_supported(int rows, int cols, int levels);
int mvsn_image_pyramid(const float *in, int n, int channels, int rows, int cols, int levels, float *const *out_levels,
                       mvsn_stream_t stream);
int mvsn_prepare_cameras(const float *K, const float *T_right_in_left, int batch, int n_sources, int levels,
                         const int *level_sizes_dev, float *K_pyr, float *T_normalised, float *T_inverse_normalised,
                         float *baseline, mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-source fusion (multi_view_stereonet.py:615-627): per chain divide by its baseline, mean
 * over the S sources; mask = mean(mask) > 0.5.
 *   raw, refined (S*B,P)  baseline (S*B)  mask (S*B,D,P) u8
 *   -> raw_out, refined_out (B,P)  mask_out (B,D,P) u8
 * `refined_aliases_raw` != 0 reproduces the reference's double division when refiner 4 is off.
 * ------------------------------------------------------------------------------------------- */
int mvsn_fuse_sources(const float *raw, const float *refined, const float *baseline, const uint8_t *mask,
                      int n_sources, int batch, int num_idepth_samples, int pixels, int refined_aliases_raw,
                      float *raw_out, float *refined_out, uint8_t *mask_out, mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Two-view consistency (the bidirectional path, multi_view_stereonet_utils.py:503-539 with
 * estimate_right_idepthmap, and what its outputs feed in multi_view_stereonet/losses.py).
 *
 * mvsn_idepth_reproject: every pixel (x, y, idepth) of THIS view is lifted to 3-D, moved into the OTHER view and
 * projected: its idepth there, its normalised pixel coordinate there, whether that leaves the image
 * (IDepthmapProjector.forward, stereo/image_predictor.py:538-576), and the other view's idepth map (and optionally a
 * mask, as float, > 0) sampled at that coordinate (grid_sample bilinear / border / align_corners=False,
 * losses.py:59-61, :129-135).  absdiff_partials (optional, batch * mvsn_idepth_reproject_blocks(pixels) floats)
 * receives per-workgroup sums of |sampled - reprojected| for mvsn_occlusion_mask.
 *   K (B,4,4) of this pyramid level   T_other_in_this (B,4,4)   idepth, other_idepth (B,rows,cols)
 *   other_mask (B,rows,cols) u8 or NULL
 *   -> idepth_in_other, other_sampled (B,rows,cols)  other_mask_sampled u8 or NULL  invalid u8  uv (B,rows,cols,2) or NULL
 * mvsn_occlusion_mask: get_occlusion_mask (losses.py:42-82): (sampled - reprojected) > mean|sampled - reprojected|
 * of the image, or invalid.
 * mvsn_masked_l1: loss (+)= mean |a - b| over the elements with neither skip flag set (the two l1_loss terms of
 * left_right_idepthmap_consistency_losses, losses.py:137-157); NaN when nothing is selected, as the reference.
 * ------------------------------------------------------------------------------------------- */
int mvsn_idepth_reproject_blocks(int pixels);
int mvsn_idepth_reproject(const float *K, const float *T_other_in_this, const float *idepth, const float *other_idepth,
                          const uint8_t *other_mask, int batch, int rows, int cols, float *idepth_in_other,
                          float *other_sampled, uint8_t *other_mask_sampled, uint8_t *invalid, float *uv,
                          float *absdiff_partials, mvsn_stream_t stream);
int mvsn_occlusion_mask(const float *idepth_in_other, const float *other_sampled, const uint8_t *invalid,
                        const float *absdiff_partials, int batch, int pixels, uint8_t *mask, mvsn_stream_t stream);
int mvsn_masked_l1(const float *a, const float *b, const uint8_t *skip_a, const uint8_t *skip_b, long n, int accumulate,
                   float *loss, mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Plane-resident residual tower on the 16 x 32 coarse grid: ONE persistent workgroup per sample keeps the 32-channel
 * activation planes in LDS and runs  [head conv + GroupNorm + LeakyReLU] -> n_blocks x (x + LReLU(GN(conv3x3_dilated(x))))
 * -> tail  in a single launch.  Replaces, at level 4 of 512 x 256 frames,
 *   - IDepthmapRefiner.forward (multi_view_stereonet.py:468-484) with the gain trick of :607-611
 *     (head over [image 3 | features 32 | prior * fx], dilations 1,2,4,8,1,1, tail_mode 1: relu(prior*fx + conv_final)/fx),
 *   - the residual stack + conv_final of FeatureNetwork.forward (:121-129; no head, tail_mode 0),
 * i.e. the 15-21 launches of mvsn_conv_forward / mvsn_groupnorm_* the launch-per-layer form needs there.
 *   in[b] (n_b, channels[b], 16, 32): up to three channel blocks, sample n reads block b at n % sample_mod[b]
 *   block_scale (optional): block `scale_block` is multiplied by block_scale[n % scale_mod] while it is loaded
 *   weights: the layers' Winograd-transformed weights (mvsn_conv_pack_weights, MVSN_CONV_FP32_WINO) re-ordered per
 *            k-step as [cout tile][xi row][lane][xi column], layer after layer: head (9 k-steps), blocks (8 each),
 *            and for tail_mode 0 the final 32 -> 32 layer
 *   params:  [bias 32 | gamma 32 | beta 32] per layer (head, blocks), then tail_mode 0: bias 32;
 *            tail_mode 1: the 32 -> 1 layer's weight (32 x 9) and bias (1)
 *   out: tail_mode 0 (n, 32, 16, 32); tail_mode 1 (n, 1, 16, 32) with prior (n, 16, 32), fx[n % fx_mod]
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float *in[3];
  int channels[3], sample_mod[3];
  const float *block_scale;
  int scale_mod, scale_block;
  int head_chunks, n_blocks;
  int dilation[6];
  const float *weights, *params;
  int tail_mode;
  const float *prior, *fx;
  int fx_mod;
  float *out;
} mvsn_tower_desc;
int mvsn_tower_16x32(const mvsn_tower_desc *desc, int n_samples, mvsn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Depth metrics of a batch on the device: replaces the per-image host loop of test.py:210-235 + get_depth_prediction_metrics
 * (test.py:41-71).  Per image: depth_est = idepth_est / baseline, inverted where positive (:210-214); selected pixels =
 * truth in (min_depth, max_depth) AND estimate in (min_depth, max_depth) (:221,:232); per-pixel values in fp32 as numpy
 * forms them, sums in double in a fixed order.
 *   idepth_est, depth_true (B, pixels) fp32 (depth_true in metric units)   baseline (B)
 *   partials: B * mvsn_depth_metrics_blocks(pixels) * 9 doubles of scratch
 *   rows (B, 9) doubles: {n_truth, n_selected, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3}; NaN metrics when nothing is
 *   selected (numpy's mean of an empty selection); the caller drops images with n_truth == 0 (:223-225).
 * ------------------------------------------------------------------------------------------- */
int mvsn_depth_metrics_blocks(long pixels);
int mvsn_depth_metrics(const float *idepth_est, const float *depth_true, const float *baseline, int batch, long pixels,
                       float min_depth, float max_depth, double *partials, double *rows, mvsn_stream_t stream);

/* Tensor plumbing of the forward as library calls (so that a whole forward is a replayable list of C calls and nothing
 * else): a device-to-device copy on the stream (the torch.cat / repeat of poses, intrinsics and coarse source images,
 * multi_view_stereonet.py:553,:587-592) and dst[i] = src[i * stride] (the focal lengths K[:, 0, 0], :607). */
int mvsn_copy(void *dst, const void *src, size_t nbytes, mvsn_stream_t stream);
/* `count` independent device-to-device copies, eight per launch, every buffer a plain pointer argument of the kernel (the
 * input / output copies of a replayed forward plan: what torch._foreach_copy_ did with pointers the runtime cannot see) */
int mvsn_copy_many(void *const *dst, const void *const *src, const size_t *nbytes, int count, mvsn_stream_t stream);
/* fx[l * batch + b] = K_pyr[l][b][0][0] for every pyramid level in one launch (K_pyr: host array of `levels` <= 8 device
 * pointers to (batch, 4, 4) intrinsics) */
int mvsn_gather_focal(const float *const *K_pyr, int levels, int batch, float *fx, mvsn_stream_t stream);
int mvsn_gather_strided(const float *src, int count, long stride, float *dst, mvsn_stream_t stream);

/* Test hook for the banded chain (process-wide, 0 = off): bit 1 = the last band of every chain never runs (what a
 * shared device can do to a launch whose workgroups must be co-resident); bits 8.. = log2 of the spin limit of a
 * hand-off (default 2^21).  With it the other bands time out: the status word is set, the cost slice carries a NaN (so
 * the depth maps of that forward are NaN), nothing hangs.  Bit 2 (value 4) pins the 4-band plan on 16x32 (the 8-band
 * half-split plan is compared with it bit for bit); set it before asking for the workspace size. */
int mvsn_debug_set_band_flags(int flags);

/* Device self-test of the MFMA fragment mapping the conv kernels rely on (A = 16x4, B = 4x16
 * fp32, asymmetric operands); returns 0 when the on-device result matches the scalar product. */
int mvsn_selftest_mfma(mvsn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVSN_HIP_H */
