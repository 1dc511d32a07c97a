// The incremental feature chain (mvsn_incremental_cost_volume) in its STEPWISE form: one plane per round of
// full-chip launches -- warp of the previous plane's features, the refiner's three 3x3 convolutions as Winograd
// launches (GroupNorm statistics from the producing launch, LeakyReLU(GroupNorm(.)) applied on load), the
// two-raw-tensor pass of the residual block, and a tail that forms F_d = moved + delta and writes the cost slice.
//
// The fused forms run one workgroup per chain: at 16x32 (Winograd plan) a step is 40 us and nothing beats them, but a
// 30x40 or 32x64 plane (DeMoN 640x480, 1024x512 frames) neither fits one CU's LDS nor its matrix pipe -- the direct
// fused kernel needs 250 / 440 us per step whatever the number of chains below the CU count.  Here a step is eight
// dependent launches (~5 us each at these sizes) that use every CU: 30x40, 1 chain: 23.6 -> ~4 ms per 96 planes;
// 32 chains: 27.7 -> ~8 ms.  MVSN_CHAIN_AUTO takes this form when the coarse grid has no Winograd chain plan, the
// grid suits the Winograd convolutions (cols % 4 == 0) and fewer chains than CUs are in flight.
// Reference semantics: multi_view_stereonet.py:270-300 (incremental extractor), :424-440 (FeatureRefiner),
// :587-592 (cost volume); same arithmetic as the fused forms up to the summation order inside the convolutions.
#include "mvsn_common.h"
#include "mvsn_chain.h"
#include "mvsn_conv_wino.h"

namespace mvsn {

int warp_launch(const float *image, const float *H, int h_bstride, int batch, int channels, int n_planes, int rows,
                int cols, float *volume, uint8_t *mask, hipStream_t stream);   // mvsn_warp.hip

// F_d = moved + delta (delta null: plane 0), cost[n, c, d] = keep * |FL[n % B, c] - keep * F_d|, four pixels per thread
__global__ __launch_bounds__(256) void chain_step_tail_kernel(const float *__restrict__ moved,
                                                              const float *__restrict__ delta,
                                                              const float *__restrict__ fl,
                                                              const uint8_t *__restrict__ mask, size_t mask_bstride,
                                                              int B, int P, size_t vol_cstride, long total4,
                                                              float *__restrict__ f_out, float *__restrict__ cost,
                                                              float *__restrict__ fvol) {
  const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 >= total4) return;
  const long e = i4 * 4;
  const int p = (int)(e % P);
  const long nc = e / P;
  const int c = (int)(nc & 31);
  const long n = nc >> 5;
  floatx4 f = *reinterpret_cast<const floatx4 *>(moved + e);
  if (delta) f += *reinterpret_cast<const floatx4 *>(delta + e);
  if (f_out) *reinterpret_cast<floatx4 *>(f_out + e) = f;
  const uint8_t *m = mask + n * mask_bstride + p;
  const floatx4 l = *reinterpret_cast<const floatx4 *>(fl + ((n % B) * 32 + c) * (long)P + p);
  floatx4 fm, cs;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float keep = m[k] ? 0.0f : 1.0f;
    fm[k] = f[k] * keep;                       // (NaN * 0 stays NaN, as the reference's multiply)
    cs[k] = keep * fabsf(l[k] - fm[k]);
  }
  const size_t o = ((size_t)n * 32 + c) * vol_cstride + p;
  __builtin_nontemporal_store(cs, reinterpret_cast<floatx4 *>(cost + o));
  if (fvol) __builtin_nontemporal_store(fm, reinterpret_cast<floatx4 *>(fvol + o));
}

static mvsn_conv_desc step_desc(int n, int cin, int rows, int cols) {
  mvsn_conv_desc d;
  d.n = n, d.c_in = cin, d.c_out = 32, d.depth = 1, d.rows = rows, d.cols = cols;
  d.kd = 1, d.kh = 3, d.kw = 3, d.stride = 1, d.dilation = 1, d.precision = MVSN_CONV_FP32_WINO;
  return d;
}

bool chain_steps_supported(int rows, int cols) {
  WinoGeom g;
  const mvsn_conv_desc d = step_desc(1, 35, rows, cols);
  return rows > 0 && cols > 0 && cols % 4 == 0 && wino_geom(&d, &g);
}

static size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t chain_steps_workspace_bytes(int n_chains, int D, int rows, int cols) {
  if (!chain_steps_supported(rows, cols) || n_chains <= 0 || D <= 0) return 0;
  const size_t N = n_chains, P = (size_t)rows * cols;
  WinoGeom g;
  const mvsn_conv_desc d = step_desc(n_chains, 32, rows, cols);
  wino_geom(&d, &g);
  const size_t records = (size_t)g.tiles * 32;
  return up256(N * 3 * D * P * 4) + 6 * up256(N * 32 * P * 4) + 2 * up256(N * records * 12 * 4) + 2 * up256(N * 8 * 4) +
         up256(N * P);
}

int chain_steps_launch(const ChainArgs &a, int n_chains, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  const int N = n_chains, D = a.D, rows = a.rows, cols = a.cols, P = rows * cols;
  if (!chain_steps_supported(rows, cols)) {
    set_error("mvsn_incremental_cost_volume: the stepwise form needs cols %% 4 == 0 (got %dx%d)", rows, cols);
    return MVSN_E_BADARG;
  }
  const size_t need = chain_steps_workspace_bytes(N, D, rows, cols);
  if (!workspace || workspace_bytes < need || ((size_t)workspace & 15) != 0) {
    set_error("mvsn_incremental_cost_volume: stepwise form needs a 16-byte aligned workspace of %zu bytes", need);
    return MVSN_E_WORKSPACE;
  }
  mvsn_conv_desc d0 = step_desc(N, 35, rows, cols), d1 = step_desc(N, 32, rows, cols);
  WinoGeom g0, g1;
  if (!wino_geom(&d0, &g0) || !wino_geom(&d1, &g1)) return MVSN_E_BADARG;
  const int records = g1.tiles * 32;
  char *w = (char *)workspace;
  auto carve = [&](size_t bytes) { char *p = w; w += up256(bytes); return p; };
  float *img = (float *)carve((size_t)N * 3 * D * P * 4);
  float *moved = (float *)carve((size_t)N * 32 * P * 4);
  float *fa = (float *)carve((size_t)N * 32 * P * 4), *fb = (float *)carve((size_t)N * 32 * P * 4);
  float *r0 = (float *)carve((size_t)N * 32 * P * 4), *r1 = (float *)carve((size_t)N * 32 * P * 4);
  float *dl = (float *)carve((size_t)N * 32 * P * 4);
  float *part0 = (float *)carve((size_t)N * records * 12 * 4), *part1 = (float *)carve((size_t)N * records * 12 * 4);
  float *st0 = (float *)carve((size_t)N * 8 * 4), *st1 = (float *)carve((size_t)N * 8 * 4);
  uint8_t *scratch_mask = (uint8_t *)carve((size_t)N * P);

  // parameters inside the packed buffer (mvsn_pack_feature_refiner)
  const float *sp = a.packed + CH_W0_FLOATS + 2 * CH_W1_FLOATS;
  const float *bias0 = sp, *gn0w = sp + 32, *gn0b = sp + 64, *bias1 = sp + 96, *gn1w = sp + 128, *gn1b = sp + 160,
              *bias2 = sp + 192;
  const float *U0 = a.packed + CH_STEPS_OFFSET, *U1 = U0 + CS_U0_FLOATS, *U2 = U1 + CS_U1_FLOATS;

  const long total4 = (long)N * 32 * P / 4;
  const dim3 tgrid((unsigned)((total4 + 255) / 256));
  auto tail = [&](int d, const float *mv, const float *delta, float *f_out) {
    hipLaunchKernelGGL(chain_step_tail_kernel, tgrid, dim3(256), 0, stream, mv, delta, a.fl, a.mask + (size_t)d * P,
                       (size_t)D * P, a.B, P, (size_t)D * P, total4, f_out, a.cost + (size_t)d * P,
                       a.fvol ? a.fvol + (size_t)d * P : (float *)nullptr);
    return check_launch("mvsn_incremental_cost_volume(stepwise tail)");
  };
  // the source image on every plane + the mask volume (the output itself)
  if (int rc = warp_launch(a.src, a.H, 9 * D, N, 3, D, rows, cols, img, a.mask, stream)) return rc;
  if (int rc = tail(0, a.f0, nullptr, nullptr)) return rc;
  const float *prev = a.f0;
  for (int d = 1; d < D; ++d) {
    float *cur = (d & 1) ? fa : fb;
    if (int rc = warp_launch(prev, a.Hinc + (size_t)d * 9, 9 * D, N, 32, 1, rows, cols, moved, scratch_mask, stream))
      return rc;
    WinoBlocks blk;
    blk.cb0 = 3, blk.cb1 = 32, blk.in1 = moved, blk.in2 = moved;
    blk.bs0 = (size_t)3 * D * P, blk.cs0 = (size_t)D * P;        // image plane d of (N, 3, D, P)
    if (int rc = wino_launch(g0, img + (size_t)d * P, U0, bias0, nullptr, nullptr, nullptr, r0, part0, stream, &blk))
      return rc;
    if (int rc = mvsn_groupnorm_finalize(part0, N, records, st0, stream)) return rc;
    if (int rc = wino_launch(g1, r0, U1, bias1, st0, gn0w, gn0b, r1, part1, stream)) return rc;
    if (int rc = mvsn_groupnorm_finalize(part1, N, records, st1, stream)) return rc;
    if (int rc = mvsn_groupnorm_lrelu_add2(r1, st1, gn1w, gn1b, r0, st0, gn0w, gn0b, N, P, r1, stream)) return rc;
    if (int rc = wino_launch(g1, r1, U2, bias2, nullptr, nullptr, nullptr, dl, nullptr, stream)) return rc;
    if (int rc = tail(d, moved, dl, cur)) return rc;
    prev = cur;
  }
  return 0;
}

}  // namespace mvsn
