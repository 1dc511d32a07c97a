// Small HBM-bound kernels either side of the cost-volume regulariser: soft-argmin, bilinear
// resize of idepth maps and hypothesis masks, multi-source fusion.
#include "mvsn_common.h"

namespace mvsn {

// ---- soft-argmin over D -----------------------------------------------------------------
// cost (N,D,P): one thread per pixel walks D with stride P, so every load is a coalesced row
// segment; max-subtracted softmax of -cost (what torch.softmin computes).
__global__ __launch_bounds__(256) void soft_argmin_kernel(const float *__restrict__ cost,
                                                          const float *__restrict__ samples, int D, int P,
                                                          float *__restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= P) return;
  const float *c = cost + (size_t)n * D * P + p;
  const float *s = samples + (size_t)n * D;
  float m = -INFINITY;
  for (int d = 0; d < D; ++d) m = fmaxf(m, -c[(size_t)d * P]);
  float den = 0.0f, num = 0.0f;
  for (int d = 0; d < D; ++d) {
    float e = expf(-c[(size_t)d * P] - m);
    den += e;
    num += e * s[d];
  }
  out[(size_t)n * P + p] = num / den;
}

// ---- bilinear resize, align_corners=False ------------------------------------------------
// src = (dst + 0.5) * (in/out) - 0.5, clamped at 0; the +1 tap is clamped to the last index
// (ATen upsample_bilinear2d, area_pixel_compute_source_index).
struct ResizeTap {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ ResizeTap resize_tap(int dst, int in_size, int out_size) {
  float scale = (float)in_size / (float)out_size;
  float src = ((float)dst + 0.5f) * scale - 0.5f;
  if (src < 0.0f) src = 0.0f;
  ResizeTap t;
  t.i0 = (int)src;
  if (t.i0 > in_size - 1) t.i0 = in_size - 1;
  t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.0f - t.l1;
  return t;
}

template <typename TIn, typename TOut, bool THRESH>
__global__ __launch_bounds__(256) void upsample_kernel(const TIn *__restrict__ in, int hin, int win, int hout,
                                                       int wout, TOut *__restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const size_t plane = blockIdx.z;  // n*C + c
  if (x >= wout) return;
  ResizeTap ty = resize_tap(y, hin, hout);
  ResizeTap tx = resize_tap(x, win, wout);
  const TIn *ip = in + plane * hin * win;
  float v00 = (float)ip[ty.i0 * win + tx.i0], v01 = (float)ip[ty.i0 * win + tx.i1];
  float v10 = (float)ip[ty.i1 * win + tx.i0], v11 = (float)ip[ty.i1 * win + tx.i1];
  float v = ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11);
  if (THRESH)
    out[(plane * hout + y) * wout + x] = (TOut)(v > 0.5f ? 1 : 0);
  else
    out[(plane * hout + y) * wout + x] = (TOut)v;
}

// ---- multi-source fusion -------------------------------------------------------------------
__global__ __launch_bounds__(256) void fuse_idepth_kernel(const float *__restrict__ raw,
                                                          const float *__restrict__ refined,
                                                          const float *__restrict__ baseline, int S, int B, int P,
                                                          int alias, float *__restrict__ raw_out,
                                                          float *__restrict__ refined_out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= P) return;
  float sr = 0.0f, sf = 0.0f;
  for (int s = 0; s < S; ++s) {
    const size_t n = (size_t)s * B + b;
    const float base = baseline[n];
    float r = raw[n * P + p] / base;
    float f;
    if (alias) {  // reference quirk: refined aliases raw, which is divided in place twice
      r = r / base;
      f = r;
    } else {
      f = refined[n * P + p] / base;
    }
    sr += r;
    sf += f;
  }
  raw_out[(size_t)b * P + p] = sr / (float)S;
  refined_out[(size_t)b * P + p] = sf / (float)S;
}

__global__ __launch_bounds__(256) void fuse_mask_kernel(const uint8_t *__restrict__ mask, int S, int B, size_t DP,
                                                        uint8_t *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= DP) return;
  float acc = 0.0f;
  for (int s = 0; s < S; ++s) acc += (float)mask[((size_t)s * B + b) * DP + i];
  out[(size_t)b * DP + i] = (acc / (float)S) > 0.5f ? 1 : 0;
}

}  // namespace mvsn

extern "C" int mvsn_soft_argmin(const float *cost, const float *idepth_samples, int n, int D, int pixels,
                                float *idepth, mvsn_stream_t stream) {
  MVSN_REQUIRE(cost && idepth_samples && idepth, MVSN_E_BADARG, "mvsn_soft_argmin: null pointer");
  MVSN_REQUIRE(n > 0 && D > 0 && pixels > 0 && n <= 65535, MVSN_E_BADARG, "mvsn_soft_argmin: bad sizes");
  hipLaunchKernelGGL(mvsn::soft_argmin_kernel, dim3((pixels + 255) / 256, n), dim3(256), 0, (hipStream_t)stream,
                     cost, idepth_samples, D, pixels, idepth);
  return mvsn::check_launch("mvsn_soft_argmin");
}

extern "C" int mvsn_upsample_bilinear(const float *in, int n, int channels, int rows_in, int cols_in, int rows_out,
                                      int cols_out, float *out, mvsn_stream_t stream) {
  MVSN_REQUIRE(in && out, MVSN_E_BADARG, "mvsn_upsample_bilinear: null pointer");
  MVSN_REQUIRE(n > 0 && channels > 0 && rows_in > 0 && cols_in > 0 && rows_out > 0 && cols_out > 0, MVSN_E_BADARG,
               "mvsn_upsample_bilinear: bad sizes");
  MVSN_REQUIRE((long)n * channels <= 65535 && rows_out <= 65535, MVSN_E_TOOLARGE, "mvsn_upsample_bilinear: grid");
  dim3 grid((cols_out + 255) / 256, rows_out, n * channels);
  hipLaunchKernelGGL((mvsn::upsample_kernel<float, float, false>), grid, dim3(256), 0, (hipStream_t)stream, in,
                     rows_in, cols_in, rows_out, cols_out, out);
  return mvsn::check_launch("mvsn_upsample_bilinear");
}

extern "C" int mvsn_upsample_mask(const uint8_t *in, int n, int channels, int rows_in, int cols_in, int rows_out,
                                  int cols_out, uint8_t *out, mvsn_stream_t stream) {
  MVSN_REQUIRE(in && out, MVSN_E_BADARG, "mvsn_upsample_mask: null pointer");
  MVSN_REQUIRE(n > 0 && channels > 0 && rows_in > 0 && cols_in > 0 && rows_out > 0 && cols_out > 0, MVSN_E_BADARG,
               "mvsn_upsample_mask: bad sizes");
  MVSN_REQUIRE((long)n * channels <= 65535 && rows_out <= 65535, MVSN_E_TOOLARGE, "mvsn_upsample_mask: grid");
  dim3 grid((cols_out + 255) / 256, rows_out, n * channels);
  hipLaunchKernelGGL((mvsn::upsample_kernel<uint8_t, uint8_t, true>), grid, dim3(256), 0, (hipStream_t)stream, in,
                     rows_in, cols_in, rows_out, cols_out, out);
  return mvsn::check_launch("mvsn_upsample_mask");
}

extern "C" int mvsn_fuse_sources(const float *raw, const float *refined, const float *baseline, const uint8_t *mask,
                                 int n_sources, int batch, int D, int pixels, int refined_aliases_raw,
                                 float *raw_out, float *refined_out, uint8_t *mask_out, mvsn_stream_t stream) {
  MVSN_REQUIRE(raw && baseline && mask && raw_out && refined_out && mask_out, MVSN_E_BADARG,
               "mvsn_fuse_sources: null pointer");
  MVSN_REQUIRE(refined || refined_aliases_raw, MVSN_E_BADARG, "mvsn_fuse_sources: refined is null");
  MVSN_REQUIRE(n_sources > 0 && batch > 0 && batch <= 65535 && D > 0 && pixels > 0, MVSN_E_BADARG,
               "mvsn_fuse_sources: bad sizes");
  hipLaunchKernelGGL(mvsn::fuse_idepth_kernel, dim3((pixels + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream,
                     raw, refined, baseline, n_sources, batch, pixels, refined_aliases_raw, raw_out, refined_out);
  int rc = mvsn::check_launch("mvsn_fuse_sources(idepth)");
  if (rc) return rc;
  const size_t DP = (size_t)D * pixels;
  hipLaunchKernelGGL(mvsn::fuse_mask_kernel, dim3((unsigned)((DP + 255) / 256), batch), dim3(256), 0,
                     (hipStream_t)stream, mask, n_sources, batch, DP, mask_out);
  return mvsn::check_launch("mvsn_fuse_sources(mask)");
}
