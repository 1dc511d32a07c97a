// Depth metrics on the device (see include/mvsn_hip.h: mvsn_depth_metrics).
//
// The reference's evaluation loop (test.py:188-280) copies two full-resolution maps per image to the host and forms
// the KITTI-style metrics in numpy, one image at a time (test.py:41-71, 210-235).  At a few thousand depth maps per
// second that loop, not the forward, sets the evaluation rate.  Here the per-image work is one HBM pass: the network's
// idepth map is turned into metric depth exactly as test.py:210-214 does (idepth / baseline, inverted where positive;
// every per-pixel value in fp32 with IEEE division, as numpy evaluates float32 arrays), both validity masks are applied
// (truth inside (min, max) and estimate inside (min, max), test.py:221-232), and seven sums + two counts per image
// leave the kernel.  Sums are accumulated in double in a fixed order (thread, wave, block): deterministic, and within
// 1e-7 relative of numpy's float32 pairwise means.  Nine doubles per image stay on the device for the all-gather.
//
// Two launches without any inter-workgroup synchronisation: MT_BLOCKS workgroups per image write partial records,
// one workgroup per image adds them in order and forms the means.
#include "mvsn_common.h"

namespace mvsn {

constexpr int MT_THREADS = 256;
constexpr int MT_VALUES = 9;   // n_truth, n_valid, abs_rel, sq_rel, sq, log_sq, a1, a2, a3

__global__ __launch_bounds__(MT_THREADS) void depth_metrics_partial_kernel(
    const float *__restrict__ idepth, const float *__restrict__ depth_true, const float *__restrict__ baseline,
    long pixels, float min_depth, float max_depth, int blocks, double *__restrict__ partials) {
  const int b = blockIdx.y, blk = blockIdx.x;
  const float *ie = idepth + (size_t)b * pixels;
  const float *tt = depth_true + (size_t)b * pixels;
  const float base = baseline[b];
  double acc[MT_VALUES];
#pragma unroll
  for (int k = 0; k < MT_VALUES; ++k) acc[k] = 0.0;
  const long per = (pixels + blocks - 1) / blocks;
  const long lo = (long)blk * per, hi = lo + per < pixels ? lo + per : pixels;
  for (long i = lo + threadIdx.x; i < hi; i += MT_THREADS) {
    const float t = __builtin_nontemporal_load(tt + i);
    const float scaled = __builtin_nontemporal_load(ie + i) / base;     // test.py:210-211
    const float e = scaled > 0.0f ? 1.0f / scaled : scaled;             // test.py:212
    const bool truth = t > min_depth && t < max_depth;                  // test.py:221
    if (truth) acc[0] += 1.0;
    if (truth && e > min_depth && e < max_depth) {                      // test.py:232
      const float diff = t - e;
      const float r0 = t / e, r1 = e / t;
      const float ratio = r0 > r1 ? r0 : r1;
      const float lg = logf(t) - logf(e);
      acc[1] += 1.0;
      acc[2] += (double)(fabsf(diff) / t);
      acc[3] += (double)((diff * diff) / t);
      acc[4] += (double)(diff * diff);
      acc[5] += (double)(lg * lg);
      acc[6] += ratio < 1.25f ? 1.0 : 0.0;
      acc[7] += ratio < 1.5625f ? 1.0 : 0.0;
      acc[8] += ratio < 1.953125f ? 1.0 : 0.0;
    }
  }
  __shared__ double sred[MT_THREADS / 64][MT_VALUES];
#pragma unroll
  for (int k = 0; k < MT_VALUES; ++k) {
    double v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < MT_VALUES) {
    double v = 0.0;
    for (int w = 0; w < MT_THREADS / 64; ++w) v += sred[w][threadIdx.x];
    partials[((size_t)b * blocks + blk) * MT_VALUES + threadIdx.x] = v;
  }
}

// rows[b] = {n_truth, n_valid, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3}; an image without a selected pixel gets
// NaN metrics (numpy's mean over an empty selection), the caller skips images with n_truth == 0 (test.py:223-225)
__global__ void depth_metrics_final_kernel(const double *__restrict__ partials, int blocks, double *__restrict__ rows) {
  const int b = blockIdx.x, k = threadIdx.x;
  if (k >= MT_VALUES) return;
  double v = 0.0;
  for (int i = 0; i < blocks; ++i) v += partials[((size_t)b * blocks + i) * MT_VALUES + k];
  double n = 0.0;
  for (int i = 0; i < blocks; ++i) n += partials[((size_t)b * blocks + i) * MT_VALUES + 1];
  double out = v;
  if (k >= 2) {
    out = v / n;
    if (k == 4 || k == 5) out = sqrt(out);
  }
  rows[(size_t)b * MT_VALUES + k] = out;
}

}  // namespace mvsn

extern "C" int mvsn_depth_metrics_blocks(long pixels) {
  if (pixels <= 0) return 0;
  const long b = (pixels + 8191) / 8192;     // >= 32 pixels per thread
  return (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
}

extern "C" int mvsn_depth_metrics(const float *idepth_est, const float *depth_true, const float *baseline, int batch,
                                  long pixels, float min_depth, float max_depth, double *partials, double *rows,
                                  mvsn_stream_t stream) {
  MVSN_REQUIRE(idepth_est && depth_true && baseline && partials && rows, MVSN_E_BADARG, "mvsn_depth_metrics: null pointer");
  MVSN_REQUIRE(batch > 0 && batch <= 65535 && pixels > 0, MVSN_E_BADARG, "mvsn_depth_metrics: bad sizes");
  const int blocks = mvsn_depth_metrics_blocks(pixels);
  hipLaunchKernelGGL(mvsn::depth_metrics_partial_kernel, dim3(blocks, batch), dim3(mvsn::MT_THREADS), 0,
                     (hipStream_t)stream, idepth_est, depth_true, baseline, pixels, min_depth, max_depth, blocks, partials);
  if (int rc = mvsn::check_launch("mvsn_depth_metrics")) return rc;
  hipLaunchKernelGGL(mvsn::depth_metrics_final_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, partials, blocks,
                     rows);
  return mvsn::check_launch("mvsn_depth_metrics(final)");
}
