// Small HBM-bound kernels either side of the cost-volume regulariser: soft-argmin, bilinear
// resize of idepth maps and hypothesis masks, multi-source fusion.
#include <type_traits>

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
  // (sixteen independent loads in flight per round: at batch 1 the kernel is one latency chain per pixel)
  float m = -INFINITY;
  int d0 = 0;
  for (; d0 + 16 <= D; d0 += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = c[(size_t)(d0 + j) * P];
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, -v[j]);
  }
  for (int d = d0; d < D; ++d) m = fmaxf(m, -c[(size_t)d * P]);
  float den = 0.0f, num = 0.0f;
  for (d0 = 0; d0 + 16 <= D; d0 += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = c[(size_t)(d0 + j) * P];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float e = expf(-v[j] - m);
      den += e;
      num += e * s[d0 + j];
    }
  }
  for (int d = d0; d < D; ++d) {
    float e = expf(-c[(size_t)d * P] - m);
    den += e;
    num += e * s[d];
  }
  out[(size_t)n * P + p] = num / den;
}

// ---- elementwise steps of the flag branches -----------------------------------------------------
__global__ __launch_bounds__(256) void channel_l2_norm_kernel(const float *__restrict__ x, int channels, long P,
                                                              float *__restrict__ out) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float *xp = x + (size_t)blockIdx.y * channels * P + p;
  float acc = 0.0f;
  for (int c = 0; c < channels; ++c) {
    const float v = xp[(size_t)c * P];
    acc += v * v;
  }
  out[(size_t)blockIdx.y * P + p] = sqrtf(acc);
}

// EPI false: out = prior * fx;  EPI true: out = relu(prior * fx + delta) / fx
template <bool EPI>
__global__ __launch_bounds__(256) void idepth_gain_kernel(const float *__restrict__ prior, const float *__restrict__ fx,
                                                          const float *__restrict__ delta, long P,
                                                          float *__restrict__ out) {
#pragma clang fp contract(off)   // prior*fx and the add are two separately rounded ATen ops in the reference
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float f = fx[blockIdx.y];
  const size_t i = (size_t)blockIdx.y * P + p;
  const float scaled = prior[i] * f;
  if constexpr (EPI) out[i] = relu_nan(scaled + delta[i]) / f;
  else out[i] = scaled;
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

// The coarse-to-fine step's prior: the upsampled idepth map AND its fx-scaled copy (the refiner's input channel,
// multi_view_stereonet.py:607-611) from one pass -- the scaling no longer costs a launch per level.
__global__ __launch_bounds__(256) void upsample_gain_kernel(const float *__restrict__ in, const float *__restrict__ gain,
                                                            int hin, int win, int hout, int wout,
                                                            float *__restrict__ out, float *__restrict__ out_scaled) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const size_t plane = blockIdx.z;
  if (x >= wout) return;
  ResizeTap ty = resize_tap(y, hin, hout);
  ResizeTap tx = resize_tap(x, win, wout);
  const float *ip = in + plane * hin * win;
  float v00 = ip[ty.i0 * win + tx.i0], v01 = ip[ty.i0 * win + tx.i1];
  float v10 = ip[ty.i1 * win + tx.i0], v11 = ip[ty.i1 * win + tx.i1];
  float v = ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11);
  out[(plane * hout + y) * wout + x] = v;
  out_scaled[(plane * hout + y) * wout + x] = v * gain[plane];
}

// Mask variant: each thread produces 16 consecutive output columns of one row and stores them as
// one 128-bit word (the D-channel masks are the largest HBM stream of a forward; byte stores waste
// most of every write transaction).  1-D grid over (plane, row, 16-column group).
__global__ __launch_bounds__(256) void upsample_mask16_kernel(const uint8_t *__restrict__ in, int hin, int win,
                                                              int hout, int wout, int groups_per_row,
                                                              size_t total_groups, uint8_t *__restrict__ out) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total_groups) return;
  const int gx = (int)(gid % groups_per_row);
  const size_t rowid = gid / groups_per_row;  // plane * hout + y
  const int y = (int)(rowid % hout);
  const size_t plane = rowid / hout;
  const int x16 = gx * 16;
  ResizeTap ty = resize_tap(y, hin, hout);
  const uint8_t *r0 = in + (plane * hin + ty.i0) * win;
  const uint8_t *r1 = in + (plane * hin + ty.i1) * win;
  uint32_t w[4] = {0, 0, 0, 0};
  uint8_t vals[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int x = x16 + k < wout ? x16 + k : wout - 1;
    ResizeTap tx = resize_tap(x, win, wout);
    const float v00 = (float)r0[tx.i0], v01 = (float)r0[tx.i1], v10 = (float)r1[tx.i0], v11 = (float)r1[tx.i1];
    const float v = ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11);
    vals[k] = v > 0.5f ? 1 : 0;
    w[k >> 2] |= (uint32_t)vals[k] << (8 * (k & 3));
  }
  uint8_t *orow = out + rowid * wout;
  if (x16 + 15 < wout && (((size_t)(orow + x16)) & 15) == 0) {
    *reinterpret_cast<uint4 *>(orow + x16) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (x16 + k < wout) orow[x16 + k] = vals[k];
  }
}

// Exact 2x mask upsampling.  The masks are boolean (bytes 0 / 1): with the 2x bilinear weights an output pixel is
// (9 a + 3 b + 3 c + d) / 16 of its nearest input pixel a and three neighbours, and "> 0.5" of that is a itself
// (a = 1: at least 9/16; a = 0: at most 7/16), also where the clamped edge taps coincide -- so the float -> bilinear
// -> threshold of MaskUpsampler (:389-396) is a 2 x 2 replication.  One thread: 8 input bytes -> 2 rows x 16 bytes.
__global__ __launch_bounds__(256) void upsample_mask2x_kernel(const uint8_t *__restrict__ in, int hin, int win,
                                                              int groups_per_row, size_t total_groups,
                                                              uint8_t *__restrict__ out) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (plane, input row, 8-column group)
  if (gid >= total_groups) return;
  const int gx = (int)(gid % groups_per_row);
  const size_t rowid = gid / groups_per_row;                           // plane * hin + y
  const uint2 w = *reinterpret_cast<const uint2 *>(in + rowid * win + gx * 8);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 o;
  o[0] = __builtin_amdgcn_perm(w.x, w.x, 0x01010000u);   // bytes b0 b0 b1 b1
  o[1] = __builtin_amdgcn_perm(w.x, w.x, 0x03030202u);   //       b2 b2 b3 b3
  o[2] = __builtin_amdgcn_perm(w.y, w.y, 0x01010000u);
  o[3] = __builtin_amdgcn_perm(w.y, w.y, 0x03030202u);
  uint8_t *dst = out + (rowid * 2) * (size_t)(win * 2) + gx * 16;
  __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(dst));
  __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(dst + win * 2));
}

// ---- area (adaptive average) downsample: one pyramid level ---------------------------------
// out size ((h+1)/2, (w+1)/2); window [floor(i*in/out), ceil((i+1)*in/out)) as ATen's
// adaptive_avg_pool2d (= interpolate(mode="area"), utils/image_utils.py:118-126): an exact 2x2
// mean for even sizes, overlapping windows for odd ones.
__global__ __launch_bounds__(256) void area_downsample_kernel(const float *__restrict__ in, int hin, int win,
                                                              int hout, int wout, float *__restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const size_t plane = blockIdx.z;
  if (x >= wout) return;
  const int y0 = (y * hin) / hout, y1 = ((y + 1) * hin + hout - 1) / hout;
  const int x0 = (x * win) / wout, x1 = ((x + 1) * win + wout - 1) / wout;
  const float *ip = in + plane * hin * win;
  float sum = 0.0f;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) sum += ip[yy * win + xx];
  out[(plane * hout + y) * wout + x] = sum / (float)((y1 - y0) * (x1 - x0));
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

namespace mvsn {
__global__ void gather_strided_kernel(const float *__restrict__ src, int count, long stride, float *__restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = src[(size_t)i * stride];
}
}  // namespace mvsn

namespace mvsn {
struct FocalSrc {
  const float *K[8];
};
__global__ void gather_focal_kernel(FocalSrc src, int levels, int batch, float *__restrict__ out, MVSN_VIS10) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < levels * batch) out[i] = src.K[i / batch][(size_t)(i % batch) * 16];
}
}  // namespace mvsn

extern "C" int mvsn_gather_focal(const float *const *K_pyr, int levels, int batch, float *fx, mvsn_stream_t stream) {
  MVSN_REQUIRE(K_pyr && fx && levels >= 1 && levels <= 8 && batch > 0, MVSN_E_BADARG, "mvsn_gather_focal: bad arguments");
  mvsn::FocalSrc src = {};
  for (int l = 0; l < levels; ++l) {
    MVSN_REQUIRE(K_pyr[l], MVSN_E_BADARG, "mvsn_gather_focal: null pointer");
    src.K[l] = K_pyr[l];
  }
  const int total = levels * batch;
  hipLaunchKernelGGL(mvsn::gather_focal_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, levels,
                     batch, fx, (const void *)src.K[0], (const void *)src.K[1], (const void *)src.K[2],
                     (const void *)src.K[3], (const void *)src.K[4], (const void *)src.K[5], (const void *)src.K[6],
                     (const void *)src.K[7], (const void *)nullptr, (const void *)nullptr);   // MVSN_VIS10
  return mvsn::check_launch("mvsn_gather_focal");
}

extern "C" int mvsn_copy(void *dst, const void *src, size_t nbytes, mvsn_stream_t stream) {
  MVSN_REQUIRE(dst && src, MVSN_E_BADARG, "mvsn_copy: null pointer");
  if (nbytes == 0) return 0;
  hipError_t e = hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) {
    mvsn::set_error("mvsn_copy: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

namespace mvsn {
// Up to eight independent device-to-device copies in ONE launch, every buffer a plain pointer argument (the runtime
// sees what is written and read: MVSN_VIS10's rule; ATen's multi-tensor copy carries its pointers inside a struct).
// grid.y = pair; 16-byte accesses where both pointers and the size allow it, bytes otherwise.
struct CopySizes {
  size_t n[8];
};
__global__ __launch_bounds__(256) void copy8_kernel(void *d0, const void *s0, void *d1, const void *s1, void *d2,
                                                    const void *s2, void *d3, const void *s3, void *d4, const void *s4,
                                                    void *d5, const void *s5, void *d6, const void *s6, void *d7,
                                                    const void *s7, CopySizes sz) {
  void *const d[8] = {d0, d1, d2, d3, d4, d5, d6, d7};
  const void *const s[8] = {s0, s1, s2, s3, s4, s5, s6, s7};
  const int k = blockIdx.y;
  char *dst = nullptr;
  const char *src = nullptr;
  size_t n = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i == k) dst = (char *)d[i], src = (const char *)s[i], n = sz.n[i];
  const size_t stride = (size_t)gridDim.x * 256, t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if ((((size_t)dst | (size_t)src | n) & 15) == 0) {
    for (size_t i = t; i < n / 16; i += stride) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
  } else {
    for (size_t i = t; i < n; i += stride) dst[i] = src[i];
  }
}
}  // namespace mvsn

extern "C" int mvsn_copy_many(void *const *dst, const void *const *src, const size_t *nbytes, int count,
                              mvsn_stream_t stream) {
  MVSN_REQUIRE(count >= 0 && (count == 0 || (dst && src && nbytes)), MVSN_E_BADARG, "mvsn_copy_many: bad arguments");
  for (int at = 0; at < count; at += 8) {
    const int m = count - at < 8 ? count - at : 8;
    void *d[8] = {};
    const void *s[8] = {};
    mvsn::CopySizes sz = {};
    size_t largest = 0;
    for (int i = 0; i < m; ++i) {
      MVSN_REQUIRE(nbytes[at + i] == 0 || (dst[at + i] && src[at + i]), MVSN_E_BADARG, "mvsn_copy_many: null pointer");
      d[i] = dst[at + i], s[i] = src[at + i], sz.n[i] = nbytes[at + i];
      if (sz.n[i] > largest) largest = sz.n[i];
    }
    if (largest == 0) continue;
    size_t blocks = (largest / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(mvsn::copy8_kernel, dim3((unsigned)blocks, m), dim3(256), 0, (hipStream_t)stream, d[0], s[0], d[1],
                       s[1], d[2], s[2], d[3], s[3], d[4], s[4], d[5], s[5], d[6], s[6], d[7], s[7], sz);
  }
  return mvsn::check_launch("mvsn_copy_many");
}

extern "C" int mvsn_gather_strided(const float *src, int count, long stride, float *dst, mvsn_stream_t stream) {
  MVSN_REQUIRE(src && dst && count > 0 && stride > 0, MVSN_E_BADARG, "mvsn_gather_strided: bad arguments");
  hipLaunchKernelGGL(mvsn::gather_strided_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, count,
                     stride, dst);
  return mvsn::check_launch("mvsn_gather_strided");
}

extern "C" int mvsn_channel_l2_norm(const float *x, int n, int channels, long pixels, float *out,
                                    mvsn_stream_t stream) {
  MVSN_REQUIRE(x && out, MVSN_E_BADARG, "mvsn_channel_l2_norm: null pointer");
  MVSN_REQUIRE(n > 0 && n <= 65535 && channels > 0 && pixels > 0, MVSN_E_BADARG, "mvsn_channel_l2_norm: bad sizes");
  hipLaunchKernelGGL(mvsn::channel_l2_norm_kernel, dim3((unsigned)((pixels + 255) / 256), n), dim3(256), 0,
                     (hipStream_t)stream, x, channels, pixels, out);
  return mvsn::check_launch("mvsn_channel_l2_norm");
}

extern "C" int mvsn_idepth_scale(const float *prior, const float *fx, int n, long pixels, float *out,
                                 mvsn_stream_t stream) {
  MVSN_REQUIRE(prior && fx && out, MVSN_E_BADARG, "mvsn_idepth_scale: null pointer");
  MVSN_REQUIRE(n > 0 && n <= 65535 && pixels > 0, MVSN_E_BADARG, "mvsn_idepth_scale: bad sizes");
  hipLaunchKernelGGL(mvsn::idepth_gain_kernel<false>, dim3((unsigned)((pixels + 255) / 256), n), dim3(256), 0,
                     (hipStream_t)stream, prior, fx, (const float *)nullptr, pixels, out);
  return mvsn::check_launch("mvsn_idepth_scale");
}

extern "C" int mvsn_refiner_epilogue(const float *prior, const float *fx, const float *delta, int n, long pixels,
                                     float *out, mvsn_stream_t stream) {
  MVSN_REQUIRE(prior && fx && delta && out, MVSN_E_BADARG, "mvsn_refiner_epilogue: null pointer");
  MVSN_REQUIRE(n > 0 && n <= 65535 && pixels > 0, MVSN_E_BADARG, "mvsn_refiner_epilogue: bad sizes");
  hipLaunchKernelGGL(mvsn::idepth_gain_kernel<true>, dim3((unsigned)((pixels + 255) / 256), n), dim3(256), 0,
                     (hipStream_t)stream, prior, fx, delta, pixels, out);
  return mvsn::check_launch("mvsn_refiner_epilogue");
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

extern "C" int mvsn_upsample_prior(const float *in, const float *fx, int n, int rows_in, int cols_in, int rows_out,
                                   int cols_out, float *out, float *out_scaled, mvsn_stream_t stream) {
  MVSN_REQUIRE(in && fx && out && out_scaled, MVSN_E_BADARG, "mvsn_upsample_prior: null pointer");
  MVSN_REQUIRE(n > 0 && rows_in > 0 && cols_in > 0 && rows_out > 0 && cols_out > 0, MVSN_E_BADARG,
               "mvsn_upsample_prior: bad sizes");
  MVSN_REQUIRE(n <= 65535 && rows_out <= 65535, MVSN_E_TOOLARGE, "mvsn_upsample_prior: grid");
  dim3 grid((cols_out + 255) / 256, rows_out, n);
  hipLaunchKernelGGL(mvsn::upsample_gain_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, fx, rows_in, cols_in,
                     rows_out, cols_out, out, out_scaled);
  return mvsn::check_launch("mvsn_upsample_prior");
}

extern "C" int mvsn_upsample_mask(const uint8_t *in, int n, int channels, int rows_in, int cols_in, int rows_out,
                                  int cols_out, uint8_t *out, mvsn_stream_t stream) {
  MVSN_REQUIRE(in && out, MVSN_E_BADARG, "mvsn_upsample_mask: null pointer");
  MVSN_REQUIRE(n > 0 && channels > 0 && rows_in > 0 && cols_in > 0 && rows_out > 0 && cols_out > 0, MVSN_E_BADARG,
               "mvsn_upsample_mask: bad sizes");
  const int groups_per_row = (cols_out + 15) / 16;
  const size_t total = (size_t)n * channels * rows_out * groups_per_row;
  MVSN_REQUIRE((total + 255) / 256 < 2147483647ull, MVSN_E_TOOLARGE, "mvsn_upsample_mask: grid");
  if (rows_out == 2 * rows_in && cols_out == 2 * cols_in && cols_in % 8 == 0 && (((size_t)in | (size_t)out) & 15) == 0) {
    const size_t groups = (size_t)n * channels * rows_in * (cols_in / 8);
    hipLaunchKernelGGL(mvsn::upsample_mask2x_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, in, rows_in, cols_in, cols_in / 8, groups, out);
    return mvsn::check_launch("mvsn_upsample_mask(2x)");
  }
  hipLaunchKernelGGL(mvsn::upsample_mask16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, in, rows_in, cols_in, rows_out, cols_out, groups_per_row, total, out);
  return mvsn::check_launch("mvsn_upsample_mask");
}

extern "C" int mvsn_area_downsample(const float *in, int n, int channels, int rows_in, int cols_in, float *out,
                                    mvsn_stream_t stream) {
  MVSN_REQUIRE(in && out, MVSN_E_BADARG, "mvsn_area_downsample: null pointer");
  MVSN_REQUIRE(n > 0 && channels > 0 && rows_in > 0 && cols_in > 0, MVSN_E_BADARG, "mvsn_area_downsample: bad sizes");
  const int rows_out = (rows_in + 1) / 2, cols_out = (cols_in + 1) / 2;
  MVSN_REQUIRE((long)n * channels <= 65535 && rows_out <= 65535, MVSN_E_TOOLARGE, "mvsn_area_downsample: grid");
  dim3 grid((cols_out + 255) / 256, rows_out, n * channels);
  hipLaunchKernelGGL(mvsn::area_downsample_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, rows_in, cols_in,
                     rows_out, cols_out, out);
  return mvsn::check_launch("mvsn_area_downsample");
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

// ---------------------------------------------------------------------------------------------
// 32 -> 1 channel 3x3 / 3x3x3 convolution (the last layer of every refiner and of the regulariser)
// ---------------------------------------------------------------------------------------------
// At 128 input bytes per output the layer is HBM-bound.  One output channel has no cout dimension to put
// on MFMA directly; the 3-D layer (27 taps) instead puts the TAPS on it (tap GEMM, further down), the 2-D
// layer (9 taps) runs on the vector ALUs: each thread owns 4 consecutive output
// columns of one row, reads the three neighbouring input rows of every channel as aligned
// float4 and takes the two halo columns from its lane neighbours with wave shuffles (a 16-lane group
// spans 64 columns; only the group's edge lanes touch memory for the halo).  Weights are wave-uniform
// (scalar loads).  Optional refiner epilogue: relu(prior*fx + conv + bias) / fx
// (multi_view_stereonet.py:482 with the gain trick of :607-611).
namespace mvsn {

// ---- 32 -> 1 channel 3x3x3 as a tap GEMM ------------------------------------------------------------------
// out[z][y][x] = sum_tap P[tap][z + dz - 1][y + dy - 1][x + dx - 1],  P[tap][v] = sum_c w[c][tap] in[c][v].
// P is a (27 x 32) by (32 x voxels) product: it runs on the fp32 matrix cores, every input element is read
// from HBM exactly once (one 16-byte load feeds four MFMAs), and the 27-tap shift-and-add reads P back from
// LDS.  A workgroup owns a 16 x 32 pixel tile and a slab of planes and streams through the slab: per input
// plane it forms P over the haloed tile (18 rows x 40 columns from the aligned column x0 - 4 = 720 slots),
// then every thread adds the plane's contribution to the three output planes it touches for its two pixels
// (rolling accumulators) and writes the plane that just became complete.
//   A = taps (16 per MFMA, 2 tiles) x 4 cins, held in registers for the whole kernel
//   B = 4 cins x 16 slots: lane (k, i) loads slots 64g + 4i .. + 3 of channel 4 ks + k with one dwordx4;
//       register p of that load is the B fragment of "slot tile p" = slots {64g + 4i + p}, so the four
//       accumulators of a lane are four CONSECUTIVE slots of one tap row -> one 16-byte LDS write.
__device__ __forceinline__ void t3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int T3_TY = 16, T3_TX = 32;
constexpr int T3_HY = T3_TY + 2, T3_XS = 40;          // haloed rows, row stride in slots
constexpr int T3_SLOTS = T3_HY * T3_XS;               // 720
constexpr int T3_GROUPS = (T3_SLOTS + 63) / 64;       // 12 groups of 64 slots, 3 per wave
constexpr int T3_LDS_FLOATS = 27 * T3_SLOTS;          // 77,760 bytes: two workgroups per CU

// XF: the input is a raw convolution output whose LeakyReLU(GroupNorm(.)) is applied on load (two VALU per
// element behind an HBM-bound stream) -- the regulariser's last normalise/activate pass never touches HBM.
// WHOLE: the plane IS the tile (16 x 32, the headline's coarse grid): the 208 halo slots of the 720 lie outside the
// volume for every plane -- their rows of P are zeroed once and never written, the products cover the 512 real slots
// only (two full 64-slot groups per wave instead of three partly idle ones: a third fewer multiplies -- at 22 flop per
// input byte the tap GEMM needs most of the fp32 matrix pipe at the HBM rate -- and every lane of every load carries
// data), and a plane's two groups are requested while the previous plane is gathered.
template <bool XF, bool WHOLE>
__global__ __launch_bounds__(256, 2) void conv_to1_3d_mfma_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                                  const float *__restrict__ bias,
                                                                  const float *__restrict__ in_stats,
                                                                  const float *__restrict__ in_gamma,
                                                                  const float *__restrict__ in_beta, int D, int H, int W,
                                                                  int ntx, int zslab, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float P[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.z;
  const int tile = xcd_tile_index(blockIdx.x, gridDim.x);
  const int tyi = tile / ntx, txi = tile - tyi * ntx;
  const int y0 = tyi * T3_TY, x0 = txi * T3_TX;
  const int zb = blockIdx.y * zslab, ze = min(D, zb + zslab);   // output planes [zb, ze)
  const size_t plane = (size_t)H * W, chan = (size_t)D * plane;
  const float *inn = in + (size_t)n * 32 * chan + (size_t)(lane >> 4) * chan;   // this lane's k (cin within a k-step)

  // tap weights as A fragments: a[t][ks] = w[cin = 4 ks + (lane>>4)][tap = 16 t + (lane&15)]
  float a[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int tap = t * 16 + (lane & 15), c = ks * 4 + (lane >> 4);
      a[t][ks] = tap < 27 ? w[c * 27 + tap] : 0.0f;
    }

  float xsc[8], xsh[8];   // XF: per k-step scale / shift of this lane's channel 4 ks + (lane>>4) (group ks >> 1)
  if constexpr (XF) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int c = ks * 4 + (lane >> 4);
      const float mean = in_stats[((size_t)n * 4 + (ks >> 1)) * 2], rstd = in_stats[((size_t)n * 4 + (ks >> 1)) * 2 + 1];
      xsc[ks] = rstd * in_gamma[c];
      xsh[ks] = in_beta[c] - mean * xsc[ks];
    }
  }

  // this lane's slots: group g = wave + 4 u, slots 64 g + 4 (lane&15) .. + 3
  constexpr int GPW = WHOLE ? 2 : T3_GROUPS / 4;   // 3 (WHOLE: 8 groups of 64 REAL slots, 2 per wave)
  int goff[GPW];   // global offset of the lane's four slots inside a plane (-1: outside the volume)
  int lslot[GPW];  // their first slot in P (-1: beyond the tile)
#pragma unroll
  for (int u = 0; u < GPW; ++u) {
    const int s0 = (wave + 4 * u) * 64 + 4 * (lane & 15);
    if constexpr (WHOLE) {   // H == 16, W == 32, one tile: pixel index == plane offset
      goff[u] = s0;
      lslot[u] = ((s0 >> 5) + 1) * T3_XS + (s0 & 31) + 4;
    } else {
      const int row = s0 / T3_XS, col = s0 - row * T3_XS;
      const int gy = y0 - 1 + row, gx = x0 - 4 + col;
      goff[u] = (s0 < T3_SLOTS && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
      lslot[u] = s0 < T3_SLOTS ? s0 : -1;
    }
  }
  if constexpr (WHOLE) {   // halo slots of every tap row: zero, once
    for (int i = tid * 4; i < T3_LDS_FLOATS; i += 1024) *reinterpret_cast<floatx4 *>(P + i) = floatx4{0.f, 0.f, 0.f, 0.f};
    t3_barrier();
  }

  // gather side: this thread's two output pixels (xx even)
  const int oy = tid >> 4, ox = (tid & 15) * 2;
  const bool o_ok = y0 + oy < H && x0 + ox < W;          // W % 4 == 0: both pixels or neither
  const float *pg = P + oy * T3_XS + ox + 3;             // tap (dy, dx) adds dy * T3_XS + dx
  float acc[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // output planes z-1, z, z+1 of the current input plane z
  const float b = bias ? bias[0] : 0.0f;
  float *outn = out + (size_t)n * chan + (size_t)(y0 + oy) * W + x0 + ox;

  // The stream over the slab's planes is software-pipelined ACROSS planes, two slot groups ahead: a group's 64
  // multiplies take ~0.85 us, an HBM round trip under load more than twice that, and with one request burst per plane
  // (round 2) the kernel sat at 27 % of the HBM rate.  Three groups per plane, three register buffers: group u of every
  // plane lives in bfr[u]; while group u multiplies, group u + 2 (of this plane or the next) is requested.
  floatx4 bfr[3][8];
  auto load_group = [&](int z, int u, floatx4 (&dst)[8]) {
    const float *src = inn + (size_t)z * plane + (goff[u] >= 0 ? goff[u] : 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      dst[ks] = goff[u] >= 0 ? __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(src + (size_t)ks * 4 * chan))
                             : floatx4{0.f, 0.f, 0.f, 0.f};
  };
  auto activate = [&](int u, floatx4 (&dst)[8]) {
    if constexpr (XF) {   // padding stays zero: it pads the ACTIVATED tensor
      if (goff[u] >= 0) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int p = 0; p < 4; ++p) dst[ks][p] = lrelu02(dst[ks][p] * xsc[ks] + xsh[ks]);
      }
    }
  };
  static_assert(GPW == (WHOLE ? 2 : 3), "the plane pipeline below is written for three (WHOLE: two) slot groups per wave");
  const int zfirst = zb - 1 < 0 ? 0 : zb - 1, zlast = ze < D ? ze : D - 1;   // input planes inside the volume
  if (zfirst <= zlast) {
    load_group(zfirst, 0, bfr[0]);
    load_group(zfirst, 1, bfr[1]);
  }
  auto plane_products = [&](int z) {
#pragma unroll
    for (int u = 0; u < GPW; ++u) {
      if constexpr (!WHOLE) {
        if (u == 0) load_group(z, 2, bfr[2]);
        else if (z + 1 <= zlast) load_group(z + 1, u - 1, bfr[u - 1]);   // next plane: in flight across the barriers below
      }
      activate(u, bfr[u]);
      floatx4 d[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 4; ++p) d[t][p] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          d[0][p] = mfma16x16x4(a[0][ks], bfr[u][ks][p], d[0][p]);
          d[1][p] = mfma16x16x4(a[1][ks], bfr[u][ks][p], d[1][p]);
        }
      // WHOLE: the group's registers are free again -- the same group of the next plane travels during the rest of this
      // plane (the other group's multiplies, the gather and both barriers)
      if constexpr (WHOLE) {
        if (z + 1 <= zlast) load_group(z + 1, u, bfr[u]);
      }
      // lane holds taps 16 t + 4 (lane>>4) + r of slots s0 .. s0 + 3
      const int s0 = lslot[u];
      if (s0 >= 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int tap = t * 16 + (lane >> 4) * 4 + r;
            if (tap < 27)
              *reinterpret_cast<floatx4 *>(P + tap * T3_SLOTS + s0) = floatx4{d[t][0][r], d[t][1][r], d[t][2][r], d[t][3][r]};
          }
      }
    }
  };

  for (int z = zb - 1; z <= ze; ++z) {
    if (z >= 0 && z < D) {   // uniform: planes outside the volume contribute nothing
      // ---- P = taps x slots for plane z
      plane_products(z);
      t3_barrier();   // (LDS only: the next plane's first loads stay in flight across it)
      // ---- shift-and-add: input plane z feeds output planes z + 1 (dz = 0), z (dz = 1), z - 1 (dz = 2)
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float *q = pg + (dz * 9 + dy * 3 + dx) * T3_SLOTS + dy * T3_XS + dx;
            acc[2 - dz][0] += q[0];
            acc[2 - dz][1] += q[1];
          }
    }
    // output plane z - 1 is complete
    if (z - 1 >= zb && z - 1 < ze && o_ok) {
      float2 v = make_float2(acc[0][0] + b, acc[0][1] + b);
      *reinterpret_cast<float2 *>(outn + (size_t)(z - 1) * plane) = v;
    }
    acc[0][0] = acc[1][0], acc[0][1] = acc[1][1];
    acc[1][0] = acc[2][0], acc[1][1] = acc[2][1];
    acc[2][0] = 0.f, acc[2][1] = 0.f;
    t3_barrier();   // everyone is done reading P before the next plane overwrites it
  }
}

// ---- 32 -> 1 channel 3x3 (2-D) the same way: 9 taps = one MFMA row tile ------------------------------------
// One workgroup per 16 x 32 tile.  XFORM: the input is not materialised -- the kernel reads the raw
// output r of the tower's last conv and the block input x and forms x + LeakyReLU(GN(r)) on the B
// fragments in registers (zero outside the image, as the padding of the materialised tensor would be),
// i.e. the last residual block's normalise/activate/add pass is folded into this layer's load.
// Epilogue (optional): the refiner's relu(prior * fx + conv + bias) / fx.
// Output tiles of 16 x TX: 16 x 64 (an 18 x 72-slot haloed tile = 1.27x the outputs) when there are enough tiles to
// fill the chip, 16 x 32 (1.41x, twice the workgroups) otherwise.  The halo lines of a 32-channel x 2-tensor stream
// do not all stay in the 4 MB L2, so the ratio is HBM traffic: level-0 folded block at batch 128 1.36 -> 1.12 ms
// (3.1 -> 3.9 TB/s algorithmic); 16 x 128 tiles 1.18 ms; round 1 measured 8 x 64 (the same 1.41x) within 1 % of 16 x 32.
constexpr int T2_TY = 16;
template <int TX>
struct T2 {
  static constexpr int XS = TX + 8, SLOTS = (T2_TY + 2) * XS, GROUPS = (SLOTS + 63) / 64, LDS_FLOATS = 9 * SLOTS;
};
// wide tiles once they alone give every CU four workgroups
static inline bool to1_wide_tiles(int n, int rows, int cols) {
  return (long)n * ((rows + T2_TY - 1) / T2_TY) * ((cols + 63) / 64) >= 4L * device_cus();
}

template <bool XFORM, int T2_TX>
__global__ __launch_bounds__(256) void conv_to1_2d_mfma_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                               const float *__restrict__ bias,
                                                               const float *__restrict__ in_stats,
                                                               const float *__restrict__ in_gamma,
                                                               const float *__restrict__ in_beta,
                                                               const float *__restrict__ in_residual,
                                                               const float *__restrict__ prior, const float *__restrict__ fx,
                                                               int H, int W, int ntx, int stat_tiles,
                                                               float *__restrict__ out) {
  constexpr int T2_XS = T2<T2_TX>::XS, T2_SLOTS = T2<T2_TX>::SLOTS, T2_GROUPS = T2<T2_TX>::GROUPS;
  __shared__ __attribute__((aligned(16))) float P[T2<T2_TX>::LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y;
  const int tile = xcd_tile_index(blockIdx.x, gridDim.x);   // neighbouring tiles (shared halo lines) on one XCD's L2
  const int tyi = tile / ntx, txi = tile - tyi * ntx;
  const int y0 = tyi * T2_TY, x0 = txi * T2_TX;
  const size_t plane = (size_t)H * W;
  const int kc = lane >> 4;   // this lane's cin within a k-step
  const float *inn = in + ((size_t)n * 32 + kc) * plane;
  const float *resn = (XFORM && in_residual) ? in_residual + ((size_t)n * 32 + kc) * plane : nullptr;

  float a[8];   // A fragments: w[cin = 4 ks + kc][tap = lane & 15]
  float sc[8], sh[8];
  __shared__ float gst[8];
  const float *st = nullptr;
  if constexpr (XFORM) st = gn_stats_here(in_stats, stat_tiles, n, gst);   // (stat_tiles > 0: from the producer's records)
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int tap = lane & 15, c = ks * 4 + kc;
    a[ks] = tap < 9 ? w[c * 9 + tap] : 0.0f;
    if constexpr (XFORM) {
      const float mean = st[(c >> 3) * 2 + 0];
      const float rstd = st[(c >> 3) * 2 + 1];
      sc[ks] = rstd * in_gamma[c];
      sh[ks] = in_beta[c] - mean * sc[ks];
    }
  }

  constexpr int GPW = (T2_GROUPS + 3) / 4;   // groups of 64 slots per wave (the last round may be partial)
#pragma unroll
  for (int u = 0; u < GPW; ++u) {
    if ((wave + 4 * u) * 64 >= T2_SLOTS) break;   // wave-uniform
    const int s0 = (wave + 4 * u) * 64 + 4 * (lane & 15);
    const int row = s0 / T2_XS, col = s0 - row * T2_XS;
    const int gy = y0 - 1 + row, gx = x0 - 4 + col;
    const bool ok = s0 < T2_SLOTS && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t off = ok ? (size_t)gy * W + gx : 0;
    floatx4 bfr[8], rfr[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bfr[ks] = ok ? *reinterpret_cast<const floatx4 *>(inn + (size_t)ks * 4 * plane + off) : floatx4{0.f, 0.f, 0.f, 0.f};
      if constexpr (XFORM)
        rfr[ks] = (ok && resn) ? *reinterpret_cast<const floatx4 *>(resn + (size_t)ks * 4 * plane + off)
                               : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    floatx4 d[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) d[p] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if constexpr (XFORM) {   // groups outside the image were loaded as zeros and must stay zero: shift masked once
        const float shk = ok ? sh[ks] : 0.0f;
#pragma unroll
        for (int p = 0; p < 4; ++p) bfr[ks][p] = rfr[ks][p] + lrelu02(bfr[ks][p] * sc[ks] + shk);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) d[p] = mfma16x16x4(a[ks], bfr[ks][p], d[p]);
    }
    if (s0 < T2_SLOTS) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tap = kc * 4 + r;
        if (tap < 9) *reinterpret_cast<floatx4 *>(P + tap * T2_SLOTS + s0) = floatx4{d[0][r], d[1][r], d[2][r], d[3][r]};
      }
    }
  }
  __syncthreads();
  const float b = bias ? bias[0] : 0.0f;
#pragma unroll
  for (int k = 0; k < T2_TY * T2_TX / 512; ++k) {   // two neighbouring outputs per thread and round
    const int idx = tid + 256 * k;
    const int oy = idx / (T2_TX / 2), ox = (idx % (T2_TX / 2)) * 2;
    if (y0 + oy >= H || x0 + ox >= W) continue;
    const float *pg = P + oy * T2_XS + ox + 3;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float *q = pg + (dy * 3 + dx) * T2_SLOTS + dy * T2_XS + dx;
        acc0 += q[0];
        acc1 += q[1];
      }
    const size_t o = (size_t)n * plane + (size_t)(y0 + oy) * W + x0 + ox;
    float2 res = make_float2(acc0 + b, acc1 + b);
    if (prior) {
      const float g = fx[n];
      const float2 pv = *reinterpret_cast<const float2 *>(prior + o);
      const float s0v = pv.x * g + res.x, s1v = pv.y * g + res.y;
      res = make_float2(relu_nan(s0v) / g, relu_nan(s1v) / g);
    }
    *reinterpret_cast<float2 *>(out + o) = res;
  }
}

}  // namespace mvsn

static int conv_to1_volume(const float *in, const float *weight, const float *bias, const float *in_stats,
                           const float *in_gamma, const float *in_beta, int n, int depth, int rows, int cols,
                           float *out, mvsn_stream_t stream) {
  using namespace mvsn;
  // tap GEMM on the matrix cores; slabs of planes so that even one chain fills the chip
  const int nty = (rows + T3_TY - 1) / T3_TY, ntx = (cols + T3_TX - 1) / T3_TX;
  int nslab = 1;
  while (nslab < 8 && (long)n * nty * ntx * nslab < 1024 && depth / (nslab * 2) >= 8) nslab *= 2;
  // a handful of chains on a small grid (batch 1: 2 x 1 tile): a plane is ~7 us of dependent load -> multiply -> LDS
  // round trips, so thinner slabs (down to one output plane, its two halo planes re-read) while workgroups < CUs
  while (nslab < depth && (long)n * nty * ntx * nslab * 2 <= device_cus()) nslab *= 2;
  const int zslab = (depth + nslab - 1) / nslab;
  const int nz = (depth + zslab - 1) / zslab;
  MVSN_REQUIRE(n <= 65535 && nz <= 65535, MVSN_E_TOOLARGE, "mvsn_conv_to1: grid");
  const size_t lds = (size_t)T3_LDS_FLOATS * sizeof(float);
  const bool whole = rows == T3_TY && cols == T3_TX;   // the plane is the tile
#define MVSN_TO1_3D_LAUNCH(XF_, WHOLE_)                                                                                \
  do {                                                                                                                 \
    static LdsOptIn opt;                                                                                               \
    if (int rc = ensure_lds(opt, (const void *)conv_to1_3d_mfma_kernel<XF_, WHOLE_>, lds, "mvsn_conv_to1")) return rc; \
    hipLaunchKernelGGL((conv_to1_3d_mfma_kernel<XF_, WHOLE_>), dim3(nty * ntx, nz, n), dim3(256), lds,                 \
                       (hipStream_t)stream, in, weight, bias, in_stats, in_gamma, in_beta, depth, rows, cols, ntx,     \
                       zslab, out);                                                                                    \
  } while (0)
  if (in_stats) {
    if (whole) MVSN_TO1_3D_LAUNCH(true, true);
    else MVSN_TO1_3D_LAUNCH(true, false);
  } else {
    in_gamma = in_beta = nullptr;
    if (whole) MVSN_TO1_3D_LAUNCH(false, true);
    else MVSN_TO1_3D_LAUNCH(false, false);
  }
#undef MVSN_TO1_3D_LAUNCH
  return mvsn::check_launch("mvsn_conv_to1(volume)");
}

extern "C" int mvsn_conv_to1_supported(int rows, int cols) { return (cols % 4 == 0 && rows > 0) ? 1 : 0; }

extern "C" int mvsn_conv_to1(const float *in, const float *weight, const float *bias, const float *prior,
                             const float *fx, int n, int depth, int rows, int cols, int kd, float *out,
                             mvsn_stream_t stream) {
  MVSN_REQUIRE(in && weight && out, MVSN_E_BADARG, "mvsn_conv_to1: null pointer");
  MVSN_REQUIRE(n > 0 && depth > 0 && rows > 0 && cols > 0 && (kd == 1 || kd == 3), MVSN_E_BADARG,
               "mvsn_conv_to1: bad sizes");
  MVSN_REQUIRE(cols % 4 == 0, MVSN_E_BADARG, "mvsn_conv_to1: cols must be a multiple of 4 (use mvsn_conv_forward)");
  MVSN_REQUIRE(!prior || (fx && depth == 1), MVSN_E_BADARG, "mvsn_conv_to1: refiner epilogue needs fx and 2-D input");
  if (kd == 3) {
    return conv_to1_volume(in, weight, bias, nullptr, nullptr, nullptr, n, depth, rows, cols, out, stream);
  } else {
    MVSN_REQUIRE(depth == 1, MVSN_E_BADARG, "mvsn_conv_to1: kd = 1 needs depth = 1");
    MVSN_REQUIRE(n <= 65535, MVSN_E_TOOLARGE, "mvsn_conv_to1: grid");
    const int nty = (rows + mvsn::T2_TY - 1) / mvsn::T2_TY;
    if (mvsn::to1_wide_tiles(n, rows, cols)) {
      const int ntx = (cols + 63) / 64;
      hipLaunchKernelGGL((mvsn::conv_to1_2d_mfma_kernel<false, 64>), dim3(nty * ntx, n), dim3(256), 0,
                         (hipStream_t)stream, in, weight, bias, (const float *)nullptr, (const float *)nullptr,
                         (const float *)nullptr, (const float *)nullptr, prior, fx, rows, cols, ntx, 0, out);
    } else {
      const int ntx = (cols + 31) / 32;
      hipLaunchKernelGGL((mvsn::conv_to1_2d_mfma_kernel<false, 32>), dim3(nty * ntx, n), dim3(256), 0,
                         (hipStream_t)stream, in, weight, bias, (const float *)nullptr, (const float *)nullptr,
                         (const float *)nullptr, (const float *)nullptr, prior, fx, rows, cols, ntx, 0, out);
    }
  }
  return mvsn::check_launch("mvsn_conv_to1");
}

extern "C" int mvsn_conv_to1_volume_norm(const float *in_raw, const float *in_stats, const float *in_gamma,
                                         const float *in_beta, const float *weight, const float *bias, int n, int depth,
                                         int rows, int cols, float *out, mvsn_stream_t stream) {
  MVSN_REQUIRE(in_raw && in_stats && in_gamma && in_beta && weight && out, MVSN_E_BADARG,
               "mvsn_conv_to1_volume_norm: null pointer");
  MVSN_REQUIRE(n > 0 && depth > 0 && rows > 0 && cols > 0, MVSN_E_BADARG, "mvsn_conv_to1_volume_norm: bad sizes");
  MVSN_REQUIRE(cols % 4 == 0, MVSN_E_BADARG, "mvsn_conv_to1_volume_norm: cols must be a multiple of 4");
  return conv_to1_volume(in_raw, weight, bias, in_stats, in_gamma, in_beta, n, depth, rows, cols, out, stream);
}

static int conv_to1_block_launch(const char *what, const float *in_raw, const float *in_stats, int stat_tiles,
                                 const float *in_gamma, const float *in_beta, const float *in_residual,
                                 const float *weight, const float *bias, const float *prior, const float *fx, int n,
                                 int rows, int cols, float *out, mvsn_stream_t stream) {
  MVSN_REQUIRE(in_raw && in_stats && in_gamma && in_beta && weight && out, MVSN_E_BADARG, "%s: null pointer", what);
  MVSN_REQUIRE(n > 0 && rows > 0 && cols > 0, MVSN_E_BADARG, "%s: bad sizes", what);
  MVSN_REQUIRE(cols % 4 == 0, MVSN_E_BADARG, "%s: cols must be a multiple of 4", what);
  MVSN_REQUIRE(!prior || fx, MVSN_E_BADARG, "%s: refiner epilogue needs fx", what);
  MVSN_REQUIRE(n <= 65535, MVSN_E_TOOLARGE, "%s: grid", what);
  const int nty = (rows + mvsn::T2_TY - 1) / mvsn::T2_TY;
  if (mvsn::to1_wide_tiles(n, rows, cols)) {
    const int ntx = (cols + 63) / 64;
    hipLaunchKernelGGL((mvsn::conv_to1_2d_mfma_kernel<true, 64>), dim3(nty * ntx, n), dim3(256), 0, (hipStream_t)stream,
                       in_raw, weight, bias, in_stats, in_gamma, in_beta, in_residual, prior, fx, rows, cols, ntx,
                       stat_tiles, out);
  } else {
    const int ntx = (cols + 31) / 32;
    hipLaunchKernelGGL((mvsn::conv_to1_2d_mfma_kernel<true, 32>), dim3(nty * ntx, n), dim3(256), 0, (hipStream_t)stream,
                       in_raw, weight, bias, in_stats, in_gamma, in_beta, in_residual, prior, fx, rows, cols, ntx,
                       stat_tiles, out);
  }
  return mvsn::check_launch(what);
}

extern "C" int mvsn_conv_to1_block(const float *in_raw, const float *in_stats, const float *in_gamma,
                                   const float *in_beta, const float *in_residual, const float *weight,
                                   const float *bias, const float *prior, const float *fx, int n, int rows, int cols,
                                   float *out, mvsn_stream_t stream) {
  return conv_to1_block_launch("mvsn_conv_to1_block", in_raw, in_stats, 0, in_gamma, in_beta, in_residual, weight, bias,
                               prior, fx, n, rows, cols, out, stream);
}

extern "C" int mvsn_conv_to1_block_records(const float *in_raw, const float *in_records, int tiles,
                                           const float *in_gamma, const float *in_beta, const float *in_residual,
                                           const float *weight, const float *bias, const float *prior, const float *fx,
                                           int n, int rows, int cols, float *out, mvsn_stream_t stream) {
  MVSN_REQUIRE(tiles > 0, MVSN_E_BADARG, "mvsn_conv_to1_block_records: tiles");
  return conv_to1_block_launch("mvsn_conv_to1_block_records", in_raw, in_records, tiles, in_gamma, in_beta, in_residual,
                               weight, bias, prior, fx, n, rows, cols, out, stream);
}
