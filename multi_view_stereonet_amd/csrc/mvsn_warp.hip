// Homography warp: (B,C,h,w) x (B,n,3,3) -> (B,C,n,h,w) + mask (B,n,h,w).
// One thread per output pixel of one plane; the 4-tap footprint is computed once and reused
// for every channel.  Reads are gathers from an image that stays L2-resident (one image feeds
// n planes); writes are fully coalesced along the row.
#include "mvsn_common.h"

namespace mvsn {

// warp_coord (mvsn_common.h) with its multiply-adds spelled out: the source leaves the contraction of a * b + c to the
// compiler, which fuses differently in different surroundings (the one-pixel kernel: fma(H1, y, H0 * x) + H2; four pixels
// per thread: H1 * y hoisted out of the pixel loop) -- a last-bit difference in the coordinate, so in the weights, so in
// the frame.  Both forms of this file use the order the one-pixel kernel had (its results do not change).
__device__ __forceinline__ WarpCoord warp_coord_pinned(const float *H, float x, float y, float rows, float cols) {
  const float u0 = __builtin_fmaf(H[1], y, H[0] * x) + H[2];
  const float u1 = __builtin_fmaf(H[4], y, H[3] * x) + H[5];
  const float u2 = __builtin_fmaf(H[7], y, H[6] * x) + H[8];
  const float px = u0 / u2;
  const float py = u1 / u2;
  const float hx = px + 0.5f, hy = py + 0.5f;
  const float nx = (hx + hx) / cols - 1.0f;
  const float ny = (hy + hy) / rows - 1.0f;
  WarpCoord c;
  c.outside = (fabsf(nx) > 1.0f) || (fabsf(ny) > 1.0f);
  c.ix = __builtin_fmaf(nx + 1.0f, cols, -1.0f) * 0.5f;
  c.iy = __builtin_fmaf(ny + 1.0f, rows, -1.0f) * 0.5f;
  return c;
}

// The four products of a pixel and channel in ONE fixed order (what the compiler had chosen for the one-pixel kernel),
// spelled out so that every form of the kernel rounds alike: a frame comes out with the same bits whichever form the
// launch size selects.
__device__ __forceinline__ float warp_blend(float ax, float ay, float bx, float by, float wa0, float wb0, float wa1,
                                            float wb1) {
  float t = ay * wb0;
  t = __builtin_fmaf(ax, wa0, t);
  t = __builtin_fmaf(bx, wa1, t);
  return __builtin_fmaf(by, wb1, t);
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void homography_warp_kernel(const float *__restrict__ image,
                                                              const float *__restrict__ H, int h_bstride, int C,
                                                              int n_planes, int rows, int cols, int cgroups,
                                                              float *__restrict__ volume,
                                                              uint8_t *__restrict__ mask) {
  const int P = rows * cols;
  // blockIdx.x = pixel block * cgroups + channel group: small launches (one coarse plane of a few chains, the chain's
  // stepwise form) are latency-bound on the per-thread channel loop, so the channels are dealt to several blocks
  // Whole planes (!SPLIT): an output row's lower taps are the next row's upper ones, and consecutive workgroups run on
  // DIFFERENT XCDs (b % 8), each with its own L2 -- every source row was fetched by two of them (the full-resolution warp
  // ran at 0.44 of the HBM rate on ~1.45x its algorithmic bytes).  Each XCD takes a contiguous band of rows instead.
#ifdef MVSN_WARP_NO_XCD_BANDS   // A/B aid
  const int pbx = blockIdx.x;
#else
  const int pbx = xcd_tile_index(blockIdx.x, gridDim.x);
#endif
  const int cg = SPLIT ? blockIdx.x % cgroups : 0, pb = SPLIT ? blockIdx.x / cgroups : pbx;
  const int p = pb * blockDim.x + threadIdx.x;
  const int plane = blockIdx.y;
  const int b = blockIdx.z;
  if (p >= P) return;
  const int cper = (C + cgroups - 1) / cgroups;
  const int c_lo = SPLIT ? cg * cper : 0, c_hi = SPLIT ? (c_lo + cper < C ? c_lo + cper : C) : C;
  const float *Hp = H + (size_t)b * h_bstride + plane * 9;
  float Hl[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Hl[i] = Hp[i];
  const int y = p / cols, x = p - y * cols;
  WarpCoord c = warp_coord_pinned(Hl, (float)x, (float)y, (float)rows, (float)cols);
  Bilinear t = bilinear_taps(c.ix, c.iy, rows, cols);
  if (cg == 0) mask[((size_t)b * n_planes + plane) * P + p] = c.outside ? 1 : 0;
  const float keep = c.outside ? 0.0f : 1.0f;
  const float *img = image + (size_t)b * C * P;
  float *out = volume + (((size_t)b * C) * n_planes + plane) * P + p;
  if (cols >= 2) {
    // The two taps of a row are neighbours in memory: ONE 8-byte load per row and channel (6 gathers per pixel
    // instead of 12 -- with a cache-resident working set the kernel is gather-issue-bound; for many frames see
    // homography_warp_px_kernel).  At the right border (x1 clamped onto x0, weight exactly 0) the pair starts one texel
    // earlier and the weights move to its second element: the same products, the same sum.
    const int xb = t.x0 < cols - 1 ? t.x0 : cols - 2;
    const bool sh = t.x0 != xb;
    const float wa0 = sh ? 0.0f : t.w00, wb0 = sh ? t.w00 : t.w01;
    const float wa1 = sh ? 0.0f : t.w10, wb1 = sh ? t.w10 : t.w11;
    const int r0 = t.y0 * cols + xb, r1 = t.y1 * cols + xb;
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    for (int ch = c_lo; ch < c_hi; ++ch) {
      const float *ic = img + (size_t)ch * P;
      const f2u a = *reinterpret_cast<const f2u *>(ic + r0), bb = *reinterpret_cast<const f2u *>(ic + r1);
      const float v = warp_blend(a.x, a.y, bb.x, bb.y, wa0, wb0, wa1, wb1);
      __builtin_nontemporal_store(keep * v, out + (size_t)ch * n_planes * P);   // keep*NaN stays NaN, as the reference
    }
    return;
  }
  const int o00 = t.y0 * cols + t.x0, o01 = t.y0 * cols + t.x1, o10 = t.y1 * cols + t.x0, o11 = t.y1 * cols + t.x1;
  for (int ch = c_lo; ch < c_hi; ++ch) {
    const float *ic = img + (size_t)ch * P;
    float v = ic[o00] * t.w00 + ic[o01] * t.w01 + ic[o10] * t.w10 + ic[o11] * t.w11;
    out[(size_t)ch * n_planes * P] = keep * v;  // keep*NaN stays NaN, like the reference's multiply
  }
}

// PX (4 or 8) consecutive pixels of a row per thread (cols % PX == 0): the coordinate algebra and the products per pixel are
// the one-pixel kernel's (same fp32 expression order: the same bits, the mask flips on the same pixels); the outputs leave
// as 16-byte streaming stores and the mask bytes as one 4- / 8-byte word.  In the HBM regime (more frames than the
// 256 MB memory-side cache holds) the one-pixel form's 4-byte stores reach 3.4 TB/s, this one 4.2-4.6; with few frames
// the one-pixel form's four times as many threads win (launch-latency-bound), see warp_launch.
template <int PX>
__global__ __launch_bounds__(256) void homography_warp_px_kernel(const float *__restrict__ image,
                                                                 const float *__restrict__ H, int h_bstride, int C,
                                                                 int n_planes, int rows, int cols,
                                                                 float *__restrict__ volume, uint8_t *__restrict__ mask) {
  const int P = rows * cols, Q = P / PX;
  const int q = xcd_tile_index(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const int plane = blockIdx.y;
  const int b = blockIdx.z;
  if (q >= Q) return;
  const float *Hp = H + (size_t)b * h_bstride + plane * 9;
  float Hl[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Hl[i] = Hp[i];
  const int p = q * PX;
  const int y = p / cols, x = p - y * cols;
  int r0[PX], r1[PX];
  float wa0[PX], wb0[PX], wa1[PX], wb1[PX], keep[PX];
  unsigned mbits[PX / 4] = {};
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    WarpCoord c = warp_coord_pinned(Hl, (float)(x + k), (float)y, (float)rows, (float)cols);
    Bilinear t = bilinear_taps(c.ix, c.iy, rows, cols);
    mbits[k >> 2] |= (c.outside ? 1u : 0u) << (8 * (k & 3));
    keep[k] = c.outside ? 0.0f : 1.0f;
    const int xb = t.x0 < cols - 1 ? t.x0 : cols - 2;    // (see the one-pixel kernel: the pair of a row as one 8-byte load)
    const bool sh = t.x0 != xb;
    wa0[k] = sh ? 0.0f : t.w00, wb0[k] = sh ? t.w00 : t.w01;
    wa1[k] = sh ? 0.0f : t.w10, wb1[k] = sh ? t.w10 : t.w11;
    r0[k] = t.y0 * cols + xb, r1[k] = t.y1 * cols + xb;
  }
  unsigned *mp = reinterpret_cast<unsigned *>(mask + ((size_t)b * n_planes + plane) * P + p);
#pragma unroll
  for (int k = 0; k < PX / 4; ++k) mp[k] = mbits[k];
  const float *img = image + (size_t)b * C * P;
  float *out = volume + (((size_t)b * C) * n_planes + plane) * P + p;
  typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
  for (int ch = 0; ch < C; ++ch) {
    const float *ic = img + (size_t)ch * P;
#pragma unroll
    for (int k4 = 0; k4 < PX / 4; ++k4) {
      floatx4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k4 * 4 + j;
        const f2u a = *reinterpret_cast<const f2u *>(ic + r0[k]), bb = *reinterpret_cast<const f2u *>(ic + r1[k]);
        v[j] = keep[k] * warp_blend(a.x, a.y, bb.x, bb.y, wa0[k], wb0[k], wa1[k], wb1[k]);
      }
      __builtin_nontemporal_store(v, reinterpret_cast<floatx4 *>(out + (size_t)ch * n_planes * P + k4 * 4));
    }
  }
}

// H for sample b, plane p at H + b * h_bstride + 9 p (the public entry point: h_bstride = 9 * n_planes)
int warp_launch(const float *image, const float *H, int h_bstride, int batch, int channels, int n_planes, int rows,
                int cols, float *volume, uint8_t *mask, hipStream_t stream) {
  const int P = rows * cols;
  const long threads = (long)((P + 255) / 256) * 256 * n_planes * batch;
  int cgroups = 1;   // fewer than ~4 threads per lane of the chip: split the channel loop (8 channels per block at most)
  while (cgroups * 8 < channels && threads * cgroups < 4L * 64 * 4 * device_cus()) cgroups *= 2;
  dim3 grid((unsigned)((P + 255) / 256) * cgroups, n_planes, batch);
#ifndef MVSN_WARP_PX   // pixels per thread of the many-frames form (A/B aid: 0 = the one-pixel kernel always)
#define MVSN_WARP_PX 4
#endif
  if (MVSN_WARP_PX && cgroups == 1 && cols % MVSN_WARP_PX == 0 && ((((size_t)volume | (size_t)mask) & 15) == 0) &&
      threads / MVSN_WARP_PX >= 16L * 64 * 4 * device_cus()) {   // (512x256 frames on 256 CUs: from 32 frames; 16 frames: 16.9 vs 19.7 us)
    constexpr int PX = MVSN_WARP_PX ? MVSN_WARP_PX : 4;
    hipLaunchKernelGGL(homography_warp_px_kernel<PX>, dim3((unsigned)((P / PX + 255) / 256), n_planes, batch), dim3(256), 0,
                       stream, image, H, h_bstride, channels, n_planes, rows, cols, volume, mask);
    return check_launch("mvsn_homography_warp");
  }
  if (cgroups > 1)
    hipLaunchKernelGGL(homography_warp_kernel<true>, grid, dim3(256), 0, stream, image, H, h_bstride, channels, n_planes,
                       rows, cols, cgroups, volume, mask);
  else
    hipLaunchKernelGGL(homography_warp_kernel<false>, grid, dim3(256), 0, stream, image, H, h_bstride, channels,
                       n_planes, rows, cols, 1, volume, mask);
  return check_launch("mvsn_homography_warp");
}

}  // namespace mvsn

extern "C" int mvsn_homography_warp(const float *image, const float *H, int batch, int channels, int n_planes,
                                    int rows, int cols, float *volume, uint8_t *mask, mvsn_stream_t stream) {
  MVSN_REQUIRE(image && H && volume && mask, MVSN_E_BADARG, "mvsn_homography_warp: null pointer");
  MVSN_REQUIRE(batch > 0 && channels > 0 && n_planes > 0 && rows > 0 && cols > 0, MVSN_E_BADARG,
               "mvsn_homography_warp: bad sizes");
  MVSN_REQUIRE(n_planes <= 65535 && batch <= 65535, MVSN_E_TOOLARGE, "mvsn_homography_warp: grid too large");
  return mvsn::warp_launch(image, H, 9 * n_planes, batch, channels, n_planes, rows, cols, volume, mask,
                           (hipStream_t)stream);
}
