// Homography warp: (B,C,h,w) x (B,n,3,3) -> (B,C,n,h,w) + mask (B,n,h,w).
// One thread per output pixel of one plane; the 4-tap footprint is computed once and reused
// for every channel.  Reads are gathers from an image that stays L2-resident (one image feeds
// n planes); writes are fully coalesced along the row.
#include "mvsn_common.h"

namespace mvsn {

__global__ __launch_bounds__(256) void homography_warp_kernel(const float *__restrict__ image,
                                                              const float *__restrict__ H, int C, int n_planes,
                                                              int rows, int cols, float *__restrict__ volume,
                                                              uint8_t *__restrict__ mask) {
  const int P = rows * cols;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int plane = blockIdx.y;
  const int b = blockIdx.z;
  if (p >= P) return;
  const float *Hp = H + ((size_t)b * n_planes + plane) * 9;
  float Hl[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Hl[i] = Hp[i];
  const int y = p / cols, x = p - y * cols;
  WarpCoord c = warp_coord(Hl, (float)x, (float)y, (float)rows, (float)cols);
  Bilinear t = bilinear_taps(c.ix, c.iy, rows, cols);
  mask[((size_t)b * n_planes + plane) * P + p] = c.outside ? 1 : 0;
  const float keep = c.outside ? 0.0f : 1.0f;
  const int o00 = t.y0 * cols + t.x0, o01 = t.y0 * cols + t.x1, o10 = t.y1 * cols + t.x0, o11 = t.y1 * cols + t.x1;
  const float *img = image + (size_t)b * C * P;
  float *out = volume + (((size_t)b * C) * n_planes + plane) * P + p;
  for (int ch = 0; ch < C; ++ch) {
    const float *ic = img + (size_t)ch * P;
    float v = ic[o00] * t.w00 + ic[o01] * t.w01 + ic[o10] * t.w10 + ic[o11] * t.w11;
    out[(size_t)ch * n_planes * P] = keep * v;  // keep*NaN stays NaN, like the reference's multiply
  }
}

// Four consecutive pixels of a row per thread (cols % 4 == 0, 16-byte aligned rows): the coordinate algebra per
// pixel is unchanged (same fp32 expression order, so the mask flips on the same pixels), but the outputs leave as
// one 16-byte store per channel and the four mask bytes as one dword -- a quarter of the store instructions of the
// one-pixel form; the four footprints of a thread touch neighbouring texels (the full-resolution warps are near
// the identity), i.e. the same cache lines.
__global__ __launch_bounds__(256) void homography_warp4_kernel(const float *__restrict__ image,
                                                               const float *__restrict__ H, int C, int n_planes,
                                                               int rows, int cols, float *__restrict__ volume,
                                                               uint8_t *__restrict__ mask) {
  const int P = rows * cols, Q = P >> 2;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int plane = blockIdx.y;
  const int b = blockIdx.z;
  if (q >= Q) return;
  const float *Hp = H + ((size_t)b * n_planes + plane) * 9;
  float Hl[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Hl[i] = Hp[i];
  const int p = q * 4;
  const int y = p / cols, x = p - y * cols;
  int o00[4], o01[4], o10[4], o11[4];
  float w00[4], w01[4], w10[4], w11[4];
  unsigned mbits = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    WarpCoord c = warp_coord(Hl, (float)(x + k), (float)y, (float)rows, (float)cols);
    Bilinear t = bilinear_taps(c.ix, c.iy, rows, cols);
    mbits |= (c.outside ? 1u : 0u) << (8 * k);
    o00[k] = t.y0 * cols + t.x0, o01[k] = t.y0 * cols + t.x1, o10[k] = t.y1 * cols + t.x0, o11[k] = t.y1 * cols + t.x1;
    // keep * (sum) as in the one-pixel kernel: the weights stay separate so that keep*NaN stays NaN
    w00[k] = t.w00, w01[k] = t.w01, w10[k] = t.w10, w11[k] = t.w11;
  }
  *reinterpret_cast<unsigned *>(mask + ((size_t)b * n_planes + plane) * P + p) = mbits;
  const float *img = image + (size_t)b * C * P;
  float *out = volume + (((size_t)b * C) * n_planes + plane) * P + p;
  for (int ch = 0; ch < C; ++ch) {
    const float *ic = img + (size_t)ch * P;
    floatx4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float keep = ((mbits >> (8 * k)) & 1u) ? 0.0f : 1.0f;
      v[k] = keep * (ic[o00[k]] * w00[k] + ic[o01[k]] * w01[k] + ic[o10[k]] * w10[k] + ic[o11[k]] * w11[k]);
    }
    __builtin_nontemporal_store(v, reinterpret_cast<floatx4 *>(out + (size_t)ch * n_planes * P));
  }
}

}  // namespace mvsn

extern "C" int mvsn_homography_warp(const float *image, const float *H, int batch, int channels, int n_planes,
                                    int rows, int cols, float *volume, uint8_t *mask, mvsn_stream_t stream) {
  MVSN_REQUIRE(image && H && volume && mask, MVSN_E_BADARG, "mvsn_homography_warp: null pointer");
  MVSN_REQUIRE(batch > 0 && channels > 0 && n_planes > 0 && rows > 0 && cols > 0, MVSN_E_BADARG,
               "mvsn_homography_warp: bad sizes");
  MVSN_REQUIRE(n_planes <= 65535 && batch <= 65535, MVSN_E_TOOLARGE, "mvsn_homography_warp: grid too large");
  const int P = rows * cols;
  if ((cols & 3) == 0 && ((((size_t)volume) | ((size_t)mask)) & 15) == 0 && P >= 4096) {   // full-resolution warps
    dim3 grid((P / 4 + 255) / 256, n_planes, batch);
    hipLaunchKernelGGL(mvsn::homography_warp4_kernel, grid, dim3(256), 0, (hipStream_t)stream, image, H, channels,
                       n_planes, rows, cols, volume, mask);
    return mvsn::check_launch("mvsn_homography_warp");
  }
  dim3 grid((P + 255) / 256, n_planes, batch);
  hipLaunchKernelGGL(mvsn::homography_warp_kernel, grid, dim3(256), 0, (hipStream_t)stream, image, H, channels,
                     n_planes, rows, cols, volume, mask);
  return mvsn::check_launch("mvsn_homography_warp");
}
