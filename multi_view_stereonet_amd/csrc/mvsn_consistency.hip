// Two-view consistency ops: the reprojection of one view's idepth map into the other view, the
// occlusion mask and the masked L1 the left/right consistency loss is built from.  Replace
// IDepthmapProjector.forward (stereo/image_predictor.py:538-576 over :45-73 and :82-118) + grid_sample,
// get_occlusion_mask (multi_view_stereonet/losses.py:42-82) and the per-direction body of
// left_right_idepthmap_consistency_losses (:112-160).  HBM-bound elementwise / gather work: one thread per pixel,
// coalesced reads and writes, the 4-tap gather of the other view's map through L2.
#include "mvsn_common.h"

namespace mvsn {

constexpr int RP_THREADS = 256;

// fp64 inverse of a 4x4 (Gauss-Jordan with partial pivoting); the reference inverts in fp32 (torch.inverse) and
// rounds there, here the result is rounded to fp32 once
__device__ inline void inverse4x4(const float *m, double *out) {
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = (double)m[i * 4 + j];
      a[i][4 + j] = i == j ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r)
      if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    for (int j = 0; j < 8; ++j) {
      const double t = a[c][j];
      a[c][j] = a[piv][j];
      a[piv][j] = t;
    }
    const double inv = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= inv;
    for (int r = 0; r < 4; ++r)
      if (r != c) {
        const double f = a[r][c];
        for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
      }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out[i * 4 + j] = a[i][4 + j];
}

__global__ __launch_bounds__(RP_THREADS) void reproject_kernel(const float *__restrict__ K, const float *__restrict__ T,
                                                               const float *__restrict__ idepth,
                                                               const float *__restrict__ other,
                                                               const uint8_t *__restrict__ other_mask, int rows, int cols,
                                                               float *__restrict__ id_prime, float *__restrict__ sampled,
                                                               uint8_t *__restrict__ mask_sampled,
                                                               uint8_t *__restrict__ invalid, float *__restrict__ uv,
                                                               float *__restrict__ partials) {
  __shared__ float cam[9 + 12 + 12];   // Kinv (3x3), T_left_in_right rows 0..2, (K T_left_in_right) rows 0..2
  __shared__ float red[RP_THREADS / 64];
  const int b = blockIdx.y, P = rows * cols;
  if (threadIdx.x == 0) {
    double Kinv[16], Tinv[16];
    inverse4x4(K + (size_t)b * 16, Kinv);
    inverse4x4(T + (size_t)b * 16, Tinv);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) cam[i * 3 + j] = (float)Kinv[i * 4 + j];
    float Tf[16];
    for (int i = 0; i < 16; ++i) Tf[i] = (float)Tinv[i];
    for (int i = 0; i < 12; ++i) cam[9 + i] = Tf[i];
    const float *Kb = K + (size_t)b * 16;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s += Kb[i * 4 + k] * Tf[k * 4 + j];   // torch.matmul(K, Tinv) in fp32 (:104)
        cam[21 + i * 4 + j] = s;
      }
  }
  __syncthreads();
  const int p = blockIdx.x * RP_THREADS + threadIdx.x;
  float ad = 0.f;
  if (p < P) {
    const float x = (float)(p % cols), y = (float)(p / cols);
    const float depth = 1.0f / (idepth[(size_t)b * P + p] + 1e-6f);                       // :557
    const float X = depth * (cam[0] * x + cam[1] * y + cam[2]);                           // :69-70
    const float Y = depth * (cam[3] * x + cam[4] * y + cam[5]);
    const float Z = depth * (cam[6] * x + cam[7] * y + cam[8]);
    const float *Tl = cam + 9, *Pm = cam + 21;
    const float Zr = Tl[8] * X + Tl[9] * Y + Tl[10] * Z + Tl[11];                         // :563
    const float idp = 1.0f / (Zr + 1e-6f);                                                // :564
    const float c0 = Pm[0] * X + Pm[1] * Y + Pm[2] * Z + Pm[3];                           // :105
    const float c1 = Pm[4] * X + Pm[5] * Y + Pm[6] * Z + Pm[7];
    const float c2 = Pm[8] * X + Pm[9] * Y + Pm[10] * Z + Pm[11] + 1e-7f;                 // :107
    const float nx = ((c0 / c2 + 0.5f) * 2.0f) / (float)cols - 1.0f;                      // :112-116
    const float ny = ((c1 / c2 + 0.5f) * 2.0f) / (float)rows - 1.0f;
    const bool out = fabsf(nx) > 1.0f || fabsf(ny) > 1.0f;                                // :571-573
    // grid_sample(bilinear, border, align_corners=False) of the other view's map at (nx, ny)
    const float ix = ((nx + 1.0f) * (float)cols - 1.0f) * 0.5f, iy = ((ny + 1.0f) * (float)rows - 1.0f) * 0.5f;
    const Bilinear t = bilinear_taps(ix, iy, rows, cols);
    const float *o = other + (size_t)b * P;
    const float s = o[t.y0 * cols + t.x0] * t.w00 + o[t.y0 * cols + t.x1] * t.w01 + o[t.y1 * cols + t.x0] * t.w10 +
                    o[t.y1 * cols + t.x1] * t.w11;
    const size_t i = (size_t)b * P + p;
    id_prime[i] = idp;
    sampled[i] = s;
    invalid[i] = out ? 1 : 0;
    if (uv) uv[2 * i] = nx, uv[2 * i + 1] = ny;
    if (mask_sampled) {
      const uint8_t *m = other_mask + (size_t)b * P;
      const float ms = (float)m[t.y0 * cols + t.x0] * t.w00 + (float)m[t.y0 * cols + t.x1] * t.w01 +
                       (float)m[t.y1 * cols + t.x0] * t.w10 + (float)m[t.y1 * cols + t.x1] * t.w11;
      mask_sampled[i] = ms > 0.0f ? 1 : 0;                                                // losses.py:133-135
    }
    ad = fabsf(s - idp);
  }
  if (partials) {   // per-block sum of |sampled - reprojected|: the occlusion threshold is its mean over the image
    for (int off = 32; off > 0; off >>= 1) ad += __shfl_xor(ad, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ad;
    __syncthreads();
    if (threadIdx.x == 0) partials[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

__global__ __launch_bounds__(RP_THREADS) void occlusion_kernel(const float *__restrict__ id_prime,
                                                               const float *__restrict__ sampled,
                                                               const uint8_t *__restrict__ invalid,
                                                               const float *__restrict__ partials, int nblocks, int P,
                                                               uint8_t *__restrict__ mask) {
  __shared__ float thr_s;
  const int b = blockIdx.y;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nblocks; ++i) s += (double)partials[(size_t)b * nblocks + i];   // fixed order: deterministic
    thr_s = (float)(s / (double)P);                                                      // losses.py:70
  }
  __syncthreads();
  const int p = blockIdx.x * RP_THREADS + threadIdx.x;
  if (p >= P) return;
  const size_t i = (size_t)b * P + p;
  mask[i] = ((sampled[i] - id_prime[i]) > thr_s || invalid[i]) ? 1 : 0;                   // :69, :74, :78
}

// loss (+)= mean |a - b| over elements with neither skip flag set (one workgroup: a few hundred thousand elements at
// most, fixed summation order)
__global__ __launch_bounds__(1024) void masked_l1_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                         const uint8_t *__restrict__ skip_a,
                                                         const uint8_t *__restrict__ skip_b, long n, int accumulate,
                                                         float *__restrict__ loss) {
  __shared__ double ssum[16];
  __shared__ double scnt[16];
  double s = 0.0, c = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024)
    if (!skip_a[i] && !skip_b[i]) {
      s += (double)fabsf(a[i] - b[i]);
      c += 1.0;
    }
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    c += __shfl_xor(c, off, 64);
  }
  if ((threadIdx.x & 63) == 0) ssum[threadIdx.x >> 6] = s, scnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    double S = 0.0, C = 0.0;
    for (int w = 0; w < 16; ++w) S += ssum[w], C += scnt[w];
    const float l = (float)(S / C);   // no selected element: 0/0 = NaN, as torch's mean over an empty selection
    loss[0] = accumulate ? loss[0] + l : l;
  }
}

}  // namespace mvsn

extern "C" int mvsn_idepth_reproject_blocks(int pixels) {
  return pixels > 0 ? (pixels + mvsn::RP_THREADS - 1) / mvsn::RP_THREADS : 0;
}

extern "C" int mvsn_idepth_reproject(const float *K, const float *T_other_in_this, const float *idepth,
                                     const float *other_idepth, const uint8_t *other_mask, int batch, int rows,
                                     int cols, float *idepth_in_other, float *other_sampled,
                                     uint8_t *other_mask_sampled, uint8_t *invalid, float *uv, float *absdiff_partials,
                                     mvsn_stream_t stream) {
  MVSN_REQUIRE(K && T_other_in_this && idepth && other_idepth && idepth_in_other && other_sampled && invalid,
               MVSN_E_BADARG, "mvsn_idepth_reproject: null pointer");
  MVSN_REQUIRE(!other_mask_sampled || other_mask, MVSN_E_BADARG, "mvsn_idepth_reproject: mask output without mask input");
  MVSN_REQUIRE(batch > 0 && batch <= 65535 && rows > 0 && cols > 0, MVSN_E_BADARG, "mvsn_idepth_reproject: bad sizes");
  const int nb = mvsn_idepth_reproject_blocks(rows * cols);
  hipLaunchKernelGGL(mvsn::reproject_kernel, dim3(nb, batch), dim3(mvsn::RP_THREADS), 0, (hipStream_t)stream, K,
                     T_other_in_this, idepth, other_idepth, other_mask, rows, cols, idepth_in_other, other_sampled,
                     other_mask_sampled, invalid, uv, absdiff_partials);
  return mvsn::check_launch("mvsn_idepth_reproject");
}

extern "C" int mvsn_occlusion_mask(const float *idepth_in_other, const float *other_sampled, const uint8_t *invalid,
                                   const float *absdiff_partials, int batch, int pixels, uint8_t *mask,
                                   mvsn_stream_t stream) {
  MVSN_REQUIRE(idepth_in_other && other_sampled && invalid && absdiff_partials && mask, MVSN_E_BADARG,
               "mvsn_occlusion_mask: null pointer");
  MVSN_REQUIRE(batch > 0 && batch <= 65535 && pixels > 0, MVSN_E_BADARG, "mvsn_occlusion_mask: bad sizes");
  const int nb = mvsn_idepth_reproject_blocks(pixels);
  hipLaunchKernelGGL(mvsn::occlusion_kernel, dim3(nb, batch), dim3(mvsn::RP_THREADS), 0, (hipStream_t)stream,
                     idepth_in_other, other_sampled, invalid, absdiff_partials, nb, pixels, mask);
  return mvsn::check_launch("mvsn_occlusion_mask");
}

extern "C" int mvsn_masked_l1(const float *a, const float *b, const uint8_t *skip_a, const uint8_t *skip_b, long n,
                              int accumulate, float *loss, mvsn_stream_t stream) {
  MVSN_REQUIRE(a && b && skip_a && skip_b && loss, MVSN_E_BADARG, "mvsn_masked_l1: null pointer");
  MVSN_REQUIRE(n > 0, MVSN_E_BADARG, "mvsn_masked_l1: bad size");
  hipLaunchKernelGGL(mvsn::masked_l1_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, skip_a, skip_b, n,
                     accumulate, loss);
  return mvsn::check_launch("mvsn_masked_l1");
}
