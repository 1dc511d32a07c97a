// Input preparation on the device: what multi_view_unpack_batch does to a batch before the forward
// (multi_view_stereonet/multi_view_stereonet_utils.py:541-641, utils/image_utils.py:111-128), as two launches:
//   mvsn_image_pyramid    all levels of the ceil-halving area pyramid of a frame batch in ONE pass over the frames
//                         (sizes divisible by 2^(levels-1): every level is the exact 2x2 mean of the one above, which
//                         is what interpolate(mode="area") computes for even sizes; other sizes go level by level
//                         through mvsn_area_downsample)
//   mvsn_prepare_cameras  the K pyramid (:575-581), the source poses and their inverses divided by the baseline to
//                         the FIRST source (:597-604), and that baseline
// HBM-bound: the pyramid reads every frame once (8-byte pieces per lane, coalesced) and writes 1/3 of that.
#include "mvsn_common.h"

namespace mvsn {

constexpr int PY_MAX_LEVELS = 6;
struct PyramidOut {
  float *level[PY_MAX_LEVELS];   // level[0] unused (the input)
};

// One 32x32 tile of level 0 per workgroup (16x16 threads, a 2x2 block each); level l+1 is formed from level l's
// ROUNDED values (the reference calls interpolate once per level), summed in row-major order like the pooling loop.
__global__ __launch_bounds__(256) void image_pyramid_kernel(const float *__restrict__ in, int rows, int cols, int levels,
                                                            PyramidOut out, MVSN_VIS10) {   // (MVSN_VIS10: mvsn_common.h)
  __shared__ float buf[2][16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const size_t plane = blockIdx.z;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
  const int x = x0 + 2 * tx, y = y0 + 2 * ty;
  float v = 0.0f;
  const bool ok = x < cols && y < rows;   // rows, cols even: the 2x2 block is all inside or all outside
  if (ok) {
    const float *ip = in + plane * rows * cols + (size_t)y * cols + x;
    const float2 a = *reinterpret_cast<const float2 *>(ip);
    const float2 b = *reinterpret_cast<const float2 *>(ip + cols);
    v = (((a.x + a.y) + b.x) + b.y) / 4.0f;
    out.level[1][plane * (rows / 2) * (cols / 2) + (size_t)(y / 2) * (cols / 2) + x / 2] = v;
  }
  int cur = 0, n = 16;   // level l has an n x n tile of this workgroup in buf[cur]
  buf[0][ty][tx] = v;
  for (int l = 2; l < levels; ++l) {
    __syncthreads();
    const int m = n >> 1;
    float w = 0.0f;
    const int lr = rows >> l, lc = cols >> l;
    const int gx = (x0 >> l) + tx, gy = (y0 >> l) + ty;
    const bool in_tile = tx < m && ty < m;
    if (in_tile) w = (((buf[cur][2 * ty][2 * tx] + buf[cur][2 * ty][2 * tx + 1]) + buf[cur][2 * ty + 1][2 * tx]) +
                      buf[cur][2 * ty + 1][2 * tx + 1]) / 4.0f;
    if (in_tile && gx < lc && gy < lr) out.level[l][plane * lr * lc + (size_t)gy * lc + gx] = w;
    cur ^= 1;
    if (in_tile) buf[cur][ty][tx] = w;
    n = m;
  }
}

// one thread per reference image
__global__ void prepare_cameras_kernel(const float *__restrict__ K, const float *__restrict__ T, int batch, int sources,
                                       int levels, const int *__restrict__ level_sizes, float *__restrict__ K_pyr,
                                       float *__restrict__ T_norm, float *__restrict__ Tinv_norm,
                                       float *__restrict__ baseline) {
#pragma clang fp contract(off)   // every multiply / add is a separately rounded ATen op in the reference
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const float *Kb = K + (size_t)b * 16;
  const float rows0 = (float)level_sizes[0], cols0 = (float)level_sizes[1];
  for (int l = 0; l < levels; ++l) {
    float *Ko = K_pyr + ((size_t)l * batch + b) * 16;
    for (int i = 0; i < 16; ++i) Ko[i] = Kb[i];
    if (l == 0) continue;
    // sx = float(w_l) / w_0 is a python double in the reference; the tensor op rounds it to fp32 (:575-581)
    const float sx = (float)((double)level_sizes[2 * l + 1] / (double)cols0);
    const float sy = (float)((double)level_sizes[2 * l] / (double)rows0);
    Ko[0] = Kb[0] * sx;
    Ko[5] = Kb[5] * sy;
    Ko[2] = sx * (Kb[2] + 0.5f) - 0.5f;
    Ko[6] = sy * (Kb[6] + 0.5f) - 0.5f;
  }
  const float *T0 = T + (size_t)b * 16;   // first source
  const float base = sqrtf((T0[3] * T0[3] + T0[7] * T0[7]) + T0[11] * T0[11]);
  baseline[b] = base;
  for (int s = 0; s < sources; ++s) {
    const float *Ts = T + ((size_t)s * batch + b) * 16;
    float *To = T_norm + ((size_t)s * batch + b) * 16, *Ti = Tinv_norm + ((size_t)s * batch + b) * 16;
    // inverse of the UN-normalised pose, then both translations divided by the baseline (:590, :601-604)
    double a[4][8];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) a[i][j] = (double)Ts[i * 4 + j], a[i][4 + j] = i == j ? 1.0 : 0.0;
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
      for (int j = 0; j < 4; ++j) {
        float t = Ts[i * 4 + j], ti = (float)a[i][4 + j];
        if (j == 3 && i < 3) t = t / base, ti = ti / base;
        To[i * 4 + j] = t;
        Ti[i * 4 + j] = ti;
      }
  }
}

}  // namespace mvsn

extern "C" int mvsn_image_pyramid_supported(int rows, int cols, int levels) {
  if (levels < 2 || levels > mvsn::PY_MAX_LEVELS || rows <= 0 || cols <= 0) return 0;
  const int m = (1 << (levels - 1)) - 1;
  return ((rows & m) == 0 && (cols & m) == 0) ? 1 : 0;
}

extern "C" int mvsn_image_pyramid(const float *in, int n, int channels, int rows, int cols, int levels,
                                  float *const *out_levels, mvsn_stream_t stream) {
  MVSN_REQUIRE(in && out_levels, MVSN_E_BADARG, "mvsn_image_pyramid: null pointer");
  MVSN_REQUIRE(n > 0 && channels > 0 && (long)n * channels <= 65535, MVSN_E_BADARG, "mvsn_image_pyramid: bad sizes");
  MVSN_REQUIRE(mvsn_image_pyramid_supported(rows, cols, levels), MVSN_E_BADARG,
               "mvsn_image_pyramid: %dx%d is not divisible by 2^(levels-1) (use mvsn_area_downsample per level)", rows,
               cols);
  MVSN_REQUIRE((((size_t)in) & 7) == 0, MVSN_E_BADARG, "mvsn_image_pyramid: input must be 8-byte aligned");
  mvsn::PyramidOut o;
  for (int l = 0; l < mvsn::PY_MAX_LEVELS; ++l) o.level[l] = nullptr;
  for (int l = 1; l < levels; ++l) {
    MVSN_REQUIRE(out_levels[l - 1], MVSN_E_BADARG, "mvsn_image_pyramid: null level pointer");
    o.level[l] = out_levels[l - 1];
  }
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, n * channels);
  MVSN_REQUIRE(grid.y <= 65535, MVSN_E_TOOLARGE, "mvsn_image_pyramid: grid");
  hipLaunchKernelGGL(mvsn::image_pyramid_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, rows, cols, levels, o,
                     (const void *)o.level[0], (const void *)o.level[1], (const void *)o.level[2], (const void *)o.level[3],
                     (const void *)o.level[4], (const void *)o.level[5], (const void *)nullptr, (const void *)nullptr,
                     (const void *)nullptr, (const void *)nullptr);
  return mvsn::check_launch("mvsn_image_pyramid");
}

extern "C" int mvsn_prepare_cameras(const float *K, const float *T_right_in_left, int batch, int n_sources, int levels,
                                    const int *level_sizes_dev, float *K_pyr, float *T_normalised,
                                    float *T_inverse_normalised, float *baseline, mvsn_stream_t stream) {
  MVSN_REQUIRE(K && T_right_in_left && level_sizes_dev && K_pyr && T_normalised && T_inverse_normalised && baseline,
               MVSN_E_BADARG, "mvsn_prepare_cameras: null pointer");
  MVSN_REQUIRE(batch > 0 && n_sources > 0 && levels > 0, MVSN_E_BADARG, "mvsn_prepare_cameras: bad sizes");
  hipLaunchKernelGGL(mvsn::prepare_cameras_kernel, dim3((batch + 63) / 64), dim3(64), 0, (hipStream_t)stream, K,
                     T_right_in_left, batch, n_sources, levels, level_sizes_dev, K_pyr, T_normalised,
                     T_inverse_normalised, baseline);
  return mvsn::check_launch("mvsn_prepare_cameras");
}
