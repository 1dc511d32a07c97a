// Shared helpers for libmvsn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "../../include/mvsn_hip.h"

namespace mvsn {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting: the largest size opted in so far is
// remembered per (launch site, device) under a lock, so a second GPU in the same process -- or a second host
// thread -- never launches a large-LDS kernel that only another device was opted in for.
struct LdsOptIn {
  static constexpr int kMaxDevices = 64;
  std::mutex lock;
  size_t opted[kMaxDevices] = {};
};

inline int ensure_lds(LdsOptIn &state, const void *kernel, size_t lds_bytes, const char *what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  const bool cached = dev < LdsOptIn::kMaxDevices;
  std::lock_guard<std::mutex> guard(state.lock);
  if (cached && lds_bytes <= state.opted[dev]) return 0;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) {
    set_error("%s: LDS opt-in of %zu bytes failed: %s", what, lds_bytes, hipGetErrorString(e));
    return (int)e;
  }
  if (cached) state.opted[dev] = lds_bytes;
  return 0;
}

// compute units of the current device (per device, cached)
inline int device_cus() {
  static std::mutex lock;
  static int cus[LdsOptIn::kMaxDevices] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  std::lock_guard<std::mutex> guard(lock);
  if (dev < LdsOptIn::kMaxDevices && cus[dev] > 0) return cus[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev < LdsOptIn::kMaxDevices) cus[dev] = n;
  return n;
}

#define MVSN_REQUIRE(cond, code, ...)   \
  do {                                  \
    if (!(cond)) {                      \
      ::mvsn::set_error(__VA_ARGS__);   \
      return (code);                    \
    }                                   \
  } while (0)

typedef float floatx4 __attribute__((ext_vector_type(4)));

// D(16x16) += A(16x4) * B(4x16), exact fp32.  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15];
// it receives D[(l>>4)*4 + r][l&15] in element r.
__device__ __forceinline__ floatx4 mfma16x16x4(float a, float b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v from the lane a DPP control selects (quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140, ...)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// LeakyReLU(0.2) as max(v, 0.2 v): two VALU instructions (multiply, max) instead of multiply / compare / select.
// Identical for every finite v (0.2 v < v exactly when v > 0); -0.0 maps to -0.0 either way.
__device__ __forceinline__ float lrelu02(float v) { return fmaxf(v, 0.2f * v); }

// Workgroup b is observed to run on XCD b % 8, each XCD with its own L2.  Neighbouring tiles share halo
// rows, so hand each XCD a contiguous range of tiles instead of every 8th one (bijective for any tile
// count; placement only affects speed, never results).
__device__ __forceinline__ int xcd_tile_index(int bid, int tiles) {
  const int q = tiles >> 3, r = tiles & 7;
  const int xcd = bid & 7, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Source-pixel coordinate of a homography warp, evaluated with the same fp32 expression order
// as the reference (stereo/image_predictor.py:493-516) followed by grid_sample's
// un-normalisation, so that the |n|>1 predicate flips on the same pixels.
struct WarpCoord {
  float ix, iy;  // un-normalised, NOT yet clamped
  bool outside;
};

__device__ __forceinline__ WarpCoord warp_coord(const float *H, float x, float y, float rows, float cols) {
  float u0 = H[0] * x + H[1] * y + H[2];
  float u1 = H[3] * x + H[4] * y + H[5];
  float u2 = H[6] * x + H[7] * y + H[8];
  float px = u0 / u2;
  float py = u1 / u2;
  float nx = ((px + 0.5f) * 2.0f) / cols - 1.0f;
  float ny = ((py + 0.5f) * 2.0f) / rows - 1.0f;
  WarpCoord c;
  c.outside = (fabsf(nx) > 1.0f) || (fabsf(ny) > 1.0f);
  c.ix = ((nx + 1.0f) * cols - 1.0f) * 0.5f;
  c.iy = ((ny + 1.0f) * rows - 1.0f) * 0.5f;
  return c;
}

// Clamp-to-edge bilinear footprint: integer top-left tap, the +1 taps clamped (their weight is
// exactly zero whenever the clamp acts), and the four weights.
struct Bilinear {
  int x0, y0, x1, y1;
  float w00, w01, w10, w11;
};

__device__ __forceinline__ Bilinear bilinear_taps(float ix, float iy, int rows, int cols) {
  // NaN coordinates (u2 == 0) propagate through the weights exactly as in grid_sample; fminf/fmaxf
  // would swallow them, so clamp with comparisons.
  ix = ix < 0.0f ? 0.0f : (ix > (float)(cols - 1) ? (float)(cols - 1) : ix);
  iy = iy < 0.0f ? 0.0f : (iy > (float)(rows - 1) ? (float)(rows - 1) : iy);
  float fx0 = floorf(ix), fy0 = floorf(iy);
  float fx = ix - fx0, fy = iy - fy0;
  Bilinear b;
  b.x0 = (int)fx0;
  b.y0 = (int)fy0;
  if (!(b.x0 >= 0 && b.x0 < cols)) b.x0 = 0;  // only reachable through NaN
  if (!(b.y0 >= 0 && b.y0 < rows)) b.y0 = 0;
  b.x1 = b.x0 + 1 < cols ? b.x0 + 1 : cols - 1;
  b.y1 = b.y0 + 1 < rows ? b.y0 + 1 : rows - 1;
  b.w00 = (1.0f - fx) * (1.0f - fy);
  b.w01 = fx * (1.0f - fy);
  b.w10 = (1.0f - fx) * fy;
  b.w11 = fx * fy;
  return b;
}

}  // namespace mvsn
