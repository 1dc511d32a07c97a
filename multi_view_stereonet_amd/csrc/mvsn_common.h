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
// F.relu as ATen computes it: a NaN stays a NaN (fmaxf(NaN, 0) would return 0 and turn a poisoned volume -- see
// chain_band_kernel's time-out path -- or a NaN input into a plausible-looking depth)
__device__ __forceinline__ float relu_nan(float v) { return v < 0.0f ? 0.0f : v; }

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


// ---- "visible" buffer arguments ------------------------------------------------------------------------------------
// A kernel that reaches a buffer only through a by-value argument STRUCT (ChainArgs, TowerArgs, WinoArgs::in1 ...)
// repeats the buffer as a plain pointer argument at the end of its signature.  The kernel never reads these
// arguments; the runtime does: a captured hipGraph decides the cache maintenance between two kernel nodes from the
// buffers it finds among their pointer arguments, and a pointer inside a struct is invisible to it.  Measured on
// MI355X / ROCm 7.2 (tools/soak.py, alternating inputs over graph replays of the batch-1 forward): with the banded
// chain's buffers only inside ChainArgs ~0.3 % of the replayed forwards came out wrong by 1e-3..4e-3 (a consumer on
// another XCD read the previous replay's lines from its L2); with the pointers repeated as arguments 0 of 6000.
// Stream launches were never affected (every dispatch there carries agent-scope acquire / release fences).
#define MVSN_VIS10 const void *, const void *, const void *, const void *, const void *, const void *, const void *, \
                   const void *, const void *, const void *

// ---- GroupNorm statistics of ONE sample from the producing convolution's records -------------------------------------
// Combination of the per-record (count, mean, M2) in double by a 256-thread workgroup; a thread reads whole 48-byte
// records (all four groups), one pass:  N = sum c,  S = sum c*mean,  Q = sum (M2 + c*mean^2)  ->  var = Q/N - (S/N)^2
// (the subtraction is done in double on sums of fp32 data: ~1e-16 relative, far below the fp32 inputs' own rounding).
// `out8` = [group][mean, rstd], global or LDS; written by threads 0..3, NOT yet published to the other threads.
// Every caller runs this very code with 256 threads, so a statistic is the same bit pattern wherever it is formed
// (mvsn_groupnorm_finalize's own launch, or a consumer that was handed the records: `gn_stats_here`).
constexpr float GN_FINALIZE_EPS = 1e-5f;
// ONE: just group `only`'s statistics are formed (workgroup-uniform; same operations in the same order for it).
template <bool ONE = false>
__device__ __forceinline__ void gn_finalize_block(const float *__restrict__ records, int tiles, float *out8,
                                                  int only = -1) {
  const int tid = threadIdx.x;
  const floatx4 *p = reinterpret_cast<const floatx4 *>(records);
  double acc[4][3];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g][0] = acc[g][1] = acc[g][2] = 0.0;
  auto add = [&](const floatx4 &a, const floatx4 &b, const floatx4 &c) {
    const float e[12] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3]};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (ONE && g != only) continue;
      const double cnt = (double)e[g * 3], mean = (double)e[g * 3 + 1];
      acc[g][0] += cnt;
      acc[g][1] += cnt * mean;
      acc[g][2] += (double)e[g * 3 + 2] + cnt * mean * mean;
    }
  };
  // eight records in flight per thread: a level-0 refiner layer leaves 8192 records per sample, and with one record
  // per iteration the launch was 32 dependent round trips long (25-32 us, 45 launches per forward)
  constexpr int GF_U = 8;
  int t = tid;
  for (; t + (GF_U - 1) * 256 < tiles; t += GF_U * 256) {
    floatx4 r[GF_U][3];
#pragma unroll
    for (int u = 0; u < GF_U; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) r[u][k] = p[(size_t)(t + u * 256) * 3 + k];
#pragma unroll
    for (int u = 0; u < GF_U; ++u) add(r[u][0], r[u][1], r[u][2]);   // (same order as the one-by-one loop)
  }
  for (; t < tiles; t += 256) add(p[(size_t)t * 3], p[(size_t)t * 3 + 1], p[(size_t)t * 3 + 2]);
  __shared__ double gn_red[4][12];   // per wave
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (ONE && g != only) continue;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double v = acc[g][k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      acc[g][k] = v;
    }
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) gn_red[tid >> 6][g * 3 + k] = acc[g][k];
  }
  __syncthreads();
  if (tid < 4 && (!ONE || tid == only)) {
    const int g = tid;
    const double N = gn_red[0][g * 3] + gn_red[1][g * 3] + gn_red[2][g * 3] + gn_red[3][g * 3];
    const double S = gn_red[0][g * 3 + 1] + gn_red[1][g * 3 + 1] + gn_red[2][g * 3 + 1] + gn_red[3][g * 3 + 1];
    const double Q = gn_red[0][g * 3 + 2] + gn_red[1][g * 3 + 2] + gn_red[2][g * 3 + 2] + gn_red[3][g * 3 + 2];
    const double mean = S / N;
    double var = Q / N - mean * mean;
    if (var < 0.0) var = 0.0;
    out8[g * 2 + 0] = (float)mean;
    out8[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)GN_FINALIZE_EPS));
  }
}

// First stage of a SPLIT finalize (mvsn_groupnorm_finalize_split: many records per sample, few samples -- one workgroup
// per sample reads megabytes alone): the same accumulation over a slice of a sample's records, the three double sums per
// group written out instead of finished.  out12 = [group][N, S, Q].
__device__ __forceinline__ void gn_partial_block(const float *__restrict__ records, int tiles, double *out12) {
  const int tid = threadIdx.x;
  const floatx4 *p = reinterpret_cast<const floatx4 *>(records);
  double acc[4][3];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g][0] = acc[g][1] = acc[g][2] = 0.0;
  auto add = [&](const floatx4 &a, const floatx4 &b, const floatx4 &c) {
    const float e[12] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3]};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const double cnt = (double)e[g * 3], mean = (double)e[g * 3 + 1];
      acc[g][0] += cnt;
      acc[g][1] += cnt * mean;
      acc[g][2] += (double)e[g * 3 + 2] + cnt * mean * mean;
    }
  };
  constexpr int GF_U = 8;
  int t = tid;
  for (; t + (GF_U - 1) * 256 < tiles; t += GF_U * 256) {
    floatx4 r[GF_U][3];
#pragma unroll
    for (int u = 0; u < GF_U; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) r[u][k] = p[(size_t)(t + u * 256) * 3 + k];
#pragma unroll
    for (int u = 0; u < GF_U; ++u) add(r[u][0], r[u][1], r[u][2]);
  }
  for (; t < tiles; t += 256) add(p[(size_t)t * 3], p[(size_t)t * 3 + 1], p[(size_t)t * 3 + 2]);
  __shared__ double gp_red[4][12];   // per wave
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double v = acc[g][k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      acc[g][k] = v;
    }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) gp_red[tid >> 6][g * 3 + k] = acc[g][k];
  }
  __syncthreads();
  if (tid < 12) out12[tid] = gp_red[0][tid] + gp_red[1][tid] + gp_red[2][tid] + gp_red[3][tid];
}

// The statistics a consumer of sample n works with: `stats` is either the finalised (N,4,2) array (tiles == 0) or the
// producer's records (N, tiles, 4, 3) -- then this workgroup (256 threads, all of them must call) forms them itself,
// which saves the dependent mvsn_groupnorm_finalize launch in front of it (small batches: the launch costs more than
// re-reading a few hundred records per workgroup).  Returns 8 floats [group][mean, rstd]; `lds8` is the caller's.
template <bool ONE = false>
__device__ __forceinline__ const float *gn_stats_here(const float *__restrict__ stats, int tiles, int n, float *lds8,
                                                      int only = -1) {
  if (tiles == 0) return stats + (size_t)n * 8;
  gn_finalize_block<ONE>(stats + (size_t)n * tiles * 12, tiles, lds8, only);
  __syncthreads();
  return lds8;
}

}  // namespace mvsn
