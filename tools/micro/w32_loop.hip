// Micro-benchmark (tuning aid, not part of the library): can ONE wave per SIMD keep the fp32 matrix pipe fed when the
// multiplies are v_mfma_f32_32x32x2_f32 (64 pipe cycles each, 256 accumulator registers = 16 Winograd coefficients x
// 32 patches x 32 couts, AGPRs) instead of two waves per SIMD on 16x16x4 (32 cycles each, 128 accumulators)?
// Per 2-channel k-step: a 4x4 window from LDS, B^T d B in registers (32 VALU), 16 MFMAs with U fragments from LDS.
//   variants: plain | + epilogue-sized VALU filler every 64 k-steps (a 2-D tile) | the production 16x16x4 loop
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/w32_loop.hip -o tools/micro/w32_loop && tools/micro/w32_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void transform(const float (&d)[4][4], float (&v)[16]) {
  float t[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = d[0][j] - d[2][j], t[1][j] = d[1][j] + d[2][j], t[2][j] = d[2][j] - d[1][j], t[3][j] = d[1][j] - d[3][j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i * 4 + 0] = t[i][0] - t[i][2], v[i * 4 + 1] = t[i][1] + t[i][2], v[i * 4 + 2] = t[i][2] - t[i][1], v[i * 4 + 3] = t[i][1] - t[i][3];
  }
}

// MODE 0: transform of step s+1 issued as one block before the multiplies of step s (registers double-buffered)
// MODE 1: transform interleaved, two VALU behind each MFMA
template <int MODE>
__global__ __launch_bounds__(256, 1) void w32(const float *__restrict__ in, float *__restrict__ out, int steps) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 24576; i += 256) smem[i] = in[i];
  __syncthreads();
  floatx16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
  float v[16];
  {
    float d[4][4];
    const float *w = smem + (lane >> 5) * 640 + (lane & 31) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 a = *reinterpret_cast<const float2 *>(w + i * 72), b = *reinterpret_cast<const float2 *>(w + i * 72 + 2);
      d[i][0] = a.x, d[i][1] = a.y, d[i][2] = b.x, d[i][3] = b.y;
    }
    transform(d, v);
  }
  for (int s = 0; s < steps; ++s) {
    float d[4][4];
    const float *w = smem + ((s + 1) & 7) * 1280 + (lane >> 5) * 640 + (lane & 31) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 a = *reinterpret_cast<const float2 *>(w + i * 72), b = *reinterpret_cast<const float2 *>(w + i * 72 + 2);
      d[i][0] = a.x, d[i][1] = a.y, d[i][2] = b.x, d[i][3] = b.y;
    }
    const float *ub = smem + 10240 + (s & 7) * 1024 + lane;
    if constexpr (MODE == 0) {
      float vn[16];
      float u[2];
      u[0] = ub[0];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        if (xi + 1 < 16) u[(xi + 1) & 1] = ub[(xi + 1) * 64];
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[xi], u[xi & 1], acc[xi], 0, 0, 0);
        if (xi == 7) transform(d, vn);
      }
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) v[xi] = vn[xi];
    } else {
      float t[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[0][j] = d[0][j] - d[2][j];
      float u[2];
      u[0] = ub[0];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        const int ti = xi >> 2, tj = xi & 3;
        if (xi + 1 < 16) u[(xi + 1) & 1] = ub[(xi + 1) * 64];
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[xi], u[xi & 1], acc[xi], 0, 0, 0);
        if (ti < 3) t[ti + 1][tj] = ti == 0 ? d[1][tj] + d[2][tj] : (ti == 1 ? d[2][tj] - d[1][tj] : d[1][tj] - d[3][tj]);
        // v[xi] is free now: next step's coefficient (row ti of t is complete once tj == 3 of the previous row... use t[ti] lazily)
        if (tj == 3) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int x2 = ti * 4 + q;
            v[x2] = q == 0 ? t[ti][0] - t[ti][2] : (q == 1 ? t[ti][1] + t[ti][2] : (q == 2 ? t[ti][2] - t[ti][1] : t[ti][1] - t[ti][3]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  floatx4 sum = floatx4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) sum[k & 3] += acc[i][k] * (float)(i + 1);
  *reinterpret_cast<floatx4 *>(out + (blockIdx.x * 256 + tid) * 4) = sum;
}

// the same loop with a tile structure: every TILE k-steps the 256 accumulators go through the output transform
// (A^T m A: 24 adds per patch and cout, 16 patches per lane), a 16-byte store per 4 outputs and a sum / sum of squares,
// then restart from zero -- the whole wave's epilogue with the pipe idle (what one wave per SIMD cannot hide)
template <int TILE>
__global__ __launch_bounds__(256, 1) void w32t(const float *__restrict__ in, float *__restrict__ out, int steps) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 24576; i += 256) smem[i] = in[i];
  __syncthreads();
  floatx16 acc[16];
  float v[16];
  {
    float d[4][4];
    const float *w = smem + (lane >> 5) * 640 + (lane & 31) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 a = *reinterpret_cast<const float2 *>(w + i * 72), b = *reinterpret_cast<const float2 *>(w + i * 72 + 2);
      d[i][0] = a.x, d[i][1] = a.y, d[i][2] = b.x, d[i][3] = b.y;
    }
    transform(d, v);
  }
  float ssum = 0.f, qsum = 0.f;
  float *op = out + ((size_t)blockIdx.x * 256 + tid) * 4;
  for (int s0 = 0; s0 < steps; s0 += TILE) {
    auto kstep = [&](int s, auto firstc) {
      constexpr bool FIRST = decltype(firstc)::value;
      float d[4][4];
      const float *w = smem + ((s + 1) & 7) * 1280 + (lane >> 5) * 640 + (lane & 31) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float2 a = *reinterpret_cast<const float2 *>(w + i * 72), b = *reinterpret_cast<const float2 *>(w + i * 72 + 2);
        d[i][0] = a.x, d[i][1] = a.y, d[i][2] = b.x, d[i][3] = b.y;
      }
      const float *ub = smem + 10240 + (s & 7) * 1024 + lane;
      float vn[16];
      float u[2];
      u[0] = ub[0];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        if (xi + 1 < 16) u[(xi + 1) & 1] = ub[(xi + 1) * 64];
        if constexpr (FIRST) {
          floatx16 c;
#pragma unroll
          for (int k = 0; k < 16; ++k) c[k] = 0.f;
          acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[xi], u[xi & 1], c, 0, 0, 0);
        } else {
          acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[xi], u[xi & 1], acc[xi], 0, 0, 0);
        }
        if (xi == 7) transform(d, vn);
      }
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) v[xi] = vn[xi];
    };
    kstep(s0, std::true_type{});
#pragma unroll 1
    for (int s = s0 + 1; s < s0 + TILE; ++s) kstep(s, std::false_type{});
    // epilogue: 16 patches of one cout per lane
#pragma unroll
    for (int p = 0; p < 16; p += 2) {
      float y[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float s0v[4], s1v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s0v[j] = acc[j][p + e] + acc[4 + j][p + e] + acc[8 + j][p + e];
          s1v[j] = acc[4 + j][p + e] - acc[8 + j][p + e] - acc[12 + j][p + e];
        }
        y[e][0] = s0v[0] + s0v[1] + s0v[2], y[e][1] = s0v[1] - s0v[2] - s0v[3];
        y[e][2] = s1v[0] + s1v[1] + s1v[2], y[e][3] = s1v[1] - s1v[2] - s1v[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) ssum += y[e][k], qsum += y[e][k] * y[e][k];
      }
      __builtin_nontemporal_store(floatx4{y[0][0], y[0][1], y[1][0], y[1][1]}, reinterpret_cast<floatx4 *>(op + (size_t)(p * 2) * 65536 * 4));
      __builtin_nontemporal_store(floatx4{y[0][2], y[0][3], y[1][2], y[1][3]}, reinterpret_cast<floatx4 *>(op + (size_t)(p * 2 + 1) * 65536 * 4));
    }
  }
  op[0] = ssum + qsum;
}

// production shape for comparison: two waves per SIMD, 16x16x4, 128 accumulators
__global__ __launch_bounds__(512, 2) void f23(const float *__restrict__ in, float *__restrict__ out, int steps) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 24576; i += 512) smem[i] = in[i];
  __syncthreads();
  floatx4 acc[16][2];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i][0] = acc[i][1] = floatx4{0, 0, 0, 0};
  float v[16];
  for (int s = 0; s < steps; ++s) {
    float d[4][4];
    const float *w = smem + (s & 3) * 2560 + (lane >> 4) * 640 + (lane & 15) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 a = *reinterpret_cast<const float2 *>(w + i * 40), b = *reinterpret_cast<const float2 *>(w + i * 40 + 2);
      d[i][0] = a.x, d[i][1] = a.y, d[i][2] = b.x, d[i][3] = b.y;
    }
    transform(d, v);
    const float *ub = smem + 10240 + (s & 1) * 2048 + lane;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[xi], ub[xi * 128], acc[xi][0], 0, 0, 0);
      acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[xi], ub[xi * 128 + 64], acc[xi][1], 0, 0, 0);
    }
  }
  floatx4 sum = floatx4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] * 2.f;
  *reinterpret_cast<floatx4 *>(out + (blockIdx.x * 512 + tid) * 4) = sum;
}

int main() {
  float *in, *out;
  hipMalloc(&in, 24576 * 4);
  hipMalloc(&out, (size_t)65536 * 4 * 4 * 34);
  hipMemset(in, 0, 24576 * 4);
  const void *fs[] = {(const void *)w32<0>, (const void *)w32<1>, (const void *)w32t<16>, (const void *)w32t<48>, (const void *)f23};
  for (auto f : fs) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  const int steps = 19200;
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  const char *names[] = {"32x32x2 1 wave/SIMD, transform block", "32x32x2 1 wave/SIMD, interleaved   ", "32x32x2 + epilogue / 16 k-steps (2-D)", "32x32x2 + epilogue / 48 k-steps (vol)", "16x16x4 2 waves/SIMD (production)   "};
  for (int which = 0; which < 5; ++which)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      const int st = which == 4 ? steps / 2 : steps;   // production k-step = 4 channels, the others 2
      if (which == 0) hipLaunchKernelGGL(w32<0>, dim3(256), dim3(256), 98304, 0, in, out, st);
      else if (which == 1) hipLaunchKernelGGL(w32<1>, dim3(256), dim3(256), 98304, 0, in, out, st);
      else if (which == 2) hipLaunchKernelGGL(w32t<16>, dim3(256), dim3(256), 98304, 0, in, out, st);
      else if (which == 3) hipLaunchKernelGGL(w32t<48>, dim3(256), dim3(256), 98304, 0, in, out, st);
      else hipLaunchKernelGGL(f23, dim3(256), dim3(512), 98304, 0, in, out, st);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      // flops per SIMD and k-step: 16 MFMAs x 4096 (one wave) | 2 waves x 32 MFMAs x 2048
      const double fl = (which == 4 ? 2.0 * 32 * 2048 : 16.0 * 4096) * 4 * 256 * st;
      printf("%s: %.3f ms  MFMA pipe %.1f %% of 157.3 TFLOP/s\n", names[which], ms, 100.0 * fl / (ms * 1e-3) / 157.3e12);
    }
  return 0;
}
