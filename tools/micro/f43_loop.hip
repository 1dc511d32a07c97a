// Micro-benchmark (tuning aid, not part of the library): the inner loop of a Winograd F(4x4,3x3) layer with ONE wave per
// SIMD -- 36 coefficients x 2 cout tiles = 288 accumulator registers (256 of them AGPRs), per 4-channel k-step a 6x6
// window read from LDS, B^T d B in registers (168 VALU), 72 MFMAs with U fragments from LDS -- against the F(2x2,3x3)
// loop of the production kernel's shape (two waves per SIMD, 128 accumulators, 32 MFMAs per k-step).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/f43_loop.hip -o /tmp/f43_loop && /tmp/f43_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ floatx4 mfma(float a, float b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void bt6(const float (&d)[6], float (&t)[6]) {
  t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  t[1] = (d[3] + d[4]) - 4.f * (d[1] + d[2]);
  t[2] = (d[4] - d[3]) + 4.f * (d[1] - d[2]);
  t[3] = (d[4] - d[2]) + 2.f * (d[3] - d[1]);
  t[4] = (d[4] - d[2]) - 2.f * (d[3] - d[1]);
  t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

__global__ __launch_bounds__(256, 1) void f43(const float *__restrict__ in, float *__restrict__ out, int steps) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 24576; i += 256) smem[i] = in[i];
  __syncthreads();
  floatx4 acc[36][2];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i][0] = acc[i][1] = floatx4{0, 0, 0, 0};
  float v[36];
  for (int s = 0; s < steps; ++s) {
    float d[6][6];
    const float *w = smem + (s & 3) * 2560 + (lane >> 4) * 640 + (lane & 15) * 4;   // 6 rows x 72-float stride
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const floatx4 a = *reinterpret_cast<const floatx4 *>(w + i * 72);
      const float2 b = *reinterpret_cast<const float2 *>(w + i * 72 + 4);
      d[i][0] = a[0], d[i][1] = a[1], d[i][2] = a[2], d[i][3] = a[3], d[i][4] = b.x, d[i][5] = b.y;
    }
    float t[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float c[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
      float r[6];
      bt6(c, r);
#pragma unroll
      for (int i = 0; i < 6; ++i) t[i][j] = r[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float r[6];
      bt6(t[i], r);
#pragma unroll
      for (int j = 0; j < 6; ++j) v[i * 6 + j] = r[j];
    }
    const float *ub = smem + 10240 + (s & 1) * 4608 + lane;
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) {
      acc[xi][0] = mfma(v[xi], ub[xi * 128], acc[xi][0]);
      acc[xi][1] = mfma(v[xi], ub[xi * 128 + 64], acc[xi][1]);
    }
  }
  floatx4 sum = floatx4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 36; ++i) sum += acc[i][0] + acc[i][1] * 2.f;
  *reinterpret_cast<floatx4 *>(out + (blockIdx.x * 256 + tid) * 4) = sum;
}

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
    float t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0][j] - d[2][j], t[1][j] = d[1][j] + d[2][j], t[2][j] = d[2][j] - d[1][j], t[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i * 4 + 0] = t[i][0] - t[i][2], v[i * 4 + 1] = t[i][1] + t[i][2], v[i * 4 + 2] = t[i][2] - t[i][1], v[i * 4 + 3] = t[i][1] - t[i][3];
    }
    const float *ub = smem + 10240 + (s & 1) * 2048 + lane;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      acc[xi][0] = mfma(v[xi], ub[xi * 128], acc[xi][0]);
      acc[xi][1] = mfma(v[xi], ub[xi * 128 + 64], acc[xi][1]);
    }
  }
  floatx4 sum = floatx4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] * 2.f;
  *reinterpret_cast<floatx4 *>(out + (blockIdx.x * 512 + tid) * 4) = sum;
}

// F(2x2,3x3), two waves per SIMD, the k-step's 32 U fragment values read into registers BEFORE its multiplies
// (is the production loop's one-coefficient-ahead prefetch what limits the pipe?)
__global__ __launch_bounds__(512, 2) void f23p(const float *__restrict__ in, float *__restrict__ out, int steps) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 24576; i += 512) smem[i] = in[i];
  __syncthreads();
  floatx4 acc[16][2];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i][0] = acc[i][1] = floatx4{0, 0, 0, 0};
  float v[16], u[2][16][2];
  {
    const float *ub = smem + 10240 + lane;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[0][xi][0] = ub[xi * 128], u[0][xi][1] = ub[xi * 128 + 64];
  }
  for (int s = 0; s < steps; s += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float d[4][4];
      const float *w = smem + ((s + half) & 3) * 2560 + (lane >> 4) * 640 + (lane & 15) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float2 a = *reinterpret_cast<const float2 *>(w + i * 40), b = *reinterpret_cast<const float2 *>(w + i * 40 + 2);
        d[i][0] = a.x, d[i][1] = a.y, d[i][2] = b.x, d[i][3] = b.y;
      }
      const float *ub = smem + 10240 + ((half ^ 1) & 1) * 2048 + lane;   // next k-step's U
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) u[half ^ 1][xi][0] = ub[xi * 128], u[half ^ 1][xi][1] = ub[xi * 128 + 64];
      float t[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j], t[1][j] = d[1][j] + d[2][j], t[2][j] = d[2][j] - d[1][j], t[3][j] = d[1][j] - d[3][j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i * 4 + 0] = t[i][0] - t[i][2], v[i * 4 + 1] = t[i][1] + t[i][2], v[i * 4 + 2] = t[i][2] - t[i][1], v[i * 4 + 3] = t[i][1] - t[i][3];
      }
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        acc[xi][0] = mfma(v[xi], u[half][xi][0], acc[xi][0]);
        acc[xi][1] = mfma(v[xi], u[half][xi][1], acc[xi][1]);
      }
    }
  }
  floatx4 sum = floatx4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] * 2.f;
  *reinterpret_cast<floatx4 *>(out + (blockIdx.x * 512 + tid) * 4) = sum;
}

// F(2x2,3x3), volume form with the three depth taps of a plane sharing ONE input transform: one wave per SIMD, three
// accumulator sets (384 registers), per 4-channel k-step 48 transform VALU and 96 MFMAs
__global__ __launch_bounds__(256, 1) void f23x3(const float *__restrict__ in, float *__restrict__ out, int steps) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 24576; i += 256) smem[i] = in[i];
  __syncthreads();
  floatx4 acc[3][16][2];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i][0] = acc[k][i][1] = floatx4{0, 0, 0, 0};
  float v[16];
  for (int s = 0; s < steps; ++s) {
    float d[4][4];
    const float *w = smem + (s & 3) * 2560 + (lane >> 4) * 640 + (lane & 15) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 a = *reinterpret_cast<const float2 *>(w + i * 40), b = *reinterpret_cast<const float2 *>(w + i * 40 + 2);
      d[i][0] = a.x, d[i][1] = a.y, d[i][2] = b.x, d[i][3] = b.y;
    }
    float t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0][j] - d[2][j], t[1][j] = d[1][j] + d[2][j], t[2][j] = d[2][j] - d[1][j], t[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i * 4 + 0] = t[i][0] - t[i][2], v[i * 4 + 1] = t[i][1] + t[i][2], v[i * 4 + 2] = t[i][2] - t[i][1], v[i * 4 + 3] = t[i][1] - t[i][3];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float *ub = smem + 10240 + k * 2048 + lane;
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        acc[k][xi][0] = mfma(v[xi], ub[xi * 128], acc[k][xi][0]);
        acc[k][xi][1] = mfma(v[xi], ub[xi * 128 + 64], acc[k][xi][1]);
      }
    }
  }
  floatx4 sum = floatx4{0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += acc[k][i][0] + acc[k][i][1] * 2.f;
  *reinterpret_cast<floatx4 *>(out + (blockIdx.x * 256 + tid) * 4) = sum;
}

int main() {
  float *in, *out;
  hipMalloc(&in, 24576 * 4);
  hipMalloc(&out, 256 * 512 * 16);
  hipMemset(in, 0, 24576 * 4);
  hipFuncSetAttribute((const void *)f43, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipFuncSetAttribute((const void *)f23, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipFuncSetAttribute((const void *)f23x3, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipFuncSetAttribute((const void *)f23p, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  const int steps = 20000;
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  for (int which = 0; which < 4; ++which)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      if (which == 0) hipLaunchKernelGGL(f43, dim3(256), dim3(256), 98304, 0, in, out, steps);
      else if (which == 1) hipLaunchKernelGGL(f23, dim3(256), dim3(512), 98304, 0, in, out, steps);
      else if (which == 2) hipLaunchKernelGGL(f23x3, dim3(256), dim3(256), 98304, 0, in, out, steps);
      else hipLaunchKernelGGL(f23p, dim3(256), dim3(512), 98304, 0, in, out, steps);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      // outputs (x 32 couts) per k-step per SIMD: F(4,3) one wave x 16 patches x 16; F(2,3) two waves x 16 patches x 4
      // (third kernel: one wave x 16 patches x 4 outputs x 3 depth taps' worth of multiplies)
      const double outs = which == 0 ? 256.0 : (which == 2 ? 192.0 : 128.0), mf = which == 0 ? 72.0 : (which == 2 ? 96.0 : 64.0);
      printf("%s: %.3f ms  %.0f ns per k-step  %.2f ns per output (x32 couts x4 cin)  MFMA pipe %.1f %% of 157.3 TFLOP/s\n",
             which == 0 ? "F(4x4,3x3) 1 wave/SIMD " : (which == 1 ? "F(2x2,3x3) 2 waves/SIMD" : (which == 2 ? "F(2,3) x3 taps, 1 wave " : "F(2,3) 2 waves, U ahead ")), ms, ms * 1e6 / steps, ms * 1e6 / steps / outs,
             100.0 * (mf * 2048.0 * 4 * 256 * steps / (ms * 1e-3)) / 157.3e12);
    }
  return 0;
}
