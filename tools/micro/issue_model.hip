// Issue model of the fp32 matrix pipe next to VALU / LDS work (tuning aid): a loop of MFMAs on independent accumulators
// with NV independent VALU instructions and NL LDS reads behind each one, 1 or 2 waves per SIMD, 16x16x4 (32 pipe
// cycles) or 32x32x2 (64).  Prints the pipe utilisation: where it drops below 100 % the fillers are NOT free.
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/issue_model.hip -o tools/micro/issue_model && tools/micro/issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int BIG, int NV, int NL, int THREADS>
__global__ __launch_bounds__(THREADS, (THREADS + 255) / 256) void k(const float *__restrict__ in, float *__restrict__ out, int iters) {
  __shared__ float smem[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += THREADS) smem[i] = in[i];
  __syncthreads();
  float a = in[tid], b = in[tid + 64];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = in[tid + i];
  float l[4] = {0.f, 0.f, 0.f, 0.f};
  const unsigned lp = (unsigned)(size_t)((__attribute__((address_space(3))) float *)smem) + (tid & 63) * 4;
  if constexpr (BIG) {
    floatx16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(i + v) & 7]) : "v"(a));
#pragma unroll
        for (int q = 0; q < NL; ++q) asm volatile("ds_read_b32 %0, %1" : "=v"(l[q & 3]) : "v"(lp) : "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) s += acc[i][j];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * THREADS + tid] = s + l[0] + l[1] + l[2] + l[3];
  } else {
    floatx4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(i + v) & 7]) : "v"(a));
#pragma unroll
        for (int q = 0; q < NL; ++q) asm volatile("ds_read_b32 %0, %1" : "=v"(l[q & 3]) : "v"(lp) : "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * THREADS + tid] = s + l[0] + l[1] + l[2] + l[3];
  }
}

template <int BIG, int NV, int NL, int THREADS>
void run(const float *in, float *out) {
  const int iters = BIG ? 8000 : 8000;
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<BIG, NV, NL, THREADS>), dim3(256), dim3(THREADS), 0, 0, in, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double mf = (BIG ? 8.0 * 4096 : 16.0 * 2048) * (THREADS / 64) * 256.0 * iters;
  const double cyc = best * 1e-3 * 2.4e9 / iters / (BIG ? 8 : 16) / (THREADS / 256.0);   // cycles per MFMA and SIMD at 2.4 GHz
  printf("%s %d wave/SIMD  VALU %d LDS %d per MFMA: %.3f ms  pipe %.1f %%  (%.1f cycles @2.4GHz per MFMA)\n", BIG ? "32x32x2" : "16x16x4", THREADS / 256, NV, NL,
         best, 100.0 * mf / (best * 1e-3) / 157.3e12, cyc);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 8192 * 4);
  hipMalloc(&out, 256 * 512 * 4);
  hipMemset(in, 0, 8192 * 4);
  run<0, 0, 0, 1024>(in, out); run<0, 1, 0, 1024>(in, out); run<0, 2, 0, 1024>(in, out); run<0, 4, 0, 1024>(in, out); run<0, 6, 0, 1024>(in, out);
  run<0, 0, 1, 1024>(in, out); run<0, 1, 1, 1024>(in, out); run<0, 2, 1, 1024>(in, out); run<0, 4, 1, 1024>(in, out); run<0, 2, 2, 1024>(in, out); run<0, 4, 2, 1024>(in, out);
  run<0, 2, 0, 768>(in, out); run<0, 2, 1, 768>(in, out); run<0, 4, 1, 768>(in, out);
  run<0, 2, 1, 512>(in, out); run<0, 4, 1, 512>(in, out);
  return 0;
}
