// Issue model, part 2 (tuning aid): does it matter how the VALU / LDS work is grouped between the multiplies?
// 16 MFMAs (16x16x4) per iteration on independent accumulators, NV VALU + NL ds_read_b32 per MFMA issued in groups:
// after every G MFMAs come G*NV VALU and G*NL LDS reads.  Two waves per SIMD (production shape).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int G, int NV, int NL, int THREADS, int PRIO>
__global__ __launch_bounds__(THREADS, THREADS / 256) void k(const float *__restrict__ in, float *__restrict__ out, int iters) {
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
  floatx4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16 / G; ++g) {
      if (PRIO) asm volatile("s_setprio 3");
#pragma unroll
      for (int i = 0; i < G; ++i) acc[g * G + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g * G + i], 0, 0, 0);
      if (PRIO) asm volatile("s_setprio 0");
#pragma unroll
      for (int v = 0; v < NV * G; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[v & 7]) : "v"(a));
#pragma unroll
      for (int q = 0; q < NL * G; ++q) asm volatile("ds_read_b32 %0, %1" : "=v"(l[q & 3]) : "v"(lp) : "memory");
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

template <int G, int NV, int NL, int THREADS, int PRIO = 0>
void run(const float *in, float *out) {
  const int iters = 8000;
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<G, NV, NL, THREADS, PRIO>), dim3(256), dim3(THREADS), 0, 0, in, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double mf = 16.0 * 2048 * (THREADS / 64) * 256.0 * iters;
  printf("group %2d  %d wave/SIMD  VALU %d LDS %d per MFMA prio %d: %.3f ms  pipe %.1f %%  (%.1f cycles per MFMA)\n", G, THREADS / 256, NV, NL, PRIO, best,
         100.0 * mf / (best * 1e-3) / 157.3e12, best * 1e-3 * 2.4e9 / iters / 16 / (THREADS / 256.0));
}

int main() {
  float *in, *out;
  hipMalloc(&in, 8192 * 4);
  hipMalloc(&out, 256 * 1024 * 4);
  hipMemset(in, 0, 8192 * 4);
  run<1, 2, 0, 512>(in, out); run<2, 2, 0, 512>(in, out); run<4, 2, 0, 512>(in, out); run<8, 2, 0, 512>(in, out); run<16, 2, 0, 512>(in, out);
  run<1, 2, 1, 512>(in, out); run<2, 2, 1, 512>(in, out); run<4, 2, 1, 512>(in, out); run<8, 2, 1, 512>(in, out); run<16, 2, 1, 512>(in, out);
  run<1, 4, 1, 512>(in, out); run<4, 4, 1, 512>(in, out); run<16, 4, 1, 512>(in, out);
  run<4, 2, 1, 512, 1>(in, out); run<16, 2, 1, 512, 1>(in, out); run<16, 4, 1, 512, 1>(in, out);
  run<1, 2, 1, 256>(in, out); run<4, 2, 1, 256>(in, out); run<16, 2, 1, 256>(in, out);
  run<4, 2, 1, 1024>(in, out); run<16, 2, 1, 1024>(in, out); run<16, 4, 1, 1024>(in, out);
  return 0;
}
