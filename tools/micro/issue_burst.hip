// Issue model, part 3 (tuning aid): the production step shape -- 64 MFMAs (16x16x4) per wave between two workgroup
// barriers, 2 waves per SIMD (512 threads), per MFMA NV VALU + 1 ds_read_b32 (+ raw reads) -- with the auxiliary work
//   MODE 0  fine: after every 2 MFMAs 2 ds_read + 2*NV VALU (what conv_wino_kernel does today)
//   MODE 1  burst G: G MFMAs back to back, then G ds_read + G*NV VALU
//   MODE 2  burst G with the ds_reads INSIDE the burst (one behind each MFMA), VALU behind the burst
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE, int G, int NV, int BAR>
__global__ __launch_bounds__(512, 2) void k(const float *__restrict__ in, float *__restrict__ out, int iters) {
  __shared__ float smem[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 512) smem[i] = in[i];
  __syncthreads();
  float a = in[tid], b = in[tid + 64];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = in[tid + i];
  float l[4] = {0.f, 0.f, 0.f, 0.f};
  const unsigned lp = (unsigned)(size_t)((__attribute__((address_space(3))) float *)smem) + (tid & 63) * 4;
  floatx4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {   // one step: 64 MFMAs
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
          asm volatile("ds_read_b32 %0, %1" : "=v"(l[0]) : "v"(lp) : "memory");
          asm volatile("ds_read_b32 %0, %1" : "=v"(l[1]) : "v"(lp) : "memory");
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i + 1], 0, 0, 0);
#pragma unroll
          for (int v = 0; v < 2 * NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[v & 7]) : "v"(a));
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 32 / G; ++g) {
#pragma unroll
          for (int i = 0; i < G; ++i) {
            acc[g * G + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g * G + i], 0, 0, 0);
            if (MODE == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(l[i & 3]) : "v"(lp) : "memory");
          }
          if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < G; ++q) asm volatile("ds_read_b32 %0, %1" : "=v"(l[q & 3]) : "v"(lp) : "memory");
          }
#pragma unroll
          for (int v = 0; v < NV * G; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[v & 7]) : "v"(a));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (BAR) asm volatile("s_barrier" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 512 + tid] = s + l[0] + l[1] + l[2] + l[3];
}

template <int MODE, int G, int NV, int BAR>
void run(const float *in, float *out) {
  const int iters = 3000;
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, G, NV, BAR>), dim3(256), dim3(512), 0, 0, in, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double mf = 64.0 * 2048 * 8 * 256.0 * iters;
  printf("mode %d burst %2d  VALU %d per MFMA  barrier %d: %.3f ms  pipe %.1f %%  (%.1f cycles per MFMA)\n", MODE, G, NV, BAR, best,
         100.0 * mf / (best * 1e-3) / 157.3e12, best * 1e-3 * 2.4e9 / iters / 64 / 2);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 8192 * 4);
  hipMalloc(&out, 256 * 1024 * 4);
  hipMemset(in, 0, 8192 * 4);
  run<0, 2, 1, 0>(in, out); run<0, 2, 1, 1>(in, out); run<0, 2, 2, 1>(in, out);
  run<1, 4, 1, 1>(in, out); run<1, 8, 1, 1>(in, out); run<1, 16, 1, 1>(in, out); run<1, 32, 1, 1>(in, out);
  run<1, 8, 2, 1>(in, out); run<1, 16, 2, 1>(in, out); run<1, 32, 2, 1>(in, out); run<1, 8, 2, 0>(in, out);
  run<2, 8, 1, 1>(in, out); run<2, 16, 1, 1>(in, out); run<2, 8, 2, 1>(in, out); run<2, 16, 2, 1>(in, out); run<2, 32, 2, 1>(in, out);
  return 0;
}
