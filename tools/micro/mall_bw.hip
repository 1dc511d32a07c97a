// Infinity-Cache (MALL) residency probe: bandwidth of a streaming copy / in-place update / read as a function of the
// working set, launches back to back on one stream (what a depth-first tower over a small group of images would see).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mall_bw.hip -o tools/micro/mall_bw && tools/micro/mall_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void copy_k(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
__global__ void axpy_k(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {   // b = b*0.5 + a: 2 reads 1 write
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { float4 x = a[i], y = b[i]; y.x = y.x * .5f + x.x; y.y = y.y * .5f + x.y; y.z = y.z * .5f + x.z; y.w = y.w * .5f + x.w; b[i] = y; }
}
__global__ void read_k(const float4* __restrict__ a, float* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < n; i += st) { float4 x = a[i]; s += x.x + x.y + x.z + x.w; }
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const size_t MAXB = size_t(3) << 30;
    char* buf; float* out;
    hipMalloc(&buf, MAXB); hipMalloc(&out, 64);
    hipMemset(buf, 0, MAXB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<size_t> sizes;   // bytes per tensor
    for (size_t mb : {8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024}) sizes.push_back(mb << 20);
    printf("tensor_MB,copy_ws_MB,copy_GBps,axpy_GBps,read_GBps,chain3_GBps\n");
    for (size_t s : sizes) {
        size_t n = s / 16;
        float4 *a = (float4*)buf, *b = (float4*)(buf + s), *c = (float4*)(buf + 2 * s);
        int grid = 256 * 8, reps = (int)((size_t(8) << 30) / s); if (reps > 400) reps = 400; if (reps < 6) reps = 6;
        float ms, r[4];
        for (int w = 0; w < 3; w++) copy_k<<<grid, 256>>>(a, b, n);
        hipEventRecord(e0); for (int i = 0; i < reps; i++) copy_k<<<grid, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); r[0] = 2.0 * s * reps / (ms * 1e-3) / 1e9;
        hipEventRecord(e0); for (int i = 0; i < reps; i++) axpy_k<<<grid, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); r[1] = 3.0 * s * reps / (ms * 1e-3) / 1e9;
        hipEventRecord(e0); for (int i = 0; i < reps; i++) read_k<<<grid, 256>>>(a, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); r[2] = 1.0 * s * reps / (ms * 1e-3) / 1e9;
        // the residual-block pattern on three tensors: copy a->b (conv: read x, write r), axpy a->b (pass: read r, x; write), rotate
        hipEventRecord(e0);
        for (int i = 0; i < reps; i++) { copy_k<<<grid, 256>>>(a, b, n); axpy_k<<<grid, 256>>>(a, b, n); float4* t = a; a = b; b = c; c = t; }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); r[3] = 5.0 * s * reps / (ms * 1e-3) / 1e9;
        printf("%zu,%zu,%.0f,%.0f,%.0f,%.0f\n", s >> 20, (2 * s) >> 20, r[0], r[1], r[2], r[3]);
    }
    return 0;
}
