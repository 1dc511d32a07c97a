// Internal interface between mvsn_conv.hip (C-ABI entry points) and the 3 x bf16 split kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/mvsn_hip.h"

namespace mvsn {

struct Bf16x3Geom {
  int n, D, H, W, dil, kd;
  int nprod;            // 3: hi/lo split products (fp32-equivalent); 1: plain bf16 operands
  int tzo, ty;          // output planes / rows per workgroup (32 columns)
  int HY, HX;           // staged plane tile
  int ntz, nty, ntx, tiles;
  size_t lds_bytes;
};

bool bf16x3_geom(const mvsn_conv_desc *d, Bf16x3Geom *g);
int bf16x3_pack(const mvsn_conv_desc *d, const float *weight, void *packed, hipStream_t stream);
int bf16x3_launch(const Bf16x3Geom &g, const float *in, const void *wpk, const float *bias, const float *in_stats,
                  const float *in_gamma, const float *in_beta, float *out, float *out_partials, hipStream_t stream);
int bf16_storage_launch(const Bf16x3Geom &g, const void *in, bool in16, const void *wpk, const float *bias,
                        const float *in_stats, const float *in_gamma, const float *in_beta, void *out, bool out16,
                        float *out_partials, hipStream_t stream);
int bf16_selftest(hipStream_t stream, int *dbad);

}  // namespace mvsn
