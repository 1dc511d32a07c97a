// Internal interface between mvsn_conv.hip (C-ABI entry points) and the Winograd F(2x2,3x3) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/mvsn_hip.h"

namespace mvsn {

struct WinoGeom {
  int n, cin, H, W, dil;
  int D;      // planes per sample (1 for the 2-D layers)
  bool vol;   // 3 x 3 x 3 layer (volume form)
  bool s2;    // 5 x 5 stride-2 layer on the input's four phases
  int wide;   // volume form on planes 32 < W <= 40 columns wide: 1 = tiles of 10 rows x the whole width, 2 = rolling strips
              // of six patch rows through the sample's planes (`tiles` = items per SAMPLE there; see conv_wino_kernel)
  int nty, ntx, tiles, nchunks;
  size_t packed_floats;
};
// work items (one per workgroup visit, 32 GroupNorm records each) of a sample
inline long wino_items(const WinoGeom &g) { return g.wide == 2 ? (long)g.tiles : (long)g.D * g.tiles; }

bool wino_geom(const mvsn_conv_desc *d, WinoGeom *g);
int wino_pack(const mvsn_conv_desc *d, const float *weight, float *packed, hipStream_t stream);
// input handed over as channel blocks: [0, cb0) from `in`, [cb0, cb0 + cb1) from in1, the rest from in2
struct WinoBlocks {
  int cb0, cb1;
  const float *in1, *in2;
  size_t bs0 = 0, cs0 = 0;   // block 0 strided (floats between samples / channels); 0: contiguous
};
constexpr int WINO_UFLOATS_PER_CHUNK = 16 * 128;   // packed floats per 4 input channels
int wino_pack_2d(const float *weight, int cin, float *packed, hipStream_t stream);
int wino_launch(const WinoGeom &g, const float *in, const float *upk, const float *bias, const float *in_stats,
                const float *in_gamma, const float *in_beta, float *out, float *out_partials, hipStream_t stream,
                const WinoBlocks *blocks = nullptr, const mvsn_apply_job *job = nullptr);
// a normalise / activate / add job can travel inside this layer's launch (mvsn_conv_forward_carry)
bool wino_can_carry(const WinoGeom &g, const mvsn_apply_job *job);

}  // namespace mvsn
