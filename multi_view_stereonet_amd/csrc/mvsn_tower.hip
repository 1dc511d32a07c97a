// Plane-resident residual towers on the 16x32 coarse grid (see include/mvsn_hip.h: mvsn_tower_16x32).
//
// At the coarsest level a whole activation tensor of one sample is 32 channels x 512 pixels = 64 KB: it fits one CU's
// LDS next to one layer of transformed weights, exactly like the incremental chain's state.  The launch-per-layer form
// (conv -> finalize -> normalise/activate/add, mvsn_conv_forward + mvsn_groupnorm_*) spends ~20 us per convolution
// launch and ~7 us per statistics / elementwise launch on such a plane whatever the batch -- 8 + 7 + 6 dependent
// launches for a level-4 refiner (IDepthmapRefiner.forward, multi_view_stereonet.py:468-484), 6 + 6 + 6 + 1 for the
// extractor's residual stack (FeatureNetwork.forward :121-129) -- and at batch 1 those ~45 launches are a tenth of
// the forward.  Here ONE persistent workgroup per sample walks the whole tower with the activations resident:
//
//   [head]   x0 = LReLU(GN(conv3x3(x_in) + b))                      (refiner: x_in = [image 3 | features 32 | prior*fx])
//   blocks   x  = x + LReLU(GN(conv3x3_dilated(x) + b))             (utils/resnet.py:93-109; dilations 1,2,4,8,1,1)
//   tail     features = conv3x3(x) + b                              (extractor, :129)
//        or  idepth   = relu(prior*fx + conv3x3_{32->1}(x) + b)/fx  (refiner :482-484 with the gain of :607-611)
//
// Arithmetic: Winograd F(2x2,3x3) on v_mfma_f32_16x16x4_f32 exactly as chain_wino_kernel (A = U_xi 16 couts x 4 cins,
// B = V_xi 4 cins x 16 patches, the input transform in registers, two xi halves per layer; 512 threads = 8 waves, wave
// w = patch row w, both cout tiles).  A dilated layer is DIL x DIL interleaved dilation-1 problems: lane (k, p) of wave w
// owns the "patch" of outputs (ya + a DIL, xa + b DIL), ya = (w / DIL) 2 DIL + w % DIL, xa likewise from p -- on 16x32
// every dilation of {1,2,4,8} tiles the plane with exactly 8 x 16 such patches.  Its 4x4 window is read with sixteen
// per-lane offsets (one zero slot per channel plane stands in for every tap outside the image), so one code path
// serves all dilations.  GroupNorm is the exact two-pass form (mean, then sum (x - mean)^2) over the workgroup:
// the whole sample lives here, no statistics ever leave the CU.  The next layer's transformed weights (64 / 72 KB,
// L2 hits) are fetched by LDS-DMA behind the barrier that ends a layer's multiplies and land under its GroupNorm.
//
// LDS (floats): U 18432 | params 96 x layers + tail | red 128 | planes 36 x 544 (pixel p at p, zero slot at 512;
// channel c is skewed by one float when c & 2, so that the four channels of a k-step read disjoint bank sets).
#include "mvsn_common.h"

namespace mvsn {

constexpr int TW_THREADS = 512, TW_WAVES = 8, TW_ROWS = 16, TW_COLS = 32, TW_P = 512;
constexpr int TW_CS = 544, TW_ZERO = 512;
constexpr int TW_UFLOATS = 9 * 2048;
constexpr int TW_MAX_BLOCKS = 6;
constexpr int TW_PARAMS = 96 * (TW_MAX_BLOCKS + 1) + 320;     // [bias | gamma | beta] per layer, then the tail's
constexpr int TW_RED = 128;
constexpr int TW_LDS_FLOATS = TW_UFLOATS + TW_PARAMS + TW_RED + 36 * TW_CS;
constexpr float TW_GN_EPS = 1e-5f;

#define TW_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define TW_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

__device__ __forceinline__ void tw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int tw_chan(int c) { return c * TW_CS + ((c >> 1) & 1); }

struct TowerArgs {
  const float *in[3];      // input channel blocks, (n_b, c_b, 512) each; sample n reads block b at n % mod[b]
  int c[3], mod[3];
  const float *scale;      // optional: block `scale_block` is multiplied by scale[n % scale_mod] while it is loaded
  int scale_mod, scale_block;
  int head_chunks;         // 0 = no head, else k-steps of the head layer (9 for 36 input channels)
  int n_blocks;            // residual blocks
  int dil[TW_MAX_BLOCKS];
  const float *U;          // transformed weights, chain layout, layer after layer (head, blocks, [tail mode 0])
  const float *params;     // [bias 32 | gamma 32 | beta 32] per layer (head, blocks), then the tail's
  int tail_mode;           // 0: conv 32->32 + bias -> out (n, 32, 512); 1: conv 32->1 + refiner epilogue -> out (n, 1, 512)
  const float *prior;      // tail 1: (n, 512)
  const float *fx;         // tail 1: fx[n % fx_mod]
  int fx_mod;
  float *out;
};

// one 3x3 layer: acc[ct][xi] (+)= U_xi * V_xi over NC k-steps of 4 input channels; the window through `woff`
template <int NC>
__device__ __forceinline__ void tower_layer(const float *__restrict__ planes, const float *__restrict__ U,
                                            const int (&woff)[16], int lane, float (&y)[2][4][4]) {
  const int k = lane >> 4;
  const float *ub = U + lane * 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    floatx4 acc[2][8];
    float d[2][3][4];
    floatx4 u[2][4];
    auto fetch = [&](int buf, int c4) {
      const float *cp = planes + tw_chan(c4 * 4 + k);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[buf][i][j] = cp[woff[(half + i) * 4 + j]];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int xq = 0; xq < 2; ++xq)
          u[buf][ct * 2 + xq] = *reinterpret_cast<const floatx4 *>(ub + ((c4 * 2 + ct) * 4 + half * 2 + xq) * 256);
    };
    fetch(0, 0);
#pragma unroll
    for (int c4 = 0; c4 < NC; ++c4) {
      const int cur = c4 & 1;
      float t[2][4], v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (half == 0) {
          t[0][j] = d[cur][0][j] - d[cur][2][j];
          t[1][j] = d[cur][1][j] + d[cur][2][j];
        } else {
          t[0][j] = d[cur][1][j] - d[cur][0][j];
          t[1][j] = d[cur][0][j] - d[cur][2][j];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        v[i * 4 + 0] = t[i][0] - t[i][2];
        v[i * 4 + 1] = t[i][1] + t[i][2];
        v[i * 4 + 2] = t[i][2] - t[i][1];
        v[i * 4 + 3] = t[i][1] - t[i][3];
      }
      if (c4 + 1 < NC) fetch(cur ^ 1, c4 + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int xq = 0; xq < 2; ++xq)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const floatx4 c0 = c4 == 0 ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[ct][xq * 4 + j];
            acc[ct][xq * 4 + j] = mfma16x16x4(u[cur][ct * 2 + xq][j], v[xq * 4 + j], c0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (half == 0) {
            s0[j] = acc[ct][j][r] + acc[ct][4 + j][r];
            s1[j] = acc[ct][4 + j][r];
          } else {
            s0[j] = acc[ct][j][r];
            s1[j] = -acc[ct][j][r] - acc[ct][4 + j][r];
          }
        }
        const float y0 = s0[0] + s0[1] + s0[2], y1 = s0[1] - s0[2] - s0[3];
        const float y2 = s1[0] + s1[1] + s1[2], y3 = s1[1] - s1[2] - s1[3];
        if (half == 0) y[ct][r][0] = y0, y[ct][r][1] = y1, y[ct][r][2] = y2, y[ct][r][3] = y3;
        else y[ct][r][0] += y0, y[ct][r][1] += y1, y[ct][r][2] += y2, y[ct][r][3] += y3;
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// sums of two values over the 32 lanes of each half-wave (totals in lanes 16..31 / 48..63)
__device__ __forceinline__ void tw_half_wave_sums(float (&s)[2]) {
#pragma unroll
  for (int k = 0; k < 2; ++k) s[k] += dpp_mov<0xB1>(s[k]);
#pragma unroll
  for (int k = 0; k < 2; ++k) s[k] += dpp_mov<0x4E>(s[k]);
#pragma unroll
  for (int k = 0; k < 2; ++k) s[k] += dpp_mov<0x141>(s[k]);
#pragma unroll
  for (int k = 0; k < 2; ++k) s[k] += dpp_mov<0x140>(s[k]);
#pragma unroll
  for (int k = 0; k < 2; ++k)
    s[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s[k]), 0x142, 0xA, 0xF, false));
}

__global__ __launch_bounds__(TW_THREADS) void tower_kernel(TowerArgs a, MVSN_VIS10) {   // (MVSN_VIS10: mvsn_common.h)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x;
  float *U = smem;
  float *params = U + TW_UFLOATS;
  float *red = params + TW_PARAMS;
  float *planes = red + TW_RED;

  const int lane16 = lane * 16;
  auto dma_u = [&](const float *src, int nchunks) {   // 1 KB runs, wave w takes runs w, w + 8, ...
    const int runs = nchunks * 8;
    const char *base = reinterpret_cast<const char *>(src + (size_t)wave * 256);
    for (int run = wave, i = 0; run < runs; run += TW_WAVES, ++i)
      __builtin_amdgcn_global_load_lds(TW_GPTR(base + (size_t)i * (TW_WAVES * 1024) + (unsigned)lane16),
                                       TW_LPTR(U + run * 256), 16, 0, 0);
  };
  auto dma_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // ---- set-up: first layer's weights, parameters, the input planes -------------------------------------
  const int n_layers = (a.head_chunks ? 1 : 0) + a.n_blocks;
  const float *up = a.U;
  dma_u(up, a.head_chunks ? a.head_chunks : 8);
  up += (size_t)(a.head_chunks ? a.head_chunks : 8) * 2048;
  const int n_params = 96 * n_layers + (a.tail_mode == 0 ? 32 : 32 * 9 + 1);
  for (int i = tid; i < n_params; i += TW_THREADS) params[i] = a.params[i];
  for (int i = tid; i < 36 * (TW_CS - TW_P); i += TW_THREADS) {       // zero slots (and the skew padding) of every plane
    const int c = i / (TW_CS - TW_P), o = i - c * (TW_CS - TW_P);
    // NOT index 512 of a skewed plane: that is its pixel 511 (tw_chan adds 1), which another wave's input load below
    // writes with no barrier in between.  Zeroing it here lost the race about once in 10^4 launches when this wave's
    // parameter loads (behind the U DMA in the vmcnt queue) came back late: a feature map with one pixel of every
    // other channel pair zeroed (tools/soak.py; HISTORY 11.5).
    if (o == 0 && ((c >> 1) & 1)) continue;
    planes[c * TW_CS + TW_P + o] = 0.0f;
  }
  {
    int cbase = 0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      if (a.c[b] <= 0) continue;
      const float *src = a.in[b] + (size_t)(n % a.mod[b]) * a.c[b] * TW_P;
      const float sc = (a.scale && b == a.scale_block) ? a.scale[n % a.scale_mod] : 1.0f;
      for (int i = tid; i < a.c[b] * (TW_P / 4); i += TW_THREADS) {
        const int c = i / (TW_P / 4), p4 = (i - c * (TW_P / 4)) * 4;
        const floatx4 v = *reinterpret_cast<const floatx4 *>(src + (size_t)c * TW_P + p4);
        float *dst = planes + tw_chan(cbase + c) + p4;
        dst[0] = v[0] * sc, dst[1] = v[1] * sc, dst[2] = v[2] * sc, dst[3] = v[3] * sc;
      }
      cbase += a.c[b];
    }
    // channels up to a multiple of 4 are zero (K padding of the head layer)
    for (int c = cbase; c < ((cbase + 3) & ~3); ++c)
      for (int i = tid; i < TW_P; i += TW_THREADS) planes[tw_chan(c) + i] = 0.0f;
  }

  const int p = lane & 15, k = lane >> 4;
  const int cb = k * 4;                                    // this lane's couts: ct*16 + cb + r
  const float inv_n = 1.0f / (8.0f * (float)TW_P);
  int woff[16], ooff[4];
  auto set_dilation = [&](int dil) {
    const int ya = (wave / dil) * 2 * dil + wave % dil, xa = (p / dil) * 2 * dil + p % dil;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = ya + (i - 1) * dil, c = xa + (j - 1) * dil;
        woff[i * 4 + j] = (r >= 0 && r < TW_ROWS && c >= 0 && c < TW_COLS) ? r * TW_COLS + c : TW_ZERO;
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) ooff[e] = (ya + (e >> 1) * dil) * TW_COLS + xa + (e & 1) * dil;
  };

  // y + bias -> LeakyReLU(GroupNorm(.)): exact two-pass statistics over the workgroup (two barriers)
  auto groupnorm_lrelu = [&](float (&y)[2][4][4], const float *bias, const float *gamma, const float *beta) {
    float s[2] = {0.f, 0.f};   // per cout tile
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float b = bias[ct * 16 + cb + r];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[ct][r][e] += b;
          s[ct] += y[ct][r][e];
        }
      }
    tw_half_wave_sums(s);
    if ((lane & 31) == 16) {
      red[(wave * 4 + 0 + (lane >> 5)) * 1] = s[0];          // group ct*2 + half
      red[(wave * 4 + 2 + (lane >> 5)) * 1] = s[1];
    }
    tw_barrier();
    float mean[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int g = ct * 2 + (lane >> 5);
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < TW_WAVES; ++w) t += red[w * 4 + g];
      mean[ct] = t * inv_n;
    }
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dv = y[ct][r][e] - mean[ct];
          q[ct] += dv * dv;
        }
    tw_half_wave_sums(q);
    if ((lane & 31) == 16) {
      red[64 + wave * 4 + 0 + (lane >> 5)] = q[0];
      red[64 + wave * 4 + 2 + (lane >> 5)] = q[1];
    }
    tw_barrier();
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int g = ct * 2 + (lane >> 5);
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < TW_WAVES; ++w) t += red[64 + w * 4 + g];
      const float rstd = 1.0f / sqrtf(t * inv_n + TW_GN_EPS);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = ct * 16 + cb + r;
        const float sc = rstd * gamma[c];
        const float sh = beta[c] - mean[ct] * sc;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[ct][r][e] = lrelu02(y[ct][r][e] * sc + sh);
      }
    }
  };

  float y[2][4][4];
  const float *lp = params;          // this layer's [bias | gamma | beta]
  dma_landed();
  __syncthreads();

  // ---- head -------------------------------------------------------------------------------------------
  if (a.head_chunks) {
    set_dilation(1);
    tower_layer<9>(planes, U, woff, lane, y);
    tw_barrier();                     // planes and U free
    dma_u(up, 8);
    up += 8 * 2048;
    groupnorm_lrelu(y, lp, lp + 32, lp + 64);
    lp += 96;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float *dst = planes + tw_chan(ct * 16 + cb + r);
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[ooff[e]] = y[ct][r][e];
      }
    dma_landed();
    tw_barrier();
  }

  // ---- residual blocks --------------------------------------------------------------------------------
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    set_dilation(a.dil[blk]);
    tower_layer<8>(planes, U, woff, lane, y);
    tw_barrier();
    if (blk + 1 < a.n_blocks || a.tail_mode == 0) {
      dma_u(up, 8);
      up += 8 * 2048;
    }
    groupnorm_lrelu(y, lp, lp + 32, lp + 64);
    lp += 96;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float *dst = planes + tw_chan(ct * 16 + cb + r);
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[ooff[e]] += y[ct][r][e];
      }
    dma_landed();
    tw_barrier();
  }

  // ---- tail -------------------------------------------------------------------------------------------
  if (a.tail_mode == 0) {
    set_dilation(1);
    tower_layer<8>(planes, U, woff, lane, y);
    float *out = a.out + (size_t)n * 32 * TW_P;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = ct * 16 + cb + r;
        const float b = lp[c];
        // (dilation 1: e = 0, 1 and e = 2, 3 are neighbouring pixels of two neighbouring rows)
        float2 lo, hi;
        lo.x = y[ct][r][0] + b, lo.y = y[ct][r][1] + b, hi.x = y[ct][r][2] + b, hi.y = y[ct][r][3] + b;
        *reinterpret_cast<float2 *>(out + (size_t)c * TW_P + ooff[0]) = lo;
        *reinterpret_cast<float2 *>(out + (size_t)c * TW_P + ooff[2]) = hi;
      }
  } else {
    // 32 -> 1, 3x3, one pixel per thread, then relu(prior*fx + delta) / fx  (multi_view_stereonet.py:482-484, 607-611)
    const int py = tid >> 5, px = tid & 31;
    float acc = lp[32 * 9];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int r = py + dy, c = px + dx;
        const int o = (r >= 0 && r < TW_ROWS && c >= 0 && c < TW_COLS) ? r * TW_COLS + c : TW_ZERO;
        const int tap = (dy + 1) * 3 + dx + 1;
        float t = 0.f;
#pragma unroll 8
        for (int ch = 0; ch < 32; ++ch) t += lp[ch * 9 + tap] * planes[tw_chan(ch) + o];
        acc += t;
      }
    const float fx = a.fx[n % a.fx_mod];
    const float pr = a.prior[(size_t)n * TW_P + tid];
    a.out[(size_t)n * TW_P + tid] = relu_nan(pr * fx + acc) / fx;
  }
}

}  // namespace mvsn

extern "C" int mvsn_tower_16x32(const mvsn_tower_desc *d, int n_samples, mvsn_stream_t stream) {
  using namespace mvsn;
  MVSN_REQUIRE(d && d->in[0] && d->weights && d->params && d->out, MVSN_E_BADARG, "mvsn_tower_16x32: null pointer");
  MVSN_REQUIRE(n_samples > 0 && d->n_blocks >= 0 && d->n_blocks <= TW_MAX_BLOCKS, MVSN_E_BADARG,
               "mvsn_tower_16x32: bad sizes");
  MVSN_REQUIRE(d->head_chunks == 0 || d->head_chunks == 9, MVSN_E_BADARG,
               "mvsn_tower_16x32: the head layer takes 33..36 input channels (9 k-steps)");
  int cin = 0;
  for (int b = 0; b < 3; ++b) {
    MVSN_REQUIRE(d->channels[b] >= 0 && (d->channels[b] == 0 || (d->in[b] && d->sample_mod[b] > 0)), MVSN_E_BADARG,
                 "mvsn_tower_16x32: bad input block %d", b);
    cin += d->channels[b];
  }
  MVSN_REQUIRE(d->head_chunks ? cin <= d->head_chunks * 4 && cin > d->head_chunks * 4 - 4 : cin == 32, MVSN_E_BADARG,
               "mvsn_tower_16x32: %d input channels do not match the first layer", cin);
  MVSN_REQUIRE(d->tail_mode == 0 || (d->tail_mode == 1 && d->prior && d->fx && d->fx_mod > 0), MVSN_E_BADARG,
               "mvsn_tower_16x32: bad tail");
  TowerArgs a;
  for (int b = 0; b < 3; ++b) a.in[b] = d->in[b], a.c[b] = d->channels[b], a.mod[b] = d->sample_mod[b] > 0 ? d->sample_mod[b] : 1;
  a.scale = d->block_scale;
  a.scale_mod = d->scale_mod > 0 ? d->scale_mod : 1;
  a.scale_block = d->block_scale ? d->scale_block : -1;
  a.head_chunks = d->head_chunks;
  a.n_blocks = d->n_blocks;
  for (int i = 0; i < TW_MAX_BLOCKS; ++i) {
    a.dil[i] = i < d->n_blocks ? d->dilation[i] : 1;
    MVSN_REQUIRE(a.dil[i] == 1 || a.dil[i] == 2 || a.dil[i] == 4 || a.dil[i] == 8, MVSN_E_BADARG,
                 "mvsn_tower_16x32: dilation must be 1, 2, 4 or 8");
  }
  a.U = d->weights;
  a.params = d->params;
  a.tail_mode = d->tail_mode;
  a.prior = d->prior;
  a.fx = d->fx;
  a.fx_mod = d->fx_mod > 0 ? d->fx_mod : 1;
  a.out = d->out;
  const size_t lds = (size_t)TW_LDS_FLOATS * sizeof(float);
  static LdsOptIn opt;
  if (int rc = ensure_lds(opt, (const void *)tower_kernel, lds, "mvsn_tower_16x32")) return rc;
  hipLaunchKernelGGL(tower_kernel, dim3(n_samples), dim3(TW_THREADS), lds, (hipStream_t)stream, a, (const void *)a.in[0],
                     (const void *)a.in[1], (const void *)a.in[2], (const void *)a.scale, (const void *)a.U,
                     (const void *)a.params, (const void *)a.prior, (const void *)a.fx, (const void *)a.out,
                     (const void *)nullptr);
  return check_launch("mvsn_tower_16x32");
}
