// 32 -> 32 channel 3x3 / 3x3x3 convolution on the bf16 matrix cores with fp32-equivalent accuracy
// ("3 x bf16 split"): every fp32 operand is split into hi = bf16(v) and
// lo = bf16(v - hi) (together 16 mantissa bits) and each product is taken as
//       a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (a_lo*b_lo ~ 2^-16 |a b| is dropped)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: three instructions of 16 (SIMD) cycles cover
// K = 32 input channels, against eight 32-cycle fp32 instructions -- the fp32 MFMA rate is what bounds
// the regulariser and refiner layers.
//
// Plane streaming: a workgroup owns TZO x TY x 32 outputs x 32 couts and walks the TZO+KD-1 input
// planes it needs once; the haloed plane tile sits in LDS channel-minor, one 144-byte record per
// position = [32 x bf16 hi | 32 x bf16 lo | 16 pad] (record stride 36 dwords: the sixteen 16-byte
// B-fragment reads of a lane group land on distinct banks).  Lane l of an MFMA supplies
// A[i = l&15][k = 8*(l>>4) + 0..7] (weights, read as two 16-byte global loads per part, L1/L2 resident)
// and B[k][j = l&15] (one 16-byte LDS read per part).  Input planes that fall outside the volume are
// skipped (their taps contribute zeros), so edge tiles do less work.
#include "mvsn_common.h"
#include "mvsn_conv_bf16x3.h"

namespace mvsn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

constexpr int BX_THREADS = 256;
constexpr int BX_REC = 36;  // dwords per position record

__device__ __forceinline__ unsigned int pack_bf16_pair(float a, float b) {
  // two RNE conversions in one v_cvt_pk_bf16_f32
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 p;
  p[0] = (__bf16)a;
  p[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned int, p);
}
__device__ __forceinline__ float bf16_lo_to_float(unsigned int pair) { return __uint_as_float(pair << 16); }
__device__ __forceinline__ float bf16_hi_to_float(unsigned int pair) { return __uint_as_float(pair & 0xffff0000u); }

// weights: [tap][cout tile t][part: hi, lo][lane][8 x bf16]
__global__ void conv_bf16x3_pack_kernel(const float *__restrict__ w, int ntaps, unsigned short *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one (tap, t, lane, e)
  const int total = ntaps * 2 * 64 * 8;
  if (i >= total) return;
  const int e = i & 7, lane = (i >> 3) & 63, t = (i >> 9) & 1, tap = i >> 10;
  const int co = t * 16 + (lane & 15), ci = 8 * (lane >> 4) + e;
  const float v = w[((size_t)co * 32 + ci) * ntaps + tap];
  const unsigned int hi = pack_bf16_pair(v, 0.0f) & 0xffffu;
  const float r = v - __uint_as_float(hi << 16);
  const unsigned int lo = pack_bf16_pair(r, 0.0f) & 0xffffu;
  const size_t base = ((size_t)(tap * 2 + t) * 2) * 512 + lane * 8 + e;
  out[base] = (unsigned short)hi;
  out[base + 512] = (unsigned short)lo;
}

template <int KD, int TZO, int TY, int MODE>
__global__ __launch_bounds__(BX_THREADS, 2) void conv_bf16x3_kernel(Bf16x3Geom g, const float *__restrict__ in,
                                                                    const uintx4 *__restrict__ wpk,
                                                                    const float *__restrict__ bias,
                                                                    const float *__restrict__ in_stats,
                                                                    const float *__restrict__ in_gamma,
                                                                    const float *__restrict__ in_beta,
                                                                    float *__restrict__ out,
                                                                    float *__restrict__ out_partials) {
  extern __shared__ __attribute__((aligned(16))) unsigned int plane[];  // HY*HX records of BX_REC dwords
  constexpr int NPT = TY / 2;  // pixel tiles (16 columns) per wave per output plane
  const int npos = g.HY * g.HX;
  float *scsh = reinterpret_cast<float *>(plane + (size_t)npos * BX_REC);  // 64
  float *red = scsh + 64;                                                  // 16

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y;
  int tix = blockIdx.x;
  const int txi = tix % g.ntx;
  tix /= g.ntx;
  const int tyi = tix % g.nty;
  const int tzi = tix / g.nty;
  const int z0 = tzi * TZO, y0 = tyi * TY, x0 = txi * 32;
  const int gy0 = y0 - g.dil, gx0 = x0 - g.dil;
  const size_t in_plane = (size_t)g.H * g.W, in_chan = (size_t)g.D * in_plane;
  const float *inn = in + (size_t)n * 32 * in_chan;

  if (MODE == 1 && tid < 32) {
    const int grp = tid >> 3;
    const float mean = in_stats[((size_t)n * 4 + grp) * 2 + 0];
    const float rstd = in_stats[((size_t)n * 4 + grp) * 2 + 1];
    const float sc = rstd * in_gamma[tid];
    scsh[tid] = sc;
    scsh[32 + tid] = in_beta[tid] - mean * sc;
  }

  // this lane's pixels (column j = lane & 15 of each of the wave's pixel tiles) as record offsets
  int prec[NPT];
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int pt = wave * NPT + j;  // 0 .. TY*2-1
    const int yy = pt >> 1, xx = (pt & 1) * 16 + (lane & 15);
    prec[j] = (yy * g.HX + xx) * BX_REC + (lane >> 4) * 4;  // + k-group: 8 bf16 = 4 dwords
  }

  floatx4 acc[TZO][NPT][2];
#pragma unroll
  for (int zo = 0; zo < TZO; ++zo)
#pragma unroll
    for (int j = 0; j < NPT; ++j) acc[zo][j][0] = acc[zo][j][1] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int items = npos * 4;  // (position, channel group of 8)
  for (int zi = 0; zi < TZO + KD - 1; ++zi) {
    const int gz = z0 - (KD / 2) + zi;
    if (gz < 0 || gz >= g.D) continue;  // uniform: a plane of zeros contributes nothing
    __syncthreads();                    // previous plane fully consumed (and scsh visible)
    // ---- stage the haloed plane tile: fp32 HBM -> (transform) -> hi/lo bf16 records ----------------
    for (int it = tid; it < items; it += BX_THREADS) {
      const int grp = it / npos, p = it - grp * npos;
      const int y = p / g.HX, x = p - y * g.HX;
      const int gy = gy0 + y, gx = gx0 + x;
      const bool ok = gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
      const float *src = inn + (size_t)(grp * 8) * in_chan + (size_t)gz * in_plane + (size_t)(ok ? gy : 0) * g.W +
                         (ok ? gx : 0);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ok ? src[(size_t)e * in_chan] : 0.0f;
      if (MODE == 1 && ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = lrelu02(v[e] * scsh[grp * 8 + e] + scsh[32 + grp * 8 + e]);
      }
      uintx4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned int h = pack_bf16_pair(v[2 * e], v[2 * e + 1]);
        hi[e] = h;
        lo[e] = pack_bf16_pair(v[2 * e] - bf16_lo_to_float(h), v[2 * e + 1] - bf16_hi_to_float(h));
      }
      unsigned int *rec = plane + (size_t)p * BX_REC + grp * 4;
      *reinterpret_cast<uintx4 *>(rec) = hi;
      *reinterpret_cast<uintx4 *>(rec + 16) = lo;
    }
    __syncthreads();
    // ---- every output plane that sees this input plane through some z-tap ----------------------------
#pragma unroll
    for (int zo = 0; zo < TZO; ++zo) {
      const int tz = zi - zo;              // z-tap through which output plane zo sees this input plane
      if (tz < 0 || tz >= KD) continue;    // uniform
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const int tap = (tz * 3 + ty) * 3 + tx;
          const uintx4 *wt = wpk + (size_t)tap * 256 + lane;  // [t][part][lane]
          const bf16x8 a0h = __builtin_bit_cast(bf16x8, wt[0]);
          const bf16x8 a0l = __builtin_bit_cast(bf16x8, wt[64]);
          const bf16x8 a1h = __builtin_bit_cast(bf16x8, wt[128]);
          const bf16x8 a1l = __builtin_bit_cast(bf16x8, wt[192]);
          const int toff = (ty * g.dil * g.HX + tx * g.dil) * BX_REC;
#pragma unroll
          for (int j = 0; j < NPT; ++j) {
            const unsigned int *rec = plane + prec[j] + toff;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uintx4 *>(rec));
            const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uintx4 *>(rec + 16));
            acc[zo][j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, bh, acc[zo][j][0], 0, 0, 0);
            acc[zo][j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bl, acc[zo][j][0], 0, 0, 0);
            acc[zo][j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bh, acc[zo][j][0], 0, 0, 0);
            acc[zo][j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, bh, acc[zo][j][1], 0, 0, 0);
            acc[zo][j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bl, acc[zo][j][1], 0, 0, 0);
            acc[zo][j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bh, acc[zo][j][1], 0, 0, 0);
          }
        }
    }
  }

  // ---- epilogue: bias, store, GroupNorm partials (same contract as the fp32 kernels) ------------------
  const int cbase = (lane >> 4) * 4;
  const size_t out_plane = (size_t)g.H * g.W, out_chan = (size_t)g.D * out_plane;
  float *outn = out + (size_t)n * 32 * out_chan;
  float s[2] = {0.f, 0.f};
  int cnt = 0;
#pragma unroll
  for (int zo = 0; zo < TZO; ++zo)
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int pt = wave * NPT + j;
      const int oz = z0 + zo, oy = y0 + (pt >> 1), ox = x0 + (pt & 1) * 16 + (lane & 15);
      const bool ok = oz < g.D && oy < g.H && ox < g.W;
      if (ok) cnt += 1;
      const size_t pos = (size_t)oz * out_plane + (size_t)oy * g.W + ox;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = t * 16 + cbase + r;
          const float v = acc[zo][j][t][r] + (bias ? bias[c] : 0.0f);
          acc[zo][j][t][r] = ok ? v : 0.0f;
          if (ok) {
            outn[(size_t)c * out_chan + pos] = v;
            s[t] += v;
          }
        }
    }
  if (out_partials == nullptr) return;
  auto half_wave_sum = [&](float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    return v;
  };
  const int hi = lane >> 5;
  int cnt_tile;
  {
    float c = (lane < 16) ? (float)cnt : 0.0f;
    c = half_wave_sum(c);
    __syncthreads();
    if (lane == 0) red[wave] = c;
    __syncthreads();
    cnt_tile = (int)(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
  }
  const float npos_out = (float)cnt_tile * 8.0f;
  float m[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) s[t] = half_wave_sum(s[t]);
  if ((lane & 31) == 0) {
    red[wave * 4 + 0 + hi] = s[0];
    red[wave * 4 + 2 + hi] = s[1];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float tot = 0.f;
    for (int w = 0; w < 4; ++w) tot += red[w * 4 + t * 2 + hi];
    m[t] = cnt_tile > 0 ? tot / npos_out : 0.0f;
  }
  __syncthreads();
  float q[2] = {0.f, 0.f};
#pragma unroll
  for (int zo = 0; zo < TZO; ++zo)
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int pt = wave * NPT + j;
      const bool ok = z0 + zo < g.D && y0 + (pt >> 1) < g.H && x0 + (pt & 1) * 16 + (lane & 15) < g.W;
      if (ok) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dv = acc[zo][j][t][r] - m[t];
            q[t] += dv * dv;
          }
      }
    }
#pragma unroll
  for (int t = 0; t < 2; ++t) q[t] = half_wave_sum(q[t]);
  if ((lane & 31) == 0) {
    red[wave * 4 + 0 + hi] = q[0];
    red[wave * 4 + 2 + hi] = q[1];
  }
  __syncthreads();
  if (wave == 0 && (lane & 31) == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float tot = 0.f;
      for (int w = 0; w < 4; ++w) tot += red[w * 4 + t * 2 + hi];
      float *p = out_partials + (((size_t)n * g.tiles + blockIdx.x) * 4 + (t * 2 + hi)) * 3;
      p[0] = npos_out;
      p[1] = m[t];
      p[2] = tot;
    }
  }
}

// ---- device self-test of the bf16 fragment mapping -----------------------------------------------------
__global__ void mfma_bf16_selftest_kernel(int *bad) {
  const int lane = threadIdx.x;
  bf16x8 a, b;
  // A[i][k] = (i - 2k) / 8, B[k][j] = (3k + j - 7) / 4: exactly representable in bf16, asymmetric
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * (lane >> 4) + e;
    a[e] = (__bf16)((float)((lane & 15) - 2 * k) * 0.125f);
    b[e] = (__bf16)((float)(3 * k + (lane & 15) - 7) * 0.25f);
  }
  floatx4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  int wrong = 0;
  for (int r = 0; r < 4; ++r) {
    const int i = (lane >> 4) * 4 + r, j = lane & 15;
    float ref = 0.f;
    for (int k = 0; k < 32; ++k) ref += ((float)(i - 2 * k) * 0.125f) * ((float)(3 * k + j - 7) * 0.25f);
    if (fabsf(ref - c[r]) > 1e-3f * fabsf(ref) + 1e-3f) wrong++;
  }
  if (wrong) atomicAdd(bad, wrong);
}

bool bf16x3_geom(const mvsn_conv_desc *d, Bf16x3Geom *g) {
  if (!d || d->c_in != 32 || d->c_out != 32 || d->stride != 1 || d->kh != 3 || d->kw != 3) return false;
  if (!(d->kd == 1 || d->kd == 3) || (d->kd == 1 && d->depth != 1) || d->dilation < 1) return false;
  g->n = d->n, g->D = d->depth, g->H = d->rows, g->W = d->cols, g->dil = d->dilation, g->kd = d->kd;
  g->tzo = d->kd == 3 ? 4 : 1;
  g->ty = 8;
  g->HY = g->ty + 2 * g->dil;
  g->HX = 32 + 2 * g->dil;
  g->ntz = (g->D + g->tzo - 1) / g->tzo;
  g->nty = (g->H + g->ty - 1) / g->ty;
  g->ntx = (g->W + 31) / 32;
  g->tiles = g->ntz * g->nty * g->ntx;
  g->lds_bytes = ((size_t)g->HY * g->HX * BX_REC + 64 + 16) * 4;
  return g->lds_bytes <= 160 * 1024 && (d->kd == 1 || d->dilation == 1);
}

int bf16x3_pack(const mvsn_conv_desc *d, const float *weight, void *packed, hipStream_t stream) {
  const int ntaps = d->kd * 9;
  const int total = ntaps * 1024;
  hipLaunchKernelGGL(conv_bf16x3_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, weight, ntaps,
                     (unsigned short *)packed);
  return check_launch("mvsn_conv_pack_weights(bf16x3)");
}

int bf16x3_launch(const Bf16x3Geom &g, const float *in, const void *wpk, const float *bias, const float *in_stats,
                  const float *in_gamma, const float *in_beta, float *out, float *out_partials, hipStream_t stream) {
  dim3 grid(g.tiles, g.n);
#define MVSN_BX_LAUNCH(...)                                                                                       \
  do {                                                                                                            \
    auto kern = conv_bf16x3_kernel<__VA_ARGS__>;                                                                  \
    static size_t opted = 0;                                                                                      \
    if (g.lds_bytes > opted) {                                                                                    \
      hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                         (int)g.lds_bytes);                                                       \
      if (e != hipSuccess) {                                                                                      \
        set_error("mvsn_conv_forward(bf16x3): LDS opt-in of %zu bytes failed: %s", g.lds_bytes,                   \
                  hipGetErrorString(e));                                                                          \
        return (int)e;                                                                                            \
      }                                                                                                           \
      opted = g.lds_bytes;                                                                                        \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, grid, dim3(BX_THREADS), g.lds_bytes, stream, g, in, (const uintx4 *)wpk, bias, in_stats, \
                       in_gamma, in_beta, out, out_partials);                                                     \
  } while (0)
  const bool xf = in_stats != nullptr;
  if (g.kd == 3) {
    if (xf) MVSN_BX_LAUNCH(3, 4, 8, 1); else MVSN_BX_LAUNCH(3, 4, 8, 0);
  } else {
    if (xf) MVSN_BX_LAUNCH(1, 1, 8, 1); else MVSN_BX_LAUNCH(1, 1, 8, 0);
  }
#undef MVSN_BX_LAUNCH
  return check_launch("mvsn_conv_forward(bf16x3)");
}

int bf16_selftest(hipStream_t stream, int *dbad) {
  hipLaunchKernelGGL(mfma_bf16_selftest_kernel, dim3(1), dim3(64), 0, stream, dbad);
  return check_launch("mvsn_selftest_mfma(bf16)");
}

}  // namespace mvsn
