// Direct 2-D / 3-D convolution on fp32 MFMA for the feature extractor, the cost-volume
// regulariser and the idepth refiners (see include/mvsn_hip.h: mvsn_conv_forward).
//
// Implicit GEMM with A = weights (16 couts x 4 cins per v_mfma_f32_16x16x4_f32) and
// B = activations (4 cins x 16 consecutive output columns).  A 256-thread workgroup owns an
// output tile of TZ x TY x 32 positions and all (<= 32) output channels; the input is walked in
// chunks of 4 channels: the haloed chunk tile and that chunk's weight fragments are staged in
// LDS, then every tap is a plain offset read.  The previous layer's GroupNorm + LeakyReLU can be
// applied while the tile is staged, and the epilogue emits per-tile GroupNorm partials
// (count, mean, M2) so normalisation never needs its own pass over the volume.
//
// LDS tile: [4 ch][HZ][HY][XS], channel stride CST = 16 (mod 32) -> the 16-column x 4-channel
// fragment read is bank-conflict-free at stride 1.
#include "mvsn_common.h"
#include "mvsn_conv_bf16x3.h"
#include "mvsn_conv_wino.h"

namespace mvsn {

constexpr int CV_THREADS = 256;
constexpr int CV_WAVES = 4;
constexpr int CV_TX = 32;   // output columns per tile
constexpr int CV_CK = 4;    // input channels per staged chunk = one MFMA k-step
constexpr float CV_EPS = 1e-5f;

// Division by a launch-invariant divisor: q = (mulhi(n, mul) + n) >> shift, exact for 0 <= n < 2^31.
// The kernels' prologues decompose tile and staging indices with ~30 divisions; as generic integer
// divisions (~40 instructions each, issued next to three MFMA-bound waves per SIMD) they made the
// prologue a quarter of a workgroup's lifetime.
#ifndef MVSN_TALL_TILE_MAX_DIL
#define MVSN_TALL_TILE_MAX_DIL 8   // 2-D 3x3 layers up to this dilation use 16-row tiles (measured: 2/4/8 within 1.5 %)
#endif
#ifndef MVSN_DMA_MAX_DIL
#define MVSN_DMA_MAX_DIL 8         // 2-D 3x3 layers up to this dilation run on the LDS-DMA kernel (cols % 4 == 0)
#endif

struct FastDiv {
  unsigned mul, shift;
};
static FastDiv make_fastdiv(unsigned d) {
  unsigned s = 0;
  while ((1u << s) < d) ++s;
  const unsigned long long m = ((1ull << 32) * ((1ull << s) - d)) / d + 1;
  return FastDiv{(unsigned)m, s};
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) {
  return (int)((__umulhi((unsigned)n, f.mul) + (unsigned)n) >> f.shift);
}

struct ConvGeom {
  FastDiv fd_hx, fd_hy, fd_ntx, fd_nty, fd_ty, fd_gq;
  int n, cin, cout, D, H, W, Do, Ho, Wo;
  int kd, kh, kw, stride, dil, pd, ph, pw;
  int TZ, TY;           // output tile (TX = 32)
  int HZ, HY, HX, XS;   // staged extent and padded row stride
  int CST;              // channel stride in LDS (floats)
  int ipc;              // 64-element DMA runs per staged channel
  int v4;               // register-staged kernel stages 16-byte groups from the aligned column x0 - 4 (3-D layers)
  int xoff;             // ... column of tap (.,.,0) of output column 0 inside a staged row
  int ntz, nty, ntx, tiles;
  int ntaps, nchunks;
  int wfloats_chunk;    // ntaps * 2 cout-tiles * 64
  size_t lds_bytes;
  int se;               // staged elements per thread per channel
  // LDS-DMA kernel: rows are staged as whole 16-byte pieces from the aligned column x0 - dpa (dpa = halo
  // rounded up to 4: 40 floats per row for a halo <= 4, 48 for 8), so one DMA instruction moves 1 KB
  int dma_ok;           // cols % 4 == 0, stride 1, halo <= 8
  int dpa, dq;          // aligned halo, 16-byte pieces per row
  int dXS, dCST, dipc;  // row stride (4 * dq), channel stride, piece instructions per channel
  int dma_stage_floats; // floats per pipeline stage without the residual tile
  FastDiv fd_dq;
};

static bool make_geom(const mvsn_conv_desc *d, ConvGeom *g) {
  if (!d || d->n <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->c_out > 32 || d->depth <= 0 || d->rows <= 0 ||
      d->cols <= 0)
    return false;
  if (d->kd <= 0 || d->kh <= 0 || d->kw <= 0 || !(d->kd & 1) || !(d->kh & 1) || !(d->kw & 1)) return false;
  if (d->stride != 1 && d->stride != 2) return false;
  if (d->dilation < 1) return false;
  if (d->depth > 1 && d->stride != 1) return false;
  g->n = d->n, g->cin = d->c_in, g->cout = d->c_out, g->D = d->depth, g->H = d->rows, g->W = d->cols;
  g->kd = d->kd, g->kh = d->kh, g->kw = d->kw, g->stride = d->stride, g->dil = d->dilation;
  g->pd = d->kd / 2, g->ph = d->dilation * (d->kh / 2), g->pw = d->dilation * (d->kw / 2);
  g->Do = d->depth;
  g->Ho = (d->rows - 1) / d->stride + 1;
  g->Wo = (d->cols - 1) / d->stride + 1;
  const bool is3d = d->depth > 1 || d->kd > 1;
  g->TZ = is3d ? 2 : 1;
  // 2-D 3x3 layers on tall images use 16-row tiles (less halo per output, more MFMAs per staging).  On the
  // LDS-DMA kernel (cols % 4 == 0) that holds for every dilation (measured on MI355X, B=128: 8/16 rows within
  // 1.5 % at dilation 4/8); the register-staged kernel keeps 8-row tiles above dilation 2, where the 16-row
  // halo no longer fits two workgroups' worth of staging registers (10-17 % slower).
  {
    const bool dma = d->cols % 4 == 0 && d->dilation <= MVSN_DMA_MAX_DIL;
    const int tall_dil = dma ? MVSN_TALL_TILE_MAX_DIL : 2;
    g->TY = (!is3d && d->kh == 3 && d->stride == 1 && d->dilation <= tall_dil && (d->rows - 1) / d->stride + 1 > 8) ? 16 : 8;
  }
  g->HZ = g->TZ + d->kd - 1;
  g->HY = (g->TY - 1) * d->stride + d->dilation * (d->kh - 1) + 1;
  g->HX = (CV_TX - 1) * d->stride + d->dilation * (d->kw - 1) + 1;
  g->XS = g->HX;
  // the register-staged layers with aligned rows (3-D 3x3x3; 2-D 5x5 stride 2): whole 16-byte groups starting at
  // the aligned column x0 * stride - 4 -- one load / LDS write moves 4 elements
  g->v4 = (d->cols % 4 == 0 && d->dilation == 1 && g->pw <= 4 &&
           ((is3d && d->kd == 3 && d->stride == 1) || (!is3d && d->kh == 5 && d->stride == 2))) ? 1 : 0;
  g->xoff = 0;
  if (g->v4) {
    g->xoff = 4 - g->pw;
    g->XS = (g->HX + g->xoff + 3) / 4 * 4;   // 40 floats (stride 1, 3 taps), 72 (stride 2, 5 taps)
  }
  g->fd_gq = make_fastdiv((unsigned)(g->XS / 4));
  // channel stride: whole 64-element DMA runs (the last run may overshoot the tile and lands in the
  // slot's tail) + 16 so that CST = 16 (mod 32)
  g->ipc = (g->HZ * g->HY * g->XS + 63) / 64;
  g->CST = g->ipc * 64 + 16;
  g->ntz = (g->Do + g->TZ - 1) / g->TZ;
  g->nty = (g->Ho + g->TY - 1) / g->TY;
  g->ntx = (g->Wo + CV_TX - 1) / CV_TX;
  g->tiles = g->ntz * g->nty * g->ntx;
  g->fd_hx = make_fastdiv((unsigned)g->HX), g->fd_hy = make_fastdiv((unsigned)g->HY);
  g->fd_ntx = make_fastdiv((unsigned)g->ntx), g->fd_nty = make_fastdiv((unsigned)g->nty);
  g->fd_ty = make_fastdiv((unsigned)g->TY);
  g->ntaps = d->kd * d->kh * d->kw;
  g->nchunks = (d->c_in + CV_CK - 1) / CV_CK;
  g->wfloats_chunk = g->ntaps * 2 * 64;
  g->lds_bytes = ((size_t)CV_CK * g->CST + g->wfloats_chunk + 64 /*in scale/shift*/ + 64 /*red*/) * sizeof(float);
  g->se = (g->HZ * g->HY * g->HX + CV_THREADS - 1) / CV_THREADS;
  {
    const int wslot = ((g->ntaps * 128 + 255) / 256) * 256;
    g->dma_ok = (d->cols % 4 == 0 && d->stride == 1 && g->pw <= 8 && d->kw == 3) ? 1 : 0;
    g->dpa = (g->pw + 3) / 4 * 4;
    g->dq = (CV_TX + 2 * g->dpa) / 4;
    g->dXS = 4 * g->dq;
    g->dipc = (g->HZ * g->HY * g->dq + 63) / 64;
    g->dCST = g->dipc * 256 + 16;                     // = 16 (mod 32): conflict-free 16-column x 4-channel reads
    g->dma_stage_floats = CV_CK * g->dCST + wslot;    // + CV_CK * dCST more when a residual tile is staged
    g->fd_dq = make_fastdiv((unsigned)g->dq);
  }
  if (g->se > 6) return false;
  const bool k333 = d->kd == 3 && d->kh == 3 && d->kw == 3 && d->stride == 1;
  const bool k133 = d->kd == 1 && d->kh == 3 && d->kw == 3 && d->stride == 1;
  const bool k155 = d->kd == 1 && d->kh == 5 && d->kw == 5 && d->stride == 2 && d->dilation == 1;
  if (!(k333 || k133 || k155)) return false;
  return g->lds_bytes <= 160 * 1024;
}

// packed weights: [chunk of 4 cin][tap][cout-tile 2][lane 64]; lane = k*16 + i holds
// W[cout = t*16 + i][cin = chunk*4 + k][tap], zero outside (c_in, c_out).
__global__ void conv_pack_kernel(const float *__restrict__ w, int cin, int cout, int ntaps, int nchunks,
                                 float *__restrict__ out) {
  const int total = nchunks * ntaps * 128;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = i & 63, t = (i >> 6) & 1;
  const int tap = (i >> 7) % ntaps, chunk = (i >> 7) / ntaps;
  const int co = t * 16 + (lane & 15);
  const int ci = chunk * CV_CK + (lane >> 4);
  out[i] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * ntaps + tap] : 0.0f;
}

// Shared epilogue of the fp32 kernels.  The MFMAs run with A = activations (16 pixels x 4 cins) and
// B = weights (4 cins x 16 couts), so D = pixels x couts: lane l holds, for cout t*16 + (l & 15), the four
// CONSECUTIVE pixels 4*(l>>4) + r of each of its NPT pixel tiles -- one 16-byte store per accumulator, no
// transpose.  The output offset of the first of those four pixels and how many of them are inside the
// image (0..4) are formed here from the tile origin (wave-uniform row arithmetic + one lane term), so no
// per-tile position registers stay live across the MFMA loop.  Per-wave GroupNorm partials (count, mean, M2) are reduced
// with shuffles only -- no LDS, no barrier: the group of a lane's channel is 2t + ((l>>3)&1).
template <int NPT, int CT>
__device__ __forceinline__ void conv_epilogue(const ConvGeom &g, floatx4 (&acc)[NPT][CT], int z0, int y0, int x0,
                                              int tid, int n, int tile_id, const float *__restrict__ bias,
                                              float *__restrict__ out, float *__restrict__ out_partials) {
  const int lane = tid & 63;
  const int cl = lane & 15;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  int opos[NPT], ovalid[NPT];
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int pt = wave_u * NPT + j;   // NPT is even: pixel tile j covers columns (j & 1) * 16 .. + 15
    const int zz = fdiv(pt >> 1, g.fd_ty), yy = (pt >> 1) - zz * g.TY;
    const int oz = z0 + zz, oy = y0 + yy, ox4 = x0 + (j & 1) * 16 + (lane >> 4) * 4;
    const bool ok = oz < g.Do && oy < g.Ho && ox4 < g.Wo;
    opos[j] = (oz * g.Ho + oy) * g.Wo + ox4;
    ovalid[j] = ok ? min(4, g.Wo - ox4) : 0;
  }
  const size_t out_chan = (size_t)g.Do * g.Ho * g.Wo;
  float *outn = out + (size_t)n * g.cout * out_chan;
  float s[2] = {0.f, 0.f};
  int cnt = 0;
  float bv[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) bv[t] = (bias && t * 16 + cl < g.cout) ? bias[t * 16 + cl] : 0.0f;
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    cnt += ovalid[j];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[j][t][r] + bv[t];
        acc[j][t][r] = v;
        if (r < ovalid[j]) s[t] += v;
      }
  }
  if ((g.Wo & 3) == 0) {   // uniform: the four pixels are all inside or all outside, and 16-byte aligned
#pragma unroll
    for (int j = 0; j < NPT; ++j)
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int c = t * 16 + cl;
        if (ovalid[j] == 4 && c < g.cout) *reinterpret_cast<floatx4 *>(outn + (size_t)c * out_chan + opos[j]) = acc[j][t];
      }
  } else {
#pragma unroll
    for (int j = 0; j < NPT; ++j)
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = t * 16 + cl;
          if (r < ovalid[j] && c < g.cout) outn[(size_t)c * out_chan + opos[j] + r] = acc[j][t][r];
        }
  }
  if (out_partials == nullptr || CT != 2) return;  // uniform; partials need all 32 channels (host-checked)
  // One record per (wave, 16-lane row): the 8 lanes of a half row hold the 8 couts of one GroupNorm group for the
  // same pixels (identical validity), so the reduction is three DPP adds -- no LDS shuffles across rows; the
  // records carry (count, mean, M2) and mvsn_groupnorm_finalize combines them.
  auto sum8 = [](float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1, 0, 3, 2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2, 3, 0, 1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    return v;
  };
  const int hi = (lane >> 3) & 1;
  const float npos = 8.0f * (float)cnt;
  const float rn = cnt > 0 ? 1.0f / npos : 0.0f;
  float m[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    s[t] = sum8(s[t]);
    m[t] = s[t] * rn;
  }
  float q[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NPT; ++j)
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < ovalid[j]) {
          const float dv = acc[j][t][r] - m[t];
          q[t] += dv * dv;
        }
#pragma unroll
  for (int t = 0; t < 2; ++t) q[t] = sum8(q[t]);
  if ((lane & 7) == 0) {   // lanes 0 and 8 of every row
    float *rec = out_partials + ((((size_t)n * g.tiles + tile_id) * 4 + (tid >> 6)) * 4 + (lane >> 4)) * 12;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      rec[(t * 2 + hi) * 3 + 0] = npos;
      rec[(t * 2 + hi) * 3 + 1] = m[t];
      rec[(t * 2 + hi) * 3 + 2] = q[t];
    }
  }
}

#ifdef MVSN_DMA_STAMPS   // tuning aid (tools/dma_phases.py): s_memtime stamps of one mid-launch wave
__device__ unsigned long long *g_dma_stamps = nullptr;
#define DMA_STAMP() do { if (dbg && dbg_i < 60) dbg[dbg_i++] = __builtin_readcyclecounter(); } while (0)
#else
#define DMA_STAMP() do { } while (0)
#endif

#ifdef MVSN_DMA_STAMPS   // register-staged kernel: per-phase totals of one workgroup's thread 0, kept in LDS
#define MFMA_PHASE(k) do { if (stamped) { const unsigned long long now_ = __builtin_readcyclecounter(); ph_lds[k] += now_ - ph_last; ph_last = now_; } } while (0)
#else
#define MFMA_PHASE(k) do { } while (0)
#endif

// NPT   pixel tiles (16 output columns each) per wave = TZ*TY*2/4
// KD/KH/KW/STRIDE compile-time so the tap loops unroll completely and LDS reads run ahead of the MFMAs
// SE    staged input elements per thread per channel (upper bound, ceil(HZ*HY*HX/256))
// CT    cout tiles (1 when c_out <= 16: the second MFMA of every pair is dropped)
//
// Pipeline per 4-channel chunk (one MFMA k-step per tap): the chunk's haloed tile and weight fragments are fetched from
// HBM/L2 into registers while the MFMAs of the previous chunk run, then written to LDS between
// two barriers (register-staged double buffering, one LDS copy).
// RES   residual-capable variant: the staged input is [in_residual +] LeakyReLU(GN(in)) (a residual
//       block folded into the load);
//       with out_staged != null every workgroup also writes the staged values of its own output
//       positions, so the block's output tensor is produced as a by-product (stride 1 only).
// V4    (3-D layers, cols % 4 == 0) the tile is staged as 16-byte groups: a third of the loads and LDS writes
template <int NPT, int KD, int KH, int KW, int STRIDE, int SE, int CT, bool RES, bool V4 = false>
__global__ __launch_bounds__(CV_THREADS, 2) void conv_mfma_kernel(ConvGeom g, const float *__restrict__ in,
                                                                  const float *__restrict__ wpk,
                                                                  const float *__restrict__ bias,
                                                                  const float *__restrict__ in_stats,
                                                                  const float *__restrict__ in_gamma,
                                                                  const float *__restrict__ in_beta,
                                                                  const float *__restrict__ in_residual,
                                                                  float *__restrict__ out_staged,
                                                                  float *__restrict__ out,
                                                                  float *__restrict__ out_partials) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTAPS = KD * KH * KW;
  constexpr int WFL = NTAPS * 128;            // weight floats per chunk
  constexpr int WR = (WFL / 4 + CV_THREADS - 1) / CV_THREADS;  // float4 per thread
  float *tile = smem;                               // CV_CK * CST
  float *wl = tile + (size_t)CV_CK * g.CST;         // WFL
  float *scsh = wl + WFL;                           // 32 scale + 32 shift of the input transform

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y;
  const int tile_id = xcd_tile_index(blockIdx.x, g.tiles);
  const int trow = fdiv(tile_id, g.fd_ntx), txi = tile_id - trow * g.ntx;
  const int tzi = fdiv(trow, g.fd_nty), tyi = trow - tzi * g.nty;
  const int z0 = tzi * g.TZ, y0 = tyi * g.TY, x0 = txi * CV_TX;  // output-space origin
  const int gz0 = z0 - g.pd, gy0 = y0 * STRIDE - g.ph, gx0 = x0 * STRIDE - g.pw;  // input-space origin
  const size_t in_plane = (size_t)g.H * g.W, in_chan = (size_t)g.D * in_plane;
  const float *inn = in + (size_t)n * g.cin * in_chan;

  // this lane's output positions
  int lpos[NPT];      // fragment read: offset of (z, y, x = column lane&15) in the staged tile (tap (0,0,0))
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int pt = wave * NPT + j;
    const int xt = pt & 1, zz = fdiv(pt >> 1, g.fd_ty), yy = (pt >> 1) - zz * g.TY;
    const int xx = xt * 16 + (lane & 15);
    lpos[j] = (zz * g.HY + yy * STRIDE) * g.XS + xx * STRIDE + g.xoff + (lane >> 4) * g.CST;
  }

  // staging plan: element e = tid + k*256 of the HZ x HY x HX chunk tile (same for every channel); V4: 16-byte
  // group e of the HZ x HY x 10 groups (columns x0 - 4 + 4q .. + 3: all inside or all outside the image)
  int goff[SE];
  unsigned interior = 0;  // bit k: staged element k is one of this workgroup's own output positions
  const int tile_elems = V4 ? g.HZ * g.HY * (g.XS / 4) : g.HZ * g.HY * g.HX;
#pragma unroll
  for (int k = 0; k < SE; ++k) {
    const int e = tid + k * CV_THREADS;
    int off = -2;  // -2: beyond the tile, -1: zero padding
    if (e < tile_elems) {
      int row, x;
      if constexpr (V4) {
        row = fdiv(e, g.fd_gq);
        x = (e - row * (g.XS / 4)) * 4 - 4 + g.pw;   // so that gx0 + x = x0 * stride - 4 + 4q
      } else {
        row = fdiv(e, g.fd_hx);
        x = e - row * g.HX;
      }
      const int z = fdiv(row, g.fd_hy), y = row - z * g.HY;
      const int gz = gz0 + z, gy = gy0 + y, gx = gx0 + x;
      off = (gz >= 0 && gz < g.D && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) ? (gz * g.H + gy) * g.W + gx : -1;
      if (RES && off >= 0 && z >= g.pd && z < g.pd + g.TZ && y >= g.ph && y < g.ph + g.TY && x >= g.pw &&
          x < g.pw + CV_TX)
        interior |= 1u << k;
    }
    goff[k] = off;
  }
  const float *resn = (RES && in_residual) ? in_residual + (size_t)n * g.cin * in_chan : nullptr;
  float *stgn = (RES && out_staged) ? out_staged + (size_t)n * g.cin * in_chan : nullptr;

  const bool xform = in_stats != nullptr;
  if (xform && tid < 32) {
    const int grp = tid >> 3;
    const float mean = in_stats[((size_t)n * 4 + grp) * 2 + 0];
    const float rstd = in_stats[((size_t)n * 4 + grp) * 2 + 1];
    const float sc = rstd * in_gamma[tid];
    scsh[tid] = sc;
    scsh[32 + tid] = in_beta[tid] - mean * sc;
  }

  floatx4 acc[NPT][CT];
#pragma unroll
  for (int j = 0; j < NPT; ++j)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[j][t] = floatx4{0.f, 0.f, 0.f, 0.f};

  static_assert(!(V4 && RES), "the 16-byte staging path has no residual folding");
  float sreg[V4 ? 1 : CV_CK][V4 ? 1 : SE];
  floatx4 sreg4[V4 ? CV_CK : 1][V4 ? SE : 1];
  float rreg[RES ? CV_CK : 1][RES ? SE : 1];
  floatx4 wreg[WR];
  auto stage_load = [&](int chunk) {
    const int c0 = chunk * CV_CK;
    const int cc = min(CV_CK, g.cin - c0);
#pragma unroll
    for (int c = 0; c < CV_CK; ++c) {
      const float *src = inn + (size_t)(c0 + c) * in_chan;
      if constexpr (V4) {
#pragma unroll
        for (int k = 0; k < SE; ++k)
          sreg4[c][k] = (c < cc && goff[k] >= 0) ? *reinterpret_cast<const floatx4 *>(src + goff[k])
                                                 : floatx4{0.f, 0.f, 0.f, 0.f};
        continue;
      }
#pragma unroll
      for (int k = 0; k < SE; ++k) sreg[c][k] = (c < cc && goff[k] >= 0) ? src[goff[k]] : 0.0f;
      if constexpr (RES) {
        const float *rsrc = resn + (size_t)(c0 + c) * in_chan;
#pragma unroll
        for (int k = 0; k < SE; ++k) rreg[c][k] = (resn && c < cc && goff[k] >= 0) ? rsrc[goff[k]] : 0.0f;
      }
    }
    const float *wsrc = wpk + (size_t)chunk * WFL;
#pragma unroll
    for (int k = 0; k < WR; ++k) {
      const int idx = (tid + k * CV_THREADS) * 4;
      if (idx < WFL) wreg[k] = *reinterpret_cast<const floatx4 *>(wsrc + idx);
    }
  };
  auto stage_store = [&](int chunk) {
    const int c0 = chunk * CV_CK;
#pragma unroll
    for (int c = 0; c < CV_CK; ++c) {
      float sc = 1.0f, sh = 0.0f;
      if (xform) {
        sc = scsh[c0 + c];
        sh = scsh[32 + c0 + c];
      }
      if constexpr (V4) {
#pragma unroll
        for (int k = 0; k < SE; ++k)
          if (goff[k] != -2) {
            floatx4 v = sreg4[c][k];
            if (xform) {   // uniform; groups outside the image hold zeros and must stay zero: LReLU(0 * sc + 0) = 0
              const float shk = goff[k] >= 0 ? sh : 0.0f;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = lrelu02(v[r] * sc + shk);
            }
            *reinterpret_cast<floatx4 *>(tile + c * g.CST + (tid + k * CV_THREADS) * 4) = v;
          }
        continue;
      }
#pragma unroll
      for (int k = 0; k < SE; ++k) {
        if (goff[k] != -2) {
          float v = sreg[c][k];
          if (xform && goff[k] >= 0) v = lrelu02(v * sc + sh);
          if constexpr (RES) v += rreg[c][k];
          tile[c * g.CST + tid + k * CV_THREADS] = v;
          if constexpr (RES) {
            if (stgn && ((interior >> k) & 1u) && c0 + c < g.cin) stgn[(size_t)(c0 + c) * in_chan + goff[k]] = v;
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < WR; ++k) {
      const int idx = (tid + k * CV_THREADS) * 4;
      if (idx < WFL) *reinterpret_cast<floatx4 *>(wl + idx) = wreg[k];
    }
  };

#ifdef MVSN_DMA_STAMPS
  __shared__ unsigned long long ph_lds[8];
  const bool stamped = g_dma_stamps && blockIdx.x == gridDim.x / 3 && blockIdx.y == gridDim.y / 2 && tid == 0;
  unsigned long long ph_last = __builtin_readcyclecounter();
  const unsigned long long ph_t0 = ph_last;
  if (stamped) for (int k = 0; k < 8; ++k) ph_lds[k] = 0;
#endif
  stage_load(0);
  MFMA_PHASE(0);   // prologue
  for (int chunk = 0; chunk < g.nchunks; ++chunk) {
    __syncthreads();  // the previous chunk's MFMAs are done with LDS (and scsh is visible)
    MFMA_PHASE(1);   // barrier 1
    stage_store(chunk);
    MFMA_PHASE(2);   // transform + LDS writes (waits for the staged loads)
    __syncthreads();
    MFMA_PHASE(3);   // barrier 2
    if (chunk + 1 < g.nchunks) stage_load(chunk + 1);
    MFMA_PHASE(4);   // next chunk's loads issued
    // taps software-pipelined by hand: the next tap's weight and activation fragments are read from LDS
    // before the current tap's MFMAs are issued (the compiler otherwise waits lgkmcnt(0) per fragment)
    const float *wt = wl + lane;
    float fw[2][2], fb[2][NPT];
    auto read_tap = [&](int tap, int buf) {
      const int tz = tap / (KH * KW), ty = (tap / KW) % KH, tx = tap % KW;
      const float *bp = tile + (tz * g.HY + ty * g.dil) * g.XS + tx * g.dil;
      fw[buf][0] = wt[tap * 128];
      fw[buf][1] = CT == 2 ? wt[tap * 128 + 64] : 0.0f;
#pragma unroll
      for (int j = 0; j < NPT; ++j) fb[buf][j] = bp[lpos[j]];
    };
    if constexpr (KD == 1 && KW == 5 && STRIDE == 2 && V4) {
      // 5x5 stride 2 (extractor): a lane's five taps of a kernel row are the columns 2x .. 2x + 4 of one tile row --
      // three aligned 8-byte reads instead of five 4-byte ones, and (the 16 pixel lanes of a channel are 8 bytes
      // apart) spread over all 32 banks where the strided 4-byte reads of the four channel groups all fell on the
      // 16 even ones.  Row ky + 1 is read while row ky's 40 MFMAs are issued.
      float2 er[2][NPT][3];
      auto read_row = [&](int ky, int buf) {
        const float *bp = tile + ky * g.XS;
#pragma unroll
        for (int j = 0; j < NPT; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q) er[buf][j][q] = *reinterpret_cast<const float2 *>(bp + lpos[j] + 2 * q);
      };
      read_row(0, 0);
      fw[0][0] = wt[0], fw[0][1] = CT == 2 ? wt[64] : 0.0f;
#pragma unroll
      for (int ky = 0; ky < 5; ++ky) {
        const int rb = ky & 1;
        if (ky + 1 < 5) read_row(ky + 1, rb ^ 1);
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
          const int tap = ky * 5 + kx, cur = tap & 1;
          if (tap + 1 < NTAPS) {
            fw[cur ^ 1][0] = wt[(tap + 1) * 128];
            fw[cur ^ 1][1] = CT == 2 ? wt[(tap + 1) * 128 + 64] : 0.0f;
          }
#pragma unroll
          for (int j = 0; j < NPT; ++j) {
            const float a = (kx & 1) ? er[rb][j][kx >> 1].y : er[rb][j][kx >> 1].x;
            acc[j][0] = mfma16x16x4(a, fw[cur][0], acc[j][0]);
            if (CT == 2) acc[j][CT - 1] = mfma16x16x4(a, fw[cur][1], acc[j][CT - 1]);
          }
        }
      }
    } else {
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap) {
      const int cur = tap & 1;
      if (tap + 1 < NTAPS) read_tap(tap + 1, cur ^ 1);
#pragma unroll
      for (int j = 0; j < NPT; ++j) {
        acc[j][0] = mfma16x16x4(fb[cur][j], fw[cur][0], acc[j][0]);   // D = pixels x couts
        if (CT == 2) acc[j][CT - 1] = mfma16x16x4(fb[cur][j], fw[cur][1], acc[j][CT - 1]);
      }
    }
    }
    MFMA_PHASE(5);   // MFMAs issued
  }

  conv_epilogue<NPT, CT>(g, acc, z0, y0, x0, tid, n, tile_id, bias, out, out_partials);
#ifdef MVSN_DMA_STAMPS
  MFMA_PHASE(6);   // epilogue issued
  if (stamped) {
    for (int k = 0; k < 7; ++k) g_dma_stamps[32 + k] = ph_lds[k];
    g_dma_stamps[39] = __builtin_readcyclecounter() - ph_t0;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant (3x3 / 3x3x3, stride 1): the same implicit GEMM, but the haloed chunk tiles and the
// weight fragments go HBM -> LDS with global_load_lds (no VGPR round trip, no LDS store pass) into a
// two-stage ring, one barrier per chunk.  Wave w owns channel w of every 4-channel chunk: it issues
// that channel's rows as 16-byte pieces, 1 KB per instruction (per-lane source address from the aligned
// column x0 - halo (rounded up to 4); pieces outside the image read a zero word; requires cols % 4 == 0, otherwise the
// register-staged kernel runs), and
// once its own loads have landed it applies the fused input transform IN LDS on exactly those
// elements -- LeakyReLU(GN(.)), optionally + residual (a whole SimpleBasicBlock folded into the next
// layer's load) -- and writes the block output for its own output positions as a by-product.
//   MODE 0 plain input, 1 LReLU(GN(in)), 2 in_residual + LReLU(GN(in)) [+ out_staged]
// ---------------------------------------------------------------------------------------------
__device__ floatx4 g_zero16 = {0.f, 0.f, 0.f, 0.f};

#define MVSN_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define MVSN_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

template <int NPT, int KD, int IPC, int CT, int MODE>
__global__ __launch_bounds__(CV_THREADS, MODE <= 1 ? 4 : 2) void conv_dma_kernel(ConvGeom g, const float *__restrict__ in,
                                                                 const float *__restrict__ wpk,
                                                                 const float *__restrict__ bias,
                                                                 const float *__restrict__ in_stats,
                                                                 const float *__restrict__ in_gamma,
                                                                 const float *__restrict__ in_beta,
                                                                 const float *__restrict__ in_residual,
                                                                 float *__restrict__ out_staged,
                                                                 float *__restrict__ out,
                                                                 float *__restrict__ out_partials) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTAPS = KD * 9;
  constexpr int WFL = NTAPS * 128;
  constexpr int WSLOT = ((WFL + 255) / 256) * 256;
  constexpr int WRUNS = WSLOT / 256;            // 16-byte DMA runs (256 floats each) per chunk
  const int tile_floats = CV_CK * g.dCST;
  const int stage_floats = tile_floats * (MODE == 2 ? 2 : 1) + WSLOT;
  float *scsh = smem + 2 * stage_floats;        // 64

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y;
#ifdef MVSN_DMA_STAMPS
  unsigned long long *dbg = (blockIdx.x == gridDim.x / 3 && blockIdx.y == gridDim.y / 2 && tid == 0) ? g_dma_stamps : nullptr;
  int dbg_i = 0;
#endif
  DMA_STAMP();   // entry
#ifdef MVSN_DMA_STAMPS
  if (dbg) dbg[62] = wall_clock64();   // 100 MHz reference: (s_memtime span) / (this span) = shader clock
#endif
  const int tile_id = xcd_tile_index(blockIdx.x, g.tiles);
  const int trow = fdiv(tile_id, g.fd_ntx), txi = tile_id - trow * g.ntx;
  const int tzi = fdiv(trow, g.fd_nty), tyi = trow - tzi * g.nty;
  const int z0 = tzi * g.TZ, y0 = tyi * g.TY, x0 = txi * CV_TX;
  const int gz0 = z0 - g.pd, gy0 = y0 - g.ph, gx0 = x0 - g.pw;
  const size_t in_plane = (size_t)g.H * g.W, in_chan = (size_t)g.D * in_plane;
  const float *inn = in + (size_t)n * g.cin * in_chan;
  const float *resn = (MODE == 2 && in_residual) ? in_residual + (size_t)n * g.cin * in_chan : nullptr;
  float *stgn = (MODE == 2 && out_staged) ? out_staged + (size_t)n * g.cin * in_chan : nullptr;

  int lpos[NPT];   // see conv_mfma_kernel
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int pt = wave * NPT + j;
    const int xt = pt & 1, zz = fdiv(pt >> 1, g.fd_ty), yy = (pt >> 1) - zz * g.TY;
    const int xx = xt * 16 + (lane & 15);
    lpos[j] = (zz * g.HY + yy) * g.dXS + xx + (g.dpa - g.pw) + (lane >> 4) * g.dCST;
  }

  // DMA plan of this lane: piece i covers the 16-byte groups i*64 .. i*64+63 of the wave's channel, group
  // e = row * dq + q holding columns x0 - dpa + 4q .. + 3 of tile row `row` (aligned in global memory and in LDS;
  // cols % 4 == 0, so a group is entirely inside or entirely outside the image)
  int goff[IPC];
  unsigned inimg = 0, interior = 0;
  const int tile_groups = g.HZ * g.HY * g.dq;
#pragma unroll
  for (int i = 0; i < IPC; ++i) {
    const int e = i * 64 + lane;
    int off = -1;
    if (i < g.dipc && e < tile_groups) {
      const int row = fdiv(e, g.fd_dq), q = e - row * g.dq;
      const int z = fdiv(row, g.fd_hy), y = row - z * g.HY;
      const int gz = gz0 + z, gy = gy0 + y, gx = x0 - g.dpa + 4 * q;
      if (gz >= 0 && gz < g.D && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
        off = (gz * g.H + gy) * g.W + gx;
        inimg |= 1u << i;
        if (z >= g.pd && z < g.pd + g.TZ && y >= g.ph && y < g.ph + g.TY && 4 * q >= g.dpa && 4 * q < g.dpa + CV_TX) interior |= 1u << i;
      }
    }
    goff[i] = off;
  }

  if (MODE >= 1 && tid < 32) {
    const int grp = tid >> 3;
    const float mean = in_stats[((size_t)n * 4 + grp) * 2 + 0];
    const float rstd = in_stats[((size_t)n * 4 + grp) * 2 + 1];
    const float sc = rstd * in_gamma[tid];
    scsh[tid] = sc;
    scsh[32 + tid] = in_beta[tid] - mean * sc;
  }

  const float *zero = reinterpret_cast<const float *>(&g_zero16);
  auto issue = [&](int chunk, int stage) {
    float *st = smem + stage * stage_floats;
    const int c = chunk * CV_CK + wave;                      // this wave's channel
    const bool cok = c < g.cin;
    const float *src = inn + (size_t)(cok ? c : 0) * in_chan;
    float *dst = st + wave * g.dCST;
#pragma unroll
    for (int i = 0; i < IPC; ++i) {
      if (i < g.dipc) {  // IPC is a compile-time upper bound; pieces past g.dipc would leave the slot
        const float *p = (cok && goff[i] >= 0) ? src + goff[i] : zero;
        __builtin_amdgcn_global_load_lds(MVSN_GPTR(p), MVSN_LPTR(dst + i * 256), 16, 0, 0);
      }
    }
    if constexpr (MODE == 2) {
      const float *rsrc = resn ? resn + (size_t)(cok ? c : 0) * in_chan : nullptr;
      float *rdst = st + tile_floats + wave * g.dCST;
#pragma unroll
      for (int i = 0; i < IPC; ++i) {
        if (i < g.dipc) {
          const float *p = (rsrc && cok && goff[i] >= 0) ? rsrc + goff[i] : zero;
          __builtin_amdgcn_global_load_lds(MVSN_GPTR(p), MVSN_LPTR(rdst + i * 256), 16, 0, 0);
        }
      }
    }
    const float *wsrc = wpk + (size_t)chunk * WFL;
    float *wdst = st + tile_floats * (MODE == 2 ? 2 : 1);
#pragma unroll
    for (int r = 0; r < (WRUNS + CV_WAVES - 1) / CV_WAVES; ++r) {
      const int run = r * CV_WAVES + wave;
      if (run < WRUNS) {
        const int idx = run * 256 + lane * 4;
        const float *p = idx < WFL ? wsrc + idx : zero;
        __builtin_amdgcn_global_load_lds(MVSN_GPTR(p), MVSN_LPTR(wdst + run * 256), 16, 0, 0);
      }
    }
  };

  floatx4 acc[NPT][CT];
#pragma unroll
  for (int j = 0; j < NPT; ++j)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[j][t] = floatx4{0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  DMA_STAMP();   // prologue done, first chunk issued
  for (int chunk = 0; chunk < g.nchunks; ++chunk) {
    const int stage = chunk & 1;
    float *st = smem + stage * stage_floats;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA runs for `chunk` have landed
    DMA_STAMP();   // landed
    if constexpr (MODE >= 1) {
      // fused input transform on the elements this wave loaded (its channel of the chunk)
      if (chunk == 0) __syncthreads();   // scsh visible
      const int c = chunk * CV_CK + wave;
      if (c < g.cin) {
        const float sc = scsh[c], sh = scsh[32 + c];
        float *tl = st + wave * g.dCST + lane * 4;
        float *stg = stgn ? stgn + (size_t)c * in_chan : nullptr;
#pragma unroll
        for (int i = 0; i < IPC; ++i) {
          if ((inimg >> i) & 1u) {
            floatx4 v = *reinterpret_cast<floatx4 *>(tl + i * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = lrelu02(v[r] * sc + sh);
            if constexpr (MODE == 2) {
              const floatx4 rv = *reinterpret_cast<const floatx4 *>(tl + tile_floats + i * 256);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += rv[r];
            }
            *reinterpret_cast<floatx4 *>(tl + i * 256) = v;
            if (MODE == 2 && stg && ((interior >> i) & 1u)) *reinterpret_cast<floatx4 *>(stg + goff[i]) = v;
          }
        }
      }
    }
    __syncthreads();  // every wave's channel is in place; everyone is done with the other stage
    DMA_STAMP();   // barrier passed
    if (chunk + 1 < g.nchunks) issue(chunk + 1, stage ^ 1);
    DMA_STAMP();   // next chunk issued

    const float *tile = st;
    const float *wt = st + tile_floats * (MODE == 2 ? 2 : 1) + lane;
    float fw[2][2], fb[2][NPT];
    auto read_tap = [&](int tap, int buf) {
      const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
      const float *bp = tile + (tz * g.HY + ty * g.dil) * g.dXS + tx * g.dil;
      fw[buf][0] = wt[tap * 128];
      fw[buf][1] = CT == 2 ? wt[tap * 128 + 64] : 0.0f;
#pragma unroll
      for (int j = 0; j < NPT; ++j) fb[buf][j] = bp[lpos[j]];
    };
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap) {   // next tap's fragments in flight behind this tap's MFMAs
      const int cur = tap & 1;
      if (tap + 1 < NTAPS) read_tap(tap + 1, cur ^ 1);
#pragma unroll
      for (int j = 0; j < NPT; ++j) {
        acc[j][0] = mfma16x16x4(fb[cur][j], fw[cur][0], acc[j][0]);   // D = pixels x couts
        if (CT == 2) acc[j][CT - 1] = mfma16x16x4(fb[cur][j], fw[cur][1], acc[j][CT - 1]);
      }
    }
    DMA_STAMP();   // MFMAs issued
  }

  conv_epilogue<NPT, CT>(g, acc, z0, y0, x0, tid, n, tile_id, bias, out, out_partials);
  DMA_STAMP();   // epilogue issued
#ifdef MVSN_DMA_STAMPS
  if (dbg) dbg[63] = wall_clock64();
#endif
}

// ---------------------------------------------------------------------------------------------
// The extractor's first layer: 5x5, stride 2, 3 -> 32 channels, no bias (FeatureNetwork.conv0,
// multi_view_stereonet.py:91).  In the generic kernel a 3-channel input is ONE k-chunk of 4 (a quarter of every MFMA
// multiplies the zero pad channel) and a workgroup lives for 25 taps only.  Here the reduction runs over the flattened
// K = 3 x 25 = 75 (channel, tap) pairs: 19 k-steps of 4 instead of 25, the weight fragments of all of them stay in
// 38 registers for the workgroup's life (gathered once from the generic packed layout), the whole haloed input tile
// (3 x 19 x 72 floats) is staged once, and the MFMA loop reads only its A fragments from LDS.
// A = activations (16 pixels x 4 k), B = weights (4 k x 16 couts), D = pixels x couts as everywhere (16-byte stores).
// ---------------------------------------------------------------------------------------------
constexpr int HD_TY = 8, HD_TX = 32, HD_ROWS = 2 * HD_TY + 3, HD_XS = 2 * HD_TX + 8;   // 19 rows of 72 floats
constexpr int HD_CST = HD_ROWS * HD_XS + 4;                                           // channel stride (floats)
constexpr int HD_KSTEPS = 19;
static_assert(3 * HD_CST < 65536, "k-step offsets are packed as 16-bit values");

#ifndef MVSN_HEAD_WGS_PER_CU
#define MVSN_HEAD_WGS_PER_CU 3
#endif
#ifndef MVSN_HEAD_CNTWAIT   // 1: counted wait at the top of a tile (measured equal; 0 does not lean on retirement order)
#define MVSN_HEAD_CNTWAIT 0
#endif
#ifndef MVSN_HEAD_ABLATE   // timing experiments only: 1 no stores, 2 no tile fetch, 4 no multiplies
#define MVSN_HEAD_ABLATE 0
#endif
#ifndef MVSN_HEAD_ST_AUX
#define MVSN_HEAD_ST_AUX 0
#endif
__global__ __launch_bounds__(256, MVSN_HEAD_WGS_PER_CU) void conv5x5s2_head_kernel(const float *__restrict__ in, const float *__restrict__ wpk,
                                                             const float *__restrict__ bias, int H, int W, int Ho,
                                                             int Wo, int ntx, int tiles, int total,
                                                             float *__restrict__ out) {
  // Round 4: persistent workgroups (MVSN_HEAD_WGS_PER_CU per CU) walking (image, tile) ids with a two-slot tile ring
  // filled by `buffer_load_dwordx4 .. lds`: tile i + 1 lands while tile i is multiplied, the weight fragments are
  // fetched once per workgroup.  (One short-lived workgroup per tile -- the round-2 form -- spent two thirds of its
  // life in the fragment gather, the staging through registers and the store tail: 0.41 of the fp32 roof.)
  __shared__ __attribute__((aligned(16))) float ring[2][3 * HD_CST];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t plane = (size_t)H * W;

  // weight fragments of every k-step (k = 4 ks + (lane>>4) -> channel k / 25, tap k % 25) out of the generic packed
  // layout [tap][cout tile][lane = cin*16 + cout]; the LDS offset of this lane's k inside the tile
  const int kk = lane >> 4, cl = lane & 15;
  float wf[HD_KSTEPS][2];
  unsigned kpk[(HD_KSTEPS + 1) / 2] = {};   // tile offsets of the k-steps, two 16-bit values per register
#pragma unroll
  for (int ks = 0; ks < HD_KSTEPS; ++ks) {
    const int k = 4 * ks + kk;
    const bool live = k < 75;
    const int ch = live ? k / 25 : 0, tap = live ? k - ch * 25 : 0;
    wf[ks][0] = live ? wpk[tap * 128 + ch * 16 + cl] : 0.0f;
    wf[ks][1] = live ? wpk[tap * 128 + 64 + ch * 16 + cl] : 0.0f;
    const unsigned ko = ch * HD_CST + (tap / 5) * HD_XS + (tap % 5) + 2;   // input column of tap tx: 2 xx + tx - 2 -> tile column + 2
    kpk[ks >> 1] |= (ks & 1) ? ko << 16 : ko;
  }
  const float bv0 = bias ? bias[cl] : 0.0f, bv1 = bias ? bias[16 + cl] : 0.0f;

  // The haloed tile: rows 2 y0 - 2 .. + 18, columns 2 x0 - 4 .. + 67 as 16-byte groups (W % 4 == 0: a group is entirely
  // inside or entirely outside the image).  A channel's 342 groups are contiguous in LDS, so piece k of channel c is
  // 64 consecutive groups = one DMA instruction; rows outside the image fall outside the range of the channel plane's
  // descriptor (zeros from the hardware), columns outside get the offset 0xFFFFFFFF.
  constexpr int CG = HD_ROWS * (HD_XS / 4);            // 342 groups per channel
  constexpr int CP = (CG + 63) / 64;                   // 6 pieces per channel
  constexpr int NP = (3 * CP + 3) / 4;                 // pieces per wave (the last round only waves 0, 1)
  int prel[NP], pqx[NP];                               // (row - 2) * W + 4 q - 4 and 4 q - 4 of this lane's group; < 0 row: none
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pc = wave + 4 * i, k = pc % CP;
    const int gi = k * 64 + lane, row = gi / (HD_XS / 4), q = gi - row * (HD_XS / 4);
    pqx[i] = gi < CG ? 4 * q - 4 : -0x40000000;       // a group past the channel's 342: never inside [0, W)
    prel[i] = (row - 2) * W + 4 * q - 4;
  }
  auto issue = [&](int id, int slot) {
    const int n = id / tiles, t = id - n * tiles;
    const int tyi = t / ntx, txi = t - tyi * ntx;
    const int org = 2 * (tyi * HD_TY) * W + 2 * (txi * HD_TX), x2 = 2 * (txi * HD_TX);
    const float *inn = in + (size_t)n * 3 * plane;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int pc = wave + 4 * i;                     // (uniform) piece of this wave
      if (pc < 3 * CP) {
        const int c = pc / CP, k = pc - c * CP;
        const int gx = x2 + pqx[i];
        const unsigned voff = (gx >= 0 && gx < W) ? (unsigned)((org + prel[i]) * 4) : 0xFFFFFFFFu;   // rows above wrap out of range too
        if (pqx[i] > -0x40000000 && !(MVSN_HEAD_ABLATE & 2))
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(inn + (size_t)c * plane), 0, (int)(plane * 4), 0x00020000),
              (__attribute__((address_space(3))) void *)(&ring[slot][c * HD_CST + k * 256]), 16, (int)voff, 0, 0, 0);
      }
    }
  };

  // pixel tile pt = wave*4 + j: output row pt >> 1, columns (pt & 1) * 16 .. + 15
  // (tile j of the wave sits at a constant distance from tile 0 -- one address per k-step, the rest as the reads' immediates)
  const int pbase0 = (4 * wave) * HD_XS + 2 * cl;
  const int oplane = Ho * Wo;

  int id = blockIdx.x;
  if (id < total) issue(id, 0);
  for (int it = 0; id < total; id += gridDim.x, ++it) {
    const int slot = it & 1;
    // (the counted form: behind this tile's DMA instructions the wave has issued exactly the previous tile's eight
    // stores, so the tile has landed when at most eight are outstanding -- measured equal to the full wait:
    // profiles/r04_micro)
    if (MVSN_HEAD_CNTWAIT && it > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // this tile has landed for every wave; everyone is done reading the other slot
    if (id + (int)gridDim.x < total) issue(id + gridDim.x, slot ^ 1);
    const float *tile0 = &ring[0][0] + slot * (3 * HD_CST) + pbase0;

    floatx4 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < ((MVSN_HEAD_ABLATE & 4) ? 1 : HD_KSTEPS); ++ks) {
      const int ko = (ks & 1) ? (int)(kpk[ks >> 1] >> 16) : (int)(kpk[ks >> 1] & 0xffffu);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = tile0[ko + (j >> 1) * 2 * HD_XS + (j & 1) * 32];
        acc[j][0] = mfma16x16x4(a, wf[ks][0], acc[j][0]);
        acc[j][1] = mfma16x16x4(a, wf[ks][1], acc[j][1]);
      }
    }

    // lane: cout t*16 + cl, the four consecutive pixels 4*(lane>>4) .. + 3 of each pixel tile.  Stores through a
    // descriptor of the image's 32 output planes: one 32-bit offset per lane and column half, the row / cout-tile
    // part as the instruction's scalar offset; columns past the image carry an out-of-range offset and are dropped
    // by the hardware, rows past it likewise.
    const int n = id / tiles, t = id - n * tiles;
    const int tyi = t / ntx, txi = t - tyi * ntx;
    const int y0 = tyi * HD_TY, x0 = txi * HD_TX;
    const __amdgpu_buffer_rsrc_t osrd =
        __builtin_amdgcn_make_buffer_rsrc(out + (size_t)n * 32 * oplane, 0, (int)((size_t)32 * oplane * 4), 0x00020000);
    unsigned ovoff[2];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
      const int ox = x0 + hx * 16 + 4 * kk;   // Wo % 4 == 0: all four pixels or none
      ovoff[hx] = ox < Wo ? (unsigned)((cl * oplane + ox) * 4) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const float bv = tt ? bv1 : bv0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pt = wave * 4 + j;
        const int oy = y0 + (pt >> 1);
        floatx4 v = acc[j][tt];
        v[0] += bv, v[1] += bv, v[2] += bv, v[3] += bv;
        // (a row past the image: the same store with an out-of-range offset -- always eight stores per tile)
        if (!(MVSN_HEAD_ABLATE & 1) || v[0] == 12345.f)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), osrd,
                                               oy < Ho ? (int)ovoff[pt & 1] : -1, oy < Ho ? (tt * 16 * oplane + oy * Wo) * 4 : 0, MVSN_HEAD_ST_AUX);
      }
    }
  }   // tiles of this workgroup
}

// One workgroup per sample: gn_finalize_block (mvsn_common.h).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float *__restrict__ partials, int tiles,
                                                          float *__restrict__ stats) {
  const int n = blockIdx.x;
  gn_finalize_block(partials + (size_t)n * tiles * 12, tiles, stats + (size_t)n * 8);
}

// Split finalize: slice k of sample n's records -> 12 double sums; then one thread per (sample, group) adds the K slices in
// order.  (A level-0 layer of a 1024x512 frame leaves 32768 records = 1.5 MB per sample: one workgroup per sample read
// them in ~45 us -- 14 such launches per batch-1 forward, and at 32 samples 32 of the 256 CUs were at work.)
__global__ __launch_bounds__(256) void gn_partial_kernel(const float *__restrict__ partials, int tiles, int per, int K,
                                                         double *__restrict__ scratch) {
  const int k = blockIdx.x, n = blockIdx.y;
  const int lo = k * per, cnt = tiles - lo < per ? tiles - lo : per;
  gn_partial_block(partials + ((size_t)n * tiles + lo) * 12, cnt, scratch + ((size_t)n * K + k) * 12);
}
__global__ __launch_bounds__(256) void gn_combine_kernel(const double *__restrict__ scratch, int n, int K,
                                                         float *__restrict__ stats) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 4) return;
  const int s = t >> 2, g = t & 3;
  double N = 0.0, S = 0.0, Q = 0.0;
  for (int k = 0; k < K; ++k) {
    const double *p = scratch + ((size_t)s * K + k) * 12 + g * 3;
    N += p[0], S += p[1], Q += p[2];
  }
  const double mean = S / N;
  double var = Q / N - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[(size_t)s * 8 + g * 2 + 0] = (float)mean;
  stats[(size_t)s * 8 + g * 2 + 1] = (float)(1.0 / sqrt(var + (double)GN_FINALIZE_EPS));
}
// slices per sample: a function of the records per sample ALONE (a sample's statistics must not depend on the batch it
// travels in); 1 up to 2048 records -- the range in which consumers may form the statistics themselves (gn_stats_here)
static inline int gn_split_slices(int tiles) { return tiles <= 2048 ? 1 : (tiles + 2047) / 2048 < 16 ? (tiles + 2047) / 2048 : 16; }

// out = [residual +] LeakyReLU(GroupNorm(x)), (N,32,spatial), float4 per thread.
constexpr int GN_APPLY_MAX_N = 65535 / 32;   // samples per launch (grid.y = sample * 32 + channel)

// RRAW: the residual is itself a raw conv output whose LeakyReLU(GroupNorm(.)) was never materialised
// (the head of a refiner): out = LReLU(GN(x)) + LReLU(GN_r(residual)).
template <bool RRAW>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, const float *__restrict__ stats,
                                                       const float *__restrict__ gamma,
                                                       const float *__restrict__ beta,
                                                       const float *__restrict__ residual,
                                                       const float *__restrict__ r_stats,
                                                       const float *__restrict__ r_gamma,
                                                       const float *__restrict__ r_beta, long spatial,
                                                       int stat_tiles, float *__restrict__ out) {
  const int plane = blockIdx.y;  // n*32 + c
  const int n = plane >> 5, c = plane & 31;
  __shared__ float gst[8];
  const float *st = gn_stats_here<true>(stats, stat_tiles, n, gst, c >> 3);   // (stat_tiles > 0: x's statistics from its records)
  const float mean = st[(c >> 3) * 2 + 0];
  const float rstd = st[(c >> 3) * 2 + 1];
  const float sc = rstd * gamma[c];
  const float sh = beta[c] - mean * sc;
  float rsc = 1.0f, rsh = 0.0f;
  if constexpr (RRAW) {
    const float rm = r_stats[((size_t)n * 4 + (c >> 3)) * 2 + 0];
    rsc = r_stats[((size_t)n * 4 + (c >> 3)) * 2 + 1] * r_gamma[c];
    rsh = r_beta[c] - rm * rsc;
  }
  auto res = [&](float v) { return RRAW ? lrelu02(v * rsc + rsh) : v; };
  const float *xp = x + (size_t)plane * spatial;
  const float *rp = residual ? residual + (size_t)plane * spatial : nullptr;
  float *op = out + (size_t)plane * spatial;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  const bool vec_ok = (spatial & 3) == 0;
  if (vec_ok) {
    // streaming pass: non-temporal 16-byte accesses, four independent iterations in flight per thread
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (; i + 3 * stride < spatial; i += 4 * stride) {
      floatx4 v[4], o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(xp + i + u * stride));
      if (rp) {
        floatx4 rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rv[u] = __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(rp + i + u * stride));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) o[u][k] = lrelu02(v[u][k] * sc + sh) + res(rv[u][k]);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) o[u][k] = lrelu02(v[u][k] * sc + sh);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(o[u], reinterpret_cast<floatx4 *>(op + i + u * stride));
    }
    for (; i < spatial; i += stride) {
      floatx4 v = *reinterpret_cast<const floatx4 *>(xp + i);
      floatx4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = lrelu02(v[k] * sc + sh);
      if (rp) {
        floatx4 rv = *reinterpret_cast<const floatx4 *>(rp + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] += res(rv[k]);
      }
      *reinterpret_cast<floatx4 *>(op + i) = o;
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < spatial; i += (long)gridDim.x * blockDim.x) {
      float o = lrelu02(xp[i] * sc + sh);
      if (rp) o += res(rp[i]);
      op[i] = o;
    }
  }
}

// ---- MFMA fragment self-test ---------------------------------------------------------------
__global__ void mfma_selftest_kernel(int *bad) {
  const int lane = threadIdx.x;
  // A[i][k] = 1 + i + 17k (asymmetric), B[k][j] = 3 + 5k - 2j
  const float a = 1.0f + (float)(lane & 15) + 17.0f * (float)(lane >> 4);
  const float b = 3.0f + 5.0f * (float)(lane >> 4) - 2.0f * (float)(lane & 15);
  floatx4 c = {0.f, 0.f, 0.f, 0.f};
  c = mfma16x16x4(a, b, c);
  int wrong = 0;
  for (int r = 0; r < 4; ++r) {
    const int i = (lane >> 4) * 4 + r, j = lane & 15;
    float ref = 0.f;
    for (int k = 0; k < 4; ++k) ref += (1.0f + i + 17.0f * k) * (3.0f + 5.0f * k - 2.0f * j);
    if (fabsf(ref - c[r]) > 1e-3f) wrong++;
  }
  if (wrong) atomicAdd(bad, wrong);
}

}  // namespace mvsn

extern "C" int mvsn_conv_bf16x3_supported(const mvsn_conv_desc *desc) {
  mvsn::Bf16x3Geom g;
  return mvsn::bf16x3_geom(desc, &g) ? 1 : 0;
}

extern "C" int mvsn_conv_forward_bf16_storage(const mvsn_conv_desc *desc, const void *in, int in_is_bf16,
                                              const float *weight_packed, const float *bias, const float *in_stats,
                                              const float *in_gamma, const float *in_beta, void *out, int out_is_bf16,
                                              float *out_partials, mvsn_stream_t stream) {
  using namespace mvsn;
  MVSN_REQUIRE(desc && in && weight_packed && out, MVSN_E_BADARG, "mvsn_conv_forward_bf16_storage: null pointer");
  MVSN_REQUIRE(desc->precision == MVSN_CONV_BF16 && desc->kd == 3, MVSN_E_BADARG,
               "mvsn_conv_forward_bf16_storage: 3x3x3 layers with desc.precision = MVSN_CONV_BF16");
  Bf16x3Geom bg;
  MVSN_REQUIRE(bf16x3_geom(desc, &bg), MVSN_E_BADARG, "mvsn_conv_forward_bf16_storage: layer has no bf16 form");
  MVSN_REQUIRE(!in_stats || (in_gamma && in_beta), MVSN_E_BADARG, "mvsn_conv_forward_bf16_storage: input transform needs gamma/beta");
  MVSN_REQUIRE(bg.n <= 65535, MVSN_E_TOOLARGE, "mvsn_conv_forward_bf16_storage: batch too large for one launch");
  MVSN_REQUIRE(((size_t)in & 3) == 0 && ((size_t)out & 7) == 0, MVSN_E_BADARG, "mvsn_conv_forward_bf16_storage: alignment");
  MVSN_REQUIRE(!in_is_bf16 || (desc->cols & 1) == 0, MVSN_E_BADARG,
               "mvsn_conv_forward_bf16_storage: a bf16 input needs an even number of columns (aligned two-element loads)");
  return bf16_storage_launch(bg, in, in_is_bf16 != 0, weight_packed, bias, in_stats, in_gamma, in_beta, out,
                             out_is_bf16 != 0, out_partials, (hipStream_t)stream);
}

extern "C" int mvsn_conv_winograd_supported(const mvsn_conv_desc *desc) {
  mvsn::WinoGeom g;
  return mvsn::wino_geom(desc, &g) ? 1 : 0;
}

extern "C" size_t mvsn_conv_packed_floats(const mvsn_conv_desc *desc) {
  if (desc && desc->precision == MVSN_CONV_FP32_WINO) {
    mvsn::WinoGeom wg;
    return mvsn::wino_geom(desc, &wg) ? wg.packed_floats : 0;
  }
  if (desc && (desc->precision == MVSN_CONV_BF16X3 || desc->precision == MVSN_CONV_BF16)) {
    mvsn::Bf16x3Geom bg;
    return mvsn::bf16x3_geom(desc, &bg) ? (size_t)desc->kd * 9 * 1024 : 0;   // [tap][2][2][64][8] bf16
  }
  mvsn::ConvGeom g;
  if (!mvsn::make_geom(desc, &g)) return 0;
  return (size_t)g.nchunks * g.wfloats_chunk;
}

extern "C" int mvsn_conv_num_tiles(const mvsn_conv_desc *desc) {
  // number of GroupNorm partial records per sample: one per (tile, wave, 16-lane row)
  if (desc && desc->precision == MVSN_CONV_FP32_WINO) {
    mvsn::WinoGeom wg;
    return mvsn::wino_geom(desc, &wg) ? (int)(mvsn::wino_items(wg) * 32) : 0;   // per (plane,) tile: 8 waves x 4 lane rows
  }
  if (desc && (desc->precision == MVSN_CONV_BF16X3 || desc->precision == MVSN_CONV_BF16)) {
    mvsn::Bf16x3Geom bg;
    return mvsn::bf16x3_geom(desc, &bg) ? bg.tiles * 4 : 0;
  }
  mvsn::ConvGeom g;
  if (!mvsn::make_geom(desc, &g)) return 0;
  return g.tiles * 16;   // per tile: 4 waves x 4 lane rows
}

extern "C" int mvsn_conv_pack_weights(const mvsn_conv_desc *desc, const float *weight, float *packed,
                                      mvsn_stream_t stream) {
  MVSN_REQUIRE(weight && packed, MVSN_E_BADARG, "mvsn_conv_pack_weights: null pointer");
  if (desc && desc->precision == MVSN_CONV_FP32_WINO) {
    mvsn::WinoGeom wg;
    MVSN_REQUIRE(mvsn::wino_geom(desc, &wg), MVSN_E_BADARG, "mvsn_conv_pack_weights: layer has no Winograd form");
    return mvsn::wino_pack(desc, weight, packed, (hipStream_t)stream);
  }
  if (desc && (desc->precision == MVSN_CONV_BF16X3 || desc->precision == MVSN_CONV_BF16)) {
    mvsn::Bf16x3Geom bg;
    MVSN_REQUIRE(mvsn::bf16x3_geom(desc, &bg), MVSN_E_BADARG, "mvsn_conv_pack_weights: layer has no bf16x3 form");
    return mvsn::bf16x3_pack(desc, weight, packed, (hipStream_t)stream);
  }
  mvsn::ConvGeom g;
  MVSN_REQUIRE(mvsn::make_geom(desc, &g), MVSN_E_BADARG, "mvsn_conv_pack_weights: unsupported descriptor");
  const int total = g.nchunks * g.ntaps * 128;
  hipLaunchKernelGGL(mvsn::conv_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight,
                     g.cin, g.cout, g.ntaps, g.nchunks, packed);
  return mvsn::check_launch("mvsn_conv_pack_weights");
}

extern "C" int mvsn_conv_forward_blocks(const mvsn_conv_desc *desc, const float *const *in_blocks,
                                        const int *block_channels, int num_blocks, const float *weight_packed,
                                        const float *bias, float *out, float *out_partials, mvsn_stream_t stream) {
  using namespace mvsn;
  MVSN_REQUIRE(desc && in_blocks && block_channels && weight_packed && out, MVSN_E_BADARG,
               "mvsn_conv_forward_blocks: null pointer");
  MVSN_REQUIRE(num_blocks >= 1 && num_blocks <= 3, MVSN_E_BADARG, "mvsn_conv_forward_blocks: 1..3 blocks");
  MVSN_REQUIRE(desc->precision == MVSN_CONV_FP32_WINO, MVSN_E_BADARG,
               "mvsn_conv_forward_blocks: only the Winograd form takes channel blocks");
  WinoGeom wg;
  MVSN_REQUIRE(wino_geom(desc, &wg), MVSN_E_BADARG, "mvsn_conv_forward_blocks: layer has no Winograd form");
  MVSN_REQUIRE(wg.n <= 65535, MVSN_E_TOOLARGE, "mvsn_conv_forward_blocks: batch too large for one launch");
  int sum = 0;
  for (int b = 0; b < num_blocks; ++b) {
    MVSN_REQUIRE(in_blocks[b] && block_channels[b] >= 1, MVSN_E_BADARG, "mvsn_conv_forward_blocks: empty block");
    MVSN_REQUIRE((reinterpret_cast<uintptr_t>(in_blocks[b]) & 15) == 0, MVSN_E_BADARG,
                 "mvsn_conv_forward_blocks: blocks must be 16-byte aligned");
    sum += block_channels[b];
  }
  MVSN_REQUIRE(sum == wg.cin, MVSN_E_BADARG, "mvsn_conv_forward_blocks: block channels do not add up to c_in");
  WinoBlocks wb;
  wb.cb0 = block_channels[0];
  wb.cb1 = num_blocks > 1 ? block_channels[1] : 0;
  wb.in1 = num_blocks > 1 ? in_blocks[1] : in_blocks[0];
  wb.in2 = num_blocks > 2 ? in_blocks[2] : in_blocks[0];
  return wino_launch(wg, in_blocks[0], weight_packed, bias, nullptr, nullptr, nullptr, out, out_partials,
                     (hipStream_t)stream, &wb);
}

extern "C" int mvsn_conv_forward(const mvsn_conv_desc *desc, const float *in, const float *weight_packed,
                                 const float *bias, const float *in_stats, const float *in_gamma,
                                 const float *in_beta, const float *in_residual, float *out_staged, float *out,
                                 float *out_partials, mvsn_stream_t stream) {
  using namespace mvsn;
  MVSN_REQUIRE(in && weight_packed && out, MVSN_E_BADARG, "mvsn_conv_forward: null pointer");
  if (desc && desc->precision == MVSN_CONV_FP32_WINO) {
    WinoGeom wg;
    MVSN_REQUIRE(wino_geom(desc, &wg), MVSN_E_BADARG, "mvsn_conv_forward: layer has no Winograd form");
    MVSN_REQUIRE(!in_residual && !out_staged, MVSN_E_BADARG, "mvsn_conv_forward: Winograd form has no residual folding");
    MVSN_REQUIRE(!in_stats || (in_gamma && in_beta && wg.cin == 32), MVSN_E_BADARG,
                 "mvsn_conv_forward: input transform needs gamma/beta and 32 channels");
    MVSN_REQUIRE(wg.n <= 65535, MVSN_E_TOOLARGE, "mvsn_conv_forward: batch too large for one launch");
    return wino_launch(wg, in, weight_packed, bias, in_stats, in_gamma, in_beta, out, out_partials,
                       (hipStream_t)stream);
  }
  if (desc && (desc->precision == MVSN_CONV_BF16X3 || desc->precision == MVSN_CONV_BF16)) {
    Bf16x3Geom bg;
    MVSN_REQUIRE(bf16x3_geom(desc, &bg), MVSN_E_BADARG, "mvsn_conv_forward: layer has no bf16x3 form");
    MVSN_REQUIRE(!in_residual && !out_staged, MVSN_E_BADARG, "mvsn_conv_forward: bf16x3 has no residual folding");
    MVSN_REQUIRE(!in_stats || (in_gamma && in_beta), MVSN_E_BADARG, "mvsn_conv_forward: input transform needs gamma/beta");
    MVSN_REQUIRE(bg.n <= 65535, MVSN_E_TOOLARGE, "mvsn_conv_forward: batch too large for one launch");
    return bf16x3_launch(bg, in, weight_packed, bias, in_stats, in_gamma, in_beta, out, out_partials,
                         (hipStream_t)stream);
  }
  ConvGeom g;
  MVSN_REQUIRE(make_geom(desc, &g), MVSN_E_BADARG, "mvsn_conv_forward: unsupported descriptor");
  // ---- the extractor's 3 -> 32 5x5 stride-2 head: K = 75 packing (no fused transform / statistics on this layer) ----
  if (desc->precision == MVSN_CONV_FP32 && desc->c_in == 3 && desc->c_out == 32 && desc->kd == 1 && desc->kh == 5 &&
      desc->kw == 5 && desc->stride == 2 && desc->dilation == 1 && desc->depth == 1 && (desc->cols & 7) == 0 &&
      // descriptor ranges / 32-bit offsets of the kernel: the input plane AND a sample's full 32-channel output extent
      // ((int)(32 * oplane * 4) is the output descriptor's range, (cl * oplane + ox) * 4 a store offset)
      (size_t)desc->rows * desc->cols * 4 < ((size_t)1 << 31) &&
      (size_t)32 * ((desc->rows - 1) / 2 + 1) * ((desc->cols - 1) / 2 + 1) * 4 < ((size_t)1 << 31) &&
      !in_stats && !in_residual && !out_staged && !out_partials && ((size_t)in & 15) == 0 && ((size_t)out & 15) == 0) {
    MVSN_REQUIRE(desc->n <= 65535, MVSN_E_TOOLARGE, "mvsn_conv_forward: batch too large for one launch");
    const int Ho = (desc->rows - 1) / 2 + 1, Wo = (desc->cols - 1) / 2 + 1;
    const int nty = (Ho + HD_TY - 1) / HD_TY, ntx = (Wo + HD_TX - 1) / HD_TX, tiles = nty * ntx;
    // (persistent workgroups, MVSN_HEAD_WGS_PER_CU per CU -- three without a spill -- that cover each other's barrier and
    // store tails; each walks its tiles through a two-slot DMA ring)
    MVSN_REQUIRE((long long)tiles * desc->n < (1ll << 31) - 65536, MVSN_E_TOOLARGE, "mvsn_conv_forward: batch too large for one launch");
    const int total = tiles * desc->n;
    const int gx = std::min(total, MVSN_HEAD_WGS_PER_CU * device_cus());
    hipLaunchKernelGGL(conv5x5s2_head_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, in, weight_packed, bias,
                       desc->rows, desc->cols, Ho, Wo, ntx, tiles, total, out);
    return check_launch("mvsn_conv_forward(5x5 stride-2 head)");
  }
  MVSN_REQUIRE(!in_stats || (in_gamma && in_beta), MVSN_E_BADARG, "mvsn_conv_forward: input transform needs gamma/beta");
  MVSN_REQUIRE(!in_stats || g.cin == 32, MVSN_E_BADARG, "mvsn_conv_forward: input transform needs 32 channels");
  MVSN_REQUIRE(!out_partials || g.cout == 32, MVSN_E_BADARG, "mvsn_conv_forward: partials need 32 output channels");
  MVSN_REQUIRE(!in_residual || in_stats, MVSN_E_BADARG, "mvsn_conv_forward: in_residual needs the input transform");
  MVSN_REQUIRE(!(in_residual || out_staged) || (g.kd == 1 && g.kh == 3 && g.stride == 1), MVSN_E_BADARG,
               "mvsn_conv_forward: residual / staged output only for 2-D 3x3 stride-1 layers");
  MVSN_REQUIRE(g.n <= 65535, MVSN_E_TOOLARGE, "mvsn_conv_forward: batch too large for one launch");
  dim3 grid(g.tiles, g.n);
  static_assert(CV_CK == CV_WAVES, "one DMA channel per wave");
  // ---- LDS-DMA pipeline for the 2-D 3x3 stride-1 layers with dilation <= 4 ---------------------
  // (measured on MI355X, B=128: 2-5 % faster than register staging there, 7 % slower on the 3-D
  // layers where the in-LDS transform pass is exposed, and 8 % slower at dilation 8)
  if (g.kd == 1 && g.kh == 3 && g.stride == 1 && g.dil <= MVSN_DMA_MAX_DIL && g.dma_ok) {
    const int mode = (in_residual || out_staged) ? 2 : (in_stats ? 1 : 0);
    const size_t stage = (size_t)g.dma_stage_floats + (mode == 2 ? (size_t)CV_CK * g.dCST : 0);
    const size_t lds = (2 * stage + 64 + 16) * sizeof(float);
    const bool one = g.cout <= 16;
#define MVSN_DMA_LAUNCH(...)                                                                                      \
  do {                                                                                                            \
    auto kern = conv_dma_kernel<__VA_ARGS__>;                                                                     \
    static LdsOptIn opt;                                                                                          \
    if (int rc = ensure_lds(opt, (const void *)kern, lds, "mvsn_conv_forward(dma)")) return rc;                   \
    hipLaunchKernelGGL(kern, grid, dim3(CV_THREADS), lds, (hipStream_t)stream, g, in, weight_packed, bias, in_stats, \
                       in_gamma, in_beta, in_residual, out_staged, out, out_partials);                            \
    return check_launch("mvsn_conv_forward(dma)");                                                                \
  } while (0)
#define MVSN_DMA_MODES(NPTV, KDV, IPCV)                                                          \
  do {                                                                                           \
    if (one) {                                                                                   \
      if (mode == 0) MVSN_DMA_LAUNCH(NPTV, KDV, IPCV, 1, 0);                                     \
      else if (mode == 1) MVSN_DMA_LAUNCH(NPTV, KDV, IPCV, 1, 1);                                \
      else MVSN_DMA_LAUNCH(NPTV, KDV, IPCV, 1, 2);                                               \
    } else {                                                                                     \
      if (mode == 0) MVSN_DMA_LAUNCH(NPTV, KDV, IPCV, 2, 0);                                     \
      else if (mode == 1) MVSN_DMA_LAUNCH(NPTV, KDV, IPCV, 2, 1);                                \
      else MVSN_DMA_LAUNCH(NPTV, KDV, IPCV, 2, 2);                                               \
    }                                                                                            \
  } while (0)
    if (lds <= 160 * 1024 && g.dipc <= 6) {
      if (g.TY == 16) {
        if (g.dipc <= 4) MVSN_DMA_MODES(8, 1, 4);
        else MVSN_DMA_MODES(8, 1, 6);
      } else {
        if (g.dipc <= 4) MVSN_DMA_MODES(4, 1, 4);
        else MVSN_DMA_MODES(4, 1, 6);
      }
    }
#undef MVSN_DMA_MODES
#undef MVSN_DMA_LAUNCH
  }
#define MVSN_CONV_LAUNCH(...)                                                                                    \
  do {                                                                                                           \
    auto kern = conv_mfma_kernel<__VA_ARGS__>;                                                                   \
    static LdsOptIn opt;                                                                                         \
    if (int rc = ensure_lds(opt, (const void *)kern, g.lds_bytes, "mvsn_conv_forward")) return rc;               \
    hipLaunchKernelGGL(kern, grid, dim3(CV_THREADS), g.lds_bytes, (hipStream_t)stream, g, in, weight_packed, bias, \
                       in_stats, in_gamma, in_beta, in_residual, out_staged, out, out_partials);                 \
  } while (0)
  const bool one_tile = g.cout <= 16;
  const bool se3 = g.se <= 3;
  const bool res = in_residual != nullptr || out_staged != nullptr;
#define MVSN_CONV_2D(NPTV)                                                                      \
  do {                                                                                          \
    if (one_tile) {                                                                             \
      if (se3) { if (res) MVSN_CONV_LAUNCH(NPTV, 1, 3, 3, 1, 3, 1, true); else MVSN_CONV_LAUNCH(NPTV, 1, 3, 3, 1, 3, 1, false); } \
      else     { if (res) MVSN_CONV_LAUNCH(NPTV, 1, 3, 3, 1, 6, 1, true); else MVSN_CONV_LAUNCH(NPTV, 1, 3, 3, 1, 6, 1, false); } \
    } else {                                                                                    \
      if (se3) { if (res) MVSN_CONV_LAUNCH(NPTV, 1, 3, 3, 1, 3, 2, true); else MVSN_CONV_LAUNCH(NPTV, 1, 3, 3, 1, 3, 2, false); } \
      else     { if (res) MVSN_CONV_LAUNCH(NPTV, 1, 3, 3, 1, 6, 2, true); else MVSN_CONV_LAUNCH(NPTV, 1, 3, 3, 1, 6, 2, false); } \
    }                                                                                           \
  } while (0)
  if (g.kd == 3 && g.v4) {               // 3-D 3x3x3 staged as 16-byte groups: 4*10*10 = 400 groups -> 2 per thread
    if (one_tile) MVSN_CONV_LAUNCH(8, 3, 3, 3, 1, 2, 1, false, true); else MVSN_CONV_LAUNCH(8, 3, 3, 3, 1, 2, 2, false, true);
  } else if (g.kd == 3) {                // 3-D 3x3x3, TZ=2 TY=8 -> NPT 8, SE 6
    if (one_tile) MVSN_CONV_LAUNCH(8, 3, 3, 3, 1, 6, 1, false); else MVSN_CONV_LAUNCH(8, 3, 3, 3, 1, 6, 2, false);
  } else if (g.kh == 5 && g.v4) {        // 2-D 5x5 stride 2 staged as 16-byte groups: 19 * 18 = 342 groups -> 2 per thread
    MVSN_CONV_LAUNCH(4, 1, 5, 5, 2, 2, 2, false, true);
  } else if (g.kh == 5) {                // 2-D 5x5 stride 2, TY=8 -> NPT 4
    MVSN_CONV_LAUNCH(4, 1, 5, 5, 2, 6, 2, false);
  } else if (g.TY == 16) {               // 2-D 3x3, NPT 8
    MVSN_CONV_2D(8);
  } else {                               // 2-D 3x3, NPT 4
    MVSN_CONV_2D(4);
  }
#undef MVSN_CONV_2D
#undef MVSN_CONV_LAUNCH
  return check_launch("mvsn_conv_forward");
}

#ifdef MVSN_DMA_STAMPS
extern "C" int mvsn_debug_set_dma_stamps(void *buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(mvsn::g_dma_stamps), &buf, sizeof(buf));
}
#endif

extern "C" int mvsn_groupnorm_finalize(const float *partials, int n, int tiles, float *stats, mvsn_stream_t stream) {
  MVSN_REQUIRE(partials && stats && n > 0 && tiles > 0, MVSN_E_BADARG, "mvsn_groupnorm_finalize: bad argument");
  hipLaunchKernelGGL(mvsn::gn_finalize_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, partials, tiles, stats);
  return mvsn::check_launch("mvsn_groupnorm_finalize");
}

extern "C" size_t mvsn_groupnorm_finalize_split_workspace_bytes(int n, int tiles) {
  if (n <= 0 || tiles <= 0) return 0;
  const int K = mvsn::gn_split_slices(tiles);
  return K == 1 ? 0 : (size_t)n * K * 12 * sizeof(double);
}

extern "C" int mvsn_groupnorm_finalize_split(const float *partials, int n, int tiles, float *stats, void *workspace,
                                             size_t workspace_bytes, mvsn_stream_t stream) {
  MVSN_REQUIRE(partials && stats && n > 0 && tiles > 0, MVSN_E_BADARG, "mvsn_groupnorm_finalize_split: bad argument");
  const int K = mvsn::gn_split_slices(tiles);
  if (K == 1) return mvsn_groupnorm_finalize(partials, n, tiles, stats, stream);
  const size_t need = mvsn_groupnorm_finalize_split_workspace_bytes(n, tiles);
  MVSN_REQUIRE(workspace && workspace_bytes >= need && ((size_t)workspace & 7) == 0, MVSN_E_WORKSPACE,
               "mvsn_groupnorm_finalize_split: 8-byte aligned workspace of %zu bytes required", need);
  MVSN_REQUIRE(n <= 65535, MVSN_E_TOOLARGE, "mvsn_groupnorm_finalize_split: batch too large for one launch");
  const int per = (tiles + K - 1) / K;
  hipLaunchKernelGGL(mvsn::gn_partial_kernel, dim3(K, n), dim3(256), 0, (hipStream_t)stream, partials, tiles, per, K,
                     (double *)workspace);
  hipLaunchKernelGGL(mvsn::gn_combine_kernel, dim3((n * 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const double *)workspace, n, K, stats);
  return mvsn::check_launch("mvsn_groupnorm_finalize_split");
}

// stat_tiles == 0: `stats` is the finalised (N,4,2) array; > 0: the producer's records (N, stat_tiles, 4, 3), finalised by
// every workgroup of the pass itself (gn_stats_here)
static int gn_apply_launch(const char *what, const float *x, const float *stats, int stat_tiles, const float *gamma,
                           const float *beta, const float *residual, const float *r_stats, const float *r_gamma,
                           const float *r_beta, int n, long spatial, float *out, mvsn_stream_t stream) {
  long per = (spatial + 4095) / 4096;   // 4 float4 per thread per pass
  int gx = (int)(per < 1 ? 1 : (per > 32 ? 32 : per));
  const long sstride = stat_tiles > 0 ? (long)stat_tiles * 12 : 8;
  for (int n0 = 0; n0 < n; n0 += mvsn::GN_APPLY_MAX_N) {   // grid.y carries (sample, channel): 2047 samples per launch
    const int nn = n - n0 < mvsn::GN_APPLY_MAX_N ? n - n0 : mvsn::GN_APPLY_MAX_N;
    const long off = (long)n0 * 32 * spatial;
    if (r_stats)
      hipLaunchKernelGGL(mvsn::gn_apply_kernel<true>, dim3(gx, nn * 32), dim3(256), 0, (hipStream_t)stream, x + off,
                         stats + (long)n0 * sstride, gamma, beta, residual + off, r_stats + (long)n0 * 8, r_gamma, r_beta,
                         spatial, stat_tiles, out + off);
    else
      hipLaunchKernelGGL(mvsn::gn_apply_kernel<false>, dim3(gx, nn * 32), dim3(256), 0, (hipStream_t)stream, x + off,
                         stats + (long)n0 * sstride, gamma, beta, residual ? residual + off : residual,
                         (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, spatial, stat_tiles,
                         out + off);
  }
  return mvsn::check_launch(what);
}

extern "C" int mvsn_groupnorm_lrelu_apply(const float *x, const float *stats, const float *gamma, const float *beta,
                                          const float *residual, int n, long spatial, float *out,
                                          mvsn_stream_t stream) {
  MVSN_REQUIRE(x && stats && gamma && beta && out && n > 0 && spatial > 0, MVSN_E_BADARG,
               "mvsn_groupnorm_lrelu_apply: bad argument");
  return gn_apply_launch("mvsn_groupnorm_lrelu_apply", x, stats, 0, gamma, beta, residual, nullptr, nullptr, nullptr, n,
                         spatial, out, stream);
}

extern "C" int mvsn_groupnorm_lrelu_add2(const float *x, const float *stats, const float *gamma, const float *beta,
                                         const float *r, const float *r_stats, const float *r_gamma,
                                         const float *r_beta, int n, long spatial, float *out, mvsn_stream_t stream) {
  MVSN_REQUIRE(x && stats && gamma && beta && r && r_stats && r_gamma && r_beta && out && n > 0 && spatial > 0,
               MVSN_E_BADARG, "mvsn_groupnorm_lrelu_add2: bad argument");
  return gn_apply_launch("mvsn_groupnorm_lrelu_add2", x, stats, 0, gamma, beta, r, r_stats, r_gamma, r_beta, n, spatial,
                         out, stream);
}

extern "C" int mvsn_groupnorm_lrelu_apply_records(const float *x, const float *records, int tiles, const float *gamma,
                                                  const float *beta, const float *residual, const float *r_stats,
                                                  const float *r_gamma, const float *r_beta, int n, long spatial,
                                                  float *out, mvsn_stream_t stream) {
  MVSN_REQUIRE(x && records && tiles > 0 && gamma && beta && out && n > 0 && spatial > 0, MVSN_E_BADARG,
               "mvsn_groupnorm_lrelu_apply_records: bad argument");
  MVSN_REQUIRE(!r_stats || (residual && r_gamma && r_beta), MVSN_E_BADARG,
               "mvsn_groupnorm_lrelu_apply_records: a raw residual needs its statistics, gamma and beta");
  return gn_apply_launch("mvsn_groupnorm_lrelu_apply_records", x, records, tiles, gamma, beta, residual, r_stats, r_gamma,
                         r_beta, n, spatial, out, stream);
}

extern "C" int mvsn_conv_forward_carry(const mvsn_conv_desc *desc, const float *in, const float *weight_packed,
                                       const float *bias, const float *in_stats, const float *in_gamma,
                                       const float *in_beta, float *out, float *out_partials,
                                       const mvsn_apply_job *job, int *carried, mvsn_stream_t stream) {
  using namespace mvsn;
  if (carried) *carried = 0;
  if (!job)
    return mvsn_conv_forward(desc, in, weight_packed, bias, in_stats, in_gamma, in_beta, nullptr, nullptr, out,
                             out_partials, stream);
  MVSN_REQUIRE(job->x && job->stats && job->gamma && job->beta && job->out && job->n > 0 && job->spatial > 0,
               MVSN_E_BADARG, "mvsn_conv_forward_carry: bad job");
  MVSN_REQUIRE(in && weight_packed && out, MVSN_E_BADARG, "mvsn_conv_forward_carry: null pointer");
  WinoGeom wg;
  if (desc && desc->precision == MVSN_CONV_FP32_WINO && wino_geom(desc, &wg) && wino_can_carry(wg, job) &&
      (!in_stats || (in_gamma && in_beta)) && wg.n <= 65535) {
    if (carried) *carried = 1;
    return wino_launch(wg, in, weight_packed, bias, in_stats, in_gamma, in_beta, out, out_partials,
                       (hipStream_t)stream, nullptr, job);
  }
  const int rc = job->r_stats
                     ? mvsn_groupnorm_lrelu_add2(job->x, job->stats, job->gamma, job->beta, job->residual, job->r_stats,
                                                 job->r_gamma, job->r_beta, job->n, job->spatial, job->out, stream)
                     : mvsn_groupnorm_lrelu_apply(job->x, job->stats, job->gamma, job->beta, job->residual, job->n,
                                                  job->spatial, job->out, stream);
  if (rc) return rc;
  return mvsn_conv_forward(desc, in, weight_packed, bias, in_stats, in_gamma, in_beta, nullptr, nullptr, out,
                           out_partials, stream);
}

extern "C" int mvsn_selftest_mfma(mvsn_stream_t stream) {
  int *dbad = nullptr;
  hipError_t e = hipMalloc(&dbad, sizeof(int));
  if (e != hipSuccess) return (int)e;
  (void)hipMemsetAsync(dbad, 0, sizeof(int), (hipStream_t)stream);
  hipLaunchKernelGGL(mvsn::mfma_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dbad);
  (void)mvsn::bf16_selftest((hipStream_t)stream, dbad);
  int bad = -1;
  (void)hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
  (void)hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(dbad);
  if (bad != 0) {
    mvsn::set_error("mvsn_selftest_mfma: %d fragment elements differ from the scalar product", bad);
    return MVSN_E_BADARG - 100;
  }
  return 0;
}
