// 32 -> 32 channel 3x3 / 3x3x3 convolution on the bf16 matrix cores with fp32-equivalent accuracy
// ("3 x bf16 split"): every fp32 operand is split into hi = bf16(v) and
// lo = bf16(v - hi) (together 16 mantissa bits) and each product is taken as
//       a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (a_lo*b_lo ~ 2^-16 |a b| is dropped)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: three instructions of 16 (SIMD) cycles cover
// K = 32 input channels, against eight 32-cycle fp32 instructions -- the fp32 MFMA rate is what bounds
// the regulariser and refiner layers.
//
// Plane streaming: a workgroup owns TZO x TY x 32 outputs x 32 couts and walks the TZO+KD-1 input
// planes it needs once; the haloed plane tile sits in LDS channel-minor, one 160-byte record per
// position = [32 x bf16 hi | 32 x bf16 lo | 32 pad] (record stride 40 dwords: the sixteen 16-byte
// B-fragment reads of a lane group land on distinct banks).  Lane l of an MFMA supplies
// A[i = l&15][k = 8*(l>>4) + 0..7] (weights, read as two 16-byte global loads per part, L1/L2 resident)
// and B[k][j = l&15] (one 16-byte LDS read per part).  Input planes that fall outside the volume are
// skipped (their taps contribute zeros), so edge tiles do less work.
#include <stdlib.h>

#include "mvsn_common.h"
#include "mvsn_conv_bf16x3.h"

namespace mvsn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

constexpr int BX_THREADS = 256;
constexpr int BX_REC = 40;  // dwords per position record: 160 B = 10 sixteen-byte slots -> the ds_read_b128 lane groups are conflict-free

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

// KD   1 (2-D) or 3 (3-D);  TZO output planes, TY output rows (x 32 columns) per tile
// TPW  consecutive tiles per workgroup (2-D only; a 3-D workgroup walks TZO+2 input planes instead)
// NIT  staging items per thread per stage = ceil(HY*HX*4 / 256); an item = 8 channels of one position
//
// Stages (one haloed input plane tile each) are software-pipelined: the raw fp32 values of stage s+1
// are fetched into registers while the MFMAs of stage s run, then converted to hi/lo bf16 records and
// written to the single LDS plane between two barriers.
// DIL  dilation as a compile-time constant: the tap offsets become ds_read immediates (no address VALU)
// NPROD  3: a*b ~= ah*bh + ah*bl + al*bh (fp32-equivalent split); 1: ah*bh only = plain bf16 operands with fp32
//        accumulation (MVSN_CONV_BF16: BASELINE config 5's speed tier, outside the 1e-3 parity contract)
// IN16 / OUT16  (bf16 STORAGE, mvsn_conv_forward_bf16_storage: BASELINE config 5's "bf16 features") the input / output
//        tensor holds bf16 instead of fp32 -- the same element order, half the bytes: the fetch widens 2-byte loads (a bf16
//        IS the hi operand, exactly), the epilogue rounds the biased fp32 accumulators to bf16 (RNE) as it stores them.
//        The GroupNorm partials are formed from the unrounded accumulators, statistics stay fp32.  NPROD = 1 only.
template <int KD, int TZO, int TY, int TPW, int NIT, int MODE, int DIL, int NPROD, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(BX_THREADS, 2) void conv_bf16x3_kernel(Bf16x3Geom g, const float *__restrict__ in,
                                                                    const uintx4 *__restrict__ wpk,
                                                                    const float *__restrict__ bias,
                                                                    const float *__restrict__ in_stats,
                                                                    const float *__restrict__ in_gamma,
                                                                    const float *__restrict__ in_beta,
                                                                    float *__restrict__ out,
                                                                    float *__restrict__ out_partials,
                                                                    unsigned long long *dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned int plane[];  // HY*HX records of BX_REC dwords
  int dbg_i = 0;
#define BX_STAMP()                                                                                     \
  do {                                                                                                 \
    if (dbg && blockIdx.x == 3 && blockIdx.y == 0 && threadIdx.x == 0 && dbg_i < 60) dbg[dbg_i++] = __builtin_readcyclecounter(); \
  } while (0)
  // Wave w owns cout tile (w & 1) and half (w >> 1) of the tile's pixel tiles: a weight fragment is
  // then reused for TY pixel tiles x 3 products, which halves the L1 traffic of the weight stream (the
  // first bottleneck of this kernel) at the price of more B-fragment reads from LDS, where there is room.
  constexpr int NPT = TY;  // pixel tiles (16 columns) per wave per output plane
  constexpr int NSTAGE = KD == 3 ? TZO + 2 : TPW;
  constexpr int HX = 32 + 2 * DIL, HY = TY + 2 * DIL;   // == g.HX, g.HY
  constexpr int npos = HY * HX;
  float *scsh = reinterpret_cast<float *>(plane + (size_t)npos * BX_REC);  // 64
  float *red = scsh + 64;                                                  // 16

  BX_STAMP();   // kernel entry
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tw = wave & 1, half = wave >> 1;
  const int n = blockIdx.y;
  const size_t in_plane = (size_t)g.H * g.W, in_chan = (size_t)g.D * in_plane;
  static_assert(!(IN16 || OUT16) || NPROD == 1, "bf16 storage: plain bf16 operands");
  const float *inn = in + (size_t)n * 32 * in_chan;                       // (fp32 storage)
  float *outn = out + (size_t)n * 32 * in_chan;
  const unsigned short *inn16 = reinterpret_cast<const unsigned short *>(in) + (size_t)n * 32 * in_chan;   // (bf16 storage)
  unsigned short *outn16 = reinterpret_cast<unsigned short *>(out) + (size_t)n * 32 * in_chan;

  if (MODE == 1 && tid < 32) {
    const int grp = tid >> 3;
    const float mean = in_stats[((size_t)n * 4 + grp) * 2 + 0];
    const float rstd = in_stats[((size_t)n * 4 + grp) * 2 + 1];
    const float sc = rstd * in_gamma[tid];
    scsh[tid] = sc;
    scsh[32 + tid] = in_beta[tid] - mean * sc;
  }

  // stage s -> (tile, input plane): 3-D: one tile, planes z0-1 .. z0+TZO; 2-D: TPW tiles, plane 0
  auto tile_of = [&](int s) { return KD == 3 ? (int)blockIdx.x : (int)blockIdx.x * TPW + s; };
  auto origin = [&](int tile, int &z0, int &y0, int &x0) {
    const int txi = tile % g.ntx;
    const int rest = tile / g.ntx;
    const int tyi = rest % g.nty;
    z0 = (rest / g.nty) * TZO, y0 = tyi * TY, x0 = txi * 32;
  };
  auto stage_valid = [&](int s) {
    if (KD == 3) {
      int z0, y0, x0;
      origin(tile_of(s), z0, y0, x0);
      const int gz = z0 - 1 + s;
      return gz >= 0 && gz < g.D;
    }
    return tile_of(s) < g.tiles;
  };

  // this lane's pixels (column j = lane & 15 of each of the wave's pixel tiles) as record offsets
  int prec[NPT];
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int pt = half * NPT + j;
    const int yy = pt >> 1, xx = (pt & 1) * 16 + (lane & 15);
    prec[j] = (yy * HX + xx) * BX_REC + (lane >> 4) * 4;
  }
  // staging items of this thread: position and channel group are the same for every stage
  int ipos[NIT];
  short iy[NIT], ix[NIT], igrp[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int it = tid + k * BX_THREADS;
    const int grp = it / npos, p = it - grp * npos;
    ipos[k] = it < npos * 4 ? p : -1;
    iy[k] = (short)(p / HX);
    ix[k] = (short)(p - (p / HX) * HX);
    igrp[k] = (short)grp;
  }

  // bf16 storage: an item is one ALIGNED 4-byte word of 8 channels = two x-adjacent positions per load (2-byte loads, one
  // position each, made the bf16-input layer 1.7x slower than the fp32-input one, and 4-byte loads at odd element
  // offsets 2.3x: sub-word and misaligned accesses are the slow path of the vector memory pipe).  The tile's columns
  // start at the odd element x0 - DIL (DIL = 1, x0 a multiple of 32, W even), so word j of a row holds the positions
  // ix = 2j - 1 and 2j: HX / 2 + 1 words per row, the first and the last contributing one position each.
  constexpr int HXP = HX / 2 + 1, NPAIRS = HY * HXP * 4, NITP = (NPAIRS + BX_THREADS - 1) / BX_THREADS;
  static_assert(!IN16 || (HX % 2 == 0 && DIL == 1), "aligned words of two positions");
  int ppos[IN16 ? NITP : 1];      // record index of the word's SECOND position (ix = 2j); the first is the record before
  short py[IN16 ? NITP : 1], px[IN16 ? NITP : 1], pgrp[IN16 ? NITP : 1];
  if constexpr (IN16) {
#pragma unroll
    for (int k = 0; k < NITP; ++k) {
      const int it = tid + k * BX_THREADS;
      const int grp = it / (HY * HXP), pp = it - grp * (HY * HXP);
      const int yy = pp / HXP, j = pp - yy * HXP;
      ppos[k] = it < NPAIRS ? yy * HX + 2 * j : -1;
      py[k] = (short)yy, px[k] = (short)(2 * j - 1), pgrp[k] = (short)grp;   // px: ix of the first position (-1 .. HX - 1)
    }
  }
  unsigned rawp[IN16 ? NITP : 1][8];
  float raw[IN16 ? 1 : NIT][8];
  unsigned okmask = 0;
  auto fetch = [&](int s) {
    int z0, y0, x0;
    origin(tile_of(s), z0, y0, x0);
    const int gz = KD == 3 ? z0 - 1 + s : 0;
    okmask = 0;
    if constexpr (IN16) {
#pragma unroll
      for (int k = 0; k < NITP; ++k) {
        const int gy = y0 - DIL + py[k], gx = x0 - DIL + px[k];   // gx even: the word (gx, gx + 1) is 4-byte aligned
        const bool rowok = ppos[k] >= 0 && gy >= 0 && gy < g.H;
        // a position counts if it belongs to the tile (ix in 0 .. HX - 1) and to the image
        const bool ok0 = rowok && px[k] >= 0 && gx >= 0 && gx < g.W;
        const bool ok1 = rowok && px[k] + 1 < HX && gx + 1 >= 0 && gx + 1 < g.W;
        okmask |= ((ok0 ? 1u : 0u) | (ok1 ? 2u : 0u)) << (2 * k);
        // (W even and gx even: when either position is inside the image, the whole word is)
        const bool any = ok0 || ok1;
        const unsigned int *src = reinterpret_cast<const unsigned int *>(
            inn16 + (size_t)(pgrp[k] * 8) * in_chan + (size_t)gz * in_plane + (size_t)(any ? gy : 0) * g.W + (any ? gx : 0));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned w = any ? src[((size_t)e * in_chan) >> 1] : 0u;
          rawp[k][e] = (ok0 ? w & 0xffffu : 0u) | (ok1 ? w & 0xffff0000u : 0u);
        }
      }
    } else {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int gy = y0 - DIL + iy[k], gx = x0 - DIL + ix[k];
      const bool ok = ipos[k] >= 0 && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
      if (ok) okmask |= 1u << k;
      const size_t soff = (size_t)(igrp[k] * 8) * in_chan + (size_t)gz * in_plane + (size_t)(ok ? gy : 0) * g.W + (ok ? gx : 0);
      if constexpr (!IN16) {
        const float *src = inn + soff;
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[k][e] = ok ? src[(size_t)e * in_chan] : 0.0f;
      }
    }
    }
  };
  auto commit = [&]() {
    if constexpr (IN16) {
#pragma unroll
      for (int k = 0; k < NITP; ++k) {
        if (ppos[k] < 0) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uintx4 hi;
          if (MODE == 1 && ((okmask >> (2 * k + h)) & 1u)) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float x = __uint_as_float(h ? rawp[k][e] & 0xffff0000u : rawp[k][e] << 16);
              v[e] = lrelu02(x * scsh[pgrp[k] * 8 + e] + scsh[32 + pgrp[k] * 8 + e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) hi[e] = pack_bf16_pair(v[2 * e], v[2 * e + 1]);
          } else {   // the stored bf16 values ARE the operands
#pragma unroll
            for (int e = 0; e < 4; ++e)
              hi[e] = h ? (rawp[k][2 * e] >> 16) | (rawp[k][2 * e + 1] & 0xffff0000u)
                        : (rawp[k][2 * e] & 0xffffu) | (rawp[k][2 * e + 1] << 16);
          }
          // (the word's first position is the record before ppos; the row's first / last word have only one in the tile)
          if (h ? px[k] + 1 < HX : px[k] >= 0)
            *reinterpret_cast<uintx4 *>(plane + (size_t)(ppos[k] - 1 + h) * BX_REC + pgrp[k] * 4) = hi;
        }
      }
    } else {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      if (ipos[k] < 0) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = raw[k][e];
      if (MODE == 1 && ((okmask >> k) & 1u)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = lrelu02(v[e] * scsh[igrp[k] * 8 + e] + scsh[32 + igrp[k] * 8 + e]);
      }
      uintx4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned int h = pack_bf16_pair(v[2 * e], v[2 * e + 1]);
        hi[e] = h;
        lo[e] = pack_bf16_pair(v[2 * e] - bf16_lo_to_float(h), v[2 * e + 1] - bf16_hi_to_float(h));
      }
      unsigned int *rec = plane + (size_t)ipos[k] * BX_REC + igrp[k] * 4;
      *reinterpret_cast<uintx4 *>(rec) = hi;
      *reinterpret_cast<uintx4 *>(rec + 16) = lo;
    }
    }
  };

  floatx4 acc[TZO][NPT];
  auto clear_acc = [&]() {
#pragma unroll
    for (int zo = 0; zo < TZO; ++zo)
#pragma unroll
      for (int j = 0; j < NPT; ++j) acc[zo][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  };

  // Bias of this lane's four channels, read once: a global load inside the per-tile epilogue would make
  // its s_waitcnt (vmcnt retires in order) wait for the next stage's prefetch and the previous tile's stores.
  const float bias1 = bias ? bias[tw * 16 + (lane & 15)] : 0.0f;   // this lane's cout (D = pixels x couts)

  // bias, store, GroupNorm partials of one finished tile.  This wave holds channels tw*16 + cbase + r,
  // i.e. GroupNorm groups 2*tw + (lane >> 5); the tile's positions are split over the two `half` waves.
  // The MFMAs run with A = activation records (16 pixels x 32 cins) and B = weights (32 cins x 16 couts), so
  // D = pixels x couts: lane l holds, for cout tw*16 + (l & 15), the four consecutive pixels 4*(l>>4) + r of each
  // pixel tile -- one 16-byte store per accumulator, no transpose.
  auto epilogue = [&](int tile) {
    int z0, y0, x0;
    origin(tile, z0, y0, x0);
    const int c = tw * 16 + (lane & 15);
    float s = 0.f;
    int cnt = 0;
    int ovalid[TZO][NPT];
#pragma unroll
    for (int zo = 0; zo < TZO; ++zo)
#pragma unroll
      for (int j = 0; j < NPT; ++j) {
        const int pt = half * NPT + j;
        const int oz = z0 + zo, oy = y0 + (pt >> 1), ox4 = x0 + (pt & 1) * 16 + (lane >> 4) * 4;
        const int nv = (oz < g.D && oy < g.H && ox4 < g.W) ? min(4, g.W - ox4) : 0;
        ovalid[zo][j] = nv;
        cnt += nv;
        const size_t pos = (size_t)oz * in_plane + (size_t)oy * g.W + ox4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[zo][j][r] + bias1;
          acc[zo][j][r] = r < nv ? v : 0.0f;
          if (r < nv) s += v;
        }
        if constexpr (OUT16) {
          if ((g.W & 3) == 0) {   // four pixels = 8 bytes, aligned
            if (nv == 4) {
              typedef unsigned int bx_u2 __attribute__((ext_vector_type(2)));
              bx_u2 pk;
              pk[0] = pack_bf16_pair(acc[zo][j][0], acc[zo][j][1]), pk[1] = pack_bf16_pair(acc[zo][j][2], acc[zo][j][3]);
              *reinterpret_cast<bx_u2 *>(outn16 + (size_t)c * in_chan + pos) = pk;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (r < nv) outn16[(size_t)c * in_chan + pos + r] = (unsigned short)(pack_bf16_pair(acc[zo][j][r], 0.0f) & 0xffffu);
          }
        } else if ((g.W & 3) == 0) {   // the four pixels are all inside or all outside, and 16-byte aligned
          if (nv == 4) *reinterpret_cast<floatx4 *>(outn + (size_t)c * in_chan + pos) = acc[zo][j];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nv) outn[(size_t)c * in_chan + pos + r] = acc[zo][j][r];
        }
      }
    if (out_partials == nullptr) return;
    // Per-wave GroupNorm partials (count, mean, M2), reduced with shuffles only -- no LDS, no barrier.  The
    // group of a lane's channel is tw*2 + ((l>>3)&1); record (tile, wave) holds this wave's two groups, the
    // other two are written with count 0.
    auto pixel_sum = [&](float v) {   // over the lanes holding the same cout: bits 4, 5
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      return v;
    };
    auto group_sum = [&](float v) {   // over the 8 couts of a group (bits 0-2) and all pixels
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      return pixel_sum(v);
    };
    const int hi = (lane >> 3) & 1;
    const float npos_out = pixel_sum((float)cnt) * 8.0f;
    s = group_sum(s);
    const float m = npos_out > 0.0f ? s / npos_out : 0.0f;
    float q = 0.f;
#pragma unroll
    for (int zo = 0; zo < TZO; ++zo)
#pragma unroll
      for (int j = 0; j < NPT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < ovalid[zo][j]) {
            const float dv = acc[zo][j][r] - m;
            q += dv * dv;
          }
    q = group_sum(q);
    if ((lane & 0x37) == 0) {   // lanes 0 and 8
      float *rec = out_partials + (((size_t)n * g.tiles + tile) * 4 + wave) * 12;
      float *mine = rec + (tw * 2 + hi) * 3, *other = rec + ((1 - tw) * 2 + hi) * 3;
      mine[0] = npos_out, mine[1] = m, mine[2] = q;
      other[0] = 0.0f, other[1] = 0.0f, other[2] = 0.0f;
    }
  };

  bf16x8 wh[9], wl[9];
  int slab_tz = -1;
  auto load_slab = [&](int tz) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const uintx4 *wt = wpk + (size_t)(tz * 9 + k) * 256 + tw * 128 + lane;  // [tap][t][part][lane]
      wh[k] = __builtin_bit_cast(bf16x8, wt[0]);
      if (NPROD == 3) wl[k] = __builtin_bit_cast(bf16x8, wt[64]);
    }
    slab_tz = tz;
  };
  if (KD == 1) load_slab(0);

  clear_acc();
  int next = 0;
  while (next < NSTAGE && !stage_valid(next)) ++next;  // uniform
  if (next < NSTAGE) fetch(next);
#pragma nounroll
  while (next < NSTAGE) {
    const int s = next;
    BX_STAMP();
    __syncthreads();  // the previous stage's MFMAs are done with the LDS plane (and scsh is visible)
    BX_STAMP();
    commit();
    BX_STAMP();
    __syncthreads();
    BX_STAMP();
    ++next;
    while (next < NSTAGE && !stage_valid(next)) ++next;
    bool prefetched = false;
    if (KD == 1) {
      if (next < NSTAGE) fetch(next);  // in flight behind this stage's MFMAs
      prefetched = true;
    }

    // One z-tap slab of weight fragments (9 taps x hi/lo of this wave's cout tile = 72 VGPRs) is read
    // ahead of the MFMAs that use it.  vmcnt retires in order, so a weight load issued after the
    // prefetch of the next stage would wait for that prefetch: the 2-D slab is therefore loaded once
    // per workgroup (before any prefetch), and the first 3-D slab of a stage before the stage's prefetch.
#pragma unroll
    for (int zo = 0; zo < TZO; ++zo) {
      const int tz = KD == 3 ? s - zo : 0;  // z-tap through which output plane zo sees this input plane
      if (tz < 0 || tz >= KD) continue;     // uniform
      if (KD == 3) {
        if (tz != slab_tz) load_slab(tz);
        if (!prefetched) {
          if (next < NSTAGE) fetch(next);    // in flight behind this stage's MFMAs
          prefetched = true;
        }
      }
      // 9 taps x NPT pixel tiles, flattened and taken two steps at a time: the three split products of a
      // step accumulate into the same registers (a dependent MFMA chain), so the products of two steps
      // are interleaved, and the B fragments (hi, lo: two 16-byte LDS reads per step) of the next pair
      // are already in flight -- three 16-cycle MFMAs do not cover an LDS round trip.
      constexpr int NPAIR = 9 * NPT / 2;
      uintx4 fh[2][2], fl[2][2];
      auto frag = [&](int step, uintx4 &h, uintx4 &l) {
        const int tap = step / NPT, j = step - tap * NPT;
        const int toff = ((tap / 3) * DIL * HX + (tap % 3) * DIL) * BX_REC;   // compile-time: a ds_read immediate
        const unsigned int *rec = plane + prec[j] + toff;
        h = *reinterpret_cast<const uintx4 *>(rec);
        if (NPROD == 3) l = *reinterpret_cast<const uintx4 *>(rec + 16);
      };
      frag(0, fh[0][0], fl[0][0]);
      frag(1, fh[0][1], fl[0][1]);
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr) {
        const int cur = pr & 1;
        if (pr + 1 < NPAIR) {
          frag(2 * pr + 2, fh[cur ^ 1][0], fl[cur ^ 1][0]);
          frag(2 * pr + 3, fh[cur ^ 1][1], fl[cur ^ 1][1]);
        }
        const int tap = (2 * pr) / NPT, j0 = 2 * pr - tap * NPT, j1 = j0 + 1;   // NPT is even: same tap
        const bf16x8 ah = wh[tap], al = wl[tap];
        const bf16x8 bh0 = __builtin_bit_cast(bf16x8, fh[cur][0]), bl0 = __builtin_bit_cast(bf16x8, fl[cur][0]);
        const bf16x8 bh1 = __builtin_bit_cast(bf16x8, fh[cur][1]), bl1 = __builtin_bit_cast(bf16x8, fl[cur][1]);
        if constexpr (NPROD == 3) {
          acc[zo][j0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh0, al, acc[zo][j0], 0, 0, 0);
          acc[zo][j1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh1, al, acc[zo][j1], 0, 0, 0);
          acc[zo][j0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl0, ah, acc[zo][j0], 0, 0, 0);
          acc[zo][j1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl1, ah, acc[zo][j1], 0, 0, 0);
        }
        acc[zo][j0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh0, ah, acc[zo][j0], 0, 0, 0);   // D = pixels x couts
        acc[zo][j1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh1, ah, acc[zo][j1], 0, 0, 0);
      }
    }
    if (KD == 3 && !prefetched && next < NSTAGE) fetch(next);
    BX_STAMP();
    if (KD == 1) {  // every 2-D stage is a finished tile
      epilogue(tile_of(s));
      clear_acc();
    }
  }
  BX_STAMP();   // before the (3-D) epilogue
  if (KD == 3) epilogue(tile_of(0));
  BX_STAMP();   // kernel exit
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
  g->nprod = d->precision == MVSN_CONV_BF16 ? 1 : 3;
  g->tzo = d->kd == 3 ? 4 : 1;
  g->ty = 4;
  g->HY = g->ty + 2 * g->dil;
  g->HX = 32 + 2 * g->dil;
  g->ntz = (g->D + g->tzo - 1) / g->tzo;
  g->nty = (g->H + g->ty - 1) / g->ty;
  g->ntx = (g->W + 31) / 32;
  g->tiles = g->ntz * g->nty * g->ntx;
  g->lds_bytes = ((size_t)g->HY * g->HX * BX_REC + 64 + 16) * 4;
  if (d->dilation > 2) return false;   // (measured) dilation 4 needs 92 KB of LDS per workgroup and loses to fp32 MFMA
  return g->lds_bytes <= 160 * 1024 && (d->kd == 1 || d->dilation == 1);
}

int bf16x3_pack(const mvsn_conv_desc *d, const float *weight, void *packed, hipStream_t stream) {
  const int ntaps = d->kd * 9;
  const int total = ntaps * 1024;
  hipLaunchKernelGGL(conv_bf16x3_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, weight, ntaps,
                     (unsigned short *)packed);
  return check_launch("mvsn_conv_pack_weights(bf16x3)");
}

#ifdef MVSN_BX_STAMPS
static unsigned long long *g_bx_stamps = nullptr;
#endif

// bf16 storage (volume form, plain bf16 operands): `in` / `out` hold bf16 where in16 / out16 say so
int bf16_storage_launch(const Bf16x3Geom &g, const void *in, bool in16, const void *wpk, const float *bias,
                        const float *in_stats, const float *in_gamma, const float *in_beta, void *out, bool out16,
                        float *out_partials, hipStream_t stream) {
  if (g.kd != 3 || g.nprod != 1 || !(in16 || out16)) {
    set_error("mvsn_conv_forward_bf16_storage: 3x3x3 layers with plain bf16 operands and a bf16 input or output");
    return MVSN_E_BADARG;
  }
  const bool xf = in_stats != nullptr;
  const dim3 grid(g.tiles, g.n);
#define MVSN_BS_LAUNCH(M, I, O)                                                                                     \
  do {                                                                                                              \
    auto kern = conv_bf16x3_kernel<3, 4, 4, 1, 4, M, 1, 1, I, O>;                                                   \
    static LdsOptIn opt;                                                                                            \
    if (int rc = ensure_lds(opt, (const void *)kern, g.lds_bytes, "mvsn_conv_forward_bf16_storage")) return rc;    \
    hipLaunchKernelGGL(kern, grid, dim3(BX_THREADS), g.lds_bytes, stream, g, (const float *)in, (const uintx4 *)wpk, \
                       bias, in_stats, in_gamma, in_beta, (float *)out, out_partials, (unsigned long long *)nullptr); \
  } while (0)
#define MVSN_BS_IO(M)                                        \
  do {                                                       \
    if (in16 && out16) MVSN_BS_LAUNCH(M, true, true);        \
    else if (in16) MVSN_BS_LAUNCH(M, true, false);           \
    else MVSN_BS_LAUNCH(M, false, true);                     \
  } while (0)
  if (xf) MVSN_BS_IO(1); else MVSN_BS_IO(0);
#undef MVSN_BS_IO
#undef MVSN_BS_LAUNCH
  return check_launch("mvsn_conv_forward_bf16_storage");
}

int bf16x3_launch(const Bf16x3Geom &g, const float *in, const void *wpk, const float *bias, const float *in_stats,
                  const float *in_gamma, const float *in_beta, float *out, float *out_partials, hipStream_t stream) {
#define MVSN_BX_LAUNCH(...)                                                                                       \
  do {                                                                                                            \
    auto kern = conv_bf16x3_kernel<__VA_ARGS__>;                                                                  \
    static LdsOptIn opt;                                                                                          \
    if (int rc = ensure_lds(opt, (const void *)kern, g.lds_bytes, "mvsn_conv_forward(bf16x3)")) return rc;        \
    hipLaunchKernelGGL(kern, grid, dim3(BX_THREADS), g.lds_bytes, stream, g, in, (const uintx4 *)wpk, bias, in_stats, \
                       in_gamma, in_beta, out, out_partials, dbgp);                                               \
  } while (0)
  const bool xf = in_stats != nullptr;
  unsigned long long *dbgp = nullptr;
#ifdef MVSN_BX_STAMPS      // tuning builds only (tools/bx_phases.py)
  dbgp = g_bx_stamps;
#endif
  constexpr int TPW2D = 8;
  const dim3 grid(g.kd == 3 ? g.tiles : (g.tiles + TPW2D - 1) / TPW2D, g.n);
#define MVSN_BX_BOTH(...)                                                         \
  do {                                                                            \
    if (g.nprod == 3) MVSN_BX_LAUNCH(__VA_ARGS__, 3); else MVSN_BX_LAUNCH(__VA_ARGS__, 1); \
  } while (0)
  if (g.kd == 3) {          // TZO 4, TY 4: 6 x 34 positions -> 816 items -> 4 per thread
    if (xf) MVSN_BX_BOTH(3, 4, 4, 1, 4, 1, 1); else MVSN_BX_BOTH(3, 4, 4, 1, 4, 0, 1);
  } else if (g.dil == 1) {  // TY 4: 6 x 34 -> 816 items -> 4 per thread
    if (xf) MVSN_BX_BOTH(1, 1, 4, TPW2D, 4, 1, 1); else MVSN_BX_BOTH(1, 1, 4, TPW2D, 4, 0, 1);
  } else {                  // dilation 2: 8 x 36 -> 1152 items -> 5 per thread
    if (xf) MVSN_BX_BOTH(1, 1, 4, TPW2D, 5, 1, 2); else MVSN_BX_BOTH(1, 1, 4, TPW2D, 5, 0, 2);
  }
#undef MVSN_BX_BOTH
#undef MVSN_BX_LAUNCH
  return check_launch("mvsn_conv_forward(bf16x3)");
}

int bf16_selftest(hipStream_t stream, int *dbad) {
  hipLaunchKernelGGL(mfma_bf16_selftest_kernel, dim3(1), dim3(64), 0, stream, dbad);
  return check_launch("mvsn_selftest_mfma(bf16)");
}

}  // namespace mvsn

#ifdef MVSN_BX_STAMPS
extern "C" int mvsn_debug_set_bx_stamps(void *buf) {
  mvsn::g_bx_stamps = (unsigned long long *)buf;
  return 0;
}
#endif
