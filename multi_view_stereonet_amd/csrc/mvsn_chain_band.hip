// The fused incremental chain, BANDED Winograd form: one chain on FOUR workgroups (see include/mvsn_hip.h:
// mvsn_incremental_cost_volume, form MVSN_CHAIN_BANDED).
//
// The plane-resident forms (mvsn_chain_wino.hip, mvsn_chain.hip) give a chain to one workgroup: with one chain per CU
// in flight that fills the chip, but at batch 1 (the reference's own evaluation loop, test.py:38,197-200, and BASELINE
// config 3's one image per GPU) the 63 sequential steps of multi_view_stereonet.py:279-290 run on S of the 256 CUs.
// Here the 16x32 coarse plane is cut into G = 4 bands of 4 pixel rows; workgroup m of a chain owns band m for the whole
// recurrence (same arithmetic per output as the plane-resident Winograd kernel: F(2x2,3x3) on v_mfma_f32_16x16x4_f32,
// GroupNorm over the WHOLE plane, masks from the reference's fp32 expression order), and the four exchange, per step,
//
//   E1  the new feature band F_d (gather source of step d + 1: a band's incremental-homography gather reaches into
//       its neighbours' rows),
//   E2  after conv0: this band's GroupNorm partial sums + its two boundary rows of the raw conv output (the
//       neighbours normalise and activate the halo row themselves once they have every band's sums),
//   E3  after conv1: the same for the residual block,
//
// i.e. three hand-offs per step instead of five (sums and halo rows travel together; the halo rows of the moved
// features are re-gathered by both neighbours instead of exchanged).
//
// Hand-off protocol (cdna_hip_programming.md section 6 Guideline 16, form R2): every exchanged float is ONE naturally
// aligned 8-byte granule {tag = step, value} written by ONE agent-scope relaxed store (lowers to a write-through
// `global_store_dwordx2 sc1`) and read by agent-scope relaxed loads (`sc1`: served past the reader's L1) that are
// repeated until the tag matches.  The data is its own flag: no release fence (the 14 us per launch DESIGN 3.6
// measured for buffer_wbl2), no acquire, no dependence on which XCD a workgroup landed on.  The granule region is
// zeroed by a memset node ahead of every launch; tags count steps within the call (F_d carries d + 1, the conv
// hand-offs of step d carry d), so a granule is either stale (tag - 1: keep polling) or current.  A buffer can be
// single: a band cannot publish step d + 1's version of a hand-off before every reader has consumed step d's, because
// the GroupNorm sums of the hand-off in between need all four bands (the ordering argument is spelled out in DESIGN.md 3.1).
// Spins are bounded: on a time-out the workgroup records it in the status word and stops waiting (the host reports it).
//
// Work split inside a workgroup: 256 threads = 4 waves, ONE per SIMD (512 registers each); wave (pt, ct) owns patch
// row pt of the band (16 patches of 2x2 pixels) and cout tile ct.  Per k-step and transform-row half: 8 multiplies
// against the plane-resident kernel's 16 -- the input transform is now repeated by the two cout-tile waves -- so a
// layer costs 18 / 16 x (8 MFMA + ~21 VALU / LDS) instructions per wave instead of 18 / 16 x (32 + 23) on twice the
// waves per SIMD.
//
// LDS (floats):  U 18432 (one layer's transformed weights, LDS-DMA'd per layer exactly as in chain_wino_kernel)
//                sparams 224 | red 64 (the 4 x 4 waves' GroupNorm records of a hand-off) | range 16 | maskb 128
//                tab 192 x 8: per pixel of rows lo-1 .. hi+1 the bilinear footprint of its incremental-homography
//                      gather (4 weights, window offset, row step, y0, x0), computed ONCE per pixel a step ahead
//                act 36 x 224: layer input planes, 6 rows (band + one halo row either side) x 34, channel stride
//                      224 = 32 (mod 64) so that the B-fragment reads of neighbouring channels use different banks
//                win 32 x 376: gather window = feature rows lo-3 .. hi+4 of the previous plane (own band written by
//                      the epilogue, the others fetched from the neighbours' granules when the step's gather needs
//                      them; rows outside the image stay zero and double as the bilinear taps' zero halo)
#include "mvsn_chain.h"
#include "mvsn_common.h"

namespace mvsn {

constexpr int CB_G = 4, CB_ROWS = 16, CB_COLS = 32, CB_BR = CB_ROWS / CB_G, CB_P = CB_ROWS * CB_COLS, CB_RS = CB_COLS + 2;
constexpr int CB_THREADS = 256, CB_WAVES = 4;
constexpr int CB_CSA = 224;                                  // channel stride of the layer input planes
constexpr int CB_W = 3, CB_WSLOTS = CB_BR + 2 * CB_W + 1;    // gather window: rows lo-3 .. hi+4 (11 slots)
constexpr int CB_CSW = 376;                                  // channel stride of the window (11 * 34 = 374, padded)
constexpr int CB_RED = 64, CB_RANGE = 16, CB_MASK = CB_BR * CB_COLS;
constexpr int CB_TAB = (CB_BR + 2) * CB_COLS * 8;            // bilinear footprints of the step's gathers, 8 words per pixel
constexpr int CB_LDS_FLOATS = CW_U0_FLOATS + CH_SP_FLOATS + CB_RED + CB_RANGE + CB_MASK + CB_TAB + 36 * CB_CSA + 32 * CB_CSW;
constexpr float CB_GN_EPS = 1e-5f;

// granule workspace of one chain (u64 units)
constexpr size_t CB_FG = 0;                                             // [32 ch][16 rows][32 cols]
constexpr size_t CB_RG = CB_FG + 32 * CB_P;                             // [layer 2][band 4][side 2][32 ch][32 cols]
constexpr size_t CB_SG = CB_RG + 2 * CB_G * 2 * 32 * CB_COLS;           // [layer 2][band 4][wave 4][4]
constexpr size_t CB_CHAIN_U64 = CB_SG + 2 * CB_G * CB_WAVES * 4;
constexpr unsigned CB_SPIN_LIMIT = 1u << 21;

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef float float2v __attribute__((ext_vector_type(2)));
typedef int intx4 __attribute__((ext_vector_type(4)));
#define CB_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define CB_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define CB_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

__device__ __forceinline__ void cb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void cb_publish(gu64 *g, unsigned tag, float v) {
  __hip_atomic_store(g, ((u64)tag << 32) | (u64)__builtin_bit_cast(unsigned, v), CB_RLX_AGENT);
}

// Two neighbouring granules (16-byte aligned pair) in ONE write-through store: half the fabric writes of a publish.
// Each 8-byte half is still a self-contained {value, tag} granule for the reader (8-byte halves of a 16-byte sc1
// store are observed untorn on gfx950, MI355X_MICROARCH.md section "inter-workgroup visibility"; a torn PAIR is
// harmless, the reader checks each granule's own tag).
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cb_publish2(gu64 *g, unsigned tag, float v0, float v1) {
  uintx4 q;
  q[0] = __builtin_bit_cast(unsigned, v0), q[1] = tag, q[2] = __builtin_bit_cast(unsigned, v1), q[3] = tag;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(g), "v"(q) : "memory");
}

// Re-read this lane's N granules until every tag in the wave matches.  `addr(j)` = granule j of this lane; lanes
// with !active take no part.  Returns with v[] filled; after a time-out (recorded in *status) it gives up at once.
template <int N, class Addr>
__device__ __forceinline__ void cb_sweep(Addr addr, unsigned tag, bool active, float (&v)[N], bool &dead, gu32 *status) {
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
    if (active) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const u64 x = __hip_atomic_load(addr(j), CB_RLX_AGENT);
        v[j] = __builtin_bit_cast(float, (unsigned)x);
        ok &= (unsigned)(x >> 32) == tag;
      }
    }
    if (__all(ok) || dead) return;
    if (spins >= CB_SPIN_LIMIT) {
      dead = true;
      __hip_atomic_store(status, 1u, CB_RLX_AGENT);
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// one 3x3 layer for ONE cout tile: acc[xi] (+)= U_xi * V_xi over NC k-steps, then the output transform
// (wino_layer of mvsn_chain_wino.hip with the cout-tile dimension dealt to the waves)
template <int NC>
__device__ __forceinline__ void band_layer(const float *__restrict__ act, const float *__restrict__ U, int ct, int wb,
                                           int lane, float (&y)[4][4]) {
  const float *wbase = act + (lane >> 4) * CB_CSA + wb;
  const float *ub = U + ct * 1024 + lane * 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    floatx4 acc[8];
    float d[2][3][4];
    floatx4 u[2][2];
    auto fetch = [&](int buf, int c4) {
      const float *wp = wbase + c4 * 4 * CB_CSA + half * CB_RS;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float2 lo = *reinterpret_cast<const float2 *>(wp + i * CB_RS);
        const float2 hi = *reinterpret_cast<const float2 *>(wp + i * CB_RS + 2);
        d[buf][i][0] = lo.x, d[buf][i][1] = lo.y, d[buf][i][2] = hi.x, d[buf][i][3] = hi.y;
      }
#pragma unroll
      for (int xq = 0; xq < 2; ++xq)
        u[buf][xq] = *reinterpret_cast<const floatx4 *>(ub + (c4 * 8 + half * 2 + xq) * 256);
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
#if CB_SCHED == 0
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int xq = 0; xq < 2; ++xq)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const floatx4 c0 = c4 == 0 ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[xq * 4 + j];
          acc[xq * 4 + j] = mfma16x16x4(u[cur][xq][j], v[xq * 4 + j], c0);
        }
#if CB_SCHED == 0
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s0[4], s1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (half == 0) {
          s0[j] = acc[j][r] + acc[4 + j][r];
          s1[j] = acc[4 + j][r];
        } else {
          s0[j] = acc[j][r];
          s1[j] = -acc[j][r] - acc[4 + j][r];
        }
      }
      const float y0 = s0[0] + s0[1] + s0[2], y1 = s0[1] - s0[2] - s0[3];
      const float y2 = s1[0] + s1[1] + s1[2], y3 = s1[1] - s1[2] - s1[3];
      if (half == 0) y[r][0] = y0, y[r][1] = y1, y[r][2] = y2, y[r][3] = y3;
      else y[r][0] += y0, y[r][1] += y1, y[r][2] += y2, y[r][3] += y3;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

#ifndef CB_SCHED
#define CB_SCHED 0
#endif
#if CB_SCHED == 2
// Variant: one wave per SIMD has nobody to fill the matrix pipe while it transforms, so the NEXT k-step's input
// transform (16 VALU) and the LDS reads of the k-step after it are interleaved with the current k-step's 8 multiplies
// (program order MFMA, 2 VALU, 1 DS read, ... pinned with sched_group_barrier).
template <int NC>
__device__ __forceinline__ void band_layer_il(const float *__restrict__ act, const float *__restrict__ U, int ct, int wb,
                                              int lane, float (&y)[4][4]) {
  const float *wbase = act + (lane >> 4) * CB_CSA + wb;
  const float *ub = U + ct * 1024 + lane * 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    floatx4 acc[8];
    float d[3][3][4];
    floatx4 u[2][2];
    float v[2][8];
    auto fetch_d = [&](int buf, int c4) {
      const float *wp = wbase + c4 * 4 * CB_CSA + half * CB_RS;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float2 lo = *reinterpret_cast<const float2 *>(wp + i * CB_RS);
        const float2 hi = *reinterpret_cast<const float2 *>(wp + i * CB_RS + 2);
        d[buf][i][0] = lo.x, d[buf][i][1] = lo.y, d[buf][i][2] = hi.x, d[buf][i][3] = hi.y;
      }
    };
    auto fetch_u = [&](int buf, int c4) {
#pragma unroll
      for (int xq = 0; xq < 2; ++xq)
        u[buf][xq] = *reinterpret_cast<const floatx4 *>(ub + (c4 * 8 + half * 2 + xq) * 256);
    };
    auto transform = [&](int vb, int db) {
      float t[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (half == 0) {
          t[0][j] = d[db][0][j] - d[db][2][j];
          t[1][j] = d[db][1][j] + d[db][2][j];
        } else {
          t[0][j] = d[db][1][j] - d[db][0][j];
          t[1][j] = d[db][0][j] - d[db][2][j];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        v[vb][i * 4 + 0] = t[i][0] - t[i][2];
        v[vb][i * 4 + 1] = t[i][1] + t[i][2];
        v[vb][i * 4 + 2] = t[i][2] - t[i][1];
        v[vb][i * 4 + 3] = t[i][1] - t[i][3];
      }
    };
    fetch_d(0, 0);
    fetch_u(0, 0);
    if (NC > 1) fetch_d(1, 1);
    transform(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c4 = 0; c4 < NC; ++c4) {
      const int cur = c4 & 1;
      if (c4 + 1 < NC) transform(cur ^ 1, (c4 + 1) % 3);
      if (c4 + 2 < NC) fetch_d((c4 + 2) % 3, c4 + 2);
      if (c4 + 1 < NC) fetch_u(cur ^ 1, c4 + 1);
#pragma unroll
      for (int xq = 0; xq < 2; ++xq)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const floatx4 c0 = c4 == 0 ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[xq * 4 + j];
          acc[xq * 4 + j] = mfma16x16x4(u[cur][xq][j], v[cur][xq * 4 + j], c0);
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s0[4], s1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (half == 0) {
          s0[j] = acc[j][r] + acc[4 + j][r];
          s1[j] = acc[4 + j][r];
        } else {
          s0[j] = acc[j][r];
          s1[j] = -acc[j][r] - acc[4 + j][r];
        }
      }
      const float y0 = s0[0] + s0[1] + s0[2], y1 = s0[1] - s0[2] - s0[3];
      const float y2 = s1[0] + s1[1] + s1[2], y3 = s1[1] - s1[2] - s1[3];
      if (half == 0) y[r][0] = y0, y[r][1] = y1, y[r][2] = y2, y[r][3] = y3;
      else y[r][0] += y0, y[r][1] += y1, y[r][2] += y2, y[r][3] += y3;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
#define band_layer band_layer_il
#endif

// sums of two values over the 32 lanes of each half-wave; the totals land in lanes 16..31 / 48..63
__device__ __forceinline__ void cb_half_wave_sums(float (&s)[2]) {
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

// wave-wide min / max of two ints (DPP within the 16-lane rows, then the four rows through scalar registers)
template <int CTRL>
__device__ __forceinline__ int cb_dpp_int(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ void cb_wave_minmax(int &lo, int &hi) {
  lo = min(lo, cb_dpp_int<0xB1>(lo)), hi = max(hi, cb_dpp_int<0xB1>(hi));
  lo = min(lo, cb_dpp_int<0x4E>(lo)), hi = max(hi, cb_dpp_int<0x4E>(hi));
  lo = min(lo, cb_dpp_int<0x141>(lo)), hi = max(hi, cb_dpp_int<0x141>(hi));
  lo = min(lo, cb_dpp_int<0x140>(lo)), hi = max(hi, cb_dpp_int<0x140>(hi));
  lo = min(min(__builtin_amdgcn_readlane(lo, 0), __builtin_amdgcn_readlane(lo, 16)),
           min(__builtin_amdgcn_readlane(lo, 32), __builtin_amdgcn_readlane(lo, 48)));
  hi = max(max(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(hi, 16)),
           max(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(hi, 48)));
}

__global__ __launch_bounds__(CB_THREADS) void chain_band_kernel(ChainArgs a, int flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int rows = CB_ROWS, cols = CB_COLS, P = CB_P, RS = CB_RS;
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int n = blockIdx.x / CB_G, m = blockIdx.x % CB_G;
  const int pt = wave >> 1, ct = wave & 1;
  const int lo = m * CB_BR, hi = lo + CB_BR - 1, wlo = lo - CB_W;
  const int D = a.D;
  int tid = tid0;

  float *U = smem;
  float *sparams = U + CW_U0_FLOATS;
  float *red = sparams + CH_SP_FLOATS;
  int *range = reinterpret_cast<int *>(red + CB_RED);       // [parity][min y0, max y0 + 1]
  float *maskb = red + CB_RED + CB_RANGE;
  float *tab = maskb + CB_MASK;
  int *tabi = reinterpret_cast<int *>(tab);
  float *act = tab + CB_TAB;
  float *win = act + 36 * CB_CSA;

  gu64 *ws = (gu64 *)(reinterpret_cast<u64 *>(a.workspace) + (size_t)n * CB_CHAIN_U64);
  gu64 *Fg = ws + CB_FG, *Rg = ws + CB_RG, *Sg = ws + CB_SG;
  gu32 *status = (gu32 *)(reinterpret_cast<u64 *>(a.workspace) + (size_t)(gridDim.x / CB_G) * CB_CHAIN_U64);
  bool dead = false;

  const float *upk = a.packed + CH_DIRECT_FLOATS;
  int lane16 = lane * 16;
  auto dma_u = [&](const float *src, int nchunks) {   // 1 KB runs, wave w takes runs w, w + 4, ...
    const int runs = nchunks * (CW_UCHUNK / 256);
    const char *base = reinterpret_cast<const char *>(src + (size_t)wave * 256);
    for (int run = wave, i = 0; run < runs; run += CB_WAVES, ++i)
      __builtin_amdgcn_global_load_lds(CB_GPTR(base + (size_t)i * (CB_WAVES * 1024) + (unsigned)lane16),
                                       CB_LPTR(U + run * 256), 16, 0, 0);
  };
  auto dma_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // ---- one-time set-up ---------------------------------------------------------------------
  dma_u(upk, 9);
  for (int i = tid; i < 36 * CB_CSA + 32 * CB_CSW; i += CB_THREADS) act[i] = 0.0f;
  for (int i = tid; i < CH_SP_FLOATS; i += CB_THREADS) sparams[i] = a.packed[CH_W0_FLOATS + 2 * CH_W1_FLOATS + i];
  if (tid < 4) range[tid] = (tid & 1) ? -1 : 1 << 20;
  const float *bias0 = sparams, *gn0w = sparams + 32, *gn0b = sparams + 64, *bias1 = sparams + 96,
              *gn1w = sparams + 128, *gn1b = sparams + 160, *bias2 = sparams + 192;

  // this lane's patch: patch row pt of the band, patch column p; outputs (lo + 2pt + a, 2p + b), e = a*2 + b
  const int p = lane & 15, k = lane >> 4;
  const int wb = (2 * pt) * RS + 2 * p;               // window origin inside an act plane (local row 0 = image row lo-1)
  const int ob = wb + RS + 1;                         // output (0,0)
  const int cbase = ct * 16 + k * 4;                  // this lane's couts: cbase + r
  const int gown = ct * 2 + (k >> 1);                 // their GroupNorm group
  const int py0 = lo + 2 * pt, px0 = 2 * p;
  const float inv_n = 1.0f / (8.0f * (float)P);
  // halo role: thread -> (side hs: 0 = row lo-1, 1 = row hi+1; column hx; channels 8*hcg .. 8*hcg+7)
  const int hs = tid0 >> 7, hx = tid0 & 31, hcg = (tid0 >> 5) & 3;
  const int hy = hs ? hi + 1 : lo - 1;
  const bool hvalid = hy >= 0 && hy < rows;           // the image has such a row <=> that neighbour exists
  const int hnb = hs ? m + 1 : m - 1;                 // the band that owns it
  float *hact = act + (hs ? CB_BR + 1 : 0) * RS + hx + 1;
  // image role: threads 0..191 -> one pixel of rows lo-1 .. hi+1
  const int er = tid0 >> 5, iy = lo - 1 + er, ixx = tid0 & 31;
  const bool ivalid = tid0 < (CB_BR + 2) * cols && iy >= 0 && iy < rows;
  const bool iband = ivalid && iy >= lo && iy <= hi;
  float shift0[2] = {0.f, 0.f}, shift1[2] = {0.f, 0.f};   // previous step's group means: [own group, halo group]
  __syncthreads();

  const float *Hn = a.H + (size_t)n * D * 9;
  const float *Hin = a.Hinc + (size_t)n * D * 9;
  const float *src = a.src + (size_t)n * 3 * P;
  uint8_t *maskg = a.mask + (size_t)n * D * P;
  float *costg = a.cost + (size_t)n * 32 * D * P;
  float *fvolg = a.fvol ? a.fvol + (size_t)n * 32 * D * P : nullptr;
  const float *flp = a.fl + (size_t)(n % a.B) * 32 * P;
  const float *fl_lane = flp + (size_t)cbase * P + py0 * cols + px0;
  int slice_off = (cbase * D) * P + py0 * cols + px0;

  // plane d's features of this lane (f[r][e]) -> own rows of the gather window, the granules the other bands
  // gather from (tag d + 1), and the cost slice (not mask) * |left - right| straight from the registers
  auto emit = [&](int d, const float (&f)[4][4], const float2 (&fl)[4][2]) {
    const float2 m0 = *reinterpret_cast<const float2 *>(maskb + (2 * pt) * cols + px0);
    const float2 m1 = *reinterpret_cast<const float2 *>(maskb + (2 * pt + 1) * cols + px0);
    const bool out[4] = {m0.x != 0.0f, m0.y != 0.0f, m1.x != 0.0f, m1.y != 0.0f};
    float *cd = costg + (size_t)d * P;
    float *fd = fvolg ? fvolg + (size_t)d * P : nullptr;
    // the granules first: the other bands wait for them, everything else of the epilogue travels meanwhile
    if (d + 1 < D) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gu64 *g = Fg + (size_t)(cbase + r) * P + py0 * cols + px0;
        cb_publish2(g, d + 1, f[r][0], f[r][1]);
        cb_publish2(g + cols, d + 1, f[r][2], f[r][3]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float *dst = win + (cbase + r) * CB_CSW + (CB_W + 2 * pt) * RS + px0 + 1;
      dst[0] = f[r][0], dst[1] = f[r][1], dst[RS] = f[r][2], dst[RS + 1] = f[r][3];
      float *cdst = cd + (r * D) * P + slice_off;
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2) {
        float2v c2;
        c2.x = out[a2 * 2] ? 0.0f : fabsf(fl[r][a2].x - f[r][a2 * 2]);
        c2.y = out[a2 * 2 + 1] ? 0.0f : fabsf(fl[r][a2].y - f[r][a2 * 2 + 1]);
        __builtin_nontemporal_store(c2, reinterpret_cast<float2v *>(cdst + a2 * cols));
      }
      if (fd) {
        float *fdst = fd + (r * D) * P + slice_off;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
          float2v f2;
          f2.x = out[a2 * 2] ? 0.0f : f[r][a2 * 2];
          f2.y = out[a2 * 2 + 1] ? 0.0f : f[r][a2 * 2 + 1];
          __builtin_nontemporal_store(f2, reinterpret_cast<float2v *>(fdst + a2 * cols));
        }
      }
    }
  };
  auto load_left = [&](float2 (&fl)[4][2]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2) fl[r][a2] = *reinterpret_cast<const float2 *>(fl_lane + (size_t)r * P + a2 * cols);
  };

  // ---- plane 0: mask from the plane's homography, features from the extractor ---------------------
  if (iband) {
    float Hl[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hl[i] = Hn[i];
    WarpCoord c = warp_coord(Hl, (float)ixx, (float)iy, (float)rows, (float)cols);
    maskb[(iy - lo) * cols + ixx] = c.outside ? 1.0f : 0.0f;
    maskg[iy * cols + ixx] = c.outside ? 1 : 0;
  }
  __syncthreads();
  {
    const float *f0 = a.f0 + (size_t)n * 32 * P + (size_t)cbase * P + py0 * cols + px0;
    float f[4][4];
    float2 fl[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float2 t0 = *reinterpret_cast<const float2 *>(f0 + (size_t)r * P);
      const float2 t1 = *reinterpret_cast<const float2 *>(f0 + (size_t)r * P + cols);
      f[r][0] = t0.x, f[r][1] = t0.y, f[r][2] = t1.x, f[r][3] = t1.y;
    }
    load_left(fl);
    emit(0, f, fl);
  }

#define CB_STAMP(i)                                                                                     \
  do {                                                                                                  \
    if (a.dbg && blockIdx.x == 0 && tid == 0 && d >= 2 && d <= 5) a.dbg[(d - 2) * 32 + (i)] = __builtin_readcyclecounter(); \
  } while (0)

  // Everything of a step that does not depend on the previous plane's features -- the image plane of rows
  // lo-1 .. hi+1 with the band's mask (A1: global gathers from the 6 KB source image) and where the incremental
  // homography sends this thread's pixels (A2: own 2x2 patch + one halo pixel; bilinear weights, window offsets,
  // and the rows of the previous plane the band's gathers touch) -- is computed one step AHEAD, inside the wait of
  // the previous step's second hand-off (between publishing and the first poll), where the workgroup would idle.
  float img[3] = {0.f, 0.f, 0.f}, mk = 0.f;
  auto prepare = [&](int dn) {
    int ymin = 1 << 20, ymax = -1;
    if (ivalid) {
      float Hl[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Hl[i] = Hn[dn * 9 + i];
      {
        WarpCoord c = warp_coord(Hl, (float)ixx, (float)iy, (float)rows, (float)cols);
        Bilinear b = bilinear_taps(c.ix, c.iy, rows, cols);
        const float keep = c.outside ? 0.0f : 1.0f;
        mk = c.outside ? 1.0f : 0.0f;
        const int o00 = b.y0 * cols + b.x0, o01 = b.y0 * cols + b.x1, o10 = b.y1 * cols + b.x0, o11 = b.y1 * cols + b.x1;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float *ic = src + (size_t)ch * P;
          img[ch] = keep * (ic[o00] * b.w00 + ic[o01] * b.w01 + ic[o10] * b.w10 + ic[o11] * b.w11);
        }
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) Hl[i] = Hin[dn * 9 + i];
      WarpCoord c = warp_coord(Hl, (float)ixx, (float)iy, (float)rows, (float)cols);
      Bilinear b = bilinear_taps(c.ix, c.iy, rows, cols);
      const float keep = c.outside ? 0.0f : 1.0f;
      floatx4 w4;
      intx4 i4;
      w4[0] = keep * b.w00, w4[1] = keep * b.w01, w4[2] = keep * b.w10, w4[3] = keep * b.w11;
      i4[0] = (b.y0 - wlo) * RS + b.x0 + 1;
      i4[1] = (b.y1 - b.y0) * RS;   // 0 where the +1 row is clamped (its weight is exactly zero there)
      i4[2] = b.y0;
      i4[3] = b.x0;
      *reinterpret_cast<floatx4 *>(tab + tid0 * 8) = w4;
      *reinterpret_cast<intx4 *>(tabi + tid0 * 8 + 4) = i4;
      ymin = b.y0, ymax = b.y1;
    }
    cb_wave_minmax(ymin, ymax);
    if (lane == 0) {
      atomicMin(&range[(dn & 1) * 2], ymin);
      atomicMax(&range[(dn & 1) * 2 + 1], ymax);
    }
  };
  if (D > 1) prepare(1);
  cb_barrier();   // step 1's row range is complete; plane 0 sits in the window

  // ---- the recurrence ------------------------------------------------------------------------
  for (int d = 1; d < D; ++d) {
    asm volatile("" : "+v"(tid), "+v"(lane16), "+v"(slice_off));
    const int par = d & 1;
    CB_STAMP(0);
    // (the range of this step was completed during the previous step, two barriers ago)
    const int need_lo = range[par * 2], need_hi = range[par * 2 + 1];
    const bool fast = !(flags & 1) && need_lo >= wlo && need_hi <= wlo + CB_WSLOTS - 1;   // workgroup-uniform

    float fp[4][4], hv[8];
    float tw[5][4];
    int to[5], tdy[5], ty[5], tx[5];
    auto footprints = [&]() {   // this thread's five pixels: own 2x2 patch (ext rows 1 + 2pt + a) and one halo pixel
#pragma unroll
      for (int e = 0; e < 5; ++e) {
        const int pix = e < 4 ? (1 + 2 * pt + (e >> 1)) * cols + px0 + (e & 1) : (hs ? CB_BR + 1 : 0) * cols + hx;
        const floatx4 w4 = *reinterpret_cast<const floatx4 *>(tab + pix * 8);
        const intx4 i4 = *reinterpret_cast<const intx4 *>(tabi + pix * 8 + 4);
        tw[e][0] = w4[0], tw[e][1] = w4[1], tw[e][2] = w4[2], tw[e][3] = w4[3];
        to[e] = i4[0], tdy[e] = i4[1], ty[e] = i4[2], tx[e] = i4[3];
      }
    };
    if (fast) {
      // E1 (consume): rows of F_{d-1} the gathers need from the neighbours -> window.  Thread: column tid & 31,
      // channels (tid >> 5) + 8 j.  Up to 7 rows, all loads in flight before the first tag is looked at.
      {
        const int fx = tid & 31, fc = tid >> 5;
        float v[7][4];
        int slot_of[7];
        bool need[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
          slot_of[s] = s < CB_W ? s : s + CB_BR;                 // window slots 0..2 and 7..10
          const int row = wlo + slot_of[s];
          need[s] = row >= need_lo && row <= need_hi && row >= 0 && row < rows;
        }
        for (unsigned spins = 0;; ++spins) {
          bool ok = true;
#pragma unroll
          for (int s = 0; s < 7; ++s)
            if (need[s]) {
              const gu64 *g = Fg + (size_t)fc * P + (wlo + slot_of[s]) * cols + fx;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const u64 x = __hip_atomic_load(g + (size_t)j * 8 * P, CB_RLX_AGENT);
                v[s][j] = __builtin_bit_cast(float, (unsigned)x);
                ok &= (unsigned)(x >> 32) == (unsigned)d;
              }
            }
          if (__all(ok) || dead) break;
          if (spins >= CB_SPIN_LIMIT) {
            dead = true;
            __hip_atomic_store(status, 2u, CB_RLX_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int s = 0; s < 7; ++s)
          if (need[s]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) win[(fc + 8 * j) * CB_CSW + slot_of[s] * RS + fx + 1] = v[s][j];
          }
      }
      footprints();
      CB_STAMP(2);
      cb_barrier();   // B2: window complete
      // A2 gather.  The +1 column tap is read unclamped: where the clamp would act its weight is exactly zero and the
      // slot read is the zero halo column; the +1 row tap re-reads row y0 there (weight exactly zero as well).
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float *fc = win + (cbase + r) * CB_CSW + to[e];
          fp[r][e] = fc[0] * tw[e][0] + fc[1] * tw[e][1] + fc[tdy[e]] * tw[e][2] + fc[tdy[e] + 1] * tw[e][3];
        }
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[j] = 0.0f;
      if (hvalid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float *fc = win + (hcg * 8 + j) * CB_CSW + to[4];
          hv[j] = fc[0] * tw[4][0] + fc[1] * tw[4][1] + fc[tdy[4]] * tw[4][2] + fc[tdy[4] + 1] * tw[4][3];
        }
      }
    } else {
      // the gathers reach beyond the window (large inter-plane motion): every tap straight from the granules
      footprints();
#pragma unroll
      for (int e = 0; e < 5; ++e) {
        const bool act_e = e < 4 || hvalid;
        const bool x1 = tx[e] + 1 < cols, y1 = tdy[e] != 0;   // clamped +1 taps: weight exactly zero, value unused
        constexpr int NCH_MAX = 8;
        const int nch = e < 4 ? 4 : 8;
        const int c0 = e < 4 ? cbase : hcg * 8;
        float t00[NCH_MAX], t01[NCH_MAX], t10[NCH_MAX], t11[NCH_MAX];
        const gu64 *g = Fg + (size_t)c0 * P + ty[e] * cols + tx[e];
        cb_sweep<NCH_MAX>([&](int j) { return g + (size_t)(j < nch ? j : 0) * P; }, d, act_e, t00, dead, status);
        cb_sweep<NCH_MAX>([&](int j) { return g + (size_t)(j < nch ? j : 0) * P + (x1 ? 1 : 0); }, d, act_e, t01, dead, status);
        cb_sweep<NCH_MAX>([&](int j) { return g + (size_t)(j < nch ? j : 0) * P + (y1 ? cols : 0); }, d, act_e, t10, dead, status);
        cb_sweep<NCH_MAX>([&](int j) { return g + (size_t)(j < nch ? j : 0) * P + (y1 ? cols : 0) + (x1 ? 1 : 0); }, d, act_e,
                          t11, dead, status);
#pragma unroll
        for (int j = 0; j < NCH_MAX; ++j) {
          const float v01 = x1 ? t01[j] : 0.0f, v10 = y1 ? t10[j] : 0.0f, v11 = (x1 && y1) ? t11[j] : 0.0f;
          const float val = t00[j] * tw[e][0] + v01 * tw[e][1] + v10 * tw[e][2] + v11 * tw[e][3];
          if (e < 4) {
            if (j < 4) fp[j][e] = val;
          } else {
            hv[j] = hvalid ? val : 0.0f;
          }
        }
      }
      cb_barrier();   // (keeps the barrier count of the two paths equal)
    }
    if (tid == 0) range[par * 2] = 1 << 20, range[par * 2 + 1] = -1;   // (read by everyone before B2; next use: step d+2)
    CB_STAMP(3);

    // A3: lay out the refiner input [image(3) | moved features(32)] on rows lo-1 .. hi+1
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float *dst = act + (3 + cbase + r) * CB_CSA + ob;
      dst[0] = fp[r][0], dst[1] = fp[r][1], dst[RS] = fp[r][2], dst[RS + 1] = fp[r][3];
    }
    if (hvalid) {
#pragma unroll
      for (int j = 0; j < 8; ++j) hact[(3 + hcg * 8 + j) * CB_CSA] = hv[j];
    }
    if (ivalid) {
      const int o = er * RS + ixx + 1;
      act[0 * CB_CSA + o] = img[0];
      act[1 * CB_CSA + o] = img[1];
      act[2 * CB_CSA + o] = img[2];
      if (iband) {
        maskb[(iy - lo) * cols + ixx] = mk;
        maskg[(size_t)d * P + iy * cols + ixx] = mk != 0.0f ? 1 : 0;
      }
    }
    dma_landed();   // conv0's U
    cb_barrier();   // B3
    CB_STAMP(4);

    float y[4][4];
    band_layer<9>(act, U, ct, wb, lane, y);
    CB_STAMP(5);
    cb_barrier();   // B4: act and U free
    dma_u(upk + CW_U0_FLOATS, 8);

    // E2 / E3: bias, partial GroupNorm sums (shifted by the previous step's mean, as chain_wino_kernel), publish
    // them with the band's boundary rows; collect the other bands'; normalise + activate own outputs and halo rows
    auto exchange = [&](int layer, const float *bias, const float *gamma, const float *beta, float (&shift)[2],
                        bool residual, auto &&meanwhile) {
      float s[2] = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float b = bias[cbase + r];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[r][e] += b;
          const float dv = y[r][e] - shift[0];
          s[0] += dv;
          s[1] += dv * dv;
        }
      }
      cb_half_wave_sums(s);
      gu64 *Sl = Sg + (size_t)layer * (CB_G * CB_WAVES * 4);
      if ((lane & 31) == 16) {
        gu64 *g = Sl + (m * CB_WAVES + wave) * 4 + (lane >> 5) * 2;
        cb_publish2(g, d, s[0], s[1]);
      }
      // boundary rows: patch row 0 holds the band's first pixel row (e = 0, 1), patch row 1 its last (e = 2, 3)
      gu64 *Rl = Rg + (size_t)layer * (CB_G * 2 * 32 * cols);
      if (pt == 0 ? m > 0 : m < CB_G - 1) {
        gu64 *g = Rl + ((size_t)(m * 2 + pt) * 32 + cbase) * cols + px0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cb_publish2(g + r * cols, d, y[r][pt * 2], y[r][pt * 2 + 1]);
        }
      }
      CB_STAMP(16 + layer * 4);
      meanwhile();   // work that needs none of the hand-off, placed where the workgroup would otherwise only wait
      CB_STAMP(17 + layer * 4);
      // collect, ONE sweep: every thread its 8 halo granules, wave 0 also the 64 sum granules (one per lane)
      float hr[8];
      {
        // the neighbour's row facing this band: its last row (side 1) for our row lo-1, its first (side 0) for hi+1
        const gu64 *g = Rl + ((size_t)(hnb * 2 + (hs ? 0 : 1)) * 32 + hcg * 8) * cols + hx;
        float sv = 0.f;
        for (unsigned spins = 0;; ++spins) {
          bool ok = true;
          if (wave == 0) {
            const u64 x = __hip_atomic_load(Sl + lane, CB_RLX_AGENT);
            sv = __builtin_bit_cast(float, (unsigned)x);
            ok &= (unsigned)(x >> 32) == (unsigned)d;
          }
          if (hvalid) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const u64 x = __hip_atomic_load(g + j * cols, CB_RLX_AGENT);
              hr[j] = __builtin_bit_cast(float, (unsigned)x);
              ok &= (unsigned)(x >> 32) == (unsigned)d;
            }
          }
          if (__all(ok) || dead) break;
          if (spins >= CB_SPIN_LIMIT) {
            dead = true;
            __hip_atomic_store(status, 3u + layer, CB_RLX_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (wave == 0) red[lane] = sv;
      }
      CB_STAMP(18 + layer * 4);
      cb_barrier();
      CB_STAMP(19 + layer * 4);
      // totals in a fixed order (band-major): every workgroup of the chain forms the same statistics bit for bit
      float sc[2], sh2[2];   // [0] own group, [1] halo group: rstd and mean
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int g = w == 0 ? gown : hcg;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mm = 0; mm < CB_G; ++mm)
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            const float2 rec = *reinterpret_cast<const float2 *>(red + (mm * CB_WAVES + pp * 2 + (g >> 1)) * 4 + (g & 1) * 2);
            s1 += rec.x;
            s2 += rec.y;
          }
        const float ms = s1 * inv_n;
        const float var = fmaxf(s2 * inv_n - ms * ms, 0.0f);
        const float mean = shift[w] + ms;
        shift[w] = mean;
        sc[w] = 1.0f / sqrtf(var + CB_GN_EPS);
        sh2[w] = mean;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = cbase + r;
        const float scl = sc[0] * gamma[c];
        const float sft = beta[c] - sh2[0] * scl;
        float *dst = act + c * CB_CSA + ob;
        if (residual) {
          dst[0] += lrelu02(y[r][0] * scl + sft), dst[1] += lrelu02(y[r][1] * scl + sft);
          dst[RS] += lrelu02(y[r][2] * scl + sft), dst[RS + 1] += lrelu02(y[r][3] * scl + sft);
        } else {
          dst[0] = lrelu02(y[r][0] * scl + sft), dst[1] = lrelu02(y[r][1] * scl + sft);
          dst[RS] = lrelu02(y[r][2] * scl + sft), dst[RS + 1] = lrelu02(y[r][3] * scl + sft);
        }
      }
      if (hvalid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = hcg * 8 + j;
          const float scl = sc[1] * gamma[c];
          const float sft = beta[c] - sh2[1] * scl;
          const float v = lrelu02(hr[j] * scl + sft);
          if (residual) hact[c * CB_CSA] += v;
          else hact[c * CB_CSA] = v;
        }
      }
    };
    exchange(0, bias0, gn0w, gn0b, shift0, false, [] {});
    CB_STAMP(6);
    dma_landed();
    cb_barrier();   // B6
    CB_STAMP(7);

    band_layer<8>(act, U, ct, wb, lane, y);
    CB_STAMP(8);
    cb_barrier();   // B7
    dma_u(upk + CW_U0_FLOATS + CW_U1_FLOATS, 8);
    exchange(1, bias1, gn1w, gn1b, shift1, true, [&] {   // x2 = x1 + LReLU(GN(conv1(x1)))
      if (d + 1 < D) prepare(d + 1);
    });
    CB_STAMP(9);
    dma_landed();
    cb_barrier();   // B10
    CB_STAMP(10);

    band_layer<8>(act, U, ct, wb, lane, y);
    float2 fl[4][2];
    load_left(fl);
    CB_STAMP(11);
    cb_barrier();   // B11
    dma_u(upk, 9);  // conv0 of the next step

    // epilogue: F_d = moved + conv_final(...); window rows, granules for the other bands, cost slice
    {
      float f[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float b2 = bias2[cbase + r];
#pragma unroll
        for (int e = 0; e < 4; ++e) f[r][e] = fp[r][e] + (y[r][e] + b2);
      }
      emit(d, f, fl);
    }
    CB_STAMP(12);
  }
#undef CB_STAMP
  dma_landed();       // the last step's look-ahead fetch must not outlive the workgroup's LDS
}

bool chain_band_supported(int rows, int cols) { return rows == CB_ROWS && cols == CB_COLS; }

size_t chain_band_workspace_bytes(int n_chains) { return ((size_t)n_chains * CB_CHAIN_U64 + 8) * sizeof(u64); }

int chain_band_groups() { return CB_G; }

size_t chain_band_status_offset(int n_chains) { return (size_t)n_chains * CB_CHAIN_U64 * sizeof(u64); }

// status word behind the granules: 0 = every hand-off completed; otherwise the code of the hand-off that timed out
int chain_band_launch(const ChainArgs &a, int n_chains, void *workspace, size_t workspace_bytes, int flags,
                      hipStream_t stream) {
  const size_t need = chain_band_workspace_bytes(n_chains);
  MVSN_REQUIRE(workspace && workspace_bytes >= need, MVSN_E_WORKSPACE,
               "mvsn_incremental_cost_volume(banded): workspace of %zu bytes required", need);
  MVSN_REQUIRE(n_chains * CB_G <= device_cus(), MVSN_E_TOOLARGE,
               "mvsn_incremental_cost_volume(banded): %d chains x %d bands exceed the %d CUs that must be co-resident",
               n_chains, CB_G, device_cus());
  // every polled word starts from tag 0 (no step carries it): a memset node ahead of the launch, replayed with it
  hipError_t e = hipMemsetAsync(workspace, 0, need, stream);
  if (e != hipSuccess) {
    set_error("mvsn_incremental_cost_volume(banded): memset failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  ChainArgs b = a;
  b.workspace = (float *)workspace;
  const size_t lds = (size_t)CB_LDS_FLOATS * sizeof(float);
  static LdsOptIn opt;
  if (int rc = ensure_lds(opt, (const void *)chain_band_kernel, lds, "mvsn_incremental_cost_volume(banded)")) return rc;
  hipLaunchKernelGGL(chain_band_kernel, dim3(n_chains * CB_G), dim3(CB_THREADS), lds, stream, b, flags);
  return check_launch("mvsn_incremental_cost_volume(banded)");
}

}  // namespace mvsn
