// The fused incremental chain, BANDED Winograd form: one chain on SEVERAL workgroups (see include/mvsn_hip.h:
// mvsn_incremental_cost_volume, form MVSN_CHAIN_BANDED).
//
// The plane-resident forms (mvsn_chain_wino.hip, mvsn_chain.hip) give a chain to one workgroup: with one chain per CU
// in flight that fills the chip, but at batch 1 (the reference's own evaluation loop, test.py:38,197-200, and BASELINE
// config 3's one image per GPU) the D - 1 sequential steps of multi_view_stereonet.py:279-290 run on S of the 256 CUs --
// and a 30x40 or 32x64 coarse plane (640x480 / 1024x512 frames) does not fit one CU at all.  Here the coarse plane is
// cut into G = rows / BR bands of BR pixel rows; workgroup m of a chain owns band m for the whole recurrence (same
// arithmetic per output as the plane-resident Winograd kernel: F(2x2,3x3) on v_mfma_f32_16x16x4_f32, GroupNorm over the
// WHOLE plane, masks from the reference's fp32 expression order), and the bands exchange, per step,
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
// Geometries (template BandGeo<ROWS, COLS, BR, W>; a band holds at most 32 patches of 2x2 pixels = two MFMA tiles):
//   16x32 (512x256 frames):   8 bands of 2 rows, "half split" (below), while 8 workgroups per chain fit the chip in one
//                             pass (<= CUs / 8 chains); 4 bands of 4 rows beyond; gather window +-3 rows.  The two
//                             plans give the same bits (same arithmetic per output, same GroupNorm records and order)
//   30x40 (640x480 frames):  15 bands of 2 rows (20 patches: the second tile is a quarter full), window +-3 rows
//   32x64 (1024x512 frames): 16 bands of 2 rows, window +-1 row (what fits next to one layer of transformed weights)
//
// Hand-off protocol (cdna_hip_programming.md section 6 Guideline 16, form R2): every exchanged float is ONE naturally
// aligned 8-byte granule {value, tag = step} written write-through (two neighbouring granules per 16-byte
// `global_store_dwordx4 sc1`) and read by agent-scope relaxed loads (`sc1`: served past the reader's L1) that are
// repeated until the tag matches.  The data is its own flag: no release fence (the 14 us per launch HISTORY 3.6
// measured for buffer_wbl2), no acquire, no dependence on which XCD a workgroup landed on.  The granule region is
// zeroed by a fill kernel ahead of every pass; tags count steps within the call (F_d carries d + 1, the conv
// hand-offs of step d carry d), so a granule is either stale (tag - 1: keep polling) or current.  A buffer can be
// single: a band cannot publish step d + 1's version of a hand-off before every reader has consumed step d's, because
// the GroupNorm sums of the hand-off in between need all bands (the ordering argument is spelled out in DESIGN.md 3.1).
// Spins are bounded: on a time-out the workgroup records it in the status word and stops waiting (the host reports it).
//
// Work split inside a workgroup: 256 threads = 4 waves, ONE per SIMD (512 registers each); wave (pt, ct) owns patch
// tile pt of the band (16 consecutive patches, row-major) and cout tile ct.  Per k-step and transform-row half: 8
// multiplies against the plane-resident kernel's 16 -- the input transform is repeated by the two cout-tile waves.
// Half split (a band of ONE patch tile): wave (h, ct) runs transform-row half h of every layer for the band's only
// tile; the pair exchanges its halves' outputs through LDS behind the barrier that follows the layer (both hold
// half 0 + half 1, the one-wave sum) and shares the rest of the step by pixel row (wave h: row h of every patch).
//
// LDS (floats):  U 18432 (one layer's transformed weights, LDS-DMA'd per layer exactly as in chain_wino_kernel)
//                sparams 224 | red 16 G (every band's, every wave's GroupNorm records of a hand-off) | gstat 32
//                range 16 | maskb BR x cols | tab: per pixel of rows lo-1 .. hi+1 the bilinear footprint of its
//                incremental-homography gather (4 weights + offset + row step), computed ONCE per pixel a step ahead
//                act 36 x CSA: layer input planes, BR + 2 rows (band + one halo row either side) x (cols + 2)
//                win 32 x CSW: gather window = feature rows lo-W .. hi+W+1 of the previous plane (own band written by
//                      the epilogue, the others fetched from the neighbours' granules when the step's gather needs
//                      them; rows outside the image stay zero and double as the bilinear taps' zero halo)
#include <type_traits>

#include "mvsn_chain.h"
#include "mvsn_common.h"

namespace mvsn {

constexpr int CB_THREADS = 256, CB_WAVES = 4;
constexpr float CB_GN_EPS = 1e-5f;
constexpr unsigned CB_SPIN_LIMIT = 1u << 21;

template <int ROWS, int COLS, int BR_, int W_, int CSA_, bool HS_ = false>
struct BandGeo {
  static constexpr int rows = ROWS, cols = COLS, BR = BR_, W = W_, G = ROWS / BR_, P = ROWS * COLS, RS = COLS + 2;
  // HS ("half split"): a band of ONE patch tile; the waves that would own the second tile take the second
  // transform-row half of every layer instead (wave = (half h, cout tile ct)); the two waves of a pair hand each other
  // their half's outputs through LDS (both end up with the sum) and share the rest of the step by pixel row: wave h
  // gathers, publishes, normalises and emits row h of every 2x2 patch (the GroupNorm sums stay with h = 0: same order)
  static constexpr bool HS = HS_;
  static constexpr int COMB = HS_ ? 2 * 2 * 64 * 16 : 0;          // [half][cout tile][r][lane][e]
  static constexpr int PCOLS = COLS / 2, PROWS = BR_ / 2, NPATCH = PROWS * PCOLS;
  static constexpr int AROWS = BR_ + 2, CSA = CSA_;             // layer input planes: rows, channel stride
  static constexpr int WSLOTS = BR_ + 2 * W_ + 1, CSW = WSLOTS * RS + 2;   // gather window: row slots, channel stride
  static constexpr int NOWN = WSLOTS - BR_;                      // window rows that belong to other bands
  static constexpr int EXT = AROWS * COLS;                       // pixels of rows lo-1 .. hi+1 (image / table role)
  static constexpr int RED = 16 * G;                             // GroupNorm records of a hand-off: [band][wave][4]
  static constexpr int HITEMS = (8 * COLS + CB_THREADS - 1) / CB_THREADS;   // halo items (side, channel group, column) per thread
  static constexpr int FROW = (32 * COLS) / CB_THREADS;          // granules per thread and fetched window row
  static constexpr int TABW = EXT * 4, TABI = EXT * 2;
  static constexpr int MASK = BR_ * COLS;
  static constexpr int LDS_FLOATS =
      CW_U0_FLOATS + CH_SP_FLOATS + RED + 32 + 16 + MASK + TABW + TABI + 36 * CSA + 32 * CSW + COMB;
  // granule workspace of one chain (u64 units)
  static constexpr size_t FG = 0;                                          // [32 ch][rows][cols]
  static constexpr size_t RG = FG + 32 * (size_t)P;                        // [layer 2][band][side 2][32 ch][cols]
  static constexpr size_t SG = RG + 2 * (size_t)G * 2 * 32 * COLS;         // [layer 2][band][wave 4][4]
  static constexpr size_t CHAIN_U64 = SG + 2 * (size_t)G * CB_WAVES * 4;
  static_assert(ROWS % BR_ == 0 && BR_ % 2 == 0 && COLS % 2 == 0, "bands of whole 2x2 patches");
  static_assert(NPATCH <= (HS_ ? 16 : 32), "a band is at most two MFMA patch tiles (half split: one)");
  static_assert(EXT <= CB_THREADS, "one thread per pixel of the band + halo rows");
  static_assert((32 * COLS) % CB_THREADS == 0, "window rows are fetched in whole rounds");
  static_assert(CSA % 2 == 0 && CSA >= AROWS * RS, "activation planes: 8-byte aligned rows");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS plan");
};
typedef BandGeo<16, 32, 4, 3, 224> Band16x32;   // CSA = 32 (mod 64): the four channels of a k-step read disjoint banks
typedef BandGeo<16, 32, 2, 3, 160, true> Band16x32H;   // 8 bands of 2 rows, half split (few chains: <= CUs / 8)
typedef BandGeo<30, 40, 2, 3, 168> Band30x40;
typedef BandGeo<32, 64, 2, 1, 264> Band32x64;

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef float float2v __attribute__((ext_vector_type(2)));
typedef int intx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
#define CB_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define CB_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define CB_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

__device__ __forceinline__ void cb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Two neighbouring granules (16-byte aligned pair) in ONE write-through store: half the fabric writes of a publish.
// Each 8-byte half is still a self-contained {value, tag} granule for the reader (8-byte halves of a 16-byte sc1
// store are observed untorn on gfx950, MI355X_MICROARCH.md section "inter-workgroup visibility"; a torn PAIR is
// harmless, the reader checks each granule's own tag).
__device__ __forceinline__ void cb_publish2(gu64 *g, unsigned tag, float v0, float v1) {
  uintx4 q;
  q[0] = __builtin_bit_cast(unsigned, v0), q[1] = tag, q[2] = __builtin_bit_cast(unsigned, v1), q[3] = tag;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(g), "v"(q) : "memory");
}

// Re-read this lane's N granules until every tag in the wave matches.  `addr(j)` = granule j of this lane; lanes
// with !active take no part.  Returns with v[] filled; after a time-out (recorded in *status) it gives up at once.
template <int N, class Addr>
__device__ __forceinline__ void cb_sweep(Addr addr, unsigned tag, bool active, float (&v)[N], bool &dead, gu32 *status,
                                         unsigned spin_limit) {
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
    if (spins >= spin_limit) {
      dead = true;
      __hip_atomic_store(status, 1u, CB_RLX_AGENT);
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// one 3x3 layer for ONE cout tile: acc[xi] (+)= U_xi * V_xi over NC k-steps, then the output transform
// (wino_layer of mvsn_chain_wino.hip with the cout-tile dimension dealt to the waves).  The pinned order -- transform,
// then the eight multiplies back to back -- is the measured best: interleaving the next k-step's transform with the
// multiplies (one wave per SIMD has nobody else to fill the pipe) took 24.2 us per step against 22.8 (HISTORY 3.6).
// HSEL = -1: both transform-row halves (y = half 0's outputs + half 1's); 0 / 1: that half alone (y = its outputs; the
// caller adds the two waves' results in the same order, so the sum is bit for bit the one-wave form's).
template <int NC, int CSA, int RS, int HSEL = -1>
__device__ __forceinline__ void band_layer(const float *__restrict__ act, const float *__restrict__ U, int ct, int wb,
                                           int lane, float (&y)[4][4]) {
  const float *wbase = act + (lane >> 4) * CSA + wb;
  const float *ub = U + ct * 1024 + lane * 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (HSEL >= 0 && half != HSEL) continue;
    floatx4 acc[8];
    float d[2][3][4];
    floatx4 u[2][2];
    auto fetch = [&](int buf, int c4) {
      const float *wp = wbase + c4 * 4 * CSA + half * RS;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float2 lo = *reinterpret_cast<const float2 *>(wp + i * RS);
        const float2 hi = *reinterpret_cast<const float2 *>(wp + i * RS + 2);
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
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int xq = 0; xq < 2; ++xq)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const floatx4 c0 = c4 == 0 ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[xq * 4 + j];
          acc[xq * 4 + j] = mfma16x16x4(u[cur][xq][j], v[xq * 4 + j], c0);
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
      if (half == 0 || HSEL == 1) y[r][0] = y0, y[r][1] = y1, y[r][2] = y2, y[r][3] = y3;
      else y[r][0] += y0, y[r][1] += y1, y[r][2] += y2, y[r][3] += y3;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

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

template <class GEO, bool C16 = false>   // C16: the cost volume stored as bf16 (ChainArgs::cost_bf16, the bf16 feature tier)
__global__ __launch_bounds__(CB_THREADS) void chain_band_kernel(ChainArgs a, int flags, MVSN_VIS10) {   // (mvsn_common.h)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int rows = GEO::rows, cols = GEO::cols, P = GEO::P, RS = GEO::RS, BR = GEO::BR, G = GEO::G, W = GEO::W;
  constexpr int CSA = GEO::CSA, CSW = GEO::CSW, HI = GEO::HITEMS;
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int nl = blockIdx.x / G, m = blockIdx.x % G;   // chain within this pass (workspace), band
  const int n = a.chain0 + nl;                        // chain of the call (data)
  constexpr bool HS = GEO::HS;
  const int hsel = wave >> 1;                        // half split: this wave's transform-row half (owner of the patches: 0)
  const int pt = HS ? 0 : wave >> 1, ct = wave & 1;
  const int lo = m * BR, hi = lo + BR - 1, wlo = lo - W;
  const int D = a.D;
  int tid = tid0;
  // test hooks (mvsn_debug_set_band_flags): bits 8.. = log2 of the spin limit; bit 1 = the chain's last band never runs,
  // as if the dispatcher had not found it a CU (what a shared device can do to a launch that needs co-residency)
  const unsigned spin_limit = (flags >> 8) ? 1u << ((flags >> 8) & 31) : CB_SPIN_LIMIT;
  if ((flags & 2) && m == G - 1) return;

  float *U = smem;
  float *sparams = U + CW_U0_FLOATS;
  float *red = sparams + CH_SP_FLOATS;                       // [band][wave][4]: the hand-off's partial sums
  float *gstat = red + GEO::RED;                             // [layer 2][group 4][mean, rstd]: the last statistics
  int *range = reinterpret_cast<int *>(gstat + 32);          // [parity][min y0, max y1]
  float *maskb = gstat + 32 + 16;
  float *tabw = maskb + GEO::MASK;                           // [pixel][4 weights]
  int *tabi = reinterpret_cast<int *>(tabw + GEO::TABW);     // [pixel][y0 * RS + x0 + 1, row step]
  float *act = tabw + GEO::TABW + GEO::TABI;
  float *win = act + 36 * CSA;
  float *comb = win + 32 * CSW;                              // (half split only)

  gu64 *ws = (gu64 *)(reinterpret_cast<u64 *>(a.workspace) + (size_t)nl * GEO::CHAIN_U64);
  gu64 *Fg = ws + GEO::FG, *Rg = ws + GEO::RG, *Sg = ws + GEO::SG;
  gu32 *status = (gu32 *)(reinterpret_cast<u64 *>(a.workspace) + (size_t)a.ws_chains * GEO::CHAIN_U64);
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
  for (int i = tid; i < 36 * CSA + 32 * CSW; i += CB_THREADS) act[i] = 0.0f;
  for (int i = tid; i < CH_SP_FLOATS; i += CB_THREADS) sparams[i] = a.packed[CH_W0_FLOATS + 2 * CH_W1_FLOATS + i];
  if (tid < 32) gstat[tid] = 0.0f;
  if (tid < 4) range[tid] = (tid & 1) ? -1 : 1 << 20;
  const float *bias0 = sparams, *gn0w = sparams + 32, *gn0b = sparams + 64, *bias1 = sparams + 96,
              *gn1w = sparams + 128, *gn1b = sparams + 160, *bias2 = sparams + 192;

  // this lane's patch: patch q of the band (row-major), outputs (lo + 2 prow + a, 2 pc + b), e = a*2 + b
  const int k = lane >> 4;
  const int q = pt * 16 + (lane & 15);
  const bool qvalid = q < GEO::NPATCH;
  const bool pvalid = qvalid;
  const bool sum_owner = qvalid && (!HS || hsel == 0);  // whose GroupNorm partial sums count (one wave per patch tile)
  // pixel rows of a 2x2 patch this wave gathers / publishes / normalises / emits (wave-uniform): both, or row hsel
  const bool row0 = !HS || hsel == 0, row1 = !HS || hsel == 1;
  auto mine = [&](int e) { return (e >> 1) ? row1 : row0; };
  const int qq = qvalid ? q : 0;
  const int prow = qq / GEO::PCOLS, pc = qq - prow * GEO::PCOLS;
  const bool tile_live = HS || pt * 16 < GEO::NPATCH;    // wave-uniform
  const int wb = (2 * prow) * RS + 2 * pc;            // window origin inside an act plane (local row 0 = image row lo-1)
  const int ob = wb + RS + 1;                         // output (0,0)
  const int cbase = ct * 16 + k * 4;                  // this lane's couts: cbase + r
  const int gown = ct * 2 + (k >> 1);                 // their GroupNorm group
  const int py0 = lo + 2 * prow, px0 = 2 * pc;
  const bool top_pub = pvalid && row0 && prow == 0 && m > 0;                  // this patch's first pixel row faces band m - 1
  const bool bot_pub = pvalid && row1 && prow == GEO::PROWS - 1 && m < G - 1;   // its second pixel row faces band m + 1
  const float inv_n = 1.0f / (8.0f * (float)P);
  // halo role: item i = tid + 256 it -> (side hs: 0 = row lo-1, 1 = row hi+1; channels 8 hcg .. + 7; column hx)
  int hs[HI], hx[HI], hcg[HI], hoff[HI];
  bool hvalid[HI];
#pragma unroll
  for (int it = 0; it < HI; ++it) {
    const int i = tid0 + CB_THREADS * it;
    const bool in = i < 8 * cols;
    const int ii = in ? i : 0;
    hs[it] = ii / (4 * cols);
    hcg[it] = (ii - hs[it] * 4 * cols) / cols;
    hx[it] = ii % cols;
    const int hy = hs[it] ? hi + 1 : lo - 1;
    hvalid[it] = in && hy >= 0 && hy < rows;           // the image has such a row <=> that neighbour exists
    hoff[it] = (hs[it] ? BR + 1 : 0) * RS + hx[it] + 1;
  }
  // image role: threads 0 .. EXT-1 -> one pixel of rows lo-1 .. hi+1
  const int er = tid0 / cols, iy = lo - 1 + er, ixx = tid0 - er * cols;
  const bool ivalid = tid0 < GEO::EXT && iy >= 0 && iy < rows;
  const bool iband = ivalid && iy >= lo && iy <= hi;
  __syncthreads();

  const float *Hn = a.H + (size_t)n * D * 9;
  const float *Hin = a.Hinc + (size_t)n * D * 9;
  const float *src = a.src + (size_t)n * 3 * P;
  uint8_t *maskg = a.mask + (size_t)n * D * P;
  typedef typename ChainCost<C16>::type cost_t;
  cost_t *costg = reinterpret_cast<cost_t *>(a.cost) + (size_t)n * 32 * D * P;
  float *fvolg = a.fvol ? a.fvol + (size_t)n * 32 * D * P : nullptr;
  const float *flp = a.fl + (size_t)(n % a.B) * 32 * P;
  const float *fl_lane = flp + (size_t)cbase * P + py0 * cols + px0;
  int slice_off = (cbase * D) * P + py0 * cols + px0;

  // plane d's features of this lane (f[r][e]) -> own rows of the gather window, the granules the other bands
  // gather from (tag d + 1), and the cost slice (not mask) * |left - right| straight from the registers
  auto emit = [&](int d, const float (&f)[4][4], const float2 (&fl)[4][2]) {
    if (!pvalid) return;
    const float2 m0 = *reinterpret_cast<const float2 *>(maskb + (2 * prow) * cols + px0);
    const float2 m1 = *reinterpret_cast<const float2 *>(maskb + (2 * prow + 1) * cols + px0);
    const bool out[4] = {m0.x != 0.0f, m0.y != 0.0f, m1.x != 0.0f, m1.y != 0.0f};
    cost_t *cd = costg + (size_t)d * P;
    float *fd = fvolg ? fvolg + (size_t)d * P : nullptr;
    // the granules first: the other bands wait for them, everything else of the epilogue travels meanwhile
    if (d + 1 < D) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gu64 *g = Fg + (size_t)(cbase + r) * P + py0 * cols + px0;
        if (row0) cb_publish2(g, d + 1, f[r][0], f[r][1]);
        if (row1) cb_publish2(g + cols, d + 1, f[r][2], f[r][3]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float *dst = win + (cbase + r) * CSW + (W + 2 * prow) * RS + px0 + 1;
      if (row0) dst[0] = f[r][0], dst[1] = f[r][1];
      if (row1) dst[RS] = f[r][2], dst[RS + 1] = f[r][3];
      cost_t *cdst = cd + (r * D) * P + slice_off;
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2) {
        if (!(a2 ? row1 : row0)) continue;
        float2v c2;
        c2.x = out[a2 * 2] ? 0.0f : fabsf(fl[r][a2].x - f[r][a2 * 2]);
        c2.y = out[a2 * 2 + 1] ? 0.0f : fabsf(fl[r][a2].y - f[r][a2 * 2 + 1]);
        chain_cost_nt(cdst + a2 * cols, c2);
      }
      if (fd) {
        float *fdst = fd + (r * D) * P + slice_off;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
          if (!(a2 ? row1 : row0)) continue;
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
      for (int a2 = 0; a2 < 2; ++a2)
        if (a2 ? row1 : row0) fl[r][a2] = *reinterpret_cast<const float2 *>(fl_lane + (size_t)r * P + a2 * cols);
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
    float2 fl[4][2] = {};
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
  // lo-1 .. hi+1 with the band's mask (A1: global gathers from the coarse source image, L1 / L2 hits) and where the
  // incremental homography sends each of those pixels (A2: bilinear weights, offset of the top-left tap, row step; and
  // the rows of the previous plane the band's gathers touch) -- is computed one step AHEAD, once per pixel, inside the
  // wait of the previous step's second hand-off (between publishing and the first poll), where the workgroup would idle.
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
      intx2 i2;
      w4[0] = keep * b.w00, w4[1] = keep * b.w01, w4[2] = keep * b.w10, w4[3] = keep * b.w11;
      i2[0] = b.y0 * RS + b.x0 + 1;      // (relative to image row 0: never negative; the window offset is this - wlo * RS)
      i2[1] = (b.y1 - b.y0) * RS;        // 0 where the +1 row is clamped (its weight is exactly zero there)
      *reinterpret_cast<floatx4 *>(tabw + tid0 * 4) = w4;
      *reinterpret_cast<intx2 *>(tabi + tid0 * 2) = i2;
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
    const bool fast = !(flags & 1) && need_lo >= wlo && need_hi <= wlo + GEO::WSLOTS - 1;   // workgroup-uniform

    float fp[4][4] = {}, hv[HI][8];   // (half split: only the wave's own pixel row is gathered, the rest stays zero)
    float tw[4 + HI][4];
    int to[4 + HI], tdy[4 + HI];
    auto footprints = [&]() {   // this thread's pixels: own 2x2 patch (ext rows 1 + 2 prow + a) and its halo pixels
#pragma unroll
      for (int e = 0; e < 4 + HI; ++e) {
        const int pix = e < 4 ? (1 + 2 * prow + (e >> 1)) * cols + px0 + (e & 1)
                              : (hs[e < 4 ? 0 : e - 4] ? BR + 1 : 0) * cols + hx[e < 4 ? 0 : e - 4];
        const floatx4 w4 = *reinterpret_cast<const floatx4 *>(tabw + pix * 4);
        const intx2 i2 = *reinterpret_cast<const intx2 *>(tabi + pix * 2);
        tw[e][0] = w4[0], tw[e][1] = w4[1], tw[e][2] = w4[2], tw[e][3] = w4[3];
        to[e] = i2[0], tdy[e] = i2[1];
      }
    };
    if (fast) {
      // E1 (consume): rows of F_{d-1} the gathers need from the other bands -> window.  A row = 32 x cols granules,
      // item i = tid + 256 j -> channel i / cols, column i % cols; all loads in flight before the first tag is looked at.
      {
        float v[GEO::NOWN][GEO::FROW];
        int slot_of[GEO::NOWN];
        bool need[GEO::NOWN];
#pragma unroll
        for (int s = 0; s < GEO::NOWN; ++s) {
          slot_of[s] = s < W ? s : s + BR;                       // the window slots above and below the own band
          const int row = wlo + slot_of[s];
          need[s] = row >= need_lo && row <= need_hi && row >= 0 && row < rows;
        }
        for (unsigned spins = 0;; ++spins) {
          bool ok = true;
#pragma unroll
          for (int s = 0; s < GEO::NOWN; ++s)
            if (need[s]) {
              const gu64 *g = Fg + (size_t)(wlo + slot_of[s]) * cols;
#pragma unroll
              for (int j = 0; j < GEO::FROW; ++j) {
                const int i = tid + CB_THREADS * j, c = i / cols, x = i - c * cols;
                const u64 xv = __hip_atomic_load(g + (size_t)c * P + x, CB_RLX_AGENT);
                v[s][j] = __builtin_bit_cast(float, (unsigned)xv);
                ok &= (unsigned)(xv >> 32) == (unsigned)d;
              }
            }
          if (__all(ok) || dead) break;
          if (spins >= spin_limit) {
            dead = true;
            __hip_atomic_store(status, 2u, CB_RLX_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int s = 0; s < GEO::NOWN; ++s)
          if (need[s]) {
#pragma unroll
            for (int j = 0; j < GEO::FROW; ++j) {
              const int i = tid + CB_THREADS * j, c = i / cols, x = i - c * cols;
              win[c * CSW + slot_of[s] * RS + x + 1] = v[s][j];
            }
          }
      }
      footprints();
      CB_STAMP(2);
      cb_barrier();   // B2: window complete
      // A2 gather.  The +1 column tap is read unclamped: where the clamp would act its weight is exactly zero and the
      // slot read is the zero halo column; the +1 row tap re-reads row y0 there (weight exactly zero as well).
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!mine(e)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float *fc = win + (cbase + r) * CSW + (to[e] - wlo * RS);
          fp[r][e] = fc[0] * tw[e][0] + fc[1] * tw[e][1] + fc[tdy[e]] * tw[e][2] + fc[tdy[e] + 1] * tw[e][3];
        }
      }
#pragma unroll
      for (int it = 0; it < HI; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[it][j] = 0.0f;
        if (hvalid[it]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float *fc = win + (hcg[it] * 8 + j) * CSW + (to[4 + it] - wlo * RS);
            hv[it][j] = fc[0] * tw[4 + it][0] + fc[1] * tw[4 + it][1] + fc[tdy[4 + it]] * tw[4 + it][2] +
                        fc[tdy[4 + it] + 1] * tw[4 + it][3];
          }
        }
      }
    } else {
      // the gathers reach beyond the window (large inter-plane motion, or a window of +-1 row): every tap straight from
      // the granules
      footprints();
#pragma unroll
      for (int e = 0; e < 4 + HI; ++e) {
        if (e < 4 && !mine(e)) continue;   // (wave-uniform: the other wave of the pair gathers that pixel row)
        const int it = e < 4 ? 0 : e - 4;
        const bool act_e = e < 4 ? pvalid : hvalid[it];
        const int y0 = to[e] / RS, x0 = to[e] - y0 * RS - 1;
        const bool x1 = x0 + 1 < cols, y1 = tdy[e] != 0;   // clamped +1 taps: weight exactly zero, value unused
        constexpr int NCH_MAX = 8;
        const int nch = e < 4 ? 4 : 8;
        const int c0 = e < 4 ? cbase : hcg[it] * 8;
        float t00[NCH_MAX], t01[NCH_MAX], t10[NCH_MAX], t11[NCH_MAX];
        const gu64 *g = Fg + (size_t)c0 * P + (act_e ? y0 * cols + x0 : 0);
        cb_sweep<NCH_MAX>([&](int j) { return g + (size_t)(j < nch ? j : 0) * P; }, d, act_e, t00, dead, status, spin_limit);
        cb_sweep<NCH_MAX>([&](int j) { return g + (size_t)(j < nch ? j : 0) * P + (x1 ? 1 : 0); }, d, act_e, t01, dead, status, spin_limit);
        cb_sweep<NCH_MAX>([&](int j) { return g + (size_t)(j < nch ? j : 0) * P + (y1 ? cols : 0); }, d, act_e, t10, dead, status, spin_limit);
        cb_sweep<NCH_MAX>([&](int j) { return g + (size_t)(j < nch ? j : 0) * P + (y1 ? cols : 0) + (x1 ? 1 : 0); }, d, act_e,
                          t11, dead, status, spin_limit);
#pragma unroll
        for (int j = 0; j < NCH_MAX; ++j) {
          const float v01 = x1 ? t01[j] : 0.0f, v10 = y1 ? t10[j] : 0.0f, v11 = (x1 && y1) ? t11[j] : 0.0f;
          const float val = t00[j] * tw[e][0] + v01 * tw[e][1] + v10 * tw[e][2] + v11 * tw[e][3];
          if (e < 4) {
            if (j < 4) fp[j][e] = act_e ? val : 0.0f;
          } else {
            hv[it][j] = act_e ? val : 0.0f;
          }
        }
      }
      cb_barrier();   // (keeps the barrier count of the two paths equal)
    }
    if (tid == 0) range[par * 2] = 1 << 20, range[par * 2 + 1] = -1;   // (read by everyone before B2; next use: step d+2)
    CB_STAMP(3);

    // A3: lay out the refiner input [image(3) | moved features(32)] on rows lo-1 .. hi+1
    if (pvalid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float *dst = act + (3 + cbase + r) * CSA + ob;
        if (row0) dst[0] = fp[r][0], dst[1] = fp[r][1];
        if (row1) dst[RS] = fp[r][2], dst[RS + 1] = fp[r][3];
      }
    }
#pragma unroll
    for (int it = 0; it < HI; ++it)
      if (hvalid[it]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) act[(3 + hcg[it] * 8 + j) * CSA + hoff[it]] = hv[it][j];
      }
    if (ivalid) {
      const int o = er * RS + ixx + 1;
      act[0 * CSA + o] = img[0];
      act[1 * CSA + o] = img[1];
      act[2 * CSA + o] = img[2];
      if (iband) {
        maskb[(iy - lo) * cols + ixx] = mk;
        maskg[(size_t)d * P + iy * cols + ixx] = mk != 0.0f ? 1 : 0;
      }
    }
    dma_landed();   // conv0's U
    cb_barrier();   // B3
    CB_STAMP(4);

    float y[4][4] = {};
    // half split: wave (h, ct) runs transform-row half h of the layer; half 1's outputs reach the owner through LDS
    // (written before, read after the barrier that follows every layer anyway) and are added in the one-wave order
    auto layer = [&](auto nc) {
      constexpr int NC = decltype(nc)::value;
      if constexpr (HS) {
        if (hsel == 0) band_layer<NC, CSA, RS, 0>(act, U, ct, wb, lane, y);
        else band_layer<NC, CSA, RS, 1>(act, U, ct, wb, lane, y);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<floatx4 *>(comb + (((hsel * 2 + ct) * 4 + r) * 64 + lane) * 4) =
              floatx4{y[r][0], y[r][1], y[r][2], y[r][3]};
      } else {
        if (tile_live) band_layer<NC, CSA, RS>(act, U, ct, wb, lane, y);
      }
    };
    auto combine = [&]() {
      if constexpr (HS) {   // (a + b == b + a bit for bit: both waves of the pair hold half 0's outputs + half 1's)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const floatx4 o = *reinterpret_cast<const floatx4 *>(comb + ((((hsel ^ 1) * 2 + ct) * 4 + r) * 64 + lane) * 4);
          y[r][0] += o[0], y[r][1] += o[1], y[r][2] += o[2], y[r][3] += o[3];
        }
      }
    };
    layer(std::integral_constant<int, 9>{});
    CB_STAMP(5);
    cb_barrier();   // B4: act and U free
    dma_u(upk + CW_U0_FLOATS, 8);
    combine();

    // E2 / E3: bias, partial GroupNorm sums (shifted by the previous step's mean, as chain_wino_kernel), publish
    // them with the band's boundary rows; collect the other bands'; normalise + activate own outputs and halo rows
    auto exchange = [&](int layer, const float *bias, const float *gamma, const float *beta, bool residual,
                        auto &&meanwhile) {
      float *gs = gstat + layer * 8;                  // [group][mean, rstd] of the previous step: the shift
      const float shift = gs[gown * 2];
      float hshift[HI];                               // (few bands: every thread forms the statistics it needs itself)
#pragma unroll
      for (int it = 0; it < HI; ++it) hshift[it] = gs[hcg[it] * 2];
      float s[2] = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float b = bias[cbase + r];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[r][e] += b;
          const float dv = y[r][e] - shift;
          s[0] += dv;
          s[1] += dv * dv;
        }
      }
      if (!sum_owner) s[0] = s[1] = 0.f;
      cb_half_wave_sums(s);
      gu64 *Sl = Sg + (size_t)layer * (G * CB_WAVES * 4);
      if ((lane & 31) == 16) {
        gu64 *g = Sl + (m * CB_WAVES + wave) * 4 + (lane >> 5) * 2;
        cb_publish2(g, d, s[0], s[1]);
      }
      // boundary rows: a patch of the band's first patch row publishes its first pixel row (e = 0, 1) for band m - 1,
      // one of the last patch row its second (e = 2, 3) for band m + 1 (a 2-row band: both)
      gu64 *Rl = Rg + (size_t)layer * (G * 2 * 32 * cols);
      if (top_pub) {
        gu64 *g = Rl + ((size_t)(m * 2 + 0) * 32 + cbase) * cols + px0;
#pragma unroll
        for (int r = 0; r < 4; ++r) cb_publish2(g + r * cols, d, y[r][0], y[r][1]);
      }
      if (bot_pub) {
        gu64 *g = Rl + ((size_t)(m * 2 + 1) * 32 + cbase) * cols + px0;
#pragma unroll
        for (int r = 0; r < 4; ++r) cb_publish2(g + r * cols, d, y[r][2], y[r][3]);
      }
      CB_STAMP(16 + layer * 4);
      meanwhile();   // work that needs none of the hand-off, placed where the workgroup would otherwise only wait
      CB_STAMP(17 + layer * 4);
      // collect, ONE sweep: every thread its halo granules, threads 0 .. 16 G - 1 also one sum granule each
      float hr[HI][8];
      {
        float sv = 0.f;
        const bool sum_role = tid0 < GEO::RED;
        for (unsigned spins = 0;; ++spins) {
          bool ok = true;
          if (sum_role) {
            const u64 x = __hip_atomic_load(Sl + tid0, CB_RLX_AGENT);
            sv = __builtin_bit_cast(float, (unsigned)x);
            ok &= (unsigned)(x >> 32) == (unsigned)d;
          }
#pragma unroll
          for (int it = 0; it < HI; ++it)
            if (hvalid[it]) {
              // the neighbour's row facing this band: its last row (side 1) for our row lo-1, its first (side 0) for hi+1
              const gu64 *g = Rl + ((size_t)((hs[it] ? m + 1 : m - 1) * 2 + (hs[it] ? 0 : 1)) * 32 + hcg[it] * 8) * cols + hx[it];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const u64 x = __hip_atomic_load(g + j * cols, CB_RLX_AGENT);
                hr[it][j] = __builtin_bit_cast(float, (unsigned)x);
                ok &= (unsigned)(x >> 32) == (unsigned)d;
              }
            }
          if (__all(ok) || dead) break;
          if (spins >= spin_limit) {
            dead = true;
            __hip_atomic_store(status, 3u + layer, CB_RLX_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (sum_role) red[tid0] = sv;
      }
      CB_STAMP(18 + layer * 4);
      cb_barrier();
      // totals in a fixed order: every workgroup of the chain forms the same statistics bit for bit
      float own_mean, own_rstd, h_mean[HI], h_rstd[HI];
      if constexpr (G <= 4 || HS) {
        // few bands: each thread adds the 2 G records of the groups it needs itself (band-major), no second barrier.
        // Half split: G owner records per group -- the 2-row band m is patch tile m % 2 of the 4-row band m / 2, so
        // the records AND their order are those of the 4-band geometry: the same statistics bit for bit.
        auto stats_of = [&](int g, float sh, float &mean, float &rstd) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int mm = 0; mm < G; ++mm)
#pragma unroll
            for (int pp = 0; pp < (HS ? 1 : 2); ++pp) {
              const float2 rec = *reinterpret_cast<const float2 *>(red + (mm * CB_WAVES + pp * 2 + (g >> 1)) * 4 + (g & 1) * 2);
              s1 += rec.x;
              s2 += rec.y;
            }
          const float ms = s1 * inv_n;
          const float var = fmaxf(s2 * inv_n - ms * ms, 0.0f);
          mean = sh + ms;
          rstd = 1.0f / sqrtf(var + CB_GN_EPS);
        };
        stats_of(gown, shift, own_mean, own_rstd);
#pragma unroll
        for (int it = 0; it < HI; ++it) stats_of(hcg[it], hshift[it], h_mean[it], h_rstd[it]);
        // (the next step's shift; nobody reads gs again before the barriers that follow)
        if (hsel == 0 && (lane & 31) == 0) gs[gown * 2] = own_mean;   // waves 0, 1 cover the four groups
      } else {
        // many bands: wave 0, one 16-lane row per group: lane `sub` adds the records of bands sub, sub + 16, ..., four
        // DPP steps add the row; everyone reads the result behind a second barrier
        if (wave == 0) {
          const int g = lane >> 4, sub = lane & 15;
          float s2v[2] = {0.f, 0.f};
          for (int mm = sub; mm < G; mm += 16)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
              const float2 rec = *reinterpret_cast<const float2 *>(red + (mm * CB_WAVES + pp * 2 + (g >> 1)) * 4 + (g & 1) * 2);
              s2v[0] += rec.x;
              s2v[1] += rec.y;
            }
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            s2v[q2] += dpp_mov<0xB1>(s2v[q2]);
            s2v[q2] += dpp_mov<0x4E>(s2v[q2]);
            s2v[q2] += dpp_mov<0x141>(s2v[q2]);
            s2v[q2] += dpp_mov<0x140>(s2v[q2]);
          }
          if (sub == 0) {
            const float ms = s2v[0] * inv_n;
            const float var = fmaxf(s2v[1] * inv_n - ms * ms, 0.0f);
            gs[g * 2] = gs[g * 2] + ms;                         // mean = shift + E[x - shift]
            gs[g * 2 + 1] = 1.0f / sqrtf(var + CB_GN_EPS);
          }
        }
        cb_barrier();
        own_mean = gs[gown * 2], own_rstd = gs[gown * 2 + 1];
#pragma unroll
        for (int it = 0; it < HI; ++it) h_mean[it] = gs[hcg[it] * 2], h_rstd[it] = gs[hcg[it] * 2 + 1];
      }
      CB_STAMP(19 + layer * 4);
      if (pvalid) {
        const float mean = own_mean, rstd = own_rstd;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = cbase + r;
          const float scl = rstd * gamma[c];
          const float sft = beta[c] - mean * scl;
          float *dst = act + c * CSA + ob;
          if (residual) {
            if (row0) dst[0] += lrelu02(y[r][0] * scl + sft), dst[1] += lrelu02(y[r][1] * scl + sft);
            if (row1) dst[RS] += lrelu02(y[r][2] * scl + sft), dst[RS + 1] += lrelu02(y[r][3] * scl + sft);
          } else {
            if (row0) dst[0] = lrelu02(y[r][0] * scl + sft), dst[1] = lrelu02(y[r][1] * scl + sft);
            if (row1) dst[RS] = lrelu02(y[r][2] * scl + sft), dst[RS + 1] = lrelu02(y[r][3] * scl + sft);
          }
        }
      }
#pragma unroll
      for (int it = 0; it < HI; ++it)
        if (hvalid[it]) {
          const float mean = h_mean[it], rstd = h_rstd[it];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = hcg[it] * 8 + j;
            const float scl = rstd * gamma[c];
            const float sft = beta[c] - mean * scl;
            const float v = lrelu02(hr[it][j] * scl + sft);
            if (residual) act[c * CSA + hoff[it]] += v;
            else act[c * CSA + hoff[it]] = v;
          }
        }
    };
    exchange(0, bias0, gn0w, gn0b, false, [] {});
    CB_STAMP(6);
    dma_landed();
    cb_barrier();   // B6
    CB_STAMP(7);

    layer(std::integral_constant<int, 8>{});
    CB_STAMP(8);
    cb_barrier();   // B7
    dma_u(upk + CW_U0_FLOATS + CW_U1_FLOATS, 8);
    combine();
    exchange(1, bias1, gn1w, gn1b, true, [&] {   // x2 = x1 + LReLU(GN(conv1(x1)))
      if (d + 1 < D) prepare(d + 1);
    });
    CB_STAMP(9);
    dma_landed();
    cb_barrier();   // B10
    CB_STAMP(10);

    layer(std::integral_constant<int, 8>{});
    float2 fl[4][2] = {};
    load_left(fl);
    CB_STAMP(11);
    cb_barrier();   // B11
    dma_u(upk, 9);  // conv0 of the next step
    combine();

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
  // A hand-off that timed out (the chain's workgroups were not co-resident: the device is shared with other work) leaves
  // wrong numbers behind.  Besides the status word, make that impossible to miss without a host round trip: one NaN in
  // the cost slice turns the regulariser's GroupNorm statistics -- and with them the whole depth map -- into NaN.
  // (written by the thread that owns the element: program order puts it behind the last step's own store)
  if (__syncthreads_or(dead) && pvalid && row0) chain_cost_st(costg + ((size_t)(D - 1) * P + slice_off), __builtin_nanf(""));
}

// Zero fill of the hand-off workspace (granules + status word) as a plain kernel: inside a captured graph a
// hipMemsetAsync node of a few bytes at an offset pointer was observed to leave pointer-like garbage behind on replay
// (ROCm 7.2: the status word read 0x7890... after every graph launch, never after a stream launch); a kernel node has no
// such surprises and costs the same launch.
__global__ __launch_bounds__(256) void band_zero_kernel(u64 *__restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
}

// ---- host side -------------------------------------------------------------------------------------------------
struct BandPlan {
  int G, threads, slot;   // workgroups per chain, threads per workgroup, LDS opt-in slot of the kernel
  size_t chain_u64, lds_bytes;
  void (*kernel)(ChainArgs, int, MVSN_VIS10);
  void (*kernel16)(ChainArgs, int, MVSN_VIS10);   // ... storing the cost volume as bf16 (ChainArgs::cost_bf16)
};

template <class GEO>
static BandPlan band_plan_of(int slot) {
  return BandPlan{GEO::G, CB_THREADS, slot, GEO::CHAIN_U64, (size_t)GEO::LDS_FLOATS * sizeof(float), chain_band_kernel<GEO, false>,
                  chain_band_kernel<GEO, true>};
}

static int g_band_debug_flags = 0;
void chain_band_debug_flags(int flags) { g_band_debug_flags = flags; }

// 16x32 has two plans: 8 bands of 2 rows (half split) while that many workgroups per chain fit the chip in one pass,
// 4 bands of 4 rows beyond (bit-identical results: same arithmetic per output, same GroupNorm records in the same
// order).  Debug flag bit 2 pins the 4-band plan (A/B, tests).
// 30x40 / 32x64: the thin-band plan (15 / 16 workgroups per chain) while the chains fit ONE of its passes (17 / 16 chains
// on 256 CUs: 2.4 / 4.0 ms per pass at D = 96 / 128); beyond that the SLAB plan (mvsn_chain_slab.hip: 3 / 4 fat bands
// per chain, 85 / 64 chains per pass of 4.4 / 5.9 ms -- less than two thin passes).
// Debug flag bit 4 (16) pins the slab plan (also on 16x32, where it exists for the tests only), bit 5 (32) the thin one.
static bool band_plan(int rows, int cols, int n_chains, BandPlan *p) {
  const bool known = (rows == 16 && cols == 32) || (rows == 30 && cols == 40) || (rows == 32 && cols == 64);
  if (!known) return false;
  const int thin_g = rows == 16 ? Band16x32::G : (rows == 30 ? Band30x40::G : Band32x64::G);
  const bool many = rows != 16 && n_chains > device_cus() / thin_g;
  if (((g_band_debug_flags & 16) || many) && !(g_band_debug_flags & 32)) {
    SlabPlan sp;
    if (chain_slab_plan(rows, cols, &sp)) {
      *p = BandPlan{sp.G, sp.threads, rows == 16 ? 4 : (rows == 30 ? 5 : 6), sp.chain_u64, sp.lds_bytes, sp.kernel, sp.kernel16};
      return true;
    }
  }
  if (rows == 16) {
    if (!(g_band_debug_flags & 4) && n_chains <= device_cus() / Band16x32H::G) *p = band_plan_of<Band16x32H>(3);
    else *p = band_plan_of<Band16x32>(0);
  } else if (rows == 30) *p = band_plan_of<Band30x40>(1);
  else *p = band_plan_of<Band32x64>(2);
  return true;
}

bool chain_band_supported(int rows, int cols) {
  BandPlan p;
  return band_plan(rows, cols, 1 << 20, &p);
}

int chain_band_groups(int n_chains, int rows, int cols) {
  BandPlan p;
  return band_plan(rows, cols, n_chains, &p) ? p.G : 0;
}

// chains per pass of the THIN-band plan: every workgroup of a pass must be co-resident (one per CU).  (16x32: of the
// 4-band plan, the largest number a single pass can take on that grid.)
int chain_band_chains_per_pass(int rows, int cols) {
  if (rows == 16 && cols == 32) return device_cus() / Band16x32::G;
  if (rows == 30 && cols == 40) return device_cus() / Band30x40::G;
  if (rows == 32 && cols == 64) return device_cus() / Band32x64::G;
  return 0;
}

// How a call's chains are dealt to passes.  Every workgroup of a pass must be co-resident (one per CU): `per` chains per
// pass of the main plan, equal passes -- 85 + 85 + 85 + 1 chains would cost a whole pass for the last one.  Exception
// (slab plans): a remainder small enough for ONE pass of the thin-band plan (256 chains on 30x40: 3 x 85 slabs + 1 chain
// on 15 thin bands, 13.1 + 2.5 ms, against 4 passes of 64 = 17.5 ms) runs as that thin pass behind the full slab passes.
// The thin pass's granules lie at the END of the main plan's workspace, so that both plans find the status block at
// workspace + ws_chains * CHAIN_U64 (their own ws_chains, their own CHAIN_U64).  Debug flag 16 (slab pinned) keeps the
// equal passes (A/B, tests).
struct BandSchedule {
  BandPlan main, tail;
  int per, n_main, n_tail;       // chains per main pass; chains of the main passes / of the thin tail pass
  size_t status_u64, tail_u64;   // offset of the status block / of the tail pass's granules (u64 units)
};

static bool band_schedule(int rows, int cols, int n_chains, BandSchedule *s) {
  if (!band_plan(rows, cols, n_chains, &s->main)) return false;
  const int cap = device_cus() / s->main.G;
  s->n_main = n_chains, s->n_tail = 0, s->per = n_chains, s->tail_u64 = 0;
  if (cap >= 1 && n_chains > cap) {
    const int passes = (n_chains + cap - 1) / cap;
    s->per = (n_chains + passes - 1) / passes;
    const int thin_g = rows == 16 ? 0 : (rows == 30 ? Band30x40::G : Band32x64::G);
    const int rem = n_chains % cap;
    if (thin_g && s->main.G < thin_g && !(g_band_debug_flags & 16) && rem > 0 && rem <= device_cus() / thin_g) {
      const BandPlan thin = rows == 30 ? band_plan_of<Band30x40>(1) : band_plan_of<Band32x64>(2);
      const size_t main_u64 = (size_t)cap * s->main.chain_u64, tail_need = (size_t)rem * thin.chain_u64;
      if (tail_need <= main_u64 && ((main_u64 - tail_need) & 1) == 0) {   // (16-byte publishes: an even u64 offset)
        s->tail = thin, s->per = cap, s->n_main = n_chains - rem, s->n_tail = rem, s->tail_u64 = main_u64 - tail_need;
      }
    }
  }
  s->status_u64 = (size_t)s->per * s->main.chain_u64;
  return true;
}

size_t chain_band_workspace_bytes(int n_chains, int rows, int cols) {
  BandSchedule s;
  return band_schedule(rows, cols, n_chains, &s) ? (s.status_u64 + 8) * sizeof(u64) : 0;
}

size_t chain_band_status_offset(int n_chains, int rows, int cols) {
  BandSchedule s;
  return band_schedule(rows, cols, n_chains, &s) ? s.status_u64 * sizeof(u64) : 0;
}

// status word behind the granules: 0 = every hand-off completed; otherwise the code of the hand-off that timed out.
// More chains than fit the chip at one workgroup per band run as consecutive passes over the same workspace (the
// chains are independent; within a pass every workgroup is co-resident).
int chain_band_launch(const ChainArgs &a, int n_chains, void *workspace, size_t workspace_bytes, int flags,
                      hipStream_t stream) {
  flags |= g_band_debug_flags;
  BandSchedule s;
  MVSN_REQUIRE(band_schedule(a.rows, a.cols, n_chains, &s), MVSN_E_TOOLARGE,
               "mvsn_incremental_cost_volume(banded): no plan for a %dx%d coarse grid", a.rows, a.cols);
  const size_t need = (s.status_u64 + 8) * sizeof(u64);
  MVSN_REQUIRE(workspace && workspace_bytes >= need, MVSN_E_WORKSPACE,
               "mvsn_incremental_cost_volume(banded): workspace of %zu bytes required", need);
  MVSN_REQUIRE(device_cus() / s.main.G >= 1, MVSN_E_TOOLARGE,
               "mvsn_incremental_cost_volume(banded): %d bands exceed the %d CUs", s.main.G, device_cus());
  static LdsOptIn opt[2][7];
  const int c16 = a.cost_bf16 ? 1 : 0;
  auto run_pass = [&](const BandPlan &p, int n0, int nn, u64 *ws, int ws_chains, bool with_status) -> int {
    const auto kern = c16 ? p.kernel16 : p.kernel;
    if (int rc = ensure_lds(opt[c16][p.slot], (const void *)kern, p.lds_bytes, "mvsn_incremental_cost_volume(banded)")) return rc;
    // every polled word starts from tag 0 (no step carries it): zeroed ahead of each pass -- with the first pass (which
    // fills the workspace) also the status block behind the granules, so that a later pass keeps what an earlier one
    // reported
    const size_t words = (size_t)nn * p.chain_u64 + (with_status ? 8 : 0);
    const unsigned blocks = (unsigned)((words + 1023) / 1024 < 1024 ? (words + 1023) / 1024 : 1024);
    hipLaunchKernelGGL(band_zero_kernel, dim3(blocks), dim3(256), 0, stream, ws, words);
    ChainArgs b = a;
    b.workspace = (float *)ws;
    b.chain0 = n0;
    b.ws_chains = ws_chains;
#ifdef MVSN_BAND_HIDE_PTRS   // A/B aid (see MVSN_VIS10): the buffers reach the kernel through the by-value struct only
    const ChainArgs hidden{};
    hipLaunchKernelGGL(kern, dim3(nn * p.G), dim3(p.threads), p.lds_bytes, stream, b, flags, CHAIN_VISIBLE(hidden));
#else
    hipLaunchKernelGGL(kern, dim3(nn * p.G), dim3(p.threads), p.lds_bytes, stream, b, flags, CHAIN_VISIBLE(b));
#endif
    return 0;
  };
  for (int n0 = 0; n0 < s.n_main; n0 += s.per) {
    const int nn = s.n_main - n0 < s.per ? s.n_main - n0 : s.per;   // (the first pass fills the workspace: nn == per)
    if (int rc = run_pass(s.main, n0, nn, (u64 *)workspace, s.per, n0 == 0)) return rc;
  }
  if (s.n_tail)
    if (int rc = run_pass(s.tail, s.n_main, s.n_tail, (u64 *)workspace + s.tail_u64, s.n_tail, false)) return rc;
  return check_launch("mvsn_incremental_cost_volume(banded)");
}

}  // namespace mvsn
