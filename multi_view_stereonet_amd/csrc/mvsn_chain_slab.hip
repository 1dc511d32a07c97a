// The fused incremental chain, SLAB plan of the banded form: one chain on a FEW workgroups, each a fat band of the
// coarse plane resident in its CU's LDS (see include/mvsn_hip.h: mvsn_incremental_cost_volume, form MVSN_CHAIN_BANDED
// with many chains in flight).
//
// The thin-band plan (mvsn_chain_band.hip: 2-row bands, 15 / 16 workgroups per chain) is the batch-1 form of the 30x40 and
// 32x64 coarse grids (640x480 / 1024x512 frames); with many chains in flight it needs one pass per 17 / 16 chains and
// the stepwise form (one plane per round of full-chip launches, the plane through HBM every step) took over.  Here a
// chain is cut into NB = 3 (30x40: 10 rows) / 4 (32x64: 8 rows) bands; workgroup m of a chain owns band m for the whole
// recurrence of multi_view_stereonet.py:279-290 exactly as chain_wino_kernel (mvsn_chain_wino.hip) owns a 16x32 plane:
// 512 threads, wave w = patch tile w (16 patches of 2x2 pixels) and both cout tiles, F(2x2,3x3) on
// v_mfma_f32_16x16x4_f32 with the input transform in registers, the band's 35 activation planes (rows lo-1 .. hi+1) in
// LDS, GroupNorm over the WHOLE plane, masks from the reference's fp32 expression order, cost slice from registers.
//
// What travels between the bands of a chain, per step d (tagged 8-byte granules {value, tag = d} written write-through
// and polled past L1 -- the thin-band plan's protocol, see mvsn_chain_band.hip; single buffers for the same reason):
//   E1   the two boundary rows of F_{d-1} (published by step d-1's epilogue): the bilinear gather of a band's own pixels
//        reaches one row into its neighbours;
//   E1b  the two boundary rows of the MOVED features (they are the neighbours' conv0 halo rows -- the neighbour has just
//        gathered them for itself; re-gathering them here would need F_{d-1} two rows deep);
//   E2   after conv0: per-wave GroupNorm partial sums + the two boundary rows of the raw output (normalised by the
//        reader once it has every band's sums);
//   E3   after conv1: the same for the residual block.
// The gather reads F_{d-1} from the activation planes themselves (rows lo-1 .. hi+1: the 32x64 band leaves LDS no room
// for a separate window).  A step whose incremental homography moves some pixel of the PLANE further than that (every
// workgroup evaluates the whole plane, one step ahead, so all bands of a chain agree) takes the slow path: the step
// before publishes all of F_{d-1} as granules and every tap is read from those (large inter-plane motion only).
//
// LDS (floats): sparams 224 | gstat 16 | flags 8 | maskb BR x cols |
//               U 16 x 1024 (16 of a layer's 16 / 18 half-k-step blocks of transformed weights; conv0's last two are read
//               from L2 into registers one k-step ahead -- 8 of the 800 loads of a step, no ring, no barrier) |
//               act 35 x CSA (plane 35, the K padding of conv0, is not stored: its lane re-reads plane 34 against zero
//               weights)
// 32x64: 65,536 + 896 + 96 + 2,048 + 35 x 672 x 4 = 162,656 bytes of the 163,840.
#include "mvsn_chain.h"
#include "mvsn_common.h"

namespace mvsn {

constexpr int SB_THREADS = 512, SB_WAVES = 8;
constexpr float SB_GN_EPS = 1e-5f;
constexpr unsigned SB_SPIN_LIMIT = 1u << 21;
#ifndef MVSN_SB_ABLATE   // tuning aid (wrong results): 1 no U DMA in the step loop, 2 no gather plan, 4 hand-off polls return at once
#define MVSN_SB_ABLATE 0
#endif
constexpr int SB_USLOTS = 16;                  // half-k-step blocks of U resident in LDS (4 KB each)

template <int ROWS, int COLS, int NB_, int CSA_>
struct SlabGeo {
  static constexpr int rows = ROWS, cols = COLS, NB = NB_, BR = ROWS / NB_, P = ROWS * COLS, RS = COLS + 2;
  static constexpr int ER = BR + 2, CSA = CSA_;                      // rows lo-1 .. hi+1; channel stride
  static constexpr int PCOLS = COLS / 2, PROWS = BR / 2, NPATCH = PROWS * PCOLS;
  static constexpr int EXT = ER * COLS;                              // pixels of the extended rows (image role)
  static constexpr int IMG_IT = (EXT + SB_THREADS - 1) / SB_THREADS;
  static constexpr int PLANE_IT = (P + SB_THREADS - 1) / SB_THREADS; // pixels per thread when the whole plane is walked
  static constexpr int HALO = 8 * COLS;                              // halo items (side, channel group, column)
  static constexpr int MASK = BR * COLS;
  static constexpr int LDS_FLOATS = SB_USLOTS * 1024 + CH_SP_FLOATS + 16 + 8 + MASK + 35 * CSA;
  // granule workspace of one chain (u64 units)
  static constexpr size_t ROWG = (size_t)NB_ * 2 * 32 * COLS;         // one boundary-row hand-off: [band][side][32 ch][cols]
  static constexpr size_t FG = 0;                                    // [32 ch][rows][cols] (slow path only)
  static constexpr size_t E1 = FG + 32 * (size_t)P, E1B = E1 + ROWG, E2 = E1B + ROWG, E3 = E2 + ROWG;
  static constexpr size_t SG = E3 + ROWG;                            // [layer 2][band][wave 8][group 4][2]
  static constexpr size_t CHAIN_U64 = SG + 2 * (size_t)NB_ * 64;
  static_assert(ROWS % NB_ == 0 && BR % 2 == 0 && COLS % 4 == 0, "bands of whole 2x2 patches, 16-byte rows");
  static_assert(NPATCH <= SB_WAVES * 16, "one patch tile per wave");
  static_assert(HALO <= SB_THREADS, "one halo item per thread");
  static_assert(NB_ * 64 <= SB_THREADS && NB_ <= 4, "sum granules: one wave per band");
  static_assert(CSA % 2 == 0 && CSA >= ER * RS, "activation planes: 8-byte aligned rows");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS plan");
};
// CSA = 32 (mod 64): the channels k and k + 1 of a k-step (lanes 0-15 / 16-31 of a ds_read_b64 group) hit disjoint banks
typedef SlabGeo<30, 40, 3, 544> Slab30x40;
typedef SlabGeo<32, 64, 4, 672> Slab32x64;
typedef SlabGeo<16, 32, 2, 352> Slab16x32;    // (tests: the plane-resident kernel's grid on two workgroups)

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef float sb_float2v __attribute__((ext_vector_type(2)));
typedef unsigned sb_uintx4 __attribute__((ext_vector_type(4)));
#define SB_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define SB_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define SB_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

__device__ __forceinline__ void sb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Two neighbouring granules in one 16-byte write-through store (see cb_publish2, mvsn_chain_band.hip), addressed as
// scalar base + 32-bit lane offset + immediate: no 64-bit address arithmetic in vector registers (with 512 threads a wave
// has 256 of them, and the addresses of a hand-off's sixteen stores, hoisted out of the step loop, were what spilled).
template <int OFF>
__device__ __forceinline__ void sb_publish2(const gu64 *base, unsigned voff_bytes, unsigned tag, float v0, float v1) {
  static_assert(OFF >= 0 && OFF < 4096, "13-bit signed immediate");
  sb_uintx4 q;
  q[0] = __builtin_bit_cast(unsigned, v0), q[1] = tag, q[2] = __builtin_bit_cast(unsigned, v1), q[3] = tag;
  asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 sc1\n\ts_nop 1" ::"v"(voff_bytes), "v"(q), "s"(base), "n"(OFF) : "memory");
}

// Re-read this lane's N granules base[off + j * STRIDE] until every tag in the wave matches (lanes with !active take no
// part).  After a time-out (recorded in *status) the wave gives up at once.  The loads are `global_load_dwordx2 v, v_off,
// s[base] offset:imm sc1` by hand: scalar base, one 32-bit lane offset, the granules of a lane by immediates (the compiler
// forms a 64-bit vector address per load of an agent-scope atomic and hoists them).  The wait that follows is tied to
// the loaded registers, so no use can be scheduled in front of it.
template <int J, int N, int STRIDE>
__device__ __forceinline__ void sb_issue(const gu64 *base, unsigned voff_bytes, u64 (&x)[N]) {
  if constexpr (J < N) {
    static_assert(J * STRIDE * 8 < 4096, "13-bit signed immediate");
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3 sc1" : "=v"(x[J]) : "v"(voff_bytes), "s"(base), "n"(J * STRIDE * 8) : "memory");
    sb_issue<J + 1, N, STRIDE>(base, voff_bytes, x);
  }
}
template <int N, int STRIDE>
__device__ __forceinline__ void sb_sweep(const gu64 *base, unsigned off, unsigned tag, bool active, float (&v)[N], bool &dead,
                                         gu32 *status, unsigned code, unsigned spin_limit) {
  const unsigned voff = off * 8u;
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
    if ((MVSN_SB_ABLATE & 4) && spins > 0) return;
    if (active) {
      u64 x[N];
      sb_issue<0, N, STRIDE>(base, voff, x);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < N; ++j) {
        asm volatile("" : "+v"(x[j]));
        v[j] = __builtin_bit_cast(float, (unsigned)x[j]);
        ok &= (unsigned)(x[j] >> 32) == tag;
      }
    }
    if (__all(ok) || dead) return;
    if (spins >= spin_limit) {
      dead = true;
      __hip_atomic_store(status, code, SB_RLX_AGENT);
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// One 3x3 layer (wino_layer of mvsn_chain_wino.hip; the U blocks by (transform-row half, k-step): block s = half * NC + c4
// sits in LDS slot s while s < SB_USLOTS, the others -- conv0's last two -- come from L2 straight into the registers the
// LDS reads would fill, fetched one k-step ahead like them).  `ug` = the layer's U in global memory (chain_wino layout:
// [k-step][cout tile][xi quad][lane][4]).
template <int NC, int CSA, int RS>
__device__ __forceinline__ void slab_layer(const float *__restrict__ act, const float *__restrict__ U,
                                           const float *__restrict__ ug, int wb, int lane, float (&y)[2][4][4]) {
  // conv0: channel 4 * 8 + 3 = 35 is the K padding (zero weights): its lane re-reads plane 34
  const float *wbase = act + (lane >> 4) * CSA + wb;
  const float *ub = U + lane * 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    floatx4 acc[2][8];
    float d[2][3][4];
    floatx4 u[2][4];
    auto fetch = [&](int buf, int c4) {
      const float *wp = wbase + c4 * 4 * CSA + half * RS;
      if constexpr (NC == 9) {
        if (c4 == 8) {   // (formed here, from an opaque copy of the lane id: as a fourth base address held across the layer it spilled)
          int kk = lane;
          asm volatile("" : "+v"(kk));
          wp -= (kk >> 4) == 3 ? CSA : 0;
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float2 lo = *reinterpret_cast<const float2 *>(wp + i * RS);
        const float2 hi = *reinterpret_cast<const float2 *>(wp + i * RS + 2);
        d[buf][i][0] = lo.x, d[buf][i][1] = lo.y, d[buf][i][2] = hi.x, d[buf][i][3] = hi.y;
      }
      const int s = half * NC + c4;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int xq = 0; xq < 2; ++xq) {
          if (s < SB_USLOTS) u[buf][ct * 2 + xq] = *reinterpret_cast<const floatx4 *>(ub + s * 1024 + (ct * 2 + xq) * 256);
          else   // (buffer descriptor + lane offset + scalar offset: the eight addresses are never vector registers)
            u[buf][ct * 2 + xq] = __builtin_bit_cast(
                floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                             __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ug), 0, NC * CW_UCHUNK * 4, 0x00020000),
                             lane * 16, ((c4 * 2 + ct) * 4 + half * 2 + xq) * 1024, 0));
        }
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

// sums of four values over the 32 lanes of each half-wave; totals in lanes 16..31 / 48..63 (chain_wino: half_wave_sums)
__device__ __forceinline__ void sb_half_wave_sums(float (&s)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] += dpp_mov<0xB1>(s[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] += dpp_mov<0x4E>(s[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] += dpp_mov<0x141>(s[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] += dpp_mov<0x140>(s[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    s[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s[k]), 0x142, 0xA, 0xF, false));
}

template <class GEO, bool C16 = false>   // C16: the cost volume stored as bf16 (ChainArgs::cost_bf16, the bf16 feature tier)
__global__ __launch_bounds__(SB_THREADS) void chain_slab_kernel(ChainArgs a, int flags, MVSN_VIS10) {   // (mvsn_common.h)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int rows = GEO::rows, cols = GEO::cols, P = GEO::P, RS = GEO::RS, BR = GEO::BR, NB = GEO::NB, ER = GEO::ER;
  constexpr int CSA = GEO::CSA, PCOLS = GEO::PCOLS, IMG_IT = GEO::IMG_IT;
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int nl = blockIdx.x / NB, m = blockIdx.x % NB;   // chain within this pass (workspace), band
  const int n = a.chain0 + nl;                          // chain of the call (data)
  const int lo = m * BR;
  const int D = a.D;
  int tid = tid0;
  // test hooks (mvsn_debug_set_band_flags): bit 0 = every step takes the slow gather path; bit 1 = the chain's last band
  // never runs (what a shared device can do to a launch that needs co-residency); bits 8.. = log2 of the spin limit
  const unsigned spin_limit = (flags >> 8) ? 1u << ((flags >> 8) & 31) : SB_SPIN_LIMIT;
  if ((flags & 2) && m == NB - 1) return;

  // (the small arrays first: their reads are lane offset + IMMEDIATE only below 64 KB -- behind U every (array, cout)
  // pair had its own address register, hoisted out of the step loop and spilled)
  float *sparams = smem;
  float *gstat = sparams + CH_SP_FLOATS;                       // [layer 2][group 4]: the previous step's means (the shift)
  int *fastw = reinterpret_cast<int *>(gstat + 16);           // [parity]: 1 = some pixel of the plane needs the slow path
  float *maskb = gstat + 16 + 8;
  float *U = maskb + GEO::MASK;
  float *act = U + SB_USLOTS * 1024;

  // (workgroup-uniform: scalar registers; every granule address is this base + a 32-bit offset)
  const gu64 *ws = (const gu64 *)(reinterpret_cast<const u64 *>(a.workspace) + (size_t)nl * GEO::CHAIN_U64);
  gu32 *status = (gu32 *)(reinterpret_cast<u64 *>(a.workspace) + (size_t)a.ws_chains * GEO::CHAIN_U64);
  bool dead = false;

  const float *upk = a.packed + CH_DIRECT_FLOATS;
  int lane16 = lane * 16;
  // a layer's first 16 half-k-step blocks -> slots 0..15; run i = (slot s = i >> 2, (cout tile, xi quad) r = i & 3) is the
  // 1 KB run ((c4 * 2 + ct) * 4 + half * 2 + xq) of the chain_wino layout; wave w takes runs w, w + 8, ...
  bool dma_on = true;
  auto dma_u = [&](const float *src, int nc) {
    if ((MVSN_SB_ABLATE & 1) && !dma_on) return;
    int l16 = lane16;
    asm volatile("" : "+v"(l16));   // (the eight 64-bit addresses of a call are formed here, not once per step for all three)
    for (int i = wave; i < SB_USLOTS * 4; i += SB_WAVES) {
      const int s = i >> 2, r = i & 3, half = s >= nc ? 1 : 0, c4 = s - half * nc;
      const int run = (c4 * 2 + (r >> 1)) * 4 + half * 2 + (r & 1);
      const char *g = reinterpret_cast<const char *>(src + (size_t)run * 256);
      __builtin_amdgcn_global_load_lds(SB_GPTR(g + (unsigned)l16), SB_LPTR(U + i * 256), 16, 0, 0);
    }
  };
  auto dma_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // ---- one-time set-up ---------------------------------------------------------------------
  dma_u(upk, 9);
  for (int i = tid; i < 35 * CSA; i += SB_THREADS) act[i] = 0.0f;
  for (int i = tid; i < CH_SP_FLOATS; i += SB_THREADS) sparams[i] = a.packed[CH_W0_FLOATS + 2 * CH_W1_FLOATS + i];
  if (tid < 16) gstat[tid] = 0.0f;
  if (tid < 8) fastw[tid] = 0;
  const float *bias0 = sparams, *gn0w = sparams + 32, *gn0b = sparams + 64, *bias1 = sparams + 96,
              *gn1w = sparams + 128, *gn1b = sparams + 160, *bias2 = sparams + 192;

  // This lane's patch: patch q of the band (row-major), outputs (lo + 2 pr + a, 2 pc + b), e = a * 2 + b.  Only the
  // window origin lives across the layers; everything else about the patch is re-derived per phase from an opaque copy of
  // the thread id (lane_geo below): held across the three layers those values were what spilled (a reload in front of a
  // layer's first multiply waits a scratch round trip).
  const bool tile_live = wave * 16 < GEO::NPATCH;       // wave-uniform
  int wb;                                               // window origin inside a plane (local row 0 = image row lo - 1)
  {
    const int q = wave * 16 + (lane & 15), qq = q < GEO::NPATCH ? q : 0;
    const int pr = qq / PCOLS, pc = qq - pr * PCOLS;
    wb = (2 * pr) * RS + 2 * pc;
  }
  struct LaneGeo {
    bool pvalid, top_pub, bot_pub;
    int pr, ob, cbase, px0, py0;
  };
  auto lane_geo = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    LaneGeo g;
    const int l = t & 63, q = (t >> 6) * 16 + (l & 15);
    g.pvalid = q < GEO::NPATCH;
    const int qq = g.pvalid ? q : 0;
    g.pr = qq / PCOLS;
    const int pc = qq - g.pr * PCOLS;
    g.ob = (2 * g.pr) * RS + 2 * pc + RS + 1;           // output (0, 0)
    g.cbase = (l >> 4) * 4;                             // this lane's couts: ct * 16 + cbase + r
    g.py0 = lo + 2 * g.pr, g.px0 = 2 * pc;
    g.top_pub = g.pvalid && g.pr == 0 && m > 0;                    // first pixel row of the band faces band m - 1
    g.bot_pub = g.pvalid && g.pr == GEO::PROWS - 1 && m < NB - 1;  // last pixel row faces band m + 1
    return g;
  };
  const float inv_n = 1.0f / (8.0f * (float)P);
  __syncthreads();

  const float *Hn = a.H + (size_t)n * D * 9;
  const float *Hin = a.Hinc + (size_t)n * D * 9;
  const float *src = a.src + (size_t)n * 3 * P;
  uint8_t *maskg = a.mask + (size_t)n * D * P;
  typedef typename ChainCost<C16>::type cost_t;
  cost_t *costg = reinterpret_cast<cost_t *>(a.cost) + (size_t)n * 32 * D * P;
  float *fvolg = a.fvol ? a.fvol + (size_t)n * 32 * D * P : nullptr;
  const float *flp = a.fl + (size_t)(n % a.B) * 32 * P;

  // Halo role of a thread: (side: 0 = row lo - 1, 1 = row hi + 1; channels 8 cg .. + 7; column x) of the two rows the
  // neighbours own.  Re-derived from an opaque copy of the thread id wherever it is used: kept across the layers its
  // five values would be five of a wave's 256 registers for the whole step.
  struct HaloRole {
    bool valid;
    int cg, off;      // channel group; offset of the element inside a plane
    unsigned grow;    // granule index inside a hand-off region: the neighbour's row facing this band, channel 8 cg
  };
  auto halo_role = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    HaloRole h;
    const bool in = t < GEO::HALO;
    const int ii = in ? t : 0;
    const int s = ii / (4 * cols), cg = (ii - s * 4 * cols) / cols, x = ii - (ii / cols) * cols;
    h.valid = in && (s ? m < NB - 1 : m > 0);            // that neighbour exists
    h.cg = cg;
    h.off = (s ? BR + 1 : 0) * RS + x + 1;
    // its last row (side 1) for our row lo - 1, its first (side 0) for hi + 1
    const int nb = h.valid ? (s ? m + 1 : m - 1) : 0;
    h.grow = (unsigned)(((nb * 2 + (s ? 0 : 1)) * 32 + cg * 8) * cols + x);
    return h;
  };

  // Does step dn's gather stay inside every band's rows lo-1 .. hi+1?  Every workgroup walks the WHOLE plane (all bands of
  // a chain reach the same answer); pixels that need more set fastw[dn & 1].  (NaN coordinates fall to y0 = 0: slow path.)
  auto plan_gather = [&](int dn) {
    float Hl[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hl[i] = Hin[dn * 9 + i];
    bool slow = false;
#pragma unroll
    for (int it = 0; it < GEO::PLANE_IT; ++it) {
      const int p = tid + it * SB_THREADS;
      if (p < P) {
        const int yy = p / cols, xx = p - yy * cols;
        // the row coordinate alone, by the gather's own expressions (warp_coord / bilinear_taps: the x side is dead code here)
        WarpCoord c = warp_coord(Hl, (float)xx, (float)yy, (float)rows, (float)cols);
        float iy = c.iy < 0.0f ? 0.0f : (c.iy > (float)(rows - 1) ? (float)(rows - 1) : c.iy);
        // CONSERVATIVE at integer boundaries: this copy of warp_coord is compiled in another context than the gather's
        // (x side dead, whole-plane loop), and the contraction of its multiply-adds may differ in the last bits -- a
        // coordinate within PLAN_TOL of an integer counts as BOTH rows, so the plan can say "in range" only when the
        // gather's own floor(iy) is in range whichever way its last bit fell (a step judged slow is merely slower:
        // the two paths produce the same bits).  |iy| <= 32 here: 1e-4 is ~26 ulps, far above any contraction difference.
        constexpr float PLAN_TOL = 1e-4f;
        int ya = (int)floorf(iy - PLAN_TOL), yb = (int)floorf(iy + PLAN_TOL);
        ya = ya < 0 ? 0 : ya;
        yb = yb > rows - 1 ? rows - 1 : yb;
        if (!(iy >= 0.0f)) ya = yb = 0;               // NaN coordinate: the gather's y0 = 0
        const int blo = (yy / BR) * BR;
        slow |= ya < blo - 1 || yb + 1 > blo + BR;   // (the unclamped + 1 row: a zero halo row at the image's edge)
      }
    }
    if (__any(slow) && lane == 0) atomicOr(&fastw[dn & 1], 1);
  };

  // ---- plane 0: mask from the plane's homography, features from the extractor (rows lo-1 .. hi+1) ---------------
  {
    float Hl[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hl[i] = Hn[i];
    for (int p = tid; p < BR * cols; p += SB_THREADS) {
      const int yy = lo + p / cols, xx = p % cols;
      WarpCoord c = warp_coord(Hl, (float)xx, (float)yy, (float)rows, (float)cols);
      maskb[p] = c.outside ? 1.0f : 0.0f;
      maskg[yy * cols + xx] = c.outside ? 1 : 0;
    }
    const float *f0 = a.f0 + (size_t)n * 32 * P;
    for (int i = tid; i < 32 * ER * cols; i += SB_THREADS) {
      const int c = i / (ER * cols), e = i - c * (ER * cols);
      const int er = e / cols, xx = e - er * cols, yy = lo - 1 + er;
      if (yy >= 0 && yy < rows) act[(3 + c) * CSA + er * RS + xx + 1] = f0[(size_t)c * P + yy * cols + xx];
    }
    if (D > 1) plan_gather(1);
  }
  __syncthreads();
  // cost slice of plane 0 (band rows), generic pass: 16-byte pieces
  {
    constexpr int quads = (BR * cols) >> 2;
    for (int i = tid; i < 32 * quads; i += SB_THREADS) {
      const int c = i / quads, p4 = (i - c * quads) * 4;
      const int yl = p4 / cols, xx = p4 - yl * cols;
      const float *fr = act + (3 + c) * CSA + (yl + 1) * RS + xx + 1;
      const size_t go = (size_t)c * P + (lo + yl) * cols + xx;
      const floatx4 l = *reinterpret_cast<const floatx4 *>(flp + go);
      const floatx4 mm = *reinterpret_cast<const floatx4 *>(maskb + p4);
      floatx4 cst, ftr;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float f = fr[k];
        cst[k] = mm[k] != 0.0f ? 0.0f : fabsf(l[k] - f);
        ftr[k] = mm[k] != 0.0f ? 0.0f : f;
      }
      const size_t vo = ((size_t)c * D) * P + (lo + yl) * cols + xx;
      chain_cost_nt(costg + vo, cst);
      if (fvolg) __builtin_nontemporal_store(ftr, reinterpret_cast<floatx4 *>(fvolg + vo));
    }
    // step 1 on the slow path reads every tap from the granules: plane 0 has to be there (tag 1)
    if ((fastw[1] || (flags & 1)) && D > 1) {
      const float *f0 = a.f0 + (size_t)n * 32 * P;
      for (int i = tid; i < 32 * BR * (cols / 2); i += SB_THREADS) {
        const int c = i / (BR * (cols / 2)), e = i - c * (BR * (cols / 2));
        const int yl = e / (cols / 2), x2 = (e - yl * (cols / 2)) * 2;
        const unsigned go = (unsigned)(c * P + (lo + yl) * cols + x2);
        sb_publish2<0>(ws + GEO::FG, go * 8u, 1u, f0[go], f0[go + 1]);
      }
    }
  }

#define SB_STAMP(i)                                                                                     \
  do {                                                                                                  \
    if (a.dbg && blockIdx.x == 0 && tid == 0 && d >= 2 && d <= 5) a.dbg[(d - 2) * 32 + (i)] = __builtin_readcyclecounter(); \
  } while (0)

  // boundary rows of this lane's 8 x (2x2) values -> the region of a hand-off (rows: e = 0, 1 top; e = 2, 3 bottom);
  // four byte offsets (side x cout tile), the rows of a cout tile by immediates
  auto publish_rows = [&](const LaneGeo &L, size_t region, unsigned tag, const float (&v)[2][4][4]) {
    const gu64 *R = ws + region;
    // granule index of this lane's (side 0, cout tile 0, r = 0) boundary-row store inside the region
    const int po = ((m * 2) * 32 + L.cbase) * cols + L.px0;
    if (L.top_pub) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const unsigned o = (unsigned)(po + ct * 16 * cols) * 8u;
        sb_publish2<0 * cols * 8>(R, o, tag, v[ct][0][0], v[ct][0][1]);
        sb_publish2<1 * cols * 8>(R, o, tag, v[ct][1][0], v[ct][1][1]);
        sb_publish2<2 * cols * 8>(R, o, tag, v[ct][2][0], v[ct][2][1]);
        sb_publish2<3 * cols * 8>(R, o, tag, v[ct][3][0], v[ct][3][1]);
      }
    }
    if (L.bot_pub) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const unsigned o = (unsigned)(po + 32 * cols + ct * 16 * cols) * 8u;
        sb_publish2<0 * cols * 8>(R, o, tag, v[ct][0][2], v[ct][0][3]);
        sb_publish2<1 * cols * 8>(R, o, tag, v[ct][1][2], v[ct][1][3]);
        sb_publish2<2 * cols * 8>(R, o, tag, v[ct][2][2], v[ct][2][3]);
        sb_publish2<3 * cols * 8>(R, o, tag, v[ct][3][2], v[ct][3][3]);
      }
    }
  };
  // a thread's halo item (8 channels of one column of a neighbour's boundary row) of a hand-off
  auto collect_rows = [&](size_t region, const HaloRole &h, unsigned tag, unsigned code, float (&hv)[8]) {
    sb_sweep<8, cols>(ws + region, h.grow, tag, h.valid, hv, dead, status, code, spin_limit);
  };

  // ---- the recurrence ------------------------------------------------------------------------
  dma_on = false;
  for (int d = 1; d < D; ++d) {
    asm volatile("" : "+v"(tid), "+v"(lane16));
    const int lane_s = tid & 63;
    const int par = d & 1;
    SB_STAMP(0);

    // A1: image plane d on rows lo-1 .. hi+1 and the band's mask (global gathers; the source image stays in L1 / L2)
    float img[IMG_IT][3], mk[IMG_IT];
    {
      float Hl[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Hl[i] = Hn[d * 9 + i];
#pragma unroll
      for (int it = 0; it < IMG_IT; ++it) {
        const int e = tid + it * SB_THREADS;
        const int er = e / cols, xx = e - er * cols, yy = lo - 1 + er;
        img[it][0] = img[it][1] = img[it][2] = 0.0f, mk[it] = 0.0f;
        if (e < GEO::EXT && yy >= 0 && yy < rows) {
          WarpCoord c = warp_coord(Hl, (float)xx, (float)yy, (float)rows, (float)cols);
          Bilinear b = bilinear_taps(c.ix, c.iy, rows, cols);
          const float keep = c.outside ? 0.0f : 1.0f;
          mk[it] = c.outside ? 1.0f : 0.0f;
          const int o00 = b.y0 * cols + b.x0, o01 = b.y0 * cols + b.x1, o10 = b.y1 * cols + b.x0, o11 = b.y1 * cols + b.x1;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float *ic = src + (size_t)ch * P;
            img[it][ch] = keep * (ic[o00] * b.w00 + ic[o01] * b.w01 + ic[o10] * b.w10 + ic[o11] * b.w11);
          }
        }
      }
    }
    // (fastw[par] was completed during the previous step, several barriers ago)
    const bool fast = !(flags & 1) && fastw[par] == 0;   // workgroup-uniform, chain-uniform

    // E1 (consume): the neighbours' boundary rows of F_{d-1} -> halo rows of the feature planes (plane 0: loaded above)
    if (d > 1) {
      const HaloRole h = halo_role();
      float hv[8];
      collect_rows(GEO::E1, h, (unsigned)d, 2u, hv);
      if (h.valid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) act[(3 + h.cg * 8 + j) * CSA + h.off] = hv[j];
      }
    }
    SB_STAMP(1);
    sb_barrier();   // Ba: F_{d-1} complete on rows lo-1 .. hi+1 (own rows: the previous step's epilogue)
    SB_STAMP(2);

    // A2: previous plane's features moved by the incremental homography
    float fp[2][4][4];
    {
      const LaneGeo L = lane_geo();
      const bool pvalid = L.pvalid;
      const int px0 = L.px0, py0 = L.py0, cbase = L.cbase, ob = L.ob;
      float Hl[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Hl[i] = Hin[d * 9 + i];
      if (fast) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float px = (float)(px0 + (e & 1)), py = (float)(py0 + (e >> 1));
          WarpCoord c = warp_coord(Hl, px, py, (float)rows, (float)cols);
          Bilinear b = bilinear_taps(c.ix, c.iy, rows, cols);
          const float keep = c.outside ? 0.0f : 1.0f;
          const float w00 = keep * b.w00, w01 = keep * b.w01, w10 = keep * b.w10, w11 = keep * b.w11;
          // (+1 taps unclamped: where the clamp would act their weight is exactly zero and the slot read is a zero halo)
          const int o = (b.y0 - lo + 1) * RS + b.x0 + 1;
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float *fc = act + (3 + ct * 16 + cbase + r) * CSA + o;
              fp[ct][r][e] = fc[0] * w00 + fc[1] * w01 + fc[RS] * w10 + fc[RS + 1] * w11;
            }
        }
      } else {
        // Every tap from the granules of F_{d-1} (all bands published the whole plane, tag d).  A rolled loop over (pixel,
        // cout tile): 16 granules per round, the four results written straight to the lane's own slots of the planes --
        // nobody reads the planes on this path (the gathers go to the granules), and the registers of a step are not shaped
        // by a path that large inter-plane motion alone takes.
        const gu64 *Fg = ws + GEO::FG;
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
          const int e = it >> 1, ct = it & 1;
          const float px = (float)(px0 + (e & 1)), py = (float)(py0 + (e >> 1));
          WarpCoord c = warp_coord(Hl, px, py, (float)rows, (float)cols);
          Bilinear b = bilinear_taps(c.ix, c.iy, rows, cols);
          const float keep = c.outside ? 0.0f : 1.0f;
          const float w[4] = {keep * b.w00, keep * b.w01, keep * b.w10, keep * b.w11};
          // clamped +1 taps re-read the tap they are clamped to: their weight is exactly zero
          const unsigned o00 = (unsigned)(b.y0 * cols + b.x0), dx = b.x1 - b.x0, dy = (b.y1 - b.y0) * cols;
          const unsigned base = pvalid ? (unsigned)((ct * 16 + cbase) * P) + o00 : 0u;
          float val[4] = {0.f, 0.f, 0.f, 0.f};
          for (unsigned spins = 0;; ++spins) {
            bool ok = true;
            if (pvalid) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                  const u64 x = __hip_atomic_load(Fg + (base + (unsigned)(r * P) + ((t & 1) ? dx : 0u) + ((t & 2) ? dy : 0u)), SB_RLX_AGENT);
                  ok &= (unsigned)(x >> 32) == (unsigned)d;
                  acc += __builtin_bit_cast(float, (unsigned)x) * w[t];
                }
                val[r] = acc;
              }
            }
            if (__all(ok) || dead) break;
            if (spins >= spin_limit) {
              dead = true;
              __hip_atomic_store(status, 1u, SB_RLX_AGENT);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          if (pvalid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) act[(3 + ct * 16 + cbase + r) * CSA + ob + (e >> 1) * RS + (e & 1)] = val[r];
          }
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float *fc = act + (3 + ct * 16 + cbase + r) * CSA + ob;
            fp[ct][r][0] = pvalid ? fc[0] : 0.f, fp[ct][r][1] = pvalid ? fc[1] : 0.f;
            fp[ct][r][2] = pvalid ? fc[RS] : 0.f, fp[ct][r][3] = pvalid ? fc[RS + 1] : 0.f;
          }
      }
    }
    // E1b (publish): the band's boundary rows of the moved features are the neighbours' conv0 halo rows
    publish_rows(lane_geo(), GEO::E1B, (unsigned)d, fp);
    SB_STAMP(3);
    sb_barrier();   // B1: every gather of plane d-1 is done
    SB_STAMP(4);

    // A3: lay out the refiner input [image(3) | moved features(32)] on rows lo-1 .. hi+1
    if (const LaneGeo L = lane_geo(); L.pvalid) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float *dst = act + (3 + ct * 16 + L.cbase + r) * CSA + L.ob;
          dst[0] = fp[ct][r][0], dst[1] = fp[ct][r][1], dst[RS] = fp[ct][r][2], dst[RS + 1] = fp[ct][r][3];
        }
    }
#pragma unroll
    for (int it = 0; it < IMG_IT; ++it) {
      const int e = tid + it * SB_THREADS;
      const int er = e / cols, xx = e - er * cols, yy = lo - 1 + er;
      if (e < GEO::EXT && yy >= 0 && yy < rows) {
        const int o = er * RS + xx + 1;
        act[0 * CSA + o] = img[it][0];
        act[1 * CSA + o] = img[it][1];
        act[2 * CSA + o] = img[it][2];
        if (er >= 1 && er <= BR) {
          maskb[(er - 1) * cols + xx] = mk[it];
          maskg[(size_t)d * P + yy * cols + xx] = mk[it] != 0.0f ? 1 : 0;
        }
      }
    }
    {   // E1b (consume)
      const HaloRole h = halo_role();
      float hv[8];
      collect_rows(GEO::E1B, h, (unsigned)d, 3u, hv);
      if (h.valid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) act[(3 + h.cg * 8 + j) * CSA + h.off] = hv[j];
      }
    }
    dma_landed();   // conv0's U
    SB_STAMP(5);
    sb_barrier();   // B2
    SB_STAMP(6);

    float y[2][4][4] = {};
    // E2 / E3: bias, per-wave partial GroupNorm sums (shifted by the previous step's mean, as chain_wino_kernel), published
    // with the band's boundary rows of the raw output; then the other bands' sums and rows are collected, the statistics
    // formed (every workgroup of the chain adds the same records in the same order: the same bits) and own outputs and
    // halo rows normalised + activated into the planes.
    auto exchange = [&](int layer, size_t region, const float *bias, const float *gamma, const float *beta, bool residual,
                        auto &&meanwhile) {
      const LaneGeo L = lane_geo();
      const bool pvalid = L.pvalid;
      const int cbase = L.cbase, ob = L.ob;
      float *gs = gstat + layer * 4;
      float s[4] = {0.f, 0.f, 0.f, 0.f};   // [ct][sum, sum of squares]
      float shift[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        shift[ct] = gs[ct * 2 + (lane_s >> 5)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float b = bias[ct * 16 + cbase + r];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[ct][r][e] += b;
            const float dv = y[ct][r][e] - shift[ct];
            s[ct * 2] += dv;
            s[ct * 2 + 1] += dv * dv;
          }
        }
      }
      if (!(pvalid && tile_live)) s[0] = s[1] = s[2] = s[3] = 0.f;
      sb_half_wave_sums(s);
      const gu64 *Sl = ws + GEO::SG + (size_t)layer * (NB * 64);
      if ((lane_s & 31) == 16) {   // record of (wave, group = ct * 2 + half-wave): [sum, sum of squares]
        const unsigned g8 = (unsigned)(m * 64 + wave * 8 + (lane_s >> 5) * 2) * 8u;
        sb_publish2<0>(Sl, g8, (unsigned)d, s[0], s[1]);
        sb_publish2<32>(Sl, g8, (unsigned)d, s[2], s[3]);
      }
      publish_rows(L, region, (unsigned)d, y);
      SB_STAMP(16 + layer * 4);
      sb_barrier();   // B3 / B7: planes and U free
      // (issued in front of the hand-off's polling loads; behind them measured the same: profiles/r05_slab/README.md)
      dma_u(upk + (layer == 0 ? CW_U0_FLOATS : CW_U0_FLOATS + CW_U1_FLOATS), 8);
      meanwhile();    // work that needs none of the hand-off, placed where the workgroup would otherwise only wait
      SB_STAMP(17 + layer * 4);
      // collect: every wave ALL sum records (lane l, band j: record (wave l >> 3, group (l >> 1) & 3, moment l & 1) --
      // no LDS round trip, no barrier), every thread its halo item
      float tot;   // this lane's (group, moment) summed over the waves of a band, then over the bands
      {
        float sv[NB];
        sb_sweep<NB, 64>(Sl, (unsigned)lane_s, (unsigned)d, true, sv, dead, status, 4u + layer, spin_limit);
        tot = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          float v = sv[j];
          v += __shfl_xor(v, 8, 64);
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          tot += v;
        }
      }
      const HaloRole h = halo_role();
      float hr[8];
      collect_rows(region, h, (unsigned)d, 6u + layer, hr);
      SB_STAMP(18 + layer * 4);
      auto stats_of = [&](int g, float sh, float &mean, float &rstd) {
        const float s1 = __shfl(tot, g * 2, 64), s2 = __shfl(tot, g * 2 + 1, 64);
        const float ms = s1 * inv_n;
        const float var = fmaxf(s2 * inv_n - ms * ms, 0.0f);
        mean = sh + ms;
        rstd = 1.0f / sqrtf(var + SB_GN_EPS);
      };
      // (the shuffles are wave-wide: every lane takes part, with or without a patch)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float mean, rstd;
        stats_of(ct * 2 + (lane_s >> 5), shift[ct], mean, rstd);
        if (pvalid) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = ct * 16 + cbase + r;
            const float scl = rstd * gamma[c];
            const float sft = beta[c] - mean * scl;
            float *dst = act + c * CSA + ob;
            if (residual) {
              dst[0] += lrelu02(y[ct][r][0] * scl + sft), dst[1] += lrelu02(y[ct][r][1] * scl + sft);
              dst[RS] += lrelu02(y[ct][r][2] * scl + sft), dst[RS + 1] += lrelu02(y[ct][r][3] * scl + sft);
            } else {
              dst[0] = lrelu02(y[ct][r][0] * scl + sft), dst[1] = lrelu02(y[ct][r][1] * scl + sft);
              dst[RS] = lrelu02(y[ct][r][2] * scl + sft), dst[RS + 1] = lrelu02(y[ct][r][3] * scl + sft);
            }
          }
        }
      }
      {
        float mean, rstd;
        stats_of(h.cg, gs[h.cg], mean, rstd);
        if (h.valid) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = h.cg * 8 + j;
            const float scl = rstd * gamma[c];
            const float sft = beta[c] - mean * scl;
            const float v = lrelu02(hr[j] * scl + sft);
            if (residual) act[c * CSA + h.off] += v;
            else act[c * CSA + h.off] = v;
          }
        }
      }
      // the next step's shift = this step's means.  Written behind a barrier: every wave has read gs above.
      const float s1 = __shfl(tot, (lane_s & 3) * 2, 64);
      const float newmean = gs[lane_s & 3] + s1 * inv_n;
      return newmean;
    };

    if (tile_live) slab_layer<9, CSA, RS>(act, U, upk, wb, lane, y);
    SB_STAMP(7);
    const float nm0 = exchange(0, GEO::E2, bias0, gn0w, gn0b, false, [] {});
    SB_STAMP(8);
    dma_landed();
    sb_barrier();   // B6
    if (wave == 0 && lane_s < 4) gstat[lane_s] = nm0;
    SB_STAMP(9);

    if (tile_live) slab_layer<8, CSA, RS>(act, U, upk + CW_U0_FLOATS, wb, lane, y);
    SB_STAMP(10);
    const float nm1 = exchange(1, GEO::E3, bias1, gn1w, gn1b, true, [&] {   // x2 = x1 + LReLU(GN(conv1(x1)))
      // the slow-path decision of the NEXT step (read by this step's epilogue and by the next step's gather)
      if (d + 1 < D && !(MVSN_SB_ABLATE & 2)) plan_gather(d + 1);
    });
    SB_STAMP(11);
    dma_landed();
    sb_barrier();   // B10
    if (wave == 0 && lane_s < 4) gstat[4 + lane_s] = nm1;
    SB_STAMP(12);

    if (tile_live) slab_layer<8, CSA, RS>(act, U, upk + CW_U0_FLOATS + CW_U1_FLOATS, wb, lane, y);
    float2 fl[2][4][2];   // left features of this lane's outputs, in flight across the barrier
    const LaneGeo L = lane_geo();
    const bool pvalid = L.pvalid;
    const int cbase = L.cbase, ob = L.ob, px0 = L.px0, py0 = L.py0, pr = L.pr;
    const float *fl_lane = flp + (cbase * P + py0 * cols + px0);
    const int slice_off = (cbase * D) * P + py0 * cols + px0;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
          fl[ct][r][a2] = *reinterpret_cast<const float2 *>(fl_lane + (size_t)(ct * 16 + r) * P + a2 * cols);
    SB_STAMP(13);
    sb_barrier();   // B11: planes and U free
    dma_u(upk, 9);  // conv0 of the next step
    const bool next_slow = (fastw[par ^ 1] != 0 || (flags & 1)) && d + 1 < D;   // (complete since B10)
    if (tid == 0) fastw[par] = 0;   // (read by everyone at the top of this step; next written during step d + 1)

    // epilogue: F_d = moved + conv_final(...): granules for the neighbours first (they wait for them), then the planes
    // (the next step's gather source) and the cost slice (not mask) * |left - right| straight from the registers
    {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float b2 = bias2[ct * 16 + cbase + r];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[ct][r][e] = fp[ct][r][e] + (y[ct][r][e] + b2);   // y := F_d
        }
      if (d + 1 < D) publish_rows(L, GEO::E1, (unsigned)(d + 1), y);
      if (next_slow && pvalid) {   // the next step gathers from the granules: the whole band
        const gu64 *Fg = ws + GEO::FG;
        int go = cbase * P + py0 * cols + px0;
        asm volatile("" : "+v"(go));
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned o = (unsigned)(go + (ct * 16 + r) * P) * 8u;
            sb_publish2<0>(Fg, o, (unsigned)(d + 1), y[ct][r][0], y[ct][r][1]);
            sb_publish2<cols * 8>(Fg, o, (unsigned)(d + 1), y[ct][r][2], y[ct][r][3]);
          }
      }
      if (pvalid) {
        const float2 m0 = *reinterpret_cast<const float2 *>(maskb + (2 * pr) * cols + px0);
        const float2 m1 = *reinterpret_cast<const float2 *>(maskb + (2 * pr + 1) * cols + px0);
        const bool out[4] = {m0.x != 0.0f, m0.y != 0.0f, m1.x != 0.0f, m1.y != 0.0f};
        cost_t *cd = costg + (size_t)d * P;
        float *fd = fvolg ? fvolg + (size_t)d * P : nullptr;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float(&f)[4] = y[ct][r];
            float *dst = act + (3 + ct * 16 + cbase + r) * CSA + ob;
            dst[0] = f[0], dst[1] = f[1], dst[RS] = f[2], dst[RS + 1] = f[3];
            cost_t *cdst = cd + ((ct * 16 + r) * D) * P + slice_off;
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2) {
              sb_float2v c2;
              c2.x = out[a2 * 2] ? 0.0f : fabsf(fl[ct][r][a2].x - f[a2 * 2]);
              c2.y = out[a2 * 2 + 1] ? 0.0f : fabsf(fl[ct][r][a2].y - f[a2 * 2 + 1]);
              chain_cost_nt(cdst + a2 * cols, c2);
            }
            if (fd) {
              float *fdst = fd + ((ct * 16 + r) * D) * P + slice_off;
#pragma unroll
              for (int a2 = 0; a2 < 2; ++a2) {
                sb_float2v f2;
                f2.x = out[a2 * 2] ? 0.0f : f[a2 * 2];
                f2.y = out[a2 * 2 + 1] ? 0.0f : f[a2 * 2 + 1];
                __builtin_nontemporal_store(f2, reinterpret_cast<sb_float2v *>(fdst + a2 * cols));
              }
            }
          }
      }
    }
    SB_STAMP(14);
    // (no barrier here: the next step's Ba follows the E1 hand-off and covers the epilogue's plane writes)
  }
#undef SB_STAMP
  dma_landed();       // the last step's look-ahead fetch must not outlive the workgroup's LDS
  // a hand-off that timed out leaves wrong numbers behind: poison the cost slice (see chain_band_kernel)
  if (__syncthreads_or(dead)) {
    const LaneGeo L = lane_geo();
    if (L.pvalid) chain_cost_st(costg + ((size_t)(D - 1) * P + (L.cbase * D) * P + L.py0 * cols + L.px0), __builtin_nanf(""));
  }
}

// ---- host side: the slab plans as seen by the banded form's dispatcher (mvsn_chain_band.hip) -------------------------
template <class GEO>
static SlabPlan slab_plan_of() {
  return SlabPlan{GEO::NB, SB_THREADS, GEO::CHAIN_U64, (size_t)GEO::LDS_FLOATS * sizeof(float), chain_slab_kernel<GEO, false>,
                  chain_slab_kernel<GEO, true>};
}

bool chain_slab_plan(int rows, int cols, SlabPlan *p) {
  if (rows == 30 && cols == 40) *p = slab_plan_of<Slab30x40>();
  else if (rows == 32 && cols == 64) *p = slab_plan_of<Slab32x64>();
  else if (rows == 16 && cols == 32) *p = slab_plan_of<Slab16x32>();
  else return false;
  return true;
}

}  // namespace mvsn
