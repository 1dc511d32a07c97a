// The fused incremental chain, Winograd form (see include/mvsn_hip.h: mvsn_incremental_cost_volume).
//
// Same contract and step structure as chain_kernel (mvsn_chain.hip): one persistent workgroup per (reference
// image, source view) chain, previous-plane features resident in LDS, per step the homography gather, the three
// 3x3 convolutions of FeatureRefiner (multi_view_stereonet.py:424-440) with two GroupNorms, and the cost slice.
// The convolutions run as Winograd F(2x2,3x3): Y = A^T[(G g G^T) .* (B^T d B)]A, the sum over input channels taken
// in the transformed domain as 16 small GEMMs M_xi[cout][patch] = sum_cin U_xi[cout][cin] V_xi[cin][patch] on
// v_mfma_f32_16x16x4_f32 -- 16 products per 2x2 outputs and (cin, cout) pair instead of 36, all fp32.
//
//   * 512 threads = 8 waves (2 per SIMD, 256 VGPRs each).  Wave w owns patch tile w = 16 consecutive 2x2 patches
//     (at 16x32: patch row w) and BOTH cout tiles: 2 x 16 accumulators of 4 = 128 VGPRs.
//   * A = U_xi (16 couts x 4 cins), B = V_xi (4 cins x 16 patches).  Lane (k = lane>>4, p = lane&15) reads the 4x4
//     input window of patch p, channel 4*c4 + k straight from the activation planes (four ds_read2_b64: data column
//     x is stored at index x + 1, so a window starts on an even index) and computes B^T d B in registers (32 adds);
//     the 16 coefficients it ends up with ARE its B-fragment values.  No transformed tile is ever stored.
//   * D = couts x patches: a lane ends up with 4 consecutive couts of its own patch for all 16 xi, so the output
//     transform A^T m A runs in registers and GroupNorm reduces over half-waves exactly as in the direct kernel.
//   * Transformed weights: 72 / 64 / 64 KB per layer -- one layer fits next to the activation planes.  The U of the
//     NEXT layer is fetched by LDS-DMA (global_load_lds, 1 KB per instruction, from L2) right after the barrier that
//     ends the current layer's multiplies, and lands behind the GroupNorm / layout phase that follows.
//   * GroupNorm: per-wave shifted sums by DPP reductions, written ahead of the barrier that ends the layer's multiplies
//     and combined across the 8 waves behind it (no barrier of its own).
//
// LDS plan (floats), RS = cols + 2, CS = (rows + 1) * RS:
//   U       [9 * 2048]      transformed weights of the current layer, [k-step][cout tile][xi quad][lane][4 xi]
//   sparams [224]           biases and GroupNorm affine
//   red     [2][8][4][4]    per-wave GroupNorm moments
//   maskb   [P]             out-of-image flag of the current plane
//   act     36 * CS + RS    activation planes: channel c, row y (-1..rows), column x (-1..cols) at
//                           c*CS + (y+1)*RS + (x+1); row -1 of channel c+1 doubles as row `rows` of channel c.
//                           Halo rows / columns are zeroed once and never written.
//                           ch 0..2 image plane d, ch 3..34 features, ch 35 zero (K padding).
// 16x32: 18432 + 224 + 256 + 512 + 20842 floats = 161,064 bytes of the 163,840.
#include "mvsn_chain.h"
#include "mvsn_common.h"

namespace mvsn {

constexpr int CW_THREADS = 512;
constexpr int CW_WAVES = 8;
constexpr int CW_RED_FLOATS = 2 * CW_WAVES * 4 * 4;
constexpr float CW_GN_EPS = 1e-5f;
typedef float float2v __attribute__((ext_vector_type(2)));

// Workgroup barrier that publishes LDS writes but leaves global loads / stores in flight (__syncthreads() also waits
// for vmcnt(0): the cost-slice stores and the left-feature loads would be drained at every barrier of the step).
__device__ __forceinline__ void cw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define CW_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define CW_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

static size_t chain_wino_lds_bytes(int rows, int cols) {
  const int P = rows * cols, RS = cols + 2, CS = (rows + 1) * RS;
  const size_t f = (size_t)CW_U0_FLOATS + CH_SP_FLOATS + CW_RED_FLOATS + ((P + 3) & ~3) + 36 * (size_t)CS + RS + 2;
  return f * sizeof(float);
}

bool chain_wino_supported(int rows, int cols) {
  if (rows < 2 || cols < 2 || (rows & 1) || (cols & 1)) return false;
  const int patches = (rows / 2) * (cols / 2);
  return patches <= CW_WAVES * 16 && rows * cols <= 2 * CW_THREADS && chain_wino_lds_bytes(rows, cols) <= 160 * 1024;
}

// ---------------------------------------------------------------------------------------------
// one 3x3 layer: acc[ct][xi] (+)= U_xi * V_xi over NC k-steps of 4 input channels, then the output transform
// ---------------------------------------------------------------------------------------------
// The 16 xi = (i, j) are walked in two halves by transform row i (i = 0,1 then i = 2,3): 64 accumulator registers
// at a time instead of 128, each half's output transform folded into y as soon as its multiplies are done.  The
// input transform costs the same (row i of B^T d B needs two rows of d), the window reads 3 rows per half.
// Software pipeline per k-step: transform the window that is already in registers, issue the LDS reads of the NEXT
// k-step (3 window rows + 4 quads of U), then the 16 multiplies -- no LDS round trip sits in front of an MFMA.
// Measured (tools/chain_phases.py, s_memtime stamps per wave), 9 k-steps x 2 halves of the first layer:
//   * one wave per SIMD alone: 730 cycles per k-step = 16 MFMAs x 32 + 23 VALU / LDS instructions x ~9.5;
//   * two waves per SIMD (this kernel): 1405 per pair of k-steps -- the two waves leave their barrier together, run
//     their transform sections together and then alternate on the matrix pipe, so the sections ADD instead of
//     hiding under each other (73 % of the pipe); a raised priority for one wave of each pair starves the other
//     instead (same total);
//   * the next k-step's transform interleaved instruction by instruction with the multiplies (sched_group_barrier,
//     MFMA / VALU alternating): 834 cycles per k-step for a wave alone, 7 % slower for the pair -- a VALU
//     instruction between two fp32 MFMAs costs more than its slot.
// The in-register transform costs one VALU per MFMA at 32 output channels; that ratio, not the schedule, is the limit.
template <int NC>
__device__ __forceinline__ void wino_layer(const float *__restrict__ act, const float *__restrict__ U, int CS, int RS,
                                           int wb, int lane, float (&y)[2][4][4]) {
  const float *wbase = act + (lane >> 4) * CS + wb;
  const float *ub = U + lane * 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    floatx4 acc[2][8];
    float d[2][3][4];
    floatx4 u[2][4];
    auto fetch = [&](int buf, int c4) {
      // rows (half 0: 0,1,2; half 1: 1,2,3) of this lane's 4x4 window: rows 2pr-1 .. 2pr+2, columns 2pc-1 .. 2pc+2
      const float *wp = wbase + c4 * 4 * CS + half * RS;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float2 lo = *reinterpret_cast<const float2 *>(wp + i * RS);
        const float2 hi = *reinterpret_cast<const float2 *>(wp + i * RS + 2);
        d[buf][i][0] = lo.x, d[buf][i][1] = lo.y, d[buf][i][2] = hi.x, d[buf][i][3] = hi.y;
      }
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
      // V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]; rows i = 2*half, 2*half + 1
      float t[2][4], v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (half == 0) {
          t[0][j] = d[cur][0][j] - d[cur][2][j];   // d0 - d2
          t[1][j] = d[cur][1][j] + d[cur][2][j];   // d1 + d2
        } else {
          t[0][j] = d[cur][1][j] - d[cur][0][j];   // d2 - d1
          t[1][j] = d[cur][0][j] - d[cur][2][j];   // d1 - d3
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
    // Y = A^T m A,  A^T = [[1,1,1,0],[0,1,-1,-1]]: rows m0, m1 (half 0) / m2, m3 (half 1) of m enter
    // s0 = m0 + m1 + m2 and s1 = m1 - m2 - m3; element r of acc[ct][xi] is cout ct*16 + (lane>>4)*4 + r
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
    // keep the halves apart: interleaved by the scheduler they hold all 128 accumulators at once, and whatever is
    // live across the layer (moved features, left features) spills
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Sums of four values at once over the 32 lanes of each half-wave (lanes 0..31 hold the channels of GroupNorm group
// 2ct, lanes 32..63 of group 2ct+1).  Four DPP steps leave every lane of a 16-lane row with its row's sum;
// row_bcast:15 then adds row 0 into row 1 and row 2 into row 3, so the half-wave sums sit in rows 1 and 3
// (lanes 16..31 / 48..63) -- no LDS crossbar round trip (ds_bpermute) in the chain of dependent steps.
__device__ __forceinline__ void half_wave_sums(float (&s)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] += dpp_mov<0xB1>(s[k]);    // quad_perm [1, 0, 3, 2]
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] += dpp_mov<0x4E>(s[k]);    // quad_perm [2, 3, 0, 1]
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] += dpp_mov<0x141>(s[k]);   // row_half_mirror
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] += dpp_mov<0x140>(s[k]);   // row_mirror
#pragma unroll
  for (int k = 0; k < 4; ++k)                                  // row_bcast:15 into rows 1 and 3 (row_mask 0xA)
    s[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s[k]), 0x142, 0xA, 0xF, false));
}

// y (+bias) -> LeakyReLU(GroupNorm(.)) in place, statistics over the whole workgroup, in two parts around the barrier
// that ends the layer's multiplies anyway (B3 / B7: no barrier of its own).
// Shifted single pass: every lane accumulates sum(x - c) and sum((x - c)^2) with c = the group's mean of the
// previous step (`shift`, identical in all lanes of the group; 0 at the first step), the per-wave sums are combined
// behind the barrier, and var = E[(x-c)^2] - (E[x-c])^2.  The recurrence moves slowly from plane to plane, so c sits
// within a fraction of a standard deviation of the mean and the subtraction cancels nothing of significance
// (with c = 0 it is the plain one-pass formula, still accurate here: |mean| is of the order of the deviation).
// Part 1 (before the barrier): bias, this wave's sums -> its record.  A wave that leaves the multiplies early does
// this while its SIMD partner still multiplies; the record slab of a GroupNorm is rewritten one whole step later.
__device__ __forceinline__ void wino_groupnorm_sums(float (&y)[2][4][4], bool pvalid, const float (&shift)[2],
                                                    const float *__restrict__ bias, float *red_slab, int lane, int wave) {
  const int cbase = (lane >> 4) * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};   // [ct][sum, sum of squares]
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
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
  if (!pvalid) s[0] = s[1] = s[2] = s[3] = 0.f;
  half_wave_sums(s);
  if ((lane & 31) == 16) {
    float *rec = red_slab + (wave * 4 + (lane >> 5)) * 2;       // [wave][group = ct*2 + half][2]
    rec[0] = s[0], rec[1] = s[1];
    rec[4] = s[2], rec[5] = s[3];
  }
}

// Part 2 (behind the barrier): combine the eight records, normalise, activate.
__device__ __forceinline__ void wino_groupnorm_apply(float (&y)[2][4][4], float inv_n, float (&shift)[2],
                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                     const float *red_slab, int lane) {
  const int cbase = (lane >> 4) * 4;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int g = ct * 2 + (lane >> 5);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < CW_WAVES; ++w) {
      const float2 rec = *reinterpret_cast<const float2 *>(red_slab + (w * 4 + g) * 2);
      s1 += rec.x;
      s2 += rec.y;
    }
    const float ms = s1 * inv_n;
    const float var = fmaxf(s2 * inv_n - ms * ms, 0.0f);
    const float mean = shift[ct] + ms;
    shift[ct] = mean;
    const float rstd = 1.0f / sqrtf(var + CW_GN_EPS);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = ct * 16 + cbase + r;
      const float sc = rstd * gamma[c];
      const float sh = beta[c] - mean * sc;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[ct][r][e] = lrelu02(y[ct][r][e] * sc + sh);
    }
  }
}

// ROWS x COLS: the coarse grid as compile-time constants (every LDS access becomes base register + immediate
// offset: no address registers live across the step loop); 0 x 0 = run-time sizes (same code, any supported grid).
// C16: the cost volume is stored as bf16 (ChainArgs::cost_bf16; the bf16 feature tier -- never the parity path).
template <int ROWS, int COLS, bool C16 = false>
__global__ __launch_bounds__(CW_THREADS) void chain_wino_kernel(ChainArgs a, MVSN_VIS10) {   // (MVSN_VIS10: mvsn_common.h)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (chain_gate_closed(a)) return;   // repair launch with nothing to repair (mvsn_chain.h)
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  // Chain of this workgroup.  The S chains of a reference image read the same 64 KB of left features at every step:
  // they take neighbouring slots of ONE XCD's range (workgroup b is observed on XCD b % 8), so they run at the same
  // time next to one copy in that L2 -- with n = blockIdx.x they sat on the same XCD a whole round apart, and the copy
  // was evicted and re-fetched in between.  A permutation of the chains: placement only, results identical.
  int n = blockIdx.x;
#ifndef MVSN_CW_NO_XCD_PAIRS
  {
    const int S = (int)gridDim.x / a.B;
    if (S > 1 && S * a.B == (int)gridDim.x) {
      const int t = xcd_tile_index(blockIdx.x, gridDim.x);
      n = (t % S) * a.B + t / S;
    }
  }
#endif
  const int rows = ROWS ? ROWS : a.rows, cols = COLS ? COLS : a.cols;
  const int P = rows * cols, RS = cols + 2, CS = (rows + 1) * RS, D = a.D;
  int tid = tid0;
  constexpr int IMG_IT = ROWS ? (ROWS * COLS + CW_THREADS - 1) / CW_THREADS : 2;   // pixels per thread (image plane pass)
  const int pcols = cols >> 1, NPT = (rows >> 1) * pcols;

  float *U = smem;
  float *sparams = U + CW_U0_FLOATS;
  float *red = sparams + CH_SP_FLOATS;
  float *maskb = red + CW_RED_FLOATS;
  const int Ppad = (P + 3) & ~3;
  float *act = maskb + Ppad;
  const int act_floats = 36 * CS + RS + 2;

  const float *upk = a.packed + CH_DIRECT_FLOATS;
  int lane16 = lane * 16;   // byte offset of this lane inside a 1 KB DMA run (made opaque per step, see below)
  auto dma_u = [&](const float *src, int nchunks) {   // 1 KB runs, wave w takes runs w, w + 8, ...
    const int runs = nchunks * (CW_UCHUNK / 256);
    const char *base = reinterpret_cast<const char *>(src + (size_t)wave * 256);   // wave-uniform (SGPR pair)
    for (int run = wave, i = 0; run < runs; run += CW_WAVES, ++i)
      __builtin_amdgcn_global_load_lds(CW_GPTR(base + (size_t)i * (CW_WAVES * 1024) + (unsigned)lane16),
                                       CW_LPTR(U + run * 256), 16, 0, 0);
  };
  auto dma_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // ---- one-time set-up ---------------------------------------------------------------------
  dma_u(upk, 9);
  for (int i = tid; i < act_floats; i += CW_THREADS) act[i] = 0.0f;
  for (int i = tid; i < CH_SP_FLOATS; i += CW_THREADS) sparams[i] = a.packed[CH_W0_FLOATS + 2 * CH_W1_FLOATS + i];
  const float *bias0 = sparams, *gn0w = sparams + 32, *gn0b = sparams + 64, *bias1 = sparams + 96,
              *gn1w = sparams + 128, *gn1b = sparams + 160, *bias2 = sparams + 192;

  // this lane's patch: q = wave*16 + (lane & 15); its 2x2 outputs (2pr + a, 2pc + b), element e = a*2 + b
  const int q = wave * 16 + (lane & 15);
  const bool pvalid = q < NPT;
  const int qq = pvalid ? q : 0;
  const int pr = qq / pcols, pc = qq - pr * pcols;
  const int wb = (2 * pr) * RS + 2 * pc;                 // window origin inside a channel plane
  const int ob = wb + RS + 1;                            // output (0,0); (a,b) at ob + a*RS + b
  const int cbase = (lane >> 4) * 4;                     // this lane's couts: ct*16 + cbase + r
  const bool tile_live = wave * 16 < NPT;                // wave-uniform
  const float inv_n = 1.0f / (8.0f * (float)P);          // values per GroupNorm group: 8 channels x P pixels
  float shift0[2] = {0.f, 0.f}, shift1[2] = {0.f, 0.f};  // previous step's group means (GroupNorm 0 / 1, per cout tile)
  __syncthreads();

  const float *f0 = a.f0 + (size_t)n * 32 * P;
  const float *flp = a.fl + (size_t)(n % a.B) * 32 * P;
  for (int i = tid; i < 32 * P; i += CW_THREADS) {
    const int c = i / P, p = i - c * P;
    const int yy = p / cols, xx = p - yy * cols;
    act[(3 + c) * CS + (yy + 1) * RS + xx + 1] = f0[i];
  }

  const float *Hn = a.H + (size_t)n * D * 9;
  const float *Hin = a.Hinc + (size_t)n * D * 9;
  const float *src = a.src + (size_t)n * 3 * P;
  uint8_t *maskg = a.mask + (size_t)n * D * P;
  typedef typename ChainCost<C16>::type cost_t;
  cost_t *costg = reinterpret_cast<cost_t *>(a.cost) + (size_t)n * 32 * D * P;
  float *fvolg = a.fvol ? a.fvol + (size_t)n * 32 * D * P : nullptr;

  // ---- plane 0: mask from the plane's homography ------------------------------------------------
  {
    float Hl[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hl[i] = Hn[i];
    for (int p = tid; p < P; p += CW_THREADS) {
      WarpCoord c = warp_coord(Hl, (float)(p % cols), (float)(p / cols), (float)rows, (float)cols);
      maskb[p] = c.outside ? 1.0f : 0.0f;
      maskg[p] = c.outside ? 1 : 0;
    }
  }
  __syncthreads();

  // Cost-volume slice of plane `dd` from the LDS-resident features: coalesced 16-byte HBM traffic (left features
  // in, cost out; the stores are streaming so that the left features and the weights stay in L2).
  auto write_cost_slice = [&](int dd) {
    if ((cols & 3) == 0) {
      const int quads = P >> 2;
      for (int i = tid; i < 32 * quads; i += CW_THREADS) {
        const int c = i / quads, p4 = (i - c * quads) * 4;
        const int yy = p4 / cols, xx = p4 - yy * cols;
        const float *fr = act + (3 + c) * CS + (yy + 1) * RS + xx + 1;
        const floatx4 l = *reinterpret_cast<const floatx4 *>(flp + (size_t)c * P + p4);
        const floatx4 m = *reinterpret_cast<const floatx4 *>(maskb + p4);
        floatx4 cst, ftr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float f = fr[k];
          cst[k] = m[k] != 0.0f ? 0.0f : fabsf(l[k] - f);
          ftr[k] = m[k] != 0.0f ? 0.0f : f;
        }
        chain_cost_nt(costg + ((size_t)c * D + dd) * P + p4, cst);
        if (fvolg) __builtin_nontemporal_store(ftr, reinterpret_cast<floatx4 *>(fvolg + ((size_t)c * D + dd) * P + p4));
      }
    } else {
      for (int i = tid; i < 32 * P; i += CW_THREADS) {
        const int c = i / P, p = i - c * P;
        const int yy = p / cols, xx = p - yy * cols;
        const float f = act[(3 + c) * CS + (yy + 1) * RS + xx + 1];
        const bool out = maskb[p] != 0.0f;
        chain_cost_st(costg + ((size_t)c * D + dd) * P + p, out ? 0.0f : fabsf(flp[(size_t)c * P + p] - f));
        if (fvolg) fvolg[((size_t)c * D + dd) * P + p] = out ? 0.0f : f;
      }
    }
  };

  // plane 0 (the extractor's features) goes out through the generic pass; planes 1..D-1 are written by the lanes
  // that produce them, straight from registers: each lane fetches the left features of its own (8 channels x 2x2
  // pixels) outputs as 8-byte pieces (L2 hits: 64 KB per reference image, shared by its S chains; the cost stores
  // are streaming and do not evict them) ahead of the barrier that ends the last layer
  write_cost_slice(0);
  // (qq = 0 for lanes without a patch: always a valid address, the loads below are unconditional)
  int fl_off = cbase * P + (2 * pr) * cols + 2 * pc;   // (an offset, not a pointer: an opaque pointer loses its address space)
  int slice_off = (cbase * D) * P + (2 * pr) * cols + 2 * pc;   // this lane's origin inside a chain's cost volume

  // ---- the recurrence ------------------------------------------------------------------------
#define CW_STAMP(i)                                                                     \
  do {                                                                                  \
    if (a.dbg && blockIdx.x == 0 && tid == 0 && d <= 4) a.dbg[(d - 1) * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#define CW_WSTAMP(i)                                                                    \
  do {                                                                                  \
    if (a.dbg && blockIdx.x == 0 && lane == 0 && d == 3) a.dbg[64 + wave * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
  for (int d = 1; d < D; ++d) {
    // per-thread index arithmetic of the pixel / slice loops is recomputed every step from an opaque copy of the
    // thread id: hoisted out of the loop it would occupy dozens of VGPRs across the convolutions (and spill)
    // ... and the left-feature pointer: the sixteen 64-bit addresses derived from it (offsets beyond the immediate
    // range) were hoisted and spilled, and every reload in front of a load is an s_waitcnt vmcnt(0) -- the sixteen
    // loads ran one round trip after the other (9 k cycles per step); likewise the lane id the GroupNorm's LDS
    // addresses derive from (their reloads waited for the next layer's U to land)
    asm volatile("" : "+v"(tid), "+v"(lane16), "+v"(slice_off), "+v"(fl_off));
    const int lane_s = tid & 63;
    const float *fl_lane = flp + fl_off;
    CW_STAMP(0);
#ifndef MVSN_CW_NO_HTOUCH
    // The two 36-byte homographies of a step are scalar loads at the top of the step; every other step they start a new
    // 64-byte line.  Touch the NEXT step's lines now: four scalar loads into registers nothing reads, held (tied into the
    // first barrier's wait below) until they have landed.  Measured (tools/chain_bench.py, three interleaved runs):
    // 2.403-2.407 ms per 256 chains against 2.413-2.423 without (profiles/r05_slab/README.md).
    float ht0 = 0.f, ht1 = 0.f, ht2 = 0.f, ht3 = 0.f;
    if (d + 1 < D)
      asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x20\n\ts_load_dword %2, %5, 0x0\n\ts_load_dword %3, %5, 0x20"
                   : "=s"(ht0), "=s"(ht1), "=s"(ht2), "=s"(ht3)
                   : "s"(Hn + (d + 1) * 9), "s"(Hin + (d + 1) * 9));
#endif

    // A1: image plane d and its mask (global gathers; the 6 KB source image stays in L1/L2)
    float img[IMG_IT][3], mk[IMG_IT];
    {
      float Hl[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Hl[i] = Hn[d * 9 + i];
#pragma unroll
      for (int it = 0; it < IMG_IT; ++it) {
        const int p = tid + it * CW_THREADS;
        if (p < P) {
          WarpCoord c = warp_coord(Hl, (float)(p % cols), (float)(p / cols), (float)rows, (float)cols);
          Bilinear b = bilinear_taps(c.ix, c.iy, rows, cols);
          const float keep = c.outside ? 0.0f : 1.0f;
          mk[it] = c.outside ? 1.0f : 0.0f;
          const int o00 = b.y0 * cols + b.x0, o01 = b.y0 * cols + b.x1, o10 = b.y1 * cols + b.x0,
                    o11 = b.y1 * cols + b.x1;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float *ic = src + (size_t)ch * P;
            img[it][ch] = keep * (ic[o00] * b.w00 + ic[o01] * b.w01 + ic[o10] * b.w10 + ic[o11] * b.w11);
          }
        }
      }
    }

    // A2: previous plane's features moved by the incremental homography (gather from LDS).  The +1 taps are read
    // unclamped: when the clamp would act their weight is exactly zero and the slot read is a zero halo.
    float fp[2][4][4];
    {
      float Hl[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Hl[i] = Hin[d * 9 + i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float px = (float)(2 * pc + (e & 1)), py = (float)(2 * pr + (e >> 1));
        WarpCoord c = warp_coord(Hl, px, py, (float)rows, (float)cols);
        Bilinear b = bilinear_taps(c.ix, c.iy, rows, cols);
        const float keep = c.outside ? 0.0f : 1.0f;
        const float w00 = keep * b.w00, w01 = keep * b.w01, w10 = keep * b.w10, w11 = keep * b.w11;
        const int o = (b.y0 + 1) * RS + b.x0 + 1;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float *fc = act + (3 + ct * 16 + cbase + r) * CS + o;
            fp[ct][r][e] = fc[0] * w00 + fc[1] * w01 + fc[RS] * w10 + fc[RS + 1] * w11;
          }
      }
    }
    CW_STAMP(1);
    cw_barrier();  // B1: every gather of plane d-1 is done
#ifndef MVSN_CW_NO_HTOUCH
    asm volatile("" : "+s"(ht0), "+s"(ht1), "+s"(ht2), "+s"(ht3));   // (behind the barrier's s_waitcnt lgkmcnt(0))
#endif
    CW_STAMP(2);

    // A3: lay out the refiner input [image(3) | moved features(32)]
    if (pvalid) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float *dst = act + (3 + ct * 16 + cbase + r) * CS + ob;
          dst[0] = fp[ct][r][0], dst[1] = fp[ct][r][1], dst[RS] = fp[ct][r][2], dst[RS + 1] = fp[ct][r][3];
        }
    }
#pragma unroll
    for (int it = 0; it < IMG_IT; ++it) {
      const int p = tid + it * CW_THREADS;
      if (p < P) {
        const int yy = p / cols, xx = p - yy * cols;
        const int o = (yy + 1) * RS + xx + 1;
        act[0 * CS + o] = img[it][0];
        act[1 * CS + o] = img[it][1];
        act[2 * CS + o] = img[it][2];
        maskb[p] = mk[it];
        maskg[(size_t)d * P + p] = mk[it] != 0.0f ? 1 : 0;
      }
    }
    dma_landed();     // conv0's U (issued behind the previous step's conv2, or in the set-up)
    cw_barrier();  // B2
    CW_STAMP(3);

    float y[2][4][4] = {};
    CW_WSTAMP(0);
    if (tile_live) wino_layer<9>(act, U, CS, RS, wb, lane, y);
    wino_groupnorm_sums(y, pvalid && tile_live, shift0, bias0, red, lane_s, wave);
    CW_STAMP(4);
    CW_WSTAMP(1);
    cw_barrier();  // B3: act and U free, GroupNorm records published
    CW_WSTAMP(2);
    dma_u(upk + CW_U0_FLOATS, 8);
    CW_WSTAMP(6);
    CW_STAMP(5);

    wino_groupnorm_apply(y, inv_n, shift0, gn0w, gn0b, red, lane_s);
    if (pvalid) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float *dst = act + (ct * 16 + cbase + r) * CS + ob;
          dst[0] = y[ct][r][0], dst[1] = y[ct][r][1], dst[RS] = y[ct][r][2], dst[RS + 1] = y[ct][r][3];
        }
    }
    CW_WSTAMP(3);
    dma_landed();
    CW_WSTAMP(4);
    cw_barrier();  // B6
    CW_WSTAMP(5);
    CW_STAMP(6);

    if (tile_live) wino_layer<8>(act, U, CS, RS, wb, lane, y);
    wino_groupnorm_sums(y, pvalid && tile_live, shift1, bias1, red + CW_RED_FLOATS / 2, lane_s, wave);
    CW_STAMP(7);
    cw_barrier();  // B7 (+ records)
    dma_u(upk + CW_U0_FLOATS + CW_U1_FLOATS, 8);
    CW_STAMP(8);

    wino_groupnorm_apply(y, inv_n, shift1, gn1w, gn1b, red + CW_RED_FLOATS / 2, lane_s);
    if (pvalid) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // x2 = x1 + LReLU(GN(conv1(x1)))
          float *dst = act + (ct * 16 + cbase + r) * CS + ob;
          dst[0] += y[ct][r][0], dst[1] += y[ct][r][1], dst[RS] += y[ct][r][2], dst[RS + 1] += y[ct][r][3];
        }
    }
    dma_landed();
    cw_barrier();  // B10
    CW_STAMP(9);

    if (tile_live) wino_layer<8>(act, U, CS, RS, wb, lane, y);
    float2 fl[2][4][2];   // left features of this lane's outputs, in flight across the barrier
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
          fl[ct][r][a2] = *reinterpret_cast<const float2 *>(fl_lane + (size_t)(ct * 16 + r) * P + a2 * cols);
    CW_STAMP(10);
    cw_barrier();  // B11
    dma_u(upk, 9);    // conv0 of the next step
    CW_STAMP(11);

    // epilogue: the new features become the next step's gather source, and their cost slice
    // (not mask) * |left - right| leaves for HBM straight from the registers (8-byte pieces, 128-byte runs per
    // 16 lanes; streaming stores)
    if (pvalid) {
      const float2 m0 = *reinterpret_cast<const float2 *>(maskb + (2 * pr) * cols + 2 * pc);
      const float2 m1 = *reinterpret_cast<const float2 *>(maskb + (2 * pr + 1) * cols + 2 * pc);
      const bool out[4] = {m0.x != 0.0f, m0.y != 0.0f, m1.x != 0.0f, m1.y != 0.0f};
      cost_t *cd = costg + (size_t)d * P;
      float *fd = fvolg ? fvolg + (size_t)d * P : nullptr;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float b2 = bias2[ct * 16 + cbase + r];
          float f[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = fp[ct][r][e] + (y[ct][r][e] + b2);
          float *dst = act + (3 + ct * 16 + cbase + r) * CS + ob;
          dst[0] = f[0], dst[1] = f[1], dst[RS] = f[2], dst[RS + 1] = f[3];
          cost_t *cdst = cd + ((ct * 16 + r) * D) * P + slice_off;
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2) {
            float2v c2;
            c2.x = out[a2 * 2] ? 0.0f : fabsf(fl[ct][r][a2].x - f[a2 * 2]);
            c2.y = out[a2 * 2 + 1] ? 0.0f : fabsf(fl[ct][r][a2].y - f[a2 * 2 + 1]);
            chain_cost_nt(cdst + a2 * cols, c2);
          }
          if (fd) {
            float *fdst = fd + ((ct * 16 + r) * D) * P + slice_off;
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2) {
              float2v f2;
              f2.x = out[a2 * 2] ? 0.0f : f[a2 * 2];
              f2.y = out[a2 * 2 + 1] ? 0.0f : f[a2 * 2 + 1];
              __builtin_nontemporal_store(f2, reinterpret_cast<float2v *>(fdst + a2 * cols));
            }
          }
        }
    }
    CW_STAMP(12);
    cw_barrier();  // B12
    CW_STAMP(13);
  }
#undef CW_STAMP
#undef CW_WSTAMP
  dma_landed();       // the last step's look-ahead fetch must not outlive the workgroup's LDS
}

int chain_wino_launch(const ChainArgs &a, int n_chains, hipStream_t stream) {
  const size_t lds = chain_wino_lds_bytes(a.rows, a.cols);
#define CW_LAUNCH(R, C, C16)                                                                                         \
  do {                                                                                                               \
    static LdsOptIn opt;                                                                                             \
    if (int rc = ensure_lds(opt, (const void *)chain_wino_kernel<R, C, C16>, lds, "mvsn_incremental_cost_volume(winograd)")) \
      return rc;                                                                                                     \
    hipLaunchKernelGGL((chain_wino_kernel<R, C, C16>), dim3(n_chains), dim3(CW_THREADS), lds, stream, a, CHAIN_VISIBLE_G(a)); \
  } while (0)
  if (a.cost_bf16) {                                     // (bf16 feature tier)
    if (a.rows == 16 && a.cols == 32) CW_LAUNCH(16, 32, true);
    else CW_LAUNCH(0, 0, true);
  } else if (a.rows == 16 && a.cols == 32) CW_LAUNCH(16, 32, false);   // 512x256 frames (BASELINE configs 2, 3 and the headline)
  else CW_LAUNCH(0, 0, false);
#undef CW_LAUNCH
  return check_launch("mvsn_incremental_cost_volume(winograd)");
}

}  // namespace mvsn
