// The fused incremental chain (see include/mvsn_hip.h: mvsn_incremental_cost_volume).
//
// One 1024-thread workgroup (16 waves, 4 per SIMD) owns one (reference image, source view)
// chain for all D-1 sequential steps.  Per step: homography gather of the previous plane's
// features out of LDS, three 3x3 convolutions as implicit GEMMs on v_mfma_f32_16x16x4_f32
// (A = weights 16 couts x 4 cins, B = activations 4 cins x 16 pixels), two GroupNorm(4)
// reductions, and the cost-volume slice written straight to HBM.  Nothing but the cost / mask
// slices (and the optional feature volume) leaves the CU between steps.
//
// LDS plan (floats), P = rows*cols, RS = cols+1, G = RS+1, CS = rows*RS+G rounded to 16 mod 32:
//   wbuf   [10368]      weight fragments of the conv that runs next, [tap][cin/4][cout/16][lane]
//   sparams[224]        biases and GroupNorm affine
//   red    [4][16][4]   cross-wave reduction scratch (one slab per reduction of a step)
//   maskb  [P]          out-of-image flag of the current plane
//   act    G + 36*CS    activation planes.  Each channel is rows x RS with one zero column
//                       closing every row (it doubles as the left halo of the next row) and G
//                       zeros between channels (top/bottom halo), so every 3x3 tap of every
//                       pixel is a plain offset read with no bounds test.  CS = 16 (mod 32)
//                       makes the 16-pixel x 4-channel B-fragment read conflict-free.
//                       ch 0..2 image plane d, ch 3..34 features, ch 35 zero (K padding).
// When that exceeds 160 KiB the activation planes move to a per-chain global workspace
// (L2-resident); the code path is otherwise identical.
#include "mvsn_chain.h"
#include "mvsn_common.h"
#include "mvsn_conv_wino.h"

namespace mvsn {

constexpr int CH_THREADS = 1024;
constexpr int CH_WAVES = 16;
constexpr int W0_FLOATS = CH_W0_FLOATS;
constexpr int W1_FLOATS = CH_W1_FLOATS;
constexpr int SP_FLOATS = CH_SP_FLOATS;
constexpr int PACKED_FLOATS = CH_PACKED_FLOATS;
constexpr int RED_FLOATS = 4 * CH_WAVES * 4;
constexpr float GN_EPS = 1e-5f;

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
__global__ void pack_refiner_kernel(const float *c0w, const float *c0b, const float *g0w, const float *g0b,
                                    const float *c1w, const float *c1b, const float *g1w, const float *g1b,
                                    const float *c2w, const float *c2b, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= CH_STEPS_OFFSET) return;   // (the stepwise form's region is written by wino_pack_2d)
  if (i >= CH_DIRECT_FLOATS) {
    // Winograd form (mvsn_chain_wino.hip): U = G g G^T, [conv][k-step][cout tile][xi quad][lane][4 xi];
    // lane = k*16 + c holds U_xi[cout = t*16 + c][cin = 4*kstep + k] (the A fragment of the MFMA)
    int rel = i - CH_DIRECT_FLOATS, cin_total = 35;
    const float *w = c0w;
    if (rel >= CW_U0_FLOATS) {
      rel -= CW_U0_FLOATS, cin_total = 32, w = c1w;
      if (rel >= CW_U1_FLOATS) rel -= CW_U1_FLOATS, w = c2w;
    }
    const int j = rel & 3, lane = (rel >> 2) & 63, xq = (rel >> 8) & 3, t = (rel >> 10) & 1, c4 = rel >> 11;
    const int xi = xq * 4 + j, cout = t * 16 + (lane & 15), cin = c4 * 4 + (lane >> 4);
    float u = 0.0f;
    if (cin < cin_total) {
      const float *g = w + ((size_t)cout * cin_total + cin) * 9;
      const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
      const int gi = xi >> 2, gj = xi & 3;
      for (int aa = 0; aa < 3; ++aa)
        for (int bb = 0; bb < 3; ++bb) u += G[gi][aa] * g[aa * 3 + bb] * G[gj][bb];
    }
    out[i] = u;
  } else if (i < W0_FLOATS + 2 * W1_FLOATS) {
    int conv, rel, cin_total, nc;
    const float *w;
    if (i < W0_FLOATS) {
      conv = 0, rel = i, cin_total = 35, nc = 9, w = c0w;
    } else if (i < W0_FLOATS + W1_FLOATS) {
      conv = 1, rel = i - W0_FLOATS, cin_total = 32, nc = 8, w = c1w;
    } else {
      conv = 2, rel = i - W0_FLOATS - W1_FLOATS, cin_total = 32, nc = 8, w = c2w;
    }
    (void)conv;
    const int lane = rel & 63;
    const int t = (rel >> 6) & 1;
    const int c4 = (rel >> 7) % nc;
    const int tap = (rel >> 7) / nc;
    const int cout = t * 16 + (lane & 15);
    const int cin = c4 * 4 + (lane >> 4);
    out[i] = cin < cin_total ? w[((size_t)cout * cin_total + cin) * 9 + tap] : 0.0f;
  } else {
    const int r = i - (W0_FLOATS + 2 * W1_FLOATS);
    const int which = r / 32, c = r % 32;
    const float *srcs[7] = {c0b, g0w, g0b, c1b, g1w, g1b, c2b};
    out[i] = srcs[which][c];
  }
}

// ---------------------------------------------------------------------------------------------
// pieces of the step
// ---------------------------------------------------------------------------------------------
template <int TP, int NC>
__device__ __forceinline__ void conv3x3_mfma(const float *__restrict__ act, const float *__restrict__ wbuf, int CS,
                                             int RS, const int (&qb)[TP], int lane, floatx4 (&acc)[TP][2]) {
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    acc[j][0] = floatx4{0.f, 0.f, 0.f, 0.f};
    acc[j][1] = floatx4{0.f, 0.f, 0.f, 0.f};
  }
  const float *abase = act + (lane >> 4) * CS;
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int off = (tap / 3 - 1) * RS + (tap % 3 - 1);
    const float *wt = wbuf + tap * NC * 128 + lane;
    const float *ab = abase + off;
#pragma unroll
    for (int c4 = 0; c4 < NC; ++c4) {
      const float w0 = wt[c4 * 128];
      const float w1 = wt[c4 * 128 + 64];
#pragma unroll
      for (int j = 0; j < TP; ++j) {
        const float b = ab[c4 * 4 * CS + qb[j]];
        acc[j][0] = mfma16x16x4(w0, b, acc[j][0]);
        acc[j][1] = mfma16x16x4(w1, b, acc[j][1]);
      }
    }
  }
}

// Same convolution when the activation planes live in the global workspace (coarse grids that do not
// fit LDS): each 4-channel slab (with its halo guards) is staged into one of two LDS buffers while the
// previous slab's 9 taps run, so the MFMA loop still reads its B fragments from LDS.  One barrier per
// slab.  `slab` holds 2 x slab_floats, slab_floats = G + 4*CS.
template <int TP, int NC>
__device__ __forceinline__ void conv3x3_mfma_slab(const float *__restrict__ act, const float *__restrict__ wbuf,
                                                  float *__restrict__ slab, int CS, int RS, const int (&qb)[TP],
                                                  int tid, floatx4 (&acc)[TP][2]) {
  const int lane = tid & 63;
  const int G = RS + 1;
  const int slab_floats = G + 4 * CS;
  constexpr int PER = 9;  // staged floats per thread per slab (<= 9216 floats per slab)
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    acc[j][0] = floatx4{0.f, 0.f, 0.f, 0.f};
    acc[j][1] = floatx4{0.f, 0.f, 0.f, 0.f};
  }
  float stage[PER];
  auto fetch = [&](int c4) {
    const float *src = act + c4 * 4 * CS - G;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * CH_THREADS;
      stage[k] = i < slab_floats ? src[i] : 0.0f;
    }
  };
  auto commit = [&](int buf) {
    float *dst = slab + buf * slab_floats;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * CH_THREADS;
      if (i < slab_floats) dst[i] = stage[k];
    }
  };
  fetch(0);
  commit(0);
  __syncthreads();
  for (int c4 = 0; c4 < NC; ++c4) {
    if (c4 + 1 < NC) fetch(c4 + 1);
    const float *sb = slab + (c4 & 1) * slab_floats + G + (lane >> 4) * CS;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int off = (tap / 3 - 1) * RS + (tap % 3 - 1);
      const float *wt = wbuf + (tap * NC + c4) * 128 + lane;
      const float w0 = wt[0];
      const float w1 = wt[64];
#pragma unroll
      for (int j = 0; j < TP; ++j) {
        const float b = sb[qb[j] + off];
        acc[j][0] = mfma16x16x4(w0, b, acc[j][0]);
        acc[j][1] = mfma16x16x4(w1, b, acc[j][1]);
      }
    }
    if (c4 + 1 < NC) commit((c4 + 1) & 1);
    __syncthreads();
  }
}

// Sum `v[t]` (t = 0,1: the two cout tiles) over the whole workgroup, per GroupNorm group.
// Lanes 0..31 of a wave hold channels of group 2t, lanes 32..63 of group 2t+1.
__device__ __forceinline__ void block_group_sum(float (&v)[2], float *red_slab, int lane, int wave) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float s = v[t];
    s += dpp_mov<0xB1>(s);    // quad_perm [1, 0, 3, 2]
    s += dpp_mov<0x4E>(s);    // quad_perm [2, 3, 0, 1]
    s += dpp_mov<0x141>(s);   // row_half_mirror: the other quad of the half row
    s += dpp_mov<0x140>(s);   // row_mirror: the other half of the 16-lane row
    s += __shfl_xor(s, 16, 64);
    v[t] = s;
  }
  if ((lane & 31) == 0) {
    const int hi = lane >> 5;
    red_slab[wave * 4 + 0 + hi] = v[0];
    red_slab[wave * 4 + 2 + hi] = v[1];
  }
  __syncthreads();
  const int hi = lane >> 5;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int w = 0; w < CH_WAVES; ++w) {
    s0 += red_slab[w * 4 + 0 + hi];
    s1 += red_slab[w * 4 + 2 + hi];
  }
  v[0] = s0;
  v[1] = s1;
}

// acc(+bias) -> LeakyReLU(GroupNorm(.)) in place.
template <int TP>
__device__ __forceinline__ void groupnorm_lrelu(floatx4 (&acc)[TP][2], const bool (&valid)[TP],
                                                const float *__restrict__ bias, const float *__restrict__ gamma,
                                                const float *__restrict__ beta, float *red_a, float *red_b,
                                                float inv_count, int lane, int wave) {
  const int cbase = (lane >> 4) * 4;
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < TP; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[j][t][r] += bias[t * 16 + cbase + r];
        if (valid[j]) s[t] += acc[j][t][r];
      }
  block_group_sum(s, red_a, lane, wave);
  const float mean[2] = {s[0] * inv_count, s[1] * inv_count};
  float q[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < TP; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dv = acc[j][t][r] - mean[t];
        if (valid[j]) q[t] += dv * dv;
      }
  block_group_sum(q, red_b, lane, wave);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float rstd = 1.0f / sqrtf(q[t] * inv_count + GN_EPS);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = t * 16 + cbase + r;
      const float sc = rstd * gamma[c];
      const float sh = beta[c] - mean[t] * sc;
#pragma unroll
      for (int j = 0; j < TP; ++j) acc[j][t][r] = lrelu02(acc[j][t][r] * sc + sh);
    }
  }
}

template <int WFLOATS>
__device__ __forceinline__ void load_weights_to_regs(const float *__restrict__ g, floatx4 (&wreg)[3], int tid) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int idx = (tid + k * CH_THREADS) * 4;
    if (idx < WFLOATS) wreg[k] = *reinterpret_cast<const floatx4 *>(g + idx);
  }
}
template <int WFLOATS>
__device__ __forceinline__ void store_weights_to_lds(float *wbuf, const floatx4 (&wreg)[3], int tid) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int idx = (tid + k * CH_THREADS) * 4;
    if (idx < WFLOATS) *reinterpret_cast<floatx4 *>(wbuf + idx) = wreg[k];
  }
}

template <int TP, bool LDS_ACT, bool C16 = false>   // C16: cost volume stored as bf16 (ChainArgs::cost_bf16, bf16 feature tier)
__global__ __launch_bounds__(CH_THREADS) void chain_kernel(ChainArgs a, MVSN_VIS10) {   // (MVSN_VIS10: mvsn_common.h)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (chain_gate_closed(a)) return;   // repair launch with nothing to repair (mvsn_chain.h)
  int tid = threadIdx.x;              // (made opaque at the top of every step, see the step loop)
  const int lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x;
  const int rows = a.rows, cols = a.cols, P = rows * cols, RS = cols + 1, G = RS + 1, CS = a.CS, D = a.D;
  constexpr int IMG_IT = (TP * 256 + CH_THREADS - 1) / CH_THREADS;

  float *wbuf = smem;
  float *sparams = wbuf + W0_FLOATS;
  float *red = sparams + SP_FLOATS;
  float *maskb = red + RED_FLOATS;
  const int Ppad = (P + 3) & ~3;
  float *act;
  float *slab = nullptr;
  const int act_floats = G + 36 * CS;
  // moved features of the step (the epilogue adds the refiner's output to them): 32 x P floats per chain in the global
  // workspace, written and read back by the same thread -- held in registers across the three convolutions they cost
  // 8 VGPRs per tile, which a 1024-thread workgroup (128 registers per lane) does not have
  float *mv;
  if constexpr (LDS_ACT) {
    act = maskb + Ppad + G;
    mv = a.workspace + (size_t)n * 32 * P;
  } else {
    act = a.workspace + (size_t)n * (act_floats + 32 * P) + G;
    mv = a.workspace + (size_t)n * (act_floats + 32 * P) + act_floats;
    slab = maskb + Ppad;   // 2 x (G + 4*CS) floats
  }

  // ---- one-time set-up ---------------------------------------------------------------------
  for (int i = tid; i < act_floats; i += CH_THREADS) act[i - G] = 0.0f;
  for (int i = tid; i < SP_FLOATS; i += CH_THREADS) sparams[i] = a.packed[W0_FLOATS + 2 * W1_FLOATS + i];
  const float *bias0 = sparams, *gn0w = sparams + 32, *gn0b = sparams + 64, *bias1 = sparams + 96,
              *gn1w = sparams + 128, *gn1b = sparams + 160, *bias2 = sparams + 192;

  int pb[TP], qb[TP];
  bool valid[TP];
  int cbase;   // this lane's channels: t*16 + cbase + r
  auto tile_indices = [&](int t_) {   // this thread's pixels of its wave's TP tiles (recomputed per step, see the loop)
    const int lane_ = t_ & 63, wave_ = t_ >> 6;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
      const int p = (wave_ * TP + j) * 16 + (lane_ & 15);
      valid[j] = p < P;
      pb[j] = valid[j] ? p : 0;
      qb[j] = (pb[j] / cols) * RS + (pb[j] % cols);
    }
    cbase = (lane_ >> 4) * 4;
  };
  tile_indices(tid);
  __syncthreads();

  const float *f0 = a.f0 + (size_t)n * 32 * P;
  const float *flp = a.fl + (size_t)(n % a.B) * 32 * P;
#pragma unroll
  for (int j = 0; j < TP; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = t * 16 + cbase + r;
        if (valid[j]) act[(3 + c) * CS + qb[j]] = f0[(size_t)c * P + pb[j]];
      }

  const float *Hn = a.H + (size_t)n * D * 9;
  const float *Hin = a.Hinc + (size_t)n * D * 9;
  const float *src = a.src + (size_t)n * 3 * P;
  uint8_t *maskg = a.mask + (size_t)n * D * P;
  typedef typename ChainCost<C16>::type cost_t;
  cost_t *costg = reinterpret_cast<cost_t *>(a.cost) + (size_t)n * 32 * D * P;
  float *fvolg = a.fvol ? a.fvol + (size_t)n * 32 * D * P : nullptr;

  // ---- plane 0: mask and cost from the extractor's features ---------------------------------
  {
    float Hl[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hl[i] = Hn[i];
    for (int p = tid; p < P; p += CH_THREADS) {
      WarpCoord c = warp_coord(Hl, (float)(p % cols), (float)(p / cols), (float)rows, (float)cols);
      maskb[p] = c.outside ? 1.0f : 0.0f;
      maskg[p] = c.outside ? 1 : 0;
    }
  }
  __syncthreads();

  // Cost-volume slice of plane `dd` from the LDS-resident features: coalesced 16-byte HBM traffic
  // (left features in, cost out), rows of the padded plane read back as 4 scalars.  Runs while the
  // plane's features and mask are still in place, i.e. before barrier B1 of the following step.
  auto write_cost_slice = [&](int dd) {
    if ((cols & 3) == 0) {
      const int quads = P >> 2;
      for (int i = tid; i < 32 * quads; i += CH_THREADS) {
        const int c = i / quads, p4 = (i - c * quads) * 4;
        const int y = p4 / cols, x = p4 - y * cols;
        const float *fr = act + (3 + c) * CS + y * RS + x;
        const floatx4 l = *reinterpret_cast<const floatx4 *>(flp + (size_t)c * P + p4);
        const floatx4 m = *reinterpret_cast<const floatx4 *>(maskb + p4);
        floatx4 cst, ftr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float f = fr[k];
          cst[k] = m[k] != 0.0f ? 0.0f : fabsf(l[k] - f);
          ftr[k] = m[k] != 0.0f ? 0.0f : f;
        }
        chain_cost_st(costg + ((size_t)c * D + dd) * P + p4, cst);
        if (fvolg) *reinterpret_cast<floatx4 *>(fvolg + ((size_t)c * D + dd) * P + p4) = ftr;
      }
    } else {
      for (int i = tid; i < 32 * P; i += CH_THREADS) {
        const int c = i / P, p = i - c * P;
        const float f = act[(3 + c) * CS + (p / cols) * RS + (p % cols)];
        const bool out = maskb[p] != 0.0f;
        chain_cost_st(costg + ((size_t)c * D + dd) * P + p, out ? 0.0f : fabsf(flp[(size_t)c * P + p] - f));
        if (fvolg) fvolg[((size_t)c * D + dd) * P + p] = out ? 0.0f : f;
      }
    }
  };
  const float inv_count = 1.0f / (8.0f * (float)P);

  // ---- the recurrence ------------------------------------------------------------------------
#define MVSN_STAMP(i)                                                                   \
  do {                                                                                  \
    if (a.dbg && blockIdx.x == 0 && tid == 0 && d <= 4) a.dbg[(d - 1) * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
  for (int d = 1; d < D; ++d) {
    // Per-thread addresses are recomputed every step from an opaque copy of the thread id: hoisted out of the loop
    // (72 of them in the smallest instantiation) they do not fit the 128 registers of a 1024-thread workgroup and go
    // to scratch -- and every reload in front of a load is an s_waitcnt vmcnt(0) (HISTORY 11.9, same disease).
    asm volatile("" : "+v"(tid));
    tile_indices(tid);
    const int lane = tid & 63, wave = tid >> 6;
    MVSN_STAMP(0);
    floatx4 wreg[3];
    load_weights_to_regs<W0_FLOATS>(a.packed, wreg, tid);
    write_cost_slice(d - 1);

    // A1: image plane d and its mask (global gathers; the 6 KB source image stays in L1/L2)
    float img[IMG_IT][3];
    float mk[IMG_IT];
    {
      float Hl[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Hl[i] = Hn[d * 9 + i];
#pragma unroll
      for (int it = 0; it < IMG_IT; ++it) {
        const int p = tid + it * CH_THREADS;
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

    // A2: previous plane's features moved by the incremental homography (gather from the activation planes) -> mv
    {
      float Hl[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Hl[i] = Hin[d * 9 + i];
#pragma unroll
      for (int j = 0; j < TP; ++j) {
        WarpCoord c = warp_coord(Hl, (float)(pb[j] % cols), (float)(pb[j] / cols), (float)rows, (float)cols);
        Bilinear b = bilinear_taps(c.ix, c.iy, rows, cols);
        const float keep = c.outside ? 0.0f : 1.0f;
        const int o00 = b.y0 * RS + b.x0, o01 = b.y0 * RS + b.x1, o10 = b.y1 * RS + b.x0, o11 = b.y1 * RS + b.x1;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float *fc = act + (3 + t * 16 + cbase + r) * CS;
            const float v = keep * (fc[o00] * b.w00 + fc[o01] * b.w01 + fc[o10] * b.w10 + fc[o11] * b.w11);
            if (valid[j]) mv[(size_t)(t * 16 + cbase + r) * P + pb[j]] = v;
          }
        __builtin_amdgcn_sched_barrier(0);   // one tile's 32 taps in flight at a time
      }
    }
    MVSN_STAMP(1);
    __syncthreads();  // B1: every gather of plane d-1 is done
    MVSN_STAMP(2);

    // A3: lay out the refiner input [image(3) | moved features(32)]
#pragma unroll
    for (int j = 0; j < TP; ++j)
      if (valid[j]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            act[(3 + t * 16 + cbase + r) * CS + qb[j]] = mv[(size_t)(t * 16 + cbase + r) * P + pb[j]];
      }
#pragma unroll
    for (int it = 0; it < IMG_IT; ++it) {
      const int p = tid + it * CH_THREADS;
      if (p < P) {
        const int q = (p / cols) * RS + (p % cols);
        act[0 * CS + q] = img[it][0];
        act[1 * CS + q] = img[it][1];
        act[2 * CS + q] = img[it][2];
        maskb[p] = mk[it];
        maskg[(size_t)d * P + p] = mk[it] != 0.0f ? 1 : 0;
      }
    }
    store_weights_to_lds<W0_FLOATS>(wbuf, wreg, tid);
    __syncthreads();  // B2
    MVSN_STAMP(3);

    floatx4 acc[TP][2];
    load_weights_to_regs<W1_FLOATS>(a.packed + W0_FLOATS, wreg, tid);  // in flight behind the MFMAs
    if constexpr (LDS_ACT) conv3x3_mfma<TP, 9>(act, wbuf, CS, RS, qb, lane, acc);
    else conv3x3_mfma_slab<TP, 9>(act, wbuf, slab, CS, RS, qb, tid, acc);
    MVSN_STAMP(4);
    __syncthreads();  // B3: act and wbuf free
    MVSN_STAMP(5);

    groupnorm_lrelu<TP>(acc, valid, bias0, gn0w, gn0b, red, red + 64, inv_count, lane, wave);
#pragma unroll
    for (int j = 0; j < TP; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (valid[j]) {
#pragma unroll
          for (int r = 0; r < 4; ++r) act[(t * 16 + cbase + r) * CS + qb[j]] = acc[j][t][r];
        }
      }
    store_weights_to_lds<W1_FLOATS>(wbuf, wreg, tid);
    __syncthreads();  // B6
    MVSN_STAMP(6);

    load_weights_to_regs<W1_FLOATS>(a.packed + W0_FLOATS + W1_FLOATS, wreg, tid);
    if constexpr (LDS_ACT) conv3x3_mfma<TP, 8>(act, wbuf, CS, RS, qb, lane, acc);
    else conv3x3_mfma_slab<TP, 8>(act, wbuf, slab, CS, RS, qb, tid, acc);
    MVSN_STAMP(7);
    __syncthreads();  // B7
    MVSN_STAMP(8);

    groupnorm_lrelu<TP>(acc, valid, bias1, gn1w, gn1b, red + 128, red + 192, inv_count, lane, wave);
#pragma unroll
    for (int j = 0; j < TP; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (valid[j]) {
#pragma unroll
          for (int r = 0; r < 4; ++r) act[(t * 16 + cbase + r) * CS + qb[j]] += acc[j][t][r];  // x2 = x1 + ...
        }
      }
    store_weights_to_lds<W1_FLOATS>(wbuf, wreg, tid);
    __syncthreads();  // B10
    MVSN_STAMP(9);

    if constexpr (LDS_ACT) conv3x3_mfma<TP, 8>(act, wbuf, CS, RS, qb, lane, acc);
    else conv3x3_mfma_slab<TP, 8>(act, wbuf, slab, CS, RS, qb, tid, acc);
    MVSN_STAMP(10);
    __syncthreads();  // B11
    MVSN_STAMP(11);

    // epilogue: the new features become the next step's gather source; their cost slice is
    // written (coalesced) at the top of the next step / after the loop
#pragma unroll
    for (int j = 0; j < TP; ++j) {
      if (!valid[j]) continue;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = t * 16 + cbase + r;
          act[(3 + c) * CS + qb[j]] = mv[(size_t)c * P + pb[j]] + (acc[j][t][r] + bias2[c]);
        }
    }
    MVSN_STAMP(12);
    __syncthreads();  // B12
    MVSN_STAMP(13);
  }
#undef MVSN_STAMP
  write_cost_slice(D - 1);
}

static size_t chain_lds_bytes(int P, int act_floats, bool lds_act, int slab_floats = 0) {
  const int Ppad = (P + 3) & ~3;
  size_t f = (size_t)W0_FLOATS + SP_FLOATS + RED_FLOATS + Ppad;
  if (lds_act) f += act_floats;
  else f += 2 * (size_t)slab_floats;
  return f * sizeof(float);
}

static int chain_cs(int rows, int cols) {
  const int RS = cols + 1, G = RS + 1;
  int cs = rows * RS + G;
  while ((cs & 31) != 16) ++cs;
  return cs;
}

}  // namespace mvsn

extern "C" size_t mvsn_feature_refiner_packed_floats(void) { return mvsn::PACKED_FLOATS; }

extern "C" int mvsn_pack_feature_refiner(const float *conv0_w, const float *conv0_b, const float *bn0_w,
                                         const float *bn0_b, const float *res0_w, const float *res0_b,
                                         const float *res0_bn_w, const float *res0_bn_b, const float *final_w,
                                         const float *final_b, float *packed, mvsn_stream_t stream) {
  MVSN_REQUIRE(conv0_w && conv0_b && bn0_w && bn0_b && res0_w && res0_b && res0_bn_w && res0_bn_b && final_w &&
                   final_b && packed,
               MVSN_E_BADARG, "mvsn_pack_feature_refiner: null pointer");
  hipLaunchKernelGGL(mvsn::pack_refiner_kernel, dim3((mvsn::CH_STEPS_OFFSET + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, conv0_w, conv0_b, bn0_w, bn0_b, res0_w, res0_b, res0_bn_w, res0_bn_b,
                     final_w, final_b, packed);
  if (int rc = mvsn::check_launch("mvsn_pack_feature_refiner")) return rc;
  // the same three layers in the convolution kernels' Winograd layout (stepwise form)
  float *u = packed + mvsn::CH_STEPS_OFFSET;
  if (int rc = mvsn::wino_pack_2d(conv0_w, 35, u, (hipStream_t)stream)) return rc;
  if (int rc = mvsn::wino_pack_2d(res0_w, 32, u + mvsn::CS_U0_FLOATS, (hipStream_t)stream)) return rc;
  return mvsn::wino_pack_2d(final_w, 32, u + mvsn::CS_U0_FLOATS + mvsn::CS_U1_FLOATS, (hipStream_t)stream);
}

extern "C" int mvsn_incremental_cost_volume_form(int rows, int cols) {
  return mvsn::chain_wino_supported(rows, cols) ? MVSN_CHAIN_WINOGRAD : MVSN_CHAIN_DIRECT;
}

extern "C" size_t mvsn_incremental_cost_volume_workspace_bytes(int n_chains, int rows, int cols) {
  if (n_chains <= 0 || rows <= 0 || cols <= 0) return 0;
  const int CS = mvsn::chain_cs(rows, cols);
  const int act_floats = (cols + 2) + 36 * CS;
  // per chain: the step's moved features (32 x P, always) + the activation planes where they do not fit LDS
  const size_t mv_floats = (size_t)32 * rows * cols;
  if (mvsn::chain_lds_bytes(rows * cols, act_floats, true) <= 160 * 1024) return (size_t)n_chains * mv_floats * sizeof(float);
  return (size_t)n_chains * (act_floats + mv_floats) * sizeof(float);
}

// What MVSN_CHAIN_AUTO resolves to for this many chains on this coarse grid -- in the entry points that put a gated repair
// launch behind a banded call (mvsn_incremental_cost_volume_guarded / _bf16; what the module calls).
static int chain_auto_form(int n_chains, int rows, int cols) {
  if (mvsn::chain_band_supported(rows, cols)) {
    // 30x40 / 32x64 (no plane-resident plan): the banded form for any number of chains -- thin bands while they fit one
    // pass, the slab plan (3 / 4 fat bands per chain resident in LDS, mvsn_chain_slab.hip) beyond; the stepwise form, which
    // sends the plane through HBM every step, is no longer AUTO's choice there
    if (!mvsn::chain_wino_supported(rows, cols)) return MVSN_CHAIN_BANDED;
    // 16x32: several workgroups per chain only while all chains fit the chip in ONE pass (beyond 64 chains the
    // plane-resident kernel is faster)
    const int cap = mvsn::chain_band_chains_per_pass(rows, cols);
    if (cap > 0 && n_chains <= cap) return MVSN_CHAIN_BANDED;
  }
  if (mvsn::chain_wino_supported(rows, cols)) return MVSN_CHAIN_WINOGRAD;
  // no plane-resident plan: one plane per round of full-chip launches, whatever the number of chains (re-measured in
  // round 3 with tools/chain_bench.py: 30x40, 256 / 512 chains 25.4 / 48.0 ms against the direct form's 41.5 / 86.8;
  // the direct form -- one workgroup per chain, planes in a global workspace -- remains for cols % 4 != 0)
  if (mvsn::chain_steps_supported(rows, cols)) return MVSN_CHAIN_STEPWISE;
  return MVSN_CHAIN_DIRECT;
}

// ... and in the PLAIN entry point (mvsn_incremental_cost_volume), which has no repair launch behind it: the multi-pass
// slab launches need the device to themselves for milliseconds at a stretch and a time-out there would leave a
// NaN-poisoned cost slice, so beyond ONE thin-band pass the plain AUTO stays on the co-residency-free forms (the
// round-4 policy).  A caller that wants the slab plan asks for MVSN_CHAIN_BANDED or uses the guarded entry.
static int chain_auto_form_unguarded(int n_chains, int rows, int cols) {
  const int form = chain_auto_form(n_chains, rows, cols);
  if (form == MVSN_CHAIN_BANDED && !mvsn::chain_wino_supported(rows, cols)) {
    const int cap = mvsn::chain_band_chains_per_pass(rows, cols);
    if (cap <= 0 || n_chains > cap) return mvsn::chain_steps_supported(rows, cols) ? MVSN_CHAIN_STEPWISE : MVSN_CHAIN_DIRECT;
  }
  return form;
}

extern "C" int mvsn_incremental_cost_volume_form_for(int n_chains, int rows, int cols) {
  if (n_chains <= 0 || rows <= 0 || cols <= 0) return MVSN_CHAIN_DIRECT;
  return chain_auto_form(n_chains, rows, cols);
}

extern "C" size_t mvsn_incremental_cost_volume_workspace_bytes_for(int n_chains, int num_idepth_samples, int rows,
                                                                   int cols, int form) {
  if (n_chains <= 0 || rows <= 0 || cols <= 0 || num_idepth_samples <= 0) return 0;
  if (form == MVSN_CHAIN_AUTO) {      // whichever entry point resolves it (they differ beyond one thin-band pass)
    const int fg = chain_auto_form(n_chains, rows, cols), fu = chain_auto_form_unguarded(n_chains, rows, cols);
    const size_t g = mvsn_incremental_cost_volume_workspace_bytes_for(n_chains, num_idepth_samples, rows, cols, fg);
    const size_t u = fu == fg ? g : mvsn_incremental_cost_volume_workspace_bytes_for(n_chains, num_idepth_samples, rows, cols, fu);
    return g > u ? g : u;
  }
  if (form == MVSN_CHAIN_STEPWISE) return mvsn::chain_steps_workspace_bytes(n_chains, num_idepth_samples, rows, cols);
  if (form == MVSN_CHAIN_BANDED) return mvsn::chain_band_workspace_bytes(n_chains, rows, cols);
  if (form == MVSN_CHAIN_WINOGRAD) return 0;
  return mvsn_incremental_cost_volume_workspace_bytes(n_chains, rows, cols);
}

extern "C" size_t mvsn_incremental_cost_volume_status_offset(int n_chains, int rows, int cols) {
  return n_chains > 0 ? mvsn::chain_band_status_offset(n_chains, rows, cols) : 0;
}

extern "C" int mvsn_incremental_cost_volume_banded_groups(int n_chains, int rows, int cols) {
  return n_chains > 0 ? mvsn::chain_band_groups(n_chains, rows, cols) : 0;
}

extern "C" int mvsn_debug_set_band_flags(int flags) {
  mvsn::chain_band_debug_flags(flags);
  return 0;
}

#ifdef MVSN_CHAIN_STAMPS
static unsigned long long *g_chain_stamps = nullptr;
extern "C" int mvsn_debug_set_chain_stamps(void *buf) {
  g_chain_stamps = (unsigned long long *)buf;
  return 0;
}
#endif

// The chain in `form` (resolved), optionally as a gated repair launch (gate != null: one of the single-launch forms).
static int chain_run(const float *src_image_lvl4, const float *H_lvl4, const float *H_inc, const float *plane0_features,
                     const float *left_features, const float *refiner_packed, int n_chains, int batch,
                     int num_idepth_samples, int rows, int cols, float *cost_volume, uint8_t *mask_volume,
                     float *feature_volume, void *workspace, size_t workspace_bytes, int form, const unsigned *gate,
                     unsigned *sticky, mvsn_stream_t stream, int cost_bf16 = 0) {
  using namespace mvsn;
  MVSN_REQUIRE(src_image_lvl4 && H_lvl4 && H_inc && plane0_features && left_features && refiner_packed &&
                   cost_volume && mask_volume,
               MVSN_E_BADARG, "mvsn_incremental_cost_volume: null pointer");
  MVSN_REQUIRE(n_chains > 0 && batch > 0 && num_idepth_samples >= 1 && rows > 0 && cols > 0, MVSN_E_BADARG,
               "mvsn_incremental_cost_volume: bad sizes");
  MVSN_REQUIRE(form >= 0 && form <= 4, MVSN_E_BADARG, "mvsn_incremental_cost_volume: form must be 0 .. 4");
  if (form == MVSN_CHAIN_AUTO) form = chain_auto_form_unguarded(n_chains, rows, cols);   // (the guarded entries resolve AUTO themselves)
  MVSN_REQUIRE(form != MVSN_CHAIN_WINOGRAD || chain_wino_supported(rows, cols), MVSN_E_TOOLARGE,
               "mvsn_incremental_cost_volume: no Winograd plan for a %dx%d coarse grid", rows, cols);
  const bool wino = form == MVSN_CHAIN_WINOGRAD || (form == MVSN_CHAIN_AUTO && chain_wino_supported(rows, cols));
  const int P = rows * cols;
  const int tiles = (P + 15) / 16;
  const int TP = (tiles + CH_WAVES - 1) / CH_WAVES;
  MVSN_REQUIRE(form != MVSN_CHAIN_BANDED || chain_band_supported(rows, cols), MVSN_E_TOOLARGE,
               "mvsn_incremental_cost_volume: the banded form has no plan for a %dx%d coarse grid (16x32, 30x40, 32x64)", rows, cols);
  MVSN_REQUIRE(wino || form == MVSN_CHAIN_STEPWISE || form == MVSN_CHAIN_BANDED || TP <= 8, MVSN_E_TOOLARGE,
               "mvsn_incremental_cost_volume: %dx%d coarse grid (%d px) exceeds the 2048 px plan", rows, cols, P);
  MVSN_REQUIRE(gate == nullptr || wino || form == MVSN_CHAIN_DIRECT, MVSN_E_BADARG,
               "mvsn_incremental_cost_volume: a repair launch runs the Winograd or the direct form");
  MVSN_REQUIRE(!cost_bf16 || form != MVSN_CHAIN_STEPWISE, MVSN_E_BADARG,
               "mvsn_incremental_cost_volume_bf16: the stepwise form has no bf16 cost-volume variant");
  ChainArgs a;
  a.cost_bf16 = cost_bf16 ? 1 : 0;
  a.chain0 = 0;
  a.ws_chains = 0;
  a.gate = gate;
  a.sticky = sticky;
  a.src = src_image_lvl4;
  a.H = H_lvl4;
  a.Hinc = H_inc;
  a.f0 = plane0_features;
  a.fl = left_features;
  a.packed = refiner_packed;
  a.cost = cost_volume;
  a.mask = mask_volume;
  a.fvol = feature_volume;
  a.B = batch;
  a.D = num_idepth_samples;
  a.rows = rows;
  a.cols = cols;
  a.CS = chain_cs(rows, cols);
  a.dbg = nullptr;
#ifdef MVSN_CHAIN_STAMPS   // tuning builds only (tools/chain_phases.py): device pointer to 64 x u64 cycle stamps
  a.dbg = g_chain_stamps;
#endif
  if (form == MVSN_CHAIN_STEPWISE) {
    a.workspace = nullptr;
    return chain_steps_launch(a, n_chains, workspace, workspace_bytes, (hipStream_t)stream);
  }
  if (form == MVSN_CHAIN_BANDED) return chain_band_launch(a, n_chains, workspace, workspace_bytes, 0, (hipStream_t)stream);
  if (wino) {
    a.workspace = nullptr;
    return chain_wino_launch(a, n_chains, (hipStream_t)stream);
  }
  const int act_floats = (cols + 2) + 36 * a.CS;
  const bool lds_act = chain_lds_bytes(P, act_floats, true) <= 160 * 1024;
  const size_t need = mvsn_incremental_cost_volume_workspace_bytes(n_chains, rows, cols);
  MVSN_REQUIRE(workspace && workspace_bytes >= need, MVSN_E_WORKSPACE,
               "mvsn_incremental_cost_volume: workspace of %zu bytes required", need);
  a.workspace = (float *)workspace;
  const int slab_floats = (cols + 2) + 4 * a.CS;
  MVSN_REQUIRE(lds_act || slab_floats <= 9 * CH_THREADS, MVSN_E_TOOLARGE,
               "mvsn_incremental_cost_volume: coarse grid too large for the slab staging plan");
  const size_t lds = chain_lds_bytes(P, act_floats, lds_act, slab_floats);

#define MVSN_CHAIN_LAUNCH(TPV, LDSV)                                                                           \
  do {                                                                                                         \
    if (a.cost_bf16) {   /* (bf16 feature tier: its own instantiation and LDS opt-in) */                        \
      auto kern16 = chain_kernel<TPV, LDSV, true>;                                                             \
      static LdsOptIn opt16;                                                                                   \
      if (int rc = ensure_lds(opt16, (const void *)kern16, lds, "mvsn_incremental_cost_volume")) return rc;    \
      hipLaunchKernelGGL(kern16, dim3(n_chains), dim3(CH_THREADS), lds, (hipStream_t)stream, a, CHAIN_VISIBLE_G(a)); \
      break;                                                                                                   \
    }                                                                                                          \
    auto kern = chain_kernel<TPV, LDSV, false>;                                                                \
    static LdsOptIn opt;                                                                                       \
    if (int rc = ensure_lds(opt, (const void *)kern, lds, "mvsn_incremental_cost_volume")) return rc;          \
    hipLaunchKernelGGL(kern, dim3(n_chains), dim3(CH_THREADS), lds, (hipStream_t)stream, a, CHAIN_VISIBLE_G(a));    \
  } while (0)

  MVSN_REQUIRE(!lds_act || TP <= 3, MVSN_E_TOOLARGE, "mvsn_incremental_cost_volume: internal plan error");
  if (lds_act) {
    switch (TP) {
      case 1: MVSN_CHAIN_LAUNCH(1, true); break;
      case 2: MVSN_CHAIN_LAUNCH(2, true); break;
      default: MVSN_CHAIN_LAUNCH(3, true); break;
    }
  } else {
    switch (TP) {
      case 1: MVSN_CHAIN_LAUNCH(1, false); break;
      case 2: MVSN_CHAIN_LAUNCH(2, false); break;
      case 3: MVSN_CHAIN_LAUNCH(3, false); break;
      case 4: MVSN_CHAIN_LAUNCH(4, false); break;
      case 5: MVSN_CHAIN_LAUNCH(5, false); break;
      case 6: MVSN_CHAIN_LAUNCH(6, false); break;
      case 7: MVSN_CHAIN_LAUNCH(7, false); break;
      default: MVSN_CHAIN_LAUNCH(8, false); break;
    }
  }
#undef MVSN_CHAIN_LAUNCH
  return check_launch("mvsn_incremental_cost_volume");
}

extern "C" int mvsn_incremental_cost_volume(const float *src_image_lvl4, const float *H_lvl4, const float *H_inc,
                                            const float *plane0_features, const float *left_features,
                                            const float *refiner_packed, int n_chains, int batch,
                                            int num_idepth_samples, int rows, int cols, float *cost_volume,
                                            uint8_t *mask_volume, float *feature_volume, void *workspace,
                                            size_t workspace_bytes, int form, mvsn_stream_t stream) {
  return chain_run(src_image_lvl4, H_lvl4, H_inc, plane0_features, left_features, refiner_packed, n_chains, batch,
                   num_idepth_samples, rows, cols, cost_volume, mask_volume, feature_volume, workspace, workspace_bytes,
                   form, nullptr, nullptr, stream);
}

// The single-launch form a banded call is repaired with: the plane-resident Winograd kernel where the grid has that plan
// (16x32), the direct kernel (planes in a global workspace) elsewhere.
static int chain_repair_form(int rows, int cols) {
  return mvsn::chain_wino_supported(rows, cols) ? MVSN_CHAIN_WINOGRAD : MVSN_CHAIN_DIRECT;
}

extern "C" size_t mvsn_incremental_cost_volume_repair_workspace_bytes(int n_chains, int rows, int cols) {
  if (n_chains <= 0 || rows <= 0 || cols <= 0) return 0;
  if (chain_repair_form(rows, cols) == MVSN_CHAIN_WINOGRAD) return 0;
  return mvsn_incremental_cost_volume_workspace_bytes(n_chains, rows, cols);
}

extern "C" int mvsn_incremental_cost_volume_guarded(const float *src_image_lvl4, const float *H_lvl4, const float *H_inc,
                                                    const float *plane0_features, const float *left_features,
                                                    const float *refiner_packed, int n_chains, int batch,
                                                    int num_idepth_samples, int rows, int cols, float *cost_volume,
                                                    uint8_t *mask_volume, float *feature_volume, void *workspace,
                                                    size_t workspace_bytes, int form, void *repair_workspace,
                                                    size_t repair_workspace_bytes, unsigned *sticky_status,
                                                    mvsn_stream_t stream) {
  using namespace mvsn;
  if (form == MVSN_CHAIN_AUTO && n_chains > 0 && rows > 0 && cols > 0) form = chain_auto_form(n_chains, rows, cols);
  // everything the repair launch needs is validated BEFORE the banded launch is enqueued: an error return must not
  // leave an unrepaired banded chain (NaN on a time-out) behind on the stream
  size_t need = 0;
  if (form == MVSN_CHAIN_BANDED && n_chains > 0 && rows > 0 && cols > 0) {
    need = mvsn_incremental_cost_volume_repair_workspace_bytes(n_chains, rows, cols);
    MVSN_REQUIRE(need == 0 || (repair_workspace && repair_workspace_bytes >= need), MVSN_E_WORKSPACE,
                 "mvsn_incremental_cost_volume_guarded: repair workspace of %zu bytes required", need);
  }
  if (int rc = chain_run(src_image_lvl4, H_lvl4, H_inc, plane0_features, left_features, refiner_packed, n_chains, batch,
                         num_idepth_samples, rows, cols, cost_volume, mask_volume, feature_volume, workspace,
                         workspace_bytes, form, nullptr, nullptr, stream))
    return rc;
  if (form != MVSN_CHAIN_BANDED) return 0;   // the other forms have no inter-workgroup hand-offs to time out
  const unsigned *gate = reinterpret_cast<const unsigned *>(static_cast<const char *>(workspace) +
                                                            chain_band_status_offset(n_chains, rows, cols));
  return chain_run(src_image_lvl4, H_lvl4, H_inc, plane0_features, left_features, refiner_packed, n_chains, batch,
                   num_idepth_samples, rows, cols, cost_volume, mask_volume, feature_volume, repair_workspace,
                   repair_workspace_bytes, chain_repair_form(rows, cols), gate, sticky_status, stream);
}

// The guarded call with the cost volume stored as bf16 (the bf16 feature tier, BASELINE config 5; include/mvsn_hip.h).
extern "C" int mvsn_incremental_cost_volume_bf16(const float *src_image_lvl4, const float *H_lvl4, const float *H_inc,
                                                    const float *plane0_features, const float *left_features,
                                                    const float *refiner_packed, int n_chains, int batch,
                                                    int num_idepth_samples, int rows, int cols, void *cost_volume_bf16,
                                                    uint8_t *mask_volume, float *feature_volume, void *workspace,
                                                    size_t workspace_bytes, int form, void *repair_workspace,
                                                    size_t repair_workspace_bytes, unsigned *sticky_status,
                                                    mvsn_stream_t stream) {
  using namespace mvsn;
  if (form == MVSN_CHAIN_AUTO && n_chains > 0 && rows > 0 && cols > 0) form = chain_auto_form(n_chains, rows, cols);
  // everything the repair launch needs is validated BEFORE the banded launch is enqueued: an error return must not
  // leave an unrepaired banded chain (NaN on a time-out) behind on the stream
  size_t need = 0;
  if (form == MVSN_CHAIN_BANDED && n_chains > 0 && rows > 0 && cols > 0) {
    need = mvsn_incremental_cost_volume_repair_workspace_bytes(n_chains, rows, cols);
    MVSN_REQUIRE(need == 0 || (repair_workspace && repair_workspace_bytes >= need), MVSN_E_WORKSPACE,
                 "mvsn_incremental_cost_volume_bf16: repair workspace of %zu bytes required", need);
  }
  if (int rc = chain_run(src_image_lvl4, H_lvl4, H_inc, plane0_features, left_features, refiner_packed, n_chains, batch,
                         num_idepth_samples, rows, cols, (float *)cost_volume_bf16, mask_volume, feature_volume, workspace,
                         workspace_bytes, form, nullptr, nullptr, stream, 1))
    return rc;
  if (form != MVSN_CHAIN_BANDED) return 0;   // the other forms have no inter-workgroup hand-offs to time out
  const unsigned *gate = reinterpret_cast<const unsigned *>(static_cast<const char *>(workspace) +
                                                            chain_band_status_offset(n_chains, rows, cols));
  return chain_run(src_image_lvl4, H_lvl4, H_inc, plane0_features, left_features, refiner_packed, n_chains, batch,
                   num_idepth_samples, rows, cols, (float *)cost_volume_bf16, mask_volume, feature_volume, repair_workspace,
                   repair_workspace_bytes, chain_repair_form(rows, cols), gate, sticky_status, stream, 1);
}
