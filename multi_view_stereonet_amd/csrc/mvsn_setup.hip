// Plane-sweep set-up: baseline normalisation, idepth samples, homography families.
// One workgroup per chain; the geometry is evaluated in double and rounded once, the idepth
// samples and the incremental homographies start from fp32-rounded values exactly where the
// reference's fp32 pipeline rounds them (see include/mvsn_hip.h for the call sites replaced).
#include "mvsn_common.h"

namespace mvsn {

__device__ inline void inv3(const double *m, double *o) {
  double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  double det = a * A + b * B + c * C;
  double r = 1.0 / det;
  o[0] = A * r;
  o[1] = -(b * i - c * h) * r;
  o[2] = (b * f - c * e) * r;
  o[3] = B * r;
  o[4] = (a * i - c * g) * r;
  o[5] = -(a * f - c * d) * r;
  o[6] = C * r;
  o[7] = -(a * h - b * g) * r;
  o[8] = (a * e - b * d) * r;
}

__device__ inline void mul3(const double *a, const double *b, double *o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

// H = K (R + t*idepth e3^T) K^-1
__device__ inline void plane_homography(const double *K3, const double *K3inv, const double *R, const double *t,
                                        double idepth, double *H) {
  double core[9], tmp[9];
  for (int i = 0; i < 9; ++i) core[i] = R[i];
  core[2] += t[0] * idepth;
  core[5] += t[1] * idepth;
  core[8] += t[2] * idepth;
  mul3(core, K3inv, tmp);
  mul3(K3, tmp, H);
}

constexpr int SETUP_THREADS = 256;

// Where a chain's pose and intrinsics live: per chain (T (N,4,4), K (N,4,4)) or, `per_source`, as the forward holds
// them -- one (B,4,4) pose tensor per source view and the B reference images' intrinsics shared by their S chains
// (chain n = s * B + b): no cat / repeat in front of the launch.
struct SetupSrc {
  const float *T[8];
  int B, per_source;
};

__global__ __launch_bounds__(SETUP_THREADS) void plane_sweep_setup_kernel(
    SetupSrc src, const float *__restrict__ K0_in, const float *__restrict__ K4_in, int rows4,
    int cols4, int D, float *__restrict__ samples_out, float *__restrict__ H4_out, float *__restrict__ Hinc_out,
    float *__restrict__ H0_out, float *__restrict__ baseline_out, MVSN_VIS10) {   // (MVSN_VIS10: mvsn_common.h)
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int kb = src.per_source ? n % src.B : n;
  const float *T = src.per_source ? src.T[n / src.B] + (size_t)kb * 16 : src.T[0] + (size_t)n * 16;
  const float *K0 = K0_in + (size_t)kb * 16;
  const float *K4 = K4_in + (size_t)kb * 16;

  __shared__ double s_sum[SETUP_THREADS];
  __shared__ int s_cnt[SETUP_THREADS];
  __shared__ float s_top;

  // --- baseline renormalisation in fp32, as multi_view_stereonet.py:566-571 -------------------
  float tx = T[3], ty = T[7], tz = T[11];
  float base = sqrtf(tx * tx + ty * ty + tz * tz);
  float tn[3] = {tx / base, ty / base, tz / base};

  // --- T_left_in_right = inverse([R t; 0 1]) ---------------------------------------------------
  double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  double Rl[9];
  inv3(R, Rl);
  double tl[3];
  for (int i = 0; i < 3; ++i) tl[i] = -(Rl[i * 3] * tn[0] + Rl[i * 3 + 1] * tn[1] + Rl[i * 3 + 2] * tn[2]);

  double K4d[9] = {K4[0], K4[1], K4[2], K4[4], K4[5], K4[6], K4[8], K4[9], K4[10]};
  double K4inv[9];
  inv3(K4d, K4inv);

  // --- maximum idepth: mean over pixels of the idepth that yields D-1 px of disparity ----------
  // (stereo/image_predictor.py:148-207)
  double M[9], tmp[9];
  mul3(Rl, K4inv, tmp);
  mul3(K4d, tmp, M);
  double Kt[3];
  for (int i = 0; i < 3; ++i) Kt[i] = K4d[i * 3] * tl[0] + K4d[i * 3 + 1] * tl[1] + K4d[i * 3 + 2] * tl[2] + (double)K4[i * 4 + 3];
  const double disp = (double)(D - 1);
  double acc = 0.0;
  int cnt = 0;
  const int P = rows4 * cols4;
  for (int p = tid; p < P; p += SETUP_THREADS) {
    double x = (double)(p % cols4), y = (double)(p / cols4);
    double i0 = M[0] * x + M[1] * y + M[2], i1 = M[3] * x + M[4] * y + M[5], i2 = M[6] * x + M[7] * y + M[8];
    double infx = i0 / i2, infy = i1 / i2;
    double f0 = 1e2 * i0 + Kt[0], f1 = 1e2 * i1 + Kt[1], f2 = 1e2 * i2 + Kt[2];
    double dx = f0 / f2 - infx, dy = f1 / f2 - infy;
    double nrm = sqrt(dx * dx + dy * dy);
    double ex = dx / (nrm + 1e-6), ey = dy / (nrm + 1e-6);
    double A0 = Kt[0] - Kt[2] * (infx + disp * ex);
    double A1 = Kt[1] - Kt[2] * (infy + disp * ey);
    double b0 = i2 * disp * ex, b1 = i2 * disp * ey;
    double idp = (A0 * b0 + A1 * b1) / (A0 * A0 + A1 * A1);
    float idf = (nrm < 1e-6) ? 0.0f : (float)idp;
    if (idf > 0.0f) {
      acc += (double)idf;
      cnt += 1;
    }
  }
  s_sum[tid] = acc;
  s_cnt[tid] = cnt;
  __syncthreads();
  for (int s = SETUP_THREADS / 2; s > 0; s >>= 1) {
    if (tid < s) {
      s_sum[tid] += s_sum[tid + s];
      s_cnt[tid] += s_cnt[tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) {
    float top = (float)s_sum[0] / (float)s_cnt[0];  // NaN when no pixel qualifies, as the reference
    if (top > 2.0f) top = 2.0f;
    if (1.0f / top < tn[2]) top = 1.0f / tn[2];      // keep samples in front of the source camera
    s_top = top;
    baseline_out[n] = base;
    // full-resolution homography of plane 0 (multi_view_stereonet.py:254-255)
    double K0d[9] = {K0[0], K0[1], K0[2], K0[4], K0[5], K0[6], K0[8], K0[9], K0[10]};
    double K0inv[9], H0[9];
    inv3(K0d, K0inv);
    plane_homography(K0d, K0inv, Rl, tl, 0.0, H0);
    for (int i = 0; i < 9; ++i) H0_out[(size_t)n * 9 + i] = (float)H0[i];
  }
  __syncthreads();
  const float delta = s_top / (float)(D - 1);

  for (int d = tid; d < D; d += SETUP_THREADS) {
    float sd = (float)d * delta;
    samples_out[(size_t)n * D + d] = sd;
    double H[9];
    plane_homography(K4d, K4inv, Rl, tl, (double)sd, H);
    float Hf[9];
    for (int i = 0; i < 9; ++i) {
      Hf[i] = (float)H[i];
      H4_out[((size_t)n * D + d) * 9 + i] = Hf[i];
    }
    float *inc = Hinc_out + ((size_t)n * D + d) * 9;
    if (d == 0) {
      for (int i = 0; i < 9; ++i) inc[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    } else {
      // previous plane, rounded to fp32 like the reference's stored H (multi_view_stereonet.py:281-282)
      float sp = (float)(d - 1) * delta;
      double Hp[9], Hpf[9], Hpinv[9], Hcf[9], Hi[9];
      plane_homography(K4d, K4inv, Rl, tl, (double)sp, Hp);
      for (int i = 0; i < 9; ++i) {
        Hpf[i] = (double)(float)Hp[i];
        Hcf[i] = (double)Hf[i];
      }
      inv3(Hpf, Hpinv);
      mul3(Hpinv, Hcf, Hi);
      for (int i = 0; i < 9; ++i) inc[i] = (float)Hi[i];
    }
  }
}

}  // namespace mvsn

extern "C" int mvsn_plane_sweep_setup(const float *T_right_in_left, const float *K_lvl0, const float *K_lvl4,
                                      int n_chains, int rows4, int cols4, int num_idepth_samples,
                                      float *idepth_samples, float *H_lvl4, float *H_inc, float *H_lvl0_plane0,
                                      float *baseline, mvsn_stream_t stream) {
  MVSN_REQUIRE(T_right_in_left && K_lvl0 && K_lvl4 && idepth_samples && H_lvl4 && H_inc && H_lvl0_plane0 && baseline,
               MVSN_E_BADARG, "mvsn_plane_sweep_setup: null pointer");
  MVSN_REQUIRE(n_chains > 0 && rows4 > 0 && cols4 > 0 && num_idepth_samples >= 2, MVSN_E_BADARG,
               "mvsn_plane_sweep_setup: bad sizes (chains %d, %dx%d, D %d)", n_chains, rows4, cols4, num_idepth_samples);
  mvsn::SetupSrc src = {};
  src.T[0] = T_right_in_left;
  hipLaunchKernelGGL(mvsn::plane_sweep_setup_kernel, dim3(n_chains), dim3(mvsn::SETUP_THREADS), 0,
                     (hipStream_t)stream, src, K_lvl0, K_lvl4, rows4, cols4, num_idepth_samples,
                     idepth_samples, H_lvl4, H_inc, H_lvl0_plane0, baseline, (const void *)src.T[0], (const void *)src.T[1],
                     (const void *)src.T[2], (const void *)src.T[3], (const void *)src.T[4], (const void *)src.T[5],
                     (const void *)src.T[6], (const void *)src.T[7], (const void *)nullptr, (const void *)nullptr);
  return mvsn::check_launch("mvsn_plane_sweep_setup");
}

extern "C" int mvsn_plane_sweep_setup_sources(const float *const *T_right_in_lefts, int n_sources, const float *K_lvl0,
                                              const float *K_lvl4, int batch, int rows4, int cols4,
                                              int num_idepth_samples, float *idepth_samples, float *H_lvl4, float *H_inc,
                                              float *H_lvl0_plane0, float *baseline, mvsn_stream_t stream) {
  MVSN_REQUIRE(T_right_in_lefts && K_lvl0 && K_lvl4 && idepth_samples && H_lvl4 && H_inc && H_lvl0_plane0 && baseline,
               MVSN_E_BADARG, "mvsn_plane_sweep_setup_sources: null pointer");
  MVSN_REQUIRE(n_sources >= 1 && n_sources <= 8 && batch > 0 && rows4 > 0 && cols4 > 0 && num_idepth_samples >= 2,
               MVSN_E_BADARG, "mvsn_plane_sweep_setup_sources: bad sizes (sources %d of at most 8, batch %d)", n_sources, batch);
  mvsn::SetupSrc src = {};
  for (int s = 0; s < n_sources; ++s) {
    MVSN_REQUIRE(T_right_in_lefts[s], MVSN_E_BADARG, "mvsn_plane_sweep_setup_sources: null pose pointer");
    src.T[s] = T_right_in_lefts[s];
  }
  src.B = batch, src.per_source = 1;
  hipLaunchKernelGGL(mvsn::plane_sweep_setup_kernel, dim3(n_sources * batch), dim3(mvsn::SETUP_THREADS), 0,
                     (hipStream_t)stream, src, K_lvl0, K_lvl4, rows4, cols4, num_idepth_samples, idepth_samples, H_lvl4,
                     H_inc, H_lvl0_plane0, baseline, (const void *)src.T[0], (const void *)src.T[1],
                     (const void *)src.T[2], (const void *)src.T[3], (const void *)src.T[4], (const void *)src.T[5],
                     (const void *)src.T[6], (const void *)src.T[7], (const void *)nullptr, (const void *)nullptr);
  return mvsn::check_launch("mvsn_plane_sweep_setup_sources");
}
