// Plane-sweep set-up: baseline normalisation, idepth samples, homography families.
// One workgroup per chain.  Everything the kernels downstream consume -- the idepth samples, H at levels 0 and 4,
// H_inc -- is formed in the reference's own fp32 operation order (namespace ref32 below: torch's CPU inverses, small
// products, sgemm and summation order, found by matching bits on the host and pinned by tests/golden/g11), so that the
// outputs carry the reference's bits; a double-precision evaluation, rounded once where the reference's pipeline
// rounds, remains for intrinsics of another form (and for level-4 grids outside 8 .. 8192 pixels: the samples).
// include/mvsn_hip.h names the call sites replaced.
#include "mvsn_common.h"

#ifndef MVSN_SETUP_FP64_H   // A/B aid: 1 = the homographies from the fp64 evaluation, rounded once (rounds 1-5)
#define MVSN_SETUP_FP64_H 0
#endif

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

// ---------------------------------------------------------------------------------------------------------------------
// The homographies the kernels CONSUME (H at levels 0 and 4), operation by operation as the reference's fp32 pipeline
// forms them (round 6).  The fp64 evaluation above is more accurate, but it is not what the reference computes: its
// H = K (R + t idepth e3^T) K^-1 chains two fp32 LAPACK inverses and two naive 3x3 products, and one ulp of the level-0
// translation entries (3e-5 px at ~300 px) is 4.5e-5 of a warped noise frame -- the whole forward's deviation from the
// reference entered there (profiles/r06_parity/).  What torch's CPU path does, found by matching bits on the host:
//   * inverse(T) on a contiguous (B,4,4) tensor = ATen's linalg_solve_ex shortcut: LU of the TRANSPOSE (right-looking,
//     first-maximum partial pivoting, FMA updates, a column scaled by the reciprocal pivot -- except the last,
//     one-element column, which is divided), then getrs with trans = 'T' on the identity, then the row interchanges in
//     reverse.  MKL's small-matrix solve is neither a plain FMA chain nor a plain rounded one; unknown by unknown
//     (each formula matched on 1200 of 1200 random cases, the whole inverse on 8000 of 8000 entries of random matrices
//     and every entry of the test poses, residues included -- tests/test_reference_geometry_cpu.py):
//       U^T y = e_c   reciprocal diagonal; products rounded and subtracted one by one; only y3's last term is fused;
//       L^T x = y     x3 = y3; x2 = fma(-l32, x3, y2); x1 = y1 - fma(l21, x2, l31 x3);
//                     x0 = y0 - fma(l30, x3, fma(l10, x1, l20 x2))      (a dot product, first product rounded);
//   * inverse(H[:, d-1].unsqueeze(1)) of a plane's 3x3 homographies (multi_view_stereonet.py:281; the slice of the
//     reference's permuted (D,B,3,3) family is contiguous for every batch size, so this is the same shortcut): LU of
//     the transpose as above; y without a fused operation; x2 = y2; x1 = fma(-l21, x2, y1);
//     x0 = y0 - fma(l10, x1, l20 x2)       (5400 of 5400 entries of random matrices, every H tried);
//   * inverse(K[:, :3, :3]) of an upper-triangular intrinsics matrix = LAPACK strti2: reciprocal diagonal,
//     -(c * (1 / f)) above it;
//   * (N,3,3) @ (N,3,3) = ATen's small-matrix bmm: acc = 0, acc += a[i][k] * b[k][j] for k = 0, 1, 2, every operation rounded.
// (stereo/image_predictor.py:446-459, multi_view_stereonet.py:167-194; the idepth samples keep their own path above.)
namespace ref32 {

__device__ inline void inverse_pose(const float *T, float *X) {
#pragma clang fp contract(off)
  float L[16];
  int ip[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) L[i * 4 + j] = T[j * 4 + i];
  for (int j = 0; j < 4; ++j) {
    int p = j;
    float m = fabsf(L[j * 4 + j]);
    for (int i = j + 1; i < 4; ++i)
      if (fabsf(L[i * 4 + j]) > m) m = fabsf(L[i * 4 + j]), p = i;
    ip[j] = p;
    if (p != j)
      for (int k = 0; k < 4; ++k) {
        const float t = L[j * 4 + k];
        L[j * 4 + k] = L[p * 4 + k], L[p * 4 + k] = t;
      }
    const float piv = L[j * 4 + j], r = 1.0f / piv;
    for (int i = j + 1; i < 4; ++i) L[i * 4 + j] = (3 - j <= 1) ? L[i * 4 + j] / piv : L[i * 4 + j] * r;
    for (int i = j + 1; i < 4; ++i)
      for (int k = j + 1; k < 4; ++k) L[i * 4 + k] = __builtin_fmaf(-L[i * 4 + j], L[j * 4 + k], L[i * 4 + k]);
  }
  const float r0 = 1.0f / L[0], r1 = 1.0f / L[5], r2 = 1.0f / L[10], r3 = 1.0f / L[15];
  for (int c = 0; c < 4; ++c) {
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    b[c] = 1.0f;
    // U^T y = e_c (L[k * 4 + i], k < i, is u_ki)
    const float y0 = b[0] * r0;
    const float p01 = L[1] * y0;
    const float y1 = (b[1] - p01) * r1;
    const float p12 = L[6] * y1, p02 = L[2] * y0;
    const float y2 = ((b[2] - p12) - p02) * r2;
    const float p13 = L[7] * y1, p03 = L[3] * y0;
    const float y3 = __builtin_fmaf(-L[11], y2, (b[3] - p13) - p03) * r3;
    // L^T x = y (unit diagonal; L[k * 4 + i], k > i, is l_ki)
    float z[4];
    z[3] = y3;
    z[2] = __builtin_fmaf(-L[14], z[3], y2);
    const float p31 = L[13] * z[3];
    z[1] = y1 - __builtin_fmaf(L[9], z[2], p31);
    const float p20 = L[8] * z[2];
    z[0] = y0 - __builtin_fmaf(L[12], z[3], __builtin_fmaf(L[4], z[1], p20));
    for (int j = 3; j >= 0; --j)
      if (ip[j] != j) {
        const float t = z[j];
        z[j] = z[ip[j]], z[ip[j]] = t;
      }
    for (int i = 0; i < 4; ++i) X[i * 4 + c] = z[i];
  }
}

// torch.inverse of one 3x3 homography (see above); M and X row-major
__device__ inline void inverse3(const float *M, float *X) {
#pragma clang fp contract(off)
  float L[9];
  int ip[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) L[i * 3 + j] = M[j * 3 + i];
  for (int j = 0; j < 3; ++j) {
    int p = j;
    float m = fabsf(L[j * 3 + j]);
    for (int i = j + 1; i < 3; ++i)
      if (fabsf(L[i * 3 + j]) > m) m = fabsf(L[i * 3 + j]), p = i;
    ip[j] = p;
    if (p != j)
      for (int k = 0; k < 3; ++k) {
        const float t = L[j * 3 + k];
        L[j * 3 + k] = L[p * 3 + k], L[p * 3 + k] = t;
      }
    const float piv = L[j * 3 + j], r = 1.0f / piv;
    for (int i = j + 1; i < 3; ++i) L[i * 3 + j] = (2 - j <= 1) ? L[i * 3 + j] / piv : L[i * 3 + j] * r;
    for (int i = j + 1; i < 3; ++i)
      for (int k = j + 1; k < 3; ++k) L[i * 3 + k] = __builtin_fmaf(-L[i * 3 + j], L[j * 3 + k], L[i * 3 + k]);
  }
  const float r0 = 1.0f / L[0], r1 = 1.0f / L[4], r2 = 1.0f / L[8];
  for (int c = 0; c < 3; ++c) {
    float b[3] = {0.f, 0.f, 0.f};
    b[c] = 1.0f;
    const float y0 = b[0] * r0;
    const float p01 = L[1] * y0;
    const float y1 = (b[1] - p01) * r1;
    const float p12 = L[5] * y1, p02 = L[2] * y0;
    const float y2 = ((b[2] - p12) - p02) * r2;
    float z[3];
    z[2] = y2;
    z[1] = __builtin_fmaf(-L[7], z[2], y1);
    const float p20 = L[6] * z[2];
    z[0] = y0 - __builtin_fmaf(L[3], z[1], p20);
    for (int j = 2; j >= 0; --j)
      if (ip[j] != j) {
        const float t = z[j];
        z[j] = z[ip[j]], z[ip[j]] = t;
      }
    for (int i = 0; i < 3; ++i) X[i * 3 + c] = z[i];
  }
}

// true when K3 has the reference's form [[fx,0,cx],[0,fy,cy],[0,0,1]] (anything else keeps the fp64 path)
__device__ inline bool inverse_intrinsics(const float *K3, float *Ki) {
#pragma clang fp contract(off)
  if (!(K3[1] == 0.f && K3[3] == 0.f && K3[6] == 0.f && K3[7] == 0.f && K3[8] == 1.f && K3[0] != 0.f && K3[4] != 0.f)) return false;
  for (int i = 0; i < 9; ++i) Ki[i] = 0.f;
  Ki[0] = 1.0f / K3[0], Ki[4] = 1.0f / K3[4], Ki[8] = 1.0f;
  Ki[2] = -(K3[2] * Ki[0]);
  Ki[5] = -(K3[5] * Ki[4]);
  return true;
}

__device__ inline void mm3(const float *a, const float *b, float *o) {
#pragma clang fp contract(off)
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float acc = 0.f;
      for (int k = 0; k < 3; ++k) {
        const float prod = a[i * 3 + k] * b[k * 3 + j];
        acc = acc + prod;
      }
      o[i * 3 + j] = acc;
    }
}

// H = K ((R + t idepth e3^T) K^-1), R / t = rotation / translation of the inverted pose
__device__ inline void plane_homography(const float *K3, const float *Ki, const float *Tl, float idepth, float *H) {
#pragma clang fp contract(off)
  float core[9], tmp[9];
  for (int i = 0; i < 3; ++i) {
    core[i * 3] = Tl[i * 4], core[i * 3 + 1] = Tl[i * 4 + 1];
    const float ti = Tl[i * 4 + 3] * idepth;
    core[i * 3 + 2] = Tl[i * 4 + 2] + ti;
  }
  mm3(core, Ki, tmp);
  mm3(K3, tmp, H);
}

// The maximum idepth of a chain (stereo/image_predictor.py:120-209 from multi_view_stereonet.py:139-147) as the
// reference's fp32 tensor program evaluates it, pixel by pixel: every ATen elementwise op rounds; `matmul(KRKinv,
// xyz_pix)` is MKL's sgemm -- per output a product and two fused multiply-adds in the order k = 0, 1, 2 -- unless
// 3 * 3 * P < 400, where ATen's small-matrix loop runs instead (no fused operation).  One thing is NOT reproduced:
// torch's vectorised sqrt (MKL VML) is an ulp off the correctly rounded root for 0.7 % of its arguments; the mean over
// the pixels absorbs it (tests/test_reference_geometry_cpu.py: 204 of 204 chains, tests/golden/g11: 21 of 21).
struct MaxIdepth {
  float M[9], Kt[3];   // K R K^-1 and the translation column of K T_left_in_right (both by ATen's naive small products)
  float disp;          // D - 1
  int naive;           // 9 P < 400
};

__device__ inline float max_idepth_pixel(const MaxIdepth &g, float x, float y) {
#pragma clang fp contract(off)
  float inf[3], far[3];
  const float x2 = x * 1e2f, y2 = y * 1e2f;
  for (int i = 0; i < 3; ++i) {
    const float m0 = g.M[i * 3], m1 = g.M[i * 3 + 1], m2 = g.M[i * 3 + 2];
    if (g.naive) {
      const float a0 = m0 * x, a1 = m1 * y;
      inf[i] = (a0 + a1) + m2;
      const float f0 = m0 * x2, f1 = m1 * y2, f2 = m2 * 1e2f;
      far[i] = (f0 + f1) + f2;
    } else {
      const float a0 = m0 * x;
      inf[i] = __builtin_fmaf(m2, 1.0f, __builtin_fmaf(m1, y, a0));
      const float f0 = m0 * x2;
      far[i] = __builtin_fmaf(m2, 1e2f, __builtin_fmaf(m1, y2, f0));
    }
    far[i] = far[i] + g.Kt[i];
  }
  const float infx = inf[0] / inf[2], infy = inf[1] / inf[2];
  const float farx = far[0] / far[2], fary = far[1] / far[2];
  const float dx = farx - infx, dy = fary - infy;
  const float dxx = dx * dx, dyy = dy * dy;
  const float norm = sqrtf(dxx + dyy);
  const float den = norm + 1e-6f;
  const float lx = dx / den, ly = dy / den;
  const float w0 = g.M[6] * x, w1 = g.M[7] * y;
  const float wz = (w0 + w1) + g.M[8];
  const float sx = g.disp * lx, sy = g.disp * ly;
  const float ux = infx + sx, uy = infy + sy;
  const float vx = g.Kt[2] * ux, vy = g.Kt[2] * uy;
  const float A0 = g.Kt[0] - vx, A1 = g.Kt[1] - vy;
  const float wd = wz * g.disp;
  const float b0 = wd * lx, b1 = wd * ly;
  const float n0 = A0 * b0, n1 = A1 * b1, d0 = A0 * A0, d1 = A1 * A1;
  float idp = (n0 + n1) / (d0 + d1);
  idp = (norm < 1e-6f ? 0.0f : 1.0f) * idp;       // (~mask).float() * idepth: a NaN stays a NaN, as there
  return (idp > 0.0f ? 1.0f : 0.0f) * idp;        // (x > 0).float() * x
}

}  // namespace ref32

constexpr int SETUP_THREADS = 256;
constexpr int SETUP_MAX_PIXELS = 8192;   // level-4 pixels the reference-order sample path holds in LDS (else: the double path)

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
  __shared__ float s_Tl[16], s_K4i[9], s_K0i[9];   // ref32: the inverted (normalised) pose, the intrinsics' inverses
  __shared__ int s_ref32;                // ... and whether the intrinsics have the form that path covers
  __shared__ ref32::MaxIdepth s_g;       // ref32 sample path: K R K^-1, K t, D - 1
  __shared__ float s_m[SETUP_MAX_PIXELS];   // ... its per-pixel idepths, summed in torch's order below
  __shared__ float s_lane[32];

  // --- baseline renormalisation in fp32, as multi_view_stereonet.py:566-571 -------------------
  float tx = T[3], ty = T[7], tz = T[11];
  float base, tn[3];
  {
#pragma clang fp contract(off)   // T[:, :3, 3].pow(2).sum(1).sqrt(): every operation rounded, no fused multiply-add
    const float xx = tx * tx, yy = ty * ty, zz = tz * tz;
    const float sxy = xx + yy;
    base = sqrtf(sxy + zz);
    tn[0] = tx / base, tn[1] = ty / base, tn[2] = tz / base;
  }

  // --- T_left_in_right = inverse([R t; 0 1]) ---------------------------------------------------
  double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  double Rl[9];
  inv3(R, Rl);
  double tl[3];
  for (int i = 0; i < 3; ++i) tl[i] = -(Rl[i * 3] * tn[0] + Rl[i * 3 + 1] * tn[1] + Rl[i * 3 + 2] * tn[2]);

  double K4d[9] = {K4[0], K4[1], K4[2], K4[4], K4[5], K4[6], K4[8], K4[9], K4[10]};
  double K4inv[9];
  inv3(K4d, K4inv);

  // --- the reference-order pieces every later step shares (ref32), by one thread --------------------------------------
  const int P = rows4 * cols4;
  if (tid == 0) {
    float Tn[16], K0f[9], K4f[9];
    for (int i = 0; i < 16; ++i) Tn[i] = T[i];
    Tn[3] = tn[0], Tn[7] = tn[1], Tn[11] = tn[2];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) K0f[i * 3 + j] = K0[i * 4 + j], K4f[i * 3 + j] = K4[i * 4 + j];
    ref32::inverse_pose(Tn, s_Tl);
    const bool ok = ref32::inverse_intrinsics(K0f, s_K0i) && ref32::inverse_intrinsics(K4f, s_K4i) && !MVSN_SETUP_FP64_H;
    s_ref32 = ok ? 1 : 0;
    if (ok) {
#pragma clang fp contract(off)
      // disparity_to_idepth's own matrices: inverse(K) of the 4x4, K R K^-1 and (K T_left_in_right)[:3, 3] by ATen's
      // naive small products (image_predictor.py:148-160)
      float K4x4[16], Kinv[16], Kinv3[9], Tl3[9], tmp3[9];
      for (int i = 0; i < 16; ++i) K4x4[i] = K4[i];
      ref32::inverse_pose(K4x4, Kinv);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Kinv3[i * 3 + j] = Kinv[i * 4 + j], Tl3[i * 3 + j] = s_Tl[i * 4 + j];
      ref32::mm3(Tl3, Kinv3, tmp3);
      ref32::mm3(K4f, tmp3, s_g.M);
      for (int i = 0; i < 3; ++i) {
        float acc = 0.f;
        for (int k = 0; k < 4; ++k) {
          const float prod = K4x4[i * 4 + k] * s_Tl[k * 4 + 3];
          acc = acc + prod;
        }
        s_g.Kt[i] = acc;
      }
      s_g.disp = (float)(D - 1);
      s_g.naive = 9 * P < 400;
    }
  }
  __syncthreads();
  const bool ref_samples = s_ref32 && P >= 8 && P <= SETUP_MAX_PIXELS;   // (rows shorter than one vector: another ATen sum)

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
  for (int p = tid; p < P && !ref_samples; p += SETUP_THREADS) {
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
  if (ref_samples) {   // the reference's fp32 program per pixel; the sum below in torch's order
    for (int p = tid; p < P; p += SETUP_THREADS) {
      const float m = ref32::max_idepth_pixel(s_g, (float)(p % cols4), (float)(p / cols4));
      s_m[p] = m;
      cnt += m > 0.0f ? 1 : 0;
    }
  }
  s_sum[tid] = acc;
  s_cnt[tid] = cnt;
  __syncthreads();
  if (ref_samples && tid < 32) {
    // torch.sum over a contiguous row of floats (ATen SumKernel.cpp, 8-float vectors): 4 x 8 lanes take the elements
    // l, l + 32, l + 64, ... of the first (P / 32) * 32 in order, through a four-level cascade that folds level j - 1
    // into level j every 16^j steps
#pragma clang fp contract(off)
    const int rounds = (P / 8) / 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    while (i + 16 <= rounds) {
      for (int j = 0; j < 16; ++j, ++i) a0 = a0 + s_m[i * 32 + tid];
      a1 = a1 + a0, a0 = 0.f;
      if ((i & (15 << 4)) == 0) {
        a2 = a2 + a1, a1 = 0.f;
        if ((i & (15 << 8)) == 0) a3 = a3 + a2, a2 = 0.f;
      }
    }
    for (; i < rounds; ++i) a0 = a0 + s_m[i * 32 + tid];
    a0 = a0 + a1;
    a0 = a0 + a2;
    a0 = a0 + a3;
    s_lane[tid] = a0;
  }
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
    if (ref_samples) {
#pragma clang fp contract(off)
      const int vecs = P / 8, rounds = vecs / 4;
      for (int v = rounds * 4; v < vecs; ++v)        // whole vectors beyond the four-lane rounds: into lane group 0
        for (int l = 0; l < 8; ++l) s_lane[l] = s_lane[l] + s_m[v * 8 + l];
      for (int k = 1; k < 4; ++k)
        for (int l = 0; l < 8; ++l) s_lane[l] = s_lane[l] + s_lane[k * 8 + l];
      float total = 0.f;
      for (int p = vecs * 8; p < P; ++p) total = total + s_m[p];   // the scalar tail first, then the eight partial sums
      for (int l = 0; l < 8; ++l) total = total + s_lane[l];
      top = total / (float)s_cnt[0];
    }
    if (top > 2.0f) top = 2.0f;
    if (1.0f / top < tn[2]) top = 1.0f / tn[2];      // keep samples in front of the source camera
    s_top = top;
    baseline_out[n] = base;
    // full-resolution homography of plane 0 (multi_view_stereonet.py:254-255)
    double K0d[9] = {K0[0], K0[1], K0[2], K0[4], K0[5], K0[6], K0[8], K0[9], K0[10]};
    double K0inv[9], H0[9];
    inv3(K0d, K0inv);
    plane_homography(K0d, K0inv, Rl, tl, 0.0, H0);
    // the reference's own fp32 chain where the intrinsics have its form (ref32 above); the fp64 value otherwise
    float K0f[9], H0f[9];
    for (int i = 0; i < 9; ++i) K0f[i] = (float)K0d[i];
    const bool ok = s_ref32 != 0;
    if (ok) ref32::plane_homography(K0f, s_K0i, s_Tl, 0.0f, H0f);
    for (int i = 0; i < 9; ++i) H0_out[(size_t)n * 9 + i] = ok ? H0f[i] : (float)H0[i];
  }
  __syncthreads();
  const float delta = s_top / (float)(D - 1);

  for (int d = tid; d < D; d += SETUP_THREADS) {
    float sd = (float)d * delta;
    samples_out[(size_t)n * D + d] = sd;
    double H[9];
    plane_homography(K4d, K4inv, Rl, tl, (double)sd, H);
    float Hf[9];
    for (int i = 0; i < 9; ++i) Hf[i] = (float)H[i];
    if (s_ref32) {
      float K4f[9];
      for (int i = 0; i < 9; ++i) K4f[i] = (float)K4d[i];
      ref32::plane_homography(K4f, s_K4i, s_Tl, sd, Hf);
    }
    for (int i = 0; i < 9; ++i) H4_out[((size_t)n * D + d) * 9 + i] = Hf[i];
    float *inc = Hinc_out + ((size_t)n * D + d) * 9;
    if (d == 0) {
      for (int i = 0; i < 9; ++i) inc[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    } else {
      // previous plane, rounded to fp32 like the reference's stored H (multi_view_stereonet.py:281-282)
      float sp = (float)(d - 1) * delta;
      double Hp[9], Hpf[9], Hpinv[9], Hcf[9], Hi[9];
      plane_homography(K4d, K4inv, Rl, tl, (double)sp, Hp);
      float Hpr[9];
      for (int i = 0; i < 9; ++i) Hpr[i] = (float)Hp[i];
      if (s_ref32) {
        float K4f[9];
        for (int i = 0; i < 9; ++i) K4f[i] = (float)K4d[i];
        ref32::plane_homography(K4f, s_K4i, s_Tl, sp, Hpr);
      }
      for (int i = 0; i < 9; ++i) {
        Hpf[i] = (double)Hpr[i];
        Hcf[i] = (double)Hf[i];
      }
      inv3(Hpf, Hpinv);
      mul3(Hpinv, Hcf, Hi);
      float Hif[9];
      for (int i = 0; i < 9; ++i) Hif[i] = (float)Hi[i];
      if (s_ref32) {   // inverse(H[d-1]) @ H[d] as torch evaluates it (multi_view_stereonet.py:281-282)
        float Hpi[9];
        ref32::inverse3(Hpr, Hpi);
        ref32::mm3(Hpi, Hf, Hif);
      }
      for (int i = 0; i < 9; ++i) inc[i] = Hif[i];
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
