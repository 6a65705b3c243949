// ndt_libm_f32.hip.h -- float32 cos / sin as glibc computes them (ndt_params::libm_f32 = 1).
// Part of libndt_mi355x.so: included by ndt_mi355x.hip inside its anonymous namespace.  Not a standalone header.
//
// PCL builds final_transformation_ = Translation3f * AngleAxisf(float(yaw), Z) in every trial of the line search; Eigen's
// AngleAxis::toRotationMatrix calls std::cos / std::sin on the float -- libm's cosf / sinf.  Those are NOT correctly
// rounded in glibc (1.3 % of the arguments differ by one ulp), so "correctly rounded" is a model, not the reference.
// This is glibc's own algorithm (since 2.28: sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h -- the ARM optimized
// routines: argument as double, fast range reduction by multiples of pi/2 for |x| < 120, degree-7 / degree-8 polynomials
// in double, one rounding to float at the end), restated operation for operation as the x86-64 FMA build evaluates it
// (every a + b * c is one fused multiply-add: the build glibc's ifunc selects on every CPU with FMA).  A C twin of these
// lines was run against libm's sinf / cosf on ALL 2 246 049 792 floats with |x| < 120: no difference
// (tests/libm_f32_twin.c, tests/test_libm_f32.py; without the fusing 12 / 22 arguments differ).  Arguments of 120 and
// more (a yaw never gets there) fall back to the correctly rounded value.
// Constants: __sincosf_table of glibc 2.35 (two entries: the polynomials and their negatives).

struct SinCosF32 { double c0, c1, c2, c3, c4, s1, s2, s3; };
__device__ __forceinline__ float sincosf_poly_glibc(double x, double x2, bool neg_table, int n) {
  const double sg = neg_table ? -1.0 : 1.0;      // table 1 holds the negated cosine coefficients (the sine's are the same)
  if ((n & 1) == 0) {
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2;
    const double s1 = __builtin_fma(x2, s3c, s2c);
    const double x7 = x3 * x2;
    const double s = __builtin_fma(x3, s1c, x);
    return (float)__builtin_fma(x7, s1, s);
  }
  const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5,
               c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
  const double x4 = x2 * x2;
  const double cc2 = __builtin_fma(x2, c4, c3);
  const double cc1 = __builtin_fma(x2, c2, c1);
  const double x6 = x4 * x2;
  const double c = __builtin_fma(x2, cc1, c0);
  return (float)__builtin_fma(x6, cc2, c);
}
__device__ __forceinline__ unsigned abstop12_f32(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }
// which = 0: sinf(y); 1: cosf(y)
__device__ __forceinline__ float sincosf_glibc(float y, int which) {
  const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;     // 2^24 * 2/pi, pi/2
  double x = (double)y;
  const unsigned top = abstop12_f32(y);
  if (top < abstop12_f32(0x1.921FB6p-1f)) {                                   // |y| < pi/4
    if (top < abstop12_f32(0x1p-12f)) return which ? 1.0f : y;
    return sincosf_poly_glibc(x, x * x, false, which);
  }
  if (top < abstop12_f32(120.0f)) {
    const double r = x * hpi_inv;
    const int n = ((int)r + 0x800000) >> 24;                                   // quadrant, round to nearest
    x = __builtin_fma(-(double)n, hpi, x);
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;            // sign[] = {1, -1, -1, 1}
    return sincosf_poly_glibc(x * sgn, x * x, (n & 2) != 0, n ^ which);
  }
  return (float)(which ? cos((double)y) : sin((double)y));
}

// cos / sin of a double for the angle terms of J_E / H_E (and, rounded to float, the correctly-rounded model libm_f32 = 0):
// a yaw is a small angle, so the general sincos of the device library -- ~240 instructions with its large-argument
// machinery, on ONE lane with the whole workgroup waiting behind every pass -- is replaced for |x| <= 8 by the textbook
// short form: nearest multiple of pi/2 taken off with two fused multiply-adds (pi/2 as a double-double), fdlibm's
// kernel polynomials on [-pi/4, pi/4].  Against glibc's (correctly rounded in practice) sin / cos on 2e7 arguments incl.
// the neighbourhoods of the multiples of pi/2: at most one ulp; rounded to float equal to (float)sin / cos on every
// seventh float below 4 (tests/libm_f32_twin.c `f64`).  Larger arguments take the library routine.
__device__ __forceinline__ void sincos_small(double x, double &sn, double &cs) {
  if (!(fabs(x) <= 8.0)) { sincos(x, &sn, &cs); return; }
  const double k = rint(x * 0x1.45f306dc9c883p-1);
  double r = __builtin_fma(-k, 0x1.921fb54442d18p+0, x);
  r = __builtin_fma(-k, 0x1.1a62633145c07p-54, r);
  const double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double v = z * r;
  const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2);
  const double s = __builtin_fma(v, __builtin_fma(z, rs, S1), r);
  const double rc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double c = w + (((1.0 - w) - hz) + z * rc);
  const int q = (int)k & 3;
  double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  if (q == 1 || q == 2) cc = -cc;
  if (q == 2 || q == 3) ss = -ss;
  sn = ss; cs = cc;
}

// ------------------------------------------------------------------------------------------
// The initial yaw as Eigen + glibc compute it (libm_f32 = 1).  computeTransformation's prologue reads the angles back from
// the guess matrix with Affine3f.rotation().eulerAngles(0, 1, 2); rotation() of an AFFINE transform runs a float JacobiSVD
// (U V^T with the determinant's sign folded in), and eulerAngles calls the platform's atan2f.  Both restated in float32
// operation for operation (no contraction: the file is built with -ffp-contract=off; float division and square root are
// correctly rounded on the device) -- the twin of the CPU checker's eigen_rotation_z / eigen_init_yaw, which is held
// against the reference's vendored Eigen bit for bit (tests/test_eigen_pins.py), and which the GPU parity tests hold this
// file to; atanf / atan2f are glibc 2.35's (sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c: fdlibm's float versions), whose
// C twin equals libm on every third float (atanf) and on 4e7 pairs (tests/atan2f_twin.c).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float atanf_glibc(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f,
                        -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f,
                        1.6285819933e-02f};
  const int hx = (int)__float_as_uint(x), ix = hx & 0x7fffffff;
  int id;
  float hi = 0.f, lo = 0.f;
  if (ix >= 0x4c000000) {
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) { if (ix < 0x31000000) return x; id = -1; }
  else {
    x = fabsf(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; hi = atanhi[0]; lo = atanlo[0]; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else { id = 1; hi = atanhi[1]; lo = atanlo[1]; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; hi = atanhi[2]; lo = atanlo[2]; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else { id = 3; hi = atanhi[3]; lo = atanlo[3]; x = -1.0f / x; }
    }
  }
  const float z = x * x, w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = hi - ((x * (s1 + s2) - lo) - x);
  return hx < 0 ? -r : r;
}
__device__ __forceinline__ float atan2f_glibc(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int hx = (int)__float_as_uint(x), ix = hx & 0x7fffffff, hy = (int)__float_as_uint(y), iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return atanf_glibc(y);
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000 || iy == 0x7f800000) return (float)atan2((double)y, (double)x);      // (matrix entries are finite)
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = atanf_glibc(fabsf(y / x));
  return m == 0 ? z : (m == 1 ? -z : (m == 2 ? pi - (z - pi_lo) : (z - pi_lo) - pi));
}
// (every index below is a compile-time constant -- the pair loops are unrolled through templates and the sort is written out --
//  so that the three matrices live in registers: with run-time indices they went to scratch memory, 384 bytes per lane for
//  every lane of the match kernel, and the whole kernel ran 3.5 % slower)
template <int X0, int XS, int Y0, int YS, int N>
__device__ __forceinline__ void eig_apply_rot(float (&A)[9], float c, float s) {
  if (c == 1.0f && s == 0.0f) return;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float xi = A[X0 + i * XS], yi = A[Y0 + i * YS];
    const float a = c * xi, b = s * yi, d = -s * xi, e = c * yi;
    A[X0 + i * XS] = a + b; A[Y0 + i * YS] = d + e;
  }
}
template <int P, int Q>
__device__ __forceinline__ bool eig_svd_pair(float (&W)[9], float (&U)[9], float (&V)[9], float &maxDiag) {
  const float precision = 2 * FLT_EPSILON, tiny = FLT_MIN;
  const float thr = fmaxf(tiny, precision * maxDiag);
  if (!(fabsf(W[3 * P + Q]) > thr || fabsf(W[3 * Q + P]) > thr)) return false;
  float m00 = W[3 * P + P], m01 = W[3 * P + Q], m10 = W[3 * Q + P], m11 = W[3 * Q + Q];   // real_2x2_jacobi_svd
  const float t = m00 + m11, d = m10 - m01;
  float r1c, r1s;
  if (fabsf(d) < tiny) { r1s = 0; r1c = 1; }
  else { const float u = t / d; const float tmp = sqrtf(1.0f + u * u); r1s = 1.0f / tmp; r1c = u / tmp; }
  if (!(r1c == 1.0f && r1s == 0.0f)) {                                                   // m.applyOnTheLeft(0, 1, rot1)
    const float a0 = r1c * m00, b0 = r1s * m10, d0 = -r1s * m00, e0 = r1c * m10;
    const float a1 = r1c * m01, b1 = r1s * m11, d1 = -r1s * m01, e1 = r1c * m11;
    m00 = a0 + b0; m10 = d0 + e0; m01 = a1 + b1; m11 = d1 + e1;
  }
  float jrc, jrs;                                                                        // makeJacobi(m00, m01, m11)
  {
    const float deno = 2.0f * fabsf(m01);
    if (deno < tiny) { jrc = 1; jrs = 0; }
    else {
      const float tau = (m00 - m11) / deno; const float w = sqrtf(tau * tau + 1.0f);
      const float tt = tau > 0 ? 1.0f / (tau + w) : 1.0f / (tau - w);
      const float sign_t = tt > 0 ? 1.0f : -1.0f; const float n = 1.0f / sqrtf(tt * tt + 1.0f);
      jrs = -sign_t * (m01 / fabsf(m01)) * fabsf(tt) * n; jrc = n;
    }
  }
  const float oc = jrc, os = -jrs;                                                       // rot1 * j_right^T
  const float jlc = r1c * oc - r1s * os, jls = r1c * os + r1s * oc;
  eig_apply_rot<3 * P, 1, 3 * Q, 1, 3>(W, jlc, jls);      // W.applyOnTheLeft(p, q, j_left): rows p, q
  eig_apply_rot<P, 3, Q, 3, 3>(U, jlc, jls);              // U.applyOnTheRight(p, q, j_left^T): columns p, q
  eig_apply_rot<P, 3, Q, 3, 3>(W, jrc, -jrs);             // W.applyOnTheRight(p, q, j_right)
  eig_apply_rot<P, 3, Q, 3, 3>(V, jrc, -jrs);
  maxDiag = fmaxf(maxDiag, fmaxf(fabsf(W[3 * P + P]), fabsf(W[3 * Q + Q])));
  return true;
}
template <int A, int B>
__device__ __forceinline__ void eig_swap_cols(float (&sv)[3], float (&U)[9], float (&V)[9]) {
  { const float t = sv[A]; sv[A] = sv[B]; sv[B] = t; }
#pragma unroll
  for (int r = 0; r < 3; ++r) { float a = U[3 * r + B]; U[3 * r + B] = U[3 * r + A]; U[3 * r + A] = a;
                                a = V[3 * r + B]; V[3 * r + B] = V[3 * r + A]; V[3 * r + A] = a; }
}
__device__ __forceinline__ float eigen_init_yaw(float c, float s) {
  const float omc = 1.0f - c, m22 = omc + c;               // AngleAxisf::toRotationMatrix: cos_axis.z * axis.z + c
  float W[9] = {c, -s, 0, s, c, 0, 0, 0, m22};
  float U[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  float scale = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { const float a = fabsf(W[i]); if (a > scale) scale = a; }
  if (scale == 0) scale = 1;
#pragma unroll
  for (int i = 0; i < 9; ++i) W[i] = W[i] / scale;
  float maxDiag = fmaxf(fabsf(W[0]), fmaxf(fabsf(W[4]), fabsf(W[8])));
  bool finished = false;
  for (int guard = 0; !finished && guard < 64; ++guard) {
    // (the pairs (2,0) and (2,1) of Eigen's sweep never fire here: W02, W12, W20, W21 start as exact zeros and a rotation of
    //  rows / columns 0 and 1 only ever adds zeros to them -- their tests `|W| > threshold` are false by construction)
    finished = !eig_svd_pair<1, 0>(W, U, V, maxDiag);
  }
  float sv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float a = W[4 * i]; sv[i] = fabsf(a);
    if (a < 0) { U[i] = -U[i]; U[3 + i] = -U[3 + i]; U[6 + i] = -U[6 + i]; }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) sv[i] *= scale;
  // descending order, the first maximum of the remaining ones to the front (JacobiSVD.h step 4)
  if (!(sv[0] >= sv[1] && sv[0] >= sv[2])) {
    if (sv[1] >= sv[2]) { if (sv[1] != 0) eig_swap_cols<0, 1>(sv, U, V); } else { if (sv[2] != 0) eig_swap_cols<0, 2>(sv, U, V); }
  }
  if (sv[0] != 0 && sv[2] > sv[1]) eig_swap_cols<1, 2>(sv, U, V);
  float P[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float a0 = U[3 * i] * V[3 * j], a1 = U[3 * i + 1] * V[3 * j + 1], a2 = U[3 * i + 2] * V[3 * j + 2]; const float a12 = a1 + a2; P[3 * i + j] = a0 + a12; }
  float x;
  { const float h0 = P[4] * P[8], h1 = P[5] * P[7], g0 = P[3] * P[8], g1 = P[5] * P[6], f0 = P[3] * P[7], f1 = P[4] * P[6];
    const float d0 = h0 - h1, d1 = g0 - g1, d2 = f0 - f1; const float t0 = P[0] * d0, t1 = P[1] * d1, t2 = P[2] * d2; const float u = t0 - t1; x = u + t2; }
  U[0] = U[0] / x; U[3] = U[3] / x; U[6] = U[6] / x;       // m.col(0) /= x
  // R = m V^T: only the entries eulerAngles reads -- R10, R11 (row 1), R12, R22, R20, R21
#define NDT_EIG_R(i, j) (U[3 * (i)] * V[3 * (j)] + (U[3 * (i) + 1] * V[3 * (j) + 1] + U[3 * (i) + 2] * V[3 * (j) + 2]))
  const float R10 = NDT_EIG_R(1, 0), R11 = NDT_EIG_R(1, 1), R12 = NDT_EIG_R(1, 2), R20 = NDT_EIG_R(2, 0), R21 = NDT_EIG_R(2, 1), R22 = NDT_EIG_R(2, 2);
#undef NDT_EIG_R
  const float res0 = atan2f_glibc(R12, R22);
  const float s1 = sincosf_glibc(res0, 0), c1 = sincosf_glibc(res0, 1);
  const float n0 = s1 * R20, n1 = c1 * R10, d0 = c1 * R11, d1 = s1 * R21;
  return -atan2f_glibc(n0 - n1, d0 - d1);
}
