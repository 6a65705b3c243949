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
