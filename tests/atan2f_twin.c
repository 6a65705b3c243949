/* atan2f_twin.c -- C twin of atanf_glibc / atan2f_glibc in ndt_slam_amd/csrc/ndt_libm_f32.hip.h (glibc 2.35's float atanf /
 * atan2f: fdlibm's float versions, plain float arithmetic, no fused operations) against this machine's libm: atanf on every
 * third float, atan2f on 4e7 pairs (points of the unit circle as sinf / cosf give them, random pairs, signed zeros).
 * TEST INFRASTRUCTURE (tests/test_libm_f32.py).  Build: gcc -O2 -fopenmp -ffp-contract=off atan2f_twin.c -lm */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static inline uint32_t asu(float f){uint32_t u; memcpy(&u,&f,4); return u;}
static inline float asf(uint32_t u){float f; memcpy(&f,&u,4); return f;}
static const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
static const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
static const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f,
  -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
static float my_atanf(float x) {
  float w, s1, s2, z; int32_t ix, hx, id;
  hx = (int32_t)asu(x); ix = hx & 0x7fffffff;
  if (ix >= 0x4c000000) { if (ix > 0x7f800000) return x + x; if (hx > 0) return atanhi[3] + atanlo[3]; else return -atanhi[3] - atanlo[3]; }
  if (ix < 0x3ee00000) { if (ix < 0x31000000) { return x; } id = -1; }
  else { x = fabsf(x);
    if (ix < 0x3f980000) { if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); } else { id = 1; x = (x - 1.0f) / (x + 1.0f); } }
    else { if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); } else { id = 3; x = -1.0f / x; } } }
  z = x * x; w = z * z;
  s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return (hx < 0) ? -z : z;
}
static float my_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  float z; int32_t k, m, hx, hy, ix, iy;
  hx = (int32_t)asu(x); ix = hx & 0x7fffffff; hy = (int32_t)asu(y); iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return my_atanf(y);
  m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) { switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; } }
  if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000 || iy == 0x7f800000) return atan2f(y, x);   /* infinities: never here */
  k = (iy - ix) >> 23;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = my_atanf(fabsf(y / x));
  switch (m) { case 0: return z; case 1: return -z; case 2: return pi - (z - pi_lo); default: return (z - pi_lo) - pi; }
}
int main() {
  long bad1 = 0;
#pragma omp parallel for reduction(+:bad1)
  for (long u = 0; u < 0x7f800000L; u += 3) { float x = asf((uint32_t)u); if (asu(my_atanf(x)) != asu(atanf(x))) bad1++; if (asu(my_atanf(-x)) != asu(atanf(-x))) bad1++; }
  printf("atanf mismatches (every 3rd float): %ld\n", bad1);
  long bad2 = 0; srand48(5);
  for (long i = 0; i < 40000000; ++i) { float a = (float)(drand48() * 6.4 - 3.2); float s = sinf(a), c = cosf(a); if (i % 3 == 0) { s = (float)(drand48()*2-1); c = (float)(drand48()*2-1); }
    if (i % 1001 == 0) s = 0.0f; if (i % 1003 == 0) c = 0.0f; if (i % 1007 == 0) s = -0.0f;
    if (asu(my_atan2f(s, c)) != asu(atan2f(s, c))) { if (bad2 < 5) printf("  %.9g %.9g -> %.9g vs %.9g\n", s, c, my_atan2f(s,c), atan2f(s,c)); bad2++; } }
  printf("atan2f mismatches: %ld of 40000000\n", bad2);
  return (bad1 || bad2) ? 1 : 0;
}
