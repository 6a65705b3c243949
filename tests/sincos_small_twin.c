/* sincos_small_twin.c -- C twin of sincos_small() in ndt_slam_amd/csrc/ndt_libm_f32.hip.h (the short fp64 sincos the optimiser
 * step uses for a yaw), against this machine's libm: at most one ulp from sin / cos on 2e7 arguments in [-4, 4] incl. the
 * neighbourhoods of the multiples of pi/2, and -- rounded to float -- equal to (float)sin / (float)cos on every seventh float
 * below 4 (the correctly-rounded model of ndt_params::libm_f32 = 0).  TEST INFRASTRUCTURE (tests/test_libm_f32.py).
 * Build: gcc -O2 -ffp-contract=off -mfma sincos_small_twin.c -lm */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static inline void my_sincos(double x, double *sn, double *cs) {
  /* |x| <= ~8: k = nearest multiple of pi/2, two-term Cody-Waite with fma, fdlibm kernels */
  const double k = rint(x * 0x1.45f306dc9c883p-1);
  double r = fma(-k, 0x1.921fb54442d18p+0, x);
  r = fma(-k, 0x1.1a62633145c07p-54, r);
  const double z = r * r;
  /* kernel_sin */
  const double S1=-1.66666666666666324348e-01,S2=8.33333333332248946124e-03,S3=-1.98412698298579493134e-04,S4=2.75573137070700676789e-06,S5=-2.50507602534068634195e-08,S6=1.58969099521155010221e-10;
  const double C1=4.16666666666666019037e-02,C2=-1.38888888888741095749e-03,C3=2.48015872894767294178e-05,C4=-2.75573143513906633035e-07,C5=2.08757232129817482790e-09,C6=-1.13596475577881948265e-11;
  const double v = z * r;
  const double rs = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
  const double s = fma(v, fma(z, rs, S1), r);
  const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double c = w + (((1.0 - w) - hz) + z * rc);
  const int q = (int)k & 3;
  double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  if (q == 1) { cc = -cc; } else if (q == 2) { ss = -ss; cc = -cc; } else if (q == 3) { ss = -ss; }
  *sn = ss; *cs = cc;
}
static int64_t ulps(double a, double b){ int64_t x,y; memcpy(&x,&a,8); memcpy(&y,&b,8); if(x<0)x=INT64_MIN-x; if(y<0)y=INT64_MIN-y; return llabs(x-y); }
int main(){ srand48(1); int64_t ws=0,wc=0; long n1s=0,n1c=0; long N=20000000;
  for(long i=0;i<N;++i){ double x=(drand48()*2-1)*4.0; if(i%7==0) x=(double)(float)x; if (i%11==0) x = round(x/1.5707963267948966)*1.5707963267948966 + (drand48()-0.5)*1e-6;
    double s,c; my_sincos(x,&s,&c); double rs=sin(x), rc=cos(x); int64_t a=ulps(s,rs), b=ulps(c,rc); if(a>ws)ws=a; if(b>wc)wc=b; n1s+=a>0; n1c+=b>0;
    /* float rounding agreement */ }
  printf("max ulp sin %ld cos %ld; differing %ld %ld of %ld\n",(long)ws,(long)wc,n1s,n1c,N);
  long badf=0; for(uint32_t u=0; u<0x40800000u; u+=7){ float y; memcpy(&y,&u,4); double s,c; my_sincos((double)y,&s,&c); if((float)s!=(float)sin((double)y)||(float)c!=(float)cos((double)y)) badf++; }
  printf("float-rounded results differing from (float)sin/cos(double): %ld\n", badf);
  return (ws > 1 || wc > 1 || badf) ? 1 : 0; }
