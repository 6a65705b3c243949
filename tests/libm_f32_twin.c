/* libm_f32_twin.c -- C twin of ndt_slam_amd/csrc/ndt_libm_f32.hip.h (the device's restatement of glibc >= 2.28's sinf / cosf),
 * run against THIS machine's libm on every float with |x| < 120 (tests/test_libm_f32.py; ~5 s on 8 cores).
 * TEST INFRASTRUCTURE.  Build: gcc -O2 -fopenmp -ffp-contract=off -DUSE_FMA -mfma libm_f32_twin.c -lm
 * (USE_FMA: every a + b * c fused, as glibc's x86-64 FMA build evaluates it; without it 12 / 22 arguments differ).
 * Usage: ./twin [stride]   -- stride 1 = exhaustive. */
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
static inline uint32_t asuint(float f){uint32_t u; memcpy(&u,&f,4); return u;}
static inline float asfloat(uint32_t u){float f; memcpy(&f,&u,4); return f;}
typedef struct { double sign[4]; double hpi_inv, hpi, c0,c1,c2,c3,c4, s1,s2,s3; } sincos_t;
static const sincos_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static inline uint32_t abstop12(float x){ return (asuint(x)>>20)&0x7ff; }
#ifdef USE_FMA
#define MADD(a,b,c) fma((a),(b),(c))
#else
#define MADD(a,b,c) ((a)*(b)+(c))
#endif
static inline float sinf_poly(double x, double x2, const sincos_t *p, int n){
  if ((n&1)==0){ double x3=x*x2; double s1=MADD(x2,p->s3,p->s2); double x7=x3*x2; double s=MADD(x3,p->s1,x); return (float)MADD(x7,s1,s); }
  else { double x4=x2*x2; double c2=MADD(x2,p->c4,p->c3); double c1=MADD(x2,p->c2,p->c1); double x6=x4*x2; double c=MADD(x2,c1,p->c0); return (float)MADD(x6,c2,c); }
}
static inline double reduce_fast(double x, const sincos_t *p, int *np){ double r=x*p->hpi_inv; int n=((int32_t)r+0x800000)>>24; *np=n; return MADD(-(double)n,p->hpi,x); }
static float my_sinf(float y){ double x=y,s; int n; const sincos_t*p=&T[0];
  if (abstop12(y)<abstop12(0x1.921FB6p-1f)){ s=x*x; if (abstop12(y)<abstop12(0x1p-12f)) return y; return sinf_poly(x,s,p,0);} 
  else if (abstop12(y)<abstop12(120.0f)){ x=reduce_fast(x,p,&n); s=p->sign[n&3]; if(n&2)p=&T[1]; return sinf_poly(x*s,x*x,p,n);} 
  return sinf(y); }
static float my_cosf(float y){ double x=y,s; int n; const sincos_t*p=&T[0];
  if (abstop12(y)<abstop12(0x1.921FB6p-1f)){ s=x*x; if (abstop12(y)<abstop12(0x1p-12f)) return 1.0f; return sinf_poly(x,s,p,1);} 
  else if (abstop12(y)<abstop12(120.0f)){ x=reduce_fast(x,p,&n); s=p->sign[n&3]; if(n&2)p=&T[1]; return sinf_poly(x*s,x*x,p,n^1);} 
  return cosf(y); }
int main(int argc,char**argv){ long bad_s=0,bad_c=0; uint32_t hi=asuint(120.0f); long stride = argc>1 ? atol(argv[1]) : 1; long cnt=0;
#pragma omp parallel for reduction(+:bad_s,bad_c,cnt) schedule(static)
  for (long u=0; u<(long)hi; u+=stride){ cnt+=2; for (int sg=0; sg<2; ++sg){ float y=asfloat((uint32_t)u | (sg?0x80000000u:0));
    float a=my_sinf(y), b=sinf(y); if (asuint(a)!=asuint(b)) { if(bad_s<5) {}; bad_s++; }
    float c=my_cosf(y), d=cosf(y); if (asuint(c)!=asuint(d)) bad_c++; } }
  printf("sinf %ld cosf %ld of %ld\n", bad_s, bad_c, cnt); return (bad_s||bad_c) ? 1 : 0; }
