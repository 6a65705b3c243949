// ndt_mi355x.hip -- MI355X (gfx950 / CDNA4) NDT scan-matching core + its C ABI.
//
// Replaces what the reference delegates to PCL behind PoseEstimator::estimatePose
// (/root/reference/src/PoseEstimator.cpp:17-56): voxel normal-distributions build (SURVEY.md
// 8a row a2), SE(2) transform + radius lookup + score/gradient/Hessian accumulation (a4, a5),
// Newton + More-Thuente pose update (a3, a6), fitness score (a7), final Hessian (a8) and the
// pose extraction (a9).  Written for 64-wide wavefronts; no MFMA (there is no dense
// contraction on this path); fp64 accumulation with fixed-order reductions so a run is
// deterministic and takes the same line-search branches as the CPU oracle.
//
// Floating-point contraction is OFF for the whole file (voxel statistics, float32 transform and
// the optimiser must round like the scalar reference code); the hot accumulation loop asks for
// fused multiply-adds explicitly with __builtin_fma.
//
// This file never includes or calls anything under oracle/.  Without a HIP device every entry
// point fails with NDT_E_NO_DEVICE: there is no CPU fallback.

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "ndt_mi355x.h"

#pragma clang fp contract(off)

namespace {

thread_local std::string g_last_error;

// ------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------

// Dense voxel grid padded by 2 cells on every side so that the 3x3 probe of any point whose
// voxel lies within one cell of the map's bounding box needs no bounds checks.
struct MapView {
  float inv_leaf, leaf, r2;
  int radius_inclusive, transform_sse;
  int min_bx, min_by, div_x, div_y;  // unpadded voxel grid (VoxelGridCovariance min_b_/div_b_)
  int gw, gh;                        // padded: div + 4
  const float2 *cent;                // gw*gh float32 centroids; +inf where the voxel is not in
                                     // the centroid search set (fewer than min_pts points)
  const double *rec;                 // gw*gh records of 8 doubles (64 B):
                                     // mean_x, mean_y, icov_xx, icov_xy, icov_yy, 3 pad
  const unsigned *occ;               // one bit per voxel of the unpadded grid: in the centroid search set
  const int *pt_start;               // div_x*div_y + 1 bucket offsets of the raw points
  const float2 *pts;                 // raw points bucketed by voxel, input order kept (a7)
  double d1, d2;                     // Gaussian constants (a3)
};

struct OptParams {
  double step_size, trans_eps, snap_thresh, mt_mu, mt_nu;
  int max_iter, conv_ge, stale_h_ang, mt_max_iter;
};

struct Tf32 { float c, s, tx, ty; };

// eleven partial sums of one derivative pass
struct Acc {
  double e, g0, g1, g2, hxx, hxy, hxt, hyy, hyt, htt;
  unsigned pairs;
};
constexpr int kAcc = 11;

enum Phase : int { PH_INIT = 0, PH_LS_FIRST = 1, PH_LS_INNER = 2, PH_DONE = 3 };

// Resumable optimiser state of one match.  One lane advances it after every derivative pass;
// kept free of any per-workgroup assumption so a pass can be produced by any set of waves.
struct AlignState {
  int phase, iters, evals, ref_evals, converged, step_iterations, open_interval, interval_converged;
  Tf32 T;                       // final_transformation_ (float32)
  double cj, sj, ch, sh;        // angle terms of J_E and of the (yaw,yaw) block of H_E
  double p[3], dir[3], xt[3];
  double score, g[3], H[6];     // xx xy xt yy yt tt
  double phi0, dphi0, a_l, f_l, g_l, a_u, f_u, g_u, a_t;
  double pairs;
  double n_points;
};

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ float2 load_pt(const float *xy, size_t stride, size_t i) {
  return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(xy) + i * stride);
}

// float32 matrix of the fp64 parameter vector (a4): Translation3f(float(p0), float(p1), 0) *
// AngleAxisf(float(p2), Z); std::cos/std::sin(float) modelled as correctly rounded.
__device__ __forceinline__ Tf32 tf_from_p(const double p[3]) {
  Tf32 t;
  float yaw = (float)p[2];
  double sd, cd;
  sincos((double)yaw, &sd, &cd);
  t.c = (float)cd;
  t.s = (float)sd;
  t.tx = (float)p[0];
  t.ty = (float)p[1];
  return t;
}

// pcl::transformPointCloud on a z = 0 point, float32, no contraction.
__device__ __forceinline__ void tf_apply(const Tf32 &t, int sse, float x, float y, float &ox,
                                         float &oy) {
  float ms = -t.s;
  float a = t.c * x, b = ms * y, c = t.s * x, d = t.c * y;
  if (!sse) {
    float r = a + b; ox = r + t.tx;
    float q = c + d; oy = q + t.ty;
  } else {
    float r = b + t.tx; ox = a + r;
    float q = d + t.ty; oy = c + q;
  }
}

__device__ __forceinline__ bool finite2(float x, float y) {
  return (fabsf(x) <= FLT_MAX) && (fabsf(y) <= FLT_MAX);
}

__device__ __forceinline__ void angle_cs(double snap, double yaw, double &c, double &s) {
  if (fabs(yaw) < snap) { c = 1.0; s = 0.0; }
  else { sincos(yaw, &s, &c); }
}

// ------------------------------------------------------------------------------------------
// a4 + a5: one source point -> its in-radius voxels -> score / gradient / Hessian terms
// ------------------------------------------------------------------------------------------

// Per-scan window of the voxel grid staged in LDS (the cells a scan can reach while its pose
// moves).  Two pieces share one LDS pool: a row-major rw x rh table of 16-bit slot numbers and a
// compact table of the occupied voxels' records (48 B: float32 centroid, fp64 mean, fp64 inverse
// covariance), so that the hot loop touches no global memory for map data.
struct Region { int x0, y0, rw, rh, cap, nspill; };   // origin in unpadded voxel coordinates; cap = record slots
struct __attribute__((aligned(16))) CellEntry { float2 cent; double mx, my, i00, i01, i11; };
static_assert(sizeof(CellEntry) == 48, "CellEntry layout");
constexpr int kRegionCells = 16384;           // at most 32 KiB of slot numbers
constexpr int kRegionMargin = 5;              // cells of slack around the scan's first bbox
constexpr int kPoolBytes = 147 * 1024;        // of the CU's 160 KiB LDS
// A slot number indexes the record table.  Voxels outside the search set point at the sentinel
// record `cap` (centroid = +inf, so the radius test fails by itself).  If a window holds more
// occupied voxels than the pool has room for, nspill > 0 and the whole scan reads the map from HBM.

struct Window {
  Region R;
  const unsigned short *slot;                 // LDS
  const CellEntry *ent;                       // LDS
};

// exp(x) for x <= ~0 (the NDT exponent -d2/2 * Mahalanobis^2): 2^(n/64) table * degree-5
// polynomial, ~1 ulp.  x is clamped at -800 (underflows to 0), so a NaN exponent gives 0 --
// the pair then adds nothing, exactly what the reference's `e != e` check does with it.
__constant__ double c_exp2_tab[64];
__device__ __forceinline__ double exp_neg(double x, const double *__restrict__ tab) {
  x = fmax(x, -800.0);
  const double t = rint(x * 92.332482616893657);            // 64 / ln 2
  const int n = (int)t;
  double r = __builtin_fma(-t, 0x1.62e42fefa0000p-7, x);    // ln2/64, high part (exact product)
  r = __builtin_fma(-t, 0x1.cf79abc9e3b3ap-46, r);          // low part
  double p = __builtin_fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = __builtin_fma(r, p, 1.0 / 6.0);
  p = __builtin_fma(r, p, 0.5);
  p = __builtin_fma(p, r * r, r);                           // exp(r) - 1
  const double sc = tab[n & 63];
  return ldexp(__builtin_fma(sc, p, sc), n >> 6);
}

template <bool SSE>
__device__ __forceinline__ void tf_apply_t(const Tf32 &t, float x, float y, float &ox, float &oy) {
  const float ms = -t.s;
  const float a = t.c * x, b = ms * y, c = t.s * x, d = t.c * y;
  if (!SSE) { const float r = a + b; ox = r + t.tx; const float q = c + d; oy = q + t.ty; }
  else      { const float r = b + t.tx; ox = a + r; const float q = d + t.ty; oy = c + q; }
}

// radius test of flann::L2_Simple<float> on one centroid
template <bool INCL>
__device__ __forceinline__ unsigned in_radius(float r2, float xt, float yt, float2 cc) {
  const float ex = xt - cc.x, ey = yt - cc.y;
  const float dd = ex * ex + ey * ey;
  return (INCL ? (dd <= r2) : (dd < r2)) ? 1u : 0u;
}

struct CellRec { double mx, my, i00, i01, i11; };

__device__ __forceinline__ CellRec load_rec_global(const MapView &M, size_t base, int k) {
  const int r = (k * 11) >> 5, q = k - 3 * r;                 // k / 3 for k in [0, 9)
  const double *rec = M.rec + (base + (size_t)(r * M.gw + q)) * 8;
  const double2 a = *reinterpret_cast<const double2 *>(rec);
  const double2 b = *reinterpret_cast<const double2 *>(rec + 2);
  CellRec c; c.mx = a.x; c.my = a.y; c.i00 = b.x; c.i01 = b.y; c.i11 = rec[4];
  return c;
}

struct PointTerms { double XT, YT, jx, jy, hx, hy; };

__device__ __forceinline__ PointTerms point_terms(float x, float y, float xt, float yt, double cj,
                                                  double sj, double ch, double sh) {
  // yaw column of J_E and the (yaw,yaw) block of H_E (untransformed coordinates)
  const double X = (double)x, Y = (double)y;
  PointTerms P;
  P.jx = X * (-sj) + Y * (-cj);
  P.jy = X * cj + Y * (-sj);
  P.hx = X * (-ch) + Y * sh;
  P.hy = X * (-sh) + Y * (-ch);
  P.XT = (double)xt; P.YT = (double)yt;
  return P;
}

// one (point, voxel) pair: eqs 6.9 / 6.12 / 6.13 restricted to (tx, ty, yaw)
__device__ __forceinline__ void accumulate_cell(double d2, const double *__restrict__ etab,
                                                const PointTerms &P, const CellRec &c, Acc &A) {
  const double nd2 = -d2;
  const double q0 = P.XT - c.mx, q1 = P.YT - c.my;
  const double u0 = __builtin_fma(c.i01, q1, c.i00 * q0);      // Sigma^-1 q
  const double u1 = __builtin_fma(c.i11, q1, c.i01 * q0);
  const double m = __builtin_fma(q1, u1, q0 * u0);
  double e = exp_neg(nd2 * m * 0.5, etab);
  const double e2 = d2 * e;
  if (e2 > 1.0 || e2 < 0.0) e = 0.0;                           // updateDerivatives error check
  const double at = __builtin_fma(u1, P.jy, u0 * P.jx);        // q^T Sigma^-1 dT/dyaw
  const double cx = __builtin_fma(c.i01, P.jy, c.i00 * P.jx);  // Sigma^-1 dT/dyaw
  const double cy = __builtin_fma(c.i11, P.jy, c.i01 * P.jx);
  const double v0 = nd2 * u0, v1 = nd2 * u1, vt = nd2 * at;
  A.e += e;
  A.g0 = __builtin_fma(e, u0, A.g0);
  A.g1 = __builtin_fma(e, u1, A.g1);
  A.g2 = __builtin_fma(e, at, A.g2);
  A.hxx = __builtin_fma(e, __builtin_fma(v0, u0, c.i00), A.hxx);
  A.hxy = __builtin_fma(e, __builtin_fma(v0, u1, c.i01), A.hxy);
  A.hxt = __builtin_fma(e, __builtin_fma(v0, at, cx), A.hxt);
  A.hyy = __builtin_fma(e, __builtin_fma(v1, u1, c.i11), A.hyy);
  A.hyt = __builtin_fma(e, __builtin_fma(v1, at, cy), A.hyt);
  double tt = __builtin_fma(P.jx, cx, P.jy * cy);              // J^T Sigma^-1 J
  tt = __builtin_fma(u0, P.hx, tt);                            // + q^T Sigma^-1 d2T/dyaw2
  tt = __builtin_fma(u1, P.hy, tt);
  tt = __builtin_fma(vt, at, tt);
  A.htt = __builtin_fma(e, tt, A.htt);
}

// Everything one source point contributes to a derivative pass.
// Fast path (window holds every occupied voxel, point's 3x3 neighbourhood inside it): slot
// numbers, centroids and records all come from LDS.  Otherwise the same arithmetic reads the
// global centroid grid / record array.
template <bool SSE, bool INCL>
__device__ __forceinline__ void eval_point(const MapView &M, const Window &W,
                                           const double *__restrict__ etab, const Tf32 &T, float x,
                                           float y, double cj, double sj, double ch, double sh, Acc &A) {
  float xt, yt;
  tf_apply_t<SSE>(T, x, y, xt, yt);
  const bool fin = finite2(xt, yt);
  const float fx = fminf(fmaxf(floorf(xt * M.inv_leaf), -1.0e9f), 1.0e9f);
  const float fy = fminf(fmaxf(floorf(yt * M.inv_leaf), -1.0e9f), 1.0e9f);
  const int ix = (int)fx - M.min_bx, iy = (int)fy - M.min_by;
  const bool ingrid = fin & (ix >= -1) & (ix <= M.div_x) & (iy >= -1) & (iy <= M.div_y);
  const Region &R = W.R;
  const int lx = ix - R.x0, ly = iy - R.y0;
  const bool inwin = ingrid & (lx >= 1) & (lx < R.rw - 1) & (ly >= 1) & (ly < R.rh - 1);
  // LDS probes with clamped indices (results dropped when !inwin)
  const int clx = min(max(lx, 1), max(R.rw - 2, 1)), cly = min(max(ly, 1), max(R.rh - 2, 1));
  const unsigned short *srow = W.slot + (cly - 1) * R.rw + (clx - 1);
  unsigned mask = 0;
  float lowx = INFINITY;                       // -inf <=> one of the nine voxels is occupied but not resident
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float2 cc = W.ent[srow[r * R.rw + q]].cent;
      lowx = fminf(lowx, cc.x);
      mask |= in_radius<INCL>(M.r2, xt, yt, cc) << (r * 3 + q);
    }
  if (inwin & (lowx != -INFINITY)) {
    if (!mask) return;
    A.pairs += __builtin_popcount(mask);
    const PointTerms P = point_terms(x, y, xt, yt, cj, sj, ch, sh);
#pragma nounroll
    do {
      const int k = __builtin_ctz(mask);
      mask &= mask - 1;
      const int r = (k * 11) >> 5, q = k - 3 * r;
      const CellEntry &E = W.ent[srow[r * R.rw + q]];
      CellRec c; c.mx = E.mx; c.my = E.my; c.i00 = E.i00; c.i01 = E.i01; c.i11 = E.i11;
      accumulate_cell(M.d2, etab, P, c, A);
    } while (mask);
    return;
  }
  if (!ingrid) return;
  // slow path: global centroid grid and record array
  const size_t base = (size_t)(iy + 1) * M.gw + (ix + 1);     // padded coords of (ix-1, iy-1)
  const float2 *grow = M.cent + base;
  mask = 0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) mask |= in_radius<INCL>(M.r2, xt, yt, grow[r * M.gw + q]) << (r * 3 + q);
  if (!mask) return;
  A.pairs += __builtin_popcount(mask);
  const PointTerms P = point_terms(x, y, xt, yt, cj, sj, ch, sh);
#pragma nounroll
  do {
    const int k = __builtin_ctz(mask);
    mask &= mask - 1;
    accumulate_cell(M.d2, etab, P, load_rec_global(M, base, k), A);
  } while (mask);
}

// Fixed-order sums over the workgroup: lanes by shuffle, waves through LDS in wave order.
// Totals are left in sred[nw*NV .. nw*NV+NV) (valid for every thread after the call).
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  return x;
}

template <int NV>
__device__ __forceinline__ void block_combine(double *sred, double *out) {
  const int nw = blockDim.x >> 6;
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += sred[w * NV + threadIdx.x];
    out[threadIdx.x] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ void block_reduce_acc(const Acc &A, double *sred, double *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *row = sred + wave * kAcc;
  double t;
  __syncthreads();   // sred may still be read from the previous round
  t = wave_sum(A.e);     if (lane == 0) row[0] = t;
  t = wave_sum(A.g0);    if (lane == 0) row[1] = t;
  t = wave_sum(A.g1);    if (lane == 0) row[2] = t;
  t = wave_sum(A.g2);    if (lane == 0) row[3] = t;
  t = wave_sum(A.hxx);   if (lane == 0) row[4] = t;
  t = wave_sum(A.hxy);   if (lane == 0) row[5] = t;
  t = wave_sum(A.hxt);   if (lane == 0) row[6] = t;
  t = wave_sum(A.hyy);   if (lane == 0) row[7] = t;
  t = wave_sum(A.hyt);   if (lane == 0) row[8] = t;
  t = wave_sum(A.htt);   if (lane == 0) row[9] = t;
  t = wave_sum((double)A.pairs); if (lane == 0) row[10] = t;
  block_combine<kAcc>(sred, out);
}

__device__ __forceinline__ void block_reduce2(double a, double b, double *sred, double *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double t;
  __syncthreads();
  t = wave_sum(a); if (lane == 0) sred[wave * 2 + 0] = t;
  t = wave_sum(b); if (lane == 0) sred[wave * 2 + 1] = t;
  block_combine<2>(sred, out);
}

// ------------------------------------------------------------------------------------------
// a6: Newton step + More-Thuente line search as a resumable state machine
// ------------------------------------------------------------------------------------------

// One Jacobi rotation annihilating a_pq of a symmetric 3x3 kept in scalars; r is the third index.
__device__ __forceinline__ void jacobi_rot(double &app, double &aqq, double &apq, double &arp, double &arq,
                                           double &v0p, double &v0q, double &v1p, double &v1q,
                                           double &v2p, double &v2q) {
  if (apq == 0.0) return;
  const double theta = (aqq - app) / (2.0 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  const double app0 = app, aqq0 = aqq;
  app = app0 - t * apq; aqq = aqq0 + t * apq;
  const double arp0 = arp, arq0 = arq;
  arp = c * arp0 - s * arq0; arq = s * arp0 + c * arq0;
  apq = 0.0;
  double a, b;
  a = v0p; b = v0q; v0p = c * a - s * b; v0q = s * a + c * b;
  a = v1p; b = v1q; v1p = c * a - s * b; v1q = s * a + c * b;
  a = v2p; b = v2q; v2p = c * a - s * b; v2q = s * a + c * b;
}

// Symmetric 3x3 pseudo-inverse solve (cyclic Jacobi, all state in registers); stands in for
// JacobiSVD<6x6>::solve on the block-diagonal 6x6 (SURVEY.md 8a note).  Hs = xx xy xt yy yt tt.
__device__ __forceinline__ void solve3(const double Hs[6], double b0, double b1, double b2,
                                       double &x0, double &x1, double &x2) {
  double a00 = Hs[0], a01 = Hs[1], a02 = Hs[2], a11 = Hs[3], a12 = Hs[4], a22 = Hs[5];
  if (a00 != a00 || a01 != a01 || a02 != a02 || a11 != a11 || a12 != a12 || a22 != a22) {
    x0 = x1 = x2 = NAN; return;
  }
  {
    // well-conditioned case: adjugate / determinant
    const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const double det = a00 * c00 + a01 * c01 + a02 * c02;
    const double sc = fmax(fmax(fabs(a00), fabs(a11)), fmax(fabs(a22), fmax(fabs(a01), fmax(fabs(a02), fabs(a12)))));
    if (fabs(det) > 1e-9 * sc * sc * sc && fabs(det) <= DBL_MAX) {
      x0 = (c00 * b0 + c01 * b1 + c02 * b2) / det;
      x1 = (c01 * b0 + c11 * b1 + c12 * b2) / det;
      x2 = (c02 * b0 + c12 * b1 + c22 * b2) / det;
      return;
    }
  }
  // near-singular Hessian: pseudo-inverse through the eigen-decomposition
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(a01) + fabs(a02) + fabs(a12);
    if (off == 0.0) break;
    jacobi_rot(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21);   // (p,q) = (0,1), r = 2
    jacobi_rot(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22);   // (0,2), r = 1
    jacobi_rot(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22);   // (1,2), r = 0
  }
  const double lmax = fmax(fabs(a00), fmax(fabs(a11), fabs(a22)));
  const double thr = lmax * (6.0 * DBL_EPSILON);
  x0 = x1 = x2 = 0.0;
  if (fabs(a00) > thr && !(fabs(a00) < DBL_MIN)) {
    const double pr = (v00 * b0 + v10 * b1 + v20 * b2) / a00;
    x0 += v00 * pr; x1 += v10 * pr; x2 += v20 * pr;
  }
  if (fabs(a11) > thr && !(fabs(a11) < DBL_MIN)) {
    const double pr = (v01 * b0 + v11 * b1 + v21 * b2) / a11;
    x0 += v01 * pr; x1 += v11 * pr; x2 += v21 * pr;
  }
  if (fabs(a22) > thr && !(fabs(a22) < DBL_MIN)) {
    const double pr = (v02 * b0 + v12 * b1 + v22 * b2) / a22;
    x0 += v02 * pr; x1 += v12 * pr; x2 += v22 * pr;
  }
}

// More-Thuente trial value, cases 1-4 (Sun & Yuan 2.4.2 / 2.4.5 / 2.4.52 / 2.4.56).
__device__ __noinline__ double mt_trial(double a_l, double f_l, double g_l, double a_u, double f_u,
                                        double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    if (fabs(a_c - a_l) < fabs(a_q - a_l)) return a_c;
    return 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    if (fabs(a_c - a_t) >= fabs(a_s - a_t)) return a_c;
    return a_s;
  } else if (fabs(g_t) <= fabs(g_l)) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double a_n = (fabs(a_c - a_t) < fabs(a_s - a_t)) ? a_c : a_s;
    double lim = a_t + 0.66 * (a_u - a_t);
    if (a_t > a_l) return (a_n < lim) ? a_n : lim;
    return (lim < a_n) ? a_n : lim;
  } else {
    double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
    double w = sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
  }
}

__device__ __forceinline__ int mt_update(AlignState &S, double a_t, double f_t, double g_t) {
  if (f_t > S.f_l) { S.a_u = a_t; S.f_u = f_t; S.g_u = g_t; return 0; }
  if (g_t * (S.a_l - a_t) > 0) { S.a_l = a_t; S.f_l = f_t; S.g_l = g_t; return 0; }
  if (g_t * (S.a_l - a_t) < 0) {
    S.a_u = S.a_l; S.f_u = S.f_l; S.g_u = S.g_l;
    S.a_l = a_t; S.f_l = f_t; S.g_l = g_t; return 0;
  }
  return 1;
}

__device__ __forceinline__ void set_trial(AlignState &S, const OptParams &P, bool refresh_h) {
  S.xt[0] = S.p[0] + S.dir[0] * S.a_t;
  S.xt[1] = S.p[1] + S.dir[1] * S.a_t;
  S.xt[2] = S.p[2] + S.dir[2] * S.a_t;
  S.T = tf_from_p(S.xt);
  angle_cs(P.snap_thresh, S.xt[2], S.cj, S.sj);
  if (refresh_h || !P.stale_h_ang) { S.ch = S.cj; S.sh = S.sj; }
}

// Start (or finish) outer iterations until a derivative pass is needed or the match is done.
__device__ __noinline__ void begin_outer(AlignState &S, const OptParams &P) {
  for (int guard = 0; guard < 1 << 20; ++guard) {   // every turn either asks for a pass or counts an iteration
    double dp0, dp1, dp2;
    solve3(S.H, -S.g[0], -S.g[1], -S.g[2], dp0, dp1, dp2);
    double nrm = sqrt(dp0 * dp0 + dp1 * dp1 + dp2 * dp2);
    if (nrm == 0 || nrm != nrm) { S.converged = (nrm == nrm); S.phase = PH_DONE; return; }
    S.dir[0] = dp0 / nrm; S.dir[1] = dp1 / nrm; S.dir[2] = dp2 / nrm;
    S.phi0 = -S.score;
    S.dphi0 = -(S.g[0] * S.dir[0] + S.g[1] * S.dir[1] + S.g[2] * S.dir[2]);
    double a = 0.0;
    bool need_eval = true;
    if (S.dphi0 >= 0) {
      if (S.dphi0 == 0) need_eval = false;
      else { S.dphi0 *= -1; S.dir[0] *= -1; S.dir[1] *= -1; S.dir[2] *= -1; }
    }
    if (need_eval) {
      S.step_iterations = 0;
      S.a_l = 0; S.a_u = 0;
      S.f_l = S.phi0 - S.phi0 - P.mt_mu * S.dphi0 * S.a_l;
      S.g_l = S.dphi0 - P.mt_mu * S.dphi0;
      S.f_u = S.phi0 - S.phi0 - P.mt_mu * S.dphi0 * S.a_u;
      S.g_u = S.dphi0 - P.mt_mu * S.dphi0;
      S.interval_converged = (P.step_size - P.trans_eps / 2) < 0;
      S.open_interval = 1;
      double a_t = nrm;
      a_t = (P.step_size < a_t) ? P.step_size : a_t;
      a_t = (a_t < P.trans_eps / 2) ? P.trans_eps / 2 : a_t;
      S.a_t = a_t;
      set_trial(S, P, true);
      S.phase = PH_LS_FIRST;
      return;
    }
    // zero directional derivative: step length 0, parameters unchanged
    int over = P.conv_ge ? (S.iters >= P.max_iter) : (S.iters > P.max_iter);
    bool conv = over || (S.iters && (fabs(a) < P.trans_eps));
    S.iters++;
    if (conv) { S.converged = 1; S.phase = PH_DONE; return; }
  }
}

// Consume one derivative pass (score, gradient, Hessian at the current trial transform).
__device__ __noinline__ void advance(AlignState &S, const OptParams &P, const MapView &M,
                                     const double tot[kAcc], double *trace, int trace_cap,
                                     int *trace_rows) {
  const double w = M.d1 * M.d2;
  S.score = -M.d1 * tot[0];
  S.g[0] = w * tot[1]; S.g[1] = w * tot[2]; S.g[2] = w * tot[3];
  S.H[0] = w * tot[4]; S.H[1] = w * tot[5]; S.H[2] = w * tot[6];
  S.H[3] = w * tot[7]; S.H[4] = w * tot[8]; S.H[5] = w * tot[9];
  S.pairs += tot[10];
  S.evals++; S.ref_evals++;
  if (trace) {
    int row = *trace_rows;
    if (row < trace_cap) {
      double *t = trace + 8 * (size_t)row;
      const double *pp = (S.phase == PH_INIT) ? S.p : S.xt;
      t[0] = (S.phase == PH_INIT) ? 0.0 : S.a_t; t[1] = S.score;
      t[2] = S.g[0]; t[3] = S.g[1]; t[4] = S.g[2]; t[5] = pp[0]; t[6] = pp[1]; t[7] = pp[2];
    }
    *trace_rows = row + 1;
  }
  if (S.phase == PH_INIT) { begin_outer(S, P); return; }

  const double mu = P.mt_mu, nu = P.mt_nu;
  double phi_t = -S.score;
  double d_phi_t = -(S.g[0] * S.dir[0] + S.g[1] * S.dir[1] + S.g[2] * S.dir[2]);
  double psi_t = phi_t - S.phi0 - mu * S.dphi0 * S.a_t;
  double d_psi_t = d_phi_t - mu * S.dphi0;
  if (S.phase == PH_LS_INNER) {
    if (S.open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      S.open_interval = 0;
      S.f_l = S.f_l + S.phi0 - mu * S.dphi0 * S.a_l; S.g_l = S.g_l + mu * S.dphi0;
      S.f_u = S.f_u + S.phi0 - mu * S.dphi0 * S.a_u; S.g_u = S.g_u + mu * S.dphi0;
    }
    if (S.open_interval) S.interval_converged = mt_update(S, S.a_t, psi_t, d_psi_t);
    else                 S.interval_converged = mt_update(S, S.a_t, phi_t, d_phi_t);
    S.step_iterations++;
  }
  bool more = !S.interval_converged && S.step_iterations < P.mt_max_iter &&
              !(psi_t <= 0 && d_phi_t <= -nu * S.dphi0);
  if (more) {
    double a_t;
    if (S.open_interval) a_t = mt_trial(S.a_l, S.f_l, S.g_l, S.a_u, S.f_u, S.g_u, S.a_t, psi_t, d_psi_t);
    else                 a_t = mt_trial(S.a_l, S.f_l, S.g_l, S.a_u, S.f_u, S.g_u, S.a_t, phi_t, d_phi_t);
    a_t = (P.step_size < a_t) ? P.step_size : a_t;
    a_t = (a_t < P.trans_eps / 2) ? P.trans_eps / 2 : a_t;
    S.a_t = a_t;
    set_trial(S, P, false);
    S.phase = PH_LS_INNER;
    return;
  }
  // line search done.  The reference now runs a Hessian-only pass when the inner loop ran;
  // the Hessian of the last pass (same cloud, same angle terms) is that Hessian already.
  if (S.step_iterations) S.ref_evals++;
  const double a = S.a_t;
  S.p[0] += S.dir[0] * a; S.p[1] += S.dir[1] * a; S.p[2] += S.dir[2] * a;
  int over = P.conv_ge ? (S.iters >= P.max_iter) : (S.iters > P.max_iter);
  bool conv = over || (S.iters && (fabs(a) < P.trans_eps));
  S.iters++;
  if (conv) { S.converged = 1; S.phase = PH_DONE; return; }
  begin_outer(S, P);
}

__device__ __noinline__ void init_state(AlignState &S, const OptParams &P, const double init[3],
                                        double n_points) {
  S.iters = 0; S.evals = 0; S.ref_evals = 0; S.converged = 0; S.step_iterations = 0;
  S.open_interval = 1; S.interval_converged = 0; S.pairs = 0.0; S.n_points = n_points;
  double pi[3] = {init[0], init[1], init[2]};
  S.T = tf_from_p(pi);     // init_guess = Translation3f * AngleAxisf (src/PoseEstimator.cpp:22-24)
  // p0 = (translation, eulerAngles(0,1,2)) of the float matrix: (-0, 0, atan2f(s, c))
  S.p[0] = (double)S.T.tx; S.p[1] = (double)S.T.ty;
  S.p[2] = (double)(float)atan2((double)S.T.s, (double)S.T.c);
  S.xt[0] = S.p[0]; S.xt[1] = S.p[1]; S.xt[2] = S.p[2];
  S.dir[0] = S.dir[1] = S.dir[2] = 0.0; S.a_t = 0.0;
  angle_cs(P.snap_thresh, S.p[2], S.cj, S.sj);
  S.ch = S.cj; S.sh = S.sj;
  S.score = 0.0;
  S.phase = PH_INIT;
}

// a9: src/PoseEstimator.cpp:31-35 on the float32 entries; asinf/acosf modelled as correctly rounded.
__device__ __forceinline__ double yaw_from_T(float T00, float T10) {
  if (T00 > 0 && T10 > 0) return (double)(float)asin((double)T10);
  if (T00 > 0 && T10 < 0) return (double)(float)asin((double)T10);
  if (T00 < 0 && T10 > 0) return (double)(float)acos((double)T00);
  return (double)(float)acos((double)T00) * (-1.0);
}

// ------------------------------------------------------------------------------------------
// a7: nearest raw map point, exact, no range cut: home voxel, then the ring-1 voxels that can
// still hold a closer point (box-distance pruning), then whole rings while the best distance
// exceeds the ring bound.
// ------------------------------------------------------------------------------------------
// The cost of this search is the number of (lane, cache line) look-ups of its divergent loads --
// the CU's vector L1 serves about one line per clock -- so everything is fetched as wide as the
// layout allows: a bucket's points two per 16-byte load, and the offsets of up to three
// neighbouring voxels of a row in one 16-byte load (pt_start carries 4 readable ints before its
// first entry and 3 after its last one).
struct __attribute__((packed, aligned(4))) I4u { int x, y, z, w; };
struct __attribute__((packed, aligned(4))) I2u { int x, y; };
__device__ __forceinline__ I4u ld_i4u(const int *p) { I4u v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ I2u ld_i2u(const int *p) { I2u v; __builtin_memcpy(&v, p, 8); return v; }

__device__ __forceinline__ float sq_dist(float qx, float qy, float px, float py) {
  const float ex = qx - px, ey = qy - py;
  return ex * ex + ey * ey;
}

// min over the points pts[s .. se) of the float32 squared distance to (qx, qy)
__device__ __forceinline__ float scan_bucket(const float2 *__restrict__ pts, int s, int se, float qx,
                                             float qy, float best) {
  if (s >= se) return best;
  if (s & 1) { const float2 p = pts[s]; best = fminf(best, sq_dist(qx, qy, p.x, p.y)); ++s; }
  const float4 *__restrict__ p4 = reinterpret_cast<const float4 *>(pts + s);   // 16-byte aligned
  const int npair = (se - s) >> 1;
  int i = 0;
  for (; i + 2 <= npair; i += 2) {           // four points, two loads in flight
    const float4 a = p4[i], b = p4[i + 1];
    const float d0 = sq_dist(qx, qy, a.x, a.y), d1 = sq_dist(qx, qy, a.z, a.w);
    const float d2 = sq_dist(qx, qy, b.x, b.y), d3 = sq_dist(qx, qy, b.z, b.w);
    best = fminf(best, fminf(fminf(d0, d1), fminf(d2, d3)));
  }
  if (i < npair) {
    const float4 a = p4[i];
    best = fminf(best, fminf(sq_dist(qx, qy, a.x, a.y), sq_dist(qx, qy, a.z, a.w)));
  }
  if ((se - s) & 1) { const float2 p = pts[se - 1]; best = fminf(best, sq_dist(qx, qy, p.x, p.y)); }
  return best;
}

__device__ __forceinline__ float nearest_sq(const MapView &M, float qx, float qy) {
  const int cx0 = (int)floorf(qx * M.inv_leaf) - M.min_bx, cy0 = (int)floorf(qy * M.inv_leaf) - M.min_by;
  const int cx = cx0 < 0 ? 0 : (cx0 >= M.div_x ? M.div_x - 1 : cx0);
  const int cy = cy0 < 0 ? 0 : (cy0 >= M.div_y ? M.div_y - 1 : cy0);
  const bool inside = (cx == cx0) && (cy == cy0);
  const int *__restrict__ ps = M.pt_start;
  const size_t gh = (size_t)cy * M.div_x + cx;
  // offsets of (cx-1, cx, cx+1) of the home row in one load: [left, home) [home, right) [right, end)
  const I4u h = ld_i4u(ps + gh - 1);
  float best = scan_bucket(M.pts, h.y, h.z, qx, qy, INFINITY);
  // distances from the query to the four walls of its voxel, shrunk by 1e-3 leaf so that a point
  // the float32 voxel rounding put on the other side of a wall is never pruned away
  const float L = M.leaf, slack = 1e-3f * L;
  const float fx = qx - (float)(cx + M.min_bx) * L, fy = qy - (float)(cy + M.min_by) * L;
  float wl = fmaxf(fx - slack, 0.f), wr = fmaxf(L - fx - slack, 0.f);
  float wd = fmaxf(fy - slack, 0.f), wu = fmaxf(L - fy - slack, 0.f);
  if (!inside) { wl = wr = wd = wu = 0.f; }            // clamped query: no pruning
  const float wmin = fminf(fminf(wl, wr), fminf(wd, wu));
  if (!(wmin * wmin < best)) return best;              // no other voxel can hold a closer point
  // ring 1: left / right voxel of the home row, then the rows below and above as one range each,
  // every voxel pruned by its box distance
  const bool has_l = cx > 0, has_r = cx + 1 < M.div_x;
  if (has_l && wl * wl < best) best = scan_bucket(M.pts, h.x, h.y, qx, qy, best);
  if (has_r && wr * wr < best) best = scan_bucket(M.pts, h.z, h.w, qx, qy, best);
#pragma unroll
  for (int dy = -1; dy <= 1; dy += 2) {
    const int yy = cy + dy;
    const float by = dy < 0 ? wd : wu;
    if (yy < 0 || yy >= M.div_y || !(by * by < best)) continue;
    const I4u o = ld_i4u(ps + (size_t)yy * M.div_x + cx - 1);
    const int sa = (has_l && wl * wl + by * by < best) ? o.x : o.y;
    const int sb = (has_r && wr * wr + by * by < best) ? o.w : o.z;
    best = scan_bucket(M.pts, sa, sb, qx, qy, best);
  }
  const double Ld = (double)L;
  const int rmax = M.div_x > M.div_y ? M.div_x : M.div_y;
  for (int r = 1; r <= rmax; ++r) {
    const double bound = (double)r * Ld * 0.999;        // unvisited points are farther than r*L
    if ((double)best <= bound * bound) break;
    const int R = r + 1;                                // ring R, pruned by box distances: in a row at
    const int y0 = cy - R, y1 = cy + R, x0 = cx - R, x1 = cx + R;   // distance by only the columns whose
    for (int yy = (y0 < 0 ? 0 : y0); yy <= y1 && yy < M.div_y; ++yy) {   // box is nearer than sqrt(best - by^2)
      const int dyc = yy - cy;
      const float by = dyc < 0 ? wd + (float)(-dyc - 1) * L : (dyc > 0 ? wu + (float)(dyc - 1) * L : 0.f);
      const float rem = best - by * by;
      if (!(rem > 0.f)) continue;
      const int hw = (int)fminf(sqrtf(rem) / L, 1.0e6f) + 1;   // columns farther than hw cannot matter
      const int *__restrict__ row = ps + (size_t)yy * M.div_x;
      if (yy == y0 || yy == y1) {
        int xa = x0 > cx - hw ? x0 : cx - hw, xb = x1 < cx + hw ? x1 : cx + hw;
        xa = xa < 0 ? 0 : xa; xb = xb >= M.div_x ? M.div_x - 1 : xb;
        if (xa <= xb) { const int sa = row[xa], sb = row[xb + 1]; best = scan_bucket(M.pts, sa, sb, qx, qy, best); }
      } else if (R <= hw) {
        I2u a = {0, 0}, b = {0, 0};                     // both voxels' offsets in flight together
        if (x0 >= 0) a = ld_i2u(row + x0);
        if (x1 < M.div_x) b = ld_i2u(row + x1);
        best = scan_bucket(M.pts, a.x, a.y, qx, qy, best);
        best = scan_bucket(M.pts, b.x, b.y, qx, qy, best);
      }
    }
  }
  return best;
}

// ------------------------------------------------------------------------------------------
// the match kernel
//
// One workgroup per CU.  Every scan has an OWNER workgroup that holds the optimiser state in LDS
// and runs the whole match on the device.  A derivative pass (and the fitness pass) is cut into
// kUnits units of points; each unit is reduced on its own and the pass total is the sum of the
// unit totals in a fixed order, so the result does not depend on who computed which unit.
// A workgroup whose own scans are finished becomes a HELPER: it attaches to an unfinished scan,
// stages that scan's window in its own LDS, registers, and from then on computes its static share
// of the units of every pass the owner opens.  Matches differ widely in the number of passes they
// need (mean ~12, max ~40 on the bench workload), so without helpers most of the chip idles
// behind the slowest scans.
//
// Inter-workgroup hand-off (cdna_hip_programming.md Guideline 16): every shared word (epoch word,
// arrival counter, ready counter, pose block, unit totals) is read and written ONLY with
// agent-scope relaxed atomics (sc1 loads / write-through stores), payload stores are drained
// (s_waitcnt vmcnt(0)) before the word that signals them, and the one bulk hand-off (the owner's
// ordered scan copy, marked-cell bitmap and window geometry) uses plain stores + agent release
// fence on the owner and an agent acquire fence on the helper.  No workgroup ever waits for a
// workgroup that is not running: a helper is only counted in after it has registered, at which
// point it does nothing but poll the scan's epoch word; helpers themselves only poll.
// Every spin is bounded by a watchdog that raises the abort word.
// ------------------------------------------------------------------------------------------
constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / 64;
constexpr int kSub = 4;                      // a lane's points are cut into kSub runs -> kSub units per wave
constexpr int kUnits = kWaves * kSub;        // units per pass
constexpr int kMaxHelpers = 15;              // helper workgroups per scan, hard limit (64 units: 4 each)
#ifndef NDT_IDLE_MAX
#define NDT_IDLE_MAX 800           // idle helper back-off: 4 us doubling up to 8 us (100 MHz ticks)
#endif
#ifndef NDT_HELPER_PENALTY
#define NDT_HELPER_PENALTY 12    // passes a scan must be ahead by before it gets one more helper than another
#endif
#ifndef NDT_BASE_HELPERS
#define NDT_BASE_HELPERS 7
#endif
constexpr int kBaseHelpers = NDT_BASE_HELPERS;              // ... while more scans are unfinished than workgroups / 8
constexpr unsigned kEpochDone = 0xFFFFFFFFu;
constexpr unsigned long long kWatchTicks = 400000000ull;   // ~4 s of the 100 MHz wall clock

typedef unsigned long long u64;
typedef unsigned int u32;

// Per-scan control block: four 128-byte lines, so that the words touched by different parties
// (epoch polls / arrivals / attach + ready counts / pose reads) never share a line.
//
// The epoch word describes one SEGMENT of a pass -- units [ubeg, uend) split over the owner and the
// first `h` registered helpers: participant k (0 = owner, k = helper rank + 1) computes the units
// ubeg + k + j*(h+1).  The assignment is static (no claim atomics: a same-address agent-scope
// read-modify-write costs ~0.1 us and 128 waves used to queue on it every pass); it is safe because a
// helper only counts once it has registered in `ready`, after which it does nothing but poll this word.
struct alignas(128) ScanCtl {
  u64 ticket;        // line 0: epoch << 32 | kind << 24 | h << 16 | uend << 8 | ubeg.  epoch 0: not open; kEpochDone: finished
  u64 pad0_[15];
  u32 arrive;        // line 1: units published by helpers in the open epoch (one add per helper workgroup)
  u32 pad1_[31];
  u32 helpers;       // line 2: helper workgroups attached; geometry published by the owner's release
  int region[6];
  u32 passes;        //         passes the owner has run so far (helpers go where most were needed)
  u32 ready;         //         helpers whose window is staged; rank = order of registration
  u32 phase;         //         1: the scan is in its fitness pass (a helper needs no window)
  u32 use_sorted;    //         1: passes read the scan from the sorted scratch copy
  u32 owner_wg;      //         workgroup that owns the scan (its scratch slot when every match uses scan 0)
  int pad2_[20];
  u64 pose[6];       // line 3: float32 transform (c|s, tx|ty) and the four fp64 angle terms
  u64 pad3_[10];
};
static_assert(sizeof(ScanCtl) == 512, "ScanCtl is four 128-byte lines");

struct WsHeader { u32 done; u32 abort; u32 next; u32 pad[29]; };   // next: scans handed out beyond the first gridDim.x
static_assert(sizeof(WsHeader) == 128, "WsHeader");

#define NDT_RLX __ATOMIC_RELAXED
#define NDT_AGENT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ u64 ld64(const u64 *p) { return __hip_atomic_load(p, NDT_RLX, NDT_AGENT); }
__device__ __forceinline__ u32 ld32(const u32 *p) { return __hip_atomic_load(p, NDT_RLX, NDT_AGENT); }
__device__ __forceinline__ void st64(u64 *p, u64 v) { __hip_atomic_store(p, v, NDT_RLX, NDT_AGENT); }
__device__ __forceinline__ void st32(u32 *p, u32 v) { __hip_atomic_store(p, v, NDT_RLX, NDT_AGENT); }
// reads through a memory-side read-modify-write: never served from a stale L2 line of this XCD
__device__ __forceinline__ u64 rd64_fresh(u64 *p) { return __hip_atomic_fetch_add(p, 0ull, NDT_RLX, NDT_AGENT); }
__device__ __forceinline__ u32 rd32_fresh(u32 *p) { return __hip_atomic_fetch_add(p, 0u, NDT_RLX, NDT_AGENT); }
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_down(v, o); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_down(v, o); v = t > v ? t : v; }
  return v;
}

// pose block of the pass being computed (LDS copy)
struct PassPose { Tf32 T; double cj, sj, ch, sh; int kind; };

struct Lds {
  AlignState S;
  PassPose PP;
  Region RG;
  int sbox[4];
  int swave[kWaves + 1];
  int sflag[4];
  double wpart[kUnits * 12];       // unit totals this workgroup computed in the open pass
  double wtmp[kWaves * 12];        // helper waves: the unit just computed, before it is published
  double tot[12];                  // pass totals
  unsigned long long own_mask;     // units of the open pass computed by this workgroup
  unsigned long long hpose[8];     // helper: pose block of the open epoch, staged by wave 0
  unsigned long long hword;        // helper: epoch word seen by wave 0
  int hrank;                       // helper: order of registration on its scan
  int jnext, stop;                 // units of the open segment handed out so far; close the segment
  unsigned diag[2];                // diagnostic: ticks of fill_window's first two phases
  double etab[64];
};

__device__ __forceinline__ Window window_of(const Region &R, const uint4 *pool) {
  Window W;
  W.R = R;
  W.slot = reinterpret_cast<const unsigned short *>(pool);
  W.ent = reinterpret_cast<const CellEntry *>(reinterpret_cast<const char *>(pool) +
                                              ((R.rw * R.rh * 2 + 15) / 16) * 16);
  return W;
}

// Owner: bounding box of the scan's voxel coordinates at the first pose -> window geometry.
template <bool SSE>
__device__ __forceinline__ void compute_region(const MapView &M, const Tf32 &T0, const float2 *__restrict__ scan,
                                               int n, Lds &L) {
  if (threadIdx.x == 0) { L.sbox[0] = INT_MAX; L.sbox[1] = INT_MAX; L.sbox[2] = INT_MIN; L.sbox[3] = INT_MIN; }
  __syncthreads();
  int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
  for (int i = threadIdx.x; i < n; i += kBlock) {
    const float2 pt = scan[i];
    float xt, yt;
    tf_apply_t<SSE>(T0, pt.x, pt.y, xt, yt);
    if (!finite2(xt, yt)) continue;
    const float fx = fminf(fmaxf(floorf(xt * M.inv_leaf), -1.0e9f), 1.0e9f);
    const float fy = fminf(fmaxf(floorf(yt * M.inv_leaf), -1.0e9f), 1.0e9f);
    const int ix = (int)fx - M.min_bx, iy = (int)fy - M.min_by;
    mnx = ix < mnx ? ix : mnx; mxx = ix > mxx ? ix : mxx;
    mny = iy < mny ? iy : mny; mxy = iy > mxy ? iy : mxy;
  }
  mnx = wave_min_i(mnx); mny = wave_min_i(mny); mxx = wave_max_i(mxx); mxy = wave_max_i(mxy);
  if ((threadIdx.x & 63) == 0 && mnx <= mxx) {
    atomicMin(&L.sbox[0], mnx); atomicMin(&L.sbox[1], mny); atomicMax(&L.sbox[2], mxx); atomicMax(&L.sbox[3], mxy);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Region r = {0, 0, 0, 0, 0, 0};
    if (L.sbox[0] <= L.sbox[2]) {
      // clip the bbox to the padded map grid, add the slack, then fit the slot-table budget around
      // the bbox centre
      long long x0 = (long long)L.sbox[0] - kRegionMargin, x1 = (long long)L.sbox[2] + kRegionMargin;
      long long y0 = (long long)L.sbox[1] - kRegionMargin, y1 = (long long)L.sbox[3] + kRegionMargin;
      x0 = x0 < -2 ? -2 : x0; y0 = y0 < -2 ? -2 : y0;
      x1 = x1 > M.div_x + 1 ? M.div_x + 1 : x1; y1 = y1 > M.div_y + 1 ? M.div_y + 1 : y1;
      long long w = x1 - x0 + 1, h = y1 - y0 + 1;
      if (w > 0 && h > 0) {
        if (w * h > kRegionCells) {
          long long w2 = w > 128 ? 128 : w;
          long long h2 = kRegionCells / w2; if (h2 > h) h2 = h;
          x0 += (w - w2) / 2; y0 += (h - h2) / 2; w = w2; h = h2;
        }
        r.x0 = (int)x0; r.y0 = (int)y0; r.rw = (int)w; r.rh = (int)h;
      }
    }
    const int slot_bytes = ((r.rw * r.rh * 2 + 15) / 16) * 16;
    int cap = (kPoolBytes - slot_bytes) / (int)sizeof(CellEntry) - 2;   // last two = sentinels
    r.cap = cap > 0xFFF0 ? 0xFFF0 : cap;
    L.RG = r;
  }
  __syncthreads();
}

// Owner and helpers: fill the slot table and the compact record table of window L.RG from the map
// and the marked-cell bitmap in L.wmap.  Slots are numbered in row-major order of the window, so
// the content depends only on the map, the geometry and the bitmap.  Slot values: < cap a resident
// record; cap = voxel outside the search set (centroid +inf); cap + 1 = occupied voxel without an
// LDS record (centroid -inf).  Sets L.RG.nspill = occupied voxels left without a record.
// Cells are walked 1024 at a time with consecutive lanes on consecutive cells (coalesced centroid
// and record reads); the row-major numbering comes from wave ballots kept in LDS.
__device__ __forceinline__ void fill_window(const MapView &M, Lds &L, uint4 *pool) {
  const Region r = L.RG;
  const unsigned *wmap = reinterpret_cast<const unsigned *>(L.wpart);
  unsigned short *slot = reinterpret_cast<unsigned short *>(pool);
  CellEntry *ent = reinterpret_cast<CellEntry *>(reinterpret_cast<char *>(pool) + ((r.rw * r.rh * 2 + 15) / 16) * 16);
  const u64 t_fill0 = wall_clock64();
  const int ncell = r.rw * r.rh;
  const int rounds = (ncell + kBlock - 1) / kBlock;          // <= kRegionCells / kBlock = 16
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 *keepw = reinterpret_cast<u64 *>(L.wpart) + 256;        // [rounds][kWaves] ballots (wmap uses the first 2 KiB)
  u64 *occw = keepw + 256;
  int *base = reinterpret_cast<int *>(L.wtmp);                // [256] exclusive prefix of the kept counts
  // which window cells are in the map's search set: one bit each from the map's occupancy words,
  // all rounds' loads in flight together, then one ballot per round
  unsigned ow[kRegionCells / kBlock];
  const int rw1 = max(r.rw, 1), step_y = kBlock / rw1, step_x = kBlock - step_y * rw1;   // one round further on
  int ly = (int)threadIdx.x / rw1, lx = (int)threadIdx.x - ly * rw1;
#pragma unroll
  for (int j = 0; j < kRegionCells / kBlock; ++j) {
    const int c = j * kBlock + threadIdx.x;
    ow[j] = 0u;
    if (j < rounds && c < ncell) {
      const int mx = r.x0 + lx, my = r.y0 + ly;
      if (mx >= 0 && mx < M.div_x && my >= 0 && my < M.div_y) {
        const size_t g = (size_t)my * M.div_x + mx;
        ow[j] = (M.occ[g >> 5] >> (g & 31)) & 1u;
      }
    }
    lx += step_x; ly += step_y;
    if (lx >= rw1) { lx -= rw1; ++ly; }
  }
#pragma unroll
  for (int j = 0; j < kRegionCells / kBlock; ++j) {
    const u64 ob = __ballot(ow[j] != 0u);
    if (lane == 0) occw[j * kWaves + wave] = ob;           // rounds past the window: zero
  }
  // marked cells dilated by two cells in x and y, on whole words: a voxel gets an LDS record when it
  // is in the search set and within two cells of a cell a scan point fell in.  (Rows are not word
  // aligned, so a mark in the first or last two columns of the window also reaches the end of the
  // neighbouring row: a few more records, nothing else.)
  unsigned *dx = reinterpret_cast<unsigned *>(keepw);       // 512 words, reused for the result
  constexpr int kWords = kRegionCells / 32;
  auto word_at = [&](const unsigned *a, int i) { return (i >= 0 && i < kWords) ? a[i] : 0u; };
  if (threadIdx.x < kWords) {
    const int i = threadIdx.x;
    const unsigned w = wmap[i], pv = word_at(wmap, i - 1), nx = word_at(wmap, i + 1);
    dx[i] = w | (w << 1) | (w << 2) | (w >> 1) | (w >> 2) | (pv >> 31) | (pv >> 30) | (nx << 31) | (nx << 30);
  }
  __syncthreads();
  unsigned kword = 0;
  if (threadIdx.x < kWords) {
    const int i = threadIdx.x;
    kword = dx[i];
#pragma unroll
    for (int m = 1; m <= 2; ++m) {
      const int sft = m * r.rw, q = sft >> 5, b = sft & 31;
      // bits moved towards higher cell numbers (from the row(s) above) and towards lower ones (below)
      kword |= (word_at(dx, i - q) << b) | (b ? (word_at(dx, i - q - 1) >> (32 - b)) : 0u);
      kword |= (word_at(dx, i + q) >> b) | (b ? (word_at(dx, i + q + 1) << (32 - b)) : 0u);
    }
    kword &= reinterpret_cast<const unsigned *>(occw)[i];
  }
  __syncthreads();
  if (threadIdx.x < kWords) dx[threadIdx.x] = kword;        // = keepw, two words per ballot word
  __syncthreads();
  if (threadIdx.x == 0) L.diag[0] = (unsigned)(wall_clock64() - t_fill0);
  // exclusive prefix of the kept counts over the rounds * kWaves ballot words (cell order)
  const int nword = rounds * kWaves;                           // <= 256
  if (threadIdx.x < 256) {
    const int mine = (int)threadIdx.x < nword ? __builtin_popcountll(keepw[threadIdx.x]) : 0;
    const int skip = (int)threadIdx.x < nword ? __builtin_popcountll(occw[threadIdx.x] & ~keepw[threadIdx.x]) : 0;
    int incl = mine, sk = skip;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o); if (lane >= o) incl += t;
      sk += __shfl_xor(sk, o);
    }
    base[threadIdx.x] = incl - mine;
    if (lane == 63) { L.swave[wave] = incl; L.sbox[wave] = sk; }
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    int add = 0;
    for (int w = 0; w < wave; ++w) add += L.swave[w];
    base[threadIdx.x] += add;
  }
  if (threadIdx.x == 0) {
    const int kept = L.swave[0] + L.swave[1] + L.swave[2] + L.swave[3];
    const int skipped = L.sbox[0] + L.sbox[1] + L.sbox[2] + L.sbox[3];
    CellEntry z; z.cent = make_float2(INFINITY, INFINITY); z.mx = z.my = z.i00 = z.i01 = z.i11 = 0.0;
    ent[r.cap] = z;                             // voxels outside the search set
    z.cent = make_float2(-INFINITY, -INFINITY);
    ent[r.cap + 1] = z;                         // occupied voxels without an LDS record
    L.RG.nspill = skipped + (kept > r.cap ? kept - r.cap : 0);
    L.diag[1] = (unsigned)(wall_clock64() - t_fill0);
  }
  __syncthreads();
  for (int j0 = 0; j0 < rounds; j0 += 4) {                   // four rounds' record loads in flight together
    int nx[4]; float2 cc[4]; double2 ra[4], rb[4]; double rc[4]; unsigned sl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u, c = j * kBlock + threadIdx.x;
      nx[u] = -1; sl[u] = (unsigned)r.cap;
      if (j < rounds && c < ncell) {
        const u64 kb = keepw[j * kWaves + wave], ob = occw[j * kWaves + wave];
        if ((ob >> lane) & 1ull) {
          sl[u] = (unsigned)r.cap + 1u;
          if ((kb >> lane) & 1ull) {
            const int next = base[j * kWaves + wave] + __builtin_popcountll(kb & ((1ull << lane) - 1ull));
            if (next < r.cap) {
              const int ly = c / r.rw, lx = c - ly * r.rw;
              const size_t pg = (size_t)(r.y0 + ly + 2) * M.gw + (r.x0 + lx + 2);
              const double *rec = M.rec + pg * 8;
              cc[u] = M.cent[pg];
              ra[u] = *reinterpret_cast<const double2 *>(rec); rb[u] = *reinterpret_cast<const double2 *>(rec + 2); rc[u] = rec[4];
              nx[u] = next; sl[u] = (unsigned)next;
            }
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u, c = j * kBlock + threadIdx.x;
      if (nx[u] >= 0) {
        CellEntry E; E.cent = cc[u]; E.mx = ra[u].x; E.my = ra[u].y; E.i00 = rb[u].x; E.i01 = rb[u].y; E.i11 = rc[u];
        ent[nx[u]] = E;
      }
      if (j < rounds && c < ncell) slot[c] = (unsigned short)sl[u];
    }
  }
  __syncthreads();
}

// Owner: spatial order of the scan.  The points are sorted by the window cell they fall in at the
// first pose (row-major cell order, input order kept inside a cell) and written to the scratch copy
// every pass reads.  The 64 lanes of a wave then always work on neighbouring points -- a rigid
// transform keeps neighbours together, so this holds at every later pose too -- which means: equal
// in-radius voxel counts (the pair loop runs max-over-lanes times), LDS probes that hit the same few
// slots and records (broadcast instead of bank conflicts), and in the fitness pass bucket loads that
// share cache lines.  The cell histogram also yields the marked-cell bitmap (L.wmap) that
// fill_window and the helpers use.  Uses the LDS pool as scratch (before the window is staged).
// Returns false (bitmap still produced, scratch copy not written) when the scan is too large for it.
constexpr int kSortMax = 20000;             // LDS room for one word per point; point numbers < 2^15
template <bool SSE>
__device__ __forceinline__ bool sort_points(const MapView &M, const Tf32 &T0, const float2 *__restrict__ scan,
                                            int n, Lds &L, uint4 *pool, float2 *__restrict__ sp) {
  const Region r = L.RG;
  const int ncell = r.rw * r.rh;
  unsigned *wmap = reinterpret_cast<unsigned *>(L.wpart);
  unsigned *hist = reinterpret_cast<unsigned *>(pool);                 // ncell + 1 counters (last: outside the window)
  unsigned *idx = hist + ((ncell + 1 + 3) & ~3);
  for (int i = threadIdx.x; i <= ncell; i += kBlock) hist[i] = 0u;
  for (int i = threadIdx.x; i < kRegionCells / 32; i += kBlock) wmap[i] = 0u;
  __syncthreads();
  auto key_of = [&](float2 pt) {
    float xt, yt;
    tf_apply_t<SSE>(T0, pt.x, pt.y, xt, yt);
    if (!finite2(xt, yt)) return ncell;
    const float fx = fminf(fmaxf(floorf(xt * M.inv_leaf), -1.0e9f), 1.0e9f);
    const float fy = fminf(fmaxf(floorf(yt * M.inv_leaf), -1.0e9f), 1.0e9f);
    const int lx = (int)fx - M.min_bx - r.x0, ly = (int)fy - M.min_by - r.y0;
    if (lx < 0 || lx >= r.rw || ly < 0 || ly >= r.rh) return ncell;
    return ly * r.rw + lx;
  };
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * kBlock) {      // four loads in flight
    float2 pt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pt[u] = scan[min(i0 + u * kBlock, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i0 + u * kBlock < n) atomicAdd(&hist[key_of(pt[u])], 1u);
  }
  __syncthreads();
  // marked-cell bitmap
  for (int w = threadIdx.x; w < (ncell + 31) / 32; w += kBlock) {
    unsigned bits = 0;
    const int c0 = w * 32, c1 = min(c0 + 32, ncell);
    for (int c = c0; c < c1; ++c) bits |= (hist[c] != 0u ? 1u : 0u) << (c - c0);
    wmap[w] = bits;
  }
  const bool do_sort = sp != nullptr && n <= kSortMax;
  if (!do_sort) { __syncthreads(); return false; }
  // exclusive scan of the ncell + 1 counters
  const int per = (ncell + 1 + kBlock - 1) / kBlock;
  const int c0 = min((int)threadIdx.x * per, ncell + 1), c1 = min(c0 + per, ncell + 1);
  unsigned mine = 0;
  for (int c = c0; c < c1; ++c) mine += hist[c];
  unsigned incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if ((int)(threadIdx.x & 63) >= o) incl += t; }
  if ((threadIdx.x & 63) == 63) L.swave[threadIdx.x >> 6] = (int)incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < kWaves; ++w) { const int t = L.swave[w]; L.swave[w] = run; run += t; }
  }
  __syncthreads();
  {
    unsigned run = (unsigned)L.swave[threadIdx.x >> 6] + incl - mine;
    for (int c = c0; c < c1; ++c) { const unsigned t = hist[c]; hist[c] = run; run += t; }
  }
  __syncthreads();
  // scatter (cell, point number) packed in one word; afterwards hist[c] = end of cell c
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * kBlock) {
    float2 pt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pt[u] = scan[min(i0 + u * kBlock, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u * kBlock >= n) break;
      const int key = key_of(pt[u]);
      idx[atomicAdd(&hist[key], 1u)] = ((unsigned)key << 15) | (unsigned)(i0 + u * kBlock);
    }
  }
  __syncthreads();
  // input order inside a cell (the atomics above arrive in any order): every entry finds its rank among
  // the entries of its cell -- neighbouring lanes read the same short segment -- and its point goes
  // straight to that place of the scratch copy
  for (int p0 = threadIdx.x; p0 < n; p0 += 4 * kBlock) {
    int dstpos[4]; float2 pt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = p0 + u * kBlock;
      dstpos[u] = -1;
      if (pp < n) {
        const unsigned v = idx[pp];
        const int key = (int)(v >> 15);
        const int s0 = key ? (int)hist[key - 1] : 0, s1 = (int)hist[key];
        int rank = 0;
        for (int a = s0; a < s1; ++a) rank += idx[a] < v ? 1 : 0;
        dstpos[u] = s0 + rank;
        pt[u] = scan[v & 0x7FFFu];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (dstpos[u] >= 0) sp[dstpos[u]] = pt[u];
  }
  __syncthreads();
  return true;
}

// Sum of 12 per-lane values over the 64 lanes of a wave in a fixed order, 86 instructions instead
// of 12 x 18: a butterfly in which every exchange also halves the number of values a lane carries
// (12 -> 6 -> 3 -> 2 -> 1), so only 24 cross-lane moves are needed.  The total of value j ends in
// the lanes whose bits select j; those lanes store it to dst[j] (LDS).
__device__ __forceinline__ void wave_reduce12(const double (&a)[12], int lane, double *__restrict__ dst) {
  const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
  double k[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {                       // keep values 0..5 (b5 = 0) or 6..11 (b5 = 1)
    const double keep = b5 ? a[i + 6] : a[i], send = b5 ? a[i] : a[i + 6];
    k[i] = keep + __shfl_xor(send, 32);
  }
  double m[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {                       // keep 0..2 or 3..5 of those
    const double keep = b4 ? k[i + 3] : k[i], send = b4 ? k[i] : k[i + 3];
    m[i] = keep + __shfl_xor(send, 16);
  }
  const double p0 = (b3 ? m[1] : m[0]) + __shfl_xor(b3 ? m[0] : m[1], 8);   // value 0 or 1 of the triple
  const double p1 = m[2] + __shfl_xor(m[2], 8);                            // value 2
  double r = (b2 ? p1 : p0) + __shfl_xor(b2 ? p0 : p1, 4);
  r += __shfl_xor(r, 2);
  r += __shfl_xor(r, 1);
  const int idx = (b5 ? 6 : 0) + (b4 ? 3 : 0) + (b2 ? 2 : (b3 ? 1 : 0));
  if ((lane & 3) == 0 && !(b2 && b3)) dst[idx] = r;
}

// Units of a pass: unit u = (virtual wave w = u % kWaves, run q = u / kWaves) is the lane set
// {w*64 .. w*64+63} walking the q-th run of its points i = w*64 + lane + k*kBlock,
// k in [q*run, (q+1)*run), of the (ordered) scan.  Any physical wave of any workgroup can compute
// a unit; its sums are reduced over the 64 lanes in a fixed order, and a pass total is the sum of
// the kUnits unit totals in unit order -- the same arithmetic whether the owner computed all units
// itself or helpers computed some.
// This routine computes the consecutive runs [q0, q1) of virtual wave w in ONE walk over k (the
// point prefetch keeps running across run boundaries) and leaves the 12 sums of run q at
// dst[(q - q0) * dst_stride .. +12) (LDS).
// wave-uniform values read from LDS land in VGPRs; these move them to SGPRs (the pass loop is short of VGPRs)
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uniform_d(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}

template <bool SSE, bool INCL>
__device__ __forceinline__ void unit_sums(const MapView &M, const Window &W, const double *__restrict__ etab,
                                          const PassPose &pp_in, const float2 *__restrict__ pts, int n, int w,
                                          int q0, int q1, double *__restrict__ dst, int dst_stride) {
  PassPose pp;
  pp.T.c = uniform_f(pp_in.T.c); pp.T.s = uniform_f(pp_in.T.s); pp.T.tx = uniform_f(pp_in.T.tx); pp.T.ty = uniform_f(pp_in.T.ty);
  pp.cj = uniform_d(pp_in.cj); pp.sj = uniform_d(pp_in.sj); pp.ch = uniform_d(pp_in.ch); pp.sh = uniform_d(pp_in.sh);
  pp.kind = __builtin_amdgcn_readfirstlane(pp_in.kind);
  const int lane = threadIdx.x & 63, last = n - 1;
  const int per_lane = (n + kBlock - 1) / kBlock;          // points of the longest lane
  const int run = (per_lane + kSub - 1) / kSub;
  const int kbeg = min(per_lane, q0 * run), kend = min(per_lane, q1 * run);
  const int base = w * 64 + lane;
  if (pp.kind == 0) {
    Acc A = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0u};
    float2 p0 = pts[min(base + kbeg * kBlock, last)], p1 = pts[min(base + (kbeg + 1) * kBlock, last)];
    int q = q0, kb = min(per_lane, (q0 + 1) * run);        // end of the current run
#pragma nounroll
    for (int k = kbeg; k < kend; ++k) {
      const float2 p2 = pts[min(base + (k + 2) * kBlock, last)];
      if (base + k * kBlock >= n) p0.x = NAN;              // past the end: contributes nothing
      eval_point<SSE, INCL>(M, W, etab, pp.T, p0.x, p0.y, pp.cj, pp.sj, pp.ch, pp.sh, A);
      p0 = p1; p1 = p2;
      if (k + 1 == kb) {                                   // run q complete (uniform across the wave)
        const double a[12] = {A.e, A.g0, A.g1, A.g2, A.hxx, A.hxy, A.hxt, A.hyy, A.hyt, A.htt, (double)A.pairs, 0.0};
        wave_reduce12(a, lane, dst + (q - q0) * dst_stride);
        A = Acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0u};
        ++q; kb = min(per_lane, (q + 1) * run);
      }
    }
    for (; q < q1; ++q) {                                  // empty runs (short scans)
      if (lane < 12) dst[(q - q0) * dst_stride + lane] = 0.0;
    }
  } else {
    for (int q = q0; q < q1; ++q) {
      const int k0 = min(per_lane, q * run), k1 = min(per_lane, (q + 1) * run);
      double fsum = 0.0, fcnt = 0.0;
      for (int k = k0; k < k1; ++k) {
        const int i = base + k * kBlock;
        if (i >= n) break;
        const float2 pt = pts[i];
        float qx, qy;
        tf_apply_t<SSE>(pp.T, pt.x, pt.y, qx, qy);
        if (!finite2(qx, qy)) continue;
        const float best = nearest_sq(M, qx, qy);
        if (best < INFINITY) { fsum += (double)best; fcnt += 1.0; }
      }
      fsum = wave_sum(fsum); fcnt = wave_sum(fcnt);
      if (lane == 0) { dst[(q - q0) * dst_stride] = fsum; dst[(q - q0) * dst_stride + 1] = fcnt; }
    }
  }
}

// Bound on every spin: looked at once per 64 polls (the abort word is one line shared by the chip).
__device__ __forceinline__ bool watchdog(WsHeader *hdr, u64 t_start, unsigned &polls) {
  if ((++polls & 63u) != 0u) return false;
  if (ld32(&hdr->abort)) return true;
  if (wall_clock64() - t_start > kWatchTicks) { st32(&hdr->abort, 1u); return true; }
  return false;
}

__device__ __forceinline__ u64 wave_bcast64(u64 v) {   // lane 0's value to the whole wave
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}

template <bool SSE, bool INCL>
__global__ void __launch_bounds__(kBlock)
ndt_align_kernel(MapView M, OptParams P, const float *__restrict__ scans,
                 const unsigned long long *__restrict__ offsets, int B, int shared_scan,
                 const double *__restrict__ inits, ndt_result *__restrict__ results,
                 double *__restrict__ trace, int trace_cap, int *__restrict__ trace_rows,
                 float2 *__restrict__ sorted /* scratch, same offsets as scans; may be null */,
                 unsigned char *__restrict__ ws /* WsHeader, ScanCtl[B], unit totals[B][kUnits][12], marked-cell bitmaps[B][kRegionCells/32] */,
                 int allow_helpers /* 0: none; else max helper workgroups per scan */,
                 unsigned long long *__restrict__ prof /* diagnostic: 8 words per scan */) {
  __shared__ Lds L;
  __shared__ uint4 pool[kPoolBytes / 16];
  WsHeader *hdr = reinterpret_cast<WsHeader *>(ws);
  ScanCtl *ctl = reinterpret_cast<ScanCtl *>(ws + sizeof(WsHeader));
  u64 *utot = reinterpret_cast<u64 *>(ws + sizeof(WsHeader) + (size_t)B * sizeof(ScanCtl));
  unsigned *wantmap = reinterpret_cast<unsigned *>(ws + sizeof(WsHeader) + (size_t)B * sizeof(ScanCtl) +
                                                  (size_t)B * kUnits * 12 * sizeof(double));
  const u64 t_start = wall_clock64();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 64) L.etab[threadIdx.x] = c_exp2_tab[threadIdx.x];
  bool aborted = false;

  // =========================== owner of scans blockIdx.x, + gridDim.x, ... ===========================
  // scans are taken from a queue: the first gridDim.x by workgroup number, the rest in the order
  // workgroups become free (results do not depend on who owns which scan)
  for (int b = blockIdx.x; b < B && !aborted;) {
    const u64 o0 = shared_scan ? offsets[0] : offsets[b];
    const u64 o1 = shared_scan ? offsets[1] : offsets[b + 1];
    const int n = (int)(o1 - o0);
    const float2 *scan = reinterpret_cast<const float2 *>(scans) + o0;
    double *tr = trace ? trace + (size_t)b * trace_cap * 8 : nullptr;
    ScanCtl *C = ctl + b;
    u64 *mytot = utot + (size_t)b * kUnits * 12;
    __syncthreads();
    if (threadIdx.x == 0) {
      init_state(L.S, P, inits + 3 * (size_t)b, (double)n);
      if (trace_rows) trace_rows[b] = 0;
      if (n <= 0) { L.S.phase = PH_DONE; L.S.converged = 0; }
    }
    __syncthreads();
    const float2 *pts = scan;
    if (n > 0) {
      const u64 q0 = wall_clock64();
      compute_region<SSE>(M, L.S.T, scan, n, L);
      const u64 q1 = wall_clock64();
      // scratch copy: at the scan's own offsets, or (every match uses scan 0) one slot per workgroup
      float2 *sp = sorted ? (shared_scan ? sorted + (size_t)blockIdx.x * (size_t)n : sorted + o0) : nullptr;
      if (sort_points<SSE>(M, L.S.T, scan, n, L, pool, sp)) pts = sp;
      const u64 q2 = wall_clock64();
      if (allow_helpers) {                         // helpers rebuild the same window from this bitmap
        const unsigned *wmap = reinterpret_cast<const unsigned *>(L.wpart);
        unsigned *gw = wantmap + (size_t)b * (kRegionCells / 32);
        for (int i = threadIdx.x; i < kRegionCells / 32; i += kBlock) gw[i] = wmap[i];
      }
      if (allow_helpers) {
        // publish geometry + marked cells + ordered copy before staging the own window, so that idle
        // workgroups stage theirs meanwhile: plain stores, drained by every wave, then one agent release
        if (threadIdx.x == 0) {
          const Region r = L.RG;
          C->region[0] = r.x0; C->region[1] = r.y0; C->region[2] = r.rw; C->region[3] = r.rh;
          C->region[4] = r.cap; C->region[5] = r.nspill;
          C->use_sorted = (pts != scan) ? 1u : 0u;
          C->owner_wg = blockIdx.x;
        }
        drain_vmem();
        __syncthreads();
        if (threadIdx.x == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          drain_vmem();
          st64(&C->ticket, (u64)1 << 32);                     // epoch 1: open for joining, nothing to compute (h = 0)
        }
      }
      fill_window(M, L, pool);
      const u64 q3 = wall_clock64();
      if (prof && threadIdx.x == 0) {
        const u64 q4 = wall_clock64();
        prof[8 * (size_t)B + 8 * (size_t)b + 6] = ((q1 - q0) << 32) | ((q2 - q1) & 0xFFFFFFFFull);
        prof[8 * (size_t)B + 8 * (size_t)b + 7] = ((q3 - q2) << 32) | ((u64)(L.diag[0] & 0xFFFFu) << 16) | (u64)(L.diag[1] & 0xFFFFu);
      }
    }
    const Window W = window_of(L.RG, pool);
    if (threadIdx.x == 0) L.sflag[1] = 0;            // registered helpers (refreshed during every advance)
    unsigned epoch = 1;
    u64 t_eval = 0, t_adv = 0, tt0 = 0, tt1 = 0, t_wait = 0, t_first_shared = 0, t_fit = 0;
    u64 ts1 = 0, ts2 = 0, ts3 = 0, a_pro = 0, a_own = 0, a_wait = 0, a_comb = 0, a_adv = 0, a_n = 0;   // shared derivative passes (diagnostic)
    const u64 t_scan0 = wall_clock64() - t_start;
    unsigned n_shared = 0, n_helped = 0;
    bool fitness_done = false;
    // ---- passes: derivative passes until the optimiser stops, then one fitness pass ----
    while (n > 0 && !fitness_done) {
      if (prof) tt0 = wall_clock64();
      const bool fit_pass = (L.S.phase == PH_DONE);
      // A pass is run as one or more SEGMENTS of consecutive units.  A derivative pass is one segment:
      // solo (one walk per wave) or split over the registered helpers.  The fitness pass runs once, can
      // be long (a poor match walks many rings per point) and usually starts when no helper is free:
      // solo, its units are handed to the waves one at a time from an LDS counter and the segment is
      // closed as soon as a helper has registered, so that the rest of the pass is shared.
      int pass_h = 0, ubeg = 0;
      bool pose_out = false;                       // thread 0: pose block of this pass is in the control block
      for (int seg = 0; seg <= kUnits && ubeg < kUnits; ++seg) {
        if (threadIdx.x == 0) {
          if (seg == 0) {
            L.PP.T = L.S.T; L.PP.cj = L.S.cj; L.PP.sj = L.S.sj; L.PP.ch = L.S.ch; L.PP.sh = L.S.sh;
            L.PP.kind = fit_pass ? 1 : 0;
            if (allow_helpers) { st32(&C->passes, (u32)L.S.evals); if (fit_pass) st32(&C->phase, 1u); }
          } else if (allow_helpers) {
            L.sflag[1] = (int)ld32(&C->ready);
          }
          const int h = allow_helpers ? min(L.sflag[1], kMaxHelpers) : 0;
          L.sflag[0] = h;
          L.jnext = 0; L.stop = 0;
          if (h > 0) {                          // open an epoch: pose block, then the epoch word
            const PassPose pp = L.PP;
            if (!pose_out) {
              pose_out = true;
              st64(&C->pose[0], ((u64)__float_as_uint(pp.T.s) << 32) | (u64)__float_as_uint(pp.T.c));
              st64(&C->pose[1], ((u64)__float_as_uint(pp.T.ty) << 32) | (u64)__float_as_uint(pp.T.tx));
              st64(&C->pose[2], (u64)__double_as_longlong(pp.cj)); st64(&C->pose[3], (u64)__double_as_longlong(pp.sj));
              st64(&C->pose[4], (u64)__double_as_longlong(pp.ch)); st64(&C->pose[5], (u64)__double_as_longlong(pp.sh));
            }
            st32(&C->arrive, 0u);
            drain_vmem();
            st64(&C->ticket, ((u64)(epoch + 1) << 32) | ((u64)pp.kind << 24) | ((u64)h << 16) | ((u64)kUnits << 8) | (u64)ubeg);
          }
        }
        __syncthreads();
        if (prof) ts1 = wall_clock64();
        const int nhelp = L.sflag[0];
        const PassPose pp = L.PP;
        int uend = kUnits;
        if (nhelp <= 0 && !fit_pass) {
          // solo derivative pass: wave w computes its own units (w, 0..kSub-1) in one walk
          unit_sums<SSE, INCL>(M, W, L.etab, pp, pts, n, wave, 0, kSub, L.wpart + wave * 12, kWaves * 12);
        } else {
          // this workgroup's units ubeg + j*(nhelp+1), j = 0, 1, ... handed to its waves from an LDS counter
          const bool watch = fit_pass && nhelp == 0 && allow_helpers;
          for (int it = 0; it <= kUnits; ++it) {             // counted (tools/repro/ticket2.hip)
            if (watch && L.stop) break;
            int j = 0;
            if (lane == 0) j = atomicAdd(&L.jnext, 1);
            j = __builtin_amdgcn_readfirstlane(j);
            const int u = ubeg + j * (nhelp + 1);
            if (u >= kUnits) break;
            unit_sums<SSE, INCL>(M, W, L.etab, pp, pts, n, u % kWaves, u / kWaves, u / kWaves + 1, L.wpart + u * 12, 0);
            if (watch && wave == kWaves - 1 && lane == 0) {
              // one wave looks for a registered helper between its units.  (Raising the scan's priority
              // when this pass runs long was tried: it draws helpers away from the scans that still have
              // tens of passes to go and cost 5 % of the batch rate.)
              if (ld32(&C->ready) > 0u) L.stop = 1;
            }
          }
          if (watch) {                                       // units [ubeg, ubeg + claimed) are done
            __syncthreads();
            uend = min(kUnits, ubeg + L.jnext);
          }
        }
        if (nhelp > 0) {
          ++epoch;
          pass_h = nhelp;
          __syncthreads();
          if (prof) ts2 = wall_clock64();
          // wait for the helpers' units (every counted helper is polling the epoch word or computing)
          if (threadIdx.x == 0) {
            const int total = kUnits - ubeg;
            const int mine = (total + nhelp) / (nhelp + 1);
            const u32 need = (u32)(total - mine);
            int bad = 0; unsigned polls = 0;
            const u64 w0 = wall_clock64();
            while (ld32(&C->arrive) < need) {
              if (watchdog(hdr, t_start, polls)) { bad = 1; break; }
              __builtin_amdgcn_s_sleep(2);
            }
            t_wait += wall_clock64() - w0;
            if (n_shared == 0) t_first_shared = w0 - t_start;
            n_shared += 1; n_helped += need;
            L.sflag[2] = bad;
          }
          __syncthreads();
          if (prof) ts3 = wall_clock64();
          if (L.sflag[2]) { aborted = true; break; }
          // helpers' totals of this segment: one load per lane, in flight together
          if (threadIdx.x < (kUnits - ubeg) * 12) {
            const int u = ubeg + threadIdx.x / 12;
            if ((u - ubeg) % (nhelp + 1) != 0)
              L.wpart[ubeg * 12 + threadIdx.x] = __longlong_as_double((long long)ld64(&mytot[ubeg * 12 + threadIdx.x]));
          }
        }
        __syncthreads();
        ubeg = uend;
      }
      if (aborted) break;
      // pass total: the units in four groups of 16, each summed in unit order by one lane per value,
      // then the four partial sums in order; wave 0 goes straight on to the optimiser step
      static_assert(kUnits == 64, "four groups of 16 units");
      if (threadIdx.x < 64) {
        const int j = lane % 12, grp = lane / 12;             // lanes 48..63: nothing to add
        double part = 0.0;
        if (lane < 48) for (int v = 16 * grp; v < 16 * grp + 16; ++v) part += L.wpart[v * 12 + j];
        const double p1 = __shfl(part, j + 12), p2 = __shfl(part, j + 24), p3 = __shfl(part, j + 36);
        if (lane < 12) L.tot[lane] = ((part + p1) + p2) + p3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (prof) { tt1 = wall_clock64(); }
        if (!fit_pass && lane == 0) advance(L.S, P, M, L.tot, tr, trace_cap, trace_rows ? trace_rows + b : nullptr);
      }
      // meanwhile another wave fetches the number of registered helpers for the next pass
      if (!fit_pass && threadIdx.x == 64 && allow_helpers) L.sflag[1] = (int)rd32_fresh(&C->ready);
      if (fit_pass) fitness_done = true;
      __syncthreads();
      if (prof) {
        const u64 te = wall_clock64();
        if (threadIdx.x >= 64) tt1 = te;          // (only wave 0 stamps the end of the summation)
        t_eval += tt1 - tt0; if (fit_pass) t_fit = tt1 - tt0;
        t_adv += te - tt1;
        if (pass_h > 0 && !fit_pass) { a_pro += ts1 - tt0; a_own += ts2 - ts1; a_wait += ts3 - ts2; a_comb += tt1 - ts3; a_adv += te - tt1; a_n += 1; }
      }
    }
    // ---- result record; close the scan ----
    if (threadIdx.x == 0) {
      const AlignState &S = L.S;
      const Tf32 T = S.T;
      ndt_result R_;
      R_.pose[0] = (double)T.tx; R_.pose[1] = (double)T.ty; R_.pose[2] = yaw_from_T(T.c, T.s);
      R_.T00 = T.c; R_.T10 = T.s; R_.T03 = T.tx; R_.T13 = T.ty;
      R_.fitness = (fitness_done && L.tot[1] > 0) ? L.tot[0] / L.tot[1] : DBL_MAX;
      R_.score = S.score;
      R_.trans_prob = n > 0 ? S.score / (double)n : 0.0;
      R_.H[0] = S.H[0]; R_.H[1] = S.H[1]; R_.H[2] = S.H[2];
      R_.H[3] = S.H[1]; R_.H[4] = S.H[3]; R_.H[5] = S.H[4];
      R_.H[6] = S.H[2]; R_.H[7] = S.H[4]; R_.H[8] = S.H[5];
      R_.p[0] = S.p[0]; R_.p[1] = S.p[1]; R_.p[2] = S.p[2];
      R_.iters = S.iters; R_.evals = S.evals;
      R_.ref_evals = S.ref_evals + 1;       // + the getHessian pass (src/PoseEstimator.cpp:56)
      R_.converged = S.converged;
      R_.status = aborted ? NDT_E_HIP : (n > 0 ? NDT_OK : NDT_E_ARG);
      R_.pad_ = 0;
      R_.kbar = (S.evals > 0 && n > 0) ? S.pairs / ((double)S.evals * (double)n) : 0.0;
      results[b] = R_;
      if (allow_helpers) {
        st64(&C->ticket, (u64)kEpochDone << 32);
        __hip_atomic_fetch_add(&hdr->done, 1u, NDT_RLX, NDT_AGENT);
      }
      if (prof) {
        prof[8 * b + 0] = t_eval; prof[8 * b + 1] = t_adv | (t_fit << 32) | ((u64)(L.RG.nspill > 0) << 63); prof[8 * b + 2] = (t_first_shared << 32) | (t_scan0 & 0xFFFFFFFFull);
        prof[8 * b + 3] = (unsigned long long)S.evals | ((u64)n_shared << 16) | ((u64)n_helped << 32);
        prof[8 * b + 6] = t_wait; prof[8 * b + 7] = wall_clock64() - t_start;
        u64 *p2 = prof + 8 * (size_t)B + 8 * (size_t)b;
        p2[0] = a_n; p2[1] = a_pro; p2[2] = a_own; p2[3] = a_wait; p2[4] = a_comb; p2[5] = a_adv;
      }
    }
      // next scan of the batch, if any
    __syncthreads();
    if (threadIdx.x == 0) L.sflag[3] = (int)gridDim.x + (int)__hip_atomic_fetch_add(&hdr->next, 1u, NDT_RLX, NDT_AGENT);
    __syncthreads();
    b = L.sflag[3];
  }

  // ============================================ helper ============================================
  if (!allow_helpers || aborted) return;
  u64 idle_ticks = 400;
  for (unsigned rounds = 0; rounds < 0x40000000u; ++rounds) {
    // ---- find an unfinished scan that still has room for a helper ----
    __syncthreads();
    if (threadIdx.x == 0) { L.sflag[0] = INT_MAX; L.sflag[3] = 0; }
    __syncthreads();
    const int start = (int)((blockIdx.x * 97u) % (unsigned)B);
    // helpers per scan: as many as the unfinished scans leave workgroups for (the last stragglers get
    // up to kMaxHelpers, a unit each per wave)
    const int unfinished = max(1, B - (int)ld32(&hdr->done));
    const int room = min(allow_helpers, max(min(allow_helpers, kBaseHelpers), (int)gridDim.x / unfinished - 1));
    for (int k = threadIdx.x; k < B; k += kBlock) {
      int b = start + k; if (b >= B) b -= B;
      const u32 ep = (u32)(rd64_fresh(&ctl[b].ticket) >> 32);
      if (ep == 0u || ep == kEpochDone) continue;
      const u32 h = rd32_fresh(&ctl[b].helpers);
      if (h >= (u32)room) continue;
      // a scan that already needed many passes will likely need many more: most passes first,
      // each attached helper counting like 4 passes fewer; then nearest
      const int score = (int)min(ld32(&ctl[b].passes), 200u) - NDT_HELPER_PENALTY * (int)h;
      atomicMin(&L.sflag[0], (int)(((u32)(512 - score) << 20) | (u32)k));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int code = -1;                                       // -1: nothing joinable right now
      unsigned polls = 63;
      if (ld32(&hdr->done) >= (u32)B || watchdog(hdr, t_start, polls)) code = -2;   // -2: leave
      else if (L.sflag[0] != INT_MAX) {
        int b = start + (L.sflag[0] & 0xFFFFF); if (b >= B) b -= B;
        const u32 h = __hip_atomic_fetch_add(&ctl[b].helpers, 1u, NDT_RLX, NDT_AGENT);
        if (h >= (u32)room) {
          __hip_atomic_fetch_sub(&ctl[b].helpers, 1u, NDT_RLX, NDT_AGENT);   // lost the race: look again
        } else {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // geometry + ordered copy of the owner
          drain_vmem();
          if (prof && h == 0) prof[8 * b + 4] = wall_clock64() - t_start;
          code = b;
        }
      }
      L.sflag[3] = code;
      if (code == -1) {                                    // back off: 4 us, doubling up to NDT_IDLE_MAX ticks
        const u64 t0 = wall_clock64();
        while (wall_clock64() - t0 < idle_ticks) __builtin_amdgcn_s_sleep(64);
        if (idle_ticks < NDT_IDLE_MAX) idle_ticks *= 2;
      } else {
        idle_ticks = 400;
      }
    }
    __syncthreads();
    const int vb = L.sflag[3];
    if (vb == -2) break;
    if (vb < 0) continue;
    // ---- attached to scan vb: stage its window, register, then serve its epochs until it is done ----
    ScanCtl *C = ctl + vb;
    const u64 o0 = shared_scan ? offsets[0] : offsets[vb];
    const u64 o1 = shared_scan ? offsets[1] : offsets[vb + 1];
    const int n = (int)(o1 - o0);
    if (threadIdx.x == 0) { L.sflag[1] = (int)C->use_sorted; L.sflag[0] = (int)C->owner_wg; }
    if (threadIdx.x == 0) {
      Region r; r.x0 = C->region[0]; r.y0 = C->region[1]; r.rw = C->region[2]; r.rh = C->region[3];
      r.cap = C->region[4]; r.nspill = C->region[5];
      L.RG = r;
    }
    {
      unsigned *wmap = reinterpret_cast<unsigned *>(L.wpart);
      const unsigned *gw = wantmap + (size_t)vb * (kRegionCells / 32);
      for (int i = threadIdx.x; i < kRegionCells / 32; i += kBlock) wmap[i] = gw[i];
    }
    if (threadIdx.x == 0) L.sflag[2] = (int)ld32(&C->phase);
    __syncthreads();
    const int owner_wg = L.sflag[0];
    const float2 *pts = L.sflag[1] ? (shared_scan ? sorted + (size_t)owner_wg * (size_t)n : sorted + o0)
                                   : (reinterpret_cast<const float2 *>(scans) + o0);
    if (L.sflag[2] == 0) fill_window(M, L, pool);          // a scan in its fitness pass needs no window
    const Window W = window_of(L.RG, pool);
    if (prof && threadIdx.x == 0 && prof[8 * vb + 5] == 0) prof[8 * vb + 5] = wall_clock64() - t_start;
    u64 *vtot = utot + (size_t)vb * kUnits * 12;
    // register: from now on this workgroup does nothing but watch the scan's epoch word
    if (threadIdx.x == 0) L.hrank = (int)__hip_atomic_fetch_add(&C->ready, 1u, NDT_RLX, NDT_AGENT);
    __syncthreads();
    const int rank = L.hrank;
    u32 last_ep = 0;
    for (unsigned turns = 0; turns < 0x40000000u; ++turns) {         // counted (tools/repro/ticket2.hip)
      if (wave == 0) {
        // wave 0 polls the epoch word (one load in flight per helper workgroup on the owner's line)
        u64 word = 0;
        if (lane == 0) {
          unsigned polls = 0;
          for (unsigned it = 0; it < 0x40000000u; ++it) {
            word = ld64(&C->ticket);
            if ((u32)(word >> 32) != last_ep && (u32)(word >> 32) != 0u) break;
            if (watchdog(hdr, t_start, polls)) { word = (u64)kEpochDone << 32; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        word = wave_bcast64(word);
        const int h = (int)((word >> 16) & 0xFFu);
        if ((u32)(word >> 32) != kEpochDone && rank < h && lane < 6) L.hpose[lane] = ld64(&C->pose[lane]);   // stable: this helper is counted in
        if (lane == 0) { L.hword = word; L.jnext = 0; }
      }
      __syncthreads();
      const u64 word = L.hword;
      const u32 ep = (u32)(word >> 32);
      if (ep == kEpochDone) break;
      last_ep = ep;
      const int h = (int)((word >> 16) & 0xFFu), ubeg = (int)(word & 0xFFu), uend = (int)((word >> 8) & 0xFFu);
      int done_units = 0;
      if (rank < h) {
        PassPose pp;
        const u64 w0 = L.hpose[0], w1 = L.hpose[1];
        pp.T.c = __uint_as_float((u32)w0); pp.T.s = __uint_as_float((u32)(w0 >> 32));
        pp.T.tx = __uint_as_float((u32)w1); pp.T.ty = __uint_as_float((u32)(w1 >> 32));
        pp.cj = __longlong_as_double((long long)L.hpose[2]); pp.sj = __longlong_as_double((long long)L.hpose[3]);
        pp.ch = __longlong_as_double((long long)L.hpose[4]); pp.sh = __longlong_as_double((long long)L.hpose[5]);
        pp.kind = (int)((word >> 24) & 0xFFu);
        double *wt = L.wtmp + wave * 12;
        for (int it = 0; it <= kUnits; ++it) {               // this workgroup's units, handed out from an LDS counter
          int j = 0;
          if (lane == 0) j = atomicAdd(&L.jnext, 1);
          j = __builtin_amdgcn_readfirstlane(j);
          const int u = ubeg + (rank + 1) + j * (h + 1);
          if (u >= uend) break;
          unit_sums<SSE, INCL>(M, W, L.etab, pp, pts, n, u % kWaves, u / kWaves, u / kWaves + 1, wt, 0);
          if (lane < 12) st64(&vtot[u * 12 + lane], (u64)__double_as_longlong(wt[lane]));
        }
        drain_vmem();                                        // the whole wave: its stores have landed
        const int total = uend - ubeg;
        done_units = (total - (rank + 1) + h) / (h + 1);     // units ubeg + rank+1 + j*(h+1) below uend
        if (done_units < 0) done_units = 0;
      }
      __syncthreads();
      if (threadIdx.x == 0 && done_units > 0) __hip_atomic_fetch_add(&C->arrive, (u32)done_units, NDT_RLX, NDT_AGENT);
    }
  }
}

// One derivative pass at an explicit pose (tests / profiling): grid-stride over points,
// one partial record per workgroup, summed on the host in block order.
template <bool SSE, bool INCL>
__global__ void __launch_bounds__(256)
ndt_eval_kernel(MapView M, double snap, const float *__restrict__ scan, size_t stride, int n,
                double p0, double p1, double p2, double *__restrict__ partial /* grid x kAcc */) {
  __shared__ double sred[(4 + 1) * kAcc];
  __shared__ double etab[64];
  __shared__ unsigned short no_slot[16];
  __shared__ CellEntry no_ent[1];
  if (threadIdx.x < 64) etab[threadIdx.x] = c_exp2_tab[threadIdx.x];
  if (threadIdx.x < 16) no_slot[threadIdx.x] = 0;
  if (threadIdx.x == 0) { no_ent[0].cent = make_float2(INFINITY, INFINITY); no_ent[0].mx = no_ent[0].my = 0; no_ent[0].i00 = no_ent[0].i01 = no_ent[0].i11 = 0; }
  __syncthreads();
  double p[3] = {p0, p1, p2};
  Tf32 T = tf_from_p(p);
  double cj, sj;
  angle_cs(snap, p2, cj, sj);
  Acc A = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0u};
  Window W;
  W.R = Region{0, 0, 0, 0, 0, 0}; W.slot = no_slot; W.ent = no_ent;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float2 pt = load_pt(scan, stride, i);
    eval_point<SSE, INCL>(M, W, etab, T, pt.x, pt.y, cj, sj, cj, sj, A);
  }
  block_reduce_acc(A, sred, sred + 4 * kAcc);
  if (threadIdx.x < kAcc) partial[blockIdx.x * kAcc + threadIdx.x] = sred[4 * kAcc + threadIdx.x];
}

__global__ void __launch_bounds__(256)
ndt_fitness_kernel(MapView M, const float *__restrict__ scan, size_t stride, int n, Tf32 T,
                   double *__restrict__ partial /* grid x 2 */) {
  __shared__ double sred[(4 + 1) * 2];
  double fsum = 0.0, fcnt = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float2 pt = load_pt(scan, stride, i);
    float qx, qy;
    tf_apply(T, M.transform_sse, pt.x, pt.y, qx, qy);
    if (!finite2(qx, qy)) continue;
    float best = nearest_sq(M, qx, qy);
    if (best < INFINITY) { fsum += (double)best; fcnt += 1.0; }
  }
  block_reduce2(fsum, fcnt, sred, sred + 4 * 2);
  if (threadIdx.x < 2) partial[blockIdx.x * 2 + threadIdx.x] = sred[4 * 2 + threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// a2: voxel normal-distributions build
// ------------------------------------------------------------------------------------------

// order-preserving float <-> uint for atomicMin/atomicMax
__device__ __forceinline__ unsigned f2ord(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __uint_as_float(u);
#else
  memcpy(&f, &u, 4);
#endif
  return f;
}

// getMinMax3D: bounds[0..3] = ord(min x), ord(min y), ord(max x), ord(max y)
__global__ void __launch_bounds__(256)
map_minmax_kernel(const float *__restrict__ xy, size_t stride, size_t n, unsigned *__restrict__ bounds) {
  __shared__ float sh[4][4];
  float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i0 < n; i0 += 16 * step) {   // 16 loads in flight
    float2 p[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const size_t i = i0 + u * step; p[u] = load_pt(xy, stride, i < n ? i : i0); }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (!finite2(p[u].x, p[u].y)) continue;
      mnx = fminf(mnx, p[u].x); mxx = fmaxf(mxx, p[u].x);
      mny = fminf(mny, p[u].y); mxy = fmaxf(mxy, p[u].y);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mnx = fminf(mnx, __shfl_down(mnx, o)); mny = fminf(mny, __shfl_down(mny, o));
    mxx = fmaxf(mxx, __shfl_down(mxx, o)); mxy = fmaxf(mxy, __shfl_down(mxy, o));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[w][0] = mnx; sh[w][1] = mny; sh[w][2] = mxx; sh[w][3] = mxy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      mnx = fminf(mnx, sh[k][0]); mny = fminf(mny, sh[k][1]);
      mxx = fmaxf(mxx, sh[k][2]); mxy = fmaxf(mxy, sh[k][3]);
    }
    if (mnx <= mxx) {     // one atomic set per workgroup
      atomicMin(&bounds[0], f2ord(mnx)); atomicMin(&bounds[1], f2ord(mny));
      atomicMax(&bounds[2], f2ord(mxx)); atomicMax(&bounds[3], f2ord(mxy));
    }
  }
}

struct GridDims { float inv_leaf; int min_bx, min_by, div_x, div_y, gw, gh; };

__device__ __forceinline__ int voxel_of(const GridDims &G, float2 p) {
  if (!finite2(p.x, p.y)) return -1;
  const float fx = fminf(fmaxf(floorf(p.x * G.inv_leaf), -1.0e9f), 1.0e9f), fy = fminf(fmaxf(floorf(p.y * G.inv_leaf), -1.0e9f), 1.0e9f);
  const int ix = (int)fx - G.min_bx, iy = (int)fy - G.min_by;
  // never true for the grid of this cloud's own bounding box; a build queued ahead of the bounding
  // box read-back with the previous grid (ndt_map_build_dev) must stay inside its buffers
  if (ix < 0 || ix >= G.div_x || iy < 0 || iy >= G.div_y) return -1;
  return iy * G.div_x + ix;
}

// Consecutive cloud points usually fall in the same voxel (a map is appended scan by scan, wall by
// wall), so a wave first merges runs of equal voxel keys among its 64 consecutive points and issues
// one atomic per run instead of one per point.
__device__ __forceinline__ void wave_runs(int v, int lane, int &head, int &len) {
  const int prev = __shfl_up(v, 1);
  const bool is_head = (lane == 0) || (v != prev);
  const unsigned long long heads = __ballot(is_head);
  const unsigned long long below = heads & ((2ull << lane) - 1ull);      // heads at or below this lane
  head = 63 - __builtin_clzll(below);
  const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
  len = above ? (lane + 1 + __builtin_ctzll(above)) - lane : 64 - lane;  // valid in head lanes
}

__global__ void __launch_bounds__(256)
map_count_kernel(const float *__restrict__ xy, size_t stride, size_t n, GridDims G, int *__restrict__ count) {
  const int lane = threadIdx.x & 63;
  const size_t nround = (n + 63) / 64 * 64;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nround; i += (size_t)gridDim.x * blockDim.x) {
    const int v = i < n ? voxel_of(G, load_pt(xy, stride, i)) : -2;
    int head, len;
    wave_runs(v, lane, head, len);
    if (head == lane && v >= 0) atomicAdd(&count[v], len);
  }
}

// exclusive scan of count[0..ng) into start[0..ng], three small kernels
constexpr int kScanBlock = 256, kScanPer = 8, kScanTile = kScanBlock * kScanPer;
constexpr int kBigVoxel = 16;        // voxels with more points are handled by a whole wave (order, statistics)

__global__ void __launch_bounds__(kScanBlock)
scan_tile_sums_kernel(const int *__restrict__ in, size_t n, int *__restrict__ tile_sum) {
  __shared__ int sh[kScanBlock / 64];
  size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanPer;
  int s = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) if (base + k < n) s += in[base + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < kScanBlock / 64; ++w) t += sh[w]; tile_sum[blockIdx.x] = t; }
}

__global__ void __launch_bounds__(1024)
scan_tile_offsets_kernel(int *__restrict__ tile_sum, int ntiles, int *__restrict__ total) {
  // single workgroup: exclusive scan of the tile sums, in place
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += 1024) {
    int i = base + threadIdx.x;
    int v = i < ntiles ? tile_sum[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    int incl = sh[threadIdx.x];
    if (i < ntiles) tile_sum[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(kScanBlock)
scan_apply_kernel(const int *__restrict__ in, size_t n, const int *__restrict__ tile_off,
                  int *__restrict__ out /* n + 1 */, const int *__restrict__ total,
                  int *__restrict__ big /* voxels with more than kBigVoxel points */, int *__restrict__ nbig, int big_cap) {
  __shared__ int sh[kScanBlock];
  size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanPer;
  int v[kScanPer]; int s = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < kScanBlock; o <<= 1) {
    int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  int run = tile_off[blockIdx.x] + sh[threadIdx.x] - s;
  unsigned bigmask = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
    if (v[k] > kBigVoxel) bigmask |= 1u << k;
  }
  {                                              // list of the big voxels: one atomic per wave
    const int mine = __builtin_popcount(bigmask), lane = threadIdx.x & 63;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    const int wave_total = __shfl(incl, 63);
    int q0 = 0;
    if (lane == 63 && wave_total > 0) q0 = atomicAdd(nbig, wave_total);
    q0 = __shfl(q0, 63) + incl - mine;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k)
      if ((bigmask >> k) & 1u) { if (q0 < big_cap) big[q0] = (int)(base + k); ++q0; }
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) out[n + threadIdx.x] = *total;   // out[n], + 3 readable copies
}

__global__ void __launch_bounds__(256)
map_scatter_kernel(const float *__restrict__ xy, size_t stride, size_t n, GridDims G,
                   const int *__restrict__ start, int *__restrict__ fill, int *__restrict__ perm) {
  const int lane = threadIdx.x & 63;
  const size_t nround = (n + 63) / 64 * 64;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nround; i += (size_t)gridDim.x * blockDim.x) {
    const int v = i < n ? voxel_of(G, load_pt(xy, stride, i)) : -2;
    int head, len;
    wave_runs(v, lane, head, len);
    int base = 0;
    if (head == lane && v >= 0) base = start[v] + atomicAdd(&fill[v], len);   // one slot range per run
    base = __shfl(base, head);
    if (v >= 0) perm[base + (lane - head)] = (int)i;                          // cloud order kept inside a run
  }
}

// Restore input order inside every bucket (PCL accumulates a voxel's points in cloud order and
// its float32 centroid sum depends on that order), rank by counting.  Voxels of up to kBigVoxel
// points: eight lanes per voxel (one wave per voxel spent its time launching waves, 70 % of the
// voxels being empty); the others, listed by scan_apply_kernel: one wave per voxel.
constexpr int kOrderVoxPerBlock = 256 / 8 * 4;     // 32 lane groups, 4 voxels each
__global__ void __launch_bounds__(256)
map_order_small_kernel(const int *__restrict__ start, size_t ng, const int *__restrict__ perm,
                       int *__restrict__ perm_sorted) {
  const int grp = threadIdx.x >> 3, sub = threadIdx.x & 7;
  for (int r = 0; r < 4; ++r) {
    const size_t g = (size_t)blockIdx.x * kOrderVoxPerBlock + (size_t)r * 32 + grp;
    if (g >= ng) return;
    const int s0 = start[g], n = start[g + 1] - s0;
    if (n > kBigVoxel) continue;
    for (int e = sub; e < n; e += 8) {
      const int mine = perm[s0 + e];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (perm[s0 + j] < mine) ? 1 : 0;
      perm_sorted[s0 + rank] = mine;
    }
  }
}

constexpr int kBigWavesPerBlock = 4, kBigBlocks = 1024, kBigStage = 512;   // LDS staging: point numbers per wave
__global__ void __launch_bounds__(256)
map_order_big_kernel(const int *__restrict__ start, const int *__restrict__ big, const int *__restrict__ nbig,
                     int big_cap, const int *__restrict__ perm, int *__restrict__ perm_sorted) {
  __shared__ int stage[kBigWavesPerBlock][kBigStage];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int count = min(*nbig, big_cap);
  for (int q = blockIdx.x * kBigWavesPerBlock + wv; q < count; q += gridDim.x * kBigWavesPerBlock) {
    const int g = big[q];
    const int s0 = start[g], n = start[g + 1] - s0;
    const bool staged = n <= kBigStage;
    if (staged) for (int e = lane; e < n; e += 64) stage[wv][e] = perm[s0 + e];
    __builtin_amdgcn_wave_barrier();            // one wave: its LDS writes are ordered before its later reads
    for (int e = lane; e < n; e += 64) {
      const int mine = staged ? stage[wv][e] : perm[s0 + e];
      int rank = 0;
      if (staged) { for (int j = 0; j < n; ++j) rank += (stage[wv][j] < mine) ? 1 : 0; }
      else        { for (int j = 0; j < n; ++j) rank += (perm[s0 + j] < mine) ? 1 : 0; }
      perm_sorted[s0 + rank] = mine;
    }
  }
}

struct LeafParams { int min_pts, cov_unbiased, cov_init_identity; double eig_mult; };

// Mean, regularised covariance and its inverse of one z = 0 voxel (second loop of
// VoxelGridCovariance::applyFilter); closed-form 2x2 eigen-decomposition, the z eigenpair is
// exactly (czz, e_z).  Returns 1 accepted, 0 rejected (icov = 0), -1 rejected with inf icov.
__device__ int leaf_finalize(const LeafParams &L, int n, double sx, double sy, double sxx,
                             double sxy, double syy, double szz, double mean[2], double icov[3]) {
  const double dn = (double)n;
  const double mx = sx / dn, my = sy / dn;
  mean[0] = mx; mean[1] = my;
  icov[0] = icov[1] = icov[2] = 0.0;
  double cxx, cxy, cyy, czz;
  if (!L.cov_unbiased) {
    cxx = (sxx - 2.0 * (sx * mx)) / dn + mx * mx;
    cxy = (sxy - 2.0 * (sx * my)) / dn + mx * my;
    cyy = (syy - 2.0 * (sy * my)) / dn + my * my;
    czz = szz / dn;
    const double f = (dn - 1.0) / dn;
    cxx *= f; cxy *= f; cyy *= f; czz *= f;
  } else {
    cxx = (sxx - sx * mx) / (dn - 1.0);
    cxy = (sxy - sx * my) / (dn - 1.0);
    cyy = (syy - sy * my) / (dn - 1.0);
    czz = szz / (dn - 1.0);
  }
  const double hd = 0.5 * (cxx - cyy), tr = 0.5 * (cxx + cyy);
  const double rad = sqrt(hd * hd + cxy * cxy);
  const double l1 = tr - rad, l2 = tr + rad;
  double vx, vy;
  if (rad == 0.0) { vx = 1.0; vy = 0.0; }
  else if (hd >= 0.0) { vx = hd + rad; vy = cxy; }
  else { vx = cxy; vy = rad - hd; }
  const double vn = sqrt(vx * vx + vy * vy);
  if (vn == 0.0) { vx = 1.0; vy = 0.0; } else { vx /= vn; vy /= vn; }
  // ascending order of {l1, l2, czz}; z first among equals
  double ev0, ev1, ev2; int k0, k1, k2;   // kind: 0 = l1, 1 = l2, 2 = z
  if (czz <= l1)      { ev0 = czz; k0 = 2; ev1 = l1; k1 = 0; ev2 = l2; k2 = 1; }
  else if (czz <= l2) { ev0 = l1; k0 = 0; ev1 = czz; k1 = 2; ev2 = l2; k2 = 1; }
  else                { ev0 = l1; k0 = 0; ev1 = l2; k1 = 1; ev2 = czz; k2 = 2; }
  if (ev0 < 0 || ev1 < 0 || ev2 <= 0) return 0;
  const double thr = L.eig_mult * ev2;
  bool rebuilt = false;
  if (ev0 < thr) { ev0 = thr; if (ev1 < thr) ev1 = thr; rebuilt = true; }
  double n1 = l1, n2 = l2;
  if (k0 == 0) n1 = ev0; else if (k0 == 1) n2 = ev0;
  if (k1 == 0) n1 = ev1; else if (k1 == 1) n2 = ev1;
  if (k2 == 0) n1 = ev2; else if (k2 == 1) n2 = ev2;
  if (rebuilt) {
    cxx = n1 * (vy * vy) + n2 * (vx * vx);
    cxy = -n1 * (vx * vy) + n2 * (vx * vy);
    cyy = n1 * (vx * vx) + n2 * (vy * vy);
  }
  const double det = cxx * cyy - cxy * cxy;
  icov[0] = cyy / det; icov[1] = -cxy / det; icov[2] = cxx / det;
  for (int a = 0; a < 3; ++a)
    if (icov[a] == (double)INFINITY || icov[a] == -(double)INFINITY) return -1;
  return 1;
}

// Cell record of one voxel from its sums (shared by the two kernels below).
__device__ __forceinline__ int write_voxel(const GridDims &G, const LeafParams &L, size_t g, int n, float fx, float fy,
                                           double sx, double sy, double sxx, double sxy, double syy, double szz,
                                           float2 *__restrict__ cent, double *__restrict__ rec, int *__restrict__ counters) {
  if (n < L.min_pts) return 0;
  const int ix = (int)(g % G.div_x), iy = (int)(g / G.div_x);
  const size_t pg = (size_t)(iy + 2) * G.gw + (ix + 2);
  double mean[2], icov[3];
  const int ok = leaf_finalize(L, n, sx, sy, sxx, sxy, syy, szz, mean, icov);
  cent[pg] = make_float2(fx / (float)n, fy / (float)n);
  double *r = rec + pg * 8;
  r[0] = mean[0]; r[1] = mean[1]; r[2] = icov[0]; r[3] = icov[1]; r[4] = icov[2];
  return ok > 0 ? n : -n;
}

// One lane per voxel: sequential sums in cloud order (float32 centroid, fp64 mean / Sxx), bucketed
// copy of the raw points, cell record.
__global__ void __launch_bounds__(256)
map_finalize_kernel(const float *__restrict__ xy, size_t stride, GridDims G, LeafParams L,
                          const int *__restrict__ start, const int *__restrict__ perm_sorted,
                          float2 *__restrict__ pts, float2 *__restrict__ cent, double *__restrict__ rec,
                          int *__restrict__ npts_grid, int *__restrict__ counters /* unused */,
                          unsigned *__restrict__ occ /* (ng + 31) / 32 words: voxel in the search set */) {
  const size_t ng = (size_t)G.div_x * G.div_y;
  size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const bool live = g < ng;
  int s0 = 0, s1 = 0;
  if (live) { s0 = start[g]; s1 = start[g + 1]; }
  const int n = s1 - s0;
  int flag = 0;
  if (n > 0) {
    float fx = 0.f, fy = 0.f;
    double sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0, szz = 0;
    if (L.cov_init_identity) { sxx = 1.0; syy = 1.0; szz = 1.0; }
    for (int s = s0; s < s1; s += 8) {          // eight gathers in flight
      int ib[8]; float2 pb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) ib[u] = perm_sorted[min(s + u, s1 - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) pb[u] = load_pt(xy, stride, (size_t)ib[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (s + u >= s1) break;
        const float2 p = pb[u];                 // strictly in cloud order: these sums define the voxel
        pts[s + u] = p;
        fx += p.x; fy += p.y;
        const double X = (double)p.x, Y = (double)p.y;
        sx += X; sy += Y;
        sxx += X * X; sxy += X * Y; syy += Y * Y;
      }
    }
    flag = write_voxel(G, L, g, n, fx, fy, sx, sy, sxx, sxy, syy, szz, cent, rec, counters);
  }
  if (live) npts_grid[g] = flag;
  const u64 in_set = __ballot(flag != 0);       // the wave's 64 consecutive voxels (blockDim is a multiple of 64)
  if ((threadIdx.x & 63) == 0 && live) {
    occ[g >> 5] = (unsigned)in_set;
    if ((g >> 5) + 1 < (ng + 31) / 32) occ[(g >> 5) + 1] = (unsigned)(in_set >> 32);
  }
}

__global__ void fill_f2_kernel(float2 *p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float2(v, v);
}


// ------------------------------------------------------------------------------------------
// f1: source pre-filter = pcl::ApproximateVoxelGrid::filter on a z = 0 cloud
// (src/PoseEstimator.cpp:6-10).  The filter is a sequential machine: 512
// direct-mapped slots, a point either joins the voxel its slot holds or flushes that voxel's
// centroid to the output and takes the slot; what is left is flushed in slot order at the end.
// One wave per scan replays it 64 points at a time: lanes whose points hash to different slots
// update them at once, lanes sharing a slot take turns in point order (the float32 sums of a slot
// are therefore added in cloud order), and the flushes of a step are written in point order.
// Output = the reference's output, bit for bit and in the same order.
// ------------------------------------------------------------------------------------------
constexpr int kPfSlots = 512;
struct __attribute__((aligned(8))) PfSlot { int ix, iy, cnt; float cx, cy; int pad; };   // 24 B: one b128 + one b64 LDS read
__global__ void __launch_bounds__(64)
prefilter_kernel(const float *__restrict__ xy, size_t stride, const unsigned long long *__restrict__ offsets, int B,
                 float leaf, float2 *__restrict__ tmp /* at the raw offsets */, unsigned *__restrict__ counts) {
  __shared__ PfSlot slot[kPfSlots];
  const int lane = threadIdx.x;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const float inv = 1.0f / leaf;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const unsigned long long o0 = offsets[b];
    const int n = (int)(offsets[b + 1] - o0);
    for (int h = lane; h < kPfSlots; h += 64) { PfSlot z; z.ix = 0; z.iy = 0; z.cnt = 0; z.cx = 0.f; z.cy = 0.f; z.pad = 0; slot[h] = z; }
    __builtin_amdgcn_wave_barrier();
    int nout = 0;
    float2 pnext = make_float2(0.f, 0.f);
    if (lane < n) pnext = load_pt(xy, stride, (size_t)o0 + (size_t)lane);
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const bool active = i < n;
      const float2 p = pnext;
      if (i + 64 < n) pnext = load_pt(xy, stride, (size_t)o0 + (size_t)(i + 64));   // next step's points in flight
      const int ix = (int)floorf(p.x * inv), iy = (int)floorf(p.y * inv);
      const unsigned h = ((unsigned)ix * 7171u + (unsigned)iy * 3079u) & (unsigned)(kPfSlots - 1);   // iz = 0
      // lanes of this step that use the same slot, and this lane's turn among them
      unsigned long long peers = __ballot(active);
#pragma unroll
      for (int bit = 0; bit < 9; ++bit) {
        const unsigned long long one = __ballot(active && ((h >> bit) & 1u));
        peers &= ((h >> bit) & 1u) ? one : ~one;
      }
      const int rank = __builtin_popcountll(peers & lt);
      bool flushed = false;
      float fx = 0.f, fy = 0.f;
      for (int turn = 0; turn < 64; ++turn) {
        if (!__ballot(active && rank >= turn)) break;
        if (active && rank == turn) {
          PfSlot e = slot[h];
          if (e.cnt && (ix != e.ix || iy != e.iy)) {          // another voxel holds the slot: flush it
            flushed = true; fx = e.cx / (float)e.cnt; fy = e.cy / (float)e.cnt;
            e.cnt = 0; e.cx = 0.f; e.cy = 0.f;
          }
          e.ix = ix; e.iy = iy; e.cnt += 1; e.cx += p.x; e.cy += p.y;
          slot[h] = e;
        }
        __builtin_amdgcn_wave_barrier();
      }
      const unsigned long long fb = __ballot(flushed);
      if (flushed) tmp[o0 + (unsigned long long)(nout + __builtin_popcountll(fb & lt))] = make_float2(fx, fy);
      nout += __builtin_popcountll(fb);
    }
    for (int h0 = 0; h0 < kPfSlots; h0 += 64) {              // what is left, in slot order
      const PfSlot e = slot[h0 + lane];
      const unsigned long long fb = __ballot(e.cnt > 0);
      if (e.cnt > 0) tmp[o0 + (unsigned long long)(nout + __builtin_popcountll(fb & lt))] =
          make_float2(e.cx / (float)e.cnt, e.cy / (float)e.cnt);
      nout += __builtin_popcountll(fb);
    }
    if (lane == 0) counts[b] = (unsigned)nout;
    __builtin_amdgcn_wave_barrier();
  }
}

// offsets of the filtered scans: exclusive scan of the counts (one workgroup)
__global__ void __launch_bounds__(1024)
prefilter_offsets_kernel(const unsigned *__restrict__ counts, int B, unsigned long long *__restrict__ out_offsets) {
  __shared__ unsigned long long sh[1024];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < B; base += 1024) {
    const int i = base + threadIdx.x;
    const unsigned long long v = i < B ? counts[i] : 0ull;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const unsigned long long t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0ull;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < B) out_offsets[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) out_offsets[B] = carry;
}

// filtered points from their raw offsets to the packed output
__global__ void __launch_bounds__(256)
prefilter_pack_kernel(const float2 *__restrict__ tmp, const unsigned long long *__restrict__ raw_offsets,
                      const unsigned long long *__restrict__ out_offsets, int B, float2 *__restrict__ out) {
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const unsigned long long r0 = raw_offsets[b], q0 = out_offsets[b];
    const unsigned long long n = out_offsets[b + 1] - q0;
    for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < n;
         j += (unsigned long long)gridDim.x * blockDim.x)
      out[q0 + j] = tmp[r0 + j];
  }
}


// ------------------------------------------------------------------------------------------
// f2: the steps either side of the match for a batch -- odometry prediction (Pose2D::calMotion +
// calPredPose, src/Pose2D.cpp:5-37, chained as in src/ScanMatcher.cpp:27-32) and, after the
// match, cost / NDT covariance (src/PoseEstimator.cpp:43-64), the accept test
// (src/ScanMatcher.cpp:50) and the EKF fusion or the odometry covariance alone
// (src/PoseFuser.cpp:3-61).  One lane per match, fp64, the oracle's expression order.
// Poses are (tx, ty, th) with th in degrees (include/ndt_slam/Pose2D.h:14).
// ------------------------------------------------------------------------------------------
struct FuseParams { double coe_ndt_cov, coe_vel, coe_omega, del_time, score_thre; };
__device__ __forceinline__ double f2_deg2rad(double x) { return x * M_PI / 180; }
__device__ __forceinline__ double f2_rad2deg(double x) { return x * 180 / M_PI; }
__device__ __forceinline__ double f2_add_angle(double a1, double a2) {
  double sum = a1 + a2;
  if (sum < -180) sum += 360; else if (sum >= 180) sum -= 360;
  return sum;
}
__device__ __forceinline__ double f2_sub_angle(double a1, double a2) {
  double dif = a1 - a2;
  if (dif < -180) dif += 360; else if (dif >= 180) dif -= 360;
  return dif;
}
__device__ __forceinline__ void f2_inv3(const double m[9], double out[9]) {   // Eigen's fixed 3x3 inverse
  const double c00 = m[4] * m[8] - m[5] * m[7];
  const double c10 = m[2] * m[7] - m[1] * m[8];
  const double c20 = m[1] * m[5] - m[2] * m[4];
  const double det = c00 * m[0] + c10 * m[3] + c20 * m[6];
  const double id = 1.0 / det;
  out[0] = c00 * id; out[1] = c10 * id; out[2] = c20 * id;
  out[3] = (m[5] * m[6] - m[3] * m[8]) * id;
  out[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  out[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  out[6] = (m[3] * m[7] - m[4] * m[6]) * id;
  out[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  out[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
__device__ __forceinline__ void f2_mul3(const double a[9], const double b[9], double o[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s = a[3 * i] * b[j];
      s += a[3 * i + 1] * b[3 + j];
      s += a[3 * i + 2] * b[6 + j];
      o[3 * i + j] = s;
    }
}
__device__ __forceinline__ void f2_odo_cov(const double motion[3], const double last[3], const double last_cov[9],
                                           const FuseParams &p, double cov[9]) {
  const double dt = p.del_time;
  const double v = sqrt(motion[0] * motion[0] + motion[1] * motion[1]) / dt;
  const double omega = f2_deg2rad(motion[2] / dt);
  const double m00 = p.coe_vel * v * v, m11 = p.coe_omega * omega * omega;
  const double a = f2_deg2rad(last[2]), c = cos(a), s = sin(a);
  const double F[9] = {1, 0, -v * dt * s, 0, 1, v * dt * c, 0, 0, 1};
  const double Ft[9] = {1, 0, 0, 0, 1, 0, F[2], F[5], 1};
  double t[9], flf[9];
  f2_mul3(F, last_cov, t); f2_mul3(t, Ft, flf);
  const double a0 = dt * c, a1 = dt * s;
  const double ama[9] = {a0 * m00 * a0, a0 * m00 * a1, 0, a1 * m00 * a0, a1 * m00 * a1, 0, 0, 0, dt * m11 * dt};
#pragma unroll
  for (int i = 0; i < 9; ++i) cov[i] = flf[i] + ama[i];
}

__global__ void __launch_bounds__(256)
predict_kernel(const double *__restrict__ odo_cur, const double *__restrict__ odo_prev,
               const double *__restrict__ last_pose, int B, double *__restrict__ motion_out,
               double *__restrict__ pred_out, double *__restrict__ init_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double *cur = odo_cur + 3 * b, *prev = odo_prev + 3 * b, *last = last_pose + 3 * b;
  const double ap = f2_deg2rad(prev[2]), cp = cos(ap), sp = sin(ap);
  const double dx = cur[0] - prev[0], dy = cur[1] - prev[1];
  double motion[3];
  motion[0] = cp * dx + sp * dy;
  motion[1] = -sp * dx + cp * dy;
  motion[2] = f2_sub_angle(cur[2], prev[2]);
  const double al = f2_deg2rad(last[2]), cl = cos(al), sl = sin(al);
  double pred[3];
  pred[0] = cl * motion[0] + -sl * motion[1] + last[0];
  pred[1] = sl * motion[0] + cl * motion[1] + last[1];
  pred[2] = f2_add_angle(last[2], motion[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) { motion_out[3 * b + i] = motion[i]; pred_out[3 * b + i] = pred[i]; }
  if (init_out) {                               // the guess ndt_align takes (src/PoseEstimator.cpp:22-24)
    init_out[3 * b] = pred[0]; init_out[3 * b + 1] = pred[1]; init_out[3 * b + 2] = f2_deg2rad(pred[2]);
  }
}

__global__ void __launch_bounds__(256)
fuse_kernel(const ndt_result *__restrict__ res, const double *__restrict__ pred_pose,
            const double *__restrict__ odo_motion, const double *__restrict__ last_pose,
            const double *__restrict__ last_cov, int B, FuseParams p, double *__restrict__ fused_out,
            double *__restrict__ cov_out, int *__restrict__ successful_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const ndt_result r = res[b];
  double pred[3], motion[3], last[3], lc[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) { pred[i] = pred_pose[3 * b + i]; motion[i] = odo_motion[3 * b + i]; last[i] = last_pose[3 * b + i]; }
#pragma unroll
  for (int i = 0; i < 9; ++i) lc[i] = last_cov[9 * b + i];
  const double est[3] = {r.pose[0], r.pose[1], f2_rad2deg(r.pose[2])};
  const double cost = (r.status == NDT_OK && r.converged) ? r.fitness : 10000000.0;
  const int successful = cost <= p.score_thre;
  double fused[3], cov[9];
  if (!successful) {
    f2_odo_cov(motion, last, lc, p, cov);
    fused[0] = pred[0]; fused[1] = pred[1]; fused[2] = pred[2];
  } else {
    double nh[9], Q[9], ch[9], sum[9], inv[9], K[9], imk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) nh[i] = -r.H[i];
    f2_inv3(nh, Q);
#pragma unroll
    for (int i = 0; i < 9; ++i) Q[i] *= p.coe_ndt_cov;
    f2_odo_cov(motion, last, lc, p, ch);
#pragma unroll
    for (int i = 0; i < 9; ++i) sum[i] = Q[i] + ch[i];
    f2_inv3(sum, inv);
    f2_mul3(ch, inv, K);
#pragma unroll
    for (int i = 0; i < 9; ++i) imk[i] = ((i % 4 == 0) ? 1.0 : 0.0) - K[i];
    f2_mul3(imk, ch, cov);
    const double zh[3] = {est[0] - pred[0], est[1] - pred[1], f2_deg2rad(f2_sub_angle(est[2], pred[2]))};
    const double mu_hat[3] = {pred[0], pred[1], f2_deg2rad(pred[2])};
    double mu[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double s = K[3 * i] * zh[0];
      s += K[3 * i + 1] * zh[1];
      s += K[3 * i + 2] * zh[2];
      mu[i] = s + mu_hat[i];
    }
    fused[0] = mu[0]; fused[1] = mu[1]; fused[2] = f2_rad2deg(mu[2]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) fused_out[3 * b + i] = fused[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) cov_out[9 * b + i] = cov[i];
  if (successful_out) successful_out[b] = successful;
}


// ------------------------------------------------------------------------------------------
// f3 (part): PCFilter::remove_neighborPoint (include/ndt_slam/PCFilter.h:29-56) -- keep the points
// of `base` that have no point of `list` closer than thre_neighbor, in input order.  The
// reference tests every pair (O(n*m) on the CPU, the largest cost outside NDT when moving
// objects are removed, src/PointCloudMap.cpp:15-39); here one lane per base point walks the list
// through LDS tiles.  The distance is PCLUtil::distance_points' float32 expression
// (include/ndt_slam/PCLUtil.h:21-23) compared with the double threshold, so the kept set is
// identical; a ballot prefix keeps the order.
// ------------------------------------------------------------------------------------------
constexpr int kRnBlock = 256, kRnTile = 1024;
__device__ __forceinline__ bool rn_near(float2 p, float2 q, double thre) {
  const float dx = p.x - q.x, dy = p.y - q.y;
  const float d2 = dx * dx + dy * dy;            // (+ dz*dz with dz = 0 adds nothing)
  return (double)sqrtf(d2) < thre;
}
__global__ void __launch_bounds__(kRnBlock)
remove_neighbors_flag_kernel(const float *__restrict__ base, size_t bstride, int nb, const float *__restrict__ list,
                             size_t lstride, int nl, double thre, unsigned char *__restrict__ keep,
                             int *__restrict__ block_count) {
  __shared__ float2 tile[kRnTile];
  __shared__ int wsum[kRnBlock / 64];
  const int i = blockIdx.x * kRnBlock + threadIdx.x;
  float2 p = make_float2(0.f, 0.f);
  if (i < nb) p = load_pt(base, bstride, (size_t)i);
  bool flag = i < nb;
  for (int t0 = 0; t0 < nl; t0 += kRnTile) {
    const int m = min(kRnTile, nl - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += kRnBlock) tile[j] = load_pt(list, lstride, (size_t)(t0 + j));
    __syncthreads();
    if (flag) {                                   // (the reference keeps testing; the outcome is the same)
      bool near = false;
      for (int j = 0; j < m; ++j) near = near || rn_near(p, tile[j], thre);
      flag = !near;
    }
  }
  if (i < nb) keep[i] = flag ? 1 : 0;
  const unsigned long long b = __ballot(flag);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __builtin_popcountll(b);
  __syncthreads();
  if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < kRnBlock / 64; ++w) s += wsum[w]; block_count[blockIdx.x] = s; }
}
// exclusive scan of the block counts (one workgroup), total to *n_out
__global__ void __launch_bounds__(1024)
remove_neighbors_scan_kernel(int *__restrict__ block_count, int nblocks, unsigned long long *__restrict__ n_out) {
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? block_count[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) block_count[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = (unsigned long long)carry;
}
__global__ void __launch_bounds__(kRnBlock)
remove_neighbors_pack_kernel(const float *__restrict__ base, size_t bstride, int nb, const unsigned char *__restrict__ keep,
                             const int *__restrict__ block_off, float2 *__restrict__ out) {
  __shared__ int wsum[kRnBlock / 64];
  const int i = blockIdx.x * kRnBlock + threadIdx.x;
  const bool flag = i < nb && keep[i];
  const unsigned long long b = __ballot(flag);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wsum[wv] = __builtin_popcountll(b);
  __syncthreads();
  int off = block_off[blockIdx.x];
  for (int w = 0; w < wv; ++w) off += wsum[w];
  if (flag) out[off + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = load_pt(base, bstride, (size_t)i);
}

}  // namespace

// ==========================================================================================
// host side
// ==========================================================================================

struct ndt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;       // the stream all work is ordered on
  hipStream_t own_stream = nullptr;   // created by ndt_ctx_create
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the last match launch
  hipEvent_t evm0 = nullptr, evm1 = nullptr; // around the last map build
  hipEvent_t evb = nullptr;                  // bounding box of the map build read back
  bool map_ms_pending = false;
  unsigned *h_bounds = nullptr;              // pinned: bounding box read-back of the map build
  std::string err;
  float map_ms = 0.f, align_ms = 0.f;
  // grow-only staging for the host-pointer entry points
  void *d_scan = nullptr; size_t d_scan_cap = 0;
  void *d_off = nullptr; size_t d_off_cap = 0;
  void *d_init = nullptr; size_t d_init_cap = 0;
  void *d_res = nullptr; size_t d_res_cap = 0;
  void *d_tmp = nullptr; size_t d_tmp_cap = 0;
  void *d_trace = nullptr; size_t d_trace_cap = 0;
  void *d_rows = nullptr; size_t d_rows_cap = 0;
  void *d_sorted = nullptr; size_t d_sorted_cap = 0;   // per-lane ordered copy of the scans
  void *d_ws = nullptr; size_t d_ws_cap = 0;           // WsHeader + ScanCtl[B] + chunk totals
  void *d_pf = nullptr; size_t d_pf_cap = 0;           // pre-filter: filtered points at the raw offsets + counts
  void *d_rn = nullptr; size_t d_rn_cap = 0;           // neighbour removal: block offsets + keep flags
  int num_cus = 0;
  int helpers = 1;                                     // NDT_NO_HELPERS=1 disables work sharing (diagnostic)
};

struct ndt_map {
  ndt_ctx *ctx = nullptr;
  ndt_params prm;
  ndt_map_info info;
  bool info_valid = false;
  MapView view;
  size_t n = 0, ng = 0, npad = 0;
  // device buffers (grow-only across rebuilds)
  int *count = nullptr, *start = nullptr, *fill = nullptr, *tile = nullptr, *npts_grid = nullptr;
  size_t count_cap = 0, start_cap = 0, fill_cap = 0, tile_cap = 0, npts_cap = 0;
  int *perm = nullptr, *perm_sorted = nullptr; float2 *pts = nullptr;
  size_t perm_cap = 0, perm_sorted_cap = 0, pts_cap = 0;
  float2 *cent = nullptr; double *rec = nullptr; size_t cent_cap = 0, rec_cap = 0;
  unsigned *bounds = nullptr; int *counters = nullptr; int *total = nullptr;   // counters: n_cells, n_valid, n_big
  int *big = nullptr; size_t big_cap = 0;
  unsigned *occ = nullptr; size_t occ_cap = 0;
  GridDims grid; bool have_grid = false;      // voxel grid of the last build (queued ahead of the next one's bounding box)
  void *d_xy_stage = nullptr; size_t d_xy_cap = 0;
};

namespace {

int fail(ndt_ctx *ctx, int code, const std::string &msg) {
  g_last_error = msg;
  if (ctx) ctx->err = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail((ctx), NDT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

int ensure(ndt_ctx *ctx, void **p, size_t *cap, size_t need) {
  if (need <= *cap && *p) return NDT_OK;
  if (*p) { hipError_t e = hipFree(*p); (void)e; *p = nullptr; *cap = 0; }
  size_t want = need + need / 4 + 256;
  HIP_TRY(ctx, hipMalloc(p, want));
  *cap = want;
  return NDT_OK;
}

template <typename T>
int ensure_t(ndt_ctx *ctx, T **p, size_t *cap_elems, size_t need_elems) {
  size_t cap_bytes = *cap_elems * sizeof(T);
  void *vp = *p;
  int rc = ensure(ctx, &vp, &cap_bytes, need_elems * sizeof(T));
  *p = static_cast<T *>(vp);
  *cap_elems = cap_bytes / sizeof(T);
  return rc;
}

OptParams opt_of(const ndt_params &p) {
  OptParams o;
  o.step_size = p.step_size; o.trans_eps = p.trans_eps; o.snap_thresh = p.snap_thresh;
  o.mt_mu = p.mt_mu; o.mt_nu = p.mt_nu; o.max_iter = p.max_iter; o.conv_ge = p.conv_ge;
  o.stale_h_ang = p.stale_h_ang; o.mt_max_iter = p.mt_max_iter;
  return o;
}

// a3: Gaussian constants (Magnusson 2009 eq 6.8), host libm, once per map.
void gauss_constants(const ndt_params &p, double *d1, double *d2) {
  double res = (double)p.resolution;
  double c1 = 10.0 * (1.0 - p.outlier_ratio);
  double c2 = p.outlier_ratio / std::pow(res, 3);
  double d3 = -std::log(c2);
  *d1 = -std::log(c1 + c2) - d3;
  *d2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / *d1);
}

// exp table 2^(j/64), correctly rounded, uploaded once per device
int upload_exp_table(ndt_ctx *ctx);

int grid_for(size_t n, int block, int cap = 2048) {
  size_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > (size_t)cap) g = cap;
  return (int)g;
}

int launch_align(ndt_ctx *ctx, const ndt_map *map, hipStream_t st, const float *scans,
                 const unsigned long long *offsets, int B, int shared_scan, const double *inits, ndt_result *out,
                 double *trace, int trace_cap, int *trace_rows, float2 *sorted, unsigned long long *prof) {
  const bool sse = map->prm.transform_sse != 0, incl = map->prm.radius_inclusive != 0;
  const MapView &V = map->view;
  const OptParams O = opt_of(map->prm);
  // workspace: header + one control line per scan (zeroed every launch) + chunk totals
  const size_t zero_bytes = sizeof(WsHeader) + (size_t)B * sizeof(ScanCtl);
  const size_t ws_bytes = zero_bytes + (size_t)B * kUnits * 12 * sizeof(double) + (size_t)B * (kRegionCells / 8);
  int rc = ensure(ctx, &ctx->d_ws, &ctx->d_ws_cap, ws_bytes);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_ws, 0, zero_bytes, st));
  unsigned char *ws = (unsigned char *)ctx->d_ws;
  const int helpers = ctx->helpers;
  // one workgroup per CU (the LDS window allows no more); idle workgroups help unfinished scans
  const int grid = helpers ? ctx->num_cus : (B < ctx->num_cus ? B : ctx->num_cus);
#define NDT_LAUNCH(S_, I_)                                                                           \
  ndt_align_kernel<S_, I_><<<grid, kBlock, 0, st>>>(V, O, scans, offsets, B, shared_scan, inits, out, trace, \
                                                    trace_cap, trace_rows, sorted, ws, helpers, prof)
  if (sse && incl) NDT_LAUNCH(true, true);
  else if (sse)    NDT_LAUNCH(true, false);
  else if (incl)   NDT_LAUNCH(false, true);
  else             NDT_LAUNCH(false, false);
#undef NDT_LAUNCH
  HIP_TRY(ctx, hipGetLastError());
  return NDT_OK;
}

int upload_exp_table(ndt_ctx *ctx) {
  double tab[64];
  for (int j = 0; j < 64; ++j) tab[j] = (double)exp2l((long double)j / 64.0L);
  HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_exp2_tab), tab, sizeof(tab)));
  return NDT_OK;
}

}  // namespace

extern "C" {

int ndt_default_params(ndt_params *p) {
  if (!p) return NDT_E_ARG;
  memset(p, 0, sizeof(*p));
  p->resolution = 1.0f; p->step_size = 0.1; p->trans_eps = 0.01; p->max_iter = 35;
  p->outlier_ratio = 0.55; p->min_pts = 6; p->eig_mult = 0.01;
  p->cov_unbiased = 0; p->cov_init_identity = 0; p->conv_ge = 0; p->radius_inclusive = 0;
  p->transform_sse = 0; p->stale_h_ang = 1; p->snap_thresh = 10e-5; p->mt_max_iter = 10;
  p->mt_mu = 1.e-4; p->mt_nu = 0.9;
  return NDT_OK;
}

const char *ndt_last_error(const ndt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int ndt_ctx_create(int device, ndt_ctx **out) {
  if (!out) return NDT_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, NDT_E_NO_DEVICE, "no HIP device visible: libndt_mi355x has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(nullptr, NDT_E_ARG, "device ordinal out of range");
  ndt_ctx *c = new (std::nothrow) ndt_ctx();
  if (!c) return NDT_E_NOMEM;
  c->device = device;
  HIP_TRY(c, hipSetDevice(device));
  HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  HIP_TRY(c, hipEventCreate(&c->ev0));
  HIP_TRY(c, hipEventCreate(&c->ev1));
  HIP_TRY(c, hipEventCreate(&c->evm0));
  HIP_TRY(c, hipEventCreate(&c->evm1));
  HIP_TRY(c, hipEventCreateWithFlags(&c->evb, hipEventDisableTiming));
  HIP_TRY(c, hipHostMalloc((void **)&c->h_bounds, 64, hipHostMallocDefault));
  { int rc = upload_exp_table(c); if (rc) return rc; }
  {
    hipDeviceProp_t prop;
    HIP_TRY(c, hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
    c->helpers = getenv("NDT_NO_HELPERS") ? 0 : kMaxHelpers;
    if (getenv("NDT_MAX_HELPERS")) c->helpers = atoi(getenv("NDT_MAX_HELPERS"));
    if (c->helpers > kMaxHelpers) c->helpers = kMaxHelpers;
    if (c->helpers < 0) c->helpers = 0;
    if (getenv("NDT_GRID")) c->num_cus = atoi(getenv("NDT_GRID"));
  }
  *out = c;
  return NDT_OK;
}

int ndt_ctx_destroy(ndt_ctx *c) {
  if (!c) return NDT_E_ARG;
  hipError_t e;
  e = hipSetDevice(c->device);
  e = hipStreamSynchronize(c->stream);
  if (c->own_stream) e = hipStreamDestroy(c->own_stream);
  if (c->ev0) e = hipEventDestroy(c->ev0);
  if (c->ev1) e = hipEventDestroy(c->ev1);
  if (c->evm0) e = hipEventDestroy(c->evm0);
  if (c->evm1) e = hipEventDestroy(c->evm1);
  if (c->evb) e = hipEventDestroy(c->evb);
  if (c->h_bounds) e = hipHostFree(c->h_bounds);
  void *bufs[] = {c->d_scan, c->d_off, c->d_init, c->d_res, c->d_tmp, c->d_trace, c->d_rows, c->d_sorted, c->d_ws, c->d_pf, c->d_rn};
  for (void *b : bufs) if (b) e = hipFree(b);
  (void)e;
  delete c;
  return NDT_OK;
}

void *ndt_ctx_stream(ndt_ctx *c) { return c ? (void *)c->stream : nullptr; }

int ndt_ctx_set_stream(ndt_ctx *c, void *stream) {
  if (!c) return NDT_E_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stream = stream ? (hipStream_t)stream : c->own_stream;
  return NDT_OK;
}

int ndt_last_timing(const ndt_ctx *cc, float *map_ms, float *align_ms) {
  if (!cc) return NDT_E_ARG;
  ndt_ctx *c = const_cast<ndt_ctx *>(cc);
  if (c->map_ms_pending) {                     // the device-pointer build returns before its last kernels end
    HIP_TRY(c, hipEventSynchronize(c->evm1));
    HIP_TRY(c, hipEventElapsedTime(&c->map_ms, c->evm0, c->evm1));
    c->map_ms_pending = false;
  }
  if (map_ms) *map_ms = c->map_ms;
  if (align_ms) *align_ms = c->align_ms;
  return NDT_OK;
}

int ndt_map_destroy(ndt_map *m) {
  if (!m) return NDT_E_ARG;
  hipError_t e = hipSetDevice(m->ctx->device);
  e = hipStreamSynchronize(m->ctx->stream);
  void *bufs[] = {m->occ, m->big, m->count, m->start, m->fill, m->tile, m->npts_grid, m->perm, m->perm_sorted, m->pts,
                  m->cent, m->rec, m->bounds, m->counters, m->total, m->d_xy_stage};
  for (void *b : bufs) if (b) e = hipFree(b);
  (void)e;
  delete m;
  return NDT_OK;
}

// Steps 2-4 of the map build for a given voxel grid: everything after the bounding box, queued on
// the context's stream.
static int queue_build(ndt_ctx *ctx, ndt_map *m, const float *xy, size_t n, size_t stride, const ndt_params *prm,
                       const GridDims &G) {
  hipStream_t st = ctx->stream;
  const float inv_leaf = G.inv_leaf;
  const size_t ng = (size_t)G.div_x * G.div_y, npad = (size_t)G.gw * G.gh;
  m->ng = ng; m->npad = npad;

  // 2. buffers (grow-only across rebuilds)
  {
    int rc;
    const size_t ntiles_ = (ng + kScanTile - 1) / kScanTile;
    if ((rc = ensure_t(ctx, &m->count, &m->count_cap, ng + 1))) return rc;
    if ((rc = ensure_t(ctx, &m->start, &m->start_cap, ng + 1 + 8))) return rc;   // 4 readable ints before, 3 after (nearest_sq)
    if ((rc = ensure_t(ctx, &m->fill, &m->fill_cap, ng + 1))) return rc;
    if ((rc = ensure_t(ctx, &m->npts_grid, &m->npts_cap, ng + 1))) return rc;
    if ((rc = ensure_t(ctx, &m->tile, &m->tile_cap, ntiles_ + 1))) return rc;
    if ((rc = ensure_t(ctx, &m->perm, &m->perm_cap, n))) return rc;
    if ((rc = ensure_t(ctx, &m->perm_sorted, &m->perm_sorted_cap, n))) return rc;
    if ((rc = ensure_t(ctx, &m->big, &m->big_cap, n / kBigVoxel + 1))) return rc;   // voxels with > kBigVoxel points
    if ((rc = ensure_t(ctx, &m->occ, &m->occ_cap, (ng + 31) / 32 + 2))) return rc;
    if ((rc = ensure_t(ctx, &m->pts, &m->pts_cap, n))) return rc;
    if ((rc = ensure_t(ctx, &m->cent, &m->cent_cap, npad))) return rc;
    if ((rc = ensure_t(ctx, &m->rec, &m->rec_cap, npad * 8))) return rc;
  }
  HIP_TRY(ctx, hipMemsetAsync(m->count, 0, (ng + 1) * sizeof(int), st));
  HIP_TRY(ctx, hipMemsetAsync(m->fill, 0, (ng + 1) * sizeof(int), st));
  HIP_TRY(ctx, hipMemsetAsync(m->counters, 0, 4 * sizeof(int), st));
  // (records of voxels outside the search set are never read: no clearing of m->rec)
  fill_f2_kernel<<<grid_for(npad, 256), 256, 0, st>>>(m->cent, npad, INFINITY);

  // 3. bucket the points by voxel, cloud order kept inside a bucket
  map_count_kernel<<<grid_for(n, 256), 256, 0, st>>>(xy, stride, n, G, m->count);
  const int ntiles = (int)((ng + kScanTile - 1) / kScanTile);
  scan_tile_sums_kernel<<<ntiles, kScanBlock, 0, st>>>(m->count, ng, m->tile);
  scan_tile_offsets_kernel<<<1, 1024, 0, st>>>(m->tile, ntiles, m->total);
  HIP_TRY(ctx, hipMemsetAsync(m->start, 0, 4 * sizeof(int), st));
  int *const start = m->start + 4;
  const int big_cap = (int)(n / kBigVoxel + 1);
  scan_apply_kernel<<<ntiles, kScanBlock, 0, st>>>(m->count, ng, m->tile, start, m->total, m->big, m->counters + 2, big_cap);
  map_scatter_kernel<<<grid_for(n, 256), 256, 0, st>>>(xy, stride, n, G, start, m->fill, m->perm);
  map_order_small_kernel<<<(unsigned)((ng + kOrderVoxPerBlock - 1) / kOrderVoxPerBlock), 256, 0, st>>>(start, ng, m->perm, m->perm_sorted);
  map_order_big_kernel<<<kBigBlocks, 256, 0, st>>>(start, m->big, m->counters + 2, big_cap, m->perm, m->perm_sorted);

  // 4. per-voxel statistics -> centroid grid + cell records + bucketed raw points
  LeafParams L;
  L.min_pts = prm->min_pts; L.cov_unbiased = prm->cov_unbiased; L.cov_init_identity = prm->cov_init_identity;
  L.eig_mult = prm->eig_mult;
  map_finalize_kernel<<<(unsigned)((ng + 255) / 256), 256, 0, st>>>(xy, stride, G, L, start, m->perm_sorted,
                                                                          m->pts, m->cent, m->rec, m->npts_grid,
                                                                          m->counters, m->occ);
  HIP_TRY(ctx, hipGetLastError());

  MapView &V = m->view;
  V.inv_leaf = inv_leaf; V.leaf = prm->resolution;
  V.r2 = (float)((double)prm->resolution * (double)prm->resolution);
  V.radius_inclusive = prm->radius_inclusive; V.transform_sse = prm->transform_sse;
  V.min_bx = G.min_bx; V.min_by = G.min_by; V.div_x = G.div_x; V.div_y = G.div_y; V.gw = G.gw; V.gh = G.gh;
  V.cent = m->cent; V.rec = m->rec; V.occ = m->occ; V.pt_start = start; V.pts = m->pts;
  gauss_constants(*prm, &V.d1, &V.d2);
  m->info.min_bx = G.min_bx; m->info.min_by = G.min_by; m->info.div_x = G.div_x; m->info.div_y = G.div_y;
  m->info.n_points = n;
  return NDT_OK;
}

int ndt_map_build_dev(ndt_ctx *ctx, const float *xy, size_t n, size_t stride, const ndt_params *prm,
                      ndt_map **pmap) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!xy || n == 0 || !prm || !pmap || !(prm->resolution > 0) || stride < 8 || (stride & 7))
    return fail(ctx, NDT_E_ARG, "ndt_map_build: bad arguments (need n > 0, resolution > 0, stride % 8 == 0)");
  if (n > (size_t)INT32_MAX) return fail(ctx, NDT_E_ARG, "ndt_map_build: more than 2^31 points");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  ndt_map *m = *pmap;
  if (!m) {
    m = new (std::nothrow) ndt_map();
    if (!m) return NDT_E_NOMEM;
    m->ctx = ctx;
    HIP_TRY(ctx, hipMalloc(&m->bounds, 4 * sizeof(unsigned)));
    HIP_TRY(ctx, hipMalloc(&m->counters, 4 * sizeof(int)));
    HIP_TRY(ctx, hipMalloc(&m->total, sizeof(int)));
    *pmap = m;
  }
  m->prm = *prm; m->n = n; m->info_valid = false;
  HIP_TRY(ctx, hipEventRecord(ctx->evm0, st));

  // 1. bounding box (getMinMax3D).  The grid follows from it on the host; instead of idling the GPU
  // during that round trip, the rest of the build is queued at once with the grid of the previous
  // build of this map (a SLAM local map keeps its voxel bounding box for many scans) and redone
  // only if the read-back disagrees.
  unsigned init_b[4] = {0xffffffffu, 0xffffffffu, 0u, 0u};
  HIP_TRY(ctx, hipMemcpyAsync(m->bounds, init_b, sizeof(init_b), hipMemcpyHostToDevice, st));
  map_minmax_kernel<<<grid_for(n, 256 * 32, 128), 256, 0, st>>>(xy, stride, n, m->bounds);
  unsigned *hb = ctx->h_bounds;                // pinned
  HIP_TRY(ctx, hipMemcpyAsync(hb, m->bounds, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipEventRecord(ctx->evb, st));
  const float inv_leaf = 1.0f / prm->resolution;
  bool queued = false;
  if (m->have_grid && m->grid.inv_leaf == inv_leaf) {
    int rc = queue_build(ctx, m, xy, n, stride, prm, m->grid);
    if (rc) return rc;
    queued = true;
  }
  HIP_TRY(ctx, hipEventSynchronize(ctx->evb));
  if (hb[0] == 0xffffffffu) { m->have_grid = false; return fail(ctx, NDT_E_ARG, "ndt_map_build: no finite points"); }
  const float mnx = ord2f(hb[0]), mny = ord2f(hb[1]), mxx = ord2f(hb[2]), mxy = ord2f(hb[3]);
  GridDims G;
  G.inv_leaf = inv_leaf;
  G.min_bx = (int)floorf(mnx * inv_leaf); G.min_by = (int)floorf(mny * inv_leaf);
  long long dx = (long long)(int)floorf(mxx * inv_leaf) - G.min_bx + 1;
  long long dy = (long long)(int)floorf(mxy * inv_leaf) - G.min_by + 1;
  if (dx * dy > (1LL << 28)) { m->have_grid = false; return fail(ctx, NDT_E_GRID, "ndt_map_build: voxel grid larger than 2^28 cells"); }
  G.div_x = (int)dx; G.div_y = (int)dy; G.gw = G.div_x + 4; G.gh = G.div_y + 4;
  const bool same = queued && G.min_bx == m->grid.min_bx && G.min_by == m->grid.min_by &&
                    G.div_x == m->grid.div_x && G.div_y == m->grid.div_y;
  if (!same) {
    int rc = queue_build(ctx, m, xy, n, stride, prm, G);
    if (rc) { m->have_grid = false; return rc; }
  }
  m->grid = G; m->have_grid = true;
  HIP_TRY(ctx, hipEventRecord(ctx->evm1, st));
  ctx->map_ms_pending = true;
  return NDT_OK;                               // asynchronous from here on (stream order)
}


int ndt_map_build(ndt_ctx *ctx, const float *xy_host, size_t n, size_t stride, const ndt_params *prm,
                  ndt_map **pmap) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!xy_host || n == 0 || !pmap || stride < 8 || (stride & 7)) return fail(ctx, NDT_E_ARG, "ndt_map_build: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // stage through a buffer owned by the map once it exists; first build uses a temporary
  void *stage = nullptr; size_t cap = 0;
  if (*pmap) { stage = (*pmap)->d_xy_stage; cap = (*pmap)->d_xy_cap; }
  int rc = ensure(ctx, &stage, &cap, n * stride);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(stage, xy_host, n * stride, hipMemcpyHostToDevice, ctx->stream));
  rc = ndt_map_build_dev(ctx, (const float *)stage, n, stride, prm, pmap);
  if (rc == NDT_OK) { hipError_t e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) rc = fail(ctx, NDT_E_HIP, hipGetErrorString(e)); }   // host-pointer form: synchronous
  if (*pmap) { (*pmap)->d_xy_stage = stage; (*pmap)->d_xy_cap = cap; }
  else { hipError_t e = hipFree(stage); (void)e; }
  return rc;
}

int ndt_map_info_get(const ndt_map *cm, ndt_map_info *out) {
  if (!cm || !out) return NDT_E_ARG;
  ndt_map *m = const_cast<ndt_map *>(cm);
  if (!m->info_valid) {
    // per-voxel flags: 0 not in the search set, n accepted, -n rejected covariance
    std::vector<int> flags(m->ng);
    HIP_TRY(m->ctx, hipSetDevice(m->ctx->device));
    HIP_TRY(m->ctx, hipMemcpyAsync(flags.data(), m->npts_grid, m->ng * sizeof(int), hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_TRY(m->ctx, hipStreamSynchronize(m->ctx->stream));
    int cells = 0, valid = 0;
    for (int f : flags) { cells += f != 0; valid += f > 0; }
    m->info.n_cells = cells; m->info.n_valid = valid;
    m->info_valid = true;
  }
  *out = m->info;
  return NDT_OK;
}

int ndt_map_export(const ndt_map *cm, int *cell_idx, float *cent_xy, double *mean_xy, double *icov,
                   int *npts) {
  if (!cm || !cell_idx || !cent_xy || !mean_xy || !icov || !npts) return NDT_E_ARG;
  ndt_map *m = const_cast<ndt_map *>(cm);
  ndt_ctx *ctx = m->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t ng = m->ng, npad = m->npad;
  int *hn = (int *)malloc(ng * sizeof(int));
  float2 *hc = (float2 *)malloc(npad * sizeof(float2));
  double *hr = (double *)malloc(npad * 8 * sizeof(double));
  if (!hn || !hc || !hr) { free(hn); free(hc); free(hr); return NDT_E_NOMEM; }
  hipError_t e1 = hipMemcpyAsync(hn, m->npts_grid, ng * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipMemcpyAsync(hc, m->cent, npad * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e3 = hipMemcpyAsync(hr, m->rec, npad * 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e4 = hipStreamSynchronize(ctx->stream);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
    free(hn); free(hc); free(hr);
    return fail(ctx, NDT_E_HIP, "ndt_map_export: copy failed");
  }
  size_t k = 0;
  const int gw = m->view.gw, dx = m->view.div_x;
  for (size_t g = 0; g < ng; ++g) {
    if (hn[g] == 0) continue;
    size_t pg = (size_t)(g / dx + 2) * gw + (g % dx + 2);
    cell_idx[k] = (int)g; npts[k] = hn[g];
    cent_xy[2 * k] = hc[pg].x; cent_xy[2 * k + 1] = hc[pg].y;
    mean_xy[2 * k] = hr[pg * 8]; mean_xy[2 * k + 1] = hr[pg * 8 + 1];
    icov[3 * k] = hr[pg * 8 + 2]; icov[3 * k + 1] = hr[pg * 8 + 3]; icov[3 * k + 2] = hr[pg * 8 + 4];
    ++k;
  }
  free(hn); free(hc); free(hr);
  return NDT_OK;
}

int ndt_align_batch_dev(ndt_ctx *ctx, const ndt_map *map, const float *scans, const uint64_t *offsets,
                        int B, size_t total_points, int shared_scan, const double *inits, ndt_result *out,
                        void *stream) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!map || !scans || !offsets || !inits || !out || B <= 0) return fail(ctx, NDT_E_ARG, "ndt_align_batch: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  if (st != ctx->stream) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->evm1, 0));   // the map build may still be running on the context's stream
  float2 *sorted = nullptr;
  if (total_points > 0) {                      // shared scan: one slot of the scan's size per workgroup
    int rc = ensure(ctx, &ctx->d_sorted, &ctx->d_sorted_cap, (shared_scan ? (size_t)ctx->num_cus : (size_t)1) * total_points * 8);
    if (rc) return rc;
    sorted = (float2 *)ctx->d_sorted;
  }
  return launch_align(ctx, map, st, scans, (const unsigned long long *)offsets, B, shared_scan, inits, out,
                      nullptr, 0, nullptr, sorted, nullptr);
}

int ndt_align_batch_trace(ndt_ctx *ctx, const ndt_map *map, const float *scans, const uint64_t *offsets,
                          int B, int shared_scan, const double *inits, ndt_result *out, double *trace,
                          int trace_cap, int *trace_rows) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!map || !scans || !offsets || !inits || !out || B <= 0) return fail(ctx, NDT_E_ARG, "ndt_align_batch: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t nscan = shared_scan ? 1 : (size_t)B;
  const size_t npts = (size_t)(offsets[nscan] - offsets[0]);
  if (npts == 0) return fail(ctx, NDT_E_ARG, "ndt_align_batch: empty scans");
  for (size_t b = 0; b < nscan; ++b)
    if (offsets[b + 1] < offsets[b]) return fail(ctx, NDT_E_ARG, "ndt_align_batch: offsets not monotone");
  int rc;
  if ((rc = ensure(ctx, &ctx->d_scan, &ctx->d_scan_cap, (size_t)offsets[nscan] * 8))) return rc;
  if ((rc = ensure(ctx, &ctx->d_off, &ctx->d_off_cap, (nscan + 1) * 8))) return rc;
  if ((rc = ensure(ctx, &ctx->d_init, &ctx->d_init_cap, (size_t)B * 24))) return rc;
  if ((rc = ensure(ctx, &ctx->d_res, &ctx->d_res_cap, (size_t)B * sizeof(ndt_result)))) return rc;
  double *d_trace = nullptr; int *d_rows = nullptr;
  if (trace && trace_cap > 0 && trace_rows) {
    if ((rc = ensure(ctx, &ctx->d_trace, &ctx->d_trace_cap, (size_t)B * trace_cap * 64))) return rc;
    if ((rc = ensure(ctx, &ctx->d_rows, &ctx->d_rows_cap, (size_t)B * 4))) return rc;
    d_trace = (double *)ctx->d_trace; d_rows = (int *)ctx->d_rows;
    HIP_TRY(ctx, hipMemsetAsync(d_trace, 0, (size_t)B * trace_cap * 64, st));
  }
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_scan, scans, (size_t)offsets[nscan] * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_off, offsets, (nscan + 1) * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_init, inits, (size_t)B * 24, hipMemcpyHostToDevice, st));
  // diagnostic phase timing (NDT_PROF=1): not part of the ABI, prints to stderr
  unsigned long long *d_prof = nullptr;
  const bool want_prof = getenv("NDT_PROF") != nullptr;
  if (want_prof) { HIP_TRY(ctx, hipMalloc(&d_prof, (size_t)B * 128)); HIP_TRY(ctx, hipMemsetAsync(d_prof, 0, (size_t)B * 128, st)); }
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, st));
  float2 *sorted = nullptr;
  if ((rc = ensure(ctx, &ctx->d_sorted, &ctx->d_sorted_cap, (shared_scan ? (size_t)ctx->num_cus : (size_t)1) * (size_t)offsets[nscan] * 8))) return rc;
  sorted = (float2 *)ctx->d_sorted;
  if ((rc = launch_align(ctx, map, st, (const float *)ctx->d_scan, (const unsigned long long *)ctx->d_off, B,
                         shared_scan, (const double *)ctx->d_init, (ndt_result *)ctx->d_res, d_trace, trace_cap,
                         d_rows, sorted, d_prof)))
    return rc;
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, st));
  if (want_prof) {
    unsigned long long *hp = (unsigned long long *)malloc((size_t)B * 128);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipMemcpy(hp, d_prof, (size_t)B * 128, hipMemcpyDeviceToHost));
    double te = 0, ta = 0, tw = 0, ev = 0, sh = 0, hc = 0, worst = 0;
    for (int b = 0; b < B; ++b) {
      te += hp[8 * b] * 0.01; ta += (hp[8 * b + 1] & 0xFFFFFFFFull) * 0.01; tw += hp[8 * b + 6] * 0.01;
      ev += (double)(hp[8 * b + 3] & 0xFFFF); sh += (double)((hp[8 * b + 3] >> 16) & 0xFFFF); hc += (double)(hp[8 * b + 3] >> 32);
      double tot = (hp[8 * b] + (hp[8 * b + 1] & 0xFFFFFFFFull)) * 0.01;
      if (tot > worst) worst = tot;
    }
    for (int rep = 0; rep < 6 && rep < B; ++rep) {      // the longest scans
      int best = -1; double bt = -1;
      for (int b = 0; b < B; ++b) { double tot = (hp[8 * b] + (hp[8 * b + 1] & 0xFFFFFFFFull)) * 0.01; if (tot > bt && !(hp[8 * b + 3] >> 63)) { bt = tot; best = b; } }
      if (best < 0) break;
      fprintf(stderr, "[NDT_PROF]   scan %3d: %.0f us (fitness pass %.0f us, window spilled %d), passes %llu, shared %llu, helper units %llu, first shared pass at %.0f us (scan started %.0f, first helper attached %.0f, its window ready %.0f)\n", best, bt,
              (double)((hp[8 * best + 1] >> 32) & 0x7FFFFFFFull) * 0.01, (int)(hp[8 * best + 1] >> 63),
              hp[8 * best + 3] & 0xFFFF, (hp[8 * best + 3] >> 16) & 0xFFFF, (hp[8 * best + 3] >> 32) & 0x7FFFFFFF, (double)(hp[8 * best + 2] >> 32) * 0.01, (double)(hp[8 * best + 2] & 0xFFFFFFFFull) * 0.01,
              (double)hp[8 * best + 4] * 0.01, (double)hp[8 * best + 5] * 0.01);
      hp[8 * best + 3] |= 1ull << 63;
    }
    fprintf(stderr, "[NDT_PROF] B=%d passes=%.0f (+fitness) | per pass: compute+combine %.2f us, advance %.2f us | shared passes %.0f, helper chunks %.0f, owner wait %.2f us per shared pass | slowest scan %.1f us\n",
            B, ev, te / (ev + B), ta / ev, sh, hc, sh > 0 ? tw / sh : 0.0, worst);
    if (const char *dump = getenv("NDT_PROF_DUMP")) { FILE *f = fopen(dump, "wb"); if (f) { fwrite(hp, 128, (size_t)B, f); fclose(f); } }
    free(hp);
    hipError_t e = hipFree(d_prof); (void)e;
  }
  HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_res, (size_t)B * sizeof(ndt_result), hipMemcpyDeviceToHost, st));
  if (d_trace) {
    HIP_TRY(ctx, hipMemcpyAsync(trace, d_trace, (size_t)B * trace_cap * 64, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(trace_rows, d_rows, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipEventElapsedTime(&ctx->align_ms, ctx->ev0, ctx->ev1));
  return NDT_OK;
}

int ndt_align_batch(ndt_ctx *ctx, const ndt_map *map, const float *scans, const uint64_t *offsets, int B,
                    int shared_scan, const double *inits, ndt_result *out) {
  return ndt_align_batch_trace(ctx, map, scans, offsets, B, shared_scan, inits, out, nullptr, 0, nullptr);
}

int ndt_align(ndt_ctx *ctx, const ndt_map *map, const float *scan, size_t n, size_t stride,
              const double init[3], ndt_result *out) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!map || !scan || n == 0 || !init || !out || stride < 8) return fail(ctx, NDT_E_ARG, "ndt_align: bad arguments");
  const float *packed = scan;
  float *tmp = nullptr;
  if (stride != 8) {   // repack pcl::PointXYZ-style records to float2
    tmp = (float *)malloc(n * 8);
    if (!tmp) return NDT_E_NOMEM;
    for (size_t i = 0; i < n; ++i) {
      const float *p = (const float *)((const char *)scan + i * stride);
      tmp[2 * i] = p[0]; tmp[2 * i + 1] = p[1];
    }
    packed = tmp;
  }
  uint64_t off[2] = {0, (uint64_t)n};
  int rc = ndt_align_batch(ctx, map, packed, off, 1, 0, init, out);
  free(tmp);
  return rc;
}

int ndt_eval_at(ndt_ctx *ctx, const ndt_map *map, const float *scan, size_t n, size_t stride,
                const double p[3], double *score, double g[3], double H[9], double *pairs) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!map || !scan || n == 0 || !p || stride < 8 || (stride & 7)) return fail(ctx, NDT_E_ARG, "ndt_eval_at: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  const int grid = grid_for(n, 256, 1024);
  if ((rc = ensure(ctx, &ctx->d_scan, &ctx->d_scan_cap, n * stride))) return rc;
  if ((rc = ensure(ctx, &ctx->d_tmp, &ctx->d_tmp_cap, (size_t)grid * kAcc * 8))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_scan, scan, n * stride, hipMemcpyHostToDevice, st));
  {
    const bool sse = map->prm.transform_sse != 0, incl = map->prm.radius_inclusive != 0;
    const MapView &V = map->view; const double sn = map->prm.snap_thresh;
    const float *ds = (const float *)ctx->d_scan; double *dt = (double *)ctx->d_tmp;
    if (sse && incl)       ndt_eval_kernel<true, true><<<grid, 256, 0, st>>>(V, sn, ds, stride, (int)n, p[0], p[1], p[2], dt);
    else if (sse)          ndt_eval_kernel<true, false><<<grid, 256, 0, st>>>(V, sn, ds, stride, (int)n, p[0], p[1], p[2], dt);
    else if (incl)         ndt_eval_kernel<false, true><<<grid, 256, 0, st>>>(V, sn, ds, stride, (int)n, p[0], p[1], p[2], dt);
    else                   ndt_eval_kernel<false, false><<<grid, 256, 0, st>>>(V, sn, ds, stride, (int)n, p[0], p[1], p[2], dt);
  }
  HIP_TRY(ctx, hipGetLastError());
  double *hp = (double *)malloc((size_t)grid * kAcc * 8);
  if (!hp) return NDT_E_NOMEM;
  hipError_t e = hipMemcpyAsync(hp, ctx->d_tmp, (size_t)grid * kAcc * 8, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { free(hp); return fail(ctx, NDT_E_HIP, hipGetErrorString(e)); }
  double t[kAcc] = {0};
  for (int b = 0; b < grid; ++b) for (int k = 0; k < kAcc; ++k) t[k] += hp[b * kAcc + k];
  free(hp);
  const double w = map->view.d1 * map->view.d2;
  if (score) *score = -map->view.d1 * t[0];
  if (g) { g[0] = w * t[1]; g[1] = w * t[2]; g[2] = w * t[3]; }
  if (H) {
    H[0] = w * t[4]; H[1] = H[3] = w * t[5]; H[2] = H[6] = w * t[6];
    H[4] = w * t[7]; H[5] = H[7] = w * t[8]; H[8] = w * t[9];
  }
  if (pairs) *pairs = t[10];
  return NDT_OK;
}

int ndt_fitness_at(ndt_ctx *ctx, const ndt_map *map, const float *scan, size_t n, size_t stride,
                   float c, float s, float tx, float ty, double *fitness) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!map || !scan || n == 0 || !fitness || stride < 8 || (stride & 7)) return fail(ctx, NDT_E_ARG, "ndt_fitness_at: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  const int grid = grid_for(n, 256, 1024);
  if ((rc = ensure(ctx, &ctx->d_scan, &ctx->d_scan_cap, n * stride))) return rc;
  if ((rc = ensure(ctx, &ctx->d_tmp, &ctx->d_tmp_cap, (size_t)grid * 16))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_scan, scan, n * stride, hipMemcpyHostToDevice, st));
  Tf32 T = {c, s, tx, ty};
  ndt_fitness_kernel<<<grid, 256, 0, st>>>(map->view, (const float *)ctx->d_scan, stride, (int)n, T,
                                           (double *)ctx->d_tmp);
  HIP_TRY(ctx, hipGetLastError());
  double *hp = (double *)malloc((size_t)grid * 16);
  if (!hp) return NDT_E_NOMEM;
  hipError_t e = hipMemcpyAsync(hp, ctx->d_tmp, (size_t)grid * 16, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { free(hp); return fail(ctx, NDT_E_HIP, hipGetErrorString(e)); }
  double sum = 0, cnt = 0;
  for (int b = 0; b < grid; ++b) { sum += hp[2 * b]; cnt += hp[2 * b + 1]; }
  free(hp);
  *fitness = cnt > 0 ? sum / cnt : DBL_MAX;
  return NDT_OK;
}

int ndt_prefilter_batch_dev(ndt_ctx *ctx, const float *raw_xy, size_t stride, const uint64_t *raw_offsets, int B,
                            size_t total_raw_points, float leaf, float *out_xy, uint64_t *out_offsets, void *stream) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!raw_xy || !raw_offsets || !out_xy || !out_offsets || B <= 0 || total_raw_points == 0 || !(leaf > 0) ||
      stride < 8 || (stride & 7))
    return fail(ctx, NDT_E_ARG, "ndt_prefilter_batch: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  int rc;
  // filtered points at the raw offsets, then the per-scan counts
  const size_t tmp_bytes = total_raw_points * sizeof(float2);
  if ((rc = ensure(ctx, &ctx->d_pf, &ctx->d_pf_cap, tmp_bytes + (size_t)B * sizeof(unsigned)))) return rc;
  float2 *tmp = (float2 *)ctx->d_pf;
  unsigned *counts = (unsigned *)((char *)ctx->d_pf + tmp_bytes);
  const int grid = B < 8 * ctx->num_cus ? B : 8 * ctx->num_cus;
  prefilter_kernel<<<grid, 64, 0, st>>>(raw_xy, stride, (const unsigned long long *)raw_offsets, B, leaf, tmp, counts);
  prefilter_offsets_kernel<<<1, 1024, 0, st>>>(counts, B, (unsigned long long *)out_offsets);
  const int gx = (int)std::min<size_t>(64, (total_raw_points / (size_t)B + 255) / 256 + 1);
  prefilter_pack_kernel<<<dim3((unsigned)gx, (unsigned)std::min(B, 65535)), 256, 0, st>>>(
      tmp, (const unsigned long long *)raw_offsets, (const unsigned long long *)out_offsets, B, (float2 *)out_xy);
  HIP_TRY(ctx, hipGetLastError());
  return NDT_OK;
}

int ndt_prefilter(ndt_ctx *ctx, const float *xy_host, size_t n, size_t stride, float leaf, float *out_xy_host,
                  size_t *n_out) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!xy_host || n == 0 || !out_xy_host || !n_out || stride < 8 || (stride & 7) || !(leaf > 0))
    return fail(ctx, NDT_E_ARG, "ndt_prefilter: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  if ((rc = ensure(ctx, &ctx->d_scan, &ctx->d_scan_cap, n * stride + n * sizeof(float2)))) return rc;
  if ((rc = ensure(ctx, &ctx->d_off, &ctx->d_off_cap, 4 * sizeof(uint64_t)))) return rc;
  float *d_in = (float *)ctx->d_scan;
  float *d_out = (float *)((char *)ctx->d_scan + n * stride);
  uint64_t *d_offs = (uint64_t *)ctx->d_off;                  // [0..1] raw, [2..3] filtered
  const uint64_t raw[2] = {0, (uint64_t)n};
  HIP_TRY(ctx, hipMemcpyAsync(d_in, xy_host, n * stride, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_offs, raw, sizeof(raw), hipMemcpyHostToDevice, st));
  if ((rc = ndt_prefilter_batch_dev(ctx, d_in, stride, d_offs, 1, n, leaf, d_out, d_offs + 2, st))) return rc;
  uint64_t fo[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(fo, d_offs + 2, sizeof(fo), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  *n_out = (size_t)fo[1];
  HIP_TRY(ctx, hipMemcpyAsync(out_xy_host, d_out, (size_t)fo[1] * sizeof(float2), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return NDT_OK;
}

int ndt_fuse_default_params(ndt_fuse_params *p) {
  if (!p) return NDT_E_ARG;
  p->coe_ndt_cov = 1.0; p->coe_vel = 0.1; p->coe_omega = 0.1; p->del_time = 0.5; p->score_thre = 0.0;
  return NDT_OK;
}

int ndt_predict_batch_dev(ndt_ctx *ctx, const double *odo_cur, const double *odo_prev, const double *last_pose, int B,
                          double *odo_motion, double *pred_pose, double *init_xyyaw, void *stream) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!odo_cur || !odo_prev || !last_pose || !odo_motion || !pred_pose || B <= 0)
    return fail(ctx, NDT_E_ARG, "ndt_predict_batch: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  predict_kernel<<<(B + 255) / 256, 256, 0, st>>>(odo_cur, odo_prev, last_pose, B, odo_motion, pred_pose, init_xyyaw);
  HIP_TRY(ctx, hipGetLastError());
  return NDT_OK;
}

int ndt_fuse_batch_dev(ndt_ctx *ctx, const ndt_result *results, const double *pred_pose, const double *odo_motion,
                       const double *last_pose, const double *last_cov, int B, const ndt_fuse_params *prm,
                       double *fused_pose, double *cov, int *successful, void *stream) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!results || !pred_pose || !odo_motion || !last_pose || !last_cov || !prm || !fused_pose || !cov || B <= 0 ||
      !(prm->del_time > 0))
    return fail(ctx, NDT_E_ARG, "ndt_fuse_batch: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  FuseParams P = {prm->coe_ndt_cov, prm->coe_vel, prm->coe_omega, prm->del_time, prm->score_thre};
  fuse_kernel<<<(B + 255) / 256, 256, 0, st>>>(results, pred_pose, odo_motion, last_pose, last_cov, B, P, fused_pose,
                                               cov, successful);
  HIP_TRY(ctx, hipGetLastError());
  return NDT_OK;
}

int ndt_remove_neighbors_dev(ndt_ctx *ctx, const float *base_xy, size_t base_stride, size_t n_base, const float *list_xy,
                             size_t list_stride, size_t n_list, double thre_neighbor, float *out_xy, uint64_t *n_out,
                             void *stream) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!base_xy || n_base == 0 || n_base > (size_t)INT32_MAX || n_list > (size_t)INT32_MAX || (n_list && !list_xy) ||
      !out_xy || !n_out || base_stride < 8 || (base_stride & 7) || (n_list && (list_stride < 8 || (list_stride & 7))))
    return fail(ctx, NDT_E_ARG, "ndt_remove_neighbors: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const int nblocks = (int)((n_base + kRnBlock - 1) / kRnBlock);
  int rc;
  if ((rc = ensure(ctx, &ctx->d_rn, &ctx->d_rn_cap, n_base + (size_t)nblocks * sizeof(int) + 16))) return rc;
  int *block_count = (int *)ctx->d_rn;
  unsigned char *keep = (unsigned char *)ctx->d_rn + (size_t)nblocks * sizeof(int);
  remove_neighbors_flag_kernel<<<nblocks, kRnBlock, 0, st>>>(base_xy, base_stride, (int)n_base, list_xy, list_stride,
                                                             (int)n_list, thre_neighbor, keep, block_count);
  remove_neighbors_scan_kernel<<<1, 1024, 0, st>>>(block_count, nblocks, (unsigned long long *)n_out);
  remove_neighbors_pack_kernel<<<nblocks, kRnBlock, 0, st>>>(base_xy, base_stride, (int)n_base, keep, block_count,
                                                             (float2 *)out_xy);
  HIP_TRY(ctx, hipGetLastError());
  return NDT_OK;
}

int ndt_remove_neighbors(ndt_ctx *ctx, const float *base_xy_host, size_t base_stride, size_t n_base,
                         const float *list_xy_host, size_t list_stride, size_t n_list, double thre_neighbor,
                         float *out_xy_host, size_t *n_out) {
  if (!ctx) return fail(nullptr, NDT_E_ARG, "null context");
  if (!base_xy_host || n_base == 0 || !out_xy_host || !n_out || (n_list && !list_xy_host))
    return fail(ctx, NDT_E_ARG, "ndt_remove_neighbors: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  const size_t bb = n_base * base_stride, lb = n_list * list_stride, ob = n_base * sizeof(float2);
  if ((rc = ensure(ctx, &ctx->d_scan, &ctx->d_scan_cap, bb + lb + ob + 64))) return rc;
  char *d = (char *)ctx->d_scan;
  float *d_base = (float *)d, *d_list = (float *)(d + bb), *d_out = (float *)(d + bb + ((lb + 15) & ~(size_t)15));
  if ((rc = ensure(ctx, &ctx->d_off, &ctx->d_off_cap, 4 * sizeof(uint64_t)))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(d_base, base_xy_host, bb, hipMemcpyHostToDevice, st));
  if (n_list) HIP_TRY(ctx, hipMemcpyAsync(d_list, list_xy_host, lb, hipMemcpyHostToDevice, st));
  if ((rc = ndt_remove_neighbors_dev(ctx, d_base, base_stride, n_base, n_list ? d_list : nullptr, n_list ? list_stride : 8,
                                     n_list, thre_neighbor, d_out, (uint64_t *)ctx->d_off, st)))
    return rc;
  uint64_t cnt = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&cnt, ctx->d_off, sizeof(cnt), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  *n_out = (size_t)cnt;
  HIP_TRY(ctx, hipMemcpyAsync(out_xy_host, d_out, (size_t)cnt * sizeof(float2), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return NDT_OK;
}

}  // extern "C"
