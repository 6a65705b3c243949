// ndt_point.hip.h -- rows a4 + a5: one source point -> its in-radius voxels -> score / gradient / Hessian terms.
// Part of libndt_mi355x.so: included by ndt_mi355x.hip inside its anonymous namespace (one translation
// unit; the order of the includes matters).  Not a standalone header.

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
