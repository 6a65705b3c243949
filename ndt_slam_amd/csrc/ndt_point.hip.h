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
#ifdef NDT_EXP_RINT
  const double t = rint(x * 92.332482616893657);            // 64 / ln 2
  const int n = (int)t;
#else
  // t = the integer nearest to x * 64 / ln 2, by adding and taking away 1.5 * 2^52 (|x * 64 / ln 2| < 2^17): the sum's low word IS
  // that integer (two's complement), so neither a rounding nor a conversion instruction is needed (round 5: one instruction
  // fewer per pair; the product is rounded once, inside the fma, where rint (x * c) rounded twice -- the two differ only when
  // x * c lies within an ulp of a half-integer, and then by one table step whose result agrees to the last bit or two)
  const double big = 6755399441055744.0;
  const double sum = __builtin_fma(x, 92.332482616893657, big);
  const double t = sum - big;
  const int n = (int)(unsigned)__double_as_longlong(sum);
#endif
  const double sc = tab[n & 63];
#ifndef NDT_NO_EXP_SCHED
  __builtin_amdgcn_sched_barrier(0);                        // issue the table read BEFORE the polynomial (the scheduler put it behind: a full LDS latency exposed per pair)
#endif
  double r = __builtin_fma(-t, 0x1.62e42fefa0000p-7, x);    // ln2/64, high part (exact product)
  r = __builtin_fma(-t, 0x1.cf79abc9e3b3ap-46, r);          // low part
#ifdef NDT_EXP_PLAIN
  double p = __builtin_fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = __builtin_fma(r, p, 1.0 / 6.0);
#else
  // (the same two fused multiply-adds, spelled as three-address v_fma_f64: left to itself the compiler keeps 1/24 and 1/6 in
  //  registers that share their low word and turns each step into v_mov + v_fmac -- three extra instructions per pair)
  double p, c4 = 1.0 / 24.0, c3 = 1.0 / 6.0, c5 = 1.0 / 120.0;
  //  (1/120 and 1/6 as scalar operands -- one per instruction is allowed -- 1/24 in a vector register: 1/24 and 1/6 in vector
  //   registers share their low word in the compiler's hands, which costs a v_mov per pair to put 1/6 together)
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(r), "s"(c5), "v"(c4));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(r), "v"(p), "s"(c3));
#endif
  p = __builtin_fma(r, p, 0.5);
  p = __builtin_fma(p, r * r, r);                           // exp(r) - 1
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
  const double2 a = gld_d2(rec);
  const double2 b = gld_d2(rec + 2);
  CellRec c; c.mx = a.x; c.my = a.y; c.i00 = b.x; c.i01 = b.y; c.i11 = gld_d(rec + 4);
  return c;
}

// Sums over the in-radius voxels of ONE point.  With J = dT/dp = [e_x, e_y, j] (j the yaw column, a function of
// the point only) everything a pair adds to the gradient and the Hessian is a product of point-only terms with
//   se = sum e,   a = sum e u,   B = sum e (Sigma^-1 - d2 u u^T),   u = Sigma^-1 q,  e = exp(-d2/2 q^T u)
// (eqs 6.12 / 6.13 restricted to (tx, ty, yaw): g = [a, a.j], H = [[B, B j], [., j^T B j + a.h]]), so the pair loop
// only accumulates these six numbers and the point's contribution is formed once, after the loop.
struct PointAcc { double se, a0, a1, b00, b01, b11; };

// one (point, voxel) pair
// updateDerivatives' error check `d2 e > 1 || d2 e < 0 || NaN` as one comparison: e >= 0 always (exp_neg), so for d2 > 0
// the pair is dropped exactly when e > e_hi, e_hi = the largest double with fl(d2 * e_hi) <= 1 (found on the host,
// MapView::e_hi; -1 when d2 < 0: every pair with e > 0 is dropped, +inf when d2 is 0 or NaN).
// CHK = false (round 5): the check left out where it cannot fire.  The map build hands out inverse covariances that are positive
// semi-definite by construction (closed-form eigen-decomposition, eigenvalues floored at eig_mult x the larger one; a rejected
// voxel has Sigma^-1 = 0), so m = q^T Sigma^-1 q >= -(rounding), e = exp(-d2 m / 2) <= 1 + 1e-12 for d2 > 0 -- and the launch
// picks CHK = false only when e_hi > 1 + 1e-6 (d2 < 1 - 1e-6: the 0.5 m preset has d2 = 0.756, e_hi = 1.32).  A comparison and
// two selects fewer in a 56-instruction loop that runs 3.7 times per point.
template <bool CHK = true>
__device__ __forceinline__ void accumulate_pair(double e_hi, double nd2, double nd2h, const double *__restrict__ etab,
                                                double XT, double YT, const CellRec &c, PointAcc &S) {
  const double q0 = XT - c.mx, q1 = YT - c.my;
  const double u0 = __builtin_fma(c.i01, q1, c.i00 * q0);      // Sigma^-1 q
  const double u1 = __builtin_fma(c.i11, q1, c.i01 * q0);
  const double m = __builtin_fma(q1, u1, q0 * u0);
  double e = exp_neg(nd2h * m, etab);                          // (-d2 m) / 2, the halving is exact
  if (CHK && e > e_hi) e = 0.0;                                // updateDerivatives error check
  const double v0 = nd2 * u0, v1 = nd2 * u1;
  S.se += e;
  S.a0 = __builtin_fma(e, u0, S.a0);
  S.a1 = __builtin_fma(e, u1, S.a1);
  S.b00 = __builtin_fma(e, __builtin_fma(v0, u0, c.i00), S.b00);
  S.b01 = __builtin_fma(e, __builtin_fma(v0, u1, c.i01), S.b01);
  S.b11 = __builtin_fma(e, __builtin_fma(v1, u1, c.i11), S.b11);
}

// the point's terms: yaw column of J_E and the (yaw,yaw) block of H_E (untransformed coordinates)
__device__ __forceinline__ void finish_point(float x, float y, double cj, double sj, double ch, double sh,
                                             const PointAcc &S, Acc &A) {
  const double X = (double)x, Y = (double)y;
  const double jx = X * (-sj) + Y * (-cj), jy = X * cj + Y * (-sj);
  const double hx = X * (-ch) + Y * sh, hy = X * (-sh) + Y * (-ch);
  const double bx = __builtin_fma(S.b01, jy, S.b00 * jx);      // B j
  const double by = __builtin_fma(S.b11, jy, S.b01 * jx);
  double tt = __builtin_fma(jx, bx, jy * by);                  // j^T B j
  tt = __builtin_fma(S.a0, hx, tt);                            // + a . d2T/dyaw2
  tt = __builtin_fma(S.a1, hy, tt);
  A.e += S.se;
  A.g0 += S.a0; A.g1 += S.a1;
  A.g2 += __builtin_fma(S.a1, jy, S.a0 * jx);
  A.hxx += S.b00; A.hxy += S.b01; A.hyy += S.b11;
  A.hxt += bx; A.hyt += by;
  A.htt += tt;
}

// Everything one source point contributes to a derivative pass.
// Fast path (window holds every occupied voxel, point's 3x3 neighbourhood inside it): slot
// numbers, centroids and records all come from LDS.  Otherwise the same arithmetic reads the
// global centroid grid / record array.
template <bool SSE, bool INCL, bool CHK = true>
__device__ __forceinline__ void eval_point(const MapView &M, const Window &W,
                                           const double *__restrict__ etab, const Tf32 &T, float x,
                                           float y, double cj, double sj, double ch, double sh, Acc &A) {
  float xt, yt;
  tf_apply_t<SSE>(T, x, y, xt, yt);
  const float fx = fminf(fmaxf(floorf(xt * M.inv_leaf), -1.0e9f), 1.0e9f);     // (NaN -> -1e9: outside every window and grid)
  const float fy = fminf(fmaxf(floorf(yt * M.inv_leaf), -1.0e9f), 1.0e9f);
  const Region &R = W.R;
  // window coordinates.  Window cells outside the map's grid carry the "outside the search set" slot (fill_window) and a
  // non-finite point lands at -1e9, outside every window: one unsigned comparison per axis decides the fast path.
  const int lx = (int)fx - (M.min_bx + R.x0), ly = (int)fy - (M.min_by + R.y0);
  const bool inwin = ((unsigned)(lx - 1) < (unsigned)max(R.rw - 2, 0)) & ((unsigned)(ly - 1) < (unsigned)max(R.rh - 2, 0));
  // LDS probes with clamped indices (results dropped when !inwin)
  const int clx = min(max(lx, 1), max(R.rw - 2, 1)), cly = min(max(ly, 1), max(R.rh - 2, 1));
  const unsigned short *srow = W.slot + (__mul24(cly - 1, R.rw) + (clx - 1));
  const int rw3 = R.rw - 3;                    // slot of neighbour k = 3 r + q: srow[r * rw + q] = srow[r * (rw - 3) + k]
  const int rw3x2 = 2 * rw3;                   // (in bytes)
  unsigned mask = 0;
  float lowx = INFINITY;                       // -inf <=> one of the nine voxels is occupied but not resident
  // (the nine slot numbers first, then the nine centroids: left to itself the scheduler waited for the first slot before
  //  it issued the other eight reads -- a whole LDS round trip exposed per point)
  unsigned short sl[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) sl[r * 3 + q] = srow[r * R.rw + q];
#ifndef NDT_NO_PROBE_SCHED
  __builtin_amdgcn_sched_barrier(0);
#endif
  float2 cc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) cc[k] = W.ent[sl[k]].cent;
#ifndef NDT_NO_PROBE_SCHED
  __builtin_amdgcn_sched_barrier(0);
#endif
  // (-inf centroids -- occupied voxels without an LDS record -- exist only in a window that spilled: uniform)
  if (R.nspill > 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) lowx = fminf(lowx, cc[k].x);
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) mask |= in_radius<INCL>(M.r2, xt, yt, cc[k]) << k;
  const double nd2 = -M.d2, nd2h = nd2 * 0.5;
  if (inwin & (lowx != -INFINITY)) {
    if (!mask) return;
    A.pairs += __builtin_popcount(mask);
    const double XT = (double)xt, YT = (double)yt;
    PointAcc S = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma nounroll
    do {
      const int k = __builtin_ctz(mask);
      mask &= mask - 1;
      const unsigned k2 = 2u * (unsigned)k;
      const int r = (int)__builtin_amdgcn_ubfe(0x2A540u, k2, 2u);                  // k / 3 for k in [0, 9): two bits per k, one v_bfe_u32
      const CellEntry &E = W.ent[*reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(srow) + (__mul24(r, rw3x2) + (int)k2))];
      CellRec c; c.mx = E.mx; c.my = E.my; c.i00 = E.i00; c.i01 = E.i01; c.i11 = E.i11;
      accumulate_pair<CHK>(M.e_hi, nd2, nd2h, etab, XT, YT, c, S);
    } while (mask);
    finish_point(x, y, cj, sj, ch, sh, S, A);
    return;
  }
  // slow path: global centroid grid and record array
  const int ix = lx + R.x0, iy = ly + R.y0;
  const bool ingrid = finite2(xt, yt) & (ix >= -1) & (ix <= M.div_x) & (iy >= -1) & (iy <= M.div_y);
  if (!ingrid) return;
  const size_t base = (size_t)(iy + 1) * M.gw + (ix + 1);     // padded coords of (ix-1, iy-1)
  const float2 *grow = M.cent + base;
  mask = 0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) mask |= in_radius<INCL>(M.r2, xt, yt, gld_f2(grow + (r * M.gw + q))) << (r * 3 + q);
  if (!mask) return;
  A.pairs += __builtin_popcount(mask);
  const double XT = (double)xt, YT = (double)yt;
  PointAcc S = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma nounroll
  do {
    const int k = __builtin_ctz(mask);
    mask &= mask - 1;
    accumulate_pair<CHK>(M.e_hi, nd2, nd2h, etab, XT, YT, load_rec_global(M, base, k), S);
  } while (mask);
  finish_point(x, y, cj, sj, ch, sh, S, A);
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
