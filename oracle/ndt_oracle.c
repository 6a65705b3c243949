/*
 * ndt_oracle.c -- CPU restatement (C99) of the reference's NDT hot path.  See ndt_oracle.h:
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (PCL absent; no golden vectors in the reference).
 *
 * Each function cites the reference call site it stands behind (paths under /root/reference)
 * and the SURVEY.md 8a row whose PCL semantics it restates.  The 6-DoF problem is restated
 * in its exact SE(2) reduction (SURVEY.md 8a note: every point has z = 0,
 * include/ndt_slam/PoseEstimator.h:100, so the (z, roll, pitch) block is inert).
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: sums must round as the scalar
 * reference code does).
 */
#include "ndt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* parameters                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* PoseEstimator.h:63-64 constructor values (overridden by ndt_mapping.launch:32-36 in the caller) and
 * the PCL-side defaults every version shares. */
static void params_common(ndt_oracle_params *p) {
  memset(p, 0, sizeof(*p));
  p->resolution = 1.0f;   /* PoseEstimator.h:64 */
  p->step_size = 0.1;     /* PoseEstimator.h:64 */
  p->trans_eps = 0.01;    /* PoseEstimator.h:64 */
  p->max_iter = 35;       /* PoseEstimator.h:64 */
  p->outlier_ratio = 0.55;
  p->min_pts = 6;
  p->eig_mult = 0.01;
  p->conv_ge = 0;
  p->radius_inclusive = 0;
  p->stale_h_ang = 0;    /* computeDerivatives calls computeAngleDerivatives(p) with its default compute_hessian = true */
  p->snap_thresh = 10e-5;
  p->mt_max_iter = 10;
  p->mt_mu = 1.e-4;
  p->mt_nu = 0.9;
  p->grid_margin = 0;
}

/* Presets of the version-sensitive switches.  preset 0 = PCL 1.9/1.10 (the default: the
 * reference only compiles against PCL <= 1.10, PoseEstimator.h:72-73): VoxelGridCovariance::Leaf() starts
 * cov_ at the identity, (n-1)/n normalisation, SSE transformPointCloud; 1 = PCL <= 1.8: the same with the
 * scalar transformPointCloud; 2 = PCL >= 1.11: cov_ starts at zero, /(n-1). */
void ndt_oracle_params_preset(ndt_oracle_params *p, int preset) {
  params_common(p);
  if (preset == 2) { p->cov_unbiased = 1; p->cov_init_identity = 0; p->transform_sse = 1; p->libm_f32 = 1; }
  else if (preset == 1) { p->cov_unbiased = 0; p->cov_init_identity = 1; p->transform_sse = 0; p->libm_f32 = 0; }   /* glibc 2.27: another sinf */
  else { p->cov_unbiased = 0; p->cov_init_identity = 1; p->transform_sse = 1; p->libm_f32 = 1; }
}

void ndt_oracle_default_params(ndt_oracle_params *p) { ndt_oracle_params_preset(p, 0); }

/* a3: Gaussian fitting constants, Magnusson 2009 eq 6.8, recomputed at the top of
 * computeTransformation (called from src/PoseEstimator.cpp:28 ndt.align). */
void ndt_oracle_gauss(const ndt_oracle_params *prm, double *d1, double *d2) {
  double res = (double)prm->resolution;
  double c1 = 10.0 * (1.0 - prm->outlier_ratio);
  double c2 = prm->outlier_ratio / pow(res, 3);
  double d3 = -log(c2);
  *d1 = -log(c1 + c2) - d3;
  *d2 = -2.0 * log((-log(c1 * exp(-0.5) + c2) - d3) / *d1);
}

/* ------------------------------------------------------------------------------------------ */
/* map (a2)                                                                                    */
/* ------------------------------------------------------------------------------------------ */

struct ndt_oracle_map {
  ndt_oracle_params prm;
  float inv_leaf;         /* 1.0f / leaf, float32 as VoxelGrid::setLeafSize computes it */
  float leaf;
  int min_bx, min_by, div_x, div_y;
  size_t n_points;
  /* dense grid */
  int *cell_id;           /* div_x*div_y: compact id of voxels in the centroid search set, else -1 */
  int *pt_start;          /* div_x*div_y + 1: bucket offsets of raw points (input order kept)    */
  float *pts;             /* 2*n_points raw xy bucketed by voxel (for a7)                        */
  /* compact cell table, ascending dense index */
  int n_cells, n_valid;
  int *c_idx;
  float *c_cent;          /* 2 per cell: float32 centroid (the kd-tree point)                    */
  double *c_mean;         /* 2 per cell                                                          */
  double *c_icov;         /* 3 per cell: xx, xy, yy                                              */
  int *c_npts;            /* >0 accepted, <0 rejected covariance                                 */
  double d1, d2;
  float r2;               /* float32 squared search radius                                       */
};

static inline const float *pt_at(const float *xy, size_t stride, size_t i) {
  return (const float *)((const char *)xy + i * stride);
}

/* float32 voxel coordinate as VoxelGridCovariance::applyFilter computes it:
 * static_cast<int>(floor(x * inverse_leaf_size)) with float operands. */
static inline int vox_coord(float x, float inv_leaf) { return (int)floorf(x * inv_leaf); }

void ndt_oracle_map_destroy(ndt_oracle_map *m) {
  if (!m) return;
  free(m->cell_id); free(m->pt_start); free(m->pts);
  free(m->c_idx); free(m->c_cent); free(m->c_mean); free(m->c_icov); free(m->c_npts);
  free(m);
}

/* Per-voxel statistics -> mean, regularised covariance, inverse covariance.
 * Restates the second loop of VoxelGridCovariance::applyFilter for a z = 0 voxel
 * (SURVEY.md 8a row a2).  sx,sy: fp64 point sums; sxx,sxy,syy: fp64 sums of products
 * (plus the identity when cov_init_identity).  Returns 1 accepted, 0 rejected. */
static int leaf_finalize(const ndt_oracle_params *prm, int n, double sx, double sy, double sxx,
                         double sxy, double syy, double szz, double mean[2], double icov[3]) {
  double dn = (double)n;
  double mx = sx / dn, my = sy / dn;
  mean[0] = mx; mean[1] = my;
  icov[0] = icov[1] = icov[2] = 0.0;
  double cxx, cxy, cyy, czz;
  if (!prm->cov_unbiased) {
    /* cov = (Sxx - 2 (sum mu^T)) / n + mu mu^T ; cov *= (n-1)/n */
    cxx = (sxx - 2.0 * (sx * mx)) / dn + mx * mx;
    /* pt_sum * mean^T is not symmetric in floating point; SelfAdjointEigenSolver reads the LOWER triangle,
     * entry (1,0) = (Syx - 2 (sum_y mu_x)) / n + mu_y mu_x */
    cxy = (sxy - 2.0 * (sy * mx)) / dn + my * mx;
    cyy = (syy - 2.0 * (sy * my)) / dn + my * my;
    czz = szz / dn;
    double f = (dn - 1.0) / dn;
    cxx *= f; cxy *= f; cyy *= f; czz *= f;
  } else {
    cxx = (sxx - sx * mx) / (dn - 1.0);
    cxy = (sxy - sy * mx) / (dn - 1.0);     /* lower-triangle entry, as above */
    cyy = (syy - sy * my) / (dn - 1.0);
    czz = szz / (dn - 1.0);
  }
  /* symmetric eigen-decomposition; the z eigenpair is exactly (czz, e_z) because the
   * covariance has an exactly zero z row/column (SURVEY.md 8a note).  2x2 block closed form. */
  double hd = 0.5 * (cxx - cyy);
  double tr = 0.5 * (cxx + cyy);
  double rad = sqrt(hd * hd + cxy * cxy);
  double l1 = tr - rad, l2 = tr + rad; /* l1 <= l2 */
  /* eigenvector of l2 */
  double vx, vy;
  if (rad == 0.0) { vx = 1.0; vy = 0.0; }
  else if (hd >= 0.0) { vx = hd + rad; vy = cxy; }
  else { vx = cxy; vy = rad - hd; }
  double vn = sqrt(vx * vx + vy * vy);
  if (vn == 0.0) { vx = 1.0; vy = 0.0; } else { vx /= vn; vy /= vn; }
  /* v2 = (vx,vy) for l2; v1 = (-vy,vx) for l1 */
  /* ascending sort of {l1, l2, czz} as SelfAdjointEigenSolver returns them */
  double ev[3]; int kind[3]; /* kind: 0 = l1, 1 = l2, 2 = z */
  ev[0] = l1; kind[0] = 0; ev[1] = l2; kind[1] = 1; ev[2] = czz; kind[2] = 2;
  for (int a = 1; a < 3; ++a) {         /* insertion sort, z sorts first among equals */
    double e = ev[a]; int k = kind[a]; int b = a - 1;
    while (b >= 0 && (ev[b] > e || (ev[b] == e && k == 2))) { ev[b + 1] = ev[b]; kind[b + 1] = kind[b]; --b; }
    ev[b + 1] = e; kind[b + 1] = k;
  }
  if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) return 0; /* rejected: stays searchable, icov = 0 */
  double thr = prm->eig_mult * ev[2];
  int rebuilt = 0;
  if (ev[0] < thr) {                      /* nested inflation quirk kept */
    ev[0] = thr;
    if (ev[1] < thr) ev[1] = thr;
    rebuilt = 1;
  }
  double n1 = l1, n2 = l2;
  for (int a = 0; a < 3; ++a) { if (kind[a] == 0) n1 = ev[a]; else if (kind[a] == 1) n2 = ev[a]; }
  if (rebuilt) {                          /* cov = V diag V^-1 (V orthonormal) */
    cxx = n1 * (vy * vy) + n2 * (vx * vx);
    cxy = -n1 * (vx * vy) + n2 * (vx * vy);
    cyy = n1 * (vx * vx) + n2 * (vy * vy);
  }
  double det = cxx * cyy - cxy * cxy;
  icov[0] = cyy / det;
  icov[1] = -cxy / det;
  icov[2] = cxx / det;
  /* icov inf check: voxel flagged rejected but keeps the inf entries (PCL leaves icov_ as is) */
  for (int a = 0; a < 3; ++a)
    if (icov[a] == (double)INFINITY || icov[a] == -(double)INFINITY) return -1;
  return 1;
}

ndt_oracle_map *ndt_oracle_map_build(const float *xy, size_t n, size_t stride,
                                     const ndt_oracle_params *prm) {
  if (!xy || n == 0 || !prm || !(prm->resolution > 0)) return NULL;
  ndt_oracle_map *m = (ndt_oracle_map *)calloc(1, sizeof(*m));
  m->prm = *prm;
  m->leaf = prm->resolution;
  m->inv_leaf = 1.0f / prm->resolution;
  m->n_points = n;
  ndt_oracle_gauss(prm, &m->d1, &m->d2);
  m->r2 = (float)((double)prm->resolution * (double)prm->resolution);

  /* getMinMax3D in float, then min_b = floor(min * inv_leaf) */
  float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
  size_t nfin = 0;
  for (size_t i = 0; i < n; ++i) {
    const float *p = pt_at(xy, stride, i);
    if (!isfinite(p[0]) || !isfinite(p[1])) continue;
    if (p[0] < mnx) mnx = p[0];
    if (p[0] > mxx) mxx = p[0];
    if (p[1] < mny) mny = p[1];
    if (p[1] > mxy) mxy = p[1];
    ++nfin;
  }
  if (nfin == 0) { free(m); return NULL; }
  m->min_bx = (int)floorf(mnx * m->inv_leaf);
  m->min_by = (int)floorf(mny * m->inv_leaf);
  int max_bx = (int)floorf(mxx * m->inv_leaf);
  int max_by = (int)floorf(mxy * m->inv_leaf);
  long long dx = (long long)max_bx - m->min_bx + 1, dy = (long long)max_by - m->min_by + 1;
  if (dx * dy > (1LL << 28)) { free(m); return NULL; } /* PCL: "leaf size too small", no grid */
  m->div_x = (int)dx; m->div_y = (int)dy;
  size_t ng = (size_t)(dx * dy);

  /* bucket raw points by voxel keeping input order (counting sort is stable) */
  m->pt_start = (int *)calloc(ng + 1, sizeof(int));
  int *vox = (int *)malloc(n * sizeof(int));
  for (size_t i = 0; i < n; ++i) {
    const float *p = pt_at(xy, stride, i);
    if (!isfinite(p[0]) || !isfinite(p[1])) { vox[i] = -1; continue; }
    int ix = vox_coord(p[0], m->inv_leaf) - m->min_bx;
    int iy = vox_coord(p[1], m->inv_leaf) - m->min_by;
    vox[i] = iy * m->div_x + ix;
    m->pt_start[vox[i] + 1]++;
  }
  for (size_t g = 0; g < ng; ++g) m->pt_start[g + 1] += m->pt_start[g];
  m->pts = (float *)malloc(2 * (nfin ? nfin : 1) * sizeof(float));
  int *fill = (int *)malloc(ng * sizeof(int));
  memcpy(fill, m->pt_start, ng * sizeof(int));
  for (size_t i = 0; i < n; ++i) {
    if (vox[i] < 0) continue;
    const float *p = pt_at(xy, stride, i);
    int s = fill[vox[i]]++;
    m->pts[2 * s] = p[0]; m->pts[2 * s + 1] = p[1];
  }
  free(fill); free(vox);

  /* per-voxel statistics, sequential in input order (float32 centroid, fp64 mean/cov) */
  m->cell_id = (int *)malloc(ng * sizeof(int));
  int nc = 0;
  for (size_t g = 0; g < ng; ++g) {
    int cnt = m->pt_start[g + 1] - m->pt_start[g];
    if (cnt >= prm->min_pts) m->cell_id[g] = nc++; else m->cell_id[g] = -1;
  }
  m->n_cells = nc;
  m->c_idx = (int *)malloc((nc ? nc : 1) * sizeof(int));
  m->c_cent = (float *)malloc(2 * (nc ? nc : 1) * sizeof(float));
  m->c_mean = (double *)malloc(2 * (nc ? nc : 1) * sizeof(double));
  m->c_icov = (double *)malloc(3 * (nc ? nc : 1) * sizeof(double));
  m->c_npts = (int *)malloc((nc ? nc : 1) * sizeof(int));
  m->n_valid = 0;
  for (size_t g = 0; g < ng; ++g) {
    int c = m->cell_id[g];
    if (c < 0) continue;
    int s0 = m->pt_start[g], s1 = m->pt_start[g + 1];
    float fx = 0.f, fy = 0.f;
    double sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0, szz = 0;
    if (prm->cov_init_identity) { sxx = 1.0; syy = 1.0; szz = 1.0; }
    for (int s = s0; s < s1; ++s) {
      float x = m->pts[2 * s], y = m->pts[2 * s + 1];
      fx += x; fy += y;                       /* float32 centroid accumulator */
      double X = (double)x, Y = (double)y;
      sx += X; sy += Y;
      sxx += X * X; sxy += X * Y; syy += Y * Y;
    }
    int cnt = s1 - s0;
    m->c_idx[c] = (int)g;
    m->c_cent[2 * c] = fx / (float)cnt;
    m->c_cent[2 * c + 1] = fy / (float)cnt;
    int ok = leaf_finalize(prm, cnt, sx, sy, sxx, sxy, syy, szz, &m->c_mean[2 * c], &m->c_icov[3 * c]);
    m->c_npts[c] = (ok > 0) ? cnt : -cnt;
    if (ok > 0) m->n_valid++;
  }
  return m;
}

void ndt_oracle_map_info_get(const ndt_oracle_map *m, ndt_oracle_map_info *o) {
  o->min_bx = m->min_bx; o->min_by = m->min_by; o->div_x = m->div_x; o->div_y = m->div_y;
  o->n_cells = m->n_cells; o->n_valid = m->n_valid; o->n_points = m->n_points;
}

void ndt_oracle_map_export(const ndt_oracle_map *m, int *cell_idx, float *cent, double *mean,
                           double *icov, int *npts) {
  memcpy(cell_idx, m->c_idx, m->n_cells * sizeof(int));
  memcpy(cent, m->c_cent, 2 * m->n_cells * sizeof(float));
  memcpy(mean, m->c_mean, 2 * m->n_cells * sizeof(double));
  memcpy(icov, m->c_icov, 3 * m->n_cells * sizeof(double));
  memcpy(npts, m->c_npts, m->n_cells * sizeof(int));
}

/* ------------------------------------------------------------------------------------------ */
/* a4: pcl::transformPointCloud with the float32 matrix [[c,-s,0,tx],[s,c,0,ty],...]           */
/* ------------------------------------------------------------------------------------------ */

typedef struct { float c, s, tx, ty; } tf32;

/* float32 matrix from the fp64 parameter vector, as computeStepLengthMT builds
 * final_transformation_ = Translation3f(float(p0),float(p1),0) * AngleAxisf(float(p2), Z).
 * std::cos/std::sin on a float argument: libm's cosf / sinf (libm_f32 = 1) or modelled as correctly rounded (0). */
static tf32 tf_from_p(const ndt_oracle_params *prm, const double p[3]) {
  tf32 t;
  float yaw = (float)p[2];
  if (prm->libm_f32) {            /* the platform's own float functions: what std::cos / std::sin (float) are on the reference's machine */
    t.c = cosf(yaw);
    t.s = sinf(yaw);
  } else {
    t.c = (float)cos((double)yaw);
    t.s = (float)sin((double)yaw);
  }
  t.tx = (float)p[0];
  t.ty = (float)p[1];
  return t;
}

static inline void tf_apply(const ndt_oracle_params *prm, tf32 t, float x, float y, float *ox,
                            float *oy) {
  float ms = -t.s;
  if (!prm->transform_sse) {
    float a = t.c * x, b = ms * y; float r = a + b; *ox = r + t.tx;
    float c = t.s * x, d = t.c * y; float q = c + d; *oy = q + t.ty;
  } else {
    float a = t.c * x, b = ms * y; float r = b + t.tx; *ox = a + r;
    float c = t.s * x, d = t.c * y; float q = d + t.ty; *oy = c + q;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* a5: computeDerivatives / updateDerivatives / computeHessian (HOT LOOP A)                    */
/* ------------------------------------------------------------------------------------------ */

typedef struct { double cj, sj, ch, sh; } angle_terms;

/* computeAngleDerivatives: small-angle snap in J and h only. */
static void angle_cs(const ndt_oracle_params *prm, double yaw, double *c, double *s) {
  if (fabs(yaw) < prm->snap_thresh) { *c = 1.0; *s = 0.0; }
  else { *c = cos(yaw); *s = sin(yaw); }
}

/* One pass over the scan.  trans: 2n transformed float32 coordinates; src: untransformed.
 * mode 0: score+gradient ; 1: score+gradient+Hessian ; 2: Hessian only. */
static double eval_pass(const ndt_oracle_map *m, const float *src, size_t n, size_t stride,
                        const float *trans, angle_terms at, int mode, double g[3], double H[6],
                        double *pairs) {
  const ndt_oracle_params *prm = &m->prm;
  const double d1 = m->d1, d2 = m->d2;
  double score = 0.0;
  double gg[3] = {0, 0, 0}, hh[6] = {0, 0, 0, 0, 0, 0};
  double npairs = 0;
  for (size_t i = 0; i < n; ++i) {
    float xt = trans[2 * i], yt = trans[2 * i + 1];
    if (!isfinite(xt) || !isfinite(yt)) continue;
    int ix = vox_coord(xt, m->inv_leaf) - m->min_bx;
    int iy = vox_coord(yt, m->inv_leaf) - m->min_by;
    /* radius search over voxel centroids, r = resolution: subset of the 3x3 neighbourhood */
    int cand[9]; float cd2[9]; int nc = 0;
    for (int dy = -1; dy <= 1; ++dy) {
      int yy = iy + dy; if (yy < 0 || yy >= m->div_y) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        int xx = ix + dx; if (xx < 0 || xx >= m->div_x) continue;
        int c = m->cell_id[(size_t)yy * m->div_x + xx];
        if (c < 0) continue;
        float ex = xt - m->c_cent[2 * c], ey = yt - m->c_cent[2 * c + 1];
        float dd = 0.f; dd += ex * ex; dd += ey * ey;   /* flann::L2_Simple<float> */
        int in = prm->radius_inclusive ? (dd <= m->r2) : (dd < m->r2);
        if (!in) continue;
        int k = nc++;                                   /* keep sorted by (distance, id) */
        while (k > 0 && (cd2[k - 1] > dd || (cd2[k - 1] == dd && cand[k - 1] > c))) {
          cd2[k] = cd2[k - 1]; cand[k] = cand[k - 1]; --k;
        }
        cd2[k] = dd; cand[k] = c;
      }
    }
    if (nc == 0) continue;
    const float *sp = pt_at(src, stride, i);
    double x = (double)sp[0], y = (double)sp[1];
    /* computePointDerivatives: yaw column of J_E and the (yaw,yaw) block of H_E */
    double jx = x * (-at.sj) + y * (-at.cj);
    double jy = x * at.cj + y * (-at.sj);
    double hx = x * (-at.ch) + y * at.sh;
    double hy = x * (-at.sh) + y * (-at.ch);
    for (int k = 0; k < nc; ++k) {
      int c = cand[k];
      npairs += 1.0;
      double q0 = (double)xt - m->c_mean[2 * c], q1 = (double)yt - m->c_mean[2 * c + 1];
      double i00 = m->c_icov[3 * c], i01 = m->c_icov[3 * c + 1], i11 = m->c_icov[3 * c + 2];
      double u0 = i00 * q0 + i01 * q1, u1 = i01 * q0 + i11 * q1;
      double e = exp(-d2 * (q0 * u0 + q1 * u1) / 2);
      double score_inc = -d1 * e;
      e = d2 * e;
      if (e > 1 || e < 0 || e != e) continue;           /* updateDerivatives error check */
      e *= d1;
      if (mode != 2) score += score_inc;
      /* Sigma^-1 * dT/dp_i for i = tx, ty, yaw */
      double ctx0 = i00, ctx1 = i01, cty0 = i01, cty1 = i11;
      double ctt0 = i00 * jx + i01 * jy, ctt1 = i01 * jx + i11 * jy;
      double ax = q0 * ctx0 + q1 * ctx1;
      double ay = q0 * cty0 + q1 * cty1;
      double atq = q0 * ctt0 + q1 * ctt1;
      if (mode != 2) { gg[0] += ax * e; gg[1] += ay * e; gg[2] += atq * e; }
      if (mode != 0) {
        double qh = q0 * (i00 * hx + i01 * hy) + q1 * (i01 * hx + i11 * hy);
        hh[0] += e * (-d2 * ax * ax + ctx0);                         /* xx */
        hh[1] += e * (-d2 * ax * ay + ctx1);                         /* xy */
        hh[2] += e * (-d2 * ax * atq + (jx * ctx0 + jy * ctx1));     /* xt */
        hh[3] += e * (-d2 * ay * ay + cty1);                         /* yy */
        hh[4] += e * (-d2 * ay * atq + (jx * cty0 + jy * cty1));     /* yt */
        hh[5] += e * (-d2 * atq * atq + qh + (jx * ctt0 + jy * ctt1)); /* tt */
      }
    }
  }
  if (mode != 2 && g) { g[0] = gg[0]; g[1] = gg[1]; g[2] = gg[2]; }
  if (mode != 0 && H) memcpy(H, hh, sizeof(hh));
  if (mode == 0 && H) memset(H, 0, 6 * sizeof(double));
  if (pairs) *pairs += npairs;
  return score;
}

static void transform_scan(const ndt_oracle_params *prm, const float *src, size_t n, size_t stride,
                           tf32 t, float *out) {
  for (size_t i = 0; i < n; ++i) {
    const float *p = pt_at(src, stride, i);
    tf_apply(prm, t, p[0], p[1], &out[2 * i], &out[2 * i + 1]);
  }
}

double ndt_oracle_eval_at(const ndt_oracle_map *m, const float *scan, size_t n, size_t stride,
                          const double p[3], double g[3], double H[9], double *pairs_out) {
  float *tr = (float *)malloc(2 * (n ? n : 1) * sizeof(float));
  transform_scan(&m->prm, scan, n, stride, tf_from_p(&m->prm, p), tr);
  angle_terms at; angle_cs(&m->prm, p[2], &at.cj, &at.sj); at.ch = at.cj; at.sh = at.sj;
  double H6[6], pr = 0;
  double s = eval_pass(m, scan, n, stride, tr, at, 1, g, H6, &pr);
  if (H) { H[0] = H6[0]; H[1] = H[3] = H6[1]; H[2] = H[6] = H6[2]; H[4] = H6[3]; H[5] = H[7] = H6[4]; H[8] = H6[5]; }
  if (pairs_out) *pairs_out = pr;
  free(tr);
  return s;
}

/* ------------------------------------------------------------------------------------------ */
/* a6: Newton step + More-Thuente line search                                                  */
/* ------------------------------------------------------------------------------------------ */

/* Symmetric 3x3 solve by cyclic Jacobi eigen-decomposition with pseudo-inverse thresholding.
 * Stands in for PCL's JacobiSVD<6x6>(H).solve(-g): for a symmetric matrix the SVD is the
 * eigen-decomposition up to signs and the 6x6 is block diagonal (SURVEY.md 8a note). */
void ndt_oracle_solve3(const double Hin[9], const double b[3], double x[3]) {
  double A[3][3], V[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[i][j] = 0.5 * (Hin[3 * i + j] + Hin[3 * j + i]); V[i][j] = (i == j); }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) if (A[i][j] != A[i][j]) { x[0] = x[1] = x[2] = NAN; return; }
  {
    /* well-conditioned case: adjugate / determinant (same answer as the SVD solve up to rounding) */
    double a00 = A[0][0], a01 = A[0][1], a02 = A[0][2], a11 = A[1][1], a12 = A[1][2], a22 = A[2][2];
    double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    double det = a00 * c00 + a01 * c01 + a02 * c02;
    double sc = fmax(fmax(fabs(a00), fabs(a11)), fmax(fabs(a22), fmax(fabs(a01), fmax(fabs(a02), fabs(a12)))));
    if (fabs(det) > 1e-9 * sc * sc * sc && fabs(det) <= DBL_MAX) {
      x[0] = (c00 * b[0] + c01 * b[1] + c02 * b[2]) / det;
      x[1] = (c01 * b[0] + c11 * b[1] + c12 * b[2]) / det;
      x[2] = (c02 * b[0] + c12 * b[1] + c22 * b[2]) / det;
      return;
    }
  }
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off == 0.0) break;
    for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
      double apq = A[p][q];
      if (apq == 0.0) continue;
      double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
      double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      double app = A[p][p], aqq = A[q][q];
      A[p][p] = app - t * apq; A[q][q] = aqq + t * apq; A[p][q] = A[q][p] = 0.0;
      int r = 3 - p - q;
      double arp = A[r][p], arq = A[r][q];
      A[r][p] = A[p][r] = c * arp - s * arq;
      A[r][q] = A[q][r] = s * arp + c * arq;
      for (int k = 0; k < 3; ++k) {
        double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
      }
    }
  }
  double lmax = fmax(fabs(A[0][0]), fmax(fabs(A[1][1]), fabs(A[2][2])));
  double thr = lmax * (6.0 * DBL_EPSILON);   /* JacobiSVD default threshold: diagSize * eps */
  x[0] = x[1] = x[2] = 0.0;
  for (int k = 0; k < 3; ++k) {
    double l = A[k][k];
    if (!(fabs(l) > thr) || fabs(l) < DBL_MIN) continue;
    double proj = (V[0][k] * b[0] + V[1][k] * b[1] + V[2][k] * b[2]) / l;
    x[0] += V[0][k] * proj; x[1] += V[1][k] * proj; x[2] += V[2][k] * proj;
  }
}

/* trialValueSelectionMT: More-Thuente trial value, cases 1-4 with the Sun & Yuan
 * cubic/quadratic/secant minimisers and the 0.66 safeguard. */
double ndt_oracle_mt_trial(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u,
                           double a_t, double f_t, double g_t) {
  if (f_t > f_l) {                                          /* case 1 */
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    if (fabs(a_c - a_l) < fabs(a_q - a_l)) return a_c;
    return 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {                               /* case 2 */
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    if (fabs(a_c - a_t) >= fabs(a_s - a_t)) return a_c;
    return a_s;
  } else if (fabs(g_t) <= fabs(g_l)) {                      /* case 3 */
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double a_n = (fabs(a_c - a_t) < fabs(a_s - a_t)) ? a_c : a_s;
    double lim = a_t + 0.66 * (a_u - a_t);
    if (a_t > a_l) return (a_n < lim) ? a_n : lim;          /* std::min(lim, a_n) */
    return (lim < a_n) ? a_n : lim;                         /* std::max(lim, a_n) */
  } else {                                                  /* case 4 */
    double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
    double w = sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
  }
}

/* updateIntervalMT: cases U1-U3 / a-c; returns 1 when the interval has converged. */
int ndt_oracle_mt_update(double *a_l, double *f_l, double *g_l, double *a_u, double *f_u,
                         double *g_u, double a_t, double f_t, double g_t) {
  if (f_t > *f_l) { *a_u = a_t; *f_u = f_t; *g_u = g_t; return 0; }
  if (g_t * (*a_l - a_t) > 0) { *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  if (g_t * (*a_l - a_t) < 0) {
    *a_u = *a_l; *f_u = *f_l; *g_u = *g_l;
    *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0;
  }
  return 1;
}


/* ------------------------------------------------------------------------------------------ */
/* pin hooks (tests only): let a test substitute the reference's vendored Eigen, compiled in   */
/* the build container (tests/golden/make_eigen_golden.cpp), for the hand restatements below,  */
/* so that a whole match can be replayed with the Eigen-exact step.  Not thread safe.          */
/* ------------------------------------------------------------------------------------------ */
static ndt_oracle_hooks g_hooks = {0, 0};
void ndt_oracle_set_hooks(const ndt_oracle_hooks *h) {
  if (h) g_hooks = *h; else { g_hooks.solve = 0; g_hooks.init_p = 0; }
}
static void newton_solve(const double H[9], const double b[3], double x[3]) {
  if (g_hooks.solve) g_hooks.solve(H, b, x); else ndt_oracle_solve3(H, b, x);
}

typedef struct {
  const ndt_oracle_map *m;
  const float *scan; size_t n, stride;
  float *trans;
  tf32 T;                 /* final_transformation_ */
  angle_terms at;         /* j_ang (cj,sj) and h_ang (ch,sh) member state */
  int evals, ref_evals;
  double pairs;
  int evals_run;          /* derivative passes the DEVICE path runs: passes with a gradient, minus the trials that repeat the */
  double pairs_run;       /*   step length of the pass before them (ndt_optimizer.hip.h: advance); their (point, voxel) pairs   */
  double *trace; int trace_cap, trace_n;
} align_ctx;

/* ndt_oracle_set_memoise(1): a line-search trial at the step length of the pass just run re-uses that pass's totals, as the
 * device path does (same x_t, same float32 matrix, same cloud: operation for operation the same pass).  Default 0: every pass
 * the reference runs is run.  Either way evals_run / pairs_run count the passes the device runs.  Not thread safe to flip. */
static int g_memoise = 0;
void ndt_oracle_set_memoise(int on) { g_memoise = on ? 1 : 0; }

static void trace_push(align_ctx *cx, double a_t, double score, const double g[3], const double p[3]) {
  if (!cx->trace || cx->trace_n >= cx->trace_cap) { cx->trace_n++; return; }
  double *t = cx->trace + 8 * (size_t)cx->trace_n++;
  t[0] = a_t; t[1] = score; t[2] = g[0]; t[3] = g[1]; t[4] = g[2]; t[5] = p[0]; t[6] = p[1]; t[7] = p[2];
}

/* computeDerivatives(score_gradient, hessian, trans_cloud, p, compute_hessian) */
static double derivatives(align_ctx *cx, const double p[3], int with_hessian, double g[3], double H[6]) {
  angle_cs(&cx->m->prm, p[2], &cx->at.cj, &cx->at.sj);
  if (with_hessian || !cx->m->prm.stale_h_ang) { cx->at.ch = cx->at.cj; cx->at.sh = cx->at.sj; }
  cx->evals++; cx->ref_evals++;
  return eval_pass(cx->m, cx->scan, cx->n, cx->stride, cx->trans, cx->at, with_hessian ? 1 : 0, g, H, &cx->pairs);
}
/* the same, counted as a pass the device path runs too */
static double derivatives_run(align_ctx *cx, const double p[3], int with_hessian, double g[3], double H[6]) {
  const double before = cx->pairs;
  const double s = derivatives(cx, p, with_hessian, g, H);
  cx->evals_run++; cx->pairs_run += cx->pairs - before;
  return s;
}

/* computeStepLengthMT */
static double step_length_mt(align_ctx *cx, const double x[3], double dir[3], double step_init,
                             double step_max, double step_min, double *score, double g[3], double H[6]) {
  const ndt_oracle_params *prm = &cx->m->prm;
  double phi_0 = -(*score);
  double d_phi_0 = -(g[0] * dir[0] + g[1] * dir[1] + g[2] * dir[2]);
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) return 0;
    d_phi_0 *= -1; dir[0] *= -1; dir[1] *= -1; dir[2] *= -1;
  }
  const double mu = prm->mt_mu, nu = prm->mt_nu;
  int step_iterations = 0;
  double a_l = 0, a_u = 0;
  double f_l = phi_0 - phi_0 - mu * d_phi_0 * a_l;   /* auxilaryFunction_PsiMT(a_l, phi_0, ...) */
  double g_l = d_phi_0 - mu * d_phi_0;               /* auxilaryFunction_dPsiMT               */
  double f_u = phi_0 - phi_0 - mu * d_phi_0 * a_u;
  double g_u = d_phi_0 - mu * d_phi_0;
  int interval_converged = (step_max - step_min) < 0, open_interval = 1;
  double a_t = step_init;
  a_t = (step_max < a_t) ? step_max : a_t;           /* std::min(a_t, step_max) */
  a_t = (a_t < step_min) ? step_min : a_t;           /* std::max(a_t, step_min) */
  double x_t[3] = {x[0] + dir[0] * a_t, x[1] + dir[1] * a_t, x[2] + dir[2] * a_t};
  cx->T = tf_from_p(prm, x_t);
  transform_scan(prm, cx->scan, cx->n, cx->stride, cx->T, cx->trans);
  *score = derivatives_run(cx, x_t, 1, g, H);
  trace_push(cx, a_t, *score, g, x_t);
  double phi_t = -(*score);
  double d_phi_t = -(g[0] * dir[0] + g[1] * dir[1] + g[2] * dir[2]);
  double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t;
  double d_psi_t = d_phi_t - mu * d_phi_0;
  while (!interval_converged && step_iterations < prm->mt_max_iter &&
         !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    const double a_prev = a_t;                             /* step length of the pass just run */
    if (open_interval) a_t = ndt_oracle_mt_trial(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
    else               a_t = ndt_oracle_mt_trial(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    a_t = (step_max < a_t) ? step_max : a_t;
    a_t = (a_t < step_min) ? step_min : a_t;
    x_t[0] = x[0] + dir[0] * a_t; x_t[1] = x[1] + dir[1] * a_t; x_t[2] = x[2] + dir[2] * a_t;
    if (a_t == a_prev) {
      /* the trial of the pass just run, again (the clamp at step_min does this up to mt_max_iter times): the reference
       * transforms the cloud with the same matrix and computes the same derivatives */
      if (g_memoise) cx->ref_evals++;                       /* counted, not run: score and g are that pass's */
      else {
        cx->T = tf_from_p(prm, x_t);
        transform_scan(prm, cx->scan, cx->n, cx->stride, cx->T, cx->trans);
        *score = derivatives(cx, x_t, 0, g, H);
      }
    } else {
      cx->T = tf_from_p(prm, x_t);
      transform_scan(prm, cx->scan, cx->n, cx->stride, cx->T, cx->trans);
      *score = derivatives_run(cx, x_t, 0, g, H);
    }
    trace_push(cx, a_t, *score, g, x_t);
    phi_t = -(*score);
    d_phi_t = -(g[0] * dir[0] + g[1] * dir[1] + g[2] * dir[2]);
    psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t;
    d_psi_t = d_phi_t - mu * d_phi_0;
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      open_interval = 0;
      f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
      f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
    }
    if (open_interval) interval_converged = ndt_oracle_mt_update(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, psi_t, d_psi_t);
    else               interval_converged = ndt_oracle_mt_update(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, phi_t, d_phi_t);
    step_iterations++;
  }
  if (step_iterations) {      /* computeHessian(hessian, trans_cloud, x_t): Hessian-only pass */
    cx->evals++; cx->ref_evals++;
    eval_pass(cx->m, cx->scan, cx->n, cx->stride, cx->trans, cx->at, 2, NULL, H, &cx->pairs);
  }
  return a_t;
}

/* a9: src/PoseEstimator.cpp:31-35 -- std::asin/std::acos on the float32 matrix entries
 * (float overloads, modelled as correctly rounded), four sign branches kept verbatim in intent. */
double ndt_oracle_yaw_from_T(float T00, float T10) {
  double theta;
  if (T00 > 0 && T10 > 0) theta = (double)(float)asin((double)T10);
  else if (T00 > 0 && T10 < 0) theta = (double)(float)asin((double)T10);
  else if (T00 < 0 && T10 > 0) theta = (double)(float)acos((double)T00);
  else theta = (double)(float)acos((double)T00) * (-1.0);
  return theta;
}

/* a7: Registration::getFitnessScore() -- exact 1-NN over ALL raw target points via ring
 * search on the voxel buckets; float32 squared distances, fp64 mean. */
static double fitness_pass(const ndt_oracle_map *m, const float *scan, size_t n, size_t stride, tf32 T) {
  double sum = 0; long nr = 0;
  const double L = (double)m->leaf;
  for (size_t i = 0; i < n; ++i) {
    const float *p = pt_at(scan, stride, i);
    float qx, qy; tf_apply(&m->prm, T, p[0], p[1], &qx, &qy);
    if (!isfinite(qx) || !isfinite(qy)) continue;
    int cx = vox_coord(qx, m->inv_leaf) - m->min_bx, cy = vox_coord(qy, m->inv_leaf) - m->min_by;
    if (cx < 0) cx = 0;
    if (cx >= m->div_x) cx = m->div_x - 1;
    if (cy < 0) cy = 0;
    if (cy >= m->div_y) cy = m->div_y - 1;
    float best = INFINITY;
    int rmax = (m->div_x > m->div_y ? m->div_x : m->div_y);
    for (int r = 0; r <= rmax; ++r) {
      int y0 = cy - r, y1 = cy + r, x0 = cx - r, x1 = cx + r;
      for (int yy = y0; yy <= y1; ++yy) {
        if (yy < 0 || yy >= m->div_y) continue;
        int stepx = (yy == y0 || yy == y1) ? 1 : (x1 - x0 > 0 ? x1 - x0 : 1);
        for (int xx = x0; xx <= x1; xx += stepx) {
          if (xx < 0 || xx >= m->div_x) continue;
          size_t g = (size_t)yy * m->div_x + xx;
          for (int s = m->pt_start[g]; s < m->pt_start[g + 1]; ++s) {
            float ex = qx - m->pts[2 * s], ey = qy - m->pts[2 * s + 1];
            float dd = 0.f; dd += ex * ex; dd += ey * ey;
            if (dd < best) best = dd;
          }
        }
      }
      /* every unvisited point is farther than r*L from the query */
      double bound = (double)r * L * 0.999;
      if ((double)best <= bound * bound) break;
    }
    if (best < INFINITY) { sum += (double)best; nr++; }
  }
  return nr > 0 ? sum / (double)nr : DBL_MAX;
}

double ndt_oracle_fitness(const ndt_oracle_map *m, const float *scan, size_t n, size_t stride,
                          float c, float s, float tx, float ty) {
  tf32 T = {c, s, tx, ty};
  return fitness_pass(m, scan, n, stride, T);
}

/* ------------------------------------------------------------------------------------------ */
/* The initial yaw as Eigen computes it (libm_f32 = 1).  computeTransformation's prologue reads */
/* the angles back from the guess matrix: Affine3f.rotation().eulerAngles(0, 1, 2).             */
/* rotation() of an AFFINE transform is not its linear part: computeRotationScaling runs         */
/* JacobiSVD<Matrix3f>(linear, FullU | FullV), x = det(U V^T), U.col(0) /= x, R = U V^T          */
/* (include/Eigen/src/Geometry/Transform.h:1088-1121, SVD/JacobiSVD.h:663-790,                   */
/* misc/RealSvd2x2.h:19-49, Jacobi/Jacobi.h:94-125 and :54-59) -- restated here in float32,      */
/* operation for operation, for the linear part [[c,-s,0],[s,c,0],[0,0,(1-c)+c]] of a z rotation */
/* (3-term products as a0 + (a1 + a2): Eigen's unrolled reduction, Core/Redux.h:99-113), then    */
/* eulerAngles(0,1,2) (Geometry/EulerAngles.h:87-107) with the platform's atan2f / sinf / cosf.  */
/* Equal to the vendored Eigen's answer on 2e7 yaws incl. the +-90 / +-180 degree strata         */
/* (tests/golden/make_eigen_golden.py) and on every case of tests/golden/eigen_golden.npz.       */
/* ------------------------------------------------------------------------------------------ */
static void eig_apply_rot(float *x, int ix, float *y, int iy, int n, float c, float s) {
  if (c == 1.0f && s == 0.0f) return;
  for (int i = 0; i < n; ++i) {
    float xi = x[i * ix], yi = y[i * iy];
    float a = c * xi, b = s * yi, d = -s * xi, e = c * yi;
    x[i * ix] = a + b; y[i * iy] = d + e;
  }
}
static void eigen_rotation_z(float c, float s, float m22, float R[9]) {
  float W[9] = {c, -s, 0, s, c, 0, 0, 0, m22};
  float U[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  float scale = 0;
  for (int i = 0; i < 9; ++i) { float a = fabsf(W[i]); if (a > scale) scale = a; }
  if (scale == 0) scale = 1;
  for (int i = 0; i < 9; ++i) W[i] = W[i] / scale;
  const float precision = 2 * FLT_EPSILON, tiny = FLT_MIN;
  float maxDiag = fmaxf(fabsf(W[0]), fmaxf(fabsf(W[4]), fabsf(W[8])));
  int finished = 0;
  for (int guard = 0; !finished && guard < 64; ++guard) {
    finished = 1;
    for (int p = 1; p < 3; ++p) for (int q = 0; q < p; ++q) {
      float thr = fmaxf(tiny, precision * maxDiag);
      if (!(fabsf(W[3 * p + q]) > thr || fabsf(W[3 * q + p]) > thr)) continue;
      finished = 0;
      float mm[4] = {W[3 * p + p], W[3 * p + q], W[3 * q + p], W[3 * q + q]};          /* real_2x2_jacobi_svd */
      float t = mm[0] + mm[3], d = mm[2] - mm[1], r1c, r1s;
      if (fabsf(d) < tiny) { r1s = 0; r1c = 1; }
      else { float u = t / d; float tmp = sqrtf(1.0f + u * u); r1s = 1.0f / tmp; r1c = u / tmp; }
      eig_apply_rot(&mm[0], 1, &mm[2], 1, 2, r1c, r1s);
      float jrc, jrs;                                                                    /* makeJacobi(m00, m01, m11) */
      { float deno = 2.0f * fabsf(mm[1]);
        if (deno < tiny) { jrc = 1; jrs = 0; }
        else { float tau = (mm[0] - mm[3]) / deno; float w = sqrtf(tau * tau + 1.0f); float tt;
          if (tau > 0) tt = 1.0f / (tau + w); else tt = 1.0f / (tau - w);
          float sign_t = tt > 0 ? 1.0f : -1.0f; float n = 1.0f / sqrtf(tt * tt + 1.0f);
          jrs = -sign_t * (mm[1] / fabsf(mm[1])) * fabsf(tt) * n; jrc = n; } }
      float jlc, jls;                                                                    /* rot1 * j_right^T */
      { float oc = jrc, os = -jrs; jlc = r1c * oc - r1s * os; jls = r1c * os + r1s * oc; }
      eig_apply_rot(&W[3 * p], 1, &W[3 * q], 1, 3, jlc, jls);
      eig_apply_rot(&U[p], 3, &U[q], 3, 3, jlc, jls);
      eig_apply_rot(&W[p], 3, &W[q], 3, 3, jrc, -jrs);
      eig_apply_rot(&V[p], 3, &V[q], 3, 3, jrc, -jrs);
      maxDiag = fmaxf(maxDiag, fmaxf(fabsf(W[3 * p + p]), fabsf(W[3 * q + q])));
    }
  }
  float sv[3];
  for (int i = 0; i < 3; ++i) { float a = W[4 * i]; sv[i] = fabsf(a); if (a < 0) for (int r = 0; r < 3; ++r) U[3 * r + i] = -U[3 * r + i]; }
  for (int i = 0; i < 3; ++i) sv[i] *= scale;
  for (int i = 0; i < 3; ++i) {
    int pos = 0; float mx = sv[i];
    for (int k = i + 1; k < 3; ++k) if (sv[k] > mx) { mx = sv[k]; pos = k - i; }
    if (mx == 0) break;
    if (pos) { pos += i; float tf = sv[i]; sv[i] = sv[pos]; sv[pos] = tf;
      for (int r = 0; r < 3; ++r) { float a = U[3 * r + pos]; U[3 * r + pos] = U[3 * r + i]; U[3 * r + i] = a;
                                    a = V[3 * r + pos]; V[3 * r + pos] = V[3 * r + i]; V[3 * r + i] = a; } }
  }
  float P[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    float a0 = U[3 * i] * V[3 * j], a1 = U[3 * i + 1] * V[3 * j + 1], a2 = U[3 * i + 2] * V[3 * j + 2]; float a12 = a1 + a2; P[3 * i + j] = a0 + a12; }
  float x;
  { float h0 = P[4] * P[8], h1 = P[5] * P[7], g0 = P[3] * P[8], g1 = P[5] * P[6], f0 = P[3] * P[7], f1 = P[4] * P[6];
    float d0 = h0 - h1, d1 = g0 - g1, d2 = f0 - f1; float t0 = P[0] * d0, t1 = P[1] * d1, t2 = P[2] * d2; float u = t0 - t1; x = u + t2; }
  float M[9]; memcpy(M, U, sizeof(M));
  for (int r = 0; r < 3; ++r) M[3 * r] = M[3 * r] / x;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    float a0 = M[3 * i] * V[3 * j], a1 = M[3 * i + 1] * V[3 * j + 1], a2 = M[3 * i + 2] * V[3 * j + 2]; float a12 = a1 + a2; R[3 * i + j] = a0 + a12; }
}
static float eigen_init_yaw(float c, float s) {
  float omc = 1.0f - c, m22 = omc + c;                     /* AngleAxisf::toRotationMatrix: cos_axis.z * axis.z + c */
  float R[9];
  eigen_rotation_z(c, s, m22, R);
  float res0 = atan2f(R[5], R[8]);
  float s1 = sinf(res0), c1 = cosf(res0);
  float n0 = s1 * R[6], n1 = c1 * R[3], d0 = c1 * R[4], d1 = s1 * R[7];
  return -atan2f(n0 - n1, d0 - d1);
}

/* src/PoseEstimator.cpp:22-24 init_guess = Translation3f(tx,ty,0) * AngleAxisf(yaw, Z), then the
 * computeTransformation prologue: p = (translation, rotation().eulerAngles(0,1,2)) of the float matrix;
 * for a pure-Z rotation eulerAngles gives (-0, 0, yaw) (include/Eigen/src/Geometry/EulerAngles.h:87-107) with yaw from
 * eigen_init_yaw (libm_f32 = 1: bit-equal to the vendored Eigen's answer, tests/test_eigen_pins.py) or modelled as
 * atan2f(s, c) correctly rounded (libm_f32 = 0: within 2.4e-7 rad of it). */
static void init_guess(const ndt_oracle_params *prm, const double init[3], tf32 *T, double p[3]) {
  double pinit[3] = {init[0], init[1], init[2]};
  *T = tf_from_p(prm, pinit);
  p[0] = (double)T->tx; p[1] = (double)T->ty;
  if (prm->libm_f32) p[2] = (double)eigen_init_yaw(T->c, T->s);      /* Eigen's own arithmetic + the platform's libm */
  else p[2] = (double)(float)atan2((double)T->s, (double)T->c);    /* model: rotation() = the linear part, atan2f correctly rounded */
  if (g_hooks.init_p) { float t4[4] = {T->c, T->s, T->tx, T->ty}; g_hooks.init_p(t4, p); }
}

/* src/PoseEstimator.cpp:17-64 with the source cloud already filtered (a1 upstream). */
static int align_impl(const ndt_oracle_map *m, const float *scan, size_t n, size_t stride,
                      const double init[3], ndt_oracle_result *res, double *trace, int trace_cap,
                      ndt_oracle_run_stats *st) {
  memset(res, 0, sizeof(*res));
  if (st) memset(st, 0, sizeof(*st));
  if (!m || !scan || n == 0) { res->status = -1; res->fitness = DBL_MAX; return -1; }
  const ndt_oracle_params *prm = &m->prm;
  align_ctx cx; memset(&cx, 0, sizeof(cx));
  cx.m = m; cx.scan = scan; cx.n = n; cx.stride = stride;
  cx.trans = (float *)malloc(2 * n * sizeof(float));
  cx.trace = trace; cx.trace_cap = trace_cap;

  /* src/PoseEstimator.cpp:22-24: init_guess = Translation3f(tx,ty,0) * AngleAxisf(yaw, Z) */
  double p[3];
  init_guess(prm, init, &cx.T, p);
  transform_scan(prm, scan, n, stride, cx.T, cx.trans);   /* transformPointCloud(output, output, guess) */

  double g[3], H[6], score;
  score = derivatives_run(&cx, p, 1, g, H);
  trace_push(&cx, 0.0, score, g, p);

  int converged = 0, iters = 0, nan_exit = 0;
  while (!converged) {
    double Hf[9] = {H[0], H[1], H[2], H[1], H[3], H[4], H[2], H[4], H[5]};
    double mg[3] = {-g[0], -g[1], -g[2]}, dp[3];
    newton_solve(Hf, mg, dp);
    double nrm = sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]);
    if (nrm == 0 || nrm != nrm) { converged = (nrm == nrm); nan_exit = 1; break; }
    dp[0] /= nrm; dp[1] /= nrm; dp[2] /= nrm;
    double a = step_length_mt(&cx, p, dp, nrm, prm->step_size, prm->trans_eps / 2, &score, g, H);
    dp[0] *= a; dp[1] *= a; dp[2] *= a;
    p[0] += dp[0]; p[1] += dp[1]; p[2] += dp[2];
    int over = prm->conv_ge ? (iters >= prm->max_iter) : (iters > prm->max_iter);
    if (over || (iters && (fabs(a) < prm->trans_eps))) converged = 1;
    iters++;
  }
  (void)nan_exit;
  res->iters = iters;
  res->converged = converged;
  res->score = score;
  res->trans_prob = score / (double)n;
  res->p[0] = p[0]; res->p[1] = p[1]; res->p[2] = p[2];
  res->T00 = cx.T.c; res->T10 = cx.T.s; res->T03 = cx.T.tx; res->T13 = cx.T.ty;
  /* src/PoseEstimator.cpp:29-36 */
  res->pose[0] = (double)cx.T.tx; res->pose[1] = (double)cx.T.ty;
  res->pose[2] = ndt_oracle_yaw_from_T(cx.T.c, cx.T.s);
  /* src/PoseEstimator.cpp:43 */
  res->fitness = fitness_pass(m, scan, n, stride, cx.T);
  /* src/PoseEstimator.cpp:53-56: getHessian -> computeHessian on the output cloud.  It re-uses
   * the member angle terms, so it reproduces the last Hessian of align() bit for bit; the
   * pass is executed here as the reference executes it. */
  double H8[6];
  cx.evals++; cx.ref_evals++;
  eval_pass(m, scan, n, stride, cx.trans, cx.at, 2, NULL, H8, &cx.pairs);
  res->H[0] = H8[0]; res->H[1] = res->H[3] = H8[1]; res->H[2] = res->H[6] = H8[2];
  res->H[4] = H8[3]; res->H[5] = res->H[7] = H8[4]; res->H[8] = H8[5];
  res->evals = cx.evals; res->ref_evals = cx.ref_evals;
  res->flags = cx.trace_n;   /* number of trace rows (derivative passes with a gradient) */
  res->kbar = cx.pairs / ((double)cx.evals * (double)n);
  res->status = 0;
  if (st) {
    st->evals_run = cx.evals_run; st->pairs_run = cx.pairs_run;
    st->kbar_run = cx.evals_run > 0 ? cx.pairs_run / ((double)cx.evals_run * (double)n) : 0.0;   /* = the device's ndt_result.kbar */
  }
  free(cx.trans);
  return 0;
}

int ndt_oracle_align(const ndt_oracle_map *m, const float *scan, size_t n, size_t stride,
                     const double init[3], ndt_oracle_result *res, double *trace, int trace_cap) {
  return align_impl(m, scan, n, stride, init, res, trace, trace_cap, NULL);
}
int ndt_oracle_align_ex(const ndt_oracle_map *m, const float *scan, size_t n, size_t stride, const double init[3],
                        ndt_oracle_result *res, double *trace, int trace_cap, ndt_oracle_run_stats *st) {
  return align_impl(m, scan, n, stride, init, res, trace, trace_cap, st);
}

int ndt_oracle_align_batch(const ndt_oracle_map *m, const float *scans, const uint64_t *off, int B,
                           const double *inits, ndt_oracle_result *res, int nthreads) {
  return ndt_oracle_align_batch_ex(m, scans, off, B, inits, res, nthreads, NULL);
}
int ndt_oracle_align_batch_ex(const ndt_oracle_map *m, const float *scans, const uint64_t *off, int B,
                              const double *inits, ndt_oracle_result *res, int nthreads, ndt_oracle_run_stats *st) {
  int rc = 0;
#ifdef _OPENMP
  if (nthreads > 1) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1) if (nthreads > 1)
#endif
  for (int b = 0; b < B; ++b) {
    int r = align_impl(m, scans + 2 * off[b], (size_t)(off[b + 1] - off[b]), 2 * sizeof(float),
                       inits + 3 * b, &res[b], NULL, 0, st ? st + b : NULL);
    if (r) rc = r;
  }
  (void)nthreads;
  return rc;
}

/* BASELINE.json configs[4]: B initial guesses for ONE scan (multi-hypothesis relocalisation). */
int ndt_oracle_align_seeds(const ndt_oracle_map *m, const float *scan, size_t n, int B,
                           const double *inits, ndt_oracle_result *res, int nthreads) {
  return ndt_oracle_align_seeds_ex(m, scan, n, B, inits, res, nthreads, NULL);
}
int ndt_oracle_align_seeds_ex(const ndt_oracle_map *m, const float *scan, size_t n, int B,
                              const double *inits, ndt_oracle_result *res, int nthreads, ndt_oracle_run_stats *st) {
  int rc = 0;
#ifdef _OPENMP
  if (nthreads > 1) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1) if (nthreads > 1)
#endif
  for (int b = 0; b < B; ++b) {
    int r = align_impl(m, scan, n, 2 * sizeof(float), inits + 3 * b, &res[b], NULL, 0, st ? st + b : NULL);
    if (r) rc = r;
  }
  (void)nthreads;
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* a1: pcl::ApproximateVoxelGrid::filter, z = 0 cloud (src/PoseEstimator.cpp:6-10)             */
/* ------------------------------------------------------------------------------------------ */
size_t ndt_oracle_approx_voxel_filter(const float *xy, size_t n, size_t stride, float leaf,
                                      float *out) {
  enum { HIST = 512 };
  struct he { int ix, iy, iz, count; float cx, cy; } hist[HIST];
  memset(hist, 0, sizeof(hist));
  float inv = 1.0f / leaf;
  size_t op = 0;
  for (size_t i = 0; i < n; ++i) {
    const float *p = pt_at(xy, stride, i);
    int ix = (int)floorf(p[0] * inv), iy = (int)floorf(p[1] * inv), iz = (int)floorf(0.0f * inv);
    unsigned h = (unsigned)((ix * 7171 + iy * 3079 + iz * 4231) & (HIST - 1));
    struct he *e = &hist[h];
    if (e->count && (ix != e->ix || iy != e->iy || iz != e->iz)) {
      out[2 * op] = e->cx / (float)e->count; out[2 * op + 1] = e->cy / (float)e->count; ++op;
      e->count = 0; e->cx = 0.f; e->cy = 0.f;
    }
    e->ix = ix; e->iy = iy; e->iz = iz; e->count++;
    e->cx += p[0]; e->cy += p[1];
  }
  for (int h = 0; h < HIST; ++h) {
    struct he *e = &hist[h];
    if (e->count) { out[2 * op] = e->cx / (float)e->count; out[2 * op + 1] = e->cy / (float)e->count; ++op; }
  }
  return op;
}


/* ------------------------------------------------------------------------------------------
 * odometry prediction and EKF fusion around the match.
 * ------------------------------------------------------------------------------------------ */
#define F2_DEG2RAD(x) ((x) * M_PI / 180)     /* include/ndt_slam/MyUtil.h:22 */
#define F2_RAD2DEG(x) ((x) * 180 / M_PI)     /* include/ndt_slam/MyUtil.h:23 */

void ndt_oracle_fuse_default_params(ndt_oracle_fuse_params *p) {
  p->coe_ndt_cov = 1.0; p->coe_vel = 0.1; p->coe_omega = 0.1; p->del_time = 0.5; p->score_thre = 0.0;
}

static double f2_add_angle(double a1, double a2) {   /* src/MyUtil.cpp:4-12 */
  double sum = a1 + a2;
  if (sum < -180) sum += 360; else if (sum >= 180) sum -= 360;
  return sum;
}
static double f2_sub_angle(double a1, double a2) {   /* src/MyUtil.cpp:15-23 */
  double dif = a1 - a2;
  if (dif < -180) dif += 360; else if (dif >= 180) dif -= 360;
  return dif;
}

void ndt_oracle_predict(const double cur[3], const double prev[3], const double last[3],
                        double motion[3], double pred[3]) {
  /* Pose2D::calRmat of prevPose / lastPose (include/ndt_slam/Pose2D.h:43-48) */
  double ap = F2_DEG2RAD(prev[2]), cp = cos(ap), sp = sin(ap);
  double dx = cur[0] - prev[0], dy = cur[1] - prev[1];
  motion[0] = cp * dx + sp * dy;                    /* Rmat[0][0]*dx + Rmat[1][0]*dy */
  motion[1] = -sp * dx + cp * dy;                   /* Rmat[0][1]*dx + Rmat[1][1]*dy */
  motion[2] = f2_sub_angle(cur[2], prev[2]);
  double al = F2_DEG2RAD(last[2]), cl = cos(al), sl = sin(al);
  pred[0] = cl * motion[0] + -sl * motion[1] + last[0];
  pred[1] = sl * motion[0] + cl * motion[1] + last[1];
  pred[2] = f2_add_angle(last[2], motion[2]);
}

/* Eigen's fixed-size 3x3 inverse (include/Eigen/src/LU/InverseImpl.h:140-200): cofactors, the
 * determinant along column 0, everything multiplied by 1/det. */
static void f2_inv3(const double m[9], double out[9]) {
#define M(i, j) m[3 * (i) + (j)]
  double c00 = M(1,1) * M(2,2) - M(1,2) * M(2,1);
  double c10 = M(0,2) * M(2,1) - M(0,1) * M(2,2);     /* cofactor<1,0> */
  double c20 = M(0,1) * M(1,2) - M(0,2) * M(1,1);
  double det = c00 * M(0,0) + c10 * M(1,0) + c20 * M(2,0);
  double id = 1.0 / det;
  out[0] = c00 * id; out[1] = c10 * id; out[2] = c20 * id;
  out[3] = (M(1,2) * M(2,0) - M(1,0) * M(2,2)) * id;
  out[4] = (M(0,0) * M(2,2) - M(0,2) * M(2,0)) * id;
  out[5] = (M(0,2) * M(1,0) - M(0,0) * M(1,2)) * id;
  out[6] = (M(1,0) * M(2,1) - M(1,1) * M(2,0)) * id;
  out[7] = (M(0,1) * M(2,0) - M(0,0) * M(2,1)) * id;
  out[8] = (M(0,0) * M(1,1) - M(0,1) * M(1,0)) * id;
#undef M
}
static void f2_mul3(const double a[9], const double b[9], double o[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = a[3 * i] * b[j];
      s += a[3 * i + 1] * b[3 + j];
      s += a[3 * i + 2] * b[6 + j];
      o[3 * i + j] = s;
    }
}

/* PoseFuser::calOdometryCovariance (src/PoseFuser.cpp:39-61) */
static void f2_odo_cov(const double motion[3], const double last[3], const double last_cov[9],
                       const ndt_oracle_fuse_params *p, double cov[9]) {
  double dt = p->del_time;
  double v = sqrt(motion[0] * motion[0] + motion[1] * motion[1]) / dt;   /* Pose2D::calDistance */
  double omega = F2_DEG2RAD(motion[2] / dt);
  double m00 = p->coe_vel * v * v, m11 = p->coe_omega * omega * omega;
  double a = F2_DEG2RAD(last[2]), c = cos(a), s = sin(a);
  double F[9] = {1, 0, -v * dt * s, 0, 1, v * dt * c, 0, 0, 1};
  double Ft[9] = {1, 0, 0, 0, 1, 0, F[2], F[5], 1};
  double t[9], flf[9];
  f2_mul3(F, last_cov, t); f2_mul3(t, Ft, flf);
  /* A M A^T with A = [dt c, 0; dt s, 0; 0, dt] */
  double a0 = dt * c, a1 = dt * s;
  double ama[9] = {a0 * m00 * a0, a0 * m00 * a1, 0, a1 * m00 * a0, a1 * m00 * a1, 0, 0, 0, dt * m11 * dt};
  for (int i = 0; i < 9; ++i) cov[i] = flf[i] + ama[i];
}

int ndt_oracle_fuse(const ndt_oracle_result *r, const double pred[3], const double motion[3],
                    const double last[3], const double last_cov[9], const ndt_oracle_fuse_params *p,
                    double fused[3], double cov[9]) {
  /* src/PoseEstimator.cpp:29-36,43-46: estimated pose in degrees, cost with the 1e7 sentinel */
  double est[3] = {r->pose[0], r->pose[1], F2_RAD2DEG(r->pose[2])};
  double cost = r->converged ? r->fitness : 10000000.0;
  int successful = cost <= p->score_thre;             /* src/ScanMatcher.cpp:50 */
  if (!successful) {                                  /* :63-65 */
    f2_odo_cov(motion, last, last_cov, p, cov);
    fused[0] = pred[0]; fused[1] = pred[1]; fused[2] = pred[2];
    return 0;
  }
  /* src/PoseEstimator.cpp:57-64: Qmat = (-H3)^-1 * coeNDTCov */
  double nh[9], Q[9];
  for (int i = 0; i < 9; ++i) nh[i] = -r->H[i];
  f2_inv3(nh, Q);
  for (int i = 0; i < 9; ++i) Q[i] *= p->coe_ndt_cov;
  /* PoseFuser::fusePose (src/PoseFuser.cpp:3-37) */
  double ch[9], sum[9], inv[9], K[9], imk[9];
  f2_odo_cov(motion, last, last_cov, p, ch);
  for (int i = 0; i < 9; ++i) sum[i] = Q[i] + ch[i];
  f2_inv3(sum, inv);
  f2_mul3(ch, inv, K);
  for (int i = 0; i < 9; ++i) imk[i] = ((i % 4 == 0) ? 1.0 : 0.0) - K[i];
  f2_mul3(imk, ch, cov);
  double zh[3] = {est[0] - pred[0], est[1] - pred[1], F2_DEG2RAD(f2_sub_angle(est[2], pred[2]))};
  double mu_hat[3] = {pred[0], pred[1], F2_DEG2RAD(pred[2])};
  double mu[3];
  for (int i = 0; i < 3; ++i) {
    double s = K[3 * i] * zh[0];
    s += K[3 * i + 1] * zh[1];
    s += K[3 * i + 2] * zh[2];
    mu[i] = s + mu_hat[i];
  }
  fused[0] = mu[0]; fused[1] = mu[1]; fused[2] = F2_RAD2DEG(mu[2]);
  return 1;
}


/* SURVEY.md 8f row f3 (part) */
size_t ndt_oracle_remove_neighbors(const float *base, size_t nb, const float *list, size_t nl, double thre, float *out) {
  size_t cnt = 0;
  for (size_t i = 0; i < nb; ++i) {
    int flag = 1;
    for (size_t j = 0; j < nl; ++j) {
      float dx = base[2 * i] - list[2 * j], dy = base[2 * i + 1] - list[2 * j + 1], dz = 0.0f - 0.0f;
      float d2 = dx * dx + dy * dy + dz * dz;
      if ((double)sqrtf(d2) < thre) flag = 0;
    }
    if (flag) { out[2 * cnt] = base[2 * i]; out[2 * cnt + 1] = base[2 * i + 1]; ++cnt; }
  }
  return cnt;
}


/* ------------------------------------------------------------------------------------------
 * Pin points: the hand restatements of routines whose source IS in the reference tree (its vendored
 * Eigen 3.3.90), exposed one by one for tests/test_eigen_pins.py (fixtures: tests/golden/eigen_golden.npz,
 * made by tests/golden/make_eigen_golden.{cpp,py} from that Eigen).
 * ------------------------------------------------------------------------------------------ */
int ndt_oracle_leaf(const ndt_oracle_params *prm, int n, const double sums[6], double mean[2], double icov[3]) {
  return leaf_finalize(prm, n, sums[0], sums[1], sums[2], sums[3], sums[4], sums[5], mean, icov);
}
void ndt_oracle_inv3(const double m[9], double out[9]) { f2_inv3(m, out); }
void ndt_oracle_init_guess(const ndt_oracle_params *prm, const double init[3], float T[4], double p[3]) {
  tf32 t; init_guess(prm, init, &t, p);
  T[0] = t.c; T[1] = t.s; T[2] = t.tx; T[3] = t.ty;
}
void ndt_oracle_step_matrix(const ndt_oracle_params *prm, const double p[3], float T[4]) {
  tf32 t = tf_from_p(prm, p);
  T[0] = t.c; T[1] = t.s; T[2] = t.tx; T[3] = t.ty;
}
/* replace the fp64 part of the cell table (compact order of ndt_oracle_map_export) */
void ndt_oracle_map_override_cells(ndt_oracle_map *m, const double *mean, const double *icov, const int *npts) {
  memcpy(m->c_mean, mean, 2 * (size_t)m->n_cells * sizeof(double));
  memcpy(m->c_icov, icov, 3 * (size_t)m->n_cells * sizeof(double));
  if (npts) memcpy(m->c_npts, npts, (size_t)m->n_cells * sizeof(int));
}
/* per-voxel sums exactly as the build loop forms them (cloud order, fp64; identity start per preset):
 * n_cells x {n, sx, sy, sxx, sxy, syy, szz} */
void ndt_oracle_map_export_sums(const ndt_oracle_map *m, double *out) {
  for (int c = 0; c < m->n_cells; ++c) {
    int g = m->c_idx[c];
    double sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0, szz = 0;
    if (m->prm.cov_init_identity) { sxx = 1.0; syy = 1.0; szz = 1.0; }
    for (int s = m->pt_start[g]; s < m->pt_start[g + 1]; ++s) {
      double X = (double)m->pts[2 * s], Y = (double)m->pts[2 * s + 1];
      sx += X; sy += Y; sxx += X * X; sxy += X * Y; syy += Y * Y;
    }
    double *o = out + 7 * (size_t)c;
    o[0] = (double)(m->pt_start[g + 1] - m->pt_start[g]); o[1] = sx; o[2] = sy; o[3] = sxx; o[4] = sxy; o[5] = syy; o[6] = szz;
  }
}
