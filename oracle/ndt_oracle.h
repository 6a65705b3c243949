/*
 * ndt_oracle.h -- CPU restatement of the reference's NDT scan-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ndt_slam_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in PCL
 * (pcl::NormalDistributionsTransform, pcl::VoxelGridCovariance,
 * pcl::Registration::getFitnessScore, pcl::transformPointCloud), an un-vendored,
 * version-unpinned dependency of the reference (CMakeLists.txt:21
 * `find_package(PCL 1.2 REQUIRED)`), absent from /root/reference and from this image.
 * The reference ships no tests, fixtures or golden vectors.  This file
 * restates the published algorithm (Magnusson 2009 eqs 6.8-6.21 / Algorithm 2;
 * More & Thuente 1994; Sun & Yuan 2006 eqs 2.4.2/2.4.5/2.4.52/2.4.56) with the
 * PCL <= 1.10 semantics listed in SURVEY.md 8a rows a1-a9, anchored on the
 * reference's call sites:
 *     src/PoseEstimator.cpp:4-69          (order of calls, units, 3x3 extraction)
 *     include/ndt_slam/PoseEstimator.h:63-104 (parameters, z = 0 clouds)
 *     ndt_mapping.launch:30-36            (parameter values)
 * Every version-sensitive PCL detail (SURVEY.md 8c list) is a named switch in
 * ndt_oracle_params.
 */
#ifndef NDT_ORACLE_H_
#define NDT_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Parameters.  Layout is shared (field for field) with include/ndt_mi355x.h's
 * ndt_params so tests can pass one ctypes struct to both sides. */
typedef struct ndt_oracle_params {
  float  resolution;        /* PoseEstimator.h:81 ndt.setResolution (float in PCL)            */
  double step_size;         /* PoseEstimator.h:79 ndt.setStepSize                             */
  double trans_eps;         /* PoseEstimator.h:77 ndt.setTransformationEpsilon                */
  int    max_iter;          /* PoseEstimator.h:83 ndt.setMaximumIterations                    */
  double outlier_ratio;     /* PCL default 0.55                                               */
  int    min_pts;           /* VoxelGridCovariance min_points_per_voxel_ = 6                  */
  double eig_mult;          /* VoxelGridCovariance min_covar_eigvalue_mult_ = 0.01            */
  /* ---- version-sensitive switches; defaults = PCL 1.10 (ndt_oracle_params_preset 0) ---- */
  int    cov_unbiased;      /* (1) 0: (Sxx/n - mu mu^T)(n-1)/n   1: /(n-1)                     */
  int    cov_init_identity; /* (1b) 1: per-voxel Sxx accumulator starts at I (old PCL Leaf())  */
  int    conv_ge;           /* (2) 0: stop when iter > max_iter  1: iter >= max_iter           */
  int    radius_inclusive;  /* (4) 0: d^2 < r^2                  1: d^2 <= r^2                 */
  int    transform_sse;     /* (12) 0: ((m00 x + m01 y) + m03)   1: m00 x + (m01 y + m03)      */
  int    stale_h_ang;       /* (8) 0 (every preset): computeDerivatives refreshes j_ang AND h_ang on every
                                     trial (it calls computeAngleDerivatives(p) with the default
                                     compute_hessian = true), so computeHessian after an inner loop sees
                                     the terms of the LAST trial.  1: a PCL whose inner trials would skip
                                     the h_ang refresh (terms of the line search's FIRST trial)       */
  double snap_thresh;       /* (6) small-angle snap, PCL `fabs(p) < 10e-5`                     */
  int    mt_max_iter;       /* (11) 10                                                         */
  double mt_mu;             /* (11) 1e-4                                                       */
  double mt_nu;             /* (11) 0.9                                                        */
  int    libm_f32;          /* (12b) float32 cos / sin of a trial's yaw: 1: THIS machine's libm cosf / sinf (glibc:
                                     the reference's platform; what Eigen's AngleAxisf calls); 0: correctly rounded */
  int    grid_margin;       /* layout only (the struct is shared with ndt_params): the checker's voxel grid is always the
                                     one PCL derives from the cloud's bounding box */
} ndt_oracle_params;

/* Result of one match.  Layout shared with include/ndt_mi355x.h's ndt_result. */
typedef struct ndt_oracle_result {
  double pose[3];     /* x, y, yaw[rad] recovered from the float32 matrix by the a9 branch logic
                         (src/PoseEstimator.cpp:29-36)                                           */
  float  T00, T10, T03, T13; /* float32 entries of getFinalTransformation()                        */
  double fitness;     /* getFitnessScore(): mean squared distance to nearest raw map point (m^2) */
  double trans_prob;  /* getTransformationProbability(): score / N                               */
  double score;       /* score at the last evaluation                                            */
  double H[9];        /* d2 score / dp2 (tx,ty,yaw[rad]) at the final cloud, row-major = the
                         {0,1,5} block of PCL's 6x6 (src/PoseEstimator.cpp:53-61 before the sign) */
  double p[3];        /* final fp64 parameter vector (tx,ty,yaw)                                 */
  int    iters;       /* nr_iterations_                                                          */
  int    evals;       /* derivative passes executed by THIS implementation                       */
  int    ref_evals;   /* passes the reference executes for the same path (incl. Hessian-only
                         passes and the a8 getHessian pass)                                      */
  int    converged;   /* hasConverged()                                                          */
  int    status;      /* 0 ok; <0 error                                                          */
  int    flags;   /* product: data-path bits (NDT_FLAG_*); the oracle has one path and returns the number of trace rows here */
  double kbar;        /* mean in-radius cells per point-evaluation                               */
} ndt_oracle_result;

typedef struct ndt_oracle_map ndt_oracle_map;

void ndt_oracle_default_params(ndt_oracle_params *p);          /* = preset 0 */
/* 0: PCL 1.9/1.10 (identity-initialised cov_, (n-1)/n, SSE transform); 1: PCL <= 1.8 (scalar transform);
 * 2: PCL >= 1.11 (zero-initialised cov_, /(n-1)) */
void ndt_oracle_params_preset(ndt_oracle_params *p, int preset);

/* a2: VoxelGridCovariance::filter(true) over a z=0 cloud.  xy points at byte stride. */
ndt_oracle_map *ndt_oracle_map_build(const float *xy, size_t n, size_t stride_bytes,
                                     const ndt_oracle_params *prm);
void ndt_oracle_map_destroy(ndt_oracle_map *m);

/* Introspection used by the parity tests (cell table compared field by field). */
typedef struct ndt_oracle_map_info {
  int min_bx, min_by, div_x, div_y;
  int n_cells;      /* voxels with >= min_pts points (members of the centroid search set) */
  int n_valid;      /* of those, voxels with an accepted covariance                        */
  size_t n_points;
} ndt_oracle_map_info;
void ndt_oracle_map_info_get(const ndt_oracle_map *m, ndt_oracle_map_info *out);
/* Export the cell table in ascending dense-index order.  Arrays sized n_cells.
 * cell_idx = iy*div_x+ix ; cent = float32 centroid xy ; mean = fp64 mean xy ;
 * icov = xx,xy,yy ; npts = point count (negative => rejected covariance, icov = 0). */
void ndt_oracle_map_export(const ndt_oracle_map *m, int *cell_idx, float *cent, double *mean,
                           double *icov, int *npts);

/* a4+a5 at an explicit pose: transform with the float32 matrix built from p (as the line
 * search does), then score/gradient/Hessian.  g[3], H[9].  Returns score. */
double ndt_oracle_eval_at(const ndt_oracle_map *m, const float *scan_xy, size_t n,
                          size_t stride_bytes, const double p[3], double g[3], double H[9],
                          double *pairs_out);

/* a3-a9: one complete match (align + fitness + final Hessian + a9 extraction).
 * init = (tx, ty, yaw[rad]) exactly as src/PoseEstimator.cpp:22-24 feeds them
 * (yaw = DEG2RAD(initPose.th)).  trace (optional, may be NULL): per derivative pass
 * 8 doubles {a_t, score, g0,g1,g2, p0,p1,p2}; trace_cap passes at most. */
int ndt_oracle_align(const ndt_oracle_map *m, const float *scan_xy, size_t n, size_t stride_bytes,
                     const double init[3], ndt_oracle_result *res, double *trace, int trace_cap);

/* What the DEVICE path runs of a match, counted on the CPU side (the cheapest whole-path integer check at config scale):
 * the derivative passes with a gradient minus the line-search trials that repeat the step length of the pass before them
 * (the device re-uses that pass's totals, ndt_slam_amd/csrc/ndt_optimizer.hip.h: advance; Hessian-only passes and getHessian
 * are fused away there), and the (point, voxel) pairs of exactly those passes.  kbar_run = pairs_run / (evals_run * n) is
 * the device's ndt_result.kbar, operation for operation. */
typedef struct ndt_oracle_run_stats { int evals_run; int pad_; double pairs_run; double kbar_run; } ndt_oracle_run_stats;
int ndt_oracle_align_ex(const ndt_oracle_map *m, const float *scan_xy, size_t n, size_t stride_bytes, const double init[3],
                        ndt_oracle_result *res, double *trace, int trace_cap, ndt_oracle_run_stats *st);
/* 1: a line-search trial at the step length of the pass just run re-uses that pass's totals instead of running it again
 * (bench.py's cpu_baseline.memoised: like for like with the passes the device runs).  Default 0 = what the reference runs.
 * Results (transforms, iterations, ref_evals, traces) are the same either way.  Process-wide; do not flip while matches run. */
void ndt_oracle_set_memoise(int on);

/* Batch of independent matches; offsets[B+1] in points; inits B x 3.
 * nthreads <= 1: scalar loop; > 1: OpenMP over scans when built with -fopenmp. */
int ndt_oracle_align_batch(const ndt_oracle_map *m, const float *scans_xy,
                           const uint64_t *offsets, int B, const double *inits,
                           ndt_oracle_result *res, int nthreads);
/* the same scan from B initial guesses (BASELINE.json configs[4]) */
int ndt_oracle_align_seeds(const ndt_oracle_map *m, const float *scan_xy, size_t n, int B,
                           const double *inits, ndt_oracle_result *res, int nthreads);
/* both with the run statistics of every match (st: B entries, or NULL) */
int ndt_oracle_align_batch_ex(const ndt_oracle_map *m, const float *scans_xy, const uint64_t *offsets, int B,
                              const double *inits, ndt_oracle_result *res, int nthreads, ndt_oracle_run_stats *st);
int ndt_oracle_align_seeds_ex(const ndt_oracle_map *m, const float *scan_xy, size_t n, int B,
                              const double *inits, ndt_oracle_result *res, int nthreads, ndt_oracle_run_stats *st);

/* a7 alone at an explicit float32 matrix (c, s, tx, ty). */
double ndt_oracle_fitness(const ndt_oracle_map *m, const float *scan_xy, size_t n,
                          size_t stride_bytes, float c, float s, float tx, float ty);

/* a1: ApproximateVoxelGrid::filter on a z=0 cloud (src/PoseEstimator.cpp:6-10).
 * out_xy must hold 2*n floats.  Returns number of output points. */
size_t ndt_oracle_approx_voxel_filter(const float *xy, size_t n, size_t stride_bytes, float leaf,
                                      float *out_xy);

/* a9 alone: yaw from float32 matrix entries (src/PoseEstimator.cpp:31-35). */
double ndt_oracle_yaw_from_T(float T00, float T10);

/* More-Thuente pieces exposed for known-answer tests. */
double ndt_oracle_mt_trial(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u,
                           double a_t, double f_t, double g_t);
int ndt_oracle_mt_update(double *a_l, double *f_l, double *g_l, double *a_u, double *f_u,
                         double *g_u, double a_t, double f_t, double g_t);
void ndt_oracle_gauss(const ndt_oracle_params *prm, double *d1, double *d2);
/* 3x3 symmetric pseudo-inverse solve H x = b (stands in for the 6x6 JacobiSVD solve). */
void ndt_oracle_solve3(const double H[9], const double b[3], double x[3]);

/* ---- the steps either side of the match (also parity unpinned: the
 * reference's own sources need ROS + Eigen headers, absent here; restated from src/Pose2D.cpp,
 * src/PoseFuser.cpp, src/MyUtil.cpp:4-23, src/ScanMatcher.cpp:27-67, src/PoseEstimator.cpp:43-64).
 * Poses are (tx, ty, th) with th in DEGREES as in include/ndt_slam/Pose2D.h:14. ---- */
typedef struct ndt_oracle_fuse_params {
  double coe_ndt_cov;  /* PoseEstimator.h:63 coeNDTCov (1.0)  */
  double coe_vel;      /* PoseFuser.h:19 coeVel (0.1)         */
  double coe_omega;    /* PoseFuser.h:19 coeOmega (0.1)       */
  double del_time;     /* PoseFuser.h:19 delTime (0.5)        */
  double score_thre;   /* ScanMatcher.h:49 scthre (0.0, launch file: score_thre) */
} ndt_oracle_fuse_params;
void ndt_oracle_fuse_default_params(ndt_oracle_fuse_params *p);
/* Pose2D::calMotion (src/Pose2D.cpp:5-16) then Pose2D::calPredPose (:28-37), as ScanMatcher::matchScan
 * chains them (src/ScanMatcher.cpp:27-32): odometry motion in the robot frame and the predicted pose. */
void ndt_oracle_predict(const double odo_cur[3], const double odo_prev[3], const double last_pose[3],
                        double motion[3], double pred[3]);
/* src/PoseEstimator.cpp:43-64 (cost, Qmat from the Hessian), src/ScanMatcher.cpp:50-67 (decision),
 * PoseFuser::fusePose / calOdometryCovariance (src/PoseFuser.cpp:3-61).  Returns `successful`. */
int ndt_oracle_fuse(const ndt_oracle_result *r, const double pred[3], const double motion[3],
                    const double last_pose[3], const double last_cov[9],
                    const ndt_oracle_fuse_params *prm, double fused[3], double cov[9]);

/* PCFilter::remove_neighborPoint (include/ndt_slam/PCFilter.h:29-56) with
 * PCLUtil::distance_points (include/ndt_slam/PCLUtil.h:21-23) on z = 0 clouds: all pairs, float32
 * distance, strict <, input order kept.  out_xy must hold 2*n_base floats; returns the count. */
size_t ndt_oracle_remove_neighbors(const float *base_xy, size_t n_base, const float *list_xy, size_t n_list,
                                   double thre_neighbor, float *out_xy);

/* SURVEY.md 8f row f3 (rest), oracle/ndt_oracle_octree.c: PCFilter::difference_extraction
 * (include/ndt_slam/PCFilter.h:58-94) on a literal two-buffer pointer octree restated from PCL's published
 * OctreePointCloudChangeDetector, and Submap::makeMap (src/PointCloudMap.cpp:15-39) on top of it.
 * Both return (size_t)-1 when the clouds span more than 2^30 voxels of `resol`. */
size_t ndt_oracle_difference_indices(const float *base_xy, size_t n_base, const float *test_xy, size_t n_test,
                                     double resol, int *out_idx);
size_t ndt_oracle_difference_extraction(const float *base_xy, size_t n_base, const float *test_xy, size_t n_test,
                                        double resol, float *out_xy);
size_t ndt_oracle_make_map(const float *scans_xy, const size_t *offsets, int n_scans, int first_submap, int newest,
                           int remove_moving, double resol, double thre_neighbor, float *out_xy);


/* ---- pin points and hooks (tests/test_eigen_pins.py; see the end of ndt_oracle.c) ---- */
typedef struct ndt_oracle_hooks {
  void (*solve)(const double H[9], const double b[3], double x[3]);   /* replaces ndt_oracle_solve3 in the Newton step */
  void (*init_p)(const float T[4] /* c, s, tx, ty */, double p[3]);     /* may overwrite the initial parameter vector  */
} ndt_oracle_hooks;
void ndt_oracle_set_hooks(const ndt_oracle_hooks *h);                   /* NULL: none.  Not thread safe */
int  ndt_oracle_leaf(const ndt_oracle_params *prm, int n, const double sums[6] /* sx sy sxx sxy syy szz */,
                     double mean[2], double icov[3]);                   /* 1 accepted, 0 / -1 rejected */
void ndt_oracle_inv3(const double m[9], double out[9]);
void ndt_oracle_init_guess(const ndt_oracle_params *prm, const double init[3], float T[4], double p[3]);
void ndt_oracle_step_matrix(const ndt_oracle_params *prm, const double p[3], float T[4]);
void ndt_oracle_map_override_cells(ndt_oracle_map *m, const double *mean, const double *icov, const int *npts);
void ndt_oracle_map_export_sums(const ndt_oracle_map *m, double *out /* n_cells x 7 */);

#ifdef __cplusplus
}
#endif
#endif
