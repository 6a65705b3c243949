/*
 * ndt_mi355x.h -- C ABI of the MI355X-native NDT scan-matching core (libndt_mi355x.so).
 *
 * Drop-in boundary.  The reference has no FFI: its hot path sits behind the
 * concrete C++ class PoseEstimator (include/ndt_slam/PoseEstimator.h:36-133), injected by
 * pointer (src/SlamLauncher.cpp:12 -> include/ndt_slam/FrontEnd.h:74-76 ->
 * include/ndt_slam/ScanMatcher.h:58-60) and called only from ScanMatcher::matchScan
 * (src/ScanMatcher.cpp:40,45).  A replacement PoseEstimator.{h,cpp} binds exactly the entry
 * points below; INTEGRATION.md shows that binding.  Each entry point names the reference
 * interface (file:line) it replaces.  Plain pointers and sizes only; never throws; every
 * function returns 0 on success or a negative ndt_status, with text in ndt_last_error().
 *
 * Threading: a context is bound to one device and one host thread; all GPU work runs on the
 * context's stream (or the stream passed to the *_dev calls).  Host-pointer calls are
 * synchronous on return; *_dev calls are asynchronous on their stream.  A context owns one set of
 * scratch buffers: *_dev calls of ONE context on different streams are serialised by the library (the
 * later call's stream waits for the earlier call's kernels); to have two batches in flight at the
 * same time use two contexts (a map may be read by any context of its device, from its own host thread: the
 * map's bookkeeping of who reads it is locked; a map is REBUILT from one thread at a time, that of its context).
 */
#ifndef NDT_MI355X_H_
#define NDT_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum ndt_status {
  NDT_OK = 0,
  NDT_E_ARG = -1,        /* null / empty / inconsistent arguments                     */
  NDT_E_HIP = -2,        /* a HIP runtime call failed (text in ndt_last_error)        */
  NDT_E_NO_DEVICE = -3,  /* no gfx950 device: the library never falls back to the CPU */
  NDT_E_GRID = -4,       /* map extent / resolution needs more than 2^28 voxels       */
  NDT_E_NOMEM = -5
};

/* Parameters of the path.  The first four are the ROS parameters the reference's constructor
 * reads and hands to PCL (include/ndt_slam/PoseEstimator.h:63-84; values in
 * ndt_mapping.launch:32-36); the rest are PCL defaults and the version-sensitive switches of
 * SURVEY.md 8c; ndt_default_params = ndt_params_pcl110 (presets below). */
typedef struct ndt_params {
  float  resolution;        /* PoseEstimator.h:81  ndt.setResolution             */
  double step_size;         /* PoseEstimator.h:79  ndt.setStepSize               */
  double trans_eps;         /* PoseEstimator.h:77  ndt.setTransformationEpsilon  */
  int    max_iter;          /* PoseEstimator.h:83  ndt.setMaximumIterations      */
  double outlier_ratio;     /* 0.55 */
  int    min_pts;           /* 6    */
  double eig_mult;          /* 0.01 */
  int    cov_unbiased;      /* 0: (Sxx/n - mu mu^T)(n-1)/n ; 1: /(n-1)           */
  int    cov_init_identity; /* 1: per-voxel Sxx accumulator starts at I          */
  int    conv_ge;           /* 0: stop when iter > max_iter ; 1: >=              */
  int    radius_inclusive;  /* 0: d^2 < r^2 ; 1: <=                              */
  int    transform_sse;     /* 0: (m00 x + m01 y) + m03 ; 1: m00 x + (m01 y + m03) */
  int    stale_h_ang;       /* 0 (every preset): the Hessian after an inner line search uses the
                                  2nd-derivative angle terms of the LAST trial (PCL refreshes them on
                                  every computeDerivatives); 1: those of the line search's first trial */
  double snap_thresh;       /* 10e-5 */
  int    mt_max_iter;       /* 10    */
  double mt_mu;             /* 1e-4  */
  double mt_nu;             /* 0.9   */
  int    libm_f32;          /* float32 cos / sin of a trial's yaw, the entries of final_transformation_ (Eigen's
                               AngleAxisf::toRotationMatrix calls std::cos / std::sin on a float):
                               1: glibc >= 2.28's cosf / sinf (x86-64, FMA build: what Ubuntu 20.04 / 22.04 run), restated
                                  operation for operation and equal to libm on all 2.2e9 floats |x| < 120
                                  (tests/test_libm_f32.py); 0: correctly rounded (differs from glibc by one ulp in 1.3 % of
                                  the angles).  asinf / acosf / atan2f stay modelled as correctly rounded: DESIGN.md 2 */
  int    grid_margin;       /* 0 (every preset): the voxel grid is the one PCL's VoxelGridCovariance derives from the cloud's
                               bounding box, and a rebuild whose box has moved by a voxel is queued twice (NDT_REBUILT).
                               m > 0: a grid built or re-queued for this map is that box widened by m voxels on every side,
                               and ndt_map_rebuild_end accepts the grid queued ahead as long as it still contains the
                               cloud's box and is at most 2 m voxels wider on any side -- a sliding local map
                               (src/PointCloudMap.cpp:119-131) then pays the second build once per ~m voxels of travel
                               instead of once per voxel.  Matches, fitness scores and ndt_eval_at do not depend on it,
                               to the last bit (same voxels, same statistics, and the order of every sum follows the scan
                               and the pose, never the grid's extent: tests/test_gpu_parity.py); ndt_map_info and
                               ndt_map_export describe the widened grid */
} ndt_params;

/* Result of one scan-to-map match = everything src/PoseEstimator.cpp:28-64 reads back from
 * PCL after ndt.align(). */
typedef struct ndt_result {
  double pose[3];     /* x, y, yaw[rad] by the asin/acos branch logic of src/PoseEstimator.cpp:31-35
                         applied to the float32 matrix entries below                               */
  float  T00, T10, T03, T13; /* getFinalTransformation() entries (src/PoseEstimator.cpp:29)        */
  double fitness;     /* getFitnessScore() (src/PoseEstimator.cpp:43), m^2; the 1e7 sentinel of
                         :44-46 is applied by the shim, not here                                   */
  double trans_prob;  /* getTransformationProbability() (src/PoseEstimator.cpp:48)                 */
  double score;
  double H[9];        /* rows/cols {0,1,5} of getHessian()'s 6x6 (src/PoseEstimator.cpp:53-61),
                         row-major, before the sign flip of :61                                    */
  double p[3];        /* final fp64 parameter vector (tx, ty, yaw)                                 */
  int    iters;       /* outer Newton iterations                                                   */
  int    evals;       /* derivative passes this library executed                                   */
  int    ref_evals;   /* passes the reference executes on the same path: adds its Hessian-only passes and the
                         getHessian pass (fused here) and the trials of a line search that repeat the step length
                         of the pass before them (same pose, same totals: not run again here)      */
  int    converged;   /* hasConverged() (src/PoseEstimator.cpp:44)                                 */
  int    status;      /* ndt_status of this match                                                  */
  int    flags;       /* NDT_FLAG_* bits: which data path the match took (results do not depend on it)    */
  double kbar;        /* mean in-radius cells per point-evaluation (roofline accounting)           */
} ndt_result;

/* ndt_result.flags */
enum ndt_result_flags {
  NDT_FLAG_WINDOW_SPILL  = 1,  /* occupied voxels of the scan's window had no LDS record: points that reached
                                  them read the cell table from HBM                                          */
  NDT_FLAG_REGION_CLIPPED = 2, /* the scan's voxel bounding box exceeded the LDS window (16384 cells): points
                                  outside it read the cell table from HBM                                    */
  NDT_FLAG_UNSORTED      = 4   /* scan above 20000 points: passes read it in input order                     */
};

typedef struct ndt_map_info {
  int min_bx, min_by, div_x, div_y;
  int n_cells;   /* voxels with >= min_pts points */
  int n_valid;   /* of those, accepted covariances */
  size_t n_points;
} ndt_map_info;

typedef struct ndt_ctx ndt_ctx;
typedef struct ndt_map ndt_map;

/* include/ndt_slam/PoseEstimator.h:63-64 constructor defaults + the PCL-side values, as one of three
 * presets of the version-sensitive switches (PCL is un-vendored and unpinned, CMakeLists.txt:21; the
 * reference compiles only against PCL <= 1.10, include/ndt_slam/PoseEstimator.h:72-73):
 *   ndt_params_pcl110   PCL 1.9 / 1.10 (Ubuntu 20.04, the likely build): VoxelGridCovariance::Leaf() starts
 *                       cov_ at the identity (cov_init_identity = 1), (n-1)/n normalisation
 *                       (cov_unbiased = 0), SSE transformPointCloud (transform_sse = 1), glibc 2.31 (libm_f32 = 1)
 *   ndt_params_pcl18    PCL <= 1.8: the same voxel statistics, scalar transformPointCloud (transform_sse = 0); its
 *                       platform (Ubuntu 18.04, glibc 2.27) has an older sinf / cosf: modelled (libm_f32 = 0)
 *   ndt_params_pcl_new  PCL >= 1.11: cov_ starts at zero, /(n-1) (cov_unbiased = 1), SSE transform, libm_f32 = 1
 * ndt_default_params is ndt_params_pcl110. */
int ndt_default_params(ndt_params *p);
int ndt_params_pcl110(ndt_params *p);
int ndt_params_pcl18(ndt_params *p);
int ndt_params_pcl_new(ndt_params *p);

/* One context per process and device (one process per GPU).  device = HIP ordinal. */
int ndt_ctx_create(int device, ndt_ctx **out);
int ndt_ctx_destroy(ndt_ctx *ctx);               /* destroy the context's maps first */
const char *ndt_last_error(const ndt_ctx *ctx);   /* ctx may be NULL: last global error */
void *ndt_ctx_stream(ndt_ctx *ctx);               /* hipStream_t the context works on   */
/* Tuning of the match launch (defaults are right for whole-GPU batches):
 *   NDT_OPT_MAX_HELPERS  0..15  workgroups that may join the passes of one unfinished scan; 0 = no work sharing;
 *                               -1 = the default: 8, or 2 when the launch has a scan for every workgroup
 *   NDT_OPT_WORKGROUPS   0..#CU workgroups per match launch (0 = one per CU); a smaller value leaves CUs to
 *                               other streams
 * Results never depend on either (unit totals are summed in unit order whoever computed them).
 *   NDT_OPT_INJECT_FAULT k      test instrumentation for the error paths: the k-th next match launch of the context returns
 *                               NDT_E_HIP right behind the dispatch of its first kernel (0 = off, the default).  The call
 *                               leaves the context usable: the kernel that was queued is ordered in front of whatever any
 *                               stream does with the context next.
 *   NDT_OPT_DEFER_FITNESS 0 / 1 for callers with a stream of batches (ndt_align_batch_dev only; default 0).  1: the call queues the
 *                               match kernel on the caller's stream and the fitness kernels on a stream of the context's own
 *                               behind it, so the caller's NEXT launch starts its match kernel at once and the fitness
 *                               kernels fill the CUs its idle workgroups leave.  The records of a launch -- their `fitness`
 *                               field above all -- are then complete at the LAUNCH'S END, not at the caller's stream's
 *                               position behind the call: wait for it with ndt_ctx_wait_launch (or a device-wide
 *                               synchronisation) before reading them, keep scans / offsets / initial guesses alive until
 *                               then, and give two launches in a row different `out` arrays (a launch waits for the end of
 *                               the launch before last by itself).  Every other entry point of the context waits for a
 *                               deferred launch's end before it touches the context.  Same kernels, same records. */
enum ndt_option { NDT_OPT_MAX_HELPERS = 1, NDT_OPT_WORKGROUPS = 2, NDT_OPT_INJECT_FAULT = 3, NDT_OPT_DEFER_FITNESS = 4 };
int ndt_ctx_set_option(ndt_ctx *ctx, int option, long long value);
/* Make the context work on a caller-owned hipStream_t (e.g. the stream a host framework already
 * orders its copies on); NULL restores the context's own stream. */
int ndt_ctx_set_stream(ndt_ctx *ctx, void *stream);

/* Replaces ndt.setInputTarget(target_cloud) (src/PoseEstimator.cpp:19): voxel-grid
 * normal-distributions build plus the raw-point buckets the fitness
 * score searches (replaces the kd-tree Registration::initCompute builds, row a7).
 * xy: points at `stride_bytes` (8 = packed float2, 16 = pcl::PointXYZ).  If *map is non-NULL
 * it is rebuilt in place (the reference's target cloud is refilled every scan,
 * src/PointCloudMap.cpp:119-131).  Host-pointer form copies to the device first. */
int ndt_map_build(ndt_ctx *ctx, const float *xy_host, size_t n, size_t stride_bytes,
                  const ndt_params *prm, ndt_map **map);
int ndt_map_build_dev(ndt_ctx *ctx, const float *xy_dev, size_t n, size_t stride_bytes,
                      const ndt_params *prm, ndt_map **map);
/* The same rebuild in two halves for a pipeline that must not wait on the host: _begin queues the
 * bounding box of the new cloud and, ahead of its read-back, the whole build with the voxel grid of the map's previous
 * build, and returns; launches queued behind it see that build.  _end waits for the bounding box: NDT_OK if the grid
 * was the right one (a SLAM local map keeps its voxel bounding box for many scans, src/PointCloudMap.cpp:119-131),
 * NDT_REBUILT if it was not -- the build has been queued again (behind the launches of ANY context that read the map
 * since _begin: it rewrites the tables they read), and whatever was queued between _begin and _end ran on a stale grid
 * and has to be queued again by the caller.  One _begin may be open per context (other builds on that
 * context fail with NDT_E_ARG until _end); xy_dev must stay as it is until _end has returned.  The map must have been
 * built before at the same resolution. */
#define NDT_REBUILT 1
int ndt_map_rebuild_begin(ndt_ctx *ctx, const float *xy_dev, size_t n, size_t stride_bytes,
                          const ndt_params *prm, ndt_map *map);
int ndt_map_rebuild_end(ndt_ctx *ctx, ndt_map *map);
int ndt_map_destroy(ndt_map *map);
int ndt_map_info_get(const ndt_map *map, ndt_map_info *out);
/* Cell table in ascending voxel-index order, arrays sized n_cells (parity tests). */
int ndt_map_export(const ndt_map *map, int *cell_idx, float *cent_xy, double *mean_xy,
                   double *icov_xx_xy_yy, int *npts);

/* Replaces ndt.setInputSource + ndt.align + getFinalTransformation + getFitnessScore +
 * hasConverged + getTransformationProbability + getHessian for ONE scan
 * (src/PoseEstimator.cpp:17-56).  scan = the post-filter source cloud (row a1 stays with the
 * caller); init = (tx, ty, yaw[rad]) as src/PoseEstimator.cpp:22-24 builds the guess. */
int ndt_align(ndt_ctx *ctx, const ndt_map *map, const float *scan_xy_host, size_t n,
              size_t stride_bytes, const double init_xyyaw[3], ndt_result *out);

/* Batch of B independent matches against one map (BASELINE.json configs 3-5).  Scans are
 * packed float2, concatenated; offsets[B+1] in points.  shared_scan != 0: every match uses
 * scan 0 (offsets[0..1]) with its own init pose (multi-hypothesis relocalisation).  Such launches keep a scratch
 * slot of the scan's size PER MATCH (ordered copy 8 B, distance 4 B, far-query list 4 B per point: 16 B x n x B)
 * and score the seeds that end far from the map from the map's occupancy words (same `fitness`). */
int ndt_align_batch(ndt_ctx *ctx, const ndt_map *map, const float *scans_xy_host,
                    const uint64_t *offsets_host, int B, int shared_scan,
                    const double *inits_host /* B x 3 */, ndt_result *out_host /* B */);
/* Same with every buffer resident in device memory; asynchronous on `stream`
 * (NULL = the context's stream).  out_dev receives B ndt_result records.  total_points = number
 * of points in scans_xy_dev (offsets[B]; the host knows it, the offsets live on the device): it
 * sizes the context's scratch copy in which every scan is kept in the order its lanes walk it. */
int ndt_align_batch_dev(ndt_ctx *ctx, const ndt_map *map, const float *scans_xy_dev,
                        const uint64_t *offsets_dev, int B, size_t total_points, int shared_scan,
                        const double *inits_dev, ndt_result *out_dev, void *stream);
/* Optional first half of ndt_align_batch_dev for a caller with a STREAM of batches: what an alignment needs before its first
 * derivative pass -- the optimiser's start from the initial guess (src/PoseEstimator.cpp:22-24 and computeTransformation's
 * prologue), the window of the voxel grid the scan can reach, the scan's points in voxel order -- depends
 * on the scan, its guess and the grid of `map` only.  This call queues that work for the whole batch on `stream` (NULL: the
 * context's) as a kernel of its own and returns; a later ndt_align_batch_dev(ctx, map, <the same scans, offsets, B,
 * total_points, shared_scan, inits>, ...) finds the prepared batch, orders its stream behind it and starts every scan at the
 * staging of its window.  Of `map` it uses the voxel grid's geometry as it is at the time of the call and reads nothing on the
 * device, so it needs no ordering against the map's builds: put it where the GPU has room while the previous batch's matches
 * run out (bench.py: on a stream of its own, queued before the step's rebuild).  Rules: the arrays must not change between the
 * two calls; if the map is rebuilt in between and its grid comes out different (another cloud; a two-phase rebuild whose
 * ndt_map_rebuild_end returns NDT_REBUILT) the prepared batch is simply not used; one prepared batch serves one
 * launch; two batches may be prepared ahead per context.  Results are byte-identical with and without it (the same device
 * routines run, earlier).  One scan at a time (src/ScanMatcher.cpp:40,45 as the reference calls it) gains nothing from it. */
int ndt_align_batch_prepare_dev(ndt_ctx *ctx, const ndt_map *map, const float *scans_xy_dev,
                                const uint64_t *offsets_dev, int B, size_t total_points, int shared_scan,
                                const double *inits_dev, void *stream);
/* Duration (ms) of the kernel ndt_align_batch_prepare_dev queued for the batches the context's launches used; 0 when
 * none was used.  Blocks until that kernel has run. */
int ndt_prepare_timing(ndt_ctx *ctx, float *order_ms);

/* Many matches over several GPUs from ONE process (north_star: "batch across the 8 GPUs of one node"; SURVEY.md 8b's
 * indicative multi-device context): `ctxs[r]` / `maps[r]` are a context of device r and a map built there from the same
 * cloud.  The batch is cut into n_shards contiguous, balanced shards (matches [r B / n, (r + 1) B / n), the first B % n
 * one longer -- the partition of ndt_slam_amd/shard.py), every shard is uploaded straight to its device, all launches
 * run side by side, the records come back into results[0 .. B) in batch order.  No exchange between devices: the
 * matches are independent given the read-only map, so a batch that starts on the host needs no xGMI
 * traffic at all.  `shared_scan`: the one scan goes to every device, the B seed poses are sharded.  Synchronous; same
 * results as ndt_align_batch on one device, byte for byte.  (One process per GPU over torch.distributed / RCCL is the
 * other form: ndt_slam_amd/shard.py.)  Replaces a loop of src/ScanMatcher.cpp:40,45 over independent scans.
 * Errors: every shard is attempted; the call returns the FIRST shard's error code.  The records of a shard that failed
 * are all overwritten -- zeroed, `status` = that shard's error, `converged` = 0, `fitness` = DBL_MAX (the shim's 1e7
 * case, src/PoseEstimator.cpp:44-46) -- and the records of the shards that succeeded are complete, so `status` tells
 * per record what can be used (tests/test_gpu_parity.py::test_sharded_batch_marks_every_record_of_a_failed_shard). */
int ndt_align_batch_sharded(ndt_ctx *const *ctxs, const ndt_map *const *maps, int n_shards, const float *scans_xy_host,
                            const uint64_t *offsets, int B, int shared_scan, const double *inits_xyyaw,
                            ndt_result *results);
/* As ndt_align_batch, additionally recording per derivative pass of every match
 * 8 doubles {a_t, score, g0, g1, g2, p0, p1, p2} (parity tests: same step sequence as the
 * oracle).  trace_host: B x trace_cap x 8 doubles; trace_rows_host: B ints. */
int ndt_align_batch_trace(ndt_ctx *ctx, const ndt_map *map, const float *scans_xy_host,
                          const uint64_t *offsets_host, int B, int shared_scan,
                          const double *inits_host, ndt_result *out_host, double *trace_host,
                          int trace_cap, int *trace_rows_host);

/* One derivative pass at an explicit pose (rows a4+a5; parity tests and profiling):
 * score, gradient[3], Hessian[9] of d(score)/dp at p = (tx, ty, yaw). */
int ndt_eval_at(ndt_ctx *ctx, const ndt_map *map, const float *scan_xy_host, size_t n,
                size_t stride_bytes, const double p[3], double *score, double g[3], double H[9],
                double *pairs);
/* Fitness score alone at an explicit float32 transform (row a7). */
int ndt_fitness_at(ndt_ctx *ctx, const ndt_map *map, const float *scan_xy_host, size_t n,
                   size_t stride_bytes, float c, float s, float tx, float ty, double *fitness);

/* Replaces the source pre-filter pcl::ApproximateVoxelGrid::filter (src/PoseEstimator.cpp:6-10:
 * setLeafSize(LeafSize x3), setInputCloud(source_cloud), filter(*filtered_cloud); SURVEY.md 8a row
 * a1 / 8f row f1) on z = 0 clouds: 512-slot direct-mapped voxel history, flush on collision, the
 * rest flushed in slot order -- same centroids (float32 sums in cloud order), same output order.
 * Single scan, host pointers; out_xy_host needs room for n points; *n_out = points written. */
int ndt_prefilter(ndt_ctx *ctx, const float *xy_host, size_t n, size_t stride_bytes, float leaf,
                  float *out_xy_host, size_t *n_out);
/* Batch of B raw scans resident in device memory (points at stride_bytes, raw_offsets[B+1] in
 * points).  Writes the filtered scans packed as float2 to out_xy_dev (capacity total_raw_points
 * points) and their offsets[B+1] to out_offsets_dev: exactly the inputs of ndt_align_batch_dev,
 * whose total_points may be given as total_raw_points (an upper bound).  Asynchronous on `stream`
 * (NULL = the context's stream). */
int ndt_prefilter_batch_dev(ndt_ctx *ctx, const float *raw_xy_dev, size_t stride_bytes,
                            const uint64_t *raw_offsets_dev, int B, size_t total_raw_points, float leaf,
                            float *out_xy_dev, uint64_t *out_offsets_dev, void *stream);

/* SURVEY.md 8f row f2 -- the steps either side of the match, for a batch resident in device
 * memory.  Poses are (tx, ty, th) triples of doubles with th in DEGREES (include/ndt_slam/Pose2D.h:14);
 * covariances are row-major 3x3 in (m, m, rad). */
typedef struct ndt_fuse_params {
  double coe_ndt_cov;  /* include/ndt_slam/PoseEstimator.h:63  coeNDTCov  (1.0) */
  double coe_vel;      /* include/ndt_slam/PoseFuser.h:19      coeVel     (0.1) */
  double coe_omega;    /* include/ndt_slam/PoseFuser.h:19      coeOmega   (0.1) */
  double del_time;     /* include/ndt_slam/PoseFuser.h:19      delTime    (0.5) */
  double score_thre;   /* include/ndt_slam/ScanMatcher.h:49-50 scthre     (0.0; launch file sets score_thre) */
} ndt_fuse_params;
int ndt_fuse_default_params(ndt_fuse_params *p);
/* Replaces Pose2D::calMotion (src/Pose2D.cpp:5-16) + Pose2D::calPredPose (:28-37) as
 * ScanMatcher::matchScan chains them (src/ScanMatcher.cpp:27-32): odometry motion in the robot
 * frame, predicted pose, and (if init_xyyaw_dev != NULL) the same pose as the (tx, ty, yaw[rad])
 * guess ndt_align_batch_dev takes.  All arrays B x 3 doubles in device memory; asynchronous. */
int ndt_predict_batch_dev(ndt_ctx *ctx, const double *odo_cur_dev, const double *odo_prev_dev,
                          const double *last_pose_dev, int B, double *odo_motion_dev, double *pred_pose_dev,
                          double *init_xyyaw_dev, void *stream);
/* Replaces, per match: cost with the 1e7 sentinel and Qmat = (-H)^-1 * coeNDTCov
 * (src/PoseEstimator.cpp:43-64), the accept test cost <= scthre (src/ScanMatcher.cpp:50), then
 * PoseFuser::fusePose (src/PoseFuser.cpp:3-37) or, for a rejected match, the predicted pose with
 * PoseFuser::calOdometryCovariance (src/PoseFuser.cpp:39-61; src/ScanMatcher.cpp:60-66).
 * results_dev: B records as written by ndt_align_batch_dev; last_cov_dev, cov_dev: B x 9;
 * successful_dev: B ints or NULL.  Asynchronous on `stream`. */
int ndt_fuse_batch_dev(ndt_ctx *ctx, const ndt_result *results_dev, const double *pred_pose_dev,
                       const double *odo_motion_dev, const double *last_pose_dev, const double *last_cov_dev,
                       int B, const ndt_fuse_params *prm, double *fused_pose_dev, double *cov_dev,
                       int *successful_dev, void *stream);

/* SURVEY.md 8f row f3 (the all-pairs step on its own) -- replaces PCFilter::remove_neighborPoint(cloud_base, point_list)
 * (include/ndt_slam/PCFilter.h:29-56, called from Submap::makeMap, src/PointCloudMap.cpp:27): the points
 * of `base` with no point of `list` closer than thre_neighbor (PCLUtil::distance_points' float32
 * distance, include/ndt_slam/PCLUtil.h:21-23, strict <), in input order, packed as float2.
 * out needs room for n_base points; n_list may be 0. */
int ndt_remove_neighbors(ndt_ctx *ctx, const float *base_xy_host, size_t base_stride_bytes, size_t n_base,
                         const float *list_xy_host, size_t list_stride_bytes, size_t n_list, double thre_neighbor,
                         float *out_xy_host, size_t *n_out);
/* Same with device pointers; *n_out_dev is a uint64 in device memory; asynchronous on `stream`. */
int ndt_remove_neighbors_dev(ndt_ctx *ctx, const float *base_xy_dev, size_t base_stride_bytes, size_t n_base,
                             const float *list_xy_dev, size_t list_stride_bytes, size_t n_list,
                             double thre_neighbor, float *out_xy_dev, uint64_t *n_out_dev, void *stream);

/* SURVEY.md 8f row f3 -- replaces PCFilter::difference_extraction(cloud_base, cloud_test)
 * (include/ndt_slam/PCFilter.h:58-94): the points of `test` that fall in leaf voxels (side `resol`) of
 * pcl::octree::OctreePointCloudChangeDetector which hold no point of `base`.  The voxel lattice is the
 * octree's own: anchored by the first point added (base first, then test) and carried through every
 * doubling of the bounding box with the keys computed in fp64 as PCL computes them.  z is taken as 0
 * (PointCloudMap::addPoints, src/PointCloudMap.cpp:71); non-finite points are skipped as PCL skips them.
 * The points come back in INPUT order, packed as float2; PCL returns the same set in the order of its
 * depth-first leaf walk, which nothing downstream depends on (the list only feeds remove_neighborPoint).
 * out needs room for n_test points; n_base may be 0.  NDT_E_ARG when the two clouds span more than 2^30
 * voxels per axis (PCL's own key width is 32 bits). */
int ndt_difference_extraction(ndt_ctx *ctx, const float *base_xy_host, size_t base_stride_bytes, size_t n_base,
                              const float *test_xy_host, size_t test_stride_bytes, size_t n_test, double resol,
                              float *out_xy_host, size_t *n_out);
/* Same with device pointers; *n_out_dev is a uint64 in device memory (UINT64_MAX on the span error);
 * asynchronous on `stream`. */
int ndt_difference_extraction_dev(ndt_ctx *ctx, const float *base_xy_dev, size_t base_stride_bytes, size_t n_base,
                                  const float *test_xy_dev, size_t test_stride_bytes, size_t n_test, double resol,
                                  float *out_xy_dev, uint64_t *n_out_dev, void *stream);
/* Replaces Submap::makeMap (src/PointCloudMap.cpp:15-39): the submap's cloud from its scans (already in the
 * map frame), scan i = points [offsets[i], offsets[i+1]) of scans_xy.  With remove_moving: scans[0] when
 * first_submap (cntS == 0), then for every triple (i, i+1, i+2) the points of scan i+1 that are not within
 * thre_neighbor of a point of difference_extraction(scan i ++ scan i+2, scan i+1), then the last scan when
 * `newest`; without: all scans (first submap) or scans 2.. (later ones).  All triples run side by side, one
 * workgroup each.  out needs room for every input point (twice that for n_scans == 1: the reference then
 * appends the lone scan as the first and again as the newest).  resol / thre_neighbor are PCFilter's `resol` (0.05) and `thre_neighbor` (0.1),
 * include/ndt_slam/PCFilter.h:20-23. */
int ndt_make_map(ndt_ctx *ctx, const float *scans_xy_host, size_t stride_bytes, const uint64_t *offsets, int n_scans,
                 int first_submap, int newest, int remove_moving, double resol, double thre_neighbor,
                 float *out_xy_host, size_t *n_out);
/* Same with the points in device memory (offsets stay on the host); *n_out_dev is a uint64 in device memory
 * (UINT64_MAX on the span error); asynchronous on `stream`.  The result can be handed to
 * ndt_prefilter_batch_dev / ndt_map_build_dev without leaving the device. */
int ndt_make_map_dev(ndt_ctx *ctx, const float *scans_xy_dev, size_t stride_bytes, const uint64_t *offsets,
                     int n_scans, int first_submap, int newest, int remove_moving, double resol,
                     double thre_neighbor, float *out_xy_dev, uint64_t *n_out_dev, void *stream);

/* Durations of the kernels of one of the context's last 64 match launches (`back` = 0: the most recent one):
 * the match kernel (rows a3-a6, a8, a9: start to stop of that kernel) and the fitness kernels behind it (row a7: stop of the
 * match kernel to stop of the last fitness kernel), from HIP events attached to the kernels' own dispatches on the launch's
 * stream.  Blocks until that launch has finished. */
int ndt_kernel_timing(ndt_ctx *ctx, int back, float *match_ms, float *fitness_ms);
/* Time from the start of the match kernel of launch `back + 1` to the start of the match kernel of launch `back` of
 * this context, from the same events: the step-to-step interval of a caller that issues launches back to back
 * (bench.py: ms_per_step_min / _max over the timed steps, without a further event record on the match stream).
 * Both launches must still be among the last 64.  Replaces the reference's per-scan "align time" log
 * (src/PoseEstimator.cpp:15,38-40). */
int ndt_launch_interval(ndt_ctx *ctx, int back, float *interval_ms);
/* Order another stream behind one of the context's last 64 match launches (`back` as above), fitness kernels included:
 * `stream` (hipStream_t; NULL = the context's stream) waits for the event attached to that launch's last kernel.  What a
 * caller would otherwise do with hipEventRecord on the launch's stream -- a packet of its own between two kernels
 * (6 us per launch on the stream that carries the matches) -- e.g. before the map the launch read is rebuilt on
 * another stream (the reference refills its target cloud every scan, src/ScanMatcher.cpp:40). */
int ndt_ctx_wait_launch(ndt_ctx *ctx, int back, void *stream);

/* Timing hooks used by bench.py (HIP events on the context's stream; milliseconds of the most
 * recent call of each kind, measured around the kernel launches only). */
int ndt_last_timing(const ndt_ctx *ctx, float *map_build_ms, float *align_ms);

/* Self-test of the device's float32 routines (ndt_params::libm_f32 = 1; ndt_slam_amd/csrc/ndt_libm_f32.hip.h): for n yaws
 * (host array, float32) the device's restatements of glibc's cosf / sinf and of the initial yaw computeTransformation reads
 * back from the guess matrix -- Eigen's Affine3f.rotation().eulerAngles(0,1,2)[2] with glibc's atan2f, from the cos / sin
 * just computed (src/PoseEstimator.cpp:22-24 -> the prologue of ndt.align, :28).  A caller on another platform checks them
 * against its own libm with this before trusting bit-level parity (tests/test_gpu_parity.py does, on 2e6 yaws).
 * Any output pointer may be NULL. */
int ndt_selftest_libm_f32(ndt_ctx *ctx, const float *yaw_host, size_t n, float *cos_out, float *sin_out, float *init_yaw_out);

#ifdef __cplusplus
}
#endif
#endif
