// ndt_common.hip.h -- device-side views of the map and small helpers shared by all kernels.
// Part of libndt_mi355x.so: included by ndt_mi355x.hip inside its anonymous namespace (one translation
// unit; the order of the includes matters).  Not a standalone header.

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
  const unsigned long long *tiles;   // which voxels hold raw points, 8 x 8 voxels per word (bit 8 (y & 7) + (x & 7)), tile
  int tiles_w;                       // (tx, ty) at [(ty + 1) tiles_w + tx + 1]: one tile of zeros all around (a7, far queries)
  const int *pt_start;               // div_x*div_y + 1 bucket offsets of the raw points
  const float2 *pts;                 // raw points bucketed by voxel, input order kept (a7)
  double d1, d2;                     // Gaussian constants (a3)
  double e_hi;                       // pairs with exp(..) > e_hi fail updateDerivatives' `d2 e in [0, 1]` check (ndt_point.hip.h)
};

struct OptParams {
  double step_size, trans_eps, snap_thresh, mt_mu, mt_nu;
  int max_iter, conv_ge, stale_h_ang, mt_max_iter;
  int libm_f32;                 // ndt_params::libm_f32: the float32 cos / sin of a trial as glibc computes them (ndt_libm_f32.hip.h)
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
  int need_tf;                  // set_trial: bit 0 = the trial's transforms are still to be formed (trial_transforms), bit 1 = with the Hessian angle terms
};

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------

// Loads through a pointer that is known to address device memory.  A pointer that reached a function through LDS
// (pass_units) has lost its address space: the compiler then emits flat_load, which counts on vmcnt AND lgkmcnt,
// so every LDS read behind it waits for the memory load as well (the software prefetch of the point loop was
// waited for at once).  These helpers put the address space back: global_load.
#define NDT_GLOBAL __attribute__((address_space(1)))
typedef float ndt_f2v __attribute__((ext_vector_type(2)));
typedef float ndt_f4v __attribute__((ext_vector_type(4)));
typedef double ndt_d2v __attribute__((ext_vector_type(2)));
typedef int ndt_i4v __attribute__((ext_vector_type(4)));
typedef int ndt_i2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 gld_f2(const float2 *p) {
  const ndt_f2v v = *(const NDT_GLOBAL ndt_f2v *)p;
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 gld_f4(const float4 *p) {
  const ndt_f4v v = *(const NDT_GLOBAL ndt_f4v *)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
// the same at a 32-bit byte offset from a wave-uniform base: global_load vdst, voffset, s[base] -- one VALU
// instruction per address instead of three (sign extension + 64-bit add) when a lane walks its own range
__device__ __forceinline__ float2 gld_f2_at(const void *base, unsigned byte_off) {
  const ndt_f2v v = *(const NDT_GLOBAL ndt_f2v *)((const NDT_GLOBAL char *)base + byte_off);
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 gld_f4_at(const void *base, unsigned byte_off) {
  const ndt_f4v v = *(const NDT_GLOBAL ndt_f4v *)((const NDT_GLOBAL char *)base + byte_off);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double2 gld_d2(const double *p) {
  const ndt_d2v v = *(const NDT_GLOBAL ndt_d2v *)p;
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ double gld_d(const double *p) { return *(const NDT_GLOBAL double *)p; }
__device__ __forceinline__ int gld_i(const int *p) { return *(const NDT_GLOBAL int *)p; }
// Inclusive prefix sum over the 64 lanes of a wave with DPP moves only (row shifts with zero fill, then the row totals
// carried into the next rows): no LDS round trips.
__device__ __forceinline__ unsigned wave_incl_scan(unsigned x) {
#define NDT_DPP_ADD(CTRL, ROWS) x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xF, true)
  NDT_DPP_ADD(0x111, 0xF);     // row_shr:1
  NDT_DPP_ADD(0x112, 0xF);     // row_shr:2
  NDT_DPP_ADD(0x114, 0xF);     // row_shr:4
  NDT_DPP_ADD(0x118, 0xF);     // row_shr:8
  NDT_DPP_ADD(0x142, 0xA);     // row_bcast15: the total of rows 0 / 2 into rows 1 / 3
  NDT_DPP_ADD(0x143, 0xC);     // row_bcast31: the total of rows 0 + 1 into rows 2, 3
#undef NDT_DPP_ADD
  return x;
}

// min / max over the 64 lanes with DPP moves (row shifts, then the rows' results carried along): valid in lane 63
__device__ __forceinline__ int wave_min_dpp(int x) {
#define NDT_DPP_MIN(CTRL, ROWS) { const int t_ = __builtin_amdgcn_update_dpp(x, x, CTRL, ROWS, 0xF, false); x = t_ < x ? t_ : x; }
  NDT_DPP_MIN(0x111, 0xF) NDT_DPP_MIN(0x112, 0xF) NDT_DPP_MIN(0x114, 0xF) NDT_DPP_MIN(0x118, 0xF) NDT_DPP_MIN(0x142, 0xA) NDT_DPP_MIN(0x143, 0xC)
#undef NDT_DPP_MIN
  return x;
}
__device__ __forceinline__ int wave_max_dpp(int x) {
#define NDT_DPP_MAX(CTRL, ROWS) { const int t_ = __builtin_amdgcn_update_dpp(x, x, CTRL, ROWS, 0xF, false); x = t_ > x ? t_ : x; }
  NDT_DPP_MAX(0x111, 0xF) NDT_DPP_MAX(0x112, 0xF) NDT_DPP_MAX(0x114, 0xF) NDT_DPP_MAX(0x118, 0xF) NDT_DPP_MAX(0x142, 0xA) NDT_DPP_MAX(0x143, 0xC)
#undef NDT_DPP_MAX
  return x;
}
__device__ __forceinline__ unsigned long long gld_u64(const unsigned long long *p) { return *(const NDT_GLOBAL unsigned long long *)p; }

__device__ __forceinline__ float2 load_pt(const float *xy, size_t stride, size_t i) {
  return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(xy) + i * stride);
}

// float32 matrix of the fp64 parameter vector (a4): Translation3f(float(p0), float(p1), 0) *
// AngleAxisf(float(p2), Z); std::cos / std::sin(float): glibc's cosf / sinf (libm_f32) or modelled as correctly rounded.
__device__ __forceinline__ Tf32 tf_from_p(const double p[3], int libm_f32) {
  Tf32 t;
  float yaw = (float)p[2];
  if (libm_f32) {
    t.c = sincosf_glibc(yaw, 1);
    t.s = sincosf_glibc(yaw, 0);
  } else {
    double sd, cd;
    sincos_small((double)yaw, sd, cd);
    t.c = (float)cd;
    t.s = (float)sd;
  }
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
  else { sincos_small(yaw, s, c); }
}
