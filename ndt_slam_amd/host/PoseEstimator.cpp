// PoseEstimator.cpp -- see PoseEstimator.h.  Follows src/PoseEstimator.cpp:4-69 line by line in
// meaning; every PCL call of the reference becomes one C-ABI call.
#include "PoseEstimator.h"

#include <cmath>
#include <cstring>
#include <limits>

namespace ndt_amd {

static inline double DEG2RAD(double x) { return x * M_PI / 180; }   // MyUtil.h:22
static inline double RAD2DEG(double x) { return x * 180 / M_PI; }   // MyUtil.h:23

void Pose2D::calRmat() {                                            // Pose2D.h:43-48
  const double a = DEG2RAD(th);
  Rmat[0][0] = Rmat[1][1] = std::cos(a);
  Rmat[1][0] = std::sin(a);
  Rmat[0][1] = -Rmat[1][0];
}

PoseEstimator::PoseEstimator(int device, double coeNDTCov, double TransformationEpsilon, double StepSize,
                             double Resolution, int MaximumIterations, double LeafSize)
    : coeNDTCov_(coeNDTCov), LeafSize_(LeafSize) {
  ndt_default_params(&prm_);
  prm_.trans_eps = TransformationEpsilon;     // ndt.setTransformationEpsilon  PoseEstimator.h:77
  prm_.step_size = StepSize;                  // ndt.setStepSize               :79
  prm_.resolution = (float)Resolution;        // ndt.setResolution             :81
  prm_.max_iter = MaximumIterations;          // ndt.setMaximumIterations      :83
  prm_.grid_margin = 8;                       // sliding local map: the voxel grid keeps 8 voxels to spare (ndt_mi355x.h)
  std::memset(&last_, 0, sizeof(last_));
  if (ndt_ctx_create(device, &ctx_) != NDT_OK) ctx_ = nullptr;     // no GPU: estimatePose reports 1e7
}

PoseEstimator::~PoseEstimator() {
  if (map_) ndt_map_destroy(map_);
  if (ctx_) ndt_ctx_destroy(ctx_);
}

void PoseEstimator::setScanPair(const Scan2D *curScan, const PointCloudXYZ *refScan) {
  source_.resize(2 * curScan->lps.size());
  for (size_t i = 0; i < curScan->lps.size(); ++i) {                // double -> float32, z = 0
    source_[2 * i] = (float)curScan->lps[i].x;
    source_[2 * i + 1] = (float)curScan->lps[i].y;
  }
  target_ = refScan;
}

void PoseEstimator::setScanPair(const Scan2D *curScan, const Scan2D *refScan) {
  target_own_.resize(refScan->lps.size());
  for (size_t i = 0; i < refScan->lps.size(); ++i)
    target_own_[i] = PointXYZ{(float)refScan->lps[i].x, (float)refScan->lps[i].y, 0.f, 1.f};
  setScanPair(curScan, &target_own_);
}

std::vector<float> approximateVoxelGrid(const std::vector<float> &xy, float leaf) {
  struct He { int ix, iy, count; float cx, cy; };
  He hist[512];
  std::memset(hist, 0, sizeof(hist));
  const float inv = 1.0f / leaf;
  std::vector<float> out;
  out.reserve(xy.size());
  const size_t n = xy.size() / 2;
  for (size_t i = 0; i < n; ++i) {
    const float x = xy[2 * i], y = xy[2 * i + 1];
    const int ix = (int)std::floor(x * inv), iy = (int)std::floor(y * inv);
    He &e = hist[(unsigned)((ix * 7171 + iy * 3079) & 511)];        // iz = 0
    if (e.count && (ix != e.ix || iy != e.iy)) {                    // collision: flush the old centroid
      out.push_back(e.cx / (float)e.count); out.push_back(e.cy / (float)e.count);
      e.count = 0; e.cx = 0.f; e.cy = 0.f;
    }
    e.ix = ix; e.iy = iy; e.count++;
    e.cx += x; e.cy += y;
  }
  for (int h = 0; h < 512; ++h)
    if (hist[h].count) { out.push_back(hist[h].cx / (float)hist[h].count); out.push_back(hist[h].cy / (float)hist[h].count); }
  return out;
}

double PoseEstimator::estimatePose(Pose2D &initPose, Pose2D &estPose, Matrix3d &cov) {
  const double kFailed = 10000000;                                   // src/PoseEstimator.cpp:45
  for (double &c : cov) c = std::numeric_limits<double>::quiet_NaN();
  if (!ctx_ || !target_ || target_->empty() || source_.empty()) return kFailed;
  // :6-10  approximate voxel filter of the source cloud, on the device
  std::vector<float> filtered(source_.size());
  size_t nf = 0;
  if (ndt_prefilter(ctx_, source_.data(), source_.size() / 2, 8, (float)LeafSize_, filtered.data(), &nf) != NDT_OK) return kFailed;
  filtered.resize(2 * nf);
  // :17-19 setInputSource / setInputTarget -- the target is rebuilt on every call, as the
  // reference does (its local map is refilled in place each scan, src/PointCloudMap.cpp:119-131)
  if (ndt_map_build(ctx_, &(*target_)[0].x, target_->size(), sizeof(PointXYZ), &prm_, &map_) != NDT_OK) return kFailed;
  // :22-28 init guess from the odometry prediction (degrees -> radians), align
  const double init[3] = {initPose.tx, initPose.ty, DEG2RAD(initPose.th)};
  if (ndt_align(ctx_, map_, filtered.data(), filtered.size() / 2, 8, init, &last_) != NDT_OK) return kFailed;
  // :29-36 pose from the float32 matrix: the reference's branches with THIS platform's asinf / acosf (std::asin / std::acos on
  // a float), not the correctly-rounded model behind last_.pose[2] -- what the reference reports on the same machine
  const float t00 = last_.T00, t10 = last_.T10;
  double theta;
  if (t00 > 0 && (t10 > 0 || t10 < 0)) theta = std::asin(t10);
  else if (t00 < 0 && t10 > 0) theta = std::acos(t00);
  else theta = std::acos(t00) * (-1.0);
  estPose.setPose(last_.T03, last_.T13, RAD2DEG(theta));
  // :43-46 fitness score, sentinel when not converged
  double cost = last_.fitness;
  if (!last_.converged) cost = kFailed;
  // (the theta the reference hands to getHessian at :53-56 does not enter the Hessian: PCL's computeHessian ignores its `p`
  //  argument and re-uses the angle terms of the LAST derivative pass -- those of the fp64 parameter vector, SURVEY 8a row a8 --
  //  so neither yaw enters last_.H: the covariance belongs to the final transformation, whichever asin / acos reports its angle)
  // :53-64 covariance = (-H)^-1 * coeNDTCov (fixed-size 3x3 inverse by cofactors, as Eigen does)
  double h[9];
  for (int i = 0; i < 9; ++i) h[i] = -last_.H[i];
  const double c00 = h[4] * h[8] - h[5] * h[7], c01 = h[5] * h[6] - h[3] * h[8], c02 = h[3] * h[7] - h[4] * h[6];
  const double det = h[0] * c00 + h[1] * c01 + h[2] * c02;
  const double inv[9] = {c00 / det, (h[2] * h[7] - h[1] * h[8]) / det, (h[1] * h[5] - h[2] * h[4]) / det,
                         c01 / det, (h[0] * h[8] - h[2] * h[6]) / det, (h[2] * h[3] - h[0] * h[5]) / det,
                         c02 / det, (h[1] * h[6] - h[0] * h[7]) / det, (h[0] * h[4] - h[1] * h[3]) / det};
  for (int i = 0; i < 9; ++i) cov[i] = inv[i] * coeNDTCov_;           // singular H: inf / NaN, as in the reference
  return cost;
}

}  // namespace ndt_amd
