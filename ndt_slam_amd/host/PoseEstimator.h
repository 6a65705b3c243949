// PoseEstimator.h -- host-side mirror of the reference's PoseEstimator over the C ABI.
//
// Same member names, argument meaning and error behaviour as the reference class
// (/root/reference/include/ndt_slam/PoseEstimator.h:36-133, src/PoseEstimator.cpp:4-69).  The
// reference's own value types (Pose2D, Scan2D, pcl::PointCloud, Eigen::Matrix3d) need ROS, PCL and
// Eigen, none of which exist in this image, so this mirror carries minimal stand-ins with the
// same fields; INTEGRATION.md shows the identical shim written against the real types, which is
// what a maintainer links instead of src/PoseEstimator.cpp.
//
// Units as in the reference: poses in and out in DEGREES (Pose2D.h:14), covariance in
// (m, m, rad), return value = fitness cost in m^2, 10000000 when the match did not converge
// (src/PoseEstimator.cpp:44-46).
#ifndef NDT_SLAM_AMD_HOST_POSEESTIMATOR_H_
#define NDT_SLAM_AMD_HOST_POSEESTIMATOR_H_

#include <array>
#include <cstddef>
#include <vector>

#include "ndt_mi355x.h"

namespace ndt_amd {

struct Pose2D {                       // include/ndt_slam/Pose2D.h:11-59 (tx, ty [m], th [deg])
  double tx = 0, ty = 0, th = 0;
  double Rmat[2][2] = {{1, 0}, {0, 1}};
  Pose2D() = default;
  Pose2D(double x, double y, double a) { setPose(x, y, a); }
  void calRmat();
  void setPose(double x, double y, double a) { tx = x; ty = y; th = a; calRmat(); }
};

struct LPoint2D { int sid = -1; double x = 0, y = 0; };   // LPoint2D.h:15-22, the fields the path reads
struct Scan2D { int sid = 0; Pose2D pose; std::vector<LPoint2D> lps; };   // Scan2D.h:15-35
struct PointXYZ { float x, y, z, pad; };                   // pcl::PointXYZ: 16 bytes
typedef std::vector<PointXYZ> PointCloudXYZ;
typedef std::array<double, 9> Matrix3d;                    // row-major 3x3

class PoseEstimator {
 public:
  double totalError = 0;             // PoseEstimator.h:58 (never written by the reference either)

  // Constructor defaults of PoseEstimator.h:63-64; the launch file overrides them
  // (ndt_mapping.launch:30-36: Resolution 0.3, LeafSize 0.05).
  explicit PoseEstimator(int device = 0, double coeNDTCov = 1.0, double TransformationEpsilon = 0.01,
                         double StepSize = 0.1, double Resolution = 1.0, int MaximumIterations = 35,
                         double LeafSize = 0.1);
  ~PoseEstimator();
  PoseEstimator(const PoseEstimator &) = delete;
  PoseEstimator &operator=(const PoseEstimator &) = delete;

  // PoseEstimator.h:91-104: LPoint2D doubles -> float32 cloud (z = 0); the target is taken as is.
  void setScanPair(const Scan2D *curScan, const PointCloudXYZ *refScan);
  // PoseEstimator.h:106-128
  void setScanPair(const Scan2D *curScan, const Scan2D *refScan);
  // src/PoseEstimator.cpp:4-69
  double estimatePose(Pose2D &initPose, Pose2D &estPose, Matrix3d &cov);

  const ndt_result &lastResult() const { return last_; }
  ndt_params &params() { return prm_; }     // version switches of SURVEY.md 8c
  bool ok() const { return ctx_ != nullptr; }

 private:
  ndt_ctx *ctx_ = nullptr;
  ndt_map *map_ = nullptr;
  ndt_params prm_;
  ndt_result last_;
  double coeNDTCov_, LeafSize_;
  std::vector<float> source_;        // packed xy of the current scan
  const PointCloudXYZ *target_ = nullptr;
  PointCloudXYZ target_own_;
};

// pcl::ApproximateVoxelGrid::filter on a z = 0 cloud (src/PoseEstimator.cpp:6-10): 512-slot
// direct-mapped history, flush on collision, order dependent.  Packed xy in, packed xy out.
std::vector<float> approximateVoxelGrid(const std::vector<float> &xy, float leaf);

}  // namespace ndt_amd
#endif
