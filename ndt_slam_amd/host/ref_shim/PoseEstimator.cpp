// The replacement PoseEstimator of INTEGRATION.md (sections 1 and 2), kept as a file so that it is compiled
// (tests/test_ref_shim_compiles.py); the text below the marker is identical to the listing there.
// ---- listing ----
#include "PoseEstimator.h"
#include <cmath>

// PoseEstimator.h:91-128 -- double -> float32 copy of the scan points, z = 0 (row a0)
static void lps_to_cloud(const std::vector<LPoint2D> &lps, pcl::PointCloud<pcl::PointXYZ> &cloud) {
  cloud.points.resize(lps.size());
  cloud.width = (uint32_t)lps.size();
  cloud.height = 1;
  cloud.is_dense = false;
  size_t k = 0;
  for (const LPoint2D &lp : lps) {
    pcl::PointXYZ &q = cloud.points[k++];
    q.x = (float)lp.x; q.y = (float)lp.y; q.z = 0.f;
  }
}

void PoseEstimator::setScanPair(const Scan2D *curScan, pcl::PointCloud<pcl::PointXYZ>::Ptr refScan) {
  lps_to_cloud(curScan->lps, *source_cloud);
  target_cloud = refScan;                     // shared with PointCloudMap::localMap_cloud, refilled every scan
}

void PoseEstimator::setScanPair(const Scan2D *curScan, const Scan2D *refScan) {
  lps_to_cloud(curScan->lps, *source_cloud);
  lps_to_cloud(refScan->lps, *target_cloud);
}

double PoseEstimator::estimatePose(Pose2D &initPose, Pose2D &estPose, Eigen::Matrix3d &cov) {
  const double kFailed = 10000000;                       // reference :44-46
  if (!ctx || target_cloud->empty() || source_cloud->empty()) return kFailed;

  // :6-10 -- unchanged: PCL's approximate voxel filter on the host (or ndt_prefilter(): same output on the device)
  pcl::PointCloud<pcl::PointXYZ>::Ptr filtered_cloud(new pcl::PointCloud<pcl::PointXYZ>);
  pcl::ApproximateVoxelGrid<pcl::PointXYZ> approximate_voxel_filter;
  approximate_voxel_filter.setLeafSize(LeafSize, LeafSize, LeafSize);
  approximate_voxel_filter.setInputCloud(source_cloud);
  approximate_voxel_filter.filter(*filtered_cloud);

  // :19 ndt.setInputTarget -- pcl::PointXYZ is 16 bytes {x,y,z,pad}: pass the stride, no repack.
  // The local map is refilled in place every scan (src/PointCloudMap.cpp:119-131), so rebuild.
  if (ndt_map_build(ctx, &target_cloud->points[0].x, target_cloud->size(), sizeof(pcl::PointXYZ),
                    &prm, &map) != NDT_OK) {
    ROS_ERROR("ndt_map_build: %s", ndt_last_error(ctx));
    return kFailed;
  }

  // :17 + :22-28 ndt.setInputSource / init guess / ndt.align
  const double init[3] = {initPose.tx, initPose.ty, DEG2RAD(initPose.th)};
  ndt_result r;
  if (ndt_align(ctx, map, &filtered_cloud->points[0].x, filtered_cloud->size(),
                sizeof(pcl::PointXYZ), init, &r) != NDT_OK) {
    ROS_ERROR("ndt_align: %s", ndt_last_error(ctx));
    return kFailed;
  }

  // :29-36 -- the yaw comes from the float32 matrix entries through std::asin / std::acos on a float, i.e. THIS platform's
  // asinf / acosf (glibc's are not correctly rounded: 7.5 % of the arguments differ by an ulp from the model behind
  // r.pose[2]).  Re-run here on r.T00 / r.T10 so that the drop-in reports exactly what the reference would on the same machine.
  const float t00 = r.T00, t10 = r.T10;
  double theta;
  if (t00 > 0 && (t10 > 0 || t10 < 0)) theta = std::asin(t10);
  else if (t00 < 0 && t10 > 0) theta = std::acos(t00);
  else theta = std::acos(t00) * (-1.0);
  estPose.setPose(r.T03, r.T13, RAD2DEG(theta));

  // :43-46
  double cost = r.fitness;
  if (!r.converged) cost = kFailed;

  // :53-64  (the theta handed to getHessian there does not enter the Hessian: PCL's computeHessian ignores its `p` argument and
  // re-uses the angle terms of the last derivative pass, SURVEY 8a row a8 -- r.H belongs to the final transformation whichever
  // asin / acos reports its angle)
  Eigen::Matrix3d hessian3d;
  hessian3d << r.H[0], r.H[1], r.H[2], r.H[3], r.H[4], r.H[5], r.H[6], r.H[7], r.H[8];
  hessian3d = -hessian3d;
  cov = hessian3d.inverse() * coeNDTCov;
  return cost;
}
