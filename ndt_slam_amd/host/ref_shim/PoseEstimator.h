// The replacement PoseEstimator of INTEGRATION.md (sections 1 and 2), kept as a file so that it is compiled
// (tests/test_ref_shim_compiles.py); the text below the marker is identical to the listing there.
// ---- listing ----
#ifndef POSEESTIMATOR_H_
#define POSEESTIMATOR_H_

#include <ros/ros.h>
#include <vector>
#include <pcl/point_types.h>
#include <pcl/point_cloud.h>
#include <pcl/filters/approximate_voxel_grid.h>   // a1 stays on the host
#include <Eigen/Dense>

#include "MyUtil.h"
#include "LPoint2D.h"
#include "Pose2D.h"
#include "Scan2D.h"
#include "ndt_mi355x.h"                            // this repo: include/

class PoseEstimator {
 private:
  pcl::PointCloud<pcl::PointXYZ>::Ptr source_cloud;
  pcl::PointCloud<pcl::PointXYZ>::Ptr target_cloud;
  double coeNDTCov, TransformationEpsilon, StepSize, Resolution, LeafSize;
  int MaximumIterations;
  ndt_ctx *ctx = nullptr;
  ndt_map *map = nullptr;
  ndt_params prm;

 public:
  double totalError;

  PoseEstimator() : coeNDTCov(1.0), TransformationEpsilon(0.01), StepSize(0.1), Resolution(1.0),
                    LeafSize(0.1), MaximumIterations(35) {
    ros::param::get("coeNDTCov", coeNDTCov);
    ros::param::get("TransformationEpsilon", TransformationEpsilon);
    ros::param::get("StepSize", StepSize);
    ros::param::get("Resolution", Resolution);
    ros::param::get("MaximumIterations", MaximumIterations);
    ros::param::get("LeafSize", LeafSize);
    source_cloud = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
    target_cloud = boost::make_shared<pcl::PointCloud<pcl::PointXYZ>>();

    ndt_default_params(&prm);
    prm.trans_eps  = TransformationEpsilon;
    prm.step_size  = StepSize;
    prm.resolution = (float)Resolution;
    prm.max_iter   = MaximumIterations;
    prm.grid_margin = 8;                            // the local map slides (src/PointCloudMap.cpp:119-131): a voxel grid with
                                                    // 8 voxels to spare is rebuilt in place until the box has moved that far
    int device = 0;
    ros::param::get("ndt_device", device);          // one process per GPU
    if (ndt_ctx_create(device, &ctx) != NDT_OK) {
      ROS_FATAL("ndt_mi355x: %s", ndt_last_error(nullptr));   // no CPU fallback
      ctx = nullptr;
    }
  }
  ~PoseEstimator() {
    if (map) ndt_map_destroy(map);
    if (ctx) ndt_ctx_destroy(ctx);
  }
  PoseEstimator(const PoseEstimator &) = delete;
  PoseEstimator &operator=(const PoseEstimator &) = delete;

  // identical to the reference (PoseEstimator.h:91-104 and :106-128)
  void setScanPair(const Scan2D *curScan, pcl::PointCloud<pcl::PointXYZ>::Ptr refScan);
  void setScanPair(const Scan2D *curScan, const Scan2D *refScan);

  double estimatePose(Pose2D &initPose, Pose2D &estPose, Eigen::Matrix3d &cov);
};
#endif
