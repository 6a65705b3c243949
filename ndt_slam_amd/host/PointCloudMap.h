// PointCloudMap.h -- host-side mirror of the reference's Submap / PointCloudMap over the C ABI
// (the callers of the local-map assembly).
//
// Same member names and bookkeeping as /root/reference/include/ndt_slam/PointCloudMap.h:22-151 and
// src/PointCloudMap.cpp:4-134; the two heavy members go to the device:
//   Submap::makeMap()      -> ndt_make_map   (octree change detection + moving-object removal per scan triple)
//   Submap::filterPoints() -> ndt_prefilter  (pcl::ApproximateVoxelGrid)
// Stand-in value types as in PoseEstimator.h (ROS / PCL / Eigen are not in this image); INTEGRATION.md has the
// call a maintainer writes against the real pcl::PointCloud.
#ifndef NDT_SLAM_AMD_HOST_POINTCLOUDMAP_H_
#define NDT_SLAM_AMD_HOST_POINTCLOUDMAP_H_

#include <memory>
#include <vector>

#include "PoseEstimator.h"

namespace ndt_amd {

typedef std::shared_ptr<PointCloudXYZ> CloudPtr;

struct MapParams {                 // ROS parameters the two classes and PCFilter read in their constructors
  bool removeMoving = false;       // PointCloudMap.h:40 (launch file: true)
  double LeafSize = 0.2;           // PointCloudMap.h:40 (launch file: 0.05)
  double resol = 0.05;             // PCFilter.h:20
  double thre_neighbor = 0.1;      // PCFilter.h:20 (launch file: 0.2)
  double sepThre = 30;             // PointCloudMap.h:91 (launch file: 10)
};

class Submap {
 public:
  double atdS = 0;                 // accumulated travel distance at the start of the submap
  size_t cntS = 0;                 // first scan number
  size_t cntE = (size_t)-1;        // last scan number
  bool newest = true;
  MapParams prm;
  CloudPtr p_cloud;
  std::vector<CloudPtr> scans;

  Submap(ndt_ctx *ctx, const MapParams &p, double a = 0, size_t s = 0);
  void addPoints(CloudPtr cloud) { scans.emplace_back(cloud); }       // PointCloudMap.h:62-65
  CloudPtr filterPoints();                                            // src/PointCloudMap.cpp:4-13
  void makeMap();                                                     // src/PointCloudMap.cpp:15-39
  const char *error() const { return err_; }

 private:
  ndt_ctx *ctx_;
  const char *err_ = nullptr;
};

class PointCloudMap {
 public:
  std::vector<Pose2D> poses;
  Pose2D lastPose;
  CloudPtr globalMap_cloud, localMap_cloud;
  double atd = 0;
  std::vector<Submap> submaps;
  std::vector<CloudPtr> maps;
  MapParams prm;

  explicit PointCloudMap(int device = 0, const MapParams &p = MapParams());
  ~PointCloudMap();
  PointCloudMap(const PointCloudMap &) = delete;
  PointCloudMap &operator=(const PointCloudMap &) = delete;

  void setLastPose(const Pose2D &p) { lastPose = p; }
  Pose2D getLastPose() const { return lastPose; }
  void addPose(const Pose2D &p);                                      // src/PointCloudMap.cpp:44-56
  void addPoints(const std::vector<LPoint2D> &lps);                   // :59-96
  void makeGlobalMap();                                               // :101-117
  void makeLocalMap();                                                // :119-134
  bool ok() const { return ctx_ != nullptr; }

 private:
  ndt_ctx *ctx_ = nullptr;
};

}  // namespace ndt_amd
#endif
