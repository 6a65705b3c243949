#pragma once
#include <pcl/point_cloud.h>
namespace pcl { template <typename P> struct ApproximateVoxelGrid {
  void setLeafSize(float, float, float) {}
  void setInputCloud(const typename PointCloud<P>::Ptr &) {}
  void filter(PointCloud<P> &) {} }; }
