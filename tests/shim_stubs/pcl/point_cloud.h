#pragma once
#include <cstdint>
#include <vector>
#include <boost/make_shared.hpp>
namespace pcl { template <typename P> struct PointCloud {
  typedef boost::shared_ptr<PointCloud<P>> Ptr;
  std::vector<P> points; std::uint32_t width = 0, height = 0; bool is_dense = true;
  std::size_t size() const { return points.size(); } bool empty() const { return points.empty(); } }; }
