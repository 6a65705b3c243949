#pragma once
namespace pcl { struct alignas(16) PointXYZ { float x, y, z, pad_; PointXYZ() : x(0), y(0), z(0), pad_(1) {} }; }
static_assert(sizeof(pcl::PointXYZ) == 16, "pcl::PointXYZ is 16 bytes");
