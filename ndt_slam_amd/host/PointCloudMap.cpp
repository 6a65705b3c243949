// PointCloudMap.cpp -- see PointCloudMap.h.  No GPU code here: the C ABI of libndt_mi355x.so does the work.
#include "PointCloudMap.h"

#include <cmath>
#include <cstdint>

namespace ndt_amd {

Submap::Submap(ndt_ctx *ctx, const MapParams &p, double a, size_t s) : atdS(a), cntS(s), prm(p), ctx_(ctx) {
  p_cloud = std::make_shared<PointCloudXYZ>();
}

static CloudPtr cloud_from_xy(const std::vector<float> &xy, size_t n) {
  CloudPtr c = std::make_shared<PointCloudXYZ>(n);
  for (size_t i = 0; i < n; ++i) (*c)[i] = PointXYZ{xy[2 * i], xy[2 * i + 1], 0.f, 1.f};
  return c;
}

CloudPtr Submap::filterPoints() {
  const size_t n = p_cloud->size();
  if (n == 0) return std::make_shared<PointCloudXYZ>();
  std::vector<float> out(2 * n);
  size_t m = 0;
  if (ndt_prefilter(ctx_, &(*p_cloud)[0].x, n, sizeof(PointXYZ), (float)prm.LeafSize, out.data(), &m) != NDT_OK) {
    err_ = ndt_last_error(ctx_);
    return std::make_shared<PointCloudXYZ>();
  }
  return cloud_from_xy(out, m);
}

void Submap::makeMap() {
  // the scans of the submap, packed one after the other as the C ABI takes them
  std::vector<uint64_t> off(scans.size() + 1, 0);
  for (size_t i = 0; i < scans.size(); ++i) off[i + 1] = off[i] + scans[i]->size();
  const size_t total = (size_t)off.back();
  std::vector<PointXYZ> all(total ? total : 1);
  for (size_t i = 0; i < scans.size(); ++i)
    for (size_t k = 0; k < scans[i]->size(); ++k) all[(size_t)off[i] + k] = (*scans[i])[k];
  std::vector<float> out(2 * (scans.size() == 1 ? 2 * total : total) + 2);
  size_t m = 0;
  p_cloud->clear();
  if (total == 0) return;
  if (ndt_make_map(ctx_, &all[0].x, sizeof(PointXYZ), off.data(), (int)scans.size(), cntS == 0, newest, prm.removeMoving,
                   prm.resol, prm.thre_neighbor, out.data(), &m) != NDT_OK) {
    err_ = ndt_last_error(ctx_);
    return;
  }
  p_cloud = cloud_from_xy(out, m);
}

PointCloudMap::PointCloudMap(int device, const MapParams &p) : prm(p) {
  if (ndt_ctx_create(device, &ctx_) != NDT_OK) ctx_ = nullptr;
  globalMap_cloud = std::make_shared<PointCloudXYZ>();
  localMap_cloud = std::make_shared<PointCloudXYZ>();
  submaps.emplace_back(Submap(ctx_, prm));                             // PointCloudMap.h:99-101
}

PointCloudMap::~PointCloudMap() {
  if (ctx_) ndt_ctx_destroy(ctx_);
}

void PointCloudMap::addPose(const Pose2D &p) {
  if (!poses.empty()) {
    const Pose2D &pp = poses.back();
    atd += std::sqrt((p.tx - pp.tx) * (p.tx - pp.tx) + (p.ty - pp.ty) * (p.ty - pp.ty));
  } else {
    atd = 0.0;
  }
  poses.emplace_back(p);
}

void PointCloudMap::addPoints(const std::vector<LPoint2D> &lps) {
  CloudPtr cloud = std::make_shared<PointCloudXYZ>(lps.size());
  for (size_t i = 0; i < lps.size(); ++i) (*cloud)[i] = PointXYZ{(float)lps[i].x, (float)lps[i].y, 0.f, 1.f};
  Submap &cur = submaps.back();
  if (atd - cur.atdS >= prm.sepThre) {
    const size_t size = poses.size();
    cur.cntE = size - 2;
    cur.p_cloud = cur.filterPoints();
    cur.newest = false;
    Submap sub(ctx_, prm, atd, size - 1);
    const size_t ns = cur.scans.size();
    if (ns >= 2) {                                                     // two scans of overlap for the triple test
      sub.addPoints(cur.scans[ns - 2]);
      sub.addPoints(cur.scans[ns - 1]);
    }
    sub.addPoints(cloud);
    sub.makeMap();
    submaps.emplace_back(sub);
  } else {
    cur.addPoints(cloud);
    cur.makeMap();
  }
}

void PointCloudMap::makeGlobalMap() {
  globalMap_cloud->clear();
  maps.clear();
  for (size_t i = 0; i + 1 < submaps.size(); ++i) {
    globalMap_cloud->insert(globalMap_cloud->end(), submaps[i].p_cloud->begin(), submaps[i].p_cloud->end());
    maps.emplace_back(submaps[i].p_cloud);
  }
  CloudPtr f = submaps.back().filterPoints();
  globalMap_cloud->insert(globalMap_cloud->end(), f->begin(), f->end());
  maps.emplace_back(f);
}

void PointCloudMap::makeLocalMap() {
  localMap_cloud->clear();
  if (submaps.size() >= 2) {
    const Submap &s = submaps[submaps.size() - 2];
    localMap_cloud->insert(localMap_cloud->end(), s.p_cloud->begin(), s.p_cloud->end());
  }
  CloudPtr f = submaps.back().filterPoints();
  localMap_cloud->insert(localMap_cloud->end(), f->begin(), f->end());
}

}  // namespace ndt_amd
