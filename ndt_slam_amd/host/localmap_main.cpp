// localmap_main: feeds registered scans through the C++ PointCloudMap mirror the way ScanMatcher::growMap does
// (src/ScanMatcher.cpp:92-116) and dumps the local and global maps.  Used by tests/test_host_shim.py.
// usage: localmap_main scans.bin(double xy) offsets.bin(uint64) n_scans poses.bin(double x y th) sepThre
//                      removeMoving LeafSize resol thre_neighbor out.bin
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#include "PointCloudMap.h"

int main(int argc, char **argv) {
  if (argc < 11) return 2;
  const int ns = atoi(argv[3]);
  std::vector<uint64_t> off(ns + 1);
  FILE *f = fopen(argv[2], "rb"); if (!f || fread(off.data(), 8, ns + 1, f) != (size_t)ns + 1) return 3; fclose(f);
  std::vector<double> xy(2 * off[ns]), poses(3 * ns);
  f = fopen(argv[1], "rb"); if (!f || fread(xy.data(), 8, xy.size(), f) != xy.size()) return 3; fclose(f);
  f = fopen(argv[4], "rb"); if (!f || fread(poses.data(), 8, poses.size(), f) != poses.size()) return 3; fclose(f);
  ndt_amd::MapParams p;
  p.sepThre = atof(argv[5]); p.removeMoving = atoi(argv[6]) != 0; p.LeafSize = atof(argv[7]);
  p.resol = atof(argv[8]); p.thre_neighbor = atof(argv[9]);
  ndt_amd::PointCloudMap pcmap(0, p);
  if (!pcmap.ok()) return 4;
  for (int i = 0; i < ns; ++i) {
    std::vector<ndt_amd::LPoint2D> lps(off[i + 1] - off[i]);
    for (size_t k = 0; k < lps.size(); ++k) { lps[k].x = xy[2 * (off[i] + k)]; lps[k].y = xy[2 * (off[i] + k) + 1]; }
    const ndt_amd::Pose2D pose(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]);
    pcmap.addPose(pose);
    pcmap.addPoints(lps);
    pcmap.setLastPose(pose);
    pcmap.makeLocalMap();
  }
  pcmap.makeGlobalMap();
  for (const auto &s : pcmap.submaps) if (s.error()) { fprintf(stderr, "%s\n", s.error()); return 5; }
  f = fopen(argv[10], "wb"); if (!f) return 6;
  const uint64_t nl = pcmap.localMap_cloud->size(), ng = pcmap.globalMap_cloud->size(), nsub = pcmap.submaps.size();
  fwrite(&nl, 8, 1, f); fwrite(&ng, 8, 1, f); fwrite(&nsub, 8, 1, f);
  for (const auto &q : *pcmap.localMap_cloud) fwrite(&q.x, 4, 2, f);
  for (const auto &q : *pcmap.globalMap_cloud) fwrite(&q.x, 4, 2, f);
  fclose(f);
  printf("%llu %llu %llu %.17g\n", (unsigned long long)nl, (unsigned long long)ng, (unsigned long long)nsub, pcmap.atd);
  return 0;
}
