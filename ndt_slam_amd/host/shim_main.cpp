// shim_main.cpp -- drives the C++ PoseEstimator mirror from raw files (used by tests/test_host_shim.py).
// usage: shim_main map.f32 n_map scan.f64 n_scan tx ty th_deg resolution leaf
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "PoseEstimator.h"

int main(int argc, char **argv) {
  if (argc < 10) return 2;
  const size_t nm = (size_t)atol(argv[2]), ns = (size_t)atol(argv[4]);
  std::vector<float> m(2 * nm);
  std::vector<double> s(2 * ns);
  FILE *f = fopen(argv[1], "rb"); if (!f || fread(m.data(), 4, 2 * nm, f) != 2 * nm) return 3; fclose(f);
  f = fopen(argv[3], "rb"); if (!f || fread(s.data(), 8, 2 * ns, f) != 2 * ns) return 3; fclose(f);
  ndt_amd::PointCloudXYZ target(nm);
  for (size_t i = 0; i < nm; ++i) target[i] = ndt_amd::PointXYZ{m[2 * i], m[2 * i + 1], 0.f, 1.f};
  ndt_amd::Scan2D scan;
  scan.lps.resize(ns);
  for (size_t i = 0; i < ns; ++i) { scan.lps[i].x = s[2 * i]; scan.lps[i].y = s[2 * i + 1]; }
  ndt_amd::PoseEstimator est(0, 1.0, 0.01, 0.1, atof(argv[8]), 35, atof(argv[9]));
  est.setScanPair(&scan, &target);
  ndt_amd::Pose2D init(atof(argv[5]), atof(argv[6]), atof(argv[7])), out;
  ndt_amd::Matrix3d cov;
  const double cost = est.estimatePose(init, out, cov);
  printf("%.17g %.17g %.17g %.17g", cost, out.tx, out.ty, out.th);
  for (double c : cov) printf(" %.17g", c);
  printf(" %d %d\n", est.lastResult().iters, est.lastResult().converged);
  return 0;
}
