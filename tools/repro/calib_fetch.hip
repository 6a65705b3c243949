// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths of the NDT kernels (MI355X_MICROARCH.md:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel reads the same 64 MiB buffer once; FETCH_SIZE (KB) / 65536 is the factor for that pattern.
//   hipcc --offload-arch=gfx950 -O3 -o calib_fetch tools/repro/calib_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o run -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read_dword(const float *p, size_t n, float *out) {          // 4 B per lane, coalesced
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678f) *out = s;
}
__global__ void read_dwordx2(const float2 *p, size_t n, float *out) {       // 8 B per lane, coalesced (scan points)
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float2 v = p[i]; s += v.x + v.y; }
  if (s == 12345.678f) *out = s;
}
__global__ void read_dwordx4(const float4 *p, size_t n, float *out) {       // 16 B per lane, coalesced (the guide's case)
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.w; }
  if (s == 12345.678f) *out = s;
}
__global__ void read_records(const float4 *p, size_t nrec, float *out) {    // 48 of every 64 bytes: the voxel records
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = p[4 * i], b = p[4 * i + 1], c = p[4 * i + 2];
    s += a.x + b.y + c.z;
  }
  if (s == 12345.678f) *out = s;
}
__global__ void read_gather8(const float2 *p, size_t n, float *out) {       // 8 B per lane, every lane its own 128-B line
  float s = 0.f;
  const size_t stride = 16;                                                  // 16 float2 = 128 B
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n / stride; i += (size_t)gridDim.x * blockDim.x) { float2 v = p[i * stride]; s += v.x; }
  if (s == 12345.678f) *out = s;
}

int main() {
  const size_t bytes = 64ull << 20;
  void *buf; float *out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 4);
  hipMemset(buf, 0, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    read_dword<<<2048, 256>>>((const float *)buf, bytes / 4, out);
    read_dwordx2<<<2048, 256>>>((const float2 *)buf, bytes / 8, out);
    read_dwordx4<<<2048, 256>>>((const float4 *)buf, bytes / 16, out);
    read_records<<<2048, 256>>>((const float4 *)buf, bytes / 64, out);
    read_gather8<<<2048, 256>>>((const float2 *)buf, bytes / 8, out);
  }
  hipDeviceSynchronize();
  printf("read %zu bytes per kernel (records: 3/4 of it; gather8: 1/16 of it requested, every 128-B line touched)\n", bytes);
  return 0;
}
