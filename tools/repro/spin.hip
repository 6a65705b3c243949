// A kernel that only computes (fp64 FMA chains, no memory traffic) for about `iters` x 64 dependent FMAs per thread, and
// one that only streams memory: companions for tools/interference.py (what does work on the CUs the match kernel's idle
// workgroups leave cost the scans that are still running?).  hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o spin.so spin.hip
#include <hip/hip_runtime.h>
extern "C" {
__global__ void __launch_bounds__(256) spin_alu(double *out, int iters, double seed) {
  double a = seed + threadIdx.x, b = 1.0000001, c = 0.9999999, d = a * 0.5;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) { a = a * b + c; d = d * c + b; }
  }
  if (a + d == 12345.678) out[blockIdx.x] = a;        // (never true: keeps the chains alive)
}
__global__ void __launch_bounds__(256) spin_mem(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n, int reps) {
  float4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      const float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = acc;
}
int launch_spin_alu(void *stream, int blocks, int iters, double *out) {
  hipLaunchKernelGGL(spin_alu, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0);
  return (int)hipGetLastError();
}
int launch_spin_mem(void *stream, int blocks, const void *in, void *out, size_t n_float4, int reps) {
  hipLaunchKernelGGL(spin_mem, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)in, (float4 *)out, n_float4, reps);
  return (int)hipGetLastError();
}
}
