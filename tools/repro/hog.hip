// hog.hip -- a foreign kernel that holds CUs: `blocks` workgroups, each taking `lds_bytes` of LDS (150 KB: a CU to
// itself, and no room for a match workgroup beside it), spinning for `ticks` of the 100 MHz wall clock.  Used by
// tests/test_gpu_robustness.py (built there with hipcc) to run match launches beside unrelated work.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC -o hog.so tools/repro/hog.hip
#include <hip/hip_runtime.h>

__global__ void hog_kernel(unsigned long long ticks, unsigned *sink) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  unsigned acc = 0;
  for (unsigned it = 0; it < 0x40000000u; ++it) {          // counted: ends by itself
    acc += lds[(threadIdx.x + it) & 255];
    if (wall_clock64() - t0 > ticks) break;
    __builtin_amdgcn_s_sleep(8);
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

extern "C" int hog_launch(void *stream, int blocks, unsigned long long ticks, int lds_bytes, unsigned *sink) {
  hipError_t e = hipFuncSetAttribute((const void *)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  hog_kernel<<<blocks, 256, lds_bytes, (hipStream_t)stream>>>(ticks, sink);
  return (int)hipGetLastError();
}
