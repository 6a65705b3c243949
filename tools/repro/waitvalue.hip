// Does hipStreamWaitValue32 on one stream see a counter bumped by a kernel that is still running on another stream?
// (the hand-off NDT_OPT_EARLY_FITNESS relies on).  Prints the order of events; exits non-zero if the waiting kernel
// only ran after the long kernel had ended.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void long_kernel(unsigned *counter, int scope_system, unsigned long long *t_end, volatile unsigned *stop) {
  if (threadIdx.x == 0) {
    if (scope_system) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else              __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 20000000ull) __builtin_amdgcn_s_sleep(64);       // 0.2 s
  if (blockIdx.x == 0 && threadIdx.x == 0) *t_end = wall_clock64();
}
__global__ void waiter_kernel(unsigned long long *t_run) { if (threadIdx.x == 0) *t_run = wall_clock64(); }
int main() {
  for (int scope_system = 0; scope_system < 2; ++scope_system) {
    unsigned *counter; unsigned long long *t; 
    if (hipExtMallocWithFlags((void **)&counter, 8, hipMallocSignalMemory) != hipSuccess) { printf("no signal memory\n"); return 2; }
    hipMemset(counter, 0, 8);
    hipMalloc(&t, 16); hipMemset(t, 0, 16);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    long_kernel<<<64, 64, 0, a>>>(counter, scope_system, t, nullptr);
    hipError_t e = hipStreamWaitValue32(b, counter, 64, hipStreamWaitValueGte, 0xFFFFFFFFu);
    waiter_kernel<<<1, 64, 0, b>>>(t + 1);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("scope %s: waitValue rc %d; waiter ran %.3f ms %s the long kernel ended\n", scope_system ? "system" : "agent", (int)e,
           (h[1] > h[0] ? (double)(h[1] - h[0]) : (double)(h[0] - h[1])) / 1e5, h[1] < h[0] ? "BEFORE" : "AFTER");
  }
  return 0;
}
