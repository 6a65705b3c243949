// Minimal reproducer: one lane stores a ticket word, barrier, every wave's lane 0 fetch_adds it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
#define AG __HIP_MEMORY_SCOPE_AGENT
template <int MODE>
__global__ void __launch_bounds__(512) k(u64 *ticket, u64 *out, int passes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p = 0; p < passes; ++p) {
    if (threadIdx.x == 0) {
      if (MODE == 0) __hip_atomic_store(ticket, (u64)(p + 2) << 32, __ATOMIC_RELAXED, AG);
      if (MODE == 1) { __hip_atomic_store(ticket, (u64)(p + 2) << 32, __ATOMIC_RELAXED, AG); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      if (MODE == 2) __hip_atomic_exchange(ticket, (u64)(p + 2) << 32, __ATOMIC_RELAXED, AG);
    }
    __syncthreads();
    for (int it = 0; it < 2; ++it) {
      u64 v = 0;
      if (lane == 0) v = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, AG);
      unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      if (lane == 0) out[(p * 8 + wave) * 2 + it] = ((u64)hi << 32) | lo;
    }
    __syncthreads();
  }
}
int main() {
  u64 *t, *o; hipMalloc(&t, 512); hipMalloc(&o, 8 * 2 * 4 * 8);
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(t, 0, 512); hipMemset(o, 0, 8 * 2 * 4 * 8);
    if (mode == 0) k<0><<<1, 512>>>(t, o, 4); if (mode == 1) k<1><<<1, 512>>>(t, o, 4); if (mode == 2) k<2><<<1, 512>>>(t, o, 4);
    u64 h[64]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int p = 0; p < 4; ++p) { printf("  pass %d:", p); for (int w = 0; w < 8; ++w) printf(" (%llu,%llu|%llu,%llu)", h[(p*8+w)*2] >> 32, h[(p*8+w)*2] & 0xffffffff, h[(p*8+w)*2+1] >> 32, h[(p*8+w)*2+1] & 0xffffffff); printf("\n"); }
  }
  return 0;
}
