// Reproducer 2: claim loop with break + wave-level work + LDS mask + barriers, many passes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned u32;
#define AG __HIP_MEMORY_SCOPE_AGENT
constexpr int KW = 8;
__device__ __noinline__ void work(int v, double *dst, const double *src) {
  const int lane = threadIdx.x & 63;
  double x = src[(v * 64 + lane) & 1023];
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  if (lane == 0) dst[0] = x;
}
__global__ void __launch_bounds__(512) k(u64 *ticket, u64 *out, const double *src, int passes) {
  __shared__ double wpart[KW];
  __shared__ unsigned own_mask;
  __shared__ double tot;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned epoch = 1;
  for (int p = 0; p < passes; ++p) {
    if (threadIdx.x == 0) {
      own_mask = 0u;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(ticket, (u64)(epoch + 1) << 32, __ATOMIC_RELAXED, AG);
    }
    __syncthreads();
    ++epoch;
    for (;;) {
      u64 v = 0;
      if (lane == 0) v = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, AG);
      const u32 lo = __builtin_amdgcn_readfirstlane((u32)v);
      if (lo >= (u32)KW) break;
      work((int)lo, wpart + lo, src);
      if (lane == 0) atomicOr(&own_mask, 1u << lo);
    }
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0; for (int i = 0; i < KW; ++i) s += wpart[i]; tot = s; out[p] = ((u64)own_mask << 32) | (u32)(int)s; }
    __syncthreads();
  }
}
int main() {
  u64 *t, *o; double *s; hipMalloc(&t, 512); hipMalloc(&o, 64 * 8); hipMalloc(&s, 1024 * 8);
  double hs[1024]; for (int i = 0; i < 1024; ++i) hs[i] = 1.0; hipMemcpy(s, hs, sizeof(hs), hipMemcpyHostToDevice);
  hipMemset(t, 0, 512); hipMemset(o, 0, 64 * 8);
  k<<<1, 512>>>(t, o, s, 32);
  u64 h[64]; hipError_t e = hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  printf("err %d\n", (int)e);
  for (int p = 0; p < 32; ++p) printf(" %llx/%llu", h[p] >> 32, h[p] & 0xffffffff); printf("\n");
  return 0;
}
