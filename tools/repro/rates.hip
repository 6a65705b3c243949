// rates.hip -- issue rate (cycles per wave-instruction per SIMD) of the instructions of the NDT pass loop on gfx950,
// with 4 waves per SIMD (the match kernel's occupancy) and with 1 wave per SIMD (a helper's lone wave), plus the
// dependent-issue latency of the fp64 ops and of LDS reads / cross-lane moves.
//   hipcc -O3 --offload-arch=gfx950 -o rates tools/repro/rates.hip && ./rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
typedef float d4v __attribute__((ext_vector_type(4)));   // 128-bit register tuple

#define BODY8(INSTR)                                                                                         \
  asm volatile(INSTR(0) "\n" INSTR(1) "\n" INSTR(2) "\n" INSTR(3) "\n" INSTR(4) "\n" INSTR(5) "\n" INSTR(6) "\n" INSTR(7) \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
               : "v"(b), "v"(c), "v"(ib), "v"(fb))

// independent: eight different destination registers; dependent: one chain
#define K_IND(NAME, INSTR)                                                                                   \
  __global__ void __launch_bounds__(1024) NAME(double *out, long long *cyc, double seed) {                   \
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    double b = seed * 0.5 + threadIdx.x * 1e-9, c = seed * 0.25;                                              \
    int ib = 3; float fb = 1.5f;                                                                             \
    __syncthreads();                                                                                         \
    long long t0 = clock64();                                                                                \
    _Pragma("unroll") for (int r = 0; r < REP; ++r) { BODY8(INSTR); }                                         \
    long long t1 = clock64();                                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                       \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;          \
  }

#define I_FMA(i)   "v_fma_f64 %" #i ", %" #i ", %8, %9"
#define I_MUL(i)   "v_mul_f64 %" #i ", %" #i ", %8"
#define I_ADD(i)   "v_add_f64 %" #i ", %" #i ", %8"
#define I_MAX(i)   "v_max_f64 %" #i ", %" #i ", %8"
#define I_RND(i)   "v_rndne_f64 %" #i ", %" #i
#define I_LDEXP(i) "v_ldexp_f64 %" #i ", %" #i ", %10"
#define I_CMP(i)   "v_cmp_lt_f64 vcc, %" #i ", %8"
#define I_MOV64(i) "v_mov_b64 %" #i ", %8"
K_IND(k_fma, I_FMA)
K_IND(k_mul, I_MUL)
K_IND(k_add, I_ADD)
K_IND(k_max, I_MAX)
K_IND(k_rnd, I_RND)
K_IND(k_ldexp, I_LDEXP)
K_IND(k_cmp, I_CMP)
K_IND(k_mov64, I_MOV64)

// 32-bit destination variants
#define BODY8I(INSTR)                                                                                        \
  asm volatile(INSTR(0) "\n" INSTR(1) "\n" INSTR(2) "\n" INSTR(3) "\n" INSTR(4) "\n" INSTR(5) "\n" INSTR(6) "\n" INSTR(7) \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
               : "v"(b), "v"(d0), "v"(d1))
#define K_INT(NAME, INSTR)                                                                                   \
  __global__ void __launch_bounds__(1024) NAME(double *out, long long *cyc, double seed) {                   \
    int a0 = (int)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    int b = 7 + (threadIdx.x & 3); double d0 = seed, d1 = seed * 3;                                           \
    __syncthreads();                                                                                         \
    long long t0 = clock64();                                                                                \
    _Pragma("unroll") for (int r = 0; r < REP; ++r) { BODY8I(INSTR); }                                        \
    long long t1 = clock64();                                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);             \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;          \
  }
#define I_MULLO(i)  "v_mul_lo_u32 %" #i ", %" #i ", %8"
#define I_MUL24(i)  "v_mul_u32_u24 %" #i ", %" #i ", %8"
#define I_MAD24(i)  "v_mad_u32_u24 %" #i ", %" #i ", %8, %8"
#define I_ADD32(i)  "v_add_u32 %" #i ", %" #i ", %8"
#define I_CVTI(i)   "v_cvt_i32_f64 %" #i ", %9"
#define I_CVTF(i)   "v_cvt_f32_f64 %" #i ", %9"
#define I_FMA32(i)  "v_fma_f32 %" #i ", %" #i ", %8, %8"
#define I_CNDM(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define I_DPP(i)    "v_mov_b32_dpp %" #i ", %8 row_ror:8 row_mask:0xf bank_mask:0xf"
#define I_BPERM(i)  "ds_bpermute_b32 %" #i ", %8, %" #i "\ns_waitcnt lgkmcnt(0)"
#define I_SWAP32(i) "v_permlane32_swap_b32 %" #i ", %8"
K_INT(k_mullo, I_MULLO)
K_INT(k_mul24, I_MUL24)
K_INT(k_mad24, I_MAD24)
K_INT(k_add32, I_ADD32)
K_INT(k_cvti, I_CVTI)
K_INT(k_cvtf, I_CVTF)
K_INT(k_fma32, I_FMA32)
K_INT(k_cndm, I_CNDM)
K_INT(k_dpp, I_DPP)
K_INT(k_bperm, I_BPERM)
K_INT(k_swap32, I_SWAP32)
#define I_CNDM_E64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]"
#define I_CNDM_LIT(i) "v_cndmask_b32_e64 %" #i ", 0, 16, vcc"
#define I_CMPCND(i)   "v_cmp_gt_u32 vcc, %" #i ", %8\ns_nop 1\nv_cndmask_b32_e64 %" #i ", 0, 16, vcc"
#define I_CMP32(i)    "v_cmp_gt_u32 vcc, %" #i ", %8"
#define I_CMP32S(i)   "v_cmp_gt_u32 s[10:11], %" #i ", %8"
#define I_PKMUL(i)    "v_pk_mul_f32 %" #i ", %9, %9"
#define I_MIN3(i)     "v_min3_f32 %" #i ", %" #i ", %8, %8"
#define I_OR3(i)      "v_or3_b32 %" #i ", %" #i ", %8, %8"
#define I_LSHLADD(i)  "v_lshl_add_u32 %" #i ", %" #i ", 1, %8"
#define I_FFBL(i)     "v_ffbl_b32 %" #i ", %" #i
#define I_AND(i)      "v_and_b32 %" #i ", %" #i ", %8"
#define I_DSU16(i)    "ds_read_u16 %" #i ", %8"
#define I_DSB64(i)    "ds_read_b64 %9, %8"
#define I_DS2B64(i)   "ds_read2_b64 %10, %8 offset0:1 offset1:2"
#define I_NOP(i)      "s_nop 0"
K_INT(k_cndm_e64, I_CNDM_E64)
K_INT(k_cndm_lit, I_CNDM_LIT)
K_INT(k_cmpcnd, I_CMPCND)
K_INT(k_cmp32, I_CMP32)
K_INT(k_cmp32s, I_CMP32S)
K_INT(k_min3, I_MIN3)
K_INT(k_or3, I_OR3)
K_INT(k_lshladd, I_LSHLADD)
K_INT(k_ffbl, I_FFBL)
K_INT(k_and, I_AND)
K_INT(k_snop, I_NOP)

// LDS read throughput: independent reads of one address pattern (48-byte records, lanes spread over ~16 records)
#define K_LDS(NAME, INSTR, WAIT)                                                                              \
  __global__ void __launch_bounds__(1024) NAME(double *out, long long *cyc, double seed) {                   \
    __shared__ double tab[6 * 1024];                                                                         \
    for (int i = threadIdx.x; i < 6 * 1024; i += blockDim.x) tab[i] = i;                                      \
    __syncthreads();                                                                                         \
    const unsigned addr = (unsigned)(size_t)(((threadIdx.x * 7) >> 2) % 800) * 48u;                           \
    double r0 = 0; d4v r1 = {0, 0, 0, 0}; int u = 0;                                                      \
    long long t0 = clock64();                                                                                \
    _Pragma("unroll") for (int r = 0; r < 8 * REP; ++r) asm volatile(INSTR : "+v"(r0), "+v"(r1), "+v"(u) : "v"(addr)); \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
    long long t1 = clock64();                                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + (double)r1.x + u + tab[threadIdx.x];                            \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;          \
  }
K_LDS(l_u16, "ds_read_u16 %2, %3", 0)
K_LDS(l_b64, "ds_read_b64 %0, %3", 0)
K_LDS(l_2b64, "ds_read2_b64 %1, %3 offset0:1 offset1:2", 0)

// the counter clock64() reads against the 100 MHz wall clock
__global__ void k_calib(long long *o) {
  const long long w0 = wall_clock64(), c0 = clock64();
  while (wall_clock64() - w0 < 100000) {}    // 1 ms
  o[0] = clock64() - c0; o[1] = wall_clock64() - w0;
}

// dependent chains: latency of one op when a wave has nothing else to issue
#define K_DEP(NAME, INSTR)                                                                                   \
  __global__ void __launch_bounds__(1024) NAME(double *out, long long *cyc, double seed) {                   \
    double a0 = seed, b = 1.0000001, c = 1e-9; int ib = 0;                                                    \
    __syncthreads();                                                                                         \
    long long t0 = clock64();                                                                                \
    _Pragma("unroll") for (int r = 0; r < 8 * REP; ++r) asm volatile(INSTR : "+v"(a0) : "v"(b), "v"(c), "v"(ib)); \
    long long t1 = clock64();                                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;                                                         \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;          \
  }
K_DEP(d_fma, "v_fma_f64 %0, %0, %1, %2")
K_DEP(d_mul, "v_mul_f64 %0, %0, %1")
K_DEP(d_add, "v_add_f64 %0, %0, %2")
K_DEP(d_ldexp, "v_ldexp_f64 %0, %0, %3")
K_DEP(d_rnd, "v_rndne_f64 %0, %0")

// LDS read latency: pointer chase
__global__ void __launch_bounds__(1024) d_lds(double *out, long long *cyc, double seed) {
  __shared__ int nxt[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) nxt[i] = (i * 17 + 64) & 4095;
  __syncthreads();
  int p = threadIdx.x;
  long long t0 = clock64();
#pragma unroll
  for (int r = 0; r < 8 * REP; ++r) p = nxt[p];
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = p + seed;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(double *, long long *, double);
struct Entry { const char *name; kern_t k; int n; };

int main() {
  double *out; long long *cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(double));
  hipMalloc(&cyc, 256 * 16 * sizeof(long long));
  Entry es[] = {
    {"v_fma_f64", k_fma, 8 * REP}, {"v_mul_f64", k_mul, 8 * REP}, {"v_add_f64", k_add, 8 * REP}, {"v_max_f64", k_max, 8 * REP},
    {"v_rndne_f64", k_rnd, 8 * REP}, {"v_ldexp_f64", k_ldexp, 8 * REP}, {"v_cmp_lt_f64", k_cmp, 8 * REP}, {"v_mov_b64", k_mov64, 8 * REP},
    {"v_mul_lo_u32", k_mullo, 8 * REP}, {"v_mul_u32_u24", k_mul24, 8 * REP}, {"v_mad_u32_u24", k_mad24, 8 * REP}, {"v_add_u32", k_add32, 8 * REP},
    {"v_cvt_i32_f64", k_cvti, 8 * REP}, {"v_cvt_f32_f64", k_cvtf, 8 * REP}, {"v_fma_f32", k_fma32, 8 * REP}, {"v_cndmask_b32", k_cndm, 8 * REP},
    {"v_mov_b32_dpp", k_dpp, 8 * REP}, {"ds_bpermute+wait", k_bperm, 8 * REP}, {"v_permlane32_swap", k_swap32, 8 * REP},
    {"cndmask e64 sgpr", k_cndm_e64, 8 * REP}, {"cndmask 0,16,vcc", k_cndm_lit, 8 * REP}, {"cmp+nop1+cndmask", k_cmpcnd, 8 * REP},
    {"v_cmp_gt_u32 vcc", k_cmp32, 8 * REP}, {"v_cmp_gt_u32 sgpr", k_cmp32s, 8 * REP}, {"v_min3_f32", k_min3, 8 * REP}, {"v_or3_b32", k_or3, 8 * REP},
    {"v_lshl_add_u32", k_lshladd, 8 * REP}, {"v_ffbl_b32", k_ffbl, 8 * REP}, {"v_and_b32", k_and, 8 * REP}, {"s_nop 0", k_snop, 8 * REP},
    {"ds_read_u16 (tput)", l_u16, 8 * REP}, {"ds_read_b64 (tput)", l_b64, 8 * REP}, {"ds_read2_b64 (tput)", l_2b64, 8 * REP},
    {"dep v_fma_f64", d_fma, 8 * REP}, {"dep v_mul_f64", d_mul, 8 * REP}, {"dep v_add_f64", d_add, 8 * REP}, {"dep v_ldexp_f64", d_ldexp, 8 * REP},
    {"dep v_rndne_f64", d_rnd, 8 * REP}, {"dep lds read", d_lds, 8 * REP},
  };
  const int threads[] = {1024, 256};   // 4 waves per SIMD, 1 wave per SIMD
  printf("%-20s %14s %14s   (shader-clock cycles per instruction, as one wave sees them)\n", "instruction", "4 waves/SIMD", "1 wave/SIMD");
  for (const Entry &e : es) {
    double res[2];
    for (int v = 0; v < 2; ++v) {
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads[v]), 0, 0, out, cyc, 1.0);
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads[v]), 0, 0, out, cyc, 1.0);
      hipDeviceSynchronize();
      std::vector<long long> h(256 * 16);
      hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      const int nw = 256 * (threads[v] / 64);
      double s = 0; for (int i = 0; i < nw; ++i) s += (double)h[i];
      res[v] = s / nw / e.n;
    }
    printf("%-20s %14.2f %14.2f\n", e.name, res[0], res[1]);
  }
  {
    long long *o; hipMalloc(&o, 16); long long h[2];
    hipLaunchKernelGGL(k_calib, dim3(1), dim3(1), 0, 0, o); hipDeviceSynchronize();
    hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("clock64: %lld counts in %lld ticks of the 100 MHz wall clock = %.1f MHz\n", h[0], h[1], 100.0 * h[0] / h[1]);
  }
  printf("(clock64 counts at a fixed 100 MHz on this part if the numbers look 20x too small: compare with v_add_u32 = 4 cycles per wave at 4 waves/SIMD)\n");
  return 0;
}
