// ndt_map_build.hip.h -- row a2: voxel normal-distributions build.
// Part of libndt_mi355x.so: included by ndt_mi355x.hip inside its anonymous namespace (one translation
// unit; the order of the includes matters).  Not a standalone header.

// ------------------------------------------------------------------------------------------
// a2: voxel normal-distributions build
// ------------------------------------------------------------------------------------------

// order-preserving float <-> uint for atomicMin/atomicMax
__device__ __forceinline__ unsigned f2ord(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __uint_as_float(u);
#else
  memcpy(&f, &u, 4);
#endif
  return f;
}

// getMinMax3D: out[0..3] = ord(min x), ord(min y), ord(max x), ord(max y).  bounds[0..3] is the running result and
// bounds[8] counts the workgroups that are done; the last one hands the result over and puts both back to their
// start values ({~0, ~0, 0, 0}, 0 -- set once when the map is created): no initialising copy in front of every build.
__global__ void __launch_bounds__(256)
map_minmax_kernel(const float *__restrict__ xy, size_t stride, size_t n, unsigned *__restrict__ bounds,
                  unsigned *__restrict__ out) {
  __shared__ float sh[4][4];
  float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i0 < n; i0 += 16 * step) {   // 16 loads in flight
    float2 p[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const size_t i = i0 + u * step; p[u] = load_pt(xy, stride, i < n ? i : i0); }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (!finite2(p[u].x, p[u].y)) continue;
      mnx = fminf(mnx, p[u].x); mxx = fmaxf(mxx, p[u].x);
      mny = fminf(mny, p[u].y); mxy = fmaxf(mxy, p[u].y);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mnx = fminf(mnx, __shfl_down(mnx, o)); mny = fminf(mny, __shfl_down(mny, o));
    mxx = fmaxf(mxx, __shfl_down(mxx, o)); mxy = fmaxf(mxy, __shfl_down(mxy, o));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[w][0] = mnx; sh[w][1] = mny; sh[w][2] = mxx; sh[w][3] = mxy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      mnx = fminf(mnx, sh[k][0]); mny = fminf(mny, sh[k][1]);
      mxx = fmaxf(mxx, sh[k][2]); mxy = fmaxf(mxy, sh[k][3]);
    }
    if (mnx <= mxx) {     // one atomic set per workgroup
      atomicMin(&bounds[0], f2ord(mnx)); atomicMin(&bounds[1], f2ord(mny));
      atomicMax(&bounds[2], f2ord(mxx)); atomicMax(&bounds[3], f2ord(mxy));
    }
    __threadfence();
    if (atomicAdd(&bounds[8], 1u) == gridDim.x - 1u) {      // the last workgroup: every other one's atomics are in
      __threadfence();
      out[0] = atomicExch(&bounds[0], 0xffffffffu); out[1] = atomicExch(&bounds[1], 0xffffffffu);
      out[2] = atomicExch(&bounds[2], 0u);          out[3] = atomicExch(&bounds[3], 0u);
      atomicExch(&bounds[8], 0u);
    }
  }
}

struct GridDims { float inv_leaf; int min_bx, min_by, div_x, div_y, gw, gh; };

__device__ __forceinline__ int voxel_of(const GridDims &G, float2 p) {
  if (!finite2(p.x, p.y)) return -1;
  const float fx = fminf(fmaxf(floorf(p.x * G.inv_leaf), -1.0e9f), 1.0e9f), fy = fminf(fmaxf(floorf(p.y * G.inv_leaf), -1.0e9f), 1.0e9f);
  const int ix = (int)fx - G.min_bx, iy = (int)fy - G.min_by;
  // never true for the grid of this cloud's own bounding box; a build queued ahead of the bounding
  // box read-back with the previous grid (ndt_map_build_dev) must stay inside its buffers
  if (ix < 0 || ix >= G.div_x || iy < 0 || iy >= G.div_y) return -1;
  return iy * G.div_x + ix;
}

// Consecutive cloud points usually fall in the same voxel (a map is appended scan by scan, wall by
// wall), so a wave first merges runs of equal voxel keys among its 64 consecutive points and issues
// one atomic per run instead of one per point.
__device__ __forceinline__ void wave_runs(int v, int lane, int &head, int &len) {
  const int prev = __shfl_up(v, 1);
  const bool is_head = (lane == 0) || (v != prev);
  const unsigned long long heads = __ballot(is_head);
  const unsigned long long below = heads & ((2ull << lane) - 1ull);      // heads at or below this lane
  head = 63 - __builtin_clzll(below);
  const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
  len = above ? (lane + 1 + __builtin_ctzll(above)) - lane : 64 - lane;  // valid in head lanes
}

__global__ void __launch_bounds__(256)
map_count_kernel(const float *__restrict__ xy, size_t stride, size_t n, GridDims G, int *__restrict__ count,
                 int *__restrict__ counters) {
  if (blockIdx.x == 0 && threadIdx.x < 4) counters[threadIdx.x] = 0;     // the build's small counters (cells, valid, big voxels, scan ticket), for the kernels behind this one
  const int lane = threadIdx.x & 63;
  const size_t nround = (n + 63) / 64 * 64;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nround; i += (size_t)gridDim.x * blockDim.x) {
    const int v = i < n ? voxel_of(G, load_pt(xy, stride, i)) : -2;
    int head, len;
    wave_runs(v, lane, head, len);
    if (head == lane && v >= 0) atomicAdd(&count[v], len);
  }
}

// Exclusive scan of count[0..ng) into start[0..ng] in ONE kernel (round 4; before: tile sums | offsets of the tile sums |
// apply -- three launches for 14 us of work and two launch latencies).  Single pass with decoupled look-back: a workgroup
// takes its tile number from a ticket (so every tile in front of it has been taken by a workgroup that is running or done:
// no workgroup waits for one that has not started), adds up its 2048 counters, publishes the sum as an AGGREGATE, looks back
// over its predecessors' words -- 64 at a time, one per lane of wave 0 -- adding aggregates until it meets an inclusive
// PREFIX, then publishes its own prefix.  The words validate themselves: value | tag of this build << 32 | kind << 62, written
// and read as one 64-bit word with agent-scope atomics (cdna_hip_programming.md Guideline 16, R2); words of earlier builds
// carry other tags and are waited out, so nothing is cleared between builds.  A predecessor publishes its aggregate right
// after its own loads, before it waits for anything: every wait ends.
// Two sums travel together -- the points in front of a voxel (its offset) and the voxels with more than kBigVoxel points
// in front of it (its place in the list of big voxels; an atomic per wave on one counter used to take most of this
// kernel's time: same-address atomics queue up at ~50 ns each) -- as two words per tile with a look-back each.
constexpr int kScanBlock = 1024, kScanPer = 8, kScanTile = kScanBlock * kScanPer;
constexpr int kBigVoxel = 16;        // voxels with more points are handled by a whole wave (order, statistics)
constexpr unsigned long long kScanAgg = 1ull << 62, kScanPre = 2ull << 62;

__global__ void __launch_bounds__(kScanBlock)
scan_onepass_kernel(const int *__restrict__ in, size_t n, unsigned long long *__restrict__ state /* 2 words per tile */, unsigned tag,
                    int ntiles, int *__restrict__ ticket, int *__restrict__ out /* n + 1 (+ 3 copies; 4 readable ints in front) */,
                    int *__restrict__ big /* voxels with more than kBigVoxel points, in voxel order */, int *__restrict__ nbig, int big_cap) {
  __shared__ int sh_s[kScanBlock / 64], sh_b[kScanBlock / 64];
  __shared__ int s_tile, s_base_s, s_base_b;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1);
  __syncthreads();
  const int tile = s_tile, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = (size_t)tile * kScanTile + (size_t)threadIdx.x * kScanPer;
  int v[kScanPer]; int s = 0; unsigned bigmask = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
    if (v[k] > kBigVoxel) bigmask |= 1u << k;
  }
  const int nb = __builtin_popcount(bigmask);
  const int incl_s = (int)wave_incl_scan((unsigned)s), incl_b = (int)wave_incl_scan((unsigned)nb);
  if (lane == 63) { sh_s[wave] = incl_s; sh_b[wave] = incl_b; }
  __syncthreads();
  int before_s = 0, total_s = 0, before_b = 0, total_b = 0;
#pragma unroll
  for (int w = 0; w < kScanBlock / 64; ++w) {
    before_s += (w < wave) ? sh_s[w] : 0; total_s += sh_s[w];
    before_b += (w < wave) ? sh_b[w] : 0; total_b += sh_b[w];
  }
  const unsigned long long tagged = (unsigned long long)(tag & 0x3fffffffu) << 32;
  auto publish = [&](int which, unsigned long long kind, int value) {
    __hip_atomic_store(&state[2 * (size_t)tile + which], kind | tagged | (unsigned)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto valid = [&](unsigned long long w) { return ((w ^ tagged) & (0x3fffffffull << 32)) == 0ull && (w >> 62) != 0ull; };
  if (wave == 0) {
    if (lane == 0) { publish(0, tile == 0 ? kScanPre : kScanAgg, total_s); publish(1, tile == 0 ? kScanPre : kScanAgg, total_b); }
    int excl_s = 0, excl_b = 0;
    bool open_s = true, open_b = true;                                     // (wave-uniform) still looking for a prefix
    for (int first = tile - 1; first >= 0 && (open_s || open_b); first -= 64) {      // (tile 0: no turn)
      const int j = first - lane;
      unsigned long long ws = kScanPre | tagged, wb = kScanPre | tagged;   // in front of tile 0: inclusive prefixes of zero
      if (j >= 0) {
        do {
          ws = __hip_atomic_load(&state[2 * (size_t)j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          wb = __hip_atomic_load(&state[2 * (size_t)j + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (!valid(ws) || !valid(wb));
      }
      if (open_s) {
        const unsigned long long pre = __ballot((ws >> 62) == 2ull);      // lanes (nearest tile first) that hold a prefix
        const int stop = pre ? __builtin_ctzll(pre) : 63;                 // add lanes 0 .. stop
        int part = lane <= stop ? (int)(unsigned)ws : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        excl_s += part;
        if (pre) open_s = false;
      }
      if (open_b) {
        const unsigned long long pre = __ballot((wb >> 62) == 2ull);
        const int stop = pre ? __builtin_ctzll(pre) : 63;
        int part = lane <= stop ? (int)(unsigned)wb : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        excl_b += part;
        if (pre) open_b = false;
      }
    }
    if (lane == 0) {
      s_base_s = excl_s; s_base_b = excl_b;
      if (tile > 0) { publish(0, kScanPre, excl_s + total_s); publish(1, kScanPre, excl_b + total_b); }
    }
  }
  __syncthreads();
  int run = s_base_s + before_s + incl_s - s;
  int q0 = s_base_b + before_b + incl_b - nb;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
    if ((bigmask >> k) & 1u) { if (q0 < big_cap) big[q0] = (int)(base + k); ++q0; }
  }
  if (tile == ntiles - 1 && threadIdx.x < 4) out[n + threadIdx.x] = s_base_s + total_s;   // out[n], + 3 readable copies
  if (tile == ntiles - 1 && threadIdx.x == 4) *nbig = s_base_b + total_b;
  if (tile == 0 && threadIdx.x < 4) out[(int)threadIdx.x - 4] = 0;                        // the four readable ints in front of out[0] (ndt_fitness.hip.h)
}

__global__ void __launch_bounds__(256)
map_scatter_kernel(const float *__restrict__ xy, size_t stride, size_t n, GridDims G,
                   const int *__restrict__ start, int *__restrict__ count /* in: points per voxel; out: zero */,
                   int *__restrict__ perm) {
  const int lane = threadIdx.x & 63;
  const size_t nround = (n + 63) / 64 * 64;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nround; i += (size_t)gridDim.x * blockDim.x) {
    const int v = i < n ? voxel_of(G, load_pt(xy, stride, i)) : -2;
    int head, len;
    wave_runs(v, lane, head, len);
    int base = 0;
    if (head == lane && v >= 0) base = start[v] + atomicSub(&count[v], len) - len;   // one slot range per run, from the bucket's end
    base = __shfl(base, head);
    if (v >= 0) perm[base + (lane - head)] = (int)i;                          // cloud order kept inside a run
  }
}

// Restore input order inside every bucket (PCL accumulates a voxel's points in cloud order and
// its float32 centroid sum depends on that order), rank by counting.  Round 3: the POINT goes to its ranked place (the
// bucketed copy `pts` the fitness search reads anyway), not its number: the gather by point number -- a million random
// 8-byte reads -- happens here, spread over every lane of the chip, and map_finalize_kernel streams a voxel's points from
// consecutive addresses instead of chasing number -> point through two dependent loads per batch (47 -> .. us).  Voxels of up to kBigVoxel
// points: eight lanes per voxel (one wave per voxel spent its time launching waves, 70 % of the
// voxels being empty); the others, listed by scan_onepass_kernel (the list is in voxel order): one wave per voxel.
constexpr int kOrderVoxPerBlock = 256 / 8 * 4;     // 32 lane groups, 4 voxels each
__device__ __forceinline__ void order_small_voxels(unsigned block, const int *__restrict__ start, size_t ng,
                                                   const int *__restrict__ perm, const float *__restrict__ xy, size_t stride,
                                                   float2 *__restrict__ pts) {
  // A group of eight lanes takes four voxels.  The kernel is a chain of dependent loads (offsets -> numbers -> point) on mostly
  // EMPTY voxels, so the four voxels are walked side by side: their offsets in one round (a lane each), then their numbers, then
  // their points, instead of four chains one after the other.
  const int grp = threadIdx.x >> 3, sub = threadIdx.x & 7, lane = threadIdx.x & 63;
  const size_t g0 = (size_t)block * kOrderVoxPerBlock + (size_t)grp * 4;
  int my_s = 0, my_n = 0;
  if (sub < 4 && g0 + sub < ng) { my_s = start[g0 + sub]; my_n = start[g0 + sub + 1] - my_s; }
  int s0[4], n[4], nmax = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s0[r] = __shfl(my_s, (lane & ~7) + r); n[r] = __shfl(my_n, (lane & ~7) + r);
    if (n[r] > kBigVoxel) n[r] = 0;                          // (listed for the wave-per-voxel workgroups)
    nmax = max(nmax, n[r]);
  }
  for (int e = sub; e < nmax; e += 8) {
    int mine[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[r] = e < n[r] ? perm[s0[r] + e] : 0;
    float2 p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = e < n[r] ? load_pt(xy, stride, (size_t)mine[r]) : make_float2(0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (e >= n[r]) continue;
      int rank = 0;
      // four numbers per load; what is read past the end of the bucket (perm has four spare entries) does not count
      for (int j = 0; j < n[r]; j += 4) {
        const ndt_i4v q = *(const NDT_GLOBAL ndt_i4v_u *)(perm + s0[r] + j);
        rank += (q.x < mine[r]) ? 1 : 0;
        rank += (j + 1 < n[r] && q.y < mine[r]) ? 1 : 0;
        rank += (j + 2 < n[r] && q.z < mine[r]) ? 1 : 0;
        rank += (j + 3 < n[r] && q.w < mine[r]) ? 1 : 0;
      }
      pts[s0[r] + rank] = p[r];                              // the point itself goes to its place: map_finalize_kernel streams them
    }
  }
}

constexpr int kBigWavesPerBlock = 4, kBigBlocks = 4096, kBigStage = 512;   // LDS staging: point numbers per wave
constexpr int kBigRuns = 64;                 // runs of consecutive point numbers per voxel the merge handles (else: rank by counting)
// One launch for both kinds (they do not depend on each other): workgroups [0, small_blocks) take the small voxels,
// the kBigBlocks behind them the listed big ones.
//
// Big voxels (round 4): a voxel's segment of `perm` is a handful of RUNS of consecutive point numbers -- map_scatter_kernel
// places every wave-run as a block, and a cloud is appended scan by scan, wall by wall (bench map: 2.5 runs per voxel,
// 99 % of the voxels <= 8) -- in the order their atomics arrived.  Cloud order = the runs sorted by their first number, so
// an element's place is (lengths of the runs that start before its run) + (its offset in its run): O(n + runs * n / 64)
// per voxel instead of the n^2 / 64 comparisons of rank counting (n = 25: the typical wall voxel).  A voxel in more than
// kBigRuns runs (a cloud in random order) keeps rank counting.
__global__ void __launch_bounds__(256)
map_order_kernel(const int *__restrict__ start, size_t ng, unsigned small_blocks, const int *__restrict__ big,
                 const int *__restrict__ nbig, int big_cap, const int *__restrict__ perm, const float *__restrict__ xy,
                 size_t stride, float2 *__restrict__ pts) {
  __shared__ int stage[kBigWavesPerBlock][kBigStage];
  __shared__ int run_first[kBigWavesPerBlock][kBigRuns], run_pos[kBigWavesPerBlock][kBigRuns + 1];
  if (blockIdx.x < small_blocks) { order_small_voxels(blockIdx.x, start, ng, perm, xy, stride, pts); return; }
  const unsigned bblock = blockIdx.x - small_blocks, bblocks = gridDim.x - small_blocks;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u64 lt = (1ull << lane) - 1ull;
  const int count = min(*nbig, big_cap);
  for (int q = bblock * kBigWavesPerBlock + wv; q < count; q += bblocks * kBigWavesPerBlock) {
    const int g = big[q];
    const int s0 = start[g], n = start[g + 1] - s0;
    const bool staged = n <= kBigStage;
    int nruns = 0;
    if (staged) {
      for (int e = lane; e < n; e += 64) stage[wv][e] = perm[s0 + e];
      __builtin_amdgcn_wave_barrier();            // one wave: its LDS writes are ordered before its later reads
      // the runs, in the order they lie in the segment: (first number, position)
      for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        const int idx = e < n ? stage[wv][e] : 0, prev = (e > 0 && e < n) ? stage[wv][e - 1] : -2;
        const bool head = e < n && (e == 0 || idx != prev + 1);
        const u64 hm = __ballot(head);
        if (head) {
          const int r = nruns + __builtin_popcountll(hm & lt);
          if (r < kBigRuns) { run_first[wv][r] = idx; run_pos[wv][r] = e; }
        }
        nruns += __builtin_popcountll(hm);
      }
      if (lane == 0 && nruns <= kBigRuns) run_pos[wv][nruns] = n;
      __builtin_amdgcn_wave_barrier();
    }
    if (staged && nruns <= kBigRuns) {
      int seen = 0;                               // runs in front of this chunk
      for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        const int idx = e < n ? stage[wv][e] : 0, prev = (e > 0 && e < n) ? stage[wv][e - 1] : -2;
        const bool head = e < n && (e == 0 || idx != prev + 1);
        const u64 hm = __ballot(head);
        const int myrun = seen + __builtin_popcountll(hm & (lt | (1ull << lane))) - 1;     // (e < n: some head at or below)
        seen += __builtin_popcountll(hm);
        if (e >= n) continue;
        const float2 p = load_pt(xy, stride, (size_t)idx);
        const int myfirst = run_first[wv][myrun], off = e - run_pos[wv][myrun];
        int dest = 0;
        for (int r = 0; r < nruns; ++r) {           // (uniform reads: LDS broadcasts)
          const int f = run_first[wv][r], len = run_pos[wv][r + 1] - run_pos[wv][r];
          dest += (f < myfirst) ? len : 0;
        }
        pts[s0 + dest + off] = p;
      }
      continue;
    }
    for (int e = lane; e < n; e += 64) {          // many runs / a segment beyond the staging area: rank by counting
      const int mine = staged ? stage[wv][e] : perm[s0 + e];
      int rank = 0;
      if (staged) { for (int j = 0; j < n; ++j) rank += (stage[wv][j] < mine) ? 1 : 0; }
      else        { for (int j = 0; j < n; ++j) rank += (perm[s0 + j] < mine) ? 1 : 0; }
      pts[s0 + rank] = load_pt(xy, stride, (size_t)mine);
    }
  }
}

struct LeafParams { int min_pts, cov_unbiased, cov_init_identity; double eig_mult; };

// Mean, regularised covariance and its inverse of one z = 0 voxel (second loop of
// VoxelGridCovariance::applyFilter); closed-form 2x2 eigen-decomposition, the z eigenpair is
// exactly (czz, e_z).  Returns 1 accepted, 0 rejected (icov = 0), -1 rejected with inf icov.
__device__ int leaf_finalize(const LeafParams &L, int n, double sx, double sy, double sxx,
                             double sxy, double syy, double szz, double mean[2], double icov[3]) {
  const double dn = (double)n;
  const double mx = sx / dn, my = sy / dn;
  mean[0] = mx; mean[1] = my;
  icov[0] = icov[1] = icov[2] = 0.0;
  double cxx, cxy, cyy, czz;
  if (!L.cov_unbiased) {
    cxx = (sxx - 2.0 * (sx * mx)) / dn + mx * mx;
    cxy = (sxy - 2.0 * (sy * mx)) / dn + my * mx;     // the LOWER-triangle entry (1,0): what Eigen's solver reads
    cyy = (syy - 2.0 * (sy * my)) / dn + my * my;
    czz = szz / dn;
    const double f = (dn - 1.0) / dn;
    cxx *= f; cxy *= f; cyy *= f; czz *= f;
  } else {
    cxx = (sxx - sx * mx) / (dn - 1.0);
    cxy = (sxy - sy * mx) / (dn - 1.0);
    cyy = (syy - sy * my) / (dn - 1.0);
    czz = szz / (dn - 1.0);
  }
  const double hd = 0.5 * (cxx - cyy), tr = 0.5 * (cxx + cyy);
  const double rad = sqrt(hd * hd + cxy * cxy);
  const double l1 = tr - rad, l2 = tr + rad;
  double vx, vy;
  if (rad == 0.0) { vx = 1.0; vy = 0.0; }
  else if (hd >= 0.0) { vx = hd + rad; vy = cxy; }
  else { vx = cxy; vy = rad - hd; }
  const double vn = sqrt(vx * vx + vy * vy);
  if (vn == 0.0) { vx = 1.0; vy = 0.0; } else { vx /= vn; vy /= vn; }
  // ascending order of {l1, l2, czz}; z first among equals
  double ev0, ev1, ev2; int k0, k1, k2;   // kind: 0 = l1, 1 = l2, 2 = z
  if (czz <= l1)      { ev0 = czz; k0 = 2; ev1 = l1; k1 = 0; ev2 = l2; k2 = 1; }
  else if (czz <= l2) { ev0 = l1; k0 = 0; ev1 = czz; k1 = 2; ev2 = l2; k2 = 1; }
  else                { ev0 = l1; k0 = 0; ev1 = l2; k1 = 1; ev2 = czz; k2 = 2; }
  if (ev0 < 0 || ev1 < 0 || ev2 <= 0) return 0;
  const double thr = L.eig_mult * ev2;
  bool rebuilt = false;
  if (ev0 < thr) { ev0 = thr; if (ev1 < thr) ev1 = thr; rebuilt = true; }
  double n1 = l1, n2 = l2;
  if (k0 == 0) n1 = ev0; else if (k0 == 1) n2 = ev0;
  if (k1 == 0) n1 = ev1; else if (k1 == 1) n2 = ev1;
  if (k2 == 0) n1 = ev2; else if (k2 == 1) n2 = ev2;
  if (rebuilt) {
    cxx = n1 * (vy * vy) + n2 * (vx * vx);
    cxy = -n1 * (vx * vy) + n2 * (vx * vy);
    cyy = n1 * (vx * vx) + n2 * (vy * vy);
  }
  const double det = cxx * cyy - cxy * cxy;
  icov[0] = cyy / det; icov[1] = -cxy / det; icov[2] = cxx / det;
  for (int a = 0; a < 3; ++a)
    if (icov[a] == (double)INFINITY || icov[a] == -(double)INFINITY) return -1;
  return 1;
}

// Cell record of one voxel from its sums (shared by the two kernels below).
__device__ __forceinline__ int write_voxel(const GridDims &G, const LeafParams &L, size_t g, int n, float fx, float fy,
                                           double sx, double sy, double sxx, double sxy, double syy, double szz,
                                           float2 *__restrict__ cent, double *__restrict__ rec, int *__restrict__ /* counters: unused */) {
  if (n < L.min_pts) return 0;
  const int ix = (int)(g % G.div_x), iy = (int)(g / G.div_x);
  const size_t pg = (size_t)(iy + 2) * G.gw + (ix + 2);
  double mean[2], icov[3];
  const int ok = leaf_finalize(L, n, sx, sy, sxx, sxy, syy, szz, mean, icov);
  cent[pg] = make_float2(fx / (float)n, fy / (float)n);
  double *r = rec + pg * 8;
  r[0] = mean[0]; r[1] = mean[1]; r[2] = icov[0]; r[3] = icov[1]; r[4] = icov[2];
  // the float32 centroid once more, in the record's own 64-byte line: the window staging of the match kernel then reads
  // ONE line per voxel instead of the record line + a line of the centroid grid (fill_window)
  r[5] = __longlong_as_double((long long)(((u64)__float_as_uint(fy / (float)n) << 32) | (u64)__float_as_uint(fx / (float)n)));
  return ok > 0 ? n : -n;
}

// One lane per voxel: sequential sums in cloud order (float32 centroid, fp64 mean / Sxx), bucketed
// copy of the raw points, cell record.
// Round 4: the points of a wave's 64 consecutive voxels are one contiguous range of `pts`; the wave loads it into LDS with
// coalesced reads, all in flight together, and every lane then walks its own voxel there.  Before, each lane streamed its
// voxel from memory eight points at a time, and the kernel lasted as long as its fullest voxel's chain of load round
// trips (128 points: sixteen of them).  Same additions in the same order: the sums are unchanged bit for bit.
constexpr int kFinStage = 1536;               // points per wave in LDS (12 KiB; a wave whose voxels hold more streams from memory)
__global__ void __launch_bounds__(256)
map_finalize_kernel(GridDims G, LeafParams L,
                          const int *__restrict__ start,
                          const float2 *__restrict__ pts, float2 *__restrict__ cent, double *__restrict__ rec,
                          int *__restrict__ npts_grid, int *__restrict__ counters /* unused */,
                          unsigned *__restrict__ occ /* (ng + 31) / 32 words: voxel in the search set */,
                          u64 *__restrict__ tiles, int tiles_w /* voxels with raw points, 8 x 8 per word (MapView::tiles) */) {
  __shared__ float2 lp[4][kFinStage];
  const size_t ng = (size_t)G.div_x * G.div_y;
  size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const bool live = g < ng;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int s0 = 0, s1 = 0;
  if (live) { s0 = start[g]; s1 = start[g + 1]; }
  const int n = s1 - s0;
  // the wave's range: from its first live lane's s0 to its last live lane's s1 (start[] is non-decreasing)
  const int a = __shfl(s0, 0);
  int b = s1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) b = max(b, __shfl_xor(b, o));
  const int m = b - a;
  const bool in_lds = m <= kFinStage;
  if (in_lds) {
    for (int j0 = 0; j0 < m; j0 += 8 * 64) {      // eight coalesced loads per lane in flight
      float2 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int j = j0 + u * 64 + lane; t[u] = pts[a + min(j, m - 1)]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int j = j0 + u * 64 + lane; if (j < m) lp[wv][j] = t[u]; }
    }
    __builtin_amdgcn_wave_barrier();
  }
  int flag = 0;
  if (n > 0) {
    float fx = 0.f, fy = 0.f;
    double sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0, szz = 0;
    if (L.cov_init_identity) { sxx = 1.0; syy = 1.0; szz = 1.0; }
    if (in_lds) {
      const float2 *q = &lp[wv][s0 - a];
      for (int s = 0; s < n; ++s) {
        const float2 p = q[s];                  // strictly in cloud order: these sums define the voxel
        fx += p.x; fy += p.y;
        const double X = (double)p.x, Y = (double)p.y;
        sx += X; sy += Y;
        sxx += X * X; sxy += X * Y; syy += Y * Y;
      }
    } else {
      // eight loads in flight, consecutive addresses (map_order_kernel put the points in cloud order), and the NEXT eight
      // issued before these are added up
      float2 pb[8], pn[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pb[u] = pts[min(s0 + u, s1 - 1)];
      for (int s = s0; s < s1; s += 8) {
        if (s + 8 < s1) {
#pragma unroll
          for (int u = 0; u < 8; ++u) pn[u] = pts[min(s + 8 + u, s1 - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (s + u >= s1) break;
          const float2 p = pb[u];
          fx += p.x; fy += p.y;
          const double X = (double)p.x, Y = (double)p.y;
          sx += X; sy += Y;
          sxx += X * X; sxy += X * Y; syy += Y * Y;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) pb[u] = pn[u];
      }
    }
    flag = write_voxel(G, L, g, n, fx, fy, sx, sy, sxx, sxy, syy, szz, cent, rec, counters);
    const int vy = (int)(g / (size_t)G.div_x), vx = (int)(g - (size_t)vy * G.div_x);
    atomicOr(tiles + (size_t)((vy >> 3) + 1) * tiles_w + (vx >> 3) + 1, 1ull << (8 * (vy & 7) + (vx & 7)));
  }
  if (live) npts_grid[g] = flag;
  const u64 in_set = __ballot(flag != 0);       // the wave's 64 consecutive voxels (blockDim is a multiple of 64)
  if ((threadIdx.x & 63) == 0 && live) {
    occ[g >> 5] = (unsigned)in_set;
    if ((g >> 5) + 1 < (ng + 31) / 32) occ[(g >> 5) + 1] = (unsigned)(in_set >> 32);
  }
}

// reset of the centroid grid and of the occupancy tiles (side stream, beside the bucketing chain)
__global__ void fill_f2_kernel(float2 *p, size_t n, float v, u64 *tiles, size_t ntiles) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float2(v, v);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < ntiles; i += (size_t)gridDim.x * blockDim.x)
    tiles[i] = 0ull;
}
