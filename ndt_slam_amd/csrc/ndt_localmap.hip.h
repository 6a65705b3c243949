// the local-map assembly, Submap::makeMap (src/PointCloudMap.cpp:15-39), on the device.
//
// For every scan triple (s_i, s_i+1, s_i+2) the reference builds a pcl::octree::OctreePointCloudChangeDetector
// over s_i ++ s_i+2, switches buffers, adds s_i+1 and asks for the points in new leaf voxels
// (PCFilter::difference_extraction, include/ndt_slam/PCFilter.h:58-94), then drops every point of s_i+1 closer
// than thre_neighbor to one of those (PCFilter::remove_neighborPoint, :29-56).  Only the SET of new-voxel
// points reaches the result, so the pointer octree is replaced by what it computes:
//
//   * the octree's voxel lattice is anchored by the first point and its bounding box doubles towards every
//     point that falls outside (OctreePointCloud::adoptBoundingBoxToPoint).  Growth is the only sequential
//     part; one workgroup replays it by repeatedly finding the first point outside the current box, which
//     costs one round per tree level (<= 31), not per point;
//   * a point's voxel key is computed with the box minimum of the moment it was inserted, exactly as
//     genOctreeKeyforPoint does in fp64, and moved into the final frame by the integer shifts of the later
//     growth steps (the old root becomes child (!upX, !upY, !upZ) of the new one);
//   * "leaf exists in the previous buffer" is membership in a hash set of the base cloud's final keys (in LDS
//     when the cloud occupies few enough voxels, else in HBM).
//
// One 1024-thread workgroup per triple does frame, keys, set and difference, the triples of a submap side by
// side; the neighbour removal and the concatenation then run over 256-point units on the whole chip.
// Clouds are z = 0 (src/PointCloudMap.cpp:71), so z never leaves the box -- but the box still grows DOWNWARDS in z
// at every doubling (the child index is built from the upper-bound flags alone), and the z key of a point is
// (unsigned)((0 - min_z) / resol) in fp64 with the min_z of the moment it was inserted: min_z = -resol (2^depth - 1)
// accumulates rounding for most values of resol, and the quotient then truncates to 2^depth - 2 in some frames and
// to 2^depth - 1 in others.  Two points of one (x, y) column inserted at different depths can therefore lie in
// DIFFERENT leaves; the replay carries min_z and records that deviation per frame, and it is part of the key.
// (The tests check all of this against a literal two-buffer pointer octree, also at the resolutions where this
// happens: 0.03, 0.3, 0.02, 0.07.)

constexpr int kMmBlock = 1024, kMmWaves = kMmBlock / 64, kMmEvents = 40, kMmMaxDepth = 30, kMmTile = 1024;
constexpr int kMmUnit = 256, kMmPer = 4, kMmGroup = 8, kMmIns = 8, kMmLdsTab = 16384;
constexpr unsigned long long kMmEmpty = ~0ull;

struct MmJob {
  const float *a0, *a1, *b;            // base cloud = a0 ++ a1 (scans i and i+2), test cloud = b (scan i+1)
  unsigned n0, n1, nb;
  unsigned sa, sb;                     // point strides in bytes
  unsigned tab_mask;                   // hash set capacity - 1 (capacity a power of two >= 2 (n0 + n1) + 2)
  unsigned long long *tab;             // preset to kMmEmpty
  float2 *diff;                        // points of b in new voxels, input order (room for nb)
  unsigned long long *n_diff;          // their count; ~0 when the clouds span more than 2^30 voxels
};

struct MmUnit {                        // at most kMmUnit consecutive points of the concatenated result
  const float *src; unsigned stride; unsigned n;
  int job;                             // >= 0: points of that triple's middle scan, to be filtered; < 0: copied whole
  unsigned pad_;
};

struct MmFrame {
  double minx, miny, maxx, maxy, minz;
  int depth, defined, nev, err, first, from;
  int ev_idx[kMmEvents];
  double ev_minx[kMmEvents], ev_miny[kMmEvents];
  unsigned ev_sx[kMmEvents], ev_sy[kMmEvents];
  unsigned ev_dz[kMmEvents];           // (2^depth - 1) - z key of a point inserted in this frame: 0 or 1
};

// z key deviation of the frame whose box starts at minz (genOctreeKeyforPoint's expression for z = 0)
__device__ inline unsigned mm_dz(double minz, double res, int depth) {
  const unsigned kz = (unsigned)((0.0 - minz) / res);
  const unsigned full = (1u << depth) - 1u;
  return kz <= full ? min(full - kz, 3u) : 3u;
}

__device__ inline float2 mm_fetch(const MmJob &J, int q) {
  if (q < (int)J.n0) return load_pt(J.a0, J.sa, (size_t)q);
  if (q < (int)(J.n0 + J.n1)) return load_pt(J.a1, J.sa, (size_t)(q - (int)J.n0));
  return load_pt(J.b, J.sb, (size_t)(q - (int)(J.n0 + J.n1)));
}

__device__ inline unsigned mm_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 29;
  return (unsigned)k;
}

// (unsigned)(a / res) as genOctreeKeyforPoint computes it, without the fp64 division where the answer is not
// in doubt: k = trunc(a * rinv) is right whenever the exact remainder a - k res (one fma) lies well inside
// (0, res) -- the rounded quotient of a value below 2^31 is off by less than 2^-22 -- and anything closer to a
// voxel border than 1e-6 of a voxel takes the division itself.
__device__ inline unsigned mm_cell(double a, double res, double rinv) {
  unsigned k = (unsigned)(a * rinv);
  const double rem = __builtin_fma(-(double)k, res, a);
  if (!(rem > res * 1e-6 && rem < res * (1.0 - 1e-6))) k = (unsigned)(a / res);
  return k;
}

// voxel key of point q in the octree's final frame
__device__ inline unsigned long long mm_key(const MmFrame &F, float2 p, int q, double res, double rinv) {
  int e = F.nev - 1;
  if (q < F.ev_idx[e]) {                       // inserted before the last growth step: find its frame
    e = 0;
#pragma unroll 1
    for (int k = 1; k < F.nev; ++k) e = F.ev_idx[k] <= q ? k : e;
  }
  const unsigned kx = mm_cell((double)p.x - F.ev_minx[e], res, rinv) + (F.ev_sx[F.nev - 1] - F.ev_sx[e]);
  const unsigned ky = mm_cell((double)p.y - F.ev_miny[e], res, rinv) + (F.ev_sy[F.nev - 1] - F.ev_sy[e]);
  // kx, ky < 2^30 (at most 30 tree levels); every later doubling adds the same 2^depth to every z key, so the
  // frame's deviation is the point's deviation in the final frame
  return ((unsigned long long)F.ev_dz[e] << 60) | ((unsigned long long)kx << 30) | (unsigned long long)ky;
}

// order-preserving append of up to kMmPer flagged points per thread (point k of a thread has index
// c0 + k * kMmBlock + tid, so the order is k-major); returns the new base (uniform)
__device__ inline int mm_append4(const bool flag[kMmPer], const float2 p[kMmPer], float2 *dst, int base,
                                 int (*wcnt)[kMmWaves]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long b[kMmPer];
#pragma unroll
  for (int k = 0; k < kMmPer; ++k) {
    b[k] = __ballot(flag[k]);
    if (lane == 0) wcnt[k][wv] = __builtin_popcountll(b[k]);
  }
  __syncthreads();
  int run = base;
#pragma unroll
  for (int k = 0; k < kMmPer; ++k) {
    int off = run, tot = 0;
#pragma unroll
    for (int w = 0; w < kMmWaves; ++w) { const int c = wcnt[k][w]; off += w < wv ? c : 0; tot += c; }
    if (flag[k]) dst[off + __builtin_popcountll(b[k] & ((1ull << lane) - 1ull))] = p[k];
    run += tot;
  }
  __syncthreads();
  return run;
}

__global__ void __launch_bounds__(kMmBlock)
make_map_diff_kernel(const MmJob *__restrict__ jobs, double res) {
  __shared__ MmFrame F;
  __shared__ int wcnt[kMmPer][kMmWaves];
  __shared__ unsigned long long ltab[kMmLdsTab];
  __shared__ int lfill, lover;
  const MmJob J = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63;
  const int nA = (int)(J.n0 + J.n1), N = nA + (int)J.nb;
  if (tid == 0) { F.defined = 0; F.nev = 0; F.err = 0; F.from = 0; F.depth = 0; F.first = 0x7fffffff; }
  __syncthreads();

  // ---- 1. replay of the bounding-box growth over base ++ test (adoptBoundingBoxToPoint) ----
  // The box only changes at a point outside it: kMmGroup chunks are tested against the current box at once and
  // passed over when none of their points is outside; otherwise every round finds the first such point of
  // the group and adds one tree level for it.
  for (int g0 = 0; g0 < N; g0 += kMmGroup * kMmBlock) {
    float2 p[kMmGroup]; bool fin[kMmGroup];
    bool out_any = false;
#pragma unroll
    for (int k = 0; k < kMmGroup; ++k) {
      const int q = g0 + k * kMmBlock + tid;
      p[k] = make_float2(0.f, 0.f);
      if (q < N) p[k] = mm_fetch(J, q);
      fin[k] = q < N && isfinite(p[k].x) && isfinite(p[k].y);          // addPointsFromInputCloud: isFinite
    }
#pragma unroll
    for (int k = 0; k < kMmGroup; ++k)
      out_any = out_any || (fin[k] && (!F.defined || (double)p[k].x < F.minx || (double)p[k].x >= F.maxx ||
                                      (double)p[k].y < F.miny || (double)p[k].y >= F.maxy));
    if (!__syncthreads_or(out_any ? 1 : 0)) continue;
    for (int rep = 0; rep <= kMmEvents; ++rep) {                        // every round but the last adds one tree level
      // the first point of the group (k-major order) outside the current box, among those not yet cleared
      int myq = 0x7fffffff;
#pragma unroll
      for (int k = kMmGroup - 1; k >= 0; --k) {
        const int q = g0 + k * kMmBlock + tid;
        const bool viol = fin[k] && q >= F.from &&
                          (!F.defined || (double)p[k].x < F.minx || (double)p[k].x >= F.maxx ||
                           (double)p[k].y < F.miny || (double)p[k].y >= F.maxy);
        myq = viol ? q : myq;
      }
      int wq = myq;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wq = min(wq, __shfl_xor(wq, o));
      if (lane == 0 && wq != 0x7fffffff) atomicMin(&F.first, wq);
      __syncthreads();
      const int j = F.first;
      const bool stop = j == 0x7fffffff || F.err != 0;
      __syncthreads();
      if (stop) break;
      if (myq == j) {
        float2 pk = p[0];
#pragma unroll
        for (int k = 1; k < kMmGroup; ++k) pk = (g0 + k * kMmBlock + tid == j) ? p[k] : pk;
        const bool upx = (double)pk.x >= F.maxx, upy = (double)pk.y >= F.maxy;
        const int e = F.nev;
        if (!F.defined) {
          // first point: box of one voxel around it, then getKeyBitSize: depth 1 (two voxels per axis) and the
          // box widened symmetrically to that size
          double mn[2] = {(double)pk.x - res / 2, (double)pk.y - res / 2};
          double mx[2] = {(double)pk.x + res / 2, (double)pk.y + res / 2};
          const double side = 2.0 * res;
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const double over = (side - (mx[a] - mn[a])) / 2.0;
            if (over > (double)FLT_EPSILON) { mn[a] -= over; mx[a] += over; }
          }
          F.minx = mn[0]; F.miny = mn[1]; F.maxx = mx[0]; F.maxy = mx[1];
          {                                     // the z axis of the same box (z = 0)
            double mnz = 0.0 - res / 2, mxz = 0.0 + res / 2;
            const double over = (side - (mxz - mnz)) / 2.0;
            if (over > (double)FLT_EPSILON) mnz -= over;
            F.minz = mnz;
          }
          F.depth = 1; F.defined = 1;
          F.ev_sx[0] = 0u; F.ev_sy[0] = 0u;
        } else if (F.depth >= kMmMaxDepth) {
          F.err = 1;
        } else {
          // one more tree level: the old root becomes the child on the side away from the violation
          double side = (double)(1 << F.depth) * res;
          unsigned sx = F.ev_sx[e - 1], sy = F.ev_sy[e - 1];
          if (!upx) { F.minx -= side; sx += 1u << F.depth; }
          if (!upy) { F.miny -= side; sy += 1u << F.depth; }
          F.minz -= side;                       // z = 0 is never at or above max_z: the box always grows downwards in z
          F.depth += 1;
          side = (double)(1 << F.depth) * res - (double)FLT_EPSILON;
          F.maxx = F.minx + side; F.maxy = F.miny + side;
          F.ev_sx[e] = sx; F.ev_sy[e] = sy;
        }
        if (!F.err) {
          F.ev_idx[e] = j; F.ev_minx[e] = F.minx; F.ev_miny[e] = F.miny; F.ev_dz[e] = mm_dz(F.minz, res, F.depth);
          F.nev = e + 1;
        }
        F.from = j;                           // the point itself is tested again against the larger box
        F.first = 0x7fffffff;
      }
      __syncthreads();
    }
    __syncthreads();
    if (F.err) break;
  }
  __syncthreads();
  if (F.err) {
    if (tid == 0) *J.n_diff = kMmEmpty;
    return;
  }
  const double rinv = 1.0 / res;

  // ---- 2. voxels of the base cloud into the set ----
  // A scan revisits the same few thousand voxels, so the set is first tried in LDS (16k entries, probes cost a
  // fraction of a microsecond); only when the base cloud occupies more than 12k voxels is it rebuilt in the
  // HBM table the host prepared (kMmIns probes in flight per thread there: every probe is a round trip).
  for (int i = tid; i < kMmLdsTab; i += kMmBlock) ltab[i] = kMmEmpty;
  if (tid == 0) { lfill = 0; lover = 0; }
  __syncthreads();
  for (int q0 = tid; q0 - tid < nA; q0 += kMmIns * kMmBlock) {   // (whole waves: the duplicate filter is a shuffle)
    float2 p[kMmIns];
#pragma unroll
    for (int k = 0; k < kMmIns; ++k) {                            // the loads of a batch in flight together
      const int q = q0 + k * kMmBlock;
      p[k] = make_float2(0.f, 0.f);
      if (q < nA) p[k] = mm_fetch(J, q);
    }
#pragma unroll
    for (int k = 0; k < kMmIns; ++k) {
      const int q = q0 + k * kMmBlock;
      bool open = q < nA && isfinite(p[k].x) && isfinite(p[k].y);
      const unsigned long long key = open ? mm_key(F, p[k], q, res, rinv) : kMmEmpty;
      const unsigned long long prev = __shfl_up(key, 1);
      if (lane > 0 && prev == key) open = false;
      if (open && !lover) {
        unsigned h = mm_hash(key) & (unsigned)(kMmLdsTab - 1);
        for (int t = 0; t < kMmLdsTab; ++t) {
          const unsigned long long old = atomicCAS(&ltab[h], kMmEmpty, key);
          if (old == kMmEmpty) { if (atomicAdd(&lfill, 1) >= kMmLdsTab * 3 / 4) lover = 1; break; }
          if (old == key || lover) break;
          h = (h + 1u) & (unsigned)(kMmLdsTab - 1);
        }
      }
    }
  }
  __syncthreads();
  const bool in_lds = lover == 0;
  for (int q0 = tid; !in_lds && q0 < nA; q0 += kMmIns * kMmBlock) {
    unsigned long long key[kMmIns]; unsigned h[kMmIns]; bool open[kMmIns];
#pragma unroll
    for (int k = 0; k < kMmIns; ++k) {
      const int q = q0 + k * kMmBlock;
      float2 p = make_float2(0.f, 0.f);
      if (q < nA) p = mm_fetch(J, q);
      open[k] = q < nA && isfinite(p.x) && isfinite(p.y);
      key[k] = open[k] ? mm_key(F, p, q, res, rinv) : kMmEmpty;
      // consecutive points of a scan mostly share their voxel: only the first lane of a run of equal keys
      // goes to memory (same-address atomics serialise in L2)
      const unsigned long long prev = __shfl_up(key[k], 1);
      if (lane > 0 && prev == key[k]) open[k] = false;
      h[k] = mm_hash(key[k]) & J.tab_mask;
    }
    for (unsigned t = 0; t <= J.tab_mask; ++t) {
      unsigned long long old[kMmIns];
#pragma unroll
      for (int k = 0; k < kMmIns; ++k) old[k] = open[k] ? atomicCAS(&J.tab[h[k]], kMmEmpty, key[k]) : 0ull;
      bool any = false;
#pragma unroll
      for (int k = 0; k < kMmIns; ++k) {
        if (open[k] && (old[k] == kMmEmpty || old[k] == key[k])) open[k] = false;
        h[k] = (h[k] + 1u) & J.tab_mask;
        any = any || open[k];
      }
      if (!any) break;
    }
  }
  __syncthreads();

  // ---- 3. points of the test cloud in voxels the base cloud does not occupy ----
  const unsigned tmask = in_lds ? (unsigned)(kMmLdsTab - 1) : J.tab_mask;
  int nd = 0;
  for (int c0 = 0; c0 < (int)J.nb; c0 += kMmPer * kMmBlock) {
    float2 p[kMmPer]; unsigned long long key[kMmPer]; unsigned h[kMmPer]; bool open[kMmPer], isnew[kMmPer];
#pragma unroll
    for (int k = 0; k < kMmPer; ++k) {
      const int i = c0 + k * kMmBlock + tid;
      p[k] = make_float2(0.f, 0.f);
      if (i < (int)J.nb) p[k] = load_pt(J.b, J.sb, (size_t)i);
      open[k] = i < (int)J.nb && isfinite(p[k].x) && isfinite(p[k].y);
      isnew[k] = open[k];
      key[k] = open[k] ? mm_key(F, p[k], nA + i, res, rinv) : 0ull;
      h[k] = mm_hash(key[k]) & tmask;
    }
    for (unsigned t = 0; t <= tmask; ++t) {
      unsigned long long cur[kMmPer];
#pragma unroll
      for (int k = 0; k < kMmPer; ++k) {
        if (in_lds) cur[k] = open[k] ? ltab[h[k]] : kMmEmpty;
        else cur[k] = open[k] ? __hip_atomic_load(&J.tab[h[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kMmEmpty;
      }
      bool any = false;
#pragma unroll
      for (int k = 0; k < kMmPer; ++k) {
        if (open[k] && cur[k] == key[k]) { isnew[k] = false; open[k] = false; }
        if (cur[k] == kMmEmpty) open[k] = false;
        h[k] = (h[k] + 1u) & tmask;
        any = any || open[k];
      }
      if (!any) break;
    }
    nd = mm_append4(isnew, p, J.diff, nd, wcnt);
  }
  if (tid == 0) *J.n_diff = (unsigned long long)nd;
}

// ---- remove_neighborPoint(test, diff) and the concatenation, spread over the chip: the result is cut into
// units of at most 256 points -- stretches of a scan that is appended whole, or of a middle scan whose points
// are tested against their triple's difference list (all pairs, float32 distance, strict <; rn_cutoff).
__global__ void __launch_bounds__(kMmUnit)
make_map_flag_kernel(const MmJob *__restrict__ jobs, const MmUnit *__restrict__ units, float cut,
                     unsigned long long *__restrict__ keepbits, unsigned *__restrict__ unit_cnt) {
  __shared__ float2 tile[kMmTile];
  __shared__ int wsum[kMmUnit / 64];
  const MmUnit U = units[blockIdx.x];
  const int tid = threadIdx.x;
  if (U.job < 0) { if (tid == 0) unit_cnt[blockIdx.x] = U.n; return; }
  const MmJob &J = jobs[U.job];
  const unsigned long long ndl = *J.n_diff;
  if (ndl == kMmEmpty) { if (tid == 0) unit_cnt[blockIdx.x] = 0xffffffffu; return; }
  const int nd = (int)ndl;
  float2 p = make_float2(0.f, 0.f);
  if (tid < (int)U.n) p = load_pt(U.src, U.stride, (size_t)tid);
  bool keep = tid < (int)U.n;
  for (int t0 = 0; t0 < nd; t0 += kMmTile) {
    const int m = min(kMmTile, nd - t0);
    __syncthreads();
    for (int j = tid; j < m; j += kMmUnit) tile[j] = J.diff[t0 + j];
    __syncthreads();
    if (keep) {
      bool near = false;
      for (int j = 0; j < m; ++j) near = near || rn_near(p, tile[j], cut);
      keep = !near;
    }
  }
  const unsigned long long b = __ballot(keep);
  if ((tid & 63) == 0) { keepbits[(size_t)blockIdx.x * (kMmUnit / 64) + (tid >> 6)] = b; wsum[tid >> 6] = __builtin_popcountll(b); }
  __syncthreads();
  if (tid == 0) { int c = 0; for (int w = 0; w < kMmUnit / 64; ++w) c += wsum[w]; unit_cnt[blockIdx.x] = (unsigned)c; }
}

// exclusive scan of the unit counts (one workgroup); total to *n_out, ~0 when a triple failed
__global__ void __launch_bounds__(1024)
make_map_offsets_kernel(const unsigned *__restrict__ unit_cnt, int nu, unsigned long long *__restrict__ unit_off,
                        unsigned long long *__restrict__ n_out) {
  __shared__ unsigned long long sh[1024];
  __shared__ unsigned long long carry;
  __shared__ int bad;
  if (threadIdx.x == 0) { carry = 0ull; bad = 0; }
  __syncthreads();
  for (int base = 0; base < nu; base += 1024) {
    const int i = base + threadIdx.x;
    unsigned c = i < nu ? unit_cnt[i] : 0u;
    if (c == 0xffffffffu) { bad = 1; c = 0u; }
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const unsigned long long t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0ull;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nu) unit_off[i] = carry + sh[threadIdx.x] - c;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = bad ? kMmEmpty : carry;
}

__global__ void __launch_bounds__(kMmUnit)
make_map_copy_kernel(const MmUnit *__restrict__ units, const unsigned long long *__restrict__ keepbits,
                     const unsigned long long *__restrict__ unit_off, const unsigned long long *__restrict__ n_out,
                     float2 *__restrict__ out) {
  if (*n_out == kMmEmpty) return;
  const MmUnit U = units[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned long long off = unit_off[blockIdx.x];
  if (U.job < 0) {
    if (tid < (int)U.n) out[off + tid] = load_pt(U.src, U.stride, (size_t)tid);
    return;
  }
  const unsigned long long *kb = keepbits + (size_t)blockIdx.x * (kMmUnit / 64);
  unsigned long long mine = 0ull;
#pragma unroll
  for (int w = 0; w < kMmUnit / 64; ++w) {
    const unsigned long long b = kb[w];
    off += w < wv ? (unsigned long long)__builtin_popcountll(b) : 0ull;
    mine = w == wv ? b : mine;
  }
  if ((mine >> lane) & 1ull)
    out[off + __builtin_popcountll(mine & ((1ull << lane) - 1ull))] = load_pt(U.src, U.stride, (size_t)tid);
}
