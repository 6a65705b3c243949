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
//   * "leaf exists in the previous buffer" is membership in a hash set of the base cloud's final keys.
//
// One 1024-thread workgroup per triple does frame, keys, set, difference and neighbour removal; the triples of
// a submap run side by side.  Clouds are z = 0 (src/PointCloudMap.cpp:71), so z never leaves the box and has
// the same key for every point.  (The tests check this against a literal two-buffer pointer octree.)

constexpr int kMmBlock = 1024, kMmWaves = kMmBlock / 64, kMmEvents = 40, kMmMaxDepth = 30, kMmTile = 1024;
constexpr unsigned long long kMmEmpty = ~0ull;

struct MmJob {
  const float *a0, *a1, *b;            // base cloud = a0 ++ a1 (scans i and i+2), test cloud = b (scan i+1)
  unsigned n0, n1, nb;
  unsigned sa, sb;                     // point strides in bytes
  unsigned tab_mask;                   // hash set capacity - 1 (capacity a power of two >= 2 (n0 + n1) + 2)
  unsigned long long *tab;             // preset to kMmEmpty
  float2 *diff;                        // points of b in new voxels, input order (room for nb)
  float2 *kept;                        // points of b that survive the removal (room for nb); null = difference only
  unsigned long long *n_diff, *n_kept; // counts; ~0 when the clouds span more than 2^30 voxels
};

struct MmSeg {                         // one piece of the concatenated result
  const float *src; unsigned stride; unsigned n;
  const unsigned long long *n_dev;     // when set, the count lives on the device (a kept list)
};

struct MmFrame {
  double minx, miny, maxx, maxy;
  int depth, defined, nev, err, first, from;
  int ev_idx[kMmEvents];
  double ev_minx[kMmEvents], ev_miny[kMmEvents];
  unsigned ev_sx[kMmEvents], ev_sy[kMmEvents];
};

__device__ inline float2 mm_fetch(const MmJob &J, int q) {
  if (q < (int)J.n0) return load_pt(J.a0, J.sa, (size_t)q);
  if (q < (int)(J.n0 + J.n1)) return load_pt(J.a1, J.sa, (size_t)(q - (int)J.n0));
  return load_pt(J.b, J.sb, (size_t)(q - (int)(J.n0 + J.n1)));
}

__device__ inline unsigned mm_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 29;
  return (unsigned)k;
}

// voxel key of point q in the octree's final frame
__device__ inline unsigned long long mm_key(const MmFrame &F, float2 p, int q, double res) {
  int e = 0;
#pragma unroll 1
  for (int k = 1; k < F.nev; ++k) e = F.ev_idx[k] <= q ? k : e;
  const unsigned kx = (unsigned)(((double)p.x - F.ev_minx[e]) / res) + (F.ev_sx[F.nev - 1] - F.ev_sx[e]);
  const unsigned ky = (unsigned)(((double)p.y - F.ev_miny[e]) / res) + (F.ev_sy[F.nev - 1] - F.ev_sy[e]);
  return ((unsigned long long)kx << 32) | (unsigned long long)ky;
}

// order-preserving append of the flagged lanes' points to dst[base ...]; returns the new base (uniform)
__device__ inline int mm_append(bool flag, float2 p, float2 *dst, int base, int *wcnt) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long b = __ballot(flag);
  if (lane == 0) wcnt[wv] = __builtin_popcountll(b);
  __syncthreads();
  int off = base, tot = 0;
#pragma unroll
  for (int w = 0; w < kMmWaves; ++w) { const int c = wcnt[w]; off += w < wv ? c : 0; tot += c; }
  if (flag) dst[off + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = p;
  __syncthreads();
  return base + tot;
}

__global__ void __launch_bounds__(kMmBlock)
make_map_triple_kernel(const MmJob *__restrict__ jobs, double res, double thre) {
  __shared__ MmFrame F;
  __shared__ float2 tile[kMmTile];
  __shared__ int wcnt[kMmWaves];
  const MmJob J = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  const int nA = (int)(J.n0 + J.n1), N = nA + (int)J.nb;
  if (tid == 0) { F.defined = 0; F.nev = 0; F.err = 0; F.from = 0; F.depth = 0; F.first = 0x7fffffff; }
  __syncthreads();

  // ---- 1. replay of the bounding-box growth over base ++ test (adoptBoundingBoxToPoint) ----
  for (int c0 = 0; c0 < N; c0 += kMmBlock) {
    const int q = c0 + tid;
    float2 p = make_float2(0.f, 0.f);
    if (q < N) p = mm_fetch(J, q);
    const bool fin = q < N && isfinite(p.x) && isfinite(p.y);          // addPointsFromInputCloud: isFinite
    for (int rep = 0; rep <= kMmEvents; ++rep) {                        // every round but the last adds one tree level
      const bool upx = (double)p.x >= F.maxx, upy = (double)p.y >= F.maxy;
      const bool viol = fin && q >= F.from &&
                        (!F.defined || (double)p.x < F.minx || upx || (double)p.y < F.miny || upy);
      const unsigned long long vb = __ballot(viol);
      if (viol && (tid & 63) == __builtin_ctzll(vb)) atomicMin(&F.first, q);     // q rises with the lane
      __syncthreads();
      const int j = F.first;
      const bool stop = j == 0x7fffffff || F.err != 0;
      __syncthreads();
      if (stop) break;
      if (q == j) {
        const int k = F.nev;
        if (!F.defined) {
          // first point: box of one voxel around it, then getKeyBitSize: depth 1 (two voxels per axis) and the
          // box widened symmetrically to that size
          double mn[2] = {(double)p.x - res / 2, (double)p.y - res / 2};
          double mx[2] = {(double)p.x + res / 2, (double)p.y + res / 2};
          const double side = 2.0 * res;
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const double over = (side - (mx[a] - mn[a])) / 2.0;
            if (over > (double)FLT_EPSILON) { mn[a] -= over; mx[a] += over; }
          }
          F.minx = mn[0]; F.miny = mn[1]; F.maxx = mx[0]; F.maxy = mx[1];
          F.depth = 1; F.defined = 1;
          F.ev_sx[0] = 0u; F.ev_sy[0] = 0u;
        } else if (F.depth >= kMmMaxDepth) {
          F.err = 1;
        } else {
          // one more tree level: the old root becomes the child on the side away from the violation
          double side = (double)(1 << F.depth) * res;
          unsigned sx = F.ev_sx[k - 1], sy = F.ev_sy[k - 1];
          if (!upx) { F.minx -= side; sx += 1u << F.depth; }
          if (!upy) { F.miny -= side; sy += 1u << F.depth; }
          F.depth += 1;
          side = (double)(1 << F.depth) * res - (double)FLT_EPSILON;
          F.maxx = F.minx + side; F.maxy = F.miny + side;
          F.ev_sx[k] = sx; F.ev_sy[k] = sy;
        }
        if (!F.err) { F.ev_idx[k] = j; F.ev_minx[k] = F.minx; F.ev_miny[k] = F.miny; F.nev = k + 1; }
        F.from = j;                         // the point itself is tested again against the larger box
        F.first = 0x7fffffff;
      }
      __syncthreads();
    }
    if (F.err) break;
  }
  __syncthreads();
  if (F.err) {
    if (tid == 0) { *J.n_diff = kMmEmpty; if (J.n_kept) *J.n_kept = kMmEmpty; }
    return;
  }

  // ---- 2. voxels of the base cloud into the set ----
  for (int q = tid; q < nA; q += kMmBlock) {
    const float2 p = mm_fetch(J, q);
    if (!(isfinite(p.x) && isfinite(p.y))) continue;
    const unsigned long long key = mm_key(F, p, q, res);
    unsigned h = mm_hash(key) & J.tab_mask;
    for (unsigned t = 0; t <= J.tab_mask; ++t) {
      const unsigned long long old = atomicCAS(&J.tab[h], kMmEmpty, key);
      if (old == kMmEmpty || old == key) break;
      h = (h + 1u) & J.tab_mask;
    }
  }
  __syncthreads();

  // ---- 3. points of the test cloud in voxels the base cloud does not occupy ----
  int nd = 0;
  for (int c0 = 0; c0 < (int)J.nb; c0 += kMmBlock) {
    const int i = c0 + tid;
    float2 p = make_float2(0.f, 0.f);
    bool isnew = false;
    if (i < (int)J.nb) {
      p = load_pt(J.b, J.sb, (size_t)i);
      if (isfinite(p.x) && isfinite(p.y)) {
        const unsigned long long key = mm_key(F, p, nA + i, res);
        unsigned h = mm_hash(key) & J.tab_mask;
        isnew = true;
        for (unsigned t = 0; t <= J.tab_mask; ++t) {
          const unsigned long long cur = __hip_atomic_load(&J.tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (cur == key) { isnew = false; break; }
          if (cur == kMmEmpty) break;
          h = (h + 1u) & J.tab_mask;
        }
      }
    }
    nd = mm_append(isnew, p, J.diff, nd, wcnt);
  }
  if (tid == 0) *J.n_diff = (unsigned long long)nd;
  if (!J.kept) return;

  // ---- 4. remove_neighborPoint(test, diff): all pairs, float32 distance, strict < ----
  int nk = 0;
  for (int c0 = 0; c0 < (int)J.nb; c0 += kMmBlock) {
    const int i = c0 + tid;
    float2 p = make_float2(0.f, 0.f);
    if (i < (int)J.nb) p = load_pt(J.b, J.sb, (size_t)i);
    bool keep = i < (int)J.nb;
    for (int t0 = 0; t0 < nd; t0 += kMmTile) {
      const int m = min(kMmTile, nd - t0);
      __syncthreads();
      for (int j = tid; j < m; j += kMmBlock) tile[j] = J.diff[t0 + j];
      __syncthreads();
      if (keep) {
        bool near = false;
        for (int j = 0; j < m; ++j) near = near || rn_near(p, tile[j], thre);
        keep = !near;
      }
    }
    nk = mm_append(keep, p, J.kept, nk, wcnt);
  }
  if (tid == 0) *J.n_kept = (unsigned long long)nk;
}

// offsets of the pieces in the concatenated cloud (one thread: a submap has tens of scans)
__global__ void make_map_offsets_kernel(const MmSeg *__restrict__ segs, int nseg, unsigned long long *__restrict__ seg_off,
                                        unsigned long long *__restrict__ n_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long off = 0; bool bad = false;
  for (int s = 0; s < nseg; ++s) {
    const unsigned long long n = segs[s].n_dev ? *segs[s].n_dev : (unsigned long long)segs[s].n;
    if (n == kMmEmpty) bad = true;
    seg_off[s] = bad ? kMmEmpty : off;
    off += bad ? 0ull : n;
  }
  *n_out = bad ? kMmEmpty : off;
}

__global__ void __launch_bounds__(256)
make_map_copy_kernel(const MmSeg *__restrict__ segs, const unsigned long long *__restrict__ seg_off,
                     float2 *__restrict__ out) {
  const MmSeg S = segs[blockIdx.y];
  const unsigned long long off = seg_off[blockIdx.y];
  if (off == kMmEmpty) return;
  const unsigned n = S.n_dev ? (unsigned)*S.n_dev : S.n;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
    out[off + i] = load_pt(S.src, S.stride, (size_t)i);
}
