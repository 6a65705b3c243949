// ndt_fitness.hip.h -- row a7: exact nearest raw map point over the voxel buckets.
// Part of libndt_mi355x.so: included by ndt_mi355x.hip inside its anonymous namespace (one translation
// unit; the order of the includes matters).  Not a standalone header.

// ------------------------------------------------------------------------------------------
// a7: nearest raw map point, exact, no range cut: home voxel, then the ring-1 voxels that can
// still hold a closer point (box-distance pruning), then whole rings while the best distance
// exceeds the ring bound.
// ------------------------------------------------------------------------------------------
// The cost of this search is the number of (lane, cache line) look-ups of its divergent loads --
// the CU's vector L1 serves about one line per clock -- so everything is fetched as wide as the
// layout allows: a bucket's points two per 16-byte load, and the offsets of up to three
// neighbouring voxels of a row in one 16-byte load (pt_start carries 4 readable ints before its
// first entry and 3 after its last one).
struct I4u { int x, y, z, w; };
struct I2u { int x, y; };
typedef ndt_i4v ndt_i4v_u __attribute__((aligned(4)));      // 16 bytes at 4-byte alignment: one dwordx4 load
typedef ndt_i2v ndt_i2v_u __attribute__((aligned(4)));
__device__ __forceinline__ I4u ld_i4u(const int *p) { const ndt_i4v v = *(const NDT_GLOBAL ndt_i4v_u *)p; return I4u{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ I2u ld_i2u(const int *p) { const ndt_i2v v = *(const NDT_GLOBAL ndt_i2v_u *)p; return I2u{v.x, v.y}; }

__device__ __forceinline__ float sq_dist(float qx, float qy, float px, float py) {
  const float ex = qx - px, ey = qy - py;
  return ex * ex + ey * ey;
}

// min over the points pts[s .. se) of the float32 squared distance to (qx, qy).
// The search is bound by the latency of its dependent loads (offsets -> bucket -> next bucket ...), so a bucket is
// fetched kFitWide 16-byte loads (two points each) at a time, all in flight together; loads past the end of the
// bucket re-read its last pair and are masked out.
#ifndef NDT_FIT_WIDE
#define NDT_FIT_WIDE 6
#endif
constexpr int kFitWide = NDT_FIT_WIDE;
__device__ __forceinline__ float scan_bucket(const float2 *__restrict__ pts, int s, int se, float qx,
                                             float qy, float best) {
  if (s >= se) return best;
  if (s & 1) { const float2 p = gld_f2(pts + s); best = fminf(best, sq_dist(qx, qy, p.x, p.y)); ++s; }
  // pairs of points, 16-byte aligned, addressed by 32-bit byte offsets from the (uniform) array base; a load past the
  // end of the bucket re-reads its last pair -- the minimum does not care about a point seen twice, so nothing is masked
  const unsigned npair = (unsigned)(se - s) >> 1, first = (unsigned)s * 8u;
  if (npair) {
    const unsigned last = first + (npair - 1u) * 16u;
    for (unsigned i = 0; i < npair; i += kFitWide) {
      float4 v[kFitWide];
#pragma unroll
      for (int u = 0; u < kFitWide; ++u) v[u] = gld_f4_at(pts, min(first + (i + u) * 16u, last));
#pragma unroll
      for (int u = 0; u < kFitWide; ++u)
        best = fminf(best, fminf(sq_dist(qx, qy, v[u].x, v[u].y), sq_dist(qx, qy, v[u].z, v[u].w)));
    }
  }
  if ((se - s) & 1) { const float2 p = gld_f2(pts + se - 1); best = fminf(best, sq_dist(qx, qy, p.x, p.y)); }
  return best;
}

// The search in two halves so that a caller can have the next query's offsets in flight while this query's
// buckets are read: nearest_prep issues the one load everything else depends on.
struct NearPrep { int cx, cy; bool inside; I4u h, dn, up; };
__device__ __forceinline__ NearPrep nearest_prep(const MapView &M, float qx, float qy) {
  NearPrep P;
  const int cx0 = (int)floorf(qx * M.inv_leaf) - M.min_bx, cy0 = (int)floorf(qy * M.inv_leaf) - M.min_by;
  P.cx = cx0 < 0 ? 0 : (cx0 >= M.div_x ? M.div_x - 1 : cx0);
  P.cy = cy0 < 0 ? 0 : (cy0 >= M.div_y ? M.div_y - 1 : cy0);
  P.inside = (P.cx == cx0) && (P.cy == cy0);
  // offsets of (cx-1, cx, cx+1) of the home row in one load: [left, home) [home, right) [right, end)
  const int *row = M.pt_start + ((size_t)P.cy * M.div_x + P.cx) - 1;
  P.h = ld_i4u(row);
  // the same for the rows below and above (clamped at the grid's edge; used only when that row exists)
  P.dn = ld_i4u(P.cy > 0 ? row - M.div_x : row);
  P.up = ld_i4u(P.cy + 1 < M.div_y ? row + M.div_x : row);
  return P;
}

// State of one query between the phases of the search.
struct NearState {
  float best;                     // squared distance to the nearest point seen so far
  int cx, cy;                     // home voxel (clamped into the grid)
  float wl, wr, wd, wu;           // distances to the walls of the home voxel (shrunk by the rounding slack; 0 if clamped)
  bool more;                      // another voxel can still hold a closer point
  int rs[4], rn[4];               // ring 1 as (up to) four ranges of the bucketed points: start, length
};

// distances from the query to the four walls of its voxel, shrunk by 1e-3 leaf so that a point
// the float32 voxel rounding put on the other side of a wall is never pruned away
// (the cell of a stored point comes from floorf(x * inv_leaf) in float32: the point can sit |x| 2^-23 beyond the wall)
__device__ __forceinline__ void near_walls(const MapView &M, float qx, float qy, int cx, int cy, bool inside, NearState &S) {
  const float L = M.leaf, slack = fmaxf(1e-3f * L, 2.5e-7f * (fabsf(qx) + fabsf(qy) + L));
  const float fx = qx - (float)(cx + M.min_bx) * L, fy = qy - (float)(cy + M.min_by) * L;
  float wl = fmaxf(fx - slack, 0.f), wr = fmaxf(L - fx - slack, 0.f);
  float wd = fmaxf(fy - slack, 0.f), wu = fmaxf(L - fy - slack, 0.f);
  if (!inside) { wl = wr = wd = wu = 0.f; }            // clamped query: no pruning
  S.wl = wl; S.wr = wr; S.wd = wd; S.wu = wu;
}

// Phase 1: the home voxel, then which ring-1 voxels can still matter -- left / right voxel of the home row and the rows
// below and above as one range each, every voxel pruned by its box distance against the home voxel's best.
// (Pruning against `best` as it was after the home voxel scans a few more points than pruning range by range; the
// minimum over a larger set of map points is the same.)
__device__ __forceinline__ NearState nearest_home(const MapView &M, float qx, float qy, const NearPrep &P) {
  NearState S;
  const int cx = P.cx, cy = P.cy;
  S.cx = cx; S.cy = cy;
  const I4u h = P.h;
  float best = scan_bucket(M.pts, h.y, h.z, qx, qy, INFINITY);
  near_walls(M, qx, qy, cx, cy, P.inside, S);
  const float wl = S.wl, wr = S.wr, wd = S.wd, wu = S.wu;
  S.best = best;
  const float wmin = fminf(fminf(wl, wr), fminf(wd, wu));
  S.more = wmin * wmin < best;                         // else: no other voxel can hold a closer point
  S.rn[0] = S.rn[1] = S.rn[2] = S.rn[3] = 0;
  S.rs[0] = S.rs[1] = S.rs[2] = S.rs[3] = 0;
  if (!S.more) return S;
  const bool has_l = cx > 0, has_r = cx + 1 < M.div_x;
  S.rs[0] = h.x; S.rn[0] = (has_l && wl * wl < best) ? h.y - h.x : 0;
  S.rs[1] = h.z; S.rn[1] = (has_r && wr * wr < best) ? h.w - h.z : 0;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int yy = cy + (d ? 1 : -1);
    const float by = d ? wu : wd;
    const I4u o = d ? P.up : P.dn;
    const bool row_ok = yy >= 0 && yy < M.div_y && (by * by < best);
    const int sa = (has_l && wl * wl + by * by < best) ? o.x : o.y;
    const int sb = (has_r && wr * wr + by * by < best) ? o.w : o.z;
    S.rs[2 + d] = sa; S.rn[2 + d] = row_ok ? sb - sa : 0;
  }
  return S;
}

// Phase 2, one lane on its own: the (up to four) ranges walked as ONE sequence, kRingWide points in flight: four loops
// one after the other cost four chains of dependent loads.
__device__ __forceinline__ float nearest_ring1_lane(const MapView &M, float qx, float qy, const NearState &S) {
  float best = S.best;
  constexpr int kRingWide = 8;
  const int c1 = S.rn[0], c2 = c1 + S.rn[1], c3 = c2 + S.rn[2], total = c3 + S.rn[3];
  // start of range j minus the number of points before it: index of sequence position t is off[j] + t
  const int o0 = S.rs[0], o1 = S.rs[1] - c1, o2 = S.rs[2] - c2, o3 = S.rs[3] - c3;
  for (int t0 = 0; t0 < total; t0 += kRingWide) {
    float2 v[kRingWide];
#pragma unroll
    for (int u = 0; u < kRingWide; ++u) {
      const int t = min(t0 + u, total - 1);            // past the end: the last point again (harmless for a minimum)
      const int off = t < c1 ? o0 : (t < c2 ? o1 : (t < c3 ? o2 : o3));
      v[u] = gld_f2_at(M.pts, (unsigned)(off + t) * 8u);
    }
#pragma unroll
    for (int u = 0; u < kRingWide; ++u) best = fminf(best, sq_dist(qx, qy, v[u].x, v[u].y));
  }
  return best;
}

// Phase 2 for the 64 queries of a wave together.  Only a few lanes of a wave need ring 1 (9 % of the queries of a
// well matched scan, two dozen points each), but some lane nearly always does, and then the whole wave walked that
// lane's sequence: as much time as the home voxels for a fourteenth of the distance evaluations.  Here the lanes that
// need it put their ranges into LDS and the (query, point) pairs of the whole wave are dealt out to all 64 lanes --
// the same points, the same float32 expression, the minimum taken with an integer atomic (non-negative floats order
// like their bit patterns), so the result is the one nearest_ring1_lane gives.
#ifndef NDT_RING_JOINT_MAX
#define NDT_RING_JOINT_MAX 12
#endif
constexpr int kRingJointMax = NDT_RING_JOINT_MAX;
struct RingLds {                 // per wave
  int start[64];                 // first item of the query with rank k (k-th lane that needs ring 1)
  float qx[64], qy[64];
  int o0[64], o1[64], o2[64], o3[64], c1[64], c2[64], c3[64];
  unsigned res[64];
};
__device__ __forceinline__ float nearest_ring1_wave(const MapView &M, RingLds &R, float qx, float qy, const NearState &S) {
  const int lane = threadIdx.x & 63;
  const int T = S.rn[0] + S.rn[1] + S.rn[2] + S.rn[3];           // 0 for lanes without a query or with nothing to look at
  const bool need = T > 0;
  const unsigned long long mask = __ballot(need);
  if (mask == 0ull) return S.best;                               // (wave-uniform)
  const int K = __popcll(mask);
  // many lanes with work of their own (poorly matched scans): each walks its own sequence, all lanes busy anyway
  if (K > kRingJointMax) return need ? nearest_ring1_lane(M, qx, qy, S) : S.best;
  const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
  const int incl = (int)wave_incl_scan((unsigned)T);             // (DPP moves: no LDS round trips)
  const int W = __builtin_amdgcn_readlane(incl, 63);             // (query, point) pairs of the wave
  if (need) {
    const int c1 = S.rn[0], c2 = c1 + S.rn[1], c3 = c2 + S.rn[2];
    R.start[rank] = incl - T;
    R.qx[rank] = qx; R.qy[rank] = qy;
    R.c1[rank] = c1; R.c2[rank] = c2; R.c3[rank] = c3;
    R.o0[rank] = S.rs[0]; R.o1[rank] = S.rs[1] - c1; R.o2[rank] = S.rs[2] - c2; R.o3[rank] = S.rs[3] - c3;
    R.res[rank] = __float_as_uint(S.best);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int steps = K > 1 ? 32 - __builtin_clz((unsigned)(K - 1)) : 0;      // binary search over the K starts
  constexpr int kWide = 4;                                       // loads of a lane in flight
  for (int i0 = 0; i0 < W; i0 += 64 * kWide) {
    float2 v[kWide]; int q[kWide];
#pragma unroll
    for (int u = 0; u < kWide; ++u) {
      const int i = min(i0 + u * 64 + lane, W - 1);              // past the end: the last pair again
      int lo = 0, hi = K - 1;                                    // largest k with start[k] <= i
      for (int s = 0; s < steps; ++s) {
        const int mid = (lo + hi + 1) >> 1;
        if (R.start[mid] <= i) lo = mid; else hi = mid - 1;
      }
      q[u] = lo;
      const int t = i - R.start[lo];
      const int off = t < R.c1[lo] ? R.o0[lo] : (t < R.c2[lo] ? R.o1[lo] : (t < R.c3[lo] ? R.o2[lo] : R.o3[lo]));
      v[u] = gld_f2_at(M.pts, (unsigned)(off + t) * 8u);
    }
#pragma unroll
    for (int u = 0; u < kWide; ++u)
      atomicMin(&R.res[q[u]], __float_as_uint(sq_dist(R.qx[q[u]], R.qy[q[u]], v[u].x, v[u].y)));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return need ? __uint_as_float(R.res[rank]) : S.best;
}

// Phase 3: whole rings while the best distance exceeds the ring bound (rare: a query farther than a voxel from every
// map point of its 3 x 3 neighbourhood).
__device__ __forceinline__ float nearest_far(const MapView &M, float qx, float qy, const NearState &S, float best, int r0 = 1) {
  if (!S.more) return best;
  const int cx = S.cx, cy = S.cy;
  const float wl = S.wl, wr = S.wr, wd = S.wd, wu = S.wu, L = M.leaf;
  (void)wl; (void)wr;
  const int *__restrict__ ps = M.pt_start;
  const double Ld = (double)L;
  const int rmax = M.div_x > M.div_y ? M.div_x : M.div_y;
  for (int r = r0; r <= rmax; ++r) {                    // (r0: everything within r0 voxels has been seen already)
    const double bound = (double)r * Ld * 0.999;        // unvisited points are farther than r*L
    if ((double)best <= bound * bound) break;
    const int R = r + 1;                                // ring R, pruned by box distances: in a row at
    const int y0 = cy - R, y1 = cy + R, x0 = cx - R, x1 = cx + R;   // distance by only the columns whose
    for (int yy = (y0 < 0 ? 0 : y0); yy <= y1 && yy < M.div_y; ++yy) {   // box is nearer than sqrt(best - by^2)
      const int dyc = yy - cy;
      const float by = dyc < 0 ? wd + (float)(-dyc - 1) * L : (dyc > 0 ? wu + (float)(dyc - 1) * L : 0.f);
      const float rem = best - by * by;
      if (!(rem > 0.f)) continue;
      const int hw = (int)fminf(sqrtf(rem) / L, 1.0e6f) + 1;   // columns farther than hw cannot matter
      const int *__restrict__ row = ps + (size_t)yy * M.div_x;
      if (yy == y0 || yy == y1) {
        int xa = x0 > cx - hw ? x0 : cx - hw, xb = x1 < cx + hw ? x1 : cx + hw;
        xa = xa < 0 ? 0 : xa; xb = xb >= M.div_x ? M.div_x - 1 : xb;
        if (xa <= xb) { const int sa = gld_i(row + xa), sb = gld_i(row + xb + 1); best = scan_bucket(M.pts, sa, sb, qx, qy, best); }
      } else if (R <= hw) {
        I2u a = {0, 0}, b = {0, 0};                     // both voxels' offsets in flight together
        if (x0 >= 0) a = ld_i2u(row + x0);
        if (x1 < M.div_x) b = ld_i2u(row + x1);
        best = scan_bucket(M.pts, a.x, a.y, qx, qy, best);
        best = scan_bucket(M.pts, b.x, b.y, qx, qy, best);
      }
    }
  }
  return best;
}

// Phase 3 from the occupancy tiles (MapView::tiles): the nine words around the home voxel's tile say which of the voxels
// up to eight away hold points at all -- one round of loads instead of two dependent ones per row of every ring -- and only
// those are looked up: row by row from the query's row outwards, in a row from the home column outwards, every voxel
// pruned by its box distance against the best so far (which closes columns, sides and rows as it shrinks).  Voxels
// farther than eight away are left to the ring walk (nearest_far from ring 9).  Same points, same float32 expression: the
// minimum is the one nearest_far finds.
struct FarLds { unsigned row[24][256]; };       // per block of 256 queries: the occupancy bits of the 24 rows around each query
__device__ __forceinline__ float nearest_far_tiles(const MapView &M, FarLds &F, float qx, float qy, const NearState &S, float best) {
  if (!S.more) return best;
  {
    const double bound = (double)M.leaf * 0.999;
    if ((double)best <= bound * bound) return best;
  }
  const int cx = S.cx, cy = S.cy;
  const float L = M.leaf;
  const int tx = cx >> 3, ty = cy >> 3;
  const unsigned long long *t = M.tiles + (size_t)ty * M.tiles_w + tx;          // tile (tx - 1, ty - 1) of the bordered array
  unsigned long long w[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) w[3 * r + k] = gld_u64(t + (size_t)r * M.tiles_w + k);
  // the 24 rows of the 3 x 3 tiles as 24-bit words (bit j = column 8 (tx - 1) + j), cut out once with constant shifts by all
  // lanes together and kept in LDS: the generator below picks its rows by number
  unsigned *const rowbits = &F.row[0][threadIdx.x];
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    const int r = i >> 3, sh = (i & 7) << 3;
    rowbits[i * 256] = ((unsigned)(w[3 * r] >> sh) & 0xffu) | (((unsigned)(w[3 * r + 1] >> sh) & 0xffu) << 8) |
                       (((unsigned)(w[3 * r + 2] >> sh) & 0xffu) << 16);
  }
  const int j0 = 8 + (cx & 7);                                   // the home column in a row's 24 bits
  const unsigned below = (1u << j0) - 1u;
  // Every lane runs a generator over its rows (cy, cy - 1, cy + 1, cy - 2, ...) and, inside a row, over the occupied voxels
  // from the home column outwards; the wave meets once per voxel to be read: finding the next voxel is cheap and differs
  // from lane to lane (the lanes of a wave come from half a dozen home voxels), reading a voxel's points is the expensive
  // part and is what all lanes do together.  (With the read inside the row loop a wave executed 5.7k vector instructions
  // for its 64 queries with 15 lanes active on average.)
  bool dn_open = true, up_open = true;
  int step = -1;                                                 // row number in the sequence: 0 -> k = 0; 2k - 1 -> cy - k; 2k -> cy + k
  unsigned cand = 0;
  int yy = cy; float by = 0.f;
  for (int it = 0; it < 24 * 17; ++it) {                         // (a counted loop: at most every voxel of the 17 rows)
    int cell = -1;
    while (cell < 0) {
      if (cand == 0u) {                                          // next row
        if (++step > 16) break;
        const int k = (step + 1) >> 1, d = (step != 0 && !(step & 1)) ? 1 : 0;
        bool &open = d ? up_open : dn_open;
        by = k == 0 ? 0.f : (d ? S.wu : S.wd) + (float)(k - 1) * L;
        if (!open || !(by * by < best)) { open = false; if (!dn_open && !up_open) { step = 17; break; } continue; }
        yy = d ? cy + k : cy - k;
        cand = rowbits[(yy - ((ty - 1) << 3)) * 256];              // row 0..23 of the 3 x 3 tiles
        if (k <= 1) cand &= ~(7u << (j0 - 1));                   // the 3 x 3 neighbourhood: phases 1 and 2
        continue;
      }
      if (!(by * by < best)) { cand = 0u; continue; }            // the row has closed since it was opened
      const unsigned lo = cand & below, hi = cand >> j0;          // left of the home column / from it on
      const int jl = lo ? 31 - __clz((int)lo) : -64, jr = hi ? j0 + __ffs((int)hi) - 1 : 128;
      const bool left = (j0 - jl) <= (jr - j0);
      const int j = left ? jl : jr, dx = j - j0;
      const float bx = dx < 0 ? S.wl + (float)(-dx - 1) * L : (dx > 0 ? S.wr + (float)(dx - 1) * L : 0.f);
      if (bx * bx + by * by < best) { cand &= ~(1u << j); cell = yy * M.div_x + cx + dx; }
      else cand &= left ? ~below : below;                        // whatever is farther out on this side is farther away
    }
    if (!__ballot(cell >= 0)) break;                             // (wave-uniform) every generator has run out
    if (cell >= 0) {
      const I2u o = ld_i2u(M.pt_start + cell);
      best = scan_bucket(M.pts, o.x, o.y, qx, qy, best);
    }
  }
  return nearest_far(M, qx, qy, S, best, 8);
}

// does nearest_far have anything to do for this query?  (its own entry conditions, for the deferred far phase)
__device__ __forceinline__ bool far_needed(const MapView &M, const NearState &S, float best) {
  const double bound = (double)M.leaf * 0.999;
  return S.more && !((double)best <= bound * bound);
}

__device__ __forceinline__ float nearest_sq(const MapView &M, float qx, float qy) {
  const NearState S = nearest_home(M, qx, qy, nearest_prep(M, qx, qy));
  return nearest_far(M, qx, qy, S, S.more ? nearest_ring1_lane(M, qx, qy, S) : S.best);
}

// ------------------------------------------------------------------------------------------
// a7 for a batch: getFitnessScore of every match (src/PoseEstimator.cpp:43), queued behind the match kernel.
// Inside the match kernel (one 16-wave workgroup per CU, all registers taken) this search ran at the latency of
// its dependent loads -- offsets, bucket, neighbouring buckets: ~110 us per 10k-point scan, a quarter of a match.
// As a kernel of its own it runs at 8 waves per SIMD on the whole chip and is bound by the vector L1 instead.
// Two steps so that the sum keeps ONE fixed order whatever the grid: the squared distance of every point (in the
// order the passes read the scan: the cell-ordered copy, so that the lanes of a wave share buckets), then per match
// the sum in the unit order of the passes (lane -> wave butterfly -> units 0..63).
// ------------------------------------------------------------------------------------------
#ifndef NDT_FIT_OCC
#define NDT_FIT_OCC 6
#endif
// DEFER (launches whose matches share one scan: hypothesis scoring, configs[4]): the queries that need phase 3 are not
// finished here but put on their match's list -- those with a point in hand from the front, those without from the back --
// and fitness_far_kernel finishes them, 64 queries of ONE kind per wave.  With phase 3 inline a wave walks the rings of its
// farthest lane while the others wait: on configs[4]'s seeds the longest lane of a wave needs 31 rounds of dependent loads,
// the average lane 8, and less than half the lanes need the phase at all.
// The mean of a match's distances, and the order it is summed in (round 5; the same for every launch shape and for both
// forms of the fitness kernels): the match's points in chunks of 64 consecutive points of the ordered copy -- chunk c's
// sum is the wave butterfly (wave_sum) over its 64 values, points without a distance adding 0 -- then lane l of ONE wave
// adds the chunks l, l + 64, l + 128 ... one after the other, and the wave butterfly adds the lanes.  Counts are whole
// numbers however they are added.  A chunk's {sum, count} is 16 bytes in the launch's FitPart array: written by the wave
// that searched the chunk (fitness_points_kernel, scans of their own) or summed it (fitness_reduce_kernel), read by the
// wave that closes the match.
struct FitPart { double sum, cnt; };
__device__ __forceinline__ size_t fit_part_of(int b, int n, unsigned long long o0, int shared_scan) {     // first chunk of match b
  return shared_scan ? (size_t)b * (size_t)((n + 63) >> 6) : (size_t)(o0 >> 6) + (size_t)b;
}
// one whole wave: the match's fitness from its chunks
__device__ __forceinline__ void fitness_close_match(const FitPart *__restrict__ part, int n, ndt_result *R) {
  const int lane = threadIdx.x & 63, nch = (n + 63) >> 6;
  double s = 0.0, c = 0.0;
  for (int k = lane; k < nch; k += 4 * 64) {                    // (four loads of each kind in flight; added in order)
    double vs[4], vc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = k + 64 * u < nch;
      vs[u] = in ? gld_d(&part[k + 64 * u].sum) : 0.0;
      vc[u] = in ? gld_d(&part[k + 64 * u].cnt) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (k + 64 * u < nch) { s += vs[u]; c += vc[u]; }
  }
  s = wave_sum(s); c = wave_sum(c);
  if (lane == 0) R->fitness = (n > 0 && c > 0) ? s / c : DBL_MAX;
}

// XCD-aware numbering of the fitness kernels' workgroups (round 5).  The dispatcher deals consecutive workgroups out to the
// eight XCDs in turn, each with an L2 of its own; a (gx, B) grid with the match's blocks along x therefore spread every match
// over all eight L2s: its ordered copy, its ~300 KB of buckets and their offsets were pulled into each of them.  Now the grid
// is one-dimensional and workgroup w works for XCD w & 7 on that XCD's (w >> 3)-th item, items numbered match-major: all gx
// blocks of a match on ONE XCD -- the XCD of the workgroup that owned the match in the match kernel (b & 7) and wrote its
// ordered copy through that L2 -- one match after the other.
__device__ __forceinline__ bool fit_block_of(int gx, int B, int &b, int &x) {
  const unsigned w = blockIdx.x, xcd = w & 7u, seq = w >> 3;
  b = (int)((seq / (unsigned)gx) * 8u + xcd);
  x = (int)(seq % (unsigned)gx);
  return b < B;
}

template <bool SSE, bool DEFER>
__global__ void __launch_bounds__(256, NDT_FIT_OCC)
fitness_points_kernel(MapView M, const float *__restrict__ scans, const unsigned long long *__restrict__ offsets, int B,
                      int shared_scan, const float2 *__restrict__ sorted, const ndt_result *__restrict__ results,
                      float *__restrict__ fit, unsigned *__restrict__ far_idx, unsigned *__restrict__ far_n, int gx,
                      FitPart *__restrict__ parts) {
  __shared__ RingLds ring[256 / 64];
  int b, bx;
  if (!fit_block_of(gx, B, b, bx)) return;
  {
    const unsigned long long o0 = shared_scan ? offsets[0] : offsets[b];
    const unsigned long long o1 = shared_scan ? offsets[1] : offsets[b + 1];
    const int n = (int)(o1 - o0);
    const ndt_result *R = results + b;
    const Tf32 T = {R->T00, R->T10, R->T03, R->T13};
    const bool use_sorted = sorted != nullptr && !(R->flags & NDT_FLAG_UNSORTED);
    const size_t slot = shared_scan ? (size_t)b * (size_t)n : (size_t)o0;
    const float2 *pts = use_sorted ? sorted + slot : reinterpret_cast<const float2 *>(scans) + o0;
    float *out = DEFER ? fit + slot : nullptr;
    FitPart *part = DEFER ? nullptr : parts + fit_part_of(b, n, o0, shared_scan);
    // whole waves stay together (the ring-1 phase is a wave's joint work): lanes past the end carry no query
    for (int i0 = bx * (int)blockDim.x + (int)(threadIdx.x & ~63u); i0 < n; i0 += gx * (int)blockDim.x) {
      const int i = i0 + (int)(threadIdx.x & 63u);
      const float2 pt = pts[min(i, n - 1)];
      float qx, qy;
      tf_apply_t<SSE>(T, pt.x, pt.y, qx, qy);
      const bool live = i < n && finite2(qx, qy);
      if (!live) { qx = 0.f; qy = 0.f; }                         // (any address inside the grid; the result is dropped)
      NearState S = nearest_home(M, qx, qy, nearest_prep(M, qx, qy));
      if (!live) { S.more = false; S.rn[0] = S.rn[1] = S.rn[2] = S.rn[3] = 0; }
      float best = nearest_ring1_wave(M, ring[threadIdx.x >> 6], qx, qy, S);
      if (DEFER) {
        const bool need = live && far_needed(M, S, best), blind = need && !(best < INFINITY);
        const unsigned long long ma = __ballot(need && !blind), mb = __ballot(blind);
        if (ma | mb) {                                           // (wave-uniform) one atomic per wave and kind
          const int lane = threadIdx.x & 63;
          unsigned base_a = 0, base_b = 0;
          if (lane == 0) {
            if (ma) base_a = atomicAdd(far_n + 2 * (size_t)b, (unsigned)__popcll(ma));
            if (mb) base_b = atomicAdd(far_n + 2 * (size_t)b + 1, (unsigned)__popcll(mb));
          }
          base_a = __builtin_amdgcn_readfirstlane(base_a); base_b = __builtin_amdgcn_readfirstlane(base_b);
          const unsigned long long below = (1ull << lane) - 1ull;
          if (need && !blind) far_idx[slot + base_a + (unsigned)__popcll(ma & below)] = (unsigned)i;
          if (blind) far_idx[slot + (size_t)(n - 1) - (base_b + (unsigned)__popcll(mb & below))] = (unsigned)i;
        }
      } else {
        best = nearest_far(M, qx, qy, S, best);
      }
      if (DEFER) {
        if (i < n) out[i] = live ? best : INFINITY;
      } else {
        // the distances of this chunk of 64 points never leave the wave: their sum and their number do (i0 is a multiple of 64)
        const bool in = live && best < INFINITY;
        const double t = wave_sum(in ? (double)best : 0.0);
        const int c = __popcll(__ballot(in));
        if ((threadIdx.x & 63u) == 0u) part[i0 >> 6] = FitPart{t, (double)c};
      }
    }
  }
}

// Phase 3 of the queries fitness_points_kernel<.., true> has listed: per match the queries with a point in hand, then
// the ones without (far_n[2b], far_n[2b + 1] of them, from the front / the back of the match's slice of far_idx).
#ifndef NDT_FAR_OCC
#define NDT_FAR_OCC 5
#endif
template <bool SSE>
__global__ void __launch_bounds__(256, NDT_FAR_OCC)
fitness_far_kernel(MapView M, const float *__restrict__ scans, const unsigned long long *__restrict__ offsets, int B,
                   int shared_scan, const float2 *__restrict__ sorted, const ndt_result *__restrict__ results,
                   float *__restrict__ fit, const unsigned *__restrict__ far_idx, const unsigned *__restrict__ far_n, int gx) {
  __shared__ FarLds far_lds;
  int b, bx;
  if (!fit_block_of(gx, B, b, bx)) return;
  {
    const unsigned na = far_n[2 * (size_t)b], nb = far_n[2 * (size_t)b + 1];
    if (na + nb == 0u) return;
    const unsigned long long o0 = shared_scan ? offsets[0] : offsets[b];
    const unsigned long long o1 = shared_scan ? offsets[1] : offsets[b + 1];
    const int n = (int)(o1 - o0);
    const ndt_result *R = results + b;
    const Tf32 T = {R->T00, R->T10, R->T03, R->T13};
    const bool use_sorted = sorted != nullptr && !(R->flags & NDT_FLAG_UNSORTED);
    const size_t slot = shared_scan ? (size_t)b * (size_t)n : (size_t)o0;
    const float2 *pts = use_sorted ? sorted + slot : reinterpret_cast<const float2 *>(scans) + o0;
    float *out = fit + slot;
    const unsigned *list = far_idx + slot;
    // whole waves of one kind: the front list rounded up to waves, then the back list
    const unsigned wa = (na + 63u) & ~63u, total = wa + nb;
    for (unsigned e = (unsigned)bx * blockDim.x + threadIdx.x; e < total; e += (unsigned)gx * blockDim.x) {
      unsigned i;
      if (e < wa) { if (e >= na) continue; i = list[e]; }
      else i = list[(unsigned)(n - 1) - (e - wa)];
      const float2 pt = pts[i];
      float qx, qy;
      tf_apply_t<SSE>(T, pt.x, pt.y, qx, qy);
      NearState S;
      const int cx0 = (int)floorf(qx * M.inv_leaf) - M.min_bx, cy0 = (int)floorf(qy * M.inv_leaf) - M.min_by;
      S.cx = cx0 < 0 ? 0 : (cx0 >= M.div_x ? M.div_x - 1 : cx0);
      S.cy = cy0 < 0 ? 0 : (cy0 >= M.div_y ? M.div_y - 1 : cy0);
      near_walls(M, qx, qy, S.cx, S.cy, (S.cx == cx0) && (S.cy == cy0), S);
      S.more = true;
      out[i] = nearest_far_tiles(M, far_lds, qx, qy, S, out[i]);
    }
  }
}

// The last kernel of a launch.  SUM (DEFER launches: the distances are complete only behind fitness_far_kernel): a
// workgroup per match sums them into the match's chunks, and its first wave closes the match.  !SUM: the chunks are there
// (fitness_points_kernel), a wave per match closes it.
constexpr int kFitBlock = 1024;
template <bool SUM>
__global__ void __launch_bounds__(kFitBlock)
fitness_reduce_kernel(const unsigned long long *__restrict__ offsets, int B, int shared_scan,
                      const float *__restrict__ fit, ndt_result *__restrict__ results, FitPart *parts,
                      uint4 *__restrict__ ws_words, unsigned n_ws_words) {
  // it also clears the match kernel's control words and epoch-tagged words for the next launch (one kernel fewer between
  // two launches than a memset in front of each)
  for (unsigned i = blockIdx.x * kFitBlock + threadIdx.x; i < n_ws_words; i += gridDim.x * kFitBlock) ws_words[i] = uint4{0u, 0u, 0u, 0u};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (!SUM) {
    for (int b = blockIdx.x * (kFitBlock / 64) + wave; b < B; b += gridDim.x * (kFitBlock / 64)) {
      const unsigned long long o0 = shared_scan ? offsets[0] : offsets[b];
      const unsigned long long o1 = shared_scan ? offsets[1] : offsets[b + 1];
      const int n = (int)(o1 - o0);
      fitness_close_match(parts + fit_part_of(b, n, o0, shared_scan), n, results + b);
    }
    return;
  }
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const unsigned long long o0 = shared_scan ? offsets[0] : offsets[b];
    const unsigned long long o1 = shared_scan ? offsets[1] : offsets[b + 1];
    const int n = (int)(o1 - o0), nch = (n + 63) >> 6;
    const float *f = fit + (shared_scan ? (size_t)b * (size_t)n : (size_t)o0);
    FitPart *part = parts + fit_part_of(b, n, o0, shared_scan);
    for (int c0 = wave; c0 < nch; c0 += 4 * (kFitBlock / 64)) {   // (four chunks of the wave in flight)
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = (c0 + u * (kFitBlock / 64)) * 64 + lane;
        v[u] = i < n ? f[i] : INFINITY;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * (kFitBlock / 64);
        if (c < nch) {                                           // (wave-uniform)
          const bool in = v[u] < INFINITY;
          const double t = wave_sum(in ? (double)v[u] : 0.0);
          const int k = __popcll(__ballot(in));
          if (lane == 0) part[c] = FitPart{t, (double)k};
        }
      }
    }
    __syncthreads();                                             // (the chunks of this workgroup's waves: stored, then read by wave 0)
    if (wave == 0) fitness_close_match(part, n, results + b);
  }
}
