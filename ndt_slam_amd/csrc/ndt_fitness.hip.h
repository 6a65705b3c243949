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
struct __attribute__((packed, aligned(4))) I4u { int x, y, z, w; };
struct __attribute__((packed, aligned(4))) I2u { int x, y; };
__device__ __forceinline__ I4u ld_i4u(const int *p) { I4u v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ I2u ld_i2u(const int *p) { I2u v; __builtin_memcpy(&v, p, 8); return v; }

__device__ __forceinline__ float sq_dist(float qx, float qy, float px, float py) {
  const float ex = qx - px, ey = qy - py;
  return ex * ex + ey * ey;
}

// min over the points pts[s .. se) of the float32 squared distance to (qx, qy)
__device__ __forceinline__ float scan_bucket(const float2 *__restrict__ pts, int s, int se, float qx,
                                             float qy, float best) {
  if (s >= se) return best;
  if (s & 1) { const float2 p = pts[s]; best = fminf(best, sq_dist(qx, qy, p.x, p.y)); ++s; }
  const float4 *__restrict__ p4 = reinterpret_cast<const float4 *>(pts + s);   // 16-byte aligned
  const int npair = (se - s) >> 1;
  int i = 0;
  for (; i + 2 <= npair; i += 2) {           // four points, two loads in flight
    const float4 a = p4[i], b = p4[i + 1];
    const float d0 = sq_dist(qx, qy, a.x, a.y), d1 = sq_dist(qx, qy, a.z, a.w);
    const float d2 = sq_dist(qx, qy, b.x, b.y), d3 = sq_dist(qx, qy, b.z, b.w);
    best = fminf(best, fminf(fminf(d0, d1), fminf(d2, d3)));
  }
  if (i < npair) {
    const float4 a = p4[i];
    best = fminf(best, fminf(sq_dist(qx, qy, a.x, a.y), sq_dist(qx, qy, a.z, a.w)));
  }
  if ((se - s) & 1) { const float2 p = pts[se - 1]; best = fminf(best, sq_dist(qx, qy, p.x, p.y)); }
  return best;
}

__device__ __forceinline__ float nearest_sq(const MapView &M, float qx, float qy) {
  const int cx0 = (int)floorf(qx * M.inv_leaf) - M.min_bx, cy0 = (int)floorf(qy * M.inv_leaf) - M.min_by;
  const int cx = cx0 < 0 ? 0 : (cx0 >= M.div_x ? M.div_x - 1 : cx0);
  const int cy = cy0 < 0 ? 0 : (cy0 >= M.div_y ? M.div_y - 1 : cy0);
  const bool inside = (cx == cx0) && (cy == cy0);
  const int *__restrict__ ps = M.pt_start;
  const size_t gh = (size_t)cy * M.div_x + cx;
  // offsets of (cx-1, cx, cx+1) of the home row in one load: [left, home) [home, right) [right, end)
  const I4u h = ld_i4u(ps + gh - 1);
  float best = scan_bucket(M.pts, h.y, h.z, qx, qy, INFINITY);
  // distances from the query to the four walls of its voxel, shrunk by 1e-3 leaf so that a point
  // the float32 voxel rounding put on the other side of a wall is never pruned away
  const float L = M.leaf, slack = 1e-3f * L;
  const float fx = qx - (float)(cx + M.min_bx) * L, fy = qy - (float)(cy + M.min_by) * L;
  float wl = fmaxf(fx - slack, 0.f), wr = fmaxf(L - fx - slack, 0.f);
  float wd = fmaxf(fy - slack, 0.f), wu = fmaxf(L - fy - slack, 0.f);
  if (!inside) { wl = wr = wd = wu = 0.f; }            // clamped query: no pruning
  const float wmin = fminf(fminf(wl, wr), fminf(wd, wu));
  if (!(wmin * wmin < best)) return best;              // no other voxel can hold a closer point
  // ring 1: left / right voxel of the home row, then the rows below and above as one range each,
  // every voxel pruned by its box distance
  const bool has_l = cx > 0, has_r = cx + 1 < M.div_x;
  if (has_l && wl * wl < best) best = scan_bucket(M.pts, h.x, h.y, qx, qy, best);
  if (has_r && wr * wr < best) best = scan_bucket(M.pts, h.z, h.w, qx, qy, best);
#pragma unroll
  for (int dy = -1; dy <= 1; dy += 2) {
    const int yy = cy + dy;
    const float by = dy < 0 ? wd : wu;
    if (yy < 0 || yy >= M.div_y || !(by * by < best)) continue;
    const I4u o = ld_i4u(ps + (size_t)yy * M.div_x + cx - 1);
    const int sa = (has_l && wl * wl + by * by < best) ? o.x : o.y;
    const int sb = (has_r && wr * wr + by * by < best) ? o.w : o.z;
    best = scan_bucket(M.pts, sa, sb, qx, qy, best);
  }
  const double Ld = (double)L;
  const int rmax = M.div_x > M.div_y ? M.div_x : M.div_y;
  for (int r = 1; r <= rmax; ++r) {
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
        if (xa <= xb) { const int sa = row[xa], sb = row[xb + 1]; best = scan_bucket(M.pts, sa, sb, qx, qy, best); }
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
