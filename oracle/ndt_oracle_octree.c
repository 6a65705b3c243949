/* CPU oracle, SURVEY.md 8f row f3: the local-map assembly of Submap::makeMap (src/PointCloudMap.cpp:15-39).
 *
 * TEST INFRASTRUCTURE ONLY (see ndt_oracle.h).  PARITY UNPINNED: the change detector is
 * pcl::octree::OctreePointCloudChangeDetector (PCL 1.8-1.10, absent from this image); what follows is a
 * literal restatement of the published algorithm of octree_pointcloud.hpp (adoptBoundingBoxToPoint,
 * getKeyBitSize, genOctreeKeyforPoint, addPointIdx) and octree2buf_base.hpp (createLeafRecursive with the
 * two child-pointer buffers, switchBuffers, serializeTreeRecursive with the new-leaf filter) as
 * PCFilter::difference_extraction drives them (include/ndt_slam/PCFilter.h:58-94): a real pointer octree,
 * so that the lattice formulation the GPU uses is checked against the tree and not against itself.
 * Clouds are z = 0 (PointCloudMap::addPoints, src/PointCloudMap.cpp:71).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "ndt_oracle.h"

typedef struct OcNode {
  struct OcNode *child[2][8];   /* BufferedBranchNode::child_node_array_[2][8] */
  int is_leaf;
  int *idx; int n, cap;         /* OctreeContainerPointIndices */
} OcNode;

typedef struct Octree {
  double res, min[3], max[3];
  int defined, depth, sel;      /* bounding_box_defined_, octree_depth_, buffer_selector_ */
  unsigned depth_mask;
  OcNode *root;
  long leaf_count;
  OcNode **all; size_t n_all, cap_all;   /* every node, for the final free */
} Octree;

static OcNode *oc_new(Octree *t, int leaf) {
  OcNode *n = (OcNode *)calloc(1, sizeof(OcNode));
  n->is_leaf = leaf;
  if (t->n_all == t->cap_all) {
    t->cap_all = t->cap_all ? 2 * t->cap_all : 1024;
    t->all = (OcNode **)realloc(t->all, t->cap_all * sizeof(OcNode *));
  }
  t->all[t->n_all++] = n;
  return n;
}

static void oc_set_depth(Octree *t, int d) {   /* OctreeBase::setTreeDepth */
  t->depth = d;
  t->depth_mask = 1u << (d - 1);
}

/* OctreePointCloud::getKeyBitSize, called once when the first point defines the box */
static void oc_key_bit_size(Octree *t) {
  const float minValue = FLT_EPSILON;
  unsigned mk[3], mv = 2;
  for (int a = 0; a < 3; ++a) {
    mk[a] = (unsigned)ceil((t->max[a] - t->min[a] - minValue) / t->res);
    if (mk[a] > mv) mv = mk[a];
  }
  double lg = log((double)mv) / log(2.0);
  unsigned d = (unsigned)ceil(lg - minValue);
  if (d > 32) d = 32;
  double side = (double)(1u << d) * t->res;
  if (t->leaf_count == 0) {
    for (int a = 0; a < 3; ++a) {
      double over = (side - (t->max[a] - t->min[a])) / 2.0;
      if (over > minValue) { t->min[a] -= over; t->max[a] += over; }
    }
  } else {
    for (int a = 0; a < 3; ++a) t->max[a] = t->min[a] + side;
  }
  oc_set_depth(t, (int)d);
}

/* OctreePointCloud::adoptBoundingBoxToPoint; returns 0 when the tree would need more than 30 levels */
static int oc_adopt(Octree *t, const float p[3]) {
  const float minValue = FLT_EPSILON;
  for (;;) {
    int lo[3], up[3], any = 0;
    for (int a = 0; a < 3; ++a) {
      lo[a] = (double)p[a] < t->min[a];
      up[a] = (double)p[a] >= t->max[a];
      any |= lo[a] | up[a];
    }
    if (!any && t->defined) return 1;
    if (t->defined) {
      if (t->depth >= 30) return 0;
      int ci = ((!up[0]) << 2) | ((!up[1]) << 1) | (!up[2]);
      OcNode *nr = oc_new(t, 0);
      nr->child[t->sel][ci] = t->root;           /* Octree2BufBase::setBranchChildPtr: current buffer only */
      t->root = nr;
      double side = (double)(1 << t->depth) * t->res;
      for (int a = 0; a < 3; ++a) if (!up[a]) t->min[a] -= side;
      oc_set_depth(t, t->depth + 1);
      side = (double)(1 << t->depth) * t->res - minValue;
      for (int a = 0; a < 3; ++a) t->max[a] = t->min[a] + side;
    } else {
      for (int a = 0; a < 3; ++a) {
        t->min[a] = (double)p[a] - t->res / 2;
        t->max[a] = (double)p[a] + t->res / 2;
      }
      oc_key_bit_size(t);
      t->defined = 1;
    }
  }
}

/* Octree2BufBase::createLeafRecursive */
static OcNode *oc_create_leaf(Octree *t, const unsigned key[3], unsigned mask, OcNode *br, int reset) {
  const int s = t->sel;
  if (reset) for (int c = 0; c < 8; ++c) br->child[s][c] = NULL;
  const int ci = ((!!(key[0] & mask)) << 2) | ((!!(key[1] & mask)) << 1) | (!!(key[2] & mask));
  if (mask > 1) {
    OcNode *cb; int do_reset = 0;
    if (!br->child[s][ci]) {
      if (br->child[!s][ci]) {
        OcNode *cn = br->child[!s][ci];
        if (!cn->is_leaf) { cb = cn; br->child[s][ci] = cn; }
        else { br->child[!s][ci] = NULL; cb = oc_new(t, 0); br->child[s][ci] = cb; }
        do_reset = 1;
      } else { cb = oc_new(t, 0); br->child[s][ci] = cb; }
    } else cb = br->child[s][ci];
    return oc_create_leaf(t, key, mask / 2, cb, do_reset);
  }
  if (!br->child[s][ci]) {
    OcNode *lf;
    if (br->child[!s][ci]) {
      OcNode *cn = br->child[!s][ci];
      if (cn->is_leaf) { lf = cn; lf->n = 0; br->child[s][ci] = cn; }     /* container cleared, node shared */
      else { br->child[!s][ci] = NULL; lf = oc_new(t, 1); br->child[s][ci] = lf; }
    } else { lf = oc_new(t, 1); br->child[s][ci] = lf; }
    t->leaf_count++;
    return lf;
  }
  return br->child[s][ci];
}

/* OctreePointCloud::addPointIdx */
static int oc_add_point(Octree *t, const float *xy, int i) {
  const float p[3] = {xy[2 * i], xy[2 * i + 1], 0.0f};
  if (!isfinite(p[0]) || !isfinite(p[1])) return 1;          /* addPointsFromInputCloud: isFinite */
  if (!oc_adopt(t, p)) return 0;
  unsigned key[3];
  for (int a = 0; a < 3; ++a) key[a] = (unsigned)(((double)p[a] - t->min[a]) / t->res);   /* genOctreeKeyforPoint */
  OcNode *lf = oc_create_leaf(t, key, t->depth_mask, t->root, 0);
  if (lf->n == lf->cap) { lf->cap = lf->cap ? 2 * lf->cap : 4; lf->idx = (int *)realloc(lf->idx, lf->cap * sizeof(int)); }
  lf->idx[lf->n++] = i;
  return 1;
}

/* Octree2BufBase::serializeTreeRecursive(new_leafs_filter = true): leaves of the current buffer whose
 * parent has no child at that slot in the previous buffer, depth first in child order. */
static void oc_new_leaves(const Octree *t, const OcNode *br, int *out, size_t *n) {
  const int s = t->sel;
  for (int c = 0; c < 8; ++c) {
    const OcNode *ch = br->child[s][c];
    if (!ch) continue;
    if (!ch->is_leaf) oc_new_leaves(t, ch, out, n);
    else if (!br->child[!s][c]) for (int k = 0; k < ch->n; ++k) out[(*n)++] = ch->idx[k];
  }
}

/* PCFilter::difference_extraction (include/ndt_slam/PCFilter.h:58-94): the indices (into `test`) of the
 * points of `test` whose leaf voxel holds no point of `base`, in the order the change detector returns
 * them.  Returns the count, or (size_t)-1 when the clouds span more than 2^30 voxels. */
size_t ndt_oracle_difference_indices(const float *base_xy, size_t n_base, const float *test_xy, size_t n_test,
                                     double resol, int *out_idx) {
  Octree t;
  memset(&t, 0, sizeof t);
  t.res = resol;
  t.root = oc_new(&t, 0);
  int ok = 1;
  for (size_t i = 0; ok && i < n_base; ++i) ok = oc_add_point(&t, base_xy, (int)i);
  /* switchBuffers */
  t.sel = !t.sel;
  t.leaf_count = 0;
  for (int c = 0; c < 8; ++c) t.root->child[t.sel][c] = NULL;
  for (size_t i = 0; ok && i < n_test; ++i) ok = oc_add_point(&t, test_xy, (int)i);
  size_t n = 0;
  if (ok) oc_new_leaves(&t, t.root, out_idx, &n);
  for (size_t k = 0; k < t.n_all; ++k) { free(t.all[k]->idx); free(t.all[k]); }
  free(t.all);
  return ok ? n : (size_t)-1;
}

size_t ndt_oracle_difference_extraction(const float *base_xy, size_t n_base, const float *test_xy, size_t n_test,
                                        double resol, float *out_xy) {
  int *idx = (int *)malloc((n_test + 1) * sizeof(int));
  size_t n = ndt_oracle_difference_indices(base_xy, n_base, test_xy, n_test, resol, idx);
  if (n != (size_t)-1)
    for (size_t k = 0; k < n; ++k) { out_xy[2 * k] = test_xy[2 * idx[k]]; out_xy[2 * k + 1] = test_xy[2 * idx[k] + 1]; }
  free(idx);
  return n;
}

/* Submap::makeMap (src/PointCloudMap.cpp:15-39).  scans: concatenated float2 points, offsets[n_scans+1].
 * first_submap = (cntS == 0).  out_xy must hold every input point (twice that for a single scan, which is
 * appended as the first and as the newest); returns the count or (size_t)-1. */
size_t ndt_oracle_make_map(const float *scans_xy, const size_t *offsets, int n_scans, int first_submap, int newest,
                           int remove_moving, double resol, double thre_neighbor, float *out_xy) {
  size_t cnt = 0;
#define APPEND(s) do { size_t m_ = offsets[(s) + 1] - offsets[(s)]; \
    memcpy(out_xy + 2 * cnt, scans_xy + 2 * offsets[(s)], m_ * 2 * sizeof(float)); cnt += m_; } while (0)
  if (n_scans <= 0) return 0;
  if (remove_moving) {
    if (first_submap) APPEND(0);
    for (int i = 0; i < n_scans - 2; ++i) {
      const size_t n1 = offsets[i + 1] - offsets[i], n3 = offsets[i + 3] - offsets[i + 2];
      const size_t n2 = offsets[i + 2] - offsets[i + 1];
      float *c13 = (float *)malloc((n1 + n3 + 1) * 2 * sizeof(float));
      memcpy(c13, scans_xy + 2 * offsets[i], n1 * 2 * sizeof(float));               /* *cloud_1and3 += scans[i]   */
      memcpy(c13 + 2 * n1, scans_xy + 2 * offsets[i + 2], n3 * 2 * sizeof(float));  /* *cloud_1and3 += scans[i+2] */
      float *diff = (float *)malloc((n2 + 1) * 2 * sizeof(float));
      size_t nd = ndt_oracle_difference_extraction(c13, n1 + n3, scans_xy + 2 * offsets[i + 1], n2, resol, diff);
      if (nd == (size_t)-1) { free(c13); free(diff); return (size_t)-1; }
      cnt += ndt_oracle_remove_neighbors(scans_xy + 2 * offsets[i + 1], n2, diff, nd, thre_neighbor, out_xy + 2 * cnt);
      free(c13); free(diff);
    }
    if (newest) APPEND(n_scans - 1);
  } else {
    for (int i = first_submap ? 0 : 2; i < n_scans; ++i) APPEND(i);
  }
#undef APPEND
  return cnt;
}
