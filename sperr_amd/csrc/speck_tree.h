// speck_tree.h -- geometry of SPECK3D's set-partitioning forest, shared by host and device code.
//
// The reference keeps explicit lists of `Set3D` boxes and splits them recursively
// (/root/reference/src/SPECK3D_INT.cpp:22-97,140-326).  Here the same forest is described
// implicitly so that any node can be addressed, enumerated and walked in O(1) by a GPU thread:
//
//   * a ROOT is one initial LIS entry (a wavelet subband box);
//   * below a root every split is the reference's XYZ split: each axis of length `len` is cut
//     into (len - len/2, len/2).  Repeating that rule `e` times cuts an axis of length L into
//     2^e intervals (some may be empty) where interval i has length
//         (L >> e) + (bitrev_e(i) < (L mod 2^e))
//     -- the ceil-first rule hands the remainder out in bit-reversed order;
//   * a NODE is (root, depth d, ix, iy, iz); on axis a the effective depth is min(d, D_a) with
//     D_a = ceil(log2 L_a), so saturated axes stop splitting exactly as the reference's
//     (len, 0) splits do;
//   * nodes with more than one sample are SETS, nodes with exactly one are PIXELS, empty nodes
//     do not exist for the coder.
//
// All functions here are pure integer arithmetic; the CPU model in tests/model and the HIP
// kernels in speck_kernels.hip call the very same code.
#ifndef SPERR_AMD_SPECK_TREE_H
#define SPERR_AMD_SPECK_TREE_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SPK_HD __host__ __device__ __forceinline__
#else
#define SPK_HD inline
#endif

namespace spk {

constexpr int kMaxRoots = 96;     // 7 per dyadic level (<=6) + packet extras, generous
constexpr int kMaxDepth = 17;     // chunk dims are u16 in the container header
constexpr int kMaxLevels = 64;    // LIS levels: 1 + sum of per-axis partitions (<= 49)
constexpr int kNodeBlock = 256;   // node ranges of a grid are padded to this many nodes

struct Root {
  uint16_t org[3];
  uint16_t len[3];
  uint8_t D[3];        // per-axis depth at which every interval is <= 1 long
  uint8_t Dmax;        // number of set depths of this root (pixels live at depth Dmax)
  uint16_t lev;        // LIS level of the root set itself
  uint16_t gridFirst;  // index of this root's depth-0 grid in Tree::grids
  uint32_t tabOff[3];  // offset of the axis' interval-start tables in Tree::tab
};

// Grid::kind bit: the root has power-of-two extents and all three axes still split at this depth,
// so every node has exactly 8 non-empty children, child j = (2ix + (j & 1), 2iy + ((j >> 1) & 1),
// 2iz + (j >> 2)) in the reference's order, and they are all pixels (deepest depth) or all sets.
// (Also requires an even x origin and an even row length: kernels move pixel pairs.)
constexpr uint8_t kGridOct = 1;
// Grid::kind bit (deepest oct grids only): a row of the root is whole 64-sample raster mask words,
// so the decoder keeps the pixel results of these leaf sets per leaf (DecBuffers::leafState) and
// folds them into the raster masks word by word instead of scattering them with atomics.
constexpr uint8_t kGridLeafWord = 2;
constexpr uint32_t kTree2D = 1;
constexpr uint32_t kTreeAllOct = 2;   // every grid is an octree (kGridOct): all of a power-of-two chunk's -- the encoder's
                                      //   node kernels then run without their any-shape path (build_tree sets it)

struct Grid {          // all nodes of one root at one depth, as a dense 3D array
  uint32_t nodeOff;    // first flat node id (multiple of kNodeBlock)
  uint16_t root;
  uint8_t depth;
  uint8_t e[3];        // log2 of the grid extent per axis = min(depth, D[a])
  uint8_t kind;        // kGridOct or 0
};

// The CODE of a set (the bits its split takes, /root/reference/src/SPECK3D_INT.cpp:140-212) depends
// on the set's extents only: every axis longer than one sample is cut into (len - len/2, len/2),
// the children come x fastest, and so on down to single samples.  Sets with the same structure
// share a SHAPE CLASS; the decoder of chunks whose lists mix shapes (k_lis_mx) keys its
// speculative tables by class instead of by list level.  Classes are numbered children first.
constexpr uint8_t kClsPixel = 0xff;   // "class" of a single sample
constexpr int kMaxCls = 254;
struct ShapeCls {
  uint8_t nk;              // children, 1..8 (empty halves do not exist)
  uint8_t h;               // 0: every child is a single sample; else 1 + the largest h of a child
  uint8_t slot;            // table column of the class (1..11), 0xff: no table (such sets are walked into)
  uint8_t nsplit;          // axes longer than one sample: the children's list level is the set's + nsplit
  uint8_t kid[8];          // class of child k in the reference's order
  uint32_t maxT;           // upper bound of the bits of a split
};

// Everything a kernel needs to know about one chunk shape.  Arrays live in device memory (or
// host memory for the CPU model); the struct itself is passed by value.
struct Tree {
  uint32_t dims[3];
  uint32_t nvals;          // dims[0]*dims[1]*dims[2]
  uint32_t nroots, ngrids;
  uint32_t nnodes;         // padded total of flat node ids
  uint32_t nlevels;        // number of LIS levels
  uint32_t maxDepth;       // max over roots of Dmax
  const Root* roots;
  const Grid* grids;
  const uint16_t* tab;     // interval start tables
  const uint16_t* blockGrid;  // grid index of every kNodeBlock-sized block of flat node ids
  // kTree2D: the forest of the 2D coder (SPECK2D_INT.cpp): the children of a set come in the reverse
  // order (bottom right first), a child's list level is its parent's + 1 whatever splits
  uint32_t flags;
  // shape classes (ncls == 0: more than kMaxCls, not available)
  uint32_t ncls, nslots;
  const ShapeCls* cls;
  const uint8_t* gridCls;  // [grid][8]: class of a node by which of its three intervals are the long ones
};

struct Node {
  uint16_t grid;           // index into Tree::grids
  uint16_t i[3];
};

// A LIS level is "regular" when every set that can sit in it has the same shape and that shape
// has power-of-two extents.  Then the code of an entry only depends on its class: a class-j set
// has arity[j] children of class j-1 (class 0: the children are pixels) living on LIS level
// lev[j-1]; the level's own entries are class K-1.
constexpr int kMaxClasses = 17;
struct LevelClass {
  uint8_t regular;
  uint8_t K;
  uint8_t arity[kMaxClasses];
  uint8_t lev[kMaxClasses];
};

SPK_HD uint32_t bitrev(uint32_t v, int e)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return e ? (__brev(v) >> (32 - e)) : 0u;
#else
  uint32_t r = 0;
  for (int k = 0; k < e; k++)
    r |= ((v >> k) & 1u) << (e - 1 - k);
  return r;
#endif
}

// length of interval i of an axis of length L cut e times
SPK_HD uint32_t axis_len(uint32_t L, int e, uint32_t i)
{
  return (L >> e) + (bitrev(i, e) < (L & ((1u << e) - 1u)) ? 1u : 0u);
}

// start tables: for e = 0..D the 2^e + 1 boundaries, concatenated
SPK_HD uint32_t tab_index(int e, uint32_t i)
{
  return ((1u << e) - 1u) + (uint32_t)e + i;
}

SPK_HD uint64_t pack_node(const Node& n)
{
  return ((uint64_t)n.grid << 48) | ((uint64_t)n.i[2] << 32) | ((uint64_t)n.i[1] << 16) | n.i[0];
}
SPK_HD Node unpack_node(uint64_t v)
{
  Node n;
  n.grid = (uint16_t)(v >> 48);
  n.i[2] = (uint16_t)(v >> 32);
  n.i[1] = (uint16_t)(v >> 16);
  n.i[0] = (uint16_t)v;
  return n;
}

SPK_HD uint32_t flat_id(const Tree& t, const Node& n)
{
  const Grid& g = t.grids[n.grid];
  return g.nodeOff + ((((uint32_t)n.i[2] << g.e[1]) + n.i[1]) << g.e[0]) + n.i[0];
}

// inverse of flat_id for ids inside real grid extents; returns false for padding ids
SPK_HD bool node_from_flat(const Tree& t, uint32_t id, Node& n)
{
  const uint32_t gi = t.blockGrid[id / kNodeBlock];
  const Grid& g = t.grids[gi];
  const uint32_t local = id - g.nodeOff;
  if (local >= (1u << (g.e[0] + g.e[1] + g.e[2])))
    return false;
  n.grid = (uint16_t)gi;
  n.i[0] = (uint16_t)(local & ((1u << g.e[0]) - 1u));
  n.i[1] = (uint16_t)((local >> g.e[0]) & ((1u << g.e[1]) - 1u));
  n.i[2] = (uint16_t)(local >> (g.e[0] + g.e[1]));
  return true;
}

struct NodeGeom {
  uint32_t len[3];
  uint32_t count;  // len[0]*len[1]*len[2]
};

SPK_HD NodeGeom node_geom(const Tree& t, const Node& n)
{
  const Grid& g = t.grids[n.grid];
  const Root& r = t.roots[g.root];
  NodeGeom q;
  for (int a = 0; a < 3; a++)
    q.len[a] = axis_len(r.len[a], g.e[a], n.i[a]);
  q.count = q.len[0] * q.len[1] * q.len[2];
  return q;
}

// raster index (x fastest) of the single sample of a pixel node given as (root, per-axis
// effective depth, per-axis index)
SPK_HD uint32_t pixel_raster(const Tree& t, const Root& r, const int e[3], const uint32_t i[3])
{
  uint32_t c[3];
  for (int a = 0; a < 3; a++) {
    const uint32_t L = r.len[a];
    if ((L & (L - 1u)) == 0)   // power of two: every interval is L >> e long, no table needed
      c[a] = (uint32_t)r.org[a] + i[a] * (L >> e[a]);
    else
      c[a] = (uint32_t)r.org[a] + t.tab[r.tabOff[a] + tab_index(e[a], i[a])];
  }
  return (c[2] * t.dims[1] + c[1]) * t.dims[0] + c[0];
}

// LIS level of a set's children = level(set) + number of axes of the set that split, where
// level(set) = root level + the number of splitting ancestors per axis
// (/root/reference/src/SPECK3D_INT.cpp:226-229).
SPK_HD uint32_t node_level(const Tree& t, const Node& n)
{
  const Grid& g = t.grids[n.grid];
  const Root& r = t.roots[g.root];
  if (t.flags & kTree2D)   // one level per partition step (/root/reference/src/SPECK2D_INT.cpp:121-147)
    return (uint32_t)r.lev + g.depth;
  uint32_t lev = r.lev;
  for (int a = 0; a < 3; a++) {
    const int Da = r.D[a];
    const int d = g.depth;
    if (Da == 0)
      continue;
    if (d < Da) {
      lev += (uint32_t)d;  // every ancestor at depth < d <= Da-1 is >= 2 long
    }
    else {
      lev += (uint32_t)(Da - 1);
      // the ancestor at depth Da-1 splits iff it is 2 long
      const uint32_t anc = (uint32_t)n.i[a] >> 1;  // effective depth is Da here
      if (axis_len(r.len[a], Da - 1, anc) >= 2)
        lev += 1;
    }
  }
  return lev;
}

// ------------------------------------------------------------------------------------------
// children of a set node, in the reference's order (x fastest, empty ones dropped)
// ------------------------------------------------------------------------------------------
struct Kids {
  int n;                 // number of non-empty children
  int childDepth;        // depth of the children
  bool deepest;          // children live at depth Dmax: they are pixels (or empty)
  uint16_t childGrid;    // grid of the children when !deepest
  uint32_t idx[8][3];    // per-axis child index at the child's effective depth
  uint32_t count[8];     // number of samples of the child
  int e[3];              // children's effective per-axis depth
};

SPK_HD void node_kids(const Tree& t, const Node& n, Kids& k)
{
  const Grid& g = t.grids[n.grid];
  const Root& r = t.roots[g.root];
  k.childDepth = g.depth + 1;
  k.deepest = (k.childDepth == r.Dmax);
  k.childGrid = (uint16_t)(n.grid + 1);
  int nsplit[3];
  uint32_t base[3];
  for (int a = 0; a < 3; a++) {
    const bool splits = g.depth < r.D[a];
    k.e[a] = splits ? g.e[a] + 1 : g.e[a];
    nsplit[a] = splits ? 2 : 1;
    base[a] = splits ? (uint32_t)n.i[a] * 2u : (uint32_t)n.i[a];
  }
  k.n = 0;
  for (int cz = 0; cz < nsplit[2]; cz++)
    for (int cy = 0; cy < nsplit[1]; cy++)
      for (int cx = 0; cx < nsplit[0]; cx++) {
        const uint32_t ci[3] = {base[0] + cx, base[1] + cy, base[2] + cz};
        uint32_t cnt = 1;
        for (int a = 0; a < 3; a++)
          cnt *= axis_len(r.len[a], k.e[a], ci[a]);
        if (cnt == 0)
          continue;
        k.idx[k.n][0] = ci[0];
        k.idx[k.n][1] = ci[1];
        k.idx[k.n][2] = ci[2];
        k.count[k.n] = cnt;
        k.n++;
      }
  if (t.flags & kTree2D)   // the 2D coder's order: from the bottom right backwards (SPECK2D_INT.cpp:109-147)
    for (int lo = 0, hi = k.n - 1; lo < hi; lo++, hi--) {
      for (int a = 0; a < 3; a++) {
        const uint32_t v = k.idx[lo][a];
        k.idx[lo][a] = k.idx[hi][a];
        k.idx[hi][a] = v;
      }
      const uint32_t cnt = k.count[lo];
      k.count[lo] = k.count[hi];
      k.count[hi] = cnt;
    }
}

// LIS level of the children of a set that are sets themselves (q = node_geom of the set)
SPK_HD uint32_t kid_level(const Tree& t, const Node& n, const NodeGeom& q)
{
  if (t.flags & kTree2D)
    return node_level(t, n) + 1u;
  return node_level(t, n) + (q.len[0] > 1) + (q.len[1] > 1) + (q.len[2] > 1);
}

SPK_HD uint32_t kid_flat(const Tree& t, const Kids& k, int j)
{
  const Grid& g = t.grids[k.childGrid];
  return g.nodeOff + (((k.idx[j][2] << g.e[1]) + k.idx[j][1]) << g.e[0]) + k.idx[j][0];
}

SPK_HD Node kid_node(const Kids& k, int j)
{
  Node c;
  c.grid = k.childGrid;
  c.i[0] = (uint16_t)k.idx[j][0];
  c.i[1] = (uint16_t)k.idx[j][1];
  c.i[2] = (uint16_t)k.idx[j][2];
  return c;
}

SPK_HD uint32_t kid_raster(const Tree& t, const Node& parent, const Kids& k, int j)
{
  const Root& r = t.roots[t.grids[parent.grid].root];
  return pixel_raster(t, r, k.e, k.idx[j]);
}

// parent of a non-root node
SPK_HD Node node_parent(const Tree& t, const Node& n)
{
  const Grid& g = t.grids[n.grid];
  const Root& r = t.roots[g.root];
  Node p;
  p.grid = (uint16_t)(n.grid - 1);
  for (int a = 0; a < 3; a++)
    p.i[a] = (g.depth - 1 < r.D[a]) ? (uint16_t)(n.i[a] >> 1) : n.i[a];
  return p;
}

SPK_HD bool node_is_root(const Tree& t, const Node& n)
{
  return t.grids[n.grid].depth == 0;
}

// ------------------------------------------------------------------------------------------
// Shape classes: class of a node, and its children by ordinal in O(1)
// ------------------------------------------------------------------------------------------
SPK_HD uint32_t node_cls(const Tree& t, const Node& n)
{
  const Grid& g = t.grids[n.grid];
  const Root& r = t.roots[g.root];
  uint32_t k = 0;
  for (int a = 0; a < 3; a++) {
    const int e = g.e[a];
    const uint32_t rem = (uint32_t)r.len[a] & ((1u << e) - 1u);
    k |= (bitrev(n.i[a], e) < rem ? 1u : 0u) << a;
  }
  return t.gridCls[(uint32_t)n.grid * 8u + k];
}

// The non-empty children of a set are a box of n[0] x n[1] x n[2] (each 1 or 2) child intervals:
// child k (x fastest) has index base[a] + its selector on every axis.
struct KidBox {
  uint32_t base[3];
  uint32_t n[3];
  int e[3];              // the children's effective depth per axis
  uint32_t nk;           // n[0] * n[1] * n[2]
  uint32_t kidlev;       // LIS level of the children that are sets
  uint16_t grid;         // their grid (meaningless when they are single samples of the deepest depth)
  uint16_t rev;          // 2D coder: child ordinals count from the far corner
};

SPK_HD void kid_box(const Tree& t, const Node& nd, KidBox& k)
{
  const Grid& g = t.grids[nd.grid];
  const Root& r = t.roots[g.root];
  k.grid = (uint16_t)(nd.grid + 1);
  k.kidlev = node_level(t, nd);
  k.rev = (uint16_t)(t.flags & kTree2D);
  for (int a = 0; a < 3; a++) {
    const bool splits = g.depth < r.D[a];
    k.e[a] = splits ? g.e[a] + 1 : g.e[a];
    k.base[a] = splits ? (uint32_t)nd.i[a] * 2u : (uint32_t)nd.i[a];
    k.n[a] = (splits && axis_len(r.len[a], k.e[a], k.base[a] + 1u) > 0) ? 2u : 1u;
    if (!k.rev)
      k.kidlev += k.n[a] - 1u;   // (an interval longer than one sample has two non-empty halves)
  }
  if (k.rev)
    k.kidlev += 1u;
  k.nk = k.n[0] * k.n[1] * k.n[2];
}

SPK_HD void kid_index(const KidBox& k, uint32_t ord, uint32_t idx[3])
{
  if (k.rev)
    ord = k.nk - 1u - ord;
  idx[0] = k.base[0] + (k.n[0] == 2 ? (ord & 1u) : 0u);
  const uint32_t o1 = k.n[0] == 2 ? ord >> 1 : ord;
  idx[1] = k.base[1] + (k.n[1] == 2 ? (o1 & 1u) : 0u);
  const uint32_t o2 = k.n[1] == 2 ? o1 >> 1 : o1;
  idx[2] = k.base[2] + o2;
}

SPK_HD uint64_t kid_packed(const KidBox& k, uint32_t ord)
{
  uint32_t idx[3];
  kid_index(k, ord, idx);
  return ((uint64_t)k.grid << 48) | ((uint64_t)idx[2] << 32) | ((uint64_t)idx[1] << 16) | idx[0];
}

SPK_HD uint32_t kid_pixel_raster(const Tree& t, const Node& parent, const KidBox& k, uint32_t ord)
{
  uint32_t idx[3];
  kid_index(k, ord, idx);
  return pixel_raster(t, t.roots[t.grids[parent.grid].root], k.e, idx);
}

// ------------------------------------------------------------------------------------------
// Encoder-side per-node logic.  `M[id]` = msb of the largest coefficient in the node (-1 when
// all zero), `E[id]` = number of bits the node's split emits (children tests, signs and nested
// splits) -- both indexed by flat node id; pixel children at the deepest depth read `msb[]`
// in raster order instead.
// ------------------------------------------------------------------------------------------
struct KidInfo {
  int8_t m[8];         // msb of each child
  uint32_t e[8];       // split length of each child set (0 for pixels)
  bool pixel[8];
};

SPK_HD void kids_info(const Tree& t, const Node& n, const Kids& k, const int8_t* M,
                      const uint32_t* E, const int8_t* msb, KidInfo& ki)
{
  for (int j = 0; j < k.n; j++) {
    ki.pixel[j] = (k.count[j] == 1);
    if (k.deepest) {
      ki.m[j] = msb[kid_raster(t, n, k, j)];
      ki.e[j] = 0;
    }
    else {
      const uint32_t id = kid_flat(t, k, j);
      ki.m[j] = M[id];
      ki.e[j] = ki.pixel[j] ? 0u : E[id];
    }
  }
}

// Number of bits a set emits when it splits at plane `m` (its own msb): one test bit per
// child except an inferred last child, one sign bit per significant pixel child, plus the
// nested split of every significant set child
// (/root/reference/src/SPECK3D_INT.cpp:140-212, SPECK3D_INT_ENC.cpp:161-199).
SPK_HD uint32_t split_bits(const Kids& k, const KidInfo& ki, int m)
{
  uint32_t bits = 0;
  bool found = false;
  for (int j = 0; j < k.n; j++) {
    const bool coded = found || (j + 1 != k.n);
    bits += coded ? 1u : 0u;
    const bool sig = coded ? (ki.m[j] == m) : true;
    if (sig) {
      found = true;
      bits += ki.pixel[j] ? 1u : ki.e[j];
    }
  }
  return bits;
}

// Offset of child `which`'s own code inside its parent's split (bits emitted for the earlier
// children), and whether `which`'s test bit is coded.
SPK_HD uint32_t kid_offset(const Kids& k, const KidInfo& ki, int m, int which, bool& coded_out)
{
  uint32_t bits = 0;
  bool found = false;
  for (int j = 0; j < which; j++) {
    bits += 1;  // never the last child, so always coded
    if (ki.m[j] == m) {
      found = true;
      bits += ki.pixel[j] ? 1u : ki.e[j];
    }
  }
  coded_out = found || (which + 1 != k.n);
  return bits;
}

}  // namespace spk

#endif
