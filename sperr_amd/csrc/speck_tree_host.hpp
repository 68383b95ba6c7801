// speck_tree_host.hpp -- host-side construction of the spk::Tree tables for one chunk shape.
//
// Follows the reference's list initialisation (/root/reference/src/SPECK3D_INT.cpp:22-97): the
// volume is split `levels` times (XYZ while both the XY and the Z transform counts last, then
// XY-only or Z-only); every split hands its non-first children to the LIS as roots and keeps
// splitting the first child; the box that is left goes to the FRONT of its list.
#ifndef SPERR_AMD_SPECK_TREE_HOST_HPP
#define SPERR_AMD_SPECK_TREE_HOST_HPP

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <map>
#include <vector>

#include "speck_tree.h"

namespace spk {

// geometry helpers restated from /root/reference/src/sperr_helper.cpp:36-68,125-146
inline size_t num_of_xforms(size_t len)
{
  size_t n = 0;
  while (len >= 9) {
    n++;
    len -= len / 2;
  }
  return std::min<size_t>(n, 6);
}
inline bool can_use_dyadic(const std::array<size_t, 3>& d, size_t& levels)
{
  if (d[2] < 2 || d[1] < 2)
    return false;
  const size_t xy = num_of_xforms(std::min(d[0], d[1])), z = num_of_xforms(d[2]);
  if (xy == z || (xy >= 5 && z >= 5)) {
    levels = std::min(xy, z);
    return true;
  }
  return false;
}
inline size_t num_of_partitions(size_t len)
{
  size_t n = 0;
  while (len > 1) {
    n++;
    len -= len / 2;
  }
  return n;
}
inline std::array<size_t, 2> approx_detail_len(size_t len, size_t lev)
{
  size_t lo = len, hi = 0;
  for (size_t i = 0; i < lev; i++) {
    hi = lo / 2;
    lo -= hi;
  }
  return {lo, hi};
}

struct HostTree {
  std::vector<Root> roots;
  std::vector<Grid> grids;
  std::vector<uint16_t> tab;
  std::vector<uint16_t> blockGrid;
  std::vector<std::vector<uint64_t>> initLIS;  // per level, packed root nodes in list order
  std::vector<uint32_t> levelCap;              // per level, number of set nodes (list capacity)
  std::vector<uint32_t> levelOff;              // exclusive prefix of levelCap (+ total at end)
  std::vector<LevelClass> levelClass;          // per level: regular shape chain (or regular = 0)
  bool allRegular = false;                     // every level that can hold sets is regular
  std::vector<ShapeCls> cls;                   // shape classes, children first (empty: too many)
  std::vector<uint8_t> gridCls;                // [grid][8]
  std::vector<uint64_t> clsCount;              // sets of the forest per class
  std::vector<uint8_t> levelGroup;             // per list level: column group of most of its entries
  std::vector<uint8_t> mxSlot;                 // k_lis_mx: column of every class (0xff: none), build_mx_columns
  std::vector<uint8_t> mxLevelGroup;           // k_lis_mx: per list level, dominant column group | highest << 4
  uint32_t nslots = 0;                         // classes that have a table slot
  uint32_t slotMaxT = 0;                       // longest split of a class with a table
  uint32_t dims[3] = {0, 0, 0};
  uint32_t nnodes = 0, nlevels = 0, maxDepth = 0, nsets = 0;
  uint32_t flags = 0;                          // spk::kTree2D
  // 2D coder: the subbands the type-I set releases, three per transform level from the coarsest on
  // (bottom right, top right, bottom left: SPECK2D_INT.cpp:149-186), as packed root nodes (kNoRoot: empty)
  static constexpr uint64_t kNoRoot = ~0ull;
  std::vector<uint64_t> iRoots;
  uint32_t iLevels = 0;                        // transform levels = part_level of the first type-I set

  Tree view() const
  {
    Tree t;
    for (int a = 0; a < 3; a++)
      t.dims[a] = dims[a];
    t.nvals = dims[0] * dims[1] * dims[2];
    t.nroots = (uint32_t)roots.size();
    t.ngrids = (uint32_t)grids.size();
    t.nnodes = nnodes;
    t.nlevels = nlevels;
    t.maxDepth = maxDepth;
    t.roots = roots.data();
    t.grids = grids.data();
    t.tab = tab.data();
    t.blockGrid = blockGrid.data();
    t.flags = flags;
    t.ncls = (uint32_t)cls.size();
    t.nslots = nslots;
    t.cls = cls.data();
    t.gridCls = gridCls.data();
    return t;
  }
};

namespace detail {
struct Box {
  uint32_t org[3], len[3];
};
inline int ceil_log2(uint32_t v)
{
  int e = 0;
  while ((1u << e) < v)
    e++;
  return e;
}
}  // namespace detail

constexpr int kClsTableH = 2;            // default: columns up to two steps above the leaf parents (measured: 1 is 5 % faster on a 250^3 chunk, 25 % slower on the 128^3-sized border chunks of a ragged volume)
constexpr uint32_t kClsTableSlots = 4;
constexpr double kClsMinShare = 0.02;       // a class needs this share of the sets of its h to get a column

// Shape classes of every set of the forest (speck_tree.h, ShapeCls).  hmax >= 0: the leaf parents
// get table slots, hmax >= 1 / 2: up to `maxSlots` (at most 4) of the classes one / two steps up.
inline void build_classes(HostTree& h, int hmax, uint32_t maxSlots, double minShare = kClsMinShare)
{
  h.cls.clear();
  h.gridCls.assign(h.grids.size() * 8, kClsPixel);
  h.nslots = 0;
  h.slotMaxT = 0;
  std::map<std::array<uint32_t, 3>, int> byDims;
  std::map<std::array<uint8_t, 9>, int> byStruct;
  bool overflow = false;
  auto intern = [&](const ShapeCls& c) -> int {
    std::array<uint8_t, 9> key{};
    key[0] = c.nk;
    for (int k = 0; k < 8; k++)
      key[1 + k] = k < c.nk ? c.kid[k] : 0;
    auto it = byStruct.find(key);
    if (it != byStruct.end())
      return it->second;
    if ((int)h.cls.size() >= kMaxCls) {
      overflow = true;
      return 0;
    }
    h.cls.push_back(c);
    byStruct[key] = (int)h.cls.size() - 1;
    return (int)h.cls.size() - 1;
  };
  // class of a set of `d` samples per axis (more than one sample in all)
  auto of_dims = [&](auto&& self, const std::array<uint32_t, 3>& d) -> int {
    auto it = byDims.find(d);
    if (it != byDims.end())
      return it->second;
    uint32_t part[3][2], n[3];
    for (int a = 0; a < 3; a++) {
      part[a][1] = d[a] > 1 ? d[a] / 2 : 0;
      part[a][0] = d[a] - part[a][1];
      n[a] = d[a] > 1 ? 2 : 1;
    }
    ShapeCls c{};
    c.slot = 0xff;
    c.nsplit = (uint8_t)((n[0] - 1) + (n[1] - 1) + (n[2] - 1));
    uint64_t maxT = 0;
    int hh = 0;
    const uint32_t nkAll = n[0] * n[1] * n[2];
    for (uint32_t ord = 0; ord < nkAll; ord++) {
          // (x fastest; the 2D coder takes its children from the far corner backwards)
          const uint32_t o = (h.flags & kTree2D) ? nkAll - 1u - ord : ord;
          const uint32_t cx = o % n[0], cy = (o / n[0]) % n[1], cz = o / (n[0] * n[1]);
          const std::array<uint32_t, 3> kd = {part[0][cx], part[1][cy], part[2][cz]};
          if (kd[0] * kd[1] * kd[2] == 1) {
            c.kid[c.nk++] = kClsPixel;
            maxT += 2;
          }
          else {
            const int kc = self(self, kd);
            c.kid[c.nk++] = (uint8_t)kc;
            if (!overflow) {
              maxT += 1 + (uint64_t)h.cls[kc].maxT;
              hh = std::max(hh, 1 + (int)h.cls[kc].h);
            }
          }
        }
    c.h = (uint8_t)std::min(hh, 255);
    c.maxT = (uint32_t)std::min<uint64_t>(maxT, 0xffffffffu);
    const int id = intern(c);
    byDims[d] = id;
    return id;
  };
  const Tree t = h.view();
  for (uint32_t gi = 0; gi < h.grids.size() && !overflow; gi++) {
    const Grid& g = h.grids[gi];
    const Root& r = h.roots[g.root];
    for (uint32_t k = 0; k < 8 && !overflow; k++) {
      std::array<uint32_t, 3> d;
      bool exists = true;
      for (int a = 0; a < 3; a++) {
        const uint32_t rem = (uint32_t)r.len[a] & ((1u << g.e[a]) - 1u);
        const uint32_t longer = (k >> a) & 1u;
        if (longer && rem == 0)
          exists = false;
        d[a] = ((uint32_t)r.len[a] >> g.e[a]) + longer;
      }
      const uint32_t cnt = d[0] * d[1] * d[2];
      if (!exists || cnt == 0)
        continue;
      if (cnt == 1) {
        if (g.depth == 0) {   // a one-sample root still is a set, with one pixel child
          ShapeCls c{};
          c.nk = 1;
          c.slot = 0xff;
          c.kid[0] = kClsPixel;
          c.maxT = 2;
          h.gridCls[gi * 8 + k] = (uint8_t)intern(c);
        }
        continue;
      }
      h.gridCls[gi * 8 + k] = (uint8_t)of_dims(of_dims, d);
    }
  }
  (void)t;
  if (overflow) {
    h.cls.clear();
    return;
  }
  // how many sets of the forest have each class
  h.clsCount.assign(h.cls.size(), 0);
  for (uint32_t gi = 0; gi < h.grids.size(); gi++) {
    const Grid& g = h.grids[gi];
    const Root& r = h.roots[g.root];
    for (uint32_t k = 0; k < 8; k++) {
      const uint8_t ci = h.gridCls[gi * 8 + k];
      if (ci == kClsPixel)
        continue;
      uint64_t cnt = 1;
      for (int a = 0; a < 3; a++) {
        const uint32_t rem = (uint32_t)r.len[a] & ((1u << g.e[a]) - 1u);
        cnt *= ((k >> a) & 1u) ? rem : (1u << g.e[a]) - rem;
      }
      h.clsCount[ci] += cnt;
    }
  }
  // Table slots = columns of the decoder's per-position rows (k_lis_mixed): 0 is the single
  // sample, 1..3 the leaf parents of 2, 4 and 8 samples, 4..7 the most frequent classes made of
  // those (h = 1), 8..11 the most frequent classes one step up (h = 2); a class must hold at least
  // `minShare` of the sets of its h to be worth a column, and needs columns for all its children.
  if (hmax >= 0)
    for (ShapeCls& c : h.cls)
      if (c.h == 0 && (c.nk == 2 || c.nk == 4 || c.nk == 8)) {
        c.slot = (uint8_t)(c.nk == 2 ? 1 : c.nk == 4 ? 2 : 3);
        h.nslots++;
        h.slotMaxT = std::max(h.slotMaxT, c.maxT);
      }
  for (int hh = 1; hh <= std::min(hmax, 2); hh++) {
    uint64_t all = 0;
    std::vector<size_t> order;
    for (size_t i = 0; i < h.cls.size(); i++)
      if (h.cls[i].h == hh) {
        all += h.clsCount[i];
        order.push_back(i);
      }
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) {
      return h.clsCount[x] != h.clsCount[y] ? h.clsCount[x] > h.clsCount[y] : x < y;
    });
    uint32_t next = 4u * (uint32_t)hh;
    for (size_t i : order) {
      ShapeCls& c = h.cls[i];
      bool ok = next < 4u * (uint32_t)hh + std::min<uint32_t>(maxSlots, 4) && c.maxT < 0x7000u &&
                (double)h.clsCount[i] >= minShare * (double)all;
      for (int k = 0; k < c.nk; k++)
        ok = ok && (c.kid[k] == kClsPixel || h.cls[c.kid[k]].slot != 0xff);
      if (!ok)
        continue;
      c.slot = (uint8_t)next++;
      h.nslots++;
      h.slotMaxT = std::max(h.slotMaxT, c.maxT);
    }
  }
  // the column group (slot / 4) most entries of each list level belong to: the walk keeps that
  // group's lengths in registers, entries of other groups are walked into
  h.levelGroup.assign(h.nlevels, 0);
  if (!h.allRegular) {
    std::vector<std::array<uint64_t, 3>> cnt(h.nlevels, {0, 0, 0});
    std::vector<uint32_t> top(h.nlevels, 0);
    const Tree tv = h.view();
    for (uint32_t gi = 0; gi < h.grids.size(); gi++) {
      const Grid& g = h.grids[gi];
      Node n;
      n.grid = (uint16_t)gi;
      for (uint32_t z = 0; z < (1u << g.e[2]); z++)
        for (uint32_t y = 0; y < (1u << g.e[1]); y++)
          for (uint32_t x = 0; x < (1u << g.e[0]); x++) {
            n.i[0] = (uint16_t)x;
            n.i[1] = (uint16_t)y;
            n.i[2] = (uint16_t)z;
            const uint32_t ci = node_cls(tv, n);
            if (ci == kClsPixel)
              continue;
            const uint32_t l = node_level(tv, n);
            top[l] = std::max<uint32_t>(top[l], std::min<uint32_t>(h.cls[ci].h, 2));
            if (h.cls[ci].slot != 0xff)
              cnt[l][h.cls[ci].slot >> 2]++;
          }
    }
    // (bits 4..5: the highest column group a window of this level's list can need: sets of h = 0
    // are never walked into and have no set children, and so on)
    for (uint32_t l = 0; l < h.nlevels; l++)
      h.levelGroup[l] = (uint8_t)((cnt[l][2] > cnt[l][1] && cnt[l][2] > cnt[l][0] ? 2 : cnt[l][1] > cnt[l][0] ? 1 : 0) |
                                  (top[l] << 4));
  }
}

// Columns of k_lis_mx (the GPU-wide decoder of lists that mix set shapes, speck_mx.hip): sixteen per
// stream position in four groups of four -- group 0: a single sample and the leaf parents of 2 / 4 / 8 samples;
// groups 1..3: twelve more classes, handed out by steps above the leaf parents (all classes one step up that
// hold at least 1 % of the sets of their step, at most eight; then two steps up, and so on while columns are
// left): a chunk with three ragged axes has eight classes of 4x4x4-sized sets of comparable frequency (232 =
// 40 x 4 + 24 x 3 per axis: 24 % of them are 4x4x4), a chunk of 250^3 four (74 % are 4x4x4), a slice four
// per step.  A class needs columns for all its children.  The walk's tight loop sees TWO groups of a list at a
// time (eight split lengths of eight bits per lane): mxLevelGroup[l] = the group most entries of list level l
// belong to | the runner-up << 2 | the most steps above the leaf parents a set of the level can be (capped at 3) << 4.
constexpr int kMxGroups = 4;
inline void build_mx_columns(HostTree& h)
{
  h.mxSlot.assign(h.cls.size(), 0xff);
  h.mxLevelGroup.assign(h.nlevels, 0);
  if (h.cls.empty())
    return;
  for (size_t i = 0; i < h.cls.size(); i++) {
    const ShapeCls& c = h.cls[i];
    if (c.h == 0 && (c.nk == 2 || c.nk == 4 || c.nk == 8))
      h.mxSlot[i] = (uint8_t)(c.nk == 2 ? 1 : c.nk == 4 ? 2 : 3);
  }
  // twelve more columns, greedily: the class that saves the most walking into sets among those whose children
  // all have columns (sets of the forest x 2^steps: a set further up costs more rounds when it is walked into,
  // and no class above it can have a column either); up to three steps above the leaf parents
  for (uint32_t next = 4; next < 16u; next++) {
    int best = -1;
    double bestW = 0.0;
    for (size_t i = 0; i < h.cls.size(); i++) {
      const ShapeCls& c = h.cls[i];
      if (h.mxSlot[i] != 0xff || c.h == 0 || c.h > 3 || c.maxT >= 0x7000u || h.clsCount[i] == 0)
        continue;
      bool ok = true;
      for (int k = 0; k < c.nk; k++)
        ok = ok && (c.kid[k] == kClsPixel || h.mxSlot[c.kid[k]] != 0xff);
      const double w = (double)h.clsCount[i] * (double)(1u << c.h);
      if (ok && w > bestW) {
        bestW = w;
        best = (int)i;
      }
    }
    if (best < 0)
      break;
    h.mxSlot[best] = (uint8_t)next;
  }
  {   // columns numbered by steps, then by frequency: the classes of one list level end up in neighbouring groups
    std::vector<size_t> sel;
    for (size_t i = 0; i < h.cls.size(); i++)
      if (h.mxSlot[i] != 0xff && h.mxSlot[i] >= 4)
        sel.push_back(i);
    std::sort(sel.begin(), sel.end(), [&](size_t x, size_t y) {
      if (h.cls[x].h != h.cls[y].h)
        return h.cls[x].h < h.cls[y].h;
      return h.clsCount[x] != h.clsCount[y] ? h.clsCount[x] > h.clsCount[y] : x < y;
    });
    for (size_t k = 0; k < sel.size(); k++)
      h.mxSlot[sel[k]] = (uint8_t)(4 + k);
  }
  {   // one more, in an array of its own (column 16): the heaviest class left, up to four steps up
    int best = -1;
    double bestW = 0.0;
    for (size_t i = 0; i < h.cls.size(); i++) {
      const ShapeCls& c = h.cls[i];
      if (h.mxSlot[i] != 0xff || c.h == 0 || c.h > 4 || c.maxT >= 0x7000u || h.clsCount[i] == 0)
        continue;
      bool ok = true;
      for (int k = 0; k < c.nk; k++)
        ok = ok && (c.kid[k] == kClsPixel || h.mxSlot[c.kid[k]] != 0xff);
      const double w = (double)h.clsCount[i] * (double)(1u << c.h);
      if (ok && w > bestW) {
        bestW = w;
        best = (int)i;
      }
    }
    if (best >= 0)
      h.mxSlot[best] = 16;
  }
  std::vector<std::array<uint64_t, kMxGroups>> cnt(h.nlevels);
  std::vector<uint32_t> top(h.nlevels, 0);   // the most steps above the leaf parents a set of the level can be
  for (auto& c : cnt)
    c.fill(0);
  const Tree tv = h.view();
  for (uint32_t gi = 0; gi < h.grids.size(); gi++) {
    const Grid& g = h.grids[gi];
    // (class and level of a node depend on which of its three intervals are the long ones, and -- the level, on
    //  saturated axes -- on whether the parent interval still split: enumerate every node, the grids are small
    //  next to the volume)
    Node n;
    n.grid = (uint16_t)gi;
    for (uint32_t z = 0; z < (1u << g.e[2]); z++)
      for (uint32_t y = 0; y < (1u << g.e[1]); y++)
        for (uint32_t x = 0; x < (1u << g.e[0]); x++) {
          n.i[0] = (uint16_t)x;
          n.i[1] = (uint16_t)y;
          n.i[2] = (uint16_t)z;
          const uint32_t ci = node_cls(tv, n);
          if (ci == kClsPixel)
            continue;
          const uint32_t l = node_level(tv, n);
          if (l >= h.nlevels)
            continue;
          top[l] = std::max<uint32_t>(top[l], std::min<uint32_t>(h.cls[ci].h, 4));
          if (h.mxSlot[ci] < 16)
            cnt[l][h.mxSlot[ci] >> 2]++;
        }
  }
  for (uint32_t l = 0; l < h.nlevels; l++) {
    uint32_t best = 0;
    for (uint32_t g = 1; g < (uint32_t)kMxGroups; g++)
      if (cnt[l][g] > cnt[l][best])
        best = g;
    uint32_t second = best == 0 ? 1u : 0u;
    for (uint32_t g = 0; g < (uint32_t)kMxGroups; g++)
      if (g != best && cnt[l][g] > cnt[l][second])
        second = g;
    h.mxLevelGroup[l] = (uint8_t)(best | (second << 2) | (top[l] << 4));
  }
}

// twoD: the forest of the 2D coder for a slice of dx x dy samples (dz = 1): the root set S is the
// coarsest approximation, at list level = the number of transform levels; the three detail subbands
// of every level are roots too (released by the type-I set, SPECK2D_INT.cpp:149-218: they are NOT
// in the initial lists), at list level = their transform level.
// mxColumns: also the columns of k_lis_mx (build_mx_columns); a caller that knows the tree goes to the table kernels
// builds them later, if ever
inline HostTree build_tree(size_t dx, size_t dy, size_t dz, bool twoD = false, bool mxColumns = true)
{
  using detail::Box;
  HostTree h;
  h.dims[0] = (uint32_t)dx;
  h.dims[1] = (uint32_t)dy;
  h.dims[2] = (uint32_t)dz;
  h.nlevels = (uint32_t)(1 + num_of_partitions(dx) + num_of_partitions(dy) + num_of_partitions(dz));
  if (twoD) {
    h.flags = kTree2D;
    h.nlevels = (uint32_t)(1 + num_of_partitions(std::max(dx, dy)));
  }

  // ---- roots, in the order the reference pushes them -----------------------------------
  struct Pending {
    Box b;
    uint32_t lev;
  };
  std::vector<std::vector<Box>> lists(h.nlevels);
  Box big{{0, 0, 0}, {(uint32_t)dx, (uint32_t)dy, (uint32_t)dz}};
  uint32_t lev = 0;
  auto split = [&](bool sx, bool sy, bool sz) {
    // cut `big` on the chosen axes; children in x-fastest order; first child stays `big`
    uint32_t part[3][2], off[3][2];
    const bool use[3] = {sx, sy, sz};
    uint32_t inc = 0;
    for (int a = 0; a < 3; a++) {
      if (use[a]) {
        part[a][1] = big.len[a] / 2;
        part[a][0] = big.len[a] - part[a][1];
      }
      else {
        part[a][0] = big.len[a];
        part[a][1] = 0;
      }
      off[a][0] = big.org[a];
      off[a][1] = big.org[a] + part[a][0];
      inc += (use[a] && part[a][1] != 0) ? 1 : 0;
    }
    lev += inc;
    Box first = big;
    for (int k = 0; k < 8; k++) {
      const int hh[3] = {k & 1, (k >> 1) & 1, (k >> 2) & 1};
      if ((!use[0] && hh[0]) || (!use[1] && hh[1]) || (!use[2] && hh[2]))
        continue;
      Box c;
      for (int a = 0; a < 3; a++) {
        c.org[a] = off[a][hh[a]];
        c.len[a] = part[a][hh[a]];
      }
      if (k == 0)
        first = c;
      else
        lists[lev].push_back(c);
    }
    big = first;
  };
  size_t dyadic = 0;
  std::vector<std::pair<uint32_t, uint32_t>> iOrder;   // (level, index in lists[level]) of the released subbands
  if (twoD) {
    const size_t xf = num_of_xforms(std::min(dx, dy));
    h.iLevels = (uint32_t)xf;
    const auto ax = approx_detail_len(dx, xf), ay = approx_detail_len(dy, xf);
    lists[xf].push_back(Box{{0, 0, 0}, {(uint32_t)ax[0], (uint32_t)ay[0], 1}});   // S
    for (size_t k = xf; k >= 1; k--) {   // the type-I set at part_level k releases BR, TR, BL of level k
      const auto lx = approx_detail_len(dx, k), ly = approx_detail_len(dy, k);
      const Box sub[3] = {{{(uint32_t)lx[0], (uint32_t)ly[0], 0}, {(uint32_t)lx[1], (uint32_t)ly[1], 1}},
                          {{(uint32_t)lx[0], 0, 0}, {(uint32_t)lx[1], (uint32_t)ly[0], 1}},
                          {{0, (uint32_t)ly[0], 0}, {(uint32_t)lx[0], (uint32_t)ly[1], 1}}};
      for (const Box& bx : sub) {
        if (bx.len[0] == 0 || bx.len[1] == 0) {
          iOrder.push_back({0xffffffffu, 0});
          continue;
        }
        iOrder.push_back({(uint32_t)k, (uint32_t)lists[k].size()});
        lists[k].push_back(bx);
      }
    }
    lev = (uint32_t)xf;
  }
  else if (can_use_dyadic({dx, dy, dz}, dyadic)) {
    for (size_t i = 0; i < dyadic; i++)
      split(true, true, true);
  }
  else {
    const size_t nxy = num_of_xforms(std::min(dx, dy)), nz = num_of_xforms(dz);
    size_t xf = 0;
    for (; xf < nxy && xf < nz; xf++)
      split(true, true, true);
    for (; xf < nxy; xf++)
      split(true, true, false);
    for (; xf < nz; xf++)
      split(false, false, true);
  }
  if (!twoD)
    lists[lev].insert(lists[lev].begin(), big);

  // ---- per-root tables -----------------------------------------------------------------
  h.initLIS.assign(h.nlevels, {});
  std::vector<std::vector<uint64_t>> rootNode(h.nlevels);
  for (uint32_t l = 0; l < h.nlevels; l++)
    for (const Box& b : lists[l]) {
      Root r{};
      uint8_t dmax = 0;
      for (int a = 0; a < 3; a++) {
        r.org[a] = (uint16_t)b.org[a];
        r.len[a] = (uint16_t)b.len[a];
        r.D[a] = (uint8_t)detail::ceil_log2(b.len[a]);
        dmax = std::max(dmax, r.D[a]);
        r.tabOff[a] = (uint32_t)h.tab.size();
        for (int e = 0; e <= r.D[a]; e++) {
          uint32_t s = 0;
          for (uint32_t i = 0; i < (1u << e); i++) {
            h.tab.push_back((uint16_t)s);
            s += axis_len(b.len[a], e, i);
          }
          h.tab.push_back((uint16_t)s);
        }
      }
      r.Dmax = std::max<uint8_t>(dmax, 1);  // a one-sample root still is a set with one pixel child
      r.lev = (uint16_t)l;
      r.gridFirst = (uint16_t)h.grids.size();
      const uint16_t ri = (uint16_t)h.roots.size();
      for (int d = 0; d < r.Dmax; d++) {
        Grid g{};
        g.root = ri;
        g.depth = (uint8_t)d;
        for (int a = 0; a < 3; a++)
          g.e[a] = (uint8_t)std::min<int>(d, r.D[a]);
        bool oct = (b.org[0] % 2 == 0) && (dx % 2 == 0);
        for (int a = 0; a < 3; a++)
          oct = oct && d < r.D[a] && b.len[a] == (1u << r.D[a]);
        g.kind = oct ? kGridOct : 0;
        if (oct && d + 1 == r.Dmax && dx % 64 == 0 && b.org[0] % 64 == 0 && b.len[0] >= 64 &&
            b.org[1] % 2 == 0 && b.org[2] % 2 == 0)
          g.kind |= kGridLeafWord;
        g.nodeOff = h.nnodes;
        const uint32_t n = 1u << (g.e[0] + g.e[1] + g.e[2]);
        const uint32_t padded = (n + kNodeBlock - 1) / kNodeBlock * kNodeBlock;
        for (uint32_t b2 = 0; b2 < padded / kNodeBlock; b2++)
          h.blockGrid.push_back((uint16_t)h.grids.size());
        h.nnodes += padded;
        h.grids.push_back(g);
      }
      h.maxDepth = std::max<uint32_t>(h.maxDepth, r.Dmax);
      Node rn{r.gridFirst, {0, 0, 0}};
      if (!twoD || (l == h.iLevels && h.initLIS[l].empty()))   // (2D: only S is listed from the start)
        h.initLIS[l].push_back(pack_node(rn));
      rootNode[l].push_back(pack_node(rn));
      h.roots.push_back(r);
    }
  for (const auto& io : iOrder)
    h.iRoots.push_back(io.first == 0xffffffffu ? HostTree::kNoRoot : rootNode[io.first][io.second]);
  {
    bool allOct = !h.grids.empty();
    for (const Grid& g : h.grids)
      allOct = allOct && (g.kind & kGridOct) != 0;
    if (allOct)
      h.flags |= kTreeAllOct;
  }

  // ---- list capacities: how many set nodes can ever sit in each LIS level ----------------
  h.levelCap.assign(h.nlevels, 0);
  std::vector<std::array<uint32_t, 3>> levelShape(h.nlevels, {0, 0, 0});
  std::vector<int> levelState(h.nlevels, 0);  // 0: no sets, 1: one shape so far, 2: mixed
  const Tree t = h.view();
  for (uint32_t gi = 0; gi < h.grids.size(); gi++) {
    const Grid& g = h.grids[gi];
    Node n;
    n.grid = (uint16_t)gi;
    for (uint32_t z = 0; z < (1u << g.e[2]); z++)
      for (uint32_t y = 0; y < (1u << g.e[1]); y++)
        for (uint32_t x = 0; x < (1u << g.e[0]); x++) {
          n.i[0] = (uint16_t)x;
          n.i[1] = (uint16_t)y;
          n.i[2] = (uint16_t)z;
          const NodeGeom q = node_geom(t, n);
          if (q.count > 1 || (g.depth == 0 && q.count == 1)) {
            const uint32_t l = node_level(t, n);
            h.levelCap[l]++;
            h.nsets++;
            const std::array<uint32_t, 3> sh = {q.len[0], q.len[1], q.len[2]};
            if (levelState[l] == 0) {
              levelShape[l] = sh;
              levelState[l] = 1;
            }
            else if (levelState[l] == 1 && levelShape[l] != sh)
              levelState[l] = 2;
          }
        }
  }
  h.levelOff.assign(h.nlevels + 1, 0);
  for (uint32_t l = 0; l < h.nlevels; l++)
    h.levelOff[l + 1] = h.levelOff[l] + h.levelCap[l];

  // ---- shape classes per level (see LevelClass) ------------------------------------------
  h.levelClass.assign(h.nlevels, LevelClass{});
  h.allRegular = true;
  for (uint32_t l = 0; l < h.nlevels; l++) {
    if (levelState[l] == 0)
      continue;
    LevelClass& c = h.levelClass[l];
    std::array<uint32_t, 3> sh = levelShape[l];
    bool ok = levelState[l] == 1 && sh[0] * sh[1] * sh[2] >= 2;
    for (int a = 0; a < 3; a++)
      ok = ok && (sh[a] & (sh[a] - 1)) == 0;
    if (!ok) {
      h.allRegular = false;
      continue;
    }
    int ar[kMaxClasses], lv[kMaxClasses], K = 0;
    uint32_t cl = l;
    for (;;) {
      const int ns = (sh[0] > 1) + (sh[1] > 1) + (sh[2] > 1);
      ar[K] = 1 << ns;
      lv[K] = (int)cl;
      K++;
      cl += (uint32_t)ns;
      for (int a = 0; a < 3; a++)
        if (sh[a] > 1)
          sh[a] /= 2;
      if (sh[0] * sh[1] * sh[2] == 1)
        break;
    }
    c.regular = 1;
    c.K = (uint8_t)K;
    for (int j = 0; j < K; j++) {  // class 0 = leaf parent = the last shape of the chain
      c.arity[j] = (uint8_t)ar[K - 1 - j];
      c.lev[j] = (uint8_t)lv[K - 1 - j];
    }
  }
  if (twoD)   // (the chains of LevelClass follow the 3D level rule: the 2D forest goes by shape classes only)
    h.allRegular = false;
  build_classes(h, kClsTableH, kClsTableSlots);
  // (the columns of k_lis_mx enumerate every node of every grid -- as long as the rest of build_tree together: 57 of
  //  116 ms for 256^3, half a second for 512^3 -- and a regular tree the table kernels take never looks at them)
  if (mxColumns)
    build_mx_columns(h);
  else {
    h.mxSlot.clear();
    h.mxLevelGroup.assign(h.nlevels, 0);
  }
  return h;
}

}  // namespace spk

#endif
