// speck_enc.hip -- SPECK3D bit-plane set-partitioning ENCODER as data-parallel HIP kernels.
//
// The reference encoder (/root/reference/src/SPECK_INT.cpp:110-163,310-357,
// src/SPECK3D_INT.cpp:99-212, src/SPECK3D_INT_ENC.cpp:141-227) is a serial, data-dependent
// traversal that appends one bit at a time.  Every bit it emits, and the position it lands on, is
// a pure function of the msb of each coefficient and of each set's largest coefficient, so the
// stream is produced here as count -> scan -> scatter (tests/model/speck_model.cpp is the CPU
// model of exactly these kernels and is pinned bit-for-bit against the oracle):
//
//   k_pyramid        bottom-up: M[node] = msb of the set's largest coefficient, E[node] = bits the
//                    set's split emits, bplane[pixel] = plane at which the pixel enters the LIP
//   k_census/_scan   per pixel tile and plane: bits of the LIP scan and of the refinement pass
//   per plane p:
//     k_list_count / k_list_scan / k_list_apply   positions of the LIS entries (list order is
//                    part of the format), their '1' test bits, list compaction
//     k_split_emit   every set with M == p writes its children's test / sign bits at
//                    (position of the list entry that started its split chain) + (walk-up sum)
//     k_mask_scan / k_born_place   children that stay insignificant join the lists in the order
//                    of their stream position (= the reference's append order)
//   k_emit_pixels    LIP-scan and refinement bits, raster order, LDS-staged
//
// Bits past the budget are dropped, but whole passes are still counted so that the header's
// total_bits equals the reference's (the bit count at the end of the pass that crossed it).
#include "speck_enc.h"

namespace sperrhip {

using namespace spk;

#define ACTIVE_OR_RETURN(s, p)                                     \
  if (!(s).active || (s).done || (int)(p) >= (s).nbp)              \
    return;

// ------------------------------------------------------------------------------------------
__global__ void k_enc_state_init(EncBuffers b, const uint64_t* initLIS, const uint32_t* initLen,
                                 uint64_t budget, int wide_pass)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b.nchunks)
    return;
  EncState& s = b.st[c];
  const CoderState& cs = b.cst[c];
  // (a chunk already flagged for 64-bit coefficients -- PSNR mode decides that before coding --
  //  skips the 32-bit pass)
  s.active = wide_pass ? (cs.need_retry != 0) : (cs.is_const == 0 && cs.need_retry == 0);
  s.nbp = 0;
  s.done = 0;
  s.plast = 0;
  s.bornCount = 0;
  s.pos = 0;
  s.total_bits = 0;
  s.budget = budget;
  s.lisBits = 0;
  s.cur = 0;
  s.iPart = b.iLevels;
  for (int q = 0; q < kMaxPlanes; q++)
    s.bucketCnt[q] = 0;
  for (uint32_t l = 0; l < b.tree.nlevels; l++) {
    s.listLen[0][l] = initLen[l];
    s.listLen[1][l] = 0;
    s.bornTot[l] = 0;
    for (uint32_t k = 0; k < initLen[l]; k++)
      b.lis[0][c * b.lisStride + b.levelOff[l] + k] = initLIS[b.levelOff[l] + k];
  }
}

// ------------------------------------------------------------------------------------------
// Nodes of spk::kGridOct grids (all of them for power-of-two chunks): the 8 children are loaded
// with fixed indices, so everything stays in registers.  `base` is the raster index of pixel
// child 0 (deepest) or the flat id of set child 0; child j is base + (j & 1) + ((j >> 1) & 1) * sy
// + (j >> 2) * sz.
// ------------------------------------------------------------------------------------------
struct OctKids {
  int m[8];         // msb of each child
  uint32_t e[8];    // split length of each child set (0 for pixels)
  uint32_t base, sy, sz;
  bool deepest;
};

__device__ __forceinline__ uint32_t oct_kid(const OctKids& k, int j)
{
  return k.base + (uint32_t)(j & 1) + (uint32_t)((j >> 1) & 1) * k.sy + (uint32_t)(j >> 2) * k.sz;
}

__device__ __forceinline__ void oct_load(const Tree& t, const Grid& g, const Root& r, const Node& nd,
                                         const int8_t* M, const uint32_t* E, const int8_t* msb,
                                         OctKids& k)
{
  k.deepest = g.depth + 1 == r.Dmax;
  if (k.deepest) {
    k.sy = t.dims[0];
    k.sz = t.dims[0] * t.dims[1];
    k.base = ((uint32_t)r.org[2] + 2u * nd.i[2]) * k.sz + ((uint32_t)r.org[1] + 2u * nd.i[1]) * k.sy +
             (uint32_t)r.org[0] + 2u * nd.i[0];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const char2 v = *reinterpret_cast<const char2*>(msb + oct_kid(k, 2 * q));
      k.m[2 * q] = v.x;
      k.m[2 * q + 1] = v.y;
      k.e[2 * q] = 0;
      k.e[2 * q + 1] = 0;
    }
  }
  else {
    const Grid& cg = t.grids[nd.grid + 1];
    k.sy = 1u << cg.e[0];
    k.sz = 1u << (cg.e[0] + cg.e[1]);
    k.base = cg.nodeOff + 2u * nd.i[2] * k.sz + 2u * nd.i[1] * k.sy + 2u * nd.i[0];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t id = oct_kid(k, 2 * q);
      const char2 v = *reinterpret_cast<const char2*>(M + id);
      const uint2 w = *reinterpret_cast<const uint2*>(E + id);
      k.m[2 * q] = v.x;
      k.m[2 * q + 1] = v.y;
      k.e[2 * q] = w.x;
      k.e[2 * q + 1] = w.y;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Nodes of any other grid (chunks whose extents are not powers of two): up to eight children by ORDINAL
// (spk::KidBox: the non-empty children of a set are a box of 1 or 2 intervals per axis), everything in
// arrays that only ever see constant indices -- fully unrolled loops guarded by `j < n` -- so they stay in
// registers.  Until round 6 these nodes went through spk::Kids / spk::KidInfo, which node_kids fills with a
// running index: 156 bytes of private segment (scratch memory) in k_pyramid and k_split_emit, set up for every
// wavefront whether its nodes take this path or not.
// ------------------------------------------------------------------------------------------
struct KidRegs {
  KidBox kb;        // the children by ordinal (kid_index / kid_packed / kid_pixel_raster, speck_tree.h)
  int n;            // number of (non-empty) children
  bool deepest;     // the children are single samples of the deepest depth
  uint32_t pixels;  // bit j: child j is a single sample
  int m[8];         // msb of each child (-1 past the last)
  uint32_t e[8];    // split length of each child set (0 for single samples)
};

__device__ __forceinline__ uint32_t kid_regs_flat(const Tree& t, const KidRegs& k, int j)
{
  uint32_t idx[3];
  kid_index(k.kb, (uint32_t)j, idx);
  const Grid& cg = t.grids[k.kb.grid];
  return cg.nodeOff + (((idx[2] << cg.e[1]) + idx[1]) << cg.e[0]) + idx[0];
}

__device__ __forceinline__ void kid_regs_load(const Tree& t, const Node& nd, const int8_t* M, const uint32_t* E,
                                              const int8_t* msb, KidRegs& k)
{
  const Grid& g = t.grids[nd.grid];
  const Root& r = t.roots[g.root];
  kid_box(t, nd, k.kb);
  k.n = (int)k.kb.nk;
  k.deepest = g.depth + 1 == r.Dmax;
  k.pixels = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    k.m[j] = -1;
    k.e[j] = 0;
    if (j < k.n) {
      uint32_t idx[3];
      kid_index(k.kb, (uint32_t)j, idx);
      const uint32_t cnt = axis_len(r.len[0], k.kb.e[0], idx[0]) * axis_len(r.len[1], k.kb.e[1], idx[1]) *
                           axis_len(r.len[2], k.kb.e[2], idx[2]);
      const bool pixel = cnt == 1;
      k.pixels |= (pixel ? 1u : 0u) << j;
      if (k.deepest)
        k.m[j] = msb[pixel_raster(t, r, k.kb.e, idx)];
      else {
        const uint32_t fid = kid_regs_flat(t, k, j);
        k.m[j] = M[fid];
        k.e[j] = pixel ? 0u : E[fid];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_pyramid: one thread per node of the grids at one depth (deepest first)
// ------------------------------------------------------------------------------------------
// h[bin] += 1 for every lane with bin >= 0 (bin < 64); returns the lane's rank inside its bin.  The lanes of a
// wavefront that share a bin send ONE LDS atomic (neighbouring nodes split at a handful of planes: 64 lanes on one
// LDS word are 64 serialised atomics).  Who shares a lane's bin comes out of six ballots, one per bit of the bin
// (round 5) -- a loop over the distinct bins of the wavefront before: a ballot, two shuffles and an LDS round trip
// per bin, a dozen times in a wavefront of the deepest depth, where k_pyramid and k_chain spend 1.4 and 1.5 ms.
__device__ __forceinline__ uint32_t wave_hist_add(uint32_t* h, int bin)
{
  static_assert(kMaxPlanes <= 64, "six bits of bin");
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t same = __ballot(bin >= 0);
  if (same == 0)
    return 0;   // (uniform)
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const uint64_t bk = __ballot((bin >> k) & 1);
    same &= ((bin >> k) & 1) ? bk : ~bk;
  }
  if (bin < 0)
    same = 1ull << lane;   // (takes no part: its own leader, no atomic)
  const int lead = __ffsll((long long)same) - 1;
  uint32_t base0 = 0;
  if ((int)lane == lead && bin >= 0)
    base0 = atomicAdd(&h[bin], (uint32_t)__popcll(same));
  base0 = (uint32_t)__shfl((int)base0, lead, 64);
  return bin >= 0 ? base0 + (uint32_t)__popcll(same & ((1ull << lane) - 1ull)) : 0u;
}

// (returns the plane at which the node splits when it is a set with a significant sample, else -1)
// (ANY: the tree has grids that are not octrees -- spk::kTreeAllOct unset; the `false` instantiation, what every
//  power-of-two chunk runs, does not carry the general path's registers: 56 against 67)
template <bool ANY>
__device__ __forceinline__ int pyramid_node(const EncBuffers& b, uint32_t c, uint32_t id)
{
  const Tree& t = b.tree;
  Node nd;
  if (!node_from_flat(t, id, nd))
    return -1;
  const NodeGeom q = node_geom(t, nd);
  if (q.count == 0)
    return -1;
  int8_t* M = b.M + c * b.nodeStride;
  uint32_t* E = b.E + c * b.nodeStride;
  const int8_t* msb = b.msb + c * b.pixStride;
  int8_t* bplane = b.bplane + c * b.pixStride;
  const Grid& g = t.grids[nd.grid];
  const Root& r = t.roots[g.root];
  if (g.kind & kGridOct) {
    OctKids k;
    oct_load(t, g, r, nd, M, b.E + c * b.nodeStride, msb, k);
    int m = -1;
#pragma unroll
    for (int j = 0; j < 8; j++)
      m = max(m, k.m[j]);
    uint32_t bits = 0, ko[8];
    bool found = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {  // split_bits and kid_offset in one sweep
      const bool coded = found || j != 7;
      ko[j] = bits + (coded ? 1u : 0u);
      bits += coded ? 1u : 0u;
      if (!coded || k.m[j] == m) {
        found = true;
        bits += k.deepest ? 1u : k.e[j];
      }
    }
    M[id] = (int8_t)m;
    E[id] = m >= 0 ? bits : 0u;
    if (k.deepest) {
      // what k_split_emit needs of the eight pixels: which ones reach the set's msb, and their signs
      const uint64_t* sign = b.sign + c * b.signStride;
      uint32_t desc = 0;
#pragma unroll
      for (int q2 = 0; q2 < 4; q2++) {
        const uint32_t ridx = oct_kid(k, 2 * q2);   // even: both pixels sit in one sign word
        const uint32_t sg = (uint32_t)(sign[ridx >> 6] >> (ridx & 63)) & 3u;
        desc |= (uint32_t)(k.m[2 * q2] == m) << (2 * q2);
        desc |= (uint32_t)(k.m[2 * q2 + 1] == m) << (2 * q2 + 1);
        desc |= sg << (8 + 2 * q2);
      }
      b.leafDesc[c * b.nodeStride + id] = (uint16_t)desc;
      const char2 v = {(char)m, (char)m};
#pragma unroll
      for (int q2 = 0; q2 < 4; q2++)
        *reinterpret_cast<char2*>(bplane + oct_kid(k, 2 * q2)) = v;
    }
    else if (m >= 0) {
      uint32_t* koff = b.koff + c * b.nodeStride;
#pragma unroll
      for (int q2 = 0; q2 < 4; q2++)
        *reinterpret_cast<uint2*>(koff + oct_kid(k, 2 * q2)) = make_uint2(ko[2 * q2], ko[2 * q2 + 1]);
    }
    return m;
  }
  if constexpr (!ANY)
    return -1;   // (not reached: every grid is an octree)
  const bool isset = q.count > 1 || g.depth == 0;
  if (!isset) {
    const int e[3] = {g.e[0], g.e[1], g.e[2]};
    const uint32_t ii[3] = {nd.i[0], nd.i[1], nd.i[2]};
    M[id] = msb[pixel_raster(t, r, e, ii)];
    return -1;
  }
  KidRegs k;
  kid_regs_load(t, nd, M, E, msb, k);
  int m = -1;
#pragma unroll
  for (int j = 0; j < 8; j++)
    m = max(m, k.m[j]);   // (-1 past the last child)
  // split_bits (spk, speck_tree.h) and kid_offset in one sweep: koff[child] = where the child's own split starts
  // inside this node's split (bits of the earlier children + the child's test bit when it is coded): a split
  // chain's position is then a sum of one word per ancestor instead of a re-evaluation of every ancestor's children
  uint32_t bits = 0, ko[8];
  bool found = false;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    ko[j] = 0;
    if (j < k.n) {
      const bool coded = found || (j + 1 != k.n);
      bits += coded ? 1u : 0u;
      ko[j] = bits;
      if (!coded || k.m[j] == m) {
        found = true;
        bits += ((k.pixels >> j) & 1u) ? 1u : k.e[j];
      }
    }
  }
  M[id] = (int8_t)m;
  E[id] = m >= 0 ? bits : 0u;
#pragma unroll
  for (int j = 0; j < 8; j++)
    if (j < k.n && ((k.pixels >> j) & 1u))
      bplane[kid_pixel_raster(t, nd, k.kb, (uint32_t)j)] = (int8_t)m;
  if (m >= 0 && !k.deepest) {
    uint32_t* koff = b.koff + c * b.nodeStride;
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (j < k.n && !((k.pixels >> j) & 1u))
        koff[kid_regs_flat(t, k, j)] = ko[j];
  }
  return m;
}

// one thread per node; the splitting sets are counted per plane on the way (the bucket histogram: a
// pass of its own over every node before, k_bucket_hist, 1.1 ms per launch of 21 chunks)
// The deepest depth of an octree grid, four leaf parents in a row per thread (round 5): their 32 samples are four
// 8-byte loads (two rows of two slices), what comes out is one store per array -- M, E, leaf descriptor, and the four
// rows of birth planes.  One node per thread moved the same bytes two at a time: 1.4 ms per launch of 21 chunks for
// under a gigabyte.  Needs rows of eight samples that start on a multiple of eight (root origin, row length).
__device__ __forceinline__ bool leaf4_block(const Tree& t, uint32_t blockId)
{
  const Grid& g = t.grids[t.blockGrid[blockId]];
  const Root& r = t.roots[g.root];
  return (g.kind & kGridOct) && g.depth + 1 == r.Dmax && g.e[0] >= 2 && ((r.org[0] | t.dims[0]) & 7u) == 0;
}
__device__ __forceinline__ void pyramid_leaf4(const EncBuffers& b, uint32_t c, uint32_t id0, uint32_t* h)
{
  const Tree& t = b.tree;
  const Grid& g = t.grids[t.blockGrid[id0 / kNodeBlock]];
  const Root& r = t.roots[g.root];
  const uint32_t local = id0 - g.nodeOff;
  if (local >= (1u << (g.e[0] + g.e[1] + g.e[2])))
    return;   // (padding ids)
  const uint32_t i0 = local & ((1u << g.e[0]) - 1u), i1 = (local >> g.e[0]) & ((1u << g.e[1]) - 1u),
                 i2 = local >> (g.e[0] + g.e[1]);
  const uint32_t sy = t.dims[0], sz = t.dims[0] * t.dims[1];
  const uint32_t base = ((uint32_t)r.org[2] + 2u * i2) * sz + ((uint32_t)r.org[1] + 2u * i1) * sy + (uint32_t)r.org[0] + 2u * i0;
  const int8_t* msb = b.msb + c * b.pixStride;
  int8_t* bplane = b.bplane + c * b.pixStride;
  const uint64_t* sign = b.sign + c * b.signStride;
  uint2 rows[4];
  uint32_t sg[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {   // row k: y + (k & 1), z + (k >> 1)
    const uint32_t ridx = base + (uint32_t)(k & 1) * sy + (uint32_t)(k >> 1) * sz;
    rows[k] = *reinterpret_cast<const uint2*>(msb + ridx);
    sg[k] = (uint32_t)(sign[ridx >> 6] >> (ridx & 63)) & 0xffu;   // (eight samples of one sign word)
  }
  uint32_t mPack = 0, ePack[4], dPack[2] = {0, 0};
  int mv[4];
#pragma unroll
  for (int n = 0; n < 4; n++) {
    int km[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {   // child j = (x + (j & 1), y + ((j >> 1) & 1), z + (j >> 2))
      const uint2 rw = rows[((j >> 1) & 1) + 2 * (j >> 2)];
      const uint32_t byte = 2u * (uint32_t)n + (uint32_t)(j & 1);
      km[j] = (int)(int8_t)(((byte < 4 ? rw.x : rw.y) >> (8u * (byte & 3u))) & 0xffu);
    }
    int m = -1;
#pragma unroll
    for (int j = 0; j < 8; j++)
      m = max(m, km[j]);
    uint32_t bits = 0, desc = 0;
    bool found = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const bool coded = found || j != 7;
      bits += coded ? 1u : 0u;
      if (!coded || km[j] == m) {
        found = true;
        bits += 1u;
      }
      desc |= (uint32_t)(km[j] == m) << j;
      desc |= ((sg[((j >> 1) & 1) + 2 * (j >> 2)] >> (2u * (uint32_t)n + (uint32_t)(j & 1))) & 1u) << (8 + j);
    }
    mv[n] = m;
    mPack |= ((uint32_t)m & 0xffu) << (8 * n);
    ePack[n] = m >= 0 ? bits : 0u;
    dPack[n >> 1] |= (desc & 0xffffu) << (16 * (n & 1));
    if (m >= 0)
      atomicAdd(&h[m], 1u);
  }
  *reinterpret_cast<uint32_t*>(b.M + c * b.nodeStride + id0) = mPack;
  *reinterpret_cast<uint4*>(b.E + c * b.nodeStride + id0) = make_uint4(ePack[0], ePack[1], ePack[2], ePack[3]);
  *reinterpret_cast<uint2*>(b.leafDesc + c * b.nodeStride + id0) = make_uint2(dPack[0], dPack[1]);
  // the samples' birth planes: their parent's msb, two samples a node
  const uint32_t pairs[4] = {((uint32_t)mv[0] & 0xffu) * 0x0101u, ((uint32_t)mv[1] & 0xffu) * 0x0101u,
                             ((uint32_t)mv[2] & 0xffu) * 0x0101u, ((uint32_t)mv[3] & 0xffu) * 0x0101u};
  const uint2 bp = make_uint2(pairs[0] | (pairs[1] << 16), pairs[2] | (pairs[3] << 16));
#pragma unroll
  for (int k = 0; k < 4; k++)
    *reinterpret_cast<uint2*>(bplane + base + (uint32_t)(k & 1) * sy + (uint32_t)(k >> 1) * sz) = bp;
}

// (round 5: a workgroup takes `per` node blocks of the depth one after the other -- eight at the deepest depths, where
//  150 K workgroups of one node per thread each paid the look-ups of their grid and a histogram's worth of global
//  atomics on the chunk's few plane counters)
constexpr int kNodePerMax = 8;
template <bool ANY>
__global__ void __launch_bounds__(kNodeBlock)
k_pyramid(EncBuffers b, const uint32_t* depthBlocks, uint32_t nblk, uint32_t per)
{
  const uint32_t c = blockIdx.y;
  EncState& s = b.st[c];
  if (!s.active)
    return;
  __shared__ uint32_t h[kMaxPlanes];
  if (threadIdx.x < kMaxPlanes)
    h[threadIdx.x] = 0;
  __syncthreads();
  static_assert(kNodeBlock == 256, "a wavefront takes a node block four nodes a lane");
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  // blocks of leaf parents that pyramid_leaf4 takes: a wavefront each, four at a time
  for (uint32_t q = wave; q < per; q += kNodeBlock / 64) {
    const uint32_t bi = blockIdx.x * per + q;
    if (bi >= nblk)
      break;
    const uint32_t blk = depthBlocks[bi];
    if (leaf4_block(b.tree, blk))   // (uniform over the wavefront)
      pyramid_leaf4(b, c, blk * kNodeBlock + lane * 4u, h);
  }
  // every other block: a node per thread
  for (uint32_t q = 0, bi = blockIdx.x * per; q < per && bi < nblk; q++, bi++) {
    const uint32_t blk = depthBlocks[bi];
    if (leaf4_block(b.tree, blk))   // (uniform over the workgroup)
      continue;
    const int sp = pyramid_node<ANY>(b, c, blk * kNodeBlock + threadIdx.x);
    wave_hist_add(h, sp);
  }
  __syncthreads();
  if (threadIdx.x < kMaxPlanes && h[threadIdx.x])
    atomicAdd(&s.bucketCnt[threadIdx.x], h[threadIdx.x]);
}

// Top-down over the depths (shallowest first): where each set's split sits relative to the list
// entry that starts its chain of nested splits.  A set whose parent splits on the same plane
// (M equal) is coded inside the parent's split; otherwise it is a list entry itself.  The sets of
// a root's deepest depth are not stored: k_split_emit takes the one step from their parent.
// On the way the splitting sets are dealt to the buckets of their planes (k_bucket_fill, a pass of its own
// over every node before: 1.2 ms per launch of 21 chunks; k_bucket_scan has run on k_pyramid's counts).
__global__ void __launch_bounds__(kNodeBlock)
k_chain(EncBuffers b, const uint32_t* depthBlocks, uint32_t nblk, uint32_t per)
{
  const uint32_t c = blockIdx.y;
  EncState& s = b.st[c];
  if (!s.active)
    return;
  __shared__ uint32_t h[kMaxPlanes], base[kMaxPlanes];
  if (threadIdx.x < kMaxPlanes)
    h[threadIdx.x] = 0;
  __syncthreads();
  const Tree& t = b.tree;
  const int8_t* M = b.M + c * b.nodeStride;
  // (`per` node blocks per workgroup, see k_pyramid: the ranks of a thread's nodes wait in registers for the
  //  workgroup's one claim per plane)
  uint32_t ids[kNodePerMax], ranks[kNodePerMax];
  int bins[kNodePerMax];
#pragma unroll
  for (int q = 0; q < kNodePerMax; q++) {
    bins[q] = -1;
    ids[q] = ranks[q] = 0;
    const uint32_t bi = blockIdx.x * per + (uint32_t)q;
    if ((uint32_t)q >= per || bi >= nblk)   // (uniform)
      continue;
    const uint32_t id = depthBlocks[bi] * kNodeBlock + threadIdx.x;
    Node nd;
    const bool valid = node_from_flat(t, id, nd);
    const int m = valid ? (int)M[id] : -1;
    bool mine = false;   // a set that splits at plane m (splitting_set)
    if (m >= 0) {
      const Grid& g = t.grids[nd.grid];
      mine = (g.kind & kGridOct) != 0;
      if (!mine) {
        const NodeGeom qg = node_geom(t, nd);
        mine = qg.count > 1 || (g.depth == 0 && qg.count == 1);
      }
      if (!(g.depth + 1 == t.roots[g.root].Dmax && g.depth != 0)) {
        uint64_t* chain = b.chain + c * b.nodeStride;
        uint64_t v = id;
        if (g.depth != 0) {
          const uint32_t pid = flat_id(t, node_parent(t, nd));
          if (M[pid] == m) {
            const uint64_t pc = chain[pid];
            v = (pc & 0xffffffffull) | ((uint64_t)((uint32_t)(pc >> 32) + b.koff[c * b.nodeStride + id]) << 32);
          }
        }
        chain[id] = v;
      }
    }
    ids[q] = id;
    bins[q] = mine ? m : -1;
    ranks[q] = wave_hist_add(h, bins[q]);
  }
  __syncthreads();
  if (threadIdx.x < kMaxPlanes && h[threadIdx.x])
    base[threadIdx.x] = atomicAdd(&s.bucketCur[threadIdx.x], h[threadIdx.x]);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kNodePerMax; q++)
    if (bins[q] >= 0)
      b.bucket[c * b.nodeStride + base[bins[q]] + ranks[q]] = ids[q];
}

__global__ void k_enc_planes_setup(EncBuffers b)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b.nchunks)
    return;
  EncState& s = b.st[c];
  if (!s.active)
    return;
  const Tree& t = b.tree;
  int nbp = 0;
  for (uint32_t r = 0; r < t.nroots; r++)
    nbp = max(nbp, (int)b.M[c * b.nodeStride + t.grids[t.roots[r].gridFirst].nodeOff] + 1);
  s.nbp = nbp;
  s.plast = nbp;
  if (nbp == 0) {  // all coefficients are zero (SPECK_INT.cpp:130-133)
    s.done = 1;
    s.total_bits = 0;
  }
}

// ------------------------------------------------------------------------------------------
// census of the pixel passes
// ------------------------------------------------------------------------------------------
// kPixPer consecutive int8 values starting at i0 (a multiple of kPixPer); -1 past the end
__device__ __forceinline__ void load_pix(const int8_t* a, uint32_t i0, uint32_t n, int v[kPixPer])
{
  static_assert(kPixPer == 16, "one 16-byte load per thread");
  if (i0 + kPixPer <= n) {
    const uint4 q = *reinterpret_cast<const uint4*>(a + i0);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < kPixPer; k++)
      v[k] = (int)(int8_t)(w[k >> 2] >> ((k & 3) * 8));
  }
  else
    for (int k = 0; k < kPixPer; k++)
      v[k] = (i0 + k < n) ? a[i0 + k] : -1;
}

// Bits each plane's LIP scan and refinement pass take inside one tile.  A sample with msb m that
// enters the LIP at plane b (b >= m) costs one LIP bit on planes m <= p < b plus a sign bit on
// plane m, and one refinement bit on every plane p < m: all of it follows from three histograms.
// (rows of all `maxPlanes` planes are written, zeros above the chunk's own plane count: the kernel may
//  run before k_enc_planes_setup has found that count -- on a second stream, launch_speck_encode)
__global__ void __launch_bounds__(kThreads) k_census(EncBuffers b, int maxPlanes)
{
  const uint32_t c = blockIdx.y;
  const EncState& s = b.st[c];
  if (!s.active)
    return;
  // bin q + 1 counts value q (-1 .. kMaxPlanes - 1); one set of histograms per wavefront
  // [0]: msb, low half = all samples, high half = those that sit in the LIP before they become
  // significant (bp > msb >= 0); [1]: birth plane.  (A wavefront's share of a tile is 1024
  // samples, so 16 bits per half are plenty.)
  __shared__ uint32_t hist[kThreads / 64][2][kMaxPlanes + 1];
  for (uint32_t i = threadIdx.x; i < (kThreads / 64) * 2 * (kMaxPlanes + 1); i += kThreads)
    (&hist[0][0][0])[i] = 0;
  __syncthreads();
  const uint32_t n = b.tree.nvals;
  const uint32_t i0 = blockIdx.x * kPixTile + threadIdx.x * kPixPer;
  const int wave = threadIdx.x >> 6;
  // most samples are zero (msb -1) and were never born (-1): those two bins are summed over the
  // wavefront and added once, instead of up to 64 lanes queueing on the same LDS word 16 times
  uint32_t zeroM = 0, zeroB = 0;
  if (i0 < n) {
    int m[kPixPer], bp[kPixPer];
    load_pix(b.msb + c * b.pixStride, i0, n, m);
    load_pix(b.bplane + c * b.pixStride, i0, n, bp);
#pragma unroll
    for (int k = 0; k < kPixPer; k++)
      if (i0 + k < n) {
        if (m[k] < 0)
          zeroM++;
        else
          atomicAdd(&hist[wave][0][m[k] + 1], bp[k] > m[k] ? 0x10001u : 1u);
      }
    // (the kernel is bound by its LDS atomics: two samples side by side are children of one set nearly everywhere and
    //  were born on the same plane -- one atomic for the pair)
#pragma unroll
    for (int k = 0; k < kPixPer; k += 2) {
      const bool a = i0 + k < n, c2 = i0 + k + 1 < n;
      if (a && c2 && bp[k] == bp[k + 1]) {
        if (bp[k] < 0)
          zeroB += 2;
        else
          atomicAdd(&hist[wave][1][bp[k] + 1], 2u);
      }
      else {
        if (a) {
          if (bp[k] < 0)
            zeroB++;
          else
            atomicAdd(&hist[wave][1][bp[k] + 1], 1u);
        }
        if (c2) {
          if (bp[k + 1] < 0)
            zeroB++;
          else
            atomicAdd(&hist[wave][1][bp[k + 1] + 1], 1u);
        }
      }
    }
  }
  uint32_t z = zeroM | (zeroB << 16);       // both at most 1024 per wavefront
  for (int d = 32; d > 0; d >>= 1)
    z += __shfl_xor(z, d, 64);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&hist[wave][0][0], z & 0xffffu);
    atomicAdd(&hist[wave][1][0], z >> 16);
  }
  __syncthreads();
  // the tile's histograms (sum over the wavefronts), then their running sums over the bins
  __shared__ uint32_t tot[2][kMaxPlanes + 1];
  __shared__ uint64_t pre[kMaxPlanes + 1];      // low half: msb bins 0..q (all samples), high: birth bins
  for (uint32_t i = threadIdx.x; i < 2 * (kMaxPlanes + 1); i += kThreads) {
    const uint32_t h = i / (kMaxPlanes + 1), q = i % (kMaxPlanes + 1);
    uint32_t v = 0;
    for (int w = 0; w < kThreads / 64; w++)
      v += hist[w][h][q];                       // (both 16-bit halves of [0] stay below 2^13)
    tot[h][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    static_assert(kMaxPlanes == 64, "one wavefront scans the bins");
    const uint64_t v = (uint64_t)(tot[0][threadIdx.x] & 0xffffu) | ((uint64_t)tot[1][threadIdx.x] << 32);
    const uint64_t inc = wave_inclusive_scan<uint64_t>(v);
    pre[threadIdx.x] = inc;
    if (threadIdx.x == 63)
      pre[64] = inc + ((uint64_t)(tot[0][64] & 0xffffu) | ((uint64_t)tot[1][64] << 32));
  }
  __syncthreads();
  const int p = threadIdx.x;
  if (p < maxPlanes) {
    const uint32_t le_m = (uint32_t)pre[p + 1], le_b = (uint32_t)(pre[p + 1] >> 32);
    const uint32_t all_m = (uint32_t)pre[kMaxPlanes], eq = tot[0][p + 1] >> 16;
    uint32_t* cnt = b.pixCnt + c * b.pixCntStride;
    cnt[(size_t)(p * 2 + 0) * b.nPixTiles + blockIdx.x] = le_m - le_b + eq;   // LIP scan bits
    cnt[(size_t)(p * 2 + 1) * b.nPixTiles + blockIdx.x] = all_m - le_m;       // refinement bits
  }
}

// one block per (plane, phase) row and chunk: exclusive scan over the pixel tiles
__global__ void __launch_bounds__(kThreads) k_census_scan(EncBuffers b)
{
  const uint32_t c = blockIdx.y;
  EncState& s = b.st[c];
  const int p = blockIdx.x >> 1, phase = blockIdx.x & 1;
  if (!s.active || s.done || p >= s.nbp)
    return;
  __shared__ uint64_t sm[kThreads / 64 + 1];
  const uint32_t* cnt = b.pixCnt + c * b.pixCntStride + (size_t)blockIdx.x * b.nPixTiles;
  uint32_t* off = b.pixOff + c * b.pixCntStride + (size_t)blockIdx.x * b.nPixTiles;
  uint64_t carry = 0;
  for (uint32_t base = 0; base < b.nPixTiles; base += kThreads * 4) {
    uint32_t v[4];
    uint64_t tsum = 0;
    for (int k = 0; k < 4; k++) {
      const uint32_t i = base + threadIdx.x * 4 + k;
      v[k] = i < b.nPixTiles ? cnt[i] : 0;
      tsum += v[k];
    }
    uint64_t total;
    uint64_t ex = block_exclusive_scan<uint64_t>(tsum, sm, &total) + carry;
    for (int k = 0; k < 4; k++) {
      const uint32_t i = base + threadIdx.x * 4 + k;
      if (i < b.nPixTiles)
        off[i] = (uint32_t)ex;
      ex += v[k];
    }
    carry += total;
  }
  if (threadIdx.x == 0) {
    if (phase == 0)
      s.lipTot[p] = carry;
    else
      s.refTot[p] = carry;
  }
}

// ------------------------------------------------------------------------------------------
// plane bookkeeping
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void plane_begin(const EncBuffers& b, EncState& s, int p)
{
  ACTIVE_OR_RETURN(s, p);
  s.rec[p].baseLIP = s.pos;
  s.pos += s.lipTot[p];
  s.rec[p].baseLIS = s.pos;
  s.rec[p].didSort = 1;
  s.rec[p].didREF = 0;
  s.plast = p;
  s.bornCount = 0;
  for (uint32_t l = 0; l < b.tree.nlevels; l++)
    s.bornTot[l] = 0;
}

__device__ __forceinline__ void plane_end(const EncBuffers& b, EncState& s, int p)
{
  ACTIVE_OR_RETURN(s, p);
  const uint32_t nx = s.cur ^ 1u;
  for (uint32_t l = 0; l < b.tree.nlevels; l++)
    s.listLen[nx][l] += s.bornTot[l];
  s.cur = nx;
  s.pos += s.lisBits;
  if (s.pos >= s.budget) {  // SPECK_INT.cpp:149
    s.done = 1;
    s.total_bits = s.pos;
    return;
  }
  s.rec[p].baseREF = s.pos;
  s.pos += s.refTot[p];
  s.rec[p].didREF = 1;
  if (s.pos >= s.budget || p == 0) {  // SPECK_INT.cpp:153,161
    s.done = 1;
    s.total_bits = s.pos;
  }
}

// the bookkeeping between two planes in one launch: plane pEnd ends (pEnd < 0: none), plane pBegin
// begins (pBegin < 0: none) -- 31 launches fewer on every batch's serial chain
__global__ void k_plane_turn(EncBuffers b, int pEnd, int pBegin)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b.nchunks)
    return;
  EncState& s = b.st[c];
  if (pEnd >= 0)
    plane_end(b, s, pEnd);
  if (pBegin >= 0)
    plane_begin(b, s, pBegin);
}

// ------------------------------------------------------------------------------------------
// LIS entries: count, scan, apply
// ------------------------------------------------------------------------------------------
struct ListItem {
  uint64_t packed;
  uint32_t id;
  uint32_t bits;   // 0 when the slot is past the end of the list
  bool sig;
};

// A list entry carries its set's msb (round 5, second session): bits 57 .. 63 of the packed node hold msb + 2 (0: not
// known -- the lists' first entries and the 2D coder's subbands, whose M is looked up as before).  An entry is tested
// on every plane from its birth to the plane it splits on: the test was a dependent grid look-up for its flat id and a
// random 1-byte read of M per entry, plane and pass (k_list_count, k_list_apply); now it is a compare on the word the
// list walk reads anyway, and the id is worked out for the entries that split.  Trees of at most 511 grids (the
// grid index keeps 9 of its 16 bits).
constexpr int kEntryMsbShift = 57;
constexpr uint64_t kEntryNodeMask = (1ull << kEntryMsbShift) - 1ull;
__device__ __forceinline__ uint64_t entry_with_msb(const Tree& t, uint64_t packed, int m)
{
  return t.ngrids <= 511u ? packed | ((uint64_t)(uint32_t)(m + 2) << kEntryMsbShift) : packed;
}

__device__ __forceinline__ void load_list4(const EncBuffers& b, uint32_t c, const EncState& s,
                                           uint32_t tile, int p, ListItem it[4])
{
  const uint32_t l = b.tileLevel[tile], start = b.tileStart[tile];
  const uint32_t n = s.listLen[s.cur][l];
  const uint64_t* list = b.lis[s.cur] + c * b.lisStride + b.levelOff[l];
  const int8_t* M = b.M + c * b.nodeStride;
  const uint32_t* E = b.E + c * b.nodeStride;
  const bool carries = b.tree.ngrids <= 511u;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t e = start + threadIdx.x * 4 + k;
    it[k].packed = e < n ? list[e] : 0ull;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t e = start + threadIdx.x * 4 + k;
    it[k].bits = 0;
    it[k].sig = false;
    it[k].id = 0;
    if (e < n) {
      const uint32_t mf = carries ? (uint32_t)(it[k].packed >> kEntryMsbShift) : 0u;
      if (mf == 0 || (int)mf - 2 == p) {   // (the entries that split on this plane, and the few that carry no msb)
        it[k].id = flat_id(b.tree, unpack_node(carries ? it[k].packed & kEntryNodeMask : it[k].packed));
        it[k].sig = mf ? true : (M[it[k].id] == p);
      }
      it[k].bits = 1u + (it[k].sig ? E[it[k].id] : 0u);
    }
  }
}

// clearPrev: first the part of the birth masks the plane before used is cleared (s.lisBits still is that
// plane's; a launch of its own until round 4 -- nothing between here and k_split_emit touches the masks)
__global__ void __launch_bounds__(kThreads) k_list_count(EncBuffers b, int p, int clearPrev)
{
  const uint32_t c = blockIdx.y;
  const EncState& s = b.st[c];
  ACTIVE_OR_RETURN(s, p);
  if (clearPrev && b.nSlots) {
    const uint64_t nbits = min(s.lisBits, (uint64_t)b.maskWords * 64);
    const uint32_t nwords = (uint32_t)((nbits + 63) / 64);
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += gridDim.x * blockDim.x)
      for (uint32_t slot = 0; slot < b.nSlots; slot++)
        b.mask[c * b.maskStride + (size_t)slot * b.maskWords + w] = 0;
  }
  __shared__ uint64_t sm[kThreads / 64 + 1];
  for (uint32_t tile = blockIdx.x; tile < b.nListTiles; tile += gridDim.x) {
    if (b.tileStart[tile] >= s.listLen[s.cur][b.tileLevel[tile]]) {   // past the end of its list
      if (threadIdx.x == 0) {
        b.tileBits[c * b.tileStride + tile] = 0;
        b.tileSurv[c * b.tileStride + tile] = 0;
      }
      continue;
    }
    ListItem it[4];
    load_list4(b, c, s, tile, p, it);
    uint64_t v = 0;  // low 40 bits: bits, high 24 bits: survivors
    for (int k = 0; k < 4; k++)
      if (it[k].bits)
        v += (uint64_t)it[k].bits + (it[k].sig ? 0ull : (1ull << 40));
    uint64_t total;
    block_exclusive_scan<uint64_t>(v, sm, &total);
    if (threadIdx.x == 0) {
      b.tileBits[c * b.tileStride + tile] = total & ((1ull << 40) - 1);
      b.tileSurv[c * b.tileStride + tile] = (uint32_t)(total >> 40);
    }
  }
}

// one block per chunk: scan the tiles in traversal order (deepest level first)
__global__ void __launch_bounds__(kThreads) k_list_scan(EncBuffers b, int p)
{
  const uint32_t c = blockIdx.x;
  EncState& s = b.st[c];
  ACTIVE_OR_RETURN(s, p);
  __shared__ uint64_t sm[kThreads / 64 + 1];
  const uint64_t* tb = b.tileBits + c * b.tileStride;
  const uint32_t* ts = b.tileSurv + c * b.tileStride;
  uint64_t* tbo = b.tileBitsOff + c * b.tileStride;
  uint32_t* tso = b.tileSurvOff + c * b.tileStride;   // first pass: global scan
  uint64_t carryB = 0, carryS = 0;
  for (uint32_t base = 0; base < b.nListTiles; base += kThreads) {
    const uint32_t i = base + threadIdx.x;
    const uint64_t vb = i < b.nListTiles ? tb[i] : 0, vs = i < b.nListTiles ? ts[i] : 0;
    uint64_t totB, totS;
    const uint64_t eb = block_exclusive_scan<uint64_t>(vb, sm, &totB) + carryB;
    const uint64_t es = block_exclusive_scan<uint64_t>(vs, sm, &totS) + carryS;
    if (i < b.nListTiles) {
      tbo[i] = eb;
      tso[i] = (uint32_t)es;
    }
    carryB += totB;
    carryS += totS;
  }
  __syncthreads();
  // survivors restart at every level: new list lengths, then make tso relative to the level
  const uint32_t nx = s.cur ^ 1u;
  __shared__ uint32_t levBase[kMaxLevels];
  for (uint32_t l = threadIdx.x; l < b.tree.nlevels; l += blockDim.x) {
    const uint32_t nt = b.levelNumTiles[l];
    levBase[l] = 0;
    s.listLen[nx][l] = 0;
    if (nt) {
      const uint32_t first = b.levelFirstTile[l], last = first + nt;
      const uint32_t beg = tso[first];
      const uint32_t end = (last < b.nListTiles) ? tso[last] : (uint32_t)carryS;
      levBase[l] = beg;
      s.listLen[nx][l] = end - beg;
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < b.nListTiles; i += blockDim.x)
    tso[i] -= levBase[b.tileLevel[i]];
  if (threadIdx.x == 0)
    s.lisBits = carryB;
}

__global__ void __launch_bounds__(kThreads) k_list_apply(EncBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const EncState& s = b.st[c];
  ACTIVE_OR_RETURN(s, p);
  __shared__ uint64_t sm[kThreads / 64 + 1];
  for (uint32_t tile = blockIdx.x; tile < b.nListTiles; tile += gridDim.x) {
    const uint32_t l = b.tileLevel[tile];
    if (b.tileStart[tile] >= s.listLen[s.cur][l])
      continue;
    ListItem it[4];
    load_list4(b, c, s, tile, p, it);
    uint64_t v = 0;
    for (int k = 0; k < 4; k++)
      if (it[k].bits)
        v += (uint64_t)it[k].bits + (it[k].sig ? 0ull : (1ull << 40));
    uint64_t total;
    const uint64_t ex = block_exclusive_scan<uint64_t>(v, sm, &total);
    uint64_t bitpos = s.rec[p].baseLIS + b.tileBitsOff[c * b.tileStride + tile] +
                      (ex & ((1ull << 40) - 1));
    uint32_t rank = b.tileSurvOff[c * b.tileStride + tile] + (uint32_t)(ex >> 40);
    uint64_t* next = b.lis[s.cur ^ 1u] + c * b.lisStride + b.levelOff[l];
    uint64_t* opos = b.opos + c * b.nodeStride;
    uint64_t* stream = b.stream + c * b.streamStride;
    for (int k = 0; k < 4; k++) {
      if (!it[k].bits)
        continue;
      if (it[k].sig) {
        opos[it[k].id] = bitpos;
        put_bits(stream, bitpos, 1, 1, s.budget);
      }
      else
        next[rank++] = it[k].packed;
      bitpos += it[k].bits;
    }
  }
}

// ------------------------------------------------------------------------------------------
// 2D coder: the type-I set -- everything outside the coarsest approximation band -- is tested at the
// end of every sorting pass (SPECK2D_INT.cpp:44-98,149-186).  When it is significant, the three
// subbands of its level are tested one after the other (a significant one splits on the spot: its
// split is written by k_split_emit like that of a list entry, an insignificant one joins the list
// of its level), and what is left of it is tested next -- a test that is implied when none of the
// three was significant.  A few bits per plane: one thread per chunk, after the lists' positions
// are known (k_list_scan) and before the splits are written.
// ------------------------------------------------------------------------------------------
__global__ void k_enc_iphase(EncBuffers b, int p)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b.nchunks)
    return;
  EncState& s = b.st[c];
  ACTIVE_OR_RETURN(s, p);
  uint32_t iPart = s.iPart;
  if (iPart == 0)
    return;
  const Tree& t = b.tree;
  const int8_t* M = b.M + c * b.nodeStride;
  const uint32_t* E = b.E + c * b.nodeStride;
  uint64_t* stream = b.stream + c * b.streamStride;
  const uint64_t baseLIS = s.rec[p].baseLIS;
  uint64_t pos = baseLIS + s.lisBits;
  bool need = true;
  while (iPart > 0) {
    if (need) {
      // the set's largest coefficient: that of the subbands it still holds
      int mi = -1;
      for (uint32_t k = (b.iLevels - iPart) * 3u; k < b.iLevels * 3u; k++)
        if (b.iRoots[k] != ~0ull)
          mi = max(mi, (int)M[flat_id(t, unpack_node(b.iRoots[k]))]);
      const bool isig = mi >= p;
      put_bits(stream, pos, isig ? 1u : 0u, 1, s.budget);
      pos++;
      if (!isig)
        break;
    }
    uint32_t counter = 0;
    for (uint32_t j = 0; j < 3; j++) {
      const uint64_t root = b.iRoots[(b.iLevels - iPart) * 3u + j];
      if (root == ~0ull)
        continue;
      const uint32_t id = flat_id(t, unpack_node(root));
      if (M[id] >= p) {
        put_bits(stream, pos, 1u, 1, s.budget);
        b.opos[c * b.nodeStride + id] = pos;
        pos += 1u + E[id];
        counter++;
      }
      else {
        const uint64_t rel = pos - baseLIS;
        const uint32_t slot = b.levelSlot[iPart];
        if (slot != 0xff && rel < (uint64_t)b.maskWords * 64) {
          const uint32_t k = atomicAdd(&s.bornCount, 1u);
          b.bornPacked[c * b.bornStride + k] = root;
          b.bornPosLev[c * b.bornStride + k] = ((uint64_t)iPart << 48) | rel;
          atomic_or64(b.mask + c * b.maskStride + (size_t)slot * b.maskWords + (rel >> 6), 1ull << (rel & 63));
        }
        pos++;
      }
    }
    iPart--;
    need = counter != 0;
  }
  s.iPart = iPart;
  s.lisBits = pos - baseLIS;
}

// ------------------------------------------------------------------------------------------
// k_split_emit: every set whose msb equals the plane splits now
// ------------------------------------------------------------------------------------------
// Nodes are bucketed by the plane at which they split, so a plane only visits its own splitting sets
// instead of scanning every node: k_pyramid counts them per plane, k_bucket_scan turns the counts into
// offsets, k_chain deals the nodes out.
__global__ void k_bucket_scan(EncBuffers b)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b.nchunks)
    return;
  EncState& s = b.st[c];
  if (!s.active)
    return;
  uint32_t off = 0;
  for (int p = 0; p < kMaxPlanes; p++) {
    s.bucketOff[p] = off;
    s.bucketCur[p] = off;
    off += s.bucketCnt[p];
  }
}

constexpr int kSplitBlocks = 512;

template <bool ANY>
__device__ __forceinline__ void split_emit_node(const EncBuffers& b, uint32_t c, EncState& s, int p,
                                                uint32_t id, bool have);

// (ANY: see pyramid_node; 73 registers without the general path, 87 with)
template <bool ANY>
__global__ void __launch_bounds__(kNodeBlock) k_split_emit(EncBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  EncState& s = b.st[c];
  ACTIVE_OR_RETURN(s, p);
  const uint32_t cnt = s.bucketCnt[p], off = s.bucketOff[p];
  const uint32_t* bucket = b.bucket + c * b.nodeStride + off;
  // (every lane of a wavefront takes part in every round: the birth slots of a wavefront's sets
  //  are claimed with one atomic)
  for (uint32_t k0 = blockIdx.x * kNodeBlock; k0 < cnt; k0 += gridDim.x * kNodeBlock) {
    const uint32_t k = k0 + threadIdx.x;
    const bool have = k < cnt;
    split_emit_node<ANY>(b, c, s, p, have ? bucket[k] : 0u, have);
  }
}

template <bool ANY>
__device__ __forceinline__ void split_emit_node(const EncBuffers& b, uint32_t c, EncState& s, int p,
                                                uint32_t id, bool have)
{
  const int8_t* M = b.M + c * b.nodeStride;
  const Tree& t = b.tree;
  Node nd;
  node_from_flat(t, id, nd);
  const uint32_t* E = b.E + c * b.nodeStride;
  const int8_t* msb = b.msb + c * b.pixStride;

  // the list entry that started this chain of splits, and the bits between its split and ours
  // (k_chain; one step from the parent for the sets it does not store)
  const uint32_t* koff = b.koff + c * b.nodeStride;
  const uint64_t* chain = b.chain + c * b.nodeStride;
  const Grid& g0 = t.grids[nd.grid];
  uint32_t topid = id, off = 0;
  if (g0.depth != 0) {
    if (g0.depth + 1 != t.roots[g0.root].Dmax) {
      const uint64_t v = chain[id];
      topid = (uint32_t)v;
      off = (uint32_t)(v >> 32);
    }
    else {
      const uint32_t pid = flat_id(t, node_parent(t, nd));
      if (M[pid] == p) {
        const uint64_t v = chain[pid];
        topid = (uint32_t)v;
        off = (uint32_t)(v >> 32) + koff[id];
      }
    }
  }
  uint64_t pos = b.opos[c * b.nodeStride + topid] + 1 + off;

  const uint64_t* sign = b.sign + c * b.signStride;
  uint64_t* stream = b.stream + c * b.streamStride;
  const uint64_t baseLIS = s.rec[p].baseLIS;
  bool found = false;
  uint64_t acc = 0;      // contiguous run of bits being assembled
  int nacc = 0;
  uint64_t accpos = pos;
  const Grid& g = t.grids[nd.grid];
  // ---- births first: how many children of this set stay insignificant sets (they get consecutive
  //      slots of the chunk's birth records; one atomic per wavefront claims them)
  const bool isOct = !ANY || (g.kind & kGridOct) != 0;
  const bool isLeafSet = isOct && g.depth + 1 == t.roots[g.root].Dmax;
  OctKids k;
  KidRegs kr;
  kr.n = 0;
  uint32_t kidlev = 0, slot = 0xff, nborn = 0;
  const uint64_t maskBits = (uint64_t)b.maskWords * 64;
  if (have && isOct && !isLeafSet) {
    oct_load(t, g, t.roots[g.root], nd, M, E, msb, k);
    kidlev = node_level(t, nd) + 3;
    slot = b.levelSlot[kidlev];
    if (!k.deepest && slot != 0xff) {
      uint64_t q2 = pos;
      bool fnd = false;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool coded = fnd || j != 7;
        const bool sig = coded ? (k.m[j] == p) : true;
        q2 += coded ? 1u : 0u;
        if (sig) {
          fnd = true;
          q2 += k.e[j];
        }
        else if (q2 - 1 - baseLIS < maskBits)
          nborn++;
      }
    }
  }
  else if (ANY && have && !isOct) {
    kid_regs_load(t, nd, M, E, msb, kr);
    kidlev = kr.kb.kidlev;
    slot = b.levelSlot[kidlev];
    uint64_t q2 = pos;
    bool fnd = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (j >= kr.n)
        break;
      const bool coded = fnd || (j + 1 != kr.n);
      const bool sig = coded ? (kr.m[j] == p) : true;
      q2 += coded ? 1u : 0u;
      if (sig) {
        fnd = true;
        q2 += ((kr.pixels >> j) & 1u) ? 1u : kr.e[j];
      }
      else if (!((kr.pixels >> j) & 1u) && slot != 0xff && q2 - 1 - baseLIS < maskBits)
        nborn++;
    }
  }
  uint32_t kslot;
  {
    const uint32_t inc = wave_inclusive_scan<uint32_t>(nborn);
    const uint32_t tot = __shfl(inc, 63, 64);
    uint32_t base = 0;
    if (tot) {
      if ((threadIdx.x & 63) == 63)
        base = atomicAdd(&s.bornCount, tot);
      base = __shfl(base, 63, 64);
    }
    kslot = base + inc - nborn;
  }
  if (!have)
    return;
  if (isOct) {  // same loop as below with the 8 children in registers
    if (isLeafSet) {   // a leaf set: everything is in its descriptor
      const uint32_t desc = b.leafDesc[c * b.nodeStride + id];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool coded = found || j != 7;
        const uint32_t sig = coded ? (desc >> j) & 1u : 1u;
        if (coded) {
          acc |= (uint64_t)sig << nacc;
          nacc++;
        }
        if (sig) {
          found = true;
          acc |= (uint64_t)((desc >> (8 + j)) & 1u) << nacc;
          nacc++;
        }
      }
      put_bits(stream, accpos, acc, nacc, s.budget);
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const bool coded = found || j != 7;
      const bool sig = coded ? (k.m[j] == p) : true;
      if (coded) {
        acc |= (uint64_t)sig << nacc;
        nacc++;
        pos++;
      }
      if (sig) {
        found = true;
        if (k.deepest) {
          const uint32_t ridx = oct_kid(k, j);
          acc |= ((sign[ridx >> 6] >> (ridx & 63)) & 1ull) << nacc;
          nacc++;
          pos++;
        }
        else {
          put_bits(stream, accpos, acc, nacc, s.budget);
          pos += k.e[j];
          acc = 0;
          nacc = 0;
          accpos = pos;
        }
      }
      else if (!k.deepest) {
        const uint64_t rel = pos - 1 - baseLIS;
        if (slot != 0xff && rel < (uint64_t)b.maskWords * 64) {
          Node kn;
          kn.grid = (uint16_t)(nd.grid + 1);
          kn.i[0] = (uint16_t)(2u * nd.i[0] + (uint32_t)(j & 1));
          kn.i[1] = (uint16_t)(2u * nd.i[1] + (uint32_t)((j >> 1) & 1));
          kn.i[2] = (uint16_t)(2u * nd.i[2] + (uint32_t)(j >> 2));
          b.bornPacked[c * b.bornStride + kslot] = entry_with_msb(t, pack_node(kn), k.m[j]);
          b.bornPosLev[c * b.bornStride + kslot] = ((uint64_t)kidlev << 48) | rel;
          kslot++;
          atomic_or64(b.mask + c * b.maskStride + (size_t)slot * b.maskWords + (rel >> 6),
                      1ull << (rel & 63));
        }
      }
    }
    put_bits(stream, accpos, acc, nacc, s.budget);
    return;
  }
  if constexpr (!ANY)
    return;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    if (j >= kr.n)
      break;
    const bool coded = found || (j + 1 != kr.n);
    const bool sig = coded ? (kr.m[j] == p) : true;
    if (coded) {
      acc |= (uint64_t)sig << nacc;
      nacc++;
      pos++;
    }
    if (sig) {
      found = true;
      if ((kr.pixels >> j) & 1u) {
        const uint32_t ridx = kid_pixel_raster(t, nd, kr.kb, (uint32_t)j);
        acc |= ((sign[ridx >> 6] >> (ridx & 63)) & 1ull) << nacc;
        nacc++;
        pos++;
      }
      else {  // the child writes its own split; flush what we have and skip over it
        put_bits(stream, accpos, acc, nacc, s.budget);
        pos += kr.e[j];
        acc = 0;
        nacc = 0;
        accpos = pos;
      }
    }
    else if (!((kr.pixels >> j) & 1u)) {  // insignificant set: joins LIS[kidlev] in stream order
      const uint64_t rel = pos - 1 - baseLIS;
      if (slot != 0xff && rel < (uint64_t)b.maskWords * 64) {
        b.bornPacked[c * b.bornStride + kslot] = entry_with_msb(t, kid_packed(kr.kb, (uint32_t)j), kr.m[j]);
        b.bornPosLev[c * b.bornStride + kslot] = ((uint64_t)kidlev << 48) | rel;
        kslot++;
        atomic_or64(b.mask + c * b.maskStride + (size_t)slot * b.maskWords + (rel >> 6),
                    1ull << (rel & 63));
      }
    }
  }
  put_bits(stream, accpos, acc, nacc, s.budget);
}

// popcount prefix of every birth mask (one block per mask slot and chunk)
__global__ void __launch_bounds__(kThreads) k_mask_scan(EncBuffers b, int p)
{
  const uint32_t c = blockIdx.y, slot = blockIdx.x;
  EncState& s = b.st[c];
  ACTIVE_OR_RETURN(s, p);
  __shared__ uint32_t sm[kThreads / 64 + 1];
  const uint64_t nbits = min(s.lisBits, (uint64_t)b.maskWords * 64);
  const uint32_t nwords = (uint32_t)((nbits + 63) / 64);
  const uint64_t* mask = b.mask + c * b.maskStride + (size_t)slot * b.maskWords;
  uint32_t* pre = b.maskPrefix + c * b.prefStride + (size_t)slot * b.prefWords;   // (one word per four mask words)
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nwords; base += kThreads * 4) {
    uint32_t v[4], tsum = 0;
    for (int k = 0; k < 4; k++) {
      const uint32_t i = base + threadIdx.x * 4 + k;
      v[k] = i < nwords ? (uint32_t)__popcll(mask[i]) : 0u;
      tsum += v[k];
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<uint32_t>(tsum, sm, &total) + carry;
    if (base + threadIdx.x * 4 < nwords)
      pre[base / 4 + threadIdx.x] = ex;   // (the thread's four words are one group)
    carry += total;
  }
  if (threadIdx.x == 0)
    s.bornTot[b.slotLevel[slot]] = carry;
}

__global__ void __launch_bounds__(kThreads) k_born_place(EncBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const EncState& s = b.st[c];
  ACTIVE_OR_RETURN(s, p);
  const uint32_t nx = s.cur ^ 1u;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < s.bornCount;
       k += gridDim.x * blockDim.x) {
    const uint64_t pl = b.bornPosLev[c * b.bornStride + k];
    const uint32_t lev = (uint32_t)(pl >> 48);
    const uint64_t rel = pl & ((1ull << 48) - 1);
    const uint32_t slot = b.levelSlot[lev];
    const uint32_t wi = (uint32_t)(rel >> 6);
    const size_t mo = c * b.maskStride + (size_t)slot * b.maskWords + wi;
    uint32_t rank = b.maskPrefix[c * b.prefStride + (size_t)slot * b.prefWords + (wi >> 2)] +
                    (uint32_t)__popcll(b.mask[mo] & ((1ull << (rel & 63)) - 1ull));
    for (uint32_t j = 1; j <= (wi & 3u); j++)   // (the words of the group in front of this one: the same 32 bytes)
      rank += (uint32_t)__popcll(b.mask[mo - j]);
    b.lis[nx][c * b.lisStride + b.levelOff[lev] + s.listLen[nx][lev] + rank] =
        b.bornPacked[c * b.bornStride + k];
  }
}


// ------------------------------------------------------------------------------------------
// k_emit_pixels: LIP-scan bits and refinement bits of every processed plane, raster order.
// Each tile's bits of one (plane, phase) are contiguous in the stream: stage them in LDS and OR
// whole words out.
// ------------------------------------------------------------------------------------------
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_emit_pixels(EncBuffers b)
{
  const uint32_t c = blockIdx.y;
  const EncState& s = b.st[c];
  if (!s.active || s.nbp == 0)
    return;
  constexpr int kWords = kPixTile * 2 / 64 + 2;
  __shared__ unsigned long long lipw[kWords], refw[kWords];
  __shared__ uint32_t sm[kThreads / 64 + 1];
  const uint32_t n = b.tree.nvals;
  const uint32_t tile = blockIdx.x;
  const uint32_t i0 = tile * kPixTile + threadIdx.x * kPixPer;
  int m[kPixPer], bp[kPixPer];
  load_pix(b.bplane + c * b.pixStride, i0, n, bp);
  int bpmax = -1;   // no sample of the thread is in the LIP or significant at planes >= bpmax
#pragma unroll
  for (int k = 0; k < kPixPer; k++)
    bpmax = max(bpmax, bp[k]);
  // A sample born at plane b gives LIP bits on planes below b and refinement bits below its msb <= b: nothing at all
  // when b is not above the last plane coded.  Such a thread -- at 2 bits per sample half of them: the fine subbands
  // whose sets never split inside the budget -- loads neither its coefficients nor their msbs (round 5, second
  // session: 5 of the 6 bytes a sample costs here).
  const bool live = bpmax > s.plast;
  CT cf[kPixPer];
  uint32_t sg = 0;
  const CT* coef = reinterpret_cast<const CT*>(b.coef) + c * b.coefStride;
  const uint64_t* sign = b.sign + c * b.signStride;
#pragma unroll
  for (int k = 0; k < kPixPer; k++) {
    m[k] = -1;
    cf[k] = (CT)0;
  }
  if (live) {
    load_pix(b.msb + c * b.pixStride, i0, n, m);
    if (i0 < n)   // kPixPer divides 64: the signs of the thread's samples sit in one word
      sg = (uint32_t)(sign[i0 >> 6] >> (i0 & 63));
#pragma unroll
    for (int k = 0; k < kPixPer; k++)
      cf[k] = (i0 + k < n) ? coef[i0 + k] : (CT)0;
  }
  else
    bpmax = -1;   // (takes no part in the plane loop)
  // The thread's 16 samples BIT-SLICED (round 3): bit k of M[j] / B[j] = bit j of msb + 1 / birth plane + 1
  // of sample k.  "msb above plane p", "msb equal to p", "born above p" for all 16 samples are then a
  // dozen logic operations on 16-bit masks per plane instead of a chain of compares, shifts and
  // adds per sample and plane.  (Measured on MI355X, 64 chunks of 256^3: the kernel takes 2.3 ms to
  // load its samples and 3.7 ms for the plane loop; two planes per round, plain stores for the
  // stream words a tile owns, LDS-only barriers and this took the loop from 4.1 ms -- what it waits
  // for is the block scan and the three barriers of a round at four workgroups per CU.)
  uint32_t M[7], B[7];
#pragma unroll
  for (int j = 0; j < 7; j++) {
    M[j] = B[j] = 0;
#pragma unroll
    for (int k = 0; k < kPixPer; k++) {
      M[j] |= (((uint32_t)(m[k] + 1) >> j) & 1u) << k;
      B[j] |= (((uint32_t)(bp[k] + 1) >> j) & 1u) << k;
    }
  }
  // masks of the samples whose value (given bit-sliced in X) is above / equal to t - 1, t in 1 .. 64
  auto above_equal = [](const uint32_t (&X)[7], uint32_t t, uint32_t& gt, uint32_t& eq) {
    gt = 0;
    eq = 0xffffu;
#pragma unroll
    for (int j = 6; j >= 0; j--) {
      if ((t >> j) & 1u)   // (uniform)
        eq &= X[j];
      else {
        gt |= eq & X[j];
        eq &= ~X[j];
      }
    }
  };
  // bits of `val` at the set positions of `mask`, packed (16-bit pext, a nibble at a time through a table)
  __shared__ uint8_t pextLut[16][16];
  {
    const uint32_t mk = threadIdx.x >> 4, vl = threadIdx.x & 15u;
    uint32_t r = 0, o = 0;
    for (int i = 0; i < 4; i++)
      if ((mk >> i) & 1u) {
        r |= ((vl >> i) & 1u) << o;
        o++;
      }
    pextLut[mk][vl] = (uint8_t)r;   // (kThreads == 256: one entry per thread)
  }
  static_assert(kThreads == 256, "one pext table entry per thread");
  auto pext16 = [&](uint32_t val, uint32_t mask) -> uint32_t {
    uint32_t r = 0, sh = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t mn = (mask >> (4 * i)) & 15u;
      r |= (uint32_t)pextLut[mn][(val >> (4 * i)) & 15u] << sh;
      sh += (uint32_t)__popc(mn);
    }
    return r;
  };
  // the LIP scan's and the refinement pass's bits of the thread's samples on plane pl
  auto plane_bits = [&](int pl, bool withRef, uint32_t& lbits, uint32_t& lval, uint32_t& rbits, uint32_t& rval) {
    uint32_t gtM, eqM, gtB, eqB;
    above_equal(M, (uint32_t)pl + 1u, gtM, eqM);
    above_equal(B, (uint32_t)pl + 1u, gtB, eqB);
    const uint32_t inLip = gtB & ~gtM;        // born above the plane, msb not above it: one '0' or '1' + sign
    lbits = lval = rbits = rval = 0;
    if (inLip) {
      const uint32_t nl = (uint32_t)__popc(inLip);
      uint32_t tok = pext16(eqM, inLip);      // token i is '1' (found significant) ...
      uint32_t sgn = pext16(sg, inLip);       // ... and then its sign follows
      lbits = nl;
      lval = tok;
      // a sign bit behind every '1', from the last token down (what lies below stays where it is)
      for (uint32_t rest = tok; rest;) {
        const uint32_t i = 31u - (uint32_t)__clz((int)rest);   // token index
        rest &= ~(1u << i);
        const uint32_t low = lval & ((2u << i) - 1u);             // tokens 0 .. i as they stand
        lval = low | (((sgn >> i) & 1u) << (i + 1)) | ((lval >> (i + 1)) << (i + 2));
        lbits++;
      }
    }
    if (withRef && gtM) {
      uint32_t cbit = 0;   // bit pl of the samples' magnitudes
#pragma unroll
      for (int k = 0; k < kPixPer; k++)
        cbit |= (uint32_t)((cf[k] >> pl) & 1) << k;
      rbits = (uint32_t)__popc(gtM);
      rval = pext16(cbit, gtM);
    }
  };
  const uint32_t* cnt = b.pixCnt + c * b.pixCntStride;
  const uint32_t* off = b.pixOff + c * b.pixCntStride;
  uint64_t* stream = b.stream + c * b.streamStride;
  // the tile's counts and stream positions of every plane, fetched at once: read plane by plane
  // they put a dependent global load in front of every iteration of the loop below
  __shared__ uint32_t tcnt[2 * kMaxPlanes];
  __shared__ uint64_t tbase[2 * kMaxPlanes];
  const int nbp = s.nbp, plast = s.plast;
  for (int t = threadIdx.x; t < 2 * nbp; t += blockDim.x) {
    const int pp = t >> 1;
    const bool ref = (t & 1) != 0;
    tcnt[t] = (ref && !s.rec[pp].didREF) ? 0u : cnt[(size_t)t * b.nPixTiles + tile];
    tbase[t] = (ref ? s.rec[pp].baseREF : s.rec[pp].baseLIP) + off[(size_t)t * b.nPixTiles + tile];
  }
  const uint64_t budget = s.budget;
  // words at or past limitWord hold no kept bit (budget may be ~0: no overflow here)
  const uint64_t limitWord = (budget >> 6) + ((budget & 63) ? 1 : 0);
  __shared__ unsigned long long lipw2[kWords], refw2[kWords];   // the second plane of a round
  for (int w = threadIdx.x; w < kWords; w += blockDim.x)
    lipw[w] = refw[w] = lipw2[w] = refw2[w] = 0;
  __shared__ uint64_t sm64[kThreads / 64 + 1];
  __syncthreads();
  // Two planes per round (round 3): their four bit counts share one block scan, which halves the
  // barriers per plane.
  for (int p = nbp - 1; p >= plast; p -= 2) {
    const int q = p - 1;                       // the round's second plane, if there is one
    const bool two = q >= plast;
    const uint32_t nlipA = tcnt[p * 2], nrefA = tcnt[p * 2 + 1];
    const uint32_t nlipB = two ? tcnt[q * 2] : 0u, nrefB = two ? tcnt[q * 2 + 1] : 0u;
    if ((nlipA | nrefA | nlipB | nrefB) == 0)
      continue;  // uniform across the block
    uint32_t lbA = 0, lvA = 0, rbA = 0, rvA = 0, lbB = 0, lvB = 0, rbB = 0, rvB = 0;   // at most 2 and 1 bits per sample
    if (q < bpmax) {   // (a sample takes part from its birth plane on: most threads hold none yet)
      plane_bits(p, nrefA != 0, lbA, lvA, rbA, rvA);
      if (two)
        plane_bits(q, nrefB != 0, lbB, lvB, rbB, rvB);
    }
    if (!nrefA)
      rbA = 0;
    if (!nrefB)
      rbB = 0;
    uint64_t total;
    const uint64_t ex = block_exclusive_scan_lds<uint64_t>((uint64_t)lbA | ((uint64_t)rbA << 16) | ((uint64_t)lbB << 32) |
                                                           ((uint64_t)rbB << 48), sm64, &total);
    const uint64_t lipBaseA = tbase[p * 2], refBaseA = tbase[p * 2 + 1];
    const uint64_t lipBaseB = two ? tbase[q * 2] : 0ull, refBaseB = two ? tbase[q * 2 + 1] : 0ull;
    auto deposit = [&](unsigned long long* w, uint32_t val, uint32_t nbits, uint32_t at) {
      atomicOr(&w[at >> 6], (unsigned long long)val << (at & 63));
      if ((at & 63) + nbits > 64)
        atomicOr(&w[(at >> 6) + 1], (unsigned long long)val >> (64 - (at & 63)));
    };
    if (lvA)
      deposit(lipw, lvA, lbA, (uint32_t)(ex & 0xffffu) + (uint32_t)(lipBaseA & 63));
    if (rvA && nrefA)
      deposit(refw, rvA, rbA, (uint32_t)((ex >> 16) & 0xffffu) + (uint32_t)(refBaseA & 63));
    if (lvB)
      deposit(lipw2, lvB, lbB, (uint32_t)((ex >> 32) & 0xffffu) + (uint32_t)(lipBaseB & 63));
    if (rvB && nrefB)
      deposit(refw2, rvB, rbB, (uint32_t)(ex >> 48) + (uint32_t)(refBaseB & 63));
    LDS_ONLY_BARRIER();
    // The tile's bits of a pass are bits [base, base + n) of the stream: only the first and the last
    // word of that range are shared with the neighbouring tiles (or passes) -- those are OR-ed in
    // atomically, the words in between are this workgroup's alone and are stored (the stream starts
    // out as zeros; before round 3 every word went through an L2 atomic).
    auto flush = [&](unsigned long long* w, uint64_t base, uint32_t n) {
      if (n == 0)
        return;
      const uint64_t firstW = base >> 6, lastW = (base + n - 1) >> 6;
      for (int i = threadIdx.x; i < kWords; i += blockDim.x) {
        uint64_t v = w[i];
        if (v == 0)
          continue;
        w[i] = 0;   // clean for the round after next
        const uint64_t gw = firstW + i;
        if (gw < limitWord) {
          if (gw == limitWord - 1 && (budget & 63))
            v &= (1ull << (budget & 63)) - 1;
          if (gw == firstW || gw >= lastW)
            atomic_or64(stream + gw, v);
          else
            stream[gw] = v;
        }
      }
    };
    flush(lipw, lipBaseA, nlipA);
    flush(refw, refBaseA, nrefA);
    if (two) {
      flush(lipw2, lipBaseB, nlipB);
      flush(refw2, refBaseB, nrefB);
    }
    LDS_ONLY_BARRIER();   // (not __syncthreads(): that would wait for the words on their way to the stream)
  }
}


// stream length and the fixed-rate retry test (SPECK_INT.cpp:264-282, SPECK_FLT.cpp:530-538)
__global__ void k_enc_finalize(EncBuffers b, uint64_t raw_budget, int rate_mode, int wide_pass)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b.nchunks)
    return;
  EncState& s = b.st[c];
  CoderState& cs = b.cst[c];
  if (cs.is_const) {
    cs.stream_len = 17;
    cs.need_retry = 0;
    return;
  }
  if (!s.active)
    return;
  const uint64_t keep = min(s.total_bits, s.budget);
  const uint64_t payload = (keep + 7) / 8;
  cs.stream_len = 17 + 9 + payload;
  cs.nbp = s.nbp;
  cs.total_bits = s.total_bits;
  if (rate_mode)   // (src/SPECK_FLT.cpp:530-538; other modes choose the width before coding)
    cs.need_retry = (!wide_pass && (9 + payload) * 8 < raw_budget) ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------
// The planes of a batch that can hold work (EncPlanHost::d_bound): out[1] = the largest plane count of its chunks,
// out[0] = the lowest plane any chunk can reach -- where the bits of the pixel passes alone (the census: LIP scan
// and refinement bits of every plane down to it) have used up the budget, the loop of src/SPECK_INT.cpp:146-158
// has ended for that chunk whatever its sorting passes took.  One workgroup; thread = chunk.
__global__ void __launch_bounds__(kThreads) k_enc_bound(EncBuffers b, uint32_t* out)
{
  __shared__ uint32_t lo, hi;
  if (threadIdx.x == 0) {
    lo = 0xffffffffu;
    hi = 0;
  }
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < b.nchunks; c += blockDim.x) {
    const EncState& s = b.st[c];
    if (!s.active || s.nbp <= 0)
      continue;
    int p = s.nbp - 1;
    uint64_t cum = 0;
    for (; p > 0; p--) {
      cum += s.lipTot[p] + s.refTot[p];
      if (cum >= s.budget)   // (the budget of a mode without one is ~0: never)
        break;
    }
    atomicMin(&lo, (uint32_t)p);
    atomicMax(&hi, (uint32_t)s.nbp);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = lo == 0xffffffffu ? 0u : lo;
    out[1] = hi;
  }
}

// node blocks a workgroup of k_pyramid / k_chain takes: as many as leave the chip four workgroups per CU and more
static uint32_t node_blocks_per_group(uint32_t nb, uint32_t nc)
{
  static const uint32_t perEnv = tune_getenv("SPERR_HIP_NODE_PER") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_NODE_PER")) : 0u;
  if (perEnv)
    return std::min<uint32_t>(std::max(1u, perEnv), (uint32_t)kNodePerMax);
  uint32_t per = 1;
  while (per < (uint32_t)kNodePerMax && (uint64_t)nb * nc / (2 * per) >= 4096)
    per *= 2;
  return per;
}

int launch_speck_encode_head(hipStream_t stream, const EncBuffers& b, const EncPlanHost& plan,
                             uint64_t raw_budget, bool rate_mode, bool wide_pass)
{
  uint64_t budget = ~0ull;
  if (raw_budget != 0) {  // SPECK_INT.cpp:48-58
    budget = raw_budget;
    while (budget % 8)
      budget++;
  }
  const uint32_t nc = b.nchunks;
  const dim3 perChunk((nc + 63) / 64);
  LAUNCH_K(k_enc_state_init, perChunk, dim3(64), 0, stream, b, plan.d_initLIS,
                     plan.d_initLen, budget, wide_pass ? 1 : 0);
  const int maxPlanes = wide_pass ? kMaxPlanes : 32;
  const bool sideCensus = plan.side && plan.evFork && plan.evJoin && b.tree.maxDepth >= 2;
  for (int d = (int)b.tree.maxDepth - 1; d >= 0; d--) {
    const uint32_t nb = plan.depthBlockOff[d + 1] - plan.depthBlockOff[d];
    if (nb) {
      const uint32_t per = node_blocks_per_group(nb, nc);
      if (b.tree.flags & kTreeAllOct)
        LAUNCH_K(k_pyramid<false>, dim3((nb + per - 1) / per, nc), dim3(kNodeBlock), 0, stream, b,
                 plan.d_depthBlocks + plan.depthBlockOff[d], nb, per);
      else
        LAUNCH_K(k_pyramid<true>, dim3((nb + per - 1) / per, nc), dim3(kNodeBlock), 0, stream, b,
                 plan.d_depthBlocks + plan.depthBlockOff[d], nb, per);
    }
    if (sideCensus && d == 0) {
      // every sample's birth plane is known now (samples are born at any depth: the roots' trees differ in
      // height, and an odd set has a single sample for a child): the census needs nothing else
      // (k_enc_planes_setup's plane count only bounds its output rows: see k_census)
      HIP_CHECK(hipEventRecord(plan.evFork, stream));
      HIP_CHECK(hipStreamWaitEvent(plan.side, plan.evFork, 0));
      LAUNCH_K(k_census, dim3(b.nPixTiles, nc), dim3(kThreads), 0, plan.side, b, maxPlanes);
    }
  }
  LAUNCH_K(k_bucket_scan, perChunk, dim3(64), 0, stream, b);   // (k_pyramid has counted the splitting sets per plane)
  for (int d = 0; d < (int)b.tree.maxDepth; d++) {
    const uint32_t nb = plan.depthBlockOff[d + 1] - plan.depthBlockOff[d];
    if (nb) {
      const uint32_t per = node_blocks_per_group(nb, nc);
      LAUNCH_K(k_chain, dim3((nb + per - 1) / per, nc), dim3(kNodeBlock), 0, stream, b,
               plan.d_depthBlocks + plan.depthBlockOff[d], nb, per);
    }
  }
  LAUNCH_K(k_enc_planes_setup, perChunk, dim3(64), 0, stream, b);
  if (sideCensus) {
    HIP_CHECK(hipEventRecord(plan.evJoin, plan.side));
    HIP_CHECK(hipStreamWaitEvent(stream, plan.evJoin, 0));
  }
  else
    LAUNCH_K(k_census, dim3(b.nPixTiles, nc), dim3(kThreads), 0, stream, b, maxPlanes);
  LAUNCH_K(k_census_scan, dim3(maxPlanes * 2, nc), dim3(kThreads), 0, stream, b);
  if (plan.d_bound && plan.h_bound && plan.evBound) {
    LAUNCH_K(k_enc_bound, dim3(1), dim3(kThreads), 0, stream, b, plan.d_bound);
    HIP_CHECK(hipMemcpyAsync(plan.h_bound, plan.d_bound, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipEventRecord(plan.evBound, stream));
    // (something behind the event: under `rocprofv3 --pmc`, which runs one kernel at a time, the copy in front of an
    //  event that is the LAST thing in its queue was never seen to complete -- the counter pass of a 64-chunk volume
    //  hung in hipEventSynchronize until its timeout; the decoder's live check, the same copy + event with planes
    //  queued behind it, never did.  A launch that does nothing costs 5 us once per batch.)
    LAUNCH_K(k_plane_turn, perChunk, dim3(64), 0, stream, b, -1, -1);
  }
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_speck_encode_planes(hipStream_t stream, const EncBuffers& b, const EncPlanHost& plan,
                               uint64_t raw_budget, bool rate_mode, bool wide_pass)
{
  const uint32_t nc = b.nchunks;
  const dim3 perChunk((nc + 63) / 64);
  const int maxPlanes = wide_pass ? kMaxPlanes : 32;
  int pTop = maxPlanes, pLow = 0;   // planes pTop - 1 .. pLow are launched
  if (plan.d_bound && plan.h_bound && plan.evBound) {
    HIP_CHECK(hipEventSynchronize(plan.evBound));
    pLow = std::min<int>(maxPlanes - 1, (int)plan.h_bound[0]);
    pTop = std::max(pLow + 1, std::min<int>(maxPlanes, (int)plan.h_bound[1]));
  }
  const uint32_t bornBlocks = (plan.nsets + kThreads - 1) / kThreads;
  // grid caps of the per-plane sweeps (workgroups over all chunks of the batch; a workgroup strides over its tiles): in a
  // light plane a launch costs what its idle workgroups take to dispatch.  Round 5: 4096 / 2048 from 16 chunks on, the wide
  // grids (16384 / 4096) below.  Round 6, with the node kernels' registers halved and four parts side by side: 1536 / 768
  // for every batch -- 64 chunks 150.0 -> 153.3 to 154.7 GB/s (1280 / 640: 154.7; 2048 / 1024: 152.4 to 152.9; 3072 / 1536:
  // 151.0; 8192 / 4096: 144.6), 8 chunks 85.6 -> 90.3 to 91.7 (6.23 -> 5.94 ms), one chunk the same 2.88 ms
  // (profiles/r6_enc_grid_ab.txt)
  static const char* wideEnv = tune_getenv("SPERR_HIP_ENC_WIDE_GRID");
  static const char* smallEnv = tune_getenv("SPERR_HIP_ENC_SMALL_GRID");
  const uint32_t wideCap = wideEnv ? (uint32_t)atoi(wideEnv) : 1536u;
  const uint32_t smallCap = smallEnv ? (uint32_t)atoi(smallEnv) : 768u;
  LAUNCH_K(k_plane_turn, perChunk, dim3(64), 0, stream, b, -1, pTop - 1);
  for (int p = pTop - 1; p >= pLow; p--) {
    LAUNCH_K(k_list_count, dim3(capped_blocks(b.nListTiles, nc, smallCap), nc), dim3(kThreads), 0, stream, b, p,
             p < pTop - 1 ? 1 : 0);
    LAUNCH_K(k_list_scan, dim3(nc), dim3(kThreads), 0, stream, b, p);
    LAUNCH_K(k_list_apply, dim3(capped_blocks(b.nListTiles, nc, wideCap), nc), dim3(kThreads), 0, stream, b, p);
    if (b.tree.flags & kTree2D)
      LAUNCH_K(k_enc_iphase, perChunk, dim3(64), 0, stream, b, p);
    if (b.tree.flags & kTreeAllOct)
      LAUNCH_K(k_split_emit<false>, dim3(capped_blocks(kSplitBlocks, nc, smallCap), nc), dim3(kNodeBlock), 0, stream, b, p);
    else
      LAUNCH_K(k_split_emit<true>, dim3(capped_blocks(kSplitBlocks, nc, smallCap), nc), dim3(kNodeBlock), 0, stream, b, p);
    if (b.nSlots) {
      LAUNCH_K(k_mask_scan, dim3(b.nSlots, nc), dim3(kThreads), 0, stream, b, p);
      LAUNCH_K(k_born_place, dim3(capped_blocks(bornBlocks, nc, smallCap), nc), dim3(kThreads), 0, stream, b, p);
    }
    LAUNCH_K(k_plane_turn, perChunk, dim3(64), 0, stream, b, p, p - 1);
  }
  if (wide_pass)
    LAUNCH_K(k_emit_pixels<uint64_t>, dim3(b.nPixTiles, nc), dim3(kThreads), 0, stream,
                       b);
  else
    LAUNCH_K(k_emit_pixels<uint32_t>, dim3(b.nPixTiles, nc), dim3(kThreads), 0, stream,
                       b);
  LAUNCH_K(k_enc_finalize, perChunk, dim3(64), 0, stream, b, raw_budget,
                     rate_mode ? 1 : 0, wide_pass ? 1 : 0);
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_speck_encode(hipStream_t stream, const EncBuffers& b, const EncPlanHost& plan,
                        uint64_t raw_budget, bool rate_mode, bool wide_pass)
{
  return launch_speck_encode_head(stream, b, plan, raw_budget, rate_mode, wide_pass) ||
                 launch_speck_encode_planes(stream, b, plan, raw_budget, rate_mode, wide_pass)
             ? -1
             : 0;
}

}  // namespace sperrhip
