// speck_mx.hip -- LIS phase of chunks whose lists MIX set shapes, GPU-WIDE (k_lis_mx, round 4).
//
// The sorting pass of a plane (/root/reference/src/SPECK3D_INT.cpp:99-138: the lists from the smallest sets to
// the largest, m_process_S / m_code_S :140-212, the split rule :214-326; 2D: /root/reference/src/SPECK2D_INT.cpp:
// 10-218) is a serial parse: what a bit means depends on every earlier bit.  k_lis_mixed (rounds 2-3, removed at
// the end of round 4) ran it with ONE workgroup per chunk that built speculative tables (rows of split lengths per
// stream position and shape class), walked the list entries with one wavefront and expanded the sets that were hopped over -- all
// three on the one workgroup's clock.  Here only the walk stays serial:
//
//   * the phase's stream is cut into fixed REGIONS of S bits, handed out by a ticket counter to the workgroups
//     of a chunk (a region is handed out only after all earlier ones: a waiting workgroup always waits for a
//     running one);
//   * OFF the chain a workgroup loads its region (+ M bits of look-ahead) and builds the rows: sixteen columns
//     per position -- a single sample, the leaf parents of 2 / 4 / 8 samples, and the twelve shape classes up to
//     three steps above them whose columns save the most (spk::build_mx_columns) -- of which a region builds
//     only those its lists can need (a hint the chain leaves behind);
//   * ON the chain its first wavefront takes the predecessor's state -- stream position, list level, entry
//     index, entries left, the stack of sets being walked into (child ordinal + "found" bit per frame, the
//     entry at the bottom), the 2D coder's type-I state -- from tagged words written and read with relaxed
//     agent-scope atomics (no fence: k_lis_hi's protocol), walks its region (a significant entry with a column
//     is 21 hand-written scalar instructions -- lane = stream bit: eight split lengths of the list's two column
//     groups, lane = entry: which of them is its class's --, any other set is walked into child by child; a set
//     whose split leaves the rows is walked into as well, so the walk always reaches the region's end) and
//     publishes the state there; the classes of the entries it will visit are loaded while it waits for its turn;
//   * OFF the chain again it expands the sets it hopped over, breadth first with all threads: leaf parents
//     become leaf events (k_leaf_apply), insignificant child sets are recorded with their stream position
//     (k_place_scan / _scatter rank them), births and events go to segments that are the workgroup's alone.
//
// Old entries are compacted by k_lis_compact, which also sets the chunk's state after the phase.
// tests/model/speck_model.cpp::model_speck3d_decode_mixed pins the formulation (rows, hops, walking into sets).
#include "speck_dec.h"

namespace sperrhip {

using namespace spk;

namespace {

constexpr int kMxThreads = 1024;
constexpr int kMxCols = 16;
constexpr int kMxColsAll = 17;   // + one column of its own array (`Tx`): the heaviest class the sixteen leave out, up to four steps up
constexpr int kMxLdsRoots = 48, kMxLdsGrids = 352;   // (host: use_mixed checks that the tree fits)
constexpr uint32_t kTInf = 0xffffu, kTNone = 0xfffeu;
constexpr uint32_t kMxTagShift = 57;
constexpr unsigned long long kMxOver = 1ull << 56;
constexpr int kMxFrames = kMaxDepth + 2;
constexpr uint32_t kNoTicket = 0xffffffffu;
constexpr int kMxPreWaves = 4;   // wavefronts that fill the ring of entry classes while the region waits for its turn
constexpr int kMxRecs = 160;   // per-word records a region's walk can leave (one per word and reload of entry classes)

// what the chain is doing (2D coder: the type-I set, /root/reference/src/SPECK2D_INT.cpp:44-98)
constexpr uint32_t kModeList = 0;      // the entries of list `level`
constexpr uint32_t kModeITest = 1;     // the type-I set's own test comes next
constexpr uint32_t kModeISub = 2;      // the test of subband iJ of the type-I set's level comes next
constexpr uint32_t kModeISubWalk = 3;  // inside a subband that was walked into

struct MxCtx {
  uint64_t parent;     // packed node of the set being walked into
  KidBox kb;
  uint8_t pc;          // its class
  uint8_t next;        // ordinal of the next child
  uint8_t found;       // an earlier child was significant
  uint8_t pad;
};

#define MX_ACTIVE_OR_RETURN(s, p)                                \
  if (!(s).active || (s).done || (int)(p) >= (s).nbp)            \
    return;

// lanes of ONE wavefront that talk through LDS (see HI_WAVE_SYNC in speck_dec.hip)
#define MX_WAVE_SYNC()                                        \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");    \
    __builtin_amdgcn_wave_barrier();                          \
  } while (0)

template <bool kStamps>
__global__ void __launch_bounds__(kMxThreads) k_lis_mx(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  DecState& s = b.st[c];
  MX_ACTIVE_OR_RETURN(s, p);
  extern __shared__ __attribute__((aligned(16))) char mx_smem[];
  __shared__ ShapeCls sh_cls[kMaxCls];
  __shared__ uint64_t sh_kcol[kMaxCls];             // byte k: column of child k (0: a single sample, 0xff: none)
  __shared__ uint8_t sh_slot[kMaxCls + 2];          // column of every class
  __shared__ Root sh_roots[kMxLdsRoots];
  __shared__ Grid sh_grids[kMxLdsGrids];
  __shared__ uint8_t sh_gridCls[kMxLdsGrids * 8];
  __shared__ MxCtx sh_ctx[kMxFrames + 1];
  __shared__ uint8_t sh_colCls[kMxColsAll + 3];     // class of every column (0xff: unused)
  __shared__ uint8_t sh_levelSlot[kMaxLevels];      // birth-mask slot of every list level
  __shared__ uint8_t sh_lgrp[kMaxLevels];           // the two column groups most entries of a list level belong to (2 bits each)
  __shared__ uint32_t sh_len[kMaxLevels], sh_lOff[kMaxLevels];
  // chain state (the first wavefront owns it; the others read it between barriers)
  __shared__ uint64_t sh_pos, sh_base;
  __shared__ uint32_t sh_level, sh_e, sh_rem, sh_mode, sh_stop, sh_needFill, sh_ringHi, sh_ticket, sh_abort, sh_published;
  __shared__ uint32_t sh_iJ, sh_iPart, sh_iCounter, sh_iNeed;
  __shared__ int sh_depth;
  __shared__ uint32_t sh_qn[2], sh_ncand, sh_hcap;
  __shared__ uint32_t sh_preLevel, sh_preLo, sh_preHi, sh_go, sh_goDone;   // list entries whose classes were put into the ring ahead of the chain
  __shared__ uint32_t sh_segBorn, sh_segLeaf, sh_segBornEnd;   // filled slots of this workgroup's segments
  __shared__ unsigned long long sh_in[8];
  __shared__ uint64_t sh_recM[kMxRecs][2];   // per-word records of the walk's tight loop: positions, entry ordinals
  __shared__ uint32_t sh_recE[kMxRecs], sh_recK[kMxRecs], sh_nrec;   //   first entry (index into the chunk's lists), word
  __shared__ uint64_t sh_tk[4];

  const int tid = threadIdx.x;
  const uint32_t lane = (uint32_t)tid & 63u, wave = (uint32_t)tid >> 6;
  const uint32_t nlevels = b.tree.nlevels;
  const bool twoD = (b.tree.flags & kTree2D) != 0;
  const uint32_t cur = s.cur;
  if (tid < kMxColsAll + 3)
    sh_colCls[tid] = 0xff;
  if (tid < kMaxLevels) {
    const bool in = (uint32_t)tid < nlevels;
    sh_levelSlot[tid] = in ? b.levelSlot[tid] : (uint8_t)0xff;
    sh_lgrp[tid] = in ? (uint8_t)(b.mxLevelGroup[tid] & 127u) : (uint8_t)4;
    sh_len[tid] = in ? s.listLen[cur][tid] : 0u;
    sh_lOff[tid] = in ? b.levelOff[tid] : 0u;
  }
  if (tid == 0) {
    sh_segBorn = sh_segLeaf = 0;
    sh_segBornEnd = 0xffffffffu;
    sh_slot[kMaxCls] = sh_slot[kMaxCls + 1] = 0xff;
  }
  __syncthreads();
  for (uint32_t i = tid; i < b.tree.nroots && i < (uint32_t)kMxLdsRoots; i += kMxThreads)
    sh_roots[i] = b.tree.roots[i];
  for (uint32_t i = tid; i < b.tree.ngrids && i < (uint32_t)kMxLdsGrids; i += kMxThreads)
    sh_grids[i] = b.tree.grids[i];
  for (uint32_t i = tid; i < b.tree.ngrids * 8 && i < (uint32_t)kMxLdsGrids * 8; i += kMxThreads)
    sh_gridCls[i] = b.tree.gridCls[i];
  for (uint32_t i = tid; i < b.tree.ncls && i < (uint32_t)kMaxCls; i += kMxThreads) {
    const ShapeCls cc = b.tree.cls[i];
    sh_cls[i] = cc;
    uint64_t kc = 0;
    for (int k = 0; k < 8; k++) {
      const uint32_t kid = k < cc.nk ? cc.kid[k] : kClsPixel;
      kc |= (uint64_t)(kid == kClsPixel ? 0u : b.mxSlot[kid]) << (8 * k);
    }
    sh_kcol[i] = kc;
    const uint8_t sl = b.mxSlot[i];
    sh_slot[i] = sl;
    if (sl < kMxColsAll)
      sh_colCls[sl] = (uint8_t)i;
  }

  const uint64_t* words = b.stream + c * b.streamStride;
  const uint64_t nwordsAvail = (s.avail + 63) / 64;
  unsigned long long* bornM = reinterpret_cast<unsigned long long*>(b.bornM + c * b.maskPixStride);
  unsigned long long* sigNew = reinterpret_cast<unsigned long long*>(b.sigNew + c * b.maskPixStride);
  unsigned long long* sign = reinterpret_cast<unsigned long long*>(b.sign + c * b.signStride);
  const uint64_t phase0 = s.lipStart + s.lipBits;
  const uint64_t maskBits = (uint64_t)b.maskWords * 64;
  uint64_t* bornPacked = b.bornPacked + c * b.bornPitch;
  uint64_t* bornPosLev = b.bornPosLev + c * b.bornPitch;
  uint64_t* sigbits = b.sigbits + c * b.sigbitsStride;
  uint64_t* leafEv = b.leafEv + c * b.leafStride;
  const uint64_t* lisCur = b.lis[cur] + c * b.lisStride;
  unsigned long long* flags = b.hiFlags + c * b.hiFlagStride;
  const unsigned long long tag = (unsigned long long)(p + 1) << kMxTagShift;

  const uint32_t S = b.mxS, W = b.mxS + b.mxM, Q = b.mxQ;
  const uint32_t kWords = ((W >> 6) + 5u) & ~1u;
  uint64_t* const wbits = reinterpret_cast<uint64_t*>(mx_smem);
  const uint32_t* const w32 = reinterpret_cast<const uint32_t*>(mx_smem);
  uint16_t* const Tr = reinterpret_cast<uint16_t*>(mx_smem + (size_t)kWords * 8);   // [W + 3][16]
  uint16_t* const Tx = Tr + (size_t)(W + 3) * kMxCols;   // [W + 3]: column 16
  char* const ldsQ = mx_smem + (size_t)kWords * 8 + (((size_t)(W + 3) * kMxColsAll * 2 + 15) & ~(size_t)15);
  auto row_at = [&](uint32_t pos, uint32_t col) -> uint32_t { return col < (uint32_t)kMxCols ? Tr[(size_t)pos * kMxCols + col] : Tx[pos]; };
  uint64_t* const qidA = reinterpret_cast<uint64_t*>(ldsQ);
  uint32_t* const qmetaA = reinterpret_cast<uint32_t*>(qidA + Q);   // first bit | class << 16 | "list entry" << 24
  uint64_t* const qidB = reinterpret_cast<uint64_t*>(qmetaA + Q);
  uint32_t* const qmetaB = reinterpret_cast<uint32_t*>(qidB + Q);
  uint16_t* const ecls = reinterpret_cast<uint16_t*>(qmetaB + Q);   // ring [kMxRing]: class | column in the list's group << 8
  uint16_t* const cand = reinterpret_cast<uint16_t*>(qidB);         // [W] (the rows are built before anything is queued)
  uint64_t a = 0;     // stream position of the region's bit 0
  uint32_t wq0 = 0;   // bit offset of region position 0 inside wbits[0]
  __syncthreads();

  auto bit_at = [&](uint32_t r) -> uint32_t {
    const uint32_t q = r + wq0;
    return (w32[q >> 5] >> (q & 31)) & 1u;
  };
  auto bits32 = [&](uint32_t r) -> uint32_t {  // 32 stream bits starting at r
    const uint32_t q = r + wq0, sh = q & 31;
    const uint32_t lo = w32[q >> 5], hi = w32[(q >> 5) + 1];
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
  };
  auto pixel_event = [&](uint32_t ridx, bool sig, uint32_t signbit) {
    atomicOr(bornM + (ridx >> 6), 1ull << (ridx & 63));
    if (sig) {
      atomicOr(sigNew + (ridx >> 6), 1ull << (ridx & 63));
      if (!signbit)
        atomicAnd(sign + (ridx >> 6), ~(1ull << (ridx & 63)));
    }
  };
  // births and leaf events go to a segment of the chunk's arrays that is this workgroup's alone (an LDS
  // counter; the shared part, with its global counter, takes what does not fit) -- k_lis_hi's scheme
  const uint32_t segB0 = (uint32_t)b.bornStride + blockIdx.x * b.bornSeg;
  const uint32_t segL0 = b.leafCap + blockIdx.x * b.leafSeg;
  auto born_slots = [&](uint32_t n) -> uint32_t {
    const uint32_t k = atomicAdd(&sh_segBorn, n);
    if (k + n <= b.bornSeg)
      return segB0 + k;
    atomicMin(&sh_segBornEnd, k);        // the segment is full from here on
    return atomicAdd(&s.bornCount, n);   // (slots at or past bornStride are dropped by write_born)
  };
  auto leaf_slot = [&]() -> uint32_t {
    const uint32_t k = atomicAdd(&sh_segLeaf, 1u);
    if (k < b.leafSeg)
      return segL0 + k;
    const uint32_t g = atomicAdd(&s.leafCount, 1u);
    return g < b.leafCap ? g : 0xffffffffu;
  };
  auto write_born = [&](uint32_t slot, uint32_t lev, uint64_t rel, uint64_t packed) {
    if (!(slot < b.bornStride || (slot >= segB0 && slot < segB0 + b.bornSeg)))
      return;   // (only a damaged stream asks for more slots than there are sets)
    bornPacked[slot] = packed;
    bornPosLev[slot] = ((uint64_t)lev << 48) | rel;
    atomic_or64(b.mask + c * b.maskStride + (size_t)sh_levelSlot[lev] * b.maskWords + (rel >> 6),
                1ull << (rel & 63));
  };
  auto born_counts = [&](uint32_t lev, uint64_t rel) -> bool {   // is this birth recorded at all
    return lev < (uint32_t)kMaxLevels && sh_levelSlot[lev] != 0xff && rel < maskBits;
  };
  // a set born insignificant at stream position abs
  auto record_born = [&](uint32_t lev, uint64_t abs, uint64_t packed) {
    const uint64_t rel = abs - phase0;
    if (!born_counts(lev, rel))
      return;  // past the usable stream: decoding stops after this plane anyway
    write_born(born_slots(1u), lev, rel, packed);
  };
  // geometry with the tree's tables in LDS (spk::kid_box / node_cls with these arrays)
  auto kid_box_l = [&](const Node& nd, KidBox& k) {
    const Grid g = sh_grids[nd.grid];
    const Root r = sh_roots[g.root];
    k.grid = (uint16_t)(nd.grid + 1);
    k.rev = (uint16_t)(b.tree.flags & kTree2D);
    uint32_t lev = r.lev;
    const int d = g.depth;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      const int Da = r.D[ax];
      if (Da != 0 && !k.rev) {   // spk::node_level
        if (d < Da)
          lev += (uint32_t)d;
        else {
          lev += (uint32_t)(Da - 1);
          if (axis_len(r.len[ax], Da - 1, (uint32_t)nd.i[ax] >> 1) >= 2)
            lev += 1;
        }
      }
      const bool splits = d < Da;
      k.e[ax] = splits ? g.e[ax] + 1 : g.e[ax];
      k.base[ax] = splits ? (uint32_t)nd.i[ax] * 2u : (uint32_t)nd.i[ax];
      k.n[ax] = (splits && axis_len(r.len[ax], k.e[ax], k.base[ax] + 1u) > 0) ? 2u : 1u;
      if (!k.rev)
        lev += k.n[ax] - 1u;
    }
    k.kidlev = k.rev ? r.lev + (uint32_t)d + 1u : lev;   // (2D coder: one level per partition step)
    k.nk = k.n[0] * k.n[1] * k.n[2];
  };
  auto node_cls_l = [&](const Node& nd) -> uint32_t {
    const Grid g = sh_grids[nd.grid];
    const Root& r = sh_roots[g.root];
    uint32_t k = 0;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      const int e = g.e[ax];
      const uint32_t rem = (uint32_t)r.len[ax] & ((1u << e) - 1u);
      k |= (bitrev(nd.i[ax], e) < rem ? 1u : 0u) << ax;
    }
    return sh_gridCls[(uint32_t)nd.grid * 8u + k];
  };
  // a significant leaf parent of nk samples whose split starts at y: ONE event word (node id,
  // significance and sign masks by child ordinal) that k_leaf_apply turns into mask updates
  auto leaf_event = [&](const Node& nd, uint32_t y, uint32_t nk) {
    const uint32_t v = bits32(y);
    uint32_t yy = 0, found = 0, sigm = 0, negm = 0;
    for (uint32_t k = 0; k < nk; k++) {
      const uint32_t coded = found | (uint32_t)(k + 1 != nk);
      const uint32_t bit = coded ? (v >> yy) & 1u : 1u;
      yy += coded;
      const uint32_t sgn = (v >> yy) & 1u;
      sigm |= bit << k;
      negm |= (bit & (sgn ^ 1u)) << k;
      found |= bit;
      yy += bit;
    }
    const Grid& g = sh_grids[nd.grid];
    const uint32_t fid = g.nodeOff + ((((uint32_t)nd.i[2] << g.e[1]) + nd.i[1]) << g.e[0]) + nd.i[0];
    const uint32_t slot = leaf_slot();
    if (slot != 0xffffffffu)
      leafEv[slot] = (uint64_t)fid | ((uint64_t)sigm << 32) | ((uint64_t)negm << 40);
  };
  // the next list after level `l` (exclusive) that holds entries
  auto next_level = [&](int l) -> int {
    for (l = l - 1; l >= 0; l--)
      if (sh_len[l] != 0)
        return l;
    return -1;
  };

  // ---- the rows of the region in view (all threads)
  auto build_rows = [&](uint32_t hcap) {
    if (tid == 0)
      sh_ncand = 0;
    __syncthreads();
    // columns 0..3 at every position (the next 16 bits give the three leaf parents' splits), the
    // other columns marked "not computed"; positions where a coded item's split can start are collected
    for (uint32_t x = tid; x <= W + 2; x += kMxThreads) {
      uint32_t T0 = kTInf, T1 = kTInf, T2 = kTInf, T3 = kTInf, rest = 0xffffffffu;
      bool isCand = false;
      if (x < W) {
        const uint32_t v = bits32(x);
        // children 0..6 coded one after the other; the last child of 2 / 4 / 8 is coded only when
        // an earlier one was significant
        uint32_t y = 0, found = 0, t2 = 0, t4 = 0, t8 = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if (k == 1 || k == 3 || k == 7) {
            const uint32_t bit = found ? (v >> y) & 1u : 1u;
            const uint32_t tl = y + found + bit;
            if (k == 1)
              t2 = tl;
            else if (k == 3)
              t4 = tl;
            else
              t8 = tl;
          }
          if (k < 7) {
            const uint32_t bit = (v >> y) & 1u;
            found |= bit;
            y += 1u + bit;
          }
        }
        T0 = 1u;
        T1 = x + t2 <= W ? t2 : kTInf;
        T2 = x + t4 <= W ? t4 : kTInf;
        T3 = x + t8 <= W ? t8 : kTInf;
        rest = kTNone | (kTNone << 16);
        // (a region may begin anywhere: position 0 can be the start of a split whose test bit the
        //  region before holds)
        isCand = x == 0 || bit_at(x - 1) != 0;
      }
      uint4* row = reinterpret_cast<uint4*>(Tr + (size_t)x * kMxCols);
      row[0] = make_uint4(T0 | (T1 << 16), T2 | (T3 << 16), rest, rest);
      row[1] = make_uint4(rest, rest, rest, rest);
      Tx[x] = (uint16_t)(x < W ? kTNone : kTInf);
      const uint64_t cm = __ballot(isCand);
      if (cm) {
        uint32_t base = 0;
        const uint32_t leader = (uint32_t)__ffsll((long long)__ballot(true)) - 1u;
        if (lane == leader)
          base = atomicAdd(&sh_ncand, (uint32_t)__popcll(cm));
        base = __shfl(base, (int)leader, 64);
        if (isCand)
          cand[base + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull))] = (uint16_t)x;
      }
    }
    __syncthreads();
    // columns 4..7, 8..11, 12..15: a chain of look-ups through the children's columns at the collected
    // positions (four per thread and round: their LDS round trips overlap)
    const uint32_t ncand = sh_ncand;
    uint32_t hPrev = 0;
    for (uint32_t g4 = 4; g4 < (uint32_t)kMxColsAll + 3u; g4 += 4) {   // (16..19: only 16 exists, behind all the others)
      bool any = false;
      for (uint32_t col = g4; col < g4 + 4; col++) {
        const uint32_t ci = sh_colCls[col];
        if (ci == 0xff || sh_cls[ci].h > hcap)   // (a column left "not computed": such a set is walked into)
          continue;
        if (any && sh_cls[ci].h != hPrev)   // (its children's columns may be this group's: columns go by steps)
          __syncthreads();
        any = true;
        hPrev = sh_cls[ci].h;
        const uint32_t nk = sh_cls[ci].nk;
        const uint64_t kc = sh_kcol[ci];
        // (one position per thread and round: sixteen wavefronts hide the LDS round trips of each other's chains --
        //  k_lis_mixed's four chains per thread in lockstep cost this kernel four times the instructions for
        //  windows that hold fewer candidates than the workgroup has threads)
        for (uint32_t i = tid; i < ncand; i += kMxThreads) {
          const uint32_t x = (uint32_t)cand[i];
          uint32_t y = x, found = 0, bad = 0;
          for (uint32_t k = 0; k < nk; k++) {
            const uint32_t ccol = (uint32_t)(kc >> (8 * k)) & 0xffu;
            const uint32_t coded = found | (uint32_t)(k + 1 != nk);
            const uint32_t yy = min(y, W + 1);
            const uint32_t bitv = coded ? bit_at(yy) : 1u;
            const uint32_t s0 = yy + coded;
            uint32_t t = row_at(s0, ccol);
            if (t >= kTNone) {
              bad |= bitv ? (t == kTInf ? 1u : 2u) : 0u;
              t = 0;
            }
            y = bitv ? s0 + t : yy + 1;
            found |= bitv;
          }
          const uint32_t t = ((bad & 1u) || y > W) ? kTInf : (bad & 2u) ? kTNone : y - x;
          if (col < (uint32_t)kMxCols)
            Tr[(size_t)x * kMxCols + col] = (uint16_t)t;
          else
            Tx[x] = (uint16_t)t;
        }
      }
      if (any)
        __syncthreads();
    }
  };

  // ---- classes of the entries [from, to) of list `l` into the ring, with their column inside the list's group
  auto fill_ring = [&](uint32_t l, uint32_t from, uint32_t to, uint32_t t0, uint32_t nt) {
    const uint64_t* list = lisCur + sh_lOff[l];
    const uint32_t ga = sh_lgrp[l] & 3u, gb = (sh_lgrp[l] >> 2) & 3u;
    auto put = [&](uint32_t i, uint64_t ent) {
      const uint32_t ci = node_cls_l(unpack_node(ent));
      const uint32_t col = sh_slot[ci];
      const uint32_t loc = col >= (uint32_t)kMxCols ? 0xffu : (col >> 2) == ga ? (col & 3u) : (col >> 2) == gb ? 4u + (col & 3u) : 0xffu;
      // (high byte: the bit offset of the class's byte in the list's pair of row words, 0x80: in neither group)
      ecls[i & (uint32_t)(kMxRing - 1)] = (uint16_t)(ci | ((loc < 8u ? loc << 3 : 0x80u) << 8));
    };
    uint32_t i = from + t0;
    for (; i + 3u * nt < to; i += 4u * nt) {   // (four loads in flight per thread)
      const uint64_t e0 = list[i], e1 = list[i + nt], e2 = list[i + 2u * nt], e3 = list[i + 3u * nt];
      put(i, e0);
      put(i + nt, e1);
      put(i + 2u * nt, e2);
      put(i + 3u * nt, e3);
    }
    for (; i < to; i += nt)
      put(i, list[i]);
  };

  // Bits of the split of a set of class ci that starts at x, worked out ON the chain from its children's columns
  // (first wavefront, uniform): for a set without a column of its own -- or whose column this region did not build --
  // all of whose significant children have one.  Such a set is then hopped over like any other, and its records are
  // the expansion's (which only ever needs the CHILDREN's columns), instead of being walked into: a round in there
  // costs four times this.  kTNone / kTInf: no luck.
  auto chain_len_with = [&](auto&& bit_of, uint32_t ci, uint32_t x) -> uint32_t {
    if (ci >= b.tree.ncls || ci >= (uint32_t)kMaxCls)
      return kTNone;
    const uint32_t nk = sh_cls[ci].nk;
    const uint64_t kc = sh_kcol[ci];
    uint32_t y = x, found = 0;
    for (uint32_t k = 0; k < nk; k++) {
      const uint32_t col = (uint32_t)(kc >> (8 * k)) & 0xffu;
      const uint32_t coded = found | (uint32_t)(k + 1 != nk);
      if (y > W)
        return kTInf;
      const uint32_t bit = coded ? bit_of(y) : 1u;
      const uint32_t s0 = y + coded;
      if (!bit) {
        y += 1;
        continue;
      }
      if (col >= (uint32_t)kMxColsAll)
        return kTNone;
      const uint32_t t = row_at(min(s0, W + 2), col);
      if (t >= kTNone)
        return t;
      y = s0 + t;
      found = 1;
    }
    return y > W ? kTInf : y - x;
  };
  auto chain_len = [&](uint32_t ci, uint32_t x) -> uint32_t { return chain_len_with(bit_at, ci, x); };

  // ---- the walk through the region (first wavefront, every lane carrying the same walker state): from
  //      sh_pos on until an item starts at or past S, or the list ends (depth 0)
  uint64_t wk_tight = 0, wk_into = 0, wk_total = 0, wk_fill = 0;
  uint32_t wk_hopsT = 0, wk_hopsG = 0, wk_rounds = 0, wk_words = 0, wk_calls = 0, wk_fills = 0, wk_zruns = 0;
  uint32_t wk_gSat = 0, wk_gView = 0, wk_gInf = 0, wk_gChain = 0, wk_enters = 0, wk_reloads = 0;
  uint64_t wk_seen = 0, wk_toWalk = 0, wk_pro = 0, wk_epi = 0, wk_dec = 0;   // (seen: the state words are in; cycles from there to the first walk; the walk's set-up; its end)
  uint32_t curRegion = 0, curMode = 0;   // (set by the chain below before it calls walk())
  auto walk = [&]() {
    const uint64_t wk0 = kStamps ? __builtin_readcyclecounter() : 0;
    // (read here, off the end of the walk: what the state words 4 and 5 hold while a list is walked)
    const unsigned long long pubBase = sh_base & ((1ull << kMxTagShift) - 1ull);
    const unsigned long long pubI = (unsigned long long)sh_iJ | ((unsigned long long)sh_iPart << 2) |
                                    ((unsigned long long)sh_iCounter << 8) | ((unsigned long long)sh_iNeed << 10);
    uint32_t r = __builtin_amdgcn_readfirstlane((uint32_t)(sh_pos - a)), e = __builtin_amdgcn_readfirstlane(sh_e),
             rem = __builtin_amdgcn_readfirstlane(sh_rem);
    uint32_t qn = __builtin_amdgcn_readfirstlane(sh_qn[0]), ns = 0;
    int depth = __builtin_amdgcn_readfirstlane(sh_depth);
    const uint32_t level = __builtin_amdgcn_readfirstlane(sh_level);
    const uint32_t lOff = sh_lOff[level];
    const uint64_t* list = lisCur + lOff;
    const uint32_t ga = sh_lgrp[level] & 3u, gb = (sh_lgrp[level] >> 2) & 3u;
    const uint32_t ringHi = __builtin_amdgcn_readfirstlane(sh_ringHi);
    uint32_t stE = 0, stM = 0;   // staged items of the list hops: lane = item
    // lane = stream word of the region
    const uint64_t sw0 = lane < kWords ? wbits[lane] : 0ull;
    uint32_t curK = 0xffffffffu;   // stream word the registers below belong to
    uint64_t m = 0;                // that word (uniform)
    // lane = bit of it: the split lengths (+ 1) of the list's two column groups one position on, eight bits each
    // (255: 254 bits and more, or none -- the general code looks at the row itself)
    uint32_t lrowA = 0, lrowB = 0;
    uint32_t lrowK = 0xffffffffu;  //   (loaded for this word)
    typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
    // four 16-bit split lengths as four bytes: length + 1, 255 for 254 bits and more (or none: kTNone, kTInf)
    auto pack8 = [&](uint64_t v) -> uint32_t {
      const us2_t lim = {254, 254};
      const us2_t lo = __builtin_elementwise_min(__builtin_bit_cast(us2_t, (uint32_t)v), lim);
      const us2_t hi = __builtin_elementwise_min(__builtin_bit_cast(us2_t, (uint32_t)(v >> 32)), lim);
      return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u) + 0x01010101u;
    };
    auto load_lrow = [&](uint32_t kk) {
      const int32_t row = (int32_t)(kk * 64u + lane) - (int32_t)wq0 + 1;
      uint64_t va = ~0ull, vb = ~0ull;
      if (row >= 0 && row <= (int32_t)W + 2) {
        va = *reinterpret_cast<const uint64_t*>(Tr + (size_t)row * kMxCols + ga * 4u);
        vb = *reinterpret_cast<const uint64_t*>(Tr + (size_t)row * kMxCols + gb * 4u);
      }
      lrowA = pack8(va);
      lrowB = pack8(vb);
    };
    uint32_t eb = 0x80000000u;     // lane = list entry eb + lane: its ecls word (nothing loaded yet)
    uint32_t ecv = 0, ecb = 0;
    uint32_t nrec = __builtin_amdgcn_readfirstlane(sh_nrec), nloc = 0;   // records in LDS, records in the registers below
    uint32_t rcL = 0, rcH = 0, riL = 0, riH = 0, rE = 0, rK = 0;         // lane = record: position mask, ordinal mask, first entry, word
    auto rec_flush = [&]() {
      if (lane < nloc && nrec + lane < (uint32_t)kMxRecs) {
        sh_recM[nrec + lane][0] = (uint64_t)rcL | ((uint64_t)rcH << 32);
        sh_recM[nrec + lane][1] = (uint64_t)riL | ((uint64_t)riH << 32);
        sh_recE[nrec + lane] = rE;
        sh_recK[nrec + lane] = rK;
      }
      nrec += nloc;
      nloc = 0;
    };
    // last stream word the tight loop takes: the one the region's last bit lies in, ALL of it -- the walk may leave a
    // region up to 63 bits late (a long split does that anyway; rows 1 .. S + 128 exist, entry classes to e + S + 128)
    const uint32_t kkLast = (S + wq0 - 1u) >> 6;
    // LDS address of lane's row entries for word 0 (+ word << 11: 64 rows of 32 bytes), then of the list's two column groups
    const uint32_t adBase = (uint32_t)(size_t)Tr + (uint32_t)((int32_t)(lane + 1u) - (int32_t)wq0) * (uint32_t)(kMxCols * 2);
    const uint32_t adBaseA = adBase + ga * 8u, adBaseB = adBase + gb * 8u;
    auto rl32 = [&](uint32_t v, uint32_t idx) -> uint32_t {
      return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)idx);
    };
    auto rl64 = [&](uint64_t v, uint32_t idx) -> uint64_t {
      return (uint64_t)rl32((uint32_t)v, idx) | ((uint64_t)rl32((uint32_t)(v >> 32), idx) << 32);
    };
    // (a stream bit out of the words in the lane registers: two v_readlane instead of an LDS round trip -- the walk
    //  into a set and the lengths worked out on the chain read a bit per child)
    auto bit_w = [&](uint32_t rr) -> uint32_t {
      const uint32_t q = rr + wq0;
      return (uint32_t)(rl64(sw0, (q >> 6) & 63u) >> (q & 63u)) & 1u;
    };
    auto chain_len_w = [&](uint32_t ci, uint32_t x) -> uint32_t { return chain_len_with(bit_w, ci, x); };
    auto flush = [&]() {
      if (lane < ns && qn + lane < Q) {
        qidA[qn + lane] = stE;
        qmetaA[qn + lane] = stM;
      }
      qn += ns;
      ns = 0;
    };
    if (kStamps)
      wk_pro += __builtin_readcyclecounter() - wk0;
    while (true) {
      if (r >= S)
        break;
      if (depth == 1) {   // the list itself
        if (rem == 0) {
          depth = 0;
          break;
        }
        const uint32_t q = r + wq0, k = q >> 6, o = q & 63u;
        if (k != curK) {
          curK = k;
          m = rl64(sw0, k & 63u);
          lrowK = 0xffffffffu;
        }
        // Stream words that lie inside the region, with at least 64 entries left: their entries in a tight
        // loop, word after word -- no end-of-list or end-of-region checks per entry; anything unusual (a set to
        // walk into) is left to the general code below.  The loop only HOPS: per significant entry one bit in
        // each of two masks of a per-word record (its position in the word, its ordinal among the entries from
        // `eb` on); the work items are made from the records off the chain (the first step of expand_all).
        if (rem >= 64u && k <= kkLast && nrec + nloc + 2u < (uint32_t)kMxRecs) {
          const uint32_t eEnd = e + rem, eStop = eEnd - 64u;
          uint32_t kk = k, oo = o;   // the word and the bit of it the walk is at
          bool unusual = false;
          const uint64_t tt0 = kStamps ? __builtin_readcyclecounter() : 0;
          if (lrowK != kk)
            load_lrow(kk);
          // The whole loop over the words is one asm statement: what a lone wavefront pays for is the length of its
          // dependency chains and where its loop lies in memory (tools/micro/hop_loop.cpp, hop_align.cpp: the loop head
          // at a 128-byte boundary and the hop below, four copies in a row, is 160 cycles per significant entry
          // where the loop of 21 instructions placed by chance was 197 to 239).  Per hop: the zeros up to the next
          // significant entry (s_lshr_b64, s_ff1), its class word (lane = ordinal mod 64) and the two row words of its
          // position (lane = position), s_bfe_u64 cuts the split's length + 1 out of the pair, and the position --
          // kept as position - 64 -- carries out of the add when the word is through.  A length of 255 ("look it up")
          // ends a word the same way and is found there, once per word; a class with no byte in the pair has bit 7
          // of its class word set.
          // The entry classes are a window of 64 ordinals that rolls along: at the end of a word every lane fetches
          // the class of the one ordinal of [idx, idx + 64) that is congruent to it (a word of 64 bits uses up some
          // fifty ordinals: reloading the window from outside the statement was every other word's exit); the load
          // is in flight while the word's record and the next word's rows are put together.
          // The statement comes back for what is rare: an unusual entry (st 1), a split that skipped a word (st 3:
          // the row registers want a load), and (st 0) the record registers full, the region's last word, the ring's
          // or the list's last entries.  Every word visited leaves one record in lane nloc of rcL .. riH and rE (an
          // empty one if nothing was significant); their words are filled in below.
          // m: vcc; position mask: s[92:93]; ordinal mask: s[94:95]; row words: s[96:97]; length: s98; next rows:
          // v[120:123]; ring address and class: v124, v125.
#define MX_HOP                                                                                     \
  "s_lshr_b64 %[mm], vcc, %[oo]\n\t"                                                               \
  "s_ff1_i32_b64 %[st], %[mm]\n\t"          /* insignificant entries: one bit each */              \
  "s_add_u32 %[idx], %[idx], %[st]\n\t"                                                            \
  "s_add_u32 %[oo], %[oo], %[st]\n\t"                                                              \
  "v_readlane_b32 %[ec], %[ecb], %[idx]\n\t"                                                       \
  "v_readlane_b32 s96, %[rlo], %[oo]\n\t"                                                          \
  "v_readlane_b32 s97, %[rhi], %[oo]\n\t"                                                          \
  "s_bitset1_b64 s[94:95], %[idx]\n\t"                                                             \
  "s_bitset1_b64 s[92:93], %[oo]\n\t"                                                              \
  "s_add_u32 %[idx], %[idx], 1\n\t"
#define MX_HOP_TEST                                                                                \
  "s_bitcmp1_b32 %[ec], 7\n\t"            /* a class with no byte in the pair */                   \
  "s_cbranch_scc1 9f\n\t"
#define MX_HOP_END                                                                                 \
  "s_bfe_u64 s[98:99], s[96:97], %[ec]\n\t" /* bits of the entry's split + 1 */                    \
  "s_add_u32 %[oo], %[oo], s98\n\t"
          static_assert(kMxRing == 8192, "the ring's index mask is a literal of the statement below");
          while (true) {
            uint32_t idx = e, st;   // (ordinals: as they are; lanes and mask bits take them mod 64)
            const uint32_t nloc0 = nloc, kk0 = kk;
            const uint32_t nlocLim = min(64u, (uint32_t)kMxRecs - 2u - nrec);
            // a word at position p with entry ordinal idx may touch the ordinals up to idx + 63 - p: all of them on the
            // list and in the ring (compared with idx - p + 64)
            const uint32_t limAbs = min(eEnd, ringHi);
            {   // the window: lane = ordinal mod 64
              const uint32_t eo = e + ((lane - e) & 63u);
              ecb = ((uint32_t)ecls[eo & (uint32_t)(kMxRing - 1)] >> 8) | (8u << 16);
            }
            // words the statement may take: to the region's last, one record each, 64 ordinals each at most
            if (e + 64u - oo > limAbs)
              break;
            uint32_t wl = (uint32_t)__builtin_amdgcn_readfirstlane(   // (one less: the count's borrow ends it)
                (int)min(min(kkLast - kk, nlocLim - nloc - 1u), (limAbs - (e + 64u - oo)) >> 6));
            if (kStamps)
              wk_enters++;
            // (rows 1 .. S + 128 exist: the next word's are fetched while a word is walked)
            uint32_t adNA = adBaseA + ((kk + 1u) << 11), adNB = adBaseB + ((kk + 1u) << 11);
            {
              uint32_t ec_;
              uint64_t mm_;
              asm volatile(
                  "s_sub_u32 %[oo], %[oo], 64\n\t"
                  "s_branch 0f\n\t"
                  ".p2align 7\n\t"
                  "1:\n\t"                                   // ---- the hops of a word
                  MX_HOP MX_HOP_END "s_cbranch_scc1 2f\n\t"
                  MX_HOP MX_HOP_END "s_cbranch_scc1 2f\n\t"
                  MX_HOP MX_HOP_END "s_cbranch_scc1 2f\n\t"
                  MX_HOP MX_HOP_END "s_cbranch_scc0 1b\n\t"
                  "2:\n\t"                                   // ---- the word is through (a split went past its end):
                  "s_lshr_b32 %[st], s93, 31\n\t"            // by a hop at bit 63 that is only the statement's own?
                  "s_cmp_gt_u32 %[st], s91\n\t"
                  "s_cbranch_scc1 3f\n\t"
                  "8:\n\t"                                   // the classes of the next 64 ordinals, the word's record, the next word's rows
                  "v_subrev_u32 v124, %[idx], %[lane]\n\t"
                  "s_lshr_b32 %[st], %[oo], 6\n\t"           // (words skipped)
                  "v_and_b32 v124, 63, v124\n\t"
                  "s_or_b32 %[oo], %[oo], 0xffffffc0\n\t"
                  "v_add_u32 v124, %[idx], v124\n\t"
                  "v_and_b32 v124, 0x1fff, v124\n\t"
                  "s_add_u32 %[kk], %[kk], 1\n\t"
                  "v_lshl_add_u32 v124, v124, 1, %[ring]\n\t"
                  "v_writelane_b32 %[rcl], s92, m0\n\t"
                  "ds_read_u16 v125, v124\n\t"
                  "v_writelane_b32 %[rch], s93, m0\n\t"
                  "v_writelane_b32 %[ril], s94, m0\n\t"
                  "v_writelane_b32 %[rih], s95, m0\n\t"
                  "s_add_u32 %[nloc], %[nloc], 1\n\t"
                  "s_waitcnt lgkmcnt(1)\n\t"
                  "s_cmp_lg_u32 %[st], 0\n\t"
                  "s_cbranch_scc1 6f\n\t"                    // (a long split skipped a word, or a length of 255)
                  "v_pk_min_u16 v120, v120, %[lim]\n\t"      // four 16-bit lengths -> four bytes: length + 1, 255 = look it up
                  "v_pk_min_u16 v121, v121, %[lim]\n\t"
                  "v_pk_min_u16 v122, v122, %[lim]\n\t"
                  "v_pk_min_u16 v123, v123, %[lim]\n\t"
                  "v_perm_b32 v120, v121, v120, %[sel]\n\t"
                  "v_perm_b32 v122, v123, v122, %[sel]\n\t"
                  "v_add_u32 %[rlo], 0x1010101, v120\n\t"
                  "v_add_u32 %[rhi], 0x1010101, v122\n\t"
                  "s_sub_u32 %[wl], %[wl], 1\n\t"            // the region's last word, the record registers full, the ring's
                  "s_cbranch_scc1 5f\n\t"                    // or the list's last entries: a count of words
                  "s_waitcnt lgkmcnt(0)\n\t"
                  "v_lshrrev_b32 v125, 8, v125\n\t"
                  "v_or_b32 %[ecb], 0x80000, v125\n\t"
                  "0:\n\t"                                   // ---- a word
                  "v_readlane_b32 vcc_lo, %[swl], %[kk]\n\t"
                  "v_readlane_b32 vcc_hi, %[swh], %[kk]\n\t"
                  "v_cmp_lt_u32_e64 s[96:97], %[nth], %[ecb]\n\t"   // (the window's entries of a class with no byte)
                  "s_mov_b32 m0, %[nloc]\n\t"
                  "ds_read_b64 v[120:121], %[ada]\n\t"
                  "ds_read_b64 v[122:123], %[adb]\n\t"
                  "s_mov_b64 s[92:93], 0\n\t"
                  "s_mov_b64 s[94:95], 0\n\t"
                  "v_writelane_b32 %[re], %[idx], m0\n\t"    // (the ordinal the word starts with)
                  "v_add_u32 %[ada], 0x800, %[ada]\n\t"
                  "v_add_u32 %[adb], 0x800, %[adb]\n\t"
                  "s_lshr_b32 s91, vcc_hi, 31\n\t"           // bit 63 of the word as it is; set for the loop, which then
                  "s_bitset1_b32 vcc_hi, 31\n\t"             // needs no test for "the rest is zeros"
                  "s_cmp_lg_u64 s[96:97], 0\n\t"
                  "s_cbranch_scc0 1b\n\t"
                  "7:\n\t"                                   // ---- the hops of a word that may meet a class with no byte
                  MX_HOP MX_HOP_TEST MX_HOP_END
                  "s_cbranch_scc0 7b\n\t"
                  "s_branch 2b\n\t"
                  "3:\n\t"                                   // bit 63 was a zero: an insignificant entry, the word's last
                  "s_sub_u32 %[st], %[idx], 1\n\t"
                  "s_bitset0_b32 s93, 31\n\t"
                  "s_bitset0_b64 s[94:95], %[st]\n\t"
                  "s_mov_b32 %[oo], 0\n\t"
                  "s_branch 8b\n\t"
                  "9:\n\t"                                   // an unusual entry: the word's record so far
                  "s_sub_u32 %[idx], %[idx], 1\n\t"
                  "s_bitset0_b64 s[92:93], %[oo]\n\t"
                  "s_bitset0_b64 s[94:95], %[idx]\n\t"
                  "s_mov_b32 %[st], 1\n\t"
                  "v_writelane_b32 %[rcl], s92, m0\n\t"
                  "v_writelane_b32 %[rch], s93, m0\n\t"
                  "v_writelane_b32 %[ril], s94, m0\n\t"
                  "v_writelane_b32 %[rih], s95, m0\n\t"
                  "s_add_u32 %[nloc], %[nloc], 1\n\t"
                  "s_branch 10f\n\t"
                  "5:\n\t"
                  "s_mov_b32 %[st], 0\n\t"
                  "s_branch 10f\n\t"
                  "6:\n\t"                                   // ---- words skipped (st of them)
                  "s_cmp_eq_u32 s98, 255\n\t"
                  "s_cbranch_scc1 4f\n\t"
                  "s_add_u32 %[kk], %[kk], %[st]\n\t"
                  "s_mov_b32 %[st], 3\n\t"
                  "s_branch 10f\n\t"
                  "4:\n\t"                                   // by a length of 255 ("look it up"): back to that entry's bit,
                  "s_lshl_b32 %[st], %[st], 6\n\t"           // the word's record once more without it
                  "s_and_b32 %[oo], %[oo], 63\n\t"
                  "s_sub_u32 %[kk], %[kk], 1\n\t"
                  "s_add_u32 %[oo], %[oo], %[st]\n\t"
                  "s_sub_u32 %[idx], %[idx], 1\n\t"
                  "s_sub_u32 %[oo], %[oo], 255\n\t"
                  "s_bitset0_b64 s[94:95], %[idx]\n\t"
                  "s_bitset0_b64 s[92:93], %[oo]\n\t"
                  "s_mov_b32 %[st], 1\n\t"
                  "v_writelane_b32 %[rcl], s92, m0\n\t"
                  "v_writelane_b32 %[rch], s93, m0\n\t"
                  "v_writelane_b32 %[ril], s94, m0\n\t"
                  "v_writelane_b32 %[rih], s95, m0\n\t"
                  "10:\n\t"
                  "s_waitcnt lgkmcnt(0)\n\t"
                  "s_add_u32 %[oo], %[oo], 64\n\t"
                  : [kk] "+s"(kk), [oo] "+s"(oo), [idx] "+s"(idx), [nloc] "+s"(nloc), [rcl] "+v"(rcL), [rch] "+v"(rcH), [ril] "+v"(riL),
                    [rih] "+v"(riH), [re] "+v"(rE), [rlo] "+v"(lrowA), [rhi] "+v"(lrowB), [ada] "+v"(adNA), [adb] "+v"(adNB), [ecb] "+v"(ecb),
                    [wl] "+s"(wl), [st] "=&s"(st), [mm] "=&s"(mm_), [ec] "=&s"(ec_)
                  : [lim] "s"(0x00fe00feu), [sel] "s"(0x06040200u),
                    [ring] "s"((uint32_t)(size_t)ecls), [nth] "s"(0x8007fu), [swl] "v"((uint32_t)sw0), [swh] "v"((uint32_t)(sw0 >> 32)), [lane] "v"(lane)
                  : "scc", "vcc", "m0", "memory", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "v120", "v121", "v122", "v123",
                    "v124", "v125");
            }
            if (lane >= nloc0 && lane < nloc) {   // the records just written: stream word | first ordinal mod 64, first entry
              rK = (kk0 + (lane - nloc0)) | ((rE & 63u) << 8);
              rE += lOff;
            }
            e = idx;
            if (kStamps) {
              wk_words += nloc - nloc0;
              for (uint32_t j = nloc0; j < nloc; j++)
                wk_hopsT += (uint32_t)__popc(rl32(rcL, j)) + (uint32_t)__popc(rl32(rcH, j));
            }
            if (nloc == 64u)
              rec_flush();
            if (st == 1u) {
              unusual = true;   // something the general code has to look at
              m = rl64(sw0, kk);
              break;
            }
            if (st == 3u)
              load_lrow(kk);
            if (kk > kkLast || e > eStop || nrec + nloc + 2u >= (uint32_t)kMxRecs)
              break;
          }
#undef MX_HOP
#undef MX_HOP_TEST
#undef MX_HOP_END
          r = kk * 64u + oo - wq0;
          rem = eEnd - e;
          if (kStamps)
            wk_tight += __builtin_readcyclecounter() - tt0;
          lrowK = kk;
          if (!unusual && !(kk == k && oo == o)) {
            curK = 0xffffffffu;
            continue;
          }
          if (!unusual)   // (nothing done -- the ring ends here: the general code takes an entry)
            m = rl64(sw0, kk);
          curK = kk;   // (an unusual entry lies in the word just walked: m and lrow are that word's)
        }
        const uint32_t q2 = r + wq0, o2 = q2 & 63u;   // (same word: the tight loop stops inside it)
        const uint64_t tt = m >> o2;
        const uint32_t z = min(min(tt ? (uint32_t)__ffsll((long long)tt) - 1u : 64u - o2, rem), S - r);
        if (z) {
          r += z;
          e += z;
          rem -= z;
          if (kStamps)
            wk_zruns++;
          continue;
        }
        if (kStamps)
          wk_hopsG++;
        if (e - eb >= 64u) {
          eb = e;
          ecv = e + lane < ringHi ? (uint32_t)ecls[(e + lane) & (uint32_t)(kMxRing - 1)] : 0xffffu;
        }
        const uint32_t ec = rl32(ecv, e - eb);
        const uint32_t ci = ec & 0xffu;
        uint32_t tl = kTNone;
        {   // its length from the rows in LDS
          const uint32_t col = ci < (uint32_t)kMaxCls ? sh_slot[ci] : 0xffu;
          if (col < (uint32_t)kMxColsAll)
            tl = row_at(r + 1u, col);
          if (kStamps) {   // why the tight loop left this entry to the general code
            if (!(ec & 0x8000u) && tl < kTNone)
              wk_gSat++;       // in the view, 254 bits and more (or a list of fewer than 64 entries)
            else if (tl < kTNone)
              wk_gView++;      // a column outside the list's two groups
            else if (col < (uint32_t)kMxCols && tl == kTInf)
              wk_gInf++;       // leaves the rows
          }
          if (tl == kTNone) {
            tl = chain_len_w(ci, r + 1u);
            if (kStamps && tl < kTNone)
              wk_gChain++;     // no column: length from the children's
          }
        }
        if (tl < kTNone) {
          if (lane == ns) {
            stE = lOff + e;
            stM = (r + 1u) | (ci << 16) | (1u << 24);
          }
          ns++;
          if (ns == 64u)
            flush();
          r += 1u + tl;
        }
        else {   // no column, or the split leaves the rows: walk into it
          const uint64_t packed = list[e];
          if (lane == 0)
            atomic_or64(sigbits + ((lOff + e) >> 6), 1ull << ((lOff + e) & 63));
          KidBox kb;
          kid_box_l(unpack_node(packed), kb);
          if (lane == 0) {
            MxCtx& nc = sh_ctx[1];
            nc.parent = packed;
            nc.kb = kb;
            nc.pc = (uint8_t)ci;
            nc.next = 0;
            nc.found = 0;
            sh_base = packed;
          }
          MX_WAVE_SYNC();
          depth = 2;
          r += 1;
        }
        e++;
        rem--;
        continue;
      }
      // ---- a set that is being walked into: its children from `next` on, one look-up each;
      //      lane k remembers what became of child k and writes its record afterwards
      const uint64_t ti0 = kStamps ? __builtin_readcyclecounter() : 0;
      if (kStamps)
        wk_rounds++;
      MxCtx& cx = sh_ctx[depth - 1];
      const uint32_t pc = cx.pc;
      const uint32_t nk = sh_cls[pc].nk;
      const uint64_t kcol = sh_kcol[pc];
      const uint64_t parent = cx.parent;
      const KidBox kb = cx.kb;
      const uint32_t k0 = cx.next;
      uint32_t found = cx.found, k = k0;
      uint32_t myAct = 0, myY = 0;   // 1: born insignificant (test bit at myY), 2: hopped over (split starts at myY)
      bool pushed = false, halted = false;
      while (k < nk) {
        if (r >= S) {
          halted = true;
          break;
        }
        const uint32_t col = (uint32_t)(kcol >> (8 * k)) & 0xffu;
        const bool coded = found || (k + 1 != nk);
        if (col == 0) {   // a single sample
          uint32_t sig = 1, sgn, len = 1;
          if (coded) {
            sig = bit_w(r);
            sgn = sig ? bit_w(r + 1) : 1u;
            len = 1u + sig;
          }
          else
            sgn = bit_w(r);
          if (lane == 0)
            pixel_event(kid_pixel_raster(b.tree, unpack_node(parent), kb, k), sig != 0, sgn);
          found |= sig;
          r += len;
          k++;
          continue;
        }
        uint32_t start = r, bit = 1;
        if (coded) {
          bit = bit_w(r);
          start = r + 1;
        }
        if (!bit) {
          if (lane == k) {
            myAct = 1;
            myY = r;
          }
          r += 1;
          k++;
          continue;
        }
        uint32_t tl = col < (uint32_t)kMxColsAll ? row_at(start, col) : kTNone;
        if (tl == kTNone)
          tl = chain_len_w(pc < b.tree.ncls ? (uint32_t)sh_cls[pc].kid[k] : 0xffu, start);
        found = 1;
        if (tl < kTNone) {
          if (lane == k) {
            myAct = 2;
            myY = start;
          }
          r = start + tl;
          k++;
          continue;
        }
        k++;   // walk into this child
        pushed = true;
        r = start;
        break;
      }
      // the records of the children handled in this round
      {
        const uint64_t rel = a + myY - phase0;
        const bool bornOk = myAct == 1 && born_counts(kb.kidlev, rel);
        const uint64_t bm = __ballot(bornOk);
        if (bm) {
          uint32_t base = 0;
          if (lane == 0)
            base = born_slots((uint32_t)__popcll(bm));
          base = rl32(base, 0);
          if (bornOk)
            write_born(base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull)), kb.kidlev, rel,
                       kid_packed(kb, lane));
        }
        const uint64_t im = __ballot(myAct == 2);
        if (im) {
          flush();
          if (myAct == 2) {
            const uint32_t idx = qn + (uint32_t)__popcll(im & ((1ull << lane) - 1ull));
            if (idx < Q) {
              qidA[idx] = kid_packed(kb, lane);
              qmetaA[idx] = myY | ((uint32_t)sh_cls[pc].kid[lane] << 16);
            }
          }
          qn += (uint32_t)__popcll(im);
        }
      }
      if (lane == 0) {
        cx.next = (uint8_t)k;
        cx.found = (uint8_t)found;
      }
      if (pushed && depth < kMxFrames) {
        const uint64_t kid = kid_packed(kb, k - 1);
        KidBox nkb;
        kid_box_l(unpack_node(kid), nkb);
        if (lane == 0) {
          MxCtx& nc = sh_ctx[depth];
          nc.parent = kid;
          nc.kb = nkb;
          nc.pc = sh_cls[pc].kid[k - 1];
          nc.next = 0;
          nc.found = 0;
        }
        depth++;
      }
      else if (pushed) {   // (cannot happen: the forest is not that deep)
        if (lane == 0)
          s.error = 1;
        r = S;
      }
      else if (halted) {
        MX_WAVE_SYNC();
        break;
      }
      else
        depth--;
      MX_WAVE_SYNC();
      if (kStamps)
        wk_into += __builtin_readcyclecounter() - ti0;
    }
    const uint64_t we0 = kStamps ? __builtin_readcyclecounter() : 0;
    // The region's end in the middle of a list, no set being walked into (the usual end): the state goes out from the
    // registers NOW -- the region after this one waits for it -- and what this workgroup still has to put away (staged
    // items, records, its own copy of the state) comes after; the chain's own publishing code below skips its stores.
    if (r >= S && depth == 1 && rem != 0 && curMode == kModeList) {
      unsigned long long f = tag;
      if (lane == 0)
        f |= (a + r - phase0) | ((unsigned long long)level << 40) | (1ull << 46) | ((unsigned long long)kModeList << 51);
      else if (lane == 1)
        f |= (unsigned long long)e;
      else if (lane == 6)
        f |= (unsigned long long)rem;
      else if (lane == 4)
        f |= pubBase;
      else if (lane == 5)
        f |= pubI;
      if (lane < 7)
        __hip_atomic_store(flags + (size_t)curRegion * kMxWordsPerRegion + lane, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(&s.mxHint),
                           tag | ((unsigned long long)level << 48) | ((unsigned long long)(curRegion & 0xfffffu) << 28) |
                               (unsigned long long)min(rem, 0xfffffffu),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_published = 1;
      }
    }
    flush();
    rec_flush();
    if (lane == 0) {
      sh_pos = a + r;
      sh_e = e;
      sh_rem = rem;
      sh_depth = depth;
      sh_qn[0] = qn;
      sh_nrec = nrec;
      if (qn > Q)
        s.error = 1;   // (only a damaged stream packs that many sets into a region)
    }
    MX_WAVE_SYNC();
    if (kStamps) {
      wk_total += __builtin_readcyclecounter() - wk0;
      wk_epi += __builtin_readcyclecounter() - we0;
      wk_calls++;
    }
  };

  // ---- expansion of the sets the walk hopped over, breadth first, all threads: leaf parents become leaf
  //      events, the other sets parse their children with the rows; significant children that are no leaf
  //      parents go to the next round
  auto expand_all = [&]() {
    __syncthreads();
    {   // the records of the walk's tight loop become work items
      const uint32_t nr = sh_nrec;
      for (uint32_t ri = (uint32_t)tid; ri < nr; ri += kMxThreads) {
        uint64_t cm = sh_recM[ri][0], im = sh_recM[ri][1];
        uint32_t slot = atomicAdd(&sh_qn[0], (uint32_t)__popcll(cm));
        // (the ordinal mask's bits are ordinals mod 64: turned so that bit 0 is the word's first ordinal)
        const uint32_t e0 = sh_recE[ri], rk = sh_recK[ri], pb = (rk & 0xffu) * 64u + 1u - wq0, r6 = (rk >> 8) & 63u;
        im = r6 ? (im >> r6) | (im << (64u - r6)) : im;
        while (cm && im) {
          const uint32_t pbit = (uint32_t)__builtin_ctzll(cm), ebit = (uint32_t)__builtin_ctzll(im);
          cm &= cm - 1;
          im &= im - 1;
          if (slot < Q) {
            qidA[slot] = e0 + ebit;
            qmetaA[slot] = (pb + pbit) | (0xffu << 16) | (1u << 24);
          }
          slot++;
        }
      }
      __syncthreads();
      if (sh_qn[0] > Q && tid == 0)
        s.error = 1;   // (only a damaged stream packs that many sets into a region)
    }
    uint32_t nin = min(sh_qn[0], Q);
    for (uint32_t round = 0; nin != 0; round++) {
      const uint64_t* qidIn = (round & 1u) ? qidB : qidA;
      const uint32_t* qmIn = (round & 1u) ? qmetaB : qmetaA;
      uint64_t* qidOut = (round & 1u) ? qidA : qidB;
      uint32_t* qmOut = (round & 1u) ? qmetaA : qmetaB;
      uint32_t* qnOut = &sh_qn[(round + 1u) & 1u];
      if (tid == 0)
        *qnOut = 0;
      __syncthreads();
      for (uint32_t i = (uint32_t)tid; i < nin; i += kMxThreads) {
        const uint64_t ident = qidIn[i];
        const uint32_t meta = qmIn[i];
        uint32_t ci = (meta >> 16) & 0xffu;
        uint32_t y = meta & 0xffffu;
        uint64_t packed = ident;
        if (meta >> 24) {
          packed = lisCur[ident];
          atomic_or64(sigbits + (ident >> 6), 1ull << (ident & 63));
        }
        const Node nd = unpack_node(packed);
        if (meta >> 24)
          ci = nd.grid < b.tree.ngrids ? node_cls_l(nd) : 0xffu;
        if (ci >= b.tree.ncls)
          continue;   // (cannot happen)
        const uint32_t nk = sh_cls[ci].nk;
        if (sh_cls[ci].h == 0) {
          leaf_event(nd, y, nk);
          continue;
        }
        KidBox kb;
        kid_box_l(nd, kb);
        const uint64_t kc = sh_kcol[ci];
        uint32_t found = 0;
        for (uint32_t k = 0; k < nk; k++) {
          const uint32_t coded = found | (uint32_t)(k + 1 != nk);
          const uint32_t col = (uint32_t)(kc >> (8 * k)) & 0xffu;
          const uint32_t bit = coded ? bit_at(y) : 1u;
          const uint32_t start = y + coded;
          if (col == 0) {   // a single sample: its sign follows
            pixel_event(kid_pixel_raster(b.tree, nd, kb, k), bit != 0, bit ? bit_at(start) : 1u);
            found |= bit;
            y = start + bit;
            continue;
          }
          if (!bit) {
            record_born(kb.kidlev, a + y, kid_packed(kb, k));
            y += 1;
            continue;
          }
          found = 1;
          // (an implied child is the last one: nothing follows it, its length is not needed)
          const uint32_t tl = (coded && col < (uint32_t)kMxColsAll) ? row_at(min(start, W + 2), col) : 0u;
          y = min(start + (tl < kTNone ? tl : 0u), W + 2);
          if (col < 4u)
            leaf_event(unpack_node(kid_packed(kb, k)), start, 1u << col);
          else {
            const uint32_t slot = atomicAdd(qnOut, 1u);
            if (slot < Q) {
              qidOut[slot] = kid_packed(kb, k);
              qmOut[slot] = start | ((uint32_t)sh_cls[ci].kid[k] << 16);
            }
          }
        }
      }
      __syncthreads();
      if (*qnOut > Q && tid == 0)
        s.error = 1;   // (only a damaged stream packs that many sets into a region)
      nin = min(*qnOut, Q);
      if (round > 8)
        break;   // (h <= 3: at most four rounds)
    }
    __syncthreads();
  };

  uint64_t tk0 = 0;
  for (;;) {
    // ---- ticket
    if (tid == 0) {
      const bool over = __hip_atomic_load(&s.hiPlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p + 1;
      sh_ticket = over ? kNoTicket : atomicAdd(&s.hiTicket, 1u);
      // Which columns this region's rows need: sets of at most so many steps above the leaf parents.  The lists come
      // smallest sets first, so most regions lie in the lists of the leaf parents and of the sets made of those, whose
      // rows need no or few chains of look-ups.  The chain leaves a hint behind (list level, region, entries left of
      // that list); a guess that falls short costs speed only: a column that is "not computed" sends the walk into
      // the set, child by child.
      {
        const unsigned long long hint = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&s.mxHint), __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
        int lv = -1;
        uint32_t left = 0, at = 0;
        if ((hint >> kMxTagShift) == (unsigned long long)(p + 1)) {
          lv = (int)((hint >> 48) & 63u);
          at = (uint32_t)(hint >> 28) & 0xfffffu;
          left = (uint32_t)hint & 0xfffffffu;
        }
        if (lv < 0 || lv >= (int)nlevels) {
          lv = next_level((int)nlevels);
          left = lv >= 0 ? sh_len[lv] : 0u;
          at = 0;
        }
        uint32_t hc = 0;
        if (lv >= 0) {
          hc = sh_lgrp[lv] >> 4;
          // entries of that list the regions in between cannot have used up (an entry takes a bit at least)
          const uint32_t span = (sh_ticket >= at ? sh_ticket - at + 2u : 2u) * (b.mxS + b.mxM);
          if (left < span) {   // the next lists (larger sets) may begin in this region
            int l2 = lv;
            for (int k = 0; k < 3; k++) {
              l2 = next_level(l2);
              if (l2 < 0)
                break;
              hc = max(hc, (uint32_t)(sh_lgrp[l2] >> 4));
            }
          }
        }
        if (twoD)
          hc = 4;   // (the subbands the type-I set releases are tested behind the lists)
        sh_hcap = hc;
      }
      sh_qn[0] = sh_qn[1] = 0;
      sh_nrec = 0;
      sh_go = sh_goDone = 0;
      sh_stop = 0;
      sh_published = 0;
      sh_abort = 0;
      sh_needFill = 0;
      if (kStamps)
        tk0 = __builtin_readcyclecounter();
    }
    __syncthreads();
    const uint32_t i = sh_ticket;
    if (i == kNoTicket)
      break;
    // (tickets are taken ahead of the chain: one past the flags is past the stream and all its padding, no
    //  phase gets there)
    if (((size_t)i + 1) * kMxWordsPerRegion > b.hiFlagStride)
      break;
    a = phase0 + (uint64_t)i * S;
    wq0 = (uint32_t)(a & 63);
    {
      const uint64_t w0 = a >> 6;
      for (uint32_t k = tid; k < kWords; k += kMxThreads) {
        const uint64_t idx = w0 + k;
        wbits[k] = idx < nwordsAvail ? words[idx] : 0ull;
      }
    }
    __syncthreads();
    build_rows(sh_hcap);
    __syncthreads();
    if (kStamps && tid == 0)
      sh_tk[0] = __builtin_readcyclecounter();

    // ---- the chain
    if (wave == 0) {
      // While this region waits for its turn: the classes of the list entries its walk will most likely visit.  The
      // region before this one starts where the one before THAT ended, and ends at most S entries further on (an entry
      // takes a bit at least): as soon as that state is out, the 2 S + 128 entries from there go into the ring.
      {
        unsigned long long f = 0;
        bool have = false;
        if (i >= 2) {
          uint32_t spins = 0;
          uint64_t spinT0 = 0;
          for (;;) {
            if (lane < 2)
              f = __hip_atomic_load(flags + (size_t)(i - 2) * kMxWordsPerRegion + lane, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
            have = __ballot(lane < 2 && (f >> kMxTagShift) == (unsigned long long)(p + 1)) == 3ull;
            if (have)
              break;
            if ((++spins & 15u) == 0 &&
                __hip_atomic_load(&s.hiPlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p + 1)
              break;
            if (spin_expired(spins, spinT0))
              break;   // (the look-back below gives up loudly)
            __builtin_amdgcn_s_sleep(1);
          }
        }
        const unsigned long long f0 = __shfl(f, 0, 64), f1 = __shfl(f, 1, 64);
        have = have && !(f0 & kMxOver) && ((uint32_t)(f0 >> 51) & 3u) == kModeList;
        const uint32_t l = (uint32_t)(f0 >> 40) & 63u, e0 = (uint32_t)f1;
        uint32_t lo = 0, hi = 0;
        if (have && l < nlevels && e0 <= sh_len[l]) {
          lo = e0;
          hi = min(sh_len[l], e0 + 2u * S + 128u);
        }
        if (lane == 0) {
          sh_preLevel = (have && hi > lo) ? l : 0xffffffffu;
          sh_preLo = lo;
          sh_preHi = hi;
          __hip_atomic_store(&sh_go, (have && hi > lo) ? 2u : 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        MX_WAVE_SYNC();
      }
    }
    // (four wavefronts put the classes into the ring: one alone takes longer than the region before this one walks)
    if (wave < (uint32_t)kMxPreWaves) {
      // (no bound on this wait: the first wavefront gets to its store whatever happens -- its own waits above are bounded --
      //  and a helper that gave up early would leave a ring that is announced as filled and is not)
      uint32_t go = 0;
      while ((go = __hip_atomic_load(&sh_go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0)
        __builtin_amdgcn_s_sleep(2);
      if (go == 2)
        fill_ring(sh_preLevel, sh_preLo, sh_preHi, (uint32_t)tid, (uint32_t)kMxPreWaves * 64u);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0)
        atomicAdd(&sh_goDone, 1u);
    }
    if (wave == 0) {
      {
        // (bounded by wall time like the look-back waits, common.h: a slow, shared or profiled device must not turn
        //  a spin count into an early exit)
        uint32_t spins = 0;
        uint64_t spinT0 = 0;
        while (__hip_atomic_load(&sh_goDone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)kMxPreWaves &&
               !spin_expired(++spins, spinT0))
          __builtin_amdgcn_s_sleep(1);
      }
      // look back (lanes 0..6 take one word each)
      if (lane < 7) {
        unsigned long long f = 0;
        if (i > 0) {
          uint32_t spins = 0;
          uint64_t spinT0 = 0;
          for (;;) {
            f = __hip_atomic_load(flags + (size_t)(i - 1) * kMxWordsPerRegion + lane, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
            if ((f >> kMxTagShift) == (unsigned long long)(p + 1))
              break;
            if ((++spins & 15u) == 0 &&
                __hip_atomic_load(&s.hiPlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p + 1) {
              sh_abort = 1;   // the phase is over
              break;
            }
            if (spin_expired(spins, spinT0)) {   // (a minute of wall time: the device has stopped making progress)
              s.error = kErrLookBackTimeout;
              __hip_atomic_store(&s.hiPlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              sh_abort = 1;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        sh_in[lane] = f;
      }
      MX_WAVE_SYNC();
      if (kStamps && lane == 0)
        sh_tk[1] = __builtin_readcyclecounter();
      if (kStamps)
        wk_seen = __builtin_readcyclecounter();
      // every lane works out the same state; lane 0 writes it
      uint32_t stop = 0, mode = kModeList, level = 0, e = 0, rem = 0, iJ = 0, iPart = 0, iCounter = 0, iNeed = 1;
      int depth = 1;
      uint64_t pos = phase0, base = 0;
      if (i == 0) {
        const int lv = next_level((int)nlevels);
        iPart = s.iPart;
        if (lv >= 0) {
          level = (uint32_t)lv;
          rem = sh_len[lv];
        }
        // (no list holds an entry: level 0 with nothing left -- the chain below ends the phase, or goes on
        //  to the type-I set)
      }
      else if (sh_abort || (sh_in[0] & kMxOver))
        stop = 1;
      else {
        // (the same in every lane: as scalars, so that the rest is scalar code)
        auto uni = [&](unsigned long long v) -> unsigned long long {
          return (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
                 ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
        };
        const unsigned long long f0 = uni(sh_in[0]), f1 = uni(sh_in[1]), f5 = uni(sh_in[5]);
        pos = phase0 + (f0 & ((1ull << 40) - 1ull));
        level = (uint32_t)(f0 >> 40) & 63u;
        depth = (int)((f0 >> 46) & 31u);
        mode = (uint32_t)(f0 >> 51) & 3u;
        e = (uint32_t)f1;
        rem = (uint32_t)uni(sh_in[6]);
        base = uni(sh_in[4]) & ((1ull << kMxTagShift) - 1ull);
        iJ = (uint32_t)f5 & 3u;
        iPart = (uint32_t)(f5 >> 2) & 63u;
        iCounter = (uint32_t)(f5 >> 8) & 3u;
        iNeed = (uint32_t)(f5 >> 10) & 1u;
        // (a state that does not add up cannot be followed: give up loudly)
        bool bad = level >= nlevels || depth < 1 || depth > kMxFrames || pos < a || pos >= a + W;
        if (mode == kModeList)
          bad = bad || e + rem != sh_len[level];
        else
          bad = bad || !twoD;
        if (bad) {
          if (lane == 0) {
            s.error = 1;
            __hip_atomic_store(&s.hiPlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          stop = 1;
          depth = 1;
        }
        // rebuild the stack of sets being walked into: frame d holds the children of the set that frame
        // d - 1 entered (frame 1: of the entry at the bottom)
        uint64_t parent = base;
        for (int d = 1; d < depth && !stop; d++) {
          const unsigned long long fw = d <= 11 ? sh_in[2] : sh_in[3];
          const uint32_t fr = (uint32_t)(fw >> (5 * ((d - 1) % 11))) & 31u;
          const Node pn = unpack_node(parent);
          if (pn.grid + 1u >= b.tree.ngrids + 1u || pn.grid >= (uint32_t)kMxLdsGrids) {   // (cannot happen)
            stop = 1;
            if (lane == 0)
              s.error = 1;
            break;
          }
          KidBox kb;
          kid_box_l(pn, kb);
          const uint32_t pc = node_cls_l(pn);
          if (lane == 0) {
            MxCtx& cx = sh_ctx[d];
            cx.parent = parent;
            cx.kb = kb;
            cx.pc = (uint8_t)pc;
            cx.next = (uint8_t)(fr & 15u);
            cx.found = (uint8_t)(fr >> 4);
          }
          if (d + 1 < depth) {
            if ((fr & 15u) == 0 || pc >= b.tree.ncls) {   // (cannot happen: a frame above was entered through a child)
              stop = 1;
              if (lane == 0)
                s.error = 1;
              break;
            }
            parent = kid_packed(kb, (fr & 15u) - 1u);
          }
        }
        if (stop == 1 && lane == 0)
          __hip_atomic_store(&s.hiPlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // (the classes of the list's entries from e on: in the ring already when the guess above held)
      const bool pre = mode == kModeList && level == sh_preLevel && e >= sh_preLo &&
                       min(sh_len[level < nlevels ? level : 0], e + S + 128u) <= sh_preHi;
      const bool needFill = stop == 0 && mode == kModeList && !pre;
      if (lane == 0) {
        sh_pos = pos;
        sh_base = base;
        sh_level = level;
        sh_e = e;
        sh_rem = rem;
        sh_depth = depth;
        sh_mode = mode;
        sh_iJ = iJ;
        sh_iPart = iPart;
        sh_iCounter = iCounter;
        sh_iNeed = iNeed;
        sh_stop = stop;
        sh_needFill = needFill ? 1u : 0u;
        if (pre)
          sh_ringHi = sh_preHi;
      }
      MX_WAVE_SYNC();
      if (kStamps && wk_seen)
        wk_dec += __builtin_readcyclecounter() - wk_seen;
    }
    for (;;) {
      __syncthreads();
      if (sh_stop)
        break;
      if (sh_needFill) {
        const uint64_t tf0 = kStamps ? __builtin_readcyclecounter() : 0;
        if (kStamps)
          wk_fills++;
        const uint32_t l = sh_level, e0 = sh_e, n = sh_len[l];
        const uint32_t to = min(n, e0 + S + 128u);
        fill_ring(l, min(e0, to), to, (uint32_t)tid, kMxThreads);
        __syncthreads();
        if (tid == 0) {
          sh_ringHi = to;
          sh_needFill = 0;
          sh_preLevel = 0xffffffffu;
        }
        if (kStamps)
          wk_fill += __builtin_readcyclecounter() - tf0;
      }
      if (wave != 0)
        continue;
      MX_WAVE_SYNC();   // (what thread 0 wrote above is this wavefront's own LDS traffic)
      // ---- first wavefront: on until the region ends, the phase ends, or a new list needs its classes
      for (;;) {
        uint32_t mode = __builtin_amdgcn_readfirstlane(sh_mode);
        uint32_t r = (uint32_t)(sh_pos - a);
        int depth = __builtin_amdgcn_readfirstlane(sh_depth);
        bool over = false;
        if (mode == kModeList && depth <= 1 && __builtin_amdgcn_readfirstlane(sh_rem) == 0) {
          // this list is through: the next one (no bits are read for that)
          const int lv = next_level((int)sh_level);
          if (lv >= 0) {
            if (lane == 0) {
              sh_level = (uint32_t)lv;
              sh_e = 0;
              sh_rem = sh_len[lv];
              sh_depth = 1;
              sh_needFill = 1;
            }
            MX_WAVE_SYNC();
            if (r < S)
              break;   // (all hands: the new list's classes)
          }
          else if (twoD) {
            if (lane == 0) {
              sh_mode = kModeITest;
              sh_depth = 1;
            }
            MX_WAVE_SYNC();
            continue;
          }
          else
            over = true;
        }
        if (!over && r >= S) {
          // ---- publish the state at the end of the region
          const uint32_t dpt = (uint32_t)max(__builtin_amdgcn_readfirstlane(sh_depth), 1);
          unsigned long long fr0 = 0, fr1 = 0;
          for (uint32_t d = 1; d < dpt; d++) {
            const unsigned long long v = (unsigned long long)((sh_ctx[d].next & 15u) | ((sh_ctx[d].found ? 1u : 0u) << 4));
            if (d <= 11)
              fr0 |= v << (5 * (d - 1));
            else
              fr1 |= v << (5 * (d - 12));
          }
          unsigned long long f = tag;
          if (lane == 0)
            f |= (sh_pos - phase0) | ((unsigned long long)sh_level << 40) | ((unsigned long long)dpt << 46) |
                 ((unsigned long long)sh_mode << 51);
          else if (lane == 1)
            f |= (unsigned long long)sh_e;
          else if (lane == 6)
            f |= (unsigned long long)sh_rem;
          else if (lane == 2)
            f |= fr0;
          else if (lane == 3)
            f |= fr1;
          else if (lane == 4)
            f |= sh_base & ((1ull << kMxTagShift) - 1ull);
          else if (lane == 5)
            f |= (unsigned long long)sh_iJ | ((unsigned long long)sh_iPart << 2) |
                 ((unsigned long long)sh_iCounter << 8) | ((unsigned long long)sh_iNeed << 10);
          const bool done = __builtin_amdgcn_readfirstlane(sh_published) != 0;   // (walk() has published this very state)
          if (lane < 7 && !done)
            __hip_atomic_store(flags + (size_t)i * kMxWordsPerRegion + lane, f, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          if (lane == 0)
            sh_stop = 3;
          if (lane == 0 && !done) {
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(&s.mxHint),
                               tag | ((unsigned long long)sh_level << 48) | ((unsigned long long)(i & 0xfffffu) << 28) |
                                   (unsigned long long)(sh_mode == kModeList ? min(sh_rem, 0xfffffffu) : 0u),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          MX_WAVE_SYNC();
          break;
        }
        if (!over && mode == kModeITest) {
          // the type-I set: nothing left of it, or its test bit says the rest is insignificant: the phase ends
          uint32_t iPart = __builtin_amdgcn_readfirstlane(sh_iPart);
          if (iPart == 0)
            over = true;
          else {
            uint32_t bit = 1;
            if (__builtin_amdgcn_readfirstlane(sh_iNeed)) {
              bit = bit_at(r);
              r += 1;
            }
            if (lane == 0)
              sh_pos = a + r;
            if (!bit)
              over = true;
            else if (lane == 0) {
              sh_iJ = 0;
              sh_iCounter = 0;
              sh_mode = kModeISub;
            }
            MX_WAVE_SYNC();
            if (!over)
              continue;
          }
        }
        if (!over && mode == kModeISub) {
          const uint32_t iJ = __builtin_amdgcn_readfirstlane(sh_iJ), iPart = __builtin_amdgcn_readfirstlane(sh_iPart);
          if (iJ >= 3u) {   // the three subbands are through: what is left of the type-I set comes next
            if (lane == 0) {
              sh_iPart = iPart - 1u;
              sh_iNeed = sh_iCounter != 0 ? 1u : 0u;
              sh_mode = kModeITest;
            }
            MX_WAVE_SYNC();
            continue;
          }
          const uint64_t root = b.iRoots[(size_t)(b.iLevels - iPart) * 3 + iJ];
          if (lane == 0)
            sh_iJ = iJ + 1u;
          if (root == ~0ull) {   // (an empty subband)
            MX_WAVE_SYNC();
            continue;
          }
          // the subband is tested like a list of one entry
          const uint32_t bit = bit_at(r);
          if (!bit) {
            if (lane == 0) {
              record_born(iPart, a + r, root);
              sh_pos = a + r + 1u;
            }
            MX_WAVE_SYNC();
            continue;
          }
          const Node rn = unpack_node(root);
          const uint32_t ci = node_cls_l(rn);
          const uint32_t col = ci < (uint32_t)kMaxCls ? sh_slot[ci] : 0xffu;
          uint32_t tl = col < (uint32_t)kMxColsAll ? row_at(r + 1u, col) : kTNone;
          if (tl == kTNone)
            tl = chain_len(ci, r + 1u);
          if (tl < kTNone) {
            if (lane == 0) {
              const uint32_t qn = sh_qn[0];
              if (qn < Q) {
                qidA[qn] = root;
                qmetaA[qn] = (r + 1u) | (ci << 16);
              }
              sh_qn[0] = qn + 1u;
              sh_iCounter = sh_iCounter + 1u;
              sh_pos = a + r + 1u + tl;
            }
            MX_WAVE_SYNC();
            continue;
          }
          KidBox kb;
          kid_box_l(rn, kb);
          if (lane == 0) {
            MxCtx& nc = sh_ctx[1];
            nc.parent = root;
            nc.kb = kb;
            nc.pc = (uint8_t)ci;
            nc.next = 0;
            nc.found = 0;
            sh_base = root;
            sh_iCounter = sh_iCounter + 1u;
            sh_pos = a + r + 1u;
            sh_depth = 2;
            sh_rem = 0;
            sh_e = 0;
            sh_mode = kModeISubWalk;
          }
          MX_WAVE_SYNC();
          continue;
        }
        if (over) {
          // ---- the phase ends here
          if (lane == 0) {
            s.hiEnd = sh_pos;
            if (twoD)
              s.iPart = sh_iPart;
          }
          if (lane < 7)
            __hip_atomic_store(flags + (size_t)i * kMxWordsPerRegion + lane, tag | kMxOver, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          if (lane == 0) {
            __hip_atomic_store(&s.hiPlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_stop = 2;
          }
          MX_WAVE_SYNC();
          break;
        }
        // ---- list entries, or the children of the sets being walked into
        curRegion = i;
        curMode = mode;
        if (kStamps && wk_seen) {
          wk_toWalk += __builtin_readcyclecounter() - wk_seen;
          wk_seen = 0;
        }
        walk();
        if (mode == kModeISubWalk && __builtin_amdgcn_readfirstlane(sh_depth) == 0) {
          if (lane == 0) {
            sh_depth = 1;
            sh_mode = kModeISub;
          }
          MX_WAVE_SYNC();
        }
      }
    }
    if (kStamps && tid == 0)
      sh_tk[2] = __builtin_readcyclecounter();
    const uint32_t stop = sh_stop;
    if (stop != 1)
      expand_all();
    if (kStamps && tid == 0 && b.lisStamps) {
      unsigned long long* out = reinterpret_cast<unsigned long long*>(b.lisStamps + (size_t)c * 64);
      const uint64_t t3 = __builtin_readcyclecounter();
      atomicAdd(out + 0, 1ull);
      atomicAdd(out + 1, (unsigned long long)(sh_tk[0] - tk0));        // load + rows
      atomicAdd(out + 2, (unsigned long long)(sh_tk[1] - sh_tk[0]));   // look-back wait
      atomicAdd(out + 3, (unsigned long long)(sh_tk[2] - sh_tk[1]));   // on the chain
      atomicAdd(out + 4, (unsigned long long)(t3 - sh_tk[2]));         // expansion
      atomicAdd(out + 5, (unsigned long long)wk_fill);
      atomicAdd(out + 6, (unsigned long long)wk_fills);
      atomicAdd(out + 7, (unsigned long long)wk_total);
      atomicAdd(out + 8, (unsigned long long)wk_calls);
      atomicAdd(out + 9, (unsigned long long)wk_tight);
      atomicAdd(out + 10, (unsigned long long)wk_hopsT);
      atomicAdd(out + 11, (unsigned long long)wk_hopsG);
      atomicAdd(out + 12, (unsigned long long)wk_rounds);
      atomicAdd(out + 13, (unsigned long long)wk_into);
      atomicAdd(out + 14, (unsigned long long)wk_words);
      atomicAdd(out + 15, (unsigned long long)wk_zruns);
      atomicAdd(out + 16, (unsigned long long)wk_gSat);
      atomicAdd(out + 17, (unsigned long long)wk_gView);
      atomicAdd(out + 18, (unsigned long long)wk_gInf);
      atomicAdd(out + 19, (unsigned long long)wk_gChain);
      atomicAdd(out + 20, (unsigned long long)wk_enters);
      atomicAdd(out + 22, (unsigned long long)wk_toWalk);
      atomicAdd(out + 23, (unsigned long long)wk_pro);
      atomicAdd(out + 24, (unsigned long long)wk_epi);
      atomicAdd(out + 25, (unsigned long long)wk_dec);
      wk_toWalk = wk_pro = wk_epi = wk_dec = 0;
      atomicAdd(out + 21, (unsigned long long)wk_reloads);
      wk_gSat = wk_gView = wk_gInf = wk_gChain = wk_enters = wk_reloads = 0;
      wk_fill = wk_total = wk_tight = wk_into = 0;
      wk_fills = wk_calls = wk_hopsT = wk_hopsG = wk_rounds = wk_words = wk_zruns = 0;
    }
    if (stop == 1 || stop == 2)
      break;
    __syncthreads();   // LDS is reused by the next region
  }
  __syncthreads();
  if (tid == 0 && blockIdx.x < 8) {
    s.hiBornCnt[blockIdx.x] = min(min(sh_segBorn, sh_segBornEnd), b.bornSeg);
    s.hiLeafCnt[blockIdx.x] = min(sh_segLeaf, b.leafSeg);
  }
}

}  // namespace

uint32_t mx_smem_bytes(uint32_t S, uint32_t M, uint32_t Q)
{
  const uint32_t W = S + M;
  const uint32_t kWords = ((W >> 6) + 5u) & ~1u;
  return kWords * 8u + (((W + 3u) * 17u * 2u + 15u) & ~15u) + 2u * Q * 12u + kMxRing * 2u;
}

int prepare_lis_mx(const DecBuffers& b)
{
  if (set_max_dyn_lds(reinterpret_cast<const void*>(&k_lis_mx<false>), (int)b.mxSmemBytes) ||
      set_max_dyn_lds(reinterpret_cast<const void*>(&k_lis_mx<true>), (int)b.mxSmemBytes))
    return -1;
  return 0;
}

int launch_lis_mx(hipStream_t stream, const DecBuffers& b, int p, uint32_t groups, bool stamps)
{
  if (stamps)
    LAUNCH_K(k_lis_mx<true>, dim3(groups, b.nchunks), dim3(kMxThreads), b.mxSmemBytes, stream, b, p);
  else
    LAUNCH_K(k_lis_mx<false>, dim3(groups, b.nchunks), dim3(kMxThreads), b.mxSmemBytes, stream, b, p);
  return 0;
}

}  // namespace sperrhip
