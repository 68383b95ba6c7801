// speck_enc.h -- buffers and launcher of the data-parallel SPECK3D encoder (speck_enc.hip)
#ifndef SPERR_AMD_SPECK_ENC_H
#define SPERR_AMD_SPECK_ENC_H

#include <vector>

#include "common.h"

namespace sperrhip {

// Device pointers of one batch of `nchunks` equally shaped chunks; chunk c's slice of an array is
// base + c * stride (strides in elements).
struct EncBuffers {
  spk::Tree tree;
  uint32_t nchunks;
  CoderState* cst;
  EncState* st;
  // quantiser output
  const void* coef;            // uint32_t or uint64_t magnitudes, raster order
  size_t coefStride;
  const uint64_t* sign;        // bit i = (value i >= 0)
  size_t signStride;
  const int8_t* msb;           // msb position of every magnitude (-1 for zero)
  int8_t* bplane;              // plane at which the pixel is first tested (-1: never)
  size_t pixStride;
  // significance pyramid
  int8_t* M;
  uint32_t* E;
  uint64_t* opos;
  uint32_t* bucket;            // flat ids of the splitting sets, grouped by plane
  uint32_t* koff;              // start of a set's own split inside its parent's split
  // chain[node] (sets above the deepest depth): the set that starts this set's chain of nested
  // splits (the list entry whose position is known) in the low half, the bits between that
  // entry's split and this set's own in the high half
  uint64_t* chain;
  // leaf sets of oct grids: children with msb == the set's msb | their sign bits << 8
  uint16_t* leafDesc;
  size_t nodeStride;
  // LIS, double buffered; level l occupies [levelOff[l], levelOff[l+1])
  uint64_t* lis[2];
  size_t lisStride;
  const uint32_t* levelOff;
  // list tiles in traversal order (deepest level first)
  uint32_t nListTiles;
  const uint16_t* tileLevel;
  const uint32_t* tileStart;
  const uint32_t* levelFirstTile;
  const uint32_t* levelNumTiles;
  uint64_t* tileBits;
  uint32_t* tileSurv;
  uint64_t* tileBitsOff;
  uint32_t* tileSurvOff;
  size_t tileStride;
  // newborn insignificant sets of one plane
  uint64_t* bornPacked;
  uint64_t* bornPosLev;
  size_t bornStride;
  // birth masks: one bit per stream position of the LIS phase, per level that can hold sets
  const uint8_t* levelSlot;    // level -> slot (0xff: none)
  const uint8_t* slotLevel;    // slot -> level
  uint32_t nSlots;
  uint32_t maskWords;
  uint64_t* mask;
  uint32_t* maskPrefix;        // popcount prefix of a slot's mask, one word per FOUR mask words (round 6: a word each before,
                               //   14 MiB per 256^3 chunk): rank = prefix of the group + the group's words in front + the bits in front
  uint32_t prefWords;          //   prefix words per slot ((maskWords + 3) / 4)
  size_t prefStride;           //   ... per chunk (nSlots * prefWords)
  size_t maskStride;           // nSlots * maskWords
  // pixel-pass census
  uint32_t nPixTiles;
  uint32_t* pixCnt;            // [plane*2 + phase][tile]
  uint32_t* pixOff;
  size_t pixCntStride;
  // output bit buffer (zero-initialised)
  uint64_t* stream;
  size_t streamStride;
  // 2D coder (spk::kTree2D): packed roots of the subbands the type-I set releases, three per level
  // from the coarsest on (~0: empty); iLevels: transform levels
  const uint64_t* iRoots;
  uint32_t iLevels;
};

struct EncPlanHost {
  const uint64_t* d_initLIS;
  const uint32_t* d_initLen;
  const uint32_t* d_depthBlocks;
  std::vector<uint32_t> depthBlockOff;   // [maxDepth + 1]
  uint32_t nsets;
  // optional: a second stream (and two events) of the caller's: the census of the pixel passes, which
  // only needs the pyramid, runs there beside the chain pass -- eight small dependent launches (round 3)
  hipStream_t side = nullptr;
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  // optional (round 5): the planes that can hold work, asked of the device before the plane loop is enqueued.
  // A fixed-rate stream runs out of budget long before plane 0 (the bench volume: 18 of 32 planes), and the seven
  // launches of a plane without work still cost 35 to 100 us on the batch's serial chain.  The pixel passes' bit
  // counts (the census) bound the planes from below: where they alone exceed the budget, no later plane runs;
  // the chunks' plane counts bound them from above.  k_enc_bound leaves both in d_bound[0..1], a copy lands in
  // h_bound (pinned) and evBound is recorded: launch_speck_encode_head() ends there, and
  // launch_speck_encode_planes() -- called once the caller has enqueued what else it has -- waits for the
  // event and launches the planes in between only.  All three null: every plane is launched.
  uint32_t* d_bound = nullptr;
  uint32_t* h_bound = nullptr;
  hipEvent_t evBound = nullptr;
};

int launch_speck_encode(hipStream_t stream, const EncBuffers& b, const EncPlanHost& plan,
                        uint64_t raw_budget, bool rate_mode, bool wide_pass);
// the same in two steps (plan.d_bound / h_bound / evBound set): everything up to the plane loop; the planes and the rest
int launch_speck_encode_head(hipStream_t stream, const EncBuffers& b, const EncPlanHost& plan,
                             uint64_t raw_budget, bool rate_mode, bool wide_pass);
int launch_speck_encode_planes(hipStream_t stream, const EncBuffers& b, const EncPlanHost& plan,
                               uint64_t raw_budget, bool rate_mode, bool wide_pass);

}  // namespace sperrhip
#endif
