// outlier.h -- point-wise error mode: outlier detection and the SPECK1D coder of the outlier list
// (outlier.hip).  Reference: src/SPECK_FLT.cpp:461-486,573-584, src/Outlier_Coder.cpp,
// src/SPECK1D_INT.cpp, src/SPECK1D_INT_ENC.cpp, src/SPECK1D_INT_DEC.cpp.
#ifndef SPERR_AMD_OUTLIER_H
#define SPERR_AMD_OUTLIER_H

#include "common.h"

namespace sperrhip {

constexpr int kO1MaxLevels = 40;   // lists of the 1D coder: num_of_partitions(N) + 1, N < 2^32

// per chunk, device resident
struct OutlierChunk {
  // encoder
  uint32_t flagged;              // values whose reconstruction error exceeds the tolerance
  uint32_t count;                // those of them whose quantised magnitude is not zero
  unsigned long long maxErrKey;  // largest |error| (bits of the double: non-negative values order as integers)
  unsigned long long widthMask;  // integer width the reference stores the magnitudes in (Outlier_Coder.cpp:88-100)
  unsigned long long maxMag;     // largest quantised magnitude
  // both directions
  int32_t nbp;                   // bit planes of the 1D stream
  uint32_t error;
  uint64_t total_bits;
  // decoder
  uint32_t has;                  // the chunk carries a complete outlier stream
  uint32_t found;                // significant values decoded
  uint64_t streamOff;            // byte offset of the stream's 9-byte header inside the container
};

// arrays of one batch; every per-chunk array is [chunk][stride]
struct OutlierBufs {
  uint32_t nchunks, N, nw;       // nw = mask words per chunk
  OutlierChunk* oc;
  size_t wordStride;             // >= nw + 2
  uint32_t* outPre;              // encoder: outliers before each 64-value word ([nw] = their number)
  uint64_t* signMask;            //          positions of the non-negative ones
  uint64_t* maskGE;              //          outliers at or above the current threshold, by position
  uint64_t* maskEQ;              //          ... whose msb is the current plane
  uint32_t* cpos;                //          set bits of maskGE before each word ([nw] = all)
  uint64_t* lip;                 // LIP bitmask (SPECK_INT.cpp:120-125)
  uint64_t* lsp;                 // decoder: LSP bitmask
  size_t kStride;                // outliers (encoder) / significant values (decoder) per chunk
  uint32_t* pos;                 // encoder: position of every outlier, ascending; decoder: of every value found
  uint64_t* mag;                 // encoder: quantised magnitudes
  uint8_t* sgn;                  // encoder: 1 = non-negative; decoder: plane | sign << 7
  uint8_t* msb;                  // encoder: msb of every magnitude
  uint32_t* posGE;               // encoder, per plane: positions / signs of the outliers at or
  uint8_t* sgnGE;                //   above the threshold, in order
  uint32_t nlists;               // LIS levels (src/SPECK1D_INT.cpp:19-34)
  uint32_t levelOff[kO1MaxLevels + 1];   // first entry of each level inside a chunk's list storage
  size_t runStride;
  uint64_t* runs;                // start | length << 32
  size_t streamStride;           // 64-bit words
  uint64_t* stream;
  uint64_t* planeBits;           // decoder: refinement bits of every plane, [plane][wordStride]
  size_t planeStride;            // = planes * wordStride
};

// encoder, three passes over the reconstructed chunk (vals) and the input volume:
//   pass 0: flagged, maxErrKey;  pass 1: signMask + per-word counts (into outPre);  pass 2 (after
//   launch_outlier_prefix): pos / mag / sgn / msb, maxMag
template <typename T>
int launch_outlier_scan(hipStream_t st, int pass, const T* vol, VolDesc vd, const ChunkGeom* geom,
                        const uint32_t cdims[3], const double* vals, size_t valsStride,
                        const CoderState* cst, double tol, const OutlierBufs& b);
int launch_outlier_prefix(hipStream_t st, const OutlierBufs& b);
int launch_speck1d_encode(hipStream_t st, const OutlierBufs& b);
// chunk slot = {u8 planes, u64 total_bits, payload}; lens2[gid] = its length (0: no outliers)
int launch_outlier_stream_out(hipStream_t st, const OutlierBufs& b, const uint32_t* gids,
                              uint8_t* slots, const uint64_t* slotOff, uint64_t* lens2);

// decoder: loads the streams (oc[c].has / streamOff / nbp / total_bits set by the host) and decodes
// them: positions, planes and signs of the values found (pos, sgn, planeBits; oc[c].found).  Needs
// nothing but the container, so it can run beside the chunks' SPECK3D decoder on another stream
int launch_speck1d_decode(hipStream_t st, const OutlierBufs& b, const uint8_t* container);
// adds the correctors of the values found to vals (tolerance q / 1.5, src/SPECK_FLT.cpp:578)
int launch_outlier_apply(hipStream_t st, const OutlierBufs& b, const CoderState* cst, double* vals,
                         size_t valsStride);

}  // namespace sperrhip
#endif
