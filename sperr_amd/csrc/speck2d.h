// speck2d.h -- the 2D integer coder behind sperr_comp_2d / sperr_decomp_2d (speck2d.hip).
// Reference: src/SPECK2D_INT.cpp, src/SPECK2D_INT_ENC.cpp, src/SPECK2D_INT_DEC.cpp, with the
// bit-plane loop of src/SPECK_INT.cpp:110-228,310-469.
#ifndef SPERR_AMD_SPECK2D_H
#define SPERR_AMD_SPECK2D_H

#include "common.h"
#include "speck_dec.h"

namespace sperrhip {

constexpr int kS2MaxLevels = 40;   // num_of_partitions(max(dx, dy)) + 1 lists

// one slice (a batch of one)
struct Speck2dBufs {
  uint32_t dx, dy, N, nw;        // dims < 65536 each; nw = mask words
  uint32_t nxforms;              // transform levels = level of the root set (SPECK2D_INT.cpp:200-207)
  uint32_t nlists;
  uint32_t wbMin;                // encoder: sets of at least this many (and at most 64) samples are expanded by the wave at once
  uint32_t levelOff[kS2MaxLevels + 1];   // list storage: first entry of every level
  uint64_t* runs;                // sx | sy << 16 | lx << 32 | ly << 48
  int8_t* sval;                  // encoder: msb of the largest coefficient inside
  uint64_t* lip;                 // LIP / LSP bitmasks, raster order
  uint64_t* lsp;
  uint32_t* fresh;               // values found significant in the current plane (LSP_new)
  void* coef;                    // uint32_t or uint64_t magnitudes (encoder: input; decoder: output)
  uint64_t* sign;                // encoder: input; decoder: initialised to all ones by the caller
  const int8_t* msb;             // encoder: msb of every coefficient (k_quantize)
  int32_t* prep;                 // encoder: [0] largest msb, [1 + 3 * lev + k] largest msb of subband k of level lev
  uint64_t* stream;              // zeroed (encoder) / loaded, zero padded (decoder)
  size_t streamWords;
  CoderState* cst;               // [1]
  DecState* dst;                 // decoder: avail / total_bits / nbp of the chunk (k_dec_header)
};

// sizes the list storage (levelOff, returns the number of entries)
size_t speck2d_list_entries(Speck2dBufs& b);

// encoder: needs b.msb; sets cst->nbp / total_bits / stream_len / need_retry like k_enc_finalize
int launch_speck2d_encode(hipStream_t st, const Speck2dBufs& b, uint64_t raw_budget, bool rate_mode,
                          bool wide_pass);
// decoder: after k_dec_header / k_dec_load_words (launch_speck_decode with no planes)
int launch_speck2d_decode(hipStream_t st, const Speck2dBufs& b, bool wide_pass);

}  // namespace sperrhip
#endif
