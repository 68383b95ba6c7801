// xform.h -- launchers of the floating-point stage kernels (xform.hip)
#ifndef SPERR_AMD_XFORM_H
#define SPERR_AMD_XFORM_H

#include <string.h>

#include "common.h"

namespace sperrhip {

struct LiftConsts {
  double alpha, beta, gamma, delta, eps, inv_eps;
};
LiftConsts lift_consts();

template <typename T>
int launch_condition(hipStream_t stream, const T* vol, VolDesc vd, const ChunkGeom* geom,
                     uint32_t nchunks, const uint32_t cdims[3], uint32_t nstrides,
                     double* strideMean, size_t strideMeanStride, double* vals,
                     size_t valsStride, CoderState* st, bool gather, bool want_range = false,
                     bool org_x_aligned = false);   // every chunk's x origin is a multiple of 16 bytes

template <typename T>
int launch_scatter(hipStream_t stream, T* vol, VolDesc vd, const ChunkGeom* geom,
                   uint32_t nchunks, const uint32_t cdims[3], const double* vals,
                   size_t valsStride, const CoderState* st);

// What a lifting pass can do on the way for the samples of its region that no LATER pass (in
// forward order) touches, i.e. those outside `inner`, the region of the next pass:
//   forward (mode 1)  they have their final value: the largest magnitude goes to
//                     CoderState::maxabs (src/SPECK_FLT.cpp:282-301), no pass of its own;
//   inverse (mode 2)  they come straight from the decoder: the pass dequantises them from the
//                     integer coefficients while it loads (src/SPECK_FLT.cpp:373-399, with the
//                     decoder's masks completing the ones never refined), so no fp64 copy of the
//                     whole chunk is written and read again.  Chunks with 64-bit coefficients keep
//                     theirs in the fp64 buffer (converted in place beforehand) and are read as ever.
struct DecState;
struct LiftFuse {
  int mode = 0;
  uint32_t inner[3] = {0, 0, 0};
  const uint32_t* coef = nullptr;
  size_t coefStride = 0;
  const uint64_t* sign = nullptr;
  size_t signStride = 0;
  const uint64_t* sigNew = nullptr;
  const uint64_t* sigOld = nullptr;
  size_t maskStride = 0;
  const DecState* dst = nullptr;
  // 1: k_ref_assemble has written the coefficients complete (never-refined ones included) and, chunk by chunk where
  // coef_scheme(dst[c]) allows it, with the sign in bit 31 (speck_dec.h): the sign and mask words are not read at all then.
  // q * double(magnitude), sign flipped = q * double(magnitude) * (+-1.0) bit for bit (a zero is positive either way)
  int coefSigned = 0;
  // k_lift_xyz_inv: do not add the chunk's mean to what it writes (the encoder's point-wise error stage compares in the
  // conditioned domain, src/SPECK_FLT.cpp:461-486)
  int noMean = 0;
  // inverse, compact chunk buffer (round 3): `vals` holds only the box the coarser levels work in,
  // rows of bufx samples and bufy rows per slice (0: the chunk's own dims); the coefficient and mask
  // arrays keep the chunk's dims
  uint32_t bufx = 0, bufy = 0;
};

// io: 0 in place; 1 / 2: the pass also reads (forward) or writes (inverse) the float / double
// volume through the chunk map -- only valid for a pass whose region is the whole chunk
int launch_lift(hipStream_t stream, bool forward, double* vals, size_t valsStride,
                uint32_t nchunks, const uint32_t cdims[3], int axis, const uint32_t region[3],
                CoderState* st, int io = 0, void* volume = nullptr, VolDesc vd = VolDesc{},
                const ChunkGeom* geom = nullptr, const LiftFuse* fuse = nullptr);

// The x and y passes of the finest level fused with the volume access (forward: volume -> vals,
// inverse: vals -> volume); only for chunks whose first two passes are the full-size x and y ones.
bool lift_xy_applicable(const uint32_t cdims[3]);
int launch_lift_xy(hipStream_t stream, bool forward, double* vals, size_t valsStride,
                   uint32_t nchunks, const uint32_t cdims[3], const CoderState* st, int io,
                   void* volume, VolDesc vd, const ChunkGeom* geom);

// The x, y AND z pass of the finest level in one kernel (k_lift_xyz_fwd / _inv): the z direction is
// a sliding window of per-position lifting pipelines in registers.  Forward: volume -> vals, with
// the largest magnitude collected when fuse->mode == 1; inverse: (coefficients dequantised on the
// way when fuse->mode == 2, never-refined ones already completed by k_dec_finish) -> volume.  Only
// for chunks whose first three passes are the full-size x, y and z ones and whose rows fit.
bool lift_xyz_applicable(const uint32_t cdims[3]);
int launch_lift_xyz(hipStream_t stream, bool forward, double* vals, size_t valsStride, uint32_t nchunks,
                    const uint32_t cdims[3], CoderState* st, int io, void* volume, VolDesc vd,
                    const ChunkGeom* geom, const LiftFuse* fuse);

// have_max: CoderState::maxabs is already there (the lifting passes collected it)
int launch_maxabs_q(hipStream_t stream, const double* vals, size_t valsStride, uint32_t nchunks,
                    uint32_t n, CoderState* st, bool have_max = false);
int launch_make_q_wide(hipStream_t stream, uint32_t nchunks, CoderState* st);
int launch_mark_wide(hipStream_t stream, uint32_t nchunks, CoderState* st);

// PSNR mode: st[c].mse = quantisation error estimate of st[c].q for the chunks with mse_active set
// (partial: n / 4096 + 1 doubles per chunk)
int launch_mse(hipStream_t stream, const double* vals, size_t valsStride, uint32_t nchunks,
               uint32_t n, double* partial, size_t partialStride, CoderState* st);

// doubles as unsigned keys of the same order (the keys of a zeroed state are below every value)
__host__ __device__ inline unsigned long long order_key(double v)
{
  unsigned long long b;
  memcpy(&b, &v, 8);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double order_key_value(unsigned long long k)
{
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double v;
  memcpy(&v, &b, 8);
  return v;
}

int launch_quantize(hipStream_t stream, bool wide, const double* vals, size_t valsStride,
                    uint32_t nchunks, uint32_t n, void* coef, size_t coefStride, uint64_t* sign,
                    size_t signStride, int8_t* msb, size_t msbStride, const CoderState* st);

// sigNew / sigOld / dst: the decoder's significance masks and state, to complete the coefficients
// that were never refined (then launch_speck_decode is told to skip its own finishing pass)
int launch_inv_quantize(hipStream_t stream, bool wide, const void* coef, size_t coefStride,
                        const uint64_t* sign, size_t signStride, uint32_t nchunks, uint32_t n,
                        double* vals, size_t valsStride, const CoderState* st,
                        const uint64_t* sigNew = nullptr, const uint64_t* sigOld = nullptr,
                        size_t maskStride = 0, const DecState* dst = nullptr);

}  // namespace sperrhip

#endif
