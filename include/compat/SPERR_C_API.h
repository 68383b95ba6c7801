/*
 * SPERR_C_API.h -- the reference's C API header name, served by libsperr_hip.so (installed also as
 * libSPERR.so).  A caller written against /root/reference/include/SPERR_C_API.h compiles unchanged
 * with -I<this directory> and links with -lSPERR: same six functions, same argument meaning,
 * ownership (*dst must be NULL on entry, the caller free()s it) and return codes
 * (/root/reference/include/SPERR_C_API.h:21-31,53-156, src/SPERR_C_API.cpp).  In C++ the
 * declarations sit inside namespace C_API with C linkage, as in the reference (:16-19,158-161),
 * so `C_API::sperr_comp_3d(...)` names the same unmangled symbol.
 *
 * The chunk pipeline runs on the MI355X; see ../sperr_hip.h for what `nthreads` means here and for
 * the device-resident entry points that have no counterpart in the reference.
 */
#ifndef SPERR_C_API_H
#define SPERR_C_API_H

#ifndef USE_VANILLA_CONFIG
#include "SperrConfig.h"
#endif

#include <stddef.h> /* size_t */
#include <stdint.h>

#ifdef __cplusplus
namespace C_API {
extern "C" {
#endif

/* modes: 1 = fixed bits per value, 2 = fixed PSNR (dB), 3 = fixed point-wise error.
 * returns: 0 ok, 1 *dst not NULL, 2 unsupported parameter, -1 anything else */

/* one 2D slice; out_inc_header != 0 prepends the 10-byte slice header */
int sperr_comp_2d(const void* src, int is_float, size_t dimx, size_t dimy, int mode, double quality,
                  int out_inc_header, void** dst, size_t* dst_len);

/* `src` is the stream WITHOUT the 10-byte header */
int sperr_decomp_2d(const void* src, size_t src_len, int output_float, size_t dimx, size_t dimy,
                    void** dst);

/* dims and precision of a 2D stream with header or a 3D container (dimz == 1 for a slice) */
void sperr_parse_header(const void* src, size_t* dimx, size_t* dimy, size_t* dimz, int* is_float);

/* a volume cut into chunks (chunk dims are a preference, remainders merge into their neighbour) */
int sperr_comp_3d(const void* src, int is_float, size_t dimx, size_t dimy, size_t dimz,
                  size_t chunk_x, size_t chunk_y, size_t chunk_z, int mode, double quality,
                  size_t nthreads, void** dst, size_t* dst_len);

int sperr_decomp_3d(const void* src, size_t src_len, int output_float, size_t nthreads,
                    size_t* dimx, size_t* dimy, size_t* dimz, void** dst);

/* keep `pct` percent of every chunk of a 3D container (progressive access) */
int sperr_trunc_3d(const void* src, size_t src_len, unsigned pct, void** dst, size_t* dst_len);

#ifdef __cplusplus
} /* extern "C" */
} /* namespace C_API */
#endif

#endif
