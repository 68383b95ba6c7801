// SPECK_FLT.h -- the reference's header name for the chunk pipeline classes (/root/reference/include/SPECK_FLT.h:17-65),
// served by the header-only mirrors over the C ABI of libsperr_hip.so (= libSPERR.so).
#ifndef SPERR_HIP_COMPAT_SPECK_FLT_H
#define SPERR_HIP_COMPAT_SPECK_FLT_H
#include "sperr_helper.h"
#endif
