// SPECK3D_FLT.h -- the reference's header name for sperr::SPECK3D_FLT (/root/reference/include/SPECK3D_FLT.h),
// served by the header-only mirrors over the C ABI of libsperr_hip.so (= libSPERR.so).
#ifndef SPERR_HIP_COMPAT_SPECK3D_FLT_H
#define SPERR_HIP_COMPAT_SPECK3D_FLT_H
#include "sperr_helper.h"
#endif
