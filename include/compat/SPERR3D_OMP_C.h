// SPERR3D_OMP_C.h -- the reference's header name for sperr::SPERR3D_OMP_C (/root/reference/include/SPERR3D_OMP_C.h:14-35),
// served by the header-only mirrors over the C ABI of libsperr_hip.so (= libSPERR.so).
#ifndef SPERR_HIP_COMPAT_SPERR3D_OMP_C_H
#define SPERR_HIP_COMPAT_SPERR3D_OMP_C_H
#include "sperr_helper.h"
#endif
