// SPERR3D_OMP_D.h -- the reference's header name for sperr::SPERR3D_OMP_D (/root/reference/include/SPERR3D_OMP_D.h:15-32),
// served by the header-only mirrors over the C ABI of libsperr_hip.so (= libSPERR.so).
#ifndef SPERR_HIP_COMPAT_SPERR3D_OMP_D_H
#define SPERR_HIP_COMPAT_SPERR3D_OMP_D_H
#include "sperr_helper.h"
#endif
