// SPERR3D_Stream_Tools.h -- the reference's header name for sperr::SPERR3D_Header / SPERR3D_Stream_Tools (/root/reference/include/SPERR3D_Stream_Tools.h:11-66),
// served by the header-only mirrors over the C ABI of libsperr_hip.so (= libSPERR.so).
#ifndef SPERR_HIP_COMPAT_SPERR3D_STREAM_TOOLS_H
#define SPERR_HIP_COMPAT_SPERR3D_STREAM_TOOLS_H
#include "sperr_helper.h"
#endif
