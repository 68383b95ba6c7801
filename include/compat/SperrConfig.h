/* SperrConfig.h -- what the reference's build pastes together from SperrConfig.h.in
 * (/root/reference/SperrConfig.h.in, CMakeLists.txt:5): the version of the stream format this
 * library reads and writes (major version byte 0, src/SPERR3D_OMP_C.cpp:188), and who built it. */
#ifndef SPERR_CONFIG
#define SPERR_CONFIG

#define SPERR_VERSION_MAJOR 0
#define SPERR_VERSION_MINOR 8
#define SPERR_VERSION_PATCH 5

#ifdef __GNUC__
#define SPERR_CONFIG_UNUSED __attribute__((unused))
#else
#define SPERR_CONFIG_UNUSED
#endif
static const char* SPERR_GIT_SHA1 SPERR_CONFIG_UNUSED = "sperr_hip";
static const char* SPERR_GIT_BRANCH SPERR_CONFIG_UNUSED = "gfx950";

/* the chunk loop is a device farm, not an OpenMP team; set_num_threads() / nthreads are accepted */
#define USE_OMP

#endif
