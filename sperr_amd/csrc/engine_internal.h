// engine_internal.h -- host-side helpers of engine.hip that the chunk farm (farm.hip) shares.
#ifndef SPERR_AMD_ENGINE_INTERNAL_H
#define SPERR_AMD_ENGINE_INTERNAL_H

#include <array>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace sperrhip {

using Dims3 = std::array<size_t, 3>;

// src/sperr_helper.cpp:542-592: {x0, lenx, y0, leny, z0, lenz} per chunk, x fastest
std::vector<std::array<size_t, 6>> host_chunk_volume(const Dims3& vol, const Dims3& chunk);

struct HostContainer {
  Dims3 vol, chunk;
  size_t nvals = 0;
  bool is_float = false, multi = false, portion = false;
  std::vector<uint64_t> off, len;   // byte offset / length of every chunk stream
};
// the whole container is in host memory; 0 ok (SPERR3D_Stream_Tools.cpp:46-105 + the checks of
// SPERR3D_OMP_D.cpp:23-49)
int host_parse_container(const uint8_t* p, size_t len, HostContainer& out);

// upper bound of one chunk stream (conditioner + SPECK headers + payload [+ outlier stream])
size_t host_chunk_stream_bound(size_t nvals, int mode, double quality);

// farm.hip: the buffers of every idle farm worker go back (sperrhip_release)
void farm_release_idle();
// farm.hip: bytes the farm's workers hold right now -- pinned staging memory, or device buffers
unsigned long long farm_footprint(bool pinned);

// Set by a thread whose device calls run beside those of other threads on the same device (the farm's
// workers): the engine then does not cut a small batch into sub-batches of its own.
extern thread_local bool t_shared_device;

}  // namespace sperrhip

#endif
