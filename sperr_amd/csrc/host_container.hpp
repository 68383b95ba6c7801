// host_container.hpp -- the byte-level host code of the container layer: chunk grid, container
// header parsing, progressive truncation.  No GPU, no HIP header: engine.hip includes it for the
// product, tests/cpp/host_fuzz.cpp compiles it with -fsanitize=address,undefined and feeds it
// damaged containers (this code parses untrusted bytes).
//
// Restated from the reference (file:line under /root/reference):
//   src/sperr_helper.cpp:542-592            chunk_volume (x fastest, short remainders merged)
//   src/SPERR3D_Stream_Tools.cpp:46-105     container header parsing (+ SPERR3D_OMP_D.cpp:23-49)
//   src/SPERR3D_Stream_Tools.cpp:134-226    progressive truncation (src/SPERR_C_API.cpp:260-280)
#ifndef SPERR_AMD_HOST_CONTAINER_HPP
#define SPERR_AMD_HOST_CONTAINER_HPP

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace sperrhip {
namespace hostc {

using Dims = std::array<size_t, 3>;

// src/sperr_helper.cpp:542-592
inline void chunk_segments(const Dims& vol, const Dims& chunk, size_t nseg[3])
{
  for (int a = 0; a < 3; a++) {
    nseg[a] = vol[a] / chunk[a];
    if (vol[a] % chunk[a] > chunk[a] / 2)
      nseg[a]++;
    if (nseg[a] == 0)
      nseg[a] = 1;
  }
}

// how many chunks chunk_volume would list, SIZE_MAX if that does not fit (dimensions read from a
// container header are checked against the container's length with this before anything is sized)
inline size_t chunk_count(const Dims& vol, const Dims& chunk)
{
  size_t nseg[3];
  chunk_segments(vol, chunk, nseg);
  unsigned __int128 n = (unsigned __int128)nseg[0] * nseg[1];
  if (n > SIZE_MAX)
    return SIZE_MAX;
  n *= nseg[2];
  return n > SIZE_MAX ? SIZE_MAX : (size_t)n;
}

inline std::vector<std::array<size_t, 6>> chunk_volume(const Dims& vol, const Dims& chunk)
{
  size_t nseg[3];
  chunk_segments(vol, chunk, nseg);
  std::vector<std::array<size_t, 6>> out;
  out.reserve(nseg[0] * nseg[1] * nseg[2]);
  for (size_t z = 0; z < nseg[2]; z++)
    for (size_t y = 0; y < nseg[1]; y++)
      for (size_t x = 0; x < nseg[0]; x++) {
        const size_t idx[3] = {x, y, z};
        std::array<size_t, 6> c;
        for (int a = 0; a < 3; a++) {
          const size_t beg = idx[a] * chunk[a];
          const size_t end = (idx[a] + 1 == nseg[a]) ? vol[a] : beg + chunk[a];
          c[2 * a] = beg;
          c[2 * a + 1] = end - beg;
        }
        out.push_back(c);
      }
  return out;
}

struct ContainerInfo {
  Dims vol, chunk;
  size_t nvals = 0;   // vol[0] * vol[1] * vol[2], checked not to wrap
  bool is_float = false, multi = false;
  std::vector<uint64_t> off, len;
};

// SPERR3D_Stream_Tools.cpp:46-105 + the checks of SPERR3D_OMP_D.cpp:23-49
inline int parse_container_host(const uint8_t* h, size_t hlen, size_t total_len, ContainerInfo& ci,
                         size_t* need)
{
  if (hlen < 14) {
    *need = 20;
    return 1;
  }
  if (h[0] != 0 || !(h[1] & 0x40))
    return -1;  // version mismatch / not 3D
  ci.is_float = (h[1] & 0x20) != 0;
  ci.multi = (h[1] & 0x10) != 0;
  uint32_t v3[3];
  memcpy(v3, h + 2, 12);
  ci.vol = {v3[0], v3[1], v3[2]};
  ci.chunk = ci.vol;
  size_t pos = 14;
  if (ci.multi) {
    if (hlen < 20) {
      *need = 20;
      return 1;
    }
    uint16_t c3[3];
    memcpy(c3, h + 14, 6);
    ci.chunk = {c3[0], c3[1], c3[2]};
    pos = 20;
  }
  for (int a = 0; a < 3; a++)
    if (ci.vol[a] == 0 || ci.chunk[a] == 0)
      return -1;
  {   // three 32-bit dims can wrap a size_t; everything downstream sizes buffers from this product
    const unsigned __int128 nv = (unsigned __int128)ci.vol[0] * ci.vol[1] * ci.vol[2];
    if (nv > (unsigned __int128)(SIZE_MAX / 8))
      return -1;
    ci.nvals = (size_t)nv;
  }
  // every chunk has a 4-byte length in the header: a damaged header must not make us list more
  // chunks than the container could hold
  const size_t nchunks = chunk_count(ci.vol, ci.chunk);
  if (nchunks > (total_len - pos) / 4)
    return -1;
  const size_t hdr = pos + 4 * nchunks;
  if (hlen < hdr) {
    *need = hdr;
    return 1;
  }
  ci.off.resize(nchunks);
  ci.len.resize(nchunks);
  uint64_t off = hdr;
  for (size_t i = 0; i < nchunks; i++) {
    uint32_t l;
    memcpy(&l, h + pos + 4 * i, 4);
    ci.off[i] = off;
    ci.len[i] = l;
    off += l;
  }
  if (off != total_len)
    return -1;  // RTNType::WrongLength
  return 0;
}

// include/SPERR_C_API.h:138-156, src/SPERR_C_API.cpp:260-280,
// src/SPERR3D_Stream_Tools.cpp:134-226: host-side byte surgery, no GPU involved.  Every chunk
// stream keeps `pct` percent of its bytes (at least 64, at most what it has), the container is
// flagged as a portion and the chunk lengths are rewritten; the decoder zero-pads what is missing.
// Returns 0 ok (*dst malloc'd), 1 *dst not NULL, -1 other.
inline int truncate_container(const uint8_t* src, size_t src_len, unsigned pct, void** dst, size_t* dst_len)
{
  if (*dst != nullptr)
    return 1;
  const uint8_t* h = static_cast<const uint8_t*>(src);
  if (src_len < 20)
    return -1;
  const bool multi = (h[1] & 0x10) != 0;
  uint32_t v3[3];
  memcpy(v3, h + 2, 12);
  Dims vol{v3[0], v3[1], v3[2]}, cd = vol;
  size_t pos = 14;
  if (multi) {
    uint16_t c3[3];
    memcpy(c3, h + 14, 6);
    cd = {c3[0], c3[1], c3[2]};
    pos = 20;
  }
  for (int a = 0; a < 3; a++)
    if (vol[a] == 0 || cd[a] == 0)
      return -1;
  const size_t nchunks = chunk_count(vol, cd);
  if (src_len < pos || nchunks > (src_len - pos) / 4)
    return -1;
  const size_t hlen = pos + 4 * nchunks;
  constexpr size_t kMinChunkBytes = 64;   // include/SPERR3D_Stream_Tools.h:54
  const bool whole = pct == 0 || pct >= 100;
  std::vector<size_t> off(nchunks), len(nchunks);
  size_t at = hlen, total = hlen, far = 0;
  for (size_t i = 0; i < nchunks; i++) {
    uint32_t l;
    memcpy(&l, h + pos + 4 * i, 4);
    off[i] = at;
    at += l;
    len[i] = l;
    if (!whole && l > kMinChunkBytes)
      len[i] = std::max(kMinChunkBytes, (size_t)((double)pct / 100.0 * (double)l));
    total += len[i];
    far = std::max(far, off[i] + len[i]);
  }
  if (src_len < far)
    return -1;
  uint8_t* out = static_cast<uint8_t*>(malloc(total));
  if (!out)
    return -1;
  memcpy(out, h, pos);
  if (!whole) {
    out[0] = 0;       // SPERR_VERSION_MAJOR (CMakeLists.txt:5)
    out[1] |= 0x80;   // portion flag: bool 0 of the packed byte (src/sperr_helper.cpp:262-273)
  }
  size_t w = hlen;
  for (size_t i = 0; i < nchunks; i++) {
    const uint32_t l = (uint32_t)len[i];
    memcpy(out + pos + 4 * i, &l, 4);
    memcpy(out + w, h + off[i], len[i]);
    w += len[i];
  }
  *dst = out;
  *dst_len = total;
  return 0;
}

}  // namespace hostc
}  // namespace sperrhip

#endif
