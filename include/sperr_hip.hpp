// sperr_hip.hpp -- header-only C++ mirrors of the reference's driver classes on top of the C ABI
// of libsperr_hip.so (include/sperr_hip.h).  Same names, methods, argument meaning and return
// codes as the reference, so code written against
//
//     /root/reference/include/SPERR3D_OMP_C.h:14-35   (sperr::SPERR3D_OMP_C)
//     /root/reference/include/SPERR3D_OMP_D.h:15-32   (sperr::SPERR3D_OMP_D)
//     /root/reference/include/SPECK_FLT.h:17-65       (sperr::SPECK3D_FLT, the per-chunk pipeline;
//                                                      sperr::SPECK2D_FLT, one slice)
//     /root/reference/include/SPERR3D_Stream_Tools.h:11-66 (SPERR3D_Header, SPERR3D_Stream_Tools)
//     /root/reference/include/sperr_helper.h:35,54-64 (dims_type, vec8_type, vecd_type, RTNType)
//
// compiles unchanged when it includes this header instead and links -lsperr_hip.  The chunk
// pipeline itself runs on the GPU; set_num_threads() is accepted and ignored.  The fixed-rate
// (set_bitrate), fixed-PSNR (set_psnr) and fixed point-wise error (set_tolerance) modes all run
// on the GPU path.
#ifndef SPERR_HIP_HPP
#define SPERR_HIP_HPP

#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sperr_hip.h"

namespace sperr {

using dims_type = std::array<size_t, 3>;
using vec8_type = std::vector<uint8_t>;
using vecd_type = std::vector<double>;
using vecf_type = std::vector<float>;

enum class RTNType {  // include/sperr_helper.h:54-64
  Good = 0,
  WrongLength,
  IOError,
  BitBudgetMet,
  VersionMismatch,
  SliceVolumeMismatch,
  CompModeUnknown,
  FE_Invalid,
  Error
};

enum class CompMode : unsigned char { PSNR, PWE, Rate, Unknown };

namespace detail {
// Width in bytes of the integer coefficients of a chunk stream (SPECK_FLT::integer_len,
// /root/reference/src/SPECK_FLT.cpp:193-213).  The reference picks it from the largest quantised
// coefficient when it encodes (:324-337: <= 255, <= 65535, <= 2^32 - 1) and from the number of
// bit planes in the SPECK header when it decodes (:64-72: <= 8, <= 16, <= 32) -- one and the same
// rule, so the stream's header byte answers for both.  A stream without coefficients (constant
// field) leaves the default, uint64 (include/SPECK_FLT.h:68).
inline size_t integer_len_of(const vec8_type& stream)
{
  if (stream.size() < 18 || (stream[0] & 0x01))
    return sizeof(uint64_t);
  const unsigned nbp = stream[17];
  return nbp <= 8 ? 1 : nbp <= 16 ? 2 : nbp <= 32 ? 4 : 8;
}
}  // namespace detail

// ---- src/SPERR3D_OMP_C.cpp:12-161 --------------------------------------------------------------
class SPERR3D_OMP_C {
 public:
  void set_num_threads(size_t) {}
  void set_dims_and_chunks(dims_type vol_dims, dims_type chunk_dims)
  {
    m_dims = vol_dims;
    for (size_t i = 0; i < 3; i++) {  // clamp to [1, vol] (SPERR3D_OMP_C.cpp:23-30)
      size_t c = chunk_dims[i] < 1 ? 1 : chunk_dims[i];
      m_chunk_dims[i] = c > vol_dims[i] ? vol_dims[i] : c;
    }
  }
  void set_psnr(double v) { m_mode = CompMode::PSNR; m_quality = v; }
  void set_tolerance(double v) { m_mode = CompMode::PWE; m_quality = v; }
  void set_bitrate(double v) { m_mode = CompMode::Rate; m_quality = v; }

  template <typename T>
  auto compress(const T* buf, size_t buf_len) -> RTNType
  {
    static_assert(std::is_floating_point<T>::value, "!! Only floating point values are supported !!");
    if (m_mode == CompMode::Unknown)
      return RTNType::CompModeUnknown;
    if (buf_len != m_dims[0] * m_dims[1] * m_dims[2])
      return RTNType::WrongLength;
    const int mode = m_mode == CompMode::Rate ? 1 : m_mode == CompMode::PSNR ? 2 : 3;
    void* dst = nullptr;
    size_t len = 0;
    const int rtn = sperr_comp_3d(buf, std::is_same<T, float>::value ? 1 : 0, m_dims[0], m_dims[1],
                                  m_dims[2], m_chunk_dims[0], m_chunk_dims[1], m_chunk_dims[2], mode,
                                  m_quality, 0, &dst, &len);
    if (rtn != 0)
      return RTNType::Error;
    m_stream.assign(static_cast<uint8_t*>(dst), static_cast<uint8_t*>(dst) + len);
    std::free(dst);
    return RTNType::Good;
  }
  auto get_encoded_bitstream() const -> vec8_type { return m_stream; }

 private:
  CompMode m_mode = CompMode::Unknown;
  double m_quality = 0.0;
  dims_type m_dims = {0, 0, 0}, m_chunk_dims = {0, 0, 0};
  vec8_type m_stream;
};

// ---- src/SPERR3D_OMP_D.cpp:13-165 --------------------------------------------------------------
class SPERR3D_OMP_D {
 public:
  void set_num_threads(size_t) {}
  auto use_bitstream(const void* p, size_t total_len) -> RTNType
  {
    const auto* u8 = static_cast<const uint8_t*>(p);
    if (total_len < 18)
      return RTNType::WrongLength;
    if (u8[0] != 0)
      return RTNType::VersionMismatch;
    if (!(u8[1] & 0x40))
      return RTNType::SliceVolumeMismatch;
    size_t dx, dy, dz;
    int is_float;
    sperr_parse_header(p, &dx, &dy, &dz, &is_float);
    m_dims = {dx, dy, dz};
    m_chunk_dims = m_dims;
    if (u8[1] & 0x10) {
      uint16_t c3[3];
      std::memcpy(c3, u8 + 14, 6);
      m_chunk_dims = {c3[0], c3[1], c3[2]};
    }
    m_ptr = u8;
    m_len = total_len;
    return RTNType::Good;
  }
  // the pointer MUST be the one given to use_bitstream (SPERR3D_OMP_D.cpp:53-56)
  auto decompress(const void* bitstream, bool multi_res = false) -> RTNType
  {
    if (bitstream == nullptr || m_ptr == nullptr || bitstream != m_ptr)
      return RTNType::Error;
    void* dst = nullptr;
    size_t dx, dy, dz;
    m_hierarchy.clear();
    if (multi_res) {   // SPERR3D_OMP_D.cpp:68-84,121-130: the volume at every coarsened resolution
      size_t nlev = 0, ldims[48];
      double* levels[16] = {};
      if (sperrhip_decomp_3d_multires(m_ptr, m_len, 0, &dx, &dy, &dz, &dst, &nlev, ldims, levels) != 0)
        return RTNType::Error;
      for (size_t h = 0; h < nlev; h++) {
        const size_t n = ldims[3 * h] * ldims[3 * h + 1] * ldims[3 * h + 2];
        m_hierarchy.emplace_back(levels[h], levels[h] + n);
        std::free(levels[h]);
      }
    }
    else if (sperr_decomp_3d(m_ptr, m_len, 0, 0, &dx, &dy, &dz, &dst) != 0)
      return RTNType::Error;
    const auto* d = static_cast<const double*>(dst);
    m_vol.assign(d, d + dx * dy * dz);
    std::free(dst);
    return RTNType::Good;
  }
  auto view_decoded_data() const -> const vecd_type& { return m_vol; }
  auto release_decoded_data() -> vecd_type&& { return std::move(m_vol); }
  auto view_hierarchy() const -> const std::vector<vecd_type>& { return m_hierarchy; }
  auto release_hierarchy() -> std::vector<vecd_type>&& { return std::move(m_hierarchy); }
  auto get_dims() const -> dims_type { return m_dims; }
  auto get_chunk_dims() const -> dims_type { return m_chunk_dims; }

 private:
  dims_type m_dims = {0, 0, 0}, m_chunk_dims = {0, 0, 0};
  const uint8_t* m_ptr = nullptr;
  size_t m_len = 0;
  vecd_type m_vol;
  std::vector<vecd_type> m_hierarchy;   // multi-resolution decoding, coarsest first
};

// ---- src/SPECK_FLT.cpp:11-606 + src/SPECK3D_FLT.cpp: one chunk ---------------------------------
// A chunk stream is the single-chunk container minus its 18-byte header
// (src/SPERR3D_OMP_C.cpp:163-234: 14 bytes + one 4-byte length).
class SPECK3D_FLT {
 public:
  template <typename T>
  void copy_data(const T* p, size_t len) { m_vals.assign(p, p + len); }
  void take_data(vecd_type&& buf) { m_vals = std::move(buf); }
  void set_dims(dims_type d) { m_dims = d; }
  void set_bitrate(double bpp) { m_mode = CompMode::Rate; m_quality = bpp; }
  void set_psnr(double v) { m_mode = CompMode::PSNR; m_quality = v; }
  void set_tolerance(double v) { m_mode = CompMode::PWE; m_quality = v; }
  auto integer_len() const -> size_t { return detail::integer_len_of(m_stream); }

  auto compress() -> RTNType
  {
    if (m_vals.empty() || m_vals.size() != m_dims[0] * m_dims[1] * m_dims[2])
      return RTNType::Error;
    if (m_mode == CompMode::Unknown)
      return RTNType::CompModeUnknown;
    const int mode = m_mode == CompMode::Rate ? 1 : m_mode == CompMode::PSNR ? 2 : 3;
    void* dst = nullptr;
    size_t len = 0;
    if (sperr_comp_3d(m_vals.data(), 0, m_dims[0], m_dims[1], m_dims[2], m_dims[0], m_dims[1],
                      m_dims[2], mode, m_quality, 0, &dst, &len) != 0)
      return RTNType::Error;
    const auto* u8 = static_cast<const uint8_t*>(dst);
    m_stream.assign(u8 + 18, u8 + len);
    std::free(dst);
    return RTNType::Good;
  }
  void append_encoded_bitstream(vec8_type& buf) const
  {
    buf.insert(buf.end(), m_stream.begin(), m_stream.end());
  }
  auto use_bitstream(const void* p, size_t len) -> RTNType
  {
    if (len < 17)
      return RTNType::WrongLength;
    const auto* u8 = static_cast<const uint8_t*>(p);
    m_stream.assign(u8, u8 + len);
    return RTNType::Good;
  }
  auto decompress(bool multi_res = false) -> RTNType
  {
    if (m_stream.empty())
      return RTNType::Error;
    // wrap the chunk stream into a single-chunk container (double output)
    vec8_type c(18 + m_stream.size());
    c[0] = 0;
    c[1] = 0x40;
    const uint32_t v3[3] = {(uint32_t)m_dims[0], (uint32_t)m_dims[1], (uint32_t)m_dims[2]};
    std::memcpy(c.data() + 2, v3, 12);
    const uint32_t l = (uint32_t)m_stream.size();
    std::memcpy(c.data() + 14, &l, 4);
    std::memcpy(c.data() + 18, m_stream.data(), m_stream.size());
    void* dst = nullptr;
    size_t dx, dy, dz;
    m_hierarchy.clear();
    if (multi_res) {   // a chunk that is not dyadic has no hierarchy (src/CDF97.cpp:150-168)
      size_t nlev = 0, ld[48];
      double* lv[16] = {};
      if (sperrhip_decomp_3d_multires(c.data(), c.size(), 0, &dx, &dy, &dz, &dst, &nlev, ld, lv) != 0)
        return RTNType::Error;
      for (size_t h = 0; h < nlev; h++) {
        m_hierarchy.emplace_back(lv[h], lv[h] + ld[3 * h] * ld[3 * h + 1] * ld[3 * h + 2]);
        std::free(lv[h]);
      }
    }
    else if (sperr_decomp_3d(c.data(), c.size(), 0, 0, &dx, &dy, &dz, &dst) != 0)
      return RTNType::Error;
    const auto* d = static_cast<const double*>(dst);
    m_vals.assign(d, d + dx * dy * dz);
    std::free(dst);
    return RTNType::Good;
  }
  auto view_hierarchy() const -> const std::vector<vecd_type>& { return m_hierarchy; }
  auto release_hierarchy() -> std::vector<vecd_type>&& { return std::move(m_hierarchy); }
  auto view_decoded_data() const -> const vecd_type& { return m_vals; }
  auto release_decoded_data() -> vecd_type&& { return std::move(m_vals); }

 private:
  CompMode m_mode = CompMode::Unknown;
  double m_quality = 0.0;
  dims_type m_dims = {0, 0, 0};
  vecd_type m_vals;
  vec8_type m_stream;
  std::vector<vecd_type> m_hierarchy;
};

// ---- src/SPECK_FLT.cpp + src/SPECK2D_FLT.cpp: one slice, dims = {x, y, 1} ------------------------
// The stream is the one sperr_comp_2d produces without its 10-byte header (src/SPERR_C_API.cpp:7-83).
class SPECK2D_FLT {
 public:
  template <typename T>
  void copy_data(const T* p, size_t len) { m_vals.assign(p, p + len); }
  void take_data(vecd_type&& buf) { m_vals = std::move(buf); }
  void set_dims(dims_type d) { m_dims = d; }
  void set_bitrate(double bpp) { m_mode = CompMode::Rate; m_quality = bpp; }
  void set_psnr(double v) { m_mode = CompMode::PSNR; m_quality = v; }
  void set_tolerance(double v) { m_mode = CompMode::PWE; m_quality = v; }
  auto integer_len() const -> size_t { return detail::integer_len_of(m_stream); }

  auto compress() -> RTNType
  {
    if (m_dims[2] != 1 || m_vals.empty() || m_vals.size() != m_dims[0] * m_dims[1])
      return RTNType::Error;
    if (m_mode == CompMode::Unknown)
      return RTNType::CompModeUnknown;
    const int mode = m_mode == CompMode::Rate ? 1 : m_mode == CompMode::PSNR ? 2 : 3;
    void* dst = nullptr;
    size_t len = 0;
    if (sperr_comp_2d(m_vals.data(), 0, m_dims[0], m_dims[1], mode, m_quality, 0, &dst, &len) != 0)
      return RTNType::Error;
    const auto* u8 = static_cast<const uint8_t*>(dst);
    m_stream.assign(u8, u8 + len);
    std::free(dst);
    return RTNType::Good;
  }
  void append_encoded_bitstream(vec8_type& buf) const
  {
    buf.insert(buf.end(), m_stream.begin(), m_stream.end());
  }
  auto use_bitstream(const void* p, size_t len) -> RTNType
  {
    if (len < 17)
      return RTNType::WrongLength;
    const auto* u8 = static_cast<const uint8_t*>(p);
    m_stream.assign(u8, u8 + len);
    return RTNType::Good;
  }
  auto decompress(bool multi_res = false) -> RTNType
  {
    if (m_stream.empty() || m_dims[2] != 1)
      return RTNType::Error;
    void* dst = nullptr;
    m_hierarchy.clear();
    if (multi_res) {
      size_t nlev = 0, ld[32];
      double* lv[16] = {};
      if (sperrhip_decomp_2d_multires(m_stream.data(), m_stream.size(), 0, m_dims[0], m_dims[1], &dst, &nlev,
                                      ld, lv) != 0)
        return RTNType::Error;
      for (size_t h = 0; h < nlev; h++) {
        m_hierarchy.emplace_back(lv[h], lv[h] + ld[2 * h] * ld[2 * h + 1]);
        std::free(lv[h]);
      }
    }
    else if (sperr_decomp_2d(m_stream.data(), m_stream.size(), 0, m_dims[0], m_dims[1], &dst) != 0)
      return RTNType::Error;
    const auto* d = static_cast<const double*>(dst);
    m_vals.assign(d, d + m_dims[0] * m_dims[1]);
    std::free(dst);
    return RTNType::Good;
  }
  auto view_decoded_data() const -> const vecd_type& { return m_vals; }
  auto release_decoded_data() -> vecd_type&& { return std::move(m_vals); }
  auto view_hierarchy() const -> const std::vector<vecd_type>& { return m_hierarchy; }
  auto release_hierarchy() -> std::vector<vecd_type>&& { return std::move(m_hierarchy); }

 private:
  CompMode m_mode = CompMode::Unknown;
  double m_quality = 0.0;
  dims_type m_dims = {0, 0, 0};
  vecd_type m_vals;
  vec8_type m_stream;
  std::vector<vecd_type> m_hierarchy;
};

// ---- include/SPERR3D_Stream_Tools.h:11-66, src/SPERR3D_Stream_Tools.cpp ---------------------------
struct SPERR3D_Header {
  uint8_t major_version = 0;
  bool is_portion = false;
  bool is_3D = false;
  bool is_float = false;
  bool multi_chunk = false;
  dims_type vol_dims = {0, 0, 0};
  dims_type chunk_dims = {0, 0, 0};
  size_t header_len = 0;
  size_t stream_len = 0;
  std::vector<size_t> chunk_offsets;   // (offset, length) of every chunk
};

class SPERR3D_Stream_Tools {
 public:
  auto get_header_len(std::array<uint8_t, 20> magic) const -> size_t
  {
    const bool multi = (magic[1] & 0x10) != 0;
    dims_type v, c;
    m_dims_of(magic.data(), multi, v, c);
    return (multi ? 20 : 14) + 4 * m_num_chunks(v, c);
  }
  auto get_stream_header(const void* p) const -> SPERR3D_Header
  {
    SPERR3D_Header h;
    const auto* u8 = static_cast<const uint8_t*>(p);
    h.major_version = u8[0];
    h.is_portion = (u8[1] & 0x80) != 0;   // pack_8_booleans: bool i at bit 7 - i
    h.is_3D = (u8[1] & 0x40) != 0;
    h.is_float = (u8[1] & 0x20) != 0;
    h.multi_chunk = (u8[1] & 0x10) != 0;
    m_dims_of(u8, h.multi_chunk, h.vol_dims, h.chunk_dims);
    const size_t n = m_num_chunks(h.vol_dims, h.chunk_dims), at = h.multi_chunk ? 20 : 14;
    h.header_len = at + 4 * n;
    h.chunk_offsets.resize(2 * n);
    size_t off = h.header_len;
    for (size_t i = 0; i < n; i++) {
      uint32_t l;
      std::memcpy(&l, u8 + at + 4 * i, 4);
      h.chunk_offsets[2 * i] = off;
      h.chunk_offsets[2 * i + 1] = l;
      off += l;
    }
    h.stream_len = off;
    return h;
  }
  // keep `pct` percent of every chunk (at least 64 bytes of each); an empty vector on failure
  auto progressive_truncate(const void* stream, size_t stream_len, unsigned pct) const -> vec8_type
  {
    vec8_type out;
    void* dst = nullptr;
    size_t len = 0;
    if (sperr_trunc_3d(stream, stream_len, pct, &dst, &len) == 0) {
      const auto* u8 = static_cast<const uint8_t*>(dst);
      out.assign(u8, u8 + len);
      std::free(dst);
    }
    return out;
  }
  auto progressive_read(const std::string& filename, unsigned pct) const -> vec8_type
  {
    vec8_type whole;
    if (std::FILE* f = std::fopen(filename.c_str(), "rb")) {
      uint8_t buf[1 << 16];
      for (size_t got; (got = std::fread(buf, 1, sizeof(buf), f)) > 0;)
        whole.insert(whole.end(), buf, buf + got);
      std::fclose(f);
    }
    return whole.size() < 18 ? vec8_type() : progressive_truncate(whole.data(), whole.size(), pct);
  }

 private:
  static void m_dims_of(const uint8_t* u8, bool multi, dims_type& v, dims_type& c)
  {
    uint32_t v3[3];
    std::memcpy(v3, u8 + 2, 12);
    v = {v3[0], v3[1], v3[2]};
    c = v;
    if (multi) {
      uint16_t c3[3];
      std::memcpy(c3, u8 + 14, 6);
      c = {c3[0], c3[1], c3[2]};
    }
  }
  // src/sperr_helper.cpp:542-592: a remainder longer than half a chunk becomes a chunk of its own
  static auto m_num_chunks(const dims_type& v, const dims_type& c) -> size_t
  {
    size_t n = 1;
    for (int a = 0; a < 3; a++) {
      size_t seg = c[a] ? v[a] / c[a] : 0;
      if (c[a] && v[a] % c[a] > c[a] / 2)
        seg++;
      n *= seg ? seg : 1;
    }
    return n;
  }
};

}  // namespace sperr

#endif
