// ref_probe.cpp -- thin extern "C" shims over the REAL reference classes, for stage-level checks.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile against the headers where they lie in
// /root/reference/include and linked with oracle/_ref/libSPERR_ref.so (itself compiled from
// /root/reference/src/*.cpp).  Nothing from the reference is copied into this repository; this
// file only calls the reference's public methods so that tests can compare the oracle
// restatement (sperr_oracle.c) and the HIP path with the reference stage by stage.
#include <cstring>
#include <vector>

#include "CDF97.h"
#include "Conditioner.h"
#include "SPECK1D_INT_DEC.h"
#include "SPECK1D_INT_ENC.h"
#include "SPECK2D_FLT.h"
#include "SPECK3D_FLT.h"
#include "SPERR3D_OMP_D.h"
#include "SPECK3D_INT_DEC.h"
#include "SPECK3D_INT_ENC.h"
#include "sperr_helper.h"

namespace {

template <typename T, typename ENC = sperr::SPECK3D_INT_ENC<T>>
int encode_as(const uint64_t* coeffs, const uint64_t* signs, sperr::dims_type dims, size_t budget,
              std::vector<uint8_t>& out)
{
  const size_t n = dims[0] * dims[1] * dims[2];
  std::vector<T> c(n);
  for (size_t i = 0; i < n; i++)
    c[i] = static_cast<T>(coeffs[i]);
  sperr::Bitmask mask(n);
  mask.use_bitstream(signs);
  ENC enc;
  enc.set_dims(dims);
  enc.set_budget(budget);
  if (enc.use_coeffs(std::move(c), std::move(mask)) != sperr::RTNType::Good)
    return 1;
  enc.encode();
  enc.append_encoded_bitstream(out);
  return 0;
}

template <typename T, typename DEC = sperr::SPECK3D_INT_DEC<T>>
int decode_as(const uint8_t* stream, size_t len, sperr::dims_type dims, uint64_t* coeffs,
              uint64_t* signs)
{
  const size_t n = dims[0] * dims[1] * dims[2];
  DEC dec;
  dec.set_dims(dims);
  dec.use_bitstream(stream, len);
  dec.decode();
  const auto& c = dec.view_coeffs();
  for (size_t i = 0; i < n; i++)
    coeffs[i] = c[i];
  const auto& words = dec.view_signs().view_buffer();
  std::memcpy(signs, words.data(), ((n + 63) / 64) * sizeof(uint64_t));
  return 0;
}

}  // namespace

extern "C" {

void refp_dwt3d(double* buf, size_t dx, size_t dy, size_t dz)
{
  sperr::CDF97 cdf;
  cdf.copy_data(buf, dx * dy * dz, {dx, dy, dz});
  cdf.dwt3d();
  const auto& v = cdf.view_data();
  std::memcpy(buf, v.data(), v.size() * sizeof(double));
}

void refp_idwt3d(double* buf, size_t dx, size_t dy, size_t dz)
{
  sperr::CDF97 cdf;
  cdf.copy_data(buf, dx * dy * dz, {dx, dy, dz});
  cdf.idwt3d();
  const auto& v = cdf.view_data();
  std::memcpy(buf, v.data(), v.size() * sizeof(double));
}

// returns 1 for a constant field
int refp_condition(double* buf, size_t dx, size_t dy, size_t dz, uint8_t header[17])
{
  sperr::Conditioner c;
  std::vector<double> v(buf, buf + dx * dy * dz);
  auto h = c.condition(v, {dx, dy, dz});
  std::memcpy(header, h.data(), 17);
  std::memcpy(buf, v.data(), v.size() * sizeof(double));
  return c.is_constant(h[0]) ? 1 : 0;
}

// width in {1,2,4,8}; *out is malloc'd
int refp_speck3d_encode(const uint64_t* coeffs, const uint64_t* signs, size_t dx, size_t dy,
                        size_t dz, size_t budget, int width, uint8_t** out, size_t* out_len)
{
  std::vector<uint8_t> s;
  const sperr::dims_type dims = {dx, dy, dz};
  int rtn = 1;
  switch (width) {
    case 1: rtn = encode_as<uint8_t>(coeffs, signs, dims, budget, s); break;
    case 2: rtn = encode_as<uint16_t>(coeffs, signs, dims, budget, s); break;
    case 4: rtn = encode_as<uint32_t>(coeffs, signs, dims, budget, s); break;
    case 8: rtn = encode_as<uint64_t>(coeffs, signs, dims, budget, s); break;
  }
  if (rtn)
    return rtn;
  *out = static_cast<uint8_t*>(std::malloc(s.size()));
  std::memcpy(*out, s.data(), s.size());
  *out_len = s.size();
  return 0;
}

// integer width is chosen from the header's num_bitplanes exactly as SPECK_FLT::use_bitstream does
int refp_speck3d_decode(const uint8_t* stream, size_t len, size_t dx, size_t dy, size_t dz,
                        uint64_t* coeffs, uint64_t* signs)
{
  const sperr::dims_type dims = {dx, dy, dz};
  const auto planes = sperr::speck_int_get_num_bitplanes(stream);
  if (planes <= 8)
    return decode_as<uint8_t>(stream, len, dims, coeffs, signs);
  if (planes <= 16)
    return decode_as<uint16_t>(stream, len, dims, coeffs, signs);
  if (planes <= 32)
    return decode_as<uint32_t>(stream, len, dims, coeffs, signs);
  return decode_as<uint64_t>(stream, len, dims, coeffs, signs);
}

// the coder of the outlier list (SPECK1D_INT_ENC / _DEC over a length-n array, no bit budget)
int refp_speck1d_encode(const uint64_t* coeffs, const uint64_t* signs, size_t n, int width,
                        uint8_t** out, size_t* out_len)
{
  std::vector<uint8_t> s;
  const sperr::dims_type dims = {n, 1, 1};
  int rtn = 1;
  switch (width) {
    case 1: rtn = encode_as<uint8_t, sperr::SPECK1D_INT_ENC<uint8_t>>(coeffs, signs, dims, 0, s); break;
    case 2: rtn = encode_as<uint16_t, sperr::SPECK1D_INT_ENC<uint16_t>>(coeffs, signs, dims, 0, s); break;
    case 4: rtn = encode_as<uint32_t, sperr::SPECK1D_INT_ENC<uint32_t>>(coeffs, signs, dims, 0, s); break;
    case 8: rtn = encode_as<uint64_t, sperr::SPECK1D_INT_ENC<uint64_t>>(coeffs, signs, dims, 0, s); break;
  }
  if (rtn)
    return rtn;
  *out = static_cast<uint8_t*>(std::malloc(s.size()));
  std::memcpy(*out, s.data(), s.size());
  *out_len = s.size();
  return 0;
}

int refp_speck1d_decode(const uint8_t* stream, size_t len, size_t n, uint64_t* coeffs,
                        uint64_t* signs)
{
  const sperr::dims_type dims = {n, 1, 1};
  const auto planes = sperr::speck_int_get_num_bitplanes(stream);
  if (planes <= 8)
    return decode_as<uint8_t, sperr::SPECK1D_INT_DEC<uint8_t>>(stream, len, dims, coeffs, signs);
  if (planes <= 16)
    return decode_as<uint16_t, sperr::SPECK1D_INT_DEC<uint16_t>>(stream, len, dims, coeffs, signs);
  if (planes <= 32)
    return decode_as<uint32_t, sperr::SPECK1D_INT_DEC<uint32_t>>(stream, len, dims, coeffs, signs);
  return decode_as<uint64_t, sperr::SPECK1D_INT_DEC<uint64_t>>(stream, len, dims, coeffs, signs);
}

// one chunk through the reference's SPECK3D_FLT in fixed-rate mode; *out is malloc'd
int refp_chunk_compress_rate(const double* vals, size_t dx, size_t dy, size_t dz, double bpp,
                             uint8_t** out, size_t* out_len)
{
  sperr::SPECK3D_FLT flt;
  flt.copy_data(vals, dx * dy * dz);
  flt.set_dims({dx, dy, dz});
  flt.set_bitrate(bpp);
  if (flt.compress() != sperr::RTNType::Good)
    return 1;
  std::vector<uint8_t> s;
  flt.append_encoded_bitstream(s);
  *out = static_cast<uint8_t*>(std::malloc(s.size()));
  std::memcpy(*out, s.data(), s.size());
  *out_len = s.size();
  return 0;
}

int refp_chunk_decompress(const uint8_t* stream, size_t len, size_t dx, size_t dy, size_t dz,
                          double* out)
{
  sperr::SPECK3D_FLT flt;
  flt.set_dims({dx, dy, dz});
  if (flt.use_bitstream(stream, len) != sperr::RTNType::Good)
    return 1;
  if (flt.decompress() != sperr::RTNType::Good)
    return 2;
  const auto& v = flt.view_decoded_data();
  std::memcpy(out, v.data(), v.size() * sizeof(double));
  return 0;
}

size_t refp_chunk_volume(size_t vx, size_t vy, size_t vz, size_t cx, size_t cy, size_t cz,
                         size_t* out6, size_t cap)
{
  auto chunks = sperr::chunk_volume({vx, vy, vz}, {cx, cy, cz});
  for (size_t i = 0; i < chunks.size() && i < cap; i++)
    for (size_t k = 0; k < 6; k++)
      out6[i * 6 + k] = chunks[i][k];
  return chunks.size();
}


// SPERR3D_OMP_D::decompress(p, multi_res = true): the volume (malloc'd doubles) and every level of
// the hierarchy (malloc'd doubles, coarsest first); returns the number of levels or -1
int refp_decomp_multi_res(const uint8_t* stream, size_t len, size_t nthreads, size_t dims[3],
                          double** vol, size_t (*ldims)[3], double** levels, size_t cap)
{
  sperr::SPERR3D_OMP_D dec;
  dec.set_num_threads(nthreads);
  if (dec.use_bitstream(stream, len) != sperr::RTNType::Good)
    return -1;
  if (dec.decompress(stream, true) != sperr::RTNType::Good)
    return -1;
  const auto d = dec.get_dims();
  const auto cd = dec.get_chunk_dims();
  for (int a = 0; a < 3; a++)
    dims[a] = d[a];
  const auto& v = dec.view_decoded_data();
  *vol = static_cast<double*>(std::malloc(v.size() * sizeof(double)));
  std::memcpy(*vol, v.data(), v.size() * sizeof(double));
  const auto res = sperr::coarsened_resolutions(d, cd);
  const auto& h = dec.view_hierarchy();
  if (h.size() != res.size() || h.size() > cap)
    return -1;
  for (size_t i = 0; i < h.size(); i++) {
    for (int a = 0; a < 3; a++)
      ldims[i][a] = res[i][a];
    levels[i] = static_cast<double*>(std::malloc(h[i].size() * sizeof(double)));
    std::memcpy(levels[i], h[i].data(), h[i].size() * sizeof(double));
  }
  return static_cast<int>(h.size());
}

// SPECK2D_FLT::decompress(multi_res = true) (what utilities/sperr2d.cpp:352-366 does): the slice
// and every level of its hierarchy, coarsest first; level_dims holds x y per level
int refp_decomp_2d_multi_res(const void* src, size_t len, size_t dimx, size_t dimy, void** dst,
                             size_t* nlev, size_t* level_dims, double** levels)
{
  sperr::SPECK2D_FLT dec;
  const sperr::dims_type dims{dimx, dimy, 1};
  dec.set_dims(dims);
  if (dec.use_bitstream(src, len) != sperr::RTNType::Good)
    return -1;
  if (dec.decompress(true) != sperr::RTNType::Good)
    return -1;
  const auto h = dec.release_hierarchy();
  const auto v = dec.release_decoded_data();
  const auto res = sperr::coarsened_resolutions(dims);
  if (h.size() != res.size() || h.size() > 16)
    return -1;
  for (size_t i = 0; i < h.size(); i++) {
    level_dims[2 * i] = res[i][0];
    level_dims[2 * i + 1] = res[i][1];
    levels[i] = static_cast<double*>(std::malloc(h[i].size() * sizeof(double)));
    std::memcpy(levels[i], h[i].data(), h[i].size() * sizeof(double));
  }
  *nlev = h.size();
  *dst = std::malloc(v.size() * sizeof(double));
  std::memcpy(*dst, v.data(), v.size() * sizeof(double));
  return 0;
}

// SPECK3D_FLT::integer_len() of the reference's own encoder and decoder for one chunk in PSNR mode
// (what test_scripts/speck3d_flt_unit_test.cpp:63-147 asserts): widths[0] = encoder, [1] = decoder.
int refp_integer_len_psnr(const double* vals, size_t dx, size_t dy, size_t dz, double psnr,
                          size_t widths[2])
{
  sperr::SPECK3D_FLT enc;
  enc.set_dims({dx, dy, dz});
  enc.set_psnr(psnr);
  enc.copy_data(vals, dx * dy * dz);
  if (enc.compress() != sperr::RTNType::Good)
    return 1;
  std::vector<uint8_t> stream;
  enc.append_encoded_bitstream(stream);
  sperr::SPECK3D_FLT dec;
  dec.set_dims({dx, dy, dz});
  if (dec.use_bitstream(stream.data(), stream.size()) != sperr::RTNType::Good ||
      dec.decompress() != sperr::RTNType::Good)
    return 1;
  widths[0] = enc.integer_len();
  widths[1] = dec.integer_len();
  return 0;
}

}  // extern "C"
