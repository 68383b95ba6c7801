// sperr_helper.h -- the reference's header name for its types and host helpers
// (/root/reference/include/sperr_helper.h:35-191), for callers such as utilities/sperr3d.cpp that
// use them next to the driver classes.  Types, RTNType and the driver classes come from
// ../sperr_hip.hpp; the helpers below are header-only host code written for this library (the
// geometry ones are the rules the chunk farm itself follows, see sperr_amd/csrc/engine.hip).
//
// Statistics note: calc_stats / calc_mean_var sum in blocks of 8192 / 16384 values and then over
// the blocks, left to right, like the reference (src/sperr_helper.cpp:428-511,616-659), so the
// figures a tool prints are the same to the last digit.
#ifndef SPERR_HIP_COMPAT_SPERR_HELPER_H
#define SPERR_HIP_COMPAT_SPERR_HELPER_H

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <optional>
#include <string>

#include "SperrConfig.h"
#include "../sperr_hip.hpp"

namespace sperr {

template <typename T>
using vec_type = std::vector<T>;

// ---- geometry (src/sperr_helper.cpp:36-146,542-592) ---------------------------------------------

// an axis of `len` samples is halved (the longer half stays) while it has at least 9; 6 at most
inline auto num_of_xforms(size_t len) -> size_t
{
  size_t n = 0;
  for (; len >= 9 && n < 6; len -= len / 2)
    n++;
  return n;
}

inline auto num_of_partitions(size_t len) -> size_t
{
  size_t n = 0;
  for (; len > 1; len -= len / 2)
    n++;
  return n;
}

// {approximation length, detail length of the last step} after `lev` halvings
inline auto calc_approx_detail_len(size_t orig_len, size_t lev) -> std::array<size_t, 2>
{
  size_t lo = orig_len, hi = 0;
  while (lev--) {
    hi = lo / 2;
    lo -= hi;
  }
  return {lo, hi};
}

// levels of the dyadic 3D transform, or nothing when the shape takes the wavelet-packet one
inline auto can_use_dyadic(dims_type d) -> std::optional<size_t>
{
  if (d[1] < 2 || d[2] < 2)
    return {};
  const size_t xy = num_of_xforms(std::min(d[0], d[1])), z = num_of_xforms(d[2]);
  if (xy == z || (xy >= 5 && z >= 5))
    return std::min(xy, z);
  return {};
}

// coarsest first; the library answers (sperrhip_multires_levels*, no device is touched)
inline auto coarsened_resolutions(dims_type d) -> std::vector<dims_type>
{
  std::vector<dims_type> out;
  size_t nlev = 0, ld[48];
  if (d[2] > 1) {
    if (sperrhip_multires_levels(d[0], d[1], d[2], d[0], d[1], d[2], &nlev, ld) == 0)
      for (size_t h = 0; h < nlev; h++)
        out.push_back({ld[3 * h], ld[3 * h + 1], ld[3 * h + 2]});
  }
  else if (sperrhip_multires_levels_2d(d[0], d[1], &nlev, ld) == 0)
    for (size_t h = 0; h < nlev; h++)
      out.push_back({ld[2 * h], ld[2 * h + 1], 1});
  return out;
}

inline auto coarsened_resolutions(dims_type vol, dims_type chunk) -> std::vector<dims_type>
{
  std::vector<dims_type> out;
  for (int a = 0; a < 3; a++)
    if (chunk[a] == 0 || vol[a] % chunk[a] != 0)
      return out;
  out = coarsened_resolutions(chunk);
  for (auto& r : out)
    for (int a = 0; a < 3; a++)
      r[a] *= vol[a] / chunk[a];
  return out;
}

// {x0, lenx, y0, leny, z0, lenz} per chunk, x fastest; a remainder of more than half a chunk is a
// chunk of its own, a shorter one joins its neighbour
inline auto chunk_volume(dims_type vol, dims_type chunk) -> std::vector<std::array<size_t, 6>>
{
  size_t n[3];
  for (int a = 0; a < 3; a++) {
    n[a] = vol[a] / chunk[a] + (vol[a] % chunk[a] > chunk[a] / 2 ? 1 : 0);
    if (n[a] == 0)
      n[a] = 1;
  }
  auto first = [&](int a, size_t i) { return i * chunk[a]; };
  auto length = [&](int a, size_t i) { return (i + 1 == n[a] ? vol[a] : (i + 1) * chunk[a]) - i * chunk[a]; };
  std::vector<std::array<size_t, 6>> out;
  out.reserve(n[0] * n[1] * n[2]);
  for (size_t z = 0; z < n[2]; z++)
    for (size_t y = 0; y < n[1]; y++)
      for (size_t x = 0; x < n[0]; x++)
        out.push_back({first(0, x), length(0, x), first(1, y), length(1, y), first(2, z), length(2, z)});
  return out;
}

// ---- flag bytes: bool i of the array is bit 7 - i of the byte ------------------------------------
inline auto pack_8_booleans(std::array<bool, 8> b) -> uint8_t
{
  unsigned v = 0;
  for (int i = 0; i < 8; i++)
    v |= (b[i] ? 1u : 0u) << (7 - i);
  return (uint8_t)v;
}

inline auto unpack_8_booleans(uint8_t v) -> std::array<bool, 8>
{
  std::array<bool, 8> b{};
  for (int i = 0; i < 8; i++)
    b[i] = ((v >> (7 - i)) & 1u) != 0;
  return b;
}

// ---- files ----------------------------------------------------------------------------------------
inline auto write_n_bytes(std::string filename, size_t n_bytes, const void* buffer) -> RTNType
{
  std::FILE* f = std::fopen(filename.c_str(), "wb");
  if (!f)
    return RTNType::IOError;
  const bool ok = std::fwrite(buffer, 1, n_bytes, f) == n_bytes;
  return (std::fclose(f) == 0 && ok) ? RTNType::Good : RTNType::IOError;
}

// the first n_bytes of a file; empty when the file is shorter or unreadable
inline auto read_n_bytes(std::string filename, size_t n_bytes) -> vec8_type
{
  vec8_type buf;
  if (std::FILE* f = std::fopen(filename.c_str(), "rb")) {
    buf.resize(n_bytes);
    if (std::fread(buf.data(), 1, n_bytes, f) != n_bytes)
      buf.clear();
    std::fclose(f);
  }
  return buf;
}

// empty when the file is unreadable or its size is not a multiple of sizeof(T)
template <typename T>
auto read_whole_file(std::string filename) -> vec_type<T>
{
  vec_type<T> buf;
  std::FILE* f = std::fopen(filename.c_str(), "rb");
  if (!f)
    return buf;
  long len = -1;
  if (std::fseek(f, 0, SEEK_END) == 0)
    len = std::ftell(f);
  if (len >= 0 && (size_t)len % sizeof(T) == 0 && std::fseek(f, 0, SEEK_SET) == 0) {
    buf.resize((size_t)len / sizeof(T));
    if (std::fread(buf.data(), sizeof(T), buf.size(), f) != buf.size())
      buf.clear();
  }
  std::fclose(f);
  return buf;
}

// ---- statistics -------------------------------------------------------------------------------------
template <typename T>
auto kahan_summation(const T* arr, size_t len) -> T
{
  T sum = 0, lost = 0;
  for (size_t i = 0; i < len; i++) {
    const T y = arr[i] - lost, t = sum + y;
    lost = (t - sum) - y;
    sum = t;
  }
  return sum;
}

namespace detail {
// sum of f(i) over [0, len): blocks of `block` values left to right, then the block sums left to
// right, the tail block last -- all in T
template <typename T, typename F>
T blocked_sum(size_t len, size_t block, F f)
{
  T total = 0;
  for (size_t b = 0; b < len; b += block) {
    if (len - b < block)
      break;
    T s = 0;
    for (size_t i = b; i < b + block; i++)
      s += f(i);
    total += s;
  }
  T tail = 0;
  for (size_t i = len - len % block; i < len; i++)
    tail += f(i);
  return total + tail;
}
}  // namespace detail

// {rmse, L-infinity, psnr (dB, against the range of arr1), min of arr1, max of arr1}
template <typename T>
auto calc_stats(const T* arr1, const T* arr2, size_t len, size_t = 0) -> std::array<T, 5>
{
  const auto mm = std::minmax_element(arr1, arr1 + len);
  const T lo = *mm.first, hi = *mm.second;
  if (std::equal(arr1, arr1 + len, arr2))
    return {T(0), T(0), std::numeric_limits<T>::infinity(), lo, hi};
  T linf = 0;
  for (size_t i = 0; i < len; i++)
    linf = std::max(linf, std::abs(arr1[i] - arr2[i]));
  const T mse = detail::blocked_sum<T>(len, 8192, [&](size_t i) {
                  const T d = std::abs(arr1[i] - arr2[i]);
                  return d * d;
                }) / T(len);
  return {std::sqrt(mse), linf, std::log10((hi - lo) * (hi - lo) / mse) * T(10), lo, hi};
}

// {mean, variance}; {NaN, NaN} for an empty array
template <typename T>
auto calc_mean_var(const T* arr, size_t len, size_t = 0) -> std::array<T, 2>
{
  if (len == 0)
    return {std::numeric_limits<T>::quiet_NaN(), std::numeric_limits<T>::quiet_NaN()};
  const T mean = detail::blocked_sum<T>(len, 16384, [&](size_t i) { return arr[i]; }) / T(len);
  const T var = detail::blocked_sum<T>(len, 16384, [&](size_t i) { return (arr[i] - mean) * (arr[i] - mean); }) / T(len);
  return {mean, var};
}

template <typename T>
auto msb_position(T v) -> int8_t
{
  int8_t pos = -1;
  for (; v; v >>= 1)
    pos++;
  return pos;
}

}  // namespace sperr

#endif
