// host_fuzz.cpp -- the byte-level host code of the container layer
// (sperr_amd/csrc/host_container.hpp: chunk grid, header parsing, progressive truncation) under
// AddressSanitizer + UndefinedBehaviorSanitizer on the CPU, fed with damaged containers: the same
// damage classes as tools/fuzz_corrupt.py applies on the GPU (bytes flipped anywhere / in the
// header, containers cut short, header fields pushed to their extremes).  The GPU cannot run under
// a sanitizer on this pool; this part of the library parses untrusted bytes and needs none.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all host_fuzz.cpp
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../sperr_amd/csrc/host_container.hpp"

using namespace sperrhip::hostc;

static int g_checked = 0, g_accepted = 0;

static void check_parse(const std::vector<uint8_t>& c)
{
  for (size_t hlen : {size_t(14), size_t(20), c.size() / 2, c.size()}) {
    if (hlen > c.size() || hlen < 14)
      continue;
    ContainerInfo ci;
    size_t need = 0;
    const int r = parse_container_host(c.data(), hlen, c.size(), ci, &need);
    g_checked++;
    if (r == 0) {
      g_accepted++;
      // what a caller relies on: the chunk streams tile the container exactly
      uint64_t at = (ci.multi ? 20 : 14) + 4 * ci.off.size();
      for (size_t i = 0; i < ci.off.size(); i++) {
        if (ci.off[i] != at || ci.off[i] + ci.len[i] > c.size())
          { std::fprintf(stderr, "invariant failed at line %d\n", __LINE__); std::abort(); }
        at += ci.len[i];
      }
      if (at != c.size() || ci.nvals != ci.vol[0] * ci.vol[1] * ci.vol[2] || ci.off.size() != chunk_count(ci.vol, ci.chunk))
        { std::fprintf(stderr, "invariant failed at line %d\n", __LINE__); std::abort(); }
    }
    else if (r == 1 && need <= hlen)
      { std::fprintf(stderr, "invariant failed at line %d\n", __LINE__); std::abort(); }   // asked for more header than it was given, but did not need it
  }
}

static void check_trunc(const std::vector<uint8_t>& c, unsigned pct)
{
  void* dst = nullptr;
  size_t n = 0;
  const int r = truncate_container(c.data(), c.size(), pct, &dst, &n);
  g_checked++;
  if (r == 0) {
    std::vector<uint8_t> t(static_cast<uint8_t*>(dst), static_cast<uint8_t*>(dst) + n);
    free(dst);
    if (n > c.size())
      { std::fprintf(stderr, "invariant failed at line %d\n", __LINE__); std::abort(); }
    // (like the reference's progressive_truncate, src/SPERR3D_Stream_Tools.cpp:134-226, truncation
    // only looks at the lengths it needs; but a container the parser accepts must stay acceptable)
    ContainerInfo ci, ci0;
    size_t need = 0;
    if (parse_container_host(c.data(), c.size(), c.size(), ci0, &need) == 0 &&
        parse_container_host(t.data(), t.size(), t.size(), ci, &need) != 0)
      { std::fprintf(stderr, "invariant failed at line %d\n", __LINE__); std::abort(); }
    // and a second call with a non-NULL *dst must be refused
    void* again = t.data();
    if (truncate_container(c.data(), c.size(), pct, &again, &n) != 1)
      { std::fprintf(stderr, "invariant failed at line %d\n", __LINE__); std::abort(); }
  }
  else if (dst != nullptr)
    { std::fprintf(stderr, "invariant failed at line %d\n", __LINE__); std::abort(); }
}

static std::vector<uint8_t> make_container(std::mt19937_64& rng, Dims vol, Dims chunk, bool is_float)
{
  Dims cd;
  for (int a = 0; a < 3; a++)
    cd[a] = std::min(std::max<size_t>(1, chunk[a]), vol[a]);
  const auto chunks = chunk_volume(vol, cd);
  const bool multi = chunks.size() > 1;
  std::vector<uint8_t> c(multi ? 20 : 14);
  c[0] = 0;
  c[1] = (uint8_t)(0x40 | (is_float ? 0x20 : 0) | (multi ? 0x10 : 0));
  const uint32_t v3[3] = {(uint32_t)vol[0], (uint32_t)vol[1], (uint32_t)vol[2]};
  memcpy(c.data() + 2, v3, 12);
  if (multi) {
    const uint16_t c3[3] = {(uint16_t)cd[0], (uint16_t)cd[1], (uint16_t)cd[2]};
    memcpy(c.data() + 14, c3, 6);
  }
  std::vector<uint32_t> lens;
  for (size_t i = 0; i < chunks.size(); i++)
    lens.push_back(17 + (uint32_t)(rng() % 400));
  for (uint32_t l : lens) {
    uint8_t b[4];
    memcpy(b, &l, 4);
    c.insert(c.end(), b, b + 4);
  }
  for (uint32_t l : lens)
    for (uint32_t k = 0; k < l; k++)
      c.push_back((uint8_t)rng());
  return c;
}

int main(int argc, char** argv)
{
  const int rounds = argc > 1 ? atoi(argv[1]) : 300;
  std::mt19937_64 rng(12345);
  const Dims vols[] = {{40, 48, 56}, {17, 17, 17}, {64, 64, 64}, {100, 70, 33}, {1, 1, 1}, {250, 3, 2}, {9, 9, 300}};
  const Dims chks[] = {{32, 32, 32}, {17, 17, 17}, {16, 16, 16}, {30, 40, 8}, {1, 1, 1}, {64, 64, 64}, {4, 4, 7}};
  for (int r = 0; r < rounds; r++) {
    const size_t k = rng() % 7;
    std::vector<uint8_t> good = make_container(rng, vols[k], chks[rng() % 7], rng() & 1);
    check_parse(good);
    for (unsigned pct : {0u, 1u, 30u, 99u, 100u, 250u})
      check_trunc(good, pct);
    for (int m = 0; m < 40; m++) {
      std::vector<uint8_t> bad = good;
      switch (rng() % 6) {
        case 0:   // bytes flipped anywhere
          for (int f = 0; f < 1 + (int)(rng() % 4); f++)
            bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8));
          break;
        case 1:   // bytes flipped in the header
          for (int f = 0; f < 1 + (int)(rng() % 3); f++)
            bad[rng() % std::min<size_t>(bad.size(), 40)] ^= (uint8_t)(1u << (rng() % 8));
          break;
        case 2:   // cut short
          bad.resize(rng() % bad.size());
          break;
        case 3: {  // a dimension pushed to an extreme
          const uint32_t ext[] = {0u, 1u, 0x7fffffffu, 0xffffffffu, 0x10000u};
          const uint32_t v = ext[rng() % 5];
          memcpy(bad.data() + 2 + 4 * (rng() % 3), &v, 4);
          break;
        }
        case 4: {  // a chunk dimension pushed to an extreme
          if (bad.size() >= 20) {
            const uint16_t ext[] = {0, 1, 0xffff, 2};
            const uint16_t v = ext[rng() % 4];
            memcpy(bad.data() + 14 + 2 * (rng() % 3), &v, 2);
          }
          break;
        }
        default: {  // a chunk length pushed to an extreme
          if (bad.size() >= 28) {
            const uint32_t ext[] = {0u, 0xffffffffu, 0x80000000u, 16u};
            const uint32_t v = ext[rng() % 4];
            memcpy(bad.data() + 20 + 4 * (rng() % 2), &v, 4);
          }
          break;
        }
      }
      if (bad.size() < 14)
        continue;   // (the entry points refuse anything below 18 / 20 bytes before parsing)
      check_parse(bad);
      if (bad.size() >= 20)
        check_trunc(bad, (unsigned)(rng() % 120));
    }
  }
  // header dimensions that imply more chunks than a size_t holds / than the container has bytes for
  for (Dims v : {Dims{0xffffffffu, 0xffffffffu, 0xffffffffu}, Dims{0x7fffffffu, 0x7fffffffu, 0x7fffffffu}}) {
    std::vector<uint8_t> c(220, 0);
    c[1] = 0x40 | 0x20 | 0x10;
    const uint32_t v3[3] = {(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2]};
    memcpy(c.data() + 2, v3, 12);
    const uint16_t c3[3] = {1, 1, 1};
    memcpy(c.data() + 14, c3, 6);
    check_parse(c);
    check_trunc(c, 50);
    if (chunk_count(v, Dims{1, 1, 1}) != SIZE_MAX)
      { std::fprintf(stderr, "invariant failed at line %d\n", __LINE__); std::abort(); }
  }
  std::printf("host fuzz ok: %d calls, %d accepted\n", g_checked, g_accepted);
  return 0;
}
