// bit_words.h -- 64 bits at a time: the word-level steps of the decoder's pixel passes (speck_dec.hip), shared with the
// CPU model (tests/model/speck_model.cpp, tests/test_speck_model.py checks them against bit-by-bit loops).
#ifndef SPERR_AMD_BIT_WORDS_H
#define SPERR_AMD_BIT_WORDS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define BW_HD __host__ __device__ __forceinline__
#else
#define BW_HD inline
#endif

namespace sperrhip {

// Bits of `x0` / `x1` (bit i: candidate i) spread to the set positions of `m` in order -- the parallel-suffix "expand" of
// Hacker's Delight 7-5, its mask half shared by the two.  (The pixel passes of the decoder walked a word's candidates
// one set bit at a time: a wavefront took as many rounds as its fullest word had candidates.)
BW_HD void spread_under_mask(uint64_t m, uint64_t& x0, uint64_t& x1)
{
  const uint64_t m0 = m;
  uint64_t mk = ~m << 1, mv[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    uint64_t mp = mk ^ (mk << 1);
    mp ^= mp << 2;
    mp ^= mp << 4;
    mp ^= mp << 8;
    mp ^= mp << 16;
    mp ^= mp << 32;
    mv[i] = mp & m;
    m = (m ^ mv[i]) | (mv[i] >> (1 << i));
    mk &= ~mp;
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    x0 = (x0 & ~mv[i]) | ((x0 << (1 << i)) & mv[i]);
    x1 = (x1 & ~mv[i]) | ((x1 << (1 << i)) & mv[i]);
  }
  x0 &= m0;
  x1 &= m0;
}

// The other way round: the bits of `x0` / `x1` at the set positions of `m`, packed to the bottom in order ("compress",
// Hacker's Delight 7-4).
BW_HD void gather_under_mask(uint64_t m, uint64_t& x0, uint64_t& x1)
{
  x0 &= m;
  x1 &= m;
  uint64_t mk = ~m << 1;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    uint64_t mp = mk ^ (mk << 1);
    mp ^= mp << 2;
    mp ^= mp << 4;
    mp ^= mp << 8;
    mp ^= mp << 16;
    mp ^= mp << 32;
    const uint64_t mv = mp & m;
    m = (m ^ mv) | (mv >> (1 << i));
    const uint64_t t0 = x0 & mv, t1 = x1 & mv;
    x0 = (x0 ^ t0) | (t0 >> (1 << i));
    x1 = (x1 ^ t1) | (t1 >> (1 << i));
    mk &= ~mp;
  }
}

// LIP scan (src/SPECK_INT.cpp:310-357 read backwards): a bit starts a token when the run of 1s in front of it is of
// even length (a 1 at a token's start is followed by its sign) -- the escaped characters of a run of backslashes,
// worked out for 64 bits at once with one addition: odd-length runs are found by the carry they send past their end.
// `parity` in: the word's first bit is a sign; out: the next word's first bit is one.
BW_HD uint64_t lip_token_starts(uint64_t x, uint32_t& parity)
{
  const uint64_t even = 0x5555555555555555ull;
  const uint64_t bs = x & ~(uint64_t)parity;
  const uint64_t follows = (bs << 1) | (uint64_t)parity;
  const uint64_t oddStarts = bs & ~even & ~follows;
  const uint64_t sum = oddStarts + bs;
  parity = sum < bs ? 1u : 0u;
  return ~((even ^ (sum << 1)) & follows);
}

}  // namespace sperrhip

#endif
