// How much the placement of a lone wavefront's hot loop in memory matters (k_lis_mx's hop loop, variant 2 of
// hop_loop.cpp): the loop head at 128-byte alignment + 4 k bytes, k = 0 .. 31.
//   hipcc --offload-arch=gfx950 -O2 -o hop_align tools/micro/hop_align.cpp && ./hop_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define STR2(x) #x
#define STR(x) STR2(x)

template <int K>
__device__ __forceinline__ void hop_word(uint64_t m, uint32_t ecb, uint32_t rlo, uint32_t rhi, uint32_t& oo, uint32_t& idx, uint64_t& cm, uint64_t& im)
{
  uint32_t z_, ec_;
  uint64_t mm_;
  asm volatile(
      ".p2align 7\n\t"
      ".fill %c[k], 4, 0xbf800000\n\t"
      "1:\n\t"
      "s_lshr_b64 %[mm], %[m], %[oo]\n\t"
      "s_ff1_i32_b64 %[z], %[mm]\n\t"
      "s_cmp_eq_u64 %[mm], 0\n\t"
      "s_cbranch_scc1 3f\n\t"
      "s_add_u32 %[idx], %[idx], %[z]\n\t"
      "s_add_u32 %[oo], %[oo], %[z]\n\t"
      "v_readlane_b32 %[ec], %[ecb], %[idx]\n\t"
      "v_readlane_b32 s96, %[rlo], %[oo]\n\t"
      "v_readlane_b32 s97, %[rhi], %[oo]\n\t"
      "s_bitset1_b64 %[im], %[idx]\n\t"
      "s_bitset1_b64 %[cm], %[oo]\n\t"
      "s_add_u32 %[idx], %[idx], 1\n\t"
      "s_bfe_u64 s[98:99], s[96:97], %[ec]\n\t"
      "s_add_u32 %[oo], %[oo], s98\n\t"
      "s_cbranch_scc0 1b\n\t"
      "3:\n\t"
      : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "=&s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_)
      : [m] "s"(m), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi), [k] "n"(K)
      : "scc", "s96", "s97", "s98", "s99");
}

template <int K>
__global__ void k_hops(uint64_t* out, const uint64_t* words, const uint32_t* rows, int nwords)
{
  const uint32_t lane = threadIdx.x;
  uint64_t cycles = 0;
  uint32_t hops = 0, sink = 0;
  for (int rep = 0; rep < 8; rep++)
    for (int w = 0; w < nwords; w++) {
      const uint64_t mv = words[w];
      const uint64_t m = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)mv) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(mv >> 32)) << 32);
      const uint32_t rlo = rows[(w * 64 + lane) * 4 + 0], rhi = rows[(w * 64 + lane) * 4 + 1];
      const uint32_t loc = rows[(w * 64 + lane) * 4 + 2] & 7u;
      uint32_t oo = 0u - 64u, idx = 0;
      uint64_t cm = 0, im = 0;
      const uint64_t t0 = __builtin_readcyclecounter();
      hop_word<K>(m, (loc << 3) | (8u << 16), rlo, rhi, oo, idx, cm, im);
      cycles += __builtin_readcyclecounter() - t0;
      hops += (uint32_t)__popcll(cm);
      sink += idx + (uint32_t)im + oo;
    }
  if (lane == 0) {
    out[0] = cycles;
    out[1] = hops;
    out[2] = sink;
  }
}

template <int K>
void run(uint64_t* d, const uint64_t* dw, const uint32_t* dr, int nwords)
{
  uint64_t h[4];
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_hops<K>, dim3(1), dim3(64), 0, 0, d, dw, dr, nwords);
    (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  }
  printf("loop head at 128 n + %3d: %8.1f cycles per hop\n", 4 * K, (double)h[0] / (double)h[1]);
}

int main()
{
  const int nwords = 512;
  uint64_t* hw = (uint64_t*)malloc(nwords * 8);
  uint32_t* hr = (uint32_t*)malloc(nwords * 64 * 16);
  srand(7);
  for (int w = 0; w < nwords; w++) {
    uint64_t m = 0;
    for (int b = 0; b < 64; b++)
      if (rand() % 5 == 0)
        m |= 1ull << b;
    hw[w] = m;
    for (int l = 0; l < 64; l++) {
      uint32_t lo = 0, hi = 0;
      for (int k = 0; k < 4; k++) {
        lo |= (uint32_t)(3 + rand() % 11) << (8 * k);
        hi |= (uint32_t)(3 + rand() % 11) << (8 * k);
      }
      hr[(w * 64 + l) * 4 + 0] = lo;
      hr[(w * 64 + l) * 4 + 1] = hi;
      hr[(w * 64 + l) * 4 + 2] = (uint32_t)rand();
      hr[(w * 64 + l) * 4 + 3] = 0;
    }
  }
  uint64_t *d, *dw;
  uint32_t* dr;
  (void)hipMalloc(&d, 32);
  (void)hipMalloc(&dw, nwords * 8);
  (void)hipMalloc(&dr, nwords * 64 * 16);
  (void)hipMemcpy(dw, hw, nwords * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dr, hr, nwords * 64 * 16, hipMemcpyHostToDevice);
#define R(k) run<k>(d, dw, dr, nwords);
  R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)
  R(16) R(17) R(18) R(19) R(20) R(21) R(22) R(23) R(24) R(25) R(26) R(27) R(28) R(29) R(30) R(31)
  return 0;
}
