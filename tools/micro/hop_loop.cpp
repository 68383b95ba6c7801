// Variants of k_lis_mx's hop loop (sperr_amd/csrc/speck_mx.hip) on one lone wavefront, synthetic words:
//   hipcc --offload-arch=gfx950 -O2 -o hop_loop tools/micro/hop_loop.cpp && ./hop_loop
// (one kernel per variant, every loop head at a 128-byte boundary: the placement alone moves a loop by 15 %, hop_align.cpp)
// A word of 64 stream bits, significant entries at the set bits of m, each followed by a split whose
// length + 1 is a byte of the lane's two row registers picked by the entry's class.  Cycles per hop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#ifndef HOP_ALIGN
#define HOP_ALIGN ".p2align 7\n\t"
#endif

template <int which>
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
      uint32_t oo = 0, idx = 0;
      uint64_t cm = 0, im = 0;
      const uint64_t t0 = __builtin_readcyclecounter();
      if constexpr (which == 0) {   // the loop as it is: select of the group, s_bfe_u32, two tests per hop
        const uint32_t ecb = ((loc & 3u) << 3) | (8u << 16) | ((loc >> 2) << 8);
        uint32_t t_, z_, ec_, lo_, hi_;
        uint64_t mm_;
        asm volatile(
            HOP_ALIGN "1:\n\t"
            "s_lshr_b64 %[mm], %[m], %[oo]\n\t"
            "s_ff1_i32_b64 %[z], %[mm]\n\t"
            "s_cmp_eq_u64 %[mm], 0\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_add_u32 %[idx], %[idx], %[z]\n\t"
            "s_add_u32 %[oo], %[oo], %[z]\n\t"
            "v_readlane_b32 %[ec], %[ecb], %[idx]\n\t"
            "v_readlane_b32 %[lo], %[rlo], %[oo]\n\t"
            "v_readlane_b32 %[hi], %[rhi], %[oo]\n\t"
            "s_bitset1_b64 %[im], %[idx]\n\t"
            "s_bitcmp1_b32 %[ec], 8\n\t"
            "s_bitset1_b64 %[cm], %[oo]\n\t"
            "s_cselect_b32 %[lo], %[hi], %[lo]\n\t"
            "s_bfe_u32 %[t], %[lo], %[ec]\n\t"
            "s_add_u32 %[idx], %[idx], 1\n\t"
            "s_add_u32 %[z], %[t], -3\n\t"
            "s_add_u32 %[oo], %[oo], %[t]\n\t"
            "s_cmp_gt_u32 %[z], 251\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_cmp_lt_u32 %[oo], 64\n\t"
            "s_cbranch_scc1 1b\n\t"
            "3:\n\t"
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "=&s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_), [lo] "=&s"(lo_),
              [hi] "=&s"(hi_), [t] "=&s"(t_)
            : [m] "s"(m), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc");
      }
      else if constexpr (which == 1) {   // s_bfe_u64 over the pair of row words: no select
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_;
        uint64_t mm_;
        asm volatile(
            HOP_ALIGN "1:\n\t"
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
            "s_add_u32 %[z], s98, -3\n\t"
            "s_add_u32 %[oo], %[oo], s98\n\t"
            "s_cmp_gt_u32 %[z], 251\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_cmp_lt_u32 %[oo], 64\n\t"
            "s_cbranch_scc1 1b\n\t"
            "3:\n\t"
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "=&s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_)
            : [m] "s"(m), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 2) {   // + the position biased by -64: the add's carry is the end of the word; no test of the length
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_;
        uint64_t mm_;
        oo -= 64u;
        asm volatile(
            HOP_ALIGN "1:\n\t"
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
            : [m] "s"(m), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 3) {   // + the rest of the word kept as a mask (no shift: s_ff1 gives the position itself)
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_, p_;
        uint64_t mm_ = m;
        asm volatile(
            "s_mov_b32 %[oo], 0\n\t"
            HOP_ALIGN "1:\n\t"
            "s_ff1_i32_b64 %[p], %[mm]\n\t"              // position of the next significant entry
            "s_cmp_eq_u64 %[mm], 0\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_sub_u32 %[z], %[p], %[oo]\n\t"
            "v_readlane_b32 s96, %[rlo], %[p]\n\t"
            "v_readlane_b32 s97, %[rhi], %[p]\n\t"
            "s_add_u32 %[idx], %[idx], %[z]\n\t"
            "v_readlane_b32 %[ec], %[ecb], %[idx]\n\t"
            "s_bitset1_b64 %[cm], %[p]\n\t"
            "s_bitset1_b64 %[im], %[idx]\n\t"
            "s_add_u32 %[idx], %[idx], 1\n\t"
            "s_bfe_u64 s[98:99], s[96:97], %[ec]\n\t"
            "s_add_u32 %[oo], %[p], s98\n\t"
            "s_cmp_gt_u32 %[oo], 63\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_lshl_b64 s[96:97], -1, %[oo]\n\t"
            "s_and_b64 %[mm], %[m], s[96:97]\n\t"
            "s_branch 1b\n\t"
            "3:\n\t"
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "+s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_), [p] "=&s"(p_)
            : [m] "s"(m), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 4) {   // variant 2 with the end-of-word test after the adds and the loads (s_ff1 of 0 is -1: undone at 3)
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_;
        uint64_t mm_;
        oo -= 64u;
        asm volatile(
            HOP_ALIGN "1:\n\t"
            "s_lshr_b64 %[mm], %[m], %[oo]\n\t"
            "s_ff1_i32_b64 %[z], %[mm]\n\t"
            "s_add_u32 %[oo], %[oo], %[z]\n\t"
            "s_add_u32 %[idx], %[idx], %[z]\n\t"
            "v_readlane_b32 s96, %[rlo], %[oo]\n\t"
            "v_readlane_b32 s97, %[rhi], %[oo]\n\t"
            "v_readlane_b32 %[ec], %[ecb], %[idx]\n\t"
            "s_cmp_eq_u64 %[mm], 0\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_bitset1_b64 %[cm], %[oo]\n\t"
            "s_bitset1_b64 %[im], %[idx]\n\t"
            "s_add_u32 %[idx], %[idx], 1\n\t"
            "s_bfe_u64 s[98:99], s[96:97], %[ec]\n\t"
            "s_add_u32 %[oo], %[oo], s98\n\t"
            "s_cbranch_scc0 1b\n\t"
            "s_branch 4f\n\t"
            "3:\n\t"
            "s_add_u32 %[oo], %[oo], 1\n\t"
            "s_add_u32 %[idx], %[idx], 1\n\t"
            "4:\n\t"
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "=&s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_)
            : [m] "s"(m), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 5) {   // the next significant position from a per-lane table (nine vector instructions per word)
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_, p_;
        const uint64_t sh = m >> lane;
        uint32_t a = (uint32_t)sh ? (uint32_t)__builtin_ctz((uint32_t)sh) : 0xffffffffu;
        uint32_t bq = (uint32_t)(sh >> 32) ? (uint32_t)__builtin_ctz((uint32_t)(sh >> 32)) : 32u;
        a = min(a, bq + 32u);
        const uint32_t nsb = min(lane + a, 64u) - 64u;   // biased by -64; 0: none
        oo -= 64u;
        asm volatile(
            HOP_ALIGN "1:\n\t"
            "v_readlane_b32 %[p], %[nsb], %[oo]\n\t"
            "s_nop 0\n\t"
            "s_sub_u32 %[z], %[p], %[oo]\n\t"
            "v_readlane_b32 s96, %[rlo], %[p]\n\t"
            "v_readlane_b32 s97, %[rhi], %[p]\n\t"
            "s_add_u32 %[idx], %[idx], %[z]\n\t"
            "s_cmp_eq_u32 %[p], 0\n\t"
            "v_readlane_b32 %[ec], %[ecb], %[idx]\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_bitset1_b64 %[cm], %[p]\n\t"
            "s_bitset1_b64 %[im], %[idx]\n\t"
            "s_add_u32 %[idx], %[idx], 1\n\t"
            "s_bfe_u64 s[98:99], s[96:97], %[ec]\n\t"
            "s_add_u32 %[oo], %[p], s98\n\t"
            "s_cbranch_scc0 1b\n\t"
            "3:\n\t"
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [z] "=&s"(z_), [ec] "=&s"(ec_), [p] "=&s"(p_)
            : [nsb] "v"(nsb), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 6) {   // variant 2 unrolled four times: one taken branch per four hops
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_;
        uint64_t mm_;
        oo -= 64u;
        asm volatile(
            HOP_ALIGN "1:\n\t"
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
            "s_cbranch_scc1 3f\n\t"
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
            "s_cbranch_scc1 3f\n\t"
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
            "s_cbranch_scc1 3f\n\t"
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
            : [m] "s"(m), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 7) {   // variant 2, the zero test's compare early and its branch late
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_;
        uint64_t mm_;
        oo -= 64u;
        asm volatile(
            HOP_ALIGN "1:\n\t"
            "s_lshr_b64 %[mm], %[m], %[oo]\n\t"
            "s_cmp_eq_u64 %[mm], 0\n\t"
            "s_ff1_i32_b64 %[z], %[mm]\n\t"
            "s_add_u32 %[oo], %[oo], %[z]\n\t"
            "s_add_u32 %[idx], %[idx], %[z]\n\t"
            "s_cbranch_scc1 3f\n\t"
            "v_readlane_b32 s96, %[rlo], %[oo]\n\t"
            "v_readlane_b32 s97, %[rhi], %[oo]\n\t"
            "v_readlane_b32 %[ec], %[ecb], %[idx]\n\t"
            "s_bitset1_b64 %[cm], %[oo]\n\t"
            "s_bitset1_b64 %[im], %[idx]\n\t"
            "s_add_u32 %[idx], %[idx], 1\n\t"
            "s_bfe_u64 s[98:99], s[96:97], %[ec]\n\t"
            "s_add_u32 %[oo], %[oo], s98\n\t"
            "s_cbranch_scc0 1b\n\t"
            "s_branch 4f\n\t"
            "3:\n\t"
            "s_add_u32 %[oo], %[oo], 1\n\t"
            "s_add_u32 %[idx], %[idx], 1\n\t"
            "4:\n\t"
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "=&s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_)
            : [m] "s"(m), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 8) {   // variant 2 without the zero test (bit 63 of the word set: what a sentinel would buy)
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_;
        uint64_t mm_;
        oo -= 64u;
        const uint64_t m_ = m | (1ull << 63);
        asm volatile(
            HOP_ALIGN "1:\n\t"
            "s_lshr_b64 %[mm], %[m], %[oo]\n\t"
            "s_ff1_i32_b64 %[z], %[mm]\n\t"
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
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "=&s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_)
            : [m] "s"(m_), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 9) {   // the word shifted along: by the zeros, then by the split (the second shift on the chain, the position off it)
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_;
        uint64_t mm_ = m;
        oo -= 64u;
        asm volatile(
            HOP_ALIGN "1:\n\t"
            "s_ff1_i32_b64 %[z], %[mm]\n\t"
            "s_cmp_eq_u64 %[mm], 0\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_add_u32 %[oo], %[oo], %[z]\n\t"
            "s_add_u32 %[idx], %[idx], %[z]\n\t"
            "s_lshr_b64 %[mm], %[mm], %[z]\n\t"
            "v_readlane_b32 s96, %[rlo], %[oo]\n\t"
            "v_readlane_b32 s97, %[rhi], %[oo]\n\t"
            "v_readlane_b32 %[ec], %[ecb], %[idx]\n\t"
            "s_bitset1_b64 %[cm], %[oo]\n\t"
            "s_bitset1_b64 %[im], %[idx]\n\t"
            "s_add_u32 %[idx], %[idx], 1\n\t"
            "s_bfe_u64 s[98:99], s[96:97], %[ec]\n\t"
            "s_lshr_b64 %[mm], %[mm], s98\n\t"
            "s_add_u32 %[oo], %[oo], s98\n\t"
            "s_cbranch_scc0 1b\n\t"
            "3:\n\t"
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "+s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_)
            : [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 10) {   // variant 9, the zero test behind the loads
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_;
        uint64_t mm_ = m;
        oo -= 64u;
        asm volatile(
            HOP_ALIGN "1:\n\t"
            "s_ff1_i32_b64 %[z], %[mm]\n\t"
            "s_cmp_eq_u64 %[mm], 0\n\t"
            "s_add_u32 %[oo], %[oo], %[z]\n\t"
            "s_add_u32 %[idx], %[idx], %[z]\n\t"
            "v_readlane_b32 s96, %[rlo], %[oo]\n\t"
            "v_readlane_b32 s97, %[rhi], %[oo]\n\t"
            "v_readlane_b32 %[ec], %[ecb], %[idx]\n\t"
            "s_lshr_b64 %[mm], %[mm], %[z]\n\t"
            "s_bitset1_b64 %[cm], %[oo]\n\t"
            "s_bitset1_b64 %[im], %[idx]\n\t"
            "s_add_u32 %[idx], %[idx], 1\n\t"
            "s_bfe_u64 s[98:99], s[96:97], %[ec]\n\t"
            "s_lshr_b64 %[mm], %[mm], s98\n\t"
            "s_add_u32 %[oo], %[oo], s98\n\t"
            "s_cbranch_scc0 1b\n\t"
            "3:\n\t"
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "+s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_)
            : [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 11) {   // variant 2 + a test for a length of 0 (an entry whose class is in neither group)
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_;
        uint64_t mm_;
        oo -= 64u;
        asm volatile(
            HOP_ALIGN "1:\n\t"
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
            "s_cmp_eq_u32 s98, 0\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_add_u32 %[oo], %[oo], s98\n\t"
            "s_cbranch_scc0 1b\n\t"
            "3:\n\t"
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "=&s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_)
            : [m] "s"(m), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
      else if constexpr (which == 12) {   // variant 2 + that test on the entry's class word, off the chain (bit 15 of it set for such a class)
        const uint32_t ecb = (loc << 3) | (8u << 16);
        uint32_t z_, ec_;
        uint64_t mm_;
        oo -= 64u;
        asm volatile(
            HOP_ALIGN "1:\n\t"
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
            "s_bitcmp1_b32 %[ec], 15\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_bfe_u64 s[98:99], s[96:97], %[ec]\n\t"
            "s_add_u32 %[oo], %[oo], s98\n\t"
            "s_cbranch_scc0 1b\n\t"
            "3:\n\t"
            : [oo] "+s"(oo), [idx] "+s"(idx), [cm] "+s"(cm), [im] "+s"(im), [mm] "=&s"(mm_), [z] "=&s"(z_), [ec] "=&s"(ec_)
            : [m] "s"(m), [ecb] "v"(ecb), [rlo] "v"(rlo), [rhi] "v"(rhi)
            : "scc", "s96", "s97", "s98", "s99");
      }
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
        lo |= (uint32_t)(3 + rand() % 11) << (8 * k);   // length + 1
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
  const char* names[] = {"as it is (21 instructions)", "s_bfe_u64 over both row words", "+ biased position, no test of the length",
                         "+ the rest of the word as a mask", "biased, end-of-word test behind the loads", "next significant position from a lane table", "variant 2 unrolled x 4", "variant 2, compare early / branch late", "variant 2 without the zero test", "the word shifted along (4 scalar steps + 1 load per hop)", "the same without a zero test (upper bound)", "variant 2 + test of the length for 0", "variant 2 + test of the class word"};
#define RUN(w)                                                                                                        \
  {                                                                                                                   \
    uint64_t h[4];                                                                                                    \
    for (int rep = 0; rep < 2; rep++) {                                                                               \
      hipLaunchKernelGGL(k_hops<w>, dim3(1), dim3(64), 0, 0, d, dw, dr, nwords);                                      \
      (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);                                                               \
    }                                                                                                                 \
    printf("%-58s %8.1f cycles per hop  (%llu hops, %.1f per word, check %llu)\n", names[w], (double)h[0] / (double)h[1], \
           (unsigned long long)h[1], (double)h[1] / (8.0 * nwords), (unsigned long long)h[2]);                        \
  }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
  return 0;
}
