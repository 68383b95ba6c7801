// What a lone wavefront pays per instruction on MI355X (the serial walks of k_lis_mx / k_lis_hi are bound by it):
//   hipcc --offload-arch=gfx950 -O2 -o lone_wave tools/micro/lone_wave.cpp && ./lone_wave
// One wavefront, loops of 256 x 16 instructions, cycles from s_memtime (the shader clock on this part).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x

__global__ void k_chain(uint64_t* out, int which)
{
  uint32_t a = threadIdx.x == 1000 ? 7u : 1u, b = 3u, c = 5u, v = threadIdx.x;
  uint64_t t0 = 0, t1 = 0;
  uint32_t s0 = __builtin_amdgcn_readfirstlane(a), s1 = __builtin_amdgcn_readfirstlane(b), s2 = 0, s3 = 0;
  t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; it++) {
    if (which == 0)        // dependent scalar adds
      asm volatile(REP16("s_add_u32 %0, %0, %1\n\t") : "+s"(s0) : "s"(s1) : "scc");
    else if (which == 1)   // independent scalar adds (four chains)
      asm volatile(REP16("s_add_u32 %0, %0, %4\n\ts_add_u32 %1, %1, %4\n\ts_add_u32 %2, %2, %4\n\ts_add_u32 %3, %3, %4\n\t")
                   : "+s"(s0), "+s"(s2), "+s"(s3), "+s"(c) : "s"(s1) : "scc");
    else if (which == 2)   // readlane whose result feeds the next lane select (VALU -> SGPR -> VALU)
      asm volatile(REP16("v_readlane_b32 %0, %1, %0\n\ts_and_b32 %0, %0, 63\n\t") : "+s"(s0) : "v"(v) : "scc");
    else if (which == 3)   // readlane + dependent scalar op, lane select independent
      asm volatile(REP16("v_readlane_b32 %0, %2, %1\n\ts_add_u32 %1, %1, %0\n\ts_and_b32 %1, %1, 63\n\t") : "+s"(s0), "+s"(s2) : "v"(v) : "scc");
    else if (which == 4)   // three readlanes, then a scalar op on each (the hop's pattern)
      asm volatile(REP16("v_readlane_b32 %0, %3, %2\n\tv_readlane_b32 %1, %4, %2\n\tv_readlane_b32 %5, %3, %2\n\ts_add_u32 %2, %0, %1\n\ts_add_u32 %2, %2, %5\n\ts_and_b32 %2, %2, 63\n\t")
                   : "+s"(s0), "+s"(s3), "+s"(s2), "+v"(v), "+v"(a), "+s"(c) : : "scc");
    else if (which == 5)   // taken branches
      asm volatile(REP16("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:\n\t") : : "s"(s1) : "scc");
    else if (which == 6)   // branches not taken
      asm volatile(REP16("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:\n\t") : : "s"(s1) : "scc");
    else if (which == 7)   // dependent 64-bit shift + ff1 (the zero skip)
      asm volatile(REP16("s_lshr_b64 %0, %1, %2\n\ts_ff1_i32_b64 %2, %0\n\ts_and_b32 %2, %2, 31\n\t") : "+s"(t1), "+s"(t0), "+s"(s2) : : "scc");
    else if (which == 8)   // dependent vector adds
      asm volatile(REP16("v_add_u32 %0, %0, %1\n\t") : "+v"(v) : "v"(a));
    else                    // LDS round trip: address depends on the value read
    {
      __shared__ uint32_t lds[64];
      lds[threadIdx.x] = (threadIdx.x * 7 + 1) & 63;
      asm volatile(REP16("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_lshlrev_b32 %0, 2, %0\n\t") : "+v"(v) : : "memory");
    }
  }
  t1 = __builtin_readcyclecounter() - t0;
  if (threadIdx.x == 0) {
    out[0] = t1;
    out[1] = s0 + s2 + s3 + c + v;
  }
}

int main()
{
  uint64_t* d;
  hipMalloc(&d, 16);
  const char* names[] = {"dependent s_add_u32", "four independent s_add_u32 chains (per instruction)", "v_readlane -> lane select of the next (pair)",
                         "v_readlane + 2 dependent scalar ops (triple)", "3 x v_readlane + 3 scalar ops (six)", "compare + taken branch + (skipped nop)",
                         "compare + branch not taken + nop", "s_lshr_b64 + s_ff1_i32_b64 + s_and (triple)", "dependent v_add_u32", "dependent ds_read_b32 + waitcnt + shift"};
  const int per[] = {16, 64, 16, 16, 16, 16, 16, 16, 16, 16};
  for (int w = 0; w < 10; w++) {
    uint64_t h[2];
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, w);
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    }
    printf("%-58s %7.1f cycles per group\n", names[w], (double)h[0] / (256.0 * per[w]));
  }
  return 0;
}
