// H2D / D2H bandwidth of the copy shapes the chunk farm can use (pinned host memory):
//   contiguous, 3D per chunk (1 KB rows), 2D bars (contiguous runs of a chunk row's y extent)
// hipcc -O2 --offload-arch=gfx950 tools/micro/copybw.cpp -o /tmp/copybw && /tmp/copybw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
  const size_t V = 1024, C = 256, esz = 4;
  const size_t volBytes = V * V * V * esz;
  char* h = nullptr;
  CK(hipHostMalloc((void**)&h, volBytes, hipHostMallocPortable));
  memset(h, 1, volBytes);
  char* d = nullptr;
  CK(hipMalloc((void**)&d, volBytes / 4));   // one z-layer of chunks
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const size_t layer = V * V * C * esz;   // 1 GB
  for (int dir = 0; dir < 2; dir++) {
    const hipMemcpyKind kind = dir ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
    for (int rep = 0; rep < 2; rep++) {
      auto t0 = now();
      if (dir) CK(hipMemcpyAsync(h, d, layer, kind, st)); else CK(hipMemcpyAsync(d, h, layer, kind, st));
      CK(hipStreamSynchronize(st));
      auto t1 = now();
      if (rep) printf("%s contiguous 1 GB: %.1f ms %.1f GB/s\n", dir ? "D2H" : "H2D", ms(t0, t1), layer / ms(t0, t1) / 1e6);
    }
    for (int rep = 0; rep < 2; rep++) {   // 16 chunks of the layer, one 3D copy each
      auto t0 = now();
      for (size_t cy = 0; cy < 4; cy++)
        for (size_t cx = 0; cx < 4; cx++) {
          hipMemcpy3DParms p;
          memset(&p, 0, sizeof(p));
          hipPitchedPtr vp = make_hipPitchedPtr(h, V * esz, V * esz, V);
          hipPitchedPtr sp = make_hipPitchedPtr(d + (cy * 4 + cx) * C * C * C * esz, C * esz, C * esz, C);
          hipPos vpos = make_hipPos(cx * C * esz, cy * C, 0), spos = make_hipPos(0, 0, 0);
          if (dir) { p.srcPtr = sp; p.srcPos = spos; p.dstPtr = vp; p.dstPos = vpos; }
          else { p.srcPtr = vp; p.srcPos = vpos; p.dstPtr = sp; p.dstPos = spos; }
          p.extent = make_hipExtent(C * esz, C, C);
          p.kind = kind;
          CK(hipMemcpy3DAsync(&p, st));
        }
      CK(hipStreamSynchronize(st));
      auto t1 = now();
      if (rep) printf("%s 16 x 3D chunk copies (1 KB rows): %.1f ms %.1f GB/s\n", dir ? "D2H" : "H2D", ms(t0, t1), layer / ms(t0, t1) / 1e6);
    }
    for (int rep = 0; rep < 2; rep++) {   // bars: y range of one chunk row, all x: runs of V*C*esz = 1 MB per z
      auto t0 = now();
      for (size_t cy = 0; cy < 4; cy++) {
        if (dir) CK(hipMemcpy2DAsync(h + cy * C * V * esz, V * V * esz, d + cy * C * V * C * esz, V * C * esz, V * C * esz, C, kind, st));
        else CK(hipMemcpy2DAsync(d + cy * C * V * C * esz, V * C * esz, h + cy * C * V * esz, V * V * esz, V * C * esz, C, kind, st));
      }
      CK(hipStreamSynchronize(st));
      auto t1 = now();
      if (rep) printf("%s 4 x 2D bar copies (1 MB runs): %.1f ms %.1f GB/s\n", dir ? "D2H" : "H2D", ms(t0, t1), layer / ms(t0, t1) / 1e6);
    }
  }
  // two streams at once, contiguous halves
  hipStream_t st2;
  CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  for (int rep = 0; rep < 2; rep++) {
    auto t0 = now();
    CK(hipMemcpyAsync(d, h, layer / 2, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(h + layer, d + layer / 2, layer / 2, hipMemcpyDeviceToHost, st2));
    CK(hipStreamSynchronize(st));
    CK(hipStreamSynchronize(st2));
    auto t1 = now();
    if (rep) printf("H2D 0.5 GB + D2H 0.5 GB on two streams: %.1f ms %.1f GB/s total\n", ms(t0, t1), layer / ms(t0, t1) / 1e6);
  }
  return 0;
}
