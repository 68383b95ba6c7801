// farm.hip -- the chunk farm: host volumes of any size through the per-chunk pipeline, on every
// device of the node, with the transfers of one batch of chunks hidden behind the kernels of others.
//
// What it replaces in the reference (file:line under /root/reference):
//   src/SPERR3D_OMP_C.cpp:61-141    the OpenMP chunk loop of compress(): gather a chunk (:236-261),
//                                   run the per-chunk pipeline, keep its stream
//   src/SPERR3D_OMP_C.cpp:145-234   header + concatenated chunk streams
//   src/SPERR3D_OMP_D.cpp:50-150    the OpenMP chunk loop of decompress(): per-chunk pipeline,
//                                   scatter into the volume (:167-184)
//   src/SPERR_C_API.cpp:135-258     sperr_comp_3d / sperr_decomp_3d (ownership, return codes)
//
// Design.  Chunks share nothing, so the volume is cut into WORK ITEMS: runs of equally shaped chunks
// in chunk_volume order (src/sperr_helper.cpp:542-592), a few hundred MB each.  A shared atomic
// counter hands the items to WORKERS, a few host threads per device (a device list with one entry
// per device; an entry may repeat).  A worker owns a HIP stream, pinned staging buffers and device
// buffers, and handles one item at a time:
//     compress:    gather the item's chunk rows into pinned memory (helper threads; or DMA them
//                  straight out of the caller's buffer when that is pinned) -> H2D -> the
//                  device-resident compressor on a volume made of the item's chunks stacked along z
//                  (sperrhip_compress_dev, which leases its own engine) -> D2H of the small
//                  container -> chunk streams to their place in the output
//     decompress:  the item's chunk streams as a small container -> H2D -> sperrhip_decompress_dev
//                  -> D2H -> rows scattered into the caller's volume
// A worker has two sets of buffers and the device two copy streams (one per direction) shared by its
// workers: item i + 1's input travels in and item i - 1's output travels out while item i computes
// (events between the streams; the host waits only when it needs an item's bytes).  Balance between
// devices comes from the queue.  No collective, no peer traffic.  The
// volume is never resident on a device as a whole, so its size is bounded by host memory only
// (BASELINE.json config 5).
#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <future>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/sperr_hip.h"
#include "common.h"
#include "engine_internal.h"
#include "numa_place.hpp"
#include "host_cpus.hpp"

namespace sperrhip {
namespace {

size_t env_size(const char* name, size_t dflt)
{
  const char* v = getenv(name);
  if (!v || !*v)
    return dflt;
  const long long x = atoll(v);
  return x > 0 ? (size_t)x : dflt;
}

// ------------------------------------------------------------------------------------------
// work items and the queue
// ------------------------------------------------------------------------------------------
struct Item {
  Dims3 shape;                               // dims of every chunk of the item
  std::vector<uint32_t> gid;                 // index in chunk_volume order
  std::vector<std::array<size_t, 3>> org;    // origin inside the volume
};

struct FarmShape {
  size_t workersPerDevice, helpers, itemBytesMax, itemChunksForced;
};

FarmShape farm_shape(size_t nthreads, size_t ndev)
{
  FarmShape f;
  // (two double-buffered workers per device since round 4; rounds 1-3: three synchronous ones)
  f.workersPerDevice = env_size("SPERR_HIP_FARM_WORKERS", 2);
  f.itemBytesMax = env_size("SPERR_HIP_FARM_ITEM_MB", 768) << 20;
  f.itemChunksForced = env_size("SPERR_HIP_FARM_ITEM", 0);
  // `nthreads` (the reference's OpenMP team size, src/SPERR3D_OMP_C.cpp:12-20) is taken as the
  // number of host threads that move rows between the caller's buffers and the staging buffers.
  // Default: half of the CPUs THIS PROCESS MAY USE shared out to the workers, at most 12 each
  // (measured on a 256-thread host with one MI355X, pageable 1024^3 volume: 4 -> 8 helpers per worker
  // gains 8 %).  "May use" is the affinity mask and the cgroup's CFS quota (host_cpus.hpp), not
  // hardware_concurrency(): the pool's box shows 256 CPUs and grants 16, and worker + helper threads
  // beyond the quota are throttled, not run -- eight devices x two workers on 16 CPUs get one helper each.
  const size_t hw = hostcpu::probe().usable;
  const size_t nworkers = std::max<size_t>(1, ndev * f.workersPerDevice);
  size_t helpers = std::min<size_t>(12, std::max<size_t>(1, hw / (2 * nworkers)));
  if (hw >= 8 * nworkers)   // (room for it: never fewer than four, as rounds 1-4 had it)
    helpers = std::max<size_t>(helpers, 4);
  if (nthreads > 0)
    helpers = std::max<size_t>(1, nthreads / nworkers);
  f.helpers = env_size("SPERR_HIP_FARM_HELPERS", helpers);
  // (the plan must fit what the process may run: on a 16-CPU quota eight devices x two workers x (1 + 1 helper) are 32
  //  runnable threads -- the workers themselves mostly wait for their device, so they stay, but it is said once)
  if (ndev * f.workersPerDevice * (1 + f.helpers) > 2 * std::max<size_t>(1, hw)) {
    static std::atomic<bool> said{false};
    if (!said.exchange(true))
      fprintf(stderr, "[sperr_hip] farm: %zu devices x %zu workers x (1 + %zu helpers) threads on %zu usable CPUs\n", ndev,
              f.workersPerDevice, f.helpers, hw);
  }
  return f;
}

// Equally shaped chunks, in chunk_volume order, `perItem` at a time.  An item is sized so that
// every worker sees a few of them (balance) but none is tiny (the per-plane kernels of the coder
// have a fixed cost per launch) or larger than the staging buffers should be.
// `perWorker`: items every worker should see.  Measured on one MI355X, 1024^3 fp32 in 256^3 chunks,
// 3 workers, pinned volume (round 3, `tools/farm_tune.py`): compression 90.5 / 87.2 / 86.8 / 87.1 /
// 91.0 / 95.1 ms for items of 3 / 4 / 5 / 6 / 8 / 11 chunks (four items per worker), decompression
// 126 / 114 / 110 / 107 / 108 ms for 4 / 5 / 6 / 8 / 11 (two: its list kernels want larger batches).
std::vector<Item> make_items(const std::vector<std::array<size_t, 6>>& chunks, size_t bytesPerValue,
                             size_t nworkers, const FarmShape& fs, size_t perWorker = 2)
{
  std::map<Dims3, std::vector<uint32_t>> groups;
  for (uint32_t i = 0; i < chunks.size(); i++)
    groups[Dims3{chunks[i][1], chunks[i][3], chunks[i][5]}].push_back(i);
  // the largest shapes first: they take longest
  std::vector<const std::pair<const Dims3, std::vector<uint32_t>>*> order;
  for (auto& g : groups)
    order.push_back(&g);
  std::stable_sort(order.begin(), order.end(), [](auto* a, auto* b) {
    return a->first[0] * a->first[1] * a->first[2] > b->first[0] * b->first[1] * b->first[2];
  });
  std::vector<Item> items;
  for (auto* g : order) {
    const Dims3& sh = g->first;
    const size_t chunkBytes = sh[0] * sh[1] * sh[2] * bytesPerValue;
    const size_t n = g->second.size();
    const size_t kMax = std::max<size_t>(1, fs.itemBytesMax / std::max<size_t>(1, chunkBytes));
    const size_t kMin = std::max<size_t>(1, (size_t(32) << 20) / std::max<size_t>(1, chunkBytes));
    size_t k = (n + perWorker * nworkers - 1) / (perWorker * nworkers);
    // (but not below four chunks while every worker still gets an item: a batch of one or two chunks
    //  is all launch latency)
    k = std::max(k, std::min<size_t>(4, (n + nworkers - 1) / nworkers));
    k = std::min(std::max(k, kMin), kMax);
    k = std::min<size_t>(k, 256);   // (a batch of the engine holds at most 256 chunks)
    if (fs.itemChunksForced)
      k = std::min<size_t>(fs.itemChunksForced, 256);
    for (size_t b0 = 0; b0 < n; b0 += k) {
      Item it;
      it.shape = sh;
      for (size_t i = b0; i < std::min(n, b0 + k); i++) {
        const uint32_t gid = g->second[i];
        it.gid.push_back(gid);
        it.org.push_back({chunks[gid][0], chunks[gid][2], chunks[gid][4]});
      }
      items.push_back(std::move(it));
    }
  }
  return items;
}

// SPERR_HIP_DEVICES: "all" (default) or a comma separated list of device ordinals; an ordinal may
// repeat (more workers on that device).  `explicitList` overrides it.
int device_list(const int* explicitList, size_t nExplicit, std::vector<int>& devs)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    fprintf(stderr, "[sperr_hip] no HIP device available; this library has no CPU fallback\n");
    return -1;
  }
  devs.clear();
  if (explicitList && nExplicit) {
    for (size_t i = 0; i < nExplicit; i++)
      devs.push_back(explicitList[i]);
  }
  else {
    const char* v = getenv("SPERR_HIP_DEVICES");
    if (v && *v && strcmp(v, "all") != 0) {
      std::string s(v);
      size_t pos = 0;
      while (pos < s.size()) {
        size_t e = s.find(',', pos);
        if (e == std::string::npos)
          e = s.size();
        if (e > pos)
          devs.push_back(atoi(s.substr(pos, e - pos).c_str()));
        pos = e + 1;
      }
    }
    if (devs.empty())
      for (int d = 0; d < ndev; d++)
        devs.push_back(d);
  }
  for (int d : devs)
    if (d < 0 || d >= ndev) {
      fprintf(stderr, "[sperr_hip] device %d is not one of the %d visible devices\n", d, ndev);
      return -1;
    }
  return 0;
}

// run fn(t) for t in [0, nthreads) on that many threads (the caller is one of them)
template <typename F>
void parallel_do(size_t nthreads, F&& fn)
{
  if (nthreads <= 1) {
    fn(size_t(0));
    return;
  }
  // a thread that cannot be started (std::system_error) must not leave started ones joinable:
  // its share is done by the caller instead
  std::vector<std::thread> th;
  th.reserve(nthreads - 1);
  std::vector<size_t> mine = {0};
  for (size_t t = 1; t < nthreads; t++) {
    try {
      th.emplace_back([&fn, t]() { fn(t); });
    }
    catch (const std::system_error&) {
      mine.push_back(t);
    }
  }
  for (size_t t : mine)
    fn(t);
  for (auto& t : th)
    t.join();
}

// 0: pageable host memory, 1: pinned host memory, 2: device memory (then *dev is its device)
int classify_ptr(const void* p, int* dev)
{
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof(a));
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();   // plain malloc'd memory is reported as an invalid value
    return 0;
  }
  if (a.type == hipMemoryTypeDevice) {
    if (dev)
      *dev = a.device;
    return 2;
  }
  return a.type == hipMemoryTypeHost ? 1 : 0;
}
bool host_ptr_is_pinned(const void* p)
{
  return classify_ptr(p, nullptr) == 1;
}

// keeps the calling thread's device across a detour to another one
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev)
  {
    if (hipGetDevice(&prev) != hipSuccess)
      prev = -1;
    (void)hipSetDevice(dev);
  }
  ~DeviceGuard()
  {
    if (prev >= 0)
      (void)hipSetDevice(prev);
  }
};

// ------------------------------------------------------------------------------------------
// worker contexts: stream + two sets of staging buffers, kept between calls (pinning memory is slow)
// ------------------------------------------------------------------------------------------
// A worker has TWO slots (round 4): while item i computes out of one, item i + 1's input is on its way into
// the other and item i - 1's output is on its way out of it.
struct Slot {
  void *pinIn = nullptr, *pinOut = nullptr, *dIn = nullptr, *dOut = nullptr;
  size_t pinInCap = 0, pinOutCap = 0, dInCap = 0, dOutCap = 0;
  hipEvent_t evIn = nullptr, evComp = nullptr, evOut = nullptr;   // input on the device / computed / output on the host
  bool usedIn = false, usedComp = false;                          // the events have been recorded at least once
  // a slot holds TWO items at a time: the one whose input is in (or on its way into) dIn, and the one before
  // the last whose output is on its way out of dOut -- so each side has its own record
  const Item* itIn = nullptr;    // set by prefetch, read by compute
  size_t lenIn = 0;              //   decompress: bytes of the small container sent in
  const Item* itOut = nullptr;   // set by compute, read by finish
  size_t lenOut = 0;             //   compress: bytes of the item's container

  static int grow_pinned(void*& p, size_t& cap, size_t bytes)
  {
    if (bytes <= cap)
      return 0;
    if (p)
      (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocPortable));
    cap = want;
    return 0;
  }
  static int grow_dev(void*& p, size_t& cap, size_t bytes)
  {
    if (bytes <= cap)
      return 0;
    if (p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + 4096;
    HIP_CHECK(hipMalloc(&p, want));
    cap = want;
    return 0;
  }
  int need_pin_in(size_t b) { return grow_pinned(pinIn, pinInCap, b); }
  int need_pin_out(size_t b) { return grow_pinned(pinOut, pinOutCap, b); }
  int need_dev_in(size_t b) { return grow_dev(dIn, dInCap, b); }
  int need_dev_out(size_t b) { return grow_dev(dOut, dOutCap, b); }
  void drop_pinned()
  {
    if (pinIn)
      (void)hipHostFree(pinIn);
    if (pinOut)
      (void)hipHostFree(pinOut);
    pinIn = pinOut = nullptr;
    pinInCap = pinOutCap = 0;
  }
  void drop_dev()
  {
    if (dIn)
      (void)hipFree(dIn);
    if (dOut)
      (void)hipFree(dOut);
    dIn = dOut = nullptr;
    dInCap = dOutCap = 0;
  }
};

struct WorkerCtx {
  int dev = -1;
  bool busy = false;
  hipStream_t st = nullptr;   // the device-resident calls of this worker
  Slot slot[2];
  size_t pinned_bytes() const { return slot[0].pinInCap + slot[0].pinOutCap + slot[1].pinInCap + slot[1].pinOutCap; }
  size_t device_bytes() const { return slot[0].dInCap + slot[0].dOutCap + slot[1].dInCap + slot[1].dOutCap; }
};

std::mutex g_ctx_mu;
std::vector<std::unique_ptr<WorkerCtx>> g_ctx;

// One copy stream per device and direction, shared by the device's workers: the large copies of the
// items go one after the other in the order they were asked for (PCIe is shared: copies side by side all
// finish late), H2D and D2H on separate streams because the link is full duplex (57 GB/s each way, 97
// GB/s together, tools/micro/copybw.cpp).  Rounds 1-3 had a mutex per direction around synchronous
// copies; now nobody waits on the host.
struct CopyLanes {
  hipStream_t h2d = nullptr, d2h = nullptr;
};
// (the calling thread has made `dev` current)
CopyLanes* copy_lanes(int dev)
{
  static std::mutex mu;
  static std::map<int, std::unique_ptr<CopyLanes>> all;
  std::lock_guard<std::mutex> lock(mu);
  auto& p = all[dev];
  if (!p) {
    auto l = std::make_unique<CopyLanes>();
    if (hipStreamCreateWithFlags(&l->h2d, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&l->d2h, hipStreamNonBlocking) != hipSuccess) {
      fprintf(stderr, "[sperr_hip] cannot create the copy streams of device %d\n", dev);
      return nullptr;
    }
    p = std::move(l);
  }
  return p.get();
}

// (the calling thread has made `dev` current)
WorkerCtx* ctx_acquire(int dev)
{
  {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    for (auto& c : g_ctx)
      if (c->dev == dev && !c->busy) {
        c->busy = true;
        return c.get();
      }
  }
  auto c = std::make_unique<WorkerCtx>();
  c->dev = dev;
  c->busy = true;
  bool ok = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) == hipSuccess;
  for (Slot& sl : c->slot)
    for (hipEvent_t* e : {&sl.evIn, &sl.evComp, &sl.evOut})
      ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    fprintf(stderr, "[sperr_hip] cannot create a stream on device %d\n", dev);
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_ctx_mu);
  g_ctx.push_back(std::move(c));
  return g_ctx.back().get();
}

}  // namespace (reopened below)
// sperrhip_release(): the buffers of every idle worker go back (the worker and its stream stay)
void farm_release_idle()
{
  std::lock_guard<std::mutex> lock(g_ctx_mu);
  for (auto& c : g_ctx) {
    if (c->busy)
      continue;
    if (hipSetDevice(c->dev) != hipSuccess)
      continue;
    for (Slot& sl : c->slot) {
      sl.drop_pinned();
      sl.drop_dev();
    }
  }
}
unsigned long long farm_footprint(bool pinned)
{
  std::lock_guard<std::mutex> lock(g_ctx_mu);
  unsigned long long n = 0;
  for (auto& c : g_ctx)
    n += pinned ? c->pinned_bytes() : c->device_bytes();
  return n;
}
namespace {

void ctx_release(WorkerCtx* c)
{
  // staging memory above SPERR_HIP_FARM_KEEP_MB (default 4096) per worker is given back
  static const size_t keep = env_size("SPERR_HIP_FARM_KEEP_MB", 4096) << 20;
  if (c->pinned_bytes() > keep)
    for (Slot& sl : c->slot)
      sl.drop_pinned();
  if (c->device_bytes() > 2 * keep)
    for (Slot& sl : c->slot)
      sl.drop_dev();
  std::lock_guard<std::mutex> lock(g_ctx_mu);
  c->busy = false;
}

// ------------------------------------------------------------------------------------------
// rows between a volume in host memory and a buffer of stacked chunks
// ------------------------------------------------------------------------------------------
// toStack: volume -> stack (compress); else stack -> volume (decompress).  Planes [p0, p1) of the
// item's nb * cz chunk planes.
void move_planes(bool toStack, uint8_t* volume, const Dims3& vol, uint8_t* stack, const Item& it,
                 size_t esz, size_t p0, size_t p1)
{
  const size_t cx = it.shape[0], cy = it.shape[1], cz = it.shape[2];
  const size_t rowBytes = cx * esz;
  for (size_t p = p0; p < p1; p++) {
    const size_t c = p / cz, z = p % cz;
    const auto& o = it.org[c];
    uint8_t* v = volume + (((o[2] + z) * vol[1] + o[1]) * vol[0] + o[0]) * esz;
    uint8_t* s = stack + p * cy * rowBytes;
    const size_t vpitch = vol[0] * esz;
    if (cx == vol[0]) {   // rows are adjacent in the volume as well
      if (toStack)
        memcpy(s, v, cy * rowBytes);
      else
        memcpy(v, s, cy * rowBytes);
      continue;
    }
    for (size_t y = 0; y < cy; y++) {
      if (toStack)
        memcpy(s + y * rowBytes, v + y * vpitch, rowBytes);
      else
        memcpy(v + y * vpitch, s + y * rowBytes, rowBytes);
    }
  }
}

void move_item(bool toStack, uint8_t* volume, const Dims3& vol, uint8_t* stack, const Item& it,
               size_t esz, size_t helpers)
{
  const size_t planes = it.gid.size() * it.shape[2];
  const size_t nt = std::max<size_t>(1, std::min(helpers, planes));
  parallel_do(nt, [&](size_t t) {
    move_planes(toStack, volume, vol, stack, it, esz, planes * t / nt, planes * (t + 1) / nt);
  });
}

// the same by DMA, when the caller's volume is pinned memory: one 3D copy per chunk
int dma_item(bool toStack, uint8_t* volume, const Dims3& vol, uint8_t* d_stack, const Item& it,
             size_t esz, hipStream_t st)
{
  const size_t cx = it.shape[0], cy = it.shape[1], cz = it.shape[2];
  for (size_t c = 0; c < it.gid.size(); c++) {
    hipMemcpy3DParms p;
    memset(&p, 0, sizeof(p));
    const hipPitchedPtr vptr = make_hipPitchedPtr(volume, vol[0] * esz, vol[0] * esz, vol[1]);
    const hipPitchedPtr sptr = make_hipPitchedPtr(d_stack + c * cx * cy * cz * esz, cx * esz, cx * esz, cy);
    const hipPos vpos = make_hipPos(it.org[c][0] * esz, it.org[c][1], it.org[c][2]);
    const hipPos spos = make_hipPos(0, 0, 0);
    if (toStack) {
      p.srcPtr = vptr;
      p.srcPos = vpos;
      p.dstPtr = sptr;
      p.dstPos = spos;
      p.kind = hipMemcpyHostToDevice;
    }
    else {
      p.srcPtr = sptr;
      p.srcPos = spos;
      p.dstPtr = vptr;
      p.dstPos = vpos;
      p.kind = hipMemcpyDeviceToHost;
    }
    p.extent = make_hipExtent(cx * esz, cy, cz);
    HIP_CHECK(hipMemcpy3DAsync(&p, st));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// jobs
// ------------------------------------------------------------------------------------------
struct Job {
  // common
  Dims3 vol, cdim;
  std::vector<Item> items;
  std::vector<int> workerDev;
  FarmShape fs;
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};
  bool direct = false;   // the caller's volume is pinned: DMA instead of staging
  // prefetch and finish of an item on helper threads beside the worker's device call (run_workers).  Not for
  // compression without a bit budget: measured (1024^3, tolerance 1e-3 of the range, three workers) 154 ms
  // with the helpers against 106 ms with every step on the worker's own thread -- its device call goes back
  // to the host a dozen times per batch, and copies issued beside it get in the way of those round trips
  bool helperThreads = true;
  // compress
  const uint8_t* src = nullptr;
  int is_float = 1, mode = 1;
  double quality = 0.0;
  size_t esz = 4;
  uint8_t* outBuf = nullptr;                 // rate mode: chunk streams land at their upper-bound offsets
  std::vector<uint64_t> slotOff;             // (rate mode)
  std::vector<std::unique_ptr<uint8_t[]>> chunkBytes;   // (other modes)
  std::vector<uint64_t> chunkLen;
  // decompress
  const uint8_t* container = nullptr;
  const HostContainer* hc = nullptr;
  int output_float = 1;
  uint8_t* dstVol = nullptr;
  Job() = default;
  Job(const Job&) = delete;
  Job& operator=(const Job&) = delete;
  ~Job() { free(outBuf); }   // (whatever ends the call early; handed over = set to nullptr)
};

// ---- the three stages of an item (round 4) -----------------------------------------------------------
//   prefetch  input -> device, on the device's H2D lane       (records slot.evIn)
//   compute   the device-resident call on the worker's stream  (waits for evIn; records evComp)
//   finish    the result -> host on the D2H lane (waits for evComp), then what is left to do on the host
// Item i's compute runs on the worker's thread while item i + 1's prefetch and item i - 1's finish run on
// two helper threads of their own: a copy call that keeps its caller until the bytes have moved (the 3D
// copies of a pinned volume do; measured: one worker with the copies issued from its own thread ran copy
// and kernels strictly one after the other) then still overlaps the kernels.
// compress ----------------------------------------------------------------------------------------
int comp_prefetch(Job& J, WorkerCtx&, CopyLanes& L, Slot& S, const Item& it)
{
  const size_t nb = it.gid.size(), cx = it.shape[0], cy = it.shape[1], cz = it.shape[2];
  const size_t inBytes = nb * cx * cy * cz * J.esz;
  S.itIn = &it;
  if (S.usedIn)
    HIP_CHECK(hipEventSynchronize(S.evIn));     // (the slot's staging buffer: its last copy left long ago)
  if (S.need_dev_in(inBytes))
    return -1;
  if (S.usedComp)
    HIP_CHECK(hipStreamWaitEvent(L.h2d, S.evComp, 0));   // the slot's last item has been computed
  if (J.direct) {
    if (dma_item(true, const_cast<uint8_t*>(J.src), J.vol, static_cast<uint8_t*>(S.dIn), it, J.esz, L.h2d))
      return -1;
  }
  else {
    if (S.need_pin_in(inBytes))
      return -1;
    move_item(true, const_cast<uint8_t*>(J.src), J.vol, static_cast<uint8_t*>(S.pinIn), it, J.esz, J.fs.helpers);
    HIP_CHECK(hipMemcpyAsync(S.dIn, S.pinIn, inBytes, hipMemcpyHostToDevice, L.h2d));
  }
  HIP_CHECK(hipEventRecord(S.evIn, L.h2d));
  S.usedIn = true;
  return 0;
}

int comp_compute(Job& J, WorkerCtx& C, CopyLanes& L, Slot& S)
{
  const Item& it = *S.itIn;
  const size_t nb = it.gid.size(), cx = it.shape[0], cy = it.shape[1], cz = it.shape[2];
  const size_t inBytes = nb * cx * cy * cz * J.esz;
  HIP_CHECK(hipStreamWaitEvent(C.st, S.evIn, 0));
  // the item's chunks stacked along z are a volume of their own, cut into exactly these chunks.
  // Without a bit budget the bound on the container is 33 bytes per value; real containers are a
  // fraction of the input, so the first attempt gets as much room as the input takes and only a
  // container that does not fit (never seen) is produced again into the full bound.
  const size_t bound = sperrhip_max_compressed_size(cx, cy, cz * nb, cx, cy, cz, J.mode, J.quality);
  size_t cap = std::min(bound, inBytes + (size_t(1) << 20));
  size_t len = 0;
  for (;;) {
    if (S.need_dev_out(cap))
      return -1;
    const int rc = sperrhip_compress_dev(S.dIn, J.is_float, cx, cy, cz * nb, cx, cy, cz, J.mode,
                                         J.quality, S.dOut, cap, &len, C.st);
    if (rc == 0)
      break;
    if (rc != -1 || cap >= bound)
      return rc;
    cap = bound;
  }
  const size_t hdr = nb > 1 ? 20 + 4 * nb : 18;
  if (len < hdr || S.need_pin_out(len))
    return -1;
  S.itOut = S.itIn;
  S.lenOut = len;
  HIP_CHECK(hipEventRecord(S.evComp, C.st));
  S.usedComp = true;
  return 0;
}

int comp_finish(Job& J, WorkerCtx&, CopyLanes& L, Slot& S)
{
  HIP_CHECK(hipStreamWaitEvent(L.d2h, S.evComp, 0));
  HIP_CHECK(hipMemcpyAsync(S.pinOut, S.dOut, S.lenOut, hipMemcpyDeviceToHost, L.d2h));
  HIP_CHECK(hipEventRecord(S.evOut, L.d2h));
  HIP_CHECK(hipEventSynchronize(S.evOut));
  const Item& it = *S.itOut;
  const size_t nb = it.gid.size(), len = S.lenOut;
  const size_t hdr = nb > 1 ? 20 + 4 * nb : 18;
  const uint8_t* h = static_cast<const uint8_t*>(S.pinOut);
  size_t at = hdr;
  for (size_t i = 0; i < nb; i++) {
    uint32_t l;
    memcpy(&l, h + (hdr - 4 * nb) + 4 * i, 4);
    if (at + l > len)
      return -1;
    const uint32_t gid = it.gid[i];
    J.chunkLen[gid] = l;
    if (J.outBuf) {
      if (l > J.slotOff[gid + 1] - J.slotOff[gid])
        return -1;
      memcpy(J.outBuf + J.slotOff[gid], h + at, l);
    }
    else {
      J.chunkBytes[gid].reset(new uint8_t[std::max<size_t>(l, 1)]);
      memcpy(J.chunkBytes[gid].get(), h + at, l);
    }
    at += l;
  }
  return 0;
}

// decompress --------------------------------------------------------------------------------------
int decomp_prefetch(Job& J, WorkerCtx&, CopyLanes& L, Slot& S, const Item& it)
{
  const HostContainer& hc = *J.hc;
  const size_t nb = it.gid.size(), cx = it.shape[0], cy = it.shape[1], cz = it.shape[2];
  const size_t hdr = nb > 1 ? 20 + 4 * nb : 18;
  size_t total = hdr;
  for (uint32_t g : it.gid)
    total += hc.len[g];
  S.itIn = &it;
  S.lenIn = total;
  if (S.usedIn)
    HIP_CHECK(hipEventSynchronize(S.evIn));     // (the staging buffer is rewritten below)
  if (S.need_pin_in(total) || S.need_dev_in(total))
    return -1;
  uint8_t* h = static_cast<uint8_t*>(S.pinIn);
  h[0] = 0;   // SPERR_VERSION_MAJOR
  h[1] = (uint8_t)(0x40 | (hc.is_float ? 0x20 : 0) | (nb > 1 ? 0x10 : 0) | (hc.portion ? 0x80 : 0));
  const uint32_t v3[3] = {(uint32_t)cx, (uint32_t)cy, (uint32_t)(cz * nb)};
  memcpy(h + 2, v3, 12);
  if (nb > 1) {
    const uint16_t c3[3] = {(uint16_t)cx, (uint16_t)cy, (uint16_t)cz};
    memcpy(h + 14, c3, 6);
  }
  size_t at = hdr;
  for (size_t i = 0; i < nb; i++) {
    const uint32_t g = it.gid[i];
    const uint32_t l = (uint32_t)hc.len[g];
    memcpy(h + (hdr - 4 * nb) + 4 * i, &l, 4);
    memcpy(h + at, J.container + hc.off[g], l);
    at += l;
  }
  if (S.usedComp)
    HIP_CHECK(hipStreamWaitEvent(L.h2d, S.evComp, 0));
  HIP_CHECK(hipMemcpyAsync(S.dIn, S.pinIn, total, hipMemcpyHostToDevice, L.h2d));
  HIP_CHECK(hipEventRecord(S.evIn, L.h2d));
  S.usedIn = true;
  return 0;
}

int decomp_compute(Job& J, WorkerCtx& C, CopyLanes& L, Slot& S)
{
  const Item& it = *S.itIn;
  S.itOut = S.itIn;
  const size_t nb = it.gid.size(), cx = it.shape[0], cy = it.shape[1], cz = it.shape[2];
  const size_t osz = J.output_float ? 4 : 8;
  const size_t outBytes = nb * cx * cy * cz * osz;
  if (S.need_dev_out(outBytes))
    return -1;
  HIP_CHECK(hipStreamWaitEvent(C.st, S.evIn, 0));
  size_t x = 0, y = 0, z = 0;
  t_shared_device = J.workerDev.size() > 1;   // (then other workers' calls run beside this one)
  const int rc = sperrhip_decompress_dev(S.dIn, S.lenIn, J.output_float, S.dOut, outBytes, &x, &y, &z, C.st);
  t_shared_device = false;
  if (rc)
    return rc;
  HIP_CHECK(hipEventRecord(S.evComp, C.st));
  S.usedComp = true;
  if (!J.direct && S.need_pin_out(outBytes))
    return -1;
  return 0;
}

int decomp_finish(Job& J, WorkerCtx&, CopyLanes& L, Slot& S)
{
  const Item& it = *S.itOut;
  const size_t osz = J.output_float ? 4 : 8;
  const size_t outBytes = it.gid.size() * it.shape[0] * it.shape[1] * it.shape[2] * osz;
  HIP_CHECK(hipStreamWaitEvent(L.d2h, S.evComp, 0));
  if (J.direct) {
    if (dma_item(false, J.dstVol, J.vol, static_cast<uint8_t*>(S.dOut), it, osz, L.d2h))
      return -1;
  }
  else
    HIP_CHECK(hipMemcpyAsync(S.pinOut, S.dOut, outBytes, hipMemcpyDeviceToHost, L.d2h));
  HIP_CHECK(hipEventRecord(S.evOut, L.d2h));
  HIP_CHECK(hipEventSynchronize(S.evOut));
  if (!J.direct)
    move_item(false, J.dstVol, J.vol, static_cast<uint8_t*>(S.pinOut), it, osz, J.fs.helpers);
  return 0;
}

// NUMA node and CPUs of a device (numa_place.hpp), looked up once per device
const numa::Place& device_place(int dev)
{
  static std::mutex mu;
  static std::map<int, numa::Place> all;
  std::lock_guard<std::mutex> lock(mu);
  auto it = all.find(dev);
  if (it != all.end())
    return it->second;
  numa::Place pl;
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), dev) == hipSuccess)
    pl = numa::probe(numa::sysfs_root(), bdf);
  else
    (void)hipGetLastError();
  return all.emplace(dev, std::move(pl)).first->second;
}

template <typename P, typename Cm, typename Fi>
int run_workers(Job& J, P&& prefetch, Cm&& compute, Fi&& finish)
{
  const size_t nw = J.workerDev.size();
  std::vector<std::thread> th;
  auto body = [&](size_t w) {
    const int dev = J.workerDev[w];
    if (hipSetDevice(dev) != hipSuccess) {
      J.failed = -1;
      return;
    }
    // this thread, the helper threads it starts (they inherit the mask) and the staging memory it
    // pins from now on stay on the socket of its device (SPERR_HIP_FARM_NUMA=0: wherever the OS puts them)
    if (numa::enabled())
      (void)numa::bind_self(device_place(dev));
    WorkerCtx* C = ctx_acquire(dev);
    CopyLanes* L = C ? copy_lanes(dev) : nullptr;
    if (!C || !L) {
      if (C)
        ctx_release(C);
      J.failed = -1;
      return;
    }
    auto take = [&]() -> const Item* {
      if (J.failed.load())
        return nullptr;
      const size_t i = J.next.fetch_add(1);
      return i < J.items.size() ? &J.items[i] : nullptr;
    };
    // (SPERR_HIP_FARM_ASYNC is read at every call: the tuning sweeps change it between calls)
    const int helpersEnv = getenv("SPERR_HIP_FARM_ASYNC") ? atoi(getenv("SPERR_HIP_FARM_ASYNC")) : -1;
    const bool helpersOn = helpersEnv < 0 ? J.helperThreads : helpersEnv != 0;
    std::future<int> fIn, fOut[2];   // item i + 1's prefetch; the slots' finishes
    // (a helper thread: the worker's device, its NUMA placement by inheritance; nothing escapes it)
    auto helper = [&](auto&& fn) -> std::future<int> {
      if (!helpersOn) {
        std::promise<int> pr;
        pr.set_value(fn());
        return pr.get_future();
      }
      try {
        return std::async(std::launch::async, [dev, fn]() -> int {
          try {
            if (hipSetDevice(dev) != hipSuccess)
              return -1;
            return fn();
          }
          catch (...) {
            return -1;
          }
        });
      }
      catch (const std::system_error&) {
        // no thread to be had (a process at its limit): the stage runs here, on the worker's own thread, like with
        // helpersOn == false -- slower, not a failed call (parallel_do and run_workers degrade the same way)
        std::promise<int> pr;
        pr.set_value(fn());
        return pr.get_future();
      }
    };
    int rc = 0;
    try {
      int cur = 0;
      const Item* it = take();
      if (!helpersOn) {
        // every step on this thread, one item after the other (rounds 1-3): no item's input is asked for
        // before the one in hand has been computed -- with a few large items per worker (compression without
        // a bit budget) two inputs up front were 49 ms of copies before the first kernel ran
        for (; it && rc == 0; it = take()) {
          Slot& S = C->slot[0];
          rc = prefetch(J, *C, *L, S, *it);
          if (rc == 0)
            rc = compute(J, *C, *L, S);
          if (rc == 0)
            rc = finish(J, *C, *L, S);
        }
      }
      else if (it)
        rc = prefetch(J, *C, *L, C->slot[cur], *it);
      while (helpersOn && it && rc == 0) {
        Slot& S = C->slot[cur];
        Slot& N = C->slot[cur ^ 1];
        const Item* nextIt = take();
        if (nextIt)   // (N's input side is free: the item that computed out of it has been computed)
          fIn = helper([&J, C, L, &N, nextIt, &prefetch]() { return prefetch(J, *C, *L, N, *nextIt); });
        if (fOut[cur].valid())   // S's output side: the item before the last must have left it
          rc = fOut[cur].get();
        if (rc == 0)
          rc = compute(J, *C, *L, S);
        if (rc == 0)
          fOut[cur] = helper([&J, C, L, &S, &finish]() { return finish(J, *C, *L, S); });
        if (nextIt) {
          const int r2 = fIn.get();
          rc = rc ? rc : r2;
        }
        cur ^= 1;
        it = nextIt;
      }
    }
    catch (...) {
      rc = -1;
    }
    for (auto& f : fOut)   // (also after a failure: a helper may still be at work on this worker's buffers)
      if (f.valid()) {
        int r2 = -1;
        try {
          r2 = f.get();
        }
        catch (...) {
        }
        rc = rc ? rc : r2;
      }
    if (fIn.valid()) {
      try {
        (void)fIn.get();
      }
      catch (...) {
      }
    }
    if (rc)
      J.failed = rc;
    // nothing of this worker may still be in flight when its buffers go back to the pool (a failed
    // item leaves copies queued on the lanes: they are waited for here as well)
    (void)hipStreamSynchronize(C->st);
    for (Slot& sl : C->slot) {
      if (sl.usedIn)
        (void)hipEventSynchronize(sl.evIn);
      if (sl.usedComp) {
        (void)hipEventSynchronize(sl.evComp);
        (void)hipEventSynchronize(sl.evOut);
      }
    }
    ctx_release(C);
  };
  // the calling thread keeps its current device: all workers are threads of their own (the items
  // come from a shared queue, so fewer workers than planned still do all of them; none at all fails)
  th.reserve(nw);
  for (size_t w = 0; w < nw; w++) {
    try {
      th.emplace_back(body, w);
    }
    catch (const std::system_error&) {
      break;
    }
  }
  if (th.empty())
    J.failed = -1;
  else if (th.size() < nw)   // (the queue still hands every item out, but maybe not to every device)
    fprintf(stderr, "[sperr_hip] chunk farm: only %zu of %zu worker threads started; devices without a worker idle\n",
            th.size(), nw);
  for (auto& t : th)
    t.join();
  return J.failed.load();
}

void assign_workers(Job& J, const std::vector<int>& devs)
{
  // worker w drives device devs[w mod ndev]; no more workers than items
  const size_t most = devs.size() * J.fs.workersPerDevice;
  const size_t nw = std::max<size_t>(1, std::min(most, J.items.size()));
  J.workerDev.clear();
  for (size_t w = 0; w < nw; w++)
    J.workerDev.push_back(devs[w % devs.size()]);
}

uint64_t rounded_up8(uint64_t bits)
{
  return (bits + 7) / 8 * 8;
}

int farm_compress(const void* src, int is_float, const Dims3& vol, const Dims3& chunkPref, int mode,
                  double quality, size_t nthreads, const int* devList, size_t nDev, void** dst,
                  size_t* dst_len)
{
  Job J;
  J.vol = vol;
  for (int a = 0; a < 3; a++) {   // SPERR3D_OMP_C.cpp:23-30
    J.cdim[a] = std::min(std::max<size_t>(1, chunkPref[a]), vol[a]);
    if (vol[a] > 0xffffffffull || J.cdim[a] > 0xffff)
      return -1;   // the header holds 32-bit volume and 16-bit chunk dims (SPERR3D_OMP_C.cpp:210-221)
  }
  std::vector<int> devs;
  if (device_list(devList, nDev, devs))
    return -1;
  const auto chunks = host_chunk_volume(vol, J.cdim);
  const size_t nchunks = chunks.size();
  if (nchunks > 0xffffffffull)
    return -1;
  J.fs = farm_shape(nthreads, devs.size());
  J.esz = is_float ? 4 : 8;
  // (items per worker: enough of them that the first item's copy in and the last item's copy out, which
  //  nothing overlaps, stay short; SPERR_HIP_FARM_PER_WORKER)
  // (the modes without a bit budget wait on the host inside the device call -- the search for q, the
  //  outlier passes --: a third worker per device hides that; measured, 1024^3 at a tolerance of 1e-3 of
  //  the range: 132 ms with two workers, 106 with three)
  if (mode != 1 && !getenv("SPERR_HIP_FARM_WORKERS"))
    J.fs.workersPerDevice = 3;
  J.helperThreads = mode == 1;
  J.items = make_items(chunks, J.esz, devs.size() * J.fs.workersPerDevice, J.fs,
                       env_size("SPERR_HIP_FARM_PER_WORKER", mode == 1 ? 6 : 2));
  assign_workers(J, devs);
  J.src = static_cast<const uint8_t*>(src);
  J.is_float = is_float;
  J.mode = mode;
  J.quality = quality;
  const bool allowDirect = !(getenv("SPERR_HIP_PINNED_COPY") && strcmp(getenv("SPERR_HIP_PINNED_COPY"), "stage") == 0);
  J.direct = allowDirect && host_ptr_is_pinned(src);
  J.chunkLen.assign(nchunks, 0);
  const size_t hdr = (nchunks > 1 ? 20 : 14) + 4 * nchunks;

  // Fixed rate: a chunk stream is 17 + 9 + budget / 8 bytes unless the chunk is constant or runs
  // out of bits first, so the streams can land where they belong in the final buffer
  // (src/SPECK_INT.cpp:54-56: the budget is rounded up to whole bytes).
  bool inPlace = mode == 1;
  if (inPlace) {
    J.slotOff.assign(nchunks + 1, hdr);
    for (size_t i = 0; i < nchunks; i++) {
      const size_t n = chunks[i][1] * chunks[i][3] * chunks[i][5];
      const double raw = quality * (double)n;
      if (!(raw < 1.8e19)) {
        inPlace = false;
        break;
      }
      const uint64_t bits = rounded_up8((uint64_t)raw);
      const size_t len = std::min<size_t>(26 + (size_t)(bits / 8), host_chunk_stream_bound(n, mode, quality));
      J.slotOff[i + 1] = J.slotOff[i] + len;
    }
    if (inPlace) {
      J.outBuf = static_cast<uint8_t*>(malloc(J.slotOff[nchunks]));
      if (!J.outBuf)
        return -1;
    }
  }
  if (!inPlace) {
    J.slotOff.clear();
    J.chunkBytes.resize(nchunks);
  }

  const int rc = run_workers(J, comp_prefetch, comp_compute, comp_finish);
  if (rc)
    return rc;

  // header (src/SPERR3D_OMP_C.cpp:163-234) + the chunk streams back to back
  size_t total = hdr;
  for (size_t i = 0; i < nchunks; i++) {
    if (J.chunkLen[i] > 0xffffffffull)
      return -1;
    total += J.chunkLen[i];
  }
  std::vector<size_t> at;
  if (!inPlace) {
    at.assign(nchunks + 1, hdr);
    for (size_t i = 0; i < nchunks; i++)
      at[i + 1] = at[i] + J.chunkLen[i];
  }
  // (parallel_do below may throw -- bad_alloc, a thread that does not start: `out`, gigabytes at
  // times, is owned until it is handed to the caller)
  struct FreeDel { void operator()(uint8_t* p) const { free(p); } };
  std::unique_ptr<uint8_t, FreeDel> outOwner(J.outBuf);
  uint8_t* out = J.outBuf;
  J.outBuf = nullptr;
  if (inPlace) {
    size_t to = hdr;
    for (size_t i = 0; i < nchunks; i++) {   // (moves nothing when every stream fills its slot)
      if (to != J.slotOff[i])
        memmove(out + to, out + J.slotOff[i], J.chunkLen[i]);
      to += J.chunkLen[i];
    }
    if (total < J.slotOff[nchunks]) {
      uint8_t* shrunk = static_cast<uint8_t*>(realloc(out, total));
      if (shrunk) {
        (void)outOwner.release();
        outOwner.reset(shrunk);
        out = shrunk;
      }
    }
  }
  else {
    outOwner.reset(static_cast<uint8_t*>(malloc(total)));   // (frees the slot buffer, if there was one)
    out = outOwner.get();
    if (!out)
      return -1;
    const size_t nt = std::max<size_t>(1, std::min<size_t>(J.fs.helpers * J.workerDev.size(), nchunks));
    parallel_do(nt, [&](size_t t) {
      for (size_t i = nchunks * t / nt; i < nchunks * (t + 1) / nt; i++)
        memcpy(out + at[i], J.chunkBytes[i].get(), J.chunkLen[i]);
    });
  }
  out[0] = 0;   // SPERR_VERSION_MAJOR (CMakeLists.txt:5)
  out[1] = (uint8_t)(0x40 | (is_float ? 0x20 : 0) | (nchunks > 1 ? 0x10 : 0));
  const uint32_t v3[3] = {(uint32_t)vol[0], (uint32_t)vol[1], (uint32_t)vol[2]};
  memcpy(out + 2, v3, 12);
  size_t pos = 14;
  if (nchunks > 1) {
    const uint16_t c3[3] = {(uint16_t)J.cdim[0], (uint16_t)J.cdim[1], (uint16_t)J.cdim[2]};
    memcpy(out + 14, c3, 6);
    pos = 20;
  }
  for (size_t i = 0; i < nchunks; i++) {
    const uint32_t l = (uint32_t)J.chunkLen[i];
    memcpy(out + pos + 4 * i, &l, 4);
  }
  *dst = outOwner.release();
  *dst_len = total;
  return 0;
}

int farm_decompress(const void* src, size_t src_len, int output_float, size_t nthreads,
                    const int* devList, size_t nDev, const HostContainer& hc, void* dstVol)
{
  Job J;
  J.vol = hc.vol;
  J.cdim = hc.chunk;
  std::vector<int> devs;
  if (device_list(devList, nDev, devs))
    return -1;
  const auto chunks = host_chunk_volume(hc.vol, hc.chunk);
  if (chunks.size() != hc.len.size())
    return -1;
  J.fs = farm_shape(nthreads, devs.size());
  // (items sized by the bytes that come out; fewer, larger items than the encoder wants: the
  // decoder's list kernels keep one workgroup per chunk busy)
  J.fs.itemBytesMax = env_size("SPERR_HIP_FARM_DEC_ITEM_MB", 1536) << 20;
  J.fs.workersPerDevice = env_size("SPERR_HIP_FARM_DEC_WORKERS", J.fs.workersPerDevice);
  J.items = make_items(chunks, output_float ? 4 : 8, devs.size() * J.fs.workersPerDevice, J.fs,
                       env_size("SPERR_HIP_FARM_DEC_PER_WORKER", 4));
  assign_workers(J, devs);
  J.container = static_cast<const uint8_t*>(src);
  J.hc = &hc;
  J.output_float = output_float;
  J.dstVol = static_cast<uint8_t*>(dstVol);
  const bool allowDirect = !(getenv("SPERR_HIP_PINNED_COPY") && strcmp(getenv("SPERR_HIP_PINNED_COPY"), "stage") == 0);
  J.direct = allowDirect && host_ptr_is_pinned(dstVol);
  (void)src_len;
  return run_workers(J, decomp_prefetch, decomp_compute, decomp_finish);
}

template <typename F>
int guarded_farm(const char* what, F&& body) noexcept
{
  try {
    return body();
  }
  catch (const std::bad_alloc&) {
    fprintf(stderr, "[sperr_hip] %s: out of host memory\n", what);
  }
  catch (const std::exception& e) {
    fprintf(stderr, "[sperr_hip] %s: %s\n", what, e.what());
  }
  catch (...) {
    fprintf(stderr, "[sperr_hip] %s: unknown exception\n", what);
  }
  return -1;
}

}  // namespace
}  // namespace sperrhip

using namespace sperrhip;

extern "C" {

int sperrhip_comp_3d_farm(const void* src, int is_float, size_t dimx, size_t dimy, size_t dimz,
                          size_t chunk_x, size_t chunk_y, size_t chunk_z, int mode, double quality,
                          size_t nthreads, const int* devices, size_t ndevices, void** dst,
                          size_t* dst_len)
{
  return guarded_farm("sperrhip_comp_3d_farm", [&]() -> int {
    if (!dst || *dst != nullptr)
      return 1;
    if (quality <= 0.0)
      return 2;
    if (mode < 1 || mode > 3)
      return 2;
    if (!src || !dst_len || dimx == 0 || dimy == 0 || dimz == 0)
      return -1;
    int dev = 0;
    if (classify_ptr(src, &dev) == 2) {
      // the volume already lives on a device: no farm, the device-resident compressor there, and
      // only the container travels
      DeviceGuard guard(dev);
      const size_t cap = sperrhip_max_compressed_size(dimx, dimy, dimz, chunk_x, chunk_y, chunk_z, mode, quality);
      void* d_out = nullptr;
      if (hipMalloc(&d_out, cap) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
      }
      size_t len = 0;
      int rc = sperrhip_compress_dev(src, is_float, dimx, dimy, dimz, chunk_x, chunk_y, chunk_z, mode, quality,
                                     d_out, cap, &len, nullptr);
      if (rc == 0) {
        void* buf = malloc(std::max<size_t>(len, 1));
        if (buf && hipMemcpy(buf, d_out, len, hipMemcpyDeviceToHost) == hipSuccess) {
          *dst = buf;
          *dst_len = len;
        }
        else {
          free(buf);
          rc = -1;
        }
      }
      (void)hipFree(d_out);
      return rc;
    }
    return farm_compress(src, is_float, Dims3{dimx, dimy, dimz}, Dims3{chunk_x, chunk_y, chunk_z},
                         mode, quality, nthreads, devices, ndevices, dst, dst_len);
  });
}

int sperrhip_decomp_3d_into(const void* src, size_t src_len, int output_float, size_t nthreads,
                            const int* devices, size_t ndevices, void* dst, size_t dst_bytes,
                            size_t* dimx, size_t* dimy, size_t* dimz)
{
  return guarded_farm("sperrhip_decomp_3d_into", [&]() -> int {
    if (!src || !dst)
      return -1;
    HostContainer hc;
    if (host_parse_container(static_cast<const uint8_t*>(src), src_len, hc))
      return -1;
    if (hc.nvals > dst_bytes / (output_float ? 4 : 8))
      return -1;
    int dev = 0;
    if (classify_ptr(dst, &dev) == 2) {
      // the volume is wanted on a device: the container goes there, the device-resident decoder runs
      DeviceGuard guard(dev);
      void* d_in = nullptr;
      if (hipMalloc(&d_in, std::max<size_t>(src_len, 1)) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
      }
      int r2 = -1;
      if (hipMemcpy(d_in, src, src_len, hipMemcpyHostToDevice) == hipSuccess)
        r2 = sperrhip_decompress_dev(d_in, src_len, output_float, dst, dst_bytes, dimx, dimy, dimz, nullptr);
      (void)hipFree(d_in);
      return r2;
    }
    const int rc = farm_decompress(src, src_len, output_float, nthreads, devices, ndevices, hc, dst);
    if (rc == 0) {
      if (dimx)
        *dimx = hc.vol[0];
      if (dimy)
        *dimy = hc.vol[1];
      if (dimz)
        *dimz = hc.vol[2];
    }
    return rc;
  });
}

int sperrhip_decomp_3d_farm(const void* src, size_t src_len, int output_float, size_t nthreads,
                            const int* devices, size_t ndevices, size_t* dimx, size_t* dimy,
                            size_t* dimz, void** dst)
{
  return guarded_farm("sperrhip_decomp_3d_farm", [&]() -> int {
    if (!dst || *dst != nullptr)
      return 1;
    if (!src)
      return -1;
    HostContainer hc;
    if (host_parse_container(static_cast<const uint8_t*>(src), src_len, hc))
      return -1;
    const size_t bytes = hc.nvals * (output_float ? 4 : 8);
    void* buf = malloc(std::max<size_t>(bytes, 1));
    if (!buf)
      return -1;
    const int rc = farm_decompress(src, src_len, output_float, nthreads, devices, ndevices, hc, buf);
    if (rc) {
      free(buf);
      return rc;
    }
    *dst = buf;
    *dimx = hc.vol[0];
    *dimy = hc.vol[1];
    *dimz = hc.vol[2];
    return 0;
  });
}

// ---- reference-compatible host API (src/SPERR_C_API.cpp:135-258) ---------------------------

int sperr_comp_3d(const void* src, int is_float, size_t dimx, size_t dimy, size_t dimz,
                  size_t chunk_x, size_t chunk_y, size_t chunk_z, int mode, double quality,
                  size_t nthreads, void** dst, size_t* dst_len)
{
  return sperrhip_comp_3d_farm(src, is_float, dimx, dimy, dimz, chunk_x, chunk_y, chunk_z, mode,
                               quality, nthreads, nullptr, 0, dst, dst_len);
}

int sperr_decomp_3d(const void* src, size_t src_len, int output_float, size_t nthreads,
                    size_t* dimx, size_t* dimy, size_t* dimz, void** dst)
{
  return sperrhip_decomp_3d_farm(src, src_len, output_float, nthreads, nullptr, 0, dimx, dimy, dimz,
                                 dst);
}

// ---- NUMA placement of the farm (numa_place.hpp) -------------------------------------------
// host only: NUMA node and CPUs of the PCI device `pci_bdf` as the sysfs tree under `sysfs_root`
// (NULL: SPERR_HIP_SYSFS_ROOT or /sys) describes them
int sperrhip_numa_probe(const char* sysfs_root, const char* pci_bdf, int* node, int* cpus, size_t cpus_cap,
                        size_t* ncpus)
{
  return guarded_farm("sperrhip_numa_probe", [&]() -> int {
    if (!pci_bdf)
      return -1;
    const numa::Place pl = numa::probe(sysfs_root && *sysfs_root ? std::string(sysfs_root) : numa::sysfs_root(), pci_bdf);
    if (node)
      *node = pl.node;
    if (ncpus)
      *ncpus = pl.cpus.size();
    if (cpus)
      for (size_t i = 0; i < std::min(cpus_cap, pl.cpus.size()); i++)
        cpus[i] = pl.cpus[i];
    return 0;
  });
}
// host only: what a farm worker of that device does to itself -- the CALLING thread is bound to the
// device's NUMA node; returns the number of CPUs it is bound to (0: left as it was, also with
// SPERR_HIP_FARM_NUMA=0), -1 on error
int sperrhip_numa_bind_self(const char* sysfs_root, const char* pci_bdf)
{
  return guarded_farm("sperrhip_numa_bind_self", [&]() -> int {
    if (!pci_bdf)
      return -1;
    if (!numa::enabled())
      return 0;
    const numa::Place pl = numa::probe(sysfs_root && *sysfs_root ? std::string(sysfs_root) : numa::sysfs_root(), pci_bdf);
    return (int)numa::bind_self(pl);
  });
}
// host only: the CPUs this process may use (host_cpus.hpp) -- what the farm sizes its threads by.
// cgroup_root / proc_cgroup NULL = /sys/fs/cgroup and /proc/self/cgroup (or SPERR_HIP_CGROUP_ROOT /
// SPERR_HIP_PROC_CGROUP).  quota_cpus 0 = no CFS limit found.
int sperrhip_host_cpus(const char* cgroup_root, const char* proc_cgroup, size_t* visible, size_t* affinity,
                       double* quota_cpus, size_t* usable)
{
  return guarded_farm("sperrhip_host_cpus", [&]() -> int {
    const hostcpu::Budget b = (cgroup_root && *cgroup_root)
                                  ? hostcpu::probe(cgroup_root, (proc_cgroup && *proc_cgroup) ? proc_cgroup : "/proc/self/cgroup")
                                  : hostcpu::probe();
    if (visible)
      *visible = b.visible;
    if (affinity)
      *affinity = b.affinity;
    if (quota_cpus)
      *quota_cpus = b.quota;
    if (usable)
      *usable = b.usable;
    return 0;
  });
}
// host only: CFS throttling of the process's cgroup so far (cpu.stat: nr_throttled, throttled_usec);
// 1 when no cpu.stat with those fields was found (then both are 0)
int sperrhip_host_throttle(const char* cgroup_root, const char* proc_cgroup, unsigned long long* nr_throttled,
                           unsigned long long* throttled_usec)
{
  return guarded_farm("sperrhip_host_throttle", [&]() -> int {
    const char* r = getenv("SPERR_HIP_CGROUP_ROOT");
    const char* q = getenv("SPERR_HIP_PROC_CGROUP");
    const std::string root = (cgroup_root && *cgroup_root) ? cgroup_root : ((r && *r) ? r : "/sys/fs/cgroup");
    const std::string proc = (proc_cgroup && *proc_cgroup) ? proc_cgroup : ((q && *q) ? q : "/proc/self/cgroup");
    unsigned long long nr = 0, us = 0;
    const bool ok = hostcpu::throttle_stat(root, proc, nr, us);
    if (nr_throttled)
      *nr_throttled = nr;
    if (throttled_usec)
      *throttled_usec = us;
    return ok ? 0 : 1;
  });
}
// host only: the farm's thread plan for `ndevices` devices and a caller's `nthreads` (0: the library's
// choice) -- workers per device (compress / decompress), helper threads per worker, and the sum of
// worker + helper threads of a compression against the CPUs the process may use
int sperrhip_farm_threads(size_t nthreads, size_t ndevices, size_t* workers_per_device, size_t* dec_workers_per_device,
                          size_t* helpers_per_worker, size_t* threads_total, size_t* cpus_usable)
{
  return guarded_farm("sperrhip_farm_threads", [&]() -> int {
    if (ndevices == 0)
      return -1;
    const FarmShape fs = farm_shape(nthreads, ndevices);
    const size_t decW = env_size("SPERR_HIP_FARM_DEC_WORKERS", fs.workersPerDevice);
    if (workers_per_device)
      *workers_per_device = fs.workersPerDevice;
    if (dec_workers_per_device)
      *dec_workers_per_device = decW;
    if (helpers_per_worker)
      *helpers_per_worker = fs.helpers;
    if (threads_total)
      *threads_total = ndevices * fs.workersPerDevice * (1 + fs.helpers);
    if (cpus_usable)
      *cpus_usable = hostcpu::probe().usable;
    return 0;
  });
}
// where the farm puts the workers of device `dev`: its PCI address (bdf_cap bytes of room), NUMA
// node (-1: unknown) and the number of CPUs of that node
int sperrhip_farm_device_place(int dev, char* bdf, size_t bdf_cap, int* node, size_t* ncpus)
{
  return guarded_farm("sperrhip_farm_device_place", [&]() -> int {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || dev < 0 || dev >= ndev) {
      (void)hipGetLastError();
      return -1;
    }
    char buf[64] = {0};
    if (hipDeviceGetPCIBusId(buf, (int)sizeof(buf), dev) != hipSuccess) {
      (void)hipGetLastError();
      return -1;
    }
    if (bdf && bdf_cap) {
      strncpy(bdf, buf, bdf_cap - 1);
      bdf[bdf_cap - 1] = 0;
    }
    const numa::Place& pl = device_place(dev);
    if (node)
      *node = pl.node;
    if (ncpus)
      *ncpus = pl.cpus.size();
    return 0;
  });
}

// The queue without any device: how the chunks of a volume are cut into items and which worker
// takes which, with `nworkers` host threads that do nothing else.  lockstep != 0: the workers take
// their items in rounds (every worker equally fast), which makes the outcome deterministic.
// per_worker_chunks[nworkers], item_of_chunk[nchunks] (either may be NULL), *nitems.
int sperrhip_farm_selftest(size_t dimx, size_t dimy, size_t dimz, size_t chunk_x, size_t chunk_y,
                           size_t chunk_z, size_t bytes_per_value, size_t ndevices,
                           size_t workers_per_device, int lockstep, uint32_t* per_worker_chunks,
                           uint32_t* item_of_chunk, uint32_t* worker_of_chunk, size_t* nitems)
{
  return guarded_farm("sperrhip_farm_selftest", [&]() -> int {
    if (dimx == 0 || dimy == 0 || dimz == 0 || ndevices == 0 || workers_per_device == 0)
      return -1;
    const Dims3 vol{dimx, dimy, dimz};
    Dims3 cd{chunk_x, chunk_y, chunk_z};
    for (int a = 0; a < 3; a++)
      cd[a] = std::min(std::max<size_t>(1, cd[a]), vol[a]);
    const auto chunks = host_chunk_volume(vol, cd);
    FarmShape fs = farm_shape(0, ndevices);
    fs.workersPerDevice = workers_per_device;
    const auto items = make_items(chunks, bytes_per_value, ndevices * workers_per_device, fs);
    const size_t nw = std::max<size_t>(1, std::min(ndevices * workers_per_device, items.size()));
    if (nitems)
      *nitems = items.size();
    if (item_of_chunk)
      for (size_t i = 0; i < items.size(); i++)
        for (uint32_t g : items[i].gid)
          item_of_chunk[g] = (uint32_t)i;
    std::vector<uint32_t> count(ndevices * workers_per_device, 0);
    std::atomic<size_t> next{0};
    std::mutex mu;
    std::condition_variable cv;
    size_t waiting = 0, alive = nw, round = 0;
    auto body = [&](size_t w) {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= items.size())
          break;
        count[w] += (uint32_t)items[i].gid.size();
        if (worker_of_chunk)
          for (uint32_t g : items[i].gid)
            worker_of_chunk[g] = (uint32_t)w;
        if (lockstep) {   // wait until every live worker has taken its item of this round
          std::unique_lock<std::mutex> lock(mu);
          const size_t my = round;
          if (++waiting == alive) {
            waiting = 0;
            round++;
            cv.notify_all();
          }
          else
            cv.wait(lock, [&]() { return round != my; });
        }
      }
      if (lockstep) {
        std::lock_guard<std::mutex> lock(mu);
        alive--;
        if (alive && waiting == alive) {
          waiting = 0;
          round++;
          cv.notify_all();
        }
      }
    };
    std::vector<std::thread> th;
    for (size_t w = 0; w < nw; w++)
      th.emplace_back(body, w);
    for (auto& t : th)
      t.join();
    if (per_worker_chunks)
      for (size_t w = 0; w < ndevices * workers_per_device; w++)
        per_worker_chunks[w] = count[w];
    return 0;
  });
}

}  // extern "C"
