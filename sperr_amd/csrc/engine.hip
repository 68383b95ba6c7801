// engine.hip -- host side of the HIP engine: chunk grid, per-shape plans, batched workspaces,
// container assembly / parsing, and the C ABI declared in include/sperr_hip.h.
//
// Host logic restated from the reference (file:line under /root/reference):
//   src/sperr_helper.cpp:542-592            chunk_volume (x fastest, short remainders merged)
//   src/SPERR3D_OMP_C.cpp:23-30,61-141      chunk loop  -> batches of equally shaped chunks
//   src/SPERR3D_OMP_C.cpp:145-234           container header + concatenated chunk streams
//   src/SPERR3D_Stream_Tools.cpp:46-105     container header parsing
//   src/SPERR3D_OMP_D.cpp:23-135            decompress driver
//   src/SPECK_FLT.cpp:401-541               per-chunk pipeline order and the fixed-rate retry
//   src/SPERR_C_API.cpp:135-258             C API semantics (ownership, return codes)
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <deque>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/sperr_hip.h"
#include "engine_internal.h"
#include "host_container.hpp"
#include "speck_dec.h"
#include "speck_enc.h"
#include "speck_tree_host.hpp"
#include "outlier.h"
#include "speck2d.h"
#include "xform.h"

// The decoder runs the shape groups of a volume side by side on up to eight streams; the ROCm
// runtime maps streams onto four hardware queues unless told otherwise, and reads the variable at
// the process's first HIP call: a host that loads this library before that gets eight.
__attribute__((constructor)) static void sperrhip_ask_for_hw_queues()
{
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

namespace sperrhip {

// ------------------------------------------------------------------------------------------
// profiling
// ------------------------------------------------------------------------------------------
namespace {

struct ProfEntry {
  double ms = 0.0;     // sum of the launch durations
  double busy = 0.0;   // time during which at least one launch of the kernel was running (launches
                       // of sub-batches on different streams overlap)
  int launches = 0;
};

struct Profiler {
  bool on = false;
  std::string only;   // when not empty: only launches of this kernel are bracketed
  std::mutex mu;      // sub-batches are enqueued from several host threads
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  hipEvent_t ref = nullptr;   // origin of the time axis of one collect() period
  struct Open {
    const char* name;
    hipEvent_t a, b;
  };
  std::vector<Open> open;
  std::map<std::string, ProfEntry> acc;

  hipEvent_t get()   // (mu held)
  {
    if (used == pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess)
        return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
  void collect()
  {
    std::lock_guard<std::mutex> lock(mu);
    std::map<std::string, std::vector<std::pair<float, float>>> spans;
    for (auto& o : open) {
      float ms = 0.f, t0 = 0.f;
      if (o.a && o.b && hipEventElapsedTime(&ms, o.a, o.b) == hipSuccess) {
        auto& e = acc[o.name];
        e.ms += ms;
        e.launches++;
        if (ref && hipEventElapsedTime(&t0, ref, o.a) == hipSuccess)
          spans[o.name].push_back({t0, t0 + ms});
        else
          e.busy += ms;
      }
    }
    for (auto& kv : spans) {
      auto& v = kv.second;
      std::sort(v.begin(), v.end());
      double busy = 0.0;
      float lo = v[0].first, hi = v[0].second;
      for (size_t i = 1; i < v.size(); i++) {
        if (v[i].first > hi) {
          busy += hi - lo;
          lo = v[i].first;
          hi = v[i].second;
        }
        else
          hi = std::max(hi, v[i].second);
      }
      acc[kv.first].busy += busy + (hi - lo);
    }
    open.clear();
    used = 0;
    ref = nullptr;
  }
};

// global switches (sperrhip_profile_enable / _only); the accumulators live in the engines
bool g_prof_on = false;
std::string g_prof_only;
std::mutex g_prof_cfg_mu;
thread_local Profiler* t_prof = nullptr;       // the profiler of the engine this thread drives
thread_local const char* t_prof_cur = nullptr;
thread_local hipEvent_t t_prof_a = nullptr;

}  // namespace

void prof_begin(const char* name, hipStream_t stream)
{
  Profiler* P = t_prof;
  if (!P || !P->on || (!P->only.empty() && P->only != name))
    return;
  std::lock_guard<std::mutex> lock(P->mu);
  if (!P->ref) {
    P->ref = P->get();
    if (P->ref)
      hipEventRecord(P->ref, stream);
  }
  t_prof_cur = name;
  t_prof_a = P->get();
  if (t_prof_a)
    hipEventRecord(t_prof_a, stream);
}

void prof_end(hipStream_t stream)
{
  Profiler* P = t_prof;
  if (!P || !P->on || !t_prof_cur)
    return;
  std::lock_guard<std::mutex> lock(P->mu);
  hipEvent_t b = P->get();
  if (b)
    hipEventRecord(b, stream);
  P->open.push_back({t_prof_cur, t_prof_a, b});
  t_prof_cur = nullptr;
}

thread_local bool t_shared_device = false;   // (engine_internal.h)

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
namespace {

struct DevBuf {
  void* p = nullptr;
  size_t n = 0;
  void drop()
  {
    if (p)
      (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  int ensure(size_t bytes)
  {
    if (bytes <= n)
      return 0;
    if (p)
      (void)hipFree(p);
    p = nullptr;
    n = 0;
    // (a quarter more than asked for, for the small buffers whose size follows the data -- outlier
    //  streams, slots --: a hipFree waits for every stream of the device, the other workers' included,
    //  and a buffer that fits exactly is too small for the next call's slightly longer streams)
    size_t want = bytes < (size_t(256) << 20) ? bytes + bytes / 4 + 4096 : bytes;
    // (callers size their requests against the free memory they saw: when the slack is what does not fit,
    //  the exact size still may)
    if (want != bytes && hipMalloc(&p, want) != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      want = bytes;
    }
    if (!p)
      HIP_CHECK(hipMalloc(&p, want));
    n = want;
    return 0;
  }
};

std::atomic<unsigned long long> g_dbg_counter[3];   // sperrhip_debug_counter

struct Arena {
  char* base = nullptr;
  size_t cap = 0, used = 0;
  template <typename T>
  T* take(size_t count)
  {
    // sizes may come from an untrusted header: nothing here may wrap
    if (used > cap || count > (cap - used) / sizeof(T))
      return nullptr;
    const size_t bytes = (count * sizeof(T) + 255) / 256 * 256;
    if (bytes > cap - used)
      return nullptr;
    T* r = reinterpret_cast<T*>(base + used);
    used += bytes;
    return r;
  }
};

// bytes that hold `bits` bits; bit counts read from a container may be anything up to 2^64 - 1
inline uint64_t bytes_of_bits(uint64_t bits)
{
  return bits / 8 + (bits % 8 != 0);
}

size_t round_up(size_t v, size_t m)
{
  return (v + m - 1) / m * m;
}

using hostc::Dims;
using hostc::chunk_count;
using hostc::chunk_volume;
using hostc::ContainerInfo;
using hostc::parse_container_host;

// src/Conditioner.cpp:137-163
uint32_t condi_num_strides(size_t len)
{
  const size_t dflt = 2048;
  if (len % dflt == 0)
    return (uint32_t)dflt;
  for (size_t n = dflt; n <= 32768; n++)
    if (len % n == 0)
      return (uint32_t)n;
  size_t n = dflt;
  while (len % n != 0)
    n--;
  return (uint32_t)n;
}

// ------------------------------------------------------------------------------------------
// per-shape plan: tree tables on the device, tile maps, DWT pass list
// ------------------------------------------------------------------------------------------
struct LiftPass {
  int axis;
  uint32_t region[3];
};

struct ShapePlan {
  uint32_t dims[3];
  uint32_t N = 0;
  spk::HostTree ht;
  spk::Tree dtree{};
  DevBuf tables;
  const uint64_t* d_initLIS = nullptr;
  const uint32_t* d_initLen = nullptr;
  const uint32_t* d_levelOff = nullptr;
  const uint16_t* d_tileLevel = nullptr;
  const uint32_t* d_tileStart = nullptr;
  const uint32_t* d_levelFirstTile = nullptr;
  const uint32_t* d_levelNumTiles = nullptr;
  const uint32_t* d_depthBlocks = nullptr;
  const uint8_t* d_levelSlot = nullptr;
  const uint64_t* d_iRoots = nullptr;   // (2D forest)
  const uint8_t* d_slotLevel = nullptr;
  const spk::LevelClass* d_levelClass = nullptr;
  const uint32_t* d_wordLeaf = nullptr;   // nullptr: no raster word lies over leaf-word grids
  const uint8_t* d_mxSlot = nullptr;      // k_lis_mx: column of every shape class
  const uint8_t* d_mxLevelGroup = nullptr;
  int l0Level = -1;                       // LIS level of 2x2x2 leaf sets that k_lis_l0 can decode
  int l1Level = -1;                       // LIS level of 4x4x4 sets that k_lis_l1 can decode
  int l2Level = -1;                       // LIS level of 8x8x8 sets that k_lis_l2 can decode
  int maxK = 0;
  std::vector<uint32_t> depthBlockOff;
  uint32_t nListTiles = 0, nSlots = 0, nPixTiles = 0, nstrides = 0;
  uint64_t maxPhaseBits = 0;
  size_t lisEntries = 0;
  std::vector<LiftPass> fwd;
};
bool use_tables(const ShapePlan& P);   // (defined with the decoder's plan logic below)


struct Blob {  // host-side staging of all tables of a plan, uploaded in one copy
  std::vector<char> bytes;
  template <typename T>
  size_t add(const std::vector<T>& v)
  {
    const size_t off = round_up(bytes.size(), 256);
    bytes.resize(off + std::max<size_t>(v.size(), 1) * sizeof(T), 0);
    if (!v.empty())
      memcpy(bytes.data() + off, v.data(), v.size() * sizeof(T));
    return off;
  }
};

// twoD: the plan of a slice's DECODER, with the forest of the 2D coder (spk::kTree2D)
int build_plan(ShapePlan& P, size_t dx, size_t dy, size_t dz, bool twoD = false)
{
  // set coordinates are packed 16 bits per axis (speck_tree.h pack_node), sample indices are 32 bits
  const unsigned __int128 samples = (unsigned __int128)dx * dy * dz;
  if (dx == 0 || dy == 0 || dz == 0 || dx > 0xffff || dy > 0xffff || dz > 0xffff || samples > (1ull << 31)) {
    fprintf(stderr, "[sperr_hip] chunk of %zu x %zu x %zu not supported (at most 65535 per axis, 2^31 samples)\n",
            dx, dy, dz);
    return -1;
  }
  P.dims[0] = (uint32_t)dx;
  P.dims[1] = (uint32_t)dy;
  P.dims[2] = (uint32_t)dz;
  P.N = (uint32_t)(dx * dy * dz);
  // (the columns of k_lis_mx only for the trees that can end up there: they cost as much as the rest of the tree)
  P.ht = spk::build_tree(dx, dy, dz, twoD, false);
  P.maxK = 0;
  for (const auto& lc : P.ht.levelClass)
    P.maxK = std::max<int>(P.maxK, lc.K);
  if (!use_tables(P))
    spk::build_mx_columns(P.ht);
  const spk::HostTree& h = P.ht;
  const uint32_t nlev = h.nlevels;

  // LIS storage: level l owns [levelOff[l], levelOff[l+1]); keep room for the roots
  std::vector<uint32_t> cap(nlev), levelOff(nlev + 1, 0), initLen(nlev);
  for (uint32_t l = 0; l < nlev; l++) {
    cap[l] = std::max<uint32_t>(h.levelCap[l], (uint32_t)h.initLIS[l].size());
    levelOff[l + 1] = levelOff[l] + cap[l];
    initLen[l] = (uint32_t)h.initLIS[l].size();
  }
  P.lisEntries = levelOff[nlev] + 8;
  std::vector<uint64_t> initLIS(levelOff[nlev] ? levelOff[nlev] : 1, 0);
  for (uint32_t l = 0; l < nlev; l++)
    for (size_t k = 0; k < h.initLIS[l].size(); k++)
      initLIS[levelOff[l] + k] = h.initLIS[l][k];

  // list tiles in traversal order: deepest level first
  std::vector<uint16_t> tileLevel;
  std::vector<uint32_t> tileStart, levelFirstTile(nlev, 0), levelNumTiles(nlev, 0);
  for (uint32_t l = nlev; l-- > 0;) {
    levelFirstTile[l] = (uint32_t)tileLevel.size();
    const uint32_t nt = (cap[l] + kListTile - 1) / kListTile;
    levelNumTiles[l] = nt;
    for (uint32_t t = 0; t < nt; t++) {
      tileLevel.push_back((uint16_t)l);
      tileStart.push_back(t * kListTile);
    }
  }
  P.nListTiles = (uint32_t)tileLevel.size();

  // node blocks grouped by depth
  std::vector<uint32_t> depthBlocks;
  P.depthBlockOff.assign(h.maxDepth + 1, 0);
  for (uint32_t d = 0; d < h.maxDepth; d++) {
    P.depthBlockOff[d] = (uint32_t)depthBlocks.size();
    for (uint32_t b = 0; b < h.blockGrid.size(); b++)
      if (h.grids[h.blockGrid[b]].depth == d)
        depthBlocks.push_back(b);
  }
  P.depthBlockOff[h.maxDepth] = (uint32_t)depthBlocks.size();

  // birth-mask slots: every level that can hold sets
  std::vector<uint8_t> levelSlot(nlev, 0xff), slotLevel;
  for (uint32_t l = 0; l < nlev; l++)
    if (cap[l]) {
      levelSlot[l] = (uint8_t)slotLevel.size();
      slotLevel.push_back((uint8_t)l);
    }
  P.nSlots = (uint32_t)slotLevel.size();
  P.maxPhaseBits = (uint64_t)h.nsets + 2ull * P.N + 64;
  P.nPixTiles = (P.N + kPixTile - 1) / kPixTile;
  P.nstrides = condi_num_strides(P.N);

  // raster mask words that lie over one row of 32 leaf sets (spk::kGridLeafWord)
  std::vector<uint32_t> wordLeaf;
  for (const spk::Root& r : h.roots) {
    const spk::Grid& g = h.grids[r.gridFirst + r.Dmax - 1];
    if (!(g.kind & spk::kGridLeafWord))
      continue;
    if (wordLeaf.empty())
      wordLeaf.assign((P.N + 63) / 64, 0xffffffffu);
    for (uint32_t z = 0; z < r.len[2]; z++)
      for (uint32_t y = 0; y < r.len[1]; y++)
        for (uint32_t x = 0; x < r.len[0]; x += 64) {
          const size_t idx = ((size_t)(r.org[2] + z) * dy + r.org[1] + y) * dx + r.org[0] + x;
          const uint32_t fid = g.nodeOff + ((((z / 2) << g.e[1]) + y / 2) << g.e[0]) + x / 2;
          wordLeaf[idx / 64] = fid | (y & 1u) | ((z & 1u) << 1);
        }
  }

  // upload
  Blob blob;
  const size_t oRoots = blob.add(h.roots), oGrids = blob.add(h.grids), oTab = blob.add(h.tab),
               oBG = blob.add(h.blockGrid), oInit = blob.add(initLIS), oInitLen = blob.add(initLen),
               oLevOff = blob.add(levelOff), oTL = blob.add(tileLevel), oTS = blob.add(tileStart),
               oLFT = blob.add(levelFirstTile), oLNT = blob.add(levelNumTiles),
               oDB = blob.add(depthBlocks), oLS = blob.add(levelSlot), oSL = blob.add(slotLevel),
               oLC = blob.add(h.levelClass), oWL = blob.add(wordLeaf), oCls = blob.add(h.cls),
               oGC = blob.add(h.gridCls), oIR = blob.add(h.iRoots),
               oMS = blob.add(h.mxSlot), oMG = blob.add(h.mxLevelGroup);
  P.maxK = 0;
  for (const auto& lc : h.levelClass)
    P.maxK = std::max<int>(P.maxK, lc.K);
  P.l0Level = -1;   // the first non-empty list the sorting pass visits, when it holds 2x2x2 sets
  for (uint32_t l = nlev; l-- > 0;) {
    if (cap[l] == 0)
      continue;
    if (h.allRegular && h.levelClass[l].regular && h.levelClass[l].K == 1 &&
        h.levelClass[l].arity[0] == 8)
      P.l0Level = (int)l;
    break;
  }
  P.l1Level = -1;   // the next non-empty list, when it holds 4x4x4 sets made of those leaf sets
  if (P.l0Level >= 0)
    for (uint32_t l = (uint32_t)P.l0Level; l-- > 0;) {
      if (cap[l] == 0)
        continue;
      const spk::LevelClass& lc = h.levelClass[l];
      if (lc.regular && lc.K == 2 && lc.arity[0] == 8 && lc.arity[1] == 8 &&
          lc.lev[0] == (uint8_t)P.l0Level)
        P.l1Level = (int)l;
      break;
    }
  P.l2Level = -1;   // and the one after it, when it holds 8x8x8 sets made of those (SPERR_HIP_LIS_L2=0: k_lis_hi takes it)
  static const bool l2Env = !(getenv("SPERR_HIP_LIS_L2") && atoi(getenv("SPERR_HIP_LIS_L2")) == 0);
  if (P.l1Level >= 0 && l2Env)
    for (uint32_t l = (uint32_t)P.l1Level; l-- > 0;) {
      if (cap[l] == 0)
        continue;
      const spk::LevelClass& lc = h.levelClass[l];
      if (lc.regular && lc.K == 3 && lc.arity[0] == 8 && lc.arity[1] == 8 && lc.arity[2] == 8 &&
          lc.lev[0] == (uint8_t)P.l0Level && lc.lev[1] == (uint8_t)P.l1Level)
        P.l2Level = (int)l;
      break;
    }
  if (P.tables.ensure(blob.bytes.size()))
    return -1;
  HIP_CHECK(hipMemcpy(P.tables.p, blob.bytes.data(), blob.bytes.size(), hipMemcpyHostToDevice));
  char* base = static_cast<char*>(P.tables.p);
  P.dtree = h.view();
  P.dtree.roots = reinterpret_cast<const spk::Root*>(base + oRoots);
  P.dtree.grids = reinterpret_cast<const spk::Grid*>(base + oGrids);
  P.dtree.tab = reinterpret_cast<const uint16_t*>(base + oTab);
  P.dtree.blockGrid = reinterpret_cast<const uint16_t*>(base + oBG);
  P.dtree.cls = reinterpret_cast<const spk::ShapeCls*>(base + oCls);
  P.dtree.gridCls = reinterpret_cast<const uint8_t*>(base + oGC);
  P.d_mxSlot = reinterpret_cast<const uint8_t*>(base + oMS);
  P.d_mxLevelGroup = reinterpret_cast<const uint8_t*>(base + oMG);
  P.d_iRoots = h.iRoots.empty() ? nullptr : reinterpret_cast<const uint64_t*>(base + oIR);
  P.d_initLIS = reinterpret_cast<const uint64_t*>(base + oInit);
  P.d_initLen = reinterpret_cast<const uint32_t*>(base + oInitLen);
  P.d_levelOff = reinterpret_cast<const uint32_t*>(base + oLevOff);
  P.d_tileLevel = reinterpret_cast<const uint16_t*>(base + oTL);
  P.d_tileStart = reinterpret_cast<const uint32_t*>(base + oTS);
  P.d_levelFirstTile = reinterpret_cast<const uint32_t*>(base + oLFT);
  P.d_levelNumTiles = reinterpret_cast<const uint32_t*>(base + oLNT);
  P.d_depthBlocks = reinterpret_cast<const uint32_t*>(base + oDB);
  P.d_levelSlot = reinterpret_cast<const uint8_t*>(base + oLS);
  P.d_slotLevel = reinterpret_cast<const uint8_t*>(base + oSL);
  P.d_levelClass = reinterpret_cast<const spk::LevelClass*>(base + oLC);
  P.d_wordLeaf = wordLeaf.empty() ? nullptr : reinterpret_cast<const uint32_t*>(base + oWL);

  // DWT pass list (src/CDF97.cpp:132-139,170-225,284-292,387-429); the inverse runs it backwards
  P.fwd.clear();
  size_t levels = 0;
  auto approx = [](size_t len, size_t lev) { return (uint32_t)spk::approx_detail_len(len, lev)[0]; };
  if (spk::can_use_dyadic({dx, dy, dz}, levels)) {
    for (size_t lev = 0; lev < levels; lev++) {
      LiftPass ps{0, {approx(dx, lev), approx(dy, lev), approx(dz, lev)}};
      for (int a = 0; a < 3; a++) {
        ps.axis = a;
        P.fwd.push_back(ps);
      }
    }
  }
  else {
    const size_t nz = spk::num_of_xforms(dz), nxy = spk::num_of_xforms(std::min(dx, dy));
    for (size_t lev = 0; lev < nz; lev++)
      P.fwd.push_back({2, {(uint32_t)dx, (uint32_t)dy, approx(dz, lev)}});
    for (size_t lev = 0; lev < nxy; lev++) {
      P.fwd.push_back({0, {approx(dx, lev), approx(dy, lev), (uint32_t)dz}});
      P.fwd.push_back({1, {approx(dx, lev), approx(dy, lev), (uint32_t)dz}});
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// engines: one per (device, concurrent caller)
// ------------------------------------------------------------------------------------------
// The reference's classes are re-entrant (one compressor object per OpenMP thread,
// src/SPERR3D_OMP_C.cpp:61-92).  Here an engine owns everything a call needs on ONE device (streams,
// per-shape plans, workspaces); a call leases an idle engine of the device that is current on the
// calling thread and a new one is made when all of them are busy (up to
// SPERR_HIP_ENGINES_PER_DEVICE, default 4; then callers wait).  The chunk farm (farm.hip) runs
// several workers per device this way.
constexpr uint32_t kSubStreams = 8;

struct Engine {
  int dev = -1;
  bool busy = false;
  bool ready = false;
  Profiler prof;
  hipStream_t sub[kSubStreams] = {};
  hipEvent_t evFork = nullptr, evJoin[kSubStreams] = {};
  hipStream_t outl = nullptr;               // the 1D decoder of outlier streams runs here, beside the chunk decoders
  hipStream_t outlQ[kSubStreams] = {};      //   (one per sub-batch; outlQ[0] == outl; a priority of their own: init_handles)
  hipStream_t sideQ[kSubStreams] = {};      // the encoder's census beside a part's pyramid (normal priority)
  hipEvent_t evOutl[kSubStreams] = {}, evOutlFork[kSubStreams] = {};   // (per sub-batch)
  hipEvent_t evPweFork = nullptr;          // encoder, point-wise error mode: the outlier stage's first half starts beside the 3D coder
  hipEvent_t evPweJoin = nullptr;          //   ... and its end is waited for by the batch's stream, not by the host (round 6)
  std::map<Dims, std::unique_ptr<ShapePlan>> plans;
  std::vector<Dims> planOrder;             // least recently used first
  DevBuf arena, slots, misc;
  DevBuf outlFixed, outlVar, outlStream;   // point-wise error mode: workspace of the outlier coder
  DevBuf pweBox;                           //   ... and the coarser levels' box of its reconstruction (pwe_outlier_stage)
  hipStream_t pweLastStream = nullptr;     //   stream the last batch's outlier stage ended on without a wait (round 6): the
                                           //   next batch of the SAME call may run its stage on another one and reuses the buffers
  DevBuf outlDec[kSubStreams];             //   (decoder: one per sub-batch of a call)
  DevBuf decBox[kSubStreams];              //   ... and the coarser levels' box of a sub-batch with outlier streams
  uint32_t* liveHost[kSubStreams] = {};    // pinned: answers to "do any chunks still decode" (DecPlanHost)
  void* pweHost = nullptr;                 // pinned: the outlier stage's first read-back (a copy into pageable memory
  size_t pweHostBytes = 0;                 //   would hold the host until the stream gets there: compress_impl)
  hipEvent_t liveEv[kSubStreams][kLiveSlots] = {};
  DevBuf slice2d;                           // 2D slices: lists and masks of the 2D coder
  DevBuf wideScratch;                       // 64-bit retry of a batch whose coder arrays lay over the chunk buffer
  std::vector<std::unique_ptr<DevBuf>> pweBufs;   // outlier streams of the batches of one call
  size_t freeMemAtInit = 0;

  // an engine whose initialisation fails is never handed out (EnginePool::acquire): what it made
  // up to the failure goes back at once
  int init()
  {
    if (ready)
      return 0;
    const int rc = init_handles();
    if (rc)
      destroy_handles();
    return rc;
  }
  void destroy_handles()
  {
    auto ds = [](hipStream_t& s) { if (s) (void)hipStreamDestroy(s); s = nullptr; };
    auto de = [](hipEvent_t& e) { if (e) (void)hipEventDestroy(e); e = nullptr; };
    for (uint32_t q = 0; q < kSubStreams; q++) {
      ds(sub[q]);
      de(evJoin[q]);
      if (q > 0)
        ds(outlQ[q]);
      ds(sideQ[q]);
      de(evOutl[q]);
      de(evOutlFork[q]);
      if (q == 0) {
        de(evPweFork);
        de(evPweJoin);
      }
      if (liveHost[q])
        (void)hipHostFree(liveHost[q]);
      liveHost[q] = nullptr;
      if (q == 0 && pweHost) {
        (void)hipHostFree(pweHost);
        pweHost = nullptr;
        pweHostBytes = 0;
      }
      for (int k = 0; k < kLiveSlots; k++)
        de(liveEv[q][k]);
    }
    de(evFork);
    ds(outl);
    outlQ[0] = nullptr;
    ready = false;
  }
  int init_handles()
  {
    size_t fr = 0, tot = 0;
    HIP_CHECK(hipMemGetInfo(&fr, &tot));
    freeMemAtInit = fr;
    for (uint32_t q = 0; q < kSubStreams; q++) {
      HIP_CHECK(hipStreamCreateWithFlags(&sub[q], hipStreamNonBlocking));
      HIP_CHECK(hipEventCreateWithFlags(&evJoin[q], hipEventDisableTiming));
    }
    HIP_CHECK(hipEventCreateWithFlags(&evFork, hipEventDisableTiming));
    // The streams of the 1D decoder (outlier lists, beside a sub-batch's chunk decoders) get a priority of
    // their own.  The runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues PER PRIORITY, and a side
    // stream that shares a hardware queue with the stream it is meant to run beside runs behind it instead
    // (round 4: with eight sub-streams and eight side streams on eight queues, eight point-wise-error chunks
    // at a tolerance of 1e-4 decoded in 44.8 ms -- the 30 ms of the 1D decoder and then the rest -- where 16
    // queues, or this, give 31; 16 queues cost the fixed-rate compression 2 %).  Not the encoder's census
    // streams: with a priority of their own 8 chunks compress at 63 instead of 75 GB/s.
    int prLeast = 0, prGreatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest) != hipSuccess) {
      (void)hipGetLastError();
      prLeast = prGreatest = 0;
    }
    static const int sidePrio = getenv("SPERR_HIP_SIDE_PRIORITY") ? atoi(getenv("SPERR_HIP_SIDE_PRIORITY")) : 1;
    const int prSide = sidePrio > 0 ? prGreatest : sidePrio < 0 ? prLeast : 0;   // (0: like every other stream)
    HIP_CHECK(hipStreamCreateWithPriority(&outl, hipStreamNonBlocking, prSide));
    outlQ[0] = outl;
    for (uint32_t q = 1; q < kSubStreams; q++)
      HIP_CHECK(hipStreamCreateWithPriority(&outlQ[q], hipStreamNonBlocking, prSide));
    for (uint32_t q = 0; q < kSubStreams; q++)
      HIP_CHECK(hipStreamCreateWithFlags(&sideQ[q], hipStreamNonBlocking));
    for (uint32_t q = 0; q < kSubStreams; q++) {
      HIP_CHECK(hipEventCreateWithFlags(&evOutl[q], hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&evOutlFork[q], hipEventDisableTiming));
      if (q == 0)
        HIP_CHECK(hipEventCreateWithFlags(&evPweFork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&evPweJoin, hipEventDisableTiming));
    }
    for (uint32_t q = 0; q < kSubStreams; q++) {
      HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&liveHost[q]), kLiveSlots * sizeof(uint32_t), hipHostMallocDefault));
      for (int k = 0; k < kLiveSlots; k++)
        HIP_CHECK(hipEventCreateWithFlags(&liveEv[q][k], hipEventDisableTiming));
    }
    ready = true;
    return 0;
  }

  // the device tables of at most kMaxPlans chunk shapes are kept (a ragged volume has 8 shapes)
  static constexpr size_t kMaxPlans = 64;
  // (dz = 0: the decoder's plan of a dx x dy slice, with the 2D coder's forest)
  ShapePlan* plan(size_t dx, size_t dy, size_t dz)
  {
    const Dims key{dx, dy, dz};
    auto it = plans.find(key);
    if (it != plans.end()) {
      auto pos = std::find(planOrder.begin(), planOrder.end(), key);
      if (pos != planOrder.end())
        planOrder.erase(pos);
      planOrder.push_back(key);
      return it->second.get();
    }
    auto p = std::make_unique<ShapePlan>();
    if (dz == 0 ? build_plan(*p, dx, dy, 1, true) : build_plan(*p, dx, dy, dz))
      return nullptr;
    ShapePlan* raw = p.get();
    plans[key] = std::move(p);
    planOrder.push_back(key);
    return raw;
  }
  // sperrhip_release(): everything an idle engine holds in HBM goes back (streams and events
  // stay; the next call sizes the workspaces again).  The engine's device is current.
  void drop_memory()
  {
    for (auto& kv : plans)
      kv.second->tables.drop();
    plans.clear();
    planOrder.clear();
    for (DevBuf* b : {&arena, &slots, &misc, &outlFixed, &outlVar, &outlStream, &pweBox, &slice2d, &wideScratch})
      b->drop();
    for (auto& b : outlDec)
      b.drop();
    for (auto& b : decBox)
      b.drop();
    for (auto& b : pweBufs)
      if (b)
        b->drop();
    pweBufs.clear();
  }
  // every device buffer of the engine: the arena, the containers' slots, and what point-wise error mode adds (the
  // outlier coder's arrays, the boxes of the coarser levels, the outlier streams) -- sperrhip_debug_counter(6)
  size_t device_bytes() const
  {
    size_t n = 0;
    for (const DevBuf* b : {&arena, &slots, &misc, &outlFixed, &outlVar, &outlStream, &pweBox, &slice2d, &wideScratch})
      n += b->n;
    for (const auto& b : outlDec)
      n += b.n;
    for (const auto& b : decBox)
      n += b.n;
    for (const auto& b : pweBufs)
      if (b)
        n += b->n;
    return n;
  }
  // called between calls only (no kernel of this engine is in flight)
  void trim_plans()
  {
    while (planOrder.size() > kMaxPlans) {
      auto it = plans.find(planOrder.front());
      if (it != plans.end()) {
        if (it->second->tables.p)
          (void)hipFree(it->second->tables.p);
        plans.erase(it);
      }
      planOrder.erase(planOrder.begin());
    }
  }
};

struct EnginePool {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::unique_ptr<Engine>> all;

  // an idle engine of the device current on this thread; nullptr: no device / initialisation failed
  Engine* acquire()
  {
    int ndev = 0, dev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
      fprintf(stderr, "[sperr_hip] no HIP device available; this library has no CPU fallback\n");
      return nullptr;
    }
    if (hipGetDevice(&dev) != hipSuccess)
      return nullptr;
    static const size_t perDev = getenv("SPERR_HIP_ENGINES_PER_DEVICE")
                                     ? (size_t)std::max(1, atoi(getenv("SPERR_HIP_ENGINES_PER_DEVICE")))
                                     : 4;
    std::unique_lock<std::mutex> lock(mu);
    for (;;) {
      size_t have = 0;
      for (auto& e : all)
        if (e->dev == dev) {
          have++;
          if (!e->busy) {
            e->busy = true;
            return e.get();
          }
        }
      if (have < perDev) {
        auto e = std::make_unique<Engine>();
        e->dev = dev;
        e->busy = true;
        Engine* raw = e.get();
        all.push_back(std::move(e));
        lock.unlock();
        if (raw->init()) {
          lock.lock();
          raw->busy = false;
          raw->dev = -1;   // never handed out again
          cv.notify_all();
          return nullptr;
        }
        return raw;
      }
      cv.wait(lock);
    }
  }
  void release(Engine* e)
  {
    {
      std::lock_guard<std::mutex> lock(mu);
      e->busy = false;
    }
    cv.notify_all();
  }
  // engines of a device that are on a call right now (the caller's own included)
  size_t busy_on(int dev)
  {
    std::lock_guard<std::mutex> lock(mu);
    size_t n = 0;
    for (auto& e : all)
      n += (e->dev == dev && e->busy) ? 1 : 0;
    return n;
  }
};

EnginePool g_pool;

// a call's hold on an engine; also makes the engine's profiler the calling thread's
struct Lease {
  Engine* e;
  Lease() : e(g_pool.acquire())
  {
    if (e) {
      std::lock_guard<std::mutex> lock(g_prof_cfg_mu);
      e->prof.on = g_prof_on;
      e->prof.only = g_prof_only;
      t_prof = &e->prof;
    }
  }
  ~Lease()
  {
    if (e) {
      t_prof = nullptr;
      e->trim_plans();
      g_pool.release(e);
    }
  }
  Lease(const Lease&) = delete;
  Lease& operator=(const Lease&) = delete;
};

// How much workspace a call may use: 80 % of what is free plus what the engine already holds --
// or SPERR_HIP_ARENA_MAX_MB when that is less (a soft cap: it bounds how many chunks of a batch
// are in flight together, one chunk is always allowed; for hosts that share the device, and for
// tests that want a volume NOT to fit).
size_t arena_cap_env()
{
  const char* v = getenv("SPERR_HIP_ARENA_MAX_MB");   // read per call: a test changes it
  return v ? (size_t)std::max(1ll, atoll(v)) << 20 : ~size_t(0);
}
// SPERR_HIP_ARENA_DEBUG=1: every array of a batch's workspace with its size, on stderr
bool arena_debug()
{
  static const bool on = getenv("SPERR_HIP_ARENA_DEBUG") && atoi(getenv("SPERR_HIP_ARENA_DEBUG")) != 0;
  return on;
}
size_t arena_room(size_t have, size_t freeNow)
{
  return std::min((size_t)((freeNow + have) * 0.80), arena_cap_env());
}
size_t arena_budget(size_t have, size_t freeNow)
{
  return std::min(std::max(have, (size_t)((freeNow + have) * 0.80)), arena_cap_env());
}

// bytes of workspace one chunk of this shape needs
struct EncSizes {
  size_t streamWords, maskWords, perChunk;
};

uint64_t rounded_budget(uint64_t raw)
{
  if (raw == 0)
    return ~0ull;
  while (raw % 8)
    raw++;
  return raw;
}

uint64_t max_payload_bits(const ShapePlan& P, uint64_t raw_budget)
{
  // every plane of a uint64 coded in full: a sample gives at most one bit per plane plus its birth
  // test and its sign, a set one test per plane
  const uint64_t unlimited = (uint64_t)P.N * 66ull + (uint64_t)P.ht.nsets * 64ull + 64ull;
  const uint64_t b = rounded_budget(raw_budget);
  return std::min(b, unlimited);
}

// ------------------------------------------------------------------------------------------
// kernels of the container layer
// ------------------------------------------------------------------------------------------

// chunk stream = conditioner header (17) [+ SPECK header (9) + payload]  (SPECK_FLT.cpp:111-124,
// Conditioner.cpp:28-63, SPECK_INT.cpp:284-308) written into the chunk's slot
__global__ void __launch_bounds__(kThreads)
k_write_slot(const CoderState* cst, const EncState* est, const uint64_t* stream,
             size_t streamStride, const uint32_t* globalId, uint8_t* slots, const uint64_t* slotOff,
             uint64_t* lens, uint32_t nvals, int wide_pass)
{
  const uint32_t c = blockIdx.y;
  const CoderState& cs = cst[c];
  if (cs.is_const ? wide_pass : ((int)cs.wide != wide_pass || (!wide_pass && cs.need_retry)))
    return;
  const uint32_t g = globalId[c];
  uint8_t* out = slots + slotOff[g];
  const uint64_t len = cs.stream_len;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    lens[g] = len;
    if (cs.is_const) {
      out[0] = 0x81;
      const uint64_t nval = nvals;
      memcpy(out + 1, &nval, 8);
      memcpy(out + 9, &cs.mean, 8);
    }
    else {
      out[0] = 0x80;
      memcpy(out + 1, &cs.mean, 8);
      memcpy(out + 9, &cs.q, 8);
      out[17] = (uint8_t)cs.nbp;
      memcpy(out + 18, &cs.total_bits, 8);
    }
  }
  if (cs.is_const)
    return;
  (void)est;
  const uint64_t payload = len - 26;
  const uint64_t* w = stream + c * streamStride;
  copy_bytes_wide(out + 26, reinterpret_cast<const uint8_t*>(w), payload,
                  (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// container header (SPERR3D_OMP_C.cpp:163-234) + chunk offsets
// (lens2: bytes of the outlier stream that follows each chunk's SPECK stream, PWE mode)
__global__ void k_container_header(uint8_t* dst, const uint64_t* lens, const uint64_t* lens2,
                                   uint64_t* offs, uint32_t nchunks, uint32_t vx, uint32_t vy,
                                   uint32_t vz, uint32_t cx, uint32_t cy, uint32_t cz,
                                   int is_float, uint64_t* total)
{
  if (blockIdx.x || threadIdx.x)
    return;
  const bool multi = nchunks > 1;
  dst[0] = 0;  // SPERR_VERSION_MAJOR (CMakeLists.txt:5)
  dst[1] = (uint8_t)(0x40 | (is_float ? 0x20 : 0) | (multi ? 0x10 : 0));
  size_t pos = 2;
  const uint32_t v3[3] = {vx, vy, vz};
  memcpy(dst + pos, v3, 12);
  pos += 12;
  if (multi) {
    const uint16_t c3[3] = {(uint16_t)cx, (uint16_t)cy, (uint16_t)cz};
    memcpy(dst + pos, c3, 6);
    pos += 6;
  }
  uint64_t off = pos + 4ull * nchunks;
  for (uint32_t i = 0; i < nchunks; i++) {
    const uint64_t both = lens[i] + lens2[i];
    const uint32_t l = (uint32_t)both;
    memcpy(dst + pos, &l, 4);
    pos += 4;
    offs[i] = off;
    off += both;
  }
  *total = off;
}

__global__ void __launch_bounds__(kThreads)
k_copy_slots(uint8_t* dst, uint64_t dst_cap, const uint8_t* slots, const uint64_t* slotOff,
             const uint64_t* lens, const uint64_t* offs, uint32_t nchunks)
{
  for (uint32_t g = blockIdx.y; g < nchunks; g += gridDim.y) {   // (grid.y is limited to 65535)
    const uint64_t len = lens[g];
    if (offs[g] + len > dst_cap)
      continue;
    const uint8_t* in = slots + slotOff[g];
    uint8_t* out = dst + offs[g];
    copy_bytes_wide(out, in, len, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
  }
}

// outlier streams of one batch (local index b): they follow the chunk's SPECK stream
__global__ void __launch_bounds__(kThreads)
k_copy_slots2(uint8_t* dst, uint64_t dst_cap, const uint8_t* slots2, const uint64_t* slotOff2,
              const uint32_t* gids, const uint64_t* lens, const uint64_t* lens2,
              const uint64_t* offs)
{
  const uint32_t b = blockIdx.y, g = gids[b];
  const uint64_t len = lens2[g], at = offs[g] + lens[g];
  if (len == 0 || at + len > dst_cap)
    return;
  const uint8_t* in = slots2 + slotOff2[b];
  uint8_t* out = dst + at;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = in[i];
}

// first 26 bytes of every chunk stream, gathered for the host
__global__ void k_gather_heads(const uint8_t* container, const uint64_t* offs, const uint64_t* lens,
                               uint8_t* heads, uint32_t nchunks)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks)
    return;
  for (int i = 0; i < 32; i++)
    heads[c * 32 + i] = (i < 26 && (uint64_t)i < lens[c]) ? container[offs[c] + i] : 0;
}

// ------------------------------------------------------------------------------------------
// compression
// ------------------------------------------------------------------------------------------
struct ChunkRef {
  uint32_t gid;
  uint32_t org[3];
};

struct EncBatchBufs {
  EncBuffers eb;
  ChunkGeom* geom;
  uint32_t* gids;
  double* vals;
  size_t valsStride;
  double* strideMean;
  size_t strideMeanStride;
  uint32_t* coef32;
  int8_t* msb;
  size_t bytesPerChunk;
  bool aliased;        // the coder's node arrays, birth records and second list lie over `vals` (carve_enc_coder)
  size_t coderBytes;   // ... and take this many bytes for the batch
};

// the arrays of the integer coder that nothing reads or writes before the quantiser is done: each from
// `first` when it fits there, else from `second` (may be null)
bool carve_enc_coder(Arena& first, Arena* second, const ShapePlan& P, uint32_t B, EncBuffers& e)
{
  const size_t nn = P.dtree.nnodes;
#define TAKE(dst, T, count)                            \
  dst = first.take<T>((size_t)(count));                \
  if (!dst && second)                                  \
    dst = second->take<T>((size_t)(count));            \
  if (!dst)                                            \
    return false;
  e.nodeStride = nn;
  e.bornStride = P.ht.nsets + 8;
  // (the large ones first: what does not fit any more is small)
  TAKE(e.opos, uint64_t, nn * B);
  TAKE(e.chain, uint64_t, nn * B);
  TAKE(e.bornPacked, uint64_t, e.bornStride * B);
  TAKE(e.bornPosLev, uint64_t, e.bornStride * B);
  TAKE(e.lis[1], uint64_t, P.lisEntries * B);
  TAKE(e.E, uint32_t, nn * B);
  TAKE(e.bucket, uint32_t, nn * B);
  TAKE(e.koff, uint32_t, nn * B);
  TAKE(e.leafDesc, uint16_t, nn * B);
  TAKE(e.M, int8_t, nn * B);
#undef TAKE
  return true;
}

// carve the arrays of one batch out of the arena; returns false when it does not fit
// alias: the coder's arrays over the chunk buffer (not in point-wise error mode, whose outlier stage writes the chunk buffer
// while the coder runs: compress_impl)
bool carve_enc(Arena& A, const ShapePlan& P, uint32_t B, uint64_t raw_budget, EncBatchBufs& o, bool alias = true)
{
  const size_t N = P.N, Npad = round_up(N, 256);
  const size_t nn = P.dtree.nnodes;
  const uint64_t payloadBits = max_payload_bits(P, raw_budget);
  const size_t streamWords = (size_t)((payloadBits + 63) / 64) + 4;
  const uint64_t maskBits = std::min<uint64_t>(P.maxPhaseBits, payloadBits + 64);
  const size_t maskWords = (size_t)((maskBits + 63) / 64) + 2;
  EncBuffers& e = o.eb;
  memset(&e, 0, sizeof(e));
  e.tree = P.dtree;
  e.nchunks = B;
#define TAKE(dst, T, count)                                                            \
  dst = A.take<T>((size_t)(count));                                                    \
  if (!dst)                                                                            \
    return false;                                                                      \
  if (arena_debug())                                                                   \
    fprintf(stderr, "[sperr_hip] arena %-18s %10.2f MB\n", #dst, (double)((size_t)(count) * sizeof(T)) / 1048576.0);
  TAKE(e.cst, CoderState, B);
  TAKE(e.st, EncState, B);
  TAKE(o.geom, ChunkGeom, B);
  TAKE(o.gids, uint32_t, B);
  o.valsStride = Npad;
  TAKE(o.vals, double, Npad * B);
  o.strideMeanStride = round_up(std::max<size_t>(P.nstrides, P.N / 4096 + 2), 32);   // (also the PSNR-mode mse partials)
  TAKE(o.strideMean, double, o.strideMeanStride * B);
  e.coefStride = Npad;
  TAKE(o.coef32, uint32_t, Npad * B);
  e.coef = o.coef32;
  e.signStride = Npad / 64;
  uint64_t* sign;
  TAKE(sign, uint64_t, e.signStride * B);
  e.sign = sign;
  e.pixStride = Npad;
  TAKE(o.msb, int8_t, Npad * B);
  e.msb = o.msb;
  TAKE(e.bplane, int8_t, Npad * B);
  // The coder's node arrays, birth records and second list are written after the quantiser has read
  // the chunk buffer for the last time: they lie over it (round 3; 127 of the 421 MB a 256^3 chunk
  // took).  A batch that needs the 64-bit retry transforms its chunks again and gives these arrays
  // memory of their own (Engine::wideScratch, compress_impl).  SPERR_HIP_ENC_ALIAS=0: no overlay.
  static const bool aliasEnv = !(tune_getenv("SPERR_HIP_ENC_ALIAS") && atoi(tune_getenv("SPERR_HIP_ENC_ALIAS")) == 0);
  {
    Arena over;
    over.base = reinterpret_cast<char*>(o.vals);
    over.cap = (aliasEnv && alias) ? Npad * B * sizeof(double) : 0;
    const size_t before = A.used;
    if (!carve_enc_coder(over, &A, P, B, e))
      return false;
    o.aliased = over.used != 0;
    o.coderBytes = over.used + (A.used - before);
    if (arena_debug())
      fprintf(stderr, "[sperr_hip] arena %-18s %10.2f MB, %.2f MB of them over o.vals\n", "coder arrays",
              (double)o.coderBytes / 1048576.0, (double)over.used / 1048576.0);
  }
  e.lisStride = P.lisEntries;
  TAKE(e.lis[0], uint64_t, P.lisEntries * B);
  e.levelOff = P.d_levelOff;
  e.nListTiles = P.nListTiles;
  e.tileLevel = P.d_tileLevel;
  e.tileStart = P.d_tileStart;
  e.levelFirstTile = P.d_levelFirstTile;
  e.levelNumTiles = P.d_levelNumTiles;
  e.tileStride = round_up(P.nListTiles, 32);
  TAKE(e.tileBits, uint64_t, e.tileStride * B);
  TAKE(e.tileSurv, uint32_t, e.tileStride * B);
  TAKE(e.tileBitsOff, uint64_t, e.tileStride * B);
  TAKE(e.tileSurvOff, uint32_t, e.tileStride * B);
  e.iRoots = P.d_iRoots;
  e.iLevels = P.ht.iLevels;
  e.levelSlot = P.d_levelSlot;
  e.slotLevel = P.d_slotLevel;
  e.nSlots = P.nSlots;
  e.maskWords = (uint32_t)maskWords;
  e.maskStride = (size_t)P.nSlots * maskWords;
  TAKE(e.mask, uint64_t, std::max<size_t>(e.maskStride, 1) * B);
  e.prefWords = (uint32_t)((maskWords + 3) / 4);
  e.prefStride = (size_t)P.nSlots * e.prefWords;
  TAKE(e.maskPrefix, uint32_t, std::max<size_t>(e.prefStride, 1) * B);
  e.nPixTiles = P.nPixTiles;
  e.pixCntStride = (size_t)kMaxPlanes * 2 * P.nPixTiles;
  TAKE(e.pixCnt, uint32_t, e.pixCntStride * B);
  TAKE(e.pixOff, uint32_t, e.pixCntStride * B);
  e.streamStride = streamWords;
  TAKE(e.stream, uint64_t, streamWords * B);
#undef TAKE
  return true;
}

// bytes carve_enc takes for a batch of B chunks.  Probed with the batch's own B: which coder arrays
// find room over the chunk buffer (carve_enc_coder, first fit) depends on the 256-byte rounding of
// B arrays, so B times the one-chunk figure can fall short for small chunks.
size_t enc_bytes_for(const ShapePlan& P, uint32_t B, uint64_t raw_budget, bool alias = true)
{
  Arena probe;
  probe.base = reinterpret_cast<char*>(uintptr_t(4096));  // never dereferenced: size probe only
  probe.cap = ~size_t(0) / 2;
  EncBatchBufs tmp;
  carve_enc(probe, P, std::max<uint32_t>(B, 1), raw_budget, tmp, alias);
  return probe.used;
}
size_t enc_bytes_per_chunk(const ShapePlan& P, uint64_t raw_budget, bool alias = true)
{
  return enc_bytes_for(P, 1, raw_budget, alias);
}

int reset_enc_pass(hipStream_t st, const EncBatchBufs& bb, uint32_t B)
{
  const EncBuffers& e = bb.eb;
  HIP_CHECK(hipMemsetAsync(e.bplane, 0xff, e.pixStride * B, st));
  HIP_CHECK(hipMemsetAsync(e.M, 0xff, e.nodeStride * B, st));
  HIP_CHECK(hipMemsetAsync(e.mask, 0, std::max<size_t>(e.maskStride, 1) * B * sizeof(uint64_t), st));
  HIP_CHECK(hipMemsetAsync(e.stream, 0, e.streamStride * B * sizeof(uint64_t), st));
  return 0;
}

// the transform starts with the full-size x pass followed by the full-size y pass (every dyadic
// shape; wavelet-packet shapes start along z), and the rows fit the fused kernel's LDS tile
bool fuse_xy(const ShapePlan& P)
{
  if (P.fwd.size() < 2 || P.fwd[0].axis != 0 || P.fwd[1].axis != 1)
    return false;
  for (int a = 0; a < 3; a++)
    if (P.fwd[0].region[a] != P.dims[a] || P.fwd[1].region[a] != P.dims[a])
      return false;
  return lift_xy_applicable(P.dims);
}

// ... and the full-size z pass follows (every dyadic shape): all three in one kernel
bool fuse_xyz(const ShapePlan& P)
{
  if (!fuse_xy(P) || P.fwd.size() < 3 || P.fwd[2].axis != 2)
    return false;
  for (int a = 0; a < 3; a++)
    if (P.fwd[2].region[a] != P.dims[a])
      return false;
  return lift_xyz_applicable(P.dims);
}

// LiftFuse of pass k (xform.h): the samples of its region that no LATER pass of the forward order
// touches.  The later passes' regions are boxes at the origin; when one of them contains all the
// others (dyadic plans: the next pass; wavelet-packet plans: the full-size x pass for every z pass)
// the samples are those outside it.  Returns 0 when there are none, 1 when there are (inner = that
// box), -1 when the later regions are not nested (no plan of build_plan is like that).
int pass_fuse(const ShapePlan& P, size_t k, uint32_t inner[3])
{
  const LiftPass& ps = P.fwd[k];
  inner[0] = inner[1] = inner[2] = 0;
  for (size_t j = k + 1; j < P.fwd.size(); j++)
    for (int a = 0; a < 3; a++)
      inner[a] = std::max(inner[a], P.fwd[j].region[a]);
  bool nested = k + 1 >= P.fwd.size();
  for (size_t j = k + 1; j < P.fwd.size(); j++)
    nested = nested || (P.fwd[j].region[0] == inner[0] && P.fwd[j].region[1] == inner[1] &&
                        P.fwd[j].region[2] == inner[2]);
  if (!nested)
    return -1;
  bool covers = true;
  for (int a = 0; a < 3; a++)
    covers = covers && inner[a] >= ps.region[a];
  return covers ? 0 : 1;
}

// Can the lifting passes collect the largest coefficient / dequantise on the way?  Not when the
// fused x-y kernel of the finest level would have to (slices: their next pass is a coarser level).
bool plan_fusable(const ShapePlan& P)
{
  if (P.fwd.empty())
    return false;
  static const bool on = !(tune_getenv("SPERR_HIP_LIFT_FUSE") && atoi(tune_getenv("SPERR_HIP_LIFT_FUSE")) == 0);
  if (!on)
    return false;
  uint32_t inner[3];
  for (size_t k = 0; k < P.fwd.size(); k++)
    if (pass_fuse(P, k, inner) < 0)
      return false;
  if (fuse_xy(P) && (pass_fuse(P, 0, inner) != 0 || pass_fuse(P, 1, inner) != 0))
    return false;
  return true;
}

// conditioner + forward transform of one batch: volume -> bb.vals (mean, constness, largest magnitude
// in CoderState).  Run once per batch -- and once more before a 64-bit retry when the coder's arrays
// lay over the chunk buffer (carve_enc): the same launches on the same input give the same bits.
template <typename T>
int float_stages(hipStream_t ss, const ShapePlan& P, EncBatchBufs& bb, uint32_t nb, const uint32_t cd[3],
                 const T* d_src, VolDesc vd, bool orgAligned, bool wantRange)
{
  EncBuffers& e = bb.eb;
  // the first lifting pass covers the whole chunk: it reads the volume itself (gather, widen,
  // subtract the mean); chunks too small to be transformed take the plain gather kernel
  const bool fuse = !P.fwd.empty();
  const int io = std::is_same<T, float>::value ? 1 : 2;
  if (launch_condition<T>(ss, d_src, vd, bb.geom, nb, cd, P.nstrides, bb.strideMean, bb.strideMeanStride,
                          bb.vals, bb.valsStride, e.cst, !fuse, wantRange, orgAligned))
    return -1;
  size_t k0 = 0;
  // the passes after which samples have their final value also collect the largest magnitude
  // (src/SPECK_FLT.cpp:282-301): no pass over the coefficients of its own
  const bool fuseMax = plan_fusable(P);
  if (fuse_xyz(P)) {   // the three full-size passes in one kernel, straight from the volume
    LiftFuse lf;
    if (fuseMax && pass_fuse(P, 2, lf.inner) > 0)
      lf.mode = 1;
    if (launch_lift_xyz(ss, true, bb.vals, bb.valsStride, nb, cd, e.cst, io, const_cast<T*>(d_src), vd,
                        bb.geom, &lf))
      return -1;
    k0 = 3;
  }
  else if (fuse_xy(P)) {   // the full-size x and y passes in one kernel, straight from the volume
    if (launch_lift_xy(ss, true, bb.vals, bb.valsStride, nb, cd, e.cst, io, const_cast<T*>(d_src), vd,
                       bb.geom))
      return -1;
    k0 = 2;
  }
  for (size_t k = k0; k < P.fwd.size(); k++) {
    const LiftPass& ps = P.fwd[k];
    LiftFuse lf;
    if (fuseMax && pass_fuse(P, k, lf.inner) > 0)
      lf.mode = 1;
    if (launch_lift(ss, true, bb.vals, bb.valsStride, nb, cd, ps.axis, ps.region, e.cst, k == 0 ? io : 0,
                    const_cast<T*>(d_src), vd, bb.geom, &lf))
      return -1;
  }
  return 0;
}

// before the 64-bit retry of a batch whose coder arrays lay over the chunk buffer: those arrays move
// to memory of their own and the buffer gets its DWT coefficients back
// the coder's arrays off the chunk buffer, into scratch memory of the engine (64-bit magnitudes are
// about to live in the buffer)
int unalias_coder(hipStream_t ss, Engine& E, const ShapePlan& P, EncBatchBufs& bb, uint32_t nb)
{
  HIP_CHECK(hipStreamSynchronize(ss));   // (the scratch buffer may grow: nothing of this stream may still use it)
  if (E.wideScratch.ensure(bb.coderBytes + 4096))
    return -1;
  Arena W;
  W.base = static_cast<char*>(E.wideScratch.p);
  W.cap = E.wideScratch.n;
  if (!carve_enc_coder(W, nullptr, P, nb, bb.eb))
    return -1;
  bb.aliased = false;
  return 0;
}

template <typename T>
int wide_retry_prepare(hipStream_t ss, Engine& E, const ShapePlan& P, EncBatchBufs& bb, uint32_t nb,
                       const uint32_t cd[3], const T* d_src, VolDesc vd, bool orgAligned, bool wantRange)
{
  if (unalias_coder(ss, E, P, bb, nb))
    return -1;
  g_dbg_counter[0]++;
  return float_stages<T>(ss, P, bb, nb, cd, d_src, vd, orgAligned, wantRange);
}

// PSNR mode (src/SPECK_FLT.cpp:268-279,431-435): per chunk q = 2 sqrt(3 t), t = range^2 10^(-psnr/10),
// divided by 2^(1/4) until the estimated quantisation error is at most t.  The libm calls run on
// the host (the same libm the reference uses), the error estimate on the device with the
// reference's summation order; chunks whose largest coefficient needs more than 32 bits are
// flagged for the 64-bit pass (SPECK_FLT.cpp:324-337).
int psnr_q_search(hipStream_t st, const ShapePlan& P, EncBatchBufs& bb, uint32_t nb, double psnr)
{
  EncBuffers& e = bb.eb;
  std::vector<CoderState> hc(nb);
  HIP_CHECK(hipMemcpyAsync(hc.data(), e.cst, nb * sizeof(CoderState), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  std::vector<double> tmse(nb, 0.0);
  bool any = false;
  for (uint32_t i = 0; i < nb; i++) {
    CoderState& c = hc[i];
    c.mse_active = 0;
    if (c.is_const)
      continue;
    const double vmax = order_key_value(c.vmaxKey), vmin = -order_key_value(c.vnegmaxKey);
    const double range = (vmax - c.mean) - (vmin - c.mean);   // of the conditioned samples
    tmse[i] = (range * range) * std::pow(10.0, -psnr / 10.0);
    c.q = 2.0 * std::sqrt(tmse[i] * 3.0);
    if (!(c.q > 0.0))
      return -1;   // (the reference asserts q > 0)
    c.mse_active = 1;
    any = true;
  }
  const double step = std::exp2(0.25);
  while (any) {
    HIP_CHECK(hipMemcpyAsync(e.cst, hc.data(), nb * sizeof(CoderState), hipMemcpyHostToDevice, st));
    if (launch_mse(st, bb.vals, bb.valsStride, nb, P.N, bb.strideMean, bb.strideMeanStride, e.cst))
      return -1;
    std::vector<CoderState> got(nb);
    HIP_CHECK(hipMemcpyAsync(got.data(), e.cst, nb * sizeof(CoderState), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    any = false;
    for (uint32_t i = 0; i < nb; i++) {
      if (!hc[i].mse_active)
        continue;
      if (got[i].mse > tmse[i]) {
        hc[i].q /= step;   // four adjustments halve q
        any = true;
      }
      else
        hc[i].mse_active = 0;
    }
  }
  for (uint32_t i = 0; i < nb; i++) {
    CoderState& c = hc[i];
    c.mse_active = 0;
    c.wide = 0;
    c.need_retry = 0;
    if (c.is_const)
      continue;
    const double m = c.maxabs / c.q;
    if (!(m < 9.3e18))
      return -1;   // llrint would raise FE_INVALID (SPECK_FLT.cpp:325-327)
    c.need_retry = std::llrint(m) > (long long)0xffffffffll ? 1u : 0u;
  }
  HIP_CHECK(hipMemcpyAsync(e.cst, hc.data(), nb * sizeof(CoderState), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));   // hc goes out of scope
  return 0;
}

// diagnostics: SPERR_HIP_TIMING=1 prints host-side wall-clock marks of the PWE stages
struct HostMarks {
  bool on = getenv("SPERR_HIP_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char* what, hipStream_t st)
  {
    if (!on)
      return;
    (void)hipStreamSynchronize(st);
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sperr_hip] %-28s %8.2f ms\n", what,
            std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// PWE mode (src/SPECK_FLT.cpp:280-281): q = 1.5 tol for every chunk; chunks whose largest
// coefficient needs more than 32 bits are flagged for the 64-bit pass (SPECK_FLT.cpp:324-337)
// (`hc`: the caller's, alive until the call's last wait -- the upload at the end is not waited for: round 6, one of the
//  four host round trips per batch that went, see pwe_stage_finish)
int pwe_q_setup(hipStream_t st, EncBatchBufs& bb, uint32_t nb, double tol, std::vector<CoderState>& hc, bool* anyWide = nullptr)
{
  EncBuffers& e = bb.eb;
  hc.assign(nb, CoderState{});
  HIP_CHECK(hipMemcpyAsync(hc.data(), e.cst, nb * sizeof(CoderState), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  for (uint32_t i = 0; i < nb; i++) {
    CoderState& c = hc[i];
    c.wide = 0;
    c.need_retry = 0;
    if (c.is_const)
      continue;
    c.q = 1.5 * tol;
    const double m = c.maxabs / c.q;
    if (!(m < 9.3e18))
      return -1;   // llrint would raise FE_INVALID (SPECK_FLT.cpp:325-327)
    c.need_retry = std::llrint(m) > (long long)0xffffffffll ? 1u : 0u;
    if (anyWide && c.need_retry)
      *anyWide = true;   // (this mode chooses the width before coding: nothing else sets the flag, k_enc_finalize)
  }
  HIP_CHECK(hipMemcpyAsync(e.cst, hc.data(), nb * sizeof(CoderState), hipMemcpyHostToDevice, st));
  return 0;
}

// outlier streams of one batch, kept on the device until the container is assembled
struct PweKeep {
  void* mem = nullptr;
  uint32_t nb = 0;
  uint32_t* gids = nullptr;
  uint64_t* slotOff = nullptr;
  uint8_t* slots = nullptr;
  std::vector<uint64_t> off2;   // host copy of slotOff: uploaded without a wait, lives until the container is out
};
// (the memory belongs to the engine, Engine::pweBufs, and is reused by later calls: a hipFree per
//  batch waits for every stream of the device, which stalls the other workers of the chunk farm)
struct PweKeepList {
  std::vector<PweKeep> v;
};

// list storage of the 1D coder: level l holds at most 2^l runs, and never more than `most`
void speck1d_level_offsets(OutlierBufs& ob, uint32_t N, uint64_t most)
{
  ob.nlists = (uint32_t)spk::num_of_partitions(N) + 1;
  uint64_t off = 0;
  for (uint32_t l = 0; l <= (uint32_t)kO1MaxLevels; l++) {
    ob.levelOff[l] = (uint32_t)std::min<uint64_t>(off, 0xffffffffull);
    if (l < ob.nlists)
      off += l < 40 ? std::min<uint64_t>(1ull << l, most) : most;
  }
  ob.runStride = round_up((size_t)off + 64, 64);
}

// PWE mode, after the integer coder (src/SPECK_FLT.cpp:461-486): rebuild the values the decoder
// will see (inverse quantiser + inverse transform, in the chunk buffer), compare them with the
// conditioned input, and code every error above the tolerance with the 1D coder.
// The stage in two halves (round 5): `begin` -- the reconstruction, the first outlier pass and its read-back, enqueued,
// not waited for -- needs nothing of the 3D coder, only the quantiser's coefficients, so it can run on a stream of its
// own BESIDE the coder (compress_impl); `finish` waits for it and does the rest.
struct PweStage {
  OutlierBufs ob;
  std::vector<OutlierChunk> hoc;
  std::vector<ChunkGeom> bricks;
  HostMarks hm;
};
template <typename T>
int pwe_stage_begin(hipStream_t st, Engine& E, const ShapePlan& P, EncBatchBufs& bb, uint32_t nb,
                    const T* d_src, VolDesc vd, const uint32_t cd[3], double tol, bool anyWide, PweStage& S)
{
  EncBuffers& e = bb.eb;
  HostMarks& hm = S.hm;
  OutlierBufs& ob = S.ob;
  std::vector<OutlierChunk>& hoc = S.hoc;
  std::vector<ChunkGeom>& bricks = S.bricks;
  if (E.pweLastStream) {   // (a second batch of one call: the first one's stage is not waited for at its end any more)
    HIP_CHECK(hipStreamSynchronize(E.pweLastStream));
    E.pweLastStream = nullptr;
  }
  hm.mark("(3D coder done)", st);
  // What the decoder will reconstruct, in the conditioned domain (src/SPECK_FLT.cpp:461-486).  Round 5: by the decoder's
  // own kernels where the plan allows it -- the coarser levels in a compact buffer of their box, dequantising as they
  // load (LiftFuse mode 2), the finest level by k_lift_xyz_inv writing doubles into the chunk buffer as if it were a
  // volume of bricks (a chunk's offset rides in org[0]; no mean added) -- instead of an inverse quantiser pass and
  // fifteen per-axis passes over the whole chunk (14.9 of the 56 ms a 1024^3 volume took to compress in this mode).
  static const bool fusedEnv = !(tune_getenv("SPERR_HIP_PWE_FUSED_INV") && atoi(tune_getenv("SPERR_HIP_PWE_FUSED_INV")) == 0);
  bricks.assign(nb, ChunkGeom{});   // (lives until the stage's next wait for the stream)
  const bool fused = fusedEnv && !anyWide && fuse_xyz(P) && plan_fusable(P) && P.fwd.size() >= 3 &&
                     (uint64_t)nb * bb.valsStride <= 0xffffffffull;
  if (fused) {
    uint32_t cbox[3] = {1, 1, 1};
    for (size_t k = 3; k < P.fwd.size(); k++)
      for (int a = 0; a < 3; a++)
        cbox[a] = std::max(cbox[a], P.fwd[k].region[a]);
    const size_t cstride = round_up((size_t)cbox[0] * cbox[1] * cbox[2], 64);
    const size_t geomOff = round_up((size_t)nb * cstride * 8, 256);
    if (E.pweBox.ensure(geomOff + (size_t)nb * sizeof(ChunkGeom) + 256))
      return -1;
    double* cvals = static_cast<double*>(E.pweBox.p);
    ChunkGeom* d_bricks = reinterpret_cast<ChunkGeom*>(static_cast<char*>(E.pweBox.p) + geomOff);
    for (uint32_t i = 0; i < nb; i++) {
      bricks[i].org[0] = (uint32_t)((size_t)i * bb.valsStride);
      bricks[i].org[1] = bricks[i].org[2] = 0;
    }
    HIP_CHECK(hipMemcpyAsync(d_bricks, bricks.data(), nb * sizeof(ChunkGeom), hipMemcpyHostToDevice, st));
    auto fuse = [&](size_t k, LiftFuse& lf) {
      if (pass_fuse(P, k, lf.inner) > 0) {
        lf.mode = 2;
        lf.coef = bb.coef32;
        lf.coefStride = e.coefStride;
        lf.sign = e.sign;
        lf.signStride = e.signStride;
      }
      lf.bufx = cbox[0];
      lf.bufy = cbox[1];
    };
    for (size_t k = P.fwd.size(); k-- > 3;) {
      const LiftPass& ps = P.fwd[k];
      LiftFuse lf;
      fuse(k, lf);
      if (launch_lift(st, false, cvals, cstride, nb, cd, ps.axis, ps.region, e.cst, 0, nullptr, vd, bb.geom, &lf))
        return -1;
    }
    LiftFuse lf;
    fuse(2, lf);
    lf.noMean = 1;
    const VolDesc brickVol{{cd[0], cd[1], cd[2]}};   // (rows of cx samples, slices of cx * cy: a brick)
    if (launch_lift_xyz(st, false, cvals, cstride, nb, cd, e.cst, 2, bb.vals, brickVol, d_bricks, &lf))
      return -1;
  }
  else {
    if (launch_inv_quantize(st, false, bb.coef32, e.coefStride, e.sign, e.signStride, nb, P.N, bb.vals,
                            bb.valsStride, e.cst) ||
        launch_inv_quantize(st, true, bb.vals, bb.valsStride, e.sign, e.signStride, nb, P.N, bb.vals,
                            bb.valsStride, e.cst))
      return -1;
    for (size_t k = P.fwd.size(); k-- > 0;) {
      const LiftPass& ps = P.fwd[k];
      if (launch_lift(st, false, bb.vals, bb.valsStride, nb, cd, ps.axis, ps.region, e.cst, 0, nullptr,
                      vd, bb.geom))
        return -1;
    }
  }
  hm.mark("inverse path", st);
  memset(&ob, 0, sizeof(ob));
  ob.nchunks = nb;
  ob.N = P.N;
  ob.nw = (P.N + 63) / 64;
  ob.wordStride = round_up((size_t)ob.nw + 2, 32);
  {
    const size_t bytes = round_up(nb * sizeof(OutlierChunk), 256) + (size_t)nb * ob.wordStride * (4 * 8 + 2 * 4) + 4096;
    if (E.outlFixed.ensure(bytes))
      return -1;
    Arena A;
    A.base = static_cast<char*>(E.outlFixed.p);
    A.cap = E.outlFixed.n;
    ob.oc = A.take<OutlierChunk>(nb);
    ob.lip = A.take<uint64_t>(nb * ob.wordStride);
    ob.signMask = A.take<uint64_t>(nb * ob.wordStride);
    ob.maskGE = A.take<uint64_t>(nb * ob.wordStride);
    ob.maskEQ = A.take<uint64_t>(nb * ob.wordStride);
    ob.outPre = A.take<uint32_t>(nb * ob.wordStride);
    ob.cpos = A.take<uint32_t>(nb * ob.wordStride);
    if (!ob.oc || !ob.lip || !ob.signMask || !ob.maskGE || !ob.maskEQ || !ob.outPre || !ob.cpos)
      return -1;
  }
  HIP_CHECK(hipMemsetAsync(ob.oc, 0, nb * sizeof(OutlierChunk), st));
  ob.kStride = 1;
  if (launch_outlier_scan<T>(st, 0, d_src, vd, bb.geom, cd, bb.vals, bb.valsStride, e.cst, tol, ob))
    return -1;
  hoc.assign(nb, OutlierChunk{});
  {   // (into pinned memory: the call returns at once and the 3D coder can be enqueued meanwhile)
    const size_t bytes = nb * sizeof(OutlierChunk);
    if (E.pweHostBytes < bytes) {
      if (E.pweHost)
        (void)hipHostFree(E.pweHost);
      E.pweHost = nullptr;
      E.pweHostBytes = 0;
      HIP_CHECK(hipHostMalloc(&E.pweHost, round_up(bytes, 4096), hipHostMallocDefault));
      E.pweHostBytes = round_up(bytes, 4096);
    }
    HIP_CHECK(hipMemcpyAsync(E.pweHost, ob.oc, bytes, hipMemcpyDeviceToHost, st));
  }
  return 0;
}

template <typename T>
int pwe_stage_finish(hipStream_t st, Engine& E, const ShapePlan& P, EncBatchBufs& bb, uint32_t nb,
                     const T* d_src, VolDesc vd, const uint32_t cd[3], double tol,
                     uint64_t* d_lens2, PweKeepList& keep, PweStage& S)
{
  EncBuffers& e = bb.eb;
  HostMarks& hm = S.hm;
  OutlierBufs& ob = S.ob;
  std::vector<OutlierChunk>& hoc = S.hoc;
  HIP_CHECK(hipStreamSynchronize(st));
  memcpy(hoc.data(), E.pweHost, nb * sizeof(OutlierChunk));
  hm.mark("outlier pass 0", st);
  bool any = false;
  for (auto& o : hoc) {
    if (!o.flagged)
      continue;
    any = true;
    // the integer type the reference keeps the magnitudes in comes from the largest ERROR, not
    // from the largest magnitude (Outlier_Coder.cpp:82-100); magnitudes wrap to that width
    double maxerr;
    memcpy(&maxerr, &o.maxErrKey, 8);
    if (!(maxerr < 9.2e18))
      return -1;
    const long long mi = std::llrint(maxerr);
    o.widthMask = mi <= 0xffll ? 0xffull : mi <= 0xffffll ? 0xffffull : mi <= 0xffffffffll ? 0xffffffffull : ~0ull;
  }
  if (!any)
    return 0;   // (d_lens2 stays zero for these chunks)
  // Round 6: everything from here to the 1D coder's result is enqueued behind ONE wait.  Until then the host came
  // back after the counting pass (for the number of outliers: the arrays' size), after the compaction (for the
  // largest magnitude: the stream's size) and after the stream's copy -- a device call went back to the host a dozen
  // times per batch, and the farm's pipeline (two workers a device) ran at the rate of those round trips.  The first
  // pass has counted the samples beyond the tolerance (`flagged`: at least the outliers that survive the wrap to
  // the integer width, Outlier_Coder.cpp:82-100) and found the largest error: both bounds are known now.
  uint32_t kmax = 0;
  int maxPlanes = 1;
  for (auto& o : hoc) {
    if (!o.flagged)
      continue;
    kmax = std::max(kmax, o.flagged);
    double maxerr;
    memcpy(&maxerr, &o.maxErrKey, 8);
    // (a magnitude is llrint(error / tol) cut to the width: below both)
    const double mq = std::min(maxerr / tol + 2.0, 1.8e19);
    const unsigned long long bound = std::min<unsigned long long>(o.widthMask, (unsigned long long)mq);
    maxPlanes = std::max(maxPlanes, 64 - __builtin_clzll(bound | 1ull));
  }
  HIP_CHECK(hipMemcpyAsync(ob.oc, hoc.data(), nb * sizeof(OutlierChunk), hipMemcpyHostToDevice, st));
  if (launch_outlier_scan<T>(st, 1, d_src, vd, bb.geom, cd, bb.vals, bb.valsStride, e.cst, tol, ob))
    return -1;
  ob.kStride = P.N;   // (only the overflow check of the prefix kernel reads it here)
  if (launch_outlier_prefix(st, ob))
    return -1;
  hm.mark("outlier pass 1 + prefix", st);
  ob.kStride = round_up(std::max<size_t>(kmax, 1), 64);
  speck1d_level_offsets(ob, P.N, 2ull * kmax + 2);
  const size_t varFixed = (size_t)nb * (ob.kStride * (4 + 8 + 1 + 1 + 4 + 1) + ob.runStride * 8) + 8192;
  if (E.outlVar.n < varFixed)
    HIP_CHECK(hipStreamSynchronize(st));   // (the buffer grows: nothing enqueued may still point into the old one)
  if (E.outlVar.ensure(varFixed))
    return -1;
  {
    Arena A;
    A.base = static_cast<char*>(E.outlVar.p);
    A.cap = E.outlVar.n;
    ob.mag = A.take<uint64_t>(nb * ob.kStride);
    ob.runs = A.take<uint64_t>(nb * ob.runStride);
    ob.pos = A.take<uint32_t>(nb * ob.kStride);
    ob.posGE = A.take<uint32_t>(nb * ob.kStride);
    ob.sgn = A.take<uint8_t>(nb * ob.kStride);
    ob.msb = A.take<uint8_t>(nb * ob.kStride);
    ob.sgnGE = A.take<uint8_t>(nb * ob.kStride);
    if (!ob.mag || !ob.runs || !ob.pos || !ob.posGE || !ob.sgn || !ob.msb || !ob.sgnGE)
      return -1;
  }
  if (launch_outlier_scan<T>(st, 2, d_src, vd, bb.geom, cd, bb.vals, bb.valsStride, e.cst, tol, ob))
    return -1;
  hm.mark("alloc + outlier pass 2", st);
  // bits of one chunk's stream: every outlier has at most nlists sets above it, each with a
  // sibling, and every one of those (and the value itself) gives at most one bit per plane, plus
  // the value's sign; and never more than every node of the whole tree doing so
  const uint64_t perOutlier = (uint64_t)(2 * ob.nlists + 2) * maxPlanes + 1;
  const uint64_t sparse = (uint64_t)kmax * perOutlier + 2ull * maxPlanes;
  const uint64_t dense = (uint64_t)P.N * (4ull * maxPlanes + 1);
  const uint64_t maxBits = std::min(sparse, dense) + 64;
  ob.streamStride = round_up((size_t)(maxBits / 64) + 4, 32);
  if (E.outlStream.n < (size_t)nb * ob.streamStride * 8 + 256)
    HIP_CHECK(hipStreamSynchronize(st));
  if (E.outlStream.ensure((size_t)nb * ob.streamStride * 8 + 256))
    return -1;
  ob.stream = static_cast<uint64_t*>(E.outlStream.p);
  HIP_CHECK(hipMemsetAsync(ob.stream, 0, (size_t)nb * ob.streamStride * 8, st));
  HIP_CHECK(hipMemsetAsync(ob.lip, 0, (size_t)nb * ob.wordStride * 8, st));
  HIP_CHECK(hipMemsetAsync(ob.maskGE, 0, (size_t)nb * ob.wordStride * 8, st));
  if (launch_speck1d_encode(st, ob))
    return -1;
  HIP_CHECK(hipMemcpyAsync(hoc.data(), ob.oc, nb * sizeof(OutlierChunk), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  hm.mark("1D coder", st);
  std::vector<uint64_t> off2(nb + 1, 0);
  for (uint32_t i = 0; i < nb; i++) {
    if (hoc[i].flagged && hoc[i].count > ob.kStride) {   // (cannot be: count <= flagged)
      fprintf(stderr, "[sperr_hip] outlier coder: more outliers than the first pass counted (chunk %u)\n", i);
      return -1;
    }
    if (hoc[i].error) {
      fprintf(stderr, "[sperr_hip] outlier coder failed (chunk %u, code %u)\n", i, hoc[i].error);
      return -1;
    }
    const uint64_t len = hoc[i].flagged ? 9 + (hoc[i].total_bits + 7) / 8 : 0;
    off2[i + 1] = off2[i] + round_up(len, 16);
  }
  PweKeep K;
  K.nb = nb;
  const size_t headBytes = round_up((size_t)nb * 8, 256) + round_up((size_t)nb * 4, 256);
  if (keep.v.size() >= E.pweBufs.size())
    E.pweBufs.push_back(std::make_unique<DevBuf>());
  DevBuf& kb = *E.pweBufs[keep.v.size()];
  if (kb.ensure(headBytes + off2[nb] + 256))
    return -1;
  K.mem = kb.p;
  keep.v.push_back(K);
  PweKeep& kk = keep.v.back();
  kk.slotOff = reinterpret_cast<uint64_t*>(kk.mem);
  kk.gids = reinterpret_cast<uint32_t*>(static_cast<char*>(kk.mem) + round_up((size_t)nb * 8, 256));
  kk.slots = reinterpret_cast<uint8_t*>(static_cast<char*>(kk.mem) + headBytes);
  kk.off2 = std::move(off2);   // (alive until the container is assembled: compress_impl waits for its stream at the end)
  HIP_CHECK(hipMemcpyAsync(kk.slotOff, kk.off2.data(), nb * 8, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(kk.gids, bb.gids, nb * 4, hipMemcpyDeviceToDevice, st));
  if (launch_outlier_stream_out(st, ob, kk.gids, kk.slots, kk.slotOff, d_lens2))
    return -1;
  E.pweLastStream = st;
  hm.mark("stream out", st);
  return 0;
}

template <typename T>
int pwe_outlier_stage(hipStream_t st, Engine& E, const ShapePlan& P, EncBatchBufs& bb, uint32_t nb,
                      const T* d_src, VolDesc vd, const uint32_t cd[3], double tol,
                      uint64_t* d_lens2, PweKeepList& keep, bool anyWide)
{
  PweStage S;
  return pwe_stage_begin<T>(st, E, P, bb, nb, d_src, vd, cd, tol, anyWide, S) ||
                 pwe_stage_finish<T>(st, E, P, bb, nb, d_src, vd, cd, tol, d_lens2, keep, S)
             ? -1
             : 0;
}

// SPERR_HIP_SLICE_MIXED=0: slices are coded by k_speck2d's quadtree walk (one workgroup) instead of
// the kernels of the 3D coder on the 2D coder's forest
bool slice_forest_enabled()
{
  static const bool on = !(getenv("SPERR_HIP_SLICE_MIXED") && atoi(getenv("SPERR_HIP_SLICE_MIXED")) == 0);
  return on;
}

// ---- 2D slices (sperr_comp_2d / sperr_decomp_2d, src/SPERR_C_API.cpp:7-134): a slice is a
// one-chunk batch of dims (x, y, 1) -- its transform plan already is dwt2d -- coded by the 2D
// coder of speck2d.hip instead of the 3D one
int carve_slice2d(Engine& E, const ShapePlan& P, Speck2dBufs& sb)
{
  memset(&sb, 0, sizeof(sb));
  sb.dx = P.dims[0];
  sb.dy = P.dims[1];
  sb.N = P.N;
  sb.nw = (P.N + 63) / 64;
  sb.nxforms = (uint32_t)spk::num_of_xforms(std::min(P.dims[0], P.dims[1]));
  sb.nlists = (uint32_t)spk::num_of_partitions(std::max(P.dims[0], P.dims[1])) + 1;
  if (sb.dx > 0xffffu || sb.dy > 0xffffu || sb.nlists > (uint32_t)kS2MaxLevels)
    return -1;
  const size_t entries = speck2d_list_entries(sb);
  const size_t words = round_up((size_t)sb.nw + 2, 32);
  const size_t bytes = round_up(entries * 8, 256) + round_up(entries, 256) + 2 * words * 8 +
                       round_up((size_t)P.N * 4, 256) + 1024 + 4096;
  if (E.slice2d.ensure(bytes))
    return -1;
  Arena A;
  A.base = static_cast<char*>(E.slice2d.p);
  A.cap = E.slice2d.n;
  sb.runs = A.take<uint64_t>(entries);
  sb.sval = A.take<int8_t>(entries);
  sb.lip = A.take<uint64_t>(words);
  sb.lsp = A.take<uint64_t>(words);
  sb.fresh = A.take<uint32_t>(P.N);
  sb.prep = A.take<int32_t>(256);
  return (sb.runs && sb.sval && sb.lip && sb.lsp && sb.fresh && sb.prep) ? 0 : -1;
}

// optional 10-byte header {version, flags, u32 dimx, u32 dimy} (SPERR_C_API.cpp:45-83), then the
// chunk stream and its outlier stream
__global__ void k_slice_header(uint8_t* dst, const uint64_t* lens, const uint64_t* lens2,
                               uint64_t* offs, uint32_t vx, uint32_t vy, int is_float,
                               int with_header, uint64_t* total)
{
  if (blockIdx.x || threadIdx.x)
    return;
  uint64_t pos = 0;
  if (with_header) {
    dst[0] = 0;
    dst[1] = (uint8_t)(is_float ? 0x20 : 0);
    const uint32_t d2[2] = {vx, vy};
    memcpy(dst + 2, d2, 8);
    pos = 10;
  }
  offs[0] = pos;
  *total = pos + lens[0] + lens2[0];
}

// Work queued by a call that fails must not outlive the call: the caller's buffers and the engine's
// arena are reused as soon as it returns.  Every exit that is not marked ok waits for the caller's
// stream and the engine's sub-streams first.
struct DrainOnError {
  Engine& E;
  hipStream_t st;
  bool ok = false;
  ~DrainOnError()
  {
    if (ok)
      return;
    (void)hipStreamSynchronize(st);
    for (uint32_t q = 0; q < kSubStreams; q++)
      if (E.sub[q])
        (void)hipStreamSynchronize(E.sub[q]);
    for (uint32_t q = 0; q < kSubStreams; q++) {
      if (E.outlQ[q])
        (void)hipStreamSynchronize(E.outlQ[q]);
      if (E.sideQ[q])
        (void)hipStreamSynchronize(E.sideQ[q]);
    }
    (void)hipGetLastError();
  }
};

template <typename T>
int compress_impl(Engine& E, const T* d_src, const Dims& vol, const Dims& chunkPref, int mode, double quality,
                  uint8_t* d_dst, size_t dst_cap, size_t* dst_len, hipStream_t st, int slice = 0)
{   // slice: 0 = a 3D container; 1 / 2 = one 2D slice without / with the 10-byte header
  // mode 1: fixed rate, `quality` bits per value; mode 2: fixed PSNR and mode 3: fixed point-wise
  // error, every bit plane is coded
  const bool rate = mode == 1;
  const double bpp = rate ? quality : 0.0;
  DrainOnError drainGuard{E, st};
  Dims cdim;
  for (int a = 0; a < 3; a++)  // SPERR3D_OMP_C.cpp:23-30
    cdim[a] = std::min(std::max<size_t>(1, chunkPref[a]), vol[a]);
  const auto chunks = chunk_volume(vol, cdim);
  const uint32_t nchunks = (uint32_t)chunks.size();
  for (int a = 0; a < 3; a++)
    if (vol[a] > 0xffffffffull || cdim[a] > 0xffff)
      return -1;

  // group chunks by shape, keeping chunk order inside a group.  (key: extents + part.)  In fixed-rate
  // mode a group of 64 and more chunks is cut into parts (four: round 5, 16 chunks each of the bench volume -- three
  // before; five and more fall under the sixteen chunks the capped grids want) that run side by side like the shape
  // groups of a ragged volume do: the per-plane chains of small launches of one part overlap the
  // bandwidth-bound kernels of the other (SPERR_HIP_ENC_PARTS=1: one part)
  using GKey = std::array<size_t, 4>;
  std::map<GKey, std::vector<ChunkRef>> groups;
  {
    std::map<Dims, uint32_t> count, seen;
    for (uint32_t i = 0; i < nchunks; i++)
      count[Dims{chunks[i][1], chunks[i][3], chunks[i][5]}]++;
    static const uint32_t partsEnv = getenv("SPERR_HIP_ENC_PARTS") ? (uint32_t)std::max(1, atoi(getenv("SPERR_HIP_ENC_PARTS"))) : 4u;
    for (uint32_t i = 0; i < nchunks; i++) {
      const auto& c = chunks[i];
      const Dims d{c[1], c[3], c[5]};
      const uint32_t n = count[d], k = seen[d]++;
      static const uint32_t partsMin = tune_getenv("SPERR_HIP_ENC_PARTS_MIN") ? (uint32_t)std::max(2, atoi(tune_getenv("SPERR_HIP_ENC_PARTS_MIN"))) : 64u;
      const uint32_t parts = (mode == 1 && !slice && n >= partsMin && n <= 512) ? std::min<uint32_t>(std::min<uint32_t>(partsEnv, n / 4), kSubStreams) : 1u;
      groups[GKey{c[1], c[3], c[5], (size_t)((uint64_t)k * parts / n)}].push_back(
          {i, {(uint32_t)c[0], (uint32_t)c[2], (uint32_t)c[4]}});
    }
  }

  // (a slice is coded on the 2D coder's forest, the plan with z extent 0; SPERR_HIP_SLICE_MIXED=0: by
  //  k_speck2d's quadtree walk)
  auto planZ = [&](const GKey& d) -> size_t { return slice && slice_forest_enabled() ? 0 : d[2]; };

  // slots for the finished chunk streams
  std::vector<uint64_t> slotOff(nchunks + 1, 0);
  {
    std::vector<uint64_t> slotLen(nchunks, 0);
    for (auto& g : groups) {
      ShapePlan* P = E.plan(g.first[0], g.first[1], planZ(g.first));
      if (!P)
        return -1;
      const uint64_t raw = (uint64_t)(bpp * (double)P->N);
      const uint64_t len = 26 + (max_payload_bits(*P, raw) + 7) / 8;
      for (auto& r : g.second)
        slotLen[r.gid] = round_up(len, 256);
    }
    for (uint32_t i = 0; i < nchunks; i++)
      slotOff[i + 1] = slotOff[i] + slotLen[i];
  }
  if (E.slots.ensure(slotOff[nchunks] + 256))
    return -1;
  const size_t miscBytes = round_up((size_t)nchunks * 8, 256) * 4 + 256;
  if (E.misc.ensure(miscBytes))
    return -1;
  uint64_t* d_slotOff = reinterpret_cast<uint64_t*>(E.misc.p);
  uint64_t* d_lens = d_slotOff + round_up(nchunks, 32);
  uint64_t* d_offs = d_lens + round_up(nchunks, 32);
  uint64_t* d_lens2 = d_offs + round_up(nchunks, 32);   // outlier streams (PWE mode)
  uint64_t* d_total = d_lens2 + round_up(nchunks, 32);
  HIP_CHECK(hipMemsetAsync(d_lens2, 0, (size_t)nchunks * 8, st));
  PweKeepList pweKeep;
  std::deque<std::vector<CoderState>> pweHcKeep;   // (mode 3 is enqueued by one host thread: no lock)
  HIP_CHECK(hipMemcpyAsync(d_slotOff, slotOff.data(), nchunks * 8, hipMemcpyHostToDevice, st));

  VolDesc vd{{vol[0], vol[1], vol[2]}};
  // Shape groups side by side (fixed-rate mode, a volume the chunk size does not divide): every
  // group gets a piece of the arena and a sub-stream of its own, the check for the rare 64-bit
  // retry (one read-back per group) waits until all groups are enqueued.  SPERR_HIP_ENC_GROUPS=0:
  // group after group.
  struct LateGroup {
    ShapePlan* P;
    EncBatchBufs bb;
    uint32_t nb, wblocks;
    uint64_t raw_budget;
    hipStream_t ss;
    std::vector<CoderState> hc;
    std::vector<ChunkGeom> hg;
    std::vector<uint32_t> hid;
    bool orgAligned;
    EncPlanHost ph;   // the plane loop of the group is enqueued after every group's first half (launch_speck_encode_planes)
  };
  std::vector<std::unique_ptr<LateGroup>> late;
  static const bool encGroupsEnv = !(tune_getenv("SPERR_HIP_ENC_GROUPS") && atoi(tune_getenv("SPERR_HIP_ENC_GROUPS")) == 0);
  bool sideBySide = encGroupsEnv && mode == 1 && !slice && groups.size() > 1;
  std::vector<size_t> groupOff;
  if (sideBySide) {
    size_t fr = 0, tot = 0, need = 0;
    HIP_CHECK(hipMemGetInfo(&fr, &tot));
    const size_t budgetBytes = arena_budget(E.arena.n, fr);
    for (auto& g : groups) {
      ShapePlan* P = E.plan(g.first[0], g.first[1], planZ(g.first));
      groupOff.push_back(need);
      need += round_up(enc_bytes_for(*P, (uint32_t)g.second.size(), (uint64_t)(bpp * (double)P->N)) + 4096, 4096);
      sideBySide = sideBySide && g.second.size() <= 256;
    }
    sideBySide = sideBySide && need <= budgetBytes;
    if (sideBySide) {
      if (E.arena.ensure(need))
        return -1;
      HIP_CHECK(hipEventRecord(E.evFork, st));
      for (uint32_t q = 0; q < kSubStreams; q++)
        HIP_CHECK(hipStreamWaitEvent(E.sub[q], E.evFork, 0));
    }
  }
  late.resize(groups.size());
  std::vector<ShapePlan*> groupPlan;   // (looked up here: the plan cache is not for several threads)
  for (auto& g : groups)
    groupPlan.push_back(E.plan(g.first[0], g.first[1], planZ(g.first)));
  auto do_group = [&](uint32_t gi, std::pair<const GKey, std::vector<ChunkRef>>& g) -> int {
    hipStream_t ss = sideBySide ? E.sub[gi % kSubStreams] : st;
    ShapePlan* P = groupPlan[gi];
    const uint64_t raw_budget = (uint64_t)(bpp * (double)P->N);  // SPECK_FLT.cpp:491
    // (point-wise error mode: the coder's arrays get memory of their own -- 127 MB more per 256^3 chunk --, so that the
    //  outlier stage's reconstruction can be written into the chunk buffer while the coder runs, see below)
    static const bool pweOverlapEnv0 = !(tune_getenv("SPERR_HIP_PWE_OVERLAP") && atoi(tune_getenv("SPERR_HIP_PWE_OVERLAP")) == 0);
    const bool encAlias = !(mode == 3 && pweOverlapEnv0 && !sideBySide);
    const size_t per = enc_bytes_per_chunk(*P, raw_budget, encAlias);
    size_t fr = 0, tot = 0;
    HIP_CHECK(hipMemGetInfo(&fr, &tot));
    const size_t budgetBytes = arena_budget(E.arena.n, fr);
    uint32_t B = (uint32_t)std::min<size_t>(g.second.size(), std::max<size_t>(1, budgetBytes / per));
    B = std::min<uint32_t>(B, 256);
    if (sideBySide)
      B = (uint32_t)g.second.size();
    else if (E.arena.ensure(std::max((size_t)B * per, enc_bytes_for(*P, B, raw_budget, encAlias)) + 4096))
      return -1;
    const uint32_t cd[3] = {P->dims[0], P->dims[1], P->dims[2]};
    for (size_t b0 = 0; b0 < g.second.size(); b0 += B) {
      const uint32_t nb = (uint32_t)std::min<size_t>(B, g.second.size() - b0);
      Arena A;
      A.base = static_cast<char*>(E.arena.p) + (sideBySide ? groupOff[gi] : 0);
      A.cap = E.arena.n - (sideBySide ? groupOff[gi] : 0);
      EncBatchBufs bb;
      if (!carve_enc(A, *P, nb, raw_budget, bb, encAlias))
        return -1;
      if (bb.aliased)
        g_dbg_counter[1]++;
      EncBuffers& e = bb.eb;
      std::vector<ChunkGeom> hg(nb);
      std::vector<uint32_t> hid(nb);
      for (uint32_t i = 0; i < nb; i++) {
        const ChunkRef& r = g.second[b0 + i];
        hid[i] = r.gid;
        for (int a = 0; a < 3; a++)
          hg[i].org[a] = r.org[a];
      }
      HIP_CHECK(hipMemcpyAsync(bb.geom, hg.data(), nb * sizeof(ChunkGeom), hipMemcpyHostToDevice, ss));
      HIP_CHECK(hipMemcpyAsync(bb.gids, hid.data(), nb * 4, hipMemcpyHostToDevice, ss));
      HIP_CHECK(hipMemsetAsync(e.cst, 0, nb * sizeof(CoderState), ss));

      // ---- float stages ----
      // the first lifting pass covers the whole chunk: it reads the volume itself (gather, widen,
      // subtract the mean); chunks too small to be transformed take the plain gather kernel
      const bool fuse = !P->fwd.empty();
      const int io = std::is_same<T, float>::value ? 1 : 2;
      bool orgAligned = true;   // (lets the conditioner stream rows with 16-byte loads)
      for (uint32_t i = 0; i < nb; i++)
        orgAligned = orgAligned && hg[i].org[0] % (16 / sizeof(T)) == 0;
      const bool fuseMax = plan_fusable(*P);
      if (float_stages<T>(ss, *P, bb, nb, cd, d_src, vd, orgAligned, mode == 2))
        return -1;
      if (launch_maxabs_q(ss, bb.vals, bb.valsStride, nb, P->N, e.cst, fuseMax))
        return -1;
      if (mode == 2 && psnr_q_search(ss, *P, bb, nb, quality))
        return -1;
      bool pweWide = false;
      if (mode == 3) {
        pweHcKeep.emplace_back();   // (the upload of the chunks' states is not waited for: alive until the call's end)
        if (pwe_q_setup(ss, bb, nb, quality, pweHcKeep.back(), &pweWide))
          return -1;
      }
      if (launch_quantize(ss, false, bb.vals, bb.valsStride, nb, P->N, bb.coef32, e.coefStride,
                          const_cast<uint64_t*>(e.sign), e.signStride, bb.msb, e.pixStride, e.cst))
        return -1;
      // (only now: the coder's arrays may lie over the chunk buffer the quantiser has just read)
      if (reset_enc_pass(ss, bb, nb))
        return -1;
      // Point-wise error mode: what the decoder will reconstruct, and which samples miss the tolerance, follows from
      // the quantiser's coefficients alone -- the first half of the outlier stage (reconstruction, first outlier
      // pass) runs on a stream of its own BESIDE the 3D coder (round 5; behind it, one stream, before: the 1D coder
      // alone is 3.2 of the 11.8 ms a batch of eight chunks took).  Not when the coder's arrays lie over the chunk
      // buffer the reconstruction is written to, and given up when a chunk turns out to need 64-bit coefficients.
      static const bool pweOverlapEnv = !(tune_getenv("SPERR_HIP_PWE_OVERLAP") && atoi(tune_getenv("SPERR_HIP_PWE_OVERLAP")) == 0);
      PweStage pweSt;
      hipStream_t pweQ = E.outlQ[1];
      const bool pweEarly = mode == 3 && pweOverlapEnv && !bb.aliased && !sideBySide && !pweWide && pweQ != nullptr &&
                            E.evPweFork != nullptr && E.evPweJoin != nullptr;
      if (pweEarly) {
        HIP_CHECK(hipEventRecord(E.evPweFork, ss));
        HIP_CHECK(hipStreamWaitEvent(pweQ, E.evPweFork, 0));
        if (pwe_stage_begin<T>(pweQ, E, *P, bb, nb, d_src, vd, cd, quality, false, pweSt))
          return -1;
      }

      // ---- integer coder, 32-bit coefficients ----
      const bool quadWalkGroup = slice && !(P->ht.flags & spk::kTree2D);   // (SPERR_HIP_SLICE_MIXED=0)
      EncPlanHost ph{P->d_initLIS, P->d_initLen, P->d_depthBlocks, P->depthBlockOff, P->ht.nsets};
      // (the census of the pixel passes on a stream of its own beside the pyramid's upper levels: the
      //  decoder's outlier streams and events are idle during a compression call)
      static const bool sideEnv = !(tune_getenv("SPERR_HIP_ENC_SIDE") && atoi(tune_getenv("SPERR_HIP_ENC_SIDE")) == 0);
      if (sideEnv) {
        ph.side = E.sideQ[gi % kSubStreams];
        ph.evFork = E.evOutlFork[gi % kSubStreams];
        ph.evJoin = E.evOutl[gi % kSubStreams];
      }
      // the planes that can hold work are asked of the device before the plane loop is enqueued (speck_enc.h;
      // the decoder's pinned words and events are idle during a compression call).  SPERR_HIP_ENC_BOUND=0: all planes
      static const bool boundEnv = !(tune_getenv("SPERR_HIP_ENC_BOUND") && atoi(tune_getenv("SPERR_HIP_ENC_BOUND")) == 0);
      if (boundEnv && !quadWalkGroup) {
        ph.d_bound = A.take<uint32_t>(64);
        // (a pinned word pair and an event per GROUP: groups gi and gi + kSubStreams share a stream, and every head
        //  is enqueued before any plane loop reads its bounds back -- a ragged volume has up to 4 parts + 7 border
        //  shapes = 11 groups.  Past kSubStreams * kLiveSlots / 2 groups: all planes are launched)
        const uint32_t lane = gi % kSubStreams, turn = gi / kSubStreams;
        if (turn < (uint32_t)kLiveSlots / 2) {
          ph.h_bound = E.liveHost[lane] + 2 * turn;
          ph.evBound = E.liveEv[lane][turn];
        }
        if (!ph.d_bound)
          ph.h_bound = nullptr;
      }
      Speck2dBufs sb;
      const bool quadWalk = quadWalkGroup;
      if (quadWalk) {
        if (carve_slice2d(E, *P, sb))
          return -1;
        sb.coef = bb.coef32;
        sb.sign = const_cast<uint64_t*>(e.sign);
        sb.msb = bb.msb;
        sb.stream = e.stream;
        sb.streamWords = e.streamStride;
        sb.cst = e.cst;
        if (launch_speck2d_encode(ss, sb, raw_budget, rate, false))
          return -1;
      }
      else if (sideBySide && ph.h_bound
                   ? launch_speck_encode_head(ss, e, ph, raw_budget, rate, false)   // (its planes: once every group's first half is enqueued)
                   : launch_speck_encode(ss, e, ph, raw_budget, rate, false))
        return -1;
      const uint32_t wblocks = (uint32_t)std::min<size_t>(4096, (e.streamStride * 8 + kThreads - 1) / kThreads);
      if (!(sideBySide && ph.h_bound))
        LAUNCH_K(k_write_slot, dim3(std::max(1u, wblocks), nb), dim3(kThreads), 0, ss, e.cst, e.st,
                 e.stream, e.streamStride, bb.gids, static_cast<uint8_t*>(E.slots.p), d_slotOff,
                 d_lens, P->N, 0);

      // (point-wise error mode: the outlier stage's second half -- its waits are for its own stream -- while the plane
      //  loop above runs)
      if (pweEarly) {
        if (pwe_stage_finish<T>(pweQ, E, *P, bb, nb, d_src, vd, cd, quality, d_lens2, pweKeep, pweSt))
          return -1;
        // (the stage no longer ends in a wait of the host: what follows on the batch's stream -- the container's
        //  kernels read the outlier streams and their lengths -- waits for it on the device)
        HIP_CHECK(hipEventRecord(E.evPweJoin, pweQ));
        HIP_CHECK(hipStreamWaitEvent(ss, E.evPweJoin, 0));
      }
      // ---- fixed-rate retry with 64-bit coefficients (SPECK_FLT.cpp:530-538) ----
      if (sideBySide) {   // (the read-back is looked at once every group is enqueued)
        std::unique_ptr<LateGroup> L(new LateGroup{P, bb, nb, wblocks, raw_budget, ss, std::vector<CoderState>(nb),
                                                   std::move(hg), std::move(hid), orgAligned, ph});
        late[gi] = std::move(L);        // (a read-back into pageable memory would block the host until
        continue;                       //  this group's stream has drained: it is done further down)
      }
      std::vector<CoderState> hc(nb);
      HIP_CHECK(hipMemcpyAsync(hc.data(), e.cst, nb * sizeof(CoderState), hipMemcpyDeviceToHost, ss));
      HIP_CHECK(hipStreamSynchronize(ss));
      bool retry = false;
      for (auto& c : hc)
        retry |= (c.need_retry != 0);
      if (retry) {
        // the DWT coefficients again when the coder's arrays were written over them, and memory of
        // their own for those arrays (the 64-bit magnitudes live in the chunk buffer)
        if (bb.aliased && wide_retry_prepare<T>(ss, E, *P, bb, nb, cd, d_src, vd, orgAligned, mode == 2))
          return -1;
        if (pweEarly) {   // (cannot be: this mode knows the width before it codes, pwe_q_setup)
          fprintf(stderr, "[sperr_hip] a chunk asked for 64-bit coefficients after its outlier stage had run\n");
          return -1;
        }
        // fixed rate: a finer q for the flagged chunks; PSNR: the same q, coefficients need 64 bits
        if ((rate ? launch_make_q_wide(ss, nb, e.cst) : launch_mark_wide(ss, nb, e.cst)) ||
            reset_enc_pass(ss, bb, nb))
          return -1;
        // 64-bit magnitudes overwrite the DWT coefficients in place (same element size)
        if (launch_quantize(ss, true, bb.vals, bb.valsStride, nb, P->N, bb.vals, bb.valsStride,
                            const_cast<uint64_t*>(e.sign), e.signStride, bb.msb, e.pixStride, e.cst))
          return -1;
        EncBuffers ew = e;
        ew.coef = bb.vals;
        ew.coefStride = bb.valsStride;
        if (quadWalk) {
          sb.coef = bb.vals;
          if (launch_speck2d_encode(ss, sb, raw_budget, rate, true))
            return -1;
        }
        else if (launch_speck_encode(ss, ew, ph, raw_budget, rate, true))
          return -1;
        LAUNCH_K(k_write_slot, dim3(std::max(1u, wblocks), nb), dim3(kThreads), 0, ss, e.cst, e.st,
                 e.stream, e.streamStride, bb.gids, static_cast<uint8_t*>(E.slots.p), d_slotOff,
                 d_lens, P->N, 1);
      }
      if (mode == 3 && pweEarly) {
        // (done above, beside the coder)
      }
      else if (mode == 3 &&
               pwe_outlier_stage<T>(ss, E, *P, bb, nb, d_src, vd, cd, quality, d_lens2, pweKeep,
                                    retry))   // (a chunk that was coded again has 64-bit coefficients)
        return -1;
    }
  
    return 0;
  };
  {
    // SPERR_HIP_ENC_THREADS=1: side by side, every group is enqueued by a host thread of its own
    // (measured on MI355X, 1000^3 in 256^3 chunks, eight groups of about 1100 launches: 64.6 ms
    // with one thread, 66 ms with eight -- the groups' chains of small launches bound the call,
    // not the enqueueing host thread; off by default)
    static const bool encThreadsEnv = tune_getenv("SPERR_HIP_ENC_THREADS") && atoi(tune_getenv("SPERR_HIP_ENC_THREADS")) != 0;
    const bool threaded = sideBySide && encThreadsEnv && !(t_prof && t_prof->on);
    std::vector<int> rcs(groups.size(), 0);
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::vector<std::thread> workers;
    uint32_t groupIdx = 0;
    for (auto& g : groups) {
      const uint32_t gi = groupIdx++;
      if (threaded)
        workers.emplace_back([&, gi, dev]() {
          rcs[gi] = hipSetDevice(dev) == hipSuccess ? do_group(gi, g) : -1;
        });
      else
        rcs[gi] = do_group(gi, g);
      if (!threaded && rcs[gi])
        return -1;
    }
    for (auto& w : workers)
      w.join();
    for (int r : rcs)
      if (r)
        return -1;
  }
  for (auto& L : late) {   // the groups that ran side by side: their plane loops, each over the planes that can hold work
    if (!L || !L->ph.h_bound)
      continue;
    EncBuffers& e = L->bb.eb;
    if (launch_speck_encode_planes(L->ss, e, L->ph, L->raw_budget, rate, false))
      return -1;
    LAUNCH_K(k_write_slot, dim3(std::max(1u, L->wblocks), L->nb), dim3(kThreads), 0, L->ss, e.cst, e.st,
             e.stream, e.streamStride, L->bb.gids, static_cast<uint8_t*>(E.slots.p), d_slotOff, d_lens, L->P->N, 0);
  }
  for (auto& L : late) {   // the 64-bit retry, where a chunk asked for it
    if (!L)
      continue;
    HIP_CHECK(hipMemcpyAsync(L->hc.data(), L->bb.eb.cst, L->nb * sizeof(CoderState), hipMemcpyDeviceToHost, L->ss));
    HIP_CHECK(hipStreamSynchronize(L->ss));
    bool retry = false;
    for (auto& cs : L->hc)
      retry |= (cs.need_retry != 0);
    if (!retry)
      continue;
    EncBatchBufs& bb = L->bb;
    EncBuffers& e = bb.eb;
    ShapePlan* P = L->P;
    EncPlanHost ph{P->d_initLIS, P->d_initLen, P->d_depthBlocks, P->depthBlockOff, P->ht.nsets};
    // (wide_retry_prepare clears bb.aliased: remember it, the guard after the launches needs it)
    const bool wasAliased = bb.aliased;
    if (wasAliased) {   // (see the retry above; the scratch memory is the engine's: one group at a time --
                        // the previous retrying group synchronised its stream below before this one starts)
      const uint32_t cdl[3] = {P->dims[0], P->dims[1], P->dims[2]};
      if (wide_retry_prepare<T>(L->ss, E, *P, bb, L->nb, cdl, d_src, vd, L->orgAligned, false))
        return -1;
    }
    if (launch_make_q_wide(L->ss, L->nb, e.cst) || reset_enc_pass(L->ss, bb, L->nb))
      return -1;
    if (launch_quantize(L->ss, true, bb.vals, bb.valsStride, L->nb, P->N, bb.vals, bb.valsStride,
                        const_cast<uint64_t*>(e.sign), e.signStride, bb.msb, e.pixStride, e.cst))
      return -1;
    EncBuffers ew = e;
    ew.coef = bb.vals;
    ew.coefStride = bb.valsStride;
    if (launch_speck_encode(L->ss, ew, ph, L->raw_budget, true, true))
      return -1;
    LAUNCH_K(k_write_slot, dim3(std::max(1u, L->wblocks), L->nb), dim3(kThreads), 0, L->ss, e.cst, e.st,
             e.stream, e.streamStride, bb.gids, static_cast<uint8_t*>(E.slots.p), d_slotOff, d_lens,
             P->N, 1);
    if (wasAliased)   // the next retrying group takes the engine's scratch memory over
      HIP_CHECK(hipStreamSynchronize(L->ss));
  }
  if (sideBySide)
    for (uint32_t q = 0; q < kSubStreams; q++) {
      HIP_CHECK(hipEventRecord(E.evJoin[q], E.sub[q]));
      HIP_CHECK(hipStreamWaitEvent(st, E.evJoin[q], 0));
    }

  // ---- container ----
  // (the header kernels write before any length is known to them: check its room here)
  if (dst_cap < (slice ? (slice == 2 ? 10u : 0u) : (nchunks > 1 ? 20u : 14u) + 4ull * nchunks)) {
    fprintf(stderr, "[sperr_hip] output buffer too small for the container header (%zu bytes)\n", dst_cap);
    return -1;
  }
  if (slice)
    LAUNCH_K(k_slice_header, dim3(1), dim3(1), 0, st, d_dst, d_lens, d_lens2, d_offs,
             (uint32_t)vol[0], (uint32_t)vol[1], std::is_same<T, float>::value ? 1 : 0,
             slice == 2 ? 1 : 0, d_total);
  else
    LAUNCH_K(k_container_header, dim3(1), dim3(1), 0, st, d_dst, d_lens, d_lens2, d_offs, nchunks,
             (uint32_t)vol[0], (uint32_t)vol[1], (uint32_t)vol[2], (uint32_t)cdim[0],
             (uint32_t)cdim[1], (uint32_t)cdim[2], std::is_same<T, float>::value ? 1 : 0, d_total);
  {
    const uint32_t gy = std::min<uint32_t>(nchunks, 32768u);
    const uint32_t gx = nchunks >= 4096 ? 4u : nchunks >= 256 ? 64u : 1024u;
    LAUNCH_K(k_copy_slots, dim3(gx, gy), dim3(kThreads), 0, st, d_dst, (uint64_t)dst_cap,
             static_cast<const uint8_t*>(E.slots.p), d_slotOff, d_lens, d_offs, nchunks);
  }
  for (auto& k : pweKeep.v)
    LAUNCH_K(k_copy_slots2, dim3(256, k.nb), dim3(kThreads), 0, st, d_dst, (uint64_t)dst_cap, k.slots,
             k.slotOff, k.gids, d_lens, d_lens2, d_offs);
  uint64_t total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  HIP_CHECK(hipGetLastError());
  E.pweLastStream = nullptr;
  E.prof.collect();
  if (total > dst_cap) {
    fprintf(stderr, "[sperr_hip] output buffer too small (%zu < %llu)\n", dst_cap,
            (unsigned long long)total);
    return -1;
  }
  *dst_len = (size_t)total;
  drainGuard.ok = true;
  return 0;
}

// ------------------------------------------------------------------------------------------
// decompression
// ------------------------------------------------------------------------------------------
// what a chunk's DecState::error says (common.h): a stream that does not add up, or a look-back wait that ran into
// its wall-time bound -- the second is the device's trouble (shared, stalled), not the container's
void report_dec_error(uint32_t code)
{
  if (code == kErrLookBackTimeout)
    fprintf(stderr, "[sperr_hip] decoder: a look-back wait timed out after %llu s (device shared, paused or stalled?) -- "
                    "the container may be fine\n", (unsigned long long)(kSpinLimitTicks / 100000000ull));
  else
    fprintf(stderr, "[sperr_hip] decoder: a chunk stream does not add up (damaged or truncated container)\n");
}

int read_container_info(const uint8_t* d_src, size_t src_len, ContainerInfo& ci, hipStream_t st)
{
  std::vector<uint8_t> h(std::min<size_t>(src_len, 20));
  HIP_CHECK(hipMemcpyAsync(h.data(), d_src, h.size(), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  size_t need = 0;
  int r = parse_container_host(h.data(), h.size(), src_len, ci, &need);
  if (r == 1) {
    if (need > src_len)
      return -1;
    h.resize(need);
    HIP_CHECK(hipMemcpyAsync(h.data(), d_src, need, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    r = parse_container_host(h.data(), h.size(), src_len, ci, &need);
  }
  return r == 0 ? 0 : -1;
}

// the lists of the larger sets GPU-wide (k_lis_hi); SPERR_HIP_LIS_HI=0 and regular trees whose geometry
// tables do not fit the kernel's LDS go to k_lis_mx, which takes any shape (k_lis_tables, one workgroup
// per chunk, was the table kernel of rounds 1-3: removed in round 4)
bool g_lis_stamps_on = false;
constexpr int kHiMaxK = 9;   // longest class chain k_lis_hi takes (chunk dims up to 1024)
bool use_lis_hi(const ShapePlan& P, bool tables)
{
  static const bool hiEnv = !(getenv("SPERR_HIP_LIS_HI") && atoi(getenv("SPERR_HIP_LIS_HI")) == 0);
  return hiEnv && tables && P.ht.grids.size() <= 288 && P.ht.roots.size() <= 48 &&
         P.maxK >= 1 && P.maxK <= kHiMaxK;
}
// every LIS level is regular and the table kernels (k_lis_l0 / _l1 / _hi) can take the shape
bool use_tables(const ShapePlan& P)
{
  if (!P.ht.allRegular || P.maxK < 1)
    return false;
  return use_lis_hi(P, true);
}
// lists that mix set shapes (any chunk extent that is not a power of two, every slice): k_lis_mx (speck_mx.hip:
// rows keyed by shape class, several workgroups per chunk, only the walk serial).  SPERR_HIP_LIS_MIXED=0, and trees
// the class machinery does not take (more than 254 classes, 48 roots, 352 grids): k_lis_walk, the serial walk.
// (k_lis_mixed, the one-workgroup-per-chunk kernel of rounds 2-3 whose formulation k_lis_mx took over, was removed
// at the end of round 4.)
bool use_mixed(const ShapePlan& P)
{
  static const bool mixEnv = !(getenv("SPERR_HIP_LIS_MIXED") && atoi(getenv("SPERR_HIP_LIS_MIXED")) == 0);
  if (!mixEnv || use_tables(P) || P.ht.cls.empty() || P.ht.roots.size() > 48 || P.ht.grids.size() > 352 ||
      P.ht.mxSlot.size() != P.ht.cls.size())
    return false;
  return 2 * kMxS + 256 <= kMxRing && ((kMxS + kMxM) >> 6) + 5 <= 64 && kMxM >= 192 &&
         mx_smem_bytes(kMxS, kMxM, kMxQ) <= 138u * 1024u;   // (k_lis_mx has 21 KB of static LDS)
}
std::vector<uint64_t> g_lis_stamps_host;   // chunk 0 of the last decoded batch

struct DecBatchBufs {
  DecBuffers db;
  ChunkGeom* geom;
  uint64_t *chunkOff, *chunkLen;
  double* vals;
  size_t valsStride;
  uint32_t* coef32;
  uint32_t* live;   // chunks that still decode (DecPlanHost::d_live)
};

// valsElems: fp64 samples per chunk of the chunk buffer (0: the whole chunk; compact_box() otherwise)
// refNPlanes: refinement bit planes per chunk (DecBuffers::refPlanes; 0: the coefficients are updated plane by plane)
bool carve_dec(Arena& A, const ShapePlan& P, uint32_t B, uint64_t maxPayloadBytes, DecBatchBufs& o,
               size_t valsElems = 0, uint32_t refNPlanes = 0)
{
  const size_t N = P.N, Npad = round_up(N, 512);   // (512: k_ref_assemble takes eight mask words per round)
  DecBuffers& d = o.db;
  memset(&d, 0, sizeof(d));
  d.tree = P.dtree;
  d.treeTabLen = (uint32_t)P.ht.tab.size();
  d.nchunks = B;
#define TAKE(dst, T, count)                                                            \
  dst = A.take<T>((size_t)(count));                                                    \
  if (!dst)                                                                            \
    return false;                                                                      \
  if (arena_debug())                                                                   \
    fprintf(stderr, "[sperr_hip] arena %-18s %10.2f MB\n", #dst, (double)((size_t)(count) * sizeof(T)) / 1048576.0);
  TAKE(d.cst, CoderState, B);
  TAKE(d.st, DecState, B);
  TAKE(o.geom, ChunkGeom, B);
  TAKE(o.chunkOff, uint64_t, B);
  TAKE(o.chunkLen, uint64_t, B);
  TAKE(o.live, uint32_t, 64);
  o.valsStride = valsElems ? round_up(valsElems, 256) : Npad;
  TAKE(o.vals, double, o.valsStride * B);
  d.coefStride = Npad;
  TAKE(o.coef32, uint32_t, Npad * B);
  d.coef = o.coef32;
  d.signStride = Npad / 64;
  TAKE(d.sign, uint64_t, d.signStride * B);
  d.maskPixStride = Npad / 64;
  d.refPlanes = nullptr;
  d.refMask = nullptr;
  d.wordTop = nullptr;
  d.refNPlanes = 0;
  d.refPlaneStride = d.wordTopStride = 0;
  if (refNPlanes) {
    // (round 6: the planes live in the coefficient array -- 32 plane slots x 8 bytes per mask word = the 64 x 4 bytes
    //  of its coefficients, speck_dec.h; Npad is a multiple of 512: whole tiles of eight mask words)
    d.refNPlanes = std::min<uint32_t>(refNPlanes, 32u);
    d.refPlaneStride = d.coefStride / 2;
    d.wordTopStride = round_up(d.maskPixStride, 64);
    d.refPlanes = reinterpret_cast<uint64_t*>(o.coef32);
    TAKE(d.refMask, uint64_t, d.maskPixStride * B);
    TAKE(d.wordTop, uint8_t, d.wordTopStride * B);
  }
  TAKE(d.bornM, uint64_t, d.maskPixStride * B);
  TAKE(d.sigOld, uint64_t, d.maskPixStride * B);
  TAKE(d.sigNew, uint64_t, d.maskPixStride * B);
  d.lisStride = P.lisEntries;
  TAKE(d.lis[0], uint64_t, P.lisEntries * B);
  TAKE(d.lis[1], uint64_t, P.lisEntries * B);
  d.levelOff = P.d_levelOff;
  d.nPixTiles = (uint32_t)((Npad / 64 + kThreads - 1) / kThreads);   // 256 mask words per tile
  d.tileStride = round_up(d.nPixTiles, 32);
  TAKE(d.tileLip, uint32_t, d.tileStride * B);
  TAKE(d.tileRef, uint32_t, d.tileStride * B);
  TAKE(d.tileLipOff, uint32_t, d.tileStride * B);
  TAKE(d.tileRefOff, uint32_t, d.tileStride * B);
  {
    // (only for the regular trees: there every birth of a sample comes through a leaf event; k_lis_mixed and
    //  k_lis_walk set mask bits themselves.  SPERR_HIP_TILE_SKIP=0: every tile swept on every plane)
    static const bool tileSkip = !(tune_getenv("SPERR_HIP_TILE_SKIP") && atoi(tune_getenv("SPERR_HIP_TILE_SKIP")) == 0);
    uint8_t* tb = nullptr;
    TAKE(tb, uint8_t, d.tileStride * B);
    d.tileBorn = (tileSkip && use_tables(P)) ? tb : nullptr;
  }
  d.lipResStride = Npad / 64 + 2;
  TAKE(d.lipSig, uint64_t, d.lipResStride * B);
  TAKE(d.lipNeg, uint64_t, d.lipResStride * B);
  d.tokStride = (2 * N + 1 + 63) / 64 + 2;
  TAKE(d.tokMask, uint64_t, d.tokStride * B);
  TAKE(d.tokCnt, uint32_t, d.tokStride * B);
  TAKE(d.tokOff, uint32_t, d.tokStride * B);
  d.tokSegStride = round_up(d.tokStride / 2048 + 2, 32);   // (kLipSeg words a segment, speck_dec.hip)
  TAKE(d.tokSegSum, uint32_t, d.tokSegStride * B);
  TAKE(d.tokSegBase, uint32_t, d.tokSegStride * B);
  d.streamStride = (size_t)(maxPayloadBytes / 8) + 4;
  TAKE(d.stream, uint64_t, d.streamStride * B);
  // table-driven LIS phase
  d.levelClass = P.d_levelClass;
  d.levelSlot = P.d_levelSlot;
  d.slotLevel = P.d_slotLevel;
  d.nSlots = P.nSlots;
  d.maskWords = (uint32_t)d.streamStride;
  d.maskStride = (size_t)P.nSlots * d.maskWords;
  TAKE(d.mask, uint64_t, std::max<size_t>(d.maskStride, 1) * B);
  d.prefWords = (d.maskWords + 3u) / 4u;
  d.prefStride = (size_t)P.nSlots * d.prefWords;
  TAKE(d.maskPrefix, uint32_t, std::max<size_t>(d.prefStride, 1) * B);
  d.bornStride = P.ht.nsets + 8;
  d.hiGroupsMax = 8;
  // (a segment that fills up sends the rest to the shared part, which holds the worst case: eight
  //  segments of a twelfth of it each were never seen to fill up at 2 to 4.5 bits per sample; round 6: a
  //  sixteenth -- a set is born once, so a plane's births are a fraction of the sets, and the lists of the three
  //  smallest set sizes, whose kernels claim slots of the shared part, hold most of them: 9.7 MB less per 256^3 chunk)
  d.bornSeg = (uint32_t)((P.ht.nsets + 8) / 16 + 64);
  d.bornPitch = d.bornStride + (size_t)d.bornSeg * d.hiGroupsMax;
  TAKE(d.bornPacked, uint64_t, d.bornPitch * B);
  TAKE(d.bornPosLev, uint64_t, d.bornPitch * B);
  d.lisStamps = nullptr;
  if (g_lis_stamps_on) {
    TAKE(d.lisStamps, uint64_t, 64 * B);
  }
  d.queueCap = 28672 + 64;
  // k_lis_hi: a pair of queues per workgroup, up to hiGroupsMax workgroups per chunk
  // k_lis_hi keeps tables for the classes of the smallest sets only (2^3 .. 32^3 by default: SPERR_HIP_HI_KCAP);
  // larger sets are walked into bit by bit, which leaves the LDS to longer regions of the stream
  static const int hiKcap = tune_getenv("SPERR_HIP_HI_KCAP") ? std::max(2, atoi(tune_getenv("SPERR_HIP_HI_KCAP"))) : 5;
  d.hiK = (uint32_t)std::min(std::max(2, P.maxK), hiKcap);
  {
    static const uint32_t ahead = tune_getenv("SPERR_HIP_HI_AHEAD") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_HI_AHEAD")) : 512u;   // (round 5, with regions of 6912 positions: eight chunks 38.9 -> 39.7 GB/s, 64 chunks the same)
    d.hiAhead = ahead;
    static const uint32_t extra = tune_getenv("SPERR_HIP_HI_EXTRA") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_HI_EXTRA")) : 1u;
    d.hiExtra = extra;
    // (round 5: off -- without the second table a region holds 6912 positions instead of 5632, a chunk has a fifth
    //  fewer regions on its serial chain, and the table it rebuilds now and then costs less than that:
    //  decompression of the bench volume 91.5 -> 94.5 GB/s, eight chunks 38.2 -> 39.3)
    static const uint32_t hop2 = tune_getenv("SPERR_HIP_HI_HOP2") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_HI_HOP2")) : 0u;
    d.hiHop2 = hop2;
    // (round 6: 1 -- class tables 48.0 K -> 36.1 K cycles per region, k_lis_hi 16.7 -> 14.7 ms summed per step,
    //  decompression of 64 chunks 113.0 -> 115.0 GB/s in alternating runs, profiles/r6_hi_candidates_ab.txt)
    static const uint32_t hiCandEnv = tune_getenv("SPERR_HIP_HI_CAND") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_HI_CAND")) : 1u;
    d.hiCand = hiCandEnv;
  }
  d.hiSmemBytes = 148 * 1024;   // (k_lis_hi has 11.5 KB of static LDS)
  d.hiW = hi_window((int)d.hiK, d.hiSmemBytes, d.hiHop2);
  d.queueStride = (size_t)d.queueCap * 4 * d.hiGroupsMax;
  TAKE(d.queue, uint64_t, d.queueStride * B);
  d.hiAhead = std::min(d.hiAhead / 64 * 64, d.hiW / 2);
  d.hiFlagStride = ((d.streamStride * 64 + N) / std::max<uint32_t>(512u, d.hiW - d.hiAhead) + 12) * 4;   // (+ the short first regions of a phase)
  d.mxSlot = P.d_mxSlot;
  d.mxLevelGroup = P.d_mxLevelGroup;
  d.mxS = kMxS;
  d.mxM = kMxM;
  d.mxQ = kMxQ;
  d.mxSmemBytes = mx_smem_bytes(kMxS, kMxM, kMxQ);
  if (use_mixed(P))   // (k_lis_mx keeps its look-back words in the same array: kMxWordsPerRegion per region of mxS bits)
    d.hiFlagStride = std::max<size_t>(d.hiFlagStride, ((d.streamStride * 64 + N) / kMxS + 4) * kMxWordsPerRegion);
  TAKE(d.hiFlags, unsigned long long, d.hiFlagStride * B);
  d.iRoots = P.d_iRoots;
  d.iLevels = P.ht.iLevels;
  d.leafCap = P.ht.nsets + 8;
  d.leafSeg = (uint32_t)((P.ht.nsets + 8) / 16 + 64);
  d.leafStride = d.leafCap + (size_t)d.leafSeg * d.hiGroupsMax;
  TAKE(d.leafEv, uint64_t, d.leafStride * B);
  d.sigbitsStride = P.lisEntries / 64 + 4;
  TAKE(d.sigbits, uint64_t, d.sigbitsStride * B);
  d.l0FlagStride = (d.streamStride * 64 + N) / 4096 + 4;   // (zero padding may be walked)
  TAKE(d.l0Flags, unsigned long long, d.l0FlagStride * B);
  {
    static const bool lbEnv = !(tune_getenv("SPERR_HIP_L01_LOOKBACK") && atoi(tune_getenv("SPERR_HIP_L01_LOOKBACK")) == 0);
    d.l0Tab = nullptr;
    if (lbEnv) {
      TAKE(d.l0Tab, unsigned long long, d.l0FlagStride * 17 * B);
    }
  }
  d.l0Level = P.l0Level;
  TAKE(d.l1Flags, unsigned long long, d.l0FlagStride * B);
  d.l1Level = P.l1Level;
  TAKE(d.l2Flags, unsigned long long, d.l0FlagStride * B);
  d.l2Level = P.l2Level;
  d.wordLeaf = P.d_wordLeaf;
  d.leafStateStride = round_up(P.ht.nnodes, 64);
  TAKE(d.leafState, uint16_t, d.leafStateStride * B);
  d.leafDirtyStride = d.leafStateStride / 32 + 1;
  TAKE(d.leafDirty, uint8_t, d.leafDirtyStride * B);
#undef TAKE
  return true;
}

// multi-resolution decoding (SPERR3D_OMP_D::decompress(p, true), src/SPERR3D_OMP_D.cpp:50-150):
// the volume at every coarsened resolution of the chunks (src/sperr_helper.cpp:70-123), coarsest
// first.  Only for dyadic chunks that tile the volume.
struct MultiRes {
  size_t nlev = 0;
  std::array<uint32_t, 3> cres[16];   // chunk resolution of level h
  std::array<uint32_t, 3> grid;       // chunks per axis
  double* d_level[16];
};

int multires_levels(const Dims& vol, const Dims& cdim, MultiRes& m)
{
  m.nlev = 0;
  size_t levels = 0;
  for (int a = 0; a < 3; a++)
    if (cdim[a] == 0 || vol[a] % cdim[a] != 0)
      return 0;
  if (!spk::can_use_dyadic({cdim[0], cdim[1], cdim[2]}, levels) || levels > 16)
    return 0;
  for (int a = 0; a < 3; a++)
    m.grid[a] = (uint32_t)(vol[a] / cdim[a]);
  for (size_t lev = levels; lev > 0; lev--)
    for (int a = 0; a < 3; a++)
      m.cres[levels - lev][a] = (uint32_t)spk::approx_detail_len(cdim[a], lev)[0];
  m.nlev = levels;
  return 0;
}

// a slice: one level per level of dwt2d (src/sperr_helper.cpp:86-95, src/CDF97.cpp:114-130)
void multires_levels_2d(size_t dx, size_t dy, MultiRes& m)
{
  const size_t levels = std::min<size_t>(spk::num_of_xforms(std::min(dx, dy)), 16);
  m.grid = {1, 1, 1};
  for (size_t lev = levels; lev > 0; lev--)
    m.cres[levels - lev] = {(uint32_t)spk::approx_detail_len(dx, lev)[0],
                            (uint32_t)spk::approx_detail_len(dy, lev)[0], 1u};
  m.nlev = levels;
}

// the approximation corner of every chunk (src/CDF97.cpp:150-168,581-593), mean added back
// (src/SPECK_FLT.cpp:592-603), placed at the chunk's position in the level's volume
__global__ void __launch_bounds__(kThreads)
k_sub_volume(const double* vals, size_t valsStride, const CoderState* cst, const ChunkGeom* geom,
             uint32_t cx, uint32_t cy, uint32_t cdx, uint32_t cdy, uint32_t cdz, uint32_t sx,
             uint32_t sy, uint32_t sz, uint32_t gx, uint32_t gy, double* level)
{
  const uint32_t c = blockIdx.y;
  const CoderState& cs = cst[c];
  const ChunkGeom g = geom[c];
  const uint32_t gi[3] = {g.org[0] / cdx, g.org[1] / cdy, g.org[2] / cdz};
  const size_t ldx = (size_t)sx * gx, ldy = (size_t)sy * gy;
  const double* in = vals + c * valsStride;
  const uint32_t n = sx * sy * sz;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t x = i % sx, r = i / sx;
    const uint32_t y = r % sy, z = r / sy;
    const double v = cs.is_const ? cs.mean : in[((size_t)z * cy + y) * cx + x] + cs.mean;
    level[(((size_t)gi[2] * sz + z) * ldy + (size_t)gi[1] * sy + y) * ldx + (size_t)gi[0] * sx + x] = v;
  }
}

template <typename T>
int decompress_impl(Engine& E, const uint8_t* d_src, size_t src_len, T* d_dst, size_t dst_cap_vals,
                    const ContainerInfo& ci, hipStream_t st, const MultiRes* mr = nullptr,
                    bool slice = false)
{   // slice: `ci` describes one chunk of dims (x, y, 1) whose stream starts at d_src (2D coder)
  DrainOnError drainGuard{E, st};
  const auto chunks = chunk_volume(ci.vol, ci.chunk);
  const uint32_t nchunks = (uint32_t)chunks.size();
  if (ci.nvals == 0 || ci.nvals > dst_cap_vals)
    return -1;
  (void)src_len;

  // chunk heads (flags, number of planes) decide the integer width and the plane count
  const size_t miscBytes = round_up((size_t)nchunks * 8, 256) * 2 + (size_t)nchunks * 32 + 256;
  if (E.misc.ensure(miscBytes))
    return -1;
  uint64_t* d_off = reinterpret_cast<uint64_t*>(E.misc.p);
  uint64_t* d_len = d_off + round_up(nchunks, 32);
  uint8_t* d_heads = reinterpret_cast<uint8_t*>(d_len + round_up(nchunks, 32));
  HIP_CHECK(hipMemcpyAsync(d_off, ci.off.data(), nchunks * 8, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(d_len, ci.len.data(), nchunks * 8, hipMemcpyHostToDevice, st));
  LAUNCH_K(k_gather_heads, dim3((nchunks + 63) / 64), dim3(64), 0, st, d_src, d_off, d_len,
           d_heads, nchunks);
  std::vector<uint8_t> heads((size_t)nchunks * 32);
  HIP_CHECK(hipMemcpyAsync(heads.data(), d_heads, heads.size(), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));

  // PWE streams: a chunk's SPECK stream may be followed by an outlier stream, which counts only
  // when all of it is there (src/SPECK_FLT.cpp:88-103)
  struct OutHead {
    bool has = false;
    uint64_t off = 0, total_bits = 0;
    int nbp = 0;
  };
  std::vector<OutHead> outHead(nchunks);
  bool anyOutlier = false;
  {
    std::vector<uint64_t> tailOff(nchunks, 0), tailLen(nchunks, 0);
    bool anyTail = false;
    for (uint32_t i = 0; i < nchunks; i++) {
      const uint8_t* hd = heads.data() + (size_t)i * 32;
      if (ci.len[i] < 26 || (hd[0] & 0x01))
        continue;
      uint64_t tb;
      memcpy(&tb, hd + 18, 8);
      const uint64_t speckLen = std::min<uint64_t>(9 + bytes_of_bits(tb), ci.len[i] - 17);
      if (17 + speckLen + 9 <= ci.len[i]) {
        tailOff[i] = ci.off[i] + 17 + speckLen;
        tailLen[i] = ci.len[i] - 17 - speckLen;
        anyTail = true;
      }
    }
    if (anyTail) {
      HIP_CHECK(hipMemcpyAsync(d_off, tailOff.data(), nchunks * 8, hipMemcpyHostToDevice, st));
      HIP_CHECK(hipMemcpyAsync(d_len, tailLen.data(), nchunks * 8, hipMemcpyHostToDevice, st));
      LAUNCH_K(k_gather_heads, dim3((nchunks + 63) / 64), dim3(64), 0, st, d_src, d_off, d_len,
               d_heads, nchunks);
      std::vector<uint8_t> tails((size_t)nchunks * 32);
      HIP_CHECK(hipMemcpyAsync(tails.data(), d_heads, tails.size(), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      for (uint32_t i = 0; i < nchunks; i++) {
        if (tailLen[i] < 9)
          continue;
        const uint8_t* t = tails.data() + (size_t)i * 32;
        uint64_t ob;
        memcpy(&ob, t + 1, 8);
        if (tailLen[i] != 9 + bytes_of_bits(ob))   // no wrap for ob near 2^64
          continue;
        outHead[i].has = true;
        outHead[i].off = tailOff[i];
        outHead[i].total_bits = ob;
        outHead[i].nbp = t[0];
        anyOutlier = true;
      }
    }
  }

  struct Ref {
    uint32_t gid;
    uint32_t org[3];
  };
  std::map<Dims, std::vector<Ref>> groups;
  for (uint32_t i = 0; i < nchunks; i++) {
    const auto& c = chunks[i];
    groups[Dims{c[1], c[3], c[5]}].push_back({i, {(uint32_t)c[0], (uint32_t)c[2], (uint32_t)c[4]}});
  }

  VolDesc vd{{ci.vol[0], ci.vol[1], ci.vol[2]}};
  struct SubHost {
    std::vector<ChunkGeom> hg;
    std::vector<uint64_t> ho, hl;
    std::vector<DecState> hs;
    DecBatchBufs bb;
    uint32_t nb = 0;
    size_t first = 0;
  };
  // Small groups (the border shapes of a volume that the chunk size does not divide; their chunks
  // decode through the serial walk, one wavefront each) are not waited for one by one: each gets
  // its own piece of the arena and one of the sub-streams, and all of them are drained together.
  // SPERR_HIP_DEFER_GROUPS=0 decodes group after group.
  static const bool deferEnv = !(tune_getenv("SPERR_HIP_DEFER_GROUPS") && atoi(tune_getenv("SPERR_HIP_DEFER_GROUPS")) == 0);
  const bool deferOK = deferEnv && !anyOutlier && !mr && !slice && groups.size() > 1;
  // (groups of 32 and more chunks of a shape the table kernels take keep the sub-batch scheme)
  auto deferrable = [](const ShapePlan& P, size_t nchunksOfShape) {
    const bool tables = use_tables(P);
    return nchunksOfShape < 32 || !tables;
  };
  std::vector<std::unique_ptr<SubHost>> pending;
  size_t deferOff = 0;
  uint32_t deferNext = 0;
  bool deferForked = false;
  hipEvent_t timingFork = nullptr;
  std::vector<std::pair<hipEvent_t, std::array<uint32_t, 4>>> timingEnds;
  auto drain = [&]() -> int {
    int rc = 0;
    for (uint32_t q = 0; q < kSubStreams; q++)
      HIP_CHECK(hipStreamSynchronize(E.sub[q]));
    HIP_CHECK(hipStreamSynchronize(E.outlQ[1]));
    for (auto& te : timingEnds) {
      float ms = 0.f;
      if (timingFork && hipEventElapsedTime(&ms, timingFork, te.first) == hipSuccess)
        fprintf(stderr, "[sperr_hip] group %u x %u x %u, %u chunks: through %.2f ms after the first group's fork\n",
                te.second[0], te.second[1], te.second[2], te.second[3], ms);
      (void)hipEventDestroy(te.first);
    }
    timingEnds.clear();
    if (timingFork) {
      (void)hipEventDestroy(timingFork);
      timingFork = nullptr;
    }
    for (auto& S : pending) {                  // stream errors (wrong lengths) surface here
      S->hs.resize(S->nb);
      HIP_CHECK(hipMemcpy(S->hs.data(), S->bb.db.st, S->nb * sizeof(DecState), hipMemcpyDeviceToHost));
      for (auto& hsx : S->hs)
        if (hsx.error) {
          report_dec_error(hsx.error);
          rc = -1;
        }
    }
    pending.clear();
    deferOff = 0;
    return rc;
  };
  auto bytes_per_chunk = [&](const ShapePlan& P, uint64_t maxPayload, size_t valsElems = 0, uint32_t refNPlanes = 0) -> size_t {
    Arena probe;
    probe.base = reinterpret_cast<char*>(uintptr_t(4096));  // size probe only
    probe.cap = ~size_t(0) / 2;
    DecBatchBufs tmp;
    carve_dec(probe, P, 1, maxPayload, tmp, valsElems, refNPlanes);
    return probe.used;
  };
  // Refinement bit planes (speck_dec.h, DecBuffers::refPlanes): as many as the chunks of a group with 32-bit
  // coefficients have planes (byte 17 of a chunk: src/SPECK_INT.cpp:284-308); not for a slice that goes through
  // the quadtree walk of speck2d.hip, which updates coefficients itself.  SPERR_HIP_REF_PLANES=0: none
  // (k_ref_apply2 updates the coefficients plane by plane, as rounds 3 and 4 did).
  static const bool refPlanesEnv = !(tune_getenv("SPERR_HIP_REF_PLANES") && atoi(tune_getenv("SPERR_HIP_REF_PLANES")) == 0);
  auto ref_planes_of = [&](const ShapePlan& P, const std::vector<Ref>& refs) -> uint32_t {
    if (!refPlanesEnv || (slice && !(P.ht.flags & spk::kTree2D)))
      return 0;
    uint32_t n = 0;
    for (const Ref& r : refs) {
      const uint8_t* hd = heads.data() + (size_t)r.gid * 32;
      if (ci.len[r.gid] >= 26 && !(hd[0] & 0x01) && hd[17] <= 32)
        n = std::max<uint32_t>(n, hd[17]);
    }
    return n;
  };
  // The fp64 chunk buffer of a group can be COMPACT (round 3): when the finest level runs as the fused
  // x-y-z kernel and every inverse pass dequantises the samples no coarser level produces straight
  // from the integer coefficients, the buffer only ever holds the box of the second level (an eighth
  // of the chunk: 17 MB instead of 134 MB for 256^3).  Not with 64-bit coefficients (they live in the
  // buffer), outlier correctors (every pass stays in the buffer), the resolution hierarchy or slices.
  static const bool compactEnv = !(tune_getenv("SPERR_HIP_DEC_COMPACT") && atoi(tune_getenv("SPERR_HIP_DEC_COMPACT")) == 0);
  auto compact_box = [&](const ShapePlan& P, const std::vector<Ref>& refs, uint32_t box[3]) -> size_t {
    box[0] = box[1] = box[2] = 0;
    if (!compactEnv || !fuse_xyz(P) || !plan_fusable(P) || mr || slice || anyOutlier || P.fwd.size() < 3)
      return 0;
    for (const Ref& r : refs) {
      const uint8_t* hd = heads.data() + (size_t)r.gid * 32;
      if (ci.len[r.gid] >= 26 && !(hd[0] & 0x01) && hd[17] > 32)
        return 0;   // a chunk with 64-bit coefficients
    }
    for (size_t k = 3; k < P.fwd.size(); k++)
      for (int a = 0; a < 3; a++)
        box[a] = std::max(box[a], P.fwd[k].region[a]);
    for (int a = 0; a < 3; a++)
      box[a] = std::max(box[a], 1u);
    return (size_t)box[0] * box[1] * box[2];
  };
  bool deferSized = false;
  // chunks of this call that decode through k_lis_mx, whatever their shape group: they run side by side, each with
  // several one-per-CU workgroups (the rows of a chunk's regions take about four workgroups to keep its serial walk
  // fed), so the groups share one budget of workgroups (SPERR_HIP_MX_WGS; with more workgroups than CUs the chunks
  // launched last wait for the first ones to END: 1000^3 in 256^3 chunks, 37 such chunks at 8 workgroups each,
  // decoded no faster than with one workgroup per chunk)
  uint32_t mxGroupsCall = 0;
  {
    size_t nmx = 0;
    for (auto& h : groups) {
      ShapePlan* Q = (slice && slice_forest_enabled()) ? E.plan(h.first[0], h.first[1], 0) : E.plan(h.first[0], h.first[1], h.first[2]);
      if (Q && use_mixed(*Q))
        nmx += h.second.size();
    }
    static const uint32_t mxBudget = getenv("SPERR_HIP_MX_WGS") ? (uint32_t)atoi(getenv("SPERR_HIP_MX_WGS")) : 208u;
    // (calls that share the device -- the chunk farm's workers, several host threads -- share the budget: each of these
    //  workgroups has a CU to itself for as long as its chunk's phase lasts)
    size_t sharers = 1;
    {
      int devNow = 0;
      if (hipGetDevice(&devNow) == hipSuccess)
        sharers = std::max<size_t>(1, g_pool.busy_on(devNow));
    }
    if (nmx)
      mxGroupsCall = std::min<uint32_t>(8u, std::max<uint32_t>(2u, (uint32_t)(mxBudget / (nmx * sharers))));
  }
  // a slice is decoded by the kernels of the 3D decoder on the 2D coder's forest (k_lis_mx and its
  // type-I phase); SPERR_HIP_SLICE_MIXED=0: by k_speck2d_decode, one workgroup walking the quadtree
  const bool sliceMixed = slice_forest_enabled();
  for (int pass = 0; pass < 2; pass++)
  for (auto& g : groups) {
    ShapePlan* P = nullptr;
    if (slice && sliceMixed) {
      P = E.plan(g.first[0], g.first[1], 0);
      if (P && !use_mixed(*P))
        P = nullptr;
    }
    if (!P)
      P = E.plan(g.first[0], g.first[1], g.first[2]);
    if (!P)
      return -1;
    const bool deferG = deferOK && deferrable(*P, g.second.size());
    if (deferG != (pass == 1))
      continue;
    if (deferG && !deferSized) {   // room for all the small groups at once, if the memory is there
      deferSized = true;
      size_t sum = 0;
      for (auto& h : groups) {
        ShapePlan* Q = E.plan(h.first[0], h.first[1], h.first[2]);
        if (!Q)
          return -1;
        if (!deferrable(*Q, h.second.size()))
          continue;
        uint64_t mp = 0;
        for (auto& r : h.second)
          mp = std::max<uint64_t>(mp, ci.len[r.gid]);
        uint32_t qbox[3];
        sum += round_up(h.second.size() * bytes_per_chunk(*Q, mp, compact_box(*Q, h.second, qbox), ref_planes_of(*Q, h.second)) + (1 << 20), 4096);
      }
      size_t fr = 0, tot = 0;
      HIP_CHECK(hipMemGetInfo(&fr, &tot));
      const size_t room = arena_room(E.arena.n, fr);
      if (E.arena.ensure(std::min(sum, room)))
        return -1;
    }
    uint64_t maxPayload = 0;
    for (auto& r : g.second)
      maxPayload = std::max<uint64_t>(maxPayload, ci.len[r.gid]);
    uint32_t cbox[3];
    const size_t compactElems = compact_box(*P, g.second, cbox);
    const uint32_t refNPlanes = ref_planes_of(*P, g.second);
    const size_t per = bytes_per_chunk(*P, maxPayload, compactElems, refNPlanes);
    size_t fr = 0, tot = 0;
    HIP_CHECK(hipMemGetInfo(&fr, &tot));
    const size_t budgetBytes = arena_budget(E.arena.n, fr);
    uint32_t B = (uint32_t)std::min<size_t>(g.second.size(), std::max<size_t>(1, budgetBytes / per));
    B = std::min<uint32_t>(B, 256);
    const size_t needBytes = (size_t)B * per + (1 << 20);
    if (deferG && deferOff + needBytes > E.arena.n && drain())   // no room beside the groups in flight
      return -1;
    if (E.arena.ensure(needBytes))   // (grows only when nothing is in flight: deferOff == 0 here)
      return -1;
    const uint32_t cd[3] = {P->dims[0], P->dims[1], P->dims[2]};
    for (size_t b0 = 0; b0 < g.second.size(); b0 += B) {
      const uint32_t nbAll = (uint32_t)std::min<size_t>(B, g.second.size() - b0);
      if (deferG && deferOff + needBytes > E.arena.n && drain())
        return -1;
      Arena A;
      A.base = static_cast<char*>(E.arena.p) + (deferG ? deferOff : 0);
      A.cap = E.arena.n - (deferG ? deferOff : 0);
      hipStream_t deferStream = nullptr;
      if (deferG) {
        if (!deferForked) {   // the sub-streams start behind what the caller's stream holds
          HIP_CHECK(hipEventRecord(E.evFork, st));
          for (uint32_t q = 0; q < kSubStreams; q++)
            HIP_CHECK(hipStreamWaitEvent(E.sub[q], E.evFork, 0));
          HIP_CHECK(hipStreamWaitEvent(E.outlQ[1], E.evFork, 0));
          deferForked = true;
        }
        // (a group of regular chunks beside groups that decode through k_lis_mx: those hold most CUs for the whole
        //  call with workgroups that mostly wait for their turn on a chunk's serial walk, and eight groups on the
        //  eight normal-priority hardware queues -- one of which the caller's stream has -- put two groups on one
        //  queue, one behind the other.  The regular chunks take the 1D decoder's idle high-priority stream: a queue
        //  pool of its own, 1000^3 in 256^3 chunks 178 -> see DESIGN.md section 9)
        if (mxGroupsCall != 0 && use_tables(*P))
          deferStream = E.outlQ[1];
        else
          deferStream = E.sub[deferNext++ % kSubStreams];
        deferOff += round_up(needBytes, 4096);
      }
      // The LIS phase of a plane keeps one latency-bound workgroup per chunk busy; sub-batches on
      // separate streams let the bandwidth-bound kernels of one sub-batch run beside the LIS
      // kernels of another.  Measured on MI355X, 64 chunks of 256^3, decompression only: 1 stream
      // 60.4 ms, 2 streams 58.0 ms, 3 streams 57.9 ms (round 1, with one workgroup per chunk in the
      // LIS phase of the larger sets: 1 stream 74 ms, 3 streams 65 ms).
      // SPERR_HIP_SUBSTREAMS=n overrides the choice (1 = a single stream).
      static const int subEnv = getenv("SPERR_HIP_SUBSTREAMS") ? atoi(getenv("SPERR_HIP_SUBSTREAMS")) : 0;
      static const bool threads = !(tune_getenv("SPERR_HIP_ENQUEUE_THREADS") && atoi(tune_getenv("SPERR_HIP_ENQUEUE_THREADS")) == 0);
      uint32_t nsub = nbAll >= 32 ? 2u : 1u;
      // A call that has the device to itself cuts a smaller batch finer (round 3): the chunks' serial
      // chains bound it, and four sub-batches side by side decode 8 chunks in 14.4 ms instead of 16.5,
      // 27 chunks in 26.7 instead of 30.1 (64 chunks: two 85.0, three 83.3, four 82.1 GB/s).  Not when
      // other calls run on the device (the chunk farm's workers: their items already overlap, and
      // sub-batches on top took its decompression from 40 to 26 GB/s).
      {
        int devNow = 0;
        if (!t_shared_device && hipGetDevice(&devNow) == hipSuccess && g_pool.busy_on(devNow) <= 1)
          nsub = nbAll >= 56 ? 3u : nbAll >= 8 ? 4u : nbAll >= 4 ? 2u : 1u;   // (round 6: three from 56 chunks on -- with this round's
                                                                               //  shorter k_lis_hi 64 chunks decode at 117.0 GB/s
                                                                               //  against 115.5 with two, 110.4 with four)
      }
      if (subEnv > 0)
        nsub = std::min<uint32_t>(kSubStreams, (uint32_t)subEnv);
      // (with outlier streams every sub-batch waits for its stream when it reads back the 1D decoder's
      //  state: several sub-batches only when each has a host thread of its own)
      if ((anyOutlier && !threads) || nbAll < 2 * nsub || deferG)   // (eight sub-batches of one chunk each: 29.5 ms for 8 chunks against 14.1 with four)
        nsub = 1;
      std::vector<SubHost> subs(nsub);
      if (nsub > 1) {
        HIP_CHECK(hipEventRecord(E.evFork, st));
        for (uint32_t q = 0; q < nsub; q++)
          HIP_CHECK(hipStreamWaitEvent(E.sub[q], E.evFork, 0));
      }
      uint32_t done = 0;
      for (uint32_t q = 0; q < nsub; q++) {
        SubHost& S = subs[q];
        S.nb = (nbAll - done + (nsub - q) - 1) / (nsub - q);
        S.first = b0 + done;
        done += S.nb;
        if (S.nb && !carve_dec(A, *P, S.nb, maxPayload, S.bb, compactElems, refNPlanes))
          return -1;
        if (S.nb && compactElems)
          g_dbg_counter[2]++;
      }
      int devId = 0;
      HIP_CHECK(hipGetDevice(&devId));
      // everything one sub-batch enqueues on its stream.  With several sub-batches each is
      // enqueued by its own host thread: one thread would start the last sub-batch only after
      // launching all kernels of the others (about 1300 launches each)
      auto enqueue = [&](uint32_t q) -> int {
        SubHost& S = subs[q];
        const uint32_t nb = S.nb;
        const size_t first = S.first;
        if (nb == 0)
          return 0;
        if (nsub > 1) {
          HIP_CHECK(hipSetDevice(devId));
          t_prof = &E.prof;   // (this may be a thread of its own)
        }
        hipStream_t ss = deferStream ? deferStream : (nsub > 1 ? E.sub[q] : st);
        DecBatchBufs& bb = S.bb;
        DecBuffers& d = bb.db;
        S.hg.resize(nb);
        S.ho.resize(nb);
        S.hl.resize(nb);
        int maxNarrow = 0, maxWide = 0;
        for (uint32_t i = 0; i < nb; i++) {
          const Ref& r = g.second[first + i];
          for (int a = 0; a < 3; a++)
            S.hg[i].org[a] = r.org[a];
          S.ho[i] = ci.off[r.gid];
          S.hl[i] = ci.len[r.gid];
          const uint8_t* hd = heads.data() + (size_t)r.gid * 32;
          if (S.hl[i] >= 26 && !(hd[0] & 0x01)) {
            const int nbp = hd[17];
            if (nbp > 32)
              maxWide = std::max(maxWide, nbp);
            else
              maxNarrow = std::max(maxNarrow, nbp);
          }
        }
        if (maxWide > kMaxPlanes || (compactElems && maxWide))
          return -1;
        // Outlier streams (point-wise error mode) are decoded by the 1D coder on a stream of their own,
        // beside everything below: it needs the container only; the correctors are added at the end
        bool batchOutliers = false;
        for (uint32_t i = 0; i < nb; i++)
          batchOutliers |= outHead[g.second[first + i].gid].has;
        std::vector<OutlierChunk> hoc;
        OutlierBufs ob;
        {
          hipStream_t so = E.outlQ[q];
          if (batchOutliers) {
            HIP_CHECK(hipEventRecord(E.evOutlFork[q], ss));
            HIP_CHECK(hipStreamWaitEvent(so, E.evOutlFork[q], 0));
          }
          if (batchOutliers) {
            hoc.assign(nb, OutlierChunk{});
            memset(hoc.data(), 0, nb * sizeof(OutlierChunk));
            uint64_t maxBits = 0;
            int maxNbp = 1;
            for (uint32_t i = 0; i < nb; i++) {
              const OutHead& oh = outHead[g.second[first + i].gid];
              if (!oh.has)
                continue;
              hoc[i].has = 1;
              hoc[i].streamOff = oh.off;
              hoc[i].total_bits = oh.total_bits;
              hoc[i].nbp = oh.nbp;
              if (oh.nbp > kMaxPlanes)
                return -1;
              maxBits = std::max(maxBits, oh.total_bits);
              maxNbp = std::max(maxNbp, oh.nbp);
            }
            memset(&ob, 0, sizeof(ob));
            ob.nchunks = nb;
            ob.N = P->N;
            ob.nw = (P->N + 63) / 64;
            ob.wordStride = round_up((size_t)ob.nw + 2, 32);
            // every value found costs at least its sign bit, every run kept on a list its test bit
            ob.kStride = round_up((size_t)std::min<uint64_t>(P->N, maxBits) + 1, 64);
            speck1d_level_offsets(ob, P->N, std::min<uint64_t>(maxBits + 2, 2ull * P->N));
            ob.streamStride = (size_t)(maxBits / 64) + 4;
            ob.planeStride = (size_t)maxNbp * ob.wordStride;
            const size_t bytes = round_up(nb * sizeof(OutlierChunk), 256) +
                                 (size_t)nb * (ob.wordStride * 16 + ob.kStride * 5 + ob.runStride * 8 +
                                               ob.streamStride * 8 + ob.planeStride * 8) + 8192;
            if (E.outlDec[q].ensure(bytes))   // (its own buffer: the sub-batches are enqueued by separate threads)
              return -1;
            Arena OA;
            OA.base = static_cast<char*>(E.outlDec[q].p);
            OA.cap = E.outlDec[q].n;
            ob.oc = OA.take<OutlierChunk>(nb);
            ob.lip = OA.take<uint64_t>(nb * ob.wordStride);
            ob.lsp = OA.take<uint64_t>(nb * ob.wordStride);
            ob.runs = OA.take<uint64_t>(nb * ob.runStride);
            ob.stream = OA.take<uint64_t>(nb * ob.streamStride);
            ob.planeBits = OA.take<uint64_t>(nb * ob.planeStride);
            ob.pos = OA.take<uint32_t>(nb * ob.kStride);
            ob.sgn = OA.take<uint8_t>(nb * ob.kStride);
            if (!ob.oc || !ob.lip || !ob.lsp || !ob.runs || !ob.stream || !ob.planeBits || !ob.pos || !ob.sgn)
              return -1;
            HIP_CHECK(hipMemcpyAsync(ob.oc, hoc.data(), nb * sizeof(OutlierChunk), hipMemcpyHostToDevice, so));
            HIP_CHECK(hipMemsetAsync(ob.lip, 0, (size_t)nb * ob.wordStride * 16, so));   // lip + lsp
            if (launch_speck1d_decode(so, ob, d_src))
              return -1;
            HIP_CHECK(hipEventRecord(E.evOutl[q], so));
          }

        }
        HIP_CHECK(hipMemcpyAsync(bb.geom, S.hg.data(), nb * sizeof(ChunkGeom), hipMemcpyHostToDevice, ss));
        HIP_CHECK(hipMemcpyAsync(bb.chunkOff, S.ho.data(), nb * 8, hipMemcpyHostToDevice, ss));
        HIP_CHECK(hipMemcpyAsync(bb.chunkLen, S.hl.data(), nb * 8, hipMemcpyHostToDevice, ss));
        HIP_CHECK(hipMemsetAsync(d.cst, 0, nb * sizeof(CoderState), ss));
        HIP_CHECK(hipMemsetAsync(d.st, 0, nb * sizeof(DecState), ss));
        DecPlanHost ph{P->d_initLIS, P->d_initLen,
                       use_tables(*P), P->l0Level >= 0 && P->ht.grids.size() <= 288, P->l1Level >= 0 && P->ht.grids.size() <= 288, P->maxK};
        ph.skipFinish = true;   // launch_inv_quantize / the dequantising inverse passes complete the coefficients
        ph.l2 = ph.l1 && P->l2Level >= 0;
        // the lists of the larger sets GPU-wide
        ph.hi = use_lis_hi(*P, ph.tables);
        ph.mixed = use_mixed(*P);
        ph.mxGroups = mxGroupsCall;
        if (!ph.tables && !ph.mixed && !(slice && !(P->ht.flags & spk::kTree2D))) {
          // (a tree neither the table kernels nor k_lis_mx take -- more than 288 / 352 grids or 48 roots: chunks of
          //  2^30 samples and more -- decodes correctly, through one serial wavefront per chunk: say so, once)
          static std::atomic<bool> warned{false};
          if (!warned.exchange(true))
            fprintf(stderr, "[sperr_hip] note: chunks of %u x %u x %u decode through the serial walk (k_lis_walk): their tree "
                            "(%zu grids, %zu roots) exceeds what the parallel list kernels hold in LDS -- expect it to be slow\n",
                    cd[0], cd[1], cd[2], P->ht.grids.size(), P->ht.roots.size());
        }
        {
          static const uint32_t gdivEnv = tune_getenv("SPERR_HIP_MX_GRID_DIV") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_MX_GRID_DIV")) : 4u;
          ph.gridDiv = (deferStream && mxGroupsCall != 0 && groups.size() >= 4) ? std::max<uint32_t>(1u, gdivEnv) : 1u;
        }
        // (the host thread may wait for this stream: it is the call's only one, or has a thread of its own)
        static const bool liveEnv = !(tune_getenv("SPERR_HIP_LIVE_CHECK") && atoi(tune_getenv("SPERR_HIP_LIVE_CHECK")) == 0);
        ph.d_live = (liveEnv && !deferStream && (nsub == 1 || threads)) ? bb.live : nullptr;
        ph.h_live = E.liveHost[q % kSubStreams];
        ph.liveEv = E.liveEv[q % kSubStreams];
        // the inverse passes dequantise on the way (not for the resolution hierarchy, whose coarsest
        // level is read before any pass has run)
        const bool fuseDq = plan_fusable(*P) && !mr && !slice;
        // diagnostics: SPERR_HIP_LIS_GPUWIDE=0 leaves every list to k_lis_hi
        static const bool gpuWide = !(getenv("SPERR_HIP_LIS_GPUWIDE") && atoi(getenv("SPERR_HIP_LIS_GPUWIDE")) == 0);
        if (!gpuWide)
          ph.l0 = ph.l1 = ph.l2 = false;
        HIP_CHECK(hipMemsetAsync(d.mask, 0, std::max<size_t>(d.maskStride, 1) * nb * 8, ss));
        HIP_CHECK(hipMemsetAsync(d.l0Flags, 0, d.l0FlagStride * nb * 8, ss));
        if (d.l0Tab)
          HIP_CHECK(hipMemsetAsync(d.l0Tab, 0, d.l0FlagStride * 17 * nb * 8, ss));
        HIP_CHECK(hipMemsetAsync(d.l1Flags, 0, d.l0FlagStride * nb * 8, ss));
        HIP_CHECK(hipMemsetAsync(d.l2Flags, 0, d.l0FlagStride * nb * 8, ss));
        HIP_CHECK(hipMemsetAsync(d.hiFlags, 0, d.hiFlagStride * nb * 8, ss));
        HIP_CHECK(hipMemsetAsync(d.sigbits, 0, d.sigbitsStride * nb * 8, ss));
        if (d.lisStamps)
          HIP_CHECK(hipMemsetAsync(d.lisStamps, 0, 64 * 8 * nb, ss));
        // 64-bit chunks first: their magnitudes are decoded into (and converted inside) the fp64
        // buffer, which the 32-bit pass then fills for the remaining chunks
        for (int wide = 1; wide >= 0; wide--) {
          if (wide && maxWide == 0)
            continue;
          HIP_CHECK(hipMemsetAsync(d.bornM, 0, d.maskPixStride * nb * 8, ss));
          if (d.tileBorn)
            HIP_CHECK(hipMemsetAsync(d.tileBorn, 0, d.tileStride * nb, ss));
          HIP_CHECK(hipMemsetAsync(d.sigOld, 0, d.maskPixStride * nb * 8, ss));
          HIP_CHECK(hipMemsetAsync(d.sigNew, 0, d.maskPixStride * nb * 8, ss));
          HIP_CHECK(hipMemsetAsync(d.sign, 0xff, d.signStride * nb * 8, ss));
          HIP_CHECK(hipMemsetAsync(d.leafState, 0, d.leafStateStride * nb * 2, ss));
          HIP_CHECK(hipMemsetAsync(d.leafDirty, 0, d.leafDirtyStride * nb, ss));
          HIP_CHECK(hipMemsetAsync(d.stream, 0, d.streamStride * nb * 8, ss));
          DecBuffers dw = d;
          if (wide) {  // 64-bit magnitudes live in the fp64 buffer, converted in place afterwards
            dw.coef = bb.vals;
            dw.coefStride = bb.valsStride;
            dw.refPlanes = nullptr;
            HIP_CHECK(hipMemsetAsync(bb.vals, 0, bb.valsStride * nb * 8, ss));
          }
          else if (d.refPlanes) { // (k_ref_assemble writes every coefficient; a plane's words are valid from wordTop down)
            HIP_CHECK(hipMemsetAsync(d.wordTop, 0, d.wordTopStride * nb, ss));
            dw.coefSigned = fuseDq ? 1u : 0u;   // (read by the dequantising inverse passes only: LiftFuse::coefSigned)
          }
          else
            HIP_CHECK(hipMemsetAsync(bb.coef32, 0, d.coefStride * nb * 4, ss));
          if (slice && !(P->ht.flags & spk::kTree2D)) {   // header + stream words by the 3D launcher (no planes), then the 2D coder
            DecPlanHost ph2 = ph;
            ph2.tables = ph2.l0 = ph2.l1 = ph2.mixed = false;
            Speck2dBufs sb;
            if (launch_speck_decode(ss, dw, ph2, d_src, bb.chunkOff, bb.chunkLen, wide != 0, 0) ||
                carve_slice2d(E, *P, sb))
              return -1;
            sb.coef = dw.coef;
            sb.sign = d.sign;
            sb.stream = d.stream;
            sb.streamWords = d.streamStride;
            sb.cst = d.cst;
            sb.dst = d.st;
            if (launch_speck2d_decode(ss, sb, wide != 0) ||
                launch_inv_quantize(ss, wide != 0, dw.coef, dw.coefStride, d.sign, d.signStride, nb,
                                    P->N, bb.vals, bb.valsStride, d.cst))
              return -1;
            continue;
          }
          // the header kernel must run even when no plane does (constant / all-zero chunks)
          if (launch_speck_decode(ss, dw, ph, d_src, bb.chunkOff, bb.chunkLen, wide != 0,
                                  wide ? maxWide : maxNarrow))
            return -1;
          // (32-bit coefficients are dequantised by the inverse passes as they load them, LiftFuse)
          if ((wide || !fuseDq) &&
              launch_inv_quantize(ss, wide != 0, dw.coef, dw.coefStride, d.sign, d.signStride, nb,
                                  P->N, bb.vals, bb.valsStride, d.cst, d.sigNew, d.sigOld,
                                  d.maskPixStride, d.st))
            return -1;
        }
        // the last inverse pass covers the whole chunk: it adds the mean, narrows and scatters --
        // unless outlier correctors have to be added to the transformed values first
        // (src/SPECK_FLT.cpp:573-590), in which case every pass stays in the chunk buffer
        const bool fxyz = fuse_xyz(*P) && !batchOutliers;
        const bool fxy = fuse_xy(*P) && !batchOutliers && !fxyz;
        // With outlier correctors the transformed values have to stay doubles a little longer -- but they can still
        // come from the fused kernels (round 5): the coarser levels in a compact buffer of their box, the finest level
        // by k_lift_xyz_inv writing doubles, no mean added, into the chunk buffer as if it were a volume of bricks (a
        // chunk's offset rides in org[0]); the correctors and the scatter pass follow as before.  (Fifteen per-axis
        // passes over the whole chunk before: 12.7 ms for 64 chunks of 256^3.)
        static const bool brickEnv = !(tune_getenv("SPERR_HIP_PWE_FUSED_INV") && atoi(tune_getenv("SPERR_HIP_PWE_FUSED_INV")) == 0);
        const bool fbrick = brickEnv && batchOutliers && fuse_xyz(*P) && fuseDq && maxWide == 0 && P->fwd.size() >= 3 &&
                            compactElems == 0 && (uint64_t)nb * bb.valsStride <= 0xffffffffull;
        uint32_t bbox[3] = {1, 1, 1};
        double* boxVals = nullptr;
        size_t boxStride = 0;
        ChunkGeom* d_bricks = nullptr;
        std::vector<ChunkGeom> bricks;   // (lives until the wait for the stream below: batchOutliers)
        if (fbrick) {
          for (size_t k = 3; k < P->fwd.size(); k++)
            for (int a = 0; a < 3; a++)
              bbox[a] = std::max(bbox[a], P->fwd[k].region[a]);
          boxStride = round_up((size_t)bbox[0] * bbox[1] * bbox[2], 64);
          const size_t geomOff = round_up((size_t)nb * boxStride * 8, 256);
          if (E.decBox[q].ensure(geomOff + (size_t)nb * sizeof(ChunkGeom) + 256))
            return -1;
          boxVals = static_cast<double*>(E.decBox[q].p);
          d_bricks = reinterpret_cast<ChunkGeom*>(static_cast<char*>(E.decBox[q].p) + geomOff);
          bricks.resize(nb);
          for (uint32_t i = 0; i < nb; i++) {
            bricks[i].org[0] = (uint32_t)((size_t)i * bb.valsStride);
            bricks[i].org[1] = bricks[i].org[2] = 0;
          }
          HIP_CHECK(hipMemcpyAsync(d_bricks, bricks.data(), nb * sizeof(ChunkGeom), hipMemcpyHostToDevice, ss));
        }
        // a level of the inverse transform is 3 passes (z y x) of a dyadic chunk, 2 (y x) of a slice
        const size_t perLevel = slice ? 2 : 3;
        auto sub_volume = [&](size_t k) -> int {   // before pass k, the first of its level
          const size_t h = mr->nlev - (k / perLevel + 1);
          const auto& r = mr->cres[h];
          const uint32_t blocks = capped_blocks((r[0] * r[1] * r[2] + kThreads - 1) / kThreads, nb);
          LAUNCH_K(k_sub_volume, dim3(blocks, nb), dim3(kThreads), 0, ss, bb.vals, bb.valsStride,
                   d.cst, bb.geom, cd[0], cd[1], cd[0], cd[1], cd[2], r[0], r[1], r[2],
                   mr->grid[0], mr->grid[1], mr->d_level[h]);
          return 0;
        };
        auto dequant_fuse = [&](size_t k, LiftFuse& lf) {
          if (fuseDq && pass_fuse(*P, k, lf.inner) > 0) {
            lf.mode = 2;
            lf.coef = bb.coef32;
            lf.coefStride = d.coefStride;
            lf.sign = d.sign;
            lf.signStride = d.signStride;
            lf.sigNew = d.sigNew;
            lf.sigOld = d.sigOld;
            lf.maskStride = d.maskPixStride;
            lf.dst = d.st;
            lf.coefSigned = d.refPlanes ? 1 : 0;
          }
          if (compactElems) {
            lf.bufx = cbox[0];
            lf.bufy = cbox[1];
          }
          if (fbrick) {
            lf.bufx = bbox[0];
            lf.bufy = bbox[1];
          }
        };
        for (size_t k = P->fwd.size(); k-- > ((fxyz || fbrick) ? 3u : fxy ? 2u : 0u);) {
          const LiftPass& ps = P->fwd[k];
          if (mr && mr->nlev && k % perLevel == perLevel - 1 && sub_volume(k))
            return -1;
          LiftFuse lf;
          dequant_fuse(k, lf);
          if (launch_lift(ss, false, fbrick ? boxVals : bb.vals, fbrick ? boxStride : bb.valsStride, nb, cd, ps.axis,
                          ps.region, d.cst,
                          (k == 0 && !batchOutliers) ? (std::is_same<T, float>::value ? 1 : 2) : 0,
                          d_dst, vd, bb.geom, &lf))
            return -1;
        }
        if (fbrick) {   // the finest level into the chunk buffer, as doubles
          LiftFuse lf;
          dequant_fuse(2, lf);
          lf.noMean = 1;
          const VolDesc brickVol{{cd[0], cd[1], cd[2]}};
          if (launch_lift_xyz(ss, false, boxVals, boxStride, nb, cd, d.cst, 2, bb.vals, brickVol, d_bricks, &lf))
            return -1;
        }
        if (fxyz) {   // the finest level: z, y and x pass in one kernel, into the volume
          if (mr && mr->nlev && sub_volume(2))
            return -1;
          LiftFuse lf;
          dequant_fuse(2, lf);
          if (launch_lift_xyz(ss, false, bb.vals, bb.valsStride, nb, cd, d.cst, std::is_same<T, float>::value ? 1 : 2,
                              d_dst, vd, bb.geom, &lf))
            return -1;
        }
        if (fxy && mr && mr->nlev && slice && sub_volume(1))   // the finest level of a slice is the fused pair
          return -1;
        if (fxy && launch_lift_xy(ss, false, bb.vals, bb.valsStride, nb, cd, d.cst,
                                  std::is_same<T, float>::value ? 1 : 2, d_dst, vd, bb.geom))
          return -1;
        if (batchOutliers) {   // the correctors of the values the 1D decoder found meanwhile
          HIP_CHECK(hipStreamWaitEvent(ss, E.evOutl[q], 0));
          if (launch_outlier_apply(ss, ob, d.cst, bb.vals, bb.valsStride))
            return -1;
          HIP_CHECK(hipMemcpyAsync(hoc.data(), ob.oc, nb * sizeof(OutlierChunk), hipMemcpyDeviceToHost, ss));
          HIP_CHECK(hipStreamSynchronize(ss));
          for (auto& o : hoc)
            if (o.error) {
              fprintf(stderr, "[sperr_hip] outlier decoder failed (code %u)\n", o.error);
              return -1;
            }
        }
        if ((P->fwd.empty() || batchOutliers) &&
            launch_scatter<T>(ss, d_dst, vd, bb.geom, nb, cd, bb.vals, bb.valsStride, d.cst))
          return -1;
        if (nsub > 1)
          HIP_CHECK(hipEventRecord(E.evJoin[q], ss));
        return 0;
      };
      if (nsub == 1) {
        static const bool enqTiming = getenv("SPERR_HIP_ENQ_TIMING") != nullptr;
        const auto tq0 = std::chrono::steady_clock::now();
        if (enqueue(0))
          return -1;
        if (enqTiming) {
          fprintf(stderr, "[sperr_hip] group %u x %u x %u, %u chunks: enqueued in %.2f ms\n", cd[0], cd[1], cd[2], nbAll,
                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count());
          // (diagnostics: when the group's stream is through, measured from the caller's stream at the fork)
          hipEvent_t evEnd = nullptr;
          if (deferStream && hipEventCreate(&evEnd) == hipSuccess) {
            if (!timingFork && hipEventCreate(&timingFork) == hipSuccess)
              (void)hipEventRecord(timingFork, st);
            (void)hipEventRecord(evEnd, deferStream);
            timingEnds.push_back({evEnd, {cd[0], cd[1], cd[2], nbAll}});
          }
        }
        if (deferG) {   // waited for in drain()
          pending.push_back(std::make_unique<SubHost>(std::move(subs[0])));
          continue;
        }
      }
      else {
        // one host thread per sub-batch (SPERR_HIP_ENQUEUE_THREADS=0: one thread for all): by itself no
        // gain on MI355X, but a thread of its own may wait for its stream, which lets the launcher
        // stop at the plane where the chunks run out of bits (DecPlanHost::d_live)
        std::vector<int> rc(nsub, 0);
        if (threads) {
          std::vector<std::thread> workers;
          for (uint32_t q = 0; q < nsub; q++)
            workers.emplace_back([&, q]() { rc[q] = enqueue(q); });
          for (auto& w : workers)
            w.join();
        }
        else
          for (uint32_t q = 0; q < nsub; q++)
            rc[q] = enqueue(q);
        for (uint32_t q = 0; q < nsub; q++) {
          if (rc[q])
            return -1;
          if (subs[q].nb)
            HIP_CHECK(hipStreamWaitEvent(st, E.evJoin[q], 0));
        }
      }
      // read-backs only after every sub-batch is enqueued: a device-to-host copy into pageable
      // memory blocks the host until its stream has drained
      for (uint32_t q = 0; q < nsub; q++) {
        SubHost& S = subs[q];
        if (S.nb == 0)
          continue;
        hipStream_t ss = nsub > 1 ? E.sub[q] : st;
        if (S.bb.db.lisStamps && q == 0) {
          g_lis_stamps_host.assign(64, 0);
          // (SPERR_HIP_STAMP_CHUNK=i: the counters of the batch's i-th chunk instead of the first)
          const uint32_t sc = getenv("SPERR_HIP_STAMP_CHUNK")
                                  ? std::min<uint32_t>(S.nb - 1, (uint32_t)atoi(getenv("SPERR_HIP_STAMP_CHUNK")))
                                  : 0u;
          HIP_CHECK(hipMemcpyAsync(g_lis_stamps_host.data(), S.bb.db.lisStamps + (size_t)sc * 64, 64 * 8,
                                   hipMemcpyDeviceToHost, ss));
        }
        S.hs.resize(S.nb);   // stream errors (wrong lengths) surface here
        HIP_CHECK(hipMemcpyAsync(S.hs.data(), S.bb.db.st, S.nb * sizeof(DecState),
                                 hipMemcpyDeviceToHost, ss));
      }
      HIP_CHECK(hipStreamSynchronize(st));
      if (nsub > 1)
        for (uint32_t q = 0; q < nsub; q++)
          HIP_CHECK(hipStreamSynchronize(E.sub[q]));
      for (auto& S : subs)
        for (auto& hsx : S.hs)
          if (hsx.error) {
            report_dec_error(hsx.error);
            return -1;
          }
    }
  }
  if (drain())
    return -1;
  HIP_CHECK(hipStreamSynchronize(st));
  HIP_CHECK(hipGetLastError());
  E.prof.collect();
  drainGuard.ok = true;
  return 0;
}


// ------------------------------------------------------------------------------------------
// stage access (parity tests): helpers
// ------------------------------------------------------------------------------------------
template <typename CT>
__global__ void k_msb_of(const CT* coef, int8_t* msb, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const unsigned long long v = coef[i];
  msb[i] = v ? (int8_t)(63 - __clzll((long long)v)) : (int8_t)-1;
}

// {u8 planes, u64 total_bits, payload} (SPECK_INT.cpp:284-308)
__global__ void __launch_bounds__(kThreads)
k_speck_stream_out(const CoderState* cst, const uint64_t* stream, uint8_t* dst, uint64_t cap,
                   uint64_t* len_out)
{
  const CoderState& cs = cst[0];
  const uint64_t len = cs.stream_len - 17;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    *len_out = len;
  if (len > cap)
    return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    dst[0] = (uint8_t)cs.nbp;
    memcpy(dst + 1, &cs.total_bits, 8);
  }
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 9 < len;
       i += (uint64_t)gridDim.x * blockDim.x)
    dst[9 + i] = (uint8_t)(stream[i >> 3] >> (8 * (i & 7)));
}

__global__ void k_fake_condi_header(uint8_t* dst)
{
  if (threadIdx.x || blockIdx.x)
    return;
  const double zero = 0.0, one = 1.0;
  dst[0] = 0x80;
  memcpy(dst + 1, &zero, 8);
  memcpy(dst + 9, &one, 8);
}

}  // namespace

// dynamic LDS above 64 KB has to be allowed per kernel and per device
std::vector<std::array<size_t, 6>> host_chunk_volume(const Dims3& vol, const Dims3& chunk)
{
  return chunk_volume(vol, chunk);
}

int host_parse_container(const uint8_t* p, size_t len, HostContainer& out)
{
  if (len < 18)
    return -1;
  ContainerInfo ci;
  size_t need = 0;
  if (parse_container_host(p, len, len, ci, &need) != 0)
    return -1;
  out.vol = ci.vol;
  out.chunk = ci.chunk;
  out.nvals = ci.nvals;
  out.is_float = ci.is_float;
  out.multi = ci.multi;
  out.portion = (p[1] & 0x80) != 0;
  out.off = std::move(ci.off);
  out.len = std::move(ci.len);
  return 0;
}

size_t host_chunk_stream_bound(size_t nvals, int mode, double quality)
{
  const double n = (double)nvals;
  const uint64_t raw = mode == 1 ? (uint64_t)(quality * n) : 0;
  // without a budget every plane is coded: at most 66 bits per sample and 64 tests per set, and
  // a chunk has fewer sets than samples (the bound of max_payload_bits, engine-side)
  const uint64_t unlimited = (uint64_t)(130.0 * n) + 64;
  const uint64_t bits = raw ? std::min(rounded_budget(raw), unlimited) : unlimited;
  size_t total = 26 + (size_t)((bits + 7) / 8) + 8;
  if (mode == 3)   // the outlier stream: the same bound for the 1D coder over n values
    total += 9 + (size_t)((unlimited + 7) / 8);
  return total;
}

int set_max_dyn_lds(const void* fn, int bytes)
{
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> done;
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find({dev, fn});
  if (it != done.end() && it->second >= bytes)
    return 0;
  HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done[{dev, fn}] = bytes;
  return 0;
}

}  // namespace sperrhip

// ==========================================================================================
// C ABI
// ==========================================================================================
using namespace sperrhip;

// nothing thrown by the host side (std::bad_alloc of a vector, std::system_error of a thread)
// may cross the C boundary
template <typename F>
static int guarded(const char* what, F&& body) noexcept
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

extern "C" {

// Gives back what the library keeps between calls: the workspaces and shape tables of every idle
// engine, the staging and device buffers of every idle farm worker, the calling thread's slice
// buffers.  Engines and workers in use by other threads are left alone.  For hosts that embed the
// library (an HDF5 filter, a long-running service) and call it rarely.
static void thread_slice_bufs_drop();
void sperrhip_release(void)
{
  (void)guarded("sperrhip_release", [&]() -> int {
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    // The idle engines are picked (and marked busy, so nobody leases them) under the pool's lock;
    // their memory goes with the lock released.  `all` may grow meanwhile (acquire() on another
    // thread): only the raw pointers collected here are used, the engines themselves never move.
    std::vector<Engine*> idle;
    {
      std::lock_guard<std::mutex> lock(g_pool.mu);
      idle.reserve(g_pool.all.size());
      for (auto& e : g_pool.all)
        if (!e->busy && e->dev >= 0) {
          e->busy = true;
          idle.push_back(e.get());
        }
    }
    for (Engine* e : idle)
      if (hipSetDevice(e->dev) == hipSuccess)
        e->drop_memory();
    {
      std::lock_guard<std::mutex> lock(g_pool.mu);
      for (Engine* e : idle)
        e->busy = false;
    }
    g_pool.cv.notify_all();
    farm_release_idle();
    if (have)
      (void)hipSetDevice(cur);
    thread_slice_bufs_drop();
    return 0;
  });
}

const char* sperrhip_version(void)
{
  return "sperr_hip 0.1 (gfx950; SPERR bitstream major version 0)";
}

void sperrhip_profile_enable(int on)
{
  std::lock_guard<std::mutex> lock(g_prof_cfg_mu);
  g_prof_on = on != 0;
}
void sperrhip_profile_only(const char* kernel)
{
  std::lock_guard<std::mutex> lock(g_prof_cfg_mu);
  g_prof_only = kernel ? kernel : "";
}
unsigned long long sperrhip_debug_counter(int which)
{
  if (which == 3) {   // bytes of the largest workspace arena an engine of this process holds
    std::lock_guard<std::mutex> lock(g_pool.mu);
    size_t most = 0;
    for (auto& e : g_pool.all)
      most = std::max(most, e->arena.n);
    return (unsigned long long)most;
  }
  if (which == 6) {   // bytes of ALL device buffers of the engine that holds most (arena + slots + the PWE buffers)
    std::lock_guard<std::mutex> lock(g_pool.mu);
    size_t most = 0;
    for (auto& e : g_pool.all)
      most = std::max(most, e->device_bytes());
    return (unsigned long long)most;
  }
  if (which == 4 || which == 5)   // pinned staging / device bytes the farm's workers hold right now
    return farm_footprint(which == 4);
  return which >= 0 && which < 3 ? g_dbg_counter[which].load() : 0ull;
}
void sperrhip_debug_lis_stamps(int on, unsigned long long* out64)
{
  g_lis_stamps_on = on != 0;
  if (out64)
    for (size_t i = 0; i < 64; i++)
      out64[i] = i < g_lis_stamps_host.size() ? g_lis_stamps_host[i] : 0;
}
void sperrhip_profile_reset(void)
{
  std::lock_guard<std::mutex> lock(g_pool.mu);
  for (auto& e : g_pool.all) {
    std::lock_guard<std::mutex> l2(e->prof.mu);
    e->prof.acc.clear();
  }
}
int sperrhip_profile_get(const char** names, double* millis, int* launches, int cap)
{
  return guarded("sperrhip_profile_get", [&]() -> int {
    return sperrhip_profile_get2(names, millis, nullptr, launches, cap);
  });
}
int sperrhip_profile_get2(const char** names, double* busy_millis, double* sum_millis, int* launches,
                          int cap)
{
  return guarded("sperrhip_profile_get2", [&]() -> int {
    // the engines' tables merged; the names stay valid until the next call
    static std::mutex mu;
    static std::map<std::string, ProfEntry> merged;
    std::lock_guard<std::mutex> lock(mu);
    merged.clear();
    {
      std::lock_guard<std::mutex> lp(g_pool.mu);
      for (auto& e : g_pool.all) {
        std::lock_guard<std::mutex> l2(e->prof.mu);
        for (auto& kv : e->prof.acc) {
          ProfEntry& m = merged[kv.first];
          m.ms += kv.second.ms;
          m.busy += kv.second.busy;
          m.launches += kv.second.launches;
        }
      }
    }
    int i = 0;
    for (auto& kv : merged) {
      if (i < cap) {
        names[i] = kv.first.c_str();
        busy_millis[i] = kv.second.busy;
        if (sum_millis)
          sum_millis[i] = kv.second.ms;
        launches[i] = kv.second.launches;
      }
      i++;
    }
    return i;
  });
}

size_t sperrhip_max_compressed_size(size_t dimx, size_t dimy, size_t dimz, size_t chunk_x,
                                    size_t chunk_y, size_t chunk_z, int mode, double quality)
{
  const Dims vol{dimx, dimy, dimz};
  Dims cd{chunk_x, chunk_y, chunk_z};
  for (int a = 0; a < 3; a++)
    cd[a] = std::min(std::max<size_t>(1, cd[a]), vol[a]);
  const auto chunks = chunk_volume(vol, cd);
  size_t total = 20 + 4 * chunks.size();
  for (auto& c : chunks)
    total += host_chunk_stream_bound(c[1] * c[3] * c[5], mode, quality);
  return total;
}

int sperrhip_compress_dev(const void* d_src, int is_float, size_t dimx, size_t dimy, size_t dimz,
                          size_t chunk_x, size_t chunk_y, size_t chunk_z, int mode, double quality,
                          void* d_dst, size_t dst_cap, size_t* dst_len, void* hip_stream)
{
  return guarded("sperrhip_compress_dev", [&]() -> int {
    if (quality <= 0.0)
      return 2;
    if (mode < 1 || mode > 3)
      return 2;
    if (!d_src || !d_dst || !dst_len || dimx == 0 || dimy == 0 || dimz == 0)
      return -1;
    Lease L;
    if (!L.e)
      return -1;
    Engine& E = *L.e;
    const Dims vol{dimx, dimy, dimz}, ch{chunk_x, chunk_y, chunk_z};
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (is_float)
      return compress_impl<float>(E, static_cast<const float*>(d_src), vol, ch, mode, quality,
                                  static_cast<uint8_t*>(d_dst), dst_cap, dst_len, st);
    return compress_impl<double>(E, static_cast<const double*>(d_src), vol, ch, mode, quality,
                                 static_cast<uint8_t*>(d_dst), dst_cap, dst_len, st);
  });
}

int sperrhip_parse_header_dev(const void* d_src, size_t src_len, size_t* dimx, size_t* dimy,
                              size_t* dimz, int* is_float, size_t* chunk_x, size_t* chunk_y,
                              size_t* chunk_z)
{
  return guarded("sperrhip_parse_header_dev", [&]() -> int {
    Lease L;
    if (!L.e)
      return -1;
    Engine& E = *L.e;
    ContainerInfo ci;
    if (read_container_info(static_cast<const uint8_t*>(d_src), src_len, ci, nullptr))
      return -1;
    *dimx = ci.vol[0];
    *dimy = ci.vol[1];
    *dimz = ci.vol[2];
    *is_float = ci.is_float ? 1 : 0;
    if (chunk_x)
      *chunk_x = ci.chunk[0];
    if (chunk_y)
      *chunk_y = ci.chunk[1];
    if (chunk_z)
      *chunk_z = ci.chunk[2];
    return 0;
  });
}

int sperrhip_decompress_dev(const void* d_src, size_t src_len, int output_float, void* d_dst,
                            size_t dst_cap_bytes, size_t* dimx, size_t* dimy, size_t* dimz,
                            void* hip_stream)
{
  return guarded("sperrhip_decompress_dev", [&]() -> int {
    if (!d_src || !d_dst)
      return -1;
    Lease L;
    if (!L.e)
      return -1;
    Engine& E = *L.e;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    ContainerInfo ci;
    if (read_container_info(static_cast<const uint8_t*>(d_src), src_len, ci, st))
      return -1;
    if (dimx)
      *dimx = ci.vol[0];
    if (dimy)
      *dimy = ci.vol[1];
    if (dimz)
      *dimz = ci.vol[2];
    if (output_float)
      return decompress_impl<float>(E, static_cast<const uint8_t*>(d_src), src_len,
                                    static_cast<float*>(d_dst), dst_cap_bytes / sizeof(float), ci, st);
    return decompress_impl<double>(E, static_cast<const uint8_t*>(d_src), src_len,
                                   static_cast<double*>(d_dst), dst_cap_bytes / sizeof(double), ci,
                                   st);
  });
}

int sperrhip_multires_levels(size_t dimx, size_t dimy, size_t dimz, size_t chunk_x, size_t chunk_y,
                             size_t chunk_z, size_t* nlev, size_t* level_dims)
{
  return guarded("sperrhip_multires_levels", [&]() -> int {
    const Dims vol{dimx, dimy, dimz};
    Dims cd{chunk_x, chunk_y, chunk_z};
    if (!nlev || dimx == 0 || dimy == 0 || dimz == 0)
      return -1;
    for (int a = 0; a < 3; a++)
      cd[a] = std::min(std::max<size_t>(1, cd[a]), vol[a]);
    MultiRes m;
    multires_levels(vol, cd, m);
    *nlev = m.nlev;
    if (level_dims)
      for (size_t h = 0; h < m.nlev; h++)
        for (int a = 0; a < 3; a++)
          level_dims[3 * h + a] = (size_t)m.cres[h][a] * m.grid[a];
    return 0;
  });
}

int sperrhip_decompress_multires_dev(const void* d_src, size_t src_len, int output_float,
                                     void* d_dst, size_t dst_cap_bytes, size_t nlev,
                                     double* const* d_levels, void* hip_stream)
{
  return guarded("sperrhip_decompress_multires_dev", [&]() -> int {
    if (!d_src || !d_dst || (nlev && !d_levels))
      return -1;
    Lease L;
    if (!L.e)
      return -1;
    Engine& E = *L.e;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    ContainerInfo ci;
    if (read_container_info(static_cast<const uint8_t*>(d_src), src_len, ci, st))
      return -1;
    MultiRes m;
    multires_levels(ci.vol, ci.chunk, m);
    if (m.nlev != nlev)
      return -1;   // (sperrhip_multires_levels tells how many there are)
    for (size_t h = 0; h < nlev; h++) {
      if (!d_levels[h])
        return -1;
      m.d_level[h] = d_levels[h];
    }
    if (output_float)
      return decompress_impl<float>(E, static_cast<const uint8_t*>(d_src), src_len,
                                    static_cast<float*>(d_dst), dst_cap_bytes / sizeof(float), ci, st,
                                    &m);
    return decompress_impl<double>(E, static_cast<const uint8_t*>(d_src), src_len,
                                   static_cast<double*>(d_dst), dst_cap_bytes / sizeof(double), ci,
                                   st, &m);
  });
}

// What a host thread that codes slices keeps: a stream of its own (calls of several threads then run
// side by side; on the null stream they would queue behind each other) and two device buffers that
// only grow (hipMalloc / hipFree per call would synchronise the device; the stream-ordered allocator
// handed out blocks whose contents the next call did not see: not used).  Freed with the thread.
struct ThreadSliceBufs {
  int dev = -1;          // the device the stream and the buffers belong to
  hipStream_t s = nullptr;
  void* p[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  hipStream_t stream()
  {
    if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
      s = nullptr;
    return s;
  }
  void* get(int i, size_t bytes)
  {
    if (bytes > cap[i]) {
      if (p[i])
        (void)hipFree(p[i]);
      p[i] = nullptr;
      cap[i] = 0;
      if (hipMalloc(&p[i], round_up(bytes, 1 << 20)) != hipSuccess)
        return nullptr;
      cap[i] = round_up(bytes, 1 << 20);
    }
    return p[i];
  }
  void drop()
  {
    for (int i = 0; i < 2; i++) {
      if (p[i])
        (void)hipFree(p[i]);
      p[i] = nullptr;
      cap[i] = 0;
    }
    if (s)
      (void)hipStreamDestroy(s);
    s = nullptr;
  }
};
// One set per device the thread has coded slices on: a thread that moves to another device
// (hipSetDevice between two calls) leases an engine of THAT device, whose arena, sub-streams and
// events must not meet a stream or buffers of the device it came from.
struct ThreadSliceCache {
  std::vector<ThreadSliceBufs> sets;
  ThreadSliceBufs& of_current_device()
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
      dev = 0;
    for (auto& b : sets)
      if (b.dev == dev)
        return b;
    sets.emplace_back();
    sets.back().dev = dev;
    return sets.back();
  }
  ~ThreadSliceCache()
  {
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (auto& b : sets) {
      if (have && b.dev != cur)
        (void)hipSetDevice(b.dev);
      b.drop();
    }
    if (have)
      (void)hipSetDevice(cur);
  }
};
static ThreadSliceCache& thread_slice_cache()
{
  static thread_local ThreadSliceCache t;
  return t;
}
static ThreadSliceBufs& thread_slice_bufs()
{
  return thread_slice_cache().of_current_device();
}
static void thread_slice_bufs_drop()
{
  ThreadSliceCache& c = thread_slice_cache();
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  for (auto& b : c.sets) {
    if (have && b.dev != cur)
      (void)hipSetDevice(b.dev);
    if (b.s)
      (void)hipStreamSynchronize(b.s);
    b.drop();
  }
  c.sets.clear();
  if (have)
    (void)hipSetDevice(cur);
}

// ---- 2D slices (include/SPERR_C_API.h:53-81, src/SPERR_C_API.cpp:7-134) ------------------------
size_t sperrhip_max_compressed_size_2d(size_t dimx, size_t dimy, int mode, double quality)
{
  return 10 + sperrhip_max_compressed_size(dimx, dimy, 1, dimx, dimy, 1, mode, quality);
}

int sperrhip_compress_2d_dev(const void* d_src, int is_float, size_t dimx, size_t dimy, int mode,
                             double quality, int out_inc_header, void* d_dst, size_t dst_cap,
                             size_t* dst_len, void* hip_stream)
{
  return guarded("sperrhip_compress_2d_dev", [&]() -> int {
    if (quality <= 0.0)
      return 2;
    if (mode < 1 || mode > 3)
      return 2;
    if (!d_src || !d_dst || !dst_len || dimx == 0 || dimy == 0)
      return -1;
    Lease L;
    if (!L.e)
      return -1;
    Engine& E = *L.e;
    const Dims vol{dimx, dimy, 1};
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int slice = out_inc_header ? 2 : 1;
    if (is_float)
      return compress_impl<float>(E, static_cast<const float*>(d_src), vol, vol, mode, quality,
                                  static_cast<uint8_t*>(d_dst), dst_cap, dst_len, st, slice);
    return compress_impl<double>(E, static_cast<const double*>(d_src), vol, vol, mode, quality,
                                 static_cast<uint8_t*>(d_dst), dst_cap, dst_len, st, slice);
  });
}

// d_src: the stream WITHOUT the optional 10-byte header, as sperr_decomp_2d takes it
int sperrhip_decompress_2d_dev(const void* d_src, size_t src_len, int output_float, size_t dimx,
                               size_t dimy, void* d_dst, size_t dst_cap_bytes, void* hip_stream)
{
  return guarded("sperrhip_decompress_2d_dev", [&]() -> int {
    if (!d_src || !d_dst || dimx == 0 || dimy == 0 || src_len < 17)
      return -1;
    Lease L;
    if (!L.e)
      return -1;
    Engine& E = *L.e;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    ContainerInfo ci;
    ci.vol = {dimx, dimy, 1};
    ci.nvals = dimx * dimy;
    ci.chunk = ci.vol;
    ci.is_float = output_float != 0;
    ci.off = {0};
    ci.len = {src_len};
    if (output_float)
      return decompress_impl<float>(E, static_cast<const uint8_t*>(d_src), src_len,
                                    static_cast<float*>(d_dst), dst_cap_bytes / sizeof(float), ci, st,
                                    nullptr, true);
    return decompress_impl<double>(E, static_cast<const uint8_t*>(d_src), src_len,
                                   static_cast<double*>(d_dst), dst_cap_bytes / sizeof(double), ci, st,
                                   nullptr, true);
  });
}

// ---- stage access for parity tests -----------------------------------------------------------

int sperrhip_dwt3d_dev(double* d_vals, size_t dimx, size_t dimy, size_t dimz, int inverse,
                       void* hip_stream)
{
  return guarded("sperrhip_dwt3d_dev", [&]() -> int {
    Lease L;
    if (!L.e)
      return -1;
    Engine& E = *L.e;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    ShapePlan* P = E.plan(dimx, dimy, dimz);
    if (!P || E.misc.ensure(4096))
      return -1;
    CoderState* cst = static_cast<CoderState*>(E.misc.p);
    HIP_CHECK(hipMemsetAsync(cst, 0, sizeof(CoderState), st));
    const uint32_t cd[3] = {P->dims[0], P->dims[1], P->dims[2]};
    if (!inverse) {
      for (const LiftPass& ps : P->fwd)
        if (launch_lift(st, true, d_vals, P->N, 1, cd, ps.axis, ps.region, cst))
          return -1;
    }
    else {
      for (size_t k = P->fwd.size(); k-- > 0;)
        if (launch_lift(st, false, d_vals, P->N, 1, cd, P->fwd[k].axis, P->fwd[k].region, cst))
          return -1;
    }
    HIP_CHECK(hipStreamSynchronize(st));
    E.prof.collect();
    return 0;
  });
}

int sperrhip_speck3d_encode_dev(const void* d_coef, int width, const uint64_t* d_sign, size_t dimx,
                                size_t dimy, size_t dimz, size_t budget_bits, void* d_dst,
                                size_t dst_cap, size_t* dst_len, void* hip_stream)
{
  return guarded("sperrhip_speck3d_encode_dev", [&]() -> int {
    if (width != 4 && width != 8)
      return 2;
    Lease L;
    if (!L.e)
      return -1;
    Engine& E = *L.e;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    ShapePlan* P = E.plan(dimx, dimy, dimz);
    if (!P)
      return -1;
    const uint64_t raw_budget = budget_bits;
    if (E.arena.ensure(enc_bytes_per_chunk(*P, raw_budget) + 4096) || E.misc.ensure(4096))
      return -1;
    Arena A;
    A.base = static_cast<char*>(E.arena.p);
    A.cap = E.arena.n;
    EncBatchBufs bb;
    if (!carve_enc(A, *P, 1, raw_budget, bb))
      return -1;
    const bool wide = width == 8;
    if (wide && bb.aliased && unalias_coder(st, E, *P, bb, 1))   // (the 64-bit magnitudes go into the chunk buffer)
      return -1;
    EncBuffers e = bb.eb;
    CoderState hcs;
    memset(&hcs, 0, sizeof(hcs));
    hcs.need_retry = wide ? 1u : 0u;  // the 64-bit pass only encodes chunks flagged for it
    hcs.wide = wide ? 1u : 0u;
    HIP_CHECK(hipMemcpyAsync(e.cst, &hcs, sizeof(CoderState), hipMemcpyHostToDevice, st));
    if (reset_enc_pass(st, bb, 1))
      return -1;
    const uint32_t n = P->N;
    HIP_CHECK(hipMemcpyAsync(const_cast<uint64_t*>(e.sign), d_sign, ((n + 63) / 64) * 8,
                             hipMemcpyDeviceToDevice, st));
    if (wide) {
      HIP_CHECK(hipMemcpyAsync(bb.vals, d_coef, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
      e.coef = bb.vals;
      e.coefStride = bb.valsStride;
      LAUNCH_K(k_msb_of<uint64_t>, dim3((n + 255) / 256), dim3(256), 0, st,
               reinterpret_cast<const uint64_t*>(bb.vals), bb.msb, n);
    }
    else {
      HIP_CHECK(hipMemcpyAsync(bb.coef32, d_coef, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
      LAUNCH_K(k_msb_of<uint32_t>, dim3((n + 255) / 256), dim3(256), 0, st,
               reinterpret_cast<const uint32_t*>(bb.coef32), bb.msb, n);
    }
    EncPlanHost ph{P->d_initLIS, P->d_initLen, P->d_depthBlocks, P->depthBlockOff, P->ht.nsets};
    if (launch_speck_encode(st, e, ph, raw_budget, false, wide))
      return -1;
    uint64_t* d_len = static_cast<uint64_t*>(E.misc.p);
    LAUNCH_K(k_speck_stream_out, dim3(1024), dim3(kThreads), 0, st, e.cst, e.stream,
             static_cast<uint8_t*>(d_dst), (uint64_t)dst_cap, d_len);
    uint64_t len = 0;
    HIP_CHECK(hipMemcpyAsync(&len, d_len, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipGetLastError());
    E.prof.collect();
    if (len > dst_cap)
      return -1;
    *dst_len = (size_t)len;
    return 0;
  });
}

int sperrhip_speck3d_decode_dev(const void* d_stream, size_t stream_len, size_t dimx, size_t dimy,
                                size_t dimz, void* d_coef, uint64_t* d_sign, int* width_out,
                                void* hip_stream)
{
  return guarded("sperrhip_speck3d_decode_dev", [&]() -> int {
    if (stream_len < 9)
      return -1;
    Lease L;
    if (!L.e)
      return -1;
    Engine& E = *L.e;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    ShapePlan* P = E.plan(dimx, dimy, dimz);
    if (!P)
      return -1;
    uint8_t head[9];
    HIP_CHECK(hipMemcpyAsync(head, d_stream, 9, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const int nbp = head[0];
    const bool wide = nbp > 32;
    if (nbp > kMaxPlanes)
      return -1;
    // wrap the bare SPECK stream into a chunk stream with a dummy conditioner header
    if (E.misc.ensure(round_up(17 + stream_len + 64, 256)))
      return -1;
    uint8_t* wrap = static_cast<uint8_t*>(E.misc.p);
    LAUNCH_K(k_fake_condi_header, dim3(1), dim3(1), 0, st, wrap);
    HIP_CHECK(hipMemcpyAsync(wrap + 17, d_stream, stream_len, hipMemcpyDeviceToDevice, st));
    Arena probe;
    probe.base = reinterpret_cast<char*>(uintptr_t(4096));  // size probe only
    probe.cap = ~size_t(0) / 2;
    DecBatchBufs tmp;
    // (refinement bit planes like decompress_impl's: SPERR_HIP_REF_PLANES=0 switches them off)
    static const bool refPlanesEnv = !(tune_getenv("SPERR_HIP_REF_PLANES") && atoi(tune_getenv("SPERR_HIP_REF_PLANES")) == 0);
    const uint32_t refNPlanes = (refPlanesEnv && !wide) ? (uint32_t)nbp : 0u;
    carve_dec(probe, *P, 1, 17 + stream_len, tmp, 0, refNPlanes);
    if (E.arena.ensure(probe.used + 4096))
      return -1;
    Arena A;
    A.base = static_cast<char*>(E.arena.p);
    A.cap = E.arena.n;
    DecBatchBufs bb;
    if (!carve_dec(A, *P, 1, 17 + stream_len, bb, 0, refNPlanes))
      return -1;
    DecBuffers d = bb.db;
    const uint64_t off = 0, len = 17 + stream_len;
    HIP_CHECK(hipMemcpyAsync(bb.chunkOff, &off, 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(bb.chunkLen, &len, 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemsetAsync(d.cst, 0, sizeof(CoderState), st));
    HIP_CHECK(hipMemsetAsync(d.st, 0, sizeof(DecState), st));
    HIP_CHECK(hipMemsetAsync(d.bornM, 0, d.maskPixStride * 8, st));
    if (d.tileBorn)
      HIP_CHECK(hipMemsetAsync(d.tileBorn, 0, d.tileStride, st));
    HIP_CHECK(hipMemsetAsync(d.sigOld, 0, d.maskPixStride * 8, st));
    HIP_CHECK(hipMemsetAsync(d.sigNew, 0, d.maskPixStride * 8, st));
    HIP_CHECK(hipMemsetAsync(d.sign, 0xff, d.signStride * 8, st));
    HIP_CHECK(hipMemsetAsync(d.leafState, 0, d.leafStateStride * 2, st));
    HIP_CHECK(hipMemsetAsync(d.leafDirty, 0, d.leafDirtyStride, st));
    HIP_CHECK(hipMemsetAsync(d.stream, 0, d.streamStride * 8, st));
    const uint32_t n = P->N;
    if (wide) {
      d.coef = bb.vals;
      d.coefStride = bb.valsStride;
      HIP_CHECK(hipMemsetAsync(bb.vals, 0, (size_t)n * 8, st));
    }
    else if (d.refPlanes)
      HIP_CHECK(hipMemsetAsync(d.wordTop, 0, d.wordTopStride, st));
    else
      HIP_CHECK(hipMemsetAsync(bb.coef32, 0, (size_t)n * 4, st));
    DecPlanHost ph{P->d_initLIS, P->d_initLen, use_tables(*P),
                   P->l0Level >= 0 && P->ht.grids.size() <= 288, P->l1Level >= 0 && P->ht.grids.size() <= 288, P->maxK};
    ph.l2 = ph.l1 && P->l2Level >= 0;
    ph.hi = use_lis_hi(*P, ph.tables);
    ph.mixed = use_mixed(*P);
    HIP_CHECK(hipMemsetAsync(d.mask, 0, std::max<size_t>(d.maskStride, 1) * 8, st));
    HIP_CHECK(hipMemsetAsync(d.l0Flags, 0, d.l0FlagStride * 8, st));
    if (d.l0Tab)
      HIP_CHECK(hipMemsetAsync(d.l0Tab, 0, d.l0FlagStride * 17 * 8, st));
    HIP_CHECK(hipMemsetAsync(d.l1Flags, 0, d.l0FlagStride * 8, st));
    HIP_CHECK(hipMemsetAsync(d.l2Flags, 0, d.l0FlagStride * 8, st));
    HIP_CHECK(hipMemsetAsync(d.hiFlags, 0, d.hiFlagStride * 8, st));
    HIP_CHECK(hipMemsetAsync(d.sigbits, 0, d.sigbitsStride * 8, st));
    if (launch_speck_decode(st, d, ph, wrap, bb.chunkOff, bb.chunkLen, wide, nbp))
      return -1;
    HIP_CHECK(hipMemcpyAsync(d_coef, d.coef, (size_t)n * (wide ? 8 : 4), hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipMemcpyAsync(d_sign, d.sign, ((n + 63) / 64) * 8, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipGetLastError());
    E.prof.collect();
    *width_out = wide ? 8 : 4;
    return 0;
  });
}

// ---- reference-compatible host API (src/SPERR_C_API.cpp:135-258) ---------------------------

void sperr_parse_header(const void* src, size_t* dimx, size_t* dimy, size_t* dimz, int* is_float)
{
  const uint8_t* p = static_cast<const uint8_t*>(src);
  const bool is_3d = (p[1] & 0x40) != 0;
  *is_float = (p[1] & 0x20) ? 1 : 0;
  uint32_t d[3] = {1, 1, 1};
  memcpy(d, p + 2, is_3d ? 12 : 8);
  *dimx = d[0];
  *dimy = d[1];
  *dimz = d[2];
}

// include/SPERR_C_API.h:138-156, src/SPERR_C_API.cpp:260-280,
// src/SPERR3D_Stream_Tools.cpp:134-226: host-side byte surgery, no GPU involved.  Every chunk
// stream keeps `pct` percent of its bytes (at least 64, at most what it has), the container is
// flagged as a portion and the chunk lengths are rewritten; the decoder zero-pads what is missing.
int sperr_trunc_3d(const void* src, size_t src_len, unsigned pct, void** dst, size_t* dst_len)
{
  return guarded("sperr_trunc_3d", [&]() -> int {
    return hostc::truncate_container(static_cast<const uint8_t*>(src), src_len, pct, dst, dst_len);
  });
}

// include/SPERR_C_API.h:53-62, src/SPERR_C_API.cpp:7-97
int sperr_comp_2d(const void* src, int is_float, size_t dimx, size_t dimy, int mode, double quality,
                  int out_inc_header, void** dst, size_t* dst_len)
{
  return guarded("sperr_comp_2d", [&]() -> int {
    if (*dst != nullptr)
      return 1;
    if (quality <= 0.0)
      return 2;
    if (mode < 1 || mode > 3)
      return 2;
    const size_t n = dimx * dimy, esz = is_float ? 4 : 8;
    const size_t cap = sperrhip_max_compressed_size_2d(dimx, dimy, mode, quality);
    // on the calling thread's own stream, with its own device buffers: slices coded from several host
    // threads run side by side
    ThreadSliceBufs& tb = thread_slice_bufs();
    hipStream_t ts = tb.stream();
    void *d_in = tb.get(0, n * esz), *d_out = tb.get(1, cap);
    if (!d_in || !d_out) {
      fprintf(stderr, "[sperr_hip] device allocation failed\n");
      return -1;
    }
    int rtn = -1;
    size_t len = 0;
    if (hipMemcpyAsync(d_in, src, n * esz, hipMemcpyHostToDevice, ts) == hipSuccess)
      rtn = sperrhip_compress_2d_dev(d_in, is_float, dimx, dimy, mode, quality, out_inc_header, d_out,
                                     cap, &len, ts);
    if (rtn == 0) {
      void* buf = malloc(len);
      if (buf && hipMemcpyAsync(buf, d_out, len, hipMemcpyDeviceToHost, ts) == hipSuccess &&
          hipStreamSynchronize(ts) == hipSuccess) {
        *dst = buf;
        *dst_len = len;
      }
      else {
        free(buf);
        rtn = -1;
      }
    }
    (void)hipStreamSynchronize(ts);
    return rtn;
  });
}

// include/SPERR_C_API.h:75-81, src/SPERR_C_API.cpp:99-134
int sperrhip_multires_levels_2d(size_t dimx, size_t dimy, size_t* nlev, size_t* level_dims)
{
  return guarded("sperrhip_multires_levels_2d", [&]() -> int {
    if (!nlev || !level_dims || dimx == 0 || dimy == 0 || dimx > 0xffff || dimy > 0xffff)
      return -1;
    MultiRes m;
    multires_levels_2d(dimx, dimy, m);
    *nlev = m.nlev;
    for (size_t h = 0; h < m.nlev; h++) {
      level_dims[2 * h] = m.cres[h][0];
      level_dims[2 * h + 1] = m.cres[h][1];
    }
    return 0;
  });
}

int sperrhip_decompress_2d_multires_dev(const void* d_src, size_t src_len, int output_float, size_t dimx,
                                        size_t dimy, void* d_dst, size_t dst_cap_bytes, size_t nlev,
                                        double* const* d_levels, void* hip_stream)
{
  return guarded("sperrhip_decompress_2d_multires_dev", [&]() -> int {
    if (!d_src || !d_dst || dimx == 0 || dimy == 0 || src_len < 17 || (nlev && !d_levels))
      return -1;
    Lease L;
    if (!L.e)
      return -1;
    Engine& E = *L.e;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    MultiRes m;
    multires_levels_2d(dimx, dimy, m);
    if (m.nlev != nlev)
      return -1;   // (sperrhip_multires_levels_2d tells how many there are)
    for (size_t h = 0; h < nlev; h++) {
      if (!d_levels[h])
        return -1;
      m.d_level[h] = d_levels[h];
    }
    ContainerInfo ci;
    ci.vol = {dimx, dimy, 1};
    ci.nvals = dimx * dimy;
    ci.chunk = ci.vol;
    ci.is_float = output_float != 0;
    ci.off = {0};
    ci.len = {src_len};
    if (output_float)
      return decompress_impl<float>(E, static_cast<const uint8_t*>(d_src), src_len,
                                    static_cast<float*>(d_dst), dst_cap_bytes / sizeof(float), ci, st, &m,
                                    true);
    return decompress_impl<double>(E, static_cast<const uint8_t*>(d_src), src_len,
                                   static_cast<double*>(d_dst), dst_cap_bytes / sizeof(double), ci, st, &m,
                                   true);
  });
}

int sperrhip_decomp_2d_multires(const void* src, size_t src_len, int output_float, size_t dimx,
                                size_t dimy, void** dst, size_t* nlev, size_t* level_dims, double** levels)
{
  return guarded("sperrhip_decomp_2d_multires", [&]() -> int {
    if (!dst || *dst != nullptr)
      return 1;
    if (src_len < 17 || !nlev || !level_dims || !levels ||
        sperrhip_multires_levels_2d(dimx, dimy, nlev, level_dims))
      return -1;
    const size_t n = dimx * dimy, esz = output_float ? 4 : 8;
    std::vector<void*> dev;
    auto release = [&]() {
      for (void* p : dev)
        (void)hipFree(p);
    };
    auto dalloc = [&](size_t bytes) -> void* {
      void* p = nullptr;
      if (hipMalloc(&p, bytes) != hipSuccess)
        return nullptr;
      dev.push_back(p);
      return p;
    };
    void* d_in = dalloc(src_len);
    void* d_out = dalloc(n * esz);
    std::vector<double*> d_lv(*nlev, nullptr);
    std::vector<size_t> lvn(*nlev, 0);
    bool ok = d_in && d_out;
    for (size_t h = 0; ok && h < *nlev; h++) {
      lvn[h] = level_dims[2 * h] * level_dims[2 * h + 1];
      d_lv[h] = static_cast<double*>(dalloc(lvn[h] * 8));
      ok = d_lv[h] != nullptr;
    }
    if (!ok) {
      fprintf(stderr, "[sperr_hip] device allocation failed\n");
      release();
      return -1;
    }
    int rtn = -1;
    if (hipMemcpy(d_in, src, src_len, hipMemcpyHostToDevice) == hipSuccess)
      rtn = sperrhip_decompress_2d_multires_dev(d_in, src_len, output_float, dimx, dimy, d_out, n * esz, *nlev,
                                                d_lv.data(), nullptr);
    if (rtn == 0) {
      void* buf = malloc(n * esz);
      if (buf && hipMemcpy(buf, d_out, n * esz, hipMemcpyDeviceToHost) == hipSuccess)
        *dst = buf;
      else {
        free(buf);
        rtn = -1;
      }
      for (size_t h = 0; rtn == 0 && h < *nlev; h++) {
        levels[h] = static_cast<double*>(malloc(lvn[h] * 8));
        if (!levels[h] || hipMemcpy(levels[h], d_lv[h], lvn[h] * 8, hipMemcpyDeviceToHost) != hipSuccess)
          rtn = -1;
      }
    }
    release();
    return rtn;
  });
}

int sperr_decomp_2d(const void* src, size_t src_len, int output_float, size_t dimx, size_t dimy,
                    void** dst)
{
  return guarded("sperr_decomp_2d", [&]() -> int {
    if (*dst != nullptr)
      return 1;
    const size_t n = dimx * dimy, esz = output_float ? 4 : 8;
    ThreadSliceBufs& tb = thread_slice_bufs();   // (see sperr_comp_2d)
    hipStream_t ts = tb.stream();
    void *d_in = src_len >= 17 ? tb.get(0, src_len) : nullptr, *d_out = tb.get(1, n * esz);
    if (!d_in || !d_out)
      return -1;
    int rtn = -1;
    if (hipMemcpyAsync(d_in, src, src_len, hipMemcpyHostToDevice, ts) == hipSuccess)
      rtn = sperrhip_decompress_2d_dev(d_in, src_len, output_float, dimx, dimy, d_out, n * esz, ts);
    if (rtn == 0) {
      void* buf = malloc(n * esz);
      if (buf && hipMemcpyAsync(buf, d_out, n * esz, hipMemcpyDeviceToHost, ts) == hipSuccess &&
          hipStreamSynchronize(ts) == hipSuccess)
        *dst = buf;
      else {
        free(buf);
        rtn = -1;
      }
    }
    (void)hipStreamSynchronize(ts);
    return rtn;
  });
}

// host buffers in, malloc'd host buffers out: the volume and the levels of the hierarchy
// (SPERR3D_OMP_D::decompress(p, true) + release_decoded_data / release_hierarchy)
int sperrhip_decomp_3d_multires(const void* src, size_t src_len, int output_float, size_t* dimx,
                                size_t* dimy, size_t* dimz, void** dst, size_t* nlev,
                                size_t* level_dims, double** levels)
{
  return guarded("sperrhip_decomp_3d_multires", [&]() -> int {
    if (!dst || *dst != nullptr)
      return 1;
    if (src_len < 18 || !nlev || !level_dims || !levels)
      return -1;
    ContainerInfo ci;
    size_t need = 0;
    if (parse_container_host(static_cast<const uint8_t*>(src), src_len, src_len, ci, &need) != 0)
      return -1;
    if (sperrhip_multires_levels(ci.vol[0], ci.vol[1], ci.vol[2], ci.chunk[0], ci.chunk[1], ci.chunk[2],
                                 nlev, level_dims))
      return -1;
    const size_t n = ci.nvals;
    const size_t esz = output_float ? 4 : 8;
    std::vector<void*> dev;
    auto release = [&]() {
      for (void* p : dev)
        (void)hipFree(p);
    };
    auto dalloc = [&](size_t bytes) -> void* {
      void* p = nullptr;
      if (hipMalloc(&p, bytes) != hipSuccess)
        return nullptr;
      dev.push_back(p);
      return p;
    };
    void* d_in = dalloc(src_len);
    void* d_out = dalloc(n * esz);
    std::vector<double*> d_lv(*nlev, nullptr);
    std::vector<size_t> lvn(*nlev, 0);
    bool ok = d_in && d_out;
    for (size_t h = 0; ok && h < *nlev; h++) {
      lvn[h] = level_dims[3 * h] * level_dims[3 * h + 1] * level_dims[3 * h + 2];
      d_lv[h] = static_cast<double*>(dalloc(lvn[h] * 8));
      ok = d_lv[h] != nullptr;
    }
    if (!ok) {
      fprintf(stderr, "[sperr_hip] device allocation failed\n");
      release();
      return -1;
    }
    int rtn = -1;
    if (hipMemcpy(d_in, src, src_len, hipMemcpyHostToDevice) == hipSuccess)
      rtn = sperrhip_decompress_multires_dev(d_in, src_len, output_float, d_out, n * esz, *nlev,
                                             d_lv.data(), nullptr);
    if (rtn == 0) {
      void* buf = malloc(n * esz);
      if (buf && hipMemcpy(buf, d_out, n * esz, hipMemcpyDeviceToHost) == hipSuccess)
        *dst = buf;
      else {
        free(buf);
        rtn = -1;
      }
      for (size_t h = 0; rtn == 0 && h < *nlev; h++) {
        levels[h] = static_cast<double*>(malloc(lvn[h] * 8));
        if (!levels[h] || hipMemcpy(levels[h], d_lv[h], lvn[h] * 8, hipMemcpyDeviceToHost) != hipSuccess)
          rtn = -1;
      }
      if (rtn == 0) {
        *dimx = ci.vol[0];
        *dimy = ci.vol[1];
        *dimz = ci.vol[2];
      }
    }
    release();
    return rtn;
  });
}

}  // extern "C"
