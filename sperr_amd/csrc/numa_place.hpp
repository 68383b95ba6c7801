// numa_place.hpp -- where the chunk farm's host threads and staging memory go on a multi-socket node.
//
// The reference's chunk loop is one OpenMP team (src/SPERR3D_OMP_C.cpp:94-130, src/SPERR3D_OMP_D.cpp:101-127)
// whose placement is the OpenMP runtime's business (OMP_PLACES / OMP_PROC_BIND).  Here a device has
// its own worker threads (farm.hip), each with helper threads that move rows and with pinned staging
// buffers the device DMAs from: on a two-socket node with eight GPUs all of that belongs on the
// socket the GPU hangs off, or every row crosses the socket interconnect twice.
//
//   device ordinal --hipDeviceGetPCIBusId--> "0000:c1:00.0"
//                  --> <sysfs>/bus/pci/devices/0000:c1:00.0/numa_node          (-1: unknown)
//                  --> <sysfs>/devices/system/node/node<N>/cpulist             ("0-63,128-191")
//                  --> sched_setaffinity of the worker thread (helper threads inherit the mask; pinned
//                      staging memory is allocated -- first touched -- by the bound worker, so the
//                      kernel's default local policy puts it on that node)
//
// Host-only code, no HIP types: the parsing and the binding are tested on the CPU over a made-up
// sysfs tree (tests/test_farm_numa.py).  SPERR_HIP_FARM_NUMA=0 switches the placement off;
// SPERR_HIP_SYSFS_ROOT points it at another tree (tests).
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sched.h>

namespace sperrhip {
namespace numa {

inline std::string sysfs_root()
{
  const char* v = getenv("SPERR_HIP_SYSFS_ROOT");
  return (v && *v) ? std::string(v) : std::string("/sys");
}

inline bool enabled()
{
  const char* v = getenv("SPERR_HIP_FARM_NUMA");
  return !(v && *v && atoi(v) == 0);
}

inline bool read_line(const std::string& path, std::string& out)
{
  FILE* f = fopen(path.c_str(), "r");
  if (!f)
    return false;
  char buf[4096];
  const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[n] = 0;
  out.assign(buf);
  while (!out.empty() && isspace((unsigned char)out.back()))
    out.pop_back();
  return true;
}

// "0-3,8,10-11" -> {0,1,2,3,8,10,11}; anything malformed ends the list where it stands
inline std::vector<int> parse_cpulist(const std::string& s)
{
  std::vector<int> cpus;
  size_t i = 0;
  auto number = [&](long& v) {
    if (i >= s.size() || !isdigit((unsigned char)s[i]))
      return false;
    v = 0;
    while (i < s.size() && isdigit((unsigned char)s[i]) && v < (1 << 20))
      v = v * 10 + (s[i++] - '0');
    return true;
  };
  while (i < s.size()) {
    long a = 0, b = 0;
    if (!number(a))
      break;
    b = a;
    if (i < s.size() && s[i] == '-') {
      i++;
      if (!number(b))
        break;
    }
    for (long c = a; c <= b && c < (1 << 16); c++)
      cpus.push_back((int)c);
    if (i < s.size() && s[i] == ',')
      i++;
    else
      break;
  }
  return cpus;
}

struct Place {
  int node = -1;            // NUMA node of the device (-1: the platform does not say)
  std::vector<int> cpus;    // that node's CPUs
};

// `bdf` as hipDeviceGetPCIBusId prints it ("0000:C1:00.0"; sysfs spells it in lower case)
inline Place probe(const std::string& root, std::string bdf)
{
  Place p;
  for (auto& c : bdf)
    c = (char)tolower((unsigned char)c);
  std::string line;
  if (!read_line(root + "/bus/pci/devices/" + bdf + "/numa_node", line))
    return p;
  char* end = nullptr;
  const long node = strtol(line.c_str(), &end, 10);
  if (end == line.c_str() || node < 0)
    return p;
  p.node = (int)node;
  if (read_line(root + "/devices/system/node/node" + std::to_string(node) + "/cpulist", line))
    p.cpus = parse_cpulist(line);
  return p;
}

// Narrows the CALLING thread's affinity to the place's CPUs (those of them the thread may run on
// already: a cpuset of the container stays in force).  Returns how many CPUs it is bound to, 0 when
// nothing was changed (unknown node, no CPU in common, or the call failed).
inline size_t bind_self(const Place& p)
{
  if (p.node < 0 || p.cpus.empty())
    return 0;
  const int most = *std::max_element(p.cpus.begin(), p.cpus.end());
  const size_t ncpu = (size_t)std::max(most + 1, 1024);
  cpu_set_t* cur = CPU_ALLOC(ncpu);
  cpu_set_t* want = CPU_ALLOC(ncpu);
  if (!cur || !want) {
    if (cur)
      CPU_FREE(cur);
    if (want)
      CPU_FREE(want);
    return 0;
  }
  const size_t sz = CPU_ALLOC_SIZE(ncpu);
  CPU_ZERO_S(sz, cur);
  CPU_ZERO_S(sz, want);
  size_t n = 0;
  if (sched_getaffinity(0, sz, cur) == 0) {
    for (int c : p.cpus)
      if (CPU_ISSET_S((size_t)c, sz, cur)) {
        CPU_SET_S((size_t)c, sz, want);
        n++;
      }
    if (n && sched_setaffinity(0, sz, want) != 0)
      n = 0;
  }
  CPU_FREE(cur);
  CPU_FREE(want);
  return n;
}

}  // namespace numa
}  // namespace sperrhip
