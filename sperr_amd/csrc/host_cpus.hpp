// host_cpus.hpp -- how many host CPUs this process may actually use.
//
// The reference sizes its chunk loop by what the caller passes, 0 meaning omp_get_max_threads()
// (src/SPERR3D_OMP_C.cpp:12-20).  Here the chunk farm starts worker and helper threads of its own
// (farm.hip), and std::thread::hardware_concurrency() is the wrong yardstick for them inside a
// container: the pool's GPU box shows 256 logical CPUs and grants the pod 16 (cgroup v2 cpu.max =
// "1600000 100000", profiles/r4_box_probe.txt).  Threads beyond the quota do not run in parallel,
// they get throttled by CFS.  The budget is therefore
//
//     min( CPUs in the affinity mask , ceil(quota / period) over the cgroup and its ancestors )
//
// cgroup v2:  <root>/cpu.max             "max 100000" | "1600000 100000"
// cgroup v1:  <root>/cpu/cpu.cfs_quota_us, cpu.cfs_period_us   (-1: no limit)
// and, below the mount point, the directories of /proc/self/cgroup's path (a nested group inherits
// the tightest limit above it).  Host-only code, no HIP types; SPERR_HIP_CGROUP_ROOT points it at a
// made-up tree and SPERR_HIP_PROC_CGROUP at a made-up /proc/self/cgroup (tests/test_host_cpus.py).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>

namespace sperrhip {
namespace hostcpu {

// a cgroup-v1 controller list ("cpu,cpuacct", "cpuset,cpu,cpuacct", "cpuset") names the CPU bandwidth controller when one
// of its comma-separated entries is exactly "cpu" or "cpuacct" (round 5 looked for the substring and skipped every list
// that STARTED with cpuset, which dropped "cpuset,cpu,cpuacct")
inline bool names_cpu_controller(const std::string& ctrl)
{
  size_t a = 0;
  while (a <= ctrl.size()) {
    size_t e = ctrl.find(',', a);
    if (e == std::string::npos)
      e = ctrl.size();
    const std::string one = ctrl.substr(a, e - a);
    if (one == "cpu" || one == "cpuacct")
      return true;
    a = e + 1;
  }
  return false;
}


struct Budget {
  size_t visible = 1;    // logical CPUs the machine shows (hardware_concurrency)
  size_t affinity = 1;   // CPUs in the calling thread's affinity mask
  double quota = 0.0;    // CPUs' worth of CFS quota (0: no limit found)
  size_t usable = 1;     // what threads should be sized by
};

inline bool slurp(const std::string& path, std::string& out)
{
  FILE* f = fopen(path.c_str(), "r");
  if (!f)
    return false;
  char buf[8192];
  const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[n] = 0;
  out.assign(buf);
  return true;
}

// "1600000 100000" -> 16.0; "max 100000" -> 0 (no limit); malformed -> 0
inline double parse_cpu_max(const std::string& s)
{
  if (s.compare(0, 3, "max") == 0)
    return 0.0;
  long long q = 0, per = 0;
  if (sscanf(s.c_str(), "%lld %lld", &q, &per) != 2 || q <= 0 || per <= 0)
    return 0.0;
  return (double)q / (double)per;
}

inline double quota_of_dir(const std::string& dir)
{
  std::string a, b;
  if (slurp(dir + "/cpu.max", a))
    return parse_cpu_max(a);
  if (slurp(dir + "/cpu.cfs_quota_us", a) && slurp(dir + "/cpu.cfs_period_us", b)) {
    const long long q = atoll(a.c_str()), per = atoll(b.c_str());
    if (q > 0 && per > 0)
      return (double)q / (double)per;
  }
  return 0.0;
}

// the tightest quota over `root` (v2) or `root`/cpu (v1) and every directory of the process's own
// cgroup path below it
inline double cgroup_quota(const std::string& root, const std::string& procCgroup)
{
  double best = 0.0;
  auto take = [&](double q) {
    if (q > 0.0 && (best == 0.0 || q < best))
      best = q;
  };
  std::vector<std::string> bases = {root, root + "/cpu", root + "/cpu,cpuacct"};
  std::string text;
  std::vector<std::string> rels;   // the process's group, per hierarchy line that concerns the cpu controller
  if (slurp(procCgroup, text)) {
    size_t i = 0;
    while (i < text.size()) {
      size_t e = text.find('\n', i);
      if (e == std::string::npos)
        e = text.size();
      const std::string line = text.substr(i, e - i);
      i = e + 1;
      const size_t c1 = line.find(':');
      const size_t c2 = c1 == std::string::npos ? c1 : line.find(':', c1 + 1);
      if (c2 == std::string::npos)
        continue;
      const std::string ctrl = line.substr(c1 + 1, c2 - c1 - 1);
      if (!(ctrl.empty() || names_cpu_controller(ctrl)))
        continue;
      std::string rel = line.substr(c2 + 1);
      if (rel.find("..") != std::string::npos)
        continue;
      rels.push_back(rel);
    }
  }
  for (const auto& base : bases) {
    take(quota_of_dir(base));
    for (const auto& rel : rels) {
      std::string dir = base;
      size_t i = 0;
      while (i < rel.size()) {
        while (i < rel.size() && rel[i] == '/')
          i++;
        size_t e = rel.find('/', i);
        if (e == std::string::npos)
          e = rel.size();
        if (e > i) {
          dir += "/" + rel.substr(i, e - i);
          take(quota_of_dir(dir));
        }
        i = e;
      }
    }
  }
  return best;
}

inline size_t affinity_count()
{
  for (size_t ncpu = 1024; ncpu <= (1u << 16); ncpu *= 4) {
    cpu_set_t* set = CPU_ALLOC(ncpu);
    if (!set)
      break;
    const size_t sz = CPU_ALLOC_SIZE(ncpu);
    CPU_ZERO_S(sz, set);
    const int rc = sched_getaffinity(0, sz, set);
    const size_t n = rc == 0 ? (size_t)CPU_COUNT_S(sz, set) : 0;
    CPU_FREE(set);
    if (rc == 0)
      return std::max<size_t>(1, n);
  }
  return 0;
}

inline Budget probe(const std::string& root, const std::string& procCgroup)
{
  Budget b;
  b.visible = std::max(1u, std::thread::hardware_concurrency());
  const size_t aff = affinity_count();
  b.affinity = aff ? aff : b.visible;
  b.quota = cgroup_quota(root, procCgroup);
  size_t u = std::min(b.visible, b.affinity);
  if (b.quota > 0.0) {
    const size_t q = (size_t)(b.quota + 0.999);   // 1.5 CPUs of quota run two threads, half the time each
    u = std::min(u, std::max<size_t>(1, q));
  }
  b.usable = std::max<size_t>(1, u);
  return b;
}

inline Budget probe()
{
  const char* r = getenv("SPERR_HIP_CGROUP_ROOT");
  const char* p = getenv("SPERR_HIP_PROC_CGROUP");
  return probe((r && *r) ? r : "/sys/fs/cgroup", (p && *p) ? p : "/proc/self/cgroup");
}

// throttling so far of the process's group: nr_throttled and throttled_usec (v2) / throttled_time in
// ns (v1) of the first cpu.stat found from the process's own group upwards; false when there is none
inline bool throttle_stat(const std::string& root, const std::string& procCgroup, unsigned long long& nr,
                          unsigned long long& usec)
{
  nr = usec = 0;
  std::vector<std::string> dirs;
  std::string text;
  if (slurp(procCgroup, text)) {
    size_t i = 0;
    while (i < text.size()) {
      size_t e = text.find('\n', i);
      if (e == std::string::npos)
        e = text.size();
      const std::string line = text.substr(i, e - i);
      i = e + 1;
      const size_t c1 = line.find(':');
      const size_t c2 = c1 == std::string::npos ? c1 : line.find(':', c1 + 1);
      if (c2 == std::string::npos)
        continue;
      const std::string ctrl = line.substr(c1 + 1, c2 - c1 - 1);
      const std::string rel = line.substr(c2 + 1);
      if (rel.find("..") != std::string::npos)
        continue;
      if (ctrl.empty())
        dirs.push_back(root + rel);
      else if (names_cpu_controller(ctrl)) {
        dirs.push_back(root + "/cpu" + rel);
        dirs.push_back(root + "/cpu,cpuacct" + rel);
      }
    }
  }
  dirs.push_back(root);
  dirs.push_back(root + "/cpu");
  dirs.push_back(root + "/cpu,cpuacct");
  for (const auto& d : dirs) {
    std::string s;
    if (!slurp(d + "/cpu.stat", s))
      continue;
    bool any = false;
    size_t i = 0;
    while (i < s.size()) {
      size_t e = s.find('\n', i);
      if (e == std::string::npos)
        e = s.size();
      char key[64];
      unsigned long long v = 0;
      if (sscanf(s.substr(i, e - i).c_str(), "%63s %llu", key, &v) == 2) {
        if (!strcmp(key, "nr_throttled")) {
          nr = v;
          any = true;
        }
        else if (!strcmp(key, "throttled_usec"))
          usec = v;
        else if (!strcmp(key, "throttled_time"))
          usec = v / 1000;
      }
      i = e + 1;
    }
    if (any)
      return true;
  }
  return false;
}

}  // namespace hostcpu
}  // namespace sperrhip
