// cli_common.hpp -- what the three command line tools share: a small option parser with the
// option surface of the reference's utilities (utilities/sperr3d.cpp:100-204,
// utilities/sperr2d.cpp:95-200, utilities/sperr3d_trunc.cpp:14-60: same names, arities, exclusions
// and messages), whole-file IO, and the quality figures of --print_stats
// (src/sperr_helper.cpp:430-517,595-640).  Everything that compresses goes through the C ABI of
// libsperr_hip.so (include/sperr_hip.h).
#ifndef SPERR_HIP_CLI_COMMON_HPP
#define SPERR_HIP_CLI_COMMON_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <vector>

namespace cli {

// ---- options -------------------------------------------------------------------------------
struct Option {
  std::string name;     // "-c", "--dims", ... ; "" = the positional file name
  int nvals;            // values it takes (0 = flag)
  std::string help, group;
  std::function<bool(const std::vector<std::string>&)> store;
  std::vector<std::string> excludes, needs;
  bool required = false, seen = false;
};

inline bool to_size(const std::string& s, size_t& v)
{
  if (s.empty() || s[0] == '-')
    return false;
  char* end = nullptr;
  const unsigned long long u = strtoull(s.c_str(), &end, 10);
  v = (size_t)u;
  return end && *end == '\0';
}

inline bool to_double(const std::string& s, double& v)
{
  if (s.empty())
    return false;
  char* end = nullptr;
  v = strtod(s.c_str(), &end);
  return end && *end == '\0';
}

class Parser {
 public:
  explicit Parser(std::string about) : about_(std::move(about)) {}

  Option& flag(const std::string& name, bool& dst, const std::string& help, const std::string& group)
  {
    return add(name, 0, help, group, [&dst](const std::vector<std::string>&) {
      dst = true;
      return true;
    });
  }
  Option& text(const std::string& name, std::string& dst, const std::string& help, const std::string& group)
  {
    return add(name, 1, help, group, [&dst](const std::vector<std::string>& v) {
      dst = v[0];
      return true;
    });
  }
  Option& real(const std::string& name, double& dst, const std::string& help, const std::string& group,
               double lo = -std::numeric_limits<double>::infinity(),
               double hi = std::numeric_limits<double>::infinity())
  {
    return add(name, 1, help, group, [&dst, lo, hi](const std::vector<std::string>& v) {
      return to_double(v[0], dst) && dst >= lo && dst <= hi;
    });
  }
  Option& count(const std::string& name, size_t& dst, const std::string& help, const std::string& group)
  {
    return add(name, 1, help, group, [&dst](const std::vector<std::string>& v) { return to_size(v[0], dst); });
  }
  Option& counts(const std::string& name, size_t* dst, int n, const std::string& help, const std::string& group)
  {
    return add(name, n, help, group, [dst, n](const std::vector<std::string>& v) {
      for (int i = 0; i < n; i++)
        if (!to_size(v[i], dst[i]))
          return false;
      return true;
    });
  }

  // 0: go on; otherwise the exit code (help: 0 is reported through `done`)
  int parse(int argc, char** argv, bool& done)
  {
    done = false;
    prog_ = argc ? argv[0] : "sperr";
    for (int i = 1; i < argc; i++) {
      std::string a = argv[i];
      if (a == "-h" || a == "--help") {
        usage(stdout);
        done = true;
        return 0;
      }
      std::vector<std::string> vals;
      Option* o = nullptr;
      const size_t eq = a.find('=');
      if (a.size() > 2 && a[0] == '-' && a[1] == '-' && eq != std::string::npos) {   // --name=value
        vals.push_back(a.substr(eq + 1));
        a = a.substr(0, eq);
      }
      if (a.size() > 1 && a[0] == '-' && !(isdigit((unsigned char)a[1]) || a[1] == '.'))
        o = find(a);
      else {
        o = find("");
        vals.push_back(a);
      }
      if (!o)
        return fail("The following argument was not expected: " + a);
      if (o->seen && o->nvals <= 1)
        return fail("Argument " + label(*o) + " given more than once");
      while ((int)vals.size() < o->nvals && i + 1 < argc)
        vals.push_back(argv[++i]);
      if ((int)vals.size() != o->nvals || (o->nvals == 0 && !vals.empty()))
        return fail(label(*o) + ": " + std::to_string(o->nvals) + " required");
      if (!o->store(vals))
        return fail("Could not convert: " + label(*o) + " = " + (vals.empty() ? "" : vals[0]));
      o->seen = true;
    }
    for (const Option& o : opts_) {
      if (o.required && !o.seen)
        return fail(label(o) + " is required");
      if (!o.seen)
        continue;
      for (const auto& x : o.excludes)
        if (find(x) && find(x)->seen)
          return fail(label(o) + " excludes " + x);
      for (const auto& x : o.needs)
        if (find(x) && !find(x)->seen)
          return fail(label(o) + " requires " + x);
    }
    return 0;
  }

  void usage(FILE* f) const
  {
    fprintf(f, "%s\nUsage: %s [OPTIONS]", about_.c_str(), prog_.c_str());
    for (const Option& o : opts_)
      if (o.name.empty())
        fprintf(f, " [filename]");
    fprintf(f, "\n\nOptions:\n  -h,--help                   Print this help message and exit\n");
    std::vector<std::string> groups;
    for (const Option& o : opts_)
      if (std::find(groups.begin(), groups.end(), o.group) == groups.end())
        groups.push_back(o.group);
    for (const auto& g : groups) {
      if (!g.empty())
        fprintf(f, "\n%s:\n", g.c_str());
      for (const Option& o : opts_) {
        if (o.group != g)
          continue;
        std::string left = "  " + label(o);
        if (o.nvals == 1 && !o.name.empty())
          left += " VALUE";
        else if (o.nvals > 1)
          left += " VALUE x " + std::to_string(o.nvals);
        if (o.required)
          left += " REQUIRED";
        std::string help = o.help;   // continuation lines line up with the first
        for (size_t at = help.find('\n'); at != std::string::npos; at = help.find('\n', at + 31))
          help.insert(at + 1, std::string(30, ' '));
        fprintf(f, "%-30s%s\n", left.c_str(), help.c_str());
      }
    }
  }

 private:
  Option& add(const std::string& name, int nvals, const std::string& help, const std::string& group,
              std::function<bool(const std::vector<std::string>&)> store)
  {
    opts_.push_back(Option{name, nvals, help, group, std::move(store), {}, {}});
    return opts_.back();
  }
  Option* find(const std::string& name)
  {
    for (Option& o : opts_)
      if (o.name == name)
        return &o;
    return nullptr;
  }
  static std::string label(const Option& o) { return o.name.empty() ? "filename" : o.name; }
  int fail(const std::string& msg) const
  {
    fprintf(stderr, "%s\nRun with --help for more information.\n", msg.c_str());
    return 106;   // the exit code of a parse error in the reference's tools
  }

  std::string about_, prog_;
  std::vector<Option> opts_;
};

// ---- files ---------------------------------------------------------------------------------
inline bool read_file(const std::string& name, std::vector<uint8_t>& buf)
{
  FILE* f = fopen(name.c_str(), "rb");
  if (!f)
    return false;
  bool ok = fseek(f, 0, SEEK_END) == 0;
  const long len = ok ? ftell(f) : -1;
  ok = ok && len >= 0 && fseek(f, 0, SEEK_SET) == 0;
  if (ok) {
    buf.resize((size_t)len);
    ok = fread(buf.data(), 1, buf.size(), f) == buf.size();
  }
  fclose(f);
  return ok;
}

inline bool write_file(const std::string& name, const void* p, size_t n)
{
  FILE* f = fopen(name.c_str(), "wb");
  if (!f)
    return false;
  const bool ok = fwrite(p, 1, n, f) == n;
  return (fclose(f) == 0) && ok;
}

// the decoded volume comes back as doubles; --decomp_f stores them narrowed, as the reference does
inline bool write_volume(const double* v, size_t n, const std::string& name_f64, const std::string& name_f32,
                         const char* what)
{
  if (!name_f64.empty() && !write_file(name_f64, v, n * sizeof(double))) {
    printf("Writing decompressed %s failed: %s\n", what, name_f64.c_str());
    return false;
  }
  if (!name_f32.empty()) {
    std::vector<float> f(v, v + n);
    if (!write_file(name_f32, f.data(), n * sizeof(float))) {
      printf("Writing decompressed %s failed: %s\n", what, name_f32.c_str());
      return false;
    }
  }
  return true;
}

// ---- quality figures -----------------------------------------------------------------------
struct Stats {
  double rmse = 0, linfty = 0, psnr = 0, lo = 0, hi = 0, sigma = 0;
};

// T = the precision of the original data: the differences are taken in that precision, as
// calc_stats<T> does (a float original is compared with the float-narrowed reconstruction)
template <typename T>
Stats quality(const T* orig, const double* recon, size_t n)
{
  Stats s;
  if (n == 0)
    return s;
  double sum = 0, sq = 0, lo = orig[0], hi = orig[0], linf = 0;
  bool same = true;
  for (size_t i = 0; i < n; i++) {
    const T r = (T)recon[i];
    const T d = std::abs(orig[i] - r);
    same = same && orig[i] == r;
    linf = std::max(linf, (double)d);
    sq += (double)d * (double)d;
    sum += orig[i];
    lo = std::min(lo, (double)orig[i]);
    hi = std::max(hi, (double)orig[i]);
  }
  const double mean = sum / (double)n;
  double var = 0;
  for (size_t i = 0; i < n; i++)
    var += ((double)orig[i] - mean) * ((double)orig[i] - mean);
  s.lo = lo;
  s.hi = hi;
  s.sigma = std::sqrt(var / (double)n);
  if (same) {
    s.psnr = std::numeric_limits<double>::infinity();
    return s;
  }
  const double mse = sq / (double)n;
  s.rmse = std::sqrt(mse);
  s.linfty = linf;
  s.psnr = 10.0 * std::log10((hi - lo) * (hi - lo) / mse);
  return s;
}

}  // namespace cli
#endif
