// sperr3d_trunc -- keep a percentage of every chunk of a SPERR3D bitstream and, given the original
// data, report the quality of what is left.  Command line and output of the reference's
// utilities/sperr3d_trunc.cpp (:14-146) over sperr_trunc_3d / sperr_decomp_3d.
#include "cli_common.hpp"
#include "sperr_hip.h"

namespace {
struct Freed {
  void* p = nullptr;
  ~Freed() { free(p); }
};
}  // namespace

int main(int argc, char** argv)
{
  // volumes that the chunk size does not divide decode several shape groups side by side, one
  // stream each: let the runtime map them to more than its default 4 hardware queues (has to be
  // set before the first HIP call; an existing setting wins)
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  std::string input_file, out_file, orig32, orig64;
  size_t pct = 0, omp = 0;
  cli::Parser app("Truncate a SPERR3D bitstream to a percentage of its length (MI355X)\n");
  app.text("", input_file, "The original SPERR3D bitstream to be truncated.", "");
  app.count("--pct", pct, "Percentage (1--100) of the original bitstream to truncate.", "Truncation settings")
      .required = true;
  app.count("--omp", omp, "Accepted for compatibility; the GPU engine has no thread team.", "Truncation settings");
  app.text("-o", out_file, "Write out the truncated bitstream.", "Output settings");
  app.text("--orig32", orig32, "Original raw data in 32-bit precision to calculate compression\n"
           "quality using the truncated bitstream.", "Input settings");
  app.text("--orig64", orig64, "Original raw data in 64-bit precision to calculate compression\n"
           "quality using the truncated bitstream.", "Input settings");
  bool done = false;
  if (int rc = app.parse(argc, argv, done))
    return rc;
  if (done)
    return 0;
  if (!orig32.empty() && !orig64.empty()) {
    printf("Is the original data in 32 or 64 bit precision?\n");
    return 1;
  }
  std::vector<uint8_t> input;
  Freed cut;
  size_t cut_len = 0;
  if (input_file.empty() || !cli::read_file(input_file, input) ||
      sperr_trunc_3d(input.data(), input.size(), (unsigned)pct, &cut.p, &cut_len) != 0) {
    printf("Error while truncating bitstream %s\n", input_file.c_str());
    return 1;
  }
  size_t dims[3] = {0, 0, 0};
  int is_float = 0;
  sperr_parse_header(cut.p, &dims[0], &dims[1], &dims[2], &is_float);
  const size_t total = dims[0] * dims[1] * dims[2];
  const double rate = (double)cut_len * 8.0 / (double)total;
  printf("Truncation resulting BPP = %.2f\n", rate);
  if (!out_file.empty() && !cli::write_file(out_file, cut.p, cut_len))
    return 1;

  if (!orig32.empty() || !orig64.empty()) {
    Freed vol;
    size_t od[3];
    if (sperr_decomp_3d(cut.p, cut_len, 0, omp, &od[0], &od[1], &od[2], &vol.p) != 0) {
      printf("Decompression failed!\n");
      return 1;
    }
    const std::string& name = orig64.empty() ? orig32 : orig64;
    std::vector<uint8_t> orig;
    if (!cli::read_file(name, orig) || orig.size() != total * (orig64.empty() ? 4 : 8)) {
      printf("Read original data failed: %s\n", name.c_str());
      return 1;
    }
    const double* recon = static_cast<const double*>(vol.p);
    const cli::Stats s = orig64.empty() ? cli::quality(reinterpret_cast<const float*>(orig.data()), recon, total)
                                        : cli::quality(reinterpret_cast<const double*>(orig.data()), recon, total);
    printf("PSNR = %.2f, L-Infty = %.2e, Accuracy Gain = %.2f\n", s.psnr, s.linfty,
           std::log2(s.sigma / s.rmse) - rate);
  }
  return 0;
}
