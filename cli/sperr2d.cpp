// sperr2d -- compress / decompress a 2D slice on the GPU.  Same command line, files and messages
// as the reference's utilities/sperr2d.cpp (options :95-200, checks :202-233, work :236-418): the
// bitstream file starts with the 10-byte header {version, flags, u32 dimx, u32 dimy}.
// Not available: --decomp_lowres_f / --decomp_lowres_d (the 2D resolution hierarchy).
#include "cli_common.hpp"
#include "sperr_hip.h"

namespace {
struct Freed {
  void* p = nullptr;
  ~Freed() { free(p); }
};
constexpr size_t kHeaderLen = 10;
}  // namespace

int main(int argc, char** argv)
{
  std::string input_file, bitstream, decomp_f32, decomp_f64, low_f32, low_f64;
  bool cflag = false, dflag = false, print_stats = false;
  size_t ftype = 0, dims[2] = {0, 0};
  double pwe = 0.0, psnr = 0.0, bpp = 0.0;

  cli::Parser app("2D SPERR compression and decompression (MI355X)\n");
  app.text("", input_file, "A data slice to be compressed, or\na bitstream to be decompressed.", "");
  const char *gx = "Execution settings", *gi = "Input properties", *go = "Output settings",
             *gc = "Compression settings";
  app.flag("-c", cflag, "Perform a compression task.", gx);
  app.flag("-d", dflag, "Perform a decompression task.", gx).excludes = {"-c"};
  app.count("--ftype", ftype, "Specify the input float type in bits. Must be 32 or 64.", gi);
  app.counts("--dims", dims, 2, "Dimensions of the input slice. E.g., `--dims 128 128`\n"
             "(The fastest-varying dimension appears first.)", gi);
  app.text("--bitstream", bitstream, "Output compressed bitstream.", go).needs = {"-c"};
  app.text("--decomp_f", decomp_f32, "Output decompressed slice in f32 precision.", go);
  app.text("--decomp_d", decomp_f64, "Output decompressed slice in f64 precision.", go);
  app.text("--decomp_lowres_f", low_f32, "(not available in this build)", go);
  app.text("--decomp_lowres_d", low_f64, "(not available in this build)", go);
  app.flag("--print_stats", print_stats, "Show statistics measuring the compression quality.", go).needs = {"-c"};
  app.real("--pwe", pwe, "Maximum point-wise error (PWE) tolerance.", gc);
  app.real("--psnr", psnr, "Target PSNR to achieve.", gc).excludes = {"--pwe"};
  app.real("--bpp", bpp, "Target bit-per-pixel (bpp) to achieve.", gc, 0.0, 64.0).excludes = {"--pwe", "--psnr"};
  bool done = false;
  if (int rc = app.parse(argc, argv, done))
    return rc;
  if (done)
    return 0;

  if (input_file.empty()) {
    printf("What's the input file?\n");
    return 1;
  }
  if (!cflag && !dflag) {
    printf("Is this compressing (-c) or decompressing (-d) ?\n");
    return 1;
  }
  if (cflag && dims[0] == 0 && dims[1] == 0) {
    printf("What's the dimensions of this 2D slice (--dims) ?\n");
    return 1;
  }
  if (cflag && ftype != 32 && ftype != 64) {
    printf("What's the floating-type precision (--ftype) ?\n");
    return 1;
  }
  if (cflag && pwe == 0.0 && psnr == 0.0 && bpp == 0.0) {
    printf("What's the compression quality (--psnr, --pwe, --bpp) ?\n");
    return 1;
  }
  if (cflag && (pwe < 0.0 || psnr < 0.0)) {
    printf("Compression quality (--psnr, --pwe) must be positive!\n");
    return 1;
  }
  if (!low_f32.empty() || !low_f64.empty()) {
    printf("The 2D resolution hierarchy (--decomp_lowres_f, --decomp_lowres_d) is not available in this build.\n");
    return 1;
  }
  if (dflag && decomp_f32.empty() && decomp_f64.empty()) {
    printf("SPERR needs an output destination when decoding!\n");
    return 1;
  }
  if (cflag && bitstream.empty())
    printf("Warning: no output file provided. Consider using --bitstream option.\n");

  std::vector<uint8_t> input;
  if (!cli::read_file(input_file, input)) {
    printf("Cannot read %s\n", input_file.c_str());
    return 1;
  }

  if (cflag) {
    const size_t total = dims[0] * dims[1];
    if (total * (ftype / 8) != input.size()) {
      printf("Input file size wrong!\n");
      return 1;
    }
    const int mode = pwe != 0.0 ? 3 : psnr != 0.0 ? 2 : 1;
    const double quality = pwe != 0.0 ? pwe : psnr != 0.0 ? psnr : bpp;
    Freed enc;
    size_t enc_len = 0;
    if (sperr_comp_2d(input.data(), ftype == 32, dims[0], dims[1], mode, quality, 1, &enc.p, &enc_len) != 0) {
      printf("Compression failed!\n");
      return 1;
    }
    if (!bitstream.empty() && !cli::write_file(bitstream, enc.p, enc_len)) {
      printf("Writing compressed bitstream failed: %s\n", bitstream.c_str());
      return 1;
    }
    if (print_stats || !decomp_f64.empty() || !decomp_f32.empty()) {
      Freed vol;
      if (sperr_decomp_2d(static_cast<const uint8_t*>(enc.p) + kHeaderLen, enc_len - kHeaderLen, 0, dims[0], dims[1],
                          &vol.p) != 0) {
        printf("Decompression failed!\n");
        return 1;
      }
      const double* recon = static_cast<const double*>(vol.p);
      if (!cli::write_volume(recon, total, decomp_f64, decomp_f32, "data"))
        return 1;
      if (print_stats) {
        const double rate = (double)enc_len * 8.0 / (double)total;
        const cli::Stats s = ftype == 32 ? cli::quality(reinterpret_cast<const float*>(input.data()), recon, total)
                                         : cli::quality(reinterpret_cast<const double*>(input.data()), recon, total);
        printf("Input range = (%.2e, %.2e), L-Infty = %.2e\n", s.lo, s.hi, s.linfty);
        printf("Bitrate = %.2f, PSNR = %.2fdB, Accuracy Gain = %.2f\n", rate, s.psnr,
               std::log2(s.sigma / s.rmse) - rate);
      }
    }
  }
  else {
    if (input.size() < kHeaderLen) {
      printf("Decompression failed!\n");
      return 1;
    }
    if (input[0] != 0) {   // SPERR_VERSION_MAJOR
      printf("This bitstream is produced by a compressor of a different version!\n");
      return 1;
    }
    if (input[1] & 0x40) {
      printf("This bitstream appears to represent a 3D volume!\n");
      return 1;
    }
    uint32_t d2[2];
    memcpy(d2, input.data() + 2, 8);
    Freed vol;
    if (sperr_decomp_2d(input.data() + kHeaderLen, input.size() - kHeaderLen, 0, d2[0], d2[1], &vol.p) != 0) {
      printf("Decompression failed!\n");
      return 1;
    }
    if (!cli::write_volume(static_cast<const double*>(vol.p), (size_t)d2[0] * d2[1], decomp_f64, decomp_f32, "data"))
      return 1;
  }
  return 0;
}
