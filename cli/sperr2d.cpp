// sperr2d -- compress / decompress a 2D slice on the GPU.  Same command line, files and messages
// as the reference's utilities/sperr2d.cpp (options :95-200, checks :202-233, work :236-418): the
// bitstream file starts with the 10-byte header {version, flags, u32 dimx, u32 dimy};
// --decomp_lowres_f / --decomp_lowres_d write one file per coarsened resolution, "name.XxY"
// (utilities/sperr2d.cpp:13-24).
#include "cli_common.hpp"
#include "sperr_hip.h"

namespace {
struct Freed {
  void* p = nullptr;
  ~Freed() { free(p); }
};
constexpr size_t kHeaderLen = 10;

// decodes the header-less `stream`, writes what was asked for; `out` keeps the slice (doubles)
int decode_and_write(const uint8_t* stream, size_t len, size_t dx, size_t dy, const std::string& f64,
                     const std::string& f32, const std::string& low64, const std::string& low32, Freed& out)
{
  const bool multi_res = !low64.empty() || !low32.empty();
  size_t nlev = 0, level_dims[2 * 16] = {};
  double* levels[16] = {};
  const int rtn = multi_res ? sperrhip_decomp_2d_multires(stream, len, 0, dx, dy, &out.p, &nlev, level_dims, levels)
                            : sperr_decomp_2d(stream, len, 0, dx, dy, &out.p);
  if (rtn != 0) {
    printf("Decompression failed!\n");
    return 1;
  }
  int bad = 0;
  for (const std::string* name : {&low64, &low32})
    for (size_t l = 0; l < nlev && !bad && !name->empty(); l++) {
      const std::string file = *name + "." + std::to_string(level_dims[2 * l]) + "x" + std::to_string(level_dims[2 * l + 1]);
      const bool as64 = name == &low64;
      if (!cli::write_volume(levels[l], level_dims[2 * l] * level_dims[2 * l + 1], as64 ? file : "", as64 ? "" : file,
                             "hierarchy"))
        bad = 1;
    }
  for (size_t l = 0; l < nlev; l++)
    free(levels[l]);
  if (bad)
    return 1;
  return cli::write_volume(static_cast<const double*>(out.p), dx * dy, f64, f32, "data") ? 0 : 1;
}
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
  app.text("--decomp_lowres_f", low_f32, "Output lower resolutions of the decompressed slice in f32 precision.", go);
  app.text("--decomp_lowres_d", low_f64, "Output lower resolutions of the decompressed slice in f64 precision.", go);
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
  const bool multi_res = !low_f32.empty() || !low_f64.empty();
  if (dflag && decomp_f32.empty() && decomp_f64.empty() && !multi_res) {
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
    if (print_stats || !decomp_f64.empty() || !decomp_f32.empty() || multi_res) {
      Freed vol;
      if (decode_and_write(static_cast<const uint8_t*>(enc.p) + kHeaderLen, enc_len - kHeaderLen, dims[0], dims[1],
                           decomp_f64, decomp_f32, low_f64, low_f32, vol))
        return 1;
      const double* recon = static_cast<const double*>(vol.p);
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
    if (decode_and_write(input.data() + kHeaderLen, input.size() - kHeaderLen, d2[0], d2[1], decomp_f64, decomp_f32,
                         low_f64, low_f32, vol))
      return 1;
  }
  return 0;
}
