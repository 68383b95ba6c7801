// sperr3d -- compress / decompress a 3D volume on the GPU.  Same command line, files, messages
// and output naming as the reference's utilities/sperr3d.cpp (options :100-204, checks :206-262,
// work :267-418); the work itself is libsperr_hip.so (include/sperr_hip.h).  --omp is the number of
// host threads that stage rows for the transfers: the chunk loop runs on the library's GPU farm.
#include "cli_common.hpp"
#include "sperr_hip.h"

namespace {

struct Freed {   // buffers the C ABI hands out are malloc'd
  void* p = nullptr;
  ~Freed() { free(p); }
};

// name + ".XxYxZ" for every coarsened resolution, coarsest first (utilities/sperr3d.cpp:15-28)
std::vector<std::string> lowres_names(const std::string& name, size_t nlev, const size_t* level_dims)
{
  std::vector<std::string> out;
  for (size_t l = 0; l < nlev; l++)
    out.push_back(name + "." + std::to_string(level_dims[3 * l]) + "x" + std::to_string(level_dims[3 * l + 1]) +
                  "x" + std::to_string(level_dims[3 * l + 2]));
  return out;
}

// decodes `stream`, writes what was asked for; `out` keeps the volume (doubles) for the statistics
int decode_and_write(const std::vector<uint8_t>& stream, bool multi_res, const std::string& f64,
                     const std::string& f32, const std::string& low64, const std::string& low32, Freed& out,
                     size_t dims[3])
{
  size_t nlev = 0, level_dims[3 * 16] = {};
  double* levels[16] = {};
  int rtn;
  if (multi_res)
    rtn = sperrhip_decomp_3d_multires(stream.data(), stream.size(), 0, &dims[0], &dims[1], &dims[2], &out.p,
                                      &nlev, level_dims, levels);
  else
    rtn = sperr_decomp_3d(stream.data(), stream.size(), 0, 0, &dims[0], &dims[1], &dims[2], &out.p);
  if (rtn != 0) {
    printf("Decompression failed!\n");
    return 1;
  }
  int bad = 0;
  for (const std::string* name : {&low64, &low32}) {
    if (name->empty())
      continue;
    const auto files = lowres_names(*name, nlev, level_dims);
    for (size_t l = 0; l < nlev && !bad; l++) {
      const size_t n = level_dims[3 * l] * level_dims[3 * l + 1] * level_dims[3 * l + 2];
      const bool as64 = name == &low64;
      if (!cli::write_volume(levels[l], n, as64 ? files[l] : "", as64 ? "" : files[l], "hierarchy"))
        bad = 1;
    }
  }
  for (size_t l = 0; l < nlev; l++)
    free(levels[l]);
  if (bad)
    return 1;
  const size_t n = dims[0] * dims[1] * dims[2];
  return cli::write_volume(static_cast<const double*>(out.p), n, f64, f32, "data") ? 0 : 1;
}

}  // namespace

int main(int argc, char** argv)
{
  // volumes that the chunk size does not divide decode several shape groups side by side, one
  // stream each: let the runtime map them to more than its default 4 hardware queues (has to be
  // set before the first HIP call; an existing setting wins)
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  std::string input_file, bitstream, decomp_f32, decomp_f64, low_f32, low_f64;
  bool cflag = false, dflag = false, print_stats = false;
  size_t omp = 0, ftype = 0, dims[3] = {0, 0, 0}, chunks[3] = {256, 256, 256};
  double pwe = 0.0, psnr = 0.0, bpp = 0.0;

  cli::Parser app("3D SPERR compression and decompression (MI355X)\n");
  app.text("", input_file, "A data volume to be compressed, or\na bitstream to be decompressed.", "");
  const char *gx = "Execution settings", *gi = "Input properties (for compression)", *go = "Output settings",
             *gc = "Compression settings";
  app.flag("-c", cflag, "Perform a compression task.", gx);
  app.flag("-d", dflag, "Perform a decompression task.", gx).excludes = {"-c"};
  app.count("--omp", omp, "Number of host threads that stage rows for the GPU(s). Default (or 0): 4 per worker.\nThe GPUs are chosen by SPERR_HIP_DEVICES (default: all).", gx);
  app.count("--ftype", ftype, "Specify the input float type in bits. Must be 32 or 64.", gi);
  app.counts("--dims", dims, 3, "Dimensions of the input volume. E.g., `--dims 128 128 128`\n"
             "(The fastest-varying dimension appears first.)", gi);
  app.text("--bitstream", bitstream, "Output compressed bitstream.", go).needs = {"-c"};
  app.text("--decomp_f", decomp_f32, "Output decompressed volume in f32 precision.", go);
  app.text("--decomp_d", decomp_f64, "Output decompressed volume in f64 precision.", go);
  app.text("--decomp_lowres_f", low_f32, "Output lower resolutions of the decompressed volume in f32 precision.", go);
  app.text("--decomp_lowres_d", low_f64, "Output lower resolutions of the decompressed volume in f64 precision.", go);
  app.flag("--print_stats", print_stats, "Print statistics measuring the compression quality.", go).needs = {"-c"};
  app.counts("--chunks", chunks, 3, "Dimensions of the preferred chunk size. Default: 256 256 256\n"
             "(Volume dims don't need to be divisible by these chunk dims.)", gc);
  app.real("--pwe", pwe, "Maximum point-wise error (PWE) tolerance.", gc);
  app.real("--psnr", psnr, "Target PSNR to achieve.", gc).excludes = {"--pwe"};
  app.real("--bpp", bpp, "Target bit-per-pixel (bpp) to achieve.", gc, 0.0, 64.0).excludes = {"--pwe", "--psnr"};
  bool done = false;
  if (int rc = app.parse(argc, argv, done))
    return rc;
  if (done)
    return 0;

  // the reference's sanity checks, same order and wording (utilities/sperr3d.cpp:206-262)
  if (input_file.empty()) {
    printf("What's the input file?\n");
    return 1;
  }
  if (!cflag && !dflag) {
    printf("Is this compressing (-c) or decompressing (-d) ?\n");
    return 1;
  }
  if (cflag && dims[0] == 0 && dims[1] == 0 && dims[2] == 0) {
    printf("What's the dimensions of this 3D volume (--dims) ?\n");
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
  if (cflag && multi_res) {
    size_t nlev = 0, level_dims[3 * 16];
    if (sperrhip_multires_levels(dims[0], dims[1], dims[2], chunks[0], chunks[1], chunks[2], &nlev, level_dims) ||
        nlev == 0) {
      printf(" Warning: the combo of volume dimension (%zu, %zu, %zu) and chunk dimension"
             " (%zu, %zu, %zu)\n cannot support multi-resolution decoding. "
             " Try to use chunk dimensions that\n are similar in length and"
             " can divide the volume dimension.\n",
             dims[0], dims[1], dims[2], chunks[0], chunks[1], chunks[2]);
      return 1;
    }
  }
  if (cflag && bitstream.empty())
    printf("Warning: no output file provided. Consider using --bitstream option.\n");

  std::vector<uint8_t> input;
  if (!cli::read_file(input_file, input)) {
    printf("Cannot read %s\n", input_file.c_str());
    return 1;
  }

  if (cflag) {
    const size_t total = dims[0] * dims[1] * dims[2];
    if (total * (ftype / 8) != input.size()) {
      printf("Input file size wrong!\n");
      return 1;
    }
    const int mode = pwe != 0.0 ? 3 : psnr != 0.0 ? 2 : 1;
    const double quality = pwe != 0.0 ? pwe : psnr != 0.0 ? psnr : bpp;
    Freed enc;
    size_t enc_len = 0;
    if (sperr_comp_3d(input.data(), ftype == 32, dims[0], dims[1], dims[2], chunks[0], chunks[1], chunks[2], mode,
                      quality, omp, &enc.p, &enc_len) != 0) {
      printf("Compression failed!\n");
      return 1;
    }
    if (!bitstream.empty() && !cli::write_file(bitstream, enc.p, enc_len)) {
      printf("Writing compressed bitstream failed: %s\n", bitstream.c_str());
      return 1;
    }
    if (print_stats || !decomp_f64.empty() || !decomp_f32.empty() || multi_res) {
      const uint8_t* e = static_cast<const uint8_t*>(enc.p);
      const std::vector<uint8_t> stream(e, e + enc_len);
      Freed vol;
      size_t od[3];
      if (decode_and_write(stream, multi_res, decomp_f64, decomp_f32, low_f64, low_f32, vol, od))
        return 1;
      if (print_stats) {
        const double rate = (double)enc_len * 8.0 / (double)total;
        const double* recon = static_cast<const double*>(vol.p);
        const cli::Stats s = ftype == 32 ? cli::quality(reinterpret_cast<const float*>(input.data()), recon, total)
                                         : cli::quality(reinterpret_cast<const double*>(input.data()), recon, total);
        printf("Input range = (%.2e, %.2e), L-Infty = %.2e\n", s.lo, s.hi, s.linfty);
        printf("Bitrate = %.2f, PSNR = %.2fdB, Accuracy Gain = %.2f\n", rate, s.psnr,
               std::log2(s.sigma / s.rmse) - rate);
      }
    }
  }
  else {
    Freed vol;
    size_t od[3];
    if (decode_and_write(input, multi_res, decomp_f64, decomp_f32, low_f64, low_f32, vol, od))
      return 1;
  }
  return 0;
}
