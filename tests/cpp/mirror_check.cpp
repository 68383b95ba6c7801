// Drives the C++ mirrors of include/sperr_hip.hpp the way code written against the reference's
// classes does (test_scripts/sperr3d_unit_test.cpp, speck2d_flt_unit_test.cpp,
// stream_tools_unit_test.cpp) and dumps what they produce; tests/test_cpp_mirrors.py compares the
// files with the oracle's.   usage: mirror_check <vol.f32> <dx> <dy> <dz> <cx> <cy> <cz> <outdir>
#include <cstdio>
#include <string>

#include "sperr_hip.hpp"

static bool dump(const std::string& name, const void* p, size_t n)
{
  std::FILE* f = std::fopen(name.c_str(), "wb");
  if (!f)
    return false;
  const bool ok = std::fwrite(p, 1, n, f) == n;
  return std::fclose(f) == 0 && ok;
}

#define CHECK(cond)                                              \
  if (!(cond)) {                                                 \
    std::fprintf(stderr, "line %d: %s\n", __LINE__, #cond);      \
    return 1;                                                    \
  }

int main(int argc, char** argv)
{
  if (argc != 9)
    return 2;
  const sperr::dims_type dims{std::stoul(argv[2]), std::stoul(argv[3]), std::stoul(argv[4])};
  const sperr::dims_type chunks{std::stoul(argv[5]), std::stoul(argv[6]), std::stoul(argv[7])};
  const std::string out = std::string(argv[8]) + "/";
  const size_t n = dims[0] * dims[1] * dims[2];
  std::vector<float> vol(n);
  std::FILE* f = std::fopen(argv[1], "rb");
  CHECK(f && std::fread(vol.data(), 4, n, f) == n);
  std::fclose(f);

  // SPERR3D_OMP_C in the three modes
  sperr::vec8_type rate_stream;
  for (int mode = 1; mode <= 3; mode++) {
    sperr::SPERR3D_OMP_C enc;
    enc.set_dims_and_chunks(dims, chunks);
    enc.set_num_threads(3);
    if (mode == 1)
      enc.set_bitrate(2.5);
    else if (mode == 2)
      enc.set_psnr(80.0);
    else
      enc.set_tolerance(1e-2);
    CHECK(enc.compress(vol.data(), n) == sperr::RTNType::Good);
    const auto stream = enc.get_encoded_bitstream();
    CHECK(dump(out + "omp_c_mode" + std::to_string(mode), stream.data(), stream.size()));
    if (mode == 1)
      rate_stream = stream;
  }

  // SPERR3D_OMP_D, with the hierarchy
  {
    sperr::SPERR3D_OMP_D dec;
    CHECK(dec.use_bitstream(rate_stream.data(), rate_stream.size()) == sperr::RTNType::Good);
    CHECK(dec.decompress(rate_stream.data(), true) == sperr::RTNType::Good);
    CHECK(dec.get_dims() == dims && dec.get_chunk_dims() == chunks);
    const auto& v = dec.view_decoded_data();
    CHECK(v.size() == n && dump(out + "omp_d_vol", v.data(), v.size() * 8));
    const auto h = dec.release_hierarchy();
    for (size_t l = 0; l < h.size(); l++)
      CHECK(dump(out + "omp_d_level" + std::to_string(l), h[l].data(), h[l].size() * 8));
  }

  // Stream tools
  {
    sperr::SPERR3D_Stream_Tools tools;
    const auto hd = tools.get_stream_header(rate_stream.data());
    CHECK(hd.is_3D && hd.is_float && !hd.is_portion && hd.vol_dims == dims && hd.chunk_dims == chunks);
    CHECK(hd.stream_len == rate_stream.size());
    CHECK(hd.chunk_offsets.size() >= 2 && hd.chunk_offsets[0] == hd.header_len);
    std::array<uint8_t, 20> magic;
    std::copy(rate_stream.begin(), rate_stream.begin() + 20, magic.begin());
    CHECK(tools.get_header_len(magic) == hd.header_len);
    const auto cut = tools.progressive_truncate(rate_stream.data(), rate_stream.size(), 30);
    CHECK(!cut.empty() && dump(out + "trunc30", cut.data(), cut.size()));
    CHECK(tools.get_stream_header(cut.data()).is_portion);
    CHECK(dump(out + "whole.sperr", rate_stream.data(), rate_stream.size()));
    CHECK(tools.progressive_read(out + "whole.sperr", 30) == cut);
  }

  // SPECK3D_FLT: the first chunk-sized corner as one chunk, PWE mode, with the hierarchy
  {
    const sperr::dims_type cd = chunks;
    std::vector<double> corner(cd[0] * cd[1] * cd[2]);
    for (size_t z = 0; z < cd[2]; z++)
      for (size_t y = 0; y < cd[1]; y++)
        for (size_t x = 0; x < cd[0]; x++)
          corner[(z * cd[1] + y) * cd[0] + x] = vol[(z * dims[1] + y) * dims[0] + x];
    sperr::SPECK3D_FLT enc;
    enc.set_dims(cd);
    enc.copy_data(corner.data(), corner.size());
    enc.set_tolerance(1e-3);
    CHECK(enc.compress() == sperr::RTNType::Good);
    sperr::vec8_type stream;
    enc.append_encoded_bitstream(stream);
    CHECK(dump(out + "speck3d_flt_stream", stream.data(), stream.size()));
    sperr::SPECK3D_FLT dec;
    dec.set_dims(cd);
    CHECK(dec.use_bitstream(stream.data(), stream.size()) == sperr::RTNType::Good);
    CHECK(dec.decompress(true) == sperr::RTNType::Good);
    CHECK(dump(out + "speck3d_flt_vol", dec.view_decoded_data().data(), corner.size() * 8));
    const auto& h = dec.view_hierarchy();
    for (size_t l = 0; l < h.size(); l++)
      CHECK(dump(out + "speck3d_flt_level" + std::to_string(l), h[l].data(), h[l].size() * 8));
  }

  // SPECK2D_FLT: the first z-slice
  {
    const sperr::dims_type sd{dims[0], dims[1], 1};
    sperr::SPECK2D_FLT enc;
    enc.set_dims(sd);
    enc.copy_data(vol.data(), sd[0] * sd[1]);
    enc.set_psnr(90.0);
    CHECK(enc.compress() == sperr::RTNType::Good);
    sperr::vec8_type stream;
    enc.append_encoded_bitstream(stream);
    CHECK(dump(out + "speck2d_flt_stream", stream.data(), stream.size()));
    sperr::SPECK2D_FLT dec;
    dec.set_dims(sd);
    CHECK(dec.use_bitstream(stream.data(), stream.size()) == sperr::RTNType::Good);
    CHECK(dec.decompress(true) == sperr::RTNType::Good);
    const auto h = dec.release_hierarchy();
    const auto v = dec.release_decoded_data();
    CHECK(v.size() == sd[0] * sd[1] && dump(out + "speck2d_flt_slice", v.data(), v.size() * 8));
    for (size_t l = 0; l < h.size(); l++)
      CHECK(dump(out + "speck2d_flt_level" + std::to_string(l), h[l].data(), h[l].size() * 8));
  }
  // integer_len() of both chunk classes (test_scripts/speck3d_flt_unit_test.cpp:63-147): the PSNR
  // targets walk through the four widths; tests/test_cpp_mirrors.py compares with the oracle's streams
  {
    std::string widths;
    const sperr::dims_type cd = chunks;
    std::vector<double> corner(cd[0] * cd[1] * cd[2]);
    for (size_t z = 0; z < cd[2]; z++)
      for (size_t y = 0; y < cd[1]; y++)
        for (size_t x = 0; x < cd[0]; x++)
          corner[(z * cd[1] + y) * cd[0] + x] = vol[(z * dims[1] + y) * dims[0] + x];
    for (double psnr : {20.0, 60.0, 120.0, 250.0}) {
      sperr::SPECK3D_FLT enc, dec;
      enc.set_dims(cd);
      enc.set_psnr(psnr);
      enc.copy_data(corner.data(), corner.size());
      CHECK(enc.compress() == sperr::RTNType::Good);
      sperr::vec8_type stream;
      enc.append_encoded_bitstream(stream);
      dec.set_dims(cd);
      CHECK(dec.use_bitstream(stream.data(), stream.size()) == sperr::RTNType::Good);
      CHECK(dec.decompress() == sperr::RTNType::Good);
      CHECK(enc.integer_len() == dec.integer_len());
      widths += std::to_string(enc.integer_len()) + " ";
      sperr::SPECK2D_FLT e2;
      e2.set_dims({dims[0], dims[1], 1});
      e2.set_psnr(psnr);
      e2.copy_data(vol.data(), dims[0] * dims[1]);
      CHECK(e2.compress() == sperr::RTNType::Good);
      widths += std::to_string(e2.integer_len()) + " ";
    }
    CHECK(dump(out + "integer_len", widths.data(), widths.size()));
  }
  std::printf("mirrors ok\n");
  return 0;
}
