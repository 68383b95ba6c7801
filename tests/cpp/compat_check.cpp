// compat_check.cpp -- a caller written against the REFERENCE's header names and spellings
// (the includes of /root/reference/utilities/sperr3d.cpp:1-2, sperr2d.cpp:1, sperr3d_trunc.cpp:1-2,
// examples/C_API/3d.c:1; the C++ spelling C_API::sperr_comp_3d of include/SPERR_C_API.h:16-19),
// compiled with -I include/compat and linked with -lSPERR.
//
//   compat_check helpers                       host helpers only (runs without a GPU): prints
//                                              geometry and statistics for the test to compare
//   compat_check run <f32 file> x y z cx cy cz <outdir>   (GPU) C API through namespace C_API and
//                                              the driver classes; writes what they produce
#include "SPERR3D_OMP_C.h"
#include "SPERR3D_OMP_D.h"
#include "SPERR3D_Stream_Tools.h"
#include "SPECK2D_FLT.h"
#include "SPECK3D_FLT.h"
#include "SPERR_C_API.h"

#include <cstdio>
#include <cstdlib>
#include <iostream>

static int helpers()
{
  std::printf("version %d.%d.%d\n", SPERR_VERSION_MAJOR, SPERR_VERSION_MINOR, SPERR_VERSION_PATCH);
  for (size_t len : {1, 8, 9, 17, 128, 250, 256, 999, 4096})
    std::printf("xforms %zu %zu parts %zu approx %zu %zu\n", len, sperr::num_of_xforms(len),
                sperr::num_of_partitions(len), sperr::calc_approx_detail_len(len, 3)[0],
                sperr::calc_approx_detail_len(len, 3)[1]);
  for (auto d : {sperr::dims_type{256, 256, 256}, {128, 128, 41}, {64, 64, 9}, {999, 999, 1}, {17, 17, 17}}) {
    auto dy = sperr::can_use_dyadic(d);
    std::printf("dyadic %zu %zu %zu %d\n", d[0], d[1], d[2], dy ? (int)*dy : -1);
    for (auto r : sperr::coarsened_resolutions(d))
      std::printf("res %zu %zu %zu\n", r[0], r[1], r[2]);
  }
  for (auto r : sperr::coarsened_resolutions({512, 256, 128}, {128, 128, 128}))
    std::printf("volres %zu %zu %zu\n", r[0], r[1], r[2]);
  std::printf("volres_indivisible %zu\n", sperr::coarsened_resolutions({500, 256, 128}, {128, 128, 128}).size());
  for (auto c : sperr::chunk_volume({1000, 300, 70}, {256, 256, 256}))
    std::printf("chunk %zu %zu %zu %zu %zu %zu\n", c[0], c[1], c[2], c[3], c[4], c[5]);
  const auto b = sperr::unpack_8_booleans(0xA4);
  std::printf("bools %d%d%d%d%d%d%d%d %u\n", b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7],
              (unsigned)sperr::pack_8_booleans({true, false, false, true, false, false, true, true}));
  sperr::vecf_type a(20000), c(20000);
  for (size_t i = 0; i < a.size(); i++) {
    a[i] = (float)((i * 2654435761u) % 1000) / 7.0f;
    c[i] = a[i] + (float)((i * 40503u) % 13) * 1e-3f;
  }
  const auto st = sperr::calc_stats(a.data(), c.data(), a.size());
  const auto mv = sperr::calc_mean_var(a.data(), a.size());
  std::printf("stats %.9g %.9g %.9g %.9g %.9g\n", st[0], st[1], st[2], st[3], st[4]);
  std::printf("meanvar %.9g %.9g\n", mv[0], mv[1]);
  std::printf("same_psnr_inf %d\n", std::isinf(sperr::calc_stats(a.data(), a.data(), a.size())[2]) ? 1 : 0);
  return 0;
}

int main(int argc, char** argv)
{
  if (argc >= 2 && std::string(argv[1]) == "helpers")
    return helpers();
  if (argc != 10 || std::string(argv[1]) != "run") {
    std::cerr << "usage: compat_check helpers | run file x y z cx cy cz outdir\n";
    return 2;
  }
  const sperr::dims_type dims = {std::stoul(argv[3]), std::stoul(argv[4]), std::stoul(argv[5])};
  const sperr::dims_type chunks = {std::stoul(argv[6]), std::stoul(argv[7]), std::stoul(argv[8])};
  const std::string out = std::string(argv[9]) + "/";
  const auto vol = sperr::read_whole_file<float>(argv[2]);
  if (vol.size() != dims[0] * dims[1] * dims[2])
    return 3;

  // the C API as a C++ caller of the reference spells it
  void* stream = nullptr;
  size_t stream_len = 0;
  int rtn = C_API::sperr_comp_3d(vol.data(), 1, dims[0], dims[1], dims[2], chunks[0], chunks[1], chunks[2], 3,
                                 1e-3, 0, &stream, &stream_len);
  if (rtn != 0)
    return 10 + rtn;
  if (C_API::sperr_comp_3d(vol.data(), 1, dims[0], dims[1], dims[2], chunks[0], chunks[1], chunks[2], 3, 1e-3, 0,
                           &stream, &stream_len) != 1)   // *dst not NULL
    return 4;
  if (sperr::write_n_bytes(out + "capi_pwe", stream_len, stream) != sperr::RTNType::Good)
    return 5;
  size_t dx = 0, dy = 0, dz = 0;
  int is_float = 0;
  C_API::sperr_parse_header(stream, &dx, &dy, &dz, &is_float);
  if (dx != dims[0] || dy != dims[1] || dz != dims[2] || is_float != 1)
    return 6;
  void* back = nullptr;
  if (C_API::sperr_decomp_3d(stream, stream_len, 1, 0, &dx, &dy, &dz, &back) != 0)
    return 7;
  sperr::write_n_bytes(out + "capi_pwe_f32", vol.size() * 4, back);
  const auto st = sperr::calc_stats(vol.data(), static_cast<const float*>(back), vol.size());
  std::printf("linfty %.9g\n", st[1]);
  std::free(back);
  void* cut = nullptr;
  size_t cut_len = 0;
  if (C_API::sperr_trunc_3d(stream, stream_len, 50, &cut, &cut_len) != 0)
    return 8;
  sperr::write_n_bytes(out + "capi_trunc50", cut_len, cut);
  std::free(cut);
  std::free(stream);

  // the driver classes, as utilities/sperr3d.cpp uses them
  sperr::SPERR3D_OMP_C enc;
  enc.set_dims_and_chunks(dims, chunks);
  enc.set_num_threads(0);
  enc.set_bitrate(2.0);
  if (enc.compress(vol.data(), vol.size()) != sperr::RTNType::Good)
    return 20;
  const auto bits = enc.get_encoded_bitstream();
  sperr::write_n_bytes(out + "omp_c_bpp2", bits.size(), bits.data());
  sperr::SPERR3D_OMP_D dec;
  dec.set_num_threads(0);
  if (dec.use_bitstream(bits.data(), bits.size()) != sperr::RTNType::Good ||
      dec.decompress(bits.data(), true) != sperr::RTNType::Good)
    return 21;
  const auto& v = dec.view_decoded_data();
  sperr::write_n_bytes(out + "omp_d_f64", v.size() * 8, v.data());
  std::printf("levels %zu\n", dec.view_hierarchy().size());
  const auto hdr = sperr::SPERR3D_Stream_Tools().get_stream_header(bits.data());
  std::printf("header %zu %zu %d\n", hdr.header_len, hdr.stream_len, (int)hdr.is_float);

  sperr::SPECK2D_FLT s2;
  s2.set_dims({dims[0], dims[1], 1});
  s2.copy_data(vol.data(), dims[0] * dims[1]);
  s2.set_psnr(90.0);
  if (s2.compress() != sperr::RTNType::Good)
    return 30;
  sperr::vec8_type s2bits;
  s2.append_encoded_bitstream(s2bits);
  sperr::write_n_bytes(out + "speck2d_psnr90", s2bits.size(), s2bits.data());
  return 0;
}
