"""The command line tools (cli/): sperr3d, sperr2d, sperr3d_trunc over the C ABI.

CPU: option handling -- names, arities, exclusions, the reference's sanity checks and messages
(utilities/sperr3d.cpp:206-262).  GPU (-m gpu): the files they write are the oracle's containers and
decoded values, byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from fields import smooth_field

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cli", "bin")


@pytest.fixture(scope="module")
def tools():
    from sperr_amd import api
    if not os.path.exists(api.LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sperr_amd", "csrc"), "-j4"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cli")])
    return BIN


def run(tools, name, *args):
    p = subprocess.run([os.path.join(tools, name)] + [str(a) for a in args], capture_output=True, text=True,
                       timeout=600)
    return p.returncode, p.stdout, p.stderr


def test_help_lists_the_reference_options(tools):
    rc, out, _ = run(tools, "sperr3d", "--help")
    assert rc == 0
    for opt in ("-c", "-d", "--omp", "--ftype", "--dims", "--bitstream", "--decomp_f", "--decomp_d",
                "--decomp_lowres_f", "--decomp_lowres_d", "--print_stats", "--chunks", "--pwe", "--psnr", "--bpp"):
        assert opt in out
    rc, out, _ = run(tools, "sperr2d", "-h")
    assert rc == 0 and "--dims VALUE x 2" in out
    rc, out, _ = run(tools, "sperr3d_trunc", "-h")
    assert rc == 0 and "--pct" in out and "--orig32" in out and "--orig64" in out


@pytest.mark.parametrize("args,message", [
    ((), "What's the input file?"),
    (("vol",), "Is this compressing (-c) or decompressing (-d) ?"),
    (("vol", "-c"), "What's the dimensions of this 3D volume (--dims) ?"),
    (("vol", "-c", "--dims", 8, 8, 8), "What's the floating-type precision (--ftype) ?"),
    (("vol", "-c", "--dims", 8, 8, 8, "--ftype", 16), "What's the floating-type precision (--ftype) ?"),
    (("vol", "-c", "--dims", 8, 8, 8, "--ftype", 32), "What's the compression quality (--psnr, --pwe, --bpp) ?"),
    (("vol", "-c", "--dims", 8, 8, 8, "--ftype", 32, "--pwe", -1.0), "must be positive"),
    (("vol", "-d"), "SPERR needs an output destination when decoding!"),
    (("vol", "-c", "--dims", 8, 8, 8, "--ftype", 32, "--bpp", 1, "--chunks", 5, 8, 8, "--decomp_lowres_f", "x"),
     "cannot support multi-resolution decoding"),
])
def test_sanity_checks_of_the_reference(tools, args, message):
    rc, out, _ = run(tools, "sperr3d", *args)
    assert rc != 0 and message in out


@pytest.mark.parametrize("args,message", [
    (("vol", "-c", "-d"), "-d excludes -c"),
    (("vol", "-c", "--bpp", 2, "--pwe", 0.1), "--bpp excludes --pwe"),
    (("vol", "-c", "--psnr", 80, "--pwe", 0.1), "--psnr excludes --pwe"),
    (("vol", "-c", "--bpp", 65), "Could not convert: --bpp"),
    (("vol", "-d", "--bitstream", "out"), "--bitstream requires -c"),
    (("vol", "-d", "--print_stats", "--decomp_f", "x"), "--print_stats requires -c"),
    (("vol", "-c", "--dims", 8, 8), "--dims: 3 required"),
    (("vol", "-c", "--nonsense"), "was not expected"),
])
def test_option_rules(tools, args, message):
    rc, _, err = run(tools, "sperr3d", *args)
    assert rc != 0 and message in err


def test_wrong_file_size_and_missing_file(tools, tmp_path):
    f = tmp_path / "v.f32"
    np.zeros(100, dtype=np.float32).tofile(f)
    rc, out, _ = run(tools, "sperr3d", f, "-c", "--dims", 8, 8, 8, "--ftype", 32, "--bpp", 2)
    assert rc != 0 and "Input file size wrong!" in out
    rc, out, _ = run(tools, "sperr3d", tmp_path / "absent", "-c", "--dims", 8, 8, 8, "--ftype", 32, "--bpp", 2)
    assert rc != 0
    rc, _, err = run(tools, "sperr3d_trunc", f)
    assert rc != 0 and "--pct is required" in err
    rc, out, _ = run(tools, "sperr3d_trunc", f, "--pct", 50, "--orig32", "a", "--orig64", "b")
    assert rc != 0 and "32 or 64 bit" in out


# ---- on the GPU -------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("ftype,mode,flag,quality", [(32, 1, "--bpp", 2.5), (64, 2, "--psnr", 85.0),
                                                      (32, 3, "--pwe", 1e-3)])
def test_sperr3d_files_match_the_oracle(tools, oracle, tmp_path, ftype, mode, flag, quality):
    dims, chunks = (40, 33, 29), (20, 16, 16)      # x y z
    v = smooth_field(dims[::-1], seed=4).astype(np.float32 if ftype == 32 else np.float64)
    src, bs = tmp_path / "vol.raw", tmp_path / "vol.sperr"
    v.tofile(src)
    rc, out, err = run(tools, "sperr3d", src, "-c", "--ftype", ftype, "--dims", *dims, "--chunks", *chunks, flag,
                       quality, "--bitstream", bs, "--decomp_f", tmp_path / "c.f32", "--print_stats")
    assert rc == 0, out + err
    want = oracle.comp_3d(v, chunks, mode, quality)
    assert bs.read_bytes() == want
    dec = oracle.decomp_3d(want, False)
    assert (tmp_path / "c.f32").read_bytes() == dec.astype(np.float32).tobytes()
    # the statistics line: bitrate and PSNR of the reconstruction
    rec = dec.astype(v.dtype).astype(np.float64).ravel()
    mse = np.mean((v.astype(np.float64).ravel() - rec) ** 2)
    psnr = 10 * np.log10((float(v.max()) - float(v.min())) ** 2 / mse)
    assert f"Bitrate = {len(want) * 8 / v.size:.2f}, PSNR = {psnr:.2f}dB" in out
    assert f"L-Infty = {np.abs(v.astype(np.float64).ravel() - rec).max():.2e}" in out
    # decoding the file
    rc, out, err = run(tools, "sperr3d", bs, "-d", "--decomp_d", tmp_path / "d.f64", "--decomp_f", tmp_path / "d.f32")
    assert rc == 0, out + err
    assert (tmp_path / "d.f64").read_bytes() == dec.tobytes()
    assert (tmp_path / "d.f32").read_bytes() == dec.astype(np.float32).tobytes()


@pytest.mark.gpu
def test_sperr3d_lowres_and_truncation(tools, oracle, tmp_path):
    dims, chunks = (64, 64, 32), (32, 32, 32)
    v = smooth_field(dims[::-1], seed=9).astype(np.float32)
    src, bs = tmp_path / "vol.f32", tmp_path / "vol.sperr"
    v.tofile(src)
    rc, out, err = run(tools, "sperr3d", src, "-c", "--ftype", 32, "--dims", *dims, "--chunks", *chunks, "--bpp", 4,
                       "--bitstream", bs)
    assert rc == 0, out + err
    want = oracle.comp_3d(v, chunks, 1, 4.0)
    assert bs.read_bytes() == want
    rc, out, err = run(tools, "sperr3d", bs, "-d", "--decomp_lowres_d", tmp_path / "low", "--decomp_lowres_f",
                       tmp_path / "lowf")
    assert rc == 0, out + err
    vol, levels = oracle.decomp_3d_multi_res(want, False)
    assert len(levels) > 0
    for lv in levels:
        z, y, x = lv.shape
        assert (tmp_path / f"low.{x}x{y}x{z}").read_bytes() == lv.tobytes()
        assert (tmp_path / f"lowf.{x}x{y}x{z}").read_bytes() == lv.astype(np.float32).tobytes()
    # truncation: the file sperr_trunc_3d would give, and the quality line
    cut = tmp_path / "cut.sperr"
    rc, out, err = run(tools, "sperr3d_trunc", bs, "--pct", 40, "-o", cut, "--orig32", src)
    assert rc == 0, out + err
    want_cut = oracle.trunc_3d(want, 40)
    assert cut.read_bytes() == want_cut
    assert f"Truncation resulting BPP = {len(want_cut) * 8 / v.size:.2f}" in out
    rec = oracle.decomp_3d(want_cut, False).astype(np.float32).astype(np.float64).ravel()
    mse = np.mean((v.astype(np.float64).ravel() - rec) ** 2)
    assert f"PSNR = {10 * np.log10((float(v.max()) - float(v.min())) ** 2 / mse):.2f}" in out


@pytest.mark.gpu
@pytest.mark.parametrize("ftype,mode,flag,quality", [(32, 1, "--bpp", 3.0), (64, 3, "--pwe", 1e-4),
                                                      (32, 2, "--psnr", 90.0)])
def test_sperr2d_files_match_the_oracle(tools, oracle, tmp_path, ftype, mode, flag, quality):
    dims = (77, 50)     # x y
    img = smooth_field((1,) + dims[::-1], seed=2)[0].astype(np.float32 if ftype == 32 else np.float64)
    src, bs = tmp_path / "img.raw", tmp_path / "img.sperr"
    img.tofile(src)
    rc, out, err = run(tools, "sperr2d", src, "-c", "--ftype", ftype, "--dims", *dims, flag, quality, "--bitstream", bs,
                       "--decomp_d", tmp_path / "c.f64", "--print_stats")
    assert rc == 0, out + err
    want = oracle.comp_2d(img, mode, quality, True)
    assert bs.read_bytes() == want
    dec = oracle.decomp_2d(want[10:], img.shape, False)
    assert (tmp_path / "c.f64").read_bytes() == dec.tobytes()
    assert f"Bitrate = {len(want) * 8 / img.size:.2f}" in out
    rc, out, err = run(tools, "sperr2d", bs, "-d", "--decomp_f", tmp_path / "d.f32", "--decomp_lowres_d",
                       tmp_path / "low", "--decomp_lowres_f", tmp_path / "lowf")
    assert rc == 0, out + err
    assert (tmp_path / "d.f32").read_bytes() == dec.astype(np.float32).tobytes()
    _, levels = oracle.decomp_2d_multi_res(want[10:], img.shape)
    assert len(levels) == 3
    for lv in levels:
        y, x = lv.shape
        assert (tmp_path / f"low.{x}x{y}").read_bytes() == lv.tobytes()
        assert (tmp_path / f"lowf.{x}x{y}").read_bytes() == lv.astype(np.float32).tobytes()
