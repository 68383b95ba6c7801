"""CPU: the host-only code of the container layer (sperr_amd/csrc/host_container.hpp -- chunk grid,
container header parsing, sperr_trunc_3d's byte surgery; restated from
/root/reference/src/sperr_helper.cpp:542-592 and src/SPERR3D_Stream_Tools.cpp:46-226) compiled for
the CPU with AddressSanitizer and UndefinedBehaviorSanitizer and run over damaged containers
(tests/cpp/host_fuzz.cpp).  Sanitizers are not available for the GPU on this pool, and this is
the part of the library that parses untrusted bytes."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_container_host_code_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_fuzz"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "cpp", "host_fuzz.cpp"), "-o", str(exe)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([str(exe), "50"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "host fuzz ok" in p.stdout
