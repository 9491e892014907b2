"""The C++ host mirror of the Rust API (instant-distance_amd/host/instant_distance.hpp): builds
tests/host/all.cpp — a transcription of the reference's tests/all.rs + examples/colors.rs — against
the C ABI and runs it.  CPU: emulated kernels, small n.  GPU: libidist.so, the reference's n = 1024."""
import os
import subprocess

import pytest

import engines

ROOT = engines.ROOT


def _build_and_run(lib_path, n, tmp_path):
    exe = str(tmp_path / "host_all")
    libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)[3:-3]
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host", "all.cpp"), "-o", exe,
                           f"-L{libdir}", f"-l{libname}", f"-Wl,-rpath,{libdir}", "-pthread"])
    out = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host api ok" in out.stdout


def test_host_cpp_emulated(tmp_path):
    _build_and_run(engines.build_emu(), 150, tmp_path)


@pytest.mark.gpu
def test_host_cpp_gpu(tmp_path):
    _build_and_run(os.path.join(ROOT, "instant-distance_amd", "csrc", "libidist.so"), 1024, tmp_path)
