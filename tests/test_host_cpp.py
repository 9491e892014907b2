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


def _build_and_run_threads(lib_path, args, tmp_path):
    exe = str(tmp_path / "host_threads")
    libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)[3:-3]
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "host", "threads.cpp"), "-o", exe,
                           f"-L{libdir}", f"-l{libname}", f"-Wl,-rpath,{libdir}", "-pthread"])
    out = subprocess.run([exe, *[str(a) for a in args]], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "threads ok" in out.stdout
    return out.stdout


def test_scalar_calls_one_thread_emulated(tmp_path):
    """(the CPU emulator is single-threaded by design: one thread, the same call sequence)"""
    _build_and_run_threads(engines.build_emu(), (220, 8, 1, 12), tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [4, 32])
def test_scalar_calls_from_native_threads_gpu(tmp_path, threads):
    """tests/host/threads.cpp: native threads, one context each, scalar calls on one shared index == one wide batch call,
    byte for byte — below and beyond the eight launches in flight from which calls are combined (idist_combine.hpp)."""
    out = _build_and_run_threads(os.path.join(ROOT, "instant-distance_amd", "csrc", "libidist.so"), (20000, 96, threads, 60), tmp_path)
    print(out)
