"""Which engine a test module runs on.

  * "emu": the product's csrc compiled for the CPU lockstep emulator (tests/simt) — test
    infrastructure that lets the real kernel source + C-ABI host code run without a GPU.
    Selected only here, by swapping the ctypes handle inside the test process; the product
    has no switch for it.
  * "gpu": instant-distance_amd/csrc/libidist.so on a real MI355X (pytest -m gpu).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "simt", "_build", "libidist_emu.so")


def engine_params():
    return [pytest.param("emu", id="emu"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]


def build_emu():
    srcs = [os.path.join(ROOT, "instant-distance_amd", "csrc", f) for f in
            ("idist_capi.hip", "idist_kernels.hpp", "idist_device.hpp", "idist_mfma.hpp", "idist_combine.hpp")]
    srcs += [os.path.join(ROOT, "tests", "simt", f) for f in ("hip_emu.hpp", "hip_emu.cpp")]
    def stale():
        return not os.path.exists(EMU_SO) or os.path.getmtime(EMU_SO) < max(os.path.getmtime(s) for s in srcs)

    if stale():
        # pytest-xdist workers race for the rebuild: one builds (into a temporary name, renamed when complete), the others wait
        import fcntl

        os.makedirs(os.path.dirname(EMU_SO), exist_ok=True)
        with open(EMU_SO + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                subprocess.check_call([os.path.join(ROOT, "tests", "simt", "build_emu.sh")], stdout=subprocess.DEVNULL)
    return EMU_SO
