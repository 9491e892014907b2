"""The drop-in boundary itself (CPU, no compute): libidist.so loads, exports exactly what include/idist.h
declares, the ctypes table of the host layer covers every declaration, the host-only entry points work
without a GPU and every compute entry point fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "idist.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:idist_status|void|const\s+char\s*\*)\s+(idist_\w+)\s*\(", src, flags=re.M)
    assert len(names) == len(set(names))
    return set(names)


@pytest.fixture(scope="module")
def real_lib():
    from instant_distance_amd import _capi

    if not os.path.exists(_capi.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.dirname(_capi.LIB_PATH)], stdout=subprocess.DEVNULL)
    return _capi.Lib(_capi.LIB_PATH)


def test_header_declarations_match_ctypes_table():
    from instant_distance_amd import _capi

    decl = declared_functions()
    assert len(decl) >= 25
    assert decl == set(_capi.SYMBOLS), (decl ^ set(_capi.SYMBOLS))


def test_library_exports_every_declared_symbol(real_lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", real_lib.path], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = declared_functions() - exported
    assert not missing, missing
    stray = {s for s in exported if s.startswith("idist_")} - declared_functions()
    assert not stray, f"exported but undeclared: {stray}"


def test_host_only_entry_points_work_without_a_gpu(real_lib):
    L = real_lib
    assert b"gfx950" in L.idist_version()
    cfg = L.default_config()
    assert (cfg.ef_search, cfg.ef_construction, cfg.has_heuristic, cfg.extend_candidates, cfg.keep_pruned) == (100, 100, 1, 0, 1)
    assert abs(cfg.ml - 1.0 / np.log(32.0)) < 1e-6                     # core/lib.rs:107
    cum = np.zeros(64, dtype=np.uint32)
    nl = C.c_uint32(0)
    L.check(L.idist_layer_sizes(1_000_000, cfg.ml, cum.ctypes.data_as(C.POINTER(C.c_uint32)), 64, C.byref(nl)))
    assert list(cum[: nl.value]) == [1000000, 288539, 83254, 24022, 6931, 1999, 576, 166, 47]
    pid = np.zeros(10, dtype=np.uint32)
    order = np.zeros(10, dtype=np.uint32)
    L.check(L.idist_permutation(7, 10, pid.ctypes.data_as(C.POINTER(C.c_uint32)), order.ctypes.data_as(C.POINTER(C.c_uint32))))
    assert sorted(pid.tolist()) == list(range(10)) and all(pid[order[i]] == i for i in range(10))


def test_compute_entry_points_fail_loudly_without_a_gpu(real_lib):
    L = real_lib
    if L.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the -m gpu tests")
    from instant_distance_amd import _capi

    pts = np.zeros((4, 3), dtype=np.float32)
    cfg = L.default_config()
    h = C.c_void_p()
    st = L.idist_index_build(pts.ctypes.data_as(C.POINTER(C.c_float)), 4, 3, C.byref(cfg), 0, C.byref(h))
    assert st == 2 and not h.value                                     # IDIST_ERR_NO_DEVICE, nothing built
    assert b"no CPU path" in L.idist_last_error()
    with pytest.raises(_capi.IdistError):
        L.check(st)


def test_header_is_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/idist.h must compile as C (not only as C++), warnings-clean."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "idist.h"\nint main(void) { idist_config c; idist_index_info i; idist_build_stats s; '
                   '(void)c; (void)i; (void)s; return 0; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(root, "include"), str(src)])


def test_product_library_reads_three_environment_knobs_only():
    """libidist.so: IDIST_COMBINE, IDIST_SYNC, IDIST_KERNEL_EVENTS.  Every other IDIST_* knob (which walk, which schedule, which
    visited set ...) is compiled out of the product (test_env() in idist_capi.hip) and lives in libidist_variants.so, the test
    build: no environment variable can change which kernels the product runs or which graph it builds."""
    import re

    from instant_distance_amd import _capi

    def knob_names(path):
        blob = open(path, "rb").read()
        return {m.decode() for m in re.findall(rb"IDIST_[A-Z0-9_]+", blob)}

    allowed = {"IDIST_COMBINE", "IDIST_SYNC", "IDIST_KERNEL_EVENTS"}
    not_knobs = {"IDIST_M", "IDIST_M2", "IDIST_TIES_DROP"}               # constants of include/idist.h named in error messages
    assert knob_names(_capi.LIB_PATH) - not_knobs == allowed
    variants = os.path.join(os.path.dirname(_capi.LIB_PATH), "libidist_variants.so")
    assert {"IDIST_WALK", "IDIST_BUILD_PIPELINE", "IDIST_BUILD_GROWTH", "IDIST_TAB_LOG2", "IDIST_W2_EF"} <= knob_names(variants)
