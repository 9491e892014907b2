import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite is dominated by the lockstep emulator and every test is independent: with pytest-xdist installed and
    no explicit -n, a `-m "not gpu"` run is spread over the host cores (a quarter of an hour becomes two minutes).  Runs that
    touch the GPU (`-m gpu`, or no marker expression at all) stay in one process; IDIST_TEST_WORKERS=0 switches this off."""
    opt = config.option
    if os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get("IDIST_TEST_WORKERS", "") == "0":
        return None
    if getattr(opt, "numprocesses", 0) is None and (opt.markexpr or "").strip() == "not gpu" and not getattr(opt, "usepdb", False):
        try:
            import xdist  # noqa: F401
        except ImportError:
            return None
        n = int(os.environ.get("IDIST_TEST_WORKERS", 0)) or min(6, os.cpu_count() or 1)
        if n > 1:
            opt.numprocesses = n
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/libidist_oracle.so on demand."""
    from oracle import pyoracle

    pyoracle.build_lib()
    pyoracle.lib()
    return pyoracle


@pytest.fixture
def engine_loader(monkeypatch):
    """Returns load(kind) -> the instant_distance_amd package bound to the requested engine."""
    import engines

    def load(kind):
        import instant_distance_amd as ida
        from instant_distance_amd import _capi

        if kind == "emu":
            monkeypatch.setattr(_capi, "_singleton", _capi.Lib(engines.build_emu()))
        else:
            monkeypatch.setattr(_capi, "_singleton", _capi.Lib(_capi.LIB_PATH))
            assert _capi.lib().device_count() >= 1, "-m gpu tests need an MI355X"
        load.kind = kind
        return ida

    return load


@pytest.fixture
def sizes(request):
    """Problem sizes: tiny under the emulator, reference-sized on the GPU."""
    kind = request.node.callspec.params.get("ida", None) if hasattr(request.node, "callspec") else None
    if kind == "gpu":
        return {"random_n": 1024, "self_n": 1024}
    return {"random_n": 160, "self_n": 48}
