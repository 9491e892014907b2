import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
