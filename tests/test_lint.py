"""GPU-marked tests never run on the CPU box: catch undefined names (e.g. a fixture used but not requested)
statically, so that a test-only edit cannot turn the MI355X run red (round-1 lesson)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_undefined_names_in_python_sources():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import lint_names
    finally:
        sys.path.pop(0)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        found = []
        for r in ["tests", "bench.py", "__graft_entry__.py", "instant-distance_amd", "oracle", "scripts"]:
            files = [r] if r.endswith(".py") else [os.path.join(d, f) for d, _, fs in os.walk(r) for f in fs if f.endswith(".py")]
            for f in files:
                found += [(f,) + b for b in lint_names.check(f)]
        assert not found, found
    finally:
        os.chdir(cwd)


def test_every_gpu_test_collects():
    """`pytest --collect-only -m gpu` must import every test module and resolve every parametrisation."""
    import subprocess

    r = subprocess.run([sys.executable, "-m", "pytest", "--collect-only", "-q", "-m", "gpu", os.path.join(ROOT, "tests")],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
