"""The launch combiner behind scalar `Hnsw::search` calls from many threads (instant-distance_amd/csrc/idist_combine.hpp;
the reference's concurrency model is one `Search` per thread on a shared index, core/lib.rs:352-356).  Host-only C++: the
CPU harness tests/host/combine_test.cpp hammers it with real threads and a fake launch.  The GPU side — scalar calls from
16 threads answered like the oracle's — is tests/test_shards.py::test_scalar_calls_from_many_threads_gpu."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("threads,calls,leaders,batch,throw_every", [(32, 150, 4, 6, 0), (3, 200, 4, 6, 0), (64, 60, 8, 96, 0), (9, 300, 1, 2, 0),
                                                                     (16, 100, 8, 3, 0), (32, 150, 4, 6, 7), (9, 300, 1, 2, 3)])
def test_combiner_serves_every_request_once_and_never_deadlocks(tmp_path, threads, calls, leaders, batch, throw_every):
    exe = str(tmp_path / "combine_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", os.path.join(ROOT, "tests", "host", "combine_test.cpp"), "-o", exe])
    out = subprocess.run([exe, str(threads), str(calls), str(leaders), str(batch), str(throw_every)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "combiner ok" in out.stdout
