"""`python bench.py --gpus N` must start itself: the driver's N = 1 command shape (`python bench.py --gpus 1 ...`) repeated at N = 8
has no launcher around it.  These run without a GPU: the command that would be started, the fail-fast line on a node with too few
devices, and the pass-through of rank 0's line (Hnsw::search takes &self, core/lib.rs:352-356)."""
import json
import os
import subprocess
import sys
import time
import types

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_command_is_the_drivers_shape():
    cmd = bench.self_launch_command(["--gpus", "8", "--steps", "20", "--warmup", "3"], 8, 29517)
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                   "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "3"]


def test_gpus_2_without_devices_fails_fast_with_one_json_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    took = time.time() - t0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 2 and len(lines) == 1, (r.returncode, r.stdout, r.stderr[-400:])
    line = json.loads(lines[0])
    assert line["value"] is None and line["n_gpus"] == 2 and "device(s) visible" in line["error"]
    assert "Traceback" not in r.stderr
    assert took < 120, took          # (the first `import torch` of a fresh container pages the image in; afterwards this is ~3 s)


def test_self_launch_passes_rank0_line_through(monkeypatch, capsys):
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return types.SimpleNamespace(returncode=0, stdout="noise\n" + json.dumps({"value": 1.0, "n_gpus": 4}) + "\n")

    monkeypatch.setattr(subprocess, "run", fake_run)
    args = bench.parse_args(["--gpus", "4"])
    assert bench.self_launch(args, ["--gpus", "4"]) == 0
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1]) == {"value": 1.0, "n_gpus": 4}
    assert seen["cmd"][:5] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4"]
    assert seen["cmd"][-2:] == ["--gpus", "4"] and seen["kw"]["timeout"] == args.launch_timeout
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"

    # ranks that die without a line: still ONE JSON line, and a non-zero exit code
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: types.SimpleNamespace(returncode=1, stdout="boom\n"))
    assert bench.self_launch(args, ["--gpus", "4"]) == 1
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["value"] is None and "no result line" in line["error"]
