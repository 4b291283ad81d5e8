"""CPU: bench.py's launcher contract — `--gpus N` is what runs, or the run is refused (no GPU call before the decision)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_must_agree_with_gpus_flag():
    import bench
    a = bench.parse_args(["--gpus", "1"])
    assert bench.resolve_world(a, {}) == ("run", 1)
    assert bench.resolve_world(bench.parse_args(["--gpus", "4"]), {}) == ("spawn", 4)          # bare `python bench.py --gpus 4`
    assert bench.resolve_world(bench.parse_args(["--gpus", "4"]), {"WORLD_SIZE": "4"}) == ("run", 4)
    with pytest.raises(SystemExit) as e:
        bench.resolve_world(a, {"WORLD_SIZE": "2"})
    assert e.value.code == 2
    with pytest.raises(SystemExit):
        bench.resolve_world(bench.parse_args(["--gpus", "8"]), {"WORLD_SIZE": "1"})


def test_gpus_1_under_world_size_2_refuses_before_touching_the_gpu():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2, r.stderr
    assert "refusing" in r.stderr and r.stdout.strip() == ""


def test_bare_gpus_n_starts_n_rank_processes(monkeypatch):
    """the spawn path hands the same argv to torch.distributed.run with --nproc-per-node N (the child processes are not run here)"""
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    rc = bench.spawn_ranks(2, ["--gpus", "2", "--backend", "gloo", "--steps", "3"])
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "2", "--backend", "gloo", "--steps", "3"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_ab_switches_and_unroll_list_parse():
    """--set NAME=VALUE (same-box A/B of ops switches) and --unrolls a,b,c (steps per replayed graph)"""
    import bench
    a = bench.parse_args(["--set", "DG64=0", "--set", "HASH_VERTEX_FUSION=1", "--unrolls", "10,4,2"])
    assert a.set == ["DG64=0", "HASH_VERTEX_FUSION=1"] and a.unrolls == (10, 4, 2)
    assert bench.parse_args([]).set == [] and 4 in bench.parse_args([]).unrolls
