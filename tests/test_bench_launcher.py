"""bench.py --gpus N: the line's n_gpus can never disagree with --gpus (invoked bare with --gpus 8 the
script used to benchmark ONE rank and print n_gpus 1).  Bare, it starts the N ranks itself as children -- the
driver's own torch.distributed.run form, rendezvous on 127.0.0.1 -- and refuses when the machine has fewer GPUs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_builds_the_contract_command_and_environment():
    import bench

    cmd = bench.launcher_command(8, 29611, ["--gpus", "8", "--steps", "20", "--warmup", "3"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]
    env = bench.launcher_env({"RANK": "3", "WORLD_SIZE": "4", "LOCAL_RANK": "3", "PATH": "/usr/bin"})
    assert "RANK" not in env and "WORLD_SIZE" not in env and "LOCAL_RANK" not in env  # the launcher sets its own
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin"
    assert 1024 < bench.free_port() < 65536


def test_gpus_are_counted_from_the_kfd_topology_without_touching_one(tmp_path, monkeypatch):
    import bench

    for i, simds in enumerate([0, 0, 1024, 1024, 1024]):  # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (96 if simds == 0 else 0, simds))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.count_gpus(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.count_gpus(str(tmp_path)) == 2
    assert bench.count_gpus(str(tmp_path / "absent")) >= 0  # (no topology: torch's count)


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_more_ranks_than_gpus_fails_loudly():
    import bench

    n = bench.count_gpus() + 1  # one more than the machine has (here: no GPU at all)
    if n < 2:
        n = 2
    p = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0 and "refusing to run fewer ranks" in p.stderr and '"metric"' not in p.stdout


def test_world_size_that_disagrees_with_gpus_is_refused():
    p = _run(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "--gpus 8 but WORLD_SIZE=1" in p.stderr and '"metric"' not in p.stdout
    p = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in p.stderr
