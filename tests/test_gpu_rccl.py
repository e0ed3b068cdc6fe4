"""-m gpu: the RCCL transport of the cross-rank step under test (round-4 verdict: `dlopen("librccl.so.1")`,
`ncclCommInitRank`, `ncclAllGather` and the grouped `ncclSend` / `ncclRecv` of csrc/allreduce.cpp had run only in the
builder's manual `bench.py --force-distributed`).  One GPU per box: world 1 -- every collective is executed by RCCL,
with itself as the only peer -- in a CHILD process (torch.distributed's NCCL process group carries the unique id), its
results held against the plain state's on the same columns.  World > 1 over the same code: the threaded-rank tests
(tests/test_gpu_distributed_sim.py) and, on CPU, gloo (tests/test_distributed_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_child(args, port):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py")] + args, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("suite,rows", [("headline", 8_000_000), ("full", 3_000_000)])
def test_world1_step_over_rccl_equals_the_plain_state(suite, rows):
    import bench

    got = run_child(["--rows", str(rows), "--suite", suite], bench.free_port())
    assert got["rows"] == rows // 64 * 64 and got["specs"] >= 34
    for i, (a, b) in enumerate(zip(got["sharded"], got["plain"])):
        assert a == b, (i, a, b)   # counts, extremes, sums (bit patterns of the doubles), distinct counts
    assert got["sharded"][-1][0] == "distinct"
    id_distinct = [r for r in got["sharded"] if r[0] == "distinct"][0]
    assert id_distinct[3] == got["rows"]   # the id column is a permutation of 0 .. n-1
    if suite == "full":
        # (the sharded state's sketch has been packed, gathered and merged into an emptied state: another sketch of the
        #  same stream -- equal weight, quantiles within the stated rank error of each other; kll_sketch.rs:397-399)
        eps = 1.65 / 200 ** 0.5
        for a, b in zip(got["kll_sharded"], got["kll_plain"]):
            assert abs(a - b) < eps * max(1.0, abs(b)), (a, b)


def test_bench_force_distributed_line():
    """`bench.py --force-distributed` (the self-test of the N > 1 path on a 1-GPU box) at a small size: a verified line"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    import bench

    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(bench.free_port())})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-distributed", "--rows", "16000000",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["verified"] is True and line["config"]["rows_total"] == 16_000_000
    assert line["roofline"]["frac"] > 0
