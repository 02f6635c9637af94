"""bench.py's multi-GPU plumbing on CPU (no GPU needed): `python bench.py --gpus N` with no launcher must start the
driver's own form -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` -- as a
child process, pass every argument through, hand each rank its RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR, and rank 0
prints ONE JSON line with n_gpus = N."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_command_is_the_drivers_multi_gpu_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(["--gpus", "4", "--steps", "7", "--warmup", "2"], 4, 29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "launch_check"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert sorted(x[0] for x in out["ranks"]) == [0, 1] and sorted(x[1] for x in out["ranks"]) == [0, 1]
    assert all(x[2] == "127.0.0.1" and x[3] == "1" for x in out["ranks"])


def test_a_rank_count_that_contradicts_gpus_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "launch_check"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)
