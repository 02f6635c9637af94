"""The C-ABI convergence-flag collective (include/dzo.h, csrc/dzo_comm.hip) on the one GPU a test box
has: a 1-rank RCCL communicator built both ways (ncclCommInitAll and unique id + ncclCommInitRank),
the all-reduce itself, and dzo_bfgs_batch_all_done on a shard created with an explicit device.  More
ranks need more GPUs (RCCL refuses two ranks on one device); the N > 1 logic is rehearsed on gloo in
tests/test_sharding_gloo.py and runs for real in bench.py --gpus N."""
import numpy as np
import pytest

from dzo_loader import dzo
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("how", ["init_all", "init_rank"])
def test_one_rank_communicator_allreduces_the_flag(how):
    dzo.init(0)
    comm = dzo.Comm.init_all([0]) if how == "init_all" else dzo.Comm.init_rank(dzo.Comm.unique_id(), 1, 0)
    assert (comm.nranks, comm.nlocal, comm.first_rank) == (1, 1, 0)
    assert comm.allreduce_min(1) == 1
    assert comm.allreduce_min(0) == 0
    assert comm.allreduce_min([1]) == 1
    assert comm.collectives == 3
    comm.close()


def test_batch_all_done_with_and_without_communicator():
    n, B = 8, 48
    X0 = np.stack([orc.pcg_fill(n, 1000 + b) for b in range(B)])
    batch = dzo.BatchedBFGS(dzo.ROSENBROCK_CHAIN, X0, 1.0, device=0)
    assert batch.device == 0
    comm = dzo.Comm.init_all([0])
    assert not comm.all_done([batch]) and not dzo.batches_all_done([batch])
    rounds = 0
    while not comm.all_done([batch]) and rounds < 200:
        batch.step(8, poll=False)
        rounds += 1
    assert comm.all_done([batch]) and dzo.batches_all_done([batch]) and batch.count_active() == 0
    with pytest.raises(dzo.DzoError):
        comm.all_done([batch, batch])                      # one shard per local rank
    comm.close()


def test_bad_arguments_are_errors_not_crashes():
    dzo.init(0)
    with pytest.raises(dzo.DzoError):
        dzo.Comm.init_all([0, 0])                          # a device listed twice
    with pytest.raises(dzo.DzoError):
        dzo.Comm.init_all([99])
    with pytest.raises(dzo.DzoError):
        dzo.Comm.init_rank(dzo.Comm.unique_id(), 2, 5)


def test_bench_gpus_2_runs_over_rccl_when_two_devices_are_visible():
    """`python bench.py --gpus 2` on a box with two GPUs: two ranks, the library's own RCCL communicator of size 2, and the
    sharded quantity (config 5) next to `value` at the top level of the line.  Skipped on the 1-GPU
    build box -- RCCL refuses two ranks on one device; the 2-rank gloo rehearsal (tests/test_sharding_gloo.py,
    tests/test_bench_launch.py) is what runs there."""
    import json
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--dim", "1000000"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["rccl_world_size"] == 2
    assert out["sharded_value"] > 0 and len(out["sharded_per_rank"]) == 2 and out["sharded_instances_total"] == 2048
    assert out["batched"]["config"]["rccl_world_size"] == 2
