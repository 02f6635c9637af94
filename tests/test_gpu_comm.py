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
