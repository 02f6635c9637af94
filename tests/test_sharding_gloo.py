"""N > 1 path on CPU: two gloo ranks exercise the instance sharding and the convergence-flag
all-reduce that bench.py uses over RCCL (the only collective in the design)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dzo_loader import dzo  # noqa: F401  (registers the package)
import importlib

sharding = importlib.import_module("dzoptimization_jl_amd.sharding")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from dzo_loader import dzo as _  # noqa
        sh = importlib.import_module("dzoptimization_jl_amd.sharding")
        mine = sh.shard_range(total, rank, world)
        # every rank "optimises" its own instances: instance b converges after (b % 7) + 3 steps
        remaining = {b: (b % 7) + 3 for b in mine}
        flag = sh.ConvergenceFlag(poll=4)
        steps = 0
        while not flag.all_done and steps < 100:
            for b in remaining:
                remaining[b] = max(0, remaining[b] - 1)
            steps += 1
            flag.update(all(v == 0 for v in remaining.values()))
        tmax = sh.max_over_ranks(float(rank + 1))
        tot = sh.sum_over_ranks(float(len(mine)))
        out.put((rank, list(mine), steps, flag.collectives, tmax, tot))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 13])
def test_two_rank_sharding_and_convergence_flag(total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = res[0][1] + res[1][1]
    assert sorted(owned) == list(range(total))                      # disjoint cover
    assert abs(len(res[0][1]) - len(res[1][1])) <= 1                # balanced
    # every instance needs <= 9 steps; the flag is polled every 4 steps -> both ranks stop at 12
    assert res[0][2] == res[1][2] == 12
    assert res[0][3] == res[1][3] == 3                              # one 4-byte collective per poll
    assert res[0][4] == res[1][4] == 2.0                            # MAX over ranks (timing reduction)
    assert res[0][5] == res[1][5] == float(total)


def test_shard_range_properties():
    for total in (0, 1, 7, 8192):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                cover += list(sharding.shard_range(total, r, world))
            assert cover == list(range(total))
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def test_single_process_flag_needs_no_process_group():
    f = sharding.ConvergenceFlag(poll=3)
    assert not f.update(True) and not f.update(True)
    assert f.update(True) and f.collectives == 0
