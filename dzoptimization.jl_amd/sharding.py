"""Host-side sharding of INDEPENDENT optimizer instances across the GPUs of one node.

The reference's model is "run multiple optimizers in parallel" (README.md:12): instances share
nothing, so the only inter-GPU exchange is the global convergence flag.  One process per GPU
(`torch.distributed`; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests);
no data-path collective exists or is needed.  A 4-byte all-reduce is latency-only, so it is
issued every `poll` steps, not every step.
"""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int) -> range:
    """Block partition of instances [0, total): rank r owns [r*total/world, (r+1)*total/world)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    lo = (rank * total) // world
    hi = ((rank + 1) * total) // world
    return range(lo, hi)


class ConvergenceFlag:
    """all_done = MIN over ranks of local_done, all-reduced every `poll` calls.

    With ``comm`` (a ``dzo.Comm``) the all-reduce is the library's own ``dzo_flag_allreduce_min`` (RCCL
    behind the C ABI -- what a Julia host calls too).  Without it the flag goes through
    ``torch.distributed`` (the gloo rehearsal of the CPU tests, or a 1-GPU box where two ranks cannot
    share a device under RCCL)."""

    def __init__(self, poll: int = 10, device=None, comm=None):
        self.poll = max(1, int(poll))
        self.comm = comm
        self._calls = 0
        self.collectives = 0
        self.all_done = False
        if comm is not None:
            self.world = comm.nranks
            self.transport = "dzo_flag_allreduce_min (RCCL behind the C ABI)"
            return
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if device is None:
            device = "cuda" if (dist.is_initialized() and dist.get_backend() == "nccl") else "cpu"
        self._buf = torch.zeros(1, dtype=torch.int32, device=device)
        self.transport = f"torch.distributed all_reduce ({dist.get_backend() if dist.is_initialized() else 'single process'})"

    def update(self, local_done: bool, force: bool = False) -> bool:
        """Call once per step; returns the most recent global flag."""
        self._calls += 1
        if force or self._calls % self.poll == 0:
            if self.world > 1 and self.comm is not None:
                self.all_done = bool(self.comm.allreduce_min(1 if local_done else 0))
                self.collectives += 1
            elif self.world > 1:
                self._buf.fill_(1 if local_done else 0)
                self._dist.all_reduce(self._buf, op=self._dist.ReduceOp.MIN)
                self.all_done = bool(int(self._buf.item()))
                self.collectives += 1
            else:
                self.all_done = bool(local_done)
        return self.all_done


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
