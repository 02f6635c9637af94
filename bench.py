#!/usr/bin/env python3
"""bench.py -- step!() calls/s and achieved HBM GB/s of the MI355X-native L-BFGS hot path.

Workload (BASELINE.json metric / configs[2]): L-BFGS, m = 20, on the N-D chained Rosenbrock
function, n = 10^7, fp64, one MI355X.  A "step" is one full step!(opt)
(src/DZOptimization.jl:454-509): two-loop direction, backtracking trial(s) with the device
objective, gradient, delta_gradient + rho, history push.  State is resident in HBM before
the timed region starts; the only per-step host traffic is the 40-byte {f_new, changed}
read-back the reference's host-side `f_new < f` test needs.

N > 1: one process per GPU (`python -m torch.distributed.run ... bench.py --gpus N`, or plain
`python bench.py --gpus N`, which starts exactly that launcher as a child process before anything
touches the GPU and exits with its return code).  Each rank owns an independent optimizer instance
(north_star: "batched-problems mode shards independent optimizer instances across the GPUs ... RCCL
only for the global convergence flag; single huge-n problems stay on one GPU"), so the headline
fields are config 3 replicated per GPU ("replicas only"), and the line carries a
`batched` object with the quantity north_star really shards: config 5, 1024 independent dense-BFGS
instances per GPU, whole-job instance-step!()/s, per-rank rates, the polls of the convergence flag.
There is no data-path collective; the flag is all-reduced (MIN over one int32) through the library's
own RCCL communicator (dzo_comm_* behind the C ABI) -- a hard requirement: if that communicator cannot
be created the run fails (BENCH_DIST_BACKEND=gloo is the explicit CPU-transport rehearsal for a 1-GPU
box, where two ranks cannot share a device under RCCL).  scaling = "weak".

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and
`cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured copy


# ------------------------------------------------------------------------------ inputs
_PCG_TABLE = {}


def _pcg_table(block):
    """A_i, C_i with state_i = A_i * state_0 + C_i (mod 2^64), i <= block; built once per process."""
    if block not in _PCG_TABLE:
        A = np.empty(block + 1, np.uint64)
        Cc = np.empty(block + 1, np.uint64)
        a, c = 1, 0
        M = (1 << 64) - 1
        mi, ii = 0x5851F42D4C957F2D, 0x14057B7EF767814F
        A[0], Cc[0] = np.uint64(1), np.uint64(0)
        for i in range(1, block + 1):
            a = (a * mi) & M
            c = (c * mi + ii) & M
            A[i], Cc[i] = a, c
        _PCG_TABLE[block] = (A, Cc)
    return _PCG_TABLE[block]


def pcg32_uniform(n, seed):
    """Vectorised PCG32 XSH-RR stream with the reference's seeding (legacy/PCG.jl:7-22):
    u_i = 2^-32 * extract(state_i), state_0 = advance(inc + seed)."""
    block = 1 << 16 if n > 4096 else 1 << 12
    A, Cc = _pcg_table(block)
    M = (1 << 64) - 1
    mi, ii = 0x5851F42D4C957F2D, 0x14057B7EF767814F
    with np.errstate(over="ignore"):
        state0 = np.uint64(((ii + seed) * mi + ii) & M)
        out = np.empty(n, np.float64)
        pos = 0
        while pos < n:
            cnt = min(block, n - pos)
            st = A[:cnt] * state0 + Cc[:cnt]
            v = (((st >> np.uint64(18)) ^ st) >> np.uint64(27)).astype(np.uint32)
            r = (st >> np.uint64(59)).astype(np.uint32)
            x = (v >> r) | (v << ((np.uint32(32) - r) & np.uint32(31)))
            out[pos:pos + cnt] = x.astype(np.float64) * 2.3283064365386962890625e-10
            state0 = A[block] * state0 + Cc[block]
            pos += cnt
    return out


def rosenbrock_chain_x0(n, seed=5):
    """C3 start: -1.2 / 1.0 alternating plus 0.01*(u - 1/2)."""
    u = pcg32_uniform(n, seed)
    base = np.where(np.arange(n) % 2 == 0, -1.2, 1.0)
    return base + 0.01 * (u - 0.5)


# ------------------------------------------------------------------------------ helpers
def launch_command(argv, gpus, port):
    """The command `python bench.py --gpus N` runs for N > 1 when no launcher started it: the driver's own
    multi-GPU form (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, gpus):
    """Start the N ranks as a CHILD process tree (never exec: a process that replaces itself after touching the GPU
    takes the box down, and this parent has not touched it and never will) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env["BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(launch_command(argv, gpus, port), env=env)


def _dist_setup(gpus, use_gpu=True):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != gpus:
        raise SystemExit(f"--gpus {gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    # rehearsal knobs (a 1-GPU box cannot run RCCL between two ranks on the same device):
    #   BENCH_DIST_BACKEND=gloo BENCH_FORCE_DEVICE=0  -> both ranks on cuda:0, flag all-reduced over gloo
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if "BENCH_FORCE_DEVICE" in os.environ:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    if not use_gpu:
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        return world, rank, local
    import torch
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return world, rank, local


def _make_comm(dzo, world):
    """N > 1: the convergence flag goes through dzo_flag_allreduce_min / dzo_bfgs_batch_all_done (the C-ABI collective
    a Julia host uses too); the unique id travels over the torch.distributed group.  A HARD requirement: when the
    communicator cannot be created every rank raises the same error (Comm.from_torch_distributed keeps the ranks'
    collectives matched while it fails) and the run ends non-zero -- a line printed over some other transport would
    look like a measurement of this one.  Only BENCH_DIST_BACKEND=gloo (the explicit rehearsal on a 1-GPU box, where
    RCCL refuses two ranks on one device) keeps torch.distributed for the flag."""
    if world <= 1 or os.environ.get("BENCH_DIST_BACKEND", "nccl") != "nccl":
        return None
    comm = dzo.Comm.from_torch_distributed()
    assert comm.nranks == world, (comm.nranks, world)
    return comm


def _barrier(world):
    import torch
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def _pmc_secondary(workload, patterns, launches_each=None):
    """HBM-side bytes per launch of a secondary workload's kernel(s) from the tracked profiles/pmc_secondary_latest.json
    (builder-side rocprofv3 --pmc passes of `bench.py --workload ...`, tools/collect_pmc_secondary.sh; NOT measured in
    this process): (bytes or None, traffic_source or None).  Several patterns: the sum (a two-loop = its kernels)."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_secondary_latest.json")))
        rows = tab.get(workload, {})
        total = 0
        for i, pat in enumerate(patterns):
            hit = [v for k, v in rows.items() if pat in k]
            if not hit:
                return None, None
            mult = 1 if launches_each is None else launches_each[i]
            total += mult * (hit[0]["read_bytes_per_launch"] + hit[0]["write_bytes_per_launch"])
        return total, tab.get("_source") + "; not measured in this process"
    except Exception:                                           # noqa: BLE001 -- the figure is optional
        return None, None


def _kernel_bytes(name, n, k, esize, layout=1, regrad=False):
    """ALGORITHMIC bytes per launch (DESIGN.md section d)."""
    if name in ("lbfgs_single_pass", "lbfgs_single_pass_retry"):
        if layout == 2 and regrad:
            # point ring, gradients recomputed in registers from the points (3-point stencil): reads the k + 1 POINTS,
            # writes the trial point into the spare slot (gradient tiles are formed on demand, when the host asks)
            return (k + 2) * n * esize
        if layout == 2:
            # point ring: reads the k + 1 points and k + 1 gradients (2k + 2), writes the trial point and its gradient
            # into the spare slot (2); step_direction is formed on demand, the pairs in registers
            return (2 * k + 4) * n * esize
        # pair ring: reads s_i, y_i (2k), g, x; writes d, the trial point and its gradient (twin buffers), delta_point,
        # delta_gradient (5)
        return (2 * k + 7) * n * esize
    if name == "lbfgs_gram_pass":
        return (2 * k + 1) * n * esize           # each s_i, y_i once, g once
    if name == "lbfgs_combine":
        return (2 * k + 1) * n * esize           # each s_i, y_i once, d written (g re-read not counted)
    if name == "lbfgs_chain_link":
        return 2 * n * esize                     # (4k+2)n over 2k+1 launches, rounded per launch
    if name == "lbfgs_trial":
        return 4 * n * esize
    if name == "lbfgs_delta_rho":
        return 4 * n * esize
    if name.startswith("objective_rosenbrock_chain"):
        return n * esize
    if name.startswith("gradient_rosenbrock_chain"):
        return 2 * n * esize
    if name == "axpby":
        return 3 * n * esize
    return None


def cpu_baseline(n, m, warm, steps, threads, steps_single=None):
    """The oracle (C restatement of the reference's unfused op sequence) timed on the host cores:
    `warm` untimed steps on all cores to fill the history, `steps` timed steps on `threads` cores, then
    `steps_single` timed steps of the same optimizer on ONE core (both)."""
    from oracle import oracle as orc
    orc.set_threads(threads)
    try:
        x0 = orc.rosenbrock_chain_x0(n)
        opt = orc.LBFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n), x0, 1.0, m)
        for _ in range(warm):
            opt.step()
        t0 = time.perf_counter()
        for _ in range(steps):
            opt.step()
        dt = time.perf_counter() - t0
        single = None
        if steps_single:
            orc.set_threads(1)
            t0 = time.perf_counter()
            for _ in range(steps_single):
                opt.step()
            single = steps_single / (time.perf_counter() - t0)
        opt.close()
    finally:
        orc.set_threads(1)
    return steps / dt, single


def cpu_baseline_object(unit, threads, fn_all, fn_single, sample):
    """{"value": all-core rate, "cores": nproc, "single_thread": {...}} from two timed closures."""
    v = fn_all()
    out = {"value": round(v, 4), "unit": unit, "cores": threads, "kind": "port", "sample": sample}
    if fn_single is not None:
        out["single_thread"] = {"value": round(fn_single(), 4), "cores": 1}
    return out


def host_cores():
    """Host cores this process may really use: the affinity mask, cut by the cgroup CPU quota when there is
    one; a GPU box of the pool reports all of the host's hardware threads but gives one GPU's job a share of 16
    (task statement), so an implausibly large count falls back to that share."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    if n > 64:
        n = int(os.environ.get("BENCH_HOST_CORES", "16"))
    return max(1, n)


# ------------------------------------------------------------------------------ secondary workloads
def _timed(fn, reps):
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return reps / (time.perf_counter() - t0)


def cpu_baseline_secondary(workload, n, m=10, A=None):
    """CPU baselines of the secondary workloads: the oracle on a bounded sample, all cores and one core."""
    from oracle import oracle as orc
    threads = host_cores()
    try:
        if workload == "bfgs_dense":
            x0 = pcg32_uniform(n, 4) - 0.5
            orc.set_threads(threads)
            ref = orc.BFGS(orc.Problem(orc.QUADRATIC, n, A=A), x0, 1.0)
            for _ in range(3):
                ref.step()
            v = _timed(ref.step, 10)
            orc.set_threads(1)
            v1 = _timed(ref.step, 4)
            return {"value": round(v, 3), "unit": "step!() calls/s", "cores": threads, "kind": "port",
                    "single_thread": {"value": round(v1, 3), "cores": 1, "steps": 4},
                    "sample": f"oracle dense BFGS on the same quadratic, n={n}: 3 untimed steps, 10 timed on OpenMP x{threads}, 4 on one core"}
        if workload == "bfgs_batched":
            # independent instances, one optimizer per host core (the reference's "run multiple optimizers in
            # parallel"), in worker PROCESSES: must run before this process touches the GPU (a spawned child
            # of a GPU-initialised process may not exec)
            import multiprocessing as mp
            steps, per = 60, 4
            its1, dt1 = orc.bfgs_rate_worker((n, 1000, 1, steps))
            with mp.get_context("spawn").Pool(threads) as pool:
                pool.map(orc.bfgs_rate_worker, [(8, 1, 1, 1)] * (4 * threads), chunksize=1)     # every worker process is up and imported
                res = pool.map(orc.bfgs_rate_worker, [(n, 1001 + w * per, per, steps) for w in range(threads)], chunksize=1)
            its, dta = sum(r[0] for r in res), max(r[1] for r in res)
            return {"value": round(its / dta, 2), "unit": "instance-step!() calls/s", "cores": threads, "kind": "port",
                    "single_thread": {"value": round(its1 / dt1, 2), "cores": 1, "steps": steps},
                    "sample": f"oracle dense BFGS, chained Rosenbrock n={n}: {threads * per} independent instances x {steps} steps, "
                              f"{per} per worker process ({threads} processes, timed inside the workers); one instance alone for the single-core rate"}
        if workload == "adgd":
            orc.set_threads(threads)
            ref = orc.AdGD(orc.Problem(orc.ROSENBROCK_CHAIN, n), orc.rosenbrock_chain_x0(n), 1.0)
            for _ in range(5):
                ref.step()
            v = _timed(ref.step, 20)
            orc.set_threads(1)
            v1 = _timed(ref.step, 10)
            return {"value": round(v, 3), "unit": "step!() calls/s", "cores": threads, "kind": "port",
                    "single_thread": {"value": round(v1, 3), "cores": 1, "steps": 10},
                    "sample": f"oracle AdGD on the chained Rosenbrock objective, n={n}: 5 untimed, 20 timed steps on OpenMP x{threads}, 10 on one core"}
        if workload == "lbfgs_lse_f32":
            g, S, Y = orc.frozen_two_loop_state(n, m, np.float32)
            rho = np.array([orc.dot(S[i], Y[i]) for i in range(m)], np.float32)
            orc.set_threads(threads)
            v = _timed(lambda: orc.lbfgs_direction(g, S, Y, rho), 40)
            orc.set_threads(1)
            v1 = _timed(lambda: orc.lbfgs_direction(g, S, Y, rho), 10)
            return {"value": round(v, 3), "unit": "compute_lbfgs_step_direction! calls/s", "cores": threads, "kind": "port",
                    "single_thread": {"value": round(v1, 3), "cores": 1, "steps": 10},
                    "sample": f"oracle two-loop (4k+3 unfused BLAS-1 calls) on the same frozen state, n={n}, k={m}, fp32: 40 calls on "
                              f"OpenMP x{threads}, 10 on one core"}
    finally:
        orc.set_threads(1)
    return None


def quadratic_matrix(n, r=8):
    """C2: A = D + U U'/r, D = diag(1 + 99 u) (seed 2), U entries u - 1/2 (seed 3)."""
    dvec = 1.0 + 99.0 * pcg32_uniform(n, 2)
    U = (pcg32_uniform(n * r, 3) - 0.5).reshape(n, r, order="F")
    A = (U @ U.T) / r
    A[np.diag_indices(n)] += dvec
    return 0.5 * (A + A.T)


def batched_leg(dzo, sharding, args, world, rank, comm, info, steps, warmup):
    """Config 5 (BASELINE configs[4]): B independent dense-BFGS instances per GPU (n = 256, chained Rosenbrock, fp64), block
    partition of the instances over the ranks, the only collective the convergence flag.  Every rank runs it; returns the
    fields of the JSON line (whole-job instance-step!()/s, per-rank rates so that a straggler is visible)."""
    n = 256 if args.n == 10_000_000 else args.n
    B = args.batch
    lo = rank * B                                              # weak scaling: B instances per GPU
    X0 = np.stack([pcg32_uniform(n, 1000 + lo + b) for b in range(B)])
    batch = dzo.BatchedBFGS(dzo.ROSENBROCK_CHAIN, X0, 1.0)
    flag = sharding.ConvergenceFlag(poll=1, comm=comm)
    if world > 1 and comm is None:
        assert os.environ.get("BENCH_DIST_BACKEND", "nccl") != "nccl", "N > 1 without the RCCL communicator"
    polls = 0
    batch.step(warmup, poll=False)
    dzo.synchronize()
    dzo.profile_reset(); dzo.profile_enable(True)
    _barrier(world)
    it0 = int(batch.iteration_count.to_host().sum())
    t0 = time.perf_counter()
    chunk = max(1, args.poll)
    done_steps = 0
    while done_steps < steps:
        k = min(chunk, steps - done_steps)
        batch.step(k, poll=False)
        done_steps += k
        if comm is not None:
            comm.all_done([batch])                             # dzo_bfgs_batch_all_done: local count + one 4-byte all-reduce
        else:
            flag.update(batch.count_active() == 0)             # (single GPU, or the gloo rehearsal)
        polls += 1
    dzo.synchronize()
    el_local = time.perf_counter() - t0                        # this rank's own time, before it waits for the others
    _barrier(world)
    el = sharding.max_over_ranks(time.perf_counter() - t0)
    dzo.profile_enable(False)
    it1 = int(batch.iteration_count.to_host().sum())
    inst_steps = sharding.sum_over_ranks(float(it1 - it0))
    per_rank = [round(sharding.sum_over_ranks((it1 - it0) / el_local if r == rank else 0.0), 1) for r in range(world)]
    tab = dzo.profile_table()
    kern = {k: {"launches": v[0], "avg_us": round(1e3 * v[1] / v[0], 2)} for k, v in tab.items() if k.startswith("bfgs_batch")}
    # the step kernel reads and writes the LOWER triangle of every H only: 1.5 n^2 T per BFGS instance-step
    # (a gradient-descent step resets the triangle: 0.5 n^2 T; counted as a BFGS step here, an upper bound)
    ach = 1.5 * n * n * 8 * (it1 - it0) / (1e-3 * tab["bfgs_batch_step"][1]) / 1e9 if "bfgs_batch_step" in tab else None
    rccl_n = comm.nranks if comm is not None else None
    if world > 1 and os.environ.get("BENCH_DIST_BACKEND", "nccl") == "nccl":
        assert rccl_n == world, "the flag did not travel over the library's RCCL communicator"
    res = {"metric": "instance-step!() calls/sec, batched dense BFGS n=256 fp64 (config 5)",
           "value": round(inst_steps / el, 1), "unit": "instance-step!() calls/s",
           "ms_per_step": round(1e3 * el / steps, 4), "dtype": "f64", "steps": steps, "warmup": warmup,
           "per_rank_instance_steps_per_s": per_rank,
           "config": {"workload": f"batched BFGS, {B} instances/GPU x n={n}, chained Rosenbrock, fp64 (BASELINE configs[4])",
                      "instances_per_gpu": B, "instances_total": B * world, "active_at_end": batch.count_active(),
                      "rccl_world_size": rccl_n, "polls": polls,
                      "parallelism": (f"instances sharded by rank (block partition), world size {world}; the only collective is "
                                      f"the convergence flag: {'dzo_bfgs_batch_all_done (RCCL behind the C ABI)' if comm is not None else flag.transport}, "
                                      f"{polls} polls in the timed region"),
                      "device": info["name"]},
           "roofline": {"bound": "hbm", "kernel": "batch_step_kernel<double, RP> (HIP-event name bfgs_batch_step)",
                        "achieved": None if ach is None else round(ach, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None if ach is None else round(ach / HBM_PEAK_GBS, 4),
                        "traffic": _pmc_secondary("batched", ["batch_step_kernel"])[0] if (B == 1024 and n == 256) else None,
                        "traffic_source": _pmc_secondary("batched", ["batch_step_kernel"])[1] if (B == 1024 and n == 256) else None,
                        "note": "rank 0's kernel; traffic = one launch of `poll` synchronous steps of the whole shard; 1.5 n^2 T per accepted instance-step (lower triangle of H: read twice, written once); line searches included in the time"},
           "kernels": kern}
    batch.close()
    return res


def launch_check(args):
    """--workload launch_check: the multi-rank plumbing (self-launch, rendezvous, rank environment, one collective)
    WITHOUT a GPU -- what tests/test_bench_launch.py runs on CPU.  Prints the ranks the group saw."""
    world, rank, local = _dist_setup(args.gpus, use_gpu=False)
    seen = [None] * world
    if world > 1:
        import torch.distributed as dist
        dist.all_gather_object(seen, (rank, local, os.environ.get("MASTER_ADDR"), os.environ.get("BENCH_SELF_LAUNCHED", "0")))
        dist.destroy_process_group()
    else:
        seen = [(rank, local, os.environ.get("MASTER_ADDR"), os.environ.get("BENCH_SELF_LAUNCHED", "0"))]
    if rank == 0:
        print(json.dumps({"metric": "launch_check", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ranks": seen}))
    return 0


def secondary_workload(args, inproc=None):
    """Configs 2, 4, 5 of BASELINE.json: same JSON shape, their own metric strings.  `inproc` = (dzo, sharding, info):
    called from the default line at N = 1 on the already initialised library (no CPU baseline, the dict is returned
    instead of printed)."""
    import importlib
    cpu_line = None
    if inproc is not None:
        dzo, sharding, info = inproc
        world, rank = 1, 0
    else:
        if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
            # first of all: the CPU baseline (worker processes for the batched one), while no GPU context exists yet
            cn = {"bfgs_dense": 4096, "bfgs_batched": 256, "lbfgs_lse_f32": 1_000_000}.get(args.workload, args.n) if args.n == 10_000_000 else args.n
            cpu_line = cpu_baseline_secondary(args.workload, cn, m=(10 if args.m == 20 else args.m),
                                              A=quadratic_matrix(cn) if args.workload == "bfgs_dense" else None)
        import torch
        world, rank, local = _dist_setup(args.gpus)
        from dzo_loader import dzo
        dzo.init(local)
        sharding = importlib.import_module("dzoptimization_jl_amd.sharding")
        info = dzo.device_info()
    out = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "data": "synthetic"}
    if args.workload == "bfgs_dense":
        n = 4096 if args.n == 10_000_000 else args.n
        A = quadratic_matrix(n)
        prob = dzo.Problem(dzo.QUADRATIC, n, A=A)
        x0 = pcg32_uniform(n, 4) - 0.5
        opt = dzo.BFGSOptimizer(prob, None, dzo.DeviceArray.from_host(x0), 1.0)
        for _ in range(args.warmup):
            opt.step()
        dzo.synchronize()
        _barrier(world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            opt.step()
        dzo.synchronize(); _barrier(world)
        el = sharding.max_over_ranks(time.perf_counter() - t0)
        # kernel-level HIP events: a second, untimed stretch of the same loop (a step is ten launches of 3-30 us: two event
        # records per launch put ~10 us between every pair of kernels, 250 against 160 us per step in a rocprofv3 trace)
        dzo.profile_reset(); dzo.profile_enable(True)
        for _ in range(min(args.steps, 50)):
            opt.step()
        dzo.synchronize()
        dzo.profile_enable(False)
        tab = dzo.profile_table()
        # isolated update + next direction (K8 + K9): 3 n^2 T algorithmic bytes
        rng = np.random.default_rng(0)
        Hd = dzo.DeviceArray.from_host(np.eye(n))
        g = dzo.DeviceArray.from_host(rng.standard_normal(n))
        scratch, dnext = dzo.DeviceArray(n), dzo.DeviceArray(n)
        times = []
        for it in range(12):
            d, y = rng.standard_normal(n), rng.standard_normal(n)
            lam = 0.1 if d @ y > 0 else -0.1
            dd, yd = dzo.DeviceArray.from_host(d), dzo.DeviceArray.from_host(y)
            dzo.synchronize()
            t1 = time.perf_counter()
            dzo.update_inverse_hessian_(Hd, lam, dd, yd, scratch, g, dnext)
            times.append(time.perf_counter() - t1)
        upd = float(np.median(times[2:]))
        # the MFMA form of the rank-2 update (no fused direction), same inputs
        times_m = []
        Hm = dzo.DeviceArray.from_host(np.eye(n))
        dzo.profile_reset(); dzo.profile_enable(2)
        for it in range(12):
            d, y = rng.standard_normal(n), rng.standard_normal(n)
            lam = 0.1 if d @ y > 0 else -0.1
            dd, yd = dzo.DeviceArray.from_host(d), dzo.DeviceArray.from_host(y)
            dzo.synchronize()
            t1 = time.perf_counter()
            dzo.update_inverse_hessian_mfma_(Hm, lam, dd, yd, scratch)
            times_m.append(time.perf_counter() - t1)
        dzo.profile_enable(0)
        tabm = dzo.profile_table()
        mfma_us = 1e3 * tabm["bfgs_update_mfma"][1] / tabm["bfgs_update_mfma"][0] if "bfgs_update_mfma" in tabm else None
        kern = {k: {"launches": v[0], "avg_us": round(1e3 * v[1] / v[0], 2)} for k, v in tab.items()}
        tri = "bfgs_tri_reduce" in tab          # step! keeps the lower triangle of H only (H >= 128 MiB)
        kbytes = {"bfgs_symv": n * n * 8 // 2, "bfgs_update": n * n * 8} if tri else {"bfgs_symv": n * n * 8, "bfgs_update": 2 * n * n * 8}
        for k, b in kbytes.items():
            if k in kern:
                kern[k]["algorithmic_GBps"] = round(b / (kern[k]["avg_us"] * 1e-6) / 1e9, 1)
        dom = "bfgs_update"
        # measured ceilings: the same read-only streaming kernel over a buffer the size of what a pass over H reads
        # (Infinity-Cache resident) and over 4 GiB (HBM)
        hbytes = n * n * 8 // (2 if tri else 1)
        ceil_llc = dzo.calibrate_read_bandwidth(hbytes, 30)
        ceil_hbm = dzo.calibrate_read_bandwidth(4 << 30, 3)
        out.update({"metric": "step!() calls/sec and achieved HBM GB/s, dense BFGS n=4096 fp64 (config 2)",
                    "value": round(world * args.steps / el, 3), "unit": "step!() calls/s",
                    "ms_per_step": round(1e3 * el / args.steps, 4), "dtype": "f64",
                    "config": {"workload": f"dense BFGS on convex quadratic 1/2 x'Ax, n={n}, fp64 (BASELINE configs[1])",
                               "objective_evals_per_step": round(opt.objective_evaluations / max(opt.iteration_count, 1), 2),
                               "line_searches": ("device-driven (one host wait per step)" if os.environ.get("DZO_TUNE_BFGS_DEV_SEARCH", "1") != "0"
                                                 else "host-driven rounds"),
                               "kernel_events": "separate untimed stretch of the same loop (the timed region carries no event records)",
                               "device": info["name"]},
                    "update_plus_direction": {"host_wall_us": round(upd * 1e6, 1), "algorithmic_bytes": 3 * n * n * 8,
                                              "algorithmic_GBps": round(3 * n * n * 8 / upd / 1e9, 1)},
                    "mfma_update_variant": {"kernel_us": None if mfma_us is None else round(mfma_us, 2),
                                            "algorithmic_GBps": None if mfma_us is None else round(2 * n * n * 8 / (mfma_us * 1e-6) / 1e9, 1),
                                            "note": "v_mfma_f64_16x16x4_f64 rank-2 update of H only (2 n^2 T); compare with kernels.bfgs_update, which also produces the next direction"},
                    "roofline": {"bound": "hbm", "kernel": dom, "achieved": kern.get(dom, {}).get("algorithmic_GBps"),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(kern.get(dom, {}).get("algorithmic_GBps", 0) / HBM_PEAK_GBS, 4),
                                 "traffic": _pmc_secondary("dense", ["tri_pass_kernel<double, true>"])[0] if (tri and n == 4096) else None,
                                 "traffic_source": _pmc_secondary("dense", ["tri_pass_kernel<double, true>"])[1] if (tri and n == 4096) else None,
                                 "measured_read_ceiling_GBps": {"resident_buffer_of_this_size": round(ceil_llc, 1), "hbm_4GiB": round(ceil_hbm, 1)},
                                 "frac_of_measured_resident_ceiling": round(kern.get(dom, {}).get("algorithmic_GBps", 0) / max(ceil_llc, 1e-9), 4),
                                 "update_plus_direction_kernel_us": round(sum(kern[k]["avg_us"] * (2 if k == "bfgs_tri_reduce" else 1)
                                                                              for k in ("bfgs_symv", "bfgs_update", "bfgs_tri_reduce") if k in kern), 2),
                                 "note": ("step! reads and writes the lower triangle of H only (1.5 n^2 T per update + direction); " if tri else "")
                                         + "H = 128 MiB fits the 256 MiB Infinity Cache: these are on-die rates, the HBM peak is "
                                           "quoted only because the contract asks for it"},
                    "kernels": kern})
    elif args.workload == "bfgs_batched":
        comm = _make_comm(dzo, world)                              # RCCL communicator behind the C ABI when N > 1 (hard requirement)
        out.update(batched_leg(dzo, sharding, args, world, rank, comm, info, args.steps, args.warmup))
        if world == 1 and args.n == 10_000_000 and inproc is None:
            # (the workload's own line only, not the default line's `secondary` object) the other instantiations of the batched kernel at their top sizes (same shard of 1024 instances): RP = 2
            # in its two forms (n = 384 narrow, n = 512 wide) -- rows of the same table, never `value`
            import copy
            rows = {}
            for nn in (384, 512):
                a2 = copy.copy(args); a2.n = nn
                r2 = batched_leg(dzo, sharding, a2, world, rank, comm, info, max(args.steps // 2, 10), args.warmup)
                rows[f"n{nn}"] = {"value": r2["value"], "unit": r2["unit"], "ms_per_step": r2["ms_per_step"],
                                  "roofline_frac": r2["roofline"]["frac"], "kernels": r2["kernels"]}
            out["other_sizes"] = rows
    elif args.workload == "adgd":
        # AdGDOptimizer (src/DZOptimization.jl:179-312) on the headline objective
        n = args.n
        x0 = rosenbrock_chain_x0(n, seed=5 + rank)
        opt = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 1.0)
        for _ in range(10 + args.warmup):
            opt.step()
        f_start = opt.current_objective_value
        fused0, rej0 = opt.fused_steps, opt.fused_rejections
        _barrier(world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            opt.step()
        dzo.synchronize(); _barrier(world)
        el = sharding.max_over_ranks(time.perf_counter() - t0)
        fused1, rej1 = opt.fused_steps, opt.fused_rejections
        # kernel-level HIP events (two event records per launch cost a few % of this 0.1-ms step): a second,
        # untimed stretch of the same loop
        dzo.profile_reset(); dzo.profile_enable(2)
        for _ in range(min(args.steps, 100)):
            opt.step()
        dzo.synchronize()
        dzo.profile_enable(False)
        tab = dzo.profile_table()
        kern = {kk: {"launches": v[0], "avg_us": round(1e3 * v[1] / v[0], 2)} for kk, v in tab.items()}
        us = 1e3 * tab["adgd_fused_step"][1] / tab["adgd_fused_step"][0] if "adgd_fused_step" in tab else None
        ach = None if us is None else 2 * n * 8 / (us * 1e-6) / 1e9
        out.update({"metric": "step!() calls/sec, AdGD n=10^7 fp64 (SURVEY 8(f) rank 1)",
                    "value": round(world * args.steps / el, 2), "unit": "step!() calls/s",
                    "ms_per_step": round(1e3 * el / args.steps, 4), "dtype": "f64",
                    "config": {"workload": f"AdGD on N-D chained Rosenbrock, n={n}, fp64", "f_start": f_start,
                               "f_end": opt.current_objective_value, "stuck": opt.is_stuck,
                               "fused_steps": fused1 - fused0, "steps_after_a_rejected_trial": rej1 - rej0,
                               "pipelined_passes": opt.pipelined_passes, "pipeline_discards": opt.pipeline_discards,
                               "kernel_events": "separate untimed stretch of the same loop (two event records per launch cost a few % of a 0.1-ms step)",
                               "device": info["name"]},
                    "roofline": {"bound": "hbm", "kernel": "adgd_fused_step", "achieved": None if ach is None else round(ach, 1),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None if ach is None else round(ach / HBM_PEAK_GBS, 4),
                                 "traffic": _pmc_secondary("adgd", ["adgd_fused_rosen_kernel"])[0] if n == 10_000_000 else None,
                                 "traffic_source": _pmc_secondary("adgd", ["adgd_fused_rosen_kernel"])[1] if n == 10_000_000 else None,
                                 "avg_launch_us": None if us is None else round(us, 2),
                                 "note": "2 n T per pass: reads x, writes the trial point into the next of three buffers; g_old is recomputed in registers from x_old (3-point stencil), g_new is not written (gradient arrays, delta_point / delta_gradient are formed on demand; the deltas enter only the two norms)"},
                    "kernels": kern})
    else:  # lbfgs_lse_f32 (config 4)
        n = 1_000_000 if args.n == 10_000_000 else args.n
        m = 10 if args.m == 20 else args.m
        c = (pcg32_uniform(n, 6) - 0.5).astype(np.float32)
        prob = dzo.Problem(dzo.LSE, n, np.float32, c=c, lam=1e-2)
        opt = dzo.LBFGSOptimizer(None, prob, None, dzo.DeviceArray.from_host(np.zeros(n, np.float32)), 1.0, m)
        # (a) the optimizer run itself: this objective is strongly convex and almost quadratic, fp32
        # L-BFGS converges (is_stuck) within a handful of steps, so it is reported, not timed
        f0 = opt.current_objective_value
        run_steps = 0
        while run_steps < 200 and not opt.is_stuck:
            opt.step()
            run_steps += 1
        # (b) K1 in isolation on the frozen synthetic state of SURVEY.md 8(d): g, s_i, y_i = u - 1/2
        # (seeds 10, 100+i, 200+i), y_i += s_i; fp32, k = m
        g = (pcg32_uniform(n, 10) - 0.5).astype(np.float32)
        S = np.empty((m, n), np.float32)
        Y = np.empty((m, n), np.float32)
        for i in range(m):
            sv = pcg32_uniform(n, 100 + i) - 0.5
            S[i] = sv.astype(np.float32)
            Y[i] = (pcg32_uniform(n, 200 + i) - 0.5 + sv).astype(np.float32)
        xz = dzo.DeviceArray.zeros(n, np.float32)
        gd = dzo.DeviceArray.from_host(g)
        fro = dzo.LBFGSOptimizer(None, lambda x_: 0.0, lambda g_, x_: None, xz, 0.0, gd, 1.0, m)
        fro.set_history(S, Y)
        for _ in range(3 + args.warmup):
            fro.compute_step_direction()
        dzo.synchronize()
        _barrier(world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fro.compute_step_direction(sync=False)               # enqueue only; the stream keeps the GPU busy
        dzo.synchronize(); _barrier(world)
        el = sharding.max_over_ranks(time.perf_counter() - t0)
        # kernel-level HIP events (two event records per launch are a measurable share of a 60-us direction): a second,
        # untimed stretch of the same loop
        dzo.profile_reset(); dzo.profile_enable(True)
        for _ in range(args.steps):
            fro.compute_step_direction(sync=False)
        dzo.synchronize()
        dzo.profile_enable(False)
        tab = dzo.profile_table()
        k = m
        kern = {kk: {"launches": v[0], "avg_us": round(1e3 * v[1] / v[0], 2)} for kk, v in tab.items()}
        stage = ("lbfgs_gram_pass", "lbfgs_gram_reduce", "lbfgs_gram_finish", "lbfgs_gram_reduce_finish", "lbfgs_combine")
        tl = sum(1e3 * tab[x][1] for x in stage if x in tab) / max(tab.get("lbfgs_combine", (1, 0))[0], 1)
        tl = max(tl, 1e-9)
        wall_us = 1e6 * el / args.steps
        out.update({"metric": "two-loop recursions/sec, L-BFGS m=10 n=10^6 fp32 (config 4, K1 on frozen state)",
                    "value": round(world * args.steps / el, 2), "unit": "compute_lbfgs_step_direction! calls/s",
                    "ms_per_step": round(1e3 * el / args.steps, 4), "dtype": "f32",
                    "config": {"workload": f"two-loop recursion, m={m}, n={n}, fp32, frozen synthetic (s, y) (BASELINE configs[3])",
                               "lse_run": {"steps_until_stuck": run_steps, "f0": f0, "f_end": opt.current_objective_value,
                                           "stuck": opt.is_stuck},
                               "device": info["name"]},
                    "roofline": {"bound": "hbm", "kernel": "two_loop (gram_pass_lanes_kernel + gram_reduce_finish_kernel + combine_kernel)",
                                 "achieved": round((4 * k + 2) * n * 4 / (tl * 1e-6) / 1e9, 1),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((4 * k + 2) * n * 4 / (tl * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                 "traffic": _pmc_secondary("lse", ["gram_pass_lanes_kernel", "combine_kernel"])[0] if (n == 1_000_000 and m == 10) else None,
                                 "traffic_source": _pmc_secondary("lse", ["gram_pass_lanes_kernel", "combine_kernel"])[1] if (n == 1_000_000 and m == 10) else None,
                                 "kernel_sum_us": round(tl, 2),
                                 "wall_us_per_direction": round(wall_us, 2),
                                 "wall_frac": round((4 * k + 2) * n * 4 / (wall_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                 "kernel_events": "separate untimed stretch of the same loop",
                                 "note": "launch-latency-bound at this size: the launches of one direction move 168 MB"},
                    "kernels": kern})
    if inproc is not None:
        return out
    if rank == 0:
        if cpu_line is not None:
            out["cpu_baseline"] = cpu_line
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


# the other BASELINE configs inside the default line (N = 1): (workload, timed steps, warm-up) -- the settings of the
# tracked profiles/rNN_bench_<workload>.json lines, so that the driver's own run carries them too
SECONDARY_IN_DEFAULT_LINE = (("bfgs_dense", 50, 5), ("lbfgs_lse_f32", 50, 5), ("adgd", 200, 10), ("bfgs_batched", 50, 5))


def secondary_in_line(args, dzo, sharding, info):
    """The default N = 1 line's `secondary` object: configs 2, 4, 5 and AdGD measured in this process AFTER the headline's
    timed region, two-pass leg and CPU baseline (nothing of it is part of `value`).  Each entry is the workload's own
    line (python3 bench.py --workload W) without its CPU baseline; a failure is reported, never fatal to the line."""
    import copy
    res = {}
    for w, steps, warm in SECONDARY_IN_DEFAULT_LINE:
        a = copy.copy(args)
        a.workload, a.steps, a.warmup, a.gpus = w, steps, warm, 1
        t0 = time.perf_counter()
        try:
            o = secondary_workload(a, inproc=(dzo, sharding, info))
            keep = {k: o[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "kernels") if k in o}
            for k in ("update_plus_direction",):
                if k in o:
                    keep[k] = o[k]
            keep["leg_wall_s"] = round(time.perf_counter() - t0, 2)
            res[w] = keep
        except Exception as e:                                  # noqa: BLE001 -- reported in the line
            res[w] = {"error": f"{type(e).__name__}: {e}"}
            try:
                dzo.synchronize()
            except Exception:                                   # noqa: BLE001
                pass
    return res


def _two_pass_leg(dzo, n, m, esize, args):
    os.environ["DZO_TUNE_SINGLE_PASS"] = "0"             # read when an optimizer is created
    try:
        x2 = dzo.DeviceArray.from_host(rosenbrock_chain_x0(n, seed=5))
        opt2 = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, x2, 1.0, m)
    finally:
        del os.environ["DZO_TUNE_SINGLE_PASS"]
    for _ in range(m + args.warmup):
        opt2.step()
    steps2 = max(20, min(args.steps, 50))
    dzo.synchronize()
    dzo.profile_reset(); dzo.profile_enable(2)          # (every kernel: the two-loop's total needs the reduce and finish kernels too)
    t0 = time.perf_counter()
    for _ in range(steps2):
        opt2.step()
    dzo.synchronize()
    el = time.perf_counter() - t0
    dzo.profile_enable(False)
    tab = dzo.profile_table()
    k = opt2.history_count
    assert opt2.single_pass_steps == 0
    out = {"steps": steps2, "ms_per_step": round(1e3 * el / steps2, 4), "step_calls_per_s": round(steps2 / el, 2),
           "note": "same workload, optimizer created with DZO_TUNE_SINGLE_PASS=0 in this process; HIP events around EVERY kernel on the "
                   "launching stream (ms_per_step includes ~8 us per bracket); algorithmic bytes: gram (2k+1) n T, combine (2k+1) n T, two-loop (4k+2) n T"}

    def leg(name, nbytes):
        if name not in tab or not tab[name][0]:
            return None
        us = 1e3 * tab[name][1] / tab[name][0]
        return {"launches": tab[name][0], "avg_us": round(us, 2), "algorithmic_bytes": nbytes,
                "achieved": round(nbytes / (us * 1e-6) / 1e9, 1), "frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    out["gram"] = leg("lbfgs_gram_pass", (2 * k + 1) * n * esize)
    out["combine"] = leg("lbfgs_combine", (2 * k + 1) * n * esize)
    names = ("lbfgs_gram_pass", "lbfgs_gram_reduce", "lbfgs_gram_finish", "lbfgs_gram_reduce_finish", "lbfgs_combine")
    if all(x in tab for x in ("lbfgs_gram_pass", "lbfgs_combine")):
        # per direction: every launch of the four kernels in the timed region / directions computed
        us = sum(1e3 * tab[x][1] for x in names if x in tab) / max(tab["lbfgs_combine"][0], 1)
        nb = (4 * k + 2) * n * esize
        out["two_loop"] = {"directions": tab["lbfgs_combine"][0], "avg_us": round(us, 1), "algorithmic_bytes": nb,
                           "achieved": round(nb / (us * 1e-6) / 1e9, 1), "frac": round(nb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    opt2.close()
    return out


def julia_host_probe(n):
    """cpu_baseline.julia (SURVEY 8(c) last row): is there a `julia` on this box?  If so, the build's own
    host module (DZOptimizationAMD.jl) is driven through tools/julia_host_check.jl -- in a child process started after this
    process has finished its own GPU work -- and its config-3 step rate is reported; nothing of /root/reference is involved.
    If not (the build container and every GPU box seen so far), the line says so."""
    import shutil
    import subprocess
    import tempfile
    julia = shutil.which("julia")
    if julia is None:
        return {"julia": "absent on this box", "kind_note": "cpu_baseline.kind stays \"port\": no Julia runtime to time the reference's own LinearAlgebra calls with"}
    try:
        with tempfile.TemporaryDirectory() as tmp:
            x0 = os.path.join(tmp, "x0.txt")
            with open(x0, "w") as f:
                f.write("\n".join(repr(float(v)) for v in rosenbrock_chain_x0(64, seed=5)) + "\n")
            r = subprocess.run([julia, os.path.join(ROOT, "tools", "julia_host_check.jl"), x0, "5", "10", str(n)],
                               capture_output=True, text=True, timeout=600)
        rate = [float(ln.split()[1]) for ln in r.stdout.splitlines() if ln.startswith("rate ")]
        return {"julia": julia, "returncode": r.returncode, "host_module_step_calls_per_s": rate[0] if rate else None,
                "stderr_tail": r.stderr[-300:] if r.returncode else ""}
    except Exception as e:                                      # noqa: BLE001 -- reported in the line
        return {"julia": julia, "error": f"{type(e).__name__}: {e}"}


def _lbfgs_variant_leg(dzo, n, m, esize, args, kind):
    """Config 3 as the callers beside the headline see it, measured after the timed region:
    `callbacks`  the reference's real API -- constraint / objective / gradient supplied as C function pointers
                 (dzo_problem_*_cb: dzo_problem_eval / dzo_problem_grad behind the callback signature, what the Julia host's
                 closures do), i.e. the GENERAL two-pass step: Gram pass + reduce + finish + combine, trial, accept, delta kernels;
    `decorated`  the built-in objective with L2 regularisation + box gradient mask + box projection (legacy :219-296) riding
                 on the point pass (its DEC instantiation);
    `ragged`     n + 1 (not a multiple of the 16-byte vector), phantom-padded point ring;
    `quadratic_chain`  the point pass's second objective: the chained quadratic (lambda = 1e-4, same start point);
    `lse`        config 4's objective and size as whole step!() calls (log-sum-exp, n = 10^6, m = 10, fp32, x0 = 0): a trial pass and a
                 dots pass over the ring of points per step; `lse_two_pass` the same with DZO_TUNE_LSE_POINTS=0 (the general path).
    Rate from a stretch without any event record; the per-kernel table from a second, untimed stretch."""
    nn = n + 1 if kind == "ragged" else n
    decor = dict(l2=1e-3, box_gradient=(-1.15, 0.95), box_constraint=(-1.15, 0.95)) if kind == "decorated" else {}
    lse = kind in ("lse", "lse_two_pass")
    if lse:
        nn, m, esize = (1_000_000 if n == 10_000_000 else n), (10 if m == 20 else m), 4
        cvec = (pcg32_uniform(nn, 6) - 0.5).astype(np.float32)
        prob = dzo.Problem(dzo.LSE, nn, np.float32, c=cvec, lam=1e-2)
        x = dzo.DeviceArray.from_host(np.zeros(nn, np.float32))
        if kind == "lse_two_pass":
            os.environ["DZO_TUNE_LSE_POINTS"] = "0"
        try:
            opt = dzo.LBFGSOptimizer(None, prob, None, x, 1.0, m)
        finally:
            os.environ.pop("DZO_TUNE_LSE_POINTS", None)
    else:
        prob = dzo.Problem(dzo.QUADRATIC_CHAIN, nn, lam=1e-4) if kind == "quadratic_chain" else dzo.Problem(dzo.ROSENBROCK_CHAIN, nn, **decor)
        x = dzo.DeviceArray.from_host(rosenbrock_chain_x0(nn, seed=5))
        if kind == "callbacks":
            opt = dzo.LBFGSOptimizer(None, prob.native_callbacks(), None, x, 1.0, m)
        else:
            opt = dzo.LBFGSOptimizer(None, prob, None, x, 1.0, m)
    # (the log-sum-exp problem of config 4 is dominated by its quadratic term: L-BFGS is at the resolution of fp32 after a handful
    # of steps and then stuck -- its leg times the steps it takes from x0 = 0 after ONE warm-up step, at most 8)
    for _ in range(1 if lse else m + args.warmup):
        opt.step()
    steps = 8 if lse else max(20, min(args.steps, 50))
    dzo.synchronize()
    trials = 0
    done = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.step()
        if opt.is_stuck:
            break
        trials += opt.last_trials
        done += 1
    dzo.synchronize()
    el = time.perf_counter() - t0
    steps = max(done, 1)
    dzo.profile_reset(); dzo.profile_enable(2)
    for _ in range(2 if lse else 10):
        opt.step()
    dzo.synchronize()
    dzo.profile_enable(False)
    tab = dzo.profile_table()
    k = opt.history_count
    kern = {nm: {"launches": c, "avg_us": round(1e3 * ms / c, 2)} for nm, (c, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1]) if c}
    out = {"metric": "step!() calls/s", "value": round(steps / el, 2), "unit": "step!() calls/s", "steps": steps, "ms_per_step": round(1e3 * el / steps, 4),
           "dtype": "f32" if lse else "f64",
           "config": {"workload": f"L-BFGS m={m} on " + ("log-sum-exp" if lse else "N-D chained quadratic" if kind == "quadratic_chain" else "N-D chained Rosenbrock") + f", n={nn}, {'fp32' if lse else 'fp64'}: {kind}", "n": nn, "m": m,
                                      "history_layout": {0: "slabs", 1: "tiles of pairs", 2: "tiles of points"}[opt.ring_layout],
                                      "objective_evals_per_step": round(trials / steps, 3), "any_stuck": bool(opt.is_stuck),
                                      "decorators": decor or None},
           "kernels": kern}
    dom = "lbfgs_single_pass" if "lbfgs_single_pass" in tab else "lbfgs_gram_pass"
    if dom in kern:
        us = kern[dom]["avg_us"]
        nb = _kernel_bytes(dom, nn, k, esize, opt.ring_layout, opt.pass_recomputes_gradients)
        if lse and dom == "lbfgs_single_pass":
            nb = (k + 3) * nn * esize                    # the trial pass: k + 1 points and c read, the trial point written
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(nb / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(nb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": nb, "avg_launch_us": us}
        names = ("lbfgs_gram_pass", "lbfgs_gram_reduce", "lbfgs_gram_finish", "lbfgs_gram_reduce_finish", "lbfgs_combine")
        if "lbfgs_combine" in tab and "lbfgs_gram_pass" in tab:
            us2 = sum(1e3 * tab[x][1] for x in names if x in tab) / max(tab["lbfgs_combine"][0], 1)
            nb2 = (4 * k + 2) * nn * esize
            out["roofline"]["two_loop"] = {"avg_us": round(us2, 1), "algorithmic_bytes": nb2, "achieved": round(nb2 / (us2 * 1e-6) / 1e9, 1),
                                           "frac": round(nb2 / (us2 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    opt.close()
    return out


# ------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dim", dest="n", type=int, default=10_000_000, help="problem dimension n")
    ap.add_argument("--history", dest="m", type=int, default=20, help="L-BFGS history length m")
    ap.add_argument("--mode", choices=["gram", "chain"], default="gram")
    ap.add_argument("--poll", type=int, default=10, help="all-reduce the convergence flag every POLL steps (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-pass", action="store_true", help="skip the two-pass (Gram + combine) roofline leg at N = 1")
    ap.add_argument("--cpu-dim", dest="cpu_n", type=int, default=None, help="n of the CPU sample (default: same n)")
    ap.add_argument("--cpu-steps", type=int, default=20, help="timed CPU steps on all cores")
    ap.add_argument("--cpu-steps-single", type=int, default=8, help="timed CPU steps on one core (the whole CPU leg: about 10 s)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-kernel HIP events in the timed region")
    ap.add_argument("--kernel-events", type=int, default=1, choices=[1, 2],
                    help="HIP events in the timed region: 1 = the two-loop (roofline) kernels only, 2 = every kernel")
    ap.add_argument("--event-sample", type=int, default=1,
                    help="lbfgs: bracket the roofline kernel's launches with HIP events in every Nth step of the timed region only (a "
                         "bracket costs about 8 us of stream time; measured in round 4, five interleaved runs on one box: 2531 step!()/s with every step "
                         "bracketed, 2565 with every 4th, i.e. +1.4 %; the default stays every step -- every launch of the timed region is then in the average)")
    ap.add_argument("--workload", default="lbfgs", choices=["lbfgs", "bfgs_dense", "bfgs_batched", "lbfgs_lse_f32", "adgd", "launch_check"],
                    help="lbfgs = BASELINE configs[2] (the headline; default). The others are the remaining "
                         "BASELINE configs, reported as secondary lines.")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default line at N = 1: skip the `secondary` object (configs 2, 4, 5 and AdGD measured after the headline)")
    ap.add_argument("--decorators", default="", help="lbfgs (sweep rows only, never the default line): decorators of legacy/DZOptimization.jl:219-296 "
                                                      "on the objective, e.g. 'l2=0.001,box=-1.15:0.95' (box = gradient mask + projection)")
    ap.add_argument("--variant", default="", choices=["", "callbacks", "decorated", "ragged", "quadratic_chain", "lse", "lse_two_pass"],
                    help="lbfgs: print only the line of one variant leg of the `secondary` object (dev)")
    ap.add_argument("--batch", type=int, default=1024, help="bfgs_batched: instances per GPU (config 5 shard)")
    ap.add_argument("--batched-steps", type=int, default=0,
                    help="N > 1: timed synchronous steps of the config-5 `batched` object (default: max(--steps, 100))")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the driver's own multi-GPU form as a child, before anything here touches the GPU
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    if args.workload == "launch_check":
        return launch_check(args)
    if args.workload != "lbfgs":
        return secondary_workload(args)

    import torch  # first: loads the ROCm runtime that libdzo_hip.so then shares
    world, rank, local = _dist_setup(args.gpus)
    from dzo_loader import dzo
    dzo.init(local)
    info = dzo.device_info()

    n, m, esize = args.n, args.m, 8
    if args.variant:
        print(json.dumps(_lbfgs_variant_leg(dzo, n, m, esize, args, args.variant)))
        return
    x0 = rosenbrock_chain_x0(n, seed=5 + rank)           # each rank optimises its own instance
    decor = {}
    for item in filter(None, args.decorators.split(",")):
        key, val = item.split("=")
        if key == "l2":
            decor["l2"] = float(val)
        elif key == "box":
            lo, hi = (float(v) for v in val.split(":"))
            decor["box_gradient"] = (lo, hi); decor["box_constraint"] = (lo, hi)
        else:
            raise SystemExit(f"--decorators: unknown item {item!r}")
    prob = dzo.Problem(dzo.ROSENBROCK_CHAIN, n, **decor)
    x_dev = dzo.DeviceArray.from_host(x0)
    del x0
    opt = dzo.LBFGSOptimizer(None, prob, None, x_dev, 1.0, m)
    opt.set_two_loop_mode(dzo.TWOLOOP_GRAM if args.mode == "gram" else dzo.TWOLOOP_CHAIN)

    # setup (untimed, not part of warmup): fill the (s, y) ring so that k = m in every timed step
    for _ in range(m):
        opt.step()
    for _ in range(args.warmup):
        opt.step()
    f_start = opt.current_objective_value

    import importlib
    sharding = importlib.import_module("dzoptimization_jl_amd.sharding")
    comm = _make_comm(dzo, world)                        # the library's own RCCL communicator (C ABI) when N > 1
    flag = sharding.ConvergenceFlag(poll=args.poll, comm=comm)   # all-reduce(MIN) of one int32
    if world > 1:
        import torch.distributed as dist

    dzo.profile_reset()
    ev_level = 0 if args.no_kernel_events else args.kernel_events
    ev_every = max(1, args.event_sample)
    dzo.profile_enable(ev_level)
    trials = 0
    passes0, retries0 = opt.single_pass_steps, opt.single_pass_retries
    _barrier(world)
    t0 = time.perf_counter()
    for s in range(args.steps):
        if ev_level and ev_every > 1:
            dzo.profile_enable(ev_level if s % ev_every == 0 else 0)    # HIP events around the kernels of every ev_every-th step
        opt.step()
        trials += opt.last_trials
        flag.update(opt.is_stuck)                       # global "everyone converged" flag
    dzo.synchronize()
    el_local = time.perf_counter() - t0                 # this rank's own time, before it waits for the others
    _barrier(world)
    elapsed = time.perf_counter() - t0
    dzo.profile_enable(False)

    elapsed = sharding.max_over_ranks(elapsed)
    # N > 1: every rank's own rate and trial count (the replicas start from different points, so their numbers of
    # rejected first trials -- a second pass each -- differ: a slower rank shows here, with its reason)
    per_rank = None
    if world > 1:
        per_rank = [{"rank": r,
                     "step_calls_per_s": round(sharding.sum_over_ranks(args.steps / el_local if r == rank else 0.0), 2),
                     "objective_evals": int(sharding.sum_over_ranks(float(trials) if r == rank else 0.0))} for r in range(world)]
    any_stuck = sharding.max_over_ranks(1.0 if opt.is_stuck else 0.0) > 0

    table = dzo.profile_table()
    # N > 1: the quantity north_star shards -- config 5, B independent dense-BFGS instances per GPU, the convergence
    # flag over the same RCCL communicator (every rank takes part)
    batched = None
    if world > 1:
        bsteps = args.batched_steps or max(args.steps, 100)
        batched = batched_leg(dzo, sharding, args, world, rank, comm, info, bsteps, args.warmup)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # The two-loop recursion proper ((4k+2) n T per direction): the same workload on the
    # two-pass path -- Gram pass + reduce + finish + combine per step, what callbacks / any other
    # objective run -- measured in this process on a second optimizer created with the single-pass step
    # switched off.  N = 1 only (a reported roofline leg, not part of `value`).
    two_pass = None
    if world == 1 and opt.single_pass_steps > 0 and not args.no_two_pass and args.mode == "gram":
        two_pass = _two_pass_leg(dzo, n, m, esize, args)

    k = opt.history_count
    kernels = {}
    for name, (launches, ms) in sorted(table.items(), key=lambda kv: -kv[1][1]):
        avg_us = 1e3 * ms / launches
        b = _kernel_bytes(name, n, k, esize, opt.ring_layout, opt.pass_recomputes_gradients)
        kernels[name] = {"launches": launches, "avg_us": round(avg_us, 2),
                         "algorithmic_GBps": None if b is None else round(b / (avg_us * 1e-6) / 1e9, 1)}
    roofline = None
    if kernels:
        dom = next((nm for nm in kernels if nm in ("lbfgs_single_pass", "lbfgs_gram_pass", "lbfgs_combine", "lbfgs_chain_link")), None)
        if dom:
            ach = kernels[dom]["algorithmic_GBps"]
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc) and n == 10_000_000 and k == 20:   # the PMC passes were taken at the headline size
                try:
                    traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            symbol = {"lbfgs_single_pass": (f"lbfgs_point_pass_kernel<double, 20, false, {opt.pass_register_sets}>" if opt.ring_layout == 2 and m > 16 else
                                            "lbfgs_point_pass_kernel" if opt.ring_layout == 2 else "lbfgs_single_pass_kernel"),
                      "lbfgs_gram_pass": "gram_pass_lanes_kernel", "lbfgs_combine": "combine_kernel",
                      "lbfgs_chain_link": "chain_link_kernel"}[dom]
            roofline = {"bound": "hbm", "kernel": symbol, "hip_event_name": dom,
                        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "algorithmic_bytes_per_launch": _kernel_bytes(dom, n, k, esize, opt.ring_layout, opt.pass_recomputes_gradients),
                        "avg_launch_us": kernels[dom]["avg_us"]}
            roofline["traffic_source"] = (None if traffic is None else
                                          "profiles/pmc_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                          "command (separate runs, gfx950 x2 read correction), not measured in this process")
            roofline["event_sample"] = (f"HIP events bracket the kernels of every {ev_every}th step of the timed region "
                                        f"({kernels[dom]['launches']} bracketed launches of this kernel in {args.steps} steps)" if ev_every > 1 else "every step")
            if "lbfgs_single_pass" in table:
                roofline["single_pass"] = {"launches": opt.single_pass_steps - passes0,
                                           "retry_passes": opt.single_pass_retries - retries0,
                                           "fallback_gram_passes": table.get("lbfgs_gram_pass", (0, 0))[0],
                                           "note": "first trial rejected: the pass runs once more at t/2 when the objective there "
                                                   "(carried by the pass) is a decrease; otherwise the step finishes on the trial "
                                                   "kernels and the next step needs a Gram pass"}
            if two_pass is not None:
                roofline["two_pass"] = two_pass
            # what the headline is, in the line
            if dom == "lbfgs_single_pass":
                contract = (4 * k + 2) * n * esize
                us_dom = kernels[dom]["avg_us"]
                roofline["frac_contract_units"] = round(contract / (us_dom * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                roofline["frac_contract_units_note"] = ("SURVEY 8(d)'s (4k+2) n T per two-loop over this kernel's time: > 1 means the pass moves FEWER bytes than 8(d) "
                                                        "assumes (it keeps the last k+1 points and recomputes their gradients in registers), it is not bandwidth; "
                                                        "`frac` is on the kernel's own bytes, and the two-loop streamed as 8(d) describes it is `two_pass`")
                passes = (opt.single_pass_steps - passes0) + (opt.single_pass_retries - retries0)
                own = roofline["algorithmic_bytes_per_launch"] * passes / args.steps
                roofline["step_frac"] = round(own / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)
                roofline["step_frac_note"] = "the passes' own bytes per step!() (retries included) over ms_per_step: what of the HBM peak a whole step sustains"

    value = world * args.steps / elapsed
    out = {
        "metric": "step!() calls/sec and achieved HBM GB/s, L-BFGS n=10^7 m=20 fp64",
        "value": round(value, 3), "unit": "step!() calls/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"L-BFGS m={m} on N-D chained Rosenbrock, n={n}, fp64 (BASELINE configs[2])" + (f" + decorators {args.decorators}" if decor else ""),
                   "n": n, "m": m, "history_full": k == m,
                   "two_loop": ((("single_pass on the point ring (one sweep over the last k+1 POINTS per trial of a step; their gradients "
                                  "are recomputed in registers from the points, pairs formed in registers)" if opt.pass_recomputes_gradients else
                                  "single_pass on the point ring (one sweep over the last k+1 points and gradients per trial of a "
                                  "step, pairs formed in registers)") if opt.ring_layout == 2 else
                                 "single_pass on the pair ring (one sweep over the history per trial; gram + combine only after "
                                 "a step that needed t <= 1/4)") if "lbfgs_single_pass" in table else args.mode),
                   "history_layout": ({0: "slabs", 1: "tiles of pairs", 2: "tiles of points"}[opt.ring_layout] +
                                      {0: "", 1: ", tile-major", 2: ", stream-major"}[opt.tile_arrangement]),
                   "fast_path_requires": ("`value` is the point pass: objective = the library's built-in chained Rosenbrock (its gradient is a 3-point "
                                          "stencil the pass recomputes), with or without the L2 / box decorators (secondary.lbfgs_decorated), any n "
                                          ">= 8 incl. ragged (secondary.lbfgs_ragged) with n T < 4 GiB, m <= 24 (fp32: 20), backtracking search, no "
                                          "descent check / fallback, GRAM mode, 16-byte aligned x0.  User CALLBACKS -- the reference's real API -- "
                                          "any other objective, Wolfe, safeguards or CHAIN mode run the general two-pass step: "
                                          "secondary.lbfgs_callbacks and roofline.two_pass are that rate"),
                   "parallelism": (f"1 optimizer instance per GPU (replicas), world size {world}; convergence flag: "
                                   f"{flag.transport}, {flag.collectives} collectives in the timed region")
                   if world > 1 else "single GPU",
                   "rccl_world_size": comm.nranks if comm is not None else None,
                   "per_rank": per_rank,
                   "objective_evals_per_step": round(trials / args.steps, 3), "any_stuck": any_stuck,
                   "f_start": f_start, "f_end": opt.current_objective_value, "device": info["name"]},
        "roofline": roofline, "kernels": kernels,
    }
    if batched is not None:
        out["batched"] = batched
        # N > 1: `value` above is N config-3 replicas (a single huge-n problem stays on one GPU: "replicas only").  The
        # quantity north_star SHARDS is config 5 -- lifted next to `value` so that a scaling record built from this
        # line's top level carries it
        out["sharded_metric"] = batched.get("metric")
        out["sharded_value"] = batched.get("value")
        out["sharded_unit"] = batched.get("unit")
        out["sharded_per_rank"] = batched.get("per_rank_instance_steps_per_s")
        out["sharded_instances_total"] = batched.get("config", {}).get("instances_total")
        out["rccl_world_size"] = batched.get("config", {}).get("rccl_world_size")
    if not args.no_cpu_baseline and world == 1:          # reported baseline: rank 0 at N = 1 only
        threads = host_cores()
        cn = args.cpu_n or n
        scale = 1.0 if cn == n else cn / n
        v, v1 = cpu_baseline(cn, m, m, args.cpu_steps, threads, args.cpu_steps_single)
        out["cpu_baseline"] = {"value": round(v * scale, 4), "unit": "step!() calls/s", "cores": threads, "kind": "port",
                               "single_thread": {"value": round(v1 * scale, 4), "cores": 1, "steps": args.cpu_steps_single},
                               "sample": f"oracle/dzo_oracle.c (unfused op sequence of the reference), n={cn}, m={m}: {m} untimed "
                                         f"steps to fill the history, then {args.cpu_steps} timed step!() calls with OpenMP x{threads} "
                                         f"(= the host cores this job may use) and {args.cpu_steps_single} more on one core"
                                         + ("" if cn == n else f"; rates scaled by {cn}/{n}")}
        out["cpu_baseline"]["julia"] = julia_host_probe(n)
    if world == 1 and not args.no_cpu_baseline and not args.no_secondary and not opt.is_stuck:
        # The same optimizer run on: 1000 more step!() calls, timed as a block without any event record.  `value` above is
        # the contract's window (W warm-up steps after the history filled, then K steps); early in this trajectory more
        # first trials are rejected (8 of 50 there against 11 % over 20 000 steps), so the rate of a long run is higher --
        # reported, never substituted for `value`.
        sus_steps, sus_trials = 1000, 0
        dzo.synchronize()
        t1 = time.perf_counter()
        for _ in range(sus_steps):
            opt.step()
            sus_trials += opt.last_trials
            if opt.is_stuck:
                break
        dzo.synchronize()
        el1 = time.perf_counter() - t1
        done1 = opt.iteration_count - (m + args.warmup + args.steps)
        out["sustained"] = {"steps": done1, "step_calls_per_s": round(done1 / el1, 2), "ms_per_step": round(1e3 * el1 / max(done1, 1), 4),
                            "objective_evals_per_step": round(sus_trials / max(done1, 1), 3), "f_end": opt.current_objective_value,
                            "note": "the same optimizer continued after the timed region (no HIP event records in this stretch); not part of `value`"}
    if world == 1 and not args.no_cpu_baseline and not args.no_secondary and n == 10_000_000 and m == 20:
        # the remaining BASELINE configs, measured by whoever runs the default line (the full default line only: the
        # profiling / A-B commands all pass --no-cpu-baseline and stay as they were)
        opt.close()
        out["secondary"] = secondary_in_line(args, dzo, sharding, info)
        # (the log-sum-exp variants stay out of the default line: config 4's problem is done after two or three steps, tools/lse_steps.py)
        for kind in ("callbacks", "decorated", "ragged", "quadratic_chain"):
            t_leg = time.perf_counter()
            try:
                leg = _lbfgs_variant_leg(dzo, n, m, esize, args, kind)
                leg["leg_wall_s"] = round(time.perf_counter() - t_leg, 2)
                out["secondary"]["lbfgs_" + kind] = leg
            except Exception as e:                              # noqa: BLE001 -- reported in the line
                out["secondary"]["lbfgs_" + kind] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
