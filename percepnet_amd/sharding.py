"""Stream sharding across the GPUs of a node: streams are independent units, so rank r owns a
contiguous block of stream ids and its own context; there is no data-path collective.  RCCL (torch.distributed
"nccl") is used only for the start/stop barrier, the MAX-over-ranks of the wall time and the report gather.

Everything bench.py does around its GPU step lives here — launching the ranks, the timed region, the aggregation and
the consistency checks of the report — so that tests/test_multiproc.py runs the very same code at world_size 2 on
CPU (gloo) with a CPU step.  The reference's own parallelism is process fan-out in a shell loop
(/root/reference/utils/run.sh:49,65,99); this is its one-process-per-GPU counterpart."""
import os
import socket
import subprocess
import sys
import time


def shard_streams(total_streams, rank, world):
    """Contiguous, balanced block of global stream ids owned by `rank`."""
    base, rem = divmod(total_streams, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_world():
    """(rank, local_rank, world) from the torch.distributed.run environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def spawn_ranks(n_ranks, script, script_args, timeout=None, env=None):
    """Re-launch `script` as n_ranks processes of one node (one per GPU) through torch.distributed.run, rendezvous on
    127.0.0.1.  stdout/stderr are inherited (rank 0 prints the report).  Returns the launcher's exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(script_args)
    e = dict(os.environ if env is None else env)
    e.setdefault("MASTER_ADDR", "127.0.0.1")
    e.setdefault("OMP_NUM_THREADS", "1")
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=e, timeout=timeout).returncode


def launched_by_torchrun():
    """True when this process was started by torch.distributed.run (it exports RANK and WORLD_SIZE, also for one rank)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def init_ranks(backend, expected_world, device=None):
    """Join the process group whenever this process was launched by torch.distributed.run — ALSO for a world of one, so
    that a 1-GPU box runs the very same RCCL calls (init, barriers, all-reduces, object gather) as the 8-GPU run — and
    REFUSE a world that is not the one asked for: a report must never say n_gpus = N unless N ranks really ran and
    synchronised.  expected_world None = adopt the launched world.  Returns torch.distributed or None (plain process)."""
    rank, _, world = launched_world()
    if expected_world is not None and expected_world != world:
        raise SystemExit(f"asked for {expected_world} rank(s) but launched with WORLD_SIZE={world}: refusing to "
                         f"report a world that did not run")
    if not launched_by_torchrun():
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    assert dist.get_world_size() == world and dist.get_rank() == rank
    return dist


def timed_steps(dist, step, warmup, steps, sync):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + device sync on both sides.
    step(t) runs step t (t = 0 .. W+K-1); sync() drains this rank's device.  Returns this rank's seconds;
    step.before_timed (optional attribute) is called after the warm-up, before the timed region."""
    for t in range(warmup):
        step(t)
    sync()
    hook = getattr(step, "before_timed", None)
    if hook:
        hook()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for t in range(warmup, warmup + steps):
        step(t)
    sync()
    if dist is not None:
        dist.barrier()
    return time.perf_counter() - t0


def aggregate_throughput(dist, my_stream_frames, my_seconds):
    """-> (whole-job stream-frames per second, max-over-ranks seconds).  `dist` is
    torch.distributed (initialised) or None for a single process."""
    if dist is None or not dist.is_initialized():
        return my_stream_frames / my_seconds, my_seconds
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"   # gloo reduces host tensors
    t = torch.tensor([my_seconds], dtype=torch.float64, device=dev)
    n = torch.tensor([float(my_stream_frames)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(n.item()) / float(t.item()), float(t.item())


def gather_report(dist, my_stream_frames, my_seconds, device_label, extra=None):
    """Whole-job report: SUM of stream-frames over ranks / MAX of seconds, plus what every rank ran on.
    Returns (on every rank) {"fps", "seconds", "n_ranks", "ranks": [{"rank", "device", "stream_frames", "seconds",
    "fps"} ...]}; raises if two ranks used the same device or the gathered world is not the launched one."""
    fps, dt = aggregate_throughput(dist, my_stream_frames, my_seconds)
    rank, _, world = launched_world()
    me = {"rank": rank, "device": device_label, "stream_frames": int(my_stream_frames), "seconds": round(my_seconds, 6),
          "fps": round(my_stream_frames / my_seconds, 1)}
    if extra:
        me.update(extra)
    if dist is None:
        ranks = [me]
    else:
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, me)
    if len(ranks) != world or sorted(r["rank"] for r in ranks) != list(range(world)):
        raise RuntimeError(f"report gathered from {len(ranks)} ranks, launched world is {world}")
    devs = [r["device"] for r in ranks]
    if len(set(devs)) != len(devs) and not os.environ.get("PN_ALLOW_SHARED_DEVICE"):
        raise RuntimeError(f"two ranks ran on the same device: {devs}")
    return {"fps": fps, "seconds": dt, "n_ranks": world, "ranks": ranks}
