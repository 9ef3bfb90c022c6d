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


# ---- one host feeding N GPUs: NUMA placement and per-rank legs of the report (round-4 verdict item 3) ----------------------
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def numa_bind_for_device(pci_bdf, sysfs_root="/sys", bind=True):
    """Bind this process to the CPUs of the NUMA node its GPU hangs off — BEFORE any pinned host buffer is allocated (first
    touch places the pages) — and say what was done.  pci_bdf: 'dddd:bb:dd.f' (lower-case hex).  The node comes from
    <sysfs>/bus/pci/devices/<bdf>/numa_node, its CPUs from <sysfs>/devices/system/node/node<N>/cpulist, intersected with the
    affinity mask the process already has.  Never fatal: an unknown node (-1, containers, single-socket hosts) leaves the
    affinity alone and the report says why.  The reference's own fan-out has no placement at all
    (/root/reference/utils/run.sh:49,65,99: a shell loop of processes); its per-stream clock is src/main.cpp:30-39."""
    info = {"pci": pci_bdf, "numa_node": None, "bound": False, "cpus": None}
    try:
        node = int(open(os.path.join(sysfs_root, "bus/pci/devices", pci_bdf, "numa_node")).read().strip())
    except (OSError, ValueError) as e:
        info["why"] = f"no numa_node for {pci_bdf}: {type(e).__name__}"
        return info
    info["numa_node"] = node
    if node < 0:
        info["why"] = "the platform reports no NUMA affinity for this device (numa_node = -1)"
        return info
    try:
        cpus = set(parse_cpulist(open(os.path.join(sysfs_root, "devices/system/node", f"node{node}", "cpulist")).read()))
    except (OSError, ValueError) as e:
        info["why"] = f"no cpulist for node {node}: {type(e).__name__}"
        return info
    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    use = sorted(cpus & allowed) if sysfs_root == "/sys" else sorted(cpus)
    info["cpus"] = len(use)
    if not use:
        info["why"] = f"node {node} has no CPU inside this process's affinity mask"
        return info
    if bind:
        try:
            os.sched_setaffinity(0, use)
            info["bound"] = True
        except (OSError, AttributeError) as e:
            info["why"] = f"sched_setaffinity failed: {type(e).__name__}"
    else:
        info["why"] = "binding not requested"
    return info


def gather_objects(dist, obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank ([obj] for a plain process)."""
    if dist is None or not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def barrier(dist):
    if dist is not None and dist.is_initialized():
        dist.barrier()


def summarize_realtime_ranks(per_rank):
    """Aggregate of the per-rank paced real-time runs (one dict per rank, bench.paced_realtime): every rank must meet the
    contract for the node to meet it; streams add up."""
    ok = [r for r in per_rank if r and "error" not in r]
    if len(ok) != len(per_rank):
        return {"ranks_failed": len(per_rank) - len(ok), "all_ranks_met_every_deadline": False}
    return {"streams_total": sum(r["streams"] for r in ok),
            "deadline_misses_total": sum(r["deadline_misses"] for r in ok),
            "deadline_misses_per_rank": [r["deadline_misses"] for r in ok],
            "delivery_latency_ms_p99_worst_rank": max(r["delivery_latency_ms"]["p99"] for r in ok) if all("delivery_latency_ms" in r for r in ok) else None,
            "all_ranks_met_every_deadline": all(r.get("met_contract") for r in ok)}


CPU_BASELINE_ENV = "PN_BENCH_CPU_BASELINE_JSON"


def cpu_baseline_handoff(compute, world, rank):
    """The CPU baseline is timed ONCE per job on an otherwise idle host and attached to rank 0's line at any N:
    * `python bench.py --gpus N` (our own launcher): the PARENT times it before it spawns the ranks and hands it over
      through a JSON file named in the environment (spawn_with_cpu_baseline);
    * ranks launched directly by torch.distributed.run: rank 0 times it before it joins the process group — the other
      ranks are then blocked in the rendezvous, i.e. idle.
    compute: () -> dict.  Returns the dict on rank 0, None elsewhere."""
    if rank != 0:
        return None
    path = os.environ.get(CPU_BASELINE_ENV)
    if path and os.path.exists(path):
        import json
        with open(path) as f:
            return json.load(f)
    return compute()


def spawn_with_cpu_baseline(n_ranks, script, script_args, compute):
    """spawn_ranks after timing the CPU baseline in this (parent) process: compute may be None (--no-cpu-baseline)."""
    import json
    import tempfile
    env = dict(os.environ)
    tmp = None
    if compute is not None:
        res = compute()
        tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
        json.dump(res, tmp); tmp.close()
        env[CPU_BASELINE_ENV] = tmp.name
    try:
        return spawn_ranks(n_ranks, script, script_args, env=env)
    finally:
        if tmp is not None:
            os.unlink(tmp.name)


def multi_rank_fields(dist, cpu, numa, sustained, realtime):
    """The parts of the report line that exist PER RANK when one host feeds N GPUs, gathered like `ranks` — the same
    keys at any N, so that an 8-GPU line is a complete record (cpu_baseline; one host's ability to pace
    N x B streams: per-rank sustained and paced real-time legs; where each rank's host memory lives: numa).
    cpu: the CPU baseline dict (rank 0) or None; numa / sustained / realtime: THIS rank's objects (or None when the leg was
    skipped).  Returns the dict of fields on every rank (rank 0 prints it)."""
    rank, _, world = launched_world()
    out = {"numa": gather_objects(dist, numa)}
    sus = gather_objects(dist, sustained)
    rt = gather_objects(dist, realtime)
    if any(s is not None for s in sus):
        out["sustained_ranks"] = sus
    if any(r is not None for r in rt):
        out["realtime_ranks"] = rt
        out["realtime_all_ranks"] = summarize_realtime_ranks(rt)
    cpus = gather_objects(dist, cpu)
    got = [c for c in cpus if c is not None]
    if got:
        out["cpu_baseline"] = got[0]
    return out
