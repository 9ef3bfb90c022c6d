"""Stream sharding across the GPUs of a node: streams are independent units, so
rank r simply owns a contiguous block of stream ids and its own context; there is no data-path
collective.  RCCL (torch.distributed "nccl") is used only for the start/stop barrier and for the
MAX-over-ranks of the wall time."""


def shard_streams(total_streams, rank, world):
    """Contiguous, balanced block of global stream ids owned by `rank`."""
    base, rem = divmod(total_streams, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def aggregate_throughput(dist, my_stream_frames, my_seconds):
    """-> (whole-job stream-frames per second, max-over-ranks seconds).  `dist` is
    torch.distributed (initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return my_stream_frames / my_seconds, my_seconds
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"   # gloo reduces host tensors
    t = torch.tensor([my_seconds], dtype=torch.float64, device=dev)
    n = torch.tensor([float(my_stream_frames)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(n.item()) / float(t.item()), float(t.item())
