"""world_size-2 CPU (gloo) test of the multi-GPU launch logic bench.py uses: streams are sharded
across ranks with no data-path collective; ranks only meet at a barrier and a MAX all-reduce of the
wall time, and rank 0 aggregates value = (sum of stream-frames over ranks) / max time."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time, json
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from percepnet_amd import synth
    from percepnet_amd.sharding import shard_streams, aggregate_throughput
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 10                                  # global stream ids 0..9, ragged over 2 ranks
    mine = shard_streams(total, rank, world)
    pcm = np.stack([synth.synth_stream(s, 2) for s in mine])     # each rank touches only its shard
    dist.barrier()
    t0 = time.perf_counter(); time.sleep(0.05 * (rank + 1)); dt = time.perf_counter() - t0
    dist.barrier()
    fps, dt_max = aggregate_throughput(dist, len(mine) * 2, dt)
    ids = [None] * world
    dist.all_gather_object(ids, list(mine))
    if rank == 0:
        print(json.dumps({"ids": ids, "fps": fps, "dt_max": dt_max, "frames": total * 2}))
    dist.destroy_process_group()
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_sharding_and_aggregation(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
        env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    flat = sorted(x for part in r["ids"] for x in part)
    assert flat == list(range(10))                       # every stream owned by exactly one rank
    assert abs(len(r["ids"][0]) - len(r["ids"][1])) <= 1  # balanced
    assert r["dt_max"] >= 0.09                            # MAX over ranks, not rank 0's own time
    assert abs(r["fps"] - r["frames"] / r["dt_max"]) < 1e-6
