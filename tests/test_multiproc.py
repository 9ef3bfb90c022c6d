"""world_size-2 CPU (gloo) tests of the multi-GPU launch path.  bench.py keeps everything around its GPU step in
percepnet_amd/sharding.py — self-launch of the ranks (spawn_ranks), joining and REFUSING a wrong world (init_ranks),
the barrier-bracketed timed region (timed_steps), SUM-frames / MAX-time aggregation and the per-rank report with its
duplicate-device check (gather_report) — and this file runs exactly that code end to end with a CPU step."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time, json
    sys.path.insert(0, %r)
    import numpy as np
    from percepnet_amd import synth, sharding
    expected = int(sys.argv[1]); share = sys.argv[2] == "share"
    dist = sharding.init_ranks("gloo", expected)
    rank, local_rank, world = sharding.launched_world()
    total = 10                                  # global stream ids 0..9, ragged over the ranks
    mine = sharding.shard_streams(total, rank, world)
    pcm = np.stack([synth.synth_stream(s, 2) for s in mine])     # each rank touches only its shard
    calls = []
    def step(t):
        calls.append(t); time.sleep(0.02 * (rank + 1))           # rank 1 is the slow one
    step.before_timed = lambda: calls.append("timed")
    dt = sharding.timed_steps(dist, step, 2, 3, lambda: None)
    rep = sharding.gather_report(dist, len(mine) * 3, dt, "cpu:0" if share else "cpu:%%d" %% local_rank, {"ids": list(mine)})
    if rank == 0:
        print(json.dumps({"rep": rep, "calls": calls, "dt0": dt}))
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
""")


def _launch(tmp_path, n, expected, share="own"):
    from percepnet_amd import sharding
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    cmd = [sys.executable, "-c",
           "import sys; sys.path.insert(0, %r); from percepnet_amd import sharding; "
           "sys.exit(sharding.spawn_ranks(%d, %r, [%r, %r], timeout=240))" % (ROOT, n, str(script), str(expected), share)]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))


def test_two_rank_launch_timing_and_report(tmp_path):
    out = _launch(tmp_path, 2, 2)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    rep = r["rep"]
    assert rep["n_ranks"] == 2 and [x["rank"] for x in rep["ranks"]] == [0, 1]
    assert [x["device"] for x in rep["ranks"]] == ["cpu:0", "cpu:1"]          # who ran where is part of the report
    flat = sorted(i for x in rep["ranks"] for i in x["ids"])
    assert flat == list(range(10))                                            # every stream owned by exactly one rank
    assert abs(len(rep["ranks"][0]["ids"]) - len(rep["ranks"][1]["ids"])) <= 1
    assert r["calls"] == [0, 1, "timed", 2, 3, 4]                             # W warm-up steps, hook, exactly K timed
    slow = max(x["seconds"] for x in rep["ranks"])
    assert rep["seconds"] >= 0.11 and abs(rep["seconds"] - slow) < 1e-3        # MAX over ranks (rank 1 sleeps 3 x 40 ms)
    assert abs(rep["fps"] - 30 / rep["seconds"]) < 1e-6 * rep["fps"] + 1e-3    # SUM of stream-frames / MAX time
    assert abs(r["dt0"] - rep["seconds"]) < 0.03                              # the closing barrier holds rank 0 for rank 1


def test_world_of_one_still_joins_a_process_group(tmp_path):
    """A single rank launched through torch.distributed.run joins a group of one and runs the same barriers,
    all-reduces and object gather as N ranks (what `bench.py --force-dist` exercises with RCCL on a 1-GPU box)."""
    out = _launch(tmp_path, 1, 1)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["rep"]["n_ranks"] == 1 and r["rep"]["ranks"][0]["ids"] == list(range(10))
    assert r["calls"] == [0, 1, "timed", 2, 3, 4]
    assert abs(r["rep"]["fps"] - 30 / r["rep"]["seconds"]) < 1e-6 * r["rep"]["fps"] + 1e-3


def test_wrong_world_is_refused(tmp_path):
    out = _launch(tmp_path, 2, 3)               # asked for 3 ranks, launched 2: never report a world that did not run
    assert out.returncode != 0
    assert "refusing to report" in out.stderr


def test_two_ranks_on_one_device_are_refused(tmp_path):
    out = _launch(tmp_path, 2, 2, share="share")
    assert out.returncode != 0
    assert "same device" in out.stderr


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts 2 ranks itself; without a GPU each rank stops at the
    no-fallback check — which shows both were launched through torch.distributed.run."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check (the GPU version is tests/test_gpu_bench.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    # the elastic agent SIGTERMs the surviving rank as soon as the first one exits, so one or two ranks get to print the
    # message; the agent's failure report names every rank it launched
    assert 1 <= out.stderr.count("bench.py needs a GPU") <= 2, out.stderr[-2000:]
    assert "local_rank: 0" in out.stderr and "local_rank: 1" in out.stderr, out.stderr[-2000:]


FIELDS_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    from percepnet_amd import sharding
    dist = sharding.init_ranks("gloo", 2)
    rank, local_rank, world = sharding.launched_world()
    cpu = sharding.cpu_baseline_handoff(lambda: {"value": -1, "kind": "computed by rank 0"}, world, rank)
    numa = sharding.numa_bind_for_device("0000:%%02x:00.0" %% (0x10 + local_rank), sysfs_root=sys.argv[1], bind=False)
    rt = {"streams": 100 + rank, "deadline_misses": rank, "met_contract": rank == 0, "delivery_latency_ms": {"p99": 12.0 + rank}}
    sus = {"ms_per_step": 9.5 + rank}
    f = sharding.multi_rank_fields(dist, cpu, numa, sus, rt)
    if rank == 0:
        print(json.dumps(f))
    sharding.barrier(dist)
    dist.destroy_process_group()
""")


def test_the_two_rank_line_is_a_complete_record(tmp_path):
    """Round-4 verdict item 3: an N > 1 line must carry cpu_baseline (timed once by the launcher parent and handed to
    rank 0), one paced real-time and one sustained object PER RANK plus their aggregate, and where each rank's host memory
    lives (numa) — through the very functions bench.py uses (sharding.spawn_with_cpu_baseline, cpu_baseline_handoff,
    numa_bind_for_device, multi_rank_fields), at world_size 2 on gloo with a fake sysfs tree."""
    sysfs = tmp_path / "sys"
    for k, node in ((0x10, 0), (0x11, 1)):
        d = sysfs / "bus/pci/devices" / ("0000:%02x:00.0" % k); d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
        n = sysfs / "devices/system/node" / f"node{node}"; n.mkdir(parents=True)
        (n / "cpulist").write_text("0-3,8\n" if node == 0 else "4-7,9-10\n")
    script = tmp_path / "fields_worker.py"
    script.write_text(FIELDS_WORKER % ROOT)
    cmd = [sys.executable, "-c",
           "import sys; sys.path.insert(0, %r); from percepnet_amd import sharding; "
           "sys.exit(sharding.spawn_with_cpu_baseline(2, %r, [%r], lambda: {'value': 33.7, 'kind': 'timed by the launcher parent'}))"
           % (ROOT, str(script), str(sysfs))]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    f = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert f["cpu_baseline"] == {"value": 33.7, "kind": "timed by the launcher parent"}      # the parent's, not recomputed
    assert [n["numa_node"] for n in f["numa"]] == [0, 1] and [n["cpus"] for n in f["numa"]] == [5, 6]
    assert [n["pci"] for n in f["numa"]] == ["0000:10:00.0", "0000:11:00.0"]
    assert [r["streams"] for r in f["realtime_ranks"]] == [100, 101]
    agg = f["realtime_all_ranks"]
    assert agg["streams_total"] == 201 and agg["deadline_misses_per_rank"] == [0, 1] and agg["all_ranks_met_every_deadline"] is False
    assert agg["delivery_latency_ms_p99_worst_rank"] == 13.0
    assert [s["ms_per_step"] for s in f["sustained_ranks"]] == [9.5, 10.5]


def test_numa_binding_reports_instead_of_failing(tmp_path):
    from percepnet_amd import sharding
    assert sharding.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    r = sharding.numa_bind_for_device("0000:aa:00.0", sysfs_root=str(tmp_path))
    assert r["bound"] is False and "no numa_node" in r["why"]
    d = tmp_path / "bus/pci/devices/0000:aa:00.0"; d.mkdir(parents=True); (d / "numa_node").write_text("-1\n")
    r = sharding.numa_bind_for_device("0000:aa:00.0", sysfs_root=str(tmp_path))
    assert r["numa_node"] == -1 and r["bound"] is False and "no NUMA affinity" in r["why"]
