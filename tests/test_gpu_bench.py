"""bench.py's own launch paths on a real GPU (-m gpu): the single-rank line carries measured parity numbers, and
`--gpus 2` with WORLD_SIZE unset launches two ranks itself (both on cuda:0 here: --share-gpu, gloo) and reports
what really ran."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_rank_line_has_measured_parity_and_roofline():
    r = _bench("--streams", "2048", "--steps", "8", "--warmup", "2", "--no-cpu-baseline")
    assert r["n_gpus"] == 1 and len(r["ranks"]) == 1
    assert isinstance(r["max_abs_delta_vs_cpu_ref_lsb"], int) and r["max_abs_delta_vs_cpu_ref_lsb"] <= 1
    assert r["max_abs_delta_gr"] <= 2e-5
    assert r["parity"]["replay_of_timed_run_bit_identical"] is True
    assert r["parity"]["pcm_samples_checked"] == 64 * 9 * 480
    assert r["roofline"]["bound"] == "mfma" and 0 < r["roofline"]["frac"] < 1
    assert abs(r["value"] * 100 - r["frames_per_s"]) < 10


def test_two_ranks_self_launched():
    r = _bench("--gpus", "2", "--share-gpu", "--backend", "gloo", "--streams", "1024", "--steps", "5", "--warmup", "2")
    assert r["n_gpus"] == 2 and [x["rank"] for x in r["ranks"]] == [0, 1]
    assert all(x["stream_frames"] == 1024 * 5 for x in r["ranks"])
    assert abs(r["frames_per_s"] - 2 * 1024 * 5 / (r["ms_per_step"] * 5e-3)) < 1e-3 * r["frames_per_s"]
    assert "cpu_baseline" not in r                       # only at N = 1
