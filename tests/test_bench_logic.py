"""CPU-only checks of the MEASUREMENT code (bench.py, tools/summarize_prof.py): the numbers in the bench line must follow from
the run — these are the pure functions behind them, exercised without a GPU."""
import csv
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import realtime_capacity as rtc  # noqa: E402


def test_spec_out_algorithmic_bytes_follow_from_the_runs_periods():
    """fe_spec_out reads the comb window [2400 - 3T, 3360 + 3T) = 960 + 6T samples of each stream (reference denoise.cpp:416-422):
    its algorithmic bytes are the sum over the run's own periods + the fixed rows, the T = 768 figure is kept beside them."""
    B = 1000
    T = np.full(B, 400, dtype=np.int32); T[:100] = 768; T[100:200] = 60
    kt = {"fe_spec_out": (0.5, 2), "fe_pitch": (1.0, 2)}
    r = bench.dsp_roofline(kt, B, tag="_no_such_profile", periods=T)
    so = r["fe_spec_out"]
    want = int(((960 + 6 * T.astype(np.int64)) * 4).sum()) + B * bench.SPEC_OUT_FIXED
    assert so["algorithmic_bytes"] == want and so["algorithmic_bytes_worst_case"] == B * (22272 + bench.SPEC_OUT_FIXED)
    assert abs(so["pitch_period_mean"] - float(T.mean())) < 0.06 and so["ms"] == 0.25
    assert abs(so["frac"] - want / 0.25e-3 / 1e12 / 8.0) < 1e-4
    assert so["traffic"] is None                                   # no profile of these kernels: refused, not invented
    worst = bench.dsp_roofline(kt, B, tag="_no_such_profile", periods=None)["fe_spec_out"]
    assert worst["algorithmic_bytes"] == worst["algorithmic_bytes_worst_case"] and "worst case" in worst["algorithmic_bytes_source"]
    assert r["dsp_total"]["algorithmic_bytes"] == want + B * bench.DSP_KERNELS["fe_pitch"][1]


def test_capacity_search_bisects_and_reports_the_next_size(monkeypatch):
    """realtime_capacity: the largest batch of the 512-stream grid whose runs ALL pass, the pass rate of the next size up, a
    fallback to smaller batches when 65 536 itself fails — with a fake paced run (a batch passes iff it is <= a limit; one size
    is flaky)."""
    class FakeCtx:
        def __init__(self, *a, **k): pass
        def close(self): pass

    class FakeApi:
        Context = FakeCtx
        NN_MFMA = 0

    calls = []

    def fake_paced(limit, flaky=None, recover_limit=10 ** 9, soak_limit=10 ** 9):
        def f(api, synth, model, dev, b, mode, seconds, ctx=None, stall=None):
            calls.append((b, seconds, stall))
            if stall is not None:                                                    # the disturbed run of a size
                ok = b <= min(limit, recover_limit)
                return {"streams": b, "met_contract": ok, "recovery": {"frames_to_recover": 120 if ok else None, "recovered": ok}}
            n = sum(1 for c in calls if c[0] == b and c[2] is None)
            ok = b <= limit and not (b == flaky and n == 2) and not (seconds >= 20 and b > soak_limit)
            return {"streams": b, "deadline_misses": 0 if ok else 7, "delivery_latency_ms": {"p99": 12.0 if ok else 80.0}, "met_contract": ok}
        return f

    monkeypatch.setattr(rtc, "paced_realtime", fake_paced(67700))
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 3, lambda m: None)
    assert r["realtime_streams_p99"] == 67584 and r["next_size"] == {"streams": 68096, "runs": 3, "passed": 0, "recovered": [False]}
    assert r["sizes"]["67584"]["passed"] == 3 and set(r["sizes"]) == {"65536", "67584", "68096", "68608"}
    assert r["sizes"]["67584"]["stall_recovery"] == [{"frames_to_recover": 120, "recovered": True}]
    calls.clear()
    monkeypatch.setattr(rtc, "paced_realtime", fake_paced(10 ** 9))
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 2, lambda m: None)
    assert r["realtime_streams_p99"] == 69632 and r["next_size"] is None
    calls.clear()
    monkeypatch.setattr(rtc, "paced_realtime", fake_paced(10 ** 9, flaky=67584))          # one run of three fails at 67 584
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 3, lambda m: None)
    assert r["realtime_streams_p99"] == 67072 and r["next_size"] == {"streams": 67584, "runs": 3, "passed": 2, "recovered": [True]}
    calls.clear()
    monkeypatch.setattr(rtc, "paced_realtime", fake_paced(60000))
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 1, lambda m: None)
    assert r["realtime_streams_p99"] == 57344 and r["next_size"]["streams"] == 65536
    # every undisturbed run passes everywhere, but only sizes <= 66 600 catch up after the injected stall
    calls.clear()
    monkeypatch.setattr(rtc, "paced_realtime", fake_paced(10 ** 9, recover_limit=66600))
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 2, lambda m: None)
    assert r["realtime_streams_p99"] == 66560 and r["next_size"] == {"streams": 67072, "runs": 2, "passed": 2, "recovered": [False]}
    # the confirmation run: 67 584 passes its 1 s runs and fails the 20 s one; the search steps down the grid to the next size that
    # passes its own runs and the long one
    calls.clear()
    monkeypatch.setattr(rtc, "paced_realtime", fake_paced(67700, soak_limit=66600))
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 2, lambda m: None, soak_seconds=20.0)
    assert r["realtime_streams_p99"] == 66560 and [x["met_contract"] for x in r["confirmation_runs"]] == [False, False, True]
    assert r["next_size"]["streams"] == 67072


def test_transient_runs_are_one_recovered_burst_and_cost_an_extra_run(monkeypatch):
    """A failed undisturbed run counts as a transient only when every late frame lies in one short burst the run recovered from; the
    capacity search answers ONE transient at a size with one more undisturbed run (which must be clean), two failures fail the size."""
    n = 600
    lat = np.full(n, 12.0); missed = np.zeros(n, bool)
    assert rtc.transient_burst(missed, lat, 2.0) == (False, None)                           # nothing late
    lat[200:230] = 35.0; missed[200:210] = True
    assert rtc.transient_burst(missed, lat, 2.0) == (True, [200, 229])                      # one 30-frame burst, recovered
    assert rtc.transient_burst(missed, lat, 40.0)[0] is False                               # ... but the run ended behind its clock
    lat2 = lat.copy(); lat2[500:520] = 30.0
    assert rtc.transient_burst(missed, lat2, 2.0)[0] is False                               # two bursts 300 frames apart
    lat3 = np.full(n, 12.0); lat3[585:] = 40.0
    assert rtc.transient_burst(np.zeros(n, bool), lat3, 2.0)[0] is False                    # still late at the end of the run
    lat4 = lat.copy(); lat4[300] = np.nan
    assert rtc.transient_burst(missed, lat4, 2.0)[0] is False                               # a frame that never came out

    class FakeCtx:
        def __init__(self, *a, **k): pass
        def close(self): pass

    class FakeApi:
        Context = FakeCtx
        NN_MFMA = 0

    def fake(plan):
        """plan[b] = outcomes of the undisturbed runs of size b in order ('ok' / 'transient' / 'fail'); anything else passes"""
        seen = {}
        def f(api, synth, model, dev, b, mode, seconds, ctx=None, stall=None):
            if stall is not None:
                return {"streams": b, "met_contract": True, "recovery": {"frames_to_recover": 60, "recovered": True}}
            k = seen.get(b, 0); seen[b] = k + 1
            o = (plan.get(b, []) + ["ok"] * 9)[k]
            return {"streams": b, "deadline_misses": 0 if o == "ok" else 9, "met_contract": o == "ok", "transient": o == "transient",
                    "delivery_latency_ms": {"p99": 12.0}}
        return f

    every = {65536 + 512 * k for k in range(9)}
    # STRICT rule (round-5 verdict item 4): a transient run is a label, it is not forgiven — 65 536 with one transient of two runs is
    # NOT the strict capacity (the search falls back), it is the figure "with one forgiven burst" (the extra run was made and was clean)
    monkeypatch.setattr(rtc, "paced_realtime", fake({65536: ["ok", "transient"], 67584: ["fail"], 66560: ["fail"], 66048: ["fail"]}))
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 2, lambda m: None)
    assert r["realtime_streams_p99"] == 65024 and r["realtime_streams_with_one_forgiven_burst"] == 65536
    assert r["sizes"]["65536"] == dict(r["sizes"]["65536"], runs=3, passed=2, transient_runs=1, strict=False, with_one_forgiven_burst=True)
    assert r["sizes"]["65024"]["strict"] is True
    monkeypatch.setattr(rtc, "paced_realtime", fake({b: ["transient", "ok", "transient"] for b in every}))   # the extra run fails too
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 2, lambda m: None)
    assert r["realtime_streams_p99"] == 65024 and r["sizes"]["65536"]["runs"] == 3 and r["sizes"]["65536"]["passed"] == 1
    assert r["realtime_streams_with_one_forgiven_burst"] == 65024
    monkeypatch.setattr(rtc, "paced_realtime", fake({b: ["transient", "transient"] for b in every}))          # two transients: no extra run
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 2, lambda m: None)
    assert r["realtime_streams_p99"] == 65024 and r["sizes"]["65536"]["runs"] == 2
    # bench.py's default: ONE size and one fallback — no grid search
    monkeypatch.setattr(rtc, "paced_realtime", fake({}))
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 2, lambda m: None, grid=[65536], fallback=(61440,))
    assert r["realtime_streams_p99"] == 65536 and set(r["sizes"]) == {"65536"} and r["next_size"] is None
    monkeypatch.setattr(rtc, "paced_realtime", fake({65536: ["ok", "fail"]}))
    r = rtc.realtime_capacity(FakeApi, None, None, 0, 0, 1.0, 2, lambda m: None, grid=[65536], fallback=(61440,))
    assert r["realtime_streams_p99"] == 61440 and set(r["sizes"]) == {"61440", "65536"} and r["realtime_streams_with_one_forgiven_burst"] == 61440


def test_stall_run_verdict_checks_deliveries_after_the_recovery_point():
    """Advisor (round 5): the disturbed run's verdict used to be the recovery of the SUBMIT clock alone.  stall_verdict's recovery
    point is the first frame after the stall from which 20 consecutive frames are undisturbed (submit on the arrival, no back-pressure,
    delivery in time): the device queue drains for some frames after the submit clock is back, and those frames count.  From the
    recovery point on the run must meet an undisturbed run's contract frame by frame; sub-millisecond submit jitter there is counted,
    not failed (it failed one box of five in round 6: a wake-up 0.6 ms late, 300 frames after a completed recovery)."""
    n, at = 600, 100
    backlog = np.zeros(n); backlog[at:at + 60] = np.linspace(0.05, 0.001, 60)        # 50 ms behind, submit clock back 60 frames later
    late = np.zeros(n); late[at:at + 30] = 0.002
    lat = np.full(n, 11.0); lat[at:at + 70] = 45.0                                     # ... deliveries late for 10 frames more
    v = rtc.stall_verdict(backlog, late, lat, at, 1.0)
    assert v["recovered"] is True and v["frames_to_recover"] == 70 and v["frames_to_recover_submit_clock"] == 60
    assert v["disturbed_frames"] == 70 and v["clean_frames_after_recovery"] == n - at - 70 and v["delivery_p99_ms_after_recovery"] == 11.0
    assert v["contract_violations_after_recovery"] == 0 and v["jitter_frames_after_recovery"] == 0
    lat_bad = lat.copy(); lat_bad[450:455] = 40.0                                      # late deliveries again 350 frames after the stall
    v = rtc.stall_verdict(backlog, late, lat_bad, at, 1.0)
    assert v["recovered"] is False and v["frames_to_recover"] == 70 and v["contract_violations_after_recovery"] == 5
    lat_nan = lat.copy(); lat_nan[590] = np.nan                                        # a frame near the end never came out
    assert rtc.stall_verdict(backlog, late, lat_nan, at, 1.0)["recovered"] is False
    late2 = late.copy(); late2[520] = 0.001                                            # a back-pressure miss long after
    assert rtc.stall_verdict(backlog, late2, lat, at, 1.0)["recovered"] is False
    jitter = backlog.copy(); jitter[420] = 0.0006; jitter[500] = 0.0009                 # two submit calls that woke 0.6 / 0.9 ms late, on time otherwise
    v = rtc.stall_verdict(jitter, late, lat, at, 1.0)
    assert v["recovered"] is True and v["frames_to_recover"] == 70 and v["jitter_frames_after_recovery"] == 2
    slow = np.zeros(n); slow[at:at + 350] = np.linspace(0.05, 0.001, 350)                # a pipeline that needs 350 frames to catch up
    v = rtc.stall_verdict(slow, late, lat, at, 1.0)
    assert v["recovered"] is False and v["frames_to_recover"] == 350
    choppy = backlog.copy(); choppy[at + 70:at + 200:10] = 0.002                         # never 20 undisturbed frames in a row until frame 291
    v = rtc.stall_verdict(choppy, late, lat, at, 1.0)
    assert v["frames_to_recover"] == 191 and v["recovered"] is True
    never = np.zeros(n); never[at:] = 0.02
    v = rtc.stall_verdict(never, late, lat, at, 1.0)
    assert v["recovered"] is False and v["frames_to_recover_submit_clock"] is None and v["frames_to_recover"] is None
    assert rtc.stall_verdict(backlog, late, lat, at, 25.0)["recovered"] is False        # ended behind its clock
    early = lat.copy(); early[40] = 30.0                                               # a late frame BEFORE the stall: not a clean run
    assert rtc.stall_verdict(backlog, late, early, at, 1.0)["recovered"] is False
    warm = backlog.copy(); warm[3] = 0.0007                                            # a submit call that woke 0.7 ms late before the stall: jitter
    v = rtc.stall_verdict(warm, late, lat, at, 1.0)
    assert v["recovered"] is True and v["jitter_frames_before_the_stall"] == 1 and v["clean_before_the_stall"] is True


REQUIRED_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                      "dtype", "data", "config", "roofline", "cpu_baseline")


def test_the_printed_line_is_compact_and_complete():
    """Round-5 verdict item 1: BENCH_r05.json came back `parsed: null` because bench.py printed a 30-38 KB line.  The printed line is
    built by compact_line() from the full record: here from RECORDED full records (round 5's own lines, the largest ones that exist) —
    under 8 KB (in fact under 4), every contract key, roofline and cpu_baseline objects with their required fields."""
    for name in ("r05_bench.json", "r05_bench_third_box.json", "r05_bench_fp16.json", "r05_bench_force_dist_rccl.json"):
        d = json.loads([l for l in open(os.path.join(ROOT, "profiles", name)) if l.startswith("{")][-1])
        d.setdefault("cpu_baseline", {"value": 30.0, "unit": "streams", "frames_per_s": 3000.0, "frames_per_s_one_core": 540.0, "cores": 128,
                                      "kind": "reference", "sample": "x" * 400})
        line = bench.compact_line(d, "gpurun_out/bench_detail.json")
        assert "\n" not in line and len(line) < 4096 < bench.LINE_LIMIT, (name, len(line))
        r = json.loads(line)
        for k in REQUIRED_LINE_KEYS:
            assert k in r, (name, k)
        assert set(r["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and r["roofline"]["bound"] in ("hbm", "mfma")
        assert set(r["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and len(r["cpu_baseline"]["sample"]) <= 160
        assert "workload" in r["config"] and "model" not in r["config"]
        assert r["value"] == d["value"] and r["ms_per_step"] == d["ms_per_step"] and r["detail"] == "gpurun_out/bench_detail.json"
        assert "paced_runs" not in line and "ranks" not in r                      # the bulk stays in the detail file
    # a pathological record (hundreds of sizes) still yields a line under the limit: optional objects are dropped, the contract stays
    d["realtime"] = {"sizes": {str(60000 + k): {"runs": 2, "passed": 2, "transient_runs": 0, "strict": True} for k in range(400)}}
    line = bench.compact_line(d, None)
    assert len(line) <= bench.LINE_LIMIT and all(k in json.loads(line) for k in REQUIRED_LINE_KEYS)


def test_profile_summaries_keep_only_the_steady_state_frames(tmp_path):
    """tools/summarize_prof.py --frames-total W+K --frames-keep K: of every (kernel, grid) only the launches of the last K frames
    count — the first frames of a run from the zero state filter at the degenerate period and move more bytes."""
    src = tmp_path / "prof"; (src / "pmc_fetch").mkdir(parents=True); (src / "stats").mkdir()
    with open(src / "pmc_fetch" / "bench_counter_collection.csv", "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
        d = 0
        for frame in range(10):                                   # 10 frames: one spec_out launch and four GRU launches each
            d += 1; w.writerow([d, "pn_fe_spec_out_kernel(args)", 196608, "FETCH_SIZE", 900.0 if frame < 6 else 600.0])
            for _ in range(4):
                d += 1; w.writerow([d, "pn_gru_mfma_p_kernel(args)", 2097152, "FETCH_SIZE", 100.0 + frame])
        w.writerow([d + 1, "pn_fe_spec_out_kernel(args)", 256, "FETCH_SIZE", 1.0])      # a self-test launch on another grid: its own key
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_prof.py"), "t", "--src", str(src), "--out", str(out),
                        "--frames-total", "10", "--frames-keep", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = open(out / "t_pmc_per_launch.csv").read().splitlines()
    assert lines[0].startswith("# kernels_snapshot=") and "last 4 of 10 frames" in lines[1]
    rows = {(x["kernel"], x["counter"]): x for x in csv.DictReader(lines[2:])}
    so = rows[("pn_fe_spec_out_kernel grid=196608", "FETCH_SIZE")]
    assert so["launches"] == "4" and float(so["avg"]) == 600.0
    gru = rows[("pn_gru_mfma_p_kernel grid=2097152", "FETCH_SIZE")]
    assert gru["launches"] == "16" and abs(float(gru["avg"]) - 107.5) < 1e-9
    assert rows[("pn_fe_spec_out_kernel grid=256", "FETCH_SIZE")]["launches"] == "1"


def test_design_table_is_generated_from_the_committed_bench_lines():
    """DESIGN.md's status table is what tools/design_table.py makes of profiles/r06_bench*_detail.json (numbers cannot go stale), and
    the committed compact line of that run is the line bench.compact_line makes of the committed full record."""
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    a, b = s.index("<!-- BEGIN GENERATED"), s.index("<!-- END GENERATED -->")
    block = s[a:b]
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_detail.json")))
    assert f"| {d['ms_per_step']} |" in block and f"{d['value']:.0f}" in block
    assert str(d["realtime_streams_p99"]) in block and d["roofline"]["kernels_snapshot"] in block
    line = json.loads([l for l in open(os.path.join(ROOT, "profiles", "r06_bench.json")) if l.startswith("{")][-1])
    assert line["value"] == d["value"] and line["roofline"]["frac"] == d["roofline"]["frac"] and line["cpu_baseline"]["value"] == d["cpu_baseline"]["value"]
    assert line["realtime_streams_p99"] == d["realtime_streams_p99"]
