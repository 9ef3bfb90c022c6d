#!/usr/bin/env python
"""The paced real-time harness and the deadline-PROVEN capacity search (measurement, not product).

The real-time contract is the reference's own (src/main.cpp:30-39: one 480-sample frame per stream every 10 ms).  bench.py
runs ONE size of it by default (two undisturbed paced runs + one run with an injected host stall at its batch size); the search
over the 512-stream grid lives here:

    python tools/realtime_capacity.py [--seconds 6] [--runs 2] [--soak-seconds 20] [--grid 65536:69632:512] [--out file.json]

Rules (round-5 verdict items 3/4, advisor note on bench.py:653):
  * a RUN passes with zero back-pressure misses, every frame delivered, delivery p99 <= DELIVERY_DEADLINE_MS and no schedule slip;
  * the STALL run passes when it was clean before the injected STALL_MS host hiccup, the pipeline is undisturbed again — 20 frames in
    a row submitted on their arrivals, back-pressuring nothing, delivered within the deadline — within RECOVERY_FRAMES of it, and from
    that point to the end no frame violates the contract (no back-pressure miss, no late or missing delivery; sub-millisecond submit
    jitter there is counted, not failed — stall_verdict says why);
  * `realtime_streams_p99` (STRICT) = the largest size whose undisturbed runs ALL pass — nothing forgiven — and whose stall run passes;
  * `realtime_streams_with_one_forgiven_burst` = the largest size that passes when ONE transient run (every late frame inside one
    burst the run recovered from — a stall of the box) is answered by one extra undisturbed run that is clean.  Never the headline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FRAME = 480
TRANSIENT_SPAN_FRAMES = 150          # a failed undisturbed run whose late frames all lie within 1.5 s and that recovered: a stall of the box
STALL_AT_FRAME, STALL_MS, RECOVERY_FRAMES = 100, 50.0, 300   # the disturbed paced run: a 50 ms host stall, back on the clock within 3 s
DELIVERY_DEADLINE_MS = 20.0          # two frame periods = the depth of the pipelined host path (two frames in flight)
DEFAULT_GRID = tuple(65536 + 512 * k for k in range(9))
DEFAULT_FALLBACK = (65024, 64512, 63488, 61440, 57344, 53248, 49152)


def gpu_clock_mhz():
    """Current shader clock (MHz) as rocm-smi reports it, or None."""
    import re
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        m = re.search(r"sclk clock level:?\s*\S*\s*\(?(\d+)\s*Mhz", out, re.I)
        return int(m.group(1)) if m else None
    except Exception:                         # noqa: BLE001 — a missing tool must not cost the bench line
        return None


class ClockSampler:
    """Shader clock (rocm-smi) sampled from a side thread while a loop runs: min / max / samples."""

    def __init__(self, period=0.5):
        import threading
        self.period, self.vals, self._stop = period, [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            v = gpu_clock_mhz()
            if v:
                self.vals.append(v)
            self._stop.wait(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set(); self._th.join(timeout=30)

    def summary(self):
        return {"sclk_mhz_min": min(self.vals), "sclk_mhz_max": max(self.vals), "sclk_samples": len(self.vals)} if self.vals else {}


def percentiles(ms):
    import numpy as np
    a = np.asarray(ms, dtype=np.float64)
    return {"p50": round(float(np.percentile(a, 50)), 4), "p99": round(float(np.percentile(a, 99)), 4), "max": round(float(a.max()), 4),
            "mean": round(float(a.mean()), 4), "frames": int(a.size)}


def transient_burst(missed, latency_ms, finished_behind_ms):
    """missed[t]: the submit call of frame t came back after frame t + 1 had arrived; latency_ms[t]: arrival-to-delivery (NaN:
    never delivered).  -> (one recovered burst?, [first, last] late frame or None): every late frame inside one window of
    TRANSIENT_SPAN_FRAMES, the last of them at least 20 frames before the end, every frame delivered, the run on its clock at the end."""
    import numpy as np
    lat = np.asarray(latency_ms, dtype=float)
    bad = np.nonzero(np.asarray(missed, dtype=bool) | (np.nan_to_num(lat, nan=1e12) > DELIVERY_DEADLINE_MS))[0]
    if not bad.size:
        return False, None
    ok = bool(not np.isnan(lat).any() and bad[-1] - bad[0] < TRANSIENT_SPAN_FRAMES and bad[-1] < lat.size - 20 and finished_behind_ms < 10.0)
    return ok, [int(bad[0]), int(bad[-1])]


def stall_verdict(backlog_s, late_s, latency_ms, stall_at, finished_behind_ms):
    """The disturbed run's verdict.  backlog_s[t]: how far behind its arrival frame t was submitted; late_s[t] > 0: the submit call
    of frame t returned after frame t + 1 had arrived; latency_ms[t]: arrival-to-delivery (NaN: never delivered).
    A frame is DISTURBED when it was submitted behind its arrival (backlog >= 0.5 ms), back-pressured the next one, was delivered
    after DELIVERY_DEADLINE_MS or never.  The stall's burst ends — the RECOVERY POINT — at the first frame from which 20 consecutive
    frames are undisturbed: submits on their arrivals AND deliveries in time again (advisor, round 5: the verdict used to look at the
    submit clock alone — the device queue drains for another 6-12 frames after the submit clock is back).
    recovered = no contract violation before the stall, the recovery point lies within RECOVERY_FRAMES of it, and from there to the end
    the run meets the CONTRACT of an undisturbed run frame by frame — no back-pressure miss, every frame delivered within the deadline
    — and ends on its clock.  (Until the second session of round 6 the recovery point was the frame after the LAST frame with a
    submit backlog >= 0.5 ms anywhere in the run: one submit call that woke 0.6 ms late hundreds of frames after a completed recovery
    — host jitter that every undisturbed run tolerates, the frame still submitted before the next arrival and delivered in time —
    failed the run, and so did one such wake-up BEFORE the stall; it happened on two boxes of seven.  Such frames are now counted
    (`jitter_frames_after_recovery`, `jitter_frames_before_the_stall`); contract violations before the stall or after the recovery
    point still fail the run.)  `frames_to_recover_submit_clock` keeps the oldest measure."""
    import numpy as np
    backlog_s, late_s = np.asarray(backlog_s, dtype=float), np.asarray(late_s, dtype=float)
    lat = np.asarray(latency_ms, dtype=float)
    n = lat.size
    after = np.nonzero(backlog_s[stall_at + 1:] < 0.0005)[0]
    violation = (late_s > 0) | (np.nan_to_num(lat, nan=1e12) > DELIVERY_DEADLINE_MS)          # what an undisturbed run may not do either
    disturbed = (backlog_s >= 0.0005) | violation
    clean_before = not bool(violation[:stall_at].any())     # a contract violation before the stall is not the stall's: it fails the run (submit jitter is counted)
    calm = ~disturbed
    point = None                                           # first frame >= stall_at that starts 20 undisturbed frames
    run = 0
    for t in range(n - 1, stall_at - 1, -1):               # run = undisturbed frames from t on (capped by the end of the run)
        run = run + 1 if calm[t] else 0
        if run >= 20:
            point = t
    rec = None if point is None else point - stall_at
    tail = lat[point:] if point is not None else lat[:0]
    bad_after = int(violation[point:].sum()) if point is not None else None
    ok = bool(clean_before and point is not None and rec <= RECOVERY_FRAMES and bad_after == 0 and finished_behind_ms < 10.0)
    return {"stall_at_frame": int(stall_at), "frames_to_recover": rec, "frames_to_recover_submit_clock": int(after[0]) + 1 if after.size else None,
            "recovery_limit_frames": RECOVERY_FRAMES, "recovered": ok,
            "disturbed_frames": int(disturbed[stall_at:point].sum()) if point is not None else int(disturbed[stall_at:].sum()),
            "clean_frames_after_recovery": int(n - point) if point is not None else 0,
            "contract_violations_after_recovery": bad_after,
            "jitter_frames_after_recovery": int((disturbed[point:] & ~violation[point:]).sum()) if point is not None else None,
            "jitter_frames_before_the_stall": int((disturbed[:stall_at] & ~violation[:stall_at]).sum()),
            "delivery_p99_ms_after_recovery": round(float(np.percentile(tail, 99)), 4) if tail.size and not np.isnan(tail).any() else None,
            "clean_before_the_stall": clean_before}


def paced_realtime(api, synth, model, dev_index, B, nn_mode, seconds=10.0, ctx=None, sample_clock=True, stall=None):
    """The real-time contract itself (reference main.cpp:30-39: one 480-sample frame per stream every 10 ms), not an
    extrapolation from a mean: a frame of B streams arrives on the HOST every 10.000 ms for `seconds` and goes through the
    pipelined host entry points (pn_submit_host_i16: pinned buffers, copy-in / compute / copy-out on three streams, two
    frames in flight).  Frame t is submitted at its arrival time a_t = t0 + 10 ms x t (or as soon as the previous call
    returns, if that is later).  Two things are measured per frame:
      * back-pressure: the submit call for frame t returns after frame t + 1 has already arrived (`deadline_misses`);
      * ARRIVAL-TO-DELIVERY latency: between arrivals the loop polls pn_host_frames_delivered (event queries) and stamps
        every frame whose output copy has landed (resolution ~0.2 ms): `delivery_latency_ms` p50 / p99 / max.
    `met_contract`: no back-pressure miss, delivery p99 within DELIVERY_DEADLINE_MS (the pipeline is two frames deep: a
    frame must be out before the frame after next arrives), and the run did not end behind its clock.
    ctx: reuse an open context of B streams (reset first); otherwise one is created and closed here.
    stall = (frame, ms): a host hiccup is INJECTED — the submit of that frame is held back by `ms` — and the run's verdict is
    stall_verdict(): back on the clock, nothing missed and everything delivered in time after that.  A size whose back-to-back rate
    is below the arrival rate passes an undisturbed run and never recovers from a disturbed one.
    `host_pipeline_back_to_back_ms`: 100 frames submitted without pacing, the rate the pipeline sustains when it is behind."""
    import ctypes
    import numpy as np
    own = ctx is None
    if own:
        ctx = api.Context(model, B, device=dev_index, nn_mode=nn_mode)
    else:
        ctx.host_wait(); ctx.reset()
    L = ctx.L
    n = B * FRAME
    bufs = []
    try:
        src = synth.synth_batch(min(B, 64), 3, base_seed=synth.BASE_SEED + 31337)
        for k in range(3):
            hin, hout = L.pn_host_alloc(n * 2), L.pn_host_alloc(n * 2)
            if not hin or not hout:
                raise RuntimeError("pinned allocation failed")
            bufs.append((hin, hout))
            fr = np.ascontiguousarray(src[np.arange(B) % src.shape[0], k * FRAME:(k + 1) * FRAME])
            ctypes.memmove(hin, fr.ctypes.data, n * 2)
        for k in range(6):                                   # warm the pipeline (streams, staging buffers, clocks)
            ctx.submit_host_i16(*bufs[k % 3])
        ctx.host_wait()
        ser = []
        for k in range(3):                                   # the serial host call, for scale: copy in + the frame + copy out
            t_s = time.perf_counter()
            if L.pn_process_host_i16(ctx.h, bufs[k][0], bufs[k][1], None):
                raise RuntimeError("pn_process_host_i16 failed")
            ser.append(time.perf_counter() - t_s)
        t_s = time.perf_counter()
        for k in range(100):                                 # back to back: what the pipeline sustains once it is behind its clock
            ctx.submit_host_i16(*bufs[k % 3])
        ctx.host_wait()
        b2b_ms = (time.perf_counter() - t_s) * 10.0
        base = L.pn_host_frames_delivered(ctx.h)             # frames delivered before the paced loop starts
        if base < 0:
            raise RuntimeError("pn_host_frames_delivered failed")
        N = int(seconds * 100)
        period = 0.010
        arrive = np.empty(N); ret = np.empty(N); start = np.empty(N); deliv = np.full(N, np.nan)
        nd = 0

        def poll(now):
            nonlocal nd
            d = L.pn_host_frames_delivered(ctx.h) - base
            while nd < min(d, N):
                deliv[nd] = now; nd += 1

        with ClockSampler(period=0.5 if sample_clock else 1e9) as clk:
            t0 = time.perf_counter() + 0.002
            for t in range(N):
                a_t = t0 + period * t
                while True:                                  # poll deliveries while waiting for the arrival; sleep in 0.2 ms steps
                    now = time.perf_counter()
                    poll(now)
                    if now >= a_t:
                        break
                    if a_t - now > 0.0004:
                        time.sleep(0.0002)
                if stall is not None and t == stall[0]:
                    # the injected host hiccup: the SUBMIT path is held back for stall[1] ms.  Deliveries keep being stamped
                    # meanwhile (a frame that lands during the hiccup landed on time: with a blind poller frames t-1 / t-2 of
                    # every batch whose frame takes more than ~10 ms door to door were stamped 50 ms late — round 6's first
                    # capacity search failed every size above 49 152 on "not clean before the stall" for that reason)
                    t_res = time.perf_counter() + stall[1] * 1e-3
                    while True:
                        now = time.perf_counter()
                        poll(now)
                        if now >= t_res:
                            break
                        time.sleep(0.0002)
                arrive[t] = a_t; start[t] = now
                ctx.submit_host_i16(*bufs[t % 3])
                ret[t] = time.perf_counter()
                poll(ret[t])
            while nd < N and time.perf_counter() < t0 + period * N + 1.0:     # the last two frames
                poll(time.perf_counter()); time.sleep(0.0002)
            ctx.host_wait()
            t_end = time.perf_counter()
            poll(t_end)
        late = ret[:-1] - arrive[1:]                          # > 0: the call for frame t came back after frame t + 1 had arrived
        backlog = start - arrive                              # how far behind its arrival a frame was submitted
        lat_all = (deliv - arrive) * 1e3
        lat = lat_all[~np.isnan(lat_all)]
        out = {"streams": B, "seconds": round(t_end - t0, 3), "frames": N, "period_ms": 10.0,
               "deadline_misses": int((late > 0).sum()), "max_lateness_ms": round(float(max(late.max(), 0.0)) * 1e3, 4),
               "delivery_latency_ms": percentiles(lat) if lat.size else None, "delivery_deadline_ms": DELIVERY_DEADLINE_MS,
               "frames_delivered_late": int((lat > DELIVERY_DEADLINE_MS).sum()),
               "submit_call_ms": percentiles((ret - start) * 1e3), "submit_backlog_ms_max": round(float(backlog.max()) * 1e3, 4),
               "finished_behind_schedule_ms": round((t_end - (t0 + period * N)) * 1e3, 4),
               "serial_host_call_ms": round(1e3 * min(ser), 3), "host_pipeline_back_to_back_ms": round(b2b_ms, 4),
               "copy_streams": ctx.pipe_streams(),
               "path": "pn_submit_host_i16 (pinned host buffers, PCIe both ways inside the loop); delivery stamped by polling "
                       "pn_host_frames_delivered between arrivals (~0.2 ms resolution)"}
        out["met_contract"] = bool(out["deadline_misses"] == 0 and lat.size == N and out["delivery_latency_ms"]["p99"] <= DELIVERY_DEADLINE_MS
                                   and out["finished_behind_schedule_ms"] < 10.0)
        # A failed undisturbed run is a TRANSIENT when everything that went wrong is one burst the pipeline recovered from (a stall
        # of the box, not a batch that is too large: that one falls behind and stays behind).  A label — the strict figure forgives none.
        tr, burst = transient_burst(np.concatenate([late > 0, [False]]), lat_all, out["finished_behind_schedule_ms"])
        out["transient"] = bool(stall is None and not out["met_contract"] and tr)
        if burst:
            out["late_burst_frames"] = burst
        if stall is not None:
            rec = stall_verdict(backlog, np.concatenate([late, [0.0]]), lat_all, stall[0], out["finished_behind_schedule_ms"])
            rec["stall_ms"] = float(stall[1])
            out["recovery"] = rec
            out["met_contract"] = bool(rec["recovered"])
        out.update(clk.summary())
        return out
    finally:
        try:
            ctx.host_wait()
        except Exception:                     # noqa: BLE001
            pass
        for hin, hout in bufs:
            L.pn_host_free(hin); L.pn_host_free(hout)
        if own:
            ctx.close()


def _split(runs_of_size):
    und = [x for x in runs_of_size if not x.get("recovery") and "error" not in x]
    dis = [x for x in runs_of_size if x.get("recovery")]
    err = [x for x in runs_of_size if "error" in x]
    return und, dis, err


def size_strict_ok(runs_of_size, runs):
    """EVERY undisturbed run met the contract (nothing forgiven), at least `runs` of them, and the stall run recovered."""
    und, dis, err = _split(runs_of_size)
    return bool(not err and len(und) >= runs and all(x.get("met_contract") for x in und) and dis and all(x.get("met_contract") for x in dis))


def size_forgiven_ok(runs_of_size, runs):
    """The stall run recovered and the undisturbed runs all passed, OR exactly one of them failed as a transient (one recovered burst)
    and was answered by one extra run — runs + 1 undisturbed runs in all — with every other run clean."""
    und, dis, err = _split(runs_of_size)
    if err or not dis or not all(x.get("met_contract") for x in dis) or len(und) < runs:
        return False
    failed = [x for x in und if not x.get("met_contract")]
    return bool(not failed or (len(failed) == 1 and failed[0].get("transient") and len(und) == runs + 1))


def realtime_capacity(api, synth, model, dev_index, nn_mode, seconds, runs, log, soak_seconds=0.0, grid=DEFAULT_GRID,
                      fallback=DEFAULT_FALLBACK, paced=None):
    """Deadline-PROVEN capacity: the largest batch of `grid` that passes the STRICT rule (module docstring), found by bisection (a
    larger batch is never easier), falling back to `fallback` sizes when grid[0] itself fails; then confirmed by one run of
    `soak_seconds` (stepping down the grid while that fails).  The figure under the forgiving rule is reported beside it.
    paced: the paced-run function (tests inject a fake); default paced_realtime."""
    paced = paced or paced_realtime
    grid = list(grid)
    tried = {}

    def probe(b):
        res = []
        try:
            ctx = api.Context(model, b, device=dev_index, nn_mode=nn_mode)
        except Exception as e:                # noqa: BLE001
            tried[b] = [{"streams": b, "error": f"{type(e).__name__}: {e}"}]
            return False
        try:
            def one(**kw):
                try:
                    res.append(paced(api, synth, model, dev_index, b, nn_mode, seconds, ctx=ctx, **kw))
                except Exception as e:        # noqa: BLE001
                    res.append({"streams": b, "error": f"{type(e).__name__}: {e}"})
            for _ in range(runs):
                one()
            failed = [r for r in res if not r.get("met_contract")]
            if len(failed) == 1 and failed[0].get("transient"):
                one()                         # evidence for the FORGIVING figure only: one more undisturbed run, which must be clean
            one(stall=(STALL_AT_FRAME, STALL_MS))
        finally:
            ctx.close()
        tried[b] = res
        und, dis, _ = _split(res)
        n_fail = sum(not r.get("met_contract") for r in und)
        n_transient = sum(bool(r.get("transient")) for r in und)
        rec = (dis[-1].get("recovery") if dis else None) or {}
        log(f"[realtime] {b} streams: {len(und) - n_fail}/{len(und)} undisturbed runs met the contract"
            + (f" ({n_transient} transient: one recovered burst — NOT forgiven by the strict figure)" if n_transient else "")
            + f"; after a {STALL_MS:.0f} ms stall the pipeline is undisturbed again (submits on their arrivals, no back-pressure, deliveries in time) {rec.get('frames_to_recover')} "
            + f"frames on (submit clock back after {rec.get('frames_to_recover_submit_clock')}; limit {RECOVERY_FRAMES}); contract violations after that point: "
            + f"{rec.get('contract_violations_after_recovery')}, jitter frames {rec.get('jitter_frames_after_recovery')}; recovered={rec.get('recovered')}")
        return size_strict_ok(res, runs)

    best = None
    if probe(grid[0]):
        lo, hi = 0, len(grid)                 # grid[lo] passes; grid[hi] (if any) fails
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if probe(grid[mid]):
                lo = mid
            else:
                hi = mid
        best = grid[lo]
        nxt = grid[hi] if hi < len(grid) else None
    else:
        nxt = grid[0]
        for b in fallback:
            if probe(b):
                best = b
                break
    soaks = []
    while best is not None and soak_seconds > 0:
        try:
            r = paced(api, synth, model, dev_index, best, nn_mode, soak_seconds)
        except Exception as e:                # noqa: BLE001
            r = {"streams": best, "error": f"{type(e).__name__}: {e}"}
        soaks.append(r)
        log(f"[realtime] {best} streams, {soak_seconds:.0f} s confirmation run: {'met' if r.get('met_contract') else 'MISSED'} the contract")
        if r.get("met_contract"):
            break
        nxt, cand, best = best, best - 512, None          # the next smaller grid size that passes its own probe
        while cand >= grid[0] and len(soaks) < 4:
            if size_strict_ok(tried[cand], runs) if cand in tried else probe(cand):
                best = cand
                break
            cand -= 512

    def size_summary(r):
        und, dis, err = _split(r)
        return {"runs": len(und), "passed": sum(bool(x.get("met_contract")) for x in und),
                "transient_runs": sum(bool(x.get("transient")) for x in und),
                "strict": size_strict_ok(r, runs), "with_one_forgiven_burst": size_forgiven_ok(r, runs),
                "deadline_misses": [x.get("deadline_misses") for x in und],
                "delivery_latency_ms_p99": [(x.get("delivery_latency_ms") or {}).get("p99") for x in und],
                "host_pipeline_back_to_back_ms": [x.get("host_pipeline_back_to_back_ms") for x in r if "error" not in x],
                "stall_recovery": [x["recovery"] for x in dis], "errors": [x["error"] for x in err]}

    summary = {b: size_summary(r) for b, r in sorted(tried.items())}
    forgiven = [b for b, r in tried.items() if size_forgiven_ok(r, runs)]
    soak_failed = {r["streams"] for r in soaks if not r.get("met_contract")}
    forgiven = [b for b in forgiven if b not in soak_failed]
    return {"realtime_streams_p99": best,
            "realtime_streams_with_one_forgiven_burst": max(forgiven) if forgiven else None,
            "next_size": None if nxt is None or nxt not in summary else {"streams": nxt, "runs": summary[nxt]["runs"], "passed": summary[nxt]["passed"],
                                                                       "recovered": [x.get("recovered") for x in summary[nxt]["stall_recovery"]]},
            "sizes": {str(b): v for b, v in summary.items()},
            "paced_runs": [r for b in sorted(tried) for r in tried[b]],
            "confirmation_runs": soaks, "confirmation_seconds": soak_seconds,
            "runs_per_size": runs, "seconds_per_run": seconds, "grid": [grid[0], grid[-1], len(grid)],
            "contract": "one 480-sample frame per stream every 10 ms (reference src/main.cpp:30-39): frames arrive on the host on a 10.000 ms "
                        f"clock, pipelined host path with PCIe in the loop; a run passes with zero back-pressure misses, delivery p99 <= "
                        f"{DELIVERY_DEADLINE_MS} ms after arrival and no schedule slip; realtime_streams_p99 (STRICT): every undisturbed run of the size "
                        f"passes, none forgiven, AND the run with a {STALL_MS:.0f} ms host stall injected at frame {STALL_AT_FRAME} is undisturbed again "
                        f"(20 frames in a row: submits on their arrivals AND deliveries in time) within {RECOVERY_FRAMES} frames and violates the contract nowhere after that; realtime_streams_with_one_forgiven_burst: "
                        f"one transient run (every late frame inside one {TRANSIENT_SPAN_FRAMES}-frame burst the run recovered from) answered by a clean extra run"}


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--seconds", type=float, default=6.0, help="length of one paced run")
    ap.add_argument("--runs", type=int, default=2, help="undisturbed paced runs per size")
    ap.add_argument("--soak-seconds", type=float, default=20.0, help="confirmation run at the capacity found (0: none)")
    ap.add_argument("--grid", default="65536:69632:512", help="lo:hi:step of the sizes searched (hi inclusive)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default=None, help="write the full record here (default: gpurun_out/realtime_capacity.json)")
    a = ap.parse_args()
    from percepnet_amd import api, synth, weights
    lo, hi, st = (int(x) for x in a.grid.split(":"))
    model = api.Model(weights.default_blob(1234))
    log = lambda m: print(m, file=sys.stderr, flush=True)       # noqa: E731
    r = realtime_capacity(api, synth, model, a.device, api.NN_MFMA, a.seconds, a.runs, log, soak_seconds=a.soak_seconds,
                          grid=list(range(lo, hi + 1, st)))
    model.close()
    out = a.out or os.path.join(ROOT, "gpurun_out", "realtime_capacity.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(r, f)
    print(json.dumps({"realtime_streams_p99": r["realtime_streams_p99"],
                      "realtime_streams_with_one_forgiven_burst": r["realtime_streams_with_one_forgiven_burst"],
                      "next_size": r["next_size"], "sizes": {b: {k: v[k] for k in ("runs", "passed", "transient_runs", "strict")} for b, v in r["sizes"].items()},
                      "detail": out}))


if __name__ == "__main__":
    main()
