"""Per-stream lifecycle and shared weights of the batched C-ABI (-m gpu; round-3 verdict item 4).

The reference's unit is ONE stream: rnnoise_create / rnnoise_init zero one DenoiseState, rnnoise_destroy retires it
(denoise.cpp:252-280,326-331), and every state borrows the one static model (49-51,267).  A batched context advances B
streams in lock-step; pn_ctx_reset_streams re-initialises chosen slots on the device while the others keep running, and
the packed device weights are shared by every context of the same model content / device / network mode."""
import os
import subprocess
import time

import numpy as np
import pytest

from percepnet_amd import api, build, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(blob):
    m = api.Model(blob)
    yield m
    m.close()


@pytest.mark.parametrize("mode", [api.NN_STRICT, api.NN_MFMA, api.NN_MFMA_X3, api.NN_MFMA_F16], ids=["strict", "mfma", "x3", "f16"])
def test_slot_reset_at_frame_37_equals_a_fresh_stream_started_there(model, oracle, mode):
    """300 streams; at frame 37 (history ring slot 1 of 12, look-ahead slot 1 of 6, conv slots 2 of 5 / 1 of 3, odd GRU
    half) slots 5, 130 and 299 end their call and start a new one.  From there on those slots must produce exactly what a
    FRESH context produces for the new streams from its frame 0 (bit for bit: same kernel family, and in STRICT mode also
    the CPU oracle's bits), and every other slot must be bit-identical to a run in which nothing was reset."""
    B, T, T0 = 300, 60, 37
    ids = [5, 130, 299]
    pcm = synth.synth_batch(B, T)
    fresh = synth.synth_batch(len(ids), T - T0, first_stream=1000)          # the new calls
    feed = pcm.copy()
    for k, s in enumerate(ids):
        feed[s, T0 * 480:] = fresh[k]
    ref_ctx = api.Context(model, B, nn_mode=mode)
    ref = [ref_ctx.process_i16(pcm[:, t * 480:(t + 1) * 480]) for t in range(T)]
    ref_ctx.close()
    small = api.Context(model, len(ids), nn_mode=mode)
    exp = [small.process_i16(fresh[:, t * 480:(t + 1) * 480]) for t in range(T - T0)]
    small.close()
    ctx = api.Context(model, B, nn_mode=mode)
    others = np.setdiff1d(np.arange(B), ids)
    for t in range(T):
        if t == T0:
            ctx.reset_streams(ids)
        out, gr = ctx.process_i16(feed[:, t * 480:(t + 1) * 480])
        assert np.array_equal(out[others], ref[t][0][others]), t
        assert np.array_equal(gr[others].view(np.uint32), ref[t][1][others].view(np.uint32)), t
        if t >= T0:
            assert np.array_equal(out[ids], exp[t - T0][0]), t
            assert np.array_equal(gr[ids].view(np.uint32), exp[t - T0][1].view(np.uint32)), t
    ctx.close()
    if mode == api.NN_STRICT:
        for k in range(len(ids)):
            ro, rg = oracle.run_pcm(fresh[k])
            got = np.concatenate([exp[t][0][k] for t in range(1, T - T0)])
            assert np.array_equal(got, ro) and np.array_equal(np.stack([exp[t][1][k] for t in range(T - T0)]), rg)


def test_reset_streams_argument_checks_and_pipelined_path(model):
    """Out-of-range ids are refused; on the pipelined host path the reset lands between the frames submitted before and
    after it (results equal the synchronous path)."""
    import ctypes
    B, T = 6, 9
    pcm = synth.synth_batch(B, T)
    ctx = api.Context(model, B, nn_mode=api.NN_STRICT)
    with pytest.raises(api.PercepNetError):
        ctx.reset_streams([0, B])
    with pytest.raises(api.PercepNetError):
        ctx.reset_streams([-1])
    ctx.reset_streams([])
    sync = []
    for t in range(T):
        if t == 4:
            ctx.reset_streams([1, 1, 4])
        sync.append(ctx.process_i16(pcm[:, t * 480:(t + 1) * 480])[0])
    ctx.reset()
    L = ctx.L
    n = B * 480
    bufs = [(L.pn_host_alloc(n * 2), L.pn_host_alloc(n * 2)) for _ in range(T)]
    for t in range(T):
        fr = np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480])          # (kept alive across the memmove)
        ctypes.memmove(bufs[t][0], fr.ctypes.data, n * 2)
        if t == 4:
            ctx.reset_streams([1, 4])
        ctx.submit_host_i16(bufs[t][0], bufs[t][1])
    ctx.host_wait()
    for t in range(T):
        got = np.ctypeslib.as_array(ctypes.cast(bufs[t][1], ctypes.POINTER(ctypes.c_int16)), shape=(B, 480))
        assert np.array_equal(got, sync[t]), t
    for a, b in bufs:
        L.pn_host_free(a); L.pn_host_free(b)
    ctx.close()


def test_contexts_of_one_model_share_the_device_weights(blob):
    """The reference binds every state to one static model (zero copies).  Here the first context of a (model content,
    device, mode) uploads and re-packs the 32 MB once; every later one — also through a second pn_model handle holding
    the same arrays — points at that copy: its footprint excludes the weights and it is created faster."""
    # a model content no other test uses (one weight bit flipped): a context another test left open cannot hold its copies
    blob = bytes(blob[:-4]) + bytes([blob[-4] ^ 1]) + bytes(blob[-3:])
    m1, m2 = api.Model(blob), api.Model(blob)
    t0 = time.perf_counter(); c1 = api.Context(m1, 64); t1 = time.perf_counter()
    c2 = api.Context(m2, 64); t2 = time.perf_counter()
    assert c1.describe()["weights"] == "own" and c2.describe()["weights"] == "shared"
    assert c1.device_bytes() - c2.device_bytes() > 30 << 20            # 31.85 MB of weights + biases
    print(f"\ncreate: first context {1e3 * (t1 - t0):.0f} ms (self-tests, re-pack, upload), second {1e3 * (t2 - t1):.0f} ms")
    c3 = api.Context(m1, 64, nn_mode=api.NN_STRICT)                    # another mode: another layout, its own copy
    assert c3.describe()["weights"] == "own"
    pcm = synth.synth_batch(64, 3)
    a = [c1.process_i16(pcm[:, t * 480:(t + 1) * 480])[0] for t in range(3)]
    c1.close()                                                         # the copy outlives its creator while c2 uses it
    c1b = api.Context(m1, 64)
    assert c1b.describe()["weights"] == "shared"
    for c in (c2, c1b):
        for t in range(3):
            assert np.array_equal(c.process_i16(pcm[:, t * 480:(t + 1) * 480])[0], a[t])
    c2.close(); c1b.close(); c3.close()
    c4 = api.Context(m2, 64)                                           # last user gone: the copy was freed, this one is new
    assert c4.describe()["weights"] == "own"
    c4.close(); m1.close(); m2.close()


def test_cli_slot_queue_reuses_slots_of_finished_pairs(blob, oracle, tmp_path):
    """percepnet_run --slots 2 with five recordings of different lengths (one shorter than a frame): two concurrent streams,
    a finished pair's slot is re-initialised on the device and taken over by the next waiting pair.  Every output equals
    the reference CLI's for that file alone (STRICT: bit for bit)."""
    exe = build.RUN
    (tmp_path / "m.pnw").write_bytes(blob)
    lens = [9, 4, 17, 0, 6]
    ins = [synth.synth_stream(40 + i, n) if n else np.zeros(100, np.int16) for i, n in enumerate(lens)]
    args = []
    for i, x in enumerate(ins):
        (tmp_path / f"i{i}.pcm").write_bytes(x.tobytes()); args += [f"i{i}.pcm", f"o{i}.pcm"]
    r = subprocess.run([exe, "--model", "m.pnw", "--strict", "--slots", "2"] + args, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for i, x in enumerate(ins):
        got = np.fromfile(tmp_path / f"o{i}.pcm", np.int16)
        if lens[i] == 0:
            assert got.size == 0
        else:
            assert np.array_equal(got, oracle.run_pcm(x)[0]), i


@pytest.mark.parametrize("mode", [api.NN_STRICT, api.NN_MFMA, api.NN_MFMA_X3, api.NN_MFMA_F16], ids=["strict", "mfma", "x3", "f16"])
def test_active_set_a_stream_that_skips_ticks_equals_a_stream_fed_only_its_frames(model, oracle, mode):
    """Round-4 verdict item 5.  In the reference a stream's state advances only when ITS rnnoise_process_frame is called
    (src/denoise.cpp:508-547, src/rnnoise.h:60).  300 streams; stream 17 receives no frame at ticks 10, 11 and 40, stream
    200 none at tick 11, stream 299 (the last row) none at ticks 0 and 59 — every ring phase of the context is crossed
    (history 12, look-ahead 6, conv 5 / 3, GRU 2).  A stream that skipped must produce, on the ticks it takes part in,
    exactly what the same kernels produce for a stream that was simply fed those frames back to back (bit for bit, every
    mode; STRICT: also the CPU oracle's bits; MFMA: within 1 LSB of it); its output rows are untouched on the ticks it
    skips; every other stream is bit-identical to a run without any skipping."""
    import torch
    B, T = 300, 60
    skips = {17: {10, 11, 40}, 200: {11}, 299: {0, 59}}
    pcm = synth.synth_batch(B, T)
    dev = torch.device("cuda:0")
    ref_ctx = api.Context(model, B, nn_mode=mode)
    ref = [ref_ctx.process_i16(pcm[:, t * 480:(t + 1) * 480]) for t in range(T)]
    ref_ctx.close()
    # what each skipping stream should produce: the frames it receives, back to back, through the same kernels
    comp = {}
    ks = sorted(skips)
    packed = np.zeros((len(ks), T * 480), np.int16)
    for k, s in enumerate(ks):
        got = [t for t in range(T) if t not in skips[s]]
        for j, t in enumerate(got):
            packed[k, j * 480:(j + 1) * 480] = pcm[s, t * 480:(t + 1) * 480]
        comp[s] = got
    small = api.Context(model, len(ks), nn_mode=mode)
    exp = [small.process_i16(packed[:, j * 480:(j + 1) * 480]) for j in range(T)]
    small.close()
    ctx = api.Context(model, B, nn_mode=mode)
    d_out = torch.full((B, 480), 12345, dtype=torch.int16, device=dev)
    d_gr = torch.full((B, 68), -7.0, dtype=torch.float32, device=dev)
    others = np.setdiff1d(np.arange(B), ks)
    seen = {s: 0 for s in ks}
    prev_out, prev_gr = d_out.cpu().numpy(), d_gr.cpu().numpy()
    for t in range(T):
        frame = pcm[:, t * 480:(t + 1) * 480].copy()
        active = [s for s in range(B) if not (s in skips and t in skips[s])]
        for s in ks:
            if t in skips[s]:
                frame[s] = 31000          # a skipped row's input must not matter
        d_in = torch.from_numpy(np.ascontiguousarray(frame)).to(dev)
        torch.cuda.synchronize()
        if t % 2:                          # any order
            active = active[::-1]
        ctx.process_i16_active_dev(d_in.data_ptr(), d_out.data_ptr(), d_gr.data_ptr(), active)
        ctx.synchronize()
        out, gr = d_out.cpu().numpy(), d_gr.cpu().numpy()
        assert np.array_equal(out[others], ref[t][0][others]), t
        assert np.array_equal(gr[others].view(np.uint32), ref[t][1][others].view(np.uint32)), t
        for k, s in enumerate(ks):
            if t in skips[s]:
                assert np.array_equal(out[s], prev_out[s]) and np.array_equal(gr[s].view(np.uint32), prev_gr[s].view(np.uint32)), (t, s)
            else:
                j = seen[s]; seen[s] += 1
                assert np.array_equal(out[s], exp[j][0][k]), (t, s)
                assert np.array_equal(gr[s].view(np.uint32), exp[j][1][k].view(np.uint32)), (t, s)
        prev_out, prev_gr = out, gr
    assert ctx.L.pn_ctx_frames_done(ctx.h) == T
    ctx.close()
    if mode in (api.NN_STRICT, api.NN_MFMA):
        for k, s in enumerate(ks):
            n = len(comp[s])
            ro, rg = oracle.run_pcm(packed[k, :n * 480])
            got = np.concatenate([exp[j][0][k] for j in range(1, n)])
            d = np.abs(got.astype(np.int32) - ro.astype(np.int32)).max()
            assert d <= (0 if mode == api.NN_STRICT else 1), (s, d)


def test_active_set_argument_checks_and_all_active_is_the_plain_call(model):
    import torch
    B = 9
    pcm = synth.synth_batch(B, 4)
    dev = torch.device("cuda:0")
    a, b = api.Context(model, B, nn_mode=api.NN_STRICT), api.Context(model, B, nn_mode=api.NN_STRICT)
    o1 = torch.zeros((B, 480), dtype=torch.int16, device=dev); o2 = torch.zeros_like(o1)
    for t in range(4):
        d_in = torch.from_numpy(np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480])).to(dev)
        torch.cuda.synchronize()
        a.process_i16_dev(d_in.data_ptr(), o1.data_ptr(), None)
        b.process_i16_active_dev(d_in.data_ptr(), o2.data_ptr(), None, np.arange(B)[::-1])
        a.synchronize(); b.synchronize()
        assert torch.equal(o1, o2)
    for bad in ([0, B], [-1], [3, 3]):
        with pytest.raises(api.PercepNetError):
            b.process_i16_active_dev(d_in.data_ptr(), o2.data_ptr(), None, bad)
    b.process_i16_active_dev(d_in.data_ptr(), o2.data_ptr(), None, [])      # nobody: a tick in which nothing happens
    b.synchronize()
    assert torch.equal(o1, o2)
    a.close(); b.close()


@pytest.mark.parametrize("mode", [api.NN_MFMA, api.NN_MFMA_X3], ids=["mfma", "x3"])
def test_a_refused_launch_fails_the_frame(model, mode):
    """Round-4 verdict item 8: a launcher that refuses its geometry used to return silently and the frame completed with
    stale layer outputs and rc 0.  With the test hook on, every entry point must fail and name the launcher."""
    B = 64
    pcm = synth.synth_batch(B, 3)
    ctx = api.Context(model, B, nn_mode=mode)
    ctx.process_i16(pcm[:, :480])
    ctx.debug_inject_launch_failure(True)
    with pytest.raises(api.PercepNetError, match="pn_launch_dense"):
        ctx.process_i16(pcm[:, 480:960])
    with pytest.raises(api.PercepNetError, match="K-tiles"):
        ctx.compute_rnn(np.zeros((B, 70), np.float32))
    ctx.debug_inject_launch_failure(False)
    ctx.reset()
    good = api.Context(model, B, nn_mode=mode)
    for t in range(3):
        assert np.array_equal(ctx.process_i16(pcm[:, t * 480:(t + 1) * 480])[0], good.process_i16(pcm[:, t * 480:(t + 1) * 480])[0])
    ctx.close(); good.close()


def test_a_failed_active_frame_leaves_the_skipped_streams_alone(model):
    """Advisor (round 5): when the frame inside pn_process_*_active is refused, the streams the call SKIPS must still get their
    output rows and in-place state back (the header promises those rows are untouched).  After the failure the hook is switched
    off and the same tick is run again: every stream — the skipped ones included, which skipped twice — continues bit-identically
    to a context that never saw the failure."""
    import torch
    B, T = 64, 8
    pcm = synth.synth_batch(B, T)
    dev = torch.device("cuda:0")
    act = [s for s in range(B) if s % 5]
    skipped = [s for s in range(B) if s % 5 == 0]
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
    good = api.Context(model, B, nn_mode=api.NN_MFMA)
    o1 = torch.zeros((B, 480), dtype=torch.int16, device=dev); o2 = torch.zeros_like(o1)
    g1 = torch.zeros((B, 68), dtype=torch.float32, device=dev); g2 = torch.zeros_like(g1)
    fr = [torch.from_numpy(np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480])).to(dev) for t in range(T)]
    torch.cuda.synchronize()
    for t in range(3):
        ctx.process_i16_dev(fr[t].data_ptr(), o1.data_ptr(), g1.data_ptr()); good.process_i16_dev(fr[t].data_ptr(), o2.data_ptr(), g2.data_ptr())
    ctx.synchronize(); good.synchronize()
    before_o, before_g = o1.clone(), g1.clone()
    ctx.debug_inject_launch_failure(True)
    with pytest.raises(api.PercepNetError, match="pn_launch_dense"):
        ctx.process_i16_active_dev(fr[3].data_ptr(), o1.data_ptr(), g1.data_ptr(), act)
    ctx.synchronize()
    assert torch.equal(o1[skipped], before_o[skipped]) and torch.equal(g1[skipped], before_g[skipped])
    ctx.debug_inject_launch_failure(False)
    for t in range(3, T):
        ids = act if t == 3 else list(range(B))
        ctx.process_i16_active_dev(fr[t].data_ptr(), o1.data_ptr(), g1.data_ptr(), ids)
        good.process_i16_active_dev(fr[t].data_ptr(), o2.data_ptr(), g2.data_ptr(), ids)
        ctx.synchronize(); good.synchronize()
        assert torch.equal(o1, o2) and torch.equal(g1, g2), t
    ctx.close(); good.close()


def test_active_set_on_the_pipelined_host_path(model):
    """pn_submit_host_i16_active = pn_process_i16_active behind the pipelined copy-in / compute / copy-out path: the rows of
    the streams that take part equal the device-pointer path's bit for bit, also after a stream has skipped ticks; a refused
    id list consumes no pipeline slot."""
    import ctypes
    import torch
    B, T = 64, 14
    skips = {5: {4, 5}, 63: {9}}
    pcm = synth.synth_batch(B, T)
    dev = torch.device("cuda:0")
    ref = api.Context(model, B, nn_mode=api.NN_STRICT)
    d_out = torch.zeros((B, 480), dtype=torch.int16, device=dev)
    want = []
    for t in range(T):
        act = [s for s in range(B) if not (s in skips and t in skips[s])]
        d_in = torch.from_numpy(np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480])).to(dev)
        torch.cuda.synchronize()
        ref.process_i16_active_dev(d_in.data_ptr(), d_out.data_ptr(), None, act)
        ref.synchronize()
        want.append((act, d_out.cpu().numpy().copy()))
    ref.close()
    ctx = api.Context(model, B, nn_mode=api.NN_STRICT)
    L = ctx.L
    n = B * 480
    bufs = [(L.pn_host_alloc(n * 2), L.pn_host_alloc(n * 2)) for _ in range(T)]
    with pytest.raises(api.PercepNetError):
        ctx.submit_host_i16_active(bufs[0][0], bufs[0][1], None, [0, 0])
    for t in range(T):
        fr = np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480])
        ctypes.memmove(bufs[t][0], fr.ctypes.data, n * 2)
        ctx.submit_host_i16_active(bufs[t][0], bufs[t][1], None, want[t][0])
    ctx.host_wait()
    assert L.pn_host_frames_delivered(ctx.h) == T
    for t in range(T):
        got = np.ctypeslib.as_array(ctypes.cast(bufs[t][1], ctypes.POINTER(ctypes.c_int16)), shape=(B, 480))
        act = want[t][0]
        assert np.array_equal(got[act], want[t][1][act]), t
    for a, b in bufs:
        L.pn_host_free(a); L.pn_host_free(b)
    ctx.close()
