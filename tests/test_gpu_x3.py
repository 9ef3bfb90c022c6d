"""Split-precision network mode (api.NN_MFMA_X3: every fp32 GEMM operand carried as an fp16 (hi, lo) pair, three
v_mfma_f32_32x32x16_f16 products per operand pair, fp32 accumulation/state/activations; pn_nn_x3.hip).

The mode claims the SAME parity bounds as the fp32 MFMA mode — north_star's +-1 LSB PCM, and 2e-5 on the g/r taps — so
every test here uses test_gpu_parity / test_gpu_longrun's PCM_TOL_LSB = 1 and GR_TOL = 2e-5, not a re-stated tolerance:
  * 40 streams x 100 frames and ragged batch sizes against the CPU oracle;
  * placement invariance over grid-stride rounds (8195 streams of 7 kinds, bit-identical replicas);
  * 1024 streams x 1000 frames (10 s), every stream against the oracle (the horizon SURVEY 8(d) asks for);
  * weight sets with 2-3x the default dynamic range and biased gates: free-running over 10 s, and ONE network step
    from identical state against the STRICT (reference-order, bit-exact) kernels, with the fp32 MFMA mode's bounds;
  * operands outside the fp16 range: a weight beyond +-65504 is refused at context creation, never silently clipped.
"""
import numpy as np
import pytest

from percepnet_amd import api, synth, weights
from test_gpu_longrun import run_long, shared_stream, compare, _record
from test_gpu_parity import PCM_TOL_LSB, GR_TOL, _oracle_batch
import test_gpu_stress_weights as stress

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(blob):
    m = api.Model(blob)
    yield m
    m.close()


def test_x3_mode_within_one_lsb(model, oracle):
    B, T = 40, 100
    pcm = synth.synth_batch(B, T)
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA_X3)
    assert ctx.describe()["nn"] == "mfma_x3"
    out, gr = ctx.run_pcm(pcm)
    ro, rg = _oracle_batch(oracle, pcm)
    d = np.abs(out.astype(np.int32) - ro.astype(np.int32))
    assert d.max() <= PCM_TOL_LSB, d.max()
    assert np.abs(gr - rg).max() <= GR_TOL, np.abs(gr - rg).max()
    ctx.close()


@pytest.mark.parametrize("B", [1, 129, 300, 513])
def test_x3_ragged_batch_sizes(model, oracle, B):
    """Batch sizes that leave the last 256-row block ragged (1 row, 129 = one row into the second 128-row operand chunk,
    300, 513): every stream against the oracle."""
    T = 12
    pcm = synth.synth_batch(min(B, 16), T)
    pcm = pcm[np.arange(B) % pcm.shape[0]]
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA_X3)
    out, gr = ctx.run_pcm(pcm)
    ro, rg = _oracle_batch(oracle, pcm[:16])
    idx = np.arange(B) % min(B, 16)
    assert np.abs(out.astype(np.int32) - ro[idx].astype(np.int32)).max() <= PCM_TOL_LSB
    assert np.abs(gr - rg[idx]).max() <= GR_TOL
    ctx.close()


def test_x3_placement_invariance_across_grid_stride_rounds(model, oracle):
    B, K, T = 8195, 7, 14
    base = synth.synth_batch(K, T, first_stream=1)
    idx = np.arange(B) % K
    pcm = base[idx]
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA_X3)
    ro, rg, rf, rs = oracle.run_batch(base)
    for t in range(T):
        frame = np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480])
        out, gr = ctx.process_i16(frame)
        assert np.abs(gr[:K] - rg[:, t]).max() <= GR_TOL, (t, "g/r vs oracle")
        if t > 0:
            assert np.abs(out[:K].astype(np.int32) - ro[:, (t - 1) * 480:t * 480].astype(np.int32)).max() <= PCM_TOL_LSB, t
        for k in range(K):
            m = idx == k
            g = gr[m].view(np.uint32)
            assert (g == g[0]).all(), (t, k, "g/r")
            assert (out[m] == out[m][0]).all(), (t, k, "pcm")
    ctx.close()


def test_x3_1024_streams_1000_frames_every_stream_vs_oracle(model, oracle):
    import torch
    B, T = 1024, 1000
    dev = torch.device("cuda:0")
    pcm = synth.synth_batch_parallel(B, T)
    ref = oracle.run_batch(pcm, group=8)
    ts = shared_stream(dev)
    with torch.cuda.stream(ts):
        d_pcm = torch.from_numpy(pcm).to(dev)
        ctx = api.Context(model, B, nn_mode=api.NN_MFMA_X3, stream=ts.cuda_stream)
        got = run_long(ctx, lambda t: d_pcm[:, t * 480:(t + 1) * 480].contiguous(), T, None, dev)
        ctx.close()
    compare("x3_1024x1000", got, ref, {
        "config": "split-precision network mode (fp16 hi/lo operand planes, 3 MFMA products, fp32 accumulate), 1024 streams x 1000 frames, every stream vs the CPU oracle"})


@pytest.mark.parametrize("name", ["scale2", "scale3", "scale2_gate_biased", "scale6_saturating"])
def test_x3_mode_on_other_weight_sets(name):
    """The stress sets of test_gpu_stress_weights.py with the fp32 MFMA mode's bounds (ONE_STEP_TOL, BOUNDS)."""
    import torch
    from oracle.oracle import Oracle
    B, T = 256, 1000
    blob = weights.pack_blob(stress.SETS[name]())
    pcm = synth.synth_batch_parallel(B, T, first_stream=700)
    ro, rg, rf, rs = Oracle(blob).run_batch(pcm, group=8)
    dev = torch.device("cuda:0")
    model = api.Model(blob)
    ts = shared_stream(dev)
    with torch.cuda.stream(ts):
        d_pcm = torch.from_numpy(pcm).to(dev)
        ctx = api.Context(model, B, nn_mode=api.NN_MFMA_X3, stream=ts.cuda_stream)
        out, gr, feat, sil = run_long(ctx, lambda t: d_pcm[:, t * 480:(t + 1) * 480].contiguous(), T, None, dev)
        ctx.close()
    TS = 200
    cx = api.Context(model, B, nn_mode=api.NN_MFMA_X3)
    cs = api.Context(model, B, nn_mode=api.NN_STRICT)
    one = np.zeros(TS)
    for t in range(TS):
        cx.set_rnn_state(cs.get_rnn_state())                # re-derives the hi/lo operand planes from the fp32 state
        gs_, gx_ = cs.compute_rnn(rf[:, t]), cx.compute_rnn(rf[:, t])
        assert np.isfinite(gx_).all(), t
        one[t] = np.abs(gs_ - gx_).max()
    cx.close(); cs.close(); model.close()
    d = np.abs(out.astype(np.int32) - ro.astype(np.int32))
    d = np.minimum(d, 65536 - d)
    dg = np.abs(gr - rg)
    by_s = [float(dg[:, k:k + 100].max()) for k in range(0, T, 100)]
    _record(f"x3_stress_{name}", {
        "mode": "split precision (NN_MFMA_X3)", "weights": name, "streams": B, "frames": T,
        "one_step_from_identical_state_max_abs_delta_gr": float(one.max()),
        "max_abs_delta_pcm_lsb_circular": int(d.max()),
        "pcm_delta_histogram_lsb": np.bincount(np.minimum(d, 16).ravel().astype(np.int64), minlength=17).tolist(),
        "max_abs_delta_gr": float(dg.max()), "mean_abs_delta_gr": float(dg.mean()),
        "p9999_abs_delta_gr": float(np.quantile(dg, 0.9999)), "max_abs_delta_gr_by_second": by_s})
    assert np.isfinite(gr).all()
    assert np.array_equal(feat.view(np.uint32), rf.view(np.uint32)) and np.array_equal(sil, rs)
    assert one.max() <= stress.ONE_STEP_TOL[name], float(one.max())
    assert max(by_s) <= 8 * max(by_s[0], 1e-6) + 1e-6, by_s
    lsb, tol = stress.BOUNDS[name]
    if lsb is not None:
        assert d.max() <= lsb, int(d.max())
    assert dg.max() <= tol, float(dg.max())


@pytest.mark.parametrize("mode", [api.NN_MFMA_X3, api.NN_MFMA_F16], ids=["x3", "f16"])
def test_x3_rnn_state_roundtrip(model, oracle, mode):
    """get/set of the RNNState arrays (nnet_data.h:28-38) in the modes that keep operand shadows (split precision: hi/lo
    planes; fp16 operands: fp16 copies): loading a state re-derives the shadows from the fp32 values, so a context
    restored from another's state continues bit-identically to it."""
    B, T = 130, 6
    rng = np.random.default_rng(3)
    feats = rng.standard_normal((T, B, 70)).astype(np.float32)
    a = api.Context(model, B, nn_mode=mode)
    for t in range(3):
        a.compute_rnn(feats[t])
    st = a.get_rnn_state()
    b = api.Context(model, B, nn_mode=mode)
    for t in range(3):
        b.compute_rnn(feats[5 - t])                          # a different history, then overwritten
    # the ring phase (which slot is "oldest") is part of the context, not of the state: bring b to the same frame count
    b.set_rnn_state(st)
    for t in range(3, T):
        ga, gb = a.compute_rnn(feats[t]), b.compute_rnn(feats[t])
        assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32)), t
    a.close(); b.close()


def test_x3_refuses_weights_outside_the_fp16_range():
    lay = weights.random_layers(5)
    w = lay["conv2"]["input_weights"].copy()
    w.reshape(-1)[17] = 1e5
    lay["conv2"]["input_weights"] = w
    m = api.Model(weights.pack_blob(lay))
    with pytest.raises(api.PercepNetError, match="fp16 range"):
        api.Context(m, 64, nn_mode=api.NN_MFMA_X3)
    c = api.Context(m, 64, nn_mode=api.NN_MFMA)              # the fp32 modes take the same model
    c.close(); m.close()


def test_x3_through_the_cli_and_the_rnnoise_entry_points(blob, oracle, tmp_path):
    """`percepnet_run --x3` (three ragged pairs) and the relinked reference main.cpp with PERCEPNET_X3=1 (one stream through
    rnnoise_create / rnnoise_process_frame): the fp32 mode's bounds against the oracle."""
    import os
    import subprocess
    from percepnet_amd import build
    (tmp_path / "m.pnw").write_bytes(blob)
    ins = [synth.synth_stream(40 + i, 15 + 4 * i) for i in range(3)]
    args = []
    for i, x in enumerate(ins):
        (tmp_path / f"i{i}.pcm").write_bytes(x.tobytes()); args += [f"i{i}.pcm", f"o{i}.pcm"]
    r = subprocess.run([build.RUN, "--model", "m.pnw", "--x3"] + args, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for i, x in enumerate(ins):
        got = np.fromfile(tmp_path / f"o{i}.pcm", np.int16)
        ref = oracle.run_pcm(x)[0]
        assert got.size == ref.size and np.abs(got.astype(np.int32) - ref.astype(np.int32)).max() <= PCM_TOL_LSB, i
    exe = os.path.join(os.path.dirname(build.RUN), "percepNet_run_relinked")
    if os.path.exists(exe):
        env = dict(os.environ, PERCEPNET_MODEL=str(tmp_path / "m.pnw"), PERCEPNET_X3="1", PERCEPNET_SELFTEST="2")
        r = subprocess.run([exe, "i0.pcm", "r0.pcm"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert "nn_mode 3" in r.stderr, r.stderr                   # the self-test line names the mode the handle really runs
        got = np.fromfile(tmp_path / "r0.pcm", np.int16)
        ref = oracle.run_pcm(ins[0])[0]
        assert np.abs(got.astype(np.int32) - ref.astype(np.int32)).max() <= PCM_TOL_LSB
        tap = np.fromfile(tmp_path / "feature_test.raw", np.float32).reshape(-1, 68)
        assert np.abs(tap - oracle.run_pcm(ins[0])[1]).max() <= GR_TOL


def test_x3_row_group_instantiations_agree_bit_for_bit(model):
    """The split-precision kernels exist with 32 and with 64 rows per wave (chosen from the batch size at context creation:
    64 from 32 768 streams).  Every output's MFMA sequence is the same in both, so the same stream must give the same bits
    in a 300-stream context (32 rows per wave) and in a 32 768-stream one (64 rows per wave)."""
    K, T = 300, 8
    base = synth.synth_batch(16, T)
    small = api.Context(model, K, nn_mode=api.NN_MFMA_X3)
    big = api.Context(model, 32768, nn_mode=api.NN_MFMA_X3)
    assert small.describe()["gru"] == "x3_rows32" and big.describe()["gru"] == "x3_rows64"
    ps, pb = base[np.arange(K) % 16], base[np.arange(32768) % 16]
    for t in range(T):
        os_, gs = small.process_i16(ps[:, t * 480:(t + 1) * 480])
        ob, gb = big.process_i16(pb[:, t * 480:(t + 1) * 480])
        assert np.array_equal(gs.view(np.uint32), gb[:K].view(np.uint32)), t
        assert np.array_equal(os_, ob[:K]), t
        assert np.array_equal(gb[32768 - 16:].view(np.uint32), gb[:16].view(np.uint32)), t      # last block = first block (same streams)
    small.close(); big.close()


@pytest.mark.parametrize("mode", [api.NN_MFMA_X3, api.NN_MFMA_F16], ids=["x3", "f16"])
def test_paired_phase_gru_kernel_agrees_bit_for_bit_with_the_one_tile_per_block_kernel(model, mode, monkeypatch):
    """PERCEPNET_X3_RG=3 runs the GRUs of a large batch on pn_gru_x3p_kernel (one persistent 8-wave block per CU: the K loop of
    one wave group beside the gating epilogue of the other, which also stages the K group's weight tiles; opt-in, DESIGN.md
    4.2f).  Same MFMAs in the same order and the same separately rounded gating arithmetic as pn_gru_x3_kernel, so the two
    must agree bit for bit — on a batch whose M tiles do not divide
    evenly over the 8 XCDs and the 64 tile walkers per XCD (33 025 streams = 130 M tiles, the last one ragged), and with
    the recurrent state carried over several frames."""
    B, K, T = 33025, 16, 6
    base = synth.synth_batch(K, T)
    pb = base[np.arange(B) % K]
    outs = {}
    for rg in ("2", "3"):
        monkeypatch.setenv("PERCEPNET_X3_RG", rg)
        ctx = api.Context(model, B, nn_mode=mode)
        assert ctx.describe()["gru"].endswith("rows64_paired" if rg == "3" else "rows64"), ctx.describe()
        outs[rg] = [ctx.process_i16(pb[:, t * 480:(t + 1) * 480]) for t in range(T)]
        ctx.close()
    for t in range(T):
        (o2, g2), (o3, g3) = outs["2"][t], outs["3"][t]
        assert np.isfinite(g3).all()
        assert np.array_equal(g2.view(np.uint32), g3.view(np.uint32)), t
        assert np.array_equal(o2, o3), t


@pytest.mark.parametrize("name", ["default", "scale3"])
def test_split_precision_is_as_close_to_exact_arithmetic_as_the_cpu_reference(name, blob):
    """The yardstick that does not depend on anyone's summation order: ONE network step from identical state evaluated in
    float64 (tools/nn_f64_model.py: the reference's formulas, every product and sum in double).  The CPU reference's own
    sequential fp32 accumulation is off by up to ~3e-6 (rms 5e-8 .. 4e-7) from that; the split-precision mode must be in
    the same class — its max error no more than 2x the reference's own, its rms no more than 2.5x — while the fp16-operand
    mode, which really does compute at reduced precision, is more than 10x further away (so the yardstick discriminates)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import x3_vs_f64
    lay = weights.unpack_blob(blob) if name == "default" else stress.SETS[name]()
    e = x3_vs_f64.measure(lay, B=96, T=40)
    _record(f"modes_vs_float64_{name}", e)
    cpu, mf, x3, f16 = e["strict (= CPU reference)"], e["fp32 MFMA"], e["split precision"], e["fp16 operands"]
    assert x3["max"] <= 2.0 * cpu["max"] and x3["rms"] <= 2.5 * cpu["rms"], (x3, cpu)
    assert mf["max"] <= 2.0 * cpu["max"] and mf["rms"] <= 1.5 * cpu["rms"], (mf, cpu)
    assert f16["rms"] >= 10 * x3["rms"], (f16, x3)


@pytest.mark.parametrize("mode", [api.NN_MFMA_X3, api.NN_MFMA_F16], ids=["x3", "f16"])
def test_shadow_operand_kernels_ragged_last_block_at_64_rows_per_wave(model, mode):
    """32 768 + 129 streams: the 64-rows-per-wave instantiation (256-row blocks) with a ragged last block whose second
    128-row operand chunk holds a single stream.  Every replica of a stream must be bit-identical wherever it sits — the
    last, ragged block included — and equal to the same stream in a small context (32 rows per wave)."""
    B, K, T = 32768 + 129, 16, 6
    base = synth.synth_batch(K, T)
    big = api.Context(model, B, nn_mode=mode)
    small = api.Context(model, K, nn_mode=mode)
    assert big.describe()["gru"].endswith("rows64") and small.describe()["gru"].endswith("rows32")
    idx = np.arange(B) % K
    pb = base[idx]
    for t in range(T):
        ob, gb = big.process_i16(pb[:, t * 480:(t + 1) * 480])
        os_, gs = small.process_i16(base[:, t * 480:(t + 1) * 480])
        assert np.isfinite(gb).all()
        for k in range(K):
            m = idx == k
            g = gb[m].view(np.uint32)
            assert (g == g[0]).all(), (t, k, "g/r replicas")
            assert (ob[m] == ob[m][0]).all(), (t, k, "pcm replicas")
        assert np.array_equal(gb[:K].view(np.uint32), gs.view(np.uint32)), t
        assert np.array_equal(ob[:K], os_), t
    big.close(); small.close()


def test_x3_65536_streams_1000_frames_256_sampled_vs_oracle(model, oracle):
    """The benchmarked configuration of the split-precision mode (65 536 concurrent streams: 64 rows per wave, 256-row
    blocks, 16 blocks per CU and launch) over the 10 s horizon: 256 distinct sampled streams scattered through the batch
    against the oracle, the other slots carrying rotated replicas as in bench.py — same layout and bounds as
    test_gpu_longrun.py::test_configs2_65536_streams_1000_frames_256_sampled_vs_oracle."""
    import torch
    from test_gpu_longrun import sample_layout
    B, T, P = 65536, 1000, 256
    dev = torch.device("cuda:0")
    pool = synth.synth_batch_parallel(P, T)
    ref = oracle.run_batch(pool, group=8)
    slots, idx, rot = sample_layout(B, P)
    ts = shared_stream(dev)
    with torch.cuda.stream(ts):
        d_pool = torch.from_numpy(pool).to(dev)
        d_idx = torch.from_numpy(idx).to(dev)
        ar = (torch.arange(480, device=dev)[None, :] + torch.from_numpy(rot).to(dev)[:, None]) % 480
        rows = torch.from_numpy(slots).to(dev)

        def frame_of(t):
            return torch.gather(d_pool[:, t * 480:(t + 1) * 480][d_idx], 1, ar).contiguous()

        ctx = api.Context(model, B, nn_mode=api.NN_MFMA_X3, stream=ts.cuda_stream)
        assert ctx.describe()["gru"] == "x3_rows64"
        got = run_long(ctx, frame_of, T, rows, dev)
        ctx.close()
    compare("x3_65536x1000_sample256", got, ref, {
        "config": "split-precision network mode, 65536 concurrent streams, 256 distinct sampled streams vs the CPU oracle"})


def test_shadow_operand_kernels_never_read_past_their_buffers():
    """PERCEPNET_GUARD=1 puts 1 MB of 0xFF (NaN as fp32 and fp16) behind every device buffer: an operand fetched past the
    end of a shadow, a packed weight array or a state buffer (clamped prefetch tiles, ragged last blocks, padding rows)
    would surface as a NaN or a large error.  33 025 streams = the 64-rows-per-wave kernels with a ragged last block."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PERCEPNET_GUARD="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "x3_check.py"), "33025", "4"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"x3 vs strict: max\|dPCM\| (\d+) LSB .* max\|dg,r\| ([0-9.e+-]+), .* finite (\w+)", r.stdout)
    assert m, r.stdout[-1000:]
    assert int(m.group(1)) <= PCM_TOL_LSB and float(m.group(2)) <= GR_TOL and m.group(3) == "True", m.group(0)


@pytest.mark.parametrize("mode", [api.NN_MFMA_X3, api.NN_MFMA_F16], ids=["x3", "f16"])
def test_operand_saturation_is_counted_not_silent(blob, mode, monkeypatch):
    """The shadow-operand modes carry GEMM operands as fp16 (pairs): an activation beyond +-65504 is CLAMPED when it
    becomes an operand (weights beyond it are refused at creation).  conv1 is a ReLU and unbounded (rnn_train.py:107), so
    a model can drive it there: with PERCEPNET_X3_SATCOUNT=1 the library counts every clamped operand value and
    pn_ctx_describe reports the total — zero for the default model, positive (with finite outputs, no NaN) for a model
    whose conv1 bias is 1e5."""
    monkeypatch.setenv("PERCEPNET_X3_SATCOUNT", "1")
    pcm = synth.synth_batch(8, 4)
    m0 = api.Model(blob)
    c0 = api.Context(m0, 8, nn_mode=mode)
    for t in range(4):
        c0.process_i16(pcm[:, t * 480:(t + 1) * 480])
    assert c0.describe()["x3_saturated"] == "0"
    c0.close(); m0.close()
    lay = weights.unpack_blob(blob)
    lay["conv1"]["bias"] = np.full_like(lay["conv1"]["bias"], 1e5)          # ReLU(1e5 + ...) > 65504 in every conv1 output
    m1 = api.Model(weights.pack_blob(lay))
    c1 = api.Context(m1, 8, nn_mode=mode)
    for t in range(4):
        out, gr = c1.process_i16(pcm[:, t * 480:(t + 1) * 480])
        assert np.isfinite(gr).all()
    n = int(c1.describe()["x3_saturated"])
    assert n >= 4 * 8 * 512, n                                              # conv1's 512 outputs of 8 streams, every frame
    c1.close(); m1.close()
    monkeypatch.delenv("PERCEPNET_X3_SATCOUNT")
    c2 = api.Context(api.Model(blob), 8, nn_mode=mode)
    assert "x3_saturated" not in c2.describe()                              # the counter is a debugging aid, off by default
    c2.close()
