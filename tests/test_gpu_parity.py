"""GPU parity tests proper (-m gpu): the HIP path, called through the C-ABI, against the CPU
oracle on identical seeded inputs, against the committed golden vectors, and — at full batch
sizes — through size-independent properties.

Tolerances (BASELINE.json north_star):
  * STRICT network mode: bit-exact PCM, g/r tap and features (integer/bit equality).
  * MFMA network mode (fused multiply-add chain instead of mul+add): PCM within +-1 LSB per sample,
    g/r within 2e-5 absolute; features (network-independent) still bit-exact.
"""
import ctypes
import os

import numpy as np
import pytest

from percepnet_amd import api, synth, weights

pytestmark = pytest.mark.gpu

PCM_TOL_LSB = 1
GR_TOL = 2e-5
# BASELINE configs[4] (fp16 GEMM operands, fp32 accumulation): north_star asks for the tolerance to be
# re-stated.  Measured on MI355X over 24 streams x 100 frames: max 3 LSB (loud streams), 1 LSB on
# normal-level streams, 8% of samples differ at all; g/r max 2.3e-4, mean 3.5e-5.
F16_PCM_TOL_LSB = 4
F16_GR_TOL = 1e-3


@pytest.fixture(scope="module")
def model(blob):
    m = api.Model(blob)
    yield m
    m.close()


def _oracle_batch(oracle, pcm):
    outs, grs = [], []
    for s in range(pcm.shape[0]):
        o, g = oracle.run_pcm(pcm[s])
        outs.append(o); grs.append(g)
    return np.stack(outs), np.stack(grs)


def test_strict_mode_bit_exact_pcm_and_taps(model, oracle):
    """8 streams incl. the burst/silence stream (7), 40 frames: everything bit-identical."""
    B, T = 8, 40
    pcm = synth.synth_batch(B, T)
    ctx = api.Context(model, B, nn_mode=api.NN_STRICT)
    out, gr = ctx.run_pcm(pcm)
    ro, rg = _oracle_batch(oracle, pcm)
    assert np.array_equal(out, ro)
    assert np.array_equal(gr, rg)
    ctx.close()


def test_hip_path_straight_against_the_compiled_reference(model, blob):
    """ONE hop: the HIP path against the reference ITSELF (oracle/_ref/libpercepnet_ref.so = the untouched sources of
    /root/reference/src compiled in place by oracle/Makefile; the .so travels to the GPU box), not against the restatement:
    rnnoise_process_frame as percepNet_run drives it (denoise.cpp:508-547, main.cpp:30-39).  8 streams x 100 frames — voiced, loud
    (non-silent branch), burst + digital silence, two-tone.  STRICT: bit-equal PCM and g,r.  MFMA: <= 1 LSB, |dg,r| <= 2e-5."""
    from oracle import oracle as orc
    if not orc.ref_available():
        pytest.skip("oracle/_ref/libpercepnet_ref.so did not travel to this box")
    ref = orc.Reference(blob)
    streams = (0, 1, 3, 7, 13, 23, 27, 33)
    pcm = np.stack([synth.synth_stream(s, 100) for s in streams])
    assert {synth.stream_kind(s) for s in streams} == {"voiced", "loud", "bursts", "twotone"}
    ro, rg = zip(*(ref.run_pcm(pcm[i]) for i in range(len(streams))))
    ro, rg = np.stack(ro), np.stack(rg)
    ctx = api.Context(model, len(streams), nn_mode=api.NN_STRICT)
    out, gr = ctx.run_pcm(pcm)
    ctx.close()
    assert np.array_equal(out, ro)
    assert np.array_equal(gr.view(np.uint32), rg.view(np.uint32))
    ctx = api.Context(model, len(streams), nn_mode=api.NN_MFMA)
    out, gr = ctx.run_pcm(pcm)
    ctx.close()
    d = np.abs(out.astype(np.int32) - ro.astype(np.int32))
    assert d.max() <= PCM_TOL_LSB, d.max()
    assert np.abs(gr - rg).max() <= GR_TOL
    assert np.abs(ro.astype(np.int32)).max() > 1000 and (ro[2] != 0).any()       # the loud stream is not silence at the output


def test_row_range_chains_bit_identical(model, oracle, monkeypatch):
    """Round-5 verdict item 3: on one in-order stream every layer of the batch GEMMs pays whole rounds of 512 co-resident blocks
    (65 536 streams = 16 rounds, 66 048 = 17).  Large fp32 contexts now run the network as row-range chains on streams of their own
    (pn_context.cpp launch_rnn; default 2 chains above 16 384 streams).  8192 + 300 streams, forced to 1, 2 and 3 chains (ragged
    last range: 8492 = 4352 + 4140 = 2944 + 2944 + 2604 rows): bit-identical PCM and g,r for every stream, also through the
    network-only entry point, and the last rows still follow the oracle."""
    B, T = 8192 + 300, 5
    pool = synth.synth_batch(64, T)
    pcm = pool[np.arange(B) % 64].copy()
    pcm[8192:] = np.roll(pcm[8192:], 11, axis=1)                    # the last rows are not copies of earlier rows
    feat = np.random.default_rng(3).standard_normal((B, 70)).astype(np.float32)
    outs = []
    for chains in ("1", "2", "3"):
        monkeypatch.setenv("PN_NN_CHAINS", chains)
        ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
        want = chains if chains != "3" else "2"                     # three chains of 2831 rows would be the small-batch regime: capped
        assert ctx.describe()["nn_chains"].split(":")[0] == want, ctx.describe()
        o, g = ctx.run_pcm(pcm)
        outs.append((o, g, ctx.compute_rnn(feat)))
        ctx.close()
    for o, g, r in outs[1:]:
        assert np.array_equal(outs[0][0], o) and np.array_equal(outs[0][1].view(np.uint32), g.view(np.uint32))
        assert np.array_equal(outs[0][2].view(np.uint32), r.view(np.uint32))
    for s in (4351, 4352, 8192, B - 1):
        ro, rg = oracle.run_pcm(pcm[s])
        assert np.abs(outs[1][0][s].astype(np.int32) - ro.astype(np.int32)).max() <= PCM_TOL_LSB
        assert np.abs(outs[1][1][s] - rg).max() <= GR_TOL


def test_postfilter_option(model, oracle):
    """the optional envelope post-filter (reference post_filtering, denoise.cpp:216-250) between
    the g/r tap and pitch_filter.  The network and its tap are untouched (bit-identical, STRICT); the PCM follows the
    oracle's post-filtered output within 1 LSB — the warped gain uses sinf, libm's on the CPU and OCML's on the GPU,
    both within 1 ULP of sin but not of each other — and really differs from the unfiltered output.  Switching the
    stage off again gives back the plain pipeline from the next frame (the synthesis overlap carries one frame)."""
    B, T = 8, 40
    pcm = synth.synth_batch(B, T)
    ctx = api.Context(model, B, nn_mode=api.NN_STRICT)
    ctx.set_postfilter(True)
    out, gr = ctx.run_pcm(pcm)
    ro = np.stack([oracle.run_pcm(pcm[s], postfilter=True)[0] for s in range(B)])
    plain, rg = _oracle_batch(oracle, pcm)
    assert np.array_equal(gr, rg)
    d = np.abs(out.astype(np.int32) - ro.astype(np.int32))
    assert d.max() <= PCM_TOL_LSB, d.max()
    assert (d != 0).mean() < 0.01
    assert np.abs(ro.astype(np.int32) - plain.astype(np.int32)).max() > 50      # the stage does something
    ctx.set_postfilter(False)
    ctx.reset()
    out2, _ = ctx.run_pcm(pcm)
    assert np.array_equal(out2, plain)
    ctx.close()


def test_golden_vectors_strict(model, golden_dir):
    """The committed outputs of the compiled reference (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(golden_dir, "pcm_golden.npz"))
    pcm = np.stack([g["in_0"], g["in_3"], g["in_7"], g["in_13"]])
    ctx = api.Context(model, 4, nn_mode=api.NN_STRICT)
    out, gr = ctx.run_pcm(pcm)
    for i, s in enumerate((0, 3, 7, 13)):
        assert np.array_equal(out[i], g[f"out_{s}"]), s
        assert np.array_equal(gr[i], g[f"gr_{s}"]), s
    ctx.close()
    # float in / float out convention of rnnoise_process_frame
    ctx = api.Context(model, 1, nn_mode=api.NN_STRICT)
    x = g["fin_0"].reshape(-1, 480)
    y = np.concatenate([ctx.process_f32(x[t])[0][0] for t in range(x.shape[0])])
    assert np.array_equal(y, g["fout_0"])
    ctx.reset()
    # samples at 4x the nominal range through the float API: nearly all frames non-silent
    x = g["fin_3"].reshape(-1, 480)
    res = [ctx.process_f32(x[t]) for t in range(x.shape[0])]
    assert np.array_equal(np.concatenate([r[0][0] for r in res]), g["fout_3"])
    assert np.array_equal(np.stack([r[1][0] for r in res]), g["fgr_3"])
    ctx.close()


def test_golden_vectors_mfma(model, golden_dir):
    g = np.load(os.path.join(golden_dir, "pcm_golden.npz"))
    pcm = np.stack([g["in_0"], g["in_3"], g["in_7"], g["in_13"]])
    ctx = api.Context(model, 4, nn_mode=api.NN_MFMA)
    out, gr = ctx.run_pcm(pcm)
    for i, s in enumerate((0, 3, 7, 13)):
        d = np.abs(out[i].astype(np.int32) - g[f"out_{s}"].astype(np.int32)).max()
        assert d <= PCM_TOL_LSB, (s, d)
        assert np.abs(gr[i] - g[f"gr_{s}"]).max() <= GR_TOL
    ctx.close()


def test_features_and_silence_bit_exact(model, oracle):
    """DSP front end alone: the 70 features and the silence decision never depend on the network
    and must match bit for bit in either mode (stream 3 is loud enough to be non-silent, 7 has
    digital silence, 13 pitch ambiguity)."""
    streams = [0, 3, 7, 13, 27, 33]
    T = 60
    pcm = np.stack([synth.synth_stream(s, T) for s in streams])
    ctx = api.Context(model, len(streams), nn_mode=api.NN_MFMA)
    feats = np.zeros((len(streams), T, 70), np.float32)
    sils = np.zeros((len(streams), T), np.int32)
    for t in range(T):
        ctx.process_i16(pcm[:, t * 480:(t + 1) * 480], want_gr=False)
        feats[:, t], sils[:, t] = ctx.read_features()
    for i, s in enumerate(streams):
        rf, rs = oracle.features(pcm[i].astype(np.float32) / np.float32(32768))
        assert np.array_equal(feats[i], rf), s
        assert np.array_equal(sils[i], rs), s
    assert 0 < sils[1].sum() < T and sils[2].all()   # both branches of `silence` were exercised
    ctx.close()


def test_mfma_mode_within_one_lsb(model, oracle):
    """configs[1]-style batch slice: 40 streams x 100 frames (1 s of audio each)."""
    B, T = 40, 100
    pcm = synth.synth_batch(B, T)
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
    out, gr = ctx.run_pcm(pcm)
    ro, rg = _oracle_batch(oracle, pcm)
    d = np.abs(out.astype(np.int32) - ro.astype(np.int32))
    assert d.max() <= PCM_TOL_LSB, d.max()
    assert np.abs(gr - rg).max() <= GR_TOL, np.abs(gr - rg).max()
    ctx.close()


def test_fp16_variant_tolerance(model, oracle):
    """configs[4]: fp16 weights/activations as MFMA operands, fp32 accumulate/state/DSP."""
    B, T = 24, 100
    pcm = synth.synth_batch(B, T)
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA_F16)
    out, gr = ctx.run_pcm(pcm)
    ro, rg = _oracle_batch(oracle, pcm)
    d = np.abs(out.astype(np.int32) - ro.astype(np.int32))
    assert d.max() <= F16_PCM_TOL_LSB, d.max()
    assert np.abs(gr - rg).max() <= F16_GR_TOL, np.abs(gr - rg).max()
    assert np.abs(gr - rg).mean() <= 1e-4
    feat, _ = ctx.read_features()            # the DSP front end does not depend on the network mode
    assert np.isfinite(feat).all()
    ctx.close()


def test_network_alone_random_features(model, oracle):
    """compute_rnn on random features with carried state, 25 steps; strict exact, MFMA 2e-5."""
    B, T = 5, 25
    rng = np.random.default_rng(11)
    feats = (rng.standard_normal((T, B, 70)) * 0.7).astype(np.float32)
    ref = np.zeros((T, B, 68), np.float32)
    for b in range(B):
        st = oracle.lib.pno_create(oracle.model)
        for t in range(T):
            g = np.zeros(34, np.float32); r = np.zeros(34, np.float32)
            f = np.ascontiguousarray(feats[t, b])
            oracle.lib.pno_compute_rnn(st, g.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                       r.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                       f.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
            ref[t, b, :34] = g; ref[t, b, 34:] = r
        oracle.lib.pno_destroy(st)
    for mode, tol in ((api.NN_STRICT, 0.0), (api.NN_MFMA, GR_TOL)):
        ctx = api.Context(model, B, nn_mode=mode)
        got = np.stack([ctx.compute_rnn(feats[t]) for t in range(T)])
        assert np.abs(got - ref).max() <= tol, (mode, np.abs(got - ref).max())
        ctx.close()


def test_saturating_weights_table_clamp(oracle):
    """Weights scaled x6 drive the tanh table index into its clamp (vec.h:62-63)."""
    from oracle.oracle import Oracle
    sat = weights.pack_blob(weights.random_layers(3, scale=6.0))
    o2 = Oracle(sat)
    m2 = api.Model(sat)
    pcm = synth.synth_batch(3, 30, first_stream=4)
    ctx = api.Context(m2, 3, nn_mode=api.NN_STRICT)
    out, gr = ctx.run_pcm(pcm)
    ro, rg = _oracle_batch(o2, pcm)
    assert np.array_equal(out, ro) and np.array_equal(gr, rg)
    assert rg.min() < 0.02 and rg.max() > 0.98
    ctx.close(); m2.close()


@pytest.mark.parametrize("B", [1, 129, 300])
def test_ragged_batch_sizes(model, oracle, B):
    """Batches that do not fill the 128-row GEMM tiles / the DSP grid: 1, 129, 300 streams."""
    T = 8
    base = synth.synth_batch(6, T)
    pcm = base[np.arange(B) % 6]
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
    out, gr = ctx.run_pcm(pcm)
    ro, rg = _oracle_batch(oracle, base)
    for s in range(B):
        assert np.abs(out[s].astype(np.int32) - ro[s % 6].astype(np.int32)).max() <= PCM_TOL_LSB
        assert np.abs(gr[s] - rg[s % 6]).max() <= GR_TOL
    ctx.close()


def test_zero_input_gives_zero_output_and_reset(model):
    B = 4
    ctx = api.Context(model, B)
    z = np.zeros((B, 480), np.int16)
    for _ in range(8):
        out, gr = ctx.process_i16(z)
        assert not out.any()
    _, sil = ctx.read_features()
    assert sil.all()
    pcm = synth.synth_batch(B, 6)
    a, _ = ctx.run_pcm(pcm)
    ctx.reset()
    b, _ = ctx.run_pcm(pcm)       # reset() restores the all-zero initial state
    ctx2 = api.Context(model, B)
    c, _ = ctx2.run_pcm(pcm)
    assert np.array_equal(b, c)
    ctx.close(); ctx2.close()


def test_full_size_properties_65536_streams(model, oracle):
    """BASELINE configs[2] size.  Properties that do not need the oracle at full size: streams are
    independent and tile/XCD placement never changes a result (identical input => bit-identical
    output wherever the stream sits), plus an oracle check on the 16 distinct inputs."""
    import torch
    B, T, P = 65536, 10, 16         # the pipeline delays audio by 5 frames: output frames 5.. carry signal
    base = synth.synth_batch(P, T)
    idx = np.arange(B) % P
    dev = torch.device("cuda:0")
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA, stream=torch.cuda.current_stream().cuda_stream)
    outs = []
    for t in range(T):
        fr = torch.from_numpy(base[:, t * 480:(t + 1) * 480][idx]).to(dev)
        o = torch.empty_like(fr)
        g = torch.empty((B, 68), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()                # the context runs on its own stream: inputs complete first
        ctx.process_i16_dev(fr.data_ptr(), o.data_ptr(), g.data_ptr())
        torch.cuda.synchronize()
        o = o.cpu().numpy(); g = g.cpu().numpy()
        assert np.array_equal(o, o[:P][idx]), "identical streams diverged across tiles"
        assert np.array_equal(g, g[:P][idx])
        outs.append((o[:P], g[:P]))
    ro, rg = _oracle_batch(oracle, base)
    assert np.abs(ro[:, 5 * 480:]).max() > 1000          # the compared PCM is real signal, not the pipeline's zeros
    for t in range(1, T):
        d = np.abs(outs[t][0].astype(np.int32) - ro[:, (t - 1) * 480:t * 480].astype(np.int32)).max()
        assert d <= PCM_TOL_LSB
        assert np.abs(outs[t][1] - rg[:, t]).max() <= GR_TOL
    ctx.close()


def test_rnnoise_drop_in_symbols(blob, oracle, tmp_path):
    """The reference's own entry points (mangled C++ names of rnnoise.h:49-68), used the way
    main.cpp:30-39 uses them, with the model loaded through rnnoise_model_from_file."""
    L = ctypes.CDLL(api.LIB_PATH)
    libc = ctypes.CDLL(None)
    libc.fopen.restype = ctypes.c_void_p
    libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [ctypes.c_void_p]
    path = tmp_path / "model.pnw"
    path.write_bytes(blob)
    f = libc.fopen(str(path).encode(), b"rb")
    mff = getattr(L, "_Z23rnnoise_model_from_fileP8_IO_FILE"); mff.restype = ctypes.c_void_p; mff.argtypes = [ctypes.c_void_p]
    rm = mff(f); libc.fclose(f)
    assert rm
    create = getattr(L, "_Z14rnnoise_createP8RNNModel"); create.restype = ctypes.c_void_p; create.argtypes = [ctypes.c_void_p]
    proc = getattr(L, "_Z21rnnoise_process_frameP12DenoiseStatePfPKfP8_IO_FILE")
    proc.restype = ctypes.c_float; proc.argtypes = [ctypes.c_void_p] * 4
    destroy = getattr(L, "_Z15rnnoise_destroyP12DenoiseState"); destroy.argtypes = [ctypes.c_void_p]
    st = create(rm)
    assert st
    tap = tmp_path / "feature_test.raw"
    ff = libc.fopen(str(tap).encode(), b"wb")
    pcm = synth.synth_stream(0, 10)
    outs = []
    for t in range(10):
        x = (pcm[t * 480:(t + 1) * 480].astype(np.float32) / np.float32(32768)).copy()
        assert proc(st, x.ctypes.data, x.ctypes.data, ff) == 0.0      # in-place, like main.cpp:35
        outs.append((x * 32768).astype(np.int32).astype(np.int16))
    libc.fclose(ff); destroy(st)
    getattr(L, "_Z18rnnoise_model_freeP8RNNModel")(ctypes.c_void_p(rm))
    ro, rg = oracle.run_pcm(pcm)
    got = np.concatenate(outs[1:])
    assert np.abs(got.astype(np.int32) - ro.astype(np.int32)).max() <= PCM_TOL_LSB
    tapdata = np.fromfile(tap, np.float32).reshape(10, 68)
    assert np.abs(tapdata - rg).max() <= GR_TOL


def test_cli_percepnet_run_matches_reference_cli_contract(blob, oracle, tmp_path):
    """percepnet_run (csrc/percepnet_run.cpp) = the reference's `percepNet_run in.pcm out.pcm`
    (main.cpp:11-44): raw int16 in, (frames-1)*480 samples out, ./feature_test.raw with 68 floats per
    frame for a single pair; N pairs = N concurrent streams of different lengths."""
    import subprocess
    from percepnet_amd import build
    exe = build.RUN
    assert os.path.exists(exe), "percepnet_run not built"
    (tmp_path / "m.pnw").write_bytes(blob)
    a = synth.synth_stream(3, 20); b = synth.synth_stream(5, 13)
    b_tail = np.concatenate([b, np.zeros(100, np.int16)])          # partial tail frame must be dropped
    (tmp_path / "a.pcm").write_bytes(a.tobytes()); (tmp_path / "b.pcm").write_bytes(b_tail.tobytes())
    r = subprocess.run([exe, "--model", "m.pnw", "a.pcm", "ao.pcm"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ao = np.fromfile(tmp_path / "ao.pcm", np.int16); tap = np.fromfile(tmp_path / "feature_test.raw", np.float32).reshape(-1, 68)
    ro, rg = oracle.run_pcm(a)
    assert ao.size == 19 * 480 and np.abs(ao.astype(np.int32) - ro.astype(np.int32)).max() <= PCM_TOL_LSB
    assert tap.shape == (20, 68) and np.abs(tap - rg).max() <= GR_TOL
    r = subprocess.run([exe, "--model", "m.pnw", "--strict", "a.pcm", "ao2.pcm", "b.pcm", "bo2.pcm"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ao2 = np.fromfile(tmp_path / "ao2.pcm", np.int16); bo2 = np.fromfile(tmp_path / "bo2.pcm", np.int16)
    rb, _ = oracle.run_pcm(b)
    assert np.array_equal(ao2, ro) and np.array_equal(bo2, rb)      # strict mode: bit-exact, ragged lengths
    r = subprocess.run([exe, "a.pcm"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    # multi-device host: --devices shards the pairs over one host thread + one context per listed device (the same
    # device twice on this 1-GPU box: two threads, two contexts, ragged shards of 3 + 2 streams of different lengths)
    ins = [synth.synth_stream(20 + i, 9 + 3 * i) for i in range(5)]
    args = []
    for i, x in enumerate(ins):
        (tmp_path / f"i{i}.pcm").write_bytes(x.tobytes()); args += [f"i{i}.pcm", f"o{i}.pcm"]
    r = subprocess.run([exe, "--model", "m.pnw", "--strict", "--verbose", "--devices", "0,0"] + args, cwd=tmp_path, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    # every device thread places itself on its GPU's NUMA node before it allocates (or says why it could not): one line each
    assert r.stderr.count("percepnet_run: device 0 (") == 2, r.stderr
    assert all(("thread bound to the" in l) or ("NUMA" in l) or ("unreadable" in l) or ("cpulist" in l) or ("affinity" in l)
               for l in r.stderr.splitlines() if l.startswith("percepnet_run: device")), r.stderr
    for i, x in enumerate(ins):
        assert np.array_equal(np.fromfile(tmp_path / f"o{i}.pcm", np.int16), oracle.run_pcm(x)[0]), i
    for bad in ("0,7", "0,,0", "0,", "x"):     # not a visible ordinal / an empty element (never read as device 0) / not a number
        r = subprocess.run([exe, "--model", "m.pnw", "--devices", bad] + args, cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and "--devices" in r.stderr, (bad, r.stderr)
    r = subprocess.run([exe, "--model", "m.pnw", "--device", "7"] + args, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "device 7" in r.stderr and "out of range" in r.stderr       # a bad shard is reported, not hidden


def test_multi_frame_device_api(model, oracle):
    """pn_process_i16_multi: n_frames frame-major [T][B][480] in one call == T single calls."""
    import torch
    B, T = 5, 6
    pcm = synth.synth_batch(B, T, first_stream=20)
    fm = np.ascontiguousarray(pcm.reshape(B, T, 480).transpose(1, 0, 2))          # [T][B][480]
    dev = torch.device("cuda:0")
    ctx = api.Context(model, B, nn_mode=api.NN_STRICT)
    d_in = torch.from_numpy(fm).to(dev); d_out = torch.empty_like(d_in)
    d_gr = torch.empty((T, B, 68), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    rc = ctx.L.pn_process_i16_multi(ctx.h, d_in.data_ptr(), d_out.data_ptr(), d_gr.data_ptr(), T)
    assert rc == 0
    ctx.synchronize()
    out = d_out.cpu().numpy(); gr = d_gr.cpu().numpy()
    ro, rg = _oracle_batch(oracle, pcm)
    for s in range(B):
        assert np.array_equal(out[1:, s].reshape(-1), ro[s])
        assert np.array_equal(gr[:, s], rg[s])
    ctx.close()


@pytest.mark.parametrize("mode", ["mfma", "f16"])
def test_placement_invariance_across_grid_stride_rounds(model, oracle, mode):
    """8195 streams built from 7 distinct ones, 14 frames (the history ring wraps): the front-end kernel's 256 blocks
    take several grid-stride rounds, the last one ragged.  Identical streams must produce bit-identical features,
    spectra-derived gains and PCM wherever they sit — catches anything that leaks from one round (or one
    stream group of a wave) into the next.  Holds for the fp16 network variant too (same arithmetic in every tile)."""
    B, K, T = 8195, 7, 14
    base = synth.synth_batch(K, T, first_stream=1)
    idx = np.arange(B) % K
    pcm = base[idx]
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA if mode == "mfma" else api.NN_MFMA_F16)
    ro, rg, rf, rs = oracle.run_batch(base)      # ... and the 7 base streams must also be the ORACLE's
    pcm_tol, gr_tol = (PCM_TOL_LSB, GR_TOL) if mode == "mfma" else (F16_PCM_TOL_LSB, F16_GR_TOL)
    for t in range(T):
        frame = np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480])
        out = np.empty_like(frame)
        gr = np.empty((B, 68), np.float32)
        assert ctx.L.pn_process_host_i16(ctx.h, frame.ctypes.data, out.ctypes.data, gr.ctypes.data) == 0
        feat, sil = ctx.read_features()
        assert np.array_equal(feat[:K].view(np.uint32), rf[:, t].view(np.uint32)), (t, "features vs oracle")
        assert np.array_equal(sil[:K], rs[:, t]), (t, "silence vs oracle")
        assert np.abs(gr[:K] - rg[:, t]).max() <= gr_tol, (t, "g/r vs oracle")
        if t > 0:
            assert np.abs(out[:K].astype(np.int32) - ro[:, (t - 1) * 480:t * 480].astype(np.int32)).max() <= pcm_tol, (t, "pcm vs oracle")
        for k in range(K):
            m = idx == k
            f = feat[m].view(np.uint32)
            assert (f == f[0]).all(), (t, k, "features")
            assert (sil[m] == sil[m][0]).all(), (t, k, "silence")
            g = gr[m].view(np.uint32)
            assert (g == g[0]).all(), (t, k, "g/r")
            assert (out[m] == out[m][0]).all(), (t, k, "pcm")
    ctx.close()


def test_pipelined_host_path_matches_synchronous(model):
    """pn_submit_host_i16 / pn_host_wait (copy-in, launches and copy-out of consecutive frames overlapped on three
    streams, double-buffered staging) == pn_process_host_i16 frame by frame, bit for bit; the families may be mixed."""
    import torch
    B, T = 37, 14
    pcm = synth.synth_batch(B, T + 1, first_stream=2)
    frames = [np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480]) for t in range(T + 1)]
    ref_ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
    ref = [ref_ctx.process_i16(f) for f in frames]
    ref_ctx.close()
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
    h_in = [torch.from_numpy(f).pin_memory() for f in frames[:T]]
    h_out = [torch.empty((B, 480), dtype=torch.int16).pin_memory() for _ in range(T)]
    h_gr = [torch.empty((B, 68), dtype=torch.float32).pin_memory() for _ in range(T)]
    for t in range(T):
        ctx.submit_host_i16(h_in[t].data_ptr(), h_out[t].data_ptr(), h_gr[t].data_ptr() if t % 3 else None)
    ctx.host_wait()
    for t in range(T):
        assert np.array_equal(h_out[t].numpy(), ref[t][0]), t
        if t % 3:
            assert np.array_equal(h_gr[t].numpy().view(np.uint32), ref[t][1].view(np.uint32)), t
    o, g = ctx.process_i16(frames[T])                      # synchronous call after pipelined ones: same state
    assert np.array_equal(o, ref[T][0]) and np.array_equal(g.view(np.uint32), ref[T][1].view(np.uint32))
    ctx.reset()                                            # and the pipeline restarts cleanly after a reset
    ctx.submit_host_i16(h_in[0].data_ptr(), h_out[0].data_ptr(), None)
    ctx.host_wait()
    assert np.array_equal(h_out[0].numpy(), ref[0][0])
    ctx.close()


def _zero_layers():
    return {name: {"bias": np.zeros(6 * nn_ if kind == weights.KIND_GRU else nn_, np.float32),
                   "input_weights": np.zeros(nin * ks * nn_ * (3 if kind == weights.KIND_GRU else 1), np.float32),
                   **({"recurrent_weights": np.zeros(nn_ * 3 * nn_, np.float32)} if kind == weights.KIND_GRU else {})}
            for name, kind, nin, nn_, ks, act in weights.LAYERS}


@pytest.mark.parametrize("mode", [api.NN_STRICT, api.NN_MFMA])
def test_reference_known_answers_through_the_hip_kernels(golden_dir, mode):
    """The reference's own gtest vectors (tests/testnnet.cpp:19-66 + tests/nnet_data_test.h -> tests/golden/nnet_kat.json,
    toy 2->3 dense / conv1d(k=3) / GRU layers, eps 1e-5) run through the PRODUCTION kernels — STRICT and MFMA — by
    zero-embedding each toy layer in the corresponding PercepNet-shaped layer (extra inputs, taps and neurons carry
    zero weights, so they add exact zeros to every chain) and feeding it the test's constant 0.5 inputs through
    layers with zero weights and a 0.5 bias."""
    import json
    k = {n: np.array(v, np.float32) for n, v in json.load(open(os.path.join(golden_dir, "nnet_kat.json"))).items()}
    SIG, TANH, LIN = weights.ACT_SIGMOID, weights.ACT_TANH, weights.ACT_LINEAR
    feat = np.zeros((1, 70), np.float32); feat[0, :2] = 0.5
    Bp = 256

    def ctx_for(layers, acts):
        m = api.Model(weights.pack_blob(layers, acts))
        return m, api.Context(m, 1, nn_mode=mode)

    # compute_dense: fc 2->3 sigmoid embedded in fc 70->128
    L = _zero_layers()
    W = L["fc"]["input_weights"].reshape(70, 128); W[:2, :3] = k["fc_weights"].reshape(2, 3)
    L["fc"]["bias"][:3] = k["fc_bias"]
    m, ctx = ctx_for(L, {"fc": SIG})
    ctx.compute_rnn(feat)
    fc_out = ctx.debug_copy(1, 5 * Bp * 128).reshape(5, Bp, 128)[0, 0, :3]      # c1ring slot tn%5 = 0
    assert np.abs(fc_out - k["fc_output"]).max() < 1e-5, fc_out
    ctx.close(); m.close()

    # compute_conv1d: 2 -> 3, kernel 3, sigmoid, embedded in conv1 128 -> 512 kernel 5 (its 3 newest taps)
    L = _zero_layers()
    L["fc"]["bias"][:2] = 0.5                                                     # fc: linear, output = [.5, .5, 0 ...]
    W = L["conv1"]["input_weights"].reshape(5, 128, 512); W[2:, :2, :3] = k["conv1_weights"].reshape(3, 2, 3)
    L["conv1"]["bias"][:3] = k["conv1_bias"]
    m, ctx = ctx_for(L, {"fc": LIN, "conv1": SIG})
    for push, row in ((1, None), (2, 0), (3, 1)):                                 # testnnet.cpp:37-48
        ctx.compute_rnn(feat)
        out = ctx.debug_copy(2, 3 * Bp * 512).reshape(3, Bp, 512)[(push - 1) % 3, 0, :3]
        if row is not None:
            assert np.abs(out - k["conv1_output"][3 * row:3 * row + 3]).max() < 1e-5, (push, out)
    ctx.close(); m.close()

    # compute_gru: 2 -> 3 reset-after tanh, embedded in gru1 512 -> 512; its input [.5, .5, 0 ...] comes from conv2's bias
    L = _zero_layers()
    L["conv2"]["bias"][:2] = 0.5
    W = L["gru1"]["input_weights"].reshape(512, 3, 512); W[:2, :, :3] = k["gru1_weights"].reshape(2, 3, 3)
    U = L["gru1"]["recurrent_weights"].reshape(512, 3, 512); U[:3, :, :3] = k["gru1_recurrent_weights"].reshape(3, 3, 3)
    b = L["gru1"]["bias"].reshape(2, 3, 512); b[:, :, :3] = k["gru1_bias"].reshape(2, 3, 3)
    m, ctx = ctx_for(L, {"fc": LIN, "conv1": LIN, "conv2": LIN, "gru1": TANH})
    for step in range(3):
        ctx.compute_rnn(feat)
        st = ctx.get_rnn_state()["gru1"][0]
        assert np.abs(st[:3] - k["gru1_output"][3 * step:3 * step + 3]).max() < 1e-5, (step, st[:3])
        assert not st[3:].any()                                                   # the padding neurons stay at exactly 0
    ctx.close(); m.close()


def test_rnn_state_roundtrip_and_compute_rnn_symbol(blob, oracle):
    """pn_ctx_get/set_rnn_state_host move the network state in the reference's RNNState layout (nnet_data.h:28-38), and
    the exported compute_rnn(RNNState*, gains, strengths, input) (rnnoise.h:68; mangled like the reference's) advances a
    caller-owned host state exactly like the reference's: STRICT mode bit-identical to the oracle, step by step."""
    rng = np.random.default_rng(5)
    T = 7
    feats = (rng.standard_normal((T, 70)) * 0.7).astype(np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    ref = np.zeros((T, 68), np.float32)
    st = oracle.lib.pno_create(oracle.model)
    for t in range(T):
        oracle.lib.pno_compute_rnn(st, ref[t, :34].ctypes.data_as(fp), ref[t, 34:].ctypes.data_as(fp), feats[t].ctypes.data_as(fp))
    oracle.lib.pno_destroy(st)

    # (1) state save / restore through the batched API: run 4 steps, move the state to a fresh context, continue
    m = api.Model(blob)
    a = api.Context(m, 1, nn_mode=api.NN_STRICT)
    for t in range(4):
        assert np.array_equal(a.compute_rnn(feats[t][None])[0], ref[t])
    saved = a.get_rnn_state()
    b = api.Context(m, 1, nn_mode=api.NN_STRICT)
    b.compute_rnn(feats[0][None])                 # put b's rings at a different phase first
    b.set_rnn_state(saved)
    for t in range(4, T):
        assert np.array_equal(b.compute_rnn(feats[t][None])[0], ref[t]), t
    a.close(); b.close(); m.close()

    # (2) the reference's entry point on a caller-owned RNNState
    L = ctypes.CDLL(api.LIB_PATH)
    class RNNState(ctypes.Structure):
        _fields_ = [("model", ctypes.c_void_p)] + [(n, fp) for n in ("c1", "c2", "g1", "g2", "g3", "gb", "rb")] + \
                   [("convout_buf", ctypes.c_float * 1536)]
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".pnw") as f:
        f.write(blob); f.flush()
        libc = ctypes.CDLL(None)
        libc.fopen.restype = ctypes.c_void_p; libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        libc.fclose.argtypes = [ctypes.c_void_p]
        fh = libc.fopen(f.name.encode(), b"rb")
        L.rnnoise_model_from_file_c.restype = ctypes.c_void_p; L.rnnoise_model_from_file_c.argtypes = [ctypes.c_void_p]
        rm = L.rnnoise_model_from_file_c(fh); libc.fclose(fh)
    assert rm
    arrs = [np.zeros(n, np.float32) for n in (5 * 128, 3 * 512, 512, 512, 512, 512, 128)]   # rnnoise_init's callocs (denoise.cpp:268-274)
    rs = RNNState(rm, *[x.ctypes.data_as(fp) for x in arrs])
    fn = getattr(L, "_Z11compute_rnnP8RNNStatePfS1_PKf")
    fn.restype = None; fn.argtypes = [ctypes.POINTER(RNNState), fp, fp, fp]
    os.environ["PERCEPNET_STRICT"] = "1"
    try:
        for t in range(T):
            g = np.zeros(34, np.float32); r = np.zeros(34, np.float32)
            fn(ctypes.byref(rs), g.ctypes.data_as(fp), r.ctypes.data_as(fp), feats[t].ctypes.data_as(fp))
            assert np.array_equal(np.concatenate([g, r]), ref[t]), t
    finally:
        del os.environ["PERCEPNET_STRICT"]
    assert arrs[2].any() and arrs[0][:512].any()          # the caller's arrays really carry the state


def test_relinked_reference_cli(blob, oracle, tmp_path):
    """INTEGRATION.md level 1 for real: the reference's UNTOUCHED src/main.cpp, compiled where it lies and linked
    against libpercepnet_hip.so (percepnet_amd/build.py -> lib/percepNet_run_relinked), run as `percepNet_run in out`
    with the model from PERCEPNET_MODEL: out.pcm and ./feature_test.raw against the oracle."""
    import subprocess
    from percepnet_amd import build
    if not os.path.exists(build.RELINKED):
        pytest.skip("lib/percepNet_run_relinked was not built (needs /root/reference at build time)")
    (tmp_path / "m.pnw").write_bytes(blob)
    a = synth.synth_stream(3, 30)
    (tmp_path / "a.pcm").write_bytes(a.tobytes())
    env = dict(os.environ, PERCEPNET_MODEL=str(tmp_path / "m.pnw"))
    r = subprocess.run([build.RELINKED, "a.pcm", "ao.pcm"], cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert "INERT" not in r.stderr
    ao = np.fromfile(tmp_path / "ao.pcm", np.int16); tap = np.fromfile(tmp_path / "feature_test.raw", np.float32).reshape(-1, 68)
    ro, rg = oracle.run_pcm(a)
    assert ao.size == 29 * 480 and np.abs(ao.astype(np.int32) - ro.astype(np.int32)).max() <= PCM_TOL_LSB
    assert np.abs(ro).max() > 1000
    assert tap.shape == (30, 68) and np.abs(tap - rg).max() <= GR_TOL
    r = subprocess.run([build.RELINKED, "a.pcm", "as.pcm"], cwd=tmp_path, capture_output=True, text=True, timeout=300,
                       env=dict(env, PERCEPNET_STRICT="1"))
    assert r.returncode == 0 and np.array_equal(np.fromfile(tmp_path / "as.pcm", np.int16), ro)      # bit-exact in STRICT


def test_small_batch_and_batch_gemm_kernel_families_agree_bit_for_bit(model, oracle):
    """Contexts of up to PERCEPNET_SMALL_ROWS streams (default 4096) run the small-batch network kernels (pn_nn_small.hip:
    one 32x32 tile and one accumulator chain per wave, a GRU's three gates on three waves), larger ones the batch-GEMM
    kernels.  Both evaluate the same k-ordered fmaf chains, so the SAME batch through either family must give bit-identical
    g/r and PCM — and both must sit within the MFMA tolerance of the oracle.  Ragged size: 2 M tiles of the batch family,
    10 of the small one, the last ragged in both."""
    B, T = 300, 12
    pcm = synth.synth_batch(B, T, first_stream=40)
    res = {}
    for fam, rows in (("small", "4096"), ("batch", "0")):
        os.environ["PERCEPNET_SMALL_ROWS"] = rows
        try:
            ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
        finally:
            del os.environ["PERCEPNET_SMALL_ROWS"]
        res[fam] = ctx.run_pcm(pcm)
        ctx.close()
    assert np.array_equal(res["small"][0], res["batch"][0])
    assert np.array_equal(res["small"][1].view(np.uint32), res["batch"][1].view(np.uint32))
    ro, rg, _, _ = oracle.run_batch(pcm, want_feat=False)
    assert np.abs(res["small"][0].astype(np.int32) - ro.astype(np.int32)).max() <= PCM_TOL_LSB
    assert np.abs(res["small"][1] - rg).max() <= GR_TOL


def test_front_end_kernel_families_agree_bit_for_bit(model, oracle):
    """Three front ends compute compute_frame_features (denoise.cpp:372-434): the phase-split kernels (pn_dsp_fe_split_*.hip:
    spec_in / pitch / spec_out, the default above 2048 streams), and the single-launch kernel with four or two streams
    per wavefront.  Different lane mappings, a differently organised FFT (three register-fused passes vs five stages), a
    sparse fine pitch search, band sums over padded band-major operands — and the same reference-order arithmetic: every
    feature word, silence flag, PCM sample and g/r tap must be bit-identical, and equal to the oracle's features.  Ragged
    batch over a full history-ring wrap (14 > 12 frames), extreme-level streams included."""
    B, T = 301, 14
    pcm = synth.synth_batch(B, T, first_stream=60)
    pcm[7] = np.where((np.arange(T * 480) // 120) % 2 == 0, 32767, -32768).astype(np.int16)     # full-scale square wave
    pcm[8] = 0
    res = {}
    for fam in ("split", "mono", "g2"):
        os.environ["PERCEPNET_FE"] = fam
        try:
            ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
        finally:
            del os.environ["PERCEPNET_FE"]
        assert ctx.describe()["frontend"] == {"split": "split", "mono": "g4", "g2": "g2"}[fam]
        outs, grs, feats, sils = [], [], [], []
        for t in range(T):
            o, g = ctx.process_i16(pcm[:, t * 480:(t + 1) * 480])
            f, s_ = ctx.read_features()
            outs.append(o); grs.append(g); feats.append(f); sils.append(s_)
        res[fam] = (np.stack(outs), np.stack(grs).view(np.uint32), np.stack(feats).view(np.uint32), np.stack(sils))
        ctx.close()
    for fam in ("mono", "g2"):
        for a, b in zip(res["split"], res[fam]):
            assert np.array_equal(a, b), fam
    _, _, rf, rs = oracle.run_batch(pcm)
    assert np.array_equal(res["split"][2], np.ascontiguousarray(rf.transpose(1, 0, 2)).view(np.uint32))
    assert np.array_equal(res["split"][3], rs.T)
    assert (rs == 0).any()                       # non-silent frames: the comb-filtered spectrum mattered


def test_create_time_self_tests_run_and_pass(blob, tmp_path):
    """pn_ctx_create runs two known-answer self-tests once per (device, kernel family) and process: the MFMA network
    kernels against the reference-order kernels on a built-in weight set, and the DSP kernels on a built-in waveform
    against the CPU oracle's stored 70 features + 14 silence flags (40 streams squeezed through ONE block's grid-stride
    loop).  PERCEPNET_SELFTEST=2 reports them on stderr; every front-end family must pass."""
    import subprocess, sys, textwrap
    (tmp_path / "m.pnw").write_bytes(blob)
    code = textwrap.dedent("""
        import sys; sys.path.insert(0, %r)
        from percepnet_amd import api
        m = api.Model(open(%r, "rb").read())
        for mode in (api.NN_MFMA, api.NN_MFMA_F16):
            c = api.Context(m, 300, nn_mode=mode); c.close()
        c = api.Context(m, 5000, nn_mode=api.NN_MFMA); c.close()
        print("created")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "m.pnw"))
    for fam in ("split", "mono", "g2"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, PERCEPNET_SELFTEST="2", PERCEPNET_FE=fam))
        assert r.returncode == 0 and "created" in r.stdout, r.stderr[-2000:]
        assert r.stderr.count("DSP self-test") == 1 and "bit-equal to the CPU oracle, 40 streams identical" in r.stderr, r.stderr[-2000:]
        assert r.stderr.count("network self-test") == 3, r.stderr[-2000:]          # small fp32, small... fp16, batch fp32 families
        assert "FAILED" not in r.stderr and "SKIPPED" not in r.stderr


def test_extreme_inputs_strict_bit_exact(model, oracle):
    """Inputs at the edges of the int16 range: full-scale square wave, alternating +-32768/32767, a lone impulse, DC at
    both rails, white noise at full scale.  The CLI's float->int16 conversion truncates and WRAPS (main.cpp:36, no
    clamp), so overshoot of the synthesis window wraps around: STRICT mode must reproduce the oracle's PCM, tap, features
    and silence flags bit for bit, MFMA mode within its tolerance on g/r (a wrapped sample may differ by 65535)."""
    T = 30
    n = T * 480
    rng = np.random.default_rng(3)
    t = np.arange(n)
    x = np.stack([
        np.where((t // 120) % 2 == 0, 32767, -32768),
        np.where(t % 2 == 0, 32767, -32768),
        np.where(t == 700, 32767, 0),
        np.full(n, 32767), np.full(n, -32768),
        rng.integers(-32768, 32768, n),
        np.where((t // 4800) % 2 == 0, 0, rng.integers(-32768, 32768, n)),      # silence <-> full-scale noise every 10 frames
    ]).astype(np.int16)
    B = x.shape[0]
    ro, rg, rf, rs = oracle.run_batch(x, threads=4)
    ctx = api.Context(model, B, nn_mode=api.NN_STRICT)
    out = np.zeros_like(ro); gr = np.zeros_like(rg)
    for f in range(T):
        o, g = ctx.process_i16(x[:, f * 480:(f + 1) * 480])
        feat, sil = ctx.read_features()
        assert np.array_equal(feat.view(np.uint32), rf[:, f].view(np.uint32)), f
        assert np.array_equal(sil, rs[:, f]), f
        gr[:, f] = g
        if f > 0:
            out[:, (f - 1) * 480:f * 480] = o
    ctx.close()
    assert np.array_equal(out, ro)
    assert np.array_equal(gr.view(np.uint32), rg.view(np.uint32))
    assert (rs == 0).any() and (rs == 1).any()
    # MFMA and split-precision modes: g/r within the mode's bound, PCM within 1 LSB as a CIRCULAR distance — the reference's
    # cast wraps (main.cpp:36), so a sample whose float value sits at the +-32768 boundary may land on either side
    for mode in (api.NN_MFMA, api.NN_MFMA_X3):
        ctx = api.Context(model, B, nn_mode=mode)
        om, gm = ctx.run_pcm(x)
        ctx.close()
        assert np.abs(gm - rg).max() <= GR_TOL, mode
        d = np.abs(om.astype(np.int32) - ro.astype(np.int32))
        assert np.minimum(d, 65536 - d).max() <= PCM_TOL_LSB, (mode, int(np.minimum(d, 65536 - d).max()))


def test_pitch_analysis_on_periodic_signals_bit_exact(model, oracle):
    """The pitch kernel (pn_dsp_fe_split_p.hip) against the oracle where its decisions are hard: pulse trains and sawtooths
    at periods across and beyond the pitch range (32 .. 800 samples: below PITCH_MIN, through every `remove_doubling`
    subharmonic k = 2..15, above PITCH_MAX), two-harmonic mixtures whose fundamental is missing, a slow chirp, period
    jumps between frames (the continuity terms prev_period / prev_gain), and amplitude steps.  Features (68 = period,
    69 = correlation), silence flags, taps and PCM must be bit-identical in STRICT mode over 24 frames; the detected
    periods must cover the range (the test is not vacuous)."""
    T = 24
    n = T * 480
    t = np.arange(n)
    rows = []
    for P in list(range(32, 128, 7)) + list(range(128, 800, 23)):
        rows.append(np.where(t % P == 0, 20000, 0))                                  # pulse train
        rows.append(((t % P) * (24000.0 / P) - 12000).astype(np.int64))              # sawtooth
    for P in (96, 150, 233, 377, 610):
        w = 2 * np.pi / P
        rows.append((6000 * np.sin(2 * w * t) + 5000 * np.sin(3 * w * t)).astype(np.int64))      # missing fundamental
        rows.append((9000 * np.sin(w * t) * (1 + 0.5 * np.sign(np.sin(2 * np.pi * t / 4800.0)))).astype(np.int64))   # amplitude steps
        Pj = np.where((t // 1440) % 2 == 0, P, P + 37)                                # period jumps every 3 frames
        rows.append(np.where(t % Pj == 0, 18000, 0))
    rows.append((8000 * np.sin(2 * np.pi * np.cumsum(1.0 / (60 + 700.0 * t / n)))).astype(np.int64))               # chirp 60 -> 760
    x = np.stack(rows).astype(np.int16)
    B = x.shape[0]
    ro, rg, rf, rs = oracle.run_batch(x, threads=8)
    ctx = api.Context(model, B, nn_mode=api.NN_STRICT)
    out = np.zeros_like(ro); gr = np.zeros_like(rg)
    periods = set()
    for f in range(T):
        o, g = ctx.process_i16(x[:, f * 480:(f + 1) * 480])
        feat, sil = ctx.read_features()
        assert np.array_equal(feat.view(np.uint32), rf[:, f].view(np.uint32)), (f, np.argwhere(feat.view(np.uint32) != rf[:, f].view(np.uint32))[:4])
        assert np.array_equal(sil, rs[:, f]), f
        periods.update(np.round(feat[:, 68] * (768 - 3 * 60)).astype(int).tolist())
        gr[:, f] = g
        if f > 0:
            out[:, (f - 1) * 480:f * 480] = o
    ctx.close()
    assert np.array_equal(out, ro)
    assert np.array_equal(gr.view(np.uint32), rg.view(np.uint32))
    assert min(periods) <= 70 and max(periods) >= 700 and len(periods) >= 60, (min(periods), max(periods), len(periods))


def test_float_api_huge_and_non_finite_inputs(model, oracle):
    """rnnoise_process_frame takes arbitrary floats (the CLI only ever feeds |x| <= 1).  The reference has a defined x86
    behaviour for everything it is fed — samples at 1e4 stay finite all the way (output ~700), at 1e9 the band energies
    overflow and the stream is NaN from its first frame, pre-activations past the tanh table hit cvttss2si's INT_MIN
    (vec.h:61) and the clamp's index 0; a NaN or an infinity poisons that stream's history and spectra for good.  STRICT mode must
    follow the CPU oracle bit for bit on the finite streams, value-or-NaN for the poisoned ones, and the clean streams
    sharing the batch (the same GEMM tiles) must not notice."""
    T = 14
    base = (synth.synth_batch(6, T).astype(np.float32) / 32768.0)
    x = base.copy()
    x[1] *= 1e4                                                   # finite all the way through (|out| ~ 700): bit-exact
    x = np.concatenate([x, (base[1] * 1e9)[None]])                # 1e9: the CPU path itself turns into NaN in the first frame
    x[2, 3 * 480 + 17] = np.nan
    x[3, 4 * 480 + 100] = np.inf
    x[4] = -1e9                                                   # DC at -1e9
    x[5, 5 * 480 + 3] = -np.inf; x[5, 5 * 480 + 4] = np.inf
    ref = [oracle.run_float(x[s]) for s in range(7)]
    ctx = api.Context(model, 7, nn_mode=api.NN_STRICT)
    same = lambda a, b: bool((((a == b) | (np.isnan(a) & np.isnan(b)))).all())
    for f in range(T):
        o, g = ctx.process_f32(x[:, f * 480:(f + 1) * 480])
        for s in range(7):
            ro, rg = ref[s][0][f * 480:(f + 1) * 480], ref[s][1][f]
            if s in (0, 1):                                         # finite streams: the bits
                assert np.array_equal(o[s].view(np.uint32), ro.view(np.uint32)), (f, s)
                assert np.array_equal(g[s].view(np.uint32), rg.view(np.uint32)), (f, s)
            else:                                                   # poisoned streams: equal values, NaN where the CPU has NaN
                assert same(o[s], ro) and same(g[s], rg), (f, s)
    ctx.close()
    assert np.isfinite(ref[0][0]).all() and np.isfinite(ref[1][0]).all() and np.abs(ref[1][0]).max() > 100          # the scenario is
    assert all(not np.isfinite(ref[s][0]).all() for s in (2, 3, 4, 5, 6))                                        # what it claims
    # the fp32 MFMA mode on the same batch: the clean stream beside the poisoned ones stays within its bounds
    ctx = api.Context(model, 7, nn_mode=api.NN_MFMA)
    for f in range(T):
        o, g = ctx.process_f32(x[:, f * 480:(f + 1) * 480])
        assert np.abs(g[0] - ref[0][1][f]).max() <= GR_TOL and np.abs(o[0] - ref[0][0][f * 480:(f + 1) * 480]).max() <= 2.0 / 32768, f
    ctx.close()


def test_maximum_size_batch_300k_streams(model, oracle):
    """Far beyond the benchmarked batch: 300 007 concurrent streams (ragged, 22.7 GiB of state; 64-bit indexing, grid limits,
    73 grid-stride rounds per front-end block) built from 16 distinct streams.  Every replica bit-identical wherever it
    sits, the first 16 slots against the oracle (features bit-equal, g/r, PCM).  One million streams (75.5 GiB) passes the
    same check: tools/big_batch_check.py, profiles/r02i_one_million_streams.log."""
    import torch
    B, T, P = 300007, 7, 16
    base = synth.synth_batch(P, T)
    ro, rg, rf, rs = oracle.run_batch(base, threads=8)
    dev = torch.device("cuda:0")
    ts = torch.cuda.Stream(dev)
    with torch.cuda.stream(ts):
        ctx = api.Context(model, B, nn_mode=api.NN_MFMA, stream=ts.cuda_stream)
        idx = torch.arange(B, device=dev) % P
        dbase = torch.from_numpy(base).to(dev)
        o = torch.empty((B, 480), dtype=torch.int16, device=dev); g = torch.empty((B, 68), dtype=torch.float32, device=dev)
        f = torch.empty((B, 70), dtype=torch.float32, device=dev); s = torch.empty((B,), dtype=torch.int32, device=dev)
        for t in range(T):
            fr = dbase[:, t * 480:(t + 1) * 480][idx].contiguous()
            ctx.process_i16_dev(fr.data_ptr(), o.data_ptr(), g.data_ptr())
            ctx.read_features_dev(f.data_ptr(), s.data_ptr())
            assert bool((o == o[:P][idx]).all()), (t, "pcm replicas")
            assert bool((g.view(torch.int32) == g[:P][idx].view(torch.int32)).all()), (t, "g/r replicas")
            assert bool((f.view(torch.int32) == f[:P][idx].view(torch.int32)).all()), (t, "feature replicas")
            assert np.array_equal(f[:P].cpu().numpy().view(np.uint32), rf[:, t].view(np.uint32)), (t, "features vs oracle")
            assert np.array_equal(s[:P].cpu().numpy(), rs[:, t])
            assert np.abs(g[:P].cpu().numpy() - rg[:, t]).max() <= GR_TOL
            if t > 0:
                assert np.abs(o[:P].cpu().numpy().astype(np.int32) - ro[:, (t - 1) * 480:t * 480].astype(np.int32)).max() <= PCM_TOL_LSB
        ctx.close()
    del o, g, f, s, dbase
    torch.cuda.empty_cache()
