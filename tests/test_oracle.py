"""CPU tests of the oracle (oracle/percepnet_oracle.c): against the committed golden vectors that
the compiled reference produced (tests/golden/make_golden.py), against the reference's own
known-answer fixture (tests/nnet_data_test.h -> nnet_kat.json), and — when oracle/_ref is present
— bit-for-bit against the compiled reference on fresh inputs."""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle.oracle import Oracle, Reference, _fp, c_f, ref_available
from percepnet_amd import synth, weights


def test_weights_match_reference_dumper(golden_dir):
    dg = json.load(open(os.path.join(golden_dir, "weights_seed1234.json")))["sha256"]
    mine = weights.layer_digest(weights.modules_to_layers(weights.build_torch_modules(1234)))
    assert mine == dg
    blob = weights.default_blob(1234)
    assert weights.layer_digest(weights.unpack_blob(blob)) == dg
    assert sum(v.size for d in weights.unpack_blob(blob).values() for v in d.values()) == weights.N_PARAMS


def test_tables_match_reference(oracle, golden_dir):
    t = np.load(os.path.join(golden_dir, "tables.npz"))
    tw, br, hw, ch, bd = oracle.tables()
    assert np.array_equal(tw.view(np.uint32), t["twiddles"].view(np.uint32))
    assert np.array_equal(br, t["bitrev"])
    assert list(t["factors"]) == [5, 192, 3, 64, 4, 16, 4, 4, 4, 1]
    assert np.array_equal(oracle.tansig_table(), t["tansig"])
    # SURVEY §8 a3 [probe]
    assert list(bd) == [0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 31, 36, 41, 48, 56, 65, 75,
                        86, 99, 115, 132, 152, 175, 201, 230, 265, 304, 349, 400]
    assert abs(float(ch.sum()) - 1) < 1e-6 and hw[0] > 0 and abs(hw[-1] - 1) < 1e-5


def test_nnet_known_answers(oracle, golden_dir):
    """The reference's own gtest cases (tests/testnnet.cpp:19-66), eps 1e-5."""
    k = {n: np.array(v, np.float32) for n, v in json.load(open(os.path.join(golden_dir, "nnet_kat.json"))).items()}
    L = oracle.lib
    half2 = np.full(2, .5, np.float32)
    out = np.zeros(3, np.float32)
    L.pno_dense(_fp(k["fc_bias"]), _fp(k["fc_weights"]), 2, 3, 1, _fp(out), _fp(half2))
    assert np.abs(out - k["fc_output"]).max() < 1e-5
    mem = np.zeros(6, np.float32)
    for push, row in ((1, None), (2, 0), (3, 1)):
        L.pno_conv1d(_fp(k["conv1_bias"]), _fp(k["conv1_weights"]), 2, 3, 3, 1, _fp(out), _fp(mem), _fp(half2))
        if row is not None:
            assert np.abs(out - k["conv1_output"][3 * row:3 * row + 3]).max() < 1e-5
    st = np.zeros(3, np.float32)
    for step in range(3):  # the reference asserts steps 0,1 one-sidedly; all three hold two-sided
        L.pno_gru(_fp(k["gru1_bias"]), _fp(k["gru1_weights"]), _fp(k["gru1_recurrent_weights"]), 2, 3, 2,
                  _fp(st), _fp(half2))
        assert np.abs(st - k["gru1_output"][3 * step:3 * step + 3]).max() < 1e-5


def test_oracle_matches_golden_pcm(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "pcm_golden.npz"))
    for s in (0, 3, 7, 13):
        assert np.array_equal(synth.synth_stream(s, 48), g[f"in_{s}"]), "synthetic generator drifted"
        out, gr = oracle.run_pcm(g[f"in_{s}"])
        assert np.array_equal(out, g[f"out_{s}"])
        assert np.array_equal(gr.view(np.uint32), g[f"gr_{s}"].view(np.uint32))
    fo, _ = oracle.run_float(g["fin_0"])
    assert np.array_equal(fo.view(np.uint32), g["fout_0"].view(np.uint32))
    fo, fgr = oracle.run_float(g["fin_3"])
    assert np.array_equal(fo.view(np.uint32), g["fout_3"].view(np.uint32))
    assert np.array_equal(fgr.view(np.uint32), g["fgr_3"].view(np.uint32))
    # the loud stream really takes both branches of `if(!silence)` (denoise.cpp:536)
    _, sil = oracle.features(g["in_3"].astype(np.float32) / np.float32(32768))
    assert 0 < sil.sum() < sil.size


def test_oracle_postfilter_option(oracle):
    """Optional envelope post-filter: off = the plain pipeline; on = same network tap, different
    PCM.  post_filtering itself is the function the train() restatement uses, pinned by the train parity tests below."""
    pcm = synth.synth_stream(3, 30)
    o0, g0 = oracle.run_pcm(pcm)
    o0b, _ = oracle.run_pcm(pcm, postfilter=False)
    o1, g1 = oracle.run_pcm(pcm, postfilter=True)
    assert np.array_equal(o0, o0b)
    assert np.array_equal(g0.view(np.uint32), g1.view(np.uint32))
    assert np.abs(o0.astype(np.int32) - o1.astype(np.int32)).max() > 50
    # warped gains are never larger than the gains (sin <= 1) and the compensation G is bounded by
    # sqrt((1+beta)/(2*sqrt(beta))) for beta = 0.02: the filtered output stays finite and of comparable level
    assert np.abs(o1.astype(np.int32)).max() < 4 * max(1, np.abs(o0.astype(np.int32)).max())


def test_activation_table_edges(oracle):
    L = oracle.lib
    assert L.pno_tansig(0.0) == 0.0
    assert abs(L.pno_tansig(100.0) - 1.0) < 1e-5 and abs(L.pno_tansig(-100.0) + 1.0) < 1e-5
    xs = np.linspace(-9, 9, 2001)
    ys = np.array([L.pno_tansig(float(x)) for x in xs])
    assert np.abs(ys - np.tanh(xs)).max() < 2e-4  # table + 2nd-order correction (vec.h:53-70)
    assert abs(L.pno_sigmoid(0.0) - .5) < 1e-7


@pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (no /root/reference)")
def test_oracle_bit_exact_vs_compiled_reference(blob, oracle):
    ref = Reference(blob)
    for s in (1, 23, 27, 33, 43):  # voiced, loud, bursts, two-tone, loud — not the golden streams
        pcm = synth.synth_stream(s, 120)
        a, ga = oracle.run_pcm(pcm)
        b, gb = ref.run_pcm(pcm)
        assert np.array_equal(a, b), s
        assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32)), s


@pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (no /root/reference)")
def test_oracle_stages_vs_compiled_reference(blob, oracle):
    ref = Reference(blob)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((960, 2)).astype(np.float32)
    a = np.zeros_like(x); b = np.zeros_like(x)
    oracle.lib.pno_fft960(_fp(x), _fp(a)); ref.lib.ref_fft960(_fp(x), _fp(b))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    sig = synth.synth_stream(2, 30).astype(np.float32) / 32768
    prev_p, prev_g = 300, np.float32(0.4)
    for off in range(0, 10000, 997):
        buf = sig[off:off + 1728].copy()
        la = np.zeros(864, np.float32); lb = np.zeros(864, np.float32)
        oracle.lib.pno_pitch_downsample(_fp(buf), _fp(la)); ref.lib.ref_pitch_downsample(_fp(buf), _fp(lb))
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))
        pa = ctypes.c_int(); pb = ctypes.c_int(); ca = ctypes.c_float(); cb = ctypes.c_float()
        oracle.lib.pno_pitch_search(_fp(la), ctypes.byref(pa), ctypes.byref(ca))
        ref.lib.ref_pitch_search(_fp(lb), ctypes.byref(pb), ctypes.byref(cb))
        assert pa.value == pb.value and ca.value == cb.value
        ta = ctypes.c_int(768 - pa.value); tb = ctypes.c_int(768 - pb.value)
        ga = oracle.lib.pno_remove_doubling(_fp(la), ctypes.byref(ta), prev_p, prev_g)
        gb = ref.lib.ref_remove_doubling(_fp(lb), ctypes.byref(tb), prev_p, prev_g)
        assert ta.value == tb.value and ga == gb
        prev_p, prev_g = ta.value, np.float32(ga)
    # saturating weights drive the tanh table into its clamp
    sat = weights.pack_blob(weights.random_layers(3, scale=6.0))
    o2 = Oracle(sat); r2 = Reference(sat)
    pcm = synth.synth_stream(4, 40)
    a, ga = o2.run_pcm(pcm); b, gb = r2.run_pcm(pcm)
    assert np.array_equal(a, b) and np.array_equal(ga.view(np.uint32), gb.view(np.uint32))
    assert ga.min() < 0.02 and ga.max() > 0.98


@pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (no /root/reference)")
def test_stage_taps_vs_compiled_reference_stage_functions(blob, oracle):
    """oracle.stages() (pno_frame_stages: the yardstick of tests/test_gpu_stages.py) pinned to the COMPILED REFERENCE's own stage
    functions: from the oracle's comb_buf tap, the window (reference tables), ref_fft960 (kiss_fft.cpp:566-586), compute_band_energy
    / compute_band_corr (denoise.cpp:89-160) and the comb filter (denoise.cpp:416-422, float32 mul then add in k order) are
    recomputed through oracle/_ref and must equal the taps bit for bit."""
    ref = Reference(blob)
    _, _, hw, hann, _ = oracle.tables()
    win = np.concatenate([hw, hw[::-1]]).astype(np.float32)

    def ref_spectrum(x960):
        z = np.zeros((960, 2), np.float32); z[:, 0] = x960 * win
        o = np.zeros_like(z)
        ref.lib.ref_fft960(_fp(z), _fp(o))
        return o[:481].copy()

    def ref_bands(X, P=None):
        e = np.zeros(34, np.float32)
        if P is None:
            ref.lib.ref_band_energy(_fp(e), _fp(X))
        else:
            ref.lib.ref_band_corr(_fp(e), _fp(X), _fp(P))
        return e

    def ri(c):                                        # complex64 [481] -> float32 [481, 2]
        return np.ascontiguousarray(c).view(np.float32).reshape(481, 2)

    for s in (3, 7, 13):
        x = synth.synth_stream(s, 24).astype(np.float32) / np.float32(32768.0)
        st = oracle.stages(x)
        for t in (0, 5, 11, 17, 23):
            cb = st["comb_buf"][t]
            X = ref_spectrum(cb[2400:3360]); Y = ref_spectrum(cb[4800:5760])
            assert np.array_equal(X.view(np.uint32), ri(st["X"][t]).view(np.uint32)), (s, t)
            assert np.array_equal(Y.view(np.uint32), ri(st["Y"][t]).view(np.uint32)), (s, t)
            T0 = int(st["period"][t])
            p = np.zeros(960, np.float32)
            for k in range(-3, 4):
                p = p + cb[2400 - T0 * k:2400 - T0 * k + 960] * hann[k + 3]
            P = ref_spectrum(p)
            assert np.array_equal(P.view(np.uint32), ri(st["P"][t]).view(np.uint32)), (s, t)
            assert np.array_equal(ref_bands(X).view(np.uint32), st["Ex"][t].view(np.uint32))
            assert np.array_equal(ref_bands(P).view(np.uint32), st["Ep"][t].view(np.uint32))
            assert np.array_equal(ref_bands(Y).view(np.uint32), st["Ey"][t].view(np.uint32))
        assert np.abs(st["X"]).max() > 0


# ---- the training-feature binary (train(), denoise.cpp:603-787) -------------------
def test_train_oracle_matches_golden_records(oracle, golden_dir):
    """Golden = the compiled reference's train() run through real files (make_golden.py featgen)."""
    g = np.load(os.path.join(golden_dir, "featgen_golden.npz"))
    for p in (0, 2, 3):
        sp, no = synth.synth_pair(p, 36)
        assert np.array_equal(sp, g[f"speech_{p}"]) and np.array_equal(no, g[f"noisy_{p}"])   # generator did not drift
        rec, pcm = oracle.train_run(sp, no)
        assert np.array_equal(rec.view(np.uint32), g[f"rec_{p}"].view(np.uint32)), p
        assert np.array_equal(pcm, g[f"pcm_{p}"]), p
    r3 = g["rec_3"]
    assert (r3[14:25, 70:104] == 0).all()             # window of frame t = input frames t-6, t-5: all-zero for t in [14, 24]
    assert np.isfinite(r3).all()
    assert (g["rec_0"][:, 104:] == np.float32(0.99)).any()      # adjust_gain_strength_by_condition branch taken


@pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (no /root/reference)")
def test_train_oracle_bit_exact_vs_compiled_reference(blob, oracle, tmp_path):
    ref = Reference(blob)
    for p in (1, 5, 7):
        sp, no = synth.synth_pair(p, 50)
        d = tmp_path / f"p{p}"; d.mkdir()
        rr, rp = ref.train(sp, no, str(d))
        orc, op = oracle.train_run(sp, no)
        assert np.array_equal(rr.view(np.uint32), orc.view(np.uint32)), p
        assert np.array_equal(rp, op), p
        # the reference's other by-product: test_input.pcm == the noisy input
        assert np.array_equal(np.fromfile(d / "test_input.pcm", np.int16), no)


def test_batched_oracle_is_bit_identical(oracle):
    """pno_run_pcm_batch (groups of streams share each sweep over the weights, groups spread over host threads) is the
    oracle the long-horizon GPU tests use: PCM, g/r tap, features and silence flags must equal the single-stream
    functions bit for bit, for full and ragged groups, every stream kind, and any thread count."""
    S, T = 23, 24
    pcm = synth.synth_batch(S, T)                     # kinds: voiced, loud (3), bursts (7), two-tone (13)
    want = [oracle.run_pcm(pcm[s]) for s in range(S)]
    wf = [oracle.features(pcm[s].astype(np.float32) / np.float32(32768)) for s in range(S)]
    for group, threads in ((8, 3), (5, 1), (32, 2), (1, 8)):
        out, gr, feat, sil = oracle.run_batch(pcm, group=group, threads=threads)
        for s in range(S):
            assert np.array_equal(out[s], want[s][0]), (group, s)
            assert np.array_equal(gr[s].view(np.uint32), want[s][1].view(np.uint32)), (group, s)
            assert np.array_equal(feat[s].view(np.uint32), wf[s][0].view(np.uint32)), (group, s)
            assert np.array_equal(sil[s], wf[s][1]), (group, s)
    assert (sil[3] == 0).any() and sil[7].any()       # both silence branches are in the sample
