"""Per-stage DSP parity on the GPU (-m gpu; SURVEY §4: "per-stage DSP parity tests against the compiled oracle", round-5 verdict
items 5 and 6).  The end-to-end tests compare features, silence flags and PCM; here the INTERMEDIATES of the DSP kernels are read
back through pn_ctx_debug_copy after every frame and compared bit for bit with the CPU oracle's stage taps (oracle.stages(), pinned
to the compiled reference's own stage functions by tests/test_oracle.py::test_stage_taps_vs_compiled_reference_stage_functions):

  tap 12  history ring          = comb_buf after the frame                  (a1/a2: denoise.cpp:383-386 as a ring)
  tap 10  look-ahead spectra    = Y(t) (window + forward transform)          (a7, a8: denoise.cpp:333-346, kiss_fft.cpp:566-586)
                                  and X(t) == Y(t-5): the ring slot the back end reads
  tap 11  comb-filtered spectrum P(t)                                        (a14: denoise.cpp:416-427)
  tap 13  pitch period                                                        (a11-a13)

Bins >= 400: the device rings hold 400 of the 481 bins.  That is not a truncation of the comparison but of the data path: every
consumer of X, P, Y in the reference stops at bin 400 — the 34 ERB bands end at nfftborder[33] = 400 (denoise.cpp:89-160) and
interp_band_gain leaves g[400..480] at the 0 it was initialised with (denoise.cpp:164-182, 517), so bins >= 400 are multiplied by
0 before the inverse transform.  test_bins_from_400_up_never_reach_the_output checks exactly that property on the oracle side with
the GPU's PCM as the witness: the oracle's output with its full 481-bin spectra equals the GPU's output built from 400."""
import numpy as np
import pytest

from percepnet_amd import api, synth, weights

pytestmark = pytest.mark.gpu

KINDS = (0, 3, 7, 13, 23, 27)          # voiced, loud (non-silent branch), bursts + digital silence, two-tone, loud, bursts
T = 30
HIST, STRIDE, BINS = 5760, 5768, 400


@pytest.fixture(scope="module")
def model(blob):
    m = api.Model(blob)
    yield m
    m.close()


def _logical_history(ring, t):
    """ring [B, 5768]: slot f % 12 holds input frame f (+ a mirror of the first 8 samples); -> comb_buf [B, 5760] after frame t
    (oldest first: frames t-11 .. t; frames before the start of the stream are zeros)."""
    B = ring.shape[0]
    out = np.zeros((B, HIST), np.float32)
    for j in range(12):
        f = t - 11 + j
        if f >= 0:
            out[:, j * 480:(j + 1) * 480] = ring[:, (f % 12) * 480:(f % 12 + 1) * 480]
    return out


def _run_with_taps(model, pcm, mode):
    B = pcm.shape[0]
    ctx = api.Context(model, B, nn_mode=mode)
    taps = {"hist": [], "Y": [], "X": [], "P": [], "period": []}
    frames = pcm.reshape(B, -1, 480)
    outs = []
    for t in range(frames.shape[1]):
        x = frames[:, t].astype(np.float32) / np.float32(32768.0)
        o, _ = ctx.process_f32(x)
        outs.append(o.copy())
        ring = ctx.debug_copy(12, B * STRIDE).reshape(B, STRIDE)
        assert np.array_equal(ring[:, HIST:], ring[:, :8])                       # the mirror of the ring's first 8 samples
        taps["hist"].append(_logical_history(ring, t))
        yr = ctx.debug_copy(10, 6 * B * BINS * 2).view(np.complex64).reshape(6, B, BINS)
        taps["Y"].append(yr[t % 6].copy()); taps["X"].append(yr[(t + 1) % 6].copy())
        taps["P"].append(ctx.debug_copy(11, B * BINS * 2).view(np.complex64).reshape(B, BINS).copy())
        taps["period"].append(ctx.debug_copy(13, B).view(np.int32).copy())
    ctx.close()
    return {k: np.stack(v, 1) for k, v in taps.items()}, np.stack(outs, 1)


@pytest.fixture(scope="module")
def taps_and_oracle(model, oracle):
    pcm = np.stack([synth.synth_stream(s, T) for s in KINDS])
    pcm[2, 8 * 480:20 * 480] = 0                          # digital silence inside the burst stream: all-zero windows and spectra
    gpu, out = _run_with_taps(model, pcm, api.NN_STRICT)
    ref = [oracle.stages(pcm[i].astype(np.float32) / np.float32(32768.0)) for i in range(len(KINDS))]
    return pcm, gpu, out, ref


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_history_ring_is_the_references_comb_buf(taps_and_oracle):
    _, gpu, _, ref = taps_and_oracle
    for i, s in enumerate(KINDS):
        assert np.array_equal(_bits(gpu["hist"][i]), _bits(ref[i]["comb_buf"])), s


def test_lookahead_and_analysis_spectra_bit_equal(taps_and_oracle):
    """window + 960-point forward transform (a7/a8): Y(t) of every frame, and the slot the back end reads as X(t) (= Y(t-5))."""
    _, gpu, _, ref = taps_and_oracle
    some_energy = 0
    for i, s in enumerate(KINDS):
        assert np.array_equal(_bits(gpu["Y"][i]), _bits(ref[i]["Y"][:, :BINS])), s
        assert np.array_equal(_bits(gpu["X"][i]), _bits(ref[i]["X"][:, :BINS])), s
        some_energy += int((np.abs(ref[i]["Y"][:, :BINS]) > 1e-4).sum())
    assert some_energy > 10000                                                       # not a comparison of zeros
    # the loud streams really take the non-silent branch, the burst streams really have all-zero windows
    assert (ref[1]["silence"] == 0).any() and (ref[2]["Y"][:, :BINS] == 0).all(axis=1).any()


def test_comb_filtered_spectrum_and_period_bit_equal(taps_and_oracle):
    """pitch search + remove_doubling give the period (a11-a13), the comb filter + window + transform give P (a14)."""
    _, gpu, _, ref = taps_and_oracle
    for i, s in enumerate(KINDS):
        assert np.array_equal(gpu["period"][i], ref[i]["period"]), s
        assert np.array_equal(_bits(gpu["P"][i]), _bits(ref[i]["P"][:, :BINS])), s
    assert len({int(p) for i in range(len(KINDS)) for p in ref[i]["period"]}) > 8       # several different comb windows were exercised


def test_bins_from_400_up_never_reach_the_output(taps_and_oracle, oracle):
    """The oracle carries all 481 bins (the reference's arrays); its spectra are non-zero above bin 400 and its PCM still equals the
    GPU's (STRICT: bit-equal), which is built from 400-bin rings."""
    pcm, _, out, ref = taps_and_oracle
    assert max(float(np.abs(r["X"][:, BINS:]).max()) for r in ref) > 1e-4
    for i in range(len(KINDS)):
        ro, _ = oracle.run_float(pcm[i].astype(np.float32) / np.float32(32768.0))
        assert np.array_equal(_bits(out[i].reshape(-1)), _bits(ro)), KINDS[i]


def test_taps_do_not_depend_on_the_network_mode(model, taps_and_oracle):
    """The DSP intermediates are the network-independent part: the MFMA mode's taps are the STRICT mode's, bit for bit."""
    pcm, gpu, _, _ = taps_and_oracle
    g2, _ = _run_with_taps(model, pcm, api.NN_MFMA)
    for k in ("hist", "Y", "X", "P", "period"):
        assert np.array_equal(_bits(gpu[k]) if gpu[k].dtype != np.int32 else gpu[k], _bits(g2[k]) if g2[k].dtype != np.int32 else g2[k]), k
