"""How far does the "+-1 LSB / 2e-5" claim of the MFMA network mode reach beyond the near-init benchmark weights?

Every long-horizon check in test_gpu_longrun.py uses the torch.manual_seed(1234) default-init network (gates g ~ r ~ 0.5).
The MFMA kernels evaluate the reference's k-ordered chains as fused multiply-adds, the CPU rounds multiply and add
separately (nnet.cpp:59-72): the per-step difference is a few ulp of the pre-activation, and how much of it reaches the
PCM depends on the weights — a network with larger dynamic range amplifies it through its GRUs and through steep
regions of the interpolated tanh/sigmoid table (vec.h:53-75), a saturating one clamps it away.  This file MEASURES it:
256 streams x 1000 frames (10 s) per weight set, MFMA mode vs the CPU oracle running the SAME weight set, the LSB
histogram and the per-second |dg,r| written to gpurun_out/parity_stress_*.json (copied to profiles/).

Two measurements per weight set:
  * FREE-RUNNING (the product's behaviour): MFMA context vs the CPU oracle over 10 s.  A random network at 2-3x the
    default weight scale is a CHAOTIC recurrence — two evaluations that differ in the last bit decorrelate after a few
    frames whatever their quality — so for those sets the free-running difference measures the network's own
    sensitivity, not the kernels; it is recorded, bounded only by "finite" and "no drift".
  * ONE STEP FROM IDENTICAL STATE: an MFMA context and a STRICT (reference-order, bit-exact to the CPU) context on the
    same GPU, the MFMA context's recurrent state overwritten with the STRICT one's before every step, both fed the
    oracle's features — the kernels' own error, without the recurrence's amplification.  This is the number that must be
    small for EVERY weight set, saturating ones included (vec.h:53-75 clamps the table index; nnet.cpp:161-179).
What must hold for every set: features + silence flags bit-equal (they never touch the network), every output finite,
the one-step bound of ONE_STEP_TOL, and the free-running tolerance stated per set in BOUNDS where one can be stated.
"""
import numpy as np
import pytest

from percepnet_amd import api, synth, weights
from test_gpu_longrun import run_long, shared_stream, _record

pytestmark = pytest.mark.gpu


def _biased(seed, scale):
    """scale-x weights plus gate biases that push z and r away from 0.5 (z towards 1: long memory; r towards 0)."""
    lay = weights.random_layers(seed, scale=scale)
    for name, kind, nin, nn_, ks, act in weights.LAYERS:
        if kind == weights.KIND_GRU:
            b = lay[name]["bias"].reshape(2, 3, nn_).copy()       # C order (z, r, h) x (input, recurrent) (dump_percepnet.py:78-80)
            b[0, 0] += 1.5                                        # z gate
            b[0, 1] -= 1.5                                        # r gate
            lay[name]["bias"] = np.ascontiguousarray(b.reshape(-1), dtype=np.float32)
    return lay


SETS = {
    "scale2": lambda: weights.random_layers(11, scale=2.0),
    "scale3": lambda: weights.random_layers(12, scale=3.0),
    "scale2_gate_biased": lambda: _biased(13, 2.0),
    "scale6_saturating": lambda: weights.random_layers(14, scale=6.0),
}
# max |dg,r| of ONE network step from identical state.  Measured (profiles/r03c_parity_stress_*.json): 2.5e-6 / 2.7e-6 /
# 3.2e-6 for the x2 / gate-biased / x3 sets, 6.0e-5 for the x6 saturating set (pre-activations of +-100: the fused
# multiply-add's rounding differs from mul-then-add by an ulp of THAT magnitude, and the steep part of the table passes it on).
ONE_STEP_TOL = {"scale2": 2e-5, "scale3": 2e-5, "scale2_gate_biased": 2e-5, "scale6_saturating": 2e-4}
# free-running (max |dPCM| in LSB, max |dg,r|) asserted per set; None = chaotic set, recorded only — measured values are
# in profiles/r03*_parity_stress_*.json
# Measured free-running over 10 s: x2, gate-biased, x3: max 1 LSB, |dg,r| <= 6.3e-6 — the north_star tolerance holds well
# beyond the near-init benchmark weights.  x6 saturating: |dg,r| up to 1.7e-3 (99.99 % below 3.7e-4); its output hits the
# int16 rails, where the reference's wrap-around cast (main.cpp:36) turns a tiny float difference into +-65535, so its PCM
# is compared as a circular distance.
BOUNDS = {
    "scale2": (1, 2e-5),
    "scale3": (1, 2e-5),
    "scale2_gate_biased": (1, 2e-5),
    "scale6_saturating": (None, 5e-3),
}


@pytest.mark.parametrize("name", list(SETS))
def test_mfma_mode_vs_oracle_on_other_weight_sets(name):
    import torch
    from oracle.oracle import Oracle
    B, T = 256, 1000
    blob = weights.pack_blob(SETS[name]())
    pcm = synth.synth_batch_parallel(B, T, first_stream=700)
    ro, rg, rf, rs = Oracle(blob).run_batch(pcm, group=8)
    dev = torch.device("cuda:0")
    model = api.Model(blob)
    ts = shared_stream(dev)
    with torch.cuda.stream(ts):
        d_pcm = torch.from_numpy(pcm).to(dev)
        ctx = api.Context(model, B, nn_mode=api.NN_MFMA, stream=ts.cuda_stream)
        out, gr, feat, sil = run_long(ctx, lambda t: d_pcm[:, t * 480:(t + 1) * 480].contiguous(), T, None, dev)
        ctx.close()
    # one step from identical state: MFMA vs STRICT kernels on the oracle's features
    TS = 200
    cm = api.Context(model, B, nn_mode=api.NN_MFMA)
    cs = api.Context(model, B, nn_mode=api.NN_STRICT)
    one = np.zeros(TS)
    for t in range(TS):
        cm.set_rnn_state(cs.get_rnn_state())
        gs_, gm_ = cs.compute_rnn(rf[:, t]), cm.compute_rnn(rf[:, t])
        assert np.isfinite(gm_).all(), t
        one[t] = np.abs(gs_ - gm_).max()
        if t < 50:
            assert np.array_equal(gs_, rg[:, t]), t          # the STRICT context IS the oracle, bit for bit (free-running from zero state)
    cm.close(); cs.close()
    model.close()
    d = np.abs(out.astype(np.int32) - ro.astype(np.int32))
    d = np.minimum(d, 65536 - d)                   # circular: the reference's float->int16 cast wraps (main.cpp:36)
    dg = np.abs(gr - rg)
    by_s = [float(dg[:, k:k + 100].max()) for k in range(0, T, 100)]
    stats = {
        "one_step_from_identical_state_max_abs_delta_gr": float(one.max()), "one_step_steps": TS,
        "one_step_max_by_50_steps": [float(one[k:k + 50].max()) for k in range(0, TS, 50)],
        "weights": name, "streams": B, "frames": T,
        "g_range": [float(rg[..., :34].min()), float(rg[..., :34].max())], "r_range": [float(rg[..., 34:].min()), float(rg[..., 34:].max())],
        "g_std_over_time": float(rg[..., :34].std()),
        "max_abs_delta_pcm_lsb_circular": int(d.max()), "pcm_samples": int(d.size),
        "pcm_delta_histogram_lsb": np.bincount(np.minimum(d, 16).ravel().astype(np.int64), minlength=17).tolist(),
        "max_abs_delta_gr": float(dg.max()), "mean_abs_delta_gr": float(dg.mean()),
        "p9999_abs_delta_gr": float(np.quantile(dg, 0.9999)),
        "max_abs_delta_gr_by_second": by_s, "non_silent_frames": int((rs == 0).sum()),
        "output_peak": int(np.abs(ro.astype(np.int32)).max()),
    }
    _record(f"stress_{name}", stats)
    assert np.isfinite(gr).all()
    assert np.array_equal(feat.view(np.uint32), rf.view(np.uint32)) and np.array_equal(sil, rs)
    assert one.max() <= ONE_STEP_TOL[name], float(one.max())
    assert max(by_s) <= 8 * max(by_s[0], 1e-6) + 1e-6, by_s          # no drift: the difference does not grow with time
    lsb, tol = BOUNDS[name]
    if lsb is not None:
        assert d.max() <= lsb, int(d.max())
    assert dg.max() <= tol, float(dg.max())


def test_fp16_bound_from_a_larger_sample(blob, oracle):
    """configs[4]: the re-stated tolerance from 1024 streams x 1000 frames (4x the round-2 sample), with max and
    99.99-percentile; the asserted bound keeps head-room over the measured maximum."""
    import torch
    B, T = 1024, 1000
    pcm = synth.synth_batch_parallel(B, T, first_stream=2000)
    ro, rg, rf, rs = oracle.run_batch(pcm, group=8)
    dev = torch.device("cuda:0")
    model = api.Model(blob)
    ts = shared_stream(dev)
    with torch.cuda.stream(ts):
        d_pcm = torch.from_numpy(pcm).to(dev)
        ctx = api.Context(model, B, nn_mode=api.NN_MFMA_F16, stream=ts.cuda_stream)
        out, gr, feat, sil = run_long(ctx, lambda t: d_pcm[:, t * 480:(t + 1) * 480].contiguous(), T, None, dev)
        ctx.close()
    model.close()
    d = np.abs(out.astype(np.int32) - ro.astype(np.int32))
    dg = np.abs(gr - rg)
    hist = np.bincount(np.minimum(d, 16).ravel().astype(np.int64), minlength=17)
    cum = np.cumsum(hist) / hist.sum()
    _record("configs4_fp16_1024x1000", {
        "config": "configs[4]: fp16 operands, 1024 streams x 1000 frames vs the CPU oracle",
        "max_abs_delta_pcm_lsb": int(d.max()), "p9999_abs_delta_pcm_lsb": int(np.searchsorted(cum, 0.9999)),
        "pcm_delta_histogram_lsb": hist.tolist(), "pcm_samples": int(d.size),
        "max_abs_delta_gr": float(dg.max()), "p9999_abs_delta_gr": float(np.quantile(dg, 0.9999)), "mean_abs_delta_gr": float(dg.mean()),
        "max_abs_delta_gr_by_second": [float(dg[:, k:k + 100].max()) for k in range(0, T, 100)]})
    assert np.array_equal(feat.view(np.uint32), rf.view(np.uint32)) and np.array_equal(sil, rs)
    assert d.max() <= 6, int(d.max())                 # measured 4 (256 streams, round 2); the stated bound is 6 LSB
    assert np.searchsorted(cum, 0.9999) <= 3
    assert dg.max() <= 1e-3, float(dg.max())
