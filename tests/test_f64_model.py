"""tools/nn_f64_model.py (the float64 yardstick of tests/test_gpu_x3.py) against the CPU oracle, no GPU needed: free-running
from the zero state on the same features, the float64 evaluation of the reference's formulas and the oracle's fp32 one must
stay within fp32 rounding of each other (a wrong weight layout, gate order or FIFO order would be off by O(0.1))."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

from percepnet_amd import synth, weights


def test_float64_model_tracks_the_oracle(blob, oracle):
    import nn_f64_model as M
    B, T = 6, 12
    lay = weights.unpack_blob(blob)
    pcm = synth.synth_batch(B, T)
    ro, rg, rf, rs = oracle.run_batch(pcm)
    tab = oracle.tansig_table()
    st = {"conv1": np.zeros((B, 4 * 128)), "conv2": np.zeros((B, 2 * 512)), "gru1": np.zeros((B, 512)), "gru2": np.zeros((B, 512)),
          "gru3": np.zeros((B, 512)), "gru_gb": np.zeros((B, 512)), "gru_rb": np.zeros((B, 128))}
    worst = 0.0
    for t in range(T):
        # one step of the model, then advance ITS state the way compute_rnn does (FIFOs shift, GRUs take the new values)
        f = rf[:, t].astype(np.float64)
        acts = {name: act for name, kind, nin, nn_, ks, act in weights.LAYERS}
        tb = np.asarray(tab, np.float64)[:201]
        fc = M._dense(lay["fc"], f, 128, acts["fc"], tb)
        c1 = M._dense(lay["conv1"], np.concatenate([st["conv1"], fc], axis=1), 512, acts["conv1"], tb)
        c2 = M._dense(lay["conv2"], np.concatenate([st["conv2"], c1], axis=1), 512, acts["conv2"], tb)
        g = M.step(lay, tab, st, rf[:, t])
        worst = max(worst, float(np.abs(g - rg[:, t]).max()))
        g1 = M._gru(lay["gru1"], c2, st["gru1"], 512, acts["gru1"], tb)
        g2 = M._gru(lay["gru2"], g1, st["gru2"], 512, acts["gru2"], tb)
        g3 = M._gru(lay["gru3"], g2, st["gru3"], 512, acts["gru3"], tb)
        gb = M._gru(lay["gru_gb"], g3, st["gru_gb"], 512, acts["gru_gb"], tb)
        rb = M._gru(lay["gru_rb"], np.concatenate([g3, c2], axis=1), st["gru_rb"], 128, acts["gru_rb"], tb)
        st = {"conv1": np.concatenate([st["conv1"][:, 128:], fc], axis=1), "conv2": np.concatenate([st["conv2"][:, 512:], c1], axis=1),
              "gru1": g1, "gru2": g2, "gru3": g3, "gru_gb": gb, "gru_rb": rb}
    assert worst < 2e-5, worst          # fp32 rounding accumulated over 12 free-running steps; a layout error is O(0.1)
