"""How far is each network mode from EXACT arithmetic?  One compute_rnn step from identical state (the STRICT context's,
= the CPU reference's bit for bit) evaluated in float64 (tools/nn_f64_model.py) and by the STRICT (= CPU reference), fp32
MFMA, split-precision and fp16-operand kernels; max / rms of |g,r - float64| per mode over B streams x T steps."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from percepnet_amd import api, synth, weights
from oracle.oracle import Oracle
import nn_f64_model as M


def measure(layers, B=128, T=60, first_stream=700):
    blob = weights.pack_blob(layers)
    orc = Oracle(blob)
    pcm = synth.synth_batch(B, T, first_stream=first_stream)
    rf = orc.run_batch(pcm)[2]
    tab = orc.tansig_table()
    model = api.Model(blob)
    modes = {"strict (= CPU reference)": api.NN_STRICT, "fp32 MFMA": api.NN_MFMA, "split precision": api.NN_MFMA_X3, "fp16 operands": api.NN_MFMA_F16}
    ctx = {k: api.Context(model, B, nn_mode=m) for k, m in modes.items()}
    ref = ctx["strict (= CPU reference)"]
    err = {k: [] for k in modes}
    for t in range(T):
        st = ref.get_rnn_state()
        exact = M.step(layers, tab, st, rf[:, t])
        for k, c in ctx.items():
            if c is not ref:
                c.set_rnn_state(st)
        for k, c in ctx.items():
            err[k].append(np.abs(c.compute_rnn(rf[:, t]).astype(np.float64) - exact))
    for c in ctx.values():
        c.close()
    model.close()
    return {k: {"max": float(np.max(v)), "rms": float(np.sqrt(np.mean(np.square(v))))} for k, v in err.items()}


if __name__ == "__main__":
    out = {}
    for name, lay in (("default (seed 1234)", weights.unpack_blob(weights.default_blob(1234))), ("scale2", weights.random_layers(11, scale=2.0)),
                      ("scale3", weights.random_layers(12, scale=3.0))):
        out[name] = measure(lay)
        print(name)
        for k, v in out[name].items():
            print(f"   {k:28s} max |g,r - float64| {v['max']:.3e}   rms {v['rms']:.3e}")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/x3_vs_f64.json", "w"), indent=1)
