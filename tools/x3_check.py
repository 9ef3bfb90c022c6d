"""Split-precision mode (NN_MFMA_X3) against STRICT (bit-exact to the CPU reference) and the fp32 MFMA mode: max |dg,r| and
max |dPCM| over B streams x T frames of synthetic speech+noise, plus per-kernel times at a large batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0
model = api.Model(weights.default_blob(1234) if scale == 0 else weights.pack_blob(weights.random_layers(11, scale=scale)))
pcm = synth.synth_batch(min(B, 64), T)
pcm = pcm[np.arange(B) % pcm.shape[0]]
out = {}
for name, mode in (("strict", api.NN_STRICT), ("mfma", api.NN_MFMA), ("x3", api.NN_MFMA_X3)):
    ctx = api.Context(model, B, nn_mode=mode)
    print(name, ctx.describe(), flush=True)
    o, g = [], []
    for t in range(T):
        y, gr = ctx.process_i16(pcm[:, t * 480:(t + 1) * 480], want_gr=True)
        o.append(y.copy()); g.append(gr.copy())
    out[name] = (np.stack(o).astype(np.int64), np.stack(g))
    ctx.close()
for a in ("mfma", "x3"):
    dp = np.abs(out[a][0] - out["strict"][0]); dg = np.abs(out[a][1] - out["strict"][1])
    print(f"{a} vs strict: max|dPCM| {dp.max()} LSB ({(dp > 0).mean() * 100:.3f} % differ), max|dg,r| {dg.max():.3e}, mean {dg.mean():.3e}, finite {np.isfinite(out[a][1]).all()}")
