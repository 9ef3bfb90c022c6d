"""Where does the MFMA network path go non-finite on the x6 'saturating' random weight set (seed 14)?  Runs MFMA and
STRICT contexts side by side on the same PCM and reports the first frame / buffer / element that is not finite, with
the STRICT value at the same place."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from percepnet_amd import api, synth, weights
B, T = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 60
blob = weights.pack_blob(weights.random_layers(14, scale=6.0))
model = api.Model(blob)
pcm = synth.synth_batch(B, T, first_stream=700)
cm = api.Context(model, B, nn_mode=api.NN_MFMA)
cs = api.Context(model, B, nn_mode=api.NN_STRICT)
names = {0: ("feat", 128), 1: ("c1ring", 5 * 128), 2: ("c2ring", 3 * 512), 3: ("c2out", 512), 4: ("gru1", 2 * 512), 5: ("gru2", 2 * 512),
         6: ("gru3", 2 * 512), 7: ("gru_gb", 2 * 512), 8: ("rb", 2 * 128), 9: ("gr", 68)}
Bp = 256
for t in range(T):
    fr = pcm[:, t * 480:(t + 1) * 480]
    om, gm = cm.process_i16(fr); os_, gs = cs.process_i16(fr)
    bad = False
    for w, (nm, per) in names.items():
        n = (B * 68) if w == 9 else Bp * per
        vm = cm.debug_copy(w, n); vs = cs.debug_copy(w, n)
        nf = ~np.isfinite(vm)
        big = np.nanmax(np.abs(vs)) if vs.size else 0
        if nf.any():
            i = int(np.argwhere(nf)[0][0])
            print(f"frame {t} buffer {nm}: {int(nf.sum())} non-finite in MFMA (first flat index {i}: mfma {vm[i]} strict {vs[i]}); strict max |v| {big:.4g}, strict finite {np.isfinite(vs).all()}")
            bad = True
        elif t % 10 == 0:
            print(f"frame {t} buffer {nm}: finite; max|mfma| {np.abs(vm).max():.4g} max|strict| {big:.4g} max|diff| {np.abs(vm - vs).max():.4g}")
    if bad:
        break
print("done")
