import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from percepnet_amd import api, synth, weights
blob = weights.default_blob(1234); model = api.Model(blob)
np.set_printoptions(linewidth=200, precision=3)
for B in (40, 256):
    T = 6
    rng = np.random.default_rng(11)
    f1 = (rng.standard_normal((T, 1, 70)) * 0.7).astype(np.float32)
    feats = np.repeat(f1, B, axis=1)
    res = []
    for rep in range(2):
        ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
        got = np.stack([ctx.compute_rnn(feats[t]) for t in range(T)])
        res.append(got); ctx.close()
    print("B", B, "run-to-run identical:", np.array_equal(res[0], res[1]))
    got = res[0]
    for t in range(T):
        d = np.abs(got[t] - got[t, :1]).max(1)
        bad = np.nonzero(d > 0)[0]
        print("  t", t, "rows differing from row 0:", len(bad), bad[:48], "max", d.max())
        if len(bad):
            cols = np.nonzero(np.abs(got[t, bad[0]] - got[t, 0]) > 0)[0]
            print("     first bad row cols", cols[:40])
