import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from percepnet_amd import api, synth, weights
from oracle.oracle import Oracle
blob = weights.default_blob(1234); orc = Oracle(blob); model = api.Model(blob)
B, T = 40, 6
rng = np.random.default_rng(11)
feats = (rng.standard_normal((T, B, 70)) * 0.7).astype(np.float32)
ref = np.zeros((T, B, 68), np.float32)
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
for b in range(B):
    st = orc.lib.pno_create(orc.model)
    for t in range(T):
        g = np.zeros(34, np.float32); r = np.zeros(34, np.float32); f = np.ascontiguousarray(feats[t, b])
        orc.lib.pno_compute_rnn(st, fp(g), fp(r), fp(f)); ref[t, b, :34] = g; ref[t, b, 34:] = r
    orc.lib.pno_destroy(st)
for mode, name in ((api.NN_STRICT, "STRICT"), (api.NN_MFMA, "MFMA")):
    ctx = api.Context(model, B, nn_mode=mode)
    got = np.stack([ctx.compute_rnn(feats[t]) for t in range(T)])
    d = np.abs(got - ref)
    print(name, "max err g", d[..., :34].max(), "r", d[..., 34:].max())
    for t in range(T):
        rows = np.nonzero(d[t].max(1) > 2e-5)[0]
        print("  t", t, "bad rows", rows[:40], "max per-row r err", np.round(d[t, :, 34:].max(1)[:40], 6)[[0, 31, 32, 33, 39]])
    ctx.close()
