import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from percepnet_amd import api, weights
blob = weights.default_blob(1234); model = api.Model(blob)
L = api.load_library()
L.pn_ctx_debug_copy.restype = ctypes.c_longlong
L.pn_ctx_debug_copy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
B = 64
rng = np.random.default_rng(11)
f1 = (rng.standard_normal((1, 70)) * 0.7).astype(np.float32)
feats = np.repeat(f1, B, axis=0)
ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
ctx.compute_rnn(feats)   # t=0 -> written slots: c1ring[0], c2ring[0], states in [1]
def grab(which, shape):
    a = np.zeros(shape, np.float32)
    n = L.pn_ctx_debug_copy(ctx.h, which, a.ctypes.data, a.nbytes); assert n == a.nbytes, n
    return a
bufs = {"feat": grab(0, (B, 96)), "fc(c1ring0)": grab(1, (5, B, 128))[0], "conv1(c2ring0)": grab(2, (3, B, 512))[0],
        "conv2": grab(3, (B, 512)), "gru1": grab(4, (2, B, 512))[1], "gru2": grab(5, (2, B, 512))[1],
        "gru3": grab(6, (2, B, 512))[1], "gru_gb": grab(7, (2, B, 512))[1], "gru_rb": grab(8, (2, B, 128))[1], "gr": grab(9, (B, 68))}
for k, v in bufs.items():
    d = np.abs(v - v[:1]).max(1)
    bad = np.nonzero(d > 0)[0]
    info = ""
    if len(bad):
        cols = np.nonzero(v[bad[0]] != v[0])[0]
        info = f" cols(n={len(cols)}) {cols[:16]} e.g. {v[bad[0], cols[0]]!r} vs {v[0, cols[0]]!r}"
    print(f"{k:16s} rows differing from row 0: {bad[:12]} max {d.max():.3e}{info}")
