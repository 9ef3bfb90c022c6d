"""Placement invariance of the front end at a batch large enough for several grid-stride rounds per block:
identical streams must give identical features wherever they sit.  PERCEPNET_LIB selects the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from percepnet_amd import api, synth, weights
B, K, T = 4099 + 4096, 7, 14
model = api.Model(weights.default_blob(1234))
base = synth.synth_batch(K, T)
idx = np.arange(B) % K
pcm = base[idx]
ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
bad = {}
for t in range(T):
    frame = np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480])
    out = np.empty_like(frame)
    ctx.L.pn_process_host_i16(ctx.h, frame.ctypes.data, out.ctypes.data, None)
    feat, sil = ctx.read_features()
    for k in range(K):
        f = feat[idx == k]
        d = np.nonzero((f.view(np.uint32) != f[0].view(np.uint32)).any(axis=1))[0]
        if len(d) and k not in bad:
            cols = np.nonzero((f[d[0]].view(np.uint32) != f[0].view(np.uint32)))[0]
            bad[k] = (t, d[:6].tolist(), cols[:12].tolist())
print(os.environ.get("PERCEPNET_LIB", "default").split("/")[-2] if os.environ.get("PERCEPNET_LIB") else "default",
      "OK" if not bad else f"MISMATCH (kind: first frame, replica indices, feature columns): {bad}")
