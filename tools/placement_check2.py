"""Which spectrum bins differ between a first-round and a later-round replica of the same stream?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from percepnet_amd import api, synth, weights
B, K, T = 8195, 7, 2
model = api.Model(weights.default_blob(1234))
base = synth.synth_batch(K, T)
idx = np.arange(B) % K
pcm = base[idx]
ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
L = ctx.L
L.pn_ctx_debug_copy.restype = ctypes.c_longlong
L.pn_ctx_debug_copy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
for t in range(T):
    frame = np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480]); out = np.empty_like(frame)
    L.pn_process_host_i16(ctx.h, frame.ctypes.data, out.ctypes.data, None)
    yr = np.empty((6, B, 400, 2), np.float32)
    assert L.pn_ctx_debug_copy(ctx.h, 10, yr.ctypes.data, yr.nbytes) > 0
    ps = np.empty((B, 400, 2), np.float32)
    assert L.pn_ctx_debug_copy(ctx.h, 11, ps.ctypes.data, ps.nbytes) > 0
    hs = np.empty((B, 5768), np.float32)
    assert L.pn_ctx_debug_copy(ctx.h, 12, hs.ctypes.data, hs.nbytes) > 0
    slot = t % 6
    for name, arr in (("Y", yr[slot].reshape(B, 800)), ("P", ps.reshape(B, 800)), ("hist", hs)):
        a = arr.view(np.uint32)
        ref = a[0]                      # stream 0 (kind 0, round 1)
        rep = np.nonzero(idx == 0)[0]
        bad = [(int(r), np.nonzero(a[r] != ref)[0]) for r in rep if (a[r] != ref).any()]
        if bad:
            r, cols = bad[0]
            print(f"frame {t} {name}: {len(bad)} of {len(rep)} replicas differ; first stream {r}: {len(cols)} floats, bins {sorted(set((cols // 2).tolist()))[:40]}")
        else:
            print(f"frame {t} {name}: all replicas identical")
