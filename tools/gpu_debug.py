"""Stage-by-stage GPU-vs-oracle diagnostics (prints, never asserts). Run on the GPU box."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from percepnet_amd import api, synth, weights
from oracle.oracle import Oracle

blob = weights.default_blob(1234)
orc = Oracle(blob)
model = api.Model(blob)
streams = [0, 3, 7, 13, 23]
T = 30
pcm = np.stack([synth.synth_stream(s, T) for s in streams])
B = len(streams)
for mode, name in ((api.NN_STRICT, "STRICT"), (api.NN_MFMA, "MFMA")):
    ctx = api.Context(model, B, nn_mode=mode)
    feats = np.zeros((B, T, 70), np.float32); sils = np.zeros((B, T), np.int32)
    outs = np.zeros((B, T, 480), np.int16); grs = np.zeros((B, T, 68), np.float32)
    for t in range(T):
        o, g = ctx.process_i16(pcm[:, t*480:(t+1)*480])
        outs[:, t] = o; grs[:, t] = g
        feats[:, t], sils[:, t] = ctx.read_features()
    for i, s in enumerate(streams):
        rf, rs = orc.features(pcm[i].astype(np.float32) / np.float32(32768))
        ro, rg = orc.run_pcm(pcm[i])
        fd = feats[i] != rf
        print(f"[{name}] stream {s}: feat mismatches {fd.sum()} (first frame {np.argmax(fd.any(1)) if fd.any() else -1}, cols {np.unique(np.nonzero(fd)[1])[:12]}), max|dfeat| {np.abs(feats[i]-rf).max():.3e}, sil mism {(sils[i]!=rs).sum()}")
        print(f"      gr max|d| {np.abs(grs[i]-rg).max():.3e}  pcm max|d| {np.abs(outs[i,1:].reshape(-1).astype(int)-ro.astype(int)).max()}  pcm neq {np.count_nonzero(outs[i,1:].reshape(-1)!=ro)}")
        if fd.any():
            t0 = np.argmax(fd.any(1))
            print("      first bad frame feats gpu", feats[i, t0, [0, 1, 33, 34, 35, 67, 68, 69]], "ref", rf[t0, [0, 1, 33, 34, 35, 67, 68, 69]])
    ctx.close()
print("done")
