"""fp16-variant tolerance + timing."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from percepnet_amd import api, synth, weights
from oracle.oracle import Oracle
blob = weights.default_blob(1234); orc = Oracle(blob); model = api.Model(blob)
B, T = 24, 100
pcm = synth.synth_batch(B, T)
ctx = api.Context(model, B, nn_mode=api.NN_MFMA_F16)
out, gr = ctx.run_pcm(pcm)
d_pcm = []; d_gr = []
for s in range(B):
    ro, rg = orc.run_pcm(pcm[s])
    d_pcm.append(np.abs(out[s].astype(np.int32) - ro.astype(np.int32)))
    d_gr.append(np.abs(gr[s] - rg))
d_pcm = np.stack(d_pcm); d_gr = np.stack(d_gr)
print("fp16 variant: max|dPCM| =", d_pcm.max(), "LSB; 99.9pct =", np.percentile(d_pcm, 99.9), "; frac nonzero =", (d_pcm > 0).mean())
print("   per-kind max:", {k: int(d_pcm[[i for i in range(B) if synth.stream_kind(i) == k]].max()) for k in ("voiced", "loud", "bursts", "twotone")})
print("   max|d g,r| =", d_gr.max(), " mean =", d_gr.mean())
ctx.close()
