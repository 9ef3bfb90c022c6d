"""A/B of two library builds in fp16 mode on bench-like data: saves PCM and g/r of each run under gpurun_out/ and compares."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from percepnet_amd import api, synth, weights
B, T = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (384, 40)
pcm = synth.synth_batch_fast(B, T)
model = api.Model(weights.default_blob(1234))
ctx = api.Context(model, B, nn_mode=api.NN_MFMA_F16)
out, gr = ctx.run_pcm(pcm)
tag = sys.argv[1]
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/f16ab_{tag}_pcm.npy", out); np.save(f"gpurun_out/f16ab_{tag}_gr.npy", gr)
other = "new" if tag == "old" else "old"
if os.path.exists(f"gpurun_out/f16ab_{other}_pcm.npy"):
    o2 = np.load(f"gpurun_out/f16ab_{other}_pcm.npy"); g2 = np.load(f"gpurun_out/f16ab_{other}_gr.npy")
    print("pcm identical:", np.array_equal(out, o2), " max|dPCM| =", int(np.abs(out.astype(np.int32) - o2.astype(np.int32)).max()),
          " gr identical:", np.array_equal(gr.view(np.uint32), g2.view(np.uint32)), " max|dgr| =", float(np.abs(gr - g2).max()))
    bad = np.argwhere(gr.view(np.uint32) != g2.view(np.uint32))
    print("first differing (stream, frame, col):", bad[:8].tolist())
