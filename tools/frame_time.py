"""ms per frame WITHOUT per-kernel events (round 5): B streams (64 pool streams tiled), 12 priming + 20 warm frames, then `reps`
repeats of K steps between synchronisations.  PERCEPNET_LIB selects a library (e.g. lib/variants/r4 = the round-4 kernels),
PN_MODE the network mode.  Prints min / median of the repeats and the event-timed kernel sum of one more pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
MODE = {'f32': api.NN_MFMA, 'f16': api.NN_MFMA_F16, 'strict': api.NN_STRICT, 'x3': api.NN_MFMA_X3}[os.environ.get('PN_MODE', 'f32')]
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = api.Context(model, B, nn_mode=MODE, stream=stream.cuda_stream)
P = min(B, 64); T = 32 + min(K, 40)
pool = torch.from_numpy(synth.synth_batch(P, T)).to(dev)
idx = torch.arange(B, device=dev) % P
frames = [pool[:, t*480:(t+1)*480][idx].contiguous() for t in range(T)]
out = torch.empty((B, 480), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
for t in range(32): ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize()
ms = []
for r in range(reps):
    t0 = time.perf_counter()
    for i in range(K): ctx.process_i16_dev(frames[32 + i % (T - 32)].data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize(); ms.append(1e3 * (time.perf_counter() - t0) / K)
ctx.reset_profile(); ctx.set_profiling(True)
for i in range(min(K, 40)): ctx.process_i16_dev(frames[32 + i % (T - 32)].data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize(); kt = ctx.kernel_times()
name = os.environ.get("PERCEPNET_LIB", "default/x").split("/")[-2]
print(f"{name:10s} {os.environ.get('PN_MODE', 'f32')} B={B}: ms/frame min {min(ms):.4f} median {sorted(ms)[len(ms)//2]:.4f} | kernel sum with events {sum(v[0]/max(v[1],1)*(4 if k=='gru512' else 1) for k, v in kt.items() if v[1]):.4f} | "
      + " ".join(f"{k}={v[0]/max(v[1],1):.4f}" for k, v in kt.items() if v[1]), flush=True)
