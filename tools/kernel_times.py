"""Per-kernel HIP-event times at a given batch size (tuning aid). PERCEPNET_LIB selects a variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
MODE = {'f32': api.NN_MFMA, 'f16': api.NN_MFMA_F16, 'strict': api.NN_STRICT, 'x3': api.NN_MFMA_X3}[os.environ.get('PN_MODE', 'f32')]
ctx = api.Context(model, B, nn_mode=MODE, stream=torch.cuda.current_stream().cuda_stream)
P = min(B, 64); T = K + 2
pool = torch.from_numpy(synth.synth_batch(P, T)).to(dev)
idx = torch.arange(B, device=dev) % P
frames = [pool[:, t*480:(t+1)*480][idx].contiguous() for t in range(T)]
torch.cuda.synchronize()      # the context has its own stream: inputs must be complete before it reads them
out = torch.empty((B, 480), dtype=torch.int16, device=dev)
for t in range(2): ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize(); ctx.reset_profile(); ctx.set_profiling(True)
t0 = time.perf_counter()
for t in range(2, T): ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
kt = ctx.kernel_times()
print(os.environ.get("PERCEPNET_LIB", "default").split("/")[-2:][0], os.environ.get("PN_MODE", "f32"), f"B={B} ms/step={1e3*dt/K:.3f} streams={B*K/dt/100:.0f} |",
      " ".join(f"{k}={v[0]/max(v[1],1):.3f}" for k, v in kt.items()))
