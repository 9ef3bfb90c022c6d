"""Experiment: do the VALU-bound front end and the MFMA-bound network overlap when two contexts run
on two HIP streams?  usage: two_ctx.py <streams_per_ctx> <n_ctx> <steps>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]); NC = int(sys.argv[2]); K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
streams = [torch.cuda.Stream() for _ in range(NC)]
MODE = {'f32': api.NN_MFMA, 'f16': api.NN_MFMA_F16, 'x3': api.NN_MFMA_X3}[os.environ.get('PN_MODE', 'f32')]
ctxs = [api.Context(model, B, nn_mode=MODE, stream=s.cuda_stream) for s in streams]
P = 64; T = K + 2
pool = torch.from_numpy(synth.synth_batch(P, T)).to(dev)
idx = torch.arange(B, device=dev) % P
frames = [pool[:, t*480:(t+1)*480][idx].contiguous() for t in range(T)]
outs = [torch.empty((B, 480), dtype=torch.int16, device=dev) for _ in range(NC)]
torch.cuda.synchronize()
def run(t0, t1):
    for t in range(t0, t1):
        for c, o in zip(ctxs, outs):
            c.process_i16_dev(frames[t].data_ptr(), o.data_ptr(), None)
run(0, 2); torch.cuda.synchronize()
SKEW = int(os.environ.get("PN_SKEW_CYCLES", "0"))          # delay the second context once: anti-phase start
if SKEW and NC > 1:
    with torch.cuda.stream(streams[1]):
        torch.cuda._sleep(SKEW)
t0 = time.perf_counter(); run(2, T); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"mode={os.environ.get('PN_MODE','f32')} lib={os.environ.get('PERCEPNET_LIB','default').split('/')[-2] if os.environ.get('PERCEPNET_LIB') else 'default'} skew={SKEW} ctx={NC} x {B}: {1e3*dt/K:.3f} ms/step-pair -> {NC*B*K/dt/100:.0f} streams")
