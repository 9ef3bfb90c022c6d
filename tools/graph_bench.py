"""Measured, not argued: one-submit multi-frame processing.  60 consecutive frames (one full period of
the ring indices: lcm(12, 6, 5, 3, 2)) of the 15-launch frame sequence are captured into ONE hipGraph on the context's
stream and replayed; compared with the eager loop over the same frames.   python tools/graph_bench.py [streams] [replays]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
F = 60
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(dev)
torch.cuda.set_stream(ts)
model = api.Model(weights.default_blob(1234))
ctx = api.Context(model, B, nn_mode=api.NN_MFMA, stream=ts.cuda_stream)
P = min(B, 64)
pool = torch.from_numpy(synth.synth_batch(P, F)).to(dev)
idx = torch.arange(B, device=dev) % P
frames = torch.stack([pool[:, t * 480:(t + 1) * 480][idx] for t in range(F)]).contiguous()      # [F][B][480]
outs = torch.empty((F, B, 480), dtype=torch.int16, device=dev)
def run_eager():
    for t in range(F):
        ctx.process_i16_dev(frames[t].data_ptr(), outs[t].data_ptr(), None)
run_eager(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R): run_eager()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / (R * F)
ref = outs.clone()
ctx.reset(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
run_eager(); torch.cuda.synchronize()           # frames 0..59 of the fresh state (also warms the capture path)
first = outs.clone()
with torch.cuda.graph(g, stream=ts):
    run_eager()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R): g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / (R * F)
print(json.dumps({"streams": B, "frames_per_submit": F, "eager_ms_per_frame": round(1e3 * eager, 4), "graph_ms_per_frame": round(1e3 * graph, 4),
                  "eager_streams": round(B / eager / 100), "graph_streams": round(B / graph / 100),
                  "note": "same 60 frames replayed; graph = one hipGraphLaunch per 60 frames (900 kernel nodes)"}))
