"""Run-to-run determinism: process the same T frames twice (fresh context each time) and compare every output sample and g/r."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]); T = int(sys.argv[2]); mode = {"f32": api.NN_MFMA, "f16": api.NN_MFMA_F16}[sys.argv[3]]
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
P = min(B, 64)
pool = torch.from_numpy(synth.synth_batch(P, T)).to(dev)
idx = torch.arange(B, device=dev) % P
rot = (torch.arange(B, device=dev) // P) * 37          # per-replica sample rotation, as bench.py does: all streams distinct
ar = (torch.arange(480, device=dev)[None, :] + rot[:, None]) % 480
frames = [torch.gather(pool[:, t*480:(t+1)*480][idx], 1, ar).contiguous() for t in range(T)]
torch.cuda.synchronize()      # the context has its own stream: inputs must be complete before it reads them
res = []
for rep in range(2):
    ctx = api.Context(model, B, nn_mode=mode, stream=torch.cuda.current_stream().cuda_stream)
    out = torch.empty((T, B, 480), dtype=torch.int16, device=dev); gr = torch.empty((T, B, 68), dtype=torch.float32, device=dev)
    for t in range(T):
        ctx.process_i16_dev(frames[t].data_ptr(), out[t].data_ptr(), gr[t].data_ptr())
    torch.cuda.synchronize()
    res.append((out.clone(), gr.clone())); ctx.close()
same_pcm = torch.equal(res[0][0], res[1][0]); same_gr = torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32))
print(f"B={B} T={T} {sys.argv[3]}: pcm identical={same_pcm} g/r identical={same_gr}")
if not same_gr:
    d = (res[0][1].view(torch.int32) != res[1][1].view(torch.int32)).nonzero()
    print("  first differing (frame, stream, col):", d[:6].tolist(), " count", len(d), " max|d|", float((res[0][1]-res[1][1]).abs().max()))
    fr = d[:, 0].min().item(); st = d[d[:, 0] == fr][:, 1]
    print(f"  first frame {fr}: streams min {st.min().item()} max {st.max().item()} n {len(st.unique())}; stream%128 hist:", torch.bincount(st.unique() % 128, minlength=128).nonzero().flatten()[:20].tolist())
